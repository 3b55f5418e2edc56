"""Data parallelism: one process per GPU, full replicas, ONE gradient all-reduce (average)
per optimizer per step over RCCL/xGMI (torch.distributed backend "nccl" on ROCm).

Mathematically this is the reference's nn.DataParallel (models/networks.py:96-102) when rank r
takes the contiguous slice [r*B/n, (r+1)*B/n) of the global batch: losses are means over the
batch, so the global gradient is the average of the per-rank gradients; BatchNorm/InstanceNorm
statistics stay per replica exactly as under DataParallel.

Payload: G 11,381,315 x 4 B = 45.5 MB, D 11.1 MB per step.  xGMI is point-to-point, so one
large flat all-reduce per optimizer (not per-tensor buckets) keeps the per-link ring cost at
~0.5 ms against a ~70 ms compute step.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist


# PCGAN_FORCE_COLLECTIVES=1: a world of ONE rank still builds the process group and runs every collective of the step (RCCL on a
# GPU box: communicator set-up, ReduceOp.AVG, the launched-now / waited-later all-reduce next to the side streams, broadcast) -- the
# rehearsal a one-GPU box allows (RCCL refuses two ranks on one device); tests/test_gpu_ddp.py::test_one_rank_over_rccl
FORCE_COLLECTIVES = os.environ.get('PCGAN_FORCE_COLLECTIVES', '0') == '1'


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment; no-op for a single process."""
    world, rank, local = env_world()
    if world <= 1 and not FORCE_COLLECTIVES:
        return world, rank, local
    if not dist.is_initialized():
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        os.environ.setdefault('MASTER_PORT', '29533')
        if backend is None:
            backend = os.environ.get('PCGAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        selfcheck_collectives()
    return world, rank, local


class CollectiveSelfCheckError(RuntimeError):
    pass


def selfcheck_collectives(n=256):
    """First thing after the process group exists (round 4: the RCCL branch had never run on more than one rank, so its first run
    must not be able to fail silently): a 1 KiB gradient-shaped all-reduce through the SAME code path the step uses
    (allreduce_mean_: ReduceOp.AVG on RCCL, SUM + scale on gloo) against (a) a plain SUM all-reduce divided by the world size and
    (b) the closed form -- rank r contributes (r + 1) * [1 .. n], so the mean is (world + 1) / 2 * [1 .. n], exact in fp32 --
    and a broadcast from rank 0.  Raises CollectiveSelfCheckError on every rank (they all see the same wrong numbers) naming what
    differed; returns the largest deviation (0.0 expected)."""
    if not is_distributed():
        return 0.0
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device('cuda', torch.cuda.current_device()) if (dist.get_backend() == 'nccl' or torch.cuda.is_available()) else torch.device('cpu')
    if dist.get_backend() == 'gloo' and os.environ.get('PCGAN_SELFCHECK_CPU'):
        dev = torch.device('cpu')
    base = torch.arange(1, n + 1, dtype=torch.float32, device=dev)
    mine = base * float(rank + 1)
    avg = allreduce_mean_(mine.clone())
    tot = mine.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    want = base * ((world + 1) / 2.0)
    e_avg = float((avg - want).abs().max())
    e_sum = float((tot / world - want).abs().max())
    b = mine.clone()
    dist.broadcast(b, 0)
    e_bc = float((b - base).abs().max())
    worst = max(e_avg, e_sum, e_bc)
    if worst != 0.0:
        raise CollectiveSelfCheckError(
            'pcgan_amd: collective self-check failed on rank %d / %d (backend %s): |AVG - closed form| = %g, |SUM / world - closed form| '
            '= %g, |broadcast - rank 0| = %g -- the gradient all-reduce cannot be trusted on this fabric / build'
            % (rank, world, dist.get_backend(), e_avg, e_sum, e_bc))
    return worst


def rank_identity(device=None):
    """what bench.py --gpus N records per rank so that the record PROVES N distinct GPUs took part: device uuid / name / PCI bus,
    host, pid, RCCL version, backend"""
    info = {'rank': env_world()[1], 'host': socket.gethostname(), 'pid': os.getpid(),
            'backend': dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None}
    if torch.cuda.is_available():
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        pr = torch.cuda.get_device_properties(idx)
        info.update({'device_index': idx, 'device_name': pr.name, 'device_uuid': str(getattr(pr, 'uuid', '')),
                     'pci_bus_id': getattr(pr, 'pci_bus_id', None), 'cus': pr.multi_processor_count})
        try:
            info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001  (diagnostics only)
            info['rccl_version'] = 'unavailable: %s' % e
    return info


# ---- PCGAN_DDP_CHECK=k: every k-th optimizer step all ranks compare a hash of their flat parameter buffers and the job ends non-zero
# on the first divergence (replicas must stay bit-identical: same start, same averaged gradients, same Adam arithmetic)
DDP_CHECK_EVERY = int(os.environ.get('PCGAN_DDP_CHECK', '0') or 0)
_ddp_check_calls = {}


def flat_hash(flat):
    """two int64 checksums of a flat fp32 buffer's BITS (plain sum, position-weighted sum): equal buffers <=> equal hashes for all
    practical purposes; torch integer ops -- a debugging aid off the hot path, not arithmetic of the step"""
    bits = flat.detach().view(torch.int32).to(torch.int64)
    w = (torch.arange(bits.numel(), device=bits.device, dtype=torch.int64) % 65521) + 1
    return torch.stack([bits.sum(), (bits * w).sum()])


class ReplicaDivergenceError(RuntimeError):
    pass


def ddp_check(optimizer, name='', every=None, exit_on_divergence=True):
    """call after optimizer.step(); every `every`-th call (default PCGAN_DDP_CHECK) compares flat_hash(optimizer.flat) across ranks
    (MIN and MAX all-reduce of the hash: identical on all ranks iff they agree).  On divergence every rank prints its own hash and
    the job exits with code 3 -- a fresh exit from the running process, never a re-exec (gpurun forbids exec after HIP init).
    Returns True when a comparison ran and passed."""
    every = DDP_CHECK_EVERY if every is None else every
    if not every or not is_distributed():
        return False
    n = _ddp_check_calls[name] = _ddp_check_calls.get(name, 0) + 1
    if n % every:
        return False
    flat = getattr(optimizer, 'flat', None)
    if flat is None:
        flat = torch.cat([p.detach().reshape(-1) for g in optimizer.param_groups for p in g['params']])
    h = flat_hash(flat)
    lo, hi = h.clone(), h.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if bool((lo != hi).any()):
        msg = ('pcgan_amd: replicas DIVERGED at optimizer %r step %d: rank %d hash %s (min %s, max %s over ranks)'
               % (name, n, dist.get_rank(), h.tolist(), lo.tolist(), hi.tolist()))
        print(msg, file=sys.stderr, flush=True)
        if exit_on_divergence:
            sys.exit(3)
        raise ReplicaDivergenceError(msg)
    return True


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def shard_batch(t, rank=None, world=None):
    """Contiguous slice of the global batch for this rank (DataParallel's scatter along dim 0)."""
    if world is None:
        world, rank = (dist.get_world_size(), dist.get_rank()) if is_distributed() else (1, 0)
    n = t.shape[0] if hasattr(t, 'shape') else len(t)
    assert n % world == 0, 'global batch %d not divisible by world size %d' % (n, world)
    per = n // world
    return t[rank * per:(rank + 1) * per]


def shard_dict(batch, world=None, rank=None):
    """shard_batch over every entry of a loader batch (tensors and per-sample lists alike)."""
    if world is None:
        world, rank = (dist.get_world_size(), dist.get_rank()) if is_distributed() else (1, 0)
    if world == 1:
        return batch
    return {k: shard_batch(v, rank, world) for k, v in batch.items()}


# bench.py --gpus N: [(start, stop)] HIP-event pairs on the launch stream around every gradient all-reduce (the collective incl. the
# wait for the slowest rank), so that the first multi-GPU run is diagnosable; None = off
COMM_TIMER = None


def allreduce_mean_(flat, async_op=False):
    """In-place average of a flat gradient buffer over all ranks.  async_op: the collective is only LAUNCHED (RCCL runs it on its own
    stream, after everything queued so far on the current stream); the returned callable makes the current stream wait for it."""
    if not is_distributed():
        return (lambda: flat) if async_op else flat
    world = dist.get_world_size()
    ev = None
    if COMM_TIMER is not None and flat.is_cuda and not async_op:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend() == 'nccl':
        work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)      # RCCL computes the average in the ring
        if async_op:
            return lambda: (work.wait(), flat)[1]
        if ev is not None:
            ev[1].record()
            COMM_TIMER.append(ev)
        return flat
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=async_op)      # gloo (CPU tests / single-GPU rehearsal): no AVG
    if async_op:
        def finish():
            work.wait()
            flat.mul_(1.0 / world)
            return flat
        return finish
    flat.mul_(1.0 / world)
    if ev is not None:
        ev[1].record()
        COMM_TIMER.append(ev)
    return flat


def sync_gradients(optimizer, async_op=False):
    """Average the optimizer's gradients across ranks: one collective when it is a FusedAdam
    (flat buffer), otherwise one flattened collective over its parameter grads.  async_op (flat buffers only): launch the
    collective and return a callable to wait for it -- the caller overlaps it with work that neither reads nor writes these
    gradients (wsgan_emb: the generator's 45.5 MB all-reduce under the discriminator's backward pass)."""
    if not is_distributed():
        return (lambda: None) if async_op else None
    from . import ops
    ops.join_side_stream()
    gflat = getattr(optimizer, 'gflat', None)
    if gflat is not None:
        if async_op:
            return allreduce_mean_(gflat, async_op=True)
        allreduce_mean_(gflat)
        return
    if async_op:
        sync_gradients(optimizer)
        return lambda: None
    grads = [p.grad for g in optimizer.param_groups for p in g['params'] if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    allreduce_mean_(flat)
    o = 0
    for g in grads:
        g.copy_(flat[o:o + g.numel()].view_as(g))
        o += g.numel()


def broadcast_parameters(net, src=0):
    """Make every replica start from rank `src`'s weights and buffers."""
    if not is_distributed():
        return
    for t in list(net.parameters()) + list(net.buffers()):
        dist.broadcast(t.data, src)
    from . import ops
    ops.invalidate_packed_weights()
