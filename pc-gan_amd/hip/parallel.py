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

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), \
        int(os.environ.get('LOCAL_RANK', '0'))


def init_process_group(backend=None):
    """Initialise torch.distributed from the torchrun environment; no-op for a single process."""
    world, rank, local = env_world()
    if world <= 1:
        return world, rank, local
    if not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('PCGAN_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


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
