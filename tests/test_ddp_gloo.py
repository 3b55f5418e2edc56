"""CPU, world_size 2 over gloo: the data-parallel plumbing of pcgan_amd.hip.parallel.

Checks (a) batch sharding = DataParallel's contiguous scatter, (b) one flat all-reduce averages
the fused optimizer's gradient buffer, (c) an N-rank step with per-rank BatchNorm statistics and
averaged gradients equals the single-process computation of the same thing (the reference's
DataParallel semantics) -- the step itself run by the oracle on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import networks_ref as N
from oracle import step_ref as S
from oracle import weights as W


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build_step():
    G = N.ResnetGeneratorRef(3, 3, 1, 8, 'instance', 2)
    D = N.NLayerDiscriminatorRef(3, 1, 8, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    IP = N.AlexNetFeatureRef(3, 'None')
    for i, net in enumerate((G, D, E, IP)):
        net.load_state_dict(W.fill_state_dict(net.state_dict(), 80 + i))
    return S.WSGANEmbStepRef(G, D, E, IP, fineSize_E=64, fineSize_IP=64)


def _half_step(m, A, B, label, sync):
    """forward + G backward (+sync) + G step + D backward (+sync) + D step, capturing the synced grads"""
    m.set_input(A, B, label)
    m.forward()
    for p in m.netD.parameters():
        p.requires_grad = False
    m.optimizer_G.zero_grad()
    m.backward_G()
    sync(m.optimizer_G)
    gG = {k: p.grad.clone() for k, p in m.netG.named_parameters()}
    m.optimizer_G.step()
    for p in m.netD.parameters():
        p.requires_grad = True
    m.optimizer_D.zero_grad()
    m.backward_D()
    sync(m.optimizer_D)
    gD = {k: p.grad.clone() for k, p in m.netD.named_parameters()}
    m.optimizer_D.step()
    return gG, gD


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2 if world <= 2 else 1)
    from pcgan_amd.hip import parallel
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    w, r, _ = parallel.init_process_group('gloo')
    assert (w, r) == (world, rank) and parallel.is_distributed()
    # (a) sharding
    full = torch.arange(8).view(8, 1)
    per8 = 8 // world
    assert parallel.shard_batch(full).flatten().tolist() == list(range(rank * per8, rank * per8 + per8))
    # (b) fused optimizer's flat gradient buffer: one all-reduce, average
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    parallel.broadcast_parameters(D)
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt.gflat.fill_(float(rank + 1))
    parallel.sync_gradients(opt)
    assert torch.allclose(opt.gflat, torch.full_like(opt.gflat, (world + 1) / 2.0))
    # ... and the launched-now / waited-later form the overlapped step uses (PCGAN_DDP_OVERLAP=1)
    opt.gflat.fill_(float(2 * rank))
    finish = parallel.sync_gradients(opt, async_op=True)
    finish()
    assert torch.allclose(opt.gflat, torch.full_like(opt.gflat, float(world - 1)))
    chk = torch.stack([p.detach().sum() for p in D.parameters()]).sum().reshape(1)
    both = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(both, chk)
    assert all(torch.equal(both[0], b) for b in both), 'replicas must start from rank 0 weights'
    # (c) the step on this rank's contiguous slice with averaged gradients
    m = _build_step()
    A = W.seeded_tensor((4, 3, 32, 32), 900)
    B = W.seeded_tensor((4, 3, 32, 32), 901)
    label = [0, 2, 2, 0]
    per = 4 // world
    sl = slice(rank * per, rank * per + per)
    gG, gD = _half_step(m, A[sl], B[sl], label[sl], parallel.sync_gradients)
    # replicas must hold identical weights after the step (same averaged gradients into the same Adam state)
    flat = torch.cat([p.detach().reshape(-1) for net in (m.netG, m.netD) for p in net.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], b) for b in both), 'replicas diverged after one step'
    if rank == 0:
        torch.save({'gG': gG, 'gD': gD, 'pG': {k: v.detach() for k, v in m.netG.named_parameters()}}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [2, 4])
def test_n_rank_step_equals_single_process_reference(tmp_path, world):
    """world 2: two samples per rank; world 4: ONE sample per rank (per-rank BatchNorm statistics over a single image: the
    extreme of DataParallel's per-replica statistics)"""
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out)
    # single-process restatement of "per-rank statistics, averaged gradients"
    torch.set_num_threads(2 if world <= 2 else 1)      # (as in the workers: oneDNN's reduction order depends on it)
    A = W.seeded_tensor((4, 3, 32, 32), 900)
    B = W.seeded_tensor((4, 3, 32, 32), 901)
    label = [0, 2, 2, 0]

    def close(g, want, what):
        if world == 2:
            assert torch.allclose(g, want, rtol=1e-4, atol=1e-6), what
        else:      # one image per rank: BatchNorm statistics over a handful of values amplify rounding; judged per tensor
            err = float((g.double() - want.double()).norm() / (want.double().norm() + 1e-12))
            assert err <= 1e-3 or float((g - want).abs().max()) <= 1e-6, '%s: relative L2 %.3e' % (what, err)
    halves = []
    per = 4 // world
    for r in range(world):
        m = _build_step()
        sl = slice(r * per, r * per + per)
        halves.append(_half_step(m, A[sl], B[sl], label[sl], lambda o: None))
    for k, g in got['gG'].items():
        want = sum(h[0][k] for h in halves) / world
        close(g, want, 'G grad ' + k)
    # backward_D runs on fake_B from forward() (made BEFORE the G step, models/wsgan_emb_model.py:478-484) and on D's own
    # weights, so the D gradients do not depend on how G was stepped in between: they are comparable term by term too
    assert set(got['gD']) == set(halves[0][1])
    for k, g in got['gD'].items():
        want = sum(h[1][k] for h in halves) / world
        close(g, want, 'D grad ' + k)


# ---- round 4: the first multi-rank run must diagnose itself -- all three pieces over gloo on the CPU
def _diag_worker(rank, world, port, mode):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from pcgan_amd.hip import parallel
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    parallel.init_process_group('gloo')          # runs selfcheck_collectives() itself
    assert parallel.selfcheck_collectives() == 0.0
    # a collective that does not average (what a broken ReduceOp.AVG would look like) is caught on every rank
    real = parallel.allreduce_mean_

    def broken(flat, async_op=False):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat
    parallel.allreduce_mean_ = broken
    try:
        with pytest.raises(parallel.CollectiveSelfCheckError, match='AVG'):
            parallel.selfcheck_collectives()
    finally:
        parallel.allreduce_mean_ = real
    ident = parallel.rank_identity()
    assert ident['rank'] == rank and ident['backend'] == 'gloo' and ident['pid'] == os.getpid() and ident['host']
    # replica hash check: equal replicas pass, a single flipped bit on one rank is caught
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    parallel.broadcast_parameters(D)
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    assert parallel.ddp_check(opt, 'D', every=2) is False          # first call: not due yet
    assert parallel.ddp_check(opt, 'D', every=2) is True           # second: compared, equal
    if rank == world - 1:
        bits = opt.flat.view(torch.int32)
        bits[12345 % bits.numel()] ^= 1                            # one ulp on one rank
    if mode == 'raise':
        with pytest.raises(parallel.ReplicaDivergenceError, match='DIVERGED'):
            parallel.ddp_check(opt, 'D2', every=1, exit_on_divergence=False)
        dist.barrier()
        dist.destroy_process_group()
    else:
        parallel.ddp_check(opt, 'D2', every=1)                     # every rank exits with code 3


@pytest.mark.timeout(300)
def test_collective_selfcheck_rank_identity_and_replica_hash_check():
    mp.spawn(_diag_worker, args=(2, _free_port(), 'raise'), nprocs=2, join=True)


@pytest.mark.timeout(300)
def test_replica_divergence_ends_the_job_with_exit_code_3():
    with pytest.raises(mp.ProcessExitedException) as e:
        mp.spawn(_diag_worker, args=(2, _free_port(), 'exit'), nprocs=2, join=True)
    assert e.value.exit_code == 3


def _one_rank_worker(rank, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), PCGAN_FORCE_COLLECTIVES='1')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    torch.set_num_threads(1)
    from pcgan_amd.hip import parallel
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    assert parallel.FORCE_COLLECTIVES
    assert parallel.init_process_group('gloo')[:2] == (1, 0) and dist.is_initialized() and parallel.is_distributed()
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    parallel.broadcast_parameters(D)
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt.gflat.copy_(torch.arange(opt.gflat.numel(), dtype=torch.float32))
    want = opt.gflat.clone()
    parallel.sync_gradients(opt)
    parallel.sync_gradients(opt, async_op=True)()
    assert torch.equal(opt.gflat, want)          # the average over one rank
    assert parallel.ddp_check(opt, 'D', every=1) is True
    assert parallel.shard_batch(torch.arange(4)).tolist() == [0, 1, 2, 3]
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_one_rank_rehearsal_runs_every_collective():
    """PCGAN_FORCE_COLLECTIVES=1: the one-GPU box's way to run the RCCL branch (tests/test_gpu_ddp.py::test_one_rank_over_rccl);
    here the same switch over gloo -- the group exists, the collectives run, and averaging over one rank changes nothing"""
    mp.spawn(_one_rank_worker, args=(_free_port(),), nprocs=1, join=True)
