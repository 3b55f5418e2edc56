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
    torch.set_num_threads(2)
    from pcgan_amd.hip import parallel
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    w, r, _ = parallel.init_process_group('gloo')
    assert (w, r) == (world, rank) and parallel.is_distributed()
    # (a) sharding
    full = torch.arange(8).view(8, 1)
    assert parallel.shard_batch(full).flatten().tolist() == list(range(rank * 4, rank * 4 + 4))
    # (b) fused optimizer's flat gradient buffer: one all-reduce, average
    D = networks.define_D(3, 1, 8, 'n_layers', 3, 'batch', True, 'normal')
    parallel.broadcast_parameters(D)
    opt = FusedAdam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    opt.gflat.fill_(float(rank + 1))
    parallel.sync_gradients(opt)
    assert torch.allclose(opt.gflat, torch.full_like(opt.gflat, 1.5))
    chk = torch.stack([p.detach().sum() for p in D.parameters()]).sum().reshape(1)
    both = [torch.zeros(1), torch.zeros(1)]
    dist.all_gather(both, chk)
    assert torch.equal(both[0], both[1]), 'replicas must start from rank 0 weights'
    # (c) the step on this rank's contiguous slice with averaged gradients
    m = _build_step()
    A = W.seeded_tensor((4, 3, 32, 32), 900)
    B = W.seeded_tensor((4, 3, 32, 32), 901)
    label = [0, 2, 2, 0]
    sl = slice(rank * 2, rank * 2 + 2)
    gG, gD = _half_step(m, A[sl], B[sl], label[sl], parallel.sync_gradients)
    # replicas must hold identical weights after the step (same averaged gradients into the same Adam state)
    flat = torch.cat([p.detach().reshape(-1) for net in (m.netG, m.netD) for p in net.parameters()])
    both = [torch.zeros_like(flat), torch.zeros_like(flat)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), 'replicas diverged after one step'
    if rank == 0:
        torch.save({'gG': gG, 'gD': gD, 'pG': {k: v.detach() for k, v in m.netG.named_parameters()}}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_step_equals_single_process_reference(tmp_path):
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    # single-process restatement of "per-rank statistics, averaged gradients"
    torch.set_num_threads(2)
    A = W.seeded_tensor((4, 3, 32, 32), 900)
    B = W.seeded_tensor((4, 3, 32, 32), 901)
    label = [0, 2, 2, 0]
    halves = []
    for r in range(2):
        m = _build_step()
        sl = slice(r * 2, r * 2 + 2)
        halves.append(_half_step(m, A[sl], B[sl], label[sl], lambda o: None))
    for k, g in got['gG'].items():
        want = 0.5 * (halves[0][0][k] + halves[1][0][k])
        assert torch.allclose(g, want, rtol=1e-4, atol=1e-6), 'G grad ' + k
    # backward_D runs on fake_B from forward() (made BEFORE the G step, models/wsgan_emb_model.py:478-484) and on D's own
    # weights, so the D gradients do not depend on how G was stepped in between: they are comparable term by term too
    assert set(got['gD']) == set(halves[0][1])
    for k, g in got['gD'].items():
        want = 0.5 * (halves[0][1][k] + halves[1][1][k])
        assert torch.allclose(g, want, rtol=1e-4, atol=1e-6), 'D grad ' + k
