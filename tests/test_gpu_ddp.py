"""GPU: the data-parallel step of the PRODUCT model (WSGANEmbModel.update_G / update_D + FusedAdam's flat gradient
buffer + parallel.sync_gradients + broadcast_parameters) under more than one rank.

The box has one GPU, so the two ranks share it and the collective runs over gloo (PCGAN_DIST_BACKEND=gloo; the RCCL
branch differs only in the all-reduce call).  Fresh child processes (tests/ddp_worker.py) -- the test process itself
never joins a process group.  Checked:
  * replicas built from DIFFERENT seeds start from rank 0's weights (they end bit-identical after the step);
  * the G and D gradient buffers at the moment of each optimizer step equal the mean of two single-rank runs over the two
    halves of the batch (per-rank BatchNorm / InstanceNorm statistics, averaged gradients: the reference's
    nn.DataParallel semantics, models/networks.py:96-102) -- relative L2 1e-6 (same kernels, same order; the only new
    arithmetic is (a + b) * 0.5);
  * both ranks' parameters are bit-equal after the step."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'ddp_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(rank, world, port, overlap='0'):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env['PCGAN_DDP_OVERLAP'] = '1' if overlap == '1' else '0'
    env['PCGAN_DDP_GRAD_STREAM'] = '1' if overlap == 'grad-stream' else '0'
    if world > 1:
        env.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   PCGAN_DIST_BACKEND='gloo')
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    return env


@pytest.mark.timeout(900)
@pytest.mark.parametrize('dtype,overlap', [('fp32', '0'), ('fp32', '1'), ('fp32', 'grad-stream'), ('bf16', '0')],
                         ids=['fp32-sequential', 'fp32-overlapped-allreduce', 'fp32-allreduce-on-gradient-stream',
                              'bf16-sequential(config 3: bf16 x ranks)'])
def test_two_ranks_product_step(dev, tmp_path, dtype, overlap):
    port = _free_port()
    outs = [str(tmp_path / ('rank%d.pt' % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, WORKER, outs[r], str(2 * r), str(2 * r + 2), dtype], cwd=ROOT, env=_env(r, 2, port, overlap),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, logs[r][-3000:])
    ranks = [torch.load(o) for o in outs]
    assert all(r['distributed'] for r in ranks)
    # single-rank runs over the same halves, both from rank 0's seed
    singles = []
    for h in range(2):
        o = str(tmp_path / ('single%d.pt' % h))
        p = subprocess.run([sys.executable, WORKER, o, str(2 * h), str(2 * h + 2), dtype], cwd=ROOT, env=_env(0, 1, 0), capture_output=True,
                           text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        singles.append(torch.load(o))
        assert not singles[-1]['distributed']
    for tag in ('gG', 'gD'):
        want = (singles[0][tag].double() + singles[1][tag].double()) * 0.5
        for r in range(2):
            err = float((ranks[r][tag].double() - want).norm() / (want.norm() + 1e-30))
            assert err <= 1e-6, '%s of rank %d vs the mean of the two single-rank runs: relative L2 %.3e' % (tag, r, err)
        assert torch.equal(ranks[0][tag], ranks[1][tag]), tag + ' differs between the ranks after the all-reduce'
    for tag in ('pG', 'pD'):
        assert torch.equal(ranks[0][tag], ranks[1][tag]), 'replicas are not bit-equal after the step (%s)' % tag
    # the losses are per-rank (each rank's half), as under the reference's rank-0 logging
    for h in range(2):
        for k, v in singles[h]['losses'].items():
            assert abs(ranks[h]['losses'][k] - v) <= 1e-5 * max(1.0, abs(v)), 'rank %d loss %s' % (h, k)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('overlap', ['0', '1', 'grad-stream'], ids=['sequential', 'overlapped-allreduce', 'allreduce-on-gradient-stream'])
def test_one_rank_over_rccl(dev, tmp_path, overlap):
    """The RCCL branch itself on the hardware there is: RCCL refuses two ranks on one device, so a world of ONE rank runs the whole
    collective side of the step (PCGAN_FORCE_COLLECTIVES=1): communicator set-up, the start-up self-check, broadcast_parameters,
    ReduceOp.AVG on the flat gradient buffers -- in the launched-now / waited-later form too, RCCL's stream beside the step's
    side and branch streams -- and the work.wait() hand-back.  Averaging over one rank is the identity, so gradients at each
    optimizer step, parameters after it and losses must be BIT-equal to the run without a process group."""
    env = _env(0, 1, 0, overlap)
    env.update(PCGAN_FORCE_COLLECTIVES='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               PCGAN_DIST_BACKEND='nccl', PCGAN_DDP_CHECK='1')
    outs = {}
    for tag, e in (('rccl', env), ('plain', _env(0, 1, 0, overlap))):
        o = str(tmp_path / (tag + '.pt'))
        p = subprocess.run([sys.executable, WORKER, o, '0', '4', 'fp32'], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, '%s run failed:\n%s' % (tag, (p.stdout + p.stderr)[-3000:])
        outs[tag] = torch.load(o)
    r, s = outs['rccl'], outs['plain']
    assert r['distributed'] and not s['distributed']
    assert r['identity']['backend'] == 'nccl' and r['identity']['rccl_version'][0].isdigit(), r['identity']
    for tag in ('gG', 'gD', 'pG', 'pD'):
        assert torch.equal(r[tag], s[tag]), '%s differs between the one-rank RCCL run and the plain run' % tag
    assert r['losses'] == s['losses']
