"""Composite C-ABI entry points (include/pcgan_hip.h "composite": pcgan_resblock_fwd / pcgan_resblock_bwd): one ResnetBlock
(reference models/networks.py:616-652) per library call instead of 6 + 10 per-op calls.  They issue the same kernels with the same
arguments in the same order, so everything must be BIT-IDENTICAL to the per-op path: block output, input gradient (the skip
connection's `grad +=` is summed in the data-gradient epilogue: one fp32 add either way), the weight / bias gradients accumulated into
the optimizer's flat buffer over two passes, the InstanceNorm running statistics -- and a whole full-size optimize_parameters()."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _block(dev, C):
    from pcgan_amd.hip import nn as hnn
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    torch.manual_seed(5)
    blk = networks.ResnetBlock(C, 'reflect', lambda c: hnn.InstanceNorm2d(c, affine=False, track_running_stats=True), 0, True).to(dev)
    with torch.no_grad():
        for p in blk.parameters():
            p.mul_(3.0)
    opt = FusedAdam(blk.parameters(), lr=2e-4, betas=(0.5, 0.999))
    return blk, opt


def _run(blk, opt, xs, dys):
    """two passes through one block (as the generator's two passes per step), gradients accumulated"""
    from pcgan_amd.hip import ops
    opt.zero_grad()
    outs, dxs = [], []
    for x0, dy in zip(xs, dys):
        x = x0.clone().requires_grad_(True)
        # the block input arrives from an instance-norm kernel in the generator: hand its operand maxima over the same way
        xin = x * 1.0
        if xin.dtype == torch.float32:
            ops._attach_amax(xin, ops.amax_of(xin.detach()))
        out = blk(xin)
        out.backward(dy)
        outs.append(out.detach().clone())
        dxs.append(x.grad.detach().clone())
    ops.join_side_stream()
    torch.cuda.synchronize()
    return outs, dxs, opt.gflat.detach().clone(), {k: v.detach().clone() for k, v in blk.state_dict().items() if 'running' in k}


@pytest.mark.parametrize('N,C,H,dt', [(4, 256, 32, 'fp32'), (2, 256, 64, 'fp32'), (32, 256, 32, 'fp32'), (4, 256, 32, 'bf16'), (32, 256, 32, 'bf16'),
                                      (2, 256, 64, 'bf16')])
def test_resblock_composite_is_the_per_op_sequence(dev, monkeypatch, N, C, H, dt):
    """fp32 tensors (fp16 two-piece route) and, since round 4, bf16 tensors (one-product kernels; the skip connection's gradient added by
    pcgan_add where the per-op path lets autograd add it: one bf16 rounding of the fp32 sum either way)"""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    g = torch.Generator().manual_seed(N + H)
    tdt = torch.float32 if dt == 'fp32' else torch.bfloat16
    xs = [(torch.randn(N, C, H, H, generator=g) * 0.7).to(dev).to(tdt) for _ in range(2)]
    dys = [torch.randn(N, C, H, H, generator=g).to(dev).to(tdt) for _ in range(2)]
    b1, o1 = _block(dev, C)
    b2, o2 = _block(dev, C)
    assert all(torch.equal(p, q) for p, q in zip(b1.parameters(), b2.parameters()))
    before = dict(ops.COMPOSITE_STATS)
    monkeypatch.setattr(ops, 'COMPOSITE', True)
    r1 = _run(b1, o1, xs, dys)
    assert ops.COMPOSITE_STATS['fwd'] == before['fwd'] + 2 and ops.COMPOSITE_STATS['bwd'] == before['bwd'] + 2, 'the composite path was not taken'
    monkeypatch.setattr(ops, 'COMPOSITE', False)
    r2 = _run(b2, o2, xs, dys)
    assert ops.COMPOSITE_STATS['fwd'] == before['fwd'] + 2, 'the per-op run must not use the composite'
    for i in range(2):
        assert torch.equal(r1[0][i], r2[0][i]), 'block output, pass %d' % i
        assert torch.equal(r1[1][i], r2[1][i]), 'input gradient (skip connection summed in the epilogue), pass %d' % i
    assert torch.equal(r1[2], r2[2]), 'weight / bias gradients in the flat buffer'
    assert float(r1[2].abs().max()) > 0
    for k in r1[3]:
        assert torch.equal(r1[3][k], r2[3][k]), k
    # no-grad call (visuals / samplers): same forward, nothing kept
    monkeypatch.setattr(ops, 'COMPOSITE', True)
    with torch.no_grad():
        xin = xs[0] * 1.0
        if dt == 'fp32':
            ops._attach_amax(xin, ops.amax_of(xin))
        y = b1(xin)
    monkeypatch.setattr(ops, 'COMPOSITE', False)
    with torch.no_grad():
        xin = xs[0] * 1.0
        if dt == 'fp32':
            ops._attach_amax(xin, ops.amax_of(xin))
        y2 = b2(xin)
    assert torch.equal(y, y2)


def test_composite_falls_back_where_it_does_not_apply(dev, monkeypatch):
    """eval-mode running statistics, frozen parameters under autograd, small channel counts: the layer-by-layer path runs (still HIP)"""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    blk, opt = _block(dev, 256)
    x = torch.randn(2, 256, 32, 32).to(dev).requires_grad_(True)
    before = dict(ops.COMPOSITE_STATS)
    for p in blk.parameters():
        p.requires_grad_(False)
    blk(x).sum().backward()                   # frozen parameters, gradient to x only
    assert ops.COMPOSITE_STATS == before and x.grad is not None
    small, _ = _block(dev, 32)
    small(torch.randn(2, 32, 16, 16).to(dev).requires_grad_(True)).sum().backward()
    assert ops.COMPOSITE_STATS == before


def test_full_size_step_composite_equals_per_op(tmp_path, dev, monkeypatch):
    """two optimize_parameters() of the config-2 networks (9-block generator, batch 2, routing threshold lowered so the residual
    blocks take the production kernels): parameters of G and D, images and losses bit-identical (a) with the nine blocks of a generator
    pass as ONE library call and autograd node per direction (round 4: pcgan_restrunk_fwd / _bwd), (b) with one composite call per
    block, (c) layer by layer"""
    import bench
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    res = []
    for comp, trunk in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(ops, 'COMPOSITE', comp)
        monkeypatch.setattr(ops, 'TRUNK', trunk)
        before = dict(ops.COMPOSITE_STATS)
        torch.manual_seed(0)
        tmp = tmp_path / ('composite%d%d' % (int(comp), int(trunk)))
        tmp.mkdir()
        model, opt = bench.build_model(0, 2, 128, str(tmp), seed=3)
        for it in range(2):
            model.set_input(bench.synthetic_batch(2, 128, 0, it))
            model.optimize_parameters()
        torch.cuda.synchronize()
        took = ops.COMPOSITE_STATS['fwd'] - before['fwd']
        assert took == (2 * 2 * 9 if comp else 0), took
        trunks = ops.COMPOSITE_STATS.get('trunk_fwd', 0) - before.get('trunk_fwd', 0), ops.COMPOSITE_STATS.get('trunk_bwd', 0) - before.get('trunk_bwd', 0)
        assert trunks == ((4, 4) if trunk else (0, 0)), trunks      # 2 steps x 2 generator passes: one call per pass and direction
        res.append({'G': model.optimizer_G.flat.detach().clone(), 'D': model.optimizer_D.flat.detach().clone(),
                    'fake_B': model.fake_B.detach().clone(), 'rec_A': model.rec_A.detach().clone(),
                    'losses': dict(model.get_current_losses()),
                    'bufs': {k: v.detach().clone() for k, v in model.netG.state_dict().items() if 'running' in k}})
    for b in res[1:]:
        a = res[0]
        for k in ('G', 'D', 'fake_B', 'rec_A'):
            assert torch.equal(a[k], b[k]), k
        assert a['losses'] == b['losses']
        for k in a['bufs']:
            assert torch.equal(a['bufs'][k], b['bufs'][k]), k


@pytest.mark.parametrize('nb,N,H,dt', [(3, 4, 32, 'fp32'), (9, 32, 32, 'fp32'), (2, 2, 64, 'fp32'), (3, 4, 32, 'bf16'), (9, 32, 32, 'bf16')])
def test_restrunk_is_the_chain_of_block_calls(dev, monkeypatch, nb, N, H, dt):
    """pcgan_restrunk_fwd / _bwd on a chain of nb ResnetBlocks inside an nn.Sequential (as in ResnetGenerator): output, input gradient,
    every weight / bias gradient accumulated over two passes and every running statistic BIT-IDENTICAL to one composite call per block;
    the chain's output carries its plane maxima for the next convolution"""
    import torch.nn as tnn
    from pcgan_amd.hip import ops, nn as hnn
    from pcgan_amd.hip.optim import FusedAdam
    from pcgan_amd.models import networks
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    C = 256

    def build():
        torch.manual_seed(9)
        seq = tnn.Sequential(*[networks.ResnetBlock(C, 'reflect', lambda c: hnn.InstanceNorm2d(c, affine=False, track_running_stats=True), 0, True)
                               for _ in range(nb)]).to(dev)
        with torch.no_grad():
            for p in seq.parameters():
                p.mul_(3.0)
        return seq, FusedAdam(seq.parameters(), lr=2e-4, betas=(0.5, 0.999))

    g = torch.Generator().manual_seed(nb + N)
    tdt = torch.float32 if dt == 'fp32' else torch.bfloat16
    xs = [(torch.randn(N, C, H, H, generator=g) * 0.7).to(dev).to(tdt) for _ in range(2)]
    dys = [torch.randn(N, C, H, H, generator=g).to(dev).to(tdt) for _ in range(2)]

    def run(trunk):
        monkeypatch.setattr(ops, 'TRUNK', trunk)
        seq, opt = build()
        before = ops.COMPOSITE_STATS.get('trunk_fwd', 0), ops.COMPOSITE_STATS.get('trunk_bwd', 0)
        opt.zero_grad()
        outs, dxs = [], []
        for x0, dy in zip(xs, dys):
            x = x0.clone().requires_grad_(True)
            xin = x * 1.0
            if dt == 'fp32':
                ops._attach_amax(xin, ops.amax_of(xin.detach()))
            out = hnn.run_sequential(seq, xin)
            assert dt != 'fp32' or ('_pcgan_amax' in out.__dict__ and out.__dict__['_pcgan_amax'][0] == out._version), 'the chain output lost its plane maxima'
            out.backward(dy)
            outs.append(out.detach().clone())
            dxs.append(x.grad.detach().clone())
        ops.join_side_stream()
        torch.cuda.synchronize()
        took = ops.COMPOSITE_STATS.get('trunk_fwd', 0) - before[0], ops.COMPOSITE_STATS.get('trunk_bwd', 0) - before[1]
        assert took == ((2, 2) if trunk else (0, 0)), took
        return outs, dxs, opt.gflat.detach().clone(), {k: v.detach().clone() for k, v in seq.state_dict().items() if 'running' in k}

    a, b = run(True), run(False)
    for i in range(2):
        assert torch.equal(a[0][i], b[0][i]), 'chain output, pass %d' % i
        assert torch.equal(a[1][i], b[1][i]), 'input gradient, pass %d' % i
    assert torch.equal(a[2], b[2]) and float(a[2].abs().max()) > 0, 'weight / bias gradients in the flat buffer'
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
