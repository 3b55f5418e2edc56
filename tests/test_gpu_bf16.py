"""GPU parity of the bf16 path (BASELINE configs[2]; `--dtype bf16`, C-ABI dtype PCGAN_BF16).

What the path is: activations and their gradients are STORED as bf16; every kernel computes in fp32 registers (fp32 MFMA
accumulators, fp32 norm statistics, fp32 loss reductions), parameters / parameter gradients / Adam stay fp32.  The
residual-block convolutions additionally round the WEIGHTS to bf16 (one bf16 product per term on the bf16 matrix pipe).
The reference has no bf16 mode, so parity is stated against the fp32 oracle with tolerances that come from the
rounding model, confirmed by measurement (profiles/r02_bf16_parity.txt):

  * one op, inputs already bf16 values: the fp32 result is rounded ONCE to bf16 on store: |err| <= 2^-9 |y| -> 4e-3 of the
    largest magnitude (kernels that also round the weights: compared with a reference that uses the same rounded weights
    at 4e-3, and with the fp32 weights at 2e-2 -- weight rounding 2^-9 relative, sqrt(K)-averaged);
  * fp32 outputs of bf16 inputs (weight gradients, bias gradients, statistics, losses): no output rounding -> 2e-5 / 1e-4 as
    in the fp32 tests, the reference fed the same bf16-rounded inputs;
  * whole networks / the whole step against the fp32 ORACLE (fp32 inputs): every stored activation and gradient carries
    2^-9 relative rounding noise, ~60-120 such layers follow each other and InstanceNorm's backward is a cancellation, so
    errors are judged by relative L2 norm AGAINST A CALIBRATION: the same oracle nets run under stock PyTorch's CPU bf16
    autocast (`autocast_bf16`).  The HIP bf16 path may lose at most twice what that loses (+ 1e-2 .. 5e-2 absolute); the
    measured pairs are printed by the tests and kept in profiles/r02_bf16_parity.txt (they agree within ~10 %).
"""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import networks_ref as N
from oracle import ops_ref as R
from oracle import weights as W
from test_gpu_ops import CONV_CASES, _conv_ref
from util_cmp import assert_close

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _r(t):
    """round a CPU fp32 tensor to bf16 values (kept as fp32)"""
    return t.to(BF).float()


def _rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_bf16(case, dev):
    from pcgan_amd.hip import ops
    name, Nb, C, H, Wd, K, Rk, stride, pad, pm = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = _r(torch.randn(Nb, C, H, Wd, generator=g))
    w = torch.randn(K, C, Rk, Rk, generator=g) * (1.0 / (C * Rk * Rk) ** 0.5)
    b = torch.randn(K, generator=g) * 0.1
    P = (H + 2 * pad - Rk) // stride + 1
    Q = (Wd + 2 * pad - Rk) // stride + 1
    dy = _r(torch.randn(Nb, K, P, Q, generator=g))
    y64, dx64, dw64, db64 = _conv_ref(x, w, b, stride, pad, pm, dy)
    xd, dyd, wd, bd = x.to(dev).to(BF), dy.to(dev).to(BF), w.to(dev), b.to(dev)
    for cache in (None, {}):
        y = ops.conv2d_fwd(xd, wd, bd, stride, pad, pm, pack_cache=cache)
        assert y.dtype == BF
        assert_close(y.float(), y64, 2e-2 if cache is not None else 4e-3, name + ' bf16 forward (packed=%s)' % (cache is not None))
        dx = ops.conv2d_bwd_data(dyd, wd, (H, Wd), stride, pad, pm, pack_cache={} if cache is not None else None)
        assert dx.dtype == BF
        assert_close(dx.float(), dx64, 2e-2 if cache is not None else 4e-3, name + ' bf16 data gradient')
    dw = ops.conv2d_bwd_weight(xd, dyd, tuple(w.shape), stride, pad, pm)
    assert dw.dtype == torch.float32
    assert_close(dw, dw64, 1e-4, name + ' weight gradient of bf16 tensors (fp32 result)')
    db = ops.channel_sum(dyd)
    assert_close(db, db64, 2e-5, name + ' bias gradient of a bf16 tensor (fp32 result)')


@pytest.mark.parametrize('Nb,H', [(32, 32), (8, 64)])
def test_residual_conv_bf16_route(Nb, H, dev):
    """the 256 -> 256 3x3 reflect convolution at full size takes the one-product bf16 MFMA kernels (forward, data gradient,
    weight gradient): against a float64 reference that uses the SAME bf16-rounded weights (what the kernel computes)"""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(7 + Nb)
    n_ref = 2                                            # images checked against the CPU reference
    x = _r(torch.randn(Nb, 256, H, H, generator=g))
    w = torch.randn(256, 256, 3, 3, generator=g) * 0.02
    dy = _r(torch.randn(Nb, 256, H, H, generator=g))
    xd, dyd, wd = x.to(dev).to(BF), dy.to(dev).to(BF), w.to(dev)
    cf, cb = {}, {}
    y = ops.conv2d_fwd(xd, wd, None, 1, 1, 1, pack_cache=cf)
    dx = ops.conv2d_bwd_data(dyd, wd, (H, H), 1, 1, 1, pack_cache=cb)
    dw = ops.conv2d_bwd_weight(xd, dyd, (256, 256, 3, 3), 1, 1, 1)
    assert any(k[0] == ops.PASS_FWD_BSPLIT for k in cf) and any(k[0] == ops.PASS_BWD_BSPLIT for k in cb), 'not on the bf16 MFMA route'
    wr = _r(w)
    xs = x[:n_ref].double().requires_grad_(True)
    yr = R.conv2d(xs, wr.double(), None, 1, 1, 1)
    yr.backward(dy[:n_ref].double())
    assert_close(y[:n_ref].float(), yr.detach(), 4e-3, 'bf16 route forward')
    assert_close(dx[:n_ref].float(), xs.grad, 4e-3, 'bf16 route data gradient')
    # weight gradient: fp32 result of bf16 x / dy (the dy pieces are exact in bf16): adjoint identity over the whole batch
    lhs = float((y.double() * dyd.double()).sum())
    # <conv_wr(x), dy> with y rounded: use the unrounded identity <w, dw> = <x, dgrad_w(dy)> instead, both in fp32 outputs
    wsum = float((wr.double().to(dev) * dw.double()).sum())
    xsum = float((xd.double() * dx.double()).sum())
    scale = float(y.double().norm() * dyd.double().norm())
    assert abs(wsum - lhs) <= 3e-3 * scale and abs(xsum - lhs) <= 3e-3 * scale, (lhs, wsum, xsum, scale)
    w8 = wr.double().requires_grad_(True)
    R.conv2d(x[:4].double(), w8, None, 1, 1, 1).backward(dy[:4].double())
    dw4 = ops.conv2d_bwd_weight(xd[:4].contiguous(), dyd[:4].contiguous(), (256, 256, 3, 3), 1, 1, 1)
    assert_close(dw4, w8.grad, 1e-4, 'bf16 route weight gradient (4 images)')


@pytest.mark.parametrize('shape', [(2, 8, 32, 32), (3, 5, 7, 7), (2, 4, 128, 128), (2, 6, 15, 15)])
@pytest.mark.parametrize('act,res', [(0, False), (1, False), (0, True)])
def test_instance_norm_bf16(shape, act, res, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(13)
    x = _r(torch.randn(shape, generator=g) * 1.5 + 0.3)
    r = _r(torch.randn(shape, generator=g)) if res else None
    dy = _r(torch.randn(shape, generator=g))
    x64 = x.double().requires_grad_(True)
    y64 = R.instance_norm(x64)
    if res:
        y64 = y64 + r.double()
    y64 = R.activation(y64, act)
    y64.backward(dy.double())
    xd = x.to(dev).to(BF)
    y, mean, m2 = ops.instnorm_fwd(xd, r.to(dev).to(BF) if res else None, 1e-5, act, 0.0)
    assert y.dtype == BF and mean.dtype == torch.float32
    assert_close(y.float(), y64, 4e-3, 'bf16 instnorm fwd')
    mean_ref = x.double().mean(dim=(2, 3)).reshape(-1)
    assert_close(mean, mean_ref, 1e-5, 'fp32 plane mean of a bf16 tensor', atol=1e-6)
    # backward: the saved y is the ROUNDED y, its ReLU mask is that of the stored value (identical sign)
    dx = ops.instnorm_bwd(dy.to(dev).to(BF), xd, y, mean, m2, 1e-5, act, 0.0)
    assert dx.dtype == BF
    assert_close(dx.float(), x64.grad, 6e-3, 'bf16 instnorm bwd')


@pytest.mark.parametrize('shape', [(4, 8, 16, 16), (3, 5, 7, 7), (2, 16, 64, 64), (40, 3, 4, 4), (32, 6, 40, 40)])
@pytest.mark.parametrize('act,slope,res', [(2, 0.2, False), (1, 0.0, True), (0, 0.0, False)])
def test_batch_norm_bf16(shape, act, slope, res, dev):
    from pcgan_amd.hip import functional as F
    g = torch.Generator().manual_seed(17)
    C = shape[1]
    x = _r(torch.randn(shape, generator=g) * 1.3 + 0.2)
    r = _r(torch.randn(shape, generator=g)) if res else None
    dy = _r(torch.randn(shape, generator=g))
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y64 = torch.nn.functional.batch_norm(x64, rm, rv, g64, b64, True, 0.1, 1e-5)
    if res:
        y64 = y64 + r.double()
    y64 = R.activation(y64, act, slope)
    y64.backward(dy.double())
    xd = x.to(dev).to(BF).requires_grad_(True)
    gd, bd = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = F.batch_norm_act(xd, gd, bd, rmd, rvd, 0.1, 1e-5, act, slope, r.to(dev).to(BF) if res else None, True, None)
    assert y.dtype == BF
    assert_close(y.float(), y64, 4e-3, 'bf16 batchnorm fwd')
    assert_close(rmd, rm, 1e-5, 'running mean (fp32 statistics of a bf16 tensor)', atol=1e-6)
    assert_close(rvd, rv, 1e-5, 'running var')
    y.backward(dy.to(dev).to(BF))
    assert_close(xd.grad.float(), x64.grad, 8e-3, 'bf16 batchnorm dx')
    # dgamma / dbeta are fp32 sums over the bf16 dy and the ROUNDED y's activation mask / fp32 xhat
    assert_close(gd.grad, g64.grad, 2e-3, 'dgamma', atol=1e-4)
    assert_close(bd.grad, b64.grad, 2e-3, 'dbeta', atol=1e-4)


def test_pointwise_pool_resize_losses_bf16(dev):
    from pcgan_amd.hip import ops, functional as F
    g = torch.Generator().manual_seed(23)
    x = _r(torch.randn(3, 6, 20, 20, generator=g))
    xd = x.to(dev).to(BF)
    # cast round trip is exact on bf16 values; fp32 -> bf16 is round-to-nearest-even like torch
    f = torch.randn(1000, generator=g) * 3
    assert torch.equal(ops.cast(f.to(dev), BF).cpu(), f.to(BF))
    assert torch.equal(ops.cast(xd, torch.float32).cpu(), x)
    for act, slope in ((1, 0.0), (2, 0.2), (3, 0.0), (4, 0.0)):
        y = ops.act_fwd(xd, act, slope)
        assert_close(y.float(), R.activation(x.double(), act, slope), 4e-3, 'bf16 activation %d' % act)
    # max pooling: values are selected, not computed -> exact; indices exact
    y, arg = ops.maxpool_fwd(xd, 3, 2, 1)
    yr, ar = torch.nn.functional.max_pool2d(x, 3, 2, 1, return_indices=True)
    assert torch.equal(y.float().cpu(), yr) and torch.equal(arg.cpu().long(), ar)
    dy = _r(torch.randn(yr.shape, generator=g))
    x64 = x.double().requires_grad_(True)
    torch.nn.functional.max_pool2d(x64, 3, 2, 1).backward(dy.double())
    assert_close(ops.maxpool_bwd(dy.to(dev).to(BF), arg, (20, 20), 3, 2, 1).float(), x64.grad, 4e-3, 'bf16 maxpool bwd')
    for is_max in (False, True):
        yp, argp = ops.global_pool_fwd(xd, is_max)
        ref = x.double().amax(dim=(2, 3), keepdim=True) if is_max else x.double().mean(dim=(2, 3), keepdim=True)
        assert_close(yp.float(), ref, 4e-3, 'bf16 global pool')
    up = ops.bilinear_fwd(xd, (33, 33))
    assert_close(up.float(), torch.nn.functional.interpolate(x.double(), size=(33, 33), mode='bilinear', align_corners=True), 4e-3, 'bf16 bilinear')
    # z concat (ratings fp32), dropout scaling
    z = torch.randn(3, 2, generator=g)
    cz = ops.concat_z(xd, z.to(dev))
    assert cz.dtype == BF and torch.equal(cz[:, :6].float().cpu(), x) and torch.equal(cz[:, 6:, 0, 0].cpu(), z.to(BF))
    # losses: fp32 value of bf16 inputs, bf16 gradient
    p = _r(torch.rand(4, 1, 14, 14, generator=g) * 0.98 + 0.01)
    t = torch.tensor([0.0, 1.0, 1.0, 0.0])
    pd = p.to(dev).to(BF).requires_grad_(True)
    loss = F.bce_loss(pd, t.to(dev))
    p64 = p.double().requires_grad_(True)
    ref = torch.nn.functional.binary_cross_entropy(p64, t.double().view(4, 1, 1, 1).expand_as(p64))
    ref.backward()
    assert loss.dtype == torch.float32 and abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-7
    loss.backward()
    assert_close(pd.grad.float(), p64.grad, 4e-3, 'bf16 BCE gradient')
    a, b = _r(torch.randn(2, 3, 16, 16, generator=g)), _r(torch.randn(2, 3, 16, 16, generator=g))
    for fn, rf in ((F.l1_loss, torch.nn.functional.l1_loss), (F.mse_loss, torch.nn.functional.mse_loss)):
        ad = a.to(dev).to(BF).requires_grad_(True)
        lv = fn(ad, b.to(dev).to(BF))
        a64 = a.double().requires_grad_(True)
        rv = rf(a64, b.double())
        rv.backward()
        assert abs(float(lv) - float(rv)) <= 2e-6 * abs(float(rv)) + 1e-7
        lv.backward()
        assert_close(ad.grad.float(), a64.grad, 4e-3, 'bf16 loss gradient')


def _run_net(net, inputs, dy):
    xs = [i.clone().requires_grad_(i.is_floating_point()) for i in inputs]
    y = net(*xs)
    y.backward(dy.to(device=y.device, dtype=y.dtype))
    return y, [x.grad for x in xs], {k: p.grad for k, p in net.named_parameters()}


def autocast_bf16(net):
    """the oracle net under stock PyTorch's CPU bf16 autocast (bf16 convolutions with fp32 accumulation, bf16 stored
    activations and activation gradients; norms in fp32): the CALIBRATION of what bf16 rounding costs on a fixture.  The
    oracle stays the reference's graph; only the arithmetic type of its torch ops changes."""
    orig = net.forward

    def forward(*a):
        with torch.autocast('cpu', dtype=torch.bfloat16):
            y = orig(*a)
        return tuple(t.float() for t in y) if isinstance(y, tuple) else y.float()
    net.forward = forward
    return net


def _param_errors(got, ref):
    """per-tensor relative L2 and the norm-weighted overall error, skipping tensors whose true gradient is 0"""
    errs, num, den = {}, 0.0, 0.0
    for k, gr in ref.items():
        if gr is None or float(gr.abs().max()) < 1e-6:
            continue
        sib = ref.get(k[:-4] + 'weight') if k.endswith('.bias') else None
        if sib is not None and float(gr.abs().max()) <= 1e-3 * float(sib.abs().max()):
            continue                                   # IN-cancelled bias: noise on both sides
        g = got[k]
        errs[k] = _rel_l2(g, gr)
        num += float((g.double().cpu() - gr.double()).norm()) ** 2
        den += float(gr.double().norm()) ** 2
    return errs, (num / den) ** 0.5


@pytest.mark.parametrize('which', ['G', 'D'])
def test_networks_bf16_vs_fp32_oracle(which, dev):
    """the whole generator / discriminator with bf16 activations against the fp32 oracle on fp32 inputs; the error budget is
    what stock PyTorch's own bf16 autocast loses on the same oracle net (x 2 + 1e-2): measured, both land within 10 % of
    each other (G: output 1.1e-2 / 1.3e-2, input gradient 0.240 / 0.244, parameter gradients 4.9e-2 / 5.0e-2 overall --
    InstanceNorm backward is a cancellation, early-layer gradients of this random-weight fixture are ill-conditioned)"""
    import copy
    from pcgan_amd.models import networks
    if which == 'G':
        ref = N.ResnetGeneratorRef(3, 3, 1, 16, 'instance', 9)
        ref.load_state_dict(W.damp_generator_head(W.fill_state_dict(ref.state_dict(), 12)))
        hip = networks.define_G(3, 3, 1, 16, 'resnet_9blocks', norm='instance', init_type='normal')
        x, z = W.seeded_tensor((2, 3, 32, 32), 101), W.seeded_normal((2, 1, 1, 1), 202)
    else:
        ref = N.NLayerDiscriminatorRef(3, 1, 16, 3, 'batch', True)
        ref.load_state_dict(W.fill_state_dict(ref.state_dict(), 20))
        hip = networks.define_D(3, 1, 16, 'n_layers', 3, 'batch', True, 'normal')
        x, z = W.seeded_tensor((4, 3, 64, 64), 103), W.seeded_normal((4, 1, 1, 1), 203)
    hip.load_state_dict({k: v.clone() for k, v in ref.state_dict().items()})
    hip.to(dev)
    sim = autocast_bf16(copy.deepcopy(ref))
    with torch.no_grad():
        shape = tuple(copy.deepcopy(ref)(x, z).shape)
    dy = W.seeded_normal(shape, 303)
    y_ref, din_ref, dp_ref = _run_net(ref, [x, z], dy)
    y_sim, din_sim, dp_sim = _run_net(sim, [x, z], dy)
    y, din, dp = _run_net(hip, [x.to(dev).to(BF), z.to(dev)], dy)
    assert y.dtype == BF and all(g.dtype == torch.float32 for g in dp.values())
    errs, overall = _param_errors(dp, dp_ref)
    errs_s, overall_s = _param_errors(dp_sim, dp_ref)
    worst = max(errs, key=errs.get)
    rows = [('output', _rel_l2(y.float(), y_ref), _rel_l2(y_sim, y_ref)),
            ('input gradient', _rel_l2(din[0].float(), din_ref[0]), _rel_l2(din_sim[0], din_ref[0])),
            ('parameter gradients overall', overall, overall_s)] + [('d' + k, errs[k], errs_s[k]) for k in errs]
    print('bf16 parity %s (relative L2 vs the fp32 oracle: HIP bf16 path / PyTorch CPU bf16 autocast of the oracle): ' % which +
          '; '.join('%s %.3e / %.3e' % r for r in rows[:3]) + '; worst tensor %s %.3e / %.3e' % (worst, errs[worst], errs_s[worst]))
    for name, e_hip, e_sim in rows:
        assert e_hip <= 2 * e_sim + 1e-2, '%s %s: relative L2 %.3e, bf16 autocast of the oracle loses %.3e' % (which, name, e_hip, e_sim)
    # running statistics are fp32 statistics of bf16 tensors
    hb = dict(hip.named_buffers())
    for k, b in ref.named_buffers():
        if 'running' in k:
            assert_close(hb[k], b, 2e-2, which + ' buffer ' + k, atol=1e-3)


def test_step_bf16_vs_fp32_oracle(tmp_path, dev):
    """one full optimize_parameters() under `--dtype bf16` (through the option parser) against the fp32 oracle step from
    the same weights: losses, images, ratings, every G / D gradient; budget = what the oracle step loses with its four nets
    under PyTorch's CPU bf16 autocast (x 2 + 2e-2); fp32 master weights and Adam state types"""
    from test_gpu_step import build_hip_model, _grab_grads
    from test_oracle_golden import build_oracle_step, oracle_set_input
    from oracle.make_golden import step_batch
    model, opt = build_hip_model('default', tmp_path, ['--dtype', 'bf16'])
    assert model.act_dtype == BF
    oracle = build_oracle_step('default')
    sim = build_oracle_step('default')
    for net in (sim.netG, sim.netD, sim.netE, sim.netIP):
        autocast_bf16(net)
    grabbed = _grab_grads(model)
    for o in (oracle, sim):
        oracle_set_input(o, 'default', 0)
        o.optimize_parameters()
    model.set_input(step_batch('default', 0))
    model.optimize_parameters()
    assert model.real_A.dtype == BF and model.fake_B.dtype == BF and model.y_B.dtype == torch.float32
    got, want, lsim = model.get_current_losses(), oracle.losses(), sim.losses()
    report = ['losses (hip / autocast / fp32) ' + ', '.join('%s %.5f/%.5f/%.5f' % (k, got[k], lsim[k], v) for k, v in want.items())]
    for k, v in want.items():
        assert abs(got[k] - v) <= 2 * abs(lsim[k] - v) + 2e-2 * abs(v) + 1e-3, \
            'bf16 step loss %s: %.6g vs fp32 oracle %.6g (bf16 autocast of the oracle: %.6g)' % (k, got[k], v, lsim[k])
    for k in ('fake_B', 'rec_A', 'y_A', 'y_B'):
        e_hip, e_sim = _rel_l2(getattr(model, k).float(), getattr(oracle, k)), _rel_l2(getattr(sim, k), getattr(oracle, k))
        report.append('%s %.3e / %.3e' % (k, e_hip, e_sim))
        assert e_hip <= 2 * e_sim + 2e-2, 'bf16 step %s: relative L2 %.3e (autocast %.3e)' % (k, e_hip, e_sim)
    for tag, ograds, sgrads in (('G', oracle.grads_G, sim.grads_G), ('D', oracle.grads_D, sim.grads_D)):
        hg = dict(grabbed[tag])
        og, sg = dict(ograds), dict(sgrads)
        if tag == 'G':      # the rating channel's filter slice has a true gradient of 0
            hg['model.1.weight'], og['model.1.weight'], sg['model.1.weight'] = (t[:, :-1] for t in (hg['model.1.weight'], og['model.1.weight'], sg['model.1.weight']))
        assert all(g.dtype == torch.float32 for g in hg.values()), 'parameter gradients stay fp32'
        errs, overall = _param_errors(hg, og)
        errs_s, overall_s = _param_errors(sg, og)
        worst = max(errs, key=errs.get)
        report.append('grad%s overall %.3e / %.3e, worst %s %.3e / %.3e' % (tag, overall, overall_s, worst, errs[worst], errs_s[worst]))
        assert overall <= 2 * overall_s + 2e-2, 'bf16 step grad%s overall %.3e (autocast %.3e)' % (tag, overall, overall_s)
        for k in errs:
            if errs_s[k] > 0.2:
                # stock bf16 already loses > 20 % of this tensor (the 3-value bias gradient of the generator head: one signed sum
                # over all pixels of rounded values): it is rounding noise on either side, only its magnitude is bounded
                assert errs[k] <= 2.0, 'bf16 step grad%s %s: relative L2 %.3e (autocast %.3e)' % (tag, k, errs[k], errs_s[k])
                continue
            assert errs[k] <= 2 * errs_s[k] + 5e-2, 'bf16 step grad%s %s: relative L2 %.3e (autocast %.3e)' % (tag, k, errs[k], errs_s[k])
    print('bf16 parity step (HIP bf16 path / PyTorch CPU bf16 autocast of the oracle nets, relative L2 vs the fp32 oracle): ' + '; '.join(report))
    for optim in (model.optimizer_G, model.optimizer_D):
        assert optim.flat.dtype == optim.gflat.dtype == optim.exp_avg.dtype == torch.float32
    # a second step runs and stays finite
    model.set_input(step_batch('default', 1))
    model.optimize_parameters()
    assert all(v == v and abs(v) < 1e4 for v in model.get_current_losses().values())


def test_mixed_storage_types_raise(dev):
    from pcgan_amd.hip import ops
    a = torch.zeros(2, 4, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        ops.add(a, a.to(BF))
    with pytest.raises(RuntimeError):
        ops.act_fwd(a.half(), 1)
