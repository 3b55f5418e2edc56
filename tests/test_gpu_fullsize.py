"""Config-2 (BASELINE.json configs[1]) FULL-SIZE checks of the hot kernels and of one whole step.

The oracle cannot restate a bs-32 256-channel convolution stack in seconds, so at full size the kernels are held to
size-independent properties of the convolution (SURVEY.md 8c / tier rule 3):
  * adjoint identities  <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>  (ties the three kernels together),
  * linearity in the input,
  * batch-slice consistency: image n of the bs-32 result equals the bs-8 result holding that image up to fp32
    re-association (small launches are cut along K -- split-K -- so the summation order may differ; 1e-5),
  * direct parity of ONE image against the oracle's CPU convolution (float64 twin),
and the full optimize_parameters() at full network size (9-block G, ngf 64, 128x128) is compared with the oracle's
CPU step on a batch of 2.

Tolerances: dot products of ~3e7 fp32 terms accumulated in float64 from fp32 results: 2e-5 relative; one-image
parity 2e-5 of the largest magnitude (forward / dgrad), 1e-4 (weight gradient, 1024-term pixel sums x 8 images).
"""
import numpy as np
import pytest
import torch

from oracle import ops_ref as R
from util_cmp import assert_close

pytestmark = pytest.mark.gpu

# the layers that carry the step: name, C, H, K, R, stride, pad, pad_mode, batch
FULL = [
    ('G.res 256->256 3x3 reflect @32', 256, 32, 256, 3, 1, 1, 1, 32),
    ('G.down2 128->256 3x3 s2 @64', 128, 64, 256, 3, 2, 1, 0, 32),
    ('D.c3 256->512 4x4 @16', 256, 16, 512, 4, 1, 1, 0, 32),
    ('G.head 64->3 7x7 reflect @128', 64, 128, 3, 7, 1, 3, 1, 32),
    # the shapes BASELINE configs 4 / 5 (256x256 images, batch 8 / 16 per GPU) put on the same kernels
    ('G.res 256->256 3x3 reflect @64 bs8 (256x256)', 256, 64, 256, 3, 1, 1, 1, 8),
    ('G.down2 128->256 3x3 s2 @128 bs8 (256x256)', 128, 128, 256, 3, 2, 1, 0, 8),
    ('G.down1 64->128 3x3 s2 @256 bs8 (256x256)', 64, 256, 128, 3, 2, 1, 0, 8),
    ('D.c3 256->512 4x4 @32 bs16 (256x256)', 256, 32, 512, 4, 1, 1, 0, 16),
]


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize('case', FULL, ids=[c[0] for c in FULL])
def test_fullsize_conv_properties(case, dev):
    from pcgan_amd.hip import ops
    name, C, H, K, Rk, stride, pad, pm, N = case
    lo, hi = N // 4, N // 2          # the batch slice re-run on its own
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = (torch.rand(N, C, H, H, generator=g) * 2 - 1).to(dev)
    x2 = (torch.rand(N, C, H, H, generator=g) * 2 - 1).to(dev)
    w = (torch.randn(K, C, Rk, Rk, generator=g) * 0.05).to(dev)
    P = (H + 2 * pad - Rk) // stride + 1
    dy = torch.randn(N, K, P, P, generator=g).to(dev)

    y = ops.conv2d_fwd(x, w, None, stride, pad, pm)
    dx = ops.conv2d_bwd_data(dy, w, (H, H), stride, pad, pm)
    dw = ops.conv2d_bwd_weight(x, dy, tuple(w.shape), stride, pad, pm)
    assert torch.isfinite(y).all() and torch.isfinite(dx).all() and torch.isfinite(dw).all()

    # adjoint identities
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(a - b) <= 2e-5 * scale, '%s: <conv x, dy> %.9g vs <x, dgrad dy> %.9g' % (name, a, b)
    assert abs(a - c) <= 2e-5 * scale, '%s: <conv x, dy> %.9g vs <w, wgrad> %.9g' % (name, a, c)

    # linearity in x
    y2 = ops.conv2d_fwd(x2, w, None, stride, pad, pm)
    ylin = ops.conv2d_fwd(0.5 * x - 2.0 * x2, w, None, stride, pad, pm)
    assert_close(ylin, (0.5 * y.double() - 2.0 * y2.double()).cpu(), 2e-5, name + ' linearity')

    # batch-slice consistency (forward, data gradient)
    y8 = ops.conv2d_fwd(x[lo:hi].contiguous(), w, None, stride, pad, pm)
    assert_close(y8, y[lo:hi].double().cpu(), 1e-5, name + ' forward of a batch slice')
    dx8 = ops.conv2d_bwd_data(dy[lo:hi].contiguous(), w, (H, H), stride, pad, pm)
    assert_close(dx8, dx[lo:hi].double().cpu(), 1e-5, name + ' data gradient of a batch slice')

    # one image against the oracle (float64 twin on the CPU)
    xc = x[5:6].double().cpu().requires_grad_(True)
    wc = w.double().cpu()
    yr = R.conv2d(xc, wc, None, stride, pad, pm)
    yr.backward(dy[5:6].double().cpu())
    assert_close(y[5:6], yr.detach(), 2e-5, name + ' image 5 forward vs oracle')
    assert_close(dx[5:6], xc.grad, 2e-5, name + ' image 5 data gradient vs oracle')
    # weight gradient of a quarter of the batch against the oracle
    x8 = x[:lo].double().cpu()
    w8 = w.double().cpu().requires_grad_(True)
    R.conv2d(x8, w8, None, stride, pad, pm).backward(dy[:lo].double().cpu())
    dw8 = ops.conv2d_bwd_weight(x[:lo].contiguous(), dy[:lo].contiguous(), tuple(w.shape), stride, pad, pm)
    assert_close(dw8, w8.grad, 1e-4, name + ' weight gradient (%d images) vs oracle' % lo)
    # ... and the FULL-batch launch itself -- the kernel the benchmark times (the sub-batch above has fewer output pixels than the
    # host's routing threshold and may take another kernel: VERDICT r3) -- per element on a slice of output channels: float64
    # oracle of dw[ks] = wgrad(x, dy[:, ks]) over all N images, every (c, r, s) column
    ks = sorted({0, 1, K // 3, K // 2 + 1, K - 2, K - 1} & set(range(K)))
    wsl = torch.zeros(len(ks), C, Rk, Rk, dtype=torch.float64, requires_grad=True)
    dys = dy[:, ks].double().cpu()
    for n0 in range(0, N, 8):          # in chunks: the float64 unfold of 32 x 256 x 32 x 32 would not fit comfortably at once
        R.conv2d(x[n0:n0 + 8].double().cpu(), wsl, None, stride, pad, pm).backward(dys[n0:n0 + 8])
    assert_close(dw[ks], wsl.grad, 5e-5, name + ' weight gradient of the full batch (production route), channels %s, vs float64' % ks)


def test_fullsize_production_routes_are_the_ones_under_test(dev):
    """the bs-32 residual weight gradient above runs hsplit_wgrad_kernel (the launch bench.py times), not a fallback"""
    from pcgan_amd.hip import ops
    L = ops._L
    assert ops._plan(L.PASS_BWD_WEIGHT, 32, 256, 32, 32, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    assert ops._plan(L.PASS_FWD, 32, 256, 32, 32, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    assert ops._plan(L.PASS_BWD_DATA, 32, 256, 32, 32, 256, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'


@pytest.mark.parametrize('route', ['as_routed', 'split_kernels', 'split_kernels_bf16x6', 'fp32_mfma'])
def test_fullsize_step_vs_oracle(tmp_path, dev, route, monkeypatch):
    """One optimize_parameters() with the FULL networks of config 2 (9-block G ngf 64, 3-layer D ndf 64, ResNet-18 E
    and AlexNet IP at 224) on a batch of 2: losses and fake_B against the oracle's CPU step from the same weights.
    The three routes of the fp32 convolutions (fp16 two-piece = default, three-piece bf16, fp32 MFMA) are held to the SAME oracle numbers.
    'split_kernels': the residual convolutions on the matrix-pipe split kernels they take at the benchmark's batch size (the host
    routes them there from 16384 output pixels; a batch of 2 has 2048) -- with the default fp16 route the operand maxima must come
    from the instance-norm kernels, not from extra passes."""
    import bench
    from oracle import networks_ref as N
    from oracle import step_ref as S
    from pcgan_amd.hip import ops
    if route.startswith('split_kernels'):
        monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    if route == 'split_kernels_bf16x6':      # PCGAN_SPLIT=bf16: the exact three-piece split on the residual convolutions, fp32 MFMA elsewhere
        monkeypatch.setattr(ops, 'HSPLIT', False)
    if route == 'fp32_mfma':                 # PCGAN_BF16X6=0: every convolution on the fp32 MFMA kernels (round 1's route)
        monkeypatch.setattr(ops, 'HSPLIT', False)
        monkeypatch.setattr(ops, 'BF16X6', False)
    amax0 = dict(ops.AMAX_STATS)
    torch.manual_seed(0)
    model, opt = bench.build_model(0, 2, 128, str(tmp_path), seed=3)
    G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
    D = N.NLayerDiscriminatorRef(3, 1, 64, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    IP = N.AlexNetFeatureRef(3, 'None')
    for ref, net in ((G, model.netG), (D, model.netD), (E, model.netE), (IP, model.netIP)):
        ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    oracle = S.WSGANEmbStepRef(G, D, E, IP)
    b = bench.synthetic_batch(2, 128, 0)
    oracle.set_input(b['A'], b['B'], [int(v) for v in b['label']])
    oracle.optimize_parameters()
    from test_gpu_configs_256 import _grab, _check_grads, _check_buffers
    grabbed = _grab(model, 'GD')
    model.set_input(b)
    model.optimize_parameters()
    torch.cuda.synchronize()
    got, want = model.get_current_losses(), oracle.losses()
    for k, v in want.items():
        assert abs(got[k] - v) <= 2e-4 * max(1.0, abs(v)), 'loss %s: hip %.7g oracle %.7g' % (k, got[k], v)
    assert_close(model.fake_B, oracle.fake_B.detach(), 2e-4, 'fake_B (full-size generator)')
    assert_close(model.rec_A, oracle.rec_A.detach(), 2e-4, 'rec_A (full-size generator)')
    # round 4 (VERDICT r3 "weak" 1): every G / D gradient tensor at the moment of its optimizer step and every running statistic
    # after the step, as the 256x256 tests do (loose band: the oracle differentiates on its own ReLU decisions, see _check_grads)
    _check_grads('G', grabbed['G'], oracle.grads_G)
    _check_grads('D', grabbed['D'], oracle.grads_D)
    for tag, hn, on in (('G', model.netG, G), ('D', model.netD, D), ('E', model.netE, E)):
        _check_buffers(tag, hn, on)
    if route == 'split_kernels' and ops.HSPLIT:
        # the norm kernels hand the operand maxima over (18 residual convolutions x 2 generator passes x (forward, data gradient,
        # 2 operands of the weight gradient) alone are 216); only tensors written by a convolution epilogue or a max-pooling
        # (the PatchGAN's second layer, AlexNet, the encoder's first block) still cost an absmax pass -- and, since the 3- / 4-channel
        # layers run on the same route (csrc/thin_conv.hip), the images themselves: generator input x 2, the encoder's three inputs,
        # the PatchGAN's four, the head's dy (small tensors: 6-19 MB)
        attached, computed = ops.AMAX_STATS['attached'] - amax0['attached'], ops.AMAX_STATS['computed'] - amax0['computed']
        assert attached >= 216 and computed <= 34, ops.AMAX_STATS


def test_fullsize_step_bf16_vs_fp32_oracle(tmp_path, dev, monkeypatch):
    """BASELINE configs[2] per GPU at FULL network size: one optimize_parameters() under --dtype bf16 (9-block G ngf 64, 3-layer D,
    ResNet-18 E and AlexNet IP at 224; residual convolutions on the window / split kernels the bs-32 launch takes) against the fp32
    oracle step from the same weights.  Budget, as in tests/test_gpu_bf16.py: what the SAME oracle step loses with its four nets
    under stock PyTorch's CPU bf16 autocast (x 2 + 2e-2) -- losses, images, ratings, every G / D gradient tensor."""
    import bench
    from oracle import networks_ref as N
    from oracle import step_ref as S
    from pcgan_amd.hip import ops
    from test_gpu_bf16 import autocast_bf16, _rel_l2, _param_errors
    from test_gpu_configs_256 import _grab
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    torch.manual_seed(0)
    model, opt = bench.build_model(0, 2, 128, str(tmp_path), seed=3, dtype='bf16')
    assert model.act_dtype == torch.bfloat16

    def build():
        G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
        D = N.NLayerDiscriminatorRef(3, 1, 64, 3, 'batch', True)
        E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
        IP = N.AlexNetFeatureRef(3, 'None')
        for ref, net in ((G, model.netG), (D, model.netD), (E, model.netE), (IP, model.netIP)):
            ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
        return G, D, E, IP
    oracle = S.WSGANEmbStepRef(*build())
    sim = S.WSGANEmbStepRef(*[autocast_bf16(n) for n in build()])
    b = bench.synthetic_batch(2, 128, 0)
    for o in (oracle, sim):
        o.set_input(b['A'], b['B'], [int(v) for v in b['label']])
        o.optimize_parameters()
    grabbed = _grab(model, 'GD')
    model.set_input(b)
    model.optimize_parameters()
    torch.cuda.synchronize()
    assert model.fake_B.dtype == torch.bfloat16 and model.y_B.dtype == torch.float32
    got, want, lsim = model.get_current_losses(), oracle.losses(), sim.losses()
    report = ['losses (hip / autocast / fp32) ' + ', '.join('%s %.5f/%.5f/%.5f' % (k, got[k], lsim[k], v) for k, v in want.items())]
    for k, v in want.items():
        assert abs(got[k] - v) <= 2 * abs(lsim[k] - v) + 2e-2 * abs(v) + 1e-3, \
            'bf16 full-size loss %s: %.6g vs fp32 oracle %.6g (autocast %.6g)' % (k, got[k], v, lsim[k])
    for k in ('fake_B', 'rec_A', 'y_A', 'y_B'):
        e_hip, e_sim = _rel_l2(getattr(model, k).float(), getattr(oracle, k)), _rel_l2(getattr(sim, k), getattr(oracle, k))
        report.append('%s %.3e / %.3e' % (k, e_hip, e_sim))
        assert e_hip <= 2 * e_sim + 2e-2, 'bf16 full-size %s: relative L2 %.3e (autocast %.3e)' % (k, e_hip, e_sim)
    for tag, ograds, sgrads in (('G', oracle.grads_G, sim.grads_G), ('D', oracle.grads_D, sim.grads_D)):
        hg = dict(grabbed[tag])
        og = {k: v for k, v in ograds.items() if v is not None}
        sg = {k: v for k, v in sgrads.items() if v is not None}
        if tag == 'G':
            hg['model.1.weight'], og['model.1.weight'], sg['model.1.weight'] = (t[:, :-1] for t in (hg['model.1.weight'], og['model.1.weight'], sg['model.1.weight']))
        assert all(g.dtype == torch.float32 for g in hg.values())
        errs, overall = _param_errors(hg, og)
        errs_s, overall_s = _param_errors(sg, og)
        worst = max(errs, key=errs.get)
        report.append('grad%s overall %.3e / %.3e, worst %s %.3e / %.3e' % (tag, overall, overall_s, worst, errs[worst], errs_s[worst]))
        assert overall <= 2 * overall_s + 2e-2, 'bf16 full-size grad%s overall %.3e (autocast %.3e)' % (tag, overall, overall_s)
        for k in errs:
            if errs_s[k] > 0.2:
                assert errs[k] <= 2.0, 'bf16 full-size grad%s %s: relative L2 %.3e (autocast %.3e)' % (tag, k, errs[k], errs_s[k])
                continue
            assert errs[k] <= 2 * errs_s[k] + 5e-2, 'bf16 full-size grad%s %s: relative L2 %.3e (autocast %.3e)' % (tag, k, errs[k], errs_s[k])
    print('bf16 full-size step (HIP bf16 / CPU bf16 autocast of the oracle, relative L2 vs the fp32 oracle): ' + '; '.join(report))


def test_partial_last_batch_between_full_batches(tmp_path, dev):
    """ADVICE r3 (high), at the level of the model: an epoch ends with a partial batch (the loader has no drop_last, as in the reference),
    so BatchNorm layers see N = 8, 5, 8 through ONE set of arrival tickets (the frozen encoder's 112^2 / 56^2 maps run the one-launch
    statistics kernels at these sizes).  Round 3's never-cleared tickets picked a wrong last arriver from the second call on -- silently
    wrong batch statistics for the rest of the run.  Three optimize_parameters() with the full networks against the oracle's CPU steps:
    the encoder's ratings (frozen weights, train-mode BatchNorm: a function of the batch alone) and the image losses at every step."""
    import bench
    from oracle import networks_ref as N
    from oracle import step_ref as S
    torch.manual_seed(0)
    model, opt = bench.build_model(0, 8, 128, str(tmp_path), seed=3)
    G = N.ResnetGeneratorRef(3, 3, 1, 64, 'instance', 9)
    D = N.NLayerDiscriminatorRef(3, 1, 64, 3, 'batch', True)
    E = N.SiameseFeatureRef(N.ResNetFeatureRef('resnet18'), 'avg', (32, 1), 1, 0.7, False)
    IP = N.AlexNetFeatureRef(3, 'None')
    for ref, net in ((G, model.netG), (D, model.netD), (E, model.netE), (IP, model.netIP)):
        ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    oracle = S.WSGANEmbStepRef(G, D, E, IP)
    for it, n in enumerate((8, 5, 8)):
        b = bench.synthetic_batch(n, 128, 7, it)
        oracle.set_input(b['A'], b['B'], [int(v) for v in b['label']])
        oracle.optimize_parameters()
        model.set_input(b)
        model.optimize_parameters()
        torch.cuda.synchronize()
        for k in ('y_A', 'y_B'):
            assert_close(getattr(model, k), getattr(oracle, k).detach(), 2e-4, 'step %d (batch %d) %s' % (it, n, k), atol=1e-5)
        got, want = model.get_current_losses(), oracle.losses()
        for k in ('G_cycle', 'G_IP', 'D_real_right'):      # (unaligned steps: Adam's O(lr) drift of the two sides is inside this band)
            assert abs(got[k] - want[k]) <= 1e-2 * max(1.0, abs(want[k])), 'step %d (batch %d) loss %s: hip %.6g oracle %.6g' % (it, n, k, got[k], want[k])
    # (the encoder's running statistics also take in fake_B, which drifts apart by O(lr) per unaligned step: not compared; its batch
    # counter is -- three passes per step, each counted once by the last arriver of channel 0)
    assert int(model.netE.state_dict()['base.model.bn1.num_batches_tracked']) == int(E.state_dict()['base.model.bn1.num_batches_tracked']) == 9
