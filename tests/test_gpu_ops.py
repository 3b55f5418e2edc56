"""GPU parity tests of the raw C-ABI kernels against the oracle's per-op restatement
(oracle/ops_ref.py), judged against its float64 twin.

Tolerances (relative to the largest reference magnitude, SURVEY.md 8c basis):
  forward / data-gradient of one op       2e-5   (fp32 fma chain over K <= 4608 terms)
  weight-gradient (reduction over pixels) 1e-4   (up to 5e5-term fp32 sums)
  index outputs (argmax)                  exact
"""
import pytest
import torch

from oracle import ops_ref as R
from util_cmp import assert_close

pytestmark = pytest.mark.gpu

# (name, N, C, H, W, K, R, stride, pad, pad_mode)
CONV_CASES = [
    ('res3x3_reflect_small', 2, 16, 8, 8, 16, 3, 1, 1, 1),
    ('res3x3_reflect_256', 2, 256, 32, 32, 256, 3, 1, 1, 1),
    ('stem7x7_reflect_c4', 2, 4, 16, 16, 8, 7, 1, 3, 1),
    ('stem7x7_reflect_c4_k64', 1, 4, 32, 32, 64, 7, 1, 3, 1),
    ('head7x7_reflect_k3', 2, 16, 16, 16, 3, 7, 1, 3, 1),
    ('down3x3_s2', 2, 8, 16, 16, 16, 3, 2, 1, 0),
    ('down3x3_s2_64_128', 2, 64, 32, 32, 128, 3, 2, 1, 0),
    ('d4x4_s2_c4', 2, 4, 16, 16, 8, 4, 2, 1, 0),
    ('d4x4_s2_64_128', 2, 64, 16, 16, 128, 4, 2, 1, 0),
    ('d4x4_s1_odd', 2, 16, 16, 16, 32, 4, 1, 1, 0),
    ('d4x4_s1_k1', 3, 32, 15, 15, 1, 4, 1, 1, 0),
    ('e7x7_s2_c3', 2, 3, 20, 20, 16, 7, 2, 3, 0),
    ('ip11x11_s4_c3', 2, 3, 35, 35, 8, 11, 4, 2, 0),
    ('ip5x5_p2', 2, 8, 13, 13, 24, 5, 1, 2, 0),
    ('e1x1_s2', 2, 16, 14, 14, 32, 1, 2, 0, 0),
    ('e3x3_7x7plane', 3, 32, 7, 7, 48, 3, 1, 1, 0),
    ('e3x3_k1', 3, 32, 7, 7, 1, 3, 1, 1, 0),
    ('e3x3_s2_odd', 2, 16, 14, 14, 24, 3, 2, 1, 0),
    ('c192', 1, 192, 13, 13, 96, 3, 1, 1, 0),
    # small-M paths (<= 4 image channels on the gradient side / <= 4 output channels) at the real channel counts
    ('e7x7_s2_c3_k64', 2, 3, 36, 36, 64, 7, 2, 3, 0),
    ('ip11x11_s4_c3_k64', 2, 3, 51, 51, 64, 11, 4, 2, 0),
    ('stem7x7_reflect_c4_k64_b', 2, 4, 24, 24, 64, 7, 1, 3, 1),
    ('d4x4_s2_c4_k64', 2, 4, 32, 32, 64, 4, 2, 1, 0),
    ('head7x7_reflect_64_3', 2, 64, 24, 24, 3, 7, 1, 3, 1),
    ('d4x4_s1_512_1', 2, 512, 15, 15, 1, 4, 1, 1, 0),
    ('e3x3_32_1', 3, 32, 7, 7, 1, 3, 1, 1, 0),
    # tiny feature maps of the encoder's late stages (fewer pixels than one tile, several stride phases)
    ('l3a_3x3_s2_128_256_8', 3, 128, 8, 8, 256, 3, 2, 1, 0),
    ('ds1x1_s2_128_256_8', 3, 128, 8, 8, 256, 1, 2, 0, 0),
    ('l3_3x3_256_4', 3, 256, 4, 4, 256, 3, 1, 1, 0),
    ('l4a_3x3_s2_256_512_4', 3, 256, 4, 4, 512, 3, 2, 1, 0),
    ('l4_3x3_512_2', 3, 512, 2, 2, 512, 3, 1, 1, 0),
    # chunked-K kernels (channels % 16 == 0): reflect data gradient with row-folded weights (3x3 pad 1) at the smallest
    # legal size, non-square, ragged pixel tiles; fused mirror gathers without the fold (5x5 pad 2); odd strided sizes
    ('refl3x3_min_4x4', 2, 16, 4, 4, 32, 3, 1, 1, 1),
    ('refl3x3_6x10', 3, 32, 6, 10, 16, 3, 1, 1, 1),
    ('refl3x3_17x9_k48', 2, 48, 17, 9, 80, 3, 1, 1, 1),
    ('refl5x5_p2_12', 2, 16, 12, 12, 32, 5, 1, 2, 1),
    ('z3x3_s2_odd_15', 3, 32, 15, 15, 48, 3, 2, 1, 0),
    ('z4x4_s2_odd_13', 2, 16, 13, 13, 32, 4, 2, 1, 0),
    ('z5x5_p2_64_27', 2, 64, 27, 27, 32, 5, 1, 2, 0),
    ('z1x1_s1_64', 2, 64, 9, 9, 128, 1, 1, 0, 0),
    # weight-gradient tile variants: 64 channels (two taps per K tile), 128-multiple channels, pixel counts not % 4
    ('w64_7x7map', 3, 64, 7, 7, 64, 3, 1, 1, 0),
    ('w128_15', 2, 128, 15, 15, 32, 3, 1, 1, 0),
    ('w384_13', 1, 384, 13, 13, 64, 3, 1, 1, 0),
    # small-M strip kernel: heights not a multiple of the 8-pixel strip, 4 and 1 output channels, 5 row taps, stride-2 phases
    ('strip_head_k4_19', 2, 32, 19, 19, 4, 7, 1, 3, 1),
    ('strip_k1_5x5_21', 2, 16, 21, 21, 1, 5, 1, 2, 0),
    ('strip_dgrad_c3_7x7s2_41', 2, 3, 41, 41, 32, 7, 2, 3, 0),
    ('strip_dgrad_c4_3x3_18', 2, 4, 18, 18, 16, 3, 1, 1, 0),
]


def _conv_ref(x, w, b, stride, pad, pad_mode, dy):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y = R.conv2d(x64, w64, b64, stride, pad, pad_mode)
    y.backward(dy.double())
    return y.detach(), x64.grad, w64.grad, b64.grad


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_fwd_bwd(case, dev):
    from pcgan_amd.hip import ops
    name, N, C, H, W, K, Rk, stride, pad, pm = case
    g = torch.Generator().manual_seed(sum(map(ord, name)))
    x = torch.rand(N, C, H, W, generator=g) * 2 - 1
    w = torch.randn(K, C, Rk, Rk, generator=g) * 0.1
    b = torch.randn(K, generator=g) * 0.1
    P = (H + 2 * pad - Rk) // stride + 1
    Q = (W + 2 * pad - Rk) // stride + 1
    dy = torch.randn(N, K, P, Q, generator=g)
    y_ref, dx_ref, dw_ref, db_ref = _conv_ref(x, w, b, stride, pad, pm, dy)

    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    y = ops.conv2d_fwd(xd, wd, bd, stride, pad, pm)
    assert_close(y, y_ref, 2e-5, name + ' fwd')
    dx = ops.conv2d_bwd_data(dyd, wd, (H, W), stride, pad, pm)
    assert_close(dx, dx_ref, 2e-5, name + ' bwd_data')
    dw = ops.conv2d_bwd_weight(xd, dyd, tuple(w.shape), stride, pad, pm)
    assert_close(dw, dw_ref, 1e-4, name + ' bwd_weight')
    db = ops.channel_sum(dyd)
    assert_close(db, db_ref, 1e-4, name + ' bias grad')
    # prepacked-weight entry points: same kernels on the same operand -> bit-identical; a second batch size
    # reuses the packed copy (it does not depend on N)
    cache = {}
    # packed calls may take another kernel of the same accuracy (matrix-pipe split kernels for eligible shapes): other low bits
    other = ops.BF16X6 or (ops.HSPLIT and ops.HGEMM)
    yp = ops.conv2d_fwd(xd, wd, bd, stride, pad, pm, pack_cache=cache)
    dxp = ops.conv2d_bwd_data(dyd, wd, (H, W), stride, pad, pm, pack_cache=cache)
    if other:
        assert_close(yp, y_ref, 2e-5, name + ' fwd_packed')
        assert_close(dxp, dx_ref, 2e-5, name + ' bwd_packed')
    else:
        assert torch.equal(yp, y), name + ' fwd_packed'
        assert torch.equal(dxp, dx), name + ' bwd_packed'
    assert len([k for k in cache if k != 'wamax']) == 2
    stamps = {k: v[0] for k, v in cache.items()}
    # a second batch size reuses the packed copy (it does not depend on N); same kernel on the same operands -> same bits
    if N > 1:
        y1 = ops.conv2d_fwd(xd[:1].contiguous(), wd, bd, stride, pad, pm, pack_cache=cache)
        dx1 = ops.conv2d_bwd_data(dyd[:1].contiguous(), wd, (H, W), stride, pad, pm, pack_cache=cache)
        if other:     # (the fp16 route scales by the largest magnitude of the tensor it is given: other bits for a sub-batch)
            assert_close(y1, y_ref[:1], 2e-5, name + ' fwd_packed, one image')
            assert_close(dx1, dx_ref[:1], 2e-5, name + ' bwd_packed, one image')
        else:
            assert torch.equal(y1, yp[:1]) and torch.equal(dx1, dxp[:1])
    assert {k: v[0] for k, v in cache.items()} == stamps


def test_packed_weight_cache_invalidation(dev):
    """The packed copy must follow the weights: torch in-place writes (version counter), the raw-pointer Adam
    kernels (global epoch) and storage moves all re-pack."""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(2, 32, 12, 12, generator=g) * 2 - 1).to(dev)
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.1).to(dev)
    cache = {}
    y0 = ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cache)
    w.mul_(2.0)                                              # version bump
    assert_close(ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cache), 2 * y0.double().cpu(), 2e-5, 'after mul_')
    grad = torch.ones_like(w)
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    w_before = w.clone()
    ops.adam_step(w, grad, m, v, 0.05, 0.5, 0.999, 1e-8, 1)  # raw-pointer update, no version bump
    assert not torch.equal(w, w_before)
    y_new = ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cache)
    assert torch.equal(y_new, ops.conv2d_fwd(x, w, None, 1, 1, 1)), 'stale packed weights after adam_step'
    dx_new = ops.conv2d_bwd_data(y0, w, (12, 12), 1, 1, 1, pack_cache=cache)
    ops.adam_step(w, grad, m, v, 0.05, 0.5, 0.999, 1e-8, 2)
    assert torch.equal(ops.conv2d_bwd_data(y0, w, (12, 12), 1, 1, 1, pack_cache=cache),
                       ops.conv2d_bwd_data(y0, w, (12, 12), 1, 1, 1)), 'stale packed weights (bwd) after adam_step'
    assert not torch.equal(dx_new, ops.conv2d_bwd_data(y0, w, (12, 12), 1, 1, 1))


@pytest.mark.parametrize('act,slope', [(1, 0.0), (2, 0.2), (3, 0.0), (4, 0.0)])
def test_conv2d_fused_activation(act, slope, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 8, 12, 12, generator=g) * 2 - 1
    w = torch.randn(16, 8, 3, 3, generator=g) * 0.2
    b = torch.randn(16, generator=g) * 0.1
    ref = R.activation(R.conv2d(x.double(), w.double(), b.double(), 1, 1, 0), act, slope)
    y = ops.conv2d_fwd(x.to(dev), w.to(dev), b.to(dev), 1, 1, 0, act, slope)
    assert_close(y, ref, 2e-5, 'conv+act %d' % act)
    dy = torch.randn(ref.shape, generator=g)
    y64 = ref.clone().requires_grad_(False)
    pre = R.conv2d(x.double(), w.double(), b.double(), 1, 1, 0).requires_grad_(True)
    R.activation(pre, act, slope).backward(dy.double())
    dpre = ops.act_bwd(dy.to(dev), y, act, slope)
    assert_close(dpre, pre.grad, 5e-5, 'act_bwd %d' % act)


@pytest.mark.parametrize('cin,cout,hw', [(16, 8, 8), (256, 128, 32), (128, 64, 16)])
def test_conv_transpose_via_conv_entry_points(cin, cout, hw, dev):
    """nn.ConvTranspose2d(k3,s2,p1,op1) = bwd_data of the conv (C=cout -> K=cin)."""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(11)
    N = 2
    x = torch.rand(N, cin, hw, hw, generator=g) * 2 - 1
    w = torch.randn(cin, cout, 3, 3, generator=g) * 0.1
    b = torch.randn(cout, generator=g) * 0.1
    x64, w64, b64 = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    y_ref = R.conv_transpose2d(x64, w64, b64)
    dy = torch.randn(y_ref.shape, generator=g)
    y_ref.backward(dy.double())
    xd, wd, bd, dyd = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)
    y = ops.conv2d_bwd_data(xd, wd, (2 * hw, 2 * hw), 2, 1, 0, bias=bd)
    assert_close(y, y_ref, 2e-5, 'convT fwd')
    dx = ops.conv2d_fwd(dyd, wd, None, 2, 1, 0)
    assert_close(dx, x64.grad, 2e-5, 'convT dgrad')
    dw = ops.conv2d_bwd_weight(dyd, xd, tuple(w.shape), 2, 1, 0)
    assert_close(dw, w64.grad, 1e-4, 'convT wgrad')


@pytest.mark.parametrize('shape', [(2, 8, 32, 32), (3, 5, 7, 7), (2, 4, 128, 128), (2, 6, 15, 15)])
@pytest.mark.parametrize('act,slope,res', [(0, 0.0, False), (1, 0.0, False), (0, 0.0, True)])
def test_instance_norm(shape, act, slope, res, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(3)
    N, C, H, W = shape
    x = torch.randn(shape, generator=g) * 1.5 + 0.3
    r = torch.randn(shape, generator=g) if res else None
    rm, rv = torch.zeros(C), torch.ones(C)
    rm64, rv64 = rm.double(), rv.double()
    x64 = x.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if res else None
    y64 = R.instance_norm(x64, rm64, rv64)
    if res:
        y64 = y64 + r64
    y64 = R.activation(y64, act, slope)
    dy = torch.randn(shape, generator=g)
    y64.backward(dy.double())

    xd = x.to(dev)
    rd = r.to(dev) if res else None
    rmd, rvd = rm.to(dev), rv.to(dev)
    mean, m2 = ops.plane_stats(xd)
    ops.in_running_update(mean, m2, rmd, rvd, N, C, H * W, 0.1)
    y = ops.norm_act_fwd(xd, mean, m2, None, None, rd, True, 1e-5, act, slope)
    assert_close(y, y64, 2e-5, 'instnorm fwd')
    assert_close(rmd, rm64, 2e-5, 'running_mean')
    assert_close(rvd, rv64, 2e-5, 'running_var')
    dyd = dy.to(dev)
    s1, s2 = ops.norm_bwd_stats(dyd, xd, y, mean, m2, True, 1e-5, act, slope)
    dx, dres = ops.norm_bwd_apply(dyd, xd, y, mean, m2, None, s1, s2, True, 1e-5, act, slope, res)
    assert_close(dx, x64.grad, 5e-5, 'instnorm bwd')
    if res:
        assert_close(dres, r64.grad, 1e-6, 'instnorm residual grad')


@pytest.mark.parametrize('shape', [(2, 8, 32, 32), (3, 5, 7, 7), (2, 4, 128, 128), (1, 2, 256, 256), (2, 6, 15, 15),
                                   (2, 3, 64, 64), (2, 3, 8, 8), (2, 16, 32, 32), (1, 32, 12, 20),
                                   # >= 1024 small planes: the wave-per-plane kernels (a plane count that is no multiple of 4, a ragged last float4 row)
                                   (8, 128, 32, 32), (13, 79, 8, 8), (4, 256, 12, 20), (4, 272, 16, 16)])
@pytest.mark.parametrize('act,res', [(0, False), (1, False), (0, True)])
def test_instance_norm_fused(shape, act, res, dev):
    """register-resident single-pass kernels (and their two-pass fallback for planes that are not a multiple of 4)"""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(13)
    x = torch.randn(shape, generator=g) * 1.5 + 0.3
    r = torch.randn(shape, generator=g) if res else None
    x64 = x.double().requires_grad_(True)
    y64 = R.instance_norm(x64)
    if res:
        y64 = y64 + r.double()
    y64 = R.activation(y64, act)
    dy = torch.randn(shape, generator=g)
    y64.backward(dy.double())
    xd = x.to(dev)
    y, mean, m2 = ops.instnorm_fwd(xd, r.to(dev) if res else None, 1e-5, act, 0.0)
    assert_close(y, y64, 2e-5, 'fused instnorm fwd')
    mean_ref, m2_ref = ops.plane_stats(xd)
    assert_close(mean, mean_ref, 1e-5, 'fused mean', atol=1e-6)
    assert_close(m2, m2_ref, 1e-5, 'fused M2')
    dx = ops.instnorm_bwd(dy.to(dev), xd, y, mean, m2, 1e-5, act, 0.0)
    assert_close(dx, x64.grad, 5e-5, 'fused instnorm bwd')
    # channel counts a matrix-pipe convolution can gather (multiples of 16): both kernels also hand out the largest magnitude of every output plane --
    # exactly the values stored -- for the fp16 route of the convolution that consumes the tensor (ops.amax_of)
    for t in (y, dx):
        ent = t.__dict__.get('_pcgan_amax')
        assert (ent is not None) == (ops.HSPLIT and shape[1] % 16 == 0 and (shape[2] * shape[3]) % 4 == 0)
        if ent is not None:
            assert torch.equal(ent[1], t.abs().amax(dim=(2, 3)).reshape(-1)), 'plane maxima'
            before = dict(ops.AMAX_STATS)
            assert ops.amax_of(t) is ent[1] and ops.AMAX_STATS['attached'] == before['attached'] + 1
            t.add_(1.0)        # an in-place change outdates them: one absmax pass instead
            assert float(ops.amax_of(t).max()) == float(t.abs().max()) and ops.AMAX_STATS['computed'] == before['computed'] + 1
            t.sub_(1.0)
    dx = ops.instnorm_bwd(dy.to(dev), xd, y, mean, m2, 1e-5, act, 0.0)
    # the register-resident backward also hands out the per-plane sums of dx; channel_sum (the bias gradient of the
    # convolution in front of the norm) finishes from them instead of re-reading dx -- same value as the full pass
    HW = shape[2] * shape[3]
    psum = getattr(dx, '_pcgan_plane_sums', None)
    assert (psum is not None) == (HW % 4 == 0)
    full = dx.double().sum(dim=(2, 3)).reshape(-1)
    scale = float(dx.double().abs().sum(dim=(2, 3)).max())
    if psum is not None:
        assert float((psum.double() - full).abs().max()) <= 1e-6 * scale + 1e-7, 'plane sums of dx'
        before = dict(ops.PLANE_SUM_STATS)
        acc = torch.ones(shape[1], device=dev)
        ops.channel_sum(dx, accumulate_into=acc)
        assert ops.PLANE_SUM_STATS['fused'] == before['fused'] + 1 and ops.PLANE_SUM_STATS['full'] == before['full']
        want = 1.0 + dx.double().sum(dim=(0, 2, 3))
        assert float((acc.double() - want).abs().max()) <= 1e-6 * scale * shape[0] + 1e-6
        plain = ops.channel_sum(dx.clone())
        assert ops.PLANE_SUM_STATS['full'] == before['full'] + 1
        assert float((plain.double() - dx.double().sum(dim=(0, 2, 3))).abs().max()) <= 1e-6 * scale * shape[0] + 1e-6


@pytest.mark.parametrize('shape', [(4, 8, 16, 16), (3, 5, 7, 7), (2, 16, 64, 64), (40, 3, 4, 4), (70, 5, 2, 2)])
@pytest.mark.parametrize('act,slope,res', [(2, 0.2, False), (1, 0.0, True), (2, 0.7, False), (0, 0.0, False)])
def test_batch_norm_train(shape, act, slope, res, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(4)
    N, C, H, W = shape
    x = torch.randn(shape, generator=g) * 2.0 - 0.5
    gamma = torch.randn(C, generator=g) * 0.2 + 1.0
    beta = torch.randn(C, generator=g) * 0.1
    r = torch.randn(shape, generator=g) if res else None
    rm64, rv64 = torch.zeros(C).double(), torch.ones(C).double()
    x64 = x.double().requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if res else None
    y64 = R.batch_norm(x64, g64, b64, rm64, rv64)
    if res:
        y64 = y64 + r64
    y64 = R.activation(y64, act, slope)
    dy = torch.randn(shape, generator=g)
    y64.backward(dy.double())

    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    rd = r.to(dev) if res else None
    rmd, rvd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    mean_nc, m2_nc = ops.plane_stats(xd)
    mean_c, var_c = ops.bn_merge(mean_nc, m2_nc, N, C, H * W, rmd, rvd, 0.1)
    y = ops.norm_act_fwd(xd, mean_c, var_c, gd, bd, rd, False, 1e-5, act, slope)
    assert_close(y, y64, 2e-5, 'bn fwd')
    assert_close(rmd, rm64, 2e-5, 'bn running_mean')
    assert_close(rvd, rv64, 2e-5, 'bn running_var')
    dyd = dy.to(dev)
    s1n, s2n = ops.norm_bwd_stats(dyd, xd, y, mean_c, var_c, False, 1e-5, act, slope)
    s1, s2 = ops.bn_bwd_reduce(s1n, s2n, N, C)
    dx, dres = ops.norm_bwd_apply(dyd, xd, y, mean_c, var_c, gd, s1, s2, False, 1e-5, act, slope, res)
    assert_close(dx, x64.grad, 5e-5, 'bn dx')
    assert_close(s2, g64.grad, 5e-5, 'bn dgamma')
    assert_close(s1, b64.grad, 5e-5, 'bn dbeta')
    if res:
        assert_close(dres, r64.grad, 1e-6, 'bn residual grad')
    # partial maxima for the fp16 route of the next convolution (multiples of 16 channels): exactly the values stored,
    # per plane from the two-pass kernels, per channel from the one-launch kernels
    def check_maxima(t, per, what):
        ent = t.__dict__.get('_pcgan_amax')
        assert (ent is not None) == (ops.HSPLIT and C % 16 == 0), what
        if ent is not None:
            want = t.abs().amax(dim=(2, 3)).reshape(-1) if per == 'plane' else t.abs().amax(dim=(0, 2, 3))
            assert torch.equal(ent[1], want), what
    check_maxima(y, 'plane', 'norm_act_fwd maxima')
    check_maxima(dx, 'plane', 'norm_bwd_apply maxima')
    if N * H * W > 1:
        yf, _, _ = ops.bn_fwd_fused(xd, gd, bd, rd, None, None, None, 0.1, 1e-5, act, slope)
        assert_close(yf, y64, 2e-5, 'bn fused fwd')
        check_maxima(yf, 'channel', 'bn_fwd_fused maxima')
        dxf, _, _, _ = ops.bn_bwd_fused(dyd, xd, yf, mean_c, var_c, gd, 1e-5, act, slope, True, res)
        assert_close(dxf, x64.grad, 5e-5, 'bn fused dx')
        check_maxima(dxf, 'channel', 'bn_bwd_fused maxima')


@pytest.mark.parametrize('shape,k,stride,pad', [((2, 4, 16, 16), 3, 2, 1), ((2, 3, 55, 55), 3, 2, 0),
                                               ((1, 2, 13, 13), 3, 2, 0), ((2, 3, 112, 112), 3, 2, 1),
                                               ((2, 3, 20, 70), 3, 1, 1), ((1, 2, 17, 33), 2, 3, 0)])      # (strides other than 2: the runtime-stride form)
def test_maxpool(shape, k, stride, pad, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(shape, generator=g)
    x64 = x.double().requires_grad_(True)
    y64 = R.max_pool2d(x64, k, stride, pad)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    y, arg = ops.maxpool_fwd(x.to(dev), k, stride, pad)
    assert torch.equal(y.cpu(), y64.detach().float()), 'maxpool values must be exact'
    _, idx = torch.nn.functional.max_pool2d(x, k, stride, pad, return_indices=True)
    assert torch.equal(arg.cpu().long(), idx), 'argmax indices must be bit-exact'
    dx = ops.maxpool_bwd(dy.to(dev), arg, shape[2:], k, stride, pad)
    assert_close(dx, x64.grad, 1e-6, 'maxpool bwd')


@pytest.mark.parametrize('is_max', [False, True])
def test_global_pool(is_max, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(7)
    x = torch.randn(5, 3, 7, 7, generator=g)
    x64 = x.double().requires_grad_(True)
    y64 = R.global_pool(x64, is_max)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    y, arg = ops.global_pool_fwd(x.to(dev), is_max)
    assert_close(y, y64, 1e-6, 'global pool fwd')
    dx = ops.global_pool_bwd(dy.to(dev), arg, (7, 7), is_max)
    assert_close(dx, x64.grad, 1e-6, 'global pool bwd')


@pytest.mark.parametrize('hin,hout', [(128, 224), (32, 64), (16, 28), (20, 9), (7, 7)])
def test_bilinear_align_corners(hin, hout, dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(8)
    x = torch.rand(2, 3, hin, hin, generator=g) * 2 - 1
    x64 = x.double().requires_grad_(True)
    y64 = torch.nn.functional.interpolate(x64, size=(hout, hout), mode='bilinear', align_corners=True)
    dy = torch.randn(y64.shape, generator=g)
    y64.backward(dy.double())
    y = ops.bilinear_fwd(x.to(dev), (hout, hout))
    assert_close(y, y64, 2e-5, 'bilinear fwd')   # fp32 source-index arithmetic vs fp64 twin
    y32 = torch.nn.functional.interpolate(x, size=(hout, hout), mode='bilinear', align_corners=True)
    assert_close(y, y32, 2e-5, 'bilinear fwd vs fp32 oracle')
    dx = ops.bilinear_bwd(dy.to(dev), (hin, hin))
    assert_close(dx, x64.grad, 5e-5, 'bilinear bwd')


def test_concat_z_and_channel_scale(dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(9)
    img = torch.randn(4, 3, 9, 9, generator=g)
    for zb in (4, 1):
        z = torch.randn(zb, 2, 1, 1, generator=g)
        ref = R.concat_z(img, z)
        out = ops.concat_z(img.to(dev), z.to(dev).contiguous())
        assert torch.equal(out.cpu(), ref)
    mask = (torch.rand(4 * 3, generator=g) > 0.3).float()
    ref = R.dropout2d_with_mask(img.double(), mask.view(4, 3).double(), 0.2)
    out = ops.channel_scale(img.to(dev), mask.to(dev), 1.0 / 0.8)
    assert_close(out, ref, 1e-6, 'dropout2d')


def test_losses(dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(10)
    pred = torch.rand(6, 1, 14, 14, generator=g) * 0.98 + 0.01
    pred[0, 0, 0, 0] = 0.0      # exercises the -100 log clamp
    pred[1, 0, 0, 0] = 1.0
    tgt = torch.tensor([1., 0., 1., 0., 0., 1.])
    p64 = pred.double().requires_grad_(True)
    l64 = R.bce_loss(p64, tgt.double())
    l64.backward()
    loss, grad = ops.bce_loss(pred.to(dev), tgt.to(dev))
    assert_close(loss, l64, 1e-5, 'bce loss')
    # the two clamped elements have a 1e12-scale gradient in torch as well; compare the rest
    gm = grad.cpu().double().clone()
    rm = p64.grad.clone()
    for t in (gm, rm):
        t[0, 0, 0, 0] = 0
        t[1, 0, 0, 0] = 0
    assert_close(gm, rm, 1e-5, 'bce grad')
    a = torch.randn(4, 3, 16, 16, generator=g)
    b = torch.randn(4, 3, 16, 16, generator=g)
    for fn, ref in ((ops.l1_loss, R.l1_loss), (ops.mse_loss, R.mse_loss)):
        a64 = a.double().requires_grad_(True)
        l = ref(a64, b.double())
        l.backward()
        loss, grad = fn(a.to(dev), b.to(dev))
        assert_close(loss, l, 1e-5, 'loss')
        assert_close(grad, a64.grad, 1e-5, 'loss grad')


def test_adam_matches_torch(dev):
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(12)
    p = torch.randn(10007, generator=g)
    grads = torch.randn(10007, generator=g) * 0.01
    ref = R.adam_reference([p], [grads], lr=2e-4, beta1=0.5, steps=3)[0]
    pd, gd = p.to(dev), grads.to(dev)
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    for step in (1, 2, 3):
        ops.adam_step(pd, gd, m, v, 2e-4, 0.5, 0.999, 1e-8, step)
    # identical grads fed to both Adams: agreement to fp32 rounding of the update
    assert (pd.cpu() - ref).abs().max().item() < 2e-7
    # device-state variant
    pd2 = p.to(dev)
    m2, v2 = torch.zeros_like(pd2), torch.zeros_like(pd2)
    lr_dev = torch.tensor([2e-4], device=dev)
    step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(3):
        ops.adam_step_dev(pd2, gd, m2, v2, lr_dev, step_dev, 0.5, 0.999, 1e-8)
    assert int(step_dev.item()) == 3
    assert torch.equal(pd2, pd)


@pytest.mark.parametrize('shape', [(32, 64, 56, 56), (8, 64, 112, 112), (5, 7, 9, 9), (3, 130, 6, 6), (70, 5, 2, 2)])
def test_batch_norm_one_launch_statistics(shape, dev):
    """the one-launch statistics of the large BatchNorm tensors (pcgan_bn_stats_merged / pcgan_bn_bwd_stats_reduced: the last-arriving
    workgroup of a channel merges its N plane results) against the two-launch pairs they stand for (plane_stats + bn_merge,
    norm_bwd_stats + bn_bwd_reduce): BIT-IDENTICAL means / variances / running statistics / gradient sums, the batch counter moves by
    one per call, and the arrival tickets (put back to zero by each call's last arriver) keep working call after call (also beside other
    kernels on a second stream)"""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    tick = torch.zeros(2 * C, dtype=torch.int32, device=dev)
    t_f, t_b = tick[:C], tick[C:]
    rm1, rv1 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    nb = torch.zeros((), dtype=torch.int64, device=dev)
    other = torch.cuda.Stream()
    junk = torch.randn(1 << 22, device=dev)
    for call in range(4):
        x = (torch.randn(shape, generator=g) * (1.0 + call) - 0.3 * call).to(dev)
        dy = torch.randn(shape, generator=g).to(dev)
        mean_nc, m2_nc = ops.plane_stats(x)
        mean_a, var_a = ops.bn_merge(mean_nc, m2_nc, N, C, H * W, rm1, rv1, 0.1)
        if call >= 2:            # company on another stream while the ticketed kernel runs
            other.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(other):
                for _ in range(4):
                    junk.mul_(1.0001)
        mean_b, var_b = ops.bn_stats_merged(x, rm2, rv2, nb, t_f, 0.1)
        torch.cuda.synchronize()
        assert torch.equal(mean_a, mean_b) and torch.equal(var_a, var_b), 'call %d' % call
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2), 'running statistics, call %d' % call
        assert int(nb) == call + 1
        assert int(t_f.min()) == int(t_f.max()) == 0          # the last arriver of every channel reset its ticket
        y = ops.norm_act_fwd(x, mean_a, var_a, None, None, None, False, 1e-5, 2, 0.2)
        s1n, s2n = ops.norm_bwd_stats(dy, x, y, mean_a, var_a, False, 1e-5, 2, 0.2)
        s1a, s2a = ops.bn_bwd_reduce(s1n, s2n, N, C)
        s1b, s2b = ops.bn_bwd_stats_reduced(dy, x, y, mean_a, var_a, 1e-5, 2, 0.2, t_b)
        torch.cuda.synchronize()
        assert torch.equal(s1a, s1b) and torch.equal(s2a, s2b), 'gradient sums, call %d' % call
        assert int(t_b.min()) == int(t_b.max()) == 0


def test_batch_norm_one_launch_statistics_alternating_batch_sizes(dev):
    """ONE ticket array shared by calls with different batch sizes (the partial last batch of an epoch: the loader has no drop_last,
    as in the reference; test() / get_current_visuals() at another N): 32, 20, 32, 1-image-short, 32 -- every call bit-identical to the
    two-launch pair.  Round 3's never-cleared tickets ((old + 1) % N == 0) chose a wrong last arriver from the second call on."""
    from pcgan_amd.hip import ops
    g = torch.Generator().manual_seed(7)
    C, H, W = 24, 20, 20
    tick = torch.zeros(2 * C, dtype=torch.int32, device=dev)
    t_f, t_b = tick[:C], tick[C:]
    rm1, rv1 = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    rm2, rv2 = rm1.clone(), rv1.clone()
    nb = torch.zeros((), dtype=torch.int64, device=dev)
    for call, N in enumerate([32, 20, 32, 31, 7, 32]):
        x = (torch.randn((N, C, H, W), generator=g) * (1.0 + call) - 0.3 * call).to(dev)
        dy = torch.randn((N, C, H, W), generator=g).to(dev)
        mean_nc, m2_nc = ops.plane_stats(x)
        mean_a, var_a = ops.bn_merge(mean_nc, m2_nc, N, C, H * W, rm1, rv1, 0.1)
        mean_b, var_b = ops.bn_stats_merged(x, rm2, rv2, nb, t_f, 0.1)
        torch.cuda.synchronize()
        assert torch.equal(mean_a, mean_b) and torch.equal(var_a, var_b), 'call %d (N = %d)' % (call, N)
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2), 'running statistics, call %d (N = %d)' % (call, N)
        assert int(nb) == call + 1
        assert int(t_f.abs().max()) == 0
        y = ops.norm_act_fwd(x, mean_a, var_a, None, None, None, False, 1e-5, 2, 0.2)
        s1n, s2n = ops.norm_bwd_stats(dy, x, y, mean_a, var_a, False, 1e-5, 2, 0.2)
        s1a, s2a = ops.bn_bwd_reduce(s1n, s2n, N, C)
        s1b, s2b = ops.bn_bwd_stats_reduced(dy, x, y, mean_a, var_a, 1e-5, 2, 0.2, t_b)
        torch.cuda.synchronize()
        assert torch.equal(s1a, s1b) and torch.equal(s2a, s2b), 'gradient sums, call %d (N = %d)' % (call, N)
        assert int(t_b.abs().max()) == 0
