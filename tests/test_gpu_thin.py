"""GPU: the convolutions that gather <= 4 channels on the f16 matrix pipe (csrc/thin_conv.hip, through the C-ABI: the 7x7 stems of
the generator and of the Elo encoder, the first PatchGAN layer, the data gradient of the generator's 64 -> 3 head) against the
oracle's convolution in float64.  Same bar as the other fp16-route kernels (tests/test_gpu_bf16x6.py): relative L2 <= 3e-6 and within
4 x the fp32-MFMA kernel's own error on the same data, per output channel when the filter rows span 2^+-20."""
import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu

# name, N, C, H, W, K, k, stride, pad, pad_mode, dgrad, bias, act
CASES = [
    ('G.stem 4->64 7x7 reflect (ragged 40x44)', 2, 4, 40, 44, 64, 7, 1, 3, 1, False, True, 0),
    ('G.stem at one tile (8x32 outputs)', 1, 4, 8, 32, 64, 7, 1, 3, 1, False, False, 0),
    ('stem with 3 channels, zero padding, 128 outputs', 2, 3, 19, 33, 128, 7, 1, 3, 0, False, True, 1),
    ('E.conv1 3->64 7x7 stride 2 (50 -> 25)', 2, 3, 50, 50, 64, 7, 2, 3, 0, False, False, 0),
    ('E.conv1 at 224 -> 112 (3.5 tiles per row)', 1, 3, 224, 224, 64, 7, 2, 3, 0, False, False, 0),
    ('D.c0 4->64 4x4 stride 2 + bias + LeakyReLU', 3, 4, 36, 36, 64, 4, 2, 1, 0, False, True, 2),
    ('1 channel', 1, 1, 16, 32, 64, 7, 1, 3, 1, False, False, 0),
    ('head data gradient 64->3 7x7 reflect', 2, 64, 40, 44, 3, 7, 1, 3, 1, True, False, 0),
    ('head data gradient at 128 channels, zero padding', 1, 128, 17, 35, 3, 7, 1, 3, 0, True, False, 0),
    ('data gradient with 1 output channel, valid convolution', 2, 64, 20, 38, 1, 7, 1, 0, 0, True, False, 0),
]


def _ref(case, src, w, b):
    name, N, C, H, W, K, k, stride, pad, pm, dgrad, bias, act = case
    if dgrad:
        xz = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
        R.conv2d(xz, w.double(), None, stride, pad, pm).backward(src.double())
        return xz.grad
    ref = R.conv2d(src.double(), w.double(), b.double() if b is not None else None, stride, pad, pm)
    if act == 1:
        ref = ref.relu()
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    return ref


def _run(dev, case, src, w, b):
    from pcgan_amd.hip import ops
    name, N, C, H, W, K, k, stride, pad, pm, dgrad, bias, act = case
    sd, wd = src.to(dev), w.to(dev)
    cache = {}
    if dgrad:
        out = ops.conv2d_bwd_data(sd, wd, (H, W), stride, pad, pm, pack_cache=cache)
    else:
        out = ops.conv2d_fwd(sd, wd, b.to(dev) if b is not None else None, stride, pad, pm, act, 0.2, pack_cache=cache)
    torch.cuda.synchronize()
    return out.double().cpu()


def _data(case, seed, row_magnitudes=False):
    name, N, C, H, W, K, k, stride, pad, pm, dgrad, bias, act = case
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(K, C, k, k, generator=g) * 0.05
    if row_magnitudes:
        rows = C if dgrad else K
        mag = torch.pow(2.0, torch.randint(-20, 21, (rows,), generator=g).float())
        mag[0], mag[1] = 2.0 ** 20, 2.0 ** -20
        w = w * (mag.view(1, C, 1, 1) if dgrad else mag.view(K, 1, 1, 1))
    if dgrad:
        P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        src = torch.randn(N, K, P, Q, generator=g)
    else:
        src = torch.randn(N, C, H, W, generator=g)
    b = torch.randn(K, generator=g) if bias else None
    return src, w, b


@pytest.mark.parametrize('case', CASES, ids=lambda c: c[0])
def test_thin_route_has_fp32_accuracy(dev, monkeypatch, case):
    from pcgan_amd.hip import ops
    src, w, b = _data(case, len(case[0]))
    ref = _ref(case, src, w, b)
    key = ('dgrad' if case[10] else 'fwd', 'thin')
    n0 = ops.ROUTE_STATS.get(key, 0)
    out = _run(dev, case, src, w, b)
    assert ops.ROUTE_STATS.get(key, 0) == n0 + 1, ops.ROUTE_STATS       # the kernel under test is the one that ran
    assert out.shape == ref.shape
    e16 = float((out - ref).norm() / ref.norm())
    monkeypatch.setattr(ops, 'THIN', False)                               # igemm2_kernel<.., 4> / the generic kernels on fp32 MFMA
    e32 = float((_run(dev, case, src, w, b) - ref).norm() / ref.norm())
    assert e16 <= 4 * e32 + 5e-7 and e16 < 3e-6, '%s: %.3e (fp32 MFMA route %.3e)' % (case[0], e16, e32)
    # every element, not only the norm: a wrong border / ragged-tile pixel must not hide in the L2
    assert float((out - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize('case', [CASES[0], CASES[3], CASES[7]], ids=lambda c: c[0])
def test_thin_route_per_channel_scales(dev, case):
    """filter rows spanning 2^-20 .. 2^20: every produced channel keeps its 22 bits (one power-of-two scale per row of the pass's
    weight matrix, applied exactly in the epilogue)"""
    case = case[:11] + (False, 0)          # no bias (it would swamp a 2^-20 row), no activation
    src, w, b = _data(case, 7, row_magnitudes=True)
    ref = _ref(case, src, w, None)
    out = _run(dev, case, src, w, None)
    rows = ref.shape[1]
    d = (out - ref).transpose(0, 1).reshape(rows, -1).norm(dim=1)
    n = ref.transpose(0, 1).reshape(rows, -1).norm(dim=1)
    e = d / n
    assert float(e.max()) < 3e-6, (float(e.max()), int(e.argmax()))


def test_thin_route_operand_range(dev):
    """inputs scaled by 1e-30 and 1e12 (the tensor's power-of-two scale puts them into fp16's range) and spanning ten decades"""
    from pcgan_amd.hip import ops
    case = CASES[0]
    src, w, b = _data(case, 3)
    for scale in (1e-30, 1e12):
        ref = _ref(case, src * scale, w, None)
        out = _run(dev, case[:11] + (False, 0), src * scale, w, None)
        assert float((out - ref).norm() / ref.norm()) < 3e-6, scale
    g = torch.Generator().manual_seed(11)
    wide = src * torch.pow(10.0, torch.rand(src.shape, generator=g) * 10 - 5)
    ref = _ref(case, wide, w, None)
    out = _run(dev, case[:11] + (False, 0), wide, w, None)
    assert float((out - ref).norm() / ref.norm()) < 3e-6
    assert ops.nonfinite_count() == 0


def test_thin_route_refuses_other_shapes(dev):
    import ctypes
    from pcgan_amd.hip import lib as L, ops
    lib = L.load()
    no = [ops.make_desc(2, 8, 32, 32, 64, 7, 7, 1, 3, 0),       # 8 input channels
          ops.make_desc(2, 4, 32, 32, 48, 7, 7, 1, 3, 0),       # 48 output channels
          ops.make_desc(2, 4, 32, 32, 64, 3, 3, 1, 1, 0),       # 3x3
          ops.make_desc(2, 4, 32, 32, 64, 7, 7, 2, 3, 1),       # reflection with stride 2
          ops.make_desc(2, 4, 32, 32, 64, 7, 7, 1, 3, 0, ops.BF16)]
    for d in no:
        assert not lib.pcgan_conv2d_thin_supported(ctypes.byref(d), L.PASS_FWD)
        assert lib.pcgan_conv2d_thin_packed_bytes(ctypes.byref(d), L.PASS_FWD) == 0
    d = ops.make_desc(2, 4, 32, 32, 64, 7, 7, 1, 3, 1)
    assert lib.pcgan_conv2d_thin_supported(ctypes.byref(d), L.PASS_FWD)
    assert not lib.pcgan_conv2d_thin_supported(ctypes.byref(d), L.PASS_BWD_DATA)        # 64 output channels: not a thin data gradient
    x = torch.zeros(2, 4, 32, 32, device=dev)
    pk = torch.empty(lib.pcgan_conv2d_thin_packed_bytes(ctypes.byref(d), L.PASS_FWD), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.pcgan_conv2d_fwd_thin(ctypes.byref(d), x.data_ptr(), None, 0, pk.data_ptr(), None, x.data_ptr(), 0, 0.0, st) != 0
    assert b'null pointer' in lib.pcgan_last_error()
