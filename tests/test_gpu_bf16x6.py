"""GPU: the opt-in bf16-split convolution forward (csrc/bf16x6_conv.hip, through the C-ABI) against the oracle's convolution
in float64: its error must be at the level of the fp32 path's own (tolerance 3e-6 relative L2, the fp32 MFMA kernel measures
~6e-7 at K = 2304), for reflect and zero padding, ragged pixel / channel tiles, 1x1 / 3x3 / 5x5 filters, bias + activation."""
import ctypes

import pytest
import torch

from oracle import ops_ref as R

pytestmark = pytest.mark.gpu


def _run(dev, N, C, H, W, K, k, pad, pad_mode, act=0, bias=True, seed=0):
    from pcgan_amd.hip import lib as L, ops
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g).relu_()
    w = torch.randn(K, C, k, k, generator=g) * 0.05
    b = torch.randn(K, generator=g) if bias else None
    d = ops.make_desc(N, C, H, W, K, k, k, 1, pad, pad_mode)
    lib = L.load()
    assert lib.pcgan_conv2d_bsplit_supported(ctypes.byref(d))
    xd, wd = x.to(dev), w.to(dev)
    bd = b.to(dev) if bias else None
    pk = torch.empty(lib.pcgan_conv2d_bsplit_packed_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.pcgan_conv2d_bsplit_pack(ctypes.byref(d), wd.data_ptr(), pk.data_ptr(), st), 'pack')
    y = torch.full((N, K, d.P, d.Q), float('nan'), device=dev)
    L.check(lib.pcgan_conv2d_fwd_bsplit(ctypes.byref(d), xd.data_ptr(), pk.data_ptr(), bd.data_ptr() if bias else None, y.data_ptr(),
                                        act, 0.2, st), 'fwd')
    ref = R.conv2d(x.double(), w.double(), b.double() if bias else None, 1, pad, pad_mode)
    if act == 1:
        ref = ref.relu()
    elif act == 2:
        ref = torch.nn.functional.leaky_relu(ref, 0.2)
    # the fp32 path of the product on the same inputs
    y32 = ops.conv2d_fwd(xd, wd, bd, 1, pad, pad_mode, act, 0.2)
    torch.cuda.synchronize()
    e = lambda t: float((t.double().cpu() - ref).norm() / ref.norm())
    return e(y), e(y32)


@pytest.mark.parametrize('N,C,H,W,K,k,pad,mode,act', [
    (2, 256, 32, 32, 256, 3, 1, 1, 0),       # the residual-block convolution (reflect)
    (2, 64, 20, 28, 128, 3, 1, 0, 1),        # zero padding, ReLU
    (3, 32, 9, 7, 40, 3, 1, 1, 2),           # ragged pixel tile (189 pixels), ragged channel tile (40 of 128)
    (1, 16, 13, 13, 32, 5, 2, 0, 0),         # 25 taps
    (2, 48, 8, 8, 200, 1, 0, 0, 0),          # 1x1, two channel tiles
    (1, 128, 6, 6, 64, 3, 0, 0, 0),          # valid convolution (output 4x4)
    (2, 32, 12, 10, 512, 3, 1, 1, 1),        # two 256-row channel tiles (8-wave workgroups), ragged pixel tile
    # shapes the window ("halo") kernel takes: width 32 / 64, whole image rows per pixel tile, channel chunks in pairs
    (1, 32, 4, 32, 256, 3, 1, 1, 1),         # one tile per image: both mirror rows in its window
    (3, 96, 12, 32, 256, 3, 1, 1, 0),        # three tiles per image, three chunk pairs
    (2, 64, 8, 64, 512, 3, 1, 1, 2),         # width 64 (two rows per tile), two 256-row channel tiles
    (1, 32, 64, 64, 256, 3, 1, 1, 0),        # the 256x256 configurations' plane
])
def test_bsplit_forward_has_fp32_accuracy(dev, N, C, H, W, K, k, pad, mode, act):
    e6, e32 = _run(dev, N, C, H, W, K, k, pad, mode, act)
    assert e6 < 3e-6 and e6 < 4 * e32 + 5e-7, (e6, e32)


def test_bsplit_unsupported_shapes_are_refused(dev):
    from pcgan_amd.hip import lib as L, ops
    lib = L.load()
    for args in [(1, 3, 8, 8, 64, 3, 3, 1, 1, 0), (1, 16, 8, 8, 64, 3, 3, 2, 1, 0), (1, 16, 8, 8, 3, 3, 3, 1, 1, 0)]:
        d = ops.make_desc(*args)
        assert not lib.pcgan_conv2d_bsplit_supported(ctypes.byref(d))
        assert lib.pcgan_conv2d_bsplit_packed_bytes(ctypes.byref(d)) == 0


def test_host_switch_routes_large_convs_and_follows_the_weights(dev, monkeypatch):
    """PCGAN_BF16X6=1 (here: the module flag) sends packed forward calls of full-tile stride-1 convolutions to the split
    kernel; the packed pieces are re-made when the weights change; small layers keep the fp32 kernel"""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BF16X6', True)
    monkeypatch.setattr(ops, 'HSPLIT', False)        # (the three-piece bf16 split; the fp16 route has its own tests below)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 128, 64, 64, generator=g).relu_().to(dev)
    w = (torch.randn(128, 128, 3, 3, generator=g) * 0.05).to(dev)
    ref = R.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1, 1)
    cache = {}
    y = ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cache)
    assert list(cache) == [(ops.PASS_FWD_BSPLIT, 1, 1, 1, 0)]      # (pass, stride, pad, pad_mode, dtype)
    assert float((y.double().cpu() - ref).norm() / ref.norm()) < 3e-6
    assert not torch.equal(y, ops.conv2d_fwd(x, w, None, 1, 1, 1))            # another kernel: other low bits
    w.mul_(2.0)
    y2 = ops.conv2d_fwd(x, w, None, 1, 1, 1, pack_cache=cache)
    assert float((y2.double().cpu() - 2 * ref).norm() / ref.norm()) < 6e-6
    small = {}
    ops.conv2d_fwd(x[:1, :, :16, :16].contiguous(), w, None, 1, 1, 1, pack_cache=small)
    assert list(small) == [(0, 1, 1, 1, 0)]


@pytest.mark.parametrize('N,C,H,W,K', [
    (2, 256, 32, 32, 256),      # the residual-block convolution
    (3, 64, 9, 7, 48),          # ragged tiles, 128-row tile variant, K not a multiple of 32
    (1, 512, 4, 4, 32),         # smallest grid (rows 0..3: both mirror rows adjacent), two 256-row tiles
    (2, 128, 6, 20, 16),        # one 16-channel K chunk
    # the window kernel: sums for rows / columns 1 and H-2 / W-2 as extra window rows / columns
    (1, 512, 4, 32, 32),        # one tile per image holds rows 1 and H-2; two 256-row tiles
    (2, 256, 4, 64, 64),        # width 64: rows {0, 1} and {2, 3} in different tiles
    (3, 256, 12, 32, 96),       # three tiles per image: first / middle / last differ in their sum rows
    (1, 256, 64, 64, 32),       # the 256x256 configurations' plane
])
def test_bsplit_reflect_data_gradient(dev, N, C, H, W, K):
    """dx of ReflectionPad2d(1) + Conv2d(3x3) through the split kernel against autograd in float64 and the fp32 kernel"""
    from pcgan_amd.hip import lib as L, ops
    g = torch.Generator().manual_seed(N * 1000 + C + K)
    dy = torch.randn(N, K, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    x = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    R.conv2d(x, w.double(), None, 1, 1, 1).backward(dy.double())
    ref = x.grad
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    lib = L.load()
    assert lib.pcgan_conv2d_bsplit_dgrad_supported(ctypes.byref(d))
    dyd, wd = dy.to(dev), w.to(dev)
    pk = torch.empty(lib.pcgan_conv2d_bsplit_dgrad_packed_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.pcgan_conv2d_bsplit_dgrad_pack(ctypes.byref(d), wd.data_ptr(), pk.data_ptr(), st), 'pack')
    dx = torch.full((N, C, H, W), float('nan'), device=dev)
    L.check(lib.pcgan_conv2d_bwd_data_bsplit(ctypes.byref(d), dyd.data_ptr(), pk.data_ptr(), dx.data_ptr(), st), 'dgrad')
    dx32 = ops.conv2d_bwd_data(dyd, wd, (H, W), 1, 1, 1)
    torch.cuda.synchronize()
    e = lambda t: float((t.double().cpu() - ref).norm() / ref.norm())
    assert e(dx) < 3e-6 and e(dx) < 4 * e(dx32) + 5e-7, (e(dx), e(dx32))


@pytest.mark.parametrize('N,C,H,W,K,acc', [
    (2, 256, 32, 32, 256, False),     # the residual-block convolution
    (3, 40, 6, 16, 128, True),        # 128-row tile, ragged column tile (360 columns), accumulate into an existing gradient
    (1, 16, 4, 16, 256, False),       # smallest grid
])
def test_bsplit_weight_gradient(dev, N, C, H, W, K, acc):
    """dW of ReflectionPad2d(1) + Conv2d(3x3) through the split kernel (roles turned, reduction over pixels) against autograd in
    float64 and the fp32 kernel"""
    from pcgan_amd.hip import lib as L, ops
    g = torch.Generator().manual_seed(N * 100 + C + K)
    x = torch.randn(N, C, H, W, generator=g)
    dy = torch.randn(N, K, H, W, generator=g)
    w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    R.conv2d(x.double(), w, None, 1, 1, 1).backward(dy.double())
    ref = w.grad
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    lib = L.load()
    assert lib.pcgan_conv2d_bsplit_wgrad_supported(ctypes.byref(d))
    xd, dyd = x.to(dev), dy.to(dev)
    ws = torch.empty(lib.pcgan_conv2d_bsplit_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    base = torch.randn(K, C, 3, 3, generator=g) if acc else None
    dw = base.to(dev).clone() if acc else torch.full((K, C, 3, 3), float('nan'), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.pcgan_conv2d_bwd_weight_bsplit(ctypes.byref(d), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), int(acc), ws.data_ptr(),
                                               ws.numel(), st), 'wgrad')
    dw32 = ops.conv2d_bwd_weight(xd, dyd, (K, C, 3, 3), 1, 1, 1)
    torch.cuda.synchronize()
    got = dw.double().cpu() - (base.double() if acc else 0)
    e = lambda t: float((t - ref).norm() / ref.norm())
    assert e(got) < 3e-6 and e(got) < 4 * e(dw32.double().cpu()) + 5e-7, (e(got), e(dw32.double().cpu()))


# ---- the fp16 two-piece route (csrc/bf16x6_conv.hip "fp16 route"; C-ABI pcgan_conv2d_*_hsplit) ------------------------------------
def _hsplit(dev, N, C, H, W, K, x, w, dgrad):
    """forward (x = input) or data gradient (x = dy) through the fp16 route, and through the fp32 kernels of the product"""
    from pcgan_amd.hip import lib as L, ops
    lib = L.load()
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    pass_ = L.PASS_BWD_DATA if dgrad else L.PASS_FWD
    assert lib.pcgan_conv2d_hsplit_supported(ctypes.byref(d), pass_)
    xd, wd = x.to(dev), w.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    pk = torch.empty(lib.pcgan_conv2d_hsplit_packed_bytes(ctypes.byref(d), pass_), dtype=torch.uint8, device=dev)
    L.check(lib.pcgan_conv2d_hsplit_pack(ctypes.byref(d), pass_, wd.data_ptr(), pk.data_ptr(), st), 'pack')
    slots = lib.pcgan_absmax_slots(xd.numel())
    amax = torch.full((slots,), float('nan'), device=dev)
    L.check(lib.pcgan_absmax(xd.data_ptr(), xd.numel(), 0, amax.data_ptr(), slots, st), 'absmax')
    assert float(amax.max()) == float(x.abs().max())
    out = torch.full((N, C if dgrad else K, H, W), float('nan'), device=dev)
    if dgrad:
        L.check(lib.pcgan_conv2d_bwd_data_hsplit(ctypes.byref(d), xd.data_ptr(), amax.data_ptr(), slots, pk.data_ptr(), out.data_ptr(), st), 'dgrad')
        ops_hs, ops.HSPLIT = ops.HSPLIT, False
        try:
            o32 = ops.conv2d_bwd_data(xd, wd, (H, W), 1, 1, 1)
        finally:
            ops.HSPLIT = ops_hs
    else:
        L.check(lib.pcgan_conv2d_fwd_hsplit(ctypes.byref(d), xd.data_ptr(), amax.data_ptr(), slots, pk.data_ptr(), None, out.data_ptr(), 0, 0.0, st), 'fwd')
        o32 = ops.conv2d_fwd(xd, wd, None, 1, 1, 1)
    torch.cuda.synchronize()
    return out.cpu(), o32.cpu()


HSPLIT_SHAPES = [
    (2, 256, 32, 32, 256),      # the residual-block convolution
    (1, 32, 4, 32, 256),        # one tile per image
    (2, 64, 8, 64, 512),        # width 64, two 256-row tiles
    (3, 96, 12, 32, 256),       # three tiles per image
]


@pytest.mark.parametrize('N,C,H,W,K', HSPLIT_SHAPES)
@pytest.mark.parametrize('data', ['relu_normal', 'wide_range', 'tiny', 'huge'])
def test_hsplit_forward_has_fp32_accuracy(dev, N, C, H, W, K, data):
    g = torch.Generator().manual_seed(N + C + K)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    if data == 'relu_normal':
        x = x.relu_()
    elif data == 'wide_range':         # magnitudes over twelve decades inside one tensor
        x = x * torch.pow(10.0, torch.rand(N, C, H, W, generator=g) * 12 - 9)
        w = w * torch.pow(10.0, torch.rand(K, C, 3, 3, generator=g) * 6 - 3)
    elif data == 'tiny':
        x, w = x * 1e-30, w * 1e-6
    else:
        x, w = x * 1e12, w * 1e8
    y, y32 = _hsplit(dev, N, C, H, W, K, x, w, False)
    ref = R.conv2d(x.double(), w.double(), None, 1, 1, 1)
    e = lambda t: float((t.double() - ref).norm() / ref.norm())
    assert e(y) < 3e-6 and e(y) < 4 * e(y32) + 5e-7, (e(y), e(y32))


@pytest.mark.parametrize('N,K,H,W,C', HSPLIT_SHAPES)
@pytest.mark.parametrize('data', ['normal', 'wide_range'])
def test_hsplit_reflect_data_gradient(dev, N, K, H, W, C, data):
    g = torch.Generator().manual_seed(N + C + K + 1)
    dy = torch.randn(N, K, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * 0.05
    if data == 'wide_range':
        dy = dy * torch.pow(10.0, torch.rand(N, K, H, W, generator=g) * 12 - 9)
    x = torch.zeros(N, C, H, W, dtype=torch.float64, requires_grad=True)
    R.conv2d(x, w.double(), None, 1, 1, 1).backward(dy.double())
    ref = x.grad
    dx, dx32 = _hsplit(dev, N, C, H, W, K, dy, w, True)
    e = lambda t: float((t.double() - ref).norm() / ref.norm())
    assert e(dx) < 3e-6 and e(dx) < 4 * e(dx32) + 5e-7, (e(dx), e(dx32))


def test_absmax_unaligned_and_ragged(dev):
    from pcgan_amd.hip import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    base = torch.randn(3000007, generator=g).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    for off, n, slots in ((0, 10007, 1), (1, 10006, 3), (3, 5, 1), (2, 4099, 64), (5, 1, 2), (0, 3000007, None), (3, 2999999, 64)):
        v = base[off:off + n]
        k = slots or lib.pcgan_absmax_slots(n)
        out = torch.full((k,), float('nan'), device=dev)
        L.check(lib.pcgan_absmax(v.data_ptr(), n, 0, out.data_ptr(), k, st), 'absmax')
        assert float(out.max()) == float(v.abs().max()) and bool((out >= 0).all()), (off, n, k)


@pytest.mark.parametrize('N,C,H,W,acc', [
    (2, 256, 32, 32, False),     # the residual-block convolution
    (3, 40, 6, 16, True),        # ragged column tile (360 columns), accumulate into an existing gradient
    (1, 16, 4, 16, False),       # smallest grid
    (2, 64, 8, 64, True),        # width 64
])
@pytest.mark.parametrize('data', ['normal', 'wide_range'])
def test_hsplit_weight_gradient(dev, N, C, H, W, acc, data):
    """dW of ReflectionPad2d(1) + Conv2d(3x3, 256 output channels) on the fp16 route against autograd in float64 and the fp32 kernel"""
    from pcgan_amd.hip import lib as L, ops
    K = 256
    g = torch.Generator().manual_seed(N * 100 + C + W)
    x = torch.randn(N, C, H, W, generator=g)
    dy = torch.randn(N, K, H, W, generator=g)
    if data == 'wide_range':
        x = x * torch.pow(10.0, torch.rand(N, C, H, W, generator=g) * 8 - 6)
        dy = dy * torch.pow(10.0, torch.rand(N, K, H, W, generator=g) * 12 - 14)
    w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    R.conv2d(x.double(), w, None, 1, 1, 1).backward(dy.double())
    ref = w.grad
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    lib = L.load()
    assert lib.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d))
    xd, dyd = x.to(dev), dy.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pcgan_conv2d_hsplit_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    base = torch.randn(K, C, 3, 3, generator=g) * float(ref.abs().max()) if acc else None
    dw = base.to(dev).clone() if acc else torch.full((K, C, 3, 3), float('nan'), device=dev)
    # operand maxima as partial values (per plane), as the instance-norm kernels hand them over
    xmax = xd.abs().amax(dim=(2, 3)).reshape(-1).contiguous()
    dmax = dyd.abs().amax(dim=(2, 3)).reshape(-1).contiguous()
    L.check(lib.pcgan_conv2d_bwd_weight_hsplit(ctypes.byref(d), xd.data_ptr(), xmax.data_ptr(), xmax.numel(), dyd.data_ptr(), dmax.data_ptr(),
                                               dmax.numel(), dw.data_ptr(), int(acc), ws.data_ptr(), ws.numel(), st), 'wgrad')
    old, ops.HSPLIT = ops.HSPLIT, False
    try:
        dw32 = ops.conv2d_bwd_weight(xd, dyd, (K, C, 3, 3), 1, 1, 1)
    finally:
        ops.HSPLIT = old
    torch.cuda.synchronize()
    got = dw.double().cpu() - (base.double() if acc else 0.0)
    e = lambda t: float((t - ref).norm() / ref.norm())
    assert e(got) < (3e-6 if not acc else 1e-5) and (acc or e(got) < 4 * e(dw32.double().cpu()) + 5e-7), (e(got), e(dw32.double().cpu()))


@pytest.mark.parametrize('N,C,H,W,K,k,stride,pad,mode', [
    (2, 64, 32, 32, 128, 3, 2, 1, 0),     # generator down-sampling: 3x3 stride 2, 128-row tile
    (2, 128, 32, 64, 256, 3, 2, 1, 0),    # 256-row tile, stride 2, non-square
    (2, 64, 32, 32, 128, 4, 2, 1, 0),     # PatchGAN 4x4 stride 2 (16 taps: 1024 columns)
    (3, 24, 16, 16, 96, 3, 1, 1, 0),      # ragged rows (96 of 128) and columns (216), zero padding stride 1
    (1, 16, 18, 34, 32, 5, 1, 1, 0),      # 25 taps, padding smaller than the filter radius (output 16 x 32)
    (2, 32, 16, 16, 64, 1, 1, 0, 0),      # 1x1, no padding (no padded copy)
    (8, 128, 16, 16, 64, 3, 1, 1, 0),     # 1024 small planes: the wave-per-plane padded copy, zero padding
    (9, 128, 16, 32, 64, 3, 1, 1, 1),     # ... reflection, a plane count that is no multiple of 4, non-square
    (2, 256, 32, 32, 256, 3, 1, 1, 1),    # the residual block's shape (bf16 tensors: 256-column workgroups, the mirror applied in the gather)
    (4, 256, 12, 16, 64, 5, 1, 2, 1),     # ... padding 2
    (2, 4, 32, 48, 64, 7, 1, 3, 1),       # the generator's stem: 4 channels x 49 taps = 196 columns (two ragged column tiles), reflection 3
    (2, 3, 20, 32, 64, 7, 1, 3, 0),       # 3 channels, zero padding
])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_hsplit_weight_gradient_general(dev, N, C, H, W, K, k, stride, pad, mode, dtype):
    """the matrix-pipe weight gradient on the other layer shapes it takes (fp16 two-piece route / bf16 one-product route) against
    autograd in float64"""
    from pcgan_amd.hip import lib as L, ops
    g = torch.Generator().manual_seed(N * 100 + C + K + k)
    x = torch.randn(N, C, H, W, generator=g)
    P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(N, K, P, Q, generator=g) * 0.01
    if dtype == 'bf16':
        x, dy = x.bfloat16().float(), dy.bfloat16().float()       # bf16-valued operands: the products are exact in fp32
    w = torch.zeros(K, C, k, k, dtype=torch.float64, requires_grad=True)
    R.conv2d(x.double(), w, None, stride, pad, mode).backward(dy.double())
    ref = w.grad
    dt = L.BF16 if dtype == 'bf16' else L.F32
    d = ops.make_desc(N, C, H, W, K, k, k, stride, pad, mode, dt)
    lib = L.load()
    assert lib.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d))
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    xd, dyd = x.to(dev).to(tdt), dy.to(dev).to(tdt)
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.pcgan_conv2d_hsplit_wgrad_workspace_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
    dw = torch.full((K, C, k, k), float('nan'), device=dev)
    if dtype == 'fp32':
        xmax, dmax = ops.amax_of(xd), ops.amax_of(dyd)
        args = (xmax.data_ptr(), xmax.numel(), dyd.data_ptr(), dmax.data_ptr(), dmax.numel())
    else:
        args = (None, 0, dyd.data_ptr(), None, 0)
    L.check(lib.pcgan_conv2d_bwd_weight_hsplit(ctypes.byref(d), xd.data_ptr(), args[0], args[1], args[2], args[3], args[4], dw.data_ptr(), 0,
                                               ws.data_ptr(), ws.numel(), st), 'wgrad')
    torch.cuda.synchronize()
    e = float((dw.double().cpu() - ref).norm() / ref.norm())
    assert e < 3e-6, e
    # the host routes these shapes there (packed or not: the weight gradient has no packed operand)
    old = ops.BSPLIT_MIN_PIXELS
    ops.BSPLIT_MIN_PIXELS = 0
    try:
        dw2 = ops.conv2d_bwd_weight(xd, dyd, (K, C, k, k), stride, pad, mode)
    finally:
        ops.BSPLIT_MIN_PIXELS = old
    assert torch.equal(dw2, dw) or float((dw2.double().cpu() - ref).norm() / ref.norm()) < 3e-6


@pytest.mark.parametrize('N,C,H,W,K,k,stride,pad', [
    (2, 256, 16, 16, 512, 4, 1, 1),      # the PatchGAN's 256 -> 512 layer (networks.py:753-775): two row tiles, output 15 x 15 (one ragged stage per row)
    (3, 32, 16, 16, 320, 4, 1, 1),       # a ragged second row tile (320 = 256 + 64)
    (2, 64, 26, 28, 64, 3, 1, 1),        # output width 28: two stages per row, the second ragged
    (1, 16, 9, 30, 32, 3, 2, 1),         # stride 2, output 5 x 15
    (2, 32, 14, 14, 64, 3, 1, 1),        # output width 14
    (2, 48, 15, 30, 96, 4, 2, 1),        # 4x4 stride 2, output 7 x 15: the last tap column falls on column W (zeroed in registers)
    (4, 64, 128, 128, 128, 3, 2, 1),     # the generator's first down-sampling layer at full size (no padded copy of the 134 MB input)
    (2, 128, 32, 32, 256, 4, 2, 1),      # PatchGAN 128 -> 256 4x4 stride 2: 2048 columns (bf16 tensors: 256-column workgroups)
    (2, 32, 16, 32, 64, 3, 1, 1),        # stride 1, whole stages: the rotate path with 2-byte elements
])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_hsplit_weight_gradient_inline_zero_padding(dev, N, C, H, W, K, k, stride, pad, dtype):
    """the general form of the matrix-pipe weight gradient (zero padding applied inside the gather, ragged output width, more than 256
    output channels; fp16 two-piece route / bf16 one-product route) against autograd in float64 and the fp32-MFMA kernel"""
    from pcgan_amd.hip import lib as L, ops
    g = torch.Generator().manual_seed(N * 100 + C + K + k + W)
    x = torch.randn(N, C, H, W, generator=g)
    P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(N, K, P, Q, generator=g) * 0.01
    half = dtype == 'bf16'
    if half:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()       # bf16-valued operands: the products are exact in fp32
    d = ops.make_desc(N, C, H, W, K, k, k, stride, pad, 0, L.BF16 if half else L.F32)
    lib = L.load()
    w = torch.zeros(K, C, k, k, dtype=torch.float64, requires_grad=True)
    R.conv2d(x.double(), w, None, stride, pad, 0).backward(dy.double())
    ref = w.grad
    assert lib.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) and lib.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d))
    tdt = torch.bfloat16 if half else torch.float32
    xd, dyd = x.to(dev).to(tdt), dy.to(dev).to(tdt)
    st = torch.cuda.current_stream().cuda_stream
    nbytes = lib.pcgan_conv2d_hsplit_wgrad_workspace_bytes(ctypes.byref(d))
    assert nbytes % (K * C * k * k * 4) == 0 and nbytes // (K * C * k * k * 4) >= 1      # whole partial sums only: no room for a padded copy
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dw = torch.full((K, C, k, k), float('nan'), device=dev)
    if half:
        args = (None, 0, dyd.data_ptr(), None, 0)
    else:
        xmax, dmax = ops.amax_of(xd), ops.amax_of(dyd)
        args = (xmax.data_ptr(), xmax.numel(), dyd.data_ptr(), dmax.data_ptr(), dmax.numel())
    L.check(lib.pcgan_conv2d_bwd_weight_hsplit(ctypes.byref(d), xd.data_ptr(), args[0], args[1], args[2], args[3], args[4], dw.data_ptr(), 0,
                                               ws.data_ptr(), ws.numel(), st), 'wgrad')
    old, ops.HSPLIT = ops.HSPLIT, False
    try:
        dw32 = ops.conv2d_bwd_weight(xd.float(), dyd.float(), (K, C, k, k), stride, pad, 0)      # the fp32-MFMA kernel
    finally:
        ops.HSPLIT = old
    torch.cuda.synchronize()
    e = lambda t: float((t.double().cpu() - ref).norm() / ref.norm())
    assert e(dw) < 3e-6 and e(dw) < 4 * e(dw32) + 5e-7, (e(dw), e(dw32))
    # every tap separately (a wrong border column or row shows in the border taps only)
    per_tap = ((dw.double().cpu() - ref) ** 2).sum(dim=(0, 1)).sqrt() / (ref ** 2).sum(dim=(0, 1)).sqrt()
    assert float(per_tap.max()) < 1e-5, per_tap
    # the host routes these shapes there -- except a ragged width under a half-empty 128-row tile (ResNet-18 layer1: the fp32 kernel wins)
    old = ops.BSPLIT_MIN_PIXELS
    ops.BSPLIT_MIN_PIXELS = 0
    try:
        route = ops._plan(L.PASS_BWD_WEIGHT, N, C, H, W, K, k, k, stride, pad, 0, L.BF16 if half else L.F32).route
        assert route == ('hsplit' if (K >= 128 or Q % 16 == 0) else 'generic'), route
        dw2 = ops.conv2d_bwd_weight(xd, dyd, (K, C, k, k), stride, pad, 0)
    finally:
        ops.BSPLIT_MIN_PIXELS = old
    assert torch.equal(dw2, dw) if route == 'hsplit' else e(dw2) < 3e-6


@pytest.mark.allow_nonfinite
def test_stale_operand_maximum_is_caught_loudly(dev, monkeypatch):
    """The fp16 route trusts `tensor._pcgan_amax` on the tensor VERSION.  A write through `.data` (or a raw-pointer kernel) after a
    norm kernel attached the maxima does not bump the version: the stale (too small) maximum makes the scaled operand overflow fp16
    and every product it enters becomes inf / NaN.  That must not pass silently: the kernels' non-finite sentinel counts it and the
    host raises at its next synchronisation point (ops.check_nonfinite, called by BaseModel.get_current_losses and after every GPU
    test).  Rewriting the tensor the honest way (an op that bumps the version) drops the stale maxima and the result is right."""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    monkeypatch.setattr(ops, 'AMAX_AUDIT_EVERY', 0)               # the kernels' own sentinel in isolation (the audit: next test)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 256, 32, 32, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.05).to(dev)
    h, _, _ = ops.instnorm_fwd(x, None, 1e-5, 0, 0.0)             # a norm kernel attaches the plane maxima of its output
    assert '_pcgan_amax' in h.__dict__
    assert ops.nonfinite_count() == 0
    cache = {}
    y0 = ops.conv2d_fwd(h, w, None, 1, 1, 1, pack_cache=cache)
    assert torch.isfinite(y0).all() and ops.nonfinite_count() == 0
    h.data.mul_(1.0e4)                                              # behind autograd's back: same version, 10^4 x larger values
    y1 = ops.conv2d_fwd(h, w, None, 1, 1, 1, pack_cache=cache)
    torch.cuda.synchronize()
    assert not torch.isfinite(y1).all(), 'the stale maximum should have overflowed the fp16 pieces'
    with pytest.raises(RuntimeError, match='inf / NaN'):
        ops.check_nonfinite('test')
    assert ops.nonfinite_count() == 0, 'the check resets the counter'
    # the same values written by an op that bumps the version: the stale maxima are ignored, an absmax pass takes their place
    h2 = h.clone()
    h2.mul_(1.0)
    assert h2.__dict__.get('_pcgan_amax') is None or h2.__dict__['_pcgan_amax'][0] != h2._version
    y2 = ops.conv2d_fwd(h2, w, None, 1, 1, 1, pack_cache=cache)
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(h2.double().cpu(), (1, 1, 1, 1), mode='reflect'), w.double().cpu())
    err = float((y2.double().cpu() - ref).norm() / ref.norm())
    assert err < 3e-6 and ops.nonfinite_count() == 0, err
    # the weight-gradient and data-gradient kernels carry the sentinel too
    dy = torch.randn(2, 256, 32, 32, generator=g).to(dev)
    ops._attach_amax(dy, torch.full((4,), 1e-3, device=dev))       # a maximum that is far too small
    ops.conv2d_bwd_data(dy, w, (32, 32), 1, 1, 1, pack_cache=cache)
    assert ops.nonfinite_count() > 0
    ops.conv2d_bwd_weight(h2, dy, (256, 256, 3, 3), 1, 1, 1)
    assert ops.nonfinite_count() > 0


@pytest.mark.allow_nonfinite
@pytest.mark.parametrize('every', [1, 3])
def test_stale_operand_maximum_that_is_too_large_is_caught_too(dev, monkeypatch, every):
    """VERDICT r3 "weak" 2: a tensor rewritten through `.data` with SMALLER values keeps a maximum that is too large; nothing overflows,
    the scaled operand just sits far below the fp16 target range, its low piece goes subnormal and the products lose bits SILENTLY --
    the non-finite sentinel cannot see that.  The audit (ops._audit_amax: every AMAX_AUDIT_EVERY-th consumption of attached maxima is
    re-taken by an absmax pass and compared on the device) does: check_nonfinite raises, naming the direction.  Honest tensors never
    trip it (the attached maxima ARE the maxima of the stored values: equality, not a tolerance)."""
    from pcgan_amd.hip import ops
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    monkeypatch.setattr(ops, 'AMAX_AUDIT_EVERY', every)
    ops.AMAX_STATS['attached'] = 0
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 256, 32, 32, generator=g).to(dev)
    w = (torch.randn(256, 256, 3, 3, generator=g) * 0.05).to(dev)
    cache = {}
    a0 = ops.AMAX_STATS['audited']
    for _ in range(2 * every):          # honest use: producer-attached and absmax-computed maxima both pass the audit
        h, _, _ = ops.instnorm_fwd(x, None, 1e-5, 1, 0.0)
        y0 = ops.conv2d_fwd(h, w, None, 1, 1, 1, pack_cache=cache)
        ops.conv2d_bwd_weight(h, y0, (256, 256, 3, 3), 1, 1, 1)      # y0: maxima taken by an absmax pass inside, then found attached
    assert ops.AMAX_STATS['audited'] - a0 >= 2
    assert ops.stale_maxima_count() == (0, 0, 0)
    ops.check_nonfinite('honest')
    ref = torch.nn.functional.conv2d(torch.nn.functional.pad(h.double().cpu(), (1, 1, 1, 1), mode='reflect'), w.double().cpu())
    e_ok = float((y0.double().cpu() - ref).norm() / ref.norm())
    h.data.mul_(1.0e-8)                 # behind the version counter: the claim is now 10^8 x too large
    ys = [ops.conv2d_fwd(h, w, None, 1, 1, 1, pack_cache=cache) for _ in range(every)]
    torch.cuda.synchronize()
    assert torch.isfinite(ys[-1]).all() and ops.nonfinite_count() == 0, 'nothing overflows: this failure is silent without the audit'
    e_bad = float((ys[-1].double().cpu() - ref * 1e-8).norm() / (ref * 1e-8).norm())
    assert e_bad > 30 * e_ok, 'the under-scaled operand should have lost precision (%.2e vs %.2e): is the scenario still the silent one?' % (e_bad, e_ok)
    small, large, other = ops.stale_maxima_count(reset=False)
    assert large >= 1 and small == 0, (small, large, other)
    with pytest.raises(RuntimeError, match='less than 2\\^-8'):
        ops.check_nonfinite('test')
    assert ops.stale_maxima_count() == (0, 0, 0), 'the check resets the counters'
    # ... and the other direction is reported by the audit as well (before / besides the overflow sentinel)
    h.data.mul_(1.0e11)
    for _ in range(every):
        ops.conv2d_fwd(h, w, None, 1, 1, 1, pack_cache=cache)
    small, large, other = ops.stale_maxima_count()
    assert small >= 1 and large == 0
    ops.nonfinite_count()               # (the overflow itself: counted by the kernels, cleared here)


@pytest.mark.parametrize('case', [
    ('res fwd (window kernel)', 2, 256, 32, 256, 3, 1, 1, 1, False),
    ('res dgrad (window kernel)', 2, 256, 32, 256, 3, 1, 1, 1, True),
    ('G.down2 fwd (hgemm, stride 2)', 2, 128, 64, 256, 3, 2, 1, 0, False),
    ('G.up1 = dgrad of a stride-2 conv (hgemm, 4 phases)', 2, 128, 64, 256, 3, 2, 1, 0, True),
    ('D.c3 fwd (hgemm, 4x4)', 2, 256, 16, 512, 4, 1, 1, 0, False),
], ids=lambda c: c[0])
def test_per_channel_weight_scales(dev, monkeypatch, case):
    """fp16-route robustness PER CHANNEL (round-2 verdict): the weights are scaled by ONE POWER OF TWO PER ROW of the pass's weight
    matrix (per output channel in the forward pass, per input channel in the data gradient), not per tensor -- so an output channel
    whose filter is 2^-20 of the tensor's largest weight is still computed with 22 significand bits (an affine-less InstanceNorm
    behind it rescales that channel to O(1)).  Filters whose per-row magnitudes span 2^-20 .. 2^20; asserted: the relative L2 error
    of EVERY output channel against float64 is at the fp32 MFMA kernel's own level (4 x its worst channel + 5e-7).  With one scale
    per tensor the small channels came out at ~1e-5 .. 1e-3."""
    from pcgan_amd.hip import ops
    name, N, C, H, K, k, stride, pad, pm, dgrad = case
    if name.startswith('res'):       # (the batch of 2 is below the host's routing threshold for the window kernel)
        monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    g = torch.Generator().manual_seed(len(name))
    w = torch.randn(K, C, k, k, generator=g)
    rows = C if dgrad else K                       # the channels this pass produces
    mag = torch.pow(2.0, torch.randint(-20, 21, (rows,), generator=g).float())
    mag[0], mag[1] = 2.0 ** 20, 2.0 ** -20
    w = w * (mag.view(1, C, 1, 1) if dgrad else mag.view(K, 1, 1, 1))
    P = (H + 2 * pad - k) // stride + 1
    if dgrad:
        src = torch.randn(N, K, P, P, generator=g)
        xz = torch.zeros(N, C, H, H, dtype=torch.float64, requires_grad=True)
        R.conv2d(xz, w.double(), None, stride, pad, pm).backward(src.double())
        ref = xz.grad
    else:
        src = torch.randn(N, C, H, H, generator=g).relu_()
        ref = R.conv2d(src.double(), w.double(), None, stride, pad, pm)

    def run():
        sd, wd = src.to(dev), w.to(dev)
        cache = {}
        if dgrad:
            out = ops.conv2d_bwd_data(sd, wd, (H, H), stride, pad, pm, pack_cache=cache)
        else:
            out = ops.conv2d_fwd(sd, wd, None, stride, pad, pm, pack_cache=cache)
        return out.double().cpu()

    def per_channel_err(out):
        d = (out - ref).transpose(0, 1).reshape(rows, -1).norm(dim=1)
        n = ref.transpose(0, 1).reshape(rows, -1).norm(dim=1)
        return d / n

    r0 = dict(ops.ROUTE_STATS)
    e16 = per_channel_err(run())
    key = ('dgrad' if dgrad else 'fwd', 'hsplit' if name.startswith('res') else 'hgemm')
    assert ops.ROUTE_STATS.get(key, 0) == r0.get(key, 0) + 1, (key, ops.ROUTE_STATS)
    monkeypatch.setattr(ops, 'HSPLIT', False)
    monkeypatch.setattr(ops, 'BF16X6', False)      # the fp32 MFMA kernels (round 1's route) on the same data
    e32 = per_channel_err(run())
    worst16, worst32 = float(e16.max()), float(e32.max())
    assert worst16 <= 4 * worst32 + 5e-7, '%s: worst channel %.3e (fp32 MFMA %.3e); channel magnitudes 2^%d: %.3e, 2^%d: %.3e' % (
        name, worst16, worst32, 20, float(e16[0]), -20, float(e16[1]))
    assert worst16 < 3e-6


@pytest.mark.parametrize('N,C,H,W', [(32, 256, 32, 32), (3, 256, 32, 32), (2, 256, 64, 64), (1, 64, 2, 16), (2, 48, 16, 32)])
def test_residual_weight_gradient_reflects_in_the_gather(dev, monkeypatch, N, C, H, W):
    """round-2 verdict 5: the weight gradient of the reflection-padded 3x3 convolution read a padded COPY of x (37 MB written and re-read
    per launch at the benchmark's size, a launch of its own).  Now the mirror is applied inside the gather -- a row select per stage and one
    register move at the two image edges -- and the result must be the SAME BITS as with the padded copy (library option
    "wgrad_padcopy" = 1: the old path, kept for this comparison), incl. the first row of the tensor where the left-edge load starts one
    element in front of it."""
    from pcgan_amd.hip import ops, lib
    monkeypatch.setattr(ops, 'BSPLIT_MIN_PIXELS', 0)
    g = torch.Generator().manual_seed(N + C + H)
    K = 256 if C >= 64 else 64
    x = torch.randn(N, C, H, W, generator=g).to(dev)
    dy = torch.randn(N, K, H, W, generator=g).to(dev)
    assert ops._plan(ops._L.PASS_BWD_WEIGHT, N, C, H, W, K, 3, 3, 1, 1, 1, ops.F32).route == 'hsplit'
    assert lib.get_option('wgrad_padcopy') == 0
    a = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
    try:
        lib.set_option('wgrad_padcopy', 1)
        ops.clear_plans()             # (the workspace of the padded-copy form is larger)
        b = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
        torch.cuda.synchronize()
    finally:
        lib.set_option('wgrad_padcopy', 0)
        ops.clear_plans()
    assert torch.equal(a, b)
    w = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    R.conv2d(x.double().cpu(), w, None, 1, 1, 1).backward(dy.double().cpu())
    err = float((a.double().cpu() - w.grad).norm() / w.grad.norm())
    assert err < 3e-6, err


@pytest.mark.parametrize('N,C,H,W', [(32, 256, 32, 32), (16, 128, 16, 32), (16, 256, 8, 20), (32, 128, 64, 64)])
def test_weight_gradient_image_innermost_form(dev, N, C, H, W):
    """round 4 experiment (csrc/wgrad_direct.hip, library option "wgrad_direct", default off): the residual convolution's weight gradient
    with every MFMA operand fragment loaded straight from memory -- a transposing pre-pass scales / splits x and dy once into 16-byte
    records of 8 images, the main kernel has no LDS, no barrier and no split arithmetic -- against float64 on a slice of output channels
    (3e-6, the bound of the per-tap kernel) and against the per-tap kernel on the whole tensor; accumulation into an existing gradient;
    odd widths (W = 20: scalar tail of the pre-pass) and the split boundaries of the pixel reduction (all shapes)."""
    import ctypes
    from pcgan_amd.hip import ops, lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(N + C + H + W)
    K = C
    x = torch.randn(N, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(N, K, H, W, generator=g) * 0.05).to(dev)
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    assert lib.pcgan_conv2d_wgrad_direct_supported(ctypes.byref(d))
    nb = int(lib.pcgan_conv2d_wgrad_direct_workspace_bytes(ctypes.byref(d)))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    xmax, dmax = ops.amax_of(x), ops.amax_of(dy)
    vp = ctypes.c_void_p
    dw = torch.full((K, C, 3, 3), 0.5, device=dev)
    for acc in (0, 1):
        L.check(lib.pcgan_conv2d_bwd_weight_direct(ctypes.byref(d), vp(x.data_ptr()), vp(xmax.data_ptr()), xmax.numel(), vp(dy.data_ptr()),
                                                   vp(dmax.data_ptr()), dmax.numel(), vp(dw.data_ptr()), acc, vp(ws.data_ptr()), nb,
                                                   vp(torch.cuda.current_stream().cuda_stream)), 'bwd_weight_direct')
    torch.cuda.synchronize()
    once = dw / 2           # overwritten, then accumulated once more: exactly twice the gradient
    ks = [0, 1, K // 3, K - 1]
    w = torch.zeros(len(ks), C, 3, 3, dtype=torch.float64, requires_grad=True)
    xp = torch.nn.functional.pad(x.double().cpu(), (1, 1, 1, 1), mode='reflect')
    for n0 in range(0, N, 8):
        torch.nn.functional.conv2d(xp[n0:n0 + 8], w).backward(dy[n0:n0 + 8, ks].double().cpu())
    err = float((once[ks].double().cpu() - w.grad).norm() / w.grad.norm())
    assert err < 3e-6, err
    ref = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
    assert float((once.double() - ref.double()).norm() / ref.double().norm()) < 3e-6
    # refusals: what the form does not take is said so, not computed wrongly
    for bad in (ops.make_desc(8, C, H, W, K, 3, 3, 1, 1, 1), ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 0), ops.make_desc(N, 48, H, W, K, 3, 3, 1, 1, 1),
                ops.make_desc(N, C, H, W, K, 3, 3, 2, 1, 1)):
        assert not lib.pcgan_conv2d_wgrad_direct_supported(ctypes.byref(bad))


@pytest.mark.parametrize('N,C,K,H,W', [(32, 256, 256, 32, 32), (3, 32, 128, 5, 16), (5, 64, 256, 7, 48), (8, 128, 128, 64, 64), (1, 32, 128, 3, 16)])
def test_weight_gradient_row_ring_form(dev, N, C, K, H, W):
    """csrc/wgrad_rowring.hip (round 4, library option "wgrad_rowring", default on): the residual convolution's weight gradient with a
    column tile of 32 input channels x all nine taps walking down a 16-pixel strip -- padded rows in a ring of four LDS slots, three
    element-shifted copies per row, dy straight from memory.  Against float64 on a slice of output channels (3e-6, the per-tap kernel's
    bound), against the per-tap kernel on the whole tensor, accumulation into an existing gradient; odd heights (the unrolled stage pair's
    tail), one image, three strips per row, a single strip (mirror columns on both sides of the same quad), the smallest height."""
    import ctypes
    from pcgan_amd.hip import ops, lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(N + C + K + H + W)
    x = torch.randn(N, C, H, W, generator=g).relu_().to(dev)
    dy = (torch.randn(N, K, H, W, generator=g) * 0.05).to(dev)
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1)
    assert lib.pcgan_conv2d_wgrad_rowring_supported(ctypes.byref(d))
    nb = int(lib.pcgan_conv2d_wgrad_rowring_workspace_bytes(ctypes.byref(d)))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    xmax, dmax = ops.amax_of(x), ops.amax_of(dy)
    vp = ctypes.c_void_p
    dw = torch.full((K, C, 3, 3), 0.5, device=dev)
    for acc in (0, 1):
        L.check(lib.pcgan_conv2d_bwd_weight_rowring(ctypes.byref(d), vp(x.data_ptr()), vp(xmax.data_ptr()), xmax.numel(), vp(dy.data_ptr()),
                                                    vp(dmax.data_ptr()), dmax.numel(), vp(dw.data_ptr()), acc, vp(ws.data_ptr()), nb,
                                                    vp(torch.cuda.current_stream().cuda_stream)), 'bwd_weight_rowring')
    torch.cuda.synchronize()
    once = dw / 2           # overwritten, then accumulated once more: exactly twice the gradient
    ks = sorted({0, 1, K // 3, K - 1})
    w = torch.zeros(len(ks), C, 3, 3, dtype=torch.float64, requires_grad=True)
    xp = torch.nn.functional.pad(x.double().cpu(), (1, 1, 1, 1), mode='reflect')
    for n0 in range(0, N, 8):
        torch.nn.functional.conv2d(xp[n0:n0 + 8], w).backward(dy[n0:n0 + 8, ks].double().cpu())
    err = float((once[ks].double().cpu() - w.grad).norm() / w.grad.norm())
    assert err < 3e-6, err
    # the per-tap kernel on the same inputs (option off), and the routed entry point with the option on = this form, bit for bit
    assert L.get_option('wgrad_rowring') == 1
    routed = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
    L.set_option('wgrad_rowring', 0)
    ops.clear_plans()
    try:
        ref = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
    finally:
        L.set_option('wgrad_rowring', 1)
        ops.clear_plans()
    assert float((once.double() - ref.double()).norm() / ref.double().norm()) < 3e-6
    if N * H * W >= ops.BSPLIT_MIN_PIXELS:      # (smaller problems are routed to other kernels by the host)
        assert torch.equal(routed, once)
    # refusals: what the form does not take is said so, not computed wrongly
    for bad in (ops.make_desc(N, C, H, 24, K, 3, 3, 1, 1, 1), ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 0), ops.make_desc(N, 48, H, W, K, 3, 3, 1, 1, 1),
                ops.make_desc(N, C, H, W, 64, 3, 3, 1, 1, 1), ops.make_desc(N, C, H, W, K, 3, 3, 2, 1, 1), ops.make_desc(N, C, 2, W, K, 3, 3, 1, 1, 1)):
        assert not lib.pcgan_conv2d_wgrad_rowring_supported(ctypes.byref(bad))


@pytest.mark.parametrize('N,C,K,H,W', [(32, 256, 256, 32, 32), (3, 32, 128, 5, 16), (5, 64, 256, 7, 48), (1, 32, 128, 3, 16), (2, 128, 128, 9, 32)])
def test_weight_gradient_row_ring_form_bf16(dev, N, C, K, H, W):
    """bf16 tensors through rowring_wgrad_bf16_kernel (one bf16 product per tap, stored patterns straight to LDS / the A fragment, dy and x
    rows loaded four stages ahead, stage loop unrolled by four: heights that are not multiples of four run zero stages at the end): against
    float64 of the SAME bf16 inputs on a slice of output channels (bf16 x bf16 products are exact in fp32: only the accumulation order
    differs, 1e-6), against the per-tap kernel, and accumulation."""
    import ctypes
    from pcgan_amd.hip import ops, lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(N + C + K + H + W)
    x = torch.randn(N, C, H, W, generator=g).relu_().to(dev).to(torch.bfloat16)
    dy = (torch.randn(N, K, H, W, generator=g) * 0.05).to(dev).to(torch.bfloat16)
    d = ops.make_desc(N, C, H, W, K, 3, 3, 1, 1, 1, ops.BF16)
    assert lib.pcgan_conv2d_wgrad_rowring_supported(ctypes.byref(d))
    nb = int(lib.pcgan_conv2d_wgrad_rowring_workspace_bytes(ctypes.byref(d)))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    vp = ctypes.c_void_p
    dw = torch.full((K, C, 3, 3), 0.5, device=dev)
    for acc in (0, 1):
        L.check(lib.pcgan_conv2d_bwd_weight_rowring(ctypes.byref(d), vp(x.data_ptr()), None, 0, vp(dy.data_ptr()), None, 0, vp(dw.data_ptr()), acc,
                                                    vp(ws.data_ptr()), nb, vp(torch.cuda.current_stream().cuda_stream)), 'bwd_weight_rowring')
    torch.cuda.synchronize()
    once = dw / 2
    ks = sorted({0, 1, K // 3, K - 1})
    w = torch.zeros(len(ks), C, 3, 3, dtype=torch.float64, requires_grad=True)
    xp = torch.nn.functional.pad(x.double().cpu(), (1, 1, 1, 1), mode='reflect')
    for n0 in range(0, N, 8):
        torch.nn.functional.conv2d(xp[n0:n0 + 8], w).backward(dy[n0:n0 + 8, ks].double().cpu())
    err = float((once[ks].double().cpu() - w.grad).norm() / w.grad.norm())
    assert err < 1e-6, err
    L.set_option('wgrad_rowring', 0)
    ops.clear_plans()
    try:
        ref = ops.conv2d_bwd_weight(x, dy, (K, C, 3, 3), 1, 1, 1)
    finally:
        L.set_option('wgrad_rowring', 1)
        ops.clear_plans()
    assert float((once.double() - ref.double()).norm() / ref.double().norm()) < 1e-6
