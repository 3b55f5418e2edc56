"""CPU: the C-ABI library loads and exports every symbol include/pcgan_hip.h declares, and the
ctypes signature table covers exactly that set (no compute calls: there is no GPU here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'pcgan_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pcgan_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_the_expected_families():
    names = _declared()
    for must in ('pcgan_conv2d_fwd', 'pcgan_conv2d_bwd_data', 'pcgan_conv2d_bwd_weight', 'pcgan_norm_act_fwd',
                 'pcgan_bce_loss', 'pcgan_adam_step', 'pcgan_bilinear_fwd', 'pcgan_maxpool_fwd'):
        assert must in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    from pcgan_amd.hip import lib
    handle = lib.load()
    for name in _declared():
        assert hasattr(handle, name), 'libpcgan_hip.so does not export %s' % name


def test_ctypes_table_matches_header():
    from pcgan_amd.hip import lib
    assert sorted(lib.SIGNATURES) == _declared()


def test_version_and_error_channel():
    from pcgan_amd.hip import lib
    handle = lib.load()
    assert handle.pcgan_version() >= 100
    assert handle.pcgan_last_error() is not None
    # argument validation happens before any launch, so it can be exercised without a GPU
    import ctypes
    d = lib.ConvDesc(1, 3, 8, 8, 4, 3, 3, 3, 1, 0, 8, 8)     # stride 3 is unsupported
    st = handle.pcgan_conv2d_fwd(ctypes.byref(d), None, None, None, None, 0, 0.0, None, 0, None)
    assert st != 0 and b'stride' in handle.pcgan_last_error()
    d = lib.ConvDesc(1, 3, 8, 8, 4, 3, 3, 1, 1, 0, 7, 7)     # wrong output size
    st = handle.pcgan_conv2d_fwd(ctypes.byref(d), None, None, None, None, 0, 0.0, None, 0, None)
    assert st != 0 and b'output dims' in handle.pcgan_last_error()
    assert handle.pcgan_conv2d_workspace_bytes(ctypes.byref(lib.ConvDesc(2, 4, 8, 8, 8, 3, 3, 1, 1, 0, 8, 8)), 0) >= 8 * 9 * 4 * 4


def test_weight_gradient_route_predicates():
    """host-side predicates of the matrix-pipe weight gradient (no launch): which shapes the route takes, which of them need no
    padded copy of x, and that the workspace of those holds whole split partials only (reference layers: models/networks.py:584-648,
    753-775)"""
    import ctypes
    from pcgan_amd.hip import lib
    h = lib.load()

    def desc(N, C, H, W, K, k, stride, pad, mode, dt):
        P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        d = lib.ConvDesc(N, C, H, W, K, k, k, stride, pad, mode, P, Q)
        d.dtype = dt
        return d

    for dt in (lib.F32, lib.BF16):
        # PatchGAN 256 -> 512, 4x4 stride 1 pad 1 at 16 x 16: output 15 x 15 (ragged stage), two row tiles, padding in the gather
        d = desc(32, 256, 16, 16, 512, 4, 1, 1, 0, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) == 1 and h.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d)) == 1
        ws = h.pcgan_conv2d_hsplit_wgrad_workspace_bytes(ctypes.byref(d))
        assert ws > 0 and ws % (512 * 256 * 16 * 4) == 0
        # the residual block's convolution: reflection applied in the gather
        d = desc(32, 256, 32, 32, 256, 3, 1, 1, 1, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) == 1 and h.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d)) == 1
        # the generator's first down-sampling layer (134 MB input with fp32 tensors): no padded copy
        d = desc(32, 64, 128, 128, 128, 3, 2, 1, 0, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d)) == 1
        # 7 x 7 maps: less than three quarters of a 16-column stage would be real columns
        d = desc(32, 512, 7, 7, 512, 3, 1, 1, 0, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) == 0 and h.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d)) == 0
        # the generator's stem: reflection padding 3 keeps the padded copy
        d = desc(32, 4, 128, 128, 64, 7, 1, 3, 1, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) == 1 and h.pcgan_conv2d_hsplit_wgrad_inline(ctypes.byref(d)) == 0
        # zero padding 2 (AlexNet's 5 x 5 layer): padded copy, so whole stages and at most 256 rows only
        d = desc(32, 64, 27, 27, 192, 5, 1, 2, 0, dt)
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(d)) == 0
        # fewer than 32 output channels / stride 4: other kernels
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(desc(32, 512, 15, 15, 1, 4, 1, 1, 0, dt))) == 0
        assert h.pcgan_conv2d_hsplit_wgrad_supported(ctypes.byref(desc(32, 3, 224, 224, 64, 11, 4, 2, 0, dt))) == 0


def test_library_reads_no_environment_and_options_are_explicit():
    """round 4: no getenv in the library's sources; the routing options are an explicit, validated table"""
    import glob
    from pcgan_amd.hip import lib
    for src in glob.glob(os.path.join(ROOT, 'pc-gan_amd', 'csrc', '*')):
        assert 'getenv' not in open(src).read(), src
    h = lib.load()
    for key, default in (('bsplit_halo', 1), ('wgrad_gen', 1), ('wgrad_padcopy', 0), ('wgrad_cw', 0), ('hgemm_bf16', 1), ('wgrad_direct', 0),
                         ('hgemm_tile', 0), ('hgemm_ks', 0), ('wgrad_rowring', 1)):
        if not any(os.environ.get(e) for e, k, _ in lib._ENV_OPTIONS if k == key):
            assert lib.get_option(key) == default, key
    lib.set_option('wgrad_cw', 256)
    assert lib.get_option('wgrad_cw') == 256
    lib.set_option('wgrad_cw', 0)
    assert h.pcgan_set_option(b'no_such_option', 1) != 0 and b'unknown option' in h.pcgan_last_error()
    assert h.pcgan_set_option(b'wgrad_cw', 77) != 0
    assert h.pcgan_set_option(b'hgemm_tile', 96096) != 0 and h.pcgan_set_option(b'hgemm_ks', 9) != 0
