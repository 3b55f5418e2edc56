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
