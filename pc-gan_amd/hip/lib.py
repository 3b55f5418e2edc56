"""ctypes binding of libpcgan_hip.so (the C-ABI declared in include/pcgan_hip.h).

The library is the product path: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a tensor is not a contiguous fp32 / bf16 tensor on an AMD GPU the
call raises -- loudly -- instead of computing something else.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCGAN_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libpcgan_hip.so')

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PASS_FWD, PASS_BWD_DATA, PASS_BWD_WEIGHT = 0, 1, 2
F32, BF16 = 0, 1          # storage type of activation tensors (include/pcgan_hip.h)


class ImageDesc(ctypes.Structure):
    """pcgan_image_desc (include/pcgan_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in ('H', 'W', 'RH', 'RW', 'FH', 'FW', 'ksize_h', 'ksize_v', 'out_channels')]


class ResBlockDesc(ctypes.Structure):
    """pcgan_resblock_desc (include/pcgan_hip.h)."""
    _fields_ = [('N', ctypes.c_int), ('C', ctypes.c_int), ('H', ctypes.c_int), ('W', ctypes.c_int), ('eps', ctypes.c_float),
                ('momentum', ctypes.c_float), ('dtype', ctypes.c_int)]


class ConvDesc(ctypes.Structure):
    """pcgan_conv_desc (include/pcgan_hip.h)."""
    _fields_ = [(n, ctypes.c_int) for n in
                ('N', 'C', 'H', 'W', 'K', 'R', 'S', 'stride', 'pad', 'pad_mode', 'P', 'Q', 'dtype')]


_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_dp = ctypes.POINTER(ConvDesc)
_ip = ctypes.POINTER(ImageDesc)
_rp = ctypes.POINTER(ResBlockDesc)

# name -> (restype, argtypes); kept in one table so tests can check that the library
# exports every symbol the header declares.
SIGNATURES = {
    'pcgan_last_error': (ctypes.c_char_p, []),
    'pcgan_version': (_i, []),
    'pcgan_device_info': (_i, [ctypes.POINTER(_i), ctypes.c_char_p, _i]),
    'pcgan_conv2d_workspace_bytes': (_sz, [_dp, _i]),
    'pcgan_conv2d_fwd': (_i, [_dp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    'pcgan_conv2d_bwd_data': (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pcgan_conv2d_packed_bytes': (_sz, [_dp, _i]),
    'pcgan_conv2d_pack_weights': (_i, [_dp, _i, _vp, _vp, _vp]),
    'pcgan_conv2d_fwd_packed': (_i, [_dp, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    'pcgan_conv2d_bwd_data_packed': (_i, [_dp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pcgan_conv2d_bwd_weight': (_i, [_dp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'pcgan_channel_sum': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'pcgan_act_bwd': (_i, [_vp, _vp, _vp, _sz, _i, _f, _i, _vp]),
    'pcgan_act_fwd': (_i, [_vp, _vp, _sz, _i, _f, _i, _vp]),
    'pcgan_concat_z': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pcgan_add': (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    'pcgan_scale': (_i, [_vp, _vp, _f, _vp, _sz, _i, _vp]),
    'pcgan_cast': (_i, [_vp, _i, _vp, _i, _sz, _vp]),
    'pcgan_channel_scale': (_i, [_vp, _vp, _vp, _i, _i, _f, _i, _vp]),
    'pcgan_plane_stats': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'pcgan_bn_merge': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    'pcgan_in_running_update': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    'pcgan_norm_act_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_norm_bwd_stats': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_norm_bwd_apply': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_bn_bwd_reduce': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'pcgan_bn_stats_merged': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    'pcgan_bn_bwd_stats_reduced': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_bn_fwd_fused': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _f, _i, _vp]),
    'pcgan_bn_bwd_fused': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_instnorm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_instnorm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _f, _i, _vp]),
    'pcgan_instnorm_fused': (_i, [_i]),
    'pcgan_sum_planes': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'pcgan_maxpool_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'pcgan_maxpool_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'pcgan_global_pool_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pcgan_global_pool_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'pcgan_bilinear_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pcgan_bilinear_bwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'pcgan_bce_loss': (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz, _i, _vp]),
    'pcgan_l1_loss': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _vp, _sz, _i, _vp]),
    'pcgan_mse_loss': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _vp, _sz, _i, _vp]),
    'pcgan_loss_workspace_bytes': (_sz, [_sz]),
    'pcgan_adam_step': (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _vp]),
    'pcgan_adam_step_dev': (_i, [_vp, _vp, _vp, _vp, _sz, _vp, _vp, _f, _f, _f, _vp]),
    'pcgan_conv2d_bsplit_supported': (_i, [_dp]),
    'pcgan_conv2d_bsplit_packed_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bsplit_pack': (_i, [_dp, _vp, _vp, _vp]),
    'pcgan_conv2d_fwd_bsplit': (_i, [_dp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    'pcgan_conv2d_bsplit_dgrad_supported': (_i, [_dp]),
    'pcgan_conv2d_bsplit_dgrad_packed_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bsplit_dgrad_pack': (_i, [_dp, _vp, _vp, _vp]),
    'pcgan_conv2d_bwd_data_bsplit': (_i, [_dp, _vp, _vp, _vp, _vp]),
    'pcgan_conv2d_bsplit_wgrad_supported': (_i, [_dp]),
    'pcgan_conv2d_bsplit_wgrad_workspace_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bwd_weight_bsplit': (_i, [_dp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    'pcgan_absmax_slots': (_i, [_sz]),
    'pcgan_absmax': (_i, [_vp, _sz, _i, _vp, _i, _vp]),
    'pcgan_amax_audit': (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    'pcgan_conv2d_hsplit_supported': (_i, [_dp, _i]),
    'pcgan_conv2d_hsplit_packed_bytes': (_sz, [_dp, _i]),
    'pcgan_conv2d_hsplit_pack': (_i, [_dp, _i, _vp, _vp, _vp]),
    'pcgan_conv2d_fwd_hsplit': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _vp]),
    'pcgan_conv2d_bwd_data_hsplit': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp]),
    'pcgan_conv2d_bwd_data_hsplit_add': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    'pcgan_event_create': (_i, [ctypes.POINTER(_vp)]),
    'pcgan_event_destroy': (_i, [_vp]),
    'pcgan_resblock_supported': (_i, [_rp]),
    'pcgan_resblock_wgrad_workspace_bytes': (_sz, [_rp]),
    'pcgan_resblock_fwd': (_i, [_rp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'pcgan_resblock_bwd': (_i, [_rp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _sz, _vp, _vp, _vp]),
    'pcgan_set_nonfinite_counter': (_i, [_vp]),
    'pcgan_set_option': (_i, [ctypes.c_char_p, _i]),
    'pcgan_get_option': (_i, [ctypes.c_char_p, _vp]),
    'pcgan_timer_enable': (_i, [_i]),
    'pcgan_timer_read': (_i, [_i, _vp, _i]),
    'pcgan_conv2d_hgemm_supported': (_i, [_dp, _i]),
    'pcgan_conv2d_hgemm_pack': (_i, [_dp, _i, _vp, _vp, _vp, _vp]),
    'pcgan_conv2d_fwd_packed_hsplit': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _sz, _vp]),
    'pcgan_conv2d_bwd_data_packed_hsplit': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'pcgan_conv2d_thin_supported': (_i, [_dp, _i]),
    'pcgan_conv2d_thin_packed_bytes': (_sz, [_dp, _i]),
    'pcgan_conv2d_thin_workspace_bytes': (_sz, [_dp, _i]),
    'pcgan_conv2d_thin_pack': (_i, [_dp, _i, _vp, _vp, _vp]),
    'pcgan_conv2d_fwd_thin': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _i, _f, _vp]),
    'pcgan_conv2d_bwd_data_thin': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    'pcgan_conv2d_hsplit_wgrad_supported': (_i, [_dp]),
    'pcgan_conv2d_hsplit_wgrad_inline': (_i, [_dp]),
    'pcgan_conv2d_hsplit_wgrad_workspace_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bwd_weight_hsplit': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    'pcgan_restrunk_fwd': (_i, [_rp, _i, _vp, _vp, _i] + [_vp] * 8 + [_vp] * 6 + [_vp]),
    'pcgan_restrunk_bwd': (_i, [_rp, _i, _vp, _vp, _vp, _i] + [_vp] * 6 + [_vp] * 6 + [_vp] * 6 + [_vp, _sz, _vp, _vp, _vp]),
    'pcgan_conv2d_wgrad_direct_supported': (_i, [_dp]),
    'pcgan_conv2d_wgrad_rowring_supported': (_i, [_dp]),
    'pcgan_conv2d_wgrad_rowring_workspace_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bwd_weight_rowring': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    'pcgan_conv2d_wgrad_direct_workspace_bytes': (_sz, [_dp]),
    'pcgan_conv2d_bwd_weight_direct': (_i, [_dp, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    'pcgan_image_transform_band': (_i, [_ip, _vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    'pcgan_image_transform': (_i, [_ip, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'pcgan_amd: %s not found -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(or `make -C pc-gan_amd/csrc`). There is no fallback path.' % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64; the kernels must run in THAT runtime instance (same
    # device context, streams and allocations), so make sure it is the one already mapped before our
    # library's DT_NEEDED entry for libamdhip64.so.7 is resolved.
    import torch
    bundled = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
    if os.path.exists(bundled):
        ctypes.CDLL(bundled, mode=ctypes.RTLD_GLOBAL)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    _options_from_environment(lib)
    return lib


# The library itself reads no environment variables (include/pcgan_hip.h); the HOST maps the A/B switches that live below the C-ABI
# onto pcgan_set_option once, when the library is loaded.  (switch, option, value parser)
_ENV_OPTIONS = (('PCGAN_BSPLIT_HALO', 'bsplit_halo', int), ('PCGAN_WGRAD_GEN', 'wgrad_gen', int), ('PCGAN_WGRAD_PADCOPY', 'wgrad_padcopy', int),
                ('PCGAN_WGRAD_CW', 'wgrad_cw', int), ('PCGAN_HGEMM', 'hgemm_bf16', int), ('PCGAN_WGRAD_DIRECT', 'wgrad_direct', int),
                ('PCGAN_WGD_LOOK', 'wgd_look', int), ('PCGAN_WGRAD_ROWRING', 'wgrad_rowring', int))


def _options_from_environment(lib):
    for env, key, conv in _ENV_OPTIONS:
        v = os.environ.get(env)
        if v not in (None, ''):
            set_option(key, conv(v), lib)


def set_option(key, value, lib=None):
    """pcgan_set_option: a routing option below the C-ABI (see the header); shape plans made under another value are the caller's to
    drop (ops.clear_plans)"""
    lib = lib or load()
    if lib.pcgan_set_option(key.encode(), int(value)) != 0:
        msg = lib.pcgan_last_error()
        raise RuntimeError('pcgan_hip set_option failed: %s' % (msg.decode() if msg else '?'))


def get_option(key):
    v = ctypes.c_int(0)
    if load().pcgan_get_option(key.encode(), ctypes.byref(v)) != 0:
        raise RuntimeError('pcgan_hip get_option: unknown option %r' % key)
    return v.value


def check(status, what):
    if status != 0:
        msg = load().pcgan_last_error()
        raise RuntimeError('pcgan_hip %s failed (status %d): %s' % (what, status, msg.decode() if msg else '?'))
