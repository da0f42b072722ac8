"""ctypes binding of libstmi355.so (the C-ABI declared in include/stmi.h).

The library is built in-tree by ``style-transfer_amd/csrc/Makefile``.  There is
no CPU fallback: if the shared object is missing, importing this module's
``load()`` raises, and every launch needs a visible gfx950 device.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STMI_LIB_PATH') or os.path.join(_HERE, '..', 'libstmi355.so')   # STMI_LIB_PATH: tuning builds

F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
PACK_FWD, PACK_DGRAD, PACK_FWD_ROWOUT, PACK_FWD_RGB, PACK_DGRAD_RGB = 0, 1, 2, 3, 4
LOSS_SLOT_FLOATS, LOSS_TERMS_MAX = 1024, 8

_c_int, _c_long, _c_float, _vp = ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/stmi.h one to one
_CONV = [_c_int] * 8  # n, h, w, ci, co, kh, kw, stride
SIGNATURES = {
    'stmi_version': (_c_int, []),
    'stmi_last_error': (ctypes.c_char_p, []),
    'stmi_device_arch': (_c_int, [ctypes.c_char_p, _c_int]),
    'stmi_last_conv_kernel': (_c_int, [ctypes.c_char_p, _c_int]),
    'stmi_conv_packed_elems': (_c_long, [_c_int] * 5),
    'stmi_pack_conv_weights': (_c_int, [_vp, _vp] + [_c_int] * 6 + [_vp]),
    'stmi_pack_conv_weights_multi': (_c_int, [_vp, _vp, _vp, _c_int, _c_int, _vp]),
    'stmi_conv2d_nhwc_fwd': (_c_int, [_vp] * 5 + [_vp, _c_long] + _CONV + [_c_int, _c_int, _vp]),
    'stmi_conv2d_ksplit_ws_floats': (_c_long, _CONV + [_c_int, _c_int]),
    'stmi_conv2d_transpose_absorb_norm_ws_bytes': (_c_long, _CONV + [_c_int]),
    'stmi_conv2d_transpose_nhwc_fwd_absorb_norm': (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp] + _CONV + [_c_int, _vp]),
    'stmi_conv2d_to3_absorb_norm_ws_bytes': (_c_long, [_c_int] * 6),
    'stmi_conv2d_to3_nhwc_fwd_absorb_norm': (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_float, _vp, _vp, _vp, _vp] + [_c_int] * 6 + [_vp]),
    'stmi_conv2d_absorb_norm_ws_bytes': (_c_long, _CONV + [_c_int]),
    'stmi_conv2d_absorb_norm_tiles': (_c_int, _CONV + [_c_int]),
    'stmi_conv2d_nhwc_fwd_absorb_norm': (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp] + _CONV + [_c_int, _vp]),
    'stmi_conv2d_nhwc_fwd_pool': (_c_int, [_vp] * 5 + _CONV + [_c_int, _c_int, _vp]),
    'stmi_conv2d_tile_stats_tiles': (_c_int, [_c_int] * 9),
    'stmi_conv2d_tile_stats_free': (_c_int, [_c_int] * 9),
    'stmi_conv2d_nhwc_fwd_tile_stats': (_c_int, [_vp] * 6 + _CONV + [_c_int, _c_int, _vp]),
    'stmi_conv2d_rgb_nhwc_fwd': (_c_int, [_vp, _c_float] + [_vp] * 5 + [_c_int] * 7 + [_vp]),
    'stmi_conv2d_rgb_tile_stats_tiles': (_c_int, [_c_int] * 4),
    'stmi_conv2d_rgb_nhwc_fwd_tile_stats': (_c_int, [_vp, _c_float] + [_vp] * 4 + [_c_int] * 6 + [_vp]),
    'stmi_conv2d_rgb_u8_nhwc_fwd': (_c_int, [_vp, _c_float] + [_vp] * 4 + [_c_int] * 6 + [_vp]),
    'stmi_conv2d_rgb_nhwc_dgrad': (_c_int, [_vp] * 3 + [_c_int] * 6 + [_vp]),
    'stmi_conv2d_to3_supported': (_c_int, [_c_int] * 3),
    'stmi_conv2d_to3_nhwc_fwd': (_c_int, [_vp] * 4 + [_c_int] * 6 + [_vp]),
    'stmi_conv2d_nhwc_dgrad': (_c_int, [_vp] * 5 + [_c_float, _vp, _c_long] + _CONV + [_c_int, _vp]),
    'stmi_conv2d_dgrad_in_sums_tiles': (_c_int, [_c_int] * 9),
    'stmi_conv2d_nhwc_dgrad_in_sums': (_c_int, [_vp] * 5 + [_c_float] + [_vp] * 4 + [_c_int, _vp] + _CONV + [_c_int, _vp]),
    'stmi_last_wgrad_kernel': (_c_int, [ctypes.c_char_p, _c_int]),
    'stmi_conv2d_wgrad_ws_floats': (_c_long, _CONV + [_c_int]),
    'stmi_conv2d_nhwc_wgrad': (_c_int, [_vp] * 4 + _CONV + [_c_int, _c_int, _vp]),
    'stmi_conv2d_transpose_nhwc_fwd': (_c_int, [_vp] * 3 + _CONV + [_c_int, _vp]),
    'stmi_conv2d_transpose_tile_stats_tiles': (_c_int, _CONV + [_c_int]),
    'stmi_conv2d_transpose_nhwc_fwd_tile_stats': (_c_int, [_vp] * 4 + _CONV + [_c_int, _vp]),
    'stmi_conv2d_transpose_nhwc_dgrad': (_c_int, [_vp] * 4 + _CONV + [_c_int, _vp]),
    'stmi_conv2d_transpose_wgrad_ws_floats': (_c_long, _CONV + [_c_int]),
    'stmi_conv2d_transpose_nhwc_wgrad': (_c_int, [_vp] * 4 + _CONV + [_c_int, _c_int, _vp]),
    'stmi_instance_norm_ws_floats': (_c_long, [_c_int] * 4),
    'stmi_instance_norm_nhwc_stats': (_c_int, [_vp, _vp, _vp, _c_float] + [_c_int] * 5 + [_vp]),
    'stmi_instance_norm_nhwc_stats_static': (_c_int, [_vp, _vp, _c_float] + [_c_int] * 5 + [_vp]),
    'stmi_instance_norm_nhwc_apply': (_c_int, [_vp] * 7 + [_c_int] * 6 + [_vp]),
    'stmi_instance_norm_nhwc_apply_u8': (_c_int, [_vp] * 5 + [_c_int] * 6 + [_vp]),
    'stmi_instance_norm_nhwc_fwd': (_c_int, [_vp, _vp, _vp, _c_float] + [_vp] * 5 + [_c_int] * 6 + [_vp]),
    'stmi_instance_norm_nhwc_apply_tiles': (_c_int, [_vp, _vp, _c_int, _vp, _c_float] + [_vp] * 5 + [_c_int] * 6 + [_vp]),
    'stmi_instance_norm_nhwc_bwd': (_c_int, [_vp] * 8 + [_c_int, _vp, _c_float] + [_c_int] * 6 + [_vp]),
    'stmi_instance_norm_nhwc_bwd_rows': (_c_int, [_vp] * 8 + [_c_int, _vp] + [_c_int] * 7 + [_vp]),
    'stmi_instance_norm_bwd_rows_count': (_c_int, [_c_int] * 5),
    'stmi_instance_norm_param_grads': (_c_int, [_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp]),
    'stmi_scale_shift': (_c_int, [_vp, _vp, _vp, _c_float, _c_long, _c_int, _c_int, _vp]),
    'stmi_maxpool2x2_same_fwd': (_c_int, [_vp, _vp] + [_c_int] * 5 + [_vp]),
    'stmi_maxpool2x2_same_bwd': (_c_int, [_vp, _vp, _vp] + [_c_int] * 5 + [_vp]),
    'stmi_gram_ws_floats': (_c_long, [_c_int] * 5),
    'stmi_gram_nhwc_fwd': (_c_int, [_vp, _vp, _vp] + [_c_int] * 5 + [_vp]),
    'stmi_gram_nhwc_fwd_add': (_c_int, [_vp, _vp, _vp] + [_c_int] * 5 + [_vp]),
    'stmi_zero_multi': (_c_int, [_vp, _vp, _c_int, _vp, _vp]),
    'stmi_gram_packed_elems': (_c_long, [_c_int, _c_int]),
    'stmi_gram_nhwc_bwd': (_c_int, [_vp, _vp, _vp] + [_c_int] * 6 + [_vp]),
    'stmi_gram_mse_fwd_bwd': (_c_int, [_vp] * 4 + [_c_float, _c_float, _c_int, _c_int, _c_int, _vp]),
    'stmi_mse_fwd_bwd': (_c_int, [_vp] * 4 + [_c_float, _c_float, _c_long, _c_int, _vp]),
    'stmi_tv_loss_fwd_bwd': (_c_int, [_vp] * 3 + [_c_float] + [_c_int] * 5 + [_vp]),
    'stmi_loss_slots_sum': (_c_int, [_vp, _vp, _c_int, _vp, _vp]),
    'stmi_adam_tf1': (_c_int, [_vp] * 4 + [_c_long] + [_c_float] * 5 + [_vp, _c_int, _vp]),
    'stmi_cast': (_c_int, [_vp, _vp, _c_long, _c_int, _c_int, _vp]),
    'stmi_pixels_to_rows': (_c_int, [_vp, _vp] + [_c_int] * 7 + [_vp]),
    'stmi_rows_to_pixels_sum': (_c_int, [_vp, _vp, _vp] + [_c_int] * 7 + [_vp]),
    'stmi_rows_wgrad_to_hwio': (_c_int, [_vp, _vp] + [_c_int] * 5 + [_vp]),
    'stmi_activation': (_c_int, [_vp, _vp, _c_long, _c_int, _c_int, _vp]),
    'stmi_clip': (_c_int, [_vp, _c_long, _c_float, _c_float, _vp]),
}



class PackDesc(ctypes.Structure):
    """stmi_pack_desc of include/stmi.h"""
    _fields_ = [('src_off', _c_long), ('dst_off', _c_long), ('kh', _c_int), ('kw', _c_int), ('ci', _c_int),
                ('co', _c_int), ('mode', _c_int), ('reserved', _c_int)]


_lib = None


class StmiError(RuntimeError):
    pass


def load():
    """Loads the shared library once and sets every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.abspath(LIB_PATH)
    if not os.path.exists(path):
        raise StmiError('libstmi355.so not found at %s; build it with `make -C style-transfer_amd/csrc` '
                        '(there is no CPU fallback)' % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def call(name, *args):
    """Calls an int-returning entry point and raises StmiError with the
    library's message on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise StmiError('%s failed (%d): %s' % (name, rc, lib.stmi_last_error().decode()))
    return rc
