"""Torch-tensor front end of the C-ABI: shape/dtype checks, output allocation,
current-stream plumbing.  Every function launches HIP kernels from
libstmi355.so; nothing here computes on the host and nothing falls back to
torch operators.

Tensors are NHWC, contiguous, on the current HIP device, float32 or bfloat16.
"""
import torch

from . import lib as _lib
from .lib import ACT_NONE, ACT_RELU, ACT_SIGMOID, PACK_DGRAD, PACK_DGRAD_RGB, PACK_FWD, PACK_FWD_RGB, PACK_FWD_ROWOUT  # noqa: F401

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError('stmi ops take float32 or bfloat16 tensors, got %s' % t.dtype)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name, dtype=None, shape=None):
    if not t.is_cuda:
        raise _lib.StmiError('%s must live on the GPU: the HIP kernels have no CPU fallback' % name)
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous' % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError('%s has shape %s, expected %s' % (name, tuple(t.shape), tuple(shape)))
    return t


def same_out(size, stride):
    return -(-size // stride)


# Optional launch log for bench.py's roofline leg: when PROFILE is a list, every stride-1 conv
# forward / backward-data call appends (name of the kernel the library launched, algorithmic
# flops, a closure that launches the same call again on the then-current stream).
PROFILE = None
# The same for the bandwidth-bound calls (instance norm): (call name, algorithmic bytes, relaunch closure).
PROFILE_HBM = None


def last_wgrad_kernel():
    """Name of the kernel the most recent weight-gradient / Gram call launched."""
    import ctypes
    buf = ctypes.create_string_buffer(64)
    _lib.call('stmi_last_wgrad_kernel', buf, 64)
    return buf.value.decode()


def last_conv_kernel():
    """Profiler name of the kernel the most recent conv forward / backward-data call launched."""
    import ctypes
    buf = ctypes.create_string_buffer(256)
    _lib.call('stmi_last_conv_kernel', buf, 256)
    return buf.value.decode()


def _conv_call(entry, args, out_channels, out_pixels, k_channels, taps, stride):
    """Launches a conv entry point (the stream is appended to args) and logs it when asked to."""
    _lib.call(entry, *args, _stream())
    if PROFILE is not None and stride == 1:
        PROFILE.append((last_conv_kernel(), 2.0 * out_channels * out_pixels * k_channels * taps,
                        lambda: _lib.call(entry, *args, _stream())))


# ---------------------------------------------------------------- weights
def packed_elems(kh, kw, ci, co, mode):
    return _lib.load().stmi_conv_packed_elems(kh, kw, ci, co, mode)


def pack_conv_weights(w_hwio, mode, dtype, out=None):
    """fp32 HWIO [kh,kw,ci,co] -> K-contiguous zero-padded copy in `dtype`."""
    _chk(w_hwio, 'w_hwio', torch.float32)
    kh, kw, ci, co = w_hwio.shape
    n = packed_elems(kh, kw, ci, co, mode)
    if out is None:
        out = torch.empty(n, dtype=dtype, device=w_hwio.device)
    _chk(out, 'packed', dtype, (n,))
    _lib.call('stmi_pack_conv_weights', w_hwio.data_ptr(), out.data_ptr(), kh, kw, ci, co, mode, _dt(out), _stream())
    return out


def make_pack_table(entries):
    """entries: iterable of (src_off, dst_off, (kh, kw, ci, co), mode) -> ctypes table for pack_conv_weights_multi."""
    entries = list(entries)
    table = (_lib.PackDesc * len(entries))()
    for d, (src, dst, (kh, kw, ci, co), mode) in zip(table, entries):
        d.src_off, d.dst_off, d.kh, d.kw, d.ci, d.co, d.mode = src, dst, kh, kw, ci, co, mode
    return table


def pack_conv_weights_multi(src_flat, dst_flat, table):
    """All tensors of `table` in one launch: fp32 flat buffer -> packed flat buffer."""
    _chk(src_flat, 'src_flat', torch.float32)
    _chk(dst_flat, 'dst_flat')
    _lib.call('stmi_pack_conv_weights_multi', src_flat.data_ptr(), dst_flat.data_ptr(), table, len(table),
              _dt(dst_flat), _stream())


# ------------------------------------------------------------------- conv
def conv_ksplit_ws_floats(x_shape, kshape, stride, dtype, dgrad=False):
    """Floats of workspace with which conv2d_fwd (conv2d_dgrad when `dgrad`; x_shape is then the shape of dx) splits
    the K loop of this geometry over several blocks per tile; 0: it never does."""
    kh, kw, ci, co = kshape
    n, h, w, _ = x_shape
    return int(_lib.load().stmi_conv2d_ksplit_ws_floats(n, h, w, ci, co, kh, kw, stride, _DT[dtype], int(dgrad)))


def conv2d_fwd(x, w_packed, kshape, stride, bias=None, relu=False, want_pre=False, out=None, out_pre=None,
               tile_stats=None, ws=None):
    """kshape = (kh, kw, ci, co) of the HWIO weight that was packed with PACK_FWD.
    ws: optional fp32 workspace; layers with too few tiles to fill the chip then split their K loop (see
    conv_ksplit_ws_floats).
    tile_stats: fp32 [n, conv_tile_stats_tiles(...), 4, co] that receives per-tile sums of the output for
    instance_norm_fwd(..., tile_stats=)."""
    kh, kw, ci, co = kshape
    n, h, w, c = x.shape
    assert c == ci, (c, ci)
    _chk(x, 'x')
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(kh, kw, ci, co, PACK_FWD),))
    oshape = (n, same_out(h, stride), same_out(w, stride), co)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    pre = None
    if want_pre:
        pre = torch.empty_like(y) if out_pre is None else _chk(out_pre, 'out_pre', x.dtype, oshape)
    if bias is not None:
        _chk(bias, 'bias', torch.float32, (co,))
    if tile_stats is not None:
        ts = conv_tile_stats_tiles(x.shape, kshape, stride, x.dtype)
        _chk(tile_stats, 'tile_stats', torch.float32, (n, ts, 4, co))
        _conv_call('stmi_conv2d_nhwc_fwd_tile_stats', (x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(),
                                                       _ptr(pre), tile_stats.data_ptr(), n, h, w, ci, co, kh, kw, stride,
                                                       int(relu), _dt(x)),
                   co, n * oshape[1] * oshape[2], ci, kh * kw, stride)
        return (y, pre) if want_pre else y
    if ws is not None:
        _chk(ws, 'ws', torch.float32)
    _conv_call('stmi_conv2d_nhwc_fwd', (x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(pre),
                                        _ptr(ws), 0 if ws is None else ws.numel(),
                                        n, h, w, ci, co, kh, kw, stride, int(relu), _dt(x)),
               co, n * oshape[1] * oshape[2], ci, kh * kw, stride)
    return (y, pre) if want_pre else y


def conv2d_fwd_pool(x, w_packed, kshape, stride, bias=None, relu=True, out=None, out_pool=None):
    """conv2d_fwd that also leaves maxpool_fwd(y) (vgg.py:99-103); 3x3 stride-1 layers of >= 128 bytes of channels
    pool inside the convolution's launch (last_conv_kernel() then ends in ' [+pool]').  Returns (y, pooled)."""
    kh, kw, ci, co = kshape
    n, h, w, c = x.shape
    assert c == ci, (c, ci)
    _chk(x, 'x')
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(kh, kw, ci, co, PACK_FWD),))
    oshape = (n, same_out(h, stride), same_out(w, stride), co)
    pshape = (n, (oshape[1] + 1) // 2, (oshape[2] + 1) // 2, co)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    yp = torch.empty(pshape, dtype=x.dtype, device=x.device) if out_pool is None else _chk(out_pool, 'out_pool', x.dtype, pshape)
    if bias is not None:
        _chk(bias, 'bias', torch.float32, (co,))
    _conv_call('stmi_conv2d_nhwc_fwd_pool', (x.data_ptr(), w_packed.data_ptr(), _ptr(bias), y.data_ptr(), yp.data_ptr(),
                                             n, h, w, ci, co, kh, kw, stride, int(relu), _dt(x)),
               co, n * oshape[1] * oshape[2], ci, kh * kw, stride)
    return y, yp


def conv_absorb_norm_ws_bytes(x_shape, kshape, stride, dtype):
    """Bytes of workspace conv2d_fwd_absorb_norm needs for this geometry; 0: not supported."""
    n, h, w, _ = x_shape
    kh, kw, ci, co = kshape
    return int(_lib.load().stmi_conv2d_absorb_norm_ws_bytes(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv_absorb_norm_tiles(x_shape, kshape, stride, dtype):
    """Rows per sample of the tile sums conv2d_fwd_absorb_norm leaves for the norm behind it."""
    n, h, w, _ = x_shape
    kh, kw, ci, co = kshape
    return int(_lib.load().stmi_conv2d_absorb_norm_tiles(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv2d_fwd_absorb_norm(x, in_tile_stats, in_scale, in_shift, w_packed, kshape, stride, tile_stats, ws, in_stats=None,
                           out=None, eps=None):
    """conv3x3(relu(IN(x))): x is the raw output of the convolution in front (in_tile_stats: the tile sums it left),
    in_scale (> 0) / in_shift the absorbed norm's parameters; the normalised tensor is never written.
    Returns (y, in_stats); tile_stats receives the tile sums of y for the norm behind it."""
    n, h, w, c = x.shape
    kh, kw, ci, co = kshape
    assert c == ci, (c, ci)
    _chk(x, 'x')
    _chk(in_tile_stats, 'in_tile_stats', torch.float32)
    assert in_tile_stats.dim() == 4 and in_tile_stats.shape[0] == n and in_tile_stats.shape[2:] == (4, ci), tuple(in_tile_stats.shape)
    _chk(in_scale, 'in_scale', torch.float32, (ci,))
    _chk(in_shift, 'in_shift', torch.float32, (ci,))
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(kh, kw, ci, co, PACK_FWD),))
    need = conv_absorb_norm_ws_bytes(x.shape, kshape, stride, x.dtype)
    if need == 0:
        raise ValueError('conv2d_fwd_absorb_norm: geometry not supported')
    _chk(ws, 'ws', torch.uint8)
    if ws.numel() < need:
        raise ValueError('conv2d_fwd_absorb_norm: workspace of %d bytes, %d needed' % (ws.numel(), need))
    ts = conv_absorb_norm_tiles(x.shape, kshape, stride, x.dtype)
    _chk(tile_stats, 'tile_stats', torch.float32, (n, ts, 4, co))
    oshape = (n, same_out(h, stride), same_out(w, stride), co)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    st = torch.empty((n, ci, 2), dtype=torch.float32, device=x.device) if in_stats is None else _chk(in_stats, 'in_stats', torch.float32, (n, ci, 2))
    _conv_call('stmi_conv2d_nhwc_fwd_absorb_norm', (x.data_ptr(), in_tile_stats.data_ptr(), in_tile_stats.shape[1],
                                                    in_scale.data_ptr(), in_shift.data_ptr(),
                                                    float(IN_EPS if eps is None else eps), st.data_ptr(),
                                                    w_packed.data_ptr(), y.data_ptr(), tile_stats.data_ptr(), ws.data_ptr(),
                                                    n, h, w, ci, co, kh, kw, stride, _dt(x)),
               co, n * oshape[1] * oshape[2], ci, kh * kw, stride)
    return y, st


def conv_tile_stats_free(x_shape, kshape, stride, dtype):
    """True when the statistics rows of this geometry cost the convolution nothing (the persistent kernel)."""
    n, h, w, ci = x_shape
    kh, kw, _, co = kshape
    return bool(_lib.load().stmi_conv2d_tile_stats_free(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv_tile_stats_tiles(x_shape, kshape, stride, dtype):
    """Tiles per sample of the statistics conv2d_fwd(..., tile_stats=) emits for this geometry; 0 when the
    kernel that runs it emits none (then instance_norm_fwd makes its own statistics pass)."""
    kh, kw, ci, co = kshape
    n, h, w, _ = x_shape
    return _lib.load().stmi_conv2d_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, _DT[dtype])


RGB_CONFIGS = ((9, 32), (3, 64), (3, 32))  # (k, co) the 3-channel kernel is built for


def conv_rgb_tile_stats_tiles(h, w, co, k):
    """Statistics rows per sample conv2d_rgb_fwd(tile_stats=...) leaves (0: none for this (k, co))."""
    return int(_lib.load().stmi_conv2d_rgb_tile_stats_tiles(h, w, co, k))


def conv2d_rgb_fwd(x_f32, w_packed, k, co, dtype, scale=1.0, mean=None, bias=None, relu=False, want_pre=False,
                   out=None, out_pre=None, tile_stats=None):
    """k x k stride-1 SAME convolution of an fp32 [n,h,w,3] image, computed in `dtype` on
    x * scale - mean[c] (the cast / VGG preprocessing happens while the image is read).
    w_packed: the [k,k,3,co] weight packed with PACK_FWD_RGB.
    tile_stats: fp32 [n, conv_rgb_tile_stats_tiles(...), 4, co] that receives the per-tile sums of y for
    instance_norm_fwd(tile_stats=...); without bias, ReLU and pre-activation copy."""
    n, h, w, c = x_f32.shape
    assert c == 3 and (k, co) in RGB_CONFIGS, (c, k, co)
    _chk(x_f32, 'x_f32', torch.float32)
    _chk(w_packed, 'w_packed', dtype, (packed_elems(k, k, 3, co, PACK_FWD_RGB),))
    oshape = (n, h, w, co)
    y = torch.empty(oshape, dtype=dtype, device=x_f32.device) if out is None else _chk(out, 'out', dtype, oshape)
    pre = None
    if want_pre:
        pre = torch.empty_like(y) if out_pre is None else _chk(out_pre, 'out_pre', dtype, oshape)
    if bias is not None:
        _chk(bias, 'bias', torch.float32, (co,))
    if mean is not None:
        _chk(mean, 'mean', torch.float32, (3,))
    if tile_stats is not None:
        assert bias is None and not relu and not want_pre, 'tile statistics: plain convolution only'
        _chk(tile_stats, 'tile_stats', torch.float32, (n, conv_rgb_tile_stats_tiles(h, w, co, k), 4, co))
        _conv_call('stmi_conv2d_rgb_nhwc_fwd_tile_stats',
                   (x_f32.data_ptr(), float(scale), _ptr(mean), w_packed.data_ptr(), y.data_ptr(), tile_stats.data_ptr(),
                    n, h, w, co, k, _dt(y)), co, n * h * w, 3, k * k, 1)
        return y
    _conv_call('stmi_conv2d_rgb_nhwc_fwd', (x_f32.data_ptr(), float(scale), _ptr(mean), w_packed.data_ptr(), _ptr(bias),
                                            y.data_ptr(), _ptr(pre), n, h, w, co, k, int(relu), _dt(y)),
               co, n * h * w, 3, k * k, 1)
    return (y, pre) if want_pre else y


def conv2d_rgb_u8_fwd(x_u8, w_packed, k, co, dtype, scale=1.0 / 255.0, mean=None, out=None, tile_stats=None):
    """conv2d_rgb_fwd on 8-bit frames [n,h,w,3] (torch.uint8): conv(x * scale - mean[c]) in `dtype`; plain
    convolution (no bias / ReLU), built for (k, co) = (9, 32).  tile_stats as for conv2d_rgb_fwd."""
    n, h, w, c = x_u8.shape
    assert c == 3 and (k, co) == RGB_CONFIGS[0], (c, k, co)
    _chk(x_u8, 'x_u8', torch.uint8)
    _chk(w_packed, 'w_packed', dtype, (packed_elems(k, k, 3, co, PACK_FWD_RGB),))
    oshape = (n, h, w, co)
    y = torch.empty(oshape, dtype=dtype, device=x_u8.device) if out is None else _chk(out, 'out', dtype, oshape)
    if mean is not None:
        _chk(mean, 'mean', torch.float32, (3,))
    if tile_stats is not None:
        _chk(tile_stats, 'tile_stats', torch.float32, (n, conv_rgb_tile_stats_tiles(h, w, co, k), 4, co))
    _conv_call('stmi_conv2d_rgb_u8_nhwc_fwd', (x_u8.data_ptr(), float(scale), _ptr(mean), w_packed.data_ptr(), y.data_ptr(),
                                               _ptr(tile_stats), n, h, w, co, k, _dt(y)), co, n * h * w, 3, k * k, 1)
    return y


def conv_transpose_absorb_norm_ws_bytes(x_shape, kshape, stride, dtype):
    """Bytes of workspace conv2d_transpose_fwd_absorb_norm needs; 0: not supported."""
    kh, kw, co, ci = kshape
    n, h, w, _ = x_shape
    return int(_lib.load().stmi_conv2d_transpose_absorb_norm_ws_bytes(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv2d_transpose_fwd_absorb_norm(x, in_tile_stats, in_scale, in_shift, w_packed, kshape, stride, tile_stats, ws,
                                     in_stats=None, out=None, eps=None):
    """conv2d_transpose_fwd(relu(IN(x))) with x the raw output of the layer in front (in_tile_stats: the tile sums it
    left): the normalised tensor is never written.  Returns (y, in_stats); tile_stats receives the tile sums of y."""
    kh, kw, co, ci = kshape
    n, h, w, c = x.shape
    assert c == ci, (c, ci)
    _chk(x, 'x')
    _chk(in_tile_stats, 'in_tile_stats', torch.float32)
    assert in_tile_stats.dim() == 4 and in_tile_stats.shape[0] == n and in_tile_stats.shape[2:] == (4, ci), tuple(in_tile_stats.shape)
    _chk(in_scale, 'in_scale', torch.float32, (ci,))
    _chk(in_shift, 'in_shift', torch.float32, (ci,))
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(kh, kw, co, ci, PACK_DGRAD),))
    need = conv_transpose_absorb_norm_ws_bytes(x.shape, kshape, stride, x.dtype)
    if need == 0:
        raise ValueError('conv2d_transpose_fwd_absorb_norm: geometry not supported')
    _chk(ws, 'ws', torch.uint8)
    if ws.numel() < need:
        raise ValueError('conv2d_transpose_fwd_absorb_norm: workspace of %d bytes, %d needed' % (ws.numel(), need))
    ts = conv_transpose_tile_stats_tiles(x.shape, kshape, stride, x.dtype)
    oshape = (n, h * stride, w * stride, co)
    _chk(tile_stats, 'tile_stats', torch.float32, (n, ts, 4, co))
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    st = torch.empty((n, ci, 2), dtype=torch.float32, device=x.device) if in_stats is None else _chk(in_stats, 'in_stats', torch.float32, (n, ci, 2))
    _lib.call('stmi_conv2d_transpose_nhwc_fwd_absorb_norm', x.data_ptr(), in_tile_stats.data_ptr(), in_tile_stats.shape[1],
              in_scale.data_ptr(), in_shift.data_ptr(), float(IN_EPS if eps is None else eps), st.data_ptr(),
              w_packed.data_ptr(), y.data_ptr(), tile_stats.data_ptr(), ws.data_ptr(), n, h, w, ci, co, kh, kw, stride,
              _dt(x), _stream())
    return y, st


def conv_to3_absorb_norm_ws_bytes(x_shape, k, dtype):
    """Bytes of workspace conv2d_to3_absorb_norm needs; 0: not supported."""
    n, h, w, ci = x_shape
    return int(_lib.load().stmi_conv2d_to3_absorb_norm_ws_bytes(n, h, w, ci, k, _DT[dtype]))


def conv2d_to3_absorb_norm(x, in_tile_stats, in_scale, in_shift, w_packed, k, ws, in_stats=None, out=None, eps=None):
    """conv2d_to3(relu(IN(x))) with x the raw output of the layer in front (in_tile_stats: the tile sums it left):
    the normalised tensor is never written.  Returns (y, in_stats)."""
    n, h, w, ci = x.shape
    _chk(x, 'x')
    _chk(in_tile_stats, 'in_tile_stats', torch.float32)
    assert in_tile_stats.dim() == 4 and in_tile_stats.shape[0] == n and in_tile_stats.shape[2:] == (4, ci), tuple(in_tile_stats.shape)
    _chk(in_scale, 'in_scale', torch.float32, (ci,))
    _chk(in_shift, 'in_shift', torch.float32, (ci,))
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(k, k, ci, 3, PACK_FWD_ROWOUT),))
    need = conv_to3_absorb_norm_ws_bytes(x.shape, k, x.dtype)
    if need == 0:
        raise ValueError('conv2d_to3_absorb_norm: geometry not supported')
    _chk(ws, 'ws', torch.uint8)
    if ws.numel() < need:
        raise ValueError('conv2d_to3_absorb_norm: workspace of %d bytes, %d needed' % (ws.numel(), need))
    y = torch.empty((n, h, w, 3), dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, (n, h, w, 3))
    st = torch.empty((n, ci, 2), dtype=torch.float32, device=x.device) if in_stats is None else _chk(in_stats, 'in_stats', torch.float32, (n, ci, 2))
    _conv_call('stmi_conv2d_to3_nhwc_fwd_absorb_norm', (x.data_ptr(), in_tile_stats.data_ptr(), in_tile_stats.shape[1],
                                                        in_scale.data_ptr(), in_shift.data_ptr(),
                                                        float(IN_EPS if eps is None else eps), st.data_ptr(),
                                                        w_packed.data_ptr(), y.data_ptr(), ws.data_ptr(), n, h, w, ci, k,
                                                        _dt(x)), 3, n * h * w, ci, k * k, 1)
    return y, st


def to3_supported(k, ci, dtype):
    return bool(_lib.load().stmi_conv2d_to3_supported(k, ci, _DT[dtype]))


def conv2d_to3(x, w_packed, k, addend=None, out=None):
    """k x k stride-1 SAME conv ci -> 3 in one launch: conv2d_fwd with PACK_FWD_ROWOUT weights and
    rows_to_pixels_sum fused (the (kx, co) intermediate stays in LDS, in fp32)."""
    n, h, w, ci = x.shape
    _chk(x, 'x')
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(k, k, ci, 3, PACK_FWD_ROWOUT),))
    oshape = (n, h, w, 3)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    if addend is not None:
        _chk(addend, 'addend', x.dtype, oshape)
    _conv_call('stmi_conv2d_to3_nhwc_fwd', (x.data_ptr(), w_packed.data_ptr(), _ptr(addend), y.data_ptr(), n, h, w, ci, k,
                                            _dt(x)), 3, n * h * w, ci, k * k, 1)
    return y


def conv2d_rgb_dgrad(dy, w_packed, k, ci, out=None):
    """Gradient w.r.t. the input of a k x k stride-1 conv ci -> 3: dy [n,h,w,3] -> dx [n,h,w,ci];
    w_packed: the [k,k,ci,3] weight packed with PACK_DGRAD_RGB."""
    n, h, w, c = dy.shape
    assert c == 3 and (k, ci) in RGB_CONFIGS[:1], (c, k, ci)
    _chk(dy, 'dy')
    _chk(w_packed, 'w_packed', dy.dtype, (packed_elems(k, k, ci, 3, PACK_DGRAD_RGB),))
    oshape = (n, h, w, ci)
    dx = torch.empty(oshape, dtype=dy.dtype, device=dy.device) if out is None else _chk(out, 'out', dy.dtype, oshape)
    _conv_call('stmi_conv2d_rgb_nhwc_dgrad', (dy.data_ptr(), w_packed.data_ptr(), dx.data_ptr(), n, h, w, ci, k, _dt(dy)),
               ci, n * h * w, 3, k * k, 1)
    return dx


def conv2d_dgrad(dy, w_packed, kshape, stride, in_hw, relu_mask=None, addend=None, alpha=1.0, out=None, in_bwd=None,
                 ws=None):
    """Gradient w.r.t. the conv input of spatial size in_hw; W packed with PACK_DGRAD.
    in_bwd = (x, stats, scale, shift, act, rows): the instance norm whose OUTPUT gradient this call writes;
    rows fp32 [n, conv_dgrad_in_sums_tiles(...), 2, ci] receives the sums instance_norm_bwd(..., rows=) folds."""
    kh, kw, ci, co = kshape
    h, w = in_hw
    n = dy.shape[0]
    _chk(dy, 'dy', shape=(n, same_out(h, stride), same_out(w, stride), co))
    _chk(w_packed, 'w_packed', dy.dtype, (packed_elems(kh, kw, ci, co, PACK_DGRAD),))
    oshape = (n, h, w, ci)
    dx = torch.empty(oshape, dtype=dy.dtype, device=dy.device) if out is None else _chk(out, 'out', dy.dtype, oshape)
    for t, nm in ((relu_mask, 'relu_mask'), (addend, 'addend')):
        if t is not None:
            _chk(t, nm, dy.dtype, oshape)
    if in_bwd is not None:
        bx, bstats, bscale, bshift, bact, rows = in_bwd
        _chk(bx, 'in_bwd x', dy.dtype, oshape)
        _chk(bstats, 'in_bwd stats', torch.float32, (n, ci, 2))
        _chk(bscale, 'in_bwd scale', torch.float32, (ci,))
        _chk(bshift, 'in_bwd shift', torch.float32, (ci,))
        _chk(rows, 'in_bwd rows', torch.float32, (n, conv_dgrad_in_sums_tiles(in_hw, n, kshape, stride, dy.dtype), 2, ci))
        _conv_call('stmi_conv2d_nhwc_dgrad_in_sums',
                   (dy.data_ptr(), w_packed.data_ptr(), dx.data_ptr(), _ptr(relu_mask), _ptr(addend), float(alpha),
                    bx.data_ptr(), bstats.data_ptr(), bscale.data_ptr(), bshift.data_ptr(), int(bact), rows.data_ptr(),
                    n, h, w, ci, co, kh, kw, stride, _dt(dy)), ci, n * h * w, co, kh * kw, stride)
        return dx
    if ws is not None:
        _chk(ws, 'ws', torch.float32)
    _conv_call('stmi_conv2d_nhwc_dgrad', (dy.data_ptr(), w_packed.data_ptr(), dx.data_ptr(), _ptr(relu_mask),
                                          _ptr(addend), float(alpha), _ptr(ws), 0 if ws is None else ws.numel(),
                                          n, h, w, ci, co, kh, kw, stride, _dt(dy)),
               ci, n * h * w, co, kh * kw, stride)
    return dx


def conv_dgrad_in_sums_tiles(in_hw, n, kshape, stride, dtype):
    """Rows per sample of the instance-norm sums conv2d_dgrad(..., in_bwd=) leaves for this geometry; 0: none."""
    kh, kw, ci, co = kshape
    return _lib.load().stmi_conv2d_dgrad_in_sums_tiles(n, in_hw[0], in_hw[1], ci, co, kh, kw, stride, _DT[dtype])


def _ws(ws, need, device):
    """The workspace a call uses: the caller's (checked) or a fresh one of `need` floats; None when need == 0."""
    if need == 0:
        return None
    if ws is None:
        return torch.empty(need, dtype=torch.float32, device=device)
    _chk(ws, 'ws', torch.float32)
    if ws.numel() < need:
        raise ValueError('ws holds %d floats, this call needs %d' % (ws.numel(), need))
    return ws


def conv2d_wgrad_ws_floats(x_shape, kshape, stride, dtype):
    """Floats of workspace conv2d_wgrad needs for the partial sums of its pixel slices (0: none)."""
    kh, kw, ci, co = kshape
    n, h, w, _ = x_shape
    return int(_lib.load().stmi_conv2d_wgrad_ws_floats(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv2d_wgrad(x, dy, kshape, stride, out=None, accumulate=False, ws=None):
    """dW fp32 [kh,kw,ci,co]; overwritten, or added to `out` when accumulate.  Bit-reproducible (no atomics):
    pixel slices leave partial sums in `ws`, folded in a fixed order."""
    kh, kw, ci, co = kshape
    n, h, w, c = x.shape
    assert c == ci
    _chk(x, 'x')
    _chk(dy, 'dy', x.dtype, (n, same_out(h, stride), same_out(w, stride), co))
    dw = torch.empty(kshape, dtype=torch.float32, device=x.device) if out is None else _chk(out, 'out', torch.float32, kshape)
    assert out is not None or not accumulate
    ws = _ws(ws, conv2d_wgrad_ws_floats(x.shape, kshape, stride, x.dtype), x.device)
    _lib.call('stmi_conv2d_nhwc_wgrad', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(ws), n, h, w, ci, co, kh, kw,
              stride, int(accumulate), _dt(x), _stream())
    return dw


def conv_transpose_tile_stats_tiles(x_shape, kshape, stride, dtype):
    """Statistics rows per sample conv2d_transpose_fwd(tile_stats=...) leaves for this geometry (0: none)."""
    kh, kw, co, ci = kshape
    n, h, w, _ = x_shape
    return int(_lib.load().stmi_conv2d_transpose_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv2d_transpose_fwd(x, w_packed, kshape, stride, out=None, tile_stats=None):
    """kshape = (kh, kw, co, ci) of the reference's transposed-conv weight; packed with
    PACK_DGRAD as if it were the HWIO weight of a conv co -> ci.
    tile_stats: fp32 [n, conv_transpose_tile_stats_tiles(...), 4, co] that receives the per-tile sums of y for
    instance_norm_fwd(tile_stats=...)."""
    kh, kw, co, ci = kshape
    n, h, w, c = x.shape
    assert c == ci
    _chk(x, 'x')
    _chk(w_packed, 'w_packed', x.dtype, (packed_elems(kh, kw, co, ci, PACK_DGRAD),))
    oshape = (n, h * stride, w * stride, co)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    if tile_stats is not None:
        _chk(tile_stats, 'tile_stats', torch.float32,
             (n, conv_transpose_tile_stats_tiles(x.shape, kshape, stride, x.dtype), 4, co))
        _lib.call('stmi_conv2d_transpose_nhwc_fwd_tile_stats', x.data_ptr(), w_packed.data_ptr(), y.data_ptr(),
                  tile_stats.data_ptr(), n, h, w, ci, co, kh, kw, stride, _dt(x), _stream())
        return y
    _lib.call('stmi_conv2d_transpose_nhwc_fwd', x.data_ptr(), w_packed.data_ptr(), y.data_ptr(), n, h, w, ci, co,
              kh, kw, stride, _dt(x), _stream())
    return y


def conv2d_transpose_dgrad(dy, w_packed, kshape, stride, addend=None, out=None):
    kh, kw, co, ci = kshape
    n, hs, ws, c = dy.shape
    assert c == co and hs % stride == 0 and ws % stride == 0
    h, w = hs // stride, ws // stride
    _chk(dy, 'dy')
    _chk(w_packed, 'w_packed', dy.dtype, (packed_elems(kh, kw, co, ci, PACK_FWD),))
    oshape = (n, h, w, ci)
    dx = torch.empty(oshape, dtype=dy.dtype, device=dy.device) if out is None else _chk(out, 'out', dy.dtype, oshape)
    if addend is not None:
        _chk(addend, 'addend', dy.dtype, oshape)
    _lib.call('stmi_conv2d_transpose_nhwc_dgrad', dy.data_ptr(), w_packed.data_ptr(), dx.data_ptr(), _ptr(addend),
              n, h, w, ci, co, kh, kw, stride, _dt(dy), _stream())
    return dx


def conv2d_transpose_wgrad_ws_floats(x_shape, kshape, stride, dtype):
    kh, kw, co, ci = kshape
    n, h, w, _ = x_shape
    return int(_lib.load().stmi_conv2d_transpose_wgrad_ws_floats(n, h, w, ci, co, kh, kw, stride, _DT[dtype]))


def conv2d_transpose_wgrad(x, dy, kshape, stride, out=None, accumulate=False, ws=None):
    kh, kw, co, ci = kshape
    n, h, w, c = x.shape
    assert c == ci
    _chk(x, 'x')
    _chk(dy, 'dy', x.dtype, (n, h * stride, w * stride, co))
    dw = torch.empty(kshape, dtype=torch.float32, device=x.device) if out is None else _chk(out, 'out', torch.float32, kshape)
    assert out is not None or not accumulate
    ws = _ws(ws, conv2d_transpose_wgrad_ws_floats(x.shape, kshape, stride, x.dtype), x.device)
    _lib.call('stmi_conv2d_transpose_nhwc_wgrad', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(ws), n, h, w, ci, co,
              kh, kw, stride, int(accumulate), _dt(x), _stream())
    return dw


# ------------------------------------------------------------ row patches
def pixels_to_rows(x, kw, dx0, out=None):
    """[n,h,w,c] (kw*c <= 32) -> [n,h,w,32]: the kw x c window starting dx0 columns from each pixel."""
    n, h, w, c = x.shape
    _chk(x, 'x')
    oshape = (n, h, w, 32)
    r = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    _lib.call('stmi_pixels_to_rows', x.data_ptr(), r.data_ptr(), n, h, w, c, kw, dx0, _dt(x), _stream())
    return r


def rows_to_pixels_sum(rows, c, kw, dx0, out=None, addend=None):
    n, h, w, k = rows.shape
    assert k == 32
    _chk(rows, 'rows')
    oshape = (n, h, w, c)
    if addend is not None:
        _chk(addend, 'addend', rows.dtype, oshape)
    y = torch.empty(oshape, dtype=rows.dtype, device=rows.device) if out is None else _chk(out, 'out', rows.dtype, oshape)
    _lib.call('stmi_rows_to_pixels_sum', rows.data_ptr(), _ptr(addend), y.data_ptr(), n, h, w, c, kw, dx0, _dt(rows),
              _stream())
    return y


def rows_wgrad_to_hwio(d, kshape, mode, out=None):
    kh, kw, ci, co = kshape
    _chk(d, 'd', torch.float32, (kh, 1, 32, co) if mode == 0 else (kh, 1, ci, 32))
    dw = torch.empty(kshape, dtype=torch.float32, device=d.device) if out is None else _chk(out, 'out', torch.float32, kshape)
    _lib.call('stmi_rows_wgrad_to_hwio', d.data_ptr(), dw.data_ptr(), kh, kw, ci, co, mode, _stream())
    return dw


# ---------------------------------------------------------- instance norm
IN_EPS = 1e-3


def instance_norm_ws(n, h, w, c, device):
    return torch.empty(_lib.load().stmi_instance_norm_ws_floats(n, h, w, c), dtype=torch.float32, device=device)


def instance_norm_fwd(x, scale, shift, act=ACT_NONE, residual=None, want_f32=False, ws=None, out=None, stats=None,
                      out_f32=None, eps=IN_EPS, tile_stats=None):
    """Returns (y, stats[, y_f32]); stats [n,c,2] = (mean, 1/sqrt(var+eps)) for the backward.
    tile_stats: the per-tile sums conv2d_fwd(..., tile_stats=) left while writing x; with them the
    statistics pass over x is skipped."""
    n, h, w, c = x.shape
    _chk(x, 'x')
    _chk(scale, 'scale', torch.float32, (c,))
    _chk(shift, 'shift', torch.float32, (c,))
    if ws is None and tile_stats is None:
        ws = instance_norm_ws(n, h, w, c, x.device)
    y = torch.empty_like(x) if out is None else _chk(out, 'out', x.dtype, x.shape)
    if stats is None:
        stats = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
    y32 = None
    if want_f32:
        y32 = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out_f32 is None else out_f32
    if residual is not None:
        _chk(residual, 'residual', x.dtype, x.shape)
    if tile_stats is not None:
        _chk(tile_stats, 'tile_stats', torch.float32)
        assert tile_stats.dim() == 4 and tuple(tile_stats.shape[2:]) == (4, c) and tile_stats.shape[0] == n, tuple(tile_stats.shape)

    def launch():
        s = _stream()
        if tile_stats is not None:
            _lib.call('stmi_instance_norm_nhwc_apply_tiles', x.data_ptr(), tile_stats.data_ptr(), tile_stats.shape[1],
                      stats.data_ptr(), float(eps), scale.data_ptr(), shift.data_ptr(), _ptr(residual), y.data_ptr(),
                      _ptr(y32), act, n, h, w, c, _dt(x), s)
            return
        _lib.call('stmi_instance_norm_nhwc_fwd', x.data_ptr(), ws.data_ptr(), stats.data_ptr(), float(eps),
                  scale.data_ptr(), shift.data_ptr(), _ptr(residual), y.data_ptr(), _ptr(y32), act, n, h, w, c,
                  _dt(x), s)
    launch()
    if PROFILE_HBM is not None:  # one read and one write of the activation (+ the skip tensor, + the fp32 copy)
        nbytes = x.numel() * x.element_size() * (2 + (residual is not None)) + (0 if y32 is None else 4 * x.numel())
        PROFILE_HBM.append(('instance_norm_fwd', nbytes, launch))
    return (y, stats, y32) if want_f32 else (y, stats)


def instance_norm_stats(x, ws=None, stats=None, eps=IN_EPS):
    """stats [n,c,2] = (mean, 1/sqrt(var+eps)) of x, as a pass of its own."""
    n, h, w, c = x.shape
    _chk(x, 'x')
    if ws is None:
        ws = instance_norm_ws(n, h, w, c, x.device)
    if stats is None:
        stats = torch.empty((n, c, 2), dtype=torch.float32, device=x.device)
    _chk(stats, 'stats', torch.float32, (n, c, 2))
    _lib.call('stmi_instance_norm_nhwc_stats', x.data_ptr(), ws.data_ptr(), stats.data_ptr(), float(eps), n, h, w, c,
              _dt(x), _stream())
    return stats


def instance_norm_apply_u8(x, stats, scale, shift, act=ACT_NONE, out=None):
    """The 3-channel last layer's apply pass straight to 8-bit frames: (uint8)(act(scale*xhat + shift) * 255),
    truncated like the reference's (img * 255).astype(uint8)."""
    n, h, w, c = x.shape
    _chk(x, 'x')
    _chk(stats, 'stats', torch.float32, (n, c, 2))
    _chk(scale, 'scale', torch.float32, (c,))
    _chk(shift, 'shift', torch.float32, (c,))
    y = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if out is None else _chk(out, 'out', torch.uint8, x.shape)
    _lib.call('stmi_instance_norm_nhwc_apply_u8', x.data_ptr(), stats.data_ptr(), scale.data_ptr(), shift.data_ptr(),
              y.data_ptr(), act, n, h, w, c, _dt(x), _stream())
    return y


def instance_norm_bwd_rows_count(x_shape, dtype):
    """Partial rows per sample instance_norm_bwd leaves at the start of its workspace ([n, rows, 2, c])."""
    n, h, w, c = x_shape
    return int(_lib.load().stmi_instance_norm_bwd_rows_count(n, h, w, c, _DT[dtype]))


def instance_norm_param_grads(rows, n, c, dscale, dshift, accumulate=False):
    """dshift / dscale fp32 [c] from the partial rows of a norm's backward reduction ([n, r, 2, c], or the
    workspace instance_norm_bwd used viewed that way), summed in a fixed order."""
    _chk(rows, 'rows', torch.float32)
    assert rows.numel() % (n * 2 * c) == 0, (tuple(rows.shape), n, c)
    nrows = rows.numel() // (n * 2 * c)
    _chk(dscale, 'dscale', torch.float32, (c,))
    _chk(dshift, 'dshift', torch.float32, (c,))
    _lib.call('stmi_instance_norm_param_grads', rows.data_ptr(), nrows, n, c, dscale.data_ptr(), dshift.data_ptr(),
              int(accumulate), _stream())


def instance_norm_bwd(x, dy, stats, scale, shift, dscale, dshift, act=ACT_NONE, ws=None, out=None, eps=IN_EPS, rows=None,
                      accumulate=True):
    """dx returned; parameter gradients are ADDED into dscale/dshift (fp32 [c]), or WRITTEN when accumulate is
    False; both None: left to a later instance_norm_param_grads on the rows this call reduced (`rows`, or the
    start of `ws`).
    rows: fp32 [n, r, 2, c] partial sums conv2d_dgrad(..., in_bwd=) left while writing dy; with them the
    reduction pass over x and dy is skipped."""
    n, h, w, c = x.shape
    _chk(x, 'x')
    _chk(dy, 'dy', x.dtype, x.shape)
    _chk(stats, 'stats', torch.float32, (n, c, 2))
    assert (dscale is None) == (dshift is None)
    for t, nm in ((scale, 'scale'), (shift, 'shift'), (dscale, 'dscale'), (dshift, 'dshift')):
        if t is not None:
            _chk(t, nm, torch.float32, (c,))
    if ws is None and rows is None:
        ws = instance_norm_ws(n, h, w, c, x.device)
    dx = torch.empty_like(x) if out is None else _chk(out, 'out', x.dtype, x.shape)
    if rows is not None:
        _chk(rows, 'rows', torch.float32)
        assert rows.dim() == 4 and rows.shape[0] == n and tuple(rows.shape[2:]) == (2, c), tuple(rows.shape)

    def launch():
        if rows is not None:
            _lib.call('stmi_instance_norm_nhwc_bwd_rows', x.data_ptr(), dy.data_ptr(), stats.data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), dx.data_ptr(), _ptr(dscale), _ptr(dshift), int(accumulate), rows.data_ptr(),
                      rows.shape[1], act, n, h, w, c, _dt(x), _stream())
            return
        _lib.call('stmi_instance_norm_nhwc_bwd', x.data_ptr(), dy.data_ptr(), stats.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), dx.data_ptr(), _ptr(dscale), _ptr(dshift), int(accumulate), ws.data_ptr(), float(eps), act,
                  n, h, w, c, _dt(x), _stream())
    launch()
    if PROFILE_HBM is not None:  # read x and dy once, write dx once
        PROFILE_HBM.append(('instance_norm_bwd', 3 * x.numel() * x.element_size(), launch))
    return dx


# -------------------------------------------------------------- VGG pieces
def scale_shift(x_f32, mean, scale, dtype, out=None):
    _chk(x_f32, 'x', torch.float32)
    c = x_f32.shape[-1]
    _chk(mean, 'mean', torch.float32, (c,))
    y = torch.empty(x_f32.shape, dtype=dtype, device=x_f32.device) if out is None else _chk(out, 'out', dtype, x_f32.shape)
    _lib.call('stmi_scale_shift', x_f32.data_ptr(), mean.data_ptr(), y.data_ptr(), float(scale), x_f32.numel(), c,
              _dt(y), _stream())
    return y


def maxpool_fwd(x, out=None):
    n, h, w, c = x.shape
    _chk(x, 'x')
    oshape = (n, (h + 1) // 2, (w + 1) // 2, c)
    y = torch.empty(oshape, dtype=x.dtype, device=x.device) if out is None else _chk(out, 'out', x.dtype, oshape)
    _lib.call('stmi_maxpool2x2_same_fwd', x.data_ptr(), y.data_ptr(), n, h, w, c, _dt(x), _stream())
    return y


def maxpool_bwd(x, dy, out=None):
    n, h, w, c = x.shape
    _chk(x, 'x')
    _chk(dy, 'dy', x.dtype, (n, (h + 1) // 2, (w + 1) // 2, c))
    dx = torch.empty_like(x) if out is None else _chk(out, 'out', x.dtype, x.shape)
    _lib.call('stmi_maxpool2x2_same_bwd', x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, h, w, c, _dt(x), _stream())
    return dx


# ------------------------------------------------------------------- Gram
def gram_ws_floats(f_shape, dtype):
    """Floats of workspace gram_fwd needs for the partial sums of its pixel slices (0: none)."""
    n, h, w, c = f_shape
    return int(_lib.load().stmi_gram_ws_floats(n, h, w, c, _DT[dtype]))


def gram_fwd(f, out=None, accumulate=False, ws=None):
    """fp32 [n,c,c] = F^T F / (h w); accumulate: ADD into `out`.  Bit-reproducible (no atomics)."""
    n, h, w, c = f.shape
    _chk(f, 'f')
    g = torch.empty((n, c, c), dtype=torch.float32, device=f.device) if out is None else _chk(out, 'out', torch.float32, (n, c, c))
    assert out is not None or not accumulate
    ws = _ws(ws, gram_ws_floats(f.shape, f.dtype), f.device)
    _lib.call('stmi_gram_nhwc_fwd_add' if accumulate else 'stmi_gram_nhwc_fwd', f.data_ptr(), g.data_ptr(), _ptr(ws),
              n, h, w, c, _dt(f), _stream())
    return g


class ZeroList:
    """Up to 16 fp32 buffers cleared in one launch (optionally advancing a device step counter);
    the ctypes tables are built once so that a step only passes pointers."""

    def __init__(self, tensors, step_dev=None):
        import ctypes
        tensors = [t for t in tensors if t is not None and t.numel() > 0]
        assert len(tensors) <= 16
        for t in tensors:
            _chk(t, 'buffer', torch.float32)
        if step_dev is not None:
            _chk(step_dev, 'step_dev', torch.int32)
        self._keep = (tensors, step_dev)
        self._ptrs = (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
        self._counts = (ctypes.c_long * len(tensors))(*[t.numel() for t in tensors])
        self._n = len(tensors)

    def run(self):
        _lib.call('stmi_zero_multi', self._ptrs, self._counts, self._n, _ptr(self._keep[1]), _stream())


def gram_packed_elems(n, c):
    return _lib.load().stmi_gram_packed_elems(n, c)


def gram_bwd(f, s_packed, df, accumulate):
    n, h, w, c = f.shape
    _chk(f, 'f')
    _chk(s_packed, 's_packed', f.dtype, (gram_packed_elems(n, c),))
    _chk(df, 'df', f.dtype, f.shape)
    _lib.call('stmi_gram_nhwc_bwd', f.data_ptr(), s_packed.data_ptr(), df.data_ptr(), int(accumulate), n, h, w, c,
              _dt(f), _stream())
    return df


# ----------------------------------------------------------------- losses
# A loss call leaves its term as per-block partial sums in a "loss slot" (fp32 [LOSS_SLOT], overwritten
# entirely); loss_slots_sum adds the slots of each term up in a fixed order.  No float atomics: loss values
# are bit-reproducible.
LOSS_SLOT = _lib.LOSS_SLOT_FLOATS


def loss_slots(count, device):
    return torch.zeros((count, LOSS_SLOT), dtype=torch.float32, device=device)


def _chk_slot(slot):
    _chk(slot, 'slot', torch.float32)
    if slot.numel() != LOSS_SLOT or slot.data_ptr() % 16:
        raise ValueError('a loss slot is %d contiguous, 16-byte aligned floats' % LOSS_SLOT)


def loss_slots_sum(slots, first, out):
    """out[t] = sum of slots[first[t] : first[t+1]] for every term t (first: host list of len(out) + 1 indices)."""
    import ctypes
    _chk(slots, 'slots', torch.float32)
    _chk(out, 'out', torch.float32)
    nterms = len(first) - 1
    assert 1 <= nterms <= _lib.LOSS_TERMS_MAX and out.numel() >= nterms
    assert first[-1] * LOSS_SLOT <= slots.numel()
    arr = (ctypes.c_int * len(first))(*[int(v) for v in first])
    _lib.call('stmi_loss_slots_sum', slots.data_ptr(), arr, nterms, out.data_ptr(), _stream())
    return out


def slot_total(slot):
    """The term a single loss call left in `slot`, as a device tensor [1]."""
    out = torch.empty(1, dtype=torch.float32, device=slot.device)
    return loss_slots_sum(slot, [0, 1], out)


def gram_mse_fwd_bwd(g, target, loss, loss_scale, grad_scale, dtype, out=None):
    """`loss`: the loss slot that receives loss_scale * sum((g - target)^2) as partial sums."""
    n, c, _ = g.shape
    _chk(g, 'g', torch.float32, (n, c, c))
    _chk(target, 'target', torch.float32)
    assert target.numel() == c * c, 'style target must be one [c,c] Gram (broadcast over the batch)'
    _chk_slot(loss)
    ne = gram_packed_elems(n, c)
    s = torch.empty(ne, dtype=dtype, device=g.device) if out is None else _chk(out, 'out', dtype, (ne,))
    _lib.call('stmi_gram_mse_fwd_bwd', g.data_ptr(), target.data_ptr(), s.data_ptr(), loss.data_ptr(),
              float(loss_scale), float(grad_scale), n, c, _dt(s), _stream())
    return s


def mse_fwd_bwd(a, b, loss, loss_scale, grad_scale, want_grad=True, out=None):
    """`loss`: the loss slot that receives loss_scale * sum((a - b)^2) as partial sums."""
    _chk(a, 'a')
    _chk(b, 'b', a.dtype, a.shape)
    _chk_slot(loss)
    grad = None
    if want_grad:
        grad = torch.empty_like(a) if out is None else _chk(out, 'out', a.dtype, a.shape)
    _lib.call('stmi_mse_fwd_bwd', a.data_ptr(), b.data_ptr(), _ptr(grad), loss.data_ptr(), float(loss_scale),
              float(grad_scale), a.numel(), _dt(a), _stream())
    return grad


def tv_loss_fwd_bwd(img_f32, loss, weight, dtype, out=None):
    """`loss`: the loss slot that receives weight * TV(img) as partial sums."""
    n, h, w, c = img_f32.shape
    _chk(img_f32, 'img', torch.float32)
    _chk_slot(loss)
    grad = torch.empty(img_f32.shape, dtype=dtype, device=img_f32.device) if out is None else _chk(out, 'out', dtype, img_f32.shape)
    _lib.call('stmi_tv_loss_fwd_bwd', img_f32.data_ptr(), grad.data_ptr(), loss.data_ptr(), float(weight), n, h, w, c,
              _dt(grad), _stream())
    return grad


# -------------------------------------------------------------- optimizer
def adam_tf1(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, step_dev=None, t=0, advance=True):
    """TF1 Adam.  With `step_dev` (int32 [1] on the device) the step number lives on the device, which
    makes the call replayable inside a HIP graph: advance=True increments it before use; advance=False
    uses it as it stands (a second slice of the same parameter set in the same step)."""
    for tns, nm in ((p, 'p'), (g, 'g'), (m, 'm'), (v, 'v')):
        _chk(tns, nm, torch.float32)
        assert tns.numel() == p.numel()
    if step_dev is not None:
        _chk(step_dev, 'step_dev', torch.int32)
        t = 0 if advance else -1
    _lib.call('stmi_adam_tf1', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
              float(beta1), float(beta2), float(eps), float(grad_scale), _ptr(step_dev), int(t), _stream())


# ------------------------------------------------------------------- misc
def cast(src, dtype, out=None):
    _chk(src, 'src')
    dst = torch.empty(src.shape, dtype=dtype, device=src.device) if out is None else _chk(out, 'out', dtype, src.shape)
    if src.dtype == torch.float32:
        _lib.call('stmi_cast', src.data_ptr(), dst.data_ptr(), src.numel(), _dt(dst), 0, _stream())
    elif dtype == torch.float32:
        _lib.call('stmi_cast', src.data_ptr(), dst.data_ptr(), src.numel(), _dt(src), 1, _stream())
    else:
        raise TypeError('cast converts between float32 and one other dtype')
    return dst


def activation(x, act, out=None):
    _chk(x, 'x')
    y = torch.empty_like(x) if out is None else _chk(out, 'out', x.dtype, x.shape)
    _lib.call('stmi_activation', x.data_ptr(), y.data_ptr(), x.numel(), act, _dt(x), _stream())
    return y


def clip_(x, lo, hi):
    _chk(x, 'x', torch.float32)
    _lib.call('stmi_clip', x.data_ptr(), x.numel(), float(lo), float(hi), _stream())
    return x
