"""Per-kernel parity of the HIP path (through the C-ABI) against the CPU oracle.

fp32 mode: max |err| / max |ref| <= 1e-3 (BASELINE.json north_star tolerance).
bf16 mode: <= 3e-2 per op (the end-to-end gates are PSNR >= 40 dB / Gram 1e-2).
"""
import numpy as np
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 1e-3, torch.bfloat16: 3e-2}
DTYPES = [torch.float32, torch.bfloat16]


def rel_err(got, ref):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float64) * scale)


def q(t, dtype):
    """value as the kernel sees it (rounded to the compute dtype), in float64"""
    return t.to(dtype).double()


CONV_CASES = [
    # n, h, w, ci, co, k, s
    (2, 12, 10, 3, 32, 9, 1),     # stylizer conv1 (row taps, RGB input)
    (2, 12, 10, 32, 64, 3, 2),    # conv2, even size: pads (0,1)
    (1, 11, 9, 64, 128, 3, 2),    # conv3 on odd size: pads (1,1)
    (2, 9, 7, 128, 128, 3, 1),    # residual conv
    (1, 10, 12, 32, 3, 9, 1),     # final conv: 3 output channels
    (1, 7, 9, 3, 64, 3, 1),       # VGG conv1_1
    (1, 6, 5, 256, 512, 3, 1),    # VGG conv4_1
    (3, 5, 5, 8, 16, 3, 1),       # reduced-width VGG layer
    # LDS-resident patch kernel (3x3, stride 1, >= 128 bytes of channels): every patch width,
    # ragged edges in both directions, several tiles per image, several channel chunks
    (1, 5, 70, 64, 64, 3, 1),
    (2, 11, 40, 128, 64, 3, 1),
    (1, 20, 130, 64, 192, 3, 1),
    (2, 17, 16, 192, 128, 3, 1),
    (1, 3, 2, 64, 64, 3, 1),
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', CONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_fwd_dgrad_wgrad(cuda, dtype, case):
    from stmi import ops
    n, h, w, ci, co, k, s = case
    x = rnd((n, h, w, ci), 1)
    wt = rnd((k, k, ci, co), 2, 1.0 / np.sqrt(k * k * ci))
    bias = rnd((co,), 3)
    xq, wq = q(x, dtype).requires_grad_(), q(wt, dtype).requires_grad_()
    y_ref = O.conv2d_same(xq, wq, s, bias=bias)
    dy = rnd(tuple(y_ref.shape), 4)
    dyq = q(dy, dtype)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, (xq, wq), dyq)

    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    w32 = wt.to(cuda, torch.float32)
    wf = ops.pack_conv_weights(w32, ops.PACK_FWD, dtype)
    wd = ops.pack_conv_weights(w32, ops.PACK_DGRAD, dtype)
    y, pre = ops.conv2d_fwd(xd, wf, (k, k, ci, co), s, bias=bias.to(cuda, torch.float32), relu=True, want_pre=True)
    assert rel_err(pre, y_ref) <= TOL[dtype]
    assert rel_err(y, torch.relu(y_ref)) <= TOL[dtype]
    dx = ops.conv2d_dgrad(dyd, wd, (k, k, ci, co), s, (h, w))
    assert rel_err(dx, gx_ref) <= TOL[dtype]
    dw = ops.conv2d_wgrad(xd, dyd, (k, k, ci, co), s)
    assert rel_err(dw, gw_ref) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('chans', [(16, 32), (64, 128)], ids=['gather', 'patch'])
def test_conv_dgrad_epilogue(cuda, dtype, chans):
    """alpha, ReLU mask and addend of the backward-data epilogue (both kernel families)."""
    from stmi import ops
    n, h, w, k, s = 2, 8, 6, 3, 1
    ci, co = chans
    wt = rnd((k, k, ci, co), 2, 0.1)
    dy = rnd((n, h, w, co), 4)
    mask = rnd((n, h, w, ci), 5)
    add = rnd((n, h, w, ci), 6)
    xq = torch.zeros((n, h, w, ci), dtype=torch.float64, requires_grad=True)
    y_ref = O.conv2d_same(xq, q(wt, dtype), s)
    (gx,) = torch.autograd.grad(y_ref, xq, q(dy, dtype))
    ref = 0.5 * gx * (q(mask, dtype) > 0) + q(add, dtype)
    wd = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    got = ops.conv2d_dgrad(dy.to(cuda, dtype), wd, (k, k, ci, co), s, (h, w), relu_mask=mask.to(cuda, dtype),
                           addend=add.to(cuda, dtype), alpha=0.5)
    assert rel_err(got, ref) <= TOL[dtype]


POOL_CASES = [
    # n, h, w, ci, co
    (2, 16, 64, 64, 64),      # one channel chunk (VGG conv1_2 form), 4 x 32 tiles
    (1, 12, 40, 128, 128),    # 128-channel tiles, ragged right edge
    (2, 11, 37, 128, 64),     # odd sizes: windows cut by the bottom and right edges
    (1, 16, 16, 256, 512),    # 8 x 16 tiles
    (3, 5, 5, 8, 16),         # not a patch-kernel layer: two launches, same answer
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', POOL_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_with_fused_pool_equals_conv_then_pool(cuda, dtype, case):
    """vgg.py:99-103: conv -> ReLU -> 2x2 max pool.  The pooled tensor a convolution's launch leaves is bit for bit
    the pool of the tensor it stores, and that tensor is the one the plain launch stores."""
    from stmi import ops
    n, h, w, ci, co = case
    ks = (3, 3, ci, co)
    x, wt, bias = rnd((n, h, w, ci), 1), rnd(ks, 2, 1.0 / np.sqrt(9 * ci)), rnd((co,), 3)
    xd = x.to(cuda, dtype)
    wf = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    b32 = bias.to(cuda, torch.float32)
    y0 = ops.conv2d_fwd(xd, wf, ks, 1, bias=b32, relu=True)
    p0 = ops.maxpool_fwd(y0)
    y1, p1 = ops.conv2d_fwd_pool(xd, wf, ks, 1, bias=b32, relu=True)
    fused = ci * (2 if dtype == torch.bfloat16 else 4) >= 128
    assert ('[+pool]' in ops.last_conv_kernel()) == fused, ops.last_conv_kernel()
    assert torch.equal(y1, y0) and torch.equal(p1, p0)
    ref = torch.relu(O.conv2d_same(q(x, dtype), q(wt, dtype), 1, bias=bias))
    assert rel_err(p1, O.max_pool_2x2_same(ref)) <= TOL[dtype]
    # without the ReLU (finished values = staged values)
    y2, p2 = ops.conv2d_fwd_pool(xd, wf, ks, 1, bias=b32, relu=False)
    assert torch.equal(p2, ops.maxpool_fwd(ops.conv2d_fwd(xd, wf, ks, 1, bias=b32)))


@pytest.mark.parametrize('case', [(1, 256, 384, 128), (2, 250, 402, 128)], ids=lambda c: 'x'.join(map(str, c)))
def test_convolution_absorbing_the_norm_in_front_of_it(cuda, case):
    """full_graph_stylizer_network.py:52-57 at inference: conv_b(relu(IN(conv_a(.)))) without the normalised tensor.
    relu(a x + b) = a (max(x, t) - t) with t = -b / a: the kernel clamps on the way in, uses per-sample weights W a and
    starts its accumulators from the constant term; compared with the three-launch route directly and, behind the next
    instance norm, with the oracle."""
    from stmi import ops
    n, h, w, c = case
    dtype = torch.bfloat16
    co = 128
    x0 = rnd((n, h, w, c), 1) * 0.5 + 0.2
    wa = rnd((3, 3, c, c), 2, 1.0 / np.sqrt(9 * c))
    wb = rnd((3, 3, c, co), 3, 1.0 / np.sqrt(9 * c))
    wb[1, 1] += 0.02
    sa, ha = rnd((c,), 4).abs() + 0.5, rnd((c,), 5)
    sb, hb = rnd((co,), 6) + 1.5, rnd((co,), 7)
    x0d = x0.to(cuda, dtype)
    wpa = ops.pack_conv_weights(wa.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    wpb = ops.pack_conv_weights(wb.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    sad, had, sbd, hbd = (t.to(cuda, torch.float32) for t in (sa, ha, sb, hb))
    ksb = (3, 3, c, co)
    need = ops.conv_absorb_norm_ws_bytes((n, h, w, c), ksb, 1, dtype)
    assert need > 0 and ops.conv_absorb_norm_ws_bytes((n, h, w, c), ksb, 1, torch.float32) == 0
    assert ops.conv_absorb_norm_ws_bytes((n, 64, 64, c), ksb, 1, dtype) == 0    # planes too small for the persistent kernel
    assert ops.conv_absorb_norm_ws_bytes((n, h, w, 64), (3, 3, 64, co), 1, dtype) == 0   # single-chunk layers are not persistent
    ts_a = ops.conv_tile_stats_tiles((n, h, w, c), (3, 3, c, c), 1, dtype)
    ts_b = ops.conv_tile_stats_tiles((n, h, w, c), (3, 3, c, co), 1, dtype)
    tiles_a = torch.zeros((n, ts_a, 4, c), device=cuda)
    xa = ops.conv2d_fwd(x0d, wpa, (3, 3, c, c), 1, tile_stats=tiles_a)          # raw output of the first convolution
    # three launches: apply pass, convolution, next norm
    ya, st_ref = ops.instance_norm_fwd(xa, sad, had, ops.ACT_RELU, tile_stats=tiles_a)
    tiles_r = torch.zeros((n, ts_b, 4, co), device=cuda)
    zr = ops.conv2d_fwd(ya, wpb, (3, 3, c, co), 1, tile_stats=tiles_r)
    out_r, _ = ops.instance_norm_fwd(zr, sbd, hbd, ops.ACT_NONE, tile_stats=tiles_r)
    # absorbed
    tiles_b = torch.full((n, ts_b, 4, co), float('nan'), device=cuda)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    z, st = ops.conv2d_fwd_absorb_norm(xa, tiles_a, sad, had, wpb, ksb, 1, tiles_b, ws)
    assert 'conv_patch_pers_kernel<__bf16, 4, 32, 4, true, true, true>' in ops.last_conv_kernel()
    assert torch.equal(st, st_ref)
    z2, _ = ops.conv2d_fwd_absorb_norm(xa, tiles_a, sad, had, wpb, ksb, 1, tiles_b, ws)
    assert torch.equal(z, z2)
    out, _ = ops.instance_norm_fwd(z, sbd, hbd, ops.ACT_NONE, tile_stats=tiles_b)
    assert rel_err(z, zr.double()) <= 3e-2       # the same tensor, up to the bf16 rounding of both routes
    assert rel_err(out, out_r.double()) <= 3e-2
    # the oracle, in fp32 on the host (the planes are large), from the stored raw tensor xa
    xa32 = xa.float().cpu()
    ya32 = torch.relu(O.instance_norm(xa32, sa.float(), ha.float()))
    ref = O.instance_norm(O.conv2d_same(ya32, q(wb, dtype).float(), 1), sb.float(), hb.float())
    assert rel_err(out, ref) <= TOL[dtype]
    assert rel_err(out_r, ref) <= TOL[dtype]


@pytest.mark.parametrize('case', [(2, 32, 128, 32, 64), (1, 26, 70, 64, 128), (2, 8, 66, 32, 64)], ids=lambda c: 'x'.join(map(str, c)))
def test_stride2_convolution_absorbing_the_norm_in_front_of_it(cuda, case):
    """The down-sampling layers at inference (full_graph_stylizer_network.py:47-49: conv -> norm -> ReLU -> strided
    conv -> norm) without the normalised tensor: the pair-patch kernel clamps on the way in.  Compared behind the next
    instance norm with the three-launch route and the oracle; ragged tiles, windows that reach past the right and
    bottom edges (the clamp value stands in for the padding there)."""
    from stmi import ops
    n, h, w, c, co = case
    dtype = torch.bfloat16
    xa64 = rnd((n, h, w, c), 1) * 0.8 + 0.3               # raw output of the convolution in front
    wb = rnd((3, 3, c, co), 3, 1.0 / np.sqrt(9 * c))
    sa, ha = rnd((c,), 4).abs() + 0.5, rnd((c,), 5)
    sb, hb = rnd((co,), 6) + 1.5, rnd((co,), 7)
    xa = xa64.to(cuda, dtype)
    wpb = ops.pack_conv_weights(wb.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    sad, had, sbd, hbd = (t.to(cuda, torch.float32) for t in (sa, ha, sb, hb))
    ksb = (3, 3, c, co)
    need = ops.conv_absorb_norm_ws_bytes((n, h, w, c), ksb, 2, dtype)
    assert need > 0 and ops.conv_absorb_norm_ws_bytes((n, h, w, c), ksb, 2, torch.float32) == 0
    assert ops.conv_absorb_norm_ws_bytes((n, h + 1, w, c), ksb, 2, dtype) == 0       # odd side: pads above as well
    # the tile sums of xa as ANY producer leaves them: here one row per sample from a statistics pass
    _, st_ref = ops.instance_norm_fwd(xa, sad, had, ops.ACT_RELU)
    xf = xa.float()
    k0 = xf[:, 0, 0, :]
    d = xf - k0[:, None, None, :]
    tiles_a = torch.stack([k0, d.sum(dim=(1, 2)), (d * d).sum(dim=(1, 2)), torch.full_like(k0, float(h * w))], dim=1)[:, None].contiguous()
    ya, _ = ops.instance_norm_fwd(xa, sad, had, ops.ACT_RELU)
    ts_b = ops.conv_absorb_norm_tiles((n, h, w, c), ksb, 2, dtype)
    assert ts_b == ops.conv_tile_stats_tiles((n, h, w, c), ksb, 2, dtype) > 0
    tiles_r = torch.zeros((n, ts_b, 4, co), device=cuda)
    zr = ops.conv2d_fwd(ya, wpb, ksb, 2, tile_stats=tiles_r)
    out_r, _ = ops.instance_norm_fwd(zr, sbd, hbd, ops.ACT_NONE, tile_stats=tiles_r)
    tiles_b = torch.full((n, ts_b, 4, co), float('nan'), device=cuda)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    z, st = ops.conv2d_fwd_absorb_norm(xa, tiles_a, sad, had, wpb, ksb, 2, tiles_b, ws)
    assert 'conv_dn2_kernel<__bf16' in ops.last_conv_kernel() and ops.last_conv_kernel().count('true') == 2
    assert rel_err(st[..., 0], st_ref[..., 0].double()) <= 1e-5 and rel_err(st[..., 1], st_ref[..., 1].double()) <= 1e-4
    out, _ = ops.instance_norm_fwd(z, sbd, hbd, ops.ACT_NONE, tile_stats=tiles_b)
    assert rel_err(z, zr.double()) <= 3e-2       # the same tensor, up to the bf16 rounding of both routes
    assert rel_err(out, out_r.double()) <= 3e-2
    y64 = torch.relu(O.instance_norm(q(xa64, dtype), sa.float().double(), ha.float().double()))
    ref = O.instance_norm(O.conv2d_same(y64, q(wb, dtype), 2), sb.float().double(), hb.float().double())
    assert rel_err(out, ref) <= TOL[dtype]


@pytest.mark.parametrize('case', [(2, 24, 64), (1, 37, 45)], ids=lambda c: 'x'.join(map(str, c)))
def test_output_layer_absorbing_the_norm_in_front_of_it(cuda, case):
    """full_graph_stylizer_network.py:76-78 at inference: the 9 x 9 layer into three channels reading the raw output
    of the last up-sampling block and applying that block's norm + ReLU on the way in."""
    from stmi import ops
    n, h, w = case
    dtype, c, k = torch.bfloat16, 32, 9
    xa64 = rnd((n, h, w, c), 1) * 0.8 + 0.3
    wb = rnd((k, k, c, 3), 3, 1.0 / np.sqrt(k * k * c))
    sa, ha = rnd((c,), 4).abs() + 0.5, rnd((c,), 5)
    xa = xa64.to(cuda, dtype)
    wpb = ops.pack_conv_weights(wb.to(cuda, torch.float32), ops.PACK_FWD_ROWOUT, dtype)
    sad, had = sa.to(cuda, torch.float32), ha.to(cuda, torch.float32)
    need = ops.conv_to3_absorb_norm_ws_bytes((n, h, w, c), k, dtype)
    assert need > 0 and ops.conv_to3_absorb_norm_ws_bytes((n, h, w, c), k, torch.float32) == 0
    xf = xa.float()
    k0 = xf[:, 0, 0, :]
    d = xf - k0[:, None, None, :]
    tiles_a = torch.stack([k0, d.sum(dim=(1, 2)), (d * d).sum(dim=(1, 2)), torch.full_like(k0, float(h * w))], dim=1)[:, None].contiguous()
    ya, st_ref = ops.instance_norm_fwd(xa, sad, had, ops.ACT_RELU)
    zr = ops.conv2d_to3(ya, wpb, k)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    z, st = ops.conv2d_to3_absorb_norm(xa, tiles_a, sad, had, wpb, k, ws)
    assert 'conv_to3_kernel<__bf16, 9, 64, 40, true>' in ops.last_conv_kernel()
    assert rel_err(st[..., 1], st_ref[..., 1].double()) <= 1e-4
    assert rel_err(z, zr.double()) <= 3e-2
    y64 = torch.relu(O.instance_norm(q(xa64, dtype), sa.float().double(), ha.float().double()))
    ref = O.conv2d_same(y64, q(wb, dtype), 1)
    assert rel_err(z, ref) <= TOL[dtype]
    assert rel_err(zr, ref) <= TOL[dtype]


@pytest.mark.parametrize('case', [(2, 16, 64), (1, 13, 45), (1, 3, 2)], ids=lambda c: 'x'.join(map(str, c)))
def test_transposed_convolution_absorbing_the_norm_in_front_of_it(cuda, case):
    """full_graph_stylizer_network.py:72-76 at inference: the second up-sampling block's transposed convolution
    reading the raw output of the first and applying that block's norm + ReLU on the way in; the four output classes
    use different taps of the kernel, so each starts from the constant terms of its own taps."""
    from stmi import ops
    n, h, w = case
    dtype, ci, co, k, s = torch.bfloat16, 64, 32, 3, 2
    xa64 = rnd((n, h, w, ci), 1) * 0.8 + 0.3
    wt = rnd((k, k, co, ci), 3, 1.0 / np.sqrt(k * k * ci))          # HWOI
    sa, ha = rnd((ci,), 4).abs() + 0.5, rnd((ci,), 5)
    sb, hb = rnd((co,), 6) + 1.5, rnd((co,), 7)
    xa = xa64.to(cuda, dtype)
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    sad, had, sbd, hbd = (t.to(cuda, torch.float32) for t in (sa, ha, sb, hb))
    ks = (k, k, co, ci)
    need = ops.conv_transpose_absorb_norm_ws_bytes((n, h, w, ci), ks, s, dtype)
    assert need > 0 and ops.conv_transpose_absorb_norm_ws_bytes((n, h, w, ci), ks, s, torch.float32) == 0
    xf = xa.float()
    k0 = xf[:, 0, 0, :]
    d = xf - k0[:, None, None, :]
    tiles_a = torch.stack([k0, d.sum(dim=(1, 2)), (d * d).sum(dim=(1, 2)), torch.full_like(k0, float(h * w))], dim=1)[:, None].contiguous()
    ya, st_ref = ops.instance_norm_fwd(xa, sad, had, ops.ACT_RELU)
    ts = ops.conv_transpose_tile_stats_tiles((n, h, w, ci), ks, s, dtype)
    tiles_r = torch.zeros((n, ts, 4, co), device=cuda)
    zr = ops.conv2d_transpose_fwd(ya, wp, ks, s, tile_stats=tiles_r)
    tiles_b = torch.full((n, ts, 4, co), float('nan'), device=cuda)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda)
    z, st = ops.conv2d_transpose_fwd_absorb_norm(xa, tiles_a, sad, had, wp, ks, s, tiles_b, ws)
    assert 'conv_up2_clamp_kernel<__bf16, 1, 2>' in ops.last_conv_kernel()
    assert rel_err(st[..., 1], st_ref[..., 1].double()) <= 1e-4
    assert rel_err(z, zr.double()) <= 3e-2
    out, _ = ops.instance_norm_fwd(z, sbd, hbd, ops.ACT_RELU, tile_stats=tiles_b)
    y64 = torch.relu(O.instance_norm(q(xa64, dtype), sa.float().double(), ha.float().double()))
    z_ref = O.conv2d_transpose_same(y64, q(wt, dtype), s)
    assert rel_err(z, z_ref) <= TOL[dtype]
    ref = torch.relu(O.instance_norm(z_ref, sb.float().double(), hb.float().double()))
    assert rel_err(out, ref) <= TOL[dtype]


KSPLIT_CASES = [
    # n, h, w, ci, co, forced slices (None: the planner's own choice, which must be a split)
    (4, 16, 16, 512, 512, None),     # VGG conv5_x of a training batch
    (1, 16, 16, 512, 512, None),     # ... of one picture
    (1, 16, 16, 512, 512, 8),        # one chunk per slice (bf16)
    (2, 9, 7, 256, 256, 2),          # ragged tiles
    (1, 5, 70, 384, 64, 3),          # a split that is not a power of two
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', KSPLIT_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_split_k_convolution_matches_the_oracle_and_the_unsplit_launch(cuda, dtype, case, monkeypatch):
    """With a workspace, layers of few tiles run their K loop in slices (kernel name says so) and a fold launch
    applies bias / ReLU / mask / addend / alpha; same answer as the oracle, bit-identical from run to run, and
    equal to the unsplit launch up to the order of fp32 additions."""
    from stmi import ops
    n, h, w, ci, co, forced = case
    if forced:
        monkeypatch.setenv('STMI_KSPLIT', str(forced))
    k = 3
    ks = (k, k, ci, co)
    x, wt, bias = rnd((n, h, w, ci), 1), rnd(ks, 2, 1.0 / np.sqrt(9 * ci)), rnd((co,), 3)
    xq, wq = q(x, dtype).requires_grad_(), q(wt, dtype)
    y_ref = O.conv2d_same(xq, wq, 1, bias=bias)
    dy, mask, add = rnd(tuple(y_ref.shape), 4), rnd((n, h, w, ci), 5), rnd((n, h, w, ci), 6)
    (gx,) = torch.autograd.grad(y_ref, xq, q(dy, dtype))
    gx_ref = 0.5 * gx * (q(mask, dtype) > 0) + q(add, dtype)

    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    w32 = wt.to(cuda, torch.float32)
    wf, wd = ops.pack_conv_weights(w32, ops.PACK_FWD, dtype), ops.pack_conv_weights(w32, ops.PACK_DGRAD, dtype)
    need_f = ops.conv_ksplit_ws_floats((n, h, w, ci), ks, 1, dtype)
    need_d = ops.conv_ksplit_ws_floats((n, h, w, ci), ks, 1, dtype, dgrad=True)
    assert need_f > 0 and need_f % (n * h * w * co) == 0 and need_d % (n * h * w * ci) == 0
    if forced:
        assert need_f == forced * n * h * w * co
    ws = torch.full((max(need_f, need_d),), float('nan'), device=cuda)   # every float that is read was written
    b32 = bias.to(cuda, torch.float32)

    y0, pre0 = ops.conv2d_fwd(xd, wf, ks, 1, bias=b32, relu=True, want_pre=True)
    assert '[split K]' not in ops.last_conv_kernel()
    y1, pre1 = ops.conv2d_fwd(xd, wf, ks, 1, bias=b32, relu=True, want_pre=True, ws=ws)
    assert '[split K]' in ops.last_conv_kernel() and 'conv_patch_kernel' in ops.last_conv_kernel()
    y2, pre2 = ops.conv2d_fwd(xd, wf, ks, 1, bias=b32, relu=True, want_pre=True, ws=ws)
    assert torch.equal(y1, y2) and torch.equal(pre1, pre2)
    assert rel_err(pre1, y_ref) <= TOL[dtype] and rel_err(y1, torch.relu(y_ref)) <= TOL[dtype]
    assert rel_err(pre1, pre0.double()) <= (2e-6 if dtype == torch.float32 else 8e-3)

    kw = dict(relu_mask=mask.to(cuda, dtype), addend=add.to(cuda, dtype), alpha=0.5)
    d0 = ops.conv2d_dgrad(dyd, wd, ks, 1, (h, w), **kw)
    d1 = ops.conv2d_dgrad(dyd, wd, ks, 1, (h, w), ws=ws, **kw)
    assert ('[split K]' in ops.last_conv_kernel()) == (need_d > 0)
    d2 = ops.conv2d_dgrad(dyd, wd, ks, 1, (h, w), ws=ws, **kw)
    assert torch.equal(d1, d2)
    assert rel_err(d1, gx_ref) <= TOL[dtype]
    assert rel_err(d1, d0.double()) <= (2e-6 if dtype == torch.float32 else 8e-3)
    # the addend may be the output itself
    acc = add.to(cuda, dtype).clone()
    ops.conv2d_dgrad(dyd, wd, ks, 1, (h, w), relu_mask=kw['relu_mask'], addend=acc, alpha=0.5, out=acc, ws=ws)
    assert torch.equal(acc, d1)

    # a workspace too small for the planes: the unsplit launch, same bits as without one
    small = torch.empty(n * h * w * co + 4, device=cuda)
    y3, _ = ops.conv2d_fwd(xd, wf, ks, 1, bias=b32, relu=True, want_pre=True, ws=small)
    assert '[split K]' not in ops.last_conv_kernel() and torch.equal(y3, y0)


def test_split_k_is_for_layers_that_do_not_fill_the_chip(cuda):
    from stmi import ops
    for dtype in DTYPES:
        assert ops.conv_ksplit_ws_floats((4, 64, 64, 256), (3, 3, 256, 256), 1, dtype) == 0      # 256+ tiles
        if dtype == torch.bfloat16:   # the planes of a larger batch cost more than the shorter K loop saves
            assert ops.conv_ksplit_ws_floats((8, 16, 16, 512), (3, 3, 512, 512), 1, dtype) == 0
        assert ops.conv_ksplit_ws_floats((4, 16, 16, 512), (3, 3, 512, 512), 2, dtype) == 0        # stride 2
        assert ops.conv_ksplit_ws_floats((4, 16, 16, 512), (1, 1, 512, 512), 1, dtype) == 0
        assert ops.conv_ksplit_ws_floats((1, 16, 16, 64), (3, 3, 64, 64), 1, dtype) == 0          # one or two chunks


RGB_CASES = [(2, 32, 64, 9, 32, False), (1, 37, 45, 9, 32, False), (2, 16, 32, 3, 64, True), (1, 21, 50, 3, 64, True),
             (1, 9, 11, 3, 32, True), (4, 256, 256, 9, 32, False)]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', RGB_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_rgb_matches_preprocess_then_conv(cuda, dtype, case):
    """The 3-channel kernel = (x * scale - mean, rounded to the compute dtype) -> SAME conv
    [+ bias, + pre-ReLU copy, + ReLU], including tiles clipped at both borders."""
    from stmi import ops
    n, h, w, k, co, vgg_like = case
    x = rnd((n, h, w, 3), 1).abs().clamp(max=1.0)
    wt = rnd((k, k, 3, co), 2, 1.0 / np.sqrt(k * k * 3))
    scale, mean = (127.5, torch.tensor([123.68, 116.78, 103.94])) if vgg_like else (1.0, None)
    bias = rnd((co,), 3) if vgg_like else None
    xin = x.double() * scale - (mean.double() if mean is not None else 0.0)
    if vgg_like:
        xin, wt = xin / 64.0, wt    # keep magnitudes comparable; applied to the kernel through `scale` / `mean`
        scale, mean = scale / 64.0, mean / 64.0
    ref = O.conv2d_same(q(xin, dtype), q(wt, dtype), 1)
    if bias is not None:
        ref = ref + bias.double()
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD_RGB, dtype)
    got = ops.conv2d_rgb_fwd(x.to(cuda, torch.float32), wp, k, co, dtype, scale=scale,
                             mean=None if mean is None else mean.to(cuda, torch.float32),
                             bias=None if bias is None else bias.to(cuda, torch.float32),
                             relu=vgg_like, want_pre=vgg_like)
    if vgg_like:
        y, pre = got
        assert rel_err(pre, ref) <= TOL[dtype]
        assert rel_err(y, torch.relu(ref)) <= TOL[dtype]
    else:
        assert rel_err(got, ref) <= TOL[dtype]
    assert 'conv_rgb_kernel' in ops.last_conv_kernel()


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 24, 40), (1, 37, 45), (4, 256, 256)])
def test_conv_rgb_dgrad_is_the_gradient_of_the_three_channel_output_layer(cuda, dtype, shape):
    from stmi import ops
    n, h, w = shape
    k, ci = 9, 32
    x = rnd((n, h, w, ci), 1)
    wt = rnd((k, k, ci, 3), 2, 1.0 / np.sqrt(k * k * ci))
    dy = rnd((n, h, w, 3), 3)
    xq = q(x, dtype).requires_grad_()
    y = O.conv2d_same(xq, q(wt, dtype), 1)
    (gx,) = torch.autograd.grad(y, xq, q(dy, dtype))
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD_RGB, dtype)
    dx = ops.conv2d_rgb_dgrad(dy.to(cuda, dtype), wp, k, ci)
    assert rel_err(dx, gx) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 16, 16, 128, 64), (1, 9, 7, 64, 32), (1, 1, 5, 64, 64)])
def test_parity_classes_in_one_launch_equal_one_launch_per_class(cuda, dtype, shape, monkeypatch):
    """A stride-2 transposed convolution / backward-data runs its four output parity classes in one
    launch; forcing one launch per class must give the same bits (also for odd sizes, where the
    classes differ in size, and for a one-row image, where two of them are empty)."""
    from stmi import ops
    n, h, w, ci, co = shape
    k, s = 3, 2
    x = rnd((n, h, w, ci), 1).to(cuda, dtype)
    wt = rnd((k, k, co, ci), 2, 1.0 / np.sqrt(k * k * ci)).to(cuda, torch.float32)
    wp = ops.pack_conv_weights(wt, ops.PACK_DGRAD, dtype)
    # three routes for the same call: the LDS-resident-patch kernel (all classes in one pass), the gather-GEMM with
    # the classes as sub-grids of one launch, one gather-GEMM launch per class
    up2 = ops.conv2d_transpose_fwd(x, wp, (k, k, co, ci), s)
    name_default = ops.last_conv_kernel()
    monkeypatch.setenv('STMI_NO_UP2', '1')
    merged = ops.conv2d_transpose_fwd(x, wp, (k, k, co, ci), s)
    name_merged = ops.last_conv_kernel()
    monkeypatch.setenv('STMI_NO_PC_MULTI', '1')
    per_class = ops.conv2d_transpose_fwd(x, wp, (k, k, co, ci), s)
    name_per_class = ops.last_conv_kernel()
    # the knobs really switched the route (the library re-reads them: STMI_ENV_LIVE is set in conftest.py)
    assert 'conv_gemm_pc_kernel' in name_merged and 'per parity class' not in name_merged, name_merged
    assert 'per parity class' in name_per_class, name_per_class
    assert name_default != name_per_class
    assert torch.equal(merged, per_class)
    tol = 1e-5 if dtype == torch.float32 else 2e-2       # another kernel, another summation order
    assert ((up2.double() - merged.double()).abs().max() / merged.double().abs().max()).item() <= tol
    # the same for the backward-data of a stride-2 convolution on an odd-sized input
    hin, win = 2 * h - 1, 2 * w - 1
    dy = rnd((n, h, w, co), 3).to(cuda, dtype)
    w2 = rnd((k, k, ci, co), 4, 1.0 / np.sqrt(k * k * ci)).to(cuda, torch.float32)
    wp2 = ops.pack_conv_weights(w2, ops.PACK_DGRAD, dtype)
    a = ops.conv2d_dgrad(dy, wp2, (k, k, ci, co), s, (hin, win))
    assert 'per parity class' in ops.last_conv_kernel()
    monkeypatch.delenv('STMI_NO_PC_MULTI')
    b = ops.conv2d_dgrad(dy, wp2, (k, k, ci, co), s, (hin, win))
    assert 'per parity class' not in ops.last_conv_kernel()
    assert torch.equal(a, b)


def test_three_channel_kernels_reject_what_they_are_not_built_for(cuda):
    """Unsupported geometries fail loudly with the library's message (callers fall back on the row-patch form
    by asking first); an empty batch is a no-op."""
    from stmi import ops, lib
    x = torch.zeros((1, 8, 8, 3), device=cuda)
    w5 = ops.pack_conv_weights(torch.zeros((5, 5, 3, 32), device=cuda), ops.PACK_FWD_RGB, torch.bfloat16)
    y = torch.empty((1, 8, 8, 32), device=cuda, dtype=torch.bfloat16)
    with pytest.raises(lib.StmiError, match='unsupported'):
        lib.call('stmi_conv2d_rgb_nhwc_fwd', x.data_ptr(), 1.0, None, w5.data_ptr(), None, y.data_ptr(), None,
                 1, 8, 8, 32, 5, 0, lib.BF16, torch.cuda.current_stream().cuda_stream)
    assert not ops.to3_supported(5, 32, torch.bfloat16) and not ops.to3_supported(3, 64, torch.float32)
    assert ops.to3_supported(9, 32, torch.float32) and ops.to3_supported(3, 64, torch.bfloat16)
    w9 = ops.pack_conv_weights(torch.zeros((9, 9, 3, 32), device=cuda), ops.PACK_FWD_RGB, torch.bfloat16)
    empty = ops.conv2d_rgb_fwd(torch.zeros((0, 8, 8, 3), device=cuda), w9, 9, 32, torch.bfloat16)
    assert tuple(empty.shape) == (0, 8, 8, 32)
    wr = ops.pack_conv_weights(torch.zeros((9, 9, 32, 3), device=cuda), ops.PACK_FWD_ROWOUT, torch.bfloat16)
    assert tuple(ops.conv2d_to3(torch.zeros((0, 8, 8, 32), device=cuda, dtype=torch.bfloat16), wr, 9).shape) == (0, 8, 8, 3)
    with pytest.raises(lib.StmiError):   # the pack mode checks its own preconditions
        ops.pack_conv_weights(torch.zeros((9, 9, 4, 32), device=cuda), ops.PACK_FWD_RGB, torch.bfloat16)


TO3_CASES = [(2, 24, 40, 9, 32, False), (1, 37, 45, 9, 32, False), (4, 256, 256, 9, 32, False),
             (2, 16, 60, 3, 64, True), (1, 21, 50, 3, 32, True), (1, 8, 30, 3, 32, False)]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', TO3_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_to3_matches_the_convolution(cuda, dtype, case):
    """One-launch ci -> 3 convolution (row patches on the output side, folded in LDS) against the
    oracle's convolution, with the optional addend and tiles clipped at both borders."""
    from stmi import ops
    n, h, w, k, ci, with_addend = case
    if not ops.to3_supported(k, ci, dtype):
        pytest.skip('no instantiation for this (k, ci, dtype): the caller keeps the two-launch path')
    x = rnd((n, h, w, ci), 1)
    wt = rnd((k, k, ci, 3), 2, 1.0 / np.sqrt(k * k * ci))
    add = rnd((n, h, w, 3), 3) if with_addend else None
    ref = O.conv2d_same(q(x, dtype), q(wt, dtype), 1)
    if add is not None:
        ref = ref + q(add, dtype)
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD_ROWOUT, dtype)
    y = ops.conv2d_to3(x.to(cuda, dtype), wp, k, addend=None if add is None else add.to(cuda, dtype))
    assert rel_err(y, ref) <= TOL[dtype]
    assert 'conv_to3_kernel' in ops.last_conv_kernel()


DECONV_CASES = [(2, 6, 5, 128, 64, 3, 2), (1, 7, 8, 64, 32, 3, 2), (1, 4, 4, 16, 8, 3, 2)]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', DECONV_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_transpose(cuda, dtype, case):
    from stmi import ops
    n, h, w, ci, co, k, s = case
    x = rnd((n, h, w, ci), 1)
    wt = rnd((k, k, co, ci), 2, 1.0 / np.sqrt(k * k * ci))
    xq, wq = q(x, dtype).requires_grad_(), q(wt, dtype).requires_grad_()
    y_ref = O.conv2d_transpose_same(xq, wq, s)
    dy = rnd(tuple(y_ref.shape), 3)
    gx_ref, gw_ref = torch.autograd.grad(y_ref, (xq, wq), q(dy, dtype))
    w32 = wt.to(cuda, torch.float32)
    # the op is the backward-data of a conv whose HWIO weight is W: pack it that way
    w_fwdpath = ops.pack_conv_weights(w32, ops.PACK_DGRAD, dtype)
    w_bwdpath = ops.pack_conv_weights(w32, ops.PACK_FWD, dtype)
    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    y = ops.conv2d_transpose_fwd(xd, w_fwdpath, (k, k, co, ci), s)
    assert rel_err(y, y_ref) <= TOL[dtype]
    dx = ops.conv2d_transpose_dgrad(dyd, w_bwdpath, (k, k, co, ci), s)
    assert rel_err(dx, gx_ref) <= TOL[dtype]
    dw = ops.conv2d_transpose_wgrad(xd, dyd, (k, k, co, ci), s)
    assert rel_err(dw, gw_ref) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 16, 64), (1, 13, 45), (3, 40, 33), (1, 3, 2), (1, 300, 130)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_rgb_conv_tile_statistics_feed_instance_norm(cuda, dtype, case):
    """conv2d_rgb_fwd(tile_stats=) + instance_norm_fwd(tile_stats=) for the stylizer's first layer (9 x 9, 3 -> 32):
    same output as the plain launch; the sums are those of the STORED tensor, so the statistics equal the separate
    statistics pass over it and the normalized result meets the oracle.  Ragged 8 x 32 tiles; more rows than
    the apply blocks fold."""
    from stmi import ops
    n, h, w = case
    k, co = 9, 32
    x = rnd((n, h, w, 3), 1).abs()
    wgt = rnd((k, k, 3, co), 2, 0.05)
    wgt[k // 2, k // 2] += 0.3     # non-zero-mean output
    scale, shift = rnd((co,), 3) + 1.5, rnd((co,), 4)
    xd = x.to(cuda, torch.float32)
    wp = ops.pack_conv_weights(wgt.to(cuda, torch.float32).contiguous(), ops.PACK_FWD_RGB, dtype)
    ts = ops.conv_rgb_tile_stats_tiles(h, w, co, k)
    assert ts == -(-h // 8) * -(-w // 32)
    tiles = torch.full((n, ts, 4, co), float('nan'), dtype=torch.float32, device=cuda)
    y_conv = ops.conv2d_rgb_fwd(xd, wp, k, co, dtype, tile_stats=tiles)
    assert torch.equal(y_conv, ops.conv2d_rgb_fwd(xd, wp, k, co, dtype)), 'the statistics must not change the output'
    assert bool(torch.isfinite(tiles).all())
    assert tiles[:, :, 3].sum(dim=1).eq(h * w).all(), 'tile pixel counts add up to the plane'
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    y_a, st_a = ops.instance_norm_fwd(y_conv, sd, hd, ops.ACT_RELU, tile_stats=tiles)
    y_b, st_b = ops.instance_norm_fwd(y_conv, sd, hd, ops.ACT_RELU)
    assert rel_err(st_a[..., 0], st_b[..., 0]) <= 1e-5
    assert rel_err(st_a[..., 1], st_b[..., 1]) <= 1e-4
    conv_ref = O.conv2d_same(q(x, dtype), q(wgt, dtype), 1)
    ref = torch.relu(O.instance_norm(conv_ref, scale.float().double(), shift.float().double()))
    assert rel_err(y_a, ref) <= TOL[dtype]
    assert rel_err(y_b, ref) <= TOL[dtype]


UP_STAT_CASES = [(2, 6, 5, 128, 64), (1, 7, 40, 64, 32), (3, 9, 70, 128, 64), (1, 1, 1, 64, 32), (1, 135, 130, 64, 32)]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', UP_STAT_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_transpose_tile_statistics_feed_instance_norm(cuda, dtype, case):
    """conv2d_transpose_fwd(tile_stats=) + instance_norm_fwd(tile_stats=): same output as the plain launch; the
    sums are those of the STORED tensor, so the statistics equal the separate statistics pass over it (to fp32
    summation order) and the normalized result meets the oracle.  Ragged 4 x 32 input tiles, one-pixel input,
    more rows than the apply blocks fold."""
    from stmi import ops
    n, h, w, ci, co = case
    k, s = 3, 2
    x = rnd((n, h, w, ci), 1) * 0.5 + 0.8
    wt = rnd((k, k, co, ci), 2, 1.0 / np.sqrt(k * k * ci))
    wt[1, 1] += 0.05           # non-zero-mean output
    scale, shift = rnd((co,), 3) + 1.5, rnd((co,), 4)
    xd = x.to(cuda, dtype)
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    ts = ops.conv_transpose_tile_stats_tiles(xd.shape, (k, k, co, ci), s, dtype)
    assert ts == -(-h // 4) * -(-w // 32), 'this geometry runs on the up-sampling kernel'
    tiles = torch.full((n, ts, 4, co), float('nan'), dtype=torch.float32, device=cuda)
    y = ops.conv2d_transpose_fwd(xd, wp, (k, k, co, ci), s, tile_stats=tiles)
    assert torch.equal(y, ops.conv2d_transpose_fwd(xd, wp, (k, k, co, ci), s)), 'the statistics must not change the output'
    assert bool(torch.isfinite(tiles).all())
    assert tiles[:, :, 3].sum(dim=1).eq(4 * h * w).all(), 'tile pixel counts add up to the plane'
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    y_a, st_a = ops.instance_norm_fwd(y, sd, hd, ops.ACT_RELU, tile_stats=tiles)
    y_b, st_b = ops.instance_norm_fwd(y, sd, hd, ops.ACT_RELU)
    assert rel_err(st_a[..., 0], st_b[..., 0]) <= 1e-5
    assert rel_err(st_a[..., 1], st_b[..., 1]) <= 1e-4
    y_ref = O.conv2d_transpose_same(q(x, dtype), q(wt, dtype), s)
    ref = torch.relu(O.instance_norm(y_ref, scale.float().double(), shift.float().double()))
    assert rel_err(y_a, ref) <= TOL[dtype]
    assert rel_err(y_b, ref) <= TOL[dtype]


DN2_CASES = [
    # n, h, w, ci, co: the stylizer's down-sampling layers (3x3, stride 2) on even image sides
    (2, 16, 64, 32, 64),       # whole 4 x 32 tiles
    (1, 10, 70, 64, 128),      # ragged right and bottom tiles (output 5 x 35)
    (3, 2, 2, 32, 64),         # one output pixel
    (1, 72, 136, 64, 128),     # more tile rows than the apply blocks fold (18 x 3 tiles)
    (2, 12, 20, 16, 64),       # fp32: one chunk per pixel pair
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', DN2_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_stride2_patch_kernel_and_its_tile_statistics(cuda, dtype, case):
    """model_utilities.py:236-237 with stride 2: the LDS-resident pair-patch kernel against the oracle and against
    the gather-GEMM form; with tile_stats= the output is unchanged, the sums are those of the stored tensor (equal to
    the separate statistics pass to fp32 summation order) and the normalized result meets the oracle."""
    from stmi import ops
    n, h, w, ci, co = case
    k, s = 3, 2
    ks = (k, k, ci, co)
    x = rnd((n, h, w, ci), 1) * 0.5 + 0.8
    wt = rnd(ks, 2, 1.0 / np.sqrt(k * k * ci))
    wt[1, 1] += 0.05           # non-zero-mean output
    scale, shift = rnd((co,), 3) + 1.5, rnd((co,), 4)
    xd = x.to(cuda, dtype)
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    mine = ci * (2 if dtype == torch.bfloat16 else 4) * 2 % 128 == 0
    y = ops.conv2d_fwd(xd, wp, ks, s)
    assert ('conv_dn2_kernel' in ops.last_conv_kernel()) == mine, ops.last_conv_kernel()
    y_ref = O.conv2d_same(q(x, dtype), q(wt, dtype), s)
    assert rel_err(y, y_ref) <= TOL[dtype]
    ts = ops.conv_tile_stats_tiles(xd.shape, ks, s, dtype)
    if not mine:
        assert ts == 0
        return
    pw = 16 if (dtype == torch.float32 and ci == 64) else 32
    assert ts == -(-(h // 2) // 4) * -(-(w // 2) // pw)
    assert ops.conv_tile_stats_free(xd.shape, ks, s, dtype)
    tiles = torch.full((n, ts, 4, co), float('nan'), dtype=torch.float32, device=cuda)
    y_t = ops.conv2d_fwd(xd, wp, ks, s, tile_stats=tiles)
    assert torch.equal(y_t, y), 'the statistics must not change the output'
    assert bool(torch.isfinite(tiles).all())
    assert tiles[:, :, 3].sum(dim=1).eq((h // 2) * (w // 2)).all(), 'tile pixel counts add up to the plane'
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    y_a, st_a = ops.instance_norm_fwd(y, sd, hd, ops.ACT_RELU, tile_stats=tiles)
    y_b, st_b = ops.instance_norm_fwd(y, sd, hd, ops.ACT_RELU)
    assert rel_err(st_a[..., 0], st_b[..., 0]) <= 1e-5
    assert rel_err(st_a[..., 1], st_b[..., 1]) <= 1e-4
    ref = torch.relu(O.instance_norm(y_ref, scale.float().double(), shift.float().double()))
    assert rel_err(y_a, ref) <= TOL[dtype]
    assert rel_err(y_b, ref) <= TOL[dtype]


def test_stride2_patch_kernel_equals_the_gather_form(cuda, monkeypatch):
    from stmi import ops
    for dtype in DTYPES:
        x, wt = rnd((2, 20, 44, 64), 1), rnd((3, 3, 64, 128), 2, 0.05)
        xd = x.to(cuda, dtype)
        wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD, dtype)
        y = ops.conv2d_fwd(xd, wp, (3, 3, 64, 128), 2)
        assert 'conv_dn2_kernel' in ops.last_conv_kernel()
        monkeypatch.setenv('STMI_NO_DN2', '1')
        y_g = ops.conv2d_fwd(xd, wp, (3, 3, 64, 128), 2)
        assert 'conv_dn2_kernel' not in ops.last_conv_kernel()
        monkeypatch.delenv('STMI_NO_DN2')
        assert rel_err(y, y_g.double()) <= (2e-6 if dtype == torch.float32 else 8e-3)


PERS_CASES = [(3, 130, 250, 128, 64), (3, 126, 260, 128, 128), (2, 254, 250, 64, 64)]   # three block rounds and more


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', PERS_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_patch_persistent(cuda, dtype, case):
    """3x3 stride-1 layers of two and more block rounds per CU run on the persistent patch kernel (a block walks
    a list of tiles, the next tile's loads issued under the current tile's epilogue): forward with bias, pre-ReLU
    copy and ReLU, and backward-data with mask and addend, against the oracle; clipped tiles on both borders."""
    from stmi import ops
    n, h, w, ci, co = case
    if ci * (4 if dtype == torch.float32 else 2) // 128 < 2:
        pytest.skip('a single 128-byte chunk: not a persistent-kernel layer')
    x = rnd((n, h, w, ci), 1)
    wt = rnd((3, 3, ci, co), 2, 1.0 / np.sqrt(9 * ci))
    bias = rnd((co,), 3)
    ref_pre = O.conv2d_same(q(x, dtype), q(wt, dtype), 1) + bias
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    y, y_pre = ops.conv2d_fwd(x.to(cuda, dtype), wp, (3, 3, ci, co), 1, bias=bias.to(cuda, torch.float32), relu=True,
                              want_pre=True)
    name = ops.last_conv_kernel()
    if dtype == torch.bfloat16 or co < 128:     # (fp32 128-channel tiles keep the one-tile form: registers)
        assert 'conv_patch_pers_kernel' in name, name
    assert rel_err(y_pre, ref_pre) <= TOL[dtype]
    assert rel_err(y, ref_pre.clamp_min(0)) <= TOL[dtype]
    # backward-data of the same layer: gathers dy [n,h,w,co], writes dx [n,h,w,ci]
    dy, mask, add = rnd((n, h, w, co), 4), rnd((n, h, w, ci), 5), rnd((n, h, w, ci), 6)
    xq = torch.zeros((n, h, w, ci), dtype=torch.float64, requires_grad=True)
    (gx,) = torch.autograd.grad(O.conv2d_same(xq, q(wt, dtype), 1), xq, q(dy, dtype))
    ref = 0.5 * gx * (q(mask, dtype) > 0) + q(add, dtype)
    wd = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    got = ops.conv2d_dgrad(dy.to(cuda, dtype), wd, (3, 3, ci, co), 1, (h, w), relu_mask=mask.to(cuda, dtype),
                           addend=add.to(cuda, dtype), alpha=0.5)
    assert rel_err(got, ref) <= TOL[dtype]


UP2_CASES = [(2, 9, 37, 64, 32, 18, 74),     # (n, h, w, ci, co, out_h, out_w): transposed conv 64 -> 32, clipped tiles
             (1, 8, 33, 128, 64, 16, 66),   # 128 -> 64
             (2, 5, 40, 128, 64, 9, 79),    # odd output: backward-data of a stride-2 conv over a 9 x 79 input
             (1, 4, 32, 64, 64, 8, 64),     # exactly one tile
             (1, 70, 50, 64, 32, 140, 100)]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', UP2_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_stride2_dgrad_patch_kernel(cuda, dtype, case):
    """Backward-data of a 3x3 stride-2 SAME convolution (= the transposed convolution of the stylizer's
    up-sampling blocks) on the LDS-resident-patch kernel, against the oracle's autograd."""
    from stmi import ops
    n, h, w, cg, cx, oh, ow = case       # gathered tensor dy [n,h,w,cg], written tensor dx [n,oh,ow,cx]
    k, s = 3, 2
    wt = rnd((k, k, cx, cg), 2, 1.0 / np.sqrt(k * k * cg))      # HWIO of the forward conv cx -> cg
    dy = rnd((n, h, w, cg), 3)
    xq = torch.zeros((n, oh, ow, cx), dtype=torch.float64, requires_grad=True)
    y_ref = O.conv2d_same(xq, q(wt, dtype), s)
    assert tuple(y_ref.shape) == (n, h, w, cg)
    (ref,) = torch.autograd.grad(y_ref, xq, q(dy, dtype))
    wd = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    dyd = dy.to(cuda, dtype)
    got = ops.conv2d_dgrad(dyd, wd, (k, k, cx, cg), s, (oh, ow), alpha=0.5)
    assert 'conv_up2_kernel' in ops.last_conv_kernel(), ops.last_conv_kernel()
    assert rel_err(got, 0.5 * ref) <= TOL[dtype]


IN_CASES = [(2, 9, 7, 32, 'relu', False), (2, 16, 16, 128, 'none', True), (1, 12, 10, 3, 'sigmoid', False),
            (1, 40, 30, 64, 'relu', False), (1, 64, 48, 128, 'relu', False),
            (4, 64, 64, 128, 'none', True),      # the residual-block shape of the training step
            (2, 128, 96, 32, 'sigmoid', False),
            (3, 70, 50, 3, 'sigmoid', False),    # scalar fold (odd channel count), several blocks
            (1, 600, 700, 32, 'none', True)]    # a large plane: the maximum number of partial rows per sample


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', IN_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_instance_norm_fwd_bwd(cuda, dtype, case):
    from stmi import ops
    n, h, w, c, act, use_res = case
    actc = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'sigmoid': ops.ACT_SIGMOID}[act]
    x = rnd((n, h, w, c), 1) * 3 + 2.0      # non-zero mean exercises the shifted sums
    scale, shift = rnd((c,), 2) + 1.5, rnd((c,), 3)
    res = rnd((n, h, w, c), 4) if use_res else None
    dy = rnd((n, h, w, c), 5)
    xq = q(x, dtype).requires_grad_()
    sq, hq = scale.float().double().requires_grad_(), shift.float().double().requires_grad_()
    z = O.instance_norm(xq, sq, hq)
    y_ref = {'none': z, 'relu': torch.relu(z), 'sigmoid': torch.sigmoid(z)}[act]
    if use_res:
        y_ref = y_ref + q(res, dtype)
    gx, gs, gh = torch.autograd.grad(y_ref, (xq, sq, hq), q(dy, dtype))

    xd = x.to(cuda, dtype)
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    y, stats, y32 = ops.instance_norm_fwd(xd, sd, hd, actc, residual=None if res is None else res.to(cuda, dtype),
                                          want_f32=True)
    assert rel_err(y32, y_ref) <= 1e-3          # fp32 copy is exact up to statistics
    assert rel_err(y, y_ref) <= TOL[dtype]
    dscale = torch.zeros(c, device=cuda)
    dshift = torch.zeros(c, device=cuda)
    dx = ops.instance_norm_bwd(xd, dy.to(cuda, dtype), stats, sd, hd, dscale, dshift, actc)
    assert rel_err(dx, gx) <= TOL[dtype]
    assert rel_err(dscale, gs) <= 1e-3
    assert rel_err(dshift, gh) <= 1e-3


TILE_STAT_CASES = [
    # n, h, w, ci, co: 3x3 stride-1 layers of the LDS-resident patch kernel; ragged tiles in both directions,
    # one tile per sample, many tiles (a finalize launch instead of the fold inside the apply blocks)
    (2, 16, 32, 128, 128), (1, 5, 70, 64, 64), (2, 11, 40, 128, 64), (1, 3, 2, 64, 64), (3, 64, 64, 128, 128),
    (1, 70, 150, 64, 128),
    (1, 270, 480, 128, 128),    # one sample alone fills three block rounds: the persistent kernel leaves the rows
    (2, 200, 330, 64, 64),      # ... (fp32: two 128-byte chunks, four row slices per tile)
    (1, 250, 410, 128, 128),    # ... clipped on both borders (bf16: the wave-private epilogue, two rows per tile)
    (1, 250, 410, 128, 96),     # ... and a channel count that does not fill the last 128-channel tile
    (2, 30, 40, 128, 96),
]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', TILE_STAT_CASES, ids=lambda c: 'x'.join(map(str, c)))
def test_conv_tile_statistics_feed_instance_norm(cuda, dtype, case):
    """conv2d_fwd(tile_stats=) + instance_norm_fwd(tile_stats=) against conv -> statistics pass -> apply, and
    against the oracle.  A large mean next to the deviation exercises the shifted tile sums."""
    from stmi import ops
    n, h, w, ci, co = case
    x = rnd((n, h, w, ci), 1)
    wgt = rnd((3, 3, ci, co), 2, 0.05)
    wgt[1, 1] += 0.2          # strong centre tap on a non-zero-mean input: |mean| of the output >> its deviation
    x = x * 0.3 + 1.0
    scale, shift = rnd((co,), 3) + 1.5, rnd((co,), 4)
    xd = x.to(cuda, dtype)
    wp = ops.pack_conv_weights(wgt.to(cuda, torch.float32).contiguous(), ops.PACK_FWD, dtype)
    ts = ops.conv_tile_stats_tiles(xd.shape, (3, 3, ci, co), 1, dtype)
    assert ts > 0, 'this geometry runs on the patch kernel'
    tiles = torch.full((n, ts, 4, co), float('nan'), dtype=torch.float32, device=cuda)
    y_conv = ops.conv2d_fwd(xd, wp, (3, 3, ci, co), 1, tile_stats=tiles)
    kname = ops.last_conv_kernel()
    assert torch.equal(y_conv, ops.conv2d_fwd(xd, wp, (3, 3, ci, co), 1)), 'the statistics must not change the output'
    assert bool(torch.isfinite(tiles).all())
    assert tiles[:, :, 3].sum(dim=1).eq(h * w).all(), 'tile pixel counts add up to the plane'
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    y_a, st_a = ops.instance_norm_fwd(y_conv, sd, hd, ops.ACT_RELU, tile_stats=tiles)
    y_b, st_b = ops.instance_norm_fwd(y_conv, sd, hd, ops.ACT_RELU)
    # The tile path takes the statistics of the fp32 accumulators (fp32, and bf16 tiles narrower than 64
    # channels) or of the stored, rounded tensor (bf16: the MFMA waves' own epilogue sums the rows it
    # stores); the separate pass always those of the stored tensor.  The first is held to the oracle
    # tightly, the stored kind as far as bf16 storage allows (with |mean| >> deviation the rounding of the
    # stored values shows in the variance) and to the separate pass.
    conv_ref = O.conv2d_same(q(x, dtype), q(wgt, dtype), 1)
    mean_ref = conv_ref.mean(dim=(1, 2))
    rstd_ref = 1.0 / torch.sqrt(conv_ref.var(dim=(1, 2), unbiased=False) + 1e-3)
    import re
    mt = re.search(r'(conv_patch\w*)<__bf16, ([^>]*)>', kname)
    stored = False
    if mt is not None:
        args = [a.strip() for a in mt.group(2).split(',')]
        if mt.group(1) == 'conv_patch_kernel':      # T, TC, PW, NSW, ONE: 64- / 128-channel tiles of several chunks
            stored = args[0] in ('2', '4') and args[3] == 'false'
        else:                                       # persistent: T, TC, PW, NSW, STATS, WEPI
            stored = args[4] == 'true'
    if stored:
        assert rel_err(st_a[..., 0], st_b[..., 0]) <= 1e-5
        assert rel_err(st_a[..., 1], st_b[..., 1]) <= 1e-4
    assert rel_err(st_a[..., 0], mean_ref) <= (2e-3 if stored else 1e-4)
    assert rel_err(st_a[..., 1], rstd_ref) <= (5e-2 if stored else 1e-3)
    assert rel_err(st_b[..., 0], mean_ref) <= (1e-4 if dtype == torch.float32 else 2e-3)
    assert rel_err(st_b[..., 1], rstd_ref) <= (1e-3 if dtype == torch.float32 else 5e-2)
    ref = torch.relu(O.instance_norm(conv_ref, scale.float().double(), shift.float().double()))
    assert rel_err(y_a, ref) <= TOL[dtype]
    assert rel_err(y_b, ref) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 16, 32, 128, 'relu', True), (1, 5, 70, 64, 'none', False),
                                  (2, 11, 40, 64, 'sigmoid', True), (3, 64, 64, 128, 'relu', False)],
                         ids=lambda c: 'x'.join(map(str, c)))
def test_dgrad_sums_feed_instance_norm_backward(cuda, dtype, case):
    """conv2d_dgrad(in_bwd=) + instance_norm_bwd(rows=) against the plain data gradient followed by the
    three-pass instance-norm backward, and both against the oracle's autograd."""
    from stmi import ops
    n, h, w, c, act, use_addend = case
    actc = {'none': ops.ACT_NONE, 'relu': ops.ACT_RELU, 'sigmoid': ops.ACT_SIGMOID}[act]
    co = 64 if c == 64 else 128
    x = rnd((n, h, w, c), 1) * 2 + 0.7            # input of the norm below
    scale, shift = rnd((c,), 2) + 1.5, rnd((c,), 3)
    wgt = rnd((3, 3, c, co), 4, 0.05)             # the conv above the norm: y2 = conv(act(IN(x)))
    dy2 = rnd((n, h, w, co), 5)
    addend = rnd((n, h, w, c), 6) if use_addend else None
    # oracle: gradient w.r.t. x of  sum(conv(act(IN(x))) * dy2) + sum(act(IN(x)) * addend)
    xq = q(x, dtype).requires_grad_()
    sq, hq = scale.float().double().requires_grad_(), shift.float().double().requires_grad_()
    z = O.instance_norm(xq, sq, hq)
    y = {'none': z, 'relu': torch.relu(z), 'sigmoid': torch.sigmoid(z)}[act]
    tot = (O.conv2d_same(y, q(wgt, dtype), 1) * q(dy2, dtype)).sum()
    if use_addend:
        tot = tot + (y * q(addend, dtype)).sum()
    gx, gs, gh = torch.autograd.grad(tot, (xq, sq, hq))

    xd = x.to(cuda, dtype)
    sd, hd = scale.to(cuda, torch.float32), shift.to(cuda, torch.float32)
    _, stats = ops.instance_norm_fwd(xd, sd, hd, actc)
    wp = ops.pack_conv_weights(wgt.to(cuda, torch.float32).contiguous(), ops.PACK_DGRAD, dtype)
    ts = ops.conv_dgrad_in_sums_tiles((h, w), n, (3, 3, c, co), 1, dtype)
    assert ts > 0
    rows = torch.full((n, ts, 2, c), float('nan'), dtype=torch.float32, device=cuda)
    ad = None if addend is None else addend.to(cuda, dtype)
    dyd = dy2.to(cuda, dtype)
    g_a = ops.conv2d_dgrad(dyd, wp, (3, 3, c, co), 1, (h, w), addend=ad, in_bwd=(xd, stats, sd, hd, actc, rows))
    g_b = ops.conv2d_dgrad(dyd, wp, (3, 3, c, co), 1, (h, w), addend=ad)
    assert torch.equal(g_a, g_b), 'the sums must not change the data gradient'
    assert bool(torch.isfinite(rows).all())
    res = {}
    for name, kw in (('rows', dict(rows=rows)), ('passes', dict())):
        dscale, dshift = torch.zeros(c, device=cuda), torch.zeros(c, device=cuda)
        dx = ops.instance_norm_bwd(xd, g_a, stats, sd, hd, dscale, dshift, actc, **kw)
        res[name] = (dx, dscale, dshift)
        assert rel_err(dx, gx) <= (2e-3 if dtype == torch.float32 else 4e-2), name
        assert rel_err(dscale, gs) <= (1e-3 if dtype == torch.float32 else 2e-2), name
        assert rel_err(dshift, gh) <= (1e-3 if dtype == torch.float32 else 2e-2), name
    # the two routes agree with each other more closely than either does with the oracle (the rows are sums
    # of the fp32 accumulators, the passes read the rounded tensor)
    assert rel_err(res['rows'][1], res['passes'][1]) <= (1e-4 if dtype == torch.float32 else 1e-2)


def test_instance_norm_workspace_is_reusable(cuda):
    """One workspace serves calls of different shapes in a row, and a repeated call gives the same bits."""
    from stmi import ops
    ws = ops.instance_norm_ws(4, 64, 64, 128, cuda)
    outs = []
    for shape in [(4, 64, 64, 128), (2, 33, 47, 64), (4, 64, 64, 128), (1, 50, 50, 3)]:
        n, h, w, c = shape
        x = (rnd(shape, 7) * 2 + 1).to(cuda, torch.bfloat16)
        sc, sh = (rnd((c,), 8) + 1.5).to(cuda, torch.float32), rnd((c,), 9).to(cuda, torch.float32)
        y, stats = ops.instance_norm_fwd(x, sc, sh, ops.ACT_RELU, ws=ws)
        ds, dh = torch.zeros(c, device=cuda), torch.zeros(c, device=cuda)
        dx = ops.instance_norm_bwd(x, y, stats, sc, sh, ds, dh, ops.ACT_RELU, ws=ws)
        assert bool(torch.isfinite(dx.float()).all()) and bool(torch.isfinite(stats).all())
        outs.append((y.clone(), dx.clone()))
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 8, 8, 64), (1, 7, 5, 16), (1, 127, 3, 8), (2, 5, 6, 3)])
def test_maxpool(cuda, dtype, shape):
    from stmi import ops
    x = torch.relu(rnd(shape, 1))
    xq = q(x, dtype).requires_grad_()
    y_ref = O.max_pool_2x2_same(xq)
    dy = rnd(tuple(y_ref.shape), 2)
    (gx,) = torch.autograd.grad(y_ref, xq, q(dy, dtype))
    gx = gx * (xq > 0)       # fused ReLU gradient of the producing layer
    xd = x.to(cuda, dtype)
    y = ops.maxpool_fwd(xd)
    assert rel_err(y, y_ref) == 0.0
    dx = ops.maxpool_bwd(xd, dy.to(cuda, dtype))
    assert rel_err(dx, gx) <= 1e-6


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 16, 16, 64), (1, 9, 7, 128), (1, 4, 4, 512), (2, 33, 17, 8), (1, 6, 6, 16)])
def test_gram_fwd_bwd(cuda, dtype, shape):
    from stmi import ops
    n, h, w, c = shape
    f = rnd(shape, 1)
    target = rnd((1, c, c), 2)
    fq = q(f, dtype).requires_grad_()
    g_ref = O.gram_matrix(fq)
    mult = n * c * c * h * w / 16.0          # keeps the gradient O(1) so the check has teeth
    loss_ref = ((g_ref - target) ** 2).mean() * mult
    (gf,) = torch.autograd.grad(loss_ref, fq)
    fd = f.to(cuda, dtype)
    g = ops.gram_fwd(fd)
    assert rel_err(g, g_ref) <= (1e-3 if dtype == torch.float32 else 1e-2)
    slot = ops.loss_slots(1, cuda)[0]
    ls = mult / (n * c * c)
    s = ops.gram_mse_fwd_bwd(g, target.to(cuda, torch.float32), slot, ls, 4.0 * ls / (h * w), dtype)
    loss = ops.slot_total(slot)
    assert abs(loss.item() - loss_ref.item()) <= (1e-3 if dtype == torch.float32 else 3e-2) * abs(loss_ref.item())
    df = torch.ones(shape, device=cuda, dtype=dtype)
    ops.gram_bwd(fd, s, df, accumulate=True)
    assert rel_err(df, gf + 1.0) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('chans', [(128, 128), (64, 32), (24, 40)], ids=['lds_dma_128', 'lds_dma_64x32', 'register_staged'])
def test_wgrad_and_gram_pixel_slices_fold_in_fixed_order(cuda, dtype, chans):
    """Geometries whose output tiles are cut into several pixel slices (a workspace is needed): the result is
    the same bits on every run, `accumulate` adds to what the output holds, and a caller workspace works."""
    from stmi import ops
    ci, co = chans
    n, h, w, k = 4, 40, 36, 3
    x, dy = rnd((n, h, w, ci), 1), rnd((n, h, w, co), 2)
    xq = q(x, dtype)
    wz = torch.zeros((k, k, ci, co), dtype=torch.float64, requires_grad=True)
    (gw_ref,) = torch.autograd.grad(O.conv2d_same(xq, wz, 1), wz, q(dy, dtype))
    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    need = ops.conv2d_wgrad_ws_floats(xd.shape, (k, k, ci, co), 1, dtype)
    assert need > 0, 'this geometry is meant to need pixel slices'
    ws = torch.full((need + 8,), float('nan'), device=cuda)   # stale workspace contents must not matter
    first = ops.conv2d_wgrad(xd, dyd, (k, k, ci, co), 1, ws=ws)
    assert rel_err(first, gw_ref) <= TOL[dtype]
    expect = {(128, 128): 'wgrad_glds_kernel', (64, 32): 'wgrad_glds_kernel', (24, 40): 'wgrad_kernel'}[chans]
    assert ops.last_wgrad_kernel() == expect
    for _ in range(3):
        assert torch.equal(ops.conv2d_wgrad(xd, dyd, (k, k, ci, co), 1), first)
    acc = torch.full_like(first, 2.0)
    ops.conv2d_wgrad(xd, dyd, (k, k, ci, co), 1, out=acc, accumulate=True, ws=ws)
    assert rel_err(acc - 2.0, gw_ref) <= TOL[dtype]
    with pytest.raises(ValueError):
        ops.conv2d_wgrad(xd, dyd, (k, k, ci, co), 1, ws=ws[:need - 1])
    # Gram matrix of the same tensor: one sample's 1440 pixels in several slices
    g_ref = O.gram_matrix(xq)
    gneed = ops.gram_ws_floats(xd.shape, dtype)
    g1 = ops.gram_fwd(xd)
    assert rel_err(g1, g_ref) <= (1e-3 if dtype == torch.float32 else 1e-2)
    for _ in range(3):
        assert torch.equal(ops.gram_fwd(xd, ws=torch.empty(max(gneed, 4), device=cuda)), g1)
    g2 = torch.full_like(g1, -1.0)
    ops.gram_fwd(xd, out=g2, accumulate=True)
    assert rel_err(g2 + 1.0, g_ref) <= (1e-3 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('case', [(4, 64, 64, 128, 128), (2, 8, 32, 32, 64), (3, 6, 128, 64, 32), (2, 3, 256, 32, 32), (2, 5, 64, 32, 32),
                                  (1, 4, 96, 96, 32)], ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_wgrad_with_the_patch_resident(cuda, case, dtype, monkeypatch):
    """3x3 stride-1 weight gradients in bf16 on rows that are multiples of 32 pixels (the residual layers of
    full_graph_stylizer_network.py:52-57 on large training crops): a block keeps the rows' halo patch in LDS and
    reads it under all nine taps.  Against the fp64 oracle, the same bits on every run; `accumulate`; samples whose
    halo rows must not leak into each other.  (The kernel is chosen from 32768 pixels on; here at any size.)"""
    from stmi import ops
    monkeypatch.setenv('STMI_WG_RES_MIN_PIX', '0')
    n, h, w, ci, co = case
    x, dy = rnd((n, h, w, ci), 1), rnd((n, h, w, co), 2)
    wz = torch.zeros((3, 3, ci, co), dtype=torch.float64, requires_grad=True)
    (gw_ref,) = torch.autograd.grad(O.conv2d_same(q(x, dtype), wz, 1), wz, q(dy, dtype))
    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    first = ops.conv2d_wgrad(xd, dyd, (3, 3, ci, co), 1)
    # (an fp32 row of 256 pixels with its halo is more than the LDS buffers hold)
    assert ops.last_wgrad_kernel() == ('wgrad_glds_kernel' if dtype == torch.float32 and w > 128 else 'wgrad_res_kernel')
    assert rel_err(first, gw_ref) <= TOL[dtype]
    for _ in range(3):
        assert torch.equal(ops.conv2d_wgrad(xd, dyd, (3, 3, ci, co), 1), first)
    need = ops.conv2d_wgrad_ws_floats(xd.shape, (3, 3, ci, co), 1, dtype)
    ws = torch.full((max(need, 4) + 8,), float('nan'), device=cuda)   # stale workspace contents must not matter
    acc = torch.full_like(first, 2.0)
    ops.conv2d_wgrad(xd, dyd, (3, 3, ci, co), 1, out=acc, accumulate=True, ws=ws)
    assert rel_err(acc - 2.0, gw_ref) <= TOL[dtype]


def test_wgrad_patch_resident_falls_back(cuda):
    """Rows too long for the LDS or not multiples of 32 pixels and strided layers keep the tap-per-block kernel, and
    so do small bf16 problems unless asked; large ones, and fp32 ones of any size, take the resident-patch kernel."""
    from stmi import ops
    xs, dys = rnd((4, 64, 64, 32), 1).to(cuda, torch.bfloat16), rnd((4, 64, 64, 32), 2).to(cuda, torch.bfloat16)
    small = ops.conv2d_wgrad(xs, dys, (3, 3, 32, 32), 1)
    assert ops.last_wgrad_kernel() == 'wgrad_glds_kernel'
    xl, dyl = xs.repeat(2, 1, 1, 1), dys.repeat(2, 1, 1, 1)       # 32768 pixels: twice the same sums
    large = ops.conv2d_wgrad(xl, dyl, (3, 3, 32, 32), 1)
    assert ops.last_wgrad_kernel() == 'wgrad_res_kernel'
    assert rel_err(large, 2 * small) <= TOL[torch.bfloat16]
    ops.conv2d_wgrad(xs.float(), dys.float(), (3, 3, 32, 32), 1)
    assert ops.last_wgrad_kernel() == 'wgrad_res_kernel'
    for (n, h, w, ci, co, s, dtype) in [(32, 4, 288, 32, 32, 1, torch.bfloat16), (16, 64, 36, 32, 32, 1, torch.bfloat16),
                                        (8, 64, 36, 32, 32, 1, torch.float32), (8, 128, 64, 32, 32, 2, torch.bfloat16)]:
        x = rnd((n, h, w, ci), 1).to(cuda, dtype)
        dy = rnd((n, (h + s - 1) // s, (w + s - 1) // s, co), 2).to(cuda, dtype)
        wz = torch.zeros((3, 3, ci, co), dtype=torch.float64, requires_grad=True)
        (gw_ref,) = torch.autograd.grad(O.conv2d_same(q(x.cpu().float(), dtype), wz, s), wz, q(dy.cpu().float(), dtype))
        got = ops.conv2d_wgrad(x, dy, (3, 3, ci, co), s)
        assert ops.last_wgrad_kernel() == 'wgrad_glds_kernel'
        assert rel_err(got, gw_ref) <= TOL[dtype]


def test_wgrad_without_workspace_is_refused(cuda):
    """Through the C-ABI: a geometry that needs pixel slices, called with ws = NULL, is an error, not a fault."""
    from stmi import lib, ops
    x = torch.zeros((4, 40, 36, 128), device=cuda)
    dy = torch.zeros((4, 40, 36, 128), device=cuda)
    dw = torch.zeros((3, 3, 128, 128), device=cuda)
    assert ops.conv2d_wgrad_ws_floats(x.shape, (3, 3, 128, 128), 1, torch.float32) > 0
    with pytest.raises(lib.StmiError, match='workspace'):
        lib.call('stmi_conv2d_nhwc_wgrad', x.data_ptr(), dy.data_ptr(), dw.data_ptr(), None, 4, 40, 36, 128, 128, 3, 3, 1,
                 0, lib.F32, torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_instance_norm_param_grads_fixed_order(cuda, dtype):
    """dscale / dshift from the backward's partial rows as a launch of its own: equal to what the backward call
    adds itself, identical on every run, and batch-order independent in value only up to rounding."""
    from stmi import ops
    n, h, w, c = 4, 24, 20, 64
    x, dy = rnd((n, h, w, c), 1), rnd((n, h, w, c), 2)
    scale, shift = rnd((c,), 3).to(cuda, torch.float32), rnd((c,), 4).to(cuda, torch.float32)
    xd, dyd = x.to(cuda, dtype), dy.to(cuda, dtype)
    _, stats = ops.instance_norm_fwd(xd, scale, shift, ops.ACT_RELU)
    ds_a, dh_a = torch.zeros(c, device=cuda), torch.zeros(c, device=cuda)
    dx_a = ops.instance_norm_bwd(xd, dyd, stats, scale, shift, ds_a, dh_a, ops.ACT_RELU)
    ws = ops.instance_norm_ws(n, h, w, c, cuda)
    dx_b = ops.instance_norm_bwd(xd, dyd, stats, scale, shift, None, None, ops.ACT_RELU, ws=ws)
    rows = ws[:n * ops.instance_norm_bwd_rows_count(xd.shape, dtype) * 2 * c]
    ds_b, dh_b = torch.full((c,), 9.0, device=cuda), torch.full((c,), 9.0, device=cuda)
    ops.instance_norm_param_grads(rows, n, c, ds_b, dh_b)
    assert torch.equal(dx_a, dx_b) and torch.equal(ds_a, ds_b) and torch.equal(dh_a, dh_b)
    ops.instance_norm_param_grads(rows, n, c, ds_b, dh_b, accumulate=True)
    assert rel_err(ds_b, 2.0 * ds_a) <= 1e-6 and rel_err(dh_b, 2.0 * dh_a) <= 1e-6


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
def test_losses(cuda, dtype):
    from stmi import ops
    a, b = rnd((2, 4, 4, 32), 1), rnd((2, 4, 4, 32), 2)
    aq = q(a, dtype).requires_grad_()
    ref = ((aq - q(b, dtype)) ** 2).mean() * 3.0
    (ga,) = torch.autograd.grad(ref, aq)
    slot = ops.loss_slots(1, cuda)[0]
    slot.fill_(7.0)   # a loss call overwrites its whole slot: stale contents must not leak into the term
    cnt = a.numel()
    g = ops.mse_fwd_bwd(a.to(cuda, dtype), b.to(cuda, dtype), slot, 3.0 / cnt, 6.0 / cnt)
    assert abs(ops.slot_total(slot).item() - ref.item()) <= 1e-4 * abs(ref.item())
    assert rel_err(g, ga) <= TOL[dtype]

    img = torch.rand((2, 9, 7, 3), dtype=torch.float64, generator=torch.Generator().manual_seed(3))
    iq = img.float().double().requires_grad_()
    tv_ref = O.total_variation_loss(iq) * 5.0
    (gi,) = torch.autograd.grad(tv_ref, iq)
    gt = ops.tv_loss_fwd_bwd(img.to(cuda, torch.float32), slot, 5.0, dtype)
    assert abs(ops.slot_total(slot).item() - tv_ref.item()) <= 1e-4 * abs(tv_ref.item())
    assert rel_err(gt, gi) <= TOL[dtype]
    # known answer: a constant image has zero total variation
    ops.tv_loss_fwd_bwd(torch.full((1, 5, 5, 3), 0.25, device=cuda), slot, 1.0, dtype)
    assert ops.slot_total(slot).item() == 0.0
    # several terms in one launch: out[t] = sum of the slots first[t] .. first[t+1]-1
    slots = ops.loss_slots(4, cuda)
    vals = torch.rand((4, ops.LOSS_SLOT), generator=torch.Generator().manual_seed(5))
    slots.copy_(vals)
    out = torch.zeros(3, device=cuda)
    ops.loss_slots_sum(slots, [0, 1, 1, 4], out)
    want = [vals[0].double().sum().item(), 0.0, vals[1:4].double().sum().item()]
    for got, w_ in zip(out.tolist(), want):
        assert abs(got - w_) <= 1e-5 * max(1.0, abs(w_))


def test_adam_tf1_and_small_ops(cuda):
    from stmi import ops
    from oracle import loops as L
    rng = np.random.default_rng(0)
    p, g = rng.standard_normal(1000), rng.standard_normal(1000)
    m, v = np.zeros(1000), np.zeros(1000)
    pd = torch.tensor(p, device=cuda, dtype=torch.float32)
    md, vd = torch.zeros(1000, device=cuda), torch.zeros(1000, device=cuda)
    step = torch.zeros(1, dtype=torch.int32, device=cuda)
    for t in range(1, 4):
        # epsilon = 1e-1 (the Gatys setting) makes the TF1 placement of epsilon visible
        p, m, v = L.adam_tf1(p, g * t, m, v, t, 0.02, 0.99, 0.999, 1e-1)
        ops.adam_tf1(pd, torch.tensor(g * t, device=cuda, dtype=torch.float32), md, vd, 0.02, 0.99, 0.999, 1e-1,
                     step_dev=step)
    assert step.item() == 3
    assert rel_err(pd, torch.tensor(p)) <= 1e-5

    x = torch.rand((2, 3, 4, 3), device=cuda)
    mean = torch.tensor([1.0, 2.0, 3.0], device=cuda)
    y = ops.scale_shift(x, mean, 127.5, torch.float32)
    assert rel_err(y, O.vgg_preprocess(x.double().cpu(), mean.double().cpu())) <= 1e-6
    yb = ops.cast(y, torch.bfloat16)
    assert torch.equal(yb.cpu(), y.cpu().to(torch.bfloat16))
    assert torch.equal(ops.cast(yb, torch.float32).cpu(), yb.cpu().float())
    z = torch.tensor([-1.0, 0.5, 2.0], device=cuda)
    assert ops.clip_(z, 0.0, 1.0).cpu().tolist() == [0.0, 0.5, 1.0]


def test_adam_and_step_begin_on_odd_and_unaligned_buffers(cuda):
    """The 16-byte forms of Adam and of the one-launch clear: lengths that are not multiples of four, views that
    start off a 16-byte boundary, buffers of very different sizes, the step counter."""
    from stmi import ops
    from oracle import loops as L
    rng = np.random.default_rng(1)
    for n, off in ((1003, 0), (1003, 1), (5, 0), (3, 2), (70001, 0)):
        p, g = rng.standard_normal(n), rng.standard_normal(n)
        hold = [torch.zeros(n + 4, device=cuda) for _ in range(4)]
        pd, gd, md, vd = (h[off:off + n] for h in hold)
        pd.copy_(torch.tensor(p, dtype=torch.float32))
        gd.copy_(torch.tensor(g, dtype=torch.float32))
        step = torch.zeros(1, dtype=torch.int32, device=cuda)
        ops.adam_tf1(pd, gd, md, vd, 0.02, 0.9, 0.999, 1e-8, step_dev=step)
        want, _, _ = L.adam_tf1(p, g, np.zeros(n), np.zeros(n), 1, 0.02, 0.9, 0.999, 1e-8)
        assert step.item() == 1
        assert rel_err(pd, torch.tensor(want)) <= 1e-5, (n, off)
        for h, view in zip(hold, (pd, gd, md, vd)):      # nothing outside the view was touched
            assert h[:off].abs().sum().item() == 0 and h[off + n:].abs().sum().item() == 0
    big = torch.ones(100003, device=cuda)
    bufs = [big[1:], torch.ones(7, device=cuda), torch.ones(4096, device=cuda)[3:4001], torch.ones(1, device=cuda),
            torch.ones(262144, device=cuda)]
    step = torch.full((1,), 41, dtype=torch.int32, device=cuda)
    ops.ZeroList(bufs, step_dev=step).run()
    assert step.item() == 42
    for b in bufs:
        assert b.abs().sum().item() == 0
    assert big[0].item() == 1.0


def test_tile_statistics_entries_say_what_they_cannot_do(cuda):
    """The tile-statistics forms of the convolutions: the query returns 0 for a geometry whose kernel leaves none,
    the launch then fails with the library's message; statistics of a two-rows-per-tile geometry come from the
    plain convolution only; an empty batch is a no-op."""
    from stmi import ops, lib
    st = torch.cuda.current_stream().cuda_stream
    # a transposed convolution the up-sampling kernel is not built for (48 output channels)
    assert ops.conv_transpose_tile_stats_tiles((1, 8, 8, 64), (3, 3, 48, 64), 2, torch.bfloat16) == 0
    x = torch.zeros((1, 8, 8, 64), device=cuda, dtype=torch.bfloat16)
    wp = torch.zeros(ops.packed_elems(3, 3, 48, 64, ops.PACK_DGRAD), device=cuda, dtype=torch.bfloat16)
    y = torch.empty((1, 16, 16, 48), device=cuda, dtype=torch.bfloat16)
    ts = torch.zeros(64, device=cuda)
    with pytest.raises(lib.StmiError, match='no tile statistics'):
        lib.call('stmi_conv2d_transpose_nhwc_fwd_tile_stats', x.data_ptr(), wp.data_ptr(), y.data_ptr(), ts.data_ptr(),
                 1, 8, 8, 64, 48, 3, 3, 2, lib.BF16, st)
    # the 3-channel kernel leaves statistics for the (9, 32) layer only
    assert ops.conv_rgb_tile_stats_tiles(16, 16, 64, 3) == 0 and ops.conv_rgb_tile_stats_tiles(16, 40, 32, 9) == 4
    # a persistent-statistics geometry in bf16 (two rows per tile): ReLU with the statistics is refused
    shape = (1, 270, 480, 128)
    n_ts = ops.conv_tile_stats_tiles(shape, (3, 3, 128, 128), 1, torch.bfloat16)
    assert n_ts == 2 * 68 * 15 and ops.conv_tile_stats_free(shape, (3, 3, 128, 128), 1, torch.bfloat16)
    xb = torch.zeros(shape, device=cuda, dtype=torch.bfloat16)
    wq = torch.zeros(ops.packed_elems(3, 3, 128, 128, ops.PACK_FWD), device=cuda, dtype=torch.bfloat16)
    yb = torch.empty(shape, device=cuda, dtype=torch.bfloat16)
    rows = torch.zeros((1, n_ts, 4, 128), device=cuda)
    with pytest.raises(lib.StmiError, match='plain convolution'):
        lib.call('stmi_conv2d_nhwc_fwd_tile_stats', xb.data_ptr(), wq.data_ptr(), None, yb.data_ptr(), None,
                 rows.data_ptr(), 1, 270, 480, 128, 128, 3, 3, 1, 1, lib.BF16, st)
    # empty batches
    for entry, args in (('stmi_conv2d_nhwc_fwd_tile_stats', (None, None, None, None, None, None, 0, 8, 8, 128, 128, 3, 3, 1, 0)),
                        ('stmi_conv2d_transpose_nhwc_fwd_tile_stats', (None, None, None, None, 0, 8, 8, 64, 32, 3, 3, 2)),
                        ('stmi_conv2d_rgb_nhwc_fwd_tile_stats', (None, 1.0, None, None, None, None, 0, 8, 8, 32, 9))):
        lib.call(entry, *args, lib.BF16, st)


def test_error_reporting(cuda):
    from stmi import lib, ops
    x = torch.zeros((1, 4, 4, 3), device=cuda)
    wp = torch.zeros(ops.packed_elems(3, 3, 3, 8, ops.PACK_FWD), device=cuda)
    with pytest.raises(lib.StmiError, match='stride 1'):
        ops.conv2d_fwd(x, wp, (3, 3, 3, 8), 2)
    with pytest.raises(lib.StmiError, match='no CPU fallback'):
        ops.maxpool_fwd(torch.zeros((1, 2, 2, 8)))


# ---------------------------------------------------------------- row patches
def _pixels_to_rows_ref(x, kw, dx0):
    """rows[n,y,x,e] = x[n,y,x+dx0+e//c,e%c] for e < kw*c and an in-range column, else 0 (include/stmi.h)"""
    n, h, w, c = x.shape
    rows = torch.zeros((n, h, w, 32), dtype=x.dtype)
    for kx in range(kw):
        for xx in range(w):
            col = xx + dx0 + kx
            if 0 <= col < w:
                rows[:, :, xx, kx * c:(kx + 1) * c] = x[:, :, col, :]
    return rows


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('case', [(2, 5, 11, 3, 9, -4), (1, 4, 7, 3, 3, -1), (1, 3, 6, 3, 9, -4 + 0)], ids=str)
def test_pixels_to_rows_and_back(cuda, dtype, case):
    from stmi import ops
    n, h, w, c, kw, dx0 = case
    x = q(rnd((n, h, w, c), 11), dtype).to(dtype)
    ref = _pixels_to_rows_ref(x, kw, dx0)
    got = ops.pixels_to_rows(x.to(cuda), kw, dx0)
    assert torch.equal(got.cpu(), ref)          # a copy: exact in both dtypes
    # rows_to_pixels_sum: y[n,yy,x,co] = sum_kx rows[n,yy,x+dx0+kx,kx*c+co] over in-range columns (+ addend)
    rows = q(rnd((n, h, w, 32), 12), dtype)
    add = q(rnd((n, h, w, c), 13), dtype)
    want = torch.zeros((n, h, w, c), dtype=torch.float64)
    for kx in range(kw):
        for xx in range(w):
            col = xx + dx0 + kx
            if 0 <= col < w:
                want[:, :, xx, :] += rows[:, :, col, kx * c:(kx + 1) * c]
    y = ops.rows_to_pixels_sum(rows.to(cuda, dtype), c, kw, dx0)
    assert rel_err(y, want) <= TOL[dtype]
    y2 = ops.rows_to_pixels_sum(rows.to(cuda, dtype), c, kw, dx0, addend=add.to(cuda, dtype))
    assert rel_err(y2, want + add) <= TOL[dtype]


@pytest.mark.parametrize('dtype', DTYPES, ids=['f32', 'bf16'])
@pytest.mark.parametrize('k', [9, 3])
def test_row_patch_convolutions_match_the_plain_ones(cuda, dtype, k):
    """The three uses of row patches (include/stmi.h): conv c->co, conv ci->c forward and its
    backward-data, and both weight gradients, against the oracle's ordinary convolution."""
    from stmi import ops
    n, h, w, c, cw = 2, 10, 13, 3, 32
    pad = (k - 1) // 2
    # c -> cw (rows on the input side)
    x = rnd((n, h, w, c), 21)
    wt = rnd((k, k, c, cw), 22, 1.0 / np.sqrt(k * k * c))
    xq, wq = q(x, dtype).requires_grad_(), q(wt, dtype).requires_grad_()
    y_ref = O.conv2d_same(xq, wq, 1)
    dy = rnd(tuple(y_ref.shape), 23)
    (gw_ref,) = torch.autograd.grad(y_ref, (wq,), q(dy, dtype))
    rows = ops.pixels_to_rows(x.to(cuda, dtype), k, -pad)
    wp = ops.pack_conv_weights(wt.to(cuda, torch.float32), ops.PACK_FWD, dtype)
    y = ops.conv2d_fwd(rows, wp, (k, 1, 32, cw), 1)
    assert rel_err(y, y_ref) <= TOL[dtype]
    d = ops.conv2d_wgrad(rows, dy.to(cuda, dtype), (k, 1, 32, cw), 1)
    gw = ops.rows_wgrad_to_hwio(d, (k, k, c, cw), 0)
    assert rel_err(gw, gw_ref) <= TOL[dtype]
    # cw -> c (rows on the output side)
    x2 = rnd((n, h, w, cw), 24)
    wt2 = rnd((k, k, cw, c), 25, 1.0 / np.sqrt(k * k * cw))
    x2q, w2q = q(x2, dtype).requires_grad_(), q(wt2, dtype).requires_grad_()
    y2_ref = O.conv2d_same(x2q, w2q, 1)
    dy2 = rnd(tuple(y2_ref.shape), 26)
    gx2_ref, gw2_ref = torch.autograd.grad(y2_ref, (x2q, w2q), q(dy2, dtype))
    w2f = ops.pack_conv_weights(wt2.to(cuda, torch.float32), ops.PACK_FWD_ROWOUT, dtype)
    w2b = ops.pack_conv_weights(wt2.to(cuda, torch.float32), ops.PACK_DGRAD, dtype)
    t = ops.conv2d_fwd(x2.to(cuda, dtype), w2f, (k, 1, cw, 32), 1)
    y2 = ops.rows_to_pixels_sum(t, c, k, -pad)
    # the 32-wide intermediate is rounded to the compute dtype before the row sum
    assert rel_err(y2, y2_ref) <= 2 * TOL[dtype]
    r = ops.pixels_to_rows(dy2.to(cuda, dtype), k, pad - (k - 1))
    dx2 = ops.conv2d_dgrad(r, w2b, (k, 1, cw, 32), 1, (h, w))
    assert rel_err(dx2, gx2_ref) <= TOL[dtype]
    d2 = ops.conv2d_wgrad(x2.to(cuda, dtype), r, (k, 1, cw, 32), 1)
    gw2 = ops.rows_wgrad_to_hwio(d2, (k, k, cw, c), 1)
    assert rel_err(gw2, gw2_ref) <= TOL[dtype]


def test_pack_multi_matches_single_packs_and_kernel_name(cuda):
    from stmi import ops
    shapes = [((3, 3, 16, 32), ops.PACK_FWD), ((3, 3, 16, 32), ops.PACK_DGRAD), ((9, 9, 3, 32), ops.PACK_FWD),
              ((9, 9, 32, 3), ops.PACK_FWD_ROWOUT), ((9, 9, 32, 3), ops.PACK_DGRAD)]
    tensors = [rnd(s, 40 + i).float() for i, (s, _) in enumerate(shapes)]
    flat = torch.cat([t.reshape(-1) for t in tensors]).to(cuda)
    singles = [ops.pack_conv_weights(t.to(cuda), m, torch.bfloat16) for t, (_, m) in zip(tensors, shapes)]
    table, src, dst = [], 0, 0
    for t, (s, m), one in zip(tensors, shapes, singles):
        table.append((src, dst, s, m))
        src += t.numel()
        dst += (one.numel() + 63) // 64 * 64
    packed = torch.zeros(dst, dtype=torch.bfloat16, device=cuda)
    ops.pack_conv_weights_multi(flat, packed, ops.make_pack_table(table))
    for (_, d, _, _), one in zip(table, singles):
        assert torch.equal(packed[d:d + one.numel()], one)
    # the library reports the kernel it launched last (used by the bench's roofline leg)
    x = torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device=cuda)
    wp = ops.pack_conv_weights(torch.zeros((3, 3, 64, 64), device=cuda), ops.PACK_FWD, torch.bfloat16)
    ops.conv2d_fwd(x, wp, (3, 3, 64, 64), 1)
    assert 'conv_patch_kernel<__bf16' in ops.last_conv_kernel()
    ops.conv2d_fwd(x, wp, (3, 3, 64, 64), 2)
    assert 'conv_dn2_kernel<__bf16' in ops.last_conv_kernel()
    ops.conv2d_fwd(x[:, :7], wp, (3, 3, 64, 64), 2)     # odd side: pads above as well, the gather form
    assert 'conv_gemm_pc_kernel<__bf16' in ops.last_conv_kernel()
