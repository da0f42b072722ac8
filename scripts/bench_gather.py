"""Timing of the gather-GEMM layers of the training step (stride-2 / transposed convolutions,
row-patch layers, Gram backward): producer/consumer kernel vs the same-wave kernels (graph replay)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16; B = 4

def timed(fn, n=20):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3

def rnd(*shape):
    return torch.randn(shape, device=dev).to(dt)

cases = []
def conv_case(name, h, w, ci, co, k, s):
    kh, kw = k if isinstance(k, tuple) else (k, k)
    x = rnd(B, h, w, ci); wt = torch.randn((kh, kw, ci, co), device=dev) * 0.05
    wf = ops.pack_conv_weights(wt, ops.PACK_FWD, dt); wd = ops.pack_conv_weights(wt, ops.PACK_DGRAD, dt)
    y = ops.conv2d_fwd(x, wf, (kh, kw, ci, co), s)
    dx = torch.empty_like(x)
    cases.append((name + ' fwd', lambda: ops.conv2d_fwd(x, wf, (kh, kw, ci, co), s, out=y), y))
    cases.append((name + ' dgrad', lambda: ops.conv2d_dgrad(y, wd, (kh, kw, ci, co), s, (h, w), out=dx), dx))
def deconv_case(name, h, w, ci, co):
    x = rnd(B, h, w, ci); wt = torch.randn((3, 3, co, ci), device=dev) * 0.05
    wf = ops.pack_conv_weights(wt, ops.PACK_DGRAD, dt); wd = ops.pack_conv_weights(wt, ops.PACK_FWD, dt)
    y = ops.conv2d_transpose_fwd(x, wf, (3, 3, co, ci), 2)
    dx = torch.empty_like(x)
    cases.append((name + ' fwd', lambda: ops.conv2d_transpose_fwd(x, wf, (3, 3, co, ci), 2, out=y), y))
    cases.append((name + ' dgrad', lambda: ops.conv2d_transpose_dgrad(y, wd, (3, 3, co, ci), 2, out=dx), dx))
conv_case('conv2 3x3s2 32->64 @256', 256, 256, 32, 64, 3, 2)
conv_case('conv3 3x3s2 64->128 @128', 128, 128, 64, 128, 3, 2)
deconv_case('deconv1 128->64 @64', 64, 64, 128, 64)
deconv_case('deconv2 64->32 @128', 128, 128, 64, 32)
conv_case('rows 9x1 32->32 @256', 256, 256, 32, 32, (9, 1), 1)
conv_case('rows 3x1 32->64 @256', 256, 256, 32, 64, (3, 1), 1)
for hh, c in ((256, 64), (128, 128), (64, 256), (32, 512)):
    f = rnd(B, hh, hh, c); sp = rnd(ops.gram_packed_elems(B, c)); g = torch.zeros_like(f)
    cases.append(('gram_bwd %dx%dx%d' % (hh, hh, c), lambda f=f, sp=sp, g=g: ops.gram_bwd(f, sp, g, accumulate=True), None))

settings = [('same-wave', {'STMI_NO_PC': '1'}), ('pc', {})]
print('%-34s' % 'case' + ' '.join('%10s' % n for n, _ in settings) + '  (us)')
for name, fn, out in cases:
    row, ref = [], None
    for nm, env in settings:
        os.environ.pop('STMI_NO_PC', None); os.environ.update(env)
        row.append(timed(fn))
        if out is not None:
            fn(); torch.cuda.synchronize()
            if ref is None: ref = out.float().clone()
            else:
                err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
                assert err < 2e-2, (name, nm, err)
    print('%-34s' % name + ' '.join('%10.1f' % v for v in row))
