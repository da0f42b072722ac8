"""Timing of the pixel-reduction GEMM (weight gradients, Gram matrices) on the training-step
shapes: register-staged kernel vs the LDS-DMA producer/consumer kernel, replayed from a graph."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0')
dtype = torch.bfloat16
B = 4

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

WG = [('res 3x3 128->128 @64', 64, 64, 128, 128, 3, 1), ('conv3 3x3s2 64->128 @128', 128, 128, 64, 128, 3, 2),
      ('conv2 3x3s2 32->64 @256', 256, 256, 32, 64, 3, 2), ('rows 9x1 32->32 @256', 256, 256, 32, 32, (9, 1), 1)]
GR = [(256, 256, 64), (128, 128, 128), (64, 64, 256), (32, 32, 512), (16, 16, 512)]
settings = [('old', {'STMI_NO_WG_GLDS': '1'}), ('new', {}), ('new/b128', {'STMI_WG_BLOCKS': '128'}), ('new/b512', {'STMI_WG_BLOCKS': '512'})]
print('%-28s' % 'case' + ' '.join('%10s' % n for n, _ in settings) + '  (us; zero-fill kernel included)')
def sweep(name, fn, out):
    row, ref = [], None
    for nm, env in settings:
        for k in ('STMI_NO_WG_GLDS', 'STMI_WG_BLOCKS'): os.environ.pop(k, None)
        os.environ.update(env)
        row.append(timed(fn))
        fn(); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        else:
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-3, (name, nm, err)
    print('%-28s' % name + ' '.join('%10.1f' % v for v in row))
for name, h, w, ci, co, k, s in WG:
    x = torch.randn((B, h, w, ci), device=dev).to(dtype)
    dy = torch.randn((B, -(-h // s), -(-w // s), co), device=dev).to(dtype)
    kh, kw = k if isinstance(k, tuple) else (k, k)
    out = torch.empty((kh, kw, ci, co), device=dev)
    sweep(name, lambda: ops.conv2d_wgrad(x, dy, (kh, kw, ci, co), s, out=out), out)
for h, w, c in GR:
    f = torch.randn((B, h, w, c), device=dev).to(dtype)
    out = torch.empty((B, c, c), device=dev)
    sweep('gram %dx%dx%d' % (h, w, c), lambda: ops.gram_fwd(f, out=out), out)
