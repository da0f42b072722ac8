"""3-channel first layers: row-patch path (cast / preprocess + pixels_to_rows + gather-GEMM) vs conv_rgb."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16

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
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3

for (n, h, w, k, co, reps) in [(4, 256, 256, 9, 32, 20), (4, 256, 256, 3, 64, 20), (16, 1080, 1920, 9, 32, 2)]:
    x = torch.rand((n, h, w, 3), device=dev)
    wt = torch.randn((k, k, 3, co), device=dev) * 0.05
    mean = torch.zeros(3, device=dev)
    xin = torch.empty((n, h, w, 3), device=dev, dtype=dt)
    rows = torch.empty((n, h, w, 32), device=dev, dtype=dt)
    y = torch.empty((n, h, w, co), device=dev, dtype=dt); pre = torch.empty_like(y)
    w_rows = ops.pack_conv_weights(torch.zeros((k, 1, 32, co), device=dev), ops.PACK_FWD, dt)
    w_rgb = ops.pack_conv_weights(wt, ops.PACK_FWD_RGB, dt)
    vgg = k == 3
    def old():
        if vgg: ops.scale_shift(x, mean, 127.5, dt, out=xin)
        else: ops.cast(x, dt, out=xin)
        ops.pixels_to_rows(xin, k, -((k - 1) // 2), out=rows)
        if vgg: ops.conv2d_fwd(rows, w_rows, (k, 1, 32, co), 1, relu=True, want_pre=True, out=y, out_pre=pre)
        else: ops.conv2d_fwd(rows, w_rows, (k, 1, 32, co), 1, out=y)
    def new():
        if vgg: ops.conv2d_rgb_fwd(x, w_rgb, k, co, dt, scale=127.5, mean=mean, relu=True, want_pre=True, out=y, out_pre=pre)
        else: ops.conv2d_rgb_fwd(x, w_rgb, k, co, dt, out=y)
    print((n, h, w, k, co), 'row-patch path %.1f us   conv_rgb %.1f us' % (timed(old, reps), timed(new, reps)))
