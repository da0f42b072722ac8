"""Cost of a kernel boundary inside a replayed HIP graph: chains of tiny and of mid-sized kernels."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0')

def timed(fn, n):
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

small = torch.zeros(64, device=dev, dtype=torch.bfloat16)
out_s = torch.empty(64, device=dev, dtype=torch.float32)
print('tiny cast kernel (64 elements): %.2f us per launch in a 200-kernel chain' % timed(lambda: ops.cast(small, torch.float32, out=out_s), 200))
for mb in (1, 4, 16, 64):
    x = torch.zeros(mb * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)
    y = torch.empty(x.shape, device=dev, dtype=torch.float32)
    t = timed(lambda: ops.cast(x, torch.float32, out=y), 100)
    print('cast %3d MB bf16 -> f32: %.2f us per launch (%.2f TB/s)' % (mb, t, mb * 3 * 1.048576 / t))
