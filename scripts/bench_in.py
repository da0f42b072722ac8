"""Per-launch time of the instance-norm kernels on the residual-block shape (graph replay)."""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops, lib
dev = torch.device('cuda:0')

def timed(fn, n=50):
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

for shape in [(4, 64, 64, 128), (4, 128, 128, 64), (4, 256, 256, 32), (4, 256, 256, 3), (16, 270, 480, 128), (16, 1080, 1920, 32)]:
    n, h, w, c = shape
    x = torch.randn(shape, device=dev).bfloat16()
    dy = torch.randn(shape, device=dev).bfloat16()
    y = torch.empty_like(x); dx = torch.empty_like(x)
    sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev)
    dsc = torch.zeros(c, device=dev); dsh = torch.zeros(c, device=dev)
    ws = ops.instance_norm_ws(n, h, w, c, dev)
    stats = torch.empty((n, c, 2), device=dev)
    s = lambda: torch.cuda.current_stream().cuda_stream
    t_stats = timed(lambda: lib.call('stmi_instance_norm_nhwc_stats', x.data_ptr(), ws.data_ptr(), stats.data_ptr(), 1e-3, n, h, w, c, lib.BF16, s()))
    t_apply = timed(lambda: lib.call('stmi_instance_norm_nhwc_apply', x.data_ptr(), stats.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, y.data_ptr(), None, 1, n, h, w, c, lib.BF16, s()))
    t_fwd = timed(lambda: ops.instance_norm_fwd(x, sc, sh, 1, ws=ws, out=y, stats=stats))
    t_bwd = timed(lambda: ops.instance_norm_bwd(x, dy, stats, sc, sh, dsc, dsh, 1, ws=ws, out=dx))
    t_cast = timed(lambda: ops.cast(x, torch.float32))
    mb = x.numel() * 2 / 1e6
    print('%s (%.1f MB): stats+finalize %.1f us, apply %.1f us, fused fwd (2 launches) %.1f us, bwd (2 launches) %.1f us' % (shape, mb, t_stats, t_apply, t_fwd, t_bwd), flush=True)
