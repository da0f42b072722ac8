"""conv + instance norm forward with and without the conv's tile statistics (graph replay, us per pair)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'scripts')]
from stmi import ops
dev = torch.device('cuda:0')


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


for shape in [(4, 64, 64, 128), (1, 64, 64, 128), (16, 270, 480, 128), (4, 128, 128, 64)]:
    n, h, w, c = shape
    dt = torch.bfloat16
    x = torch.randn(shape, device=dev).to(dt)
    wgt = (torch.randn((3, 3, c, c), device=dev) * 0.05).contiguous()
    wp = ops.pack_conv_weights(wgt, ops.PACK_FWD, dt)
    y = torch.empty_like(x); z = torch.empty_like(x)
    sc = torch.ones(c, device=dev); sh = torch.zeros(c, device=dev)
    ws = ops.instance_norm_ws(n, h, w, c, dev)
    stats = torch.empty((n, c, 2), device=dev)
    ts = ops.conv_tile_stats_tiles(shape, (3, 3, c, c), 1, dt)
    tiles = torch.zeros((n, ts, 4, c), device=dev)
    t_conv = timed(lambda: ops.conv2d_fwd(x, wp, (3, 3, c, c), 1, out=y))
    t_conv_s = timed(lambda: ops.conv2d_fwd(x, wp, (3, 3, c, c), 1, out=y, tile_stats=tiles))
    t_in = timed(lambda: ops.instance_norm_fwd(y, sc, sh, ops.ACT_RELU, ws=ws, out=z, stats=stats))
    t_in_s = timed(lambda: ops.instance_norm_fwd(y, sc, sh, ops.ACT_RELU, out=z, stats=stats, tile_stats=tiles))
    print('%s tiles/sample %d: conv %.1f -> %.1f us, instance norm %.1f -> %.1f us, pair %.1f -> %.1f us'
          % (shape, ts, t_conv, t_conv_s, t_in, t_in_s, t_conv + t_in, t_conv_s + t_in_s), flush=True)
