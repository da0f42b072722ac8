"""Splits the patch conv kernel's time: full, fills skipped, MFMA+LDS reads skipped, both."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops

dev = torch.device('cuda:0')
dtype = torch.bfloat16
B = int(os.environ.get('B', '4'))
LAYERS = [('vgg1_2', 256, 256, 64, 64, '64'), ('vgg2_2', 128, 128, 128, 128, '128'), ('vgg3_x', 64, 64, 256, 256, '128'),
          ('vgg4_x', 32, 32, 512, 512, '64'), ('res', 64, 64, 128, 128, '64')]

def run(x, wp, kshape, y, iters=20):
    """us per launch, launches replayed from a HIP graph so that host launch cost is not measured"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            ops.conv2d_fwd(x, wp, kshape, 1, out=y)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            ops.conv2d_fwd(x, wp, kshape, 1, out=y)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3

print('%-8s %6s | %9s %9s %9s %9s  (us; ideal = flops at 2.5 PF)' % ('layer', 'ideal', 'full', 'no-fill', 'no-mma', 'neither'))
for name, h, w, ci, co, pco in LAYERS:
    x = torch.randn((B, h, w, ci), device=dev).to(dtype)
    wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
    wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dtype)
    y = torch.empty((B, h, w, co), device=dev, dtype=dtype)
    flops = 2.0 * B * h * w * co * 9 * ci
    os.environ['STMI_PATCH_CO'] = pco
    row = []
    for d in '0123':
        os.environ['STMI_PATCH_DBG'] = d
        row.append(run(x, wp, (3, 3, ci, co), y))
    print('%-8s %6.1f | ' % (name, flops / 2.5e15 * 1e6) + ' '.join('%9.1f' % v for v in row))
