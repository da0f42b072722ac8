"""Per-layer timing of the conv gather-GEMM kernels on the heavy layer shapes of the
training step (batch 4, 256x256).  STMI_CONV_CFG / STMI_NO_GLDS / STMI_NO_PATCH / STMI_PATCH_CO select the variant."""
import os, sys
os.environ.setdefault('STMI_ENV_LIVE', '1')   # this script switches kernel knobs inside one process
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops

dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
B = int(os.environ.get('B', '4'))
LAYERS = [  # name, h, w, ci, co
    ('vgg1_2', 256, 256, 64, 64), ('vgg2_1', 128, 128, 64, 128), ('vgg2_2', 128, 128, 128, 128),
    ('vgg3_1', 64, 64, 128, 256), ('vgg3_x', 64, 64, 256, 256), ('vgg4_1', 32, 32, 256, 512),
    ('vgg4_x', 32, 32, 512, 512), ('vgg5_x', 16, 16, 512, 512), ('res', 64, 64, 128, 128),
]
cfgs = [None, 'v1', '1', '3', 'p32', 'p64', 'p128']

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

print('%-8s %10s | ' % ('layer', 'GFLOP') + ' '.join('%9s' % (c or 'auto') for c in cfgs) + '   (TFLOP/s)')
for name, h, w, ci, co in LAYERS:
    x = torch.randn((B, h, w, ci), device=dev).to(dtype)
    wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
    wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dtype)
    y = torch.empty((B, h, w, co), device=dev, dtype=dtype)
    flops = 2.0 * B * h * w * co * 9 * ci
    row, ref = [], None
    for c in cfgs:
        for k in ('STMI_CONV_CFG', 'STMI_NO_GLDS', 'STMI_NO_PATCH', 'STMI_PATCH_CO'): os.environ.pop(k, None)
        if c == 'v1': os.environ['STMI_NO_GLDS'] = '1'; os.environ['STMI_NO_PATCH'] = '1'
        elif c in ('p32', 'p64', 'p128'): os.environ['STMI_PATCH_CO'] = c[1:]
        elif c is not None: os.environ['STMI_CONV_CFG'] = c; os.environ['STMI_NO_PATCH'] = '1'
        t = run(x, wp, (3, 3, ci, co), y)
        if c == 'v1': ref = y.float().clone()
        elif ref is not None:
            err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-2, (name, c, err)
        row.append(flops / (t * 1e-6) / 1e12)
    print('%-8s %10.2f | ' % (name, flops / 1e9) + ' '.join('%9.1f' % v for v in row))
