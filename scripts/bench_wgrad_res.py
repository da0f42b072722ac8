"""Weight gradient of a residual layer (3x3, 128 -> 128, bf16) at training sizes: tap-per-block kernel against the
resident-patch kernel, wgrad + fold replayed from a graph (us per call)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
os.environ['STMI_ENV_LIVE'] = '1'
os.environ['STMI_WG_RES_MIN_PIX'] = '0'   # both kernels at every size
from stmi import ops
dev = torch.device('cuda:0')
dtype = torch.float32 if os.environ.get('F32') else torch.bfloat16


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
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


settings = [('taps', {'STMI_NO_WG_RES': '1'}), ('res', {}), ('res/b128', {'STMI_WG_BLOCKS': '128'}), ('res/b512', {'STMI_WG_BLOCKS': '512'})]
shapes = [(4, 64, 64, 128), (1, 64, 64, 128), (4, 128, 128, 128), (8, 64, 64, 128)]
if len(sys.argv) > 1:      # one shape (index) and one setting (name): for a profiler run
    shapes = [shapes[int(sys.argv[1])]]
    settings = [s for s in settings if s[0] == sys.argv[2]]
for (n, h, w, c) in shapes:
    x = torch.randn((n, h, w, c), device=dev).to(dtype)
    dy = torch.randn((n, h, w, c), device=dev).to(dtype)
    out = torch.empty((3, 3, c, c), device=dev)
    row, ref = [], None
    for nm, env in settings:
        for k in ('STMI_NO_WG_RES', 'STMI_WG_BLOCKS'):
            os.environ.pop(k, None)
        os.environ.update(env)
        need = ops.conv2d_wgrad_ws_floats(x.shape, (3, 3, c, c), 1, dtype)
        ws = torch.empty(max(need, 4), device=dev)
        row.append('%s %.1f (%s, %d planes)' % (nm, timed(lambda: ops.conv2d_wgrad(x, dy, (3, 3, c, c), 1, out=out, ws=ws)),
                                                ops.last_wgrad_kernel().replace('wgrad_', '').replace('_kernel', ''), need // out.numel()))
        if ref is None:
            ref = out.clone()
        else:
            assert ((out - ref).abs().max() / ref.abs().max()).item() < 2e-3
    print((n, h, w, c), ' | '.join(row), flush=True)
