"""Timing of the pixel-reduction GEMM (weight gradients, Gram matrices) on the training-step shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0')
dtype = torch.bfloat16
B = 4

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

WG = [('res 3x3 128->128 @64', 64, 64, 128, 128, 3, 1), ('conv3 3x3s2 64->128 @128', 128, 128, 64, 128, 3, 2),
      ('conv2 3x3s2 32->64 @256', 256, 256, 32, 64, 3, 2), ('conv1 9x9 3->32 @256', 256, 256, 3, 32, 9, 1),
      ('final 9x9 32->3 @256', 256, 256, 32, 3, 9, 1)]
GR = [(256, 256, 64), (128, 128, 128), (64, 64, 256), (32, 32, 512), (16, 16, 512)]
settings = [(None, None), (None, '512'), (None, '256'), ('1', '1024'), ('1', '512'), ('1', '256'), ('0', '256')]
print('%-28s' % 'case' + ' '.join('%12s' % ('cfg%s/b%s' % (c or 'A', b or '1024')) for c, b in settings) + '  (us)')
for name, h, w, ci, co, k, s in WG:
    x = torch.randn((B, h, w, ci), device=dev).to(dtype)
    dy = torch.randn((B, -(-h // s), -(-w // s), co), device=dev).to(dtype)
    out = torch.empty((k, k, ci, co), device=dev)
    row = []
    for c, b in settings:
        for kk, v in (('STMI_WG_CFG', c), ('STMI_WG_BLOCKS', b)):
            os.environ.pop(kk, None)
            if v is not None: os.environ[kk] = v
        row.append(timeit(lambda: ops.conv2d_wgrad(x, dy, (k, k, ci, co), s, out=out)))
    print('%-28s' % name + ' '.join('%12.1f' % v for v in row))
for h, w, c in GR:
    f = torch.randn((B, h, w, c), device=dev).to(dtype)
    out = torch.empty((B, c, c), device=dev)
    row = []
    for cf, b in settings:
        for kk, v in (('STMI_WG_CFG', cf), ('STMI_WG_BLOCKS', b)):
            os.environ.pop(kk, None)
            if v is not None: os.environ[kk] = v
        row.append(timeit(lambda: ops.gram_fwd(f, out=out)))
    print('%-28s' % ('gram %dx%dx%d' % (h, w, c)) + ' '.join('%12.1f' % v for v in row))
