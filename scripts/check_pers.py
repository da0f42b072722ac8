"""Persistent patch kernel against the one-tile-per-block form on shapes of several block rounds (bit-equal:
same products, same accumulation order), and its time per launch."""
import os, sys
os.environ['STMI_ENV_LIVE'] = '1'
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0')
CASES = [(2, 128, 256, 128, 128, torch.bfloat16), (2, 130, 250, 128, 64, torch.bfloat16), (1, 256, 256, 64, 64, torch.float32),
         (16, 270, 480, 128, 128, torch.bfloat16), (1, 512, 512, 128, 128, torch.bfloat16), (1, 128, 128, 256, 256, torch.bfloat16)]
if len(sys.argv) > 1:
    CASES = CASES[:int(sys.argv[1])]
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n, h, w, ci, co, dt in CASES:
    x = torch.randn((n, h, w, ci), device=dev).to(dt)
    wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
    b = torch.randn(co, device=dev)
    wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dt)
    res = {}
    for mode in ('0', '1'):
        os.environ['STMI_PATCH_PERS'] = mode
        y = ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, bias=b, relu=True)
        torch.cuda.synchronize()
        name = ops.last_conv_kernel()
        t = timed(lambda: ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, bias=b, relu=True, out=y))
        res[mode] = (y.float().clone(), name, t)
        print(' ', (n, h, w, ci, co), str(dt)[6:], 'PERS=' + mode, name[11:60], '%.1f us' % t, flush=True)
    assert 'pers' in res['1'][1], 'persistent kernel not selected'
    d = (res['0'][0] - res['1'][0]).abs().max().item()
    print('   max |difference| =', d, flush=True)
    assert d == 0.0
print('ok')
