"""conv2d_dgrad / conv2d_fwd at the layer shapes of a small-picture VGG pass against the fp64 oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from oracle import tf_ops as O
from stmi import ops
dev = torch.device('cuda:0')
rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max()).item()
g = torch.Generator().manual_seed(0)
for dtype in (torch.float32, torch.bfloat16):
    for (n, h, w, ci, co) in [(1, 4, 4, 512, 512), (1, 8, 8, 512, 512), (1, 8, 8, 256, 512), (1, 16, 16, 256, 256), (1, 16, 16, 128, 256),
                              (1, 32, 32, 128, 128), (1, 32, 32, 64, 128), (1, 64, 64, 64, 64), (2, 16, 12, 256, 256), (1, 16, 16, 256, 128)]:
        x = torch.randn((n, h, w, ci), generator=g, dtype=torch.float64)
        wt = torch.randn((3, 3, ci, co), generator=g, dtype=torch.float64) / np.sqrt(9 * ci)
        dy = torch.randn((n, h, w, co), generator=g, dtype=torch.float64)
        mask = torch.randn((n, h, w, ci), generator=g, dtype=torch.float64)
        qd = lambda t: t.to(dtype).double()
        xq = qd(x).requires_grad_()
        y = O.conv2d_same(xq, qd(wt), 1)
        (gx,) = torch.autograd.grad(y, xq, qd(dy))
        gx_m = gx * (qd(mask) > 0)
        wd = ops.pack_conv_weights(wt.to(dev, torch.float32), ops.PACK_DGRAD, dtype)
        wf = ops.pack_conv_weights(wt.to(dev, torch.float32), ops.PACK_FWD, dtype)
        yy = ops.conv2d_fwd(x.to(dev, dtype), wf, (3, 3, ci, co), 1)
        kf = ops.last_conv_kernel().split('stmi::')[-1].split('(')[0]
        d1 = ops.conv2d_dgrad(dy.to(dev, dtype), wd, (3, 3, ci, co), 1, (h, w))
        d2 = ops.conv2d_dgrad(dy.to(dev, dtype), wd, (3, 3, ci, co), 1, (h, w), relu_mask=mask.to(dev, dtype))
        kd = ops.last_conv_kernel().split('stmi::')[-1].split('(')[0]
        print('%s %s: fwd %.1e [%s]  dgrad %.1e  masked %.1e [%s]' % (str(dtype)[6:], (n, h, w, ci, co), rel(yy, y), kf, rel(d1, gx), rel(d2, gx_m), kd))
