"""Per-block timeline of the patch conv kernel (100 MHz wall clock ticks -> us)."""
import os, sys, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops, lib
h = lib.load()
dev = torch.device('cuda:0'); dtype = torch.bfloat16; B = 4
for name, hh, w, ci, co, pco in [('vgg1_2', 256, 256, 64, 64, '64'), ('vgg3_x', 64, 64, 256, 256, '128'), ('res', 64, 64, 128, 128, '64')]:
    x = torch.randn((B, hh, w, ci), device=dev).to(dtype)
    wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
    wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dtype)
    y = torch.empty((B, hh, w, co), device=dev, dtype=dtype)
    os.environ['STMI_PATCH_CO'] = pco
    os.environ['STMI_PATCH_DBG'] = '4'
    for _ in range(3):
        ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, out=y)
    torch.cuda.synchronize()
    n = 4096 * 16
    buf = (ctypes.c_long * n)()
    h.stmi_debug_patch_times(buf, n)
    full = np.frombuffer(buf, dtype=np.int64).reshape(4096, 2, 8).astype(np.float64) / 100.0
    nb = min(4096, B * (hh // (128 // min(64, w))) * (w // min(64, w)) * (co // int(pco)))
    full = full[:nb]
    t0 = full[:, :, 0].min()
    pf = full[:, 1, :]
    print('   producer setup: decode+offsets-w %.2f  W0-issue %.2f  poff %.2f  pieces+W12 %.2f' % ((pf[:, 5] - pf[:, 0]).mean(), (pf[:, 6] - pf[:, 5]).mean(), (pf[:, 7] - pf[:, 6]).mean(), (pf[:, 1] - pf[:, 7]).mean()))
    for role, nm in ((0, 'consumer'), (1, 'producer')):
        t = full[:, role, :5]
        d_ = np.diff(t, axis=1)
        print('%s %s blocks=%d span %.1f us | setup %.2f  to-first-barrier %.2f  loop %.2f  epilogue %.2f'
              % (name, nm, nb, t[:, 4].max() - t0, d_[:, 0].mean(), d_[:, 1].mean(), d_[:, 2].mean(), d_[:, 3].mean()))
