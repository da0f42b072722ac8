import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'tests')]
from test_gpu_step import small_setup
from stmi import ops
from stmi.trainer import StyleTrainer
cuda = torch.device('cuda:0')
batch, style, params, vggw, mean = small_setup()
tr = StyleTrainer(batch.shape[0], batch.shape[1], batch.shape[2], cuda, vggw, mean, 1e-2, 1e4, 1e8,
                  dtype=torch.float32, params=params, use_graph=True)
tr.set_style_image(torch.tensor(style))
xb = torch.tensor(batch, device=cuda)
for t in range(3):
    tr.step(xb)
    torch.cuda.synchronize()
    rep = tr.rep_loss
    for li, nm in enumerate(rep.style_names[:3]):
        pre = rep.pre[nm]
        g2 = ops.gram_fwd(pre)
        torch.cuda.synchronize()
        print(t, nm, 'pre max', float(pre.abs().max()), 'finite', bool(torch.isfinite(pre).all()),
              'G replay max', float(rep.grams[li].abs().max()), 'G eager max', float(g2.abs().max()),
              'stylized max', float(tr.net.out_f32.abs().max()), 'xpre max', float(rep.x_pre.abs().max()))
