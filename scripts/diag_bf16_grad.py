"""Diagnostic: per-variable cosine similarity of bf16-mode gradients vs the fp64 oracle."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'tests')]
from test_gpu_step import small_setup, oracle_step
from stmi.trainer import StyleTrainer
cuda = torch.device('cuda:0')
for (h, w) in [(64, 48), (128, 128)]:
    batch, style, params, vggw, mean = small_setup(h=h, w=w)
    sw, cw, tvw = 1e-2, 1e4, 1e8
    ref, _, _ = oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=1)
    for dt in (torch.float32, torch.bfloat16):
        tr = StyleTrainer(batch.shape[0], h, w, cuda, vggw, mean, sw, cw, tvw, dtype=dt, params=params, use_graph=False)
        tr.set_style_image(torch.tensor(style))
        tr.step(torch.tensor(batch, device=cuda))
        print(h, w, dt, 'losses', tr.read_losses(), 'ref', ref[0][0])
        cs = []
        for i, (g, r) in enumerate(zip(tr.net.grad_views, ref[0][2])):
            g = g.double().cpu().reshape(-1); r = r.reshape(-1)
            cs.append((g @ r / (g.norm() * r.norm() + 1e-300)).item())
        print('  cos W    ', ' '.join('%.3f' % c for c in cs[0::3]))
        print('  cos shift', ' '.join('%.3f' % c for c in cs[1::3]))
        print('  cos scale', ' '.join('%.3f' % c for c in cs[2::3]))
        g = tr.net.flat_grad.double().cpu(); r = torch.cat([x.reshape(-1) for x in ref[0][2]])
        print('  total cos', (g @ r / (g.norm() * r.norm())).item(), 'norm ratio', (g.norm() / r.norm()).item())
