"""Diagnostic: fp32 kernels with activations rounded to bf16 (forward noise only) vs fp64 oracle."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'tests')]
from test_gpu_step import small_setup, oracle_step
from stmi import ops
from stmi.trainer import StyleTrainer
cuda = torch.device('cuda:0')
mode = sys.argv[1]  # 'fwd' -> round forward activations; 'bwd' -> round gradients;
# 'fwd_not3' -> forward activations except 3-channel tensors (the stylized image side); 'fwd_only3' -> only those

def rounder(fn, keys):
    def wrapped(*a, **k):
        r = fn(*a, **k)
        outs = r if isinstance(r, tuple) else (r,)
        for o in outs:
            if torch.is_tensor(o) and o.dtype == torch.float32 and o.dim() == 4:
                if (mode == 'fwd_not3' and o.shape[-1] == 3) or (mode == 'fwd_only3' and o.shape[-1] != 3):
                    continue
                o.copy_(o.bfloat16().float())
        return r
    return wrapped

fwd_names = ['conv2d_fwd', 'conv2d_to3', 'conv2d_rgb_fwd', 'conv2d_transpose_fwd', 'instance_norm_fwd', 'maxpool_fwd', 'scale_shift']
bwd_names = ['conv2d_dgrad', 'conv2d_transpose_dgrad', 'instance_norm_bwd', 'maxpool_bwd', 'mse_fwd_bwd', 'tv_loss_fwd_bwd']
for nm in (fwd_names if mode.startswith('fwd') else bwd_names):
    setattr(ops, nm, rounder(getattr(ops, nm), None))

h, w = 64, 48
batch, style, params, vggw, mean = small_setup(h=h, w=w)
sw, cw, tvw = 1e-2, 1e4, 1e8
ref, _, _ = oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=1)
tr = StyleTrainer(batch.shape[0], h, w, cuda, vggw, mean, sw, cw, tvw, dtype=torch.float32, params=params, use_graph=False)
tr.set_style_image(torch.tensor(style))
tr.step(torch.tensor(batch, device=cuda))
print(mode, 'losses', tr.read_losses(), 'ref', ref[0][0])
cs = []
for i, (g, r) in enumerate(zip(tr.net.grad_views, ref[0][2])):
    g = g.double().cpu().reshape(-1); r = r.reshape(-1)
    cs.append((g @ r / (g.norm() * r.norm() + 1e-300)).item())
gf = tr.net.flat_grad.double().cpu(); rf = torch.cat([x.reshape(-1) for x in ref[0][2]])
print('  flat cos', (gf @ rf / (gf.norm() * rf.norm())).item())
print('  cos W    ', ' '.join('%.3f' % c for c in cs[0::3]))
