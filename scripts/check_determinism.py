"""Runs the first fp32 training step of the small parity problem several times from the same state and reports
the first tensor that differs between runs (statistics, activations, gradients of every stylizer block).

  python scripts/check_determinism.py [runs]
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from oracle import tf_ops as O
from stmi.trainer import StyleTrainer

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
batch = rng.random((2, 32, 24, 3)).astype(np.float32)
style = rng.random((1, 30, 22, 3)).astype(np.float32)
params = O.init_stylizer_params(seed=1)
vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
mean = np.array([103.9, 116.8, 123.7], np.float32)
snaps = []
for r in range(runs):
    tr = StyleTrainer(2, 32, 24, dev, vggw, mean, 1e-2, 1e4, 1e8, dtype=torch.float32, params=params, use_graph=False)
    tr.set_style_image(torch.tensor(style))
    tr.step(torch.tensor(batch, device=dev))
    torch.cuda.synchronize()
    snap = {}
    for i, b in enumerate(tr.net.blocks):
        snap['%02d conv_out' % i] = b.conv_out.clone()
        if b.tile_stats is not None:
            snap['%02d tile_stats' % i] = b.tile_stats.clone()
        snap['%02d stats' % i] = b.stats.clone()
        snap['%02d y' % i] = b.y.clone()
    for i, g in enumerate(tr.net.grad_views):
        snap['grad %02d' % i] = g.clone()
    snaps.append(snap)
    del tr
for r in range(1, runs):
    bad = [k for k in snaps[0] if not torch.equal(snaps[0][k], snaps[r][k])]
    if bad:
        k = bad[0]
        d = (snaps[0][k].double() - snaps[r][k].double()).abs()
        print('run %d: %d tensors differ; first: %s  max|d| %.3e  (max|x| %.3e)  at %s' % (
            r, len(bad), k, d.max().item(), snaps[0][k].abs().max().item(), np.unravel_index(int(d.argmax()), d.shape)))
        print('   all:', bad[:12])
    else:
        print('run %d: identical' % r)
