"""Diagnostic: which bf16 roundings move the training gradient away from the fp64 oracle's.

fp32 kernels, with the OUTPUT of selected front-end ops rounded to bf16 in place; per loss term
(style only / content only / TV only / all) the cosine between the flat gradient and the oracle's.

  python scripts/diag_bf16_grad3.py [h w]
"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'tests')]
from test_gpu_step import small_setup, oracle_step
from stmi import ops
from stmi.trainer import StyleTrainer
cuda = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 48)

ORIG = {}
ALL = ['conv2d_fwd', 'conv2d_to3', 'conv2d_rgb_fwd', 'conv2d_transpose_fwd', 'instance_norm_fwd', 'maxpool_fwd',
       'scale_shift']


def install(names, only3=None):
    for nm in ALL:
        if nm not in ORIG:
            ORIG[nm] = getattr(ops, nm)
        setattr(ops, nm, ORIG[nm])
    for nm in names:
        fn = ORIG[nm]

        def wrapped(*a, fn=fn, **k):
            r = fn(*a, **k)
            for o in (r if isinstance(r, tuple) else (r,)):
                if torch.is_tensor(o) and o.dtype == torch.float32 and o.dim() == 4:
                    if only3 is True and o.shape[-1] != 3:
                        continue
                    if only3 is False and o.shape[-1] == 3:
                        continue
                    o.copy_(o.bfloat16().float())
            return r
        setattr(ops, nm, wrapped)


batch, style, params, vggw, mean = small_setup(h=h, w=w)
if os.environ.get('DIAG_SMOOTH'):
    # structured pictures instead of white noise: the content difference F - T is then not a small
    # difference of two near-identical noise responses
    from test_gpu_e2e import smooth_image
    rng = np.random.default_rng(5)
    batch = np.concatenate([smooth_image(rng, h, w) for _ in range(batch.shape[0])]).astype(np.float32)
    style = smooth_image(rng, style.shape[1], style.shape[2]).astype(np.float32)
TERMS = {'all': (1e-2, 1e4, 1e8), 'style': (1e-2, 0.0, 0.0), 'content': (0.0, 1e4, 0.0), 'tv': (0.0, 0.0, 1e8)}
REFS = {}
for term, (sw, cw, tvw) in TERMS.items():
    ref, _, _ = oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=1)
    REFS[term] = torch.cat([x.reshape(-1) for x in ref[0][2]])
print('gradient norms per term (oracle):', {k: float(v.norm()) for k, v in REFS.items()})

CASES = [('none', [], None), ('all', ALL, None)]
if os.environ.get('DIAG_FULL'):
    CASES += [(nm, [nm], None) for nm in ALL] + [('all but 3-channel tensors', ALL, False),
                                                 ('3-channel tensors only', ALL, True)]
# how small is the content difference next to the features themselves (fp64 oracle)?
from oracle import tf_ops as O
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
with torch.no_grad():
    Wd = {k: (t64(a), t64(b)) for k, (a, b) in vggw.items()}
    out = O.stylizer_forward(t64(batch), [t64(p) for p in params])
    _, cf = O.vgg_taps(O.vgg_preprocess(out, t64(mean)), Wd)
    _, ct = O.vgg_taps(O.vgg_preprocess(t64(batch), t64(mean)), Wd)
    print('content tap: |F - T| / |F| = %.4f' % ((cf[0] - ct[0]).norm() / cf[0].norm()).item())
for label, names, only3 in CASES:
    install(names, only3)
    row = []
    for term, (sw, cw, tvw) in TERMS.items():
        tr = StyleTrainer(batch.shape[0], h, w, cuda, vggw, mean, sw, cw, tvw, dtype=torch.float32, params=params,
                          use_graph=False)
        tr.set_style_image(torch.tensor(style))
        tr.step(torch.tensor(batch, device=cuda))
        g = tr.net.flat_grad.double().cpu()
        r = REFS[term]
        row.append('%s %.4f' % (term, (g @ r / (g.norm() * r.norm() + 1e-300)).item()))
    print('%-28s %s' % (label, '   '.join(row)), flush=True)
install([], None)
# the real bf16 mode for comparison
row = []
for term, (sw, cw, tvw) in TERMS.items():
    tr = StyleTrainer(batch.shape[0], h, w, cuda, vggw, mean, sw, cw, tvw, dtype=torch.bfloat16, params=params,
                      use_graph=False)
    tr.set_style_image(torch.tensor(style))
    tr.step(torch.tensor(batch, device=cuda))
    g = tr.net.flat_grad.double().cpu()
    r = REFS[term]
    row.append('%s %.4f' % (term, (g @ r / (g.norm() * r.norm() + 1e-300)).item()))
print('%-28s %s' % ('bf16 mode', '   '.join(row)))
