"""bf16 mode, gradient fidelity by loss term and by network half, full-width VGG, 2 x 64 x 48:
cosine of the flat stylizer gradient with the fp64 oracle's for
  * the bf16 mode as shipped (stylizer and VGG in bf16), all terms and each term alone;
  * a hybrid with the VGG pass in fp32 and the stylizer in bf16 (what keeping the loss network exact would buy);
  * a hybrid with the stylizer in fp32 and the VGG pass in bf16."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from oracle import tf_ops as O
from stmi import ops
from stmi.stylizer import StylizerNet
from stmi.vggnet import VggFeatures

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
n, h, w = 2, 64, 48
batch = rng.random((n, h, w, 3)).astype(np.float32)
style = rng.random((1, 46, 38, 3)).astype(np.float32)
params = O.init_stylizer_params(seed=1)
mean = np.array([103.9, 116.8, 123.7], np.float32)
vggw = O.synthetic_vgg_weights(seed=7, width_div=1, bias_scale=0.1)
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
W = {k: (t64(a), t64(b)) for k, (a, b) in vggw.items()}
with torch.no_grad():
    targets64, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)


def oracle_grad(weights):
    P = [t64(p).requires_grad_() for p in params]
    loss, *_ = O.train_step_loss(t64(batch), P, W, targets64, t64(mean), *weights)
    return torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, P)])


def gpu_grad(weights, dt_net, dt_vgg):
    net = StylizerNet(n, h, w, dev, dt_net, params=params, training=True)
    vgg = VggFeatures(vggw, mean, dev, dt_vgg)
    rep_c = vgg.replica(n, h, w)
    rep = vgg.replica(n, h, w, keep_for_backward=True)
    x = torch.tensor(batch, device=dev)
    style_rep = vgg.replica(1, 46, 38)
    tg, _ = style_rep.forward(torch.tensor(style, device=dev))
    tg = [g.clone() for g in tg]
    rep_c.forward(x, want_grams=False)
    out = net.forward(x)
    slots = ops.loss_slots(rep.LOSS_SLOTS, dev)
    d = rep.loss_and_grad(out, tg, rep_c.content, *weights, slots)
    if dt_vgg != dt_net:
        d = ops.cast(d, torch.float32) if dt_vgg != torch.float32 else d
        d = ops.cast(d, dt_net) if dt_net != torch.float32 else d
    net.backward(d)
    torch.cuda.synchronize()
    return net.flat_grad.double().cpu()


cos = lambda a, b: (a @ b / (a.norm() * b.norm())).item()
SW, CW, TVW = 1e-2, 1e4, 1e8
bf, f32 = torch.bfloat16, torch.float32
for term, wts in (('all', (SW, CW, TVW)), ('style', (SW, 0.0, 0.0)), ('content', (0.0, CW, 0.0)), ('tv', (0.0, 0.0, TVW))):
    ref = oracle_grad(wts)
    print('%-8s |g| %.3e   bf16/bf16 %.4f   stylizer bf16 + VGG fp32 %.4f   stylizer fp32 + VGG bf16 %.4f   fp32/fp32 %.6f' % (
        term, ref.norm().item(), cos(gpu_grad(wts, bf, bf), ref), cos(gpu_grad(wts, bf, f32), ref),
        cos(gpu_grad(wts, f32, bf), ref), cos(gpu_grad(wts, f32, f32), ref)))
