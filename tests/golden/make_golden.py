"""Generates the committed golden fixtures from the fp64 oracle (oracle/tf_ops.py).

PARITY UNPINNED: the reference holds no golden vectors and TensorFlow cannot run here, so
these vectors pin the ORACLE against regressions (tests/test_cpu_oracle.py) and the HIP path
against the oracle's committed outputs (tests/test_gpu_golden.py); they were not produced by
TensorFlow.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import tf_ops as O  # noqa: E402

t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def stylizer_case(name, shape, seed):
    rng = np.random.default_rng(seed)
    x = rng.random(shape).astype(np.float32)
    params = O.init_stylizer_params(seed=1)
    acts = []
    out = O.stylizer_forward(t64(x), [t64(p) for p in params], acts)
    # weights come from init_stylizer_params(seed=1); only inputs/outputs are stored
    np.savez_compressed(os.path.join(HERE, name), x=x, out=out.numpy().astype(np.float32),
                        act_means=np.array([a.mean().item() for a in acts]),
                        act_stds=np.array([a.std().item() for a in acts]),
                        act_corner=np.stack([a[0, 0, 0, :3].numpy() for a in acts]))


def step_case(name, seed):
    rng = np.random.default_rng(seed)
    batch = rng.random((2, 32, 24, 3)).astype(np.float32)
    style = rng.random((1, 30, 22, 3)).astype(np.float32)
    params = O.init_stylizer_params(seed=1)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean = np.array([103.9, 116.8, 123.7])
    sw, cw, tvw = 1e-2, 1e4, 1e8
    P = [t64(p).requires_grad_() for p in params]
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    with torch.no_grad():
        targets, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)
    loss, s, c, tv, out = O.train_step_loss(t64(batch), P, W, targets, t64(mean), sw, cw, tvw)
    grads = torch.autograd.grad(loss, P)
    m = [torch.zeros_like(p) for p in P]
    v = [torch.zeros_like(p) for p in P]
    with torch.no_grad():
        O.adam_tf1_step(P, grads, m, v, 1)
    np.savez_compressed(
        os.path.join(HERE, name), batch=batch, style=style, mean=mean,
        losses=np.array([loss.item(), s.item(), c.item(), tv.item()]),
        stylized=out.detach().numpy().astype(np.float32),
        grad_norms=np.array([g.norm().item() for g in grads]),
        grad_final_w=grads[45].numpy().astype(np.float32),
        post_adam_final_w=P[45].detach().numpy().astype(np.float32),
        gram_targets_trace=np.array([g.diagonal(dim1=1, dim2=2).sum().item() for g in targets]))


def gatys_case(name, seed):
    rng = np.random.default_rng(seed)
    content = rng.random((1, 24, 20, 3)).astype(np.float32)
    style = rng.random((1, 17, 23, 3)).astype(np.float32)   # odd sizes: SAME pooling clips windows
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean = np.zeros(3)
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    sw, cw, tvw = 1e-2, 1e4, 1e8
    with torch.no_grad():
        st, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)
        _, ct = O.vgg_taps(O.vgg_preprocess(t64(content), t64(mean)), W)
    img = t64(content).clone().requires_grad_()
    m, v = [torch.zeros_like(img)], [torch.zeros_like(img)]
    losses = []
    for t in range(1, 4):
        loss, s, c, tv = O.gatys_loss(img, W, st, ct, t64(mean), sw, cw, tvw)
        (g,) = torch.autograd.grad(loss, img)
        losses.append([loss.item(), s.item(), c.item(), tv.item()])
        with torch.no_grad():
            O.adam_tf1_step([img], [g], m, v, t, lr=0.02, beta1=0.99, beta2=0.999, eps=1e-1)
    np.savez_compressed(os.path.join(HERE, name), content=content, style=style, losses=np.array(losses),
                        image_after_3=img.detach().numpy().astype(np.float32))


if __name__ == '__main__':
    stylizer_case('stylizer_fwd_1x32x32.npz', (1, 32, 32, 3), 11)
    stylizer_case('stylizer_fwd_2x64x48.npz', (2, 64, 48, 3), 12)
    step_case('train_step_2x32x24.npz', 0)
    gatys_case('gatys_3steps_24x20.npz', 5)
    print('golden fixtures written to', HERE)
