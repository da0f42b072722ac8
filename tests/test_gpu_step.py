"""End-to-end parity of the HIP executors against the CPU oracle on small shapes:
stylizer forward with its 16 activations, VGG taps, the four losses, the 48
parameter gradients, post-Adam weights, Gatys steps.

fp32 mode: max |err| / max |ref| <= 1e-3 on activations and losses (BASELINE.json).
bf16 mode: PSNR >= 40 dB on the stylized output, Gram matrices within 1e-2.
"""
import numpy as np
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    got = torch.as_tensor(np.asarray(got.detach().float().cpu() if torch.is_tensor(got) else got)).double()
    ref = torch.as_tensor(np.asarray(ref.detach().cpu() if torch.is_tensor(ref) else ref)).double()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def psnr(got, ref):
    mse = ((got.double().cpu() - ref.double().cpu()) ** 2).mean().item()
    return 10.0 * np.log10(1.0 / max(mse, 1e-30))


def small_setup(seed=0, n=2, h=32, w=24):
    rng = np.random.default_rng(seed)
    batch = rng.random((n, h, w, 3)).astype(np.float32)
    style = rng.random((1, 30, 22, 3)).astype(np.float32)   # odd pooled sizes: 15x11 -> 8x6 -> 4x3 -> 2x2
    params = O.init_stylizer_params(seed=1)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean = np.array([103.9, 116.8, 123.7], np.float32)
    return batch, style, params, vggw, mean


def oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=1):
    """fp64 oracle: returns per-step (losses, stylized, grads) and the final parameters."""
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    P = [t64(p).requires_grad_() for p in params]
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    m = [torch.zeros_like(p) for p in P]
    v = [torch.zeros_like(p) for p in P]
    with torch.no_grad():
        style_targets, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)
    out = []
    for t in range(1, steps + 1):
        loss, s, c, tv, stylized = O.train_step_loss(t64(batch), P, W, style_targets, t64(mean), sw, cw, tvw)
        grads = torch.autograd.grad(loss, P)
        out.append(([loss.item(), s.item(), c.item(), tv.item()], stylized.detach(), [g.clone() for g in grads]))
        with torch.no_grad():
            O.adam_tf1_step(P, grads, m, v, t)
    return out, [p.detach() for p in P], style_targets


def test_stylizer_forward_activations_fp32(cuda):
    from stmi.stylizer import StylizerNet
    batch, _, params, _, _ = small_setup()
    acts = []
    ref = O.stylizer_forward(torch.tensor(batch, dtype=torch.float64), [torch.tensor(p, dtype=torch.float64) for p in params], acts)
    net = StylizerNet(batch.shape[0], batch.shape[1], batch.shape[2], cuda, torch.float32, params=params, training=True)
    out = net.forward(torch.tensor(batch, device=cuda))
    assert rel_err(out, ref) <= 1e-3
    assert len(acts) == 16
    for i, (b, a) in enumerate(zip(net.blocks, acts)):
        assert rel_err(b.y, a) <= 1e-3, 'activation %d' % i
    # inference executor (buffer pool) gives the same image
    net_inf = StylizerNet(batch.shape[0], batch.shape[1], batch.shape[2], cuda, torch.float32, params=params)
    assert rel_err(net_inf.forward(torch.tensor(batch, device=cuda)), ref) <= 1e-3


def test_stylizer_forward_bf16_psnr(cuda):
    from stmi.stylizer import StylizerNet
    rng = np.random.default_rng(3)
    batch = rng.random((1, 64, 48, 3)).astype(np.float32)
    params = O.init_stylizer_params(seed=1)
    ref = O.stylizer_forward(torch.tensor(batch, dtype=torch.float64), [torch.tensor(p, dtype=torch.float64) for p in params])
    net = StylizerNet(1, 64, 48, cuda, torch.bfloat16, params=params)
    out = net.forward(torch.tensor(batch, device=cuda))
    assert psnr(out, ref) >= 40.0


@pytest.mark.parametrize('relu_taps', [False, True], ids=['conv_taps', 'relu_taps'])
def test_vgg_taps(cuda, relu_taps):
    from stmi.vggnet import VggFeatures
    _, style, _, vggw, mean = small_setup()
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    grams_ref, content_ref = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W, relu_taps=relu_taps)
    # bf16 Gram gate is 1e-2 on the training path (conv taps); the relu-tap variant is checked a
    # little looser here because this reduced-width net ends on 2x2 maps (little averaging)
    for dtype, gtol in ((torch.float32, 1e-3), (torch.bfloat16, 2e-2 if relu_taps else 1e-2)):
        net = VggFeatures(vggw, mean, cuda, dtype, with_backward=False, last='relu5_4' if relu_taps else 'conv5_2')
        rep = net.replica(1, style.shape[1], style.shape[2], relu_taps=relu_taps)
        grams, content = rep.forward(torch.tensor(style, device=cuda))
        for g, r in zip(grams, grams_ref):
            assert rel_err(g, r) <= gtol
        assert rel_err(content, content_ref[0]) <= (1e-3 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize('use_graph', [False, True], ids=['eager', 'graph'])
def test_training_step_fp32(cuda, use_graph):
    from stmi.trainer import StyleTrainer
    batch, style, params, vggw, mean = small_setup()
    sw, cw, tvw = 1e-2, 1e4, 1e8   # starry_night_transfer
    ref, ref_params, ref_targets = oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=2)
    tr = StyleTrainer(batch.shape[0], batch.shape[1], batch.shape[2], cuda, vggw, mean, sw, cw, tvw,
                      dtype=torch.float32, params=params, use_graph=use_graph)
    targets = tr.set_style_image(torch.tensor(style))
    for g, r in zip(targets, ref_targets):
        assert rel_err(g, r) <= 1e-3
    xb = torch.tensor(batch, device=cuda)
    own_grads = []
    for t in range(2):
        tr.step(xb)
        losses = tr.read_losses()
        for got, want in zip(losses, ref[t][0]):
            assert abs(got - want) <= 1e-3 * abs(want), (t, losses, ref[t][0])
        assert rel_err(tr.net.out_f32, ref[t][1]) <= 1e-3
        own_grads.append([g.double().cpu().clone() for g in tr.net.grad_views])
        if t == 0:
            for i, (g, r) in enumerate(zip(tr.net.grad_views, ref[0][2])):
                assert rel_err(g, r) <= 2e-3, 'gradient of variable %d' % i
    # Adam in situ: the oracle's TF1 Adam applied to the gradients the GPU produced must land on
    # the GPU's weights (tight).  Against the all-oracle weights only the direction of the 2-step
    # update is compared: Adam normalises every element by sqrt(v), so an element whose gradient
    # is tiny relative to its tensor turns a 1e-3 (max-norm) gradient error into a large one.
    P = [torch.tensor(p, dtype=torch.float64) for p in params]
    m = [torch.zeros_like(p) for p in P]
    v = [torch.zeros_like(p) for p in P]
    for t in range(2):
        O.adam_tf1_step(P, own_grads[t], m, v, t + 1)
    for i, (p, mine, r, p0) in enumerate(zip(tr.net.views, P, ref_params, params)):
        assert rel_err(p, mine) <= 1e-5, 'Adam on own gradients, variable %d' % i
        p0 = torch.tensor(p0, dtype=torch.float64)
        du, dr = (p.double().cpu() - p0).reshape(-1), (r - p0).reshape(-1)
        assert (du @ dr / (du.norm() * dr.norm())).item() >= 0.999, 'update direction of variable %d' % i


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_training_is_bit_reproducible(cuda, dtype):
    """No kernel of a step accumulates with float atomics (weight gradients, Gram matrices, parameter gradients
    of the norms and loss terms are partial sums folded in a fixed order), so repeated runs from the same state
    -- eager launches or graph replay -- give the same bits: weights, Adam moments and losses after two steps."""
    from stmi.trainer import StyleTrainer
    batch, style, params, vggw, mean = small_setup()
    runs = []
    for use_graph in (False, True, True):
        tr = StyleTrainer(batch.shape[0], batch.shape[1], batch.shape[2], cuda, vggw, mean, 1e-2, 1e4, 1e8,
                          dtype=dtype, params=params, use_graph=use_graph)
        tr.set_style_image(torch.tensor(style))
        xb = torch.tensor(batch, device=cuda)
        tr.step(xb)
        first = tr.losses.clone()
        tr.step(xb)
        torch.cuda.synchronize()
        runs.append((tr.net.flat.clone(), tr.m.clone(), tr.v.clone(), first, tr.losses.clone(), tr.net.flat_grad.clone()))
        del tr
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)


def test_training_step_with_resident_patch_weight_gradients(cuda, monkeypatch):
    """The residual layers' weight gradients from the resident-patch kernel (chosen by itself on large crops, asked
    for here) inside a whole training step: the same gradients as with the tap-per-block kernel up to the order of
    fp32 additions, and bit-reproducible."""
    from stmi import ops
    from stmi.trainer import StyleTrainer
    batch, style, params, vggw, mean = small_setup(h=64, w=128)    # residual layers on 16 x 32 pixel planes
    grads = []
    for min_pix in ('0', '0', '1000000000'):
        monkeypatch.setenv('STMI_WG_RES_MIN_PIX', min_pix)
        tr = StyleTrainer(batch.shape[0], 64, 128, cuda, vggw, mean, 1e-2, 1e4, 1e8, dtype=torch.bfloat16, params=params,
                          use_graph=False)
        tr.set_style_image(torch.tensor(style))
        tr.step(torch.tensor(batch, device=cuda))
        torch.cuda.synchronize()
        grads.append(tr.net.flat_grad.clone())
        del tr
    assert torch.equal(grads[0], grads[1])
    assert not torch.equal(grads[0], grads[2]), 'both runs took the same kernel'
    assert (grads[0] - grads[2]).abs().max() <= 1e-4 * grads[2].abs().max()


def test_training_step_bf16_tracks_oracle(cuda):
    from stmi.trainer import StyleTrainer
    batch, style, params, vggw, mean = small_setup(h=64, w=48)
    sw, cw, tvw = 1e-2, 1e4, 1e8
    ref, _, _ = oracle_step(batch, style, params, vggw, mean, sw, cw, tvw, steps=1)
    tr = StyleTrainer(batch.shape[0], 64, 48, cuda, vggw, mean, sw, cw, tvw, dtype=torch.bfloat16, params=params)
    tr.set_style_image(torch.tensor(style))
    tr.step(torch.tensor(batch, device=cuda))
    losses = tr.read_losses()
    assert psnr(tr.net.out_f32, ref[0][1]) >= 40.0
    for got, want in zip(losses, ref[0][0]):
        assert abs(got - want) <= 5e-2 * abs(want), (losses, ref[0][0])
    # Gradient direction vs the fp64 oracle.  The deviation comes from ROUNDING THE FORWARD ACTIVATIONS to
    # bf16, and grows with the depth of the path a loss term's gradient travels: fp32 kernels fed
    # bf16-rounded activations give per-term cosines TV 0.997, style 0.96, content 0.93 (the content
    # gradient crosses all fourteen VGG layers and the whole stylizer; ReLU masks and pool routing of a
    # deep random network shift under 2^-9 perturbations), 0.96 for the sum; rounding only the backward
    # tensors gives 1.000 (scripts/diag_bf16_grad3.py, scripts/diag_bf16_grad2.py).  The bf16 mode itself
    # measures 0.94.  It is therefore a reduced-fidelity mode by construction: the gate below guards
    # against gross errors, fp32 is the parity mode, and the 20-step trajectory test below shows that the
    # bf16 mode nevertheless drives the loss like fp32.
    g = tr.net.flat_grad.double().cpu()
    r = torch.cat([x.reshape(-1) for x in ref[0][2]])
    cos = (g @ r / (g.norm() * r.norm())).item()
    assert cos >= 0.90, cos


def test_bf16_training_follows_fp32_trajectory(cuda):
    """20 steps from the same initial weights: the bf16-MFMA mode must drive the loss down like fp32."""
    from stmi.trainer import StyleTrainer
    batch, style, params, vggw, mean = small_setup(h=64, w=48)
    sw, cw, tvw = 1e-2, 1e4, 1e8
    final = {}
    for dt in (torch.float32, torch.bfloat16):
        tr = StyleTrainer(batch.shape[0], 64, 48, cuda, vggw, mean, sw, cw, tvw, dtype=dt, params=params)
        tr.set_style_image(torch.tensor(style))
        xb = torch.tensor(batch, device=cuda)
        tr.step(xb)
        first = tr.read_losses()[0]
        for _ in range(19):
            tr.step(xb)
        final[dt] = (first, tr.read_losses()[0])
    f32, b16 = final[torch.float32], final[torch.bfloat16]
    assert f32[1] < 0.7 * f32[0], f32
    assert abs(b16[1] - f32[1]) <= 0.1 * f32[1], final


def test_gatys_steps_fp32(cuda):
    from stmi.trainer import GatysOptimizer
    batch, style, _, vggw, mean = small_setup(n=1)
    sw, cw, tvw = 1e-2, 1e4, 1e8
    t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    with torch.no_grad():
        st, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)
        _, ct = O.vgg_taps(O.vgg_preprocess(t64(batch), t64(mean)), W)
    img = t64(batch).clone().requires_grad_()
    m, v = [torch.zeros_like(img)], [torch.zeros_like(img)]
    opt = GatysOptimizer(torch.tensor(batch), cuda, vggw, mean, sw, cw, tvw, dtype=torch.float32)
    opt.set_style_image(torch.tensor(style))
    for t in range(1, 4):
        loss, s, c, tv = O.gatys_loss(img, W, st, ct, t64(mean), sw, cw, tvw)
        (g,) = torch.autograd.grad(loss, img)
        with torch.no_grad():
            O.adam_tf1_step([img], [g], m, v, t, lr=0.02, beta1=0.99, beta2=0.999, eps=1e-1)
        opt.step()
        got = opt.read_losses()
        for a, b in zip(got, [loss.item(), s.item(), c.item(), tv.item()]):
            assert abs(a - b) <= 1e-3 * abs(b) + 1e-9, (t, got)
        assert rel_err(opt.image, img) <= 1e-3
