"""CPU tests of the oracle: the torch restatement against the loop-level restatement,
against the committed golden fixtures, and against known answers.

PARITY UNPINNED (see oracle/__init__.py): the reference ships no golden vectors and
TensorFlow is not available, so these tests pin the oracle to itself and to first
principles, not to TensorFlow output.
"""
import os

import numpy as np
import pytest
import torch

from oracle import loops as L
from oracle import tf_ops as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def test_same_padding_rule():
    # out = ceil(in/s); pad_before = total // 2 (the odd element goes AFTER)
    assert O.same_pads(256, 3, 2) == (128, 0, 1)
    assert O.same_pads(255, 3, 2) == (128, 1, 1)
    assert O.same_pads(256, 9, 1) == (256, 4, 4)
    assert O.same_pads(256, 3, 1) == (256, 1, 1)
    assert O.same_pads(507, 2, 2) == (254, 0, 1)
    assert O.same_pads(5, 2, 2) == (3, 0, 1)
    assert L.same_pads(255, 3, 2) == O.same_pads(255, 3, 2)


@pytest.mark.parametrize('h,w,ci,co,k,s', [(6, 7, 3, 4, 3, 1), (6, 8, 3, 4, 3, 2), (7, 9, 2, 3, 3, 2),
                                           (10, 11, 3, 5, 9, 1), (5, 5, 2, 2, 2, 2), (1, 1, 2, 2, 3, 1)])
def test_conv_matches_index_formula(h, w, ci, co, k, s):
    rng = np.random.default_rng(0)
    x, wt = rng.standard_normal((2, h, w, ci)), rng.standard_normal((k, k, ci, co))
    got = O.conv2d_same(t64(x), t64(wt), s).numpy()
    assert np.abs(got - L.conv2d_same(x, wt, s)).max() < 1e-12


@pytest.mark.parametrize('h,w,ci,co,k,s', [(4, 5, 3, 2, 3, 2), (3, 3, 2, 4, 3, 2), (4, 4, 2, 2, 4, 2),
                                           (3, 4, 2, 3, 2, 2), (3, 4, 2, 3, 3, 1), (1, 1, 2, 2, 3, 2)])
def test_conv_transpose_matches_index_formula_and_is_the_adjoint(h, w, ci, co, k, s):
    rng = np.random.default_rng(1)
    x, wt = rng.standard_normal((2, h, w, ci)), rng.standard_normal((k, k, co, ci))
    got = O.conv2d_transpose_same(t64(x), t64(wt), s).numpy()
    assert got.shape == (2, h * s, w * s, co)
    assert np.abs(got - L.conv2d_transpose_same(x, wt, s)).max() < 1e-12
    # conv2d_transpose is DEFINED as the gradient of conv2d: <conv(y, W), x> == <y, deconv(x, W)>
    y = rng.standard_normal(got.shape)
    fwd = O.conv2d_same(t64(y), t64(wt), s).numpy()
    assert abs((fwd * x).sum() - (y * got).sum()) < 1e-9


def test_pool_norm_gram_tv_adam_match_loops():
    rng = np.random.default_rng(2)
    for h, w in [(6, 6), (7, 5), (1, 3), (127, 2)]:
        x = rng.standard_normal((2, h, w, 3))
        assert np.array_equal(O.max_pool_2x2_same(t64(x)).numpy(), L.max_pool_2x2_same(x))
    x = rng.standard_normal((2, 5, 6, 3))
    sc, sh = rng.standard_normal(3), rng.standard_normal(3)
    assert np.abs(O.instance_norm(t64(x), t64(sc), t64(sh)).numpy() - L.instance_norm(x, sc, sh)).max() < 1e-12
    assert np.abs(O.gram_matrix(t64(x)).numpy() - L.gram_matrix(x)).max() < 1e-12
    assert abs(O.total_variation_loss(t64(x)).item() - L.total_variation_loss(x)) < 1e-12
    p, g = rng.standard_normal(50), rng.standard_normal(50)
    P, m, v = [t64(p)], [torch.zeros(50, dtype=torch.float64)], [torch.zeros(50, dtype=torch.float64)]
    pm, mm, vm = p.copy(), np.zeros(50), np.zeros(50)
    for t in range(1, 4):
        O.adam_tf1_step(P, [t64(g * t)], m, v, t, lr=0.02, beta1=0.99, beta2=0.999, eps=1e-1)
        pm, mm, vm = L.adam_tf1(pm, g * t, mm, vm, t, 0.02, 0.99, 0.999, 1e-1)
    assert np.abs(P[0].numpy() - pm).max() < 1e-12
    # TF1 places epsilon outside the bias correction: differs from the textbook / torch form
    tp = torch.nn.Parameter(t64(p))
    opt = torch.optim.Adam([tp], lr=0.02, betas=(0.99, 0.999), eps=1e-1)
    for t in range(1, 4):
        tp.grad = t64(g * t)
        opt.step()
    assert np.abs(tp.detach().numpy() - pm).max() > 1e-3


def test_known_answers():
    rng = np.random.default_rng(3)
    x = t64(rng.standard_normal((2, 9, 7, 4)) * 3 + 1)
    scale, shift = t64([1.0, 2.0, 0.5, 1.5]), t64([0.0, 1.0, -1.0, 0.25])
    y = O.instance_norm(x, scale, shift)
    # per-(n,c) mean equals shift, variance equals scale^2 * var/(var+eps)
    assert torch.allclose(y.mean(dim=(1, 2)), shift.expand(2, 4), atol=1e-12)
    var = x.var(dim=(1, 2), unbiased=False)
    assert torch.allclose(y.var(dim=(1, 2), unbiased=False), scale ** 2 * var / (var + 1e-3), rtol=1e-10)
    g = O.gram_matrix(x)
    assert torch.allclose(g, g.transpose(1, 2))
    assert (torch.linalg.eigvalsh(g) > -1e-9).all()
    assert O.total_variation_loss(torch.full((1, 5, 6, 3), 0.3, dtype=torch.float64)).item() == 0.0
    # Gram divides by H*W only (model_utilities.py:380-384)
    ones = torch.ones((1, 4, 5, 2), dtype=torch.float64)
    assert torch.allclose(O.gram_matrix(ones), torch.ones((1, 2, 2), dtype=torch.float64))
    # style term is divided by len(content outputs) = 1, not by the number of style layers
    st = [torch.zeros((1, 2, 2), dtype=torch.float64)] * 5
    sv = [torch.ones((1, 2, 2), dtype=torch.float64)] * 5
    ct = [torch.zeros((1, 2, 2, 2), dtype=torch.float64)]
    total, s, c, tv = O.total_loss(ones, st, ct, sv, ct, 1.0, 1.0, 1.0)
    assert s.item() == 5.0 and c.item() == 0.0 and tv.item() == 0.0
    # 'starry_night_content' style: only the content term survives
    total, s, c, tv = O.total_loss(x[:, :, :, :3], st, ct, sv, [ct[0] + 2.0], 0.0, 1.0, 0.0)
    assert total.item() == c.item() == 4.0


def test_stylizer_shapes_and_parameter_inventory():
    shapes = O.stylizer_param_shapes()
    assert len(shapes) == 48
    assert sum(int(np.prod(s)) for s in shapes) == 1677638
    assert shapes[0] == (9, 9, 3, 32) and shapes[39] == (3, 3, 64, 128) and shapes[45] == (9, 9, 32, 3)
    params = O.init_stylizer_params(seed=1)
    assert all(np.abs(p).max() <= 0.2 + 1e-7 for p in params[0::3])     # truncated at 2 sigma
    assert all((p == 0).all() for p in params[1::3]) and all((p == 1).all() for p in params[2::3])
    out = O.stylizer_forward(t64(np.random.default_rng(0).random((1, 20, 28, 3))), [t64(p) for p in params])
    assert out.shape == (1, 20, 28, 3) and 0.0 < out.min() and out.max() < 1.0


@pytest.mark.parametrize('name', ['stylizer_fwd_1x32x32.npz', 'stylizer_fwd_2x64x48.npz'])
def test_golden_stylizer_forward(name):
    z = np.load(os.path.join(GOLD, name))
    acts = []
    out = O.stylizer_forward(t64(z['x']), [t64(p) for p in O.init_stylizer_params(seed=1)], acts)
    assert np.abs(out.numpy() - z['out']).max() < 1e-6
    assert np.allclose([a.mean().item() for a in acts], z['act_means'], rtol=1e-9, atol=1e-12)
    assert np.allclose([a.std().item() for a in acts], z['act_stds'], rtol=1e-9)
    assert np.allclose(np.stack([a[0, 0, 0, :3].numpy() for a in acts]), z['act_corner'], rtol=1e-9, atol=1e-12)


def test_golden_training_step():
    z = np.load(os.path.join(GOLD, 'train_step_2x32x24.npz'))
    params = O.init_stylizer_params(seed=1)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    P = [t64(p).requires_grad_() for p in params]
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    with torch.no_grad():
        targets, _ = O.vgg_taps(O.vgg_preprocess(t64(z['style']), t64(z['mean'])), W)
    loss, s, c, tv, out = O.train_step_loss(t64(z['batch']), P, W, targets, t64(z['mean']), 1e-2, 1e4, 1e8)
    assert np.allclose([loss.item(), s.item(), c.item(), tv.item()], z['losses'], rtol=1e-9)
    assert np.abs(out.detach().numpy() - z['stylized']).max() < 1e-6
    grads = torch.autograd.grad(loss, P)
    assert np.allclose([g.norm().item() for g in grads], z['grad_norms'], rtol=1e-8)
    assert np.allclose(grads[45].numpy(), z['grad_final_w'], rtol=1e-5, atol=1e-3)
    assert np.allclose([g.diagonal(dim1=1, dim2=2).sum().item() for g in targets], z['gram_targets_trace'], rtol=1e-9)


def test_golden_gatys_steps():
    z = np.load(os.path.join(GOLD, 'gatys_3steps_24x20.npz'))
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    mean = torch.zeros(3, dtype=torch.float64)
    with torch.no_grad():
        st, _ = O.vgg_taps(O.vgg_preprocess(t64(z['style']), mean), W)
        _, ct = O.vgg_taps(O.vgg_preprocess(t64(z['content']), mean), W)
    img = t64(z['content']).clone().requires_grad_()
    m, v = [torch.zeros_like(img)], [torch.zeros_like(img)]
    for t in range(1, 4):
        loss, s, c, tv = O.gatys_loss(img, W, st, ct, mean, 1e-2, 1e4, 1e8)
        assert np.allclose([loss.item(), s.item(), c.item(), tv.item()], z['losses'][t - 1], rtol=1e-9)
        (g,) = torch.autograd.grad(loss, img)
        with torch.no_grad():
            O.adam_tf1_step([img], [g], m, v, t, lr=0.02, beta1=0.99, beta2=0.999, eps=1e-1)
    assert np.abs(img.detach().numpy() - z['image_after_3']).max() < 1e-6
