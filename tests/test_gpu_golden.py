"""The HIP path against the committed fixtures in tests/golden/*.npz (written by
tests/golden/make_golden.py from the fp64 oracle; PARITY UNPINNED: no vector in them was
produced by TensorFlow).  fp32 mode, tolerance 1e-3 as max |err| / max |ref|; the weights are
the seeded arrays the generator used (oracle.init_stylizer_params / synthetic_vgg_weights).
"""
import os

import numpy as np
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel(got, ref):
    got = np.asarray(got.detach().double().cpu() if torch.is_tensor(got) else got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize('name', ['stylizer_fwd_1x32x32.npz', 'stylizer_fwd_2x64x48.npz'])
def test_stylizer_forward_fixture(cuda, name):
    from stmi.stylizer import StylizerNet
    fx = load(name)
    x = fx['x']
    net = StylizerNet(x.shape[0], x.shape[1], x.shape[2], cuda, torch.float32,
                      params=O.init_stylizer_params(seed=1), training=True)
    out = net.forward(torch.tensor(x, device=cuda))
    assert rel(out, fx['out']) <= 1e-3
    assert len(net.blocks) == len(fx['act_means']) == 16
    for i, b in enumerate(net.blocks):
        y = b.y.double()
        scale = max(abs(fx['act_means'][i]), fx['act_stds'][i])
        assert abs(y.mean().item() - fx['act_means'][i]) <= 1e-3 * scale, 'mean of activation %d' % i
        assert abs(y.std().item() - fx['act_stds'][i]) <= 1e-3 * scale, 'std of activation %d' % i
        corner = y[0, 0, 0, :3].cpu().numpy()
        assert np.abs(corner - fx['act_corner'][i]).max() <= 1e-3 * max(np.abs(y.cpu().numpy()).max(), 1e-30), i


def test_training_step_fixture(cuda):
    from stmi.trainer import StyleTrainer
    fx = load('train_step_2x32x24.npz')
    batch, style, mean = fx['batch'], fx['style'], fx['mean'].astype(np.float32)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    tr = StyleTrainer(batch.shape[0], batch.shape[1], batch.shape[2], cuda, vggw, mean, 1e-2, 1e4, 1e8,
                      dtype=torch.float32, params=O.init_stylizer_params(seed=1), use_graph=False)
    targets = tr.set_style_image(torch.tensor(style))
    traces = [g.double().diagonal(dim1=1, dim2=2).sum().item() for g in targets]
    for got, want in zip(traces, fx['gram_targets_trace']):
        assert abs(got - want) <= 1e-3 * abs(want)
    tr.step(torch.tensor(batch, device=cuda))
    for got, want in zip(tr.read_losses(), fx['losses']):
        assert abs(got - want) <= 1e-3 * abs(want)
    assert rel(tr.net.out_f32, fx['stylized']) <= 1e-3
    norms = np.array([g.double().norm().item() for g in tr.net.grad_views])
    assert np.abs(norms - fx['grad_norms']).max() <= 2e-3 * fx['grad_norms'].max()
    assert np.all(np.abs(norms - fx['grad_norms']) <= 2e-2 * fx['grad_norms'] + 1e-12)
    assert rel(tr.net.grad_views[45], fx['grad_final_w']) <= 2e-3
    # first TF1 Adam step moves every element by lr * g / (|g| + eps'): compare where the sign is certain
    moved = tr.net.views[45].double().cpu().numpy() - O.init_stylizer_params(seed=1)[45]
    want = fx['post_adam_final_w'].astype(np.float64) - O.init_stylizer_params(seed=1)[45]
    sure = np.abs(fx['grad_final_w']) > 1e-2 * np.abs(fx['grad_final_w']).max()
    assert np.abs(moved - want)[sure].max() <= 2e-5     # a step is 1e-3; float32 storage of the fixture


def test_gatys_fixture(cuda):
    from stmi.trainer import GatysOptimizer
    fx = load('gatys_3steps_24x20.npz')
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    opt = GatysOptimizer(torch.tensor(fx['content']), cuda, vggw, np.zeros(3, np.float32), 1e-2, 1e4, 1e8,
                         dtype=torch.float32)
    opt.set_style_image(torch.tensor(fx['style']))
    for t in range(3):
        opt.step()
        for got, want in zip(opt.read_losses(), fx['losses'][t]):
            assert abs(got - want) <= 1e-3 * abs(want) + 1e-9, (t, got, want)
    assert rel(opt.image, fx['image_after_3']) <= 1e-3
