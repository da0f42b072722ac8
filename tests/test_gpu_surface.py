"""GPU tests of the drop-in Python surface: the reference's module-level API evaluated
through the HIP kernels must reproduce the oracle (fp32 mode, 1e-3)."""
import os

import numpy as np
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def rel_err(got, ref):
    got = got.detach().double().cpu() if torch.is_tensor(got) else t64(got)
    ref = ref.detach().double().cpu() if torch.is_tensor(ref) else t64(ref)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def test_build_style_network_eager_matches_oracle(cuda):
    import full_graph_stylizer_network as fg
    import model_utilities as mu
    params = O.init_stylizer_params(seed=1)
    x = np.random.default_rng(0).random((2, 32, 24, 3)).astype(np.float32)
    ref = O.stylizer_forward(t64(x), [t64(p) for p in params])
    store = mu.VariableStore(cuda)
    store.load(params)
    with mu.variable_scope(store):
        out, cfg = fg.build_style_network(torch.tensor(x, device=cuda))
    assert rel_err(out, ref) <= 1e-3
    assert list(cfg) == ['conv_layers', 'residual_layers', 'deconv_layers']
    assert len(cfg['conv_layers']) == 3 and len(cfg['residual_layers']) == 5 and len(cfg['deconv_layers']) == 2
    # a fresh scope creates the 48 variables in the reference's order: W, shift, scale per block
    fresh = mu.VariableStore(cuda, seed=1)
    with mu.variable_scope(fresh):
        fg.build_style_network(torch.tensor(x, device=cuda))
    shapes = [tuple(v.shape) for v in fresh.variables]
    assert shapes == [tuple(s) for s in O.stylizer_param_shapes()]
    assert all((v == 0).all() for v in fresh.variables[1::3]) and all((v == 1).all() for v in fresh.variables[2::3])
    assert all(v.abs().max() <= 0.2 + 1e-6 for v in fresh.variables[0::3])
    # second pass through the same scope reuses the variables
    with mu.variable_scope(fresh):
        a, _ = fg.build_style_network(torch.tensor(x, device=cuda))
    with mu.variable_scope(fresh):
        b, _ = fg.build_style_network(torch.tensor(x, device=cuda))
    assert torch.equal(a, b) and len(fresh.variables) == 48


def test_vgg_builders_and_total_loss(cuda, tmp_path, monkeypatch):
    import model_utilities as mu
    import vgg
    w = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean_img = np.random.default_rng(1).random((224, 224, 3)).astype(np.float32) * 255
    path = str(tmp_path / 'vgg.mat')
    vgg.save_vgg_mat(path, w, mean_img)
    monkeypatch.setattr(vgg, 'VGG_PATH', path)
    monkeypatch.setattr(vgg, 'VGG_PATH_REPLICATE', path)
    monkeypatch.setattr(vgg, 'vgg_preprocessing_mean', np.array([0, 0, 0]))
    rng = np.random.default_rng(2)
    a = rng.random((1, 30, 22, 3)).astype(np.float32)      # odd pooled sizes
    b = rng.random((1, 30, 22, 3)).astype(np.float32)
    W = {k: (t64(x), t64(y)) for k, (x, y) in w.items()}
    # the reference preprocesses BEFORE the builder loads the .mat: mean is still zero here
    pa, pb = vgg.vgg_preprocess_input(torch.tensor(a)), vgg.vgg_preprocess_input(torch.tensor(b))
    (sa, sb), (ca, cb) = vgg.build_vgg_network_replicate_from_mat([pa, pb], replications=2)
    zero = torch.zeros(3, dtype=torch.float64)
    ref_sa, ref_ca = O.vgg_taps(O.vgg_preprocess(t64(a), zero), W)
    ref_sb, ref_cb = O.vgg_taps(O.vgg_preprocess(t64(b), zero), W)
    for g, r in zip(sa + sb, ref_sa + ref_sb):
        assert rel_err(g, r) <= 1e-3
    assert rel_err(ca[0], ref_ca[0]) <= 1e-3 and rel_err(cb[0], ref_cb[0]) <= 1e-3
    # ... and afterwards the global holds the .mat mean pixel
    assert np.allclose(vgg.vgg_preprocessing_mean, mean_img.mean(axis=(0, 1)))
    mean = t64(vgg.vgg_preprocessing_mean)
    s1, c1 = vgg.build_vgg_network_from_mat(vgg.vgg_preprocess_input(torch.tensor(a)))
    ref_s1, ref_c1 = O.vgg_taps(O.vgg_preprocess(t64(a), mean), W, relu_taps=True)
    for g, r in zip(s1, ref_s1):
        assert rel_err(g, r) <= 1e-3
    assert rel_err(c1[0], ref_c1[0]) <= 1e-3
    # StyleTransfer.total_loss with the reference's argument order
    img = torch.tensor(b, device=cuda)
    got = mu.StyleTransfer.total_loss(img, sa, ca, sb, cb, style_weight=1e-2, content_weight=1e4,
                                      total_variation_weight=1e8)
    want = O.total_loss(t64(b), ref_sa, ref_ca, ref_sb, ref_cb, 1e-2, 1e4, 1e8)
    for g, r in zip(got, want):
        assert abs(g.item() - r.item()) <= 1e-3 * abs(r.item())
    assert abs(mu.StyleTransfer.total_variation_loss(img).item() - O.total_variation_loss(t64(b)).item()) <= 1e-5
    assert rel_err(mu.StyleTransfer.gram_matrix(ca[0]), O.gram_matrix(ref_ca[0])) <= 1e-3
    assert mu.StyleTransfer.clip_0_1(torch.tensor([-1.0, 0.5, 2.0], device=cuda)).tolist() == [0.0, 0.5, 1.0]


def test_style_net_service_and_cli(cuda, tmp_path):
    import full_graph_stylizer_network as fg
    import image_utilities as iu
    params = O.init_stylizer_params(seed=5)
    name = 'STYLE_NET_2019-09-22-22-24'
    (tmp_path / name).mkdir()
    fg.save_checkpoint(params, str(tmp_path / name), name)
    svc = fg.StyleNetService(name, str(tmp_path))
    assert svc.loaded is False and svc.model_name == name and svc.model_dir == str(tmp_path)
    svc.start()
    svc.wait_for_ready()
    x = np.random.default_rng(0).random((1, 40, 28, 3)).astype(np.float32)
    y = svc.run_on_image(x)
    assert y.shape == x.shape and y.dtype == np.float32
    ref = O.stylizer_forward(t64(x), [t64(p) for p in params])
    assert rel_err(y, ref) <= 1e-3
    # a second image size is served by the same service (the reference's placeholder is shape-free)
    x2 = np.random.default_rng(1).random((1, 24, 36, 3)).astype(np.float32)
    assert rel_err(svc.run_on_image(x2), O.stylizer_forward(t64(x2), [t64(p) for p in params])) <= 1e-3
    svc.close()
    # CLI form "MODEL BASE_DIR SRC DST" end to end: file in, file out
    src, dst = str(tmp_path / 'in.png'), str(tmp_path / 'out.png')
    iu.save_image(x, src)
    fg.main(['prog', name, str(tmp_path), src, dst])
    out = iu.load_image(dst)
    x8 = iu.load_image(src)
    ref8 = O.stylizer_forward(t64(x8), [t64(p) for p in params]).numpy()
    assert out.shape == x.shape and np.abs(out - ref8).max() <= 1.5 / 255
