"""The reference's ENTRY FUNCTIONS executed for real on the GPU, checked against the CPU oracle:

  train_style_network   (reference full_graph_stylizer_network.py:85-232): dataset directory of JPEGs, epoch
                        loop with the ragged last batch, config.json, per-epoch sample JPEG, checkpoint
  normal_style_transfer (:235-346): the Gatys loop through the module function
  GatysOptimizer in bf16 (the dtype the 512 x 512 benchmark configuration names)
  the Django view's call sequence (ml_services/style_net_service/views.py:23-30, :46-54): three services
                        started back to back, request threads hitting them concurrently
  a TensorFlow tensor-bundle checkpoint served by StyleNetService
  batch-16 1920 x 1080 inference: per-frame independence and PSNR against the oracle at full size

Tolerances: fp32 mode 1e-3 (max |err| / max |ref|), bf16 mode PSNR >= 40 dB on images.
"""
import json
import os
import threading

import numpy as np
import pytest
import torch

from oracle import tf_ops as O

pytestmark = pytest.mark.gpu
t64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)


def rel_err(got, ref):
    got = got.detach().double().cpu() if torch.is_tensor(got) else t64(got)
    ref = ref.detach().double().cpu() if torch.is_tensor(ref) else t64(ref)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def psnr(got, ref):
    got = got.detach().double().cpu() if torch.is_tensor(got) else t64(got)
    ref = ref.detach().double().cpu() if torch.is_tensor(ref) else t64(ref)
    return 10.0 * np.log10(1.0 / max(((got - ref) ** 2).mean().item(), 1e-30))


def smooth_image(rng, h, w):
    """A low-frequency test picture (survives JPEG well, unlike white noise)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.zeros((h, w, 3), np.float32)
    for c in range(3):
        fy, fx, ph = rng.uniform(0.02, 0.2), rng.uniform(0.02, 0.2), rng.uniform(0, 6.28)
        img[:, :, c] = 0.5 + 0.4 * np.sin(fy * yy + fx * xx + ph) * np.cos(0.5 * fx * xx - ph)
    return img[None]


@pytest.fixture
def sandbox(tmp_path, monkeypatch):
    """Reference directory layout under tmp_path: models/, a clipped-COCO directory, style and content
    pictures, a synthetic reduced-width VGG .mat; module constants pointed at it."""
    import constants
    import full_graph_stylizer_network as fg
    import image_utilities as iu
    import model_utilities as mu
    import vgg
    rng = np.random.default_rng(11)
    models = tmp_path / 'models'
    coco = tmp_path / 'coco_clipped'
    stylized = tmp_path / 'stylized_images'
    for d in (models, coco, stylized):
        d.mkdir()
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean_img = rng.random((224, 224, 3)).astype(np.float32) * 255
    mat = str(tmp_path / 'vgg.mat')
    vgg.save_vgg_mat(mat, vggw, mean_img)
    monkeypatch.setattr(vgg, 'VGG_PATH', mat)
    monkeypatch.setattr(vgg, 'VGG_PATH_REPLICATE', mat)
    monkeypatch.setattr(vgg, 'vgg_preprocessing_mean', np.array([0, 0, 0]))
    monkeypatch.setattr(constants, 'MODELS_DIR', str(models))
    monkeypatch.setattr(constants, 'STYLIZED_IMAGES_DIR', str(stylized))
    monkeypatch.setattr(iu, 'COCO_DATASET_CLIPPED', str(coco))
    style_path = str(tmp_path / 'style.jpg')
    iu.save_image(smooth_image(rng, 30, 22), style_path)
    monkeypatch.setitem(mu.StyleTransfer.STYLE_CONFIG_DICT, 'unit_test_style',
                        mu.StyleConfig(style_path, 1e-2, 1e4, 1e8))
    return dict(tmp=tmp_path, models=models, coco=coco, stylized=stylized, vggw=vggw, style_path=style_path,
                rng=rng, fg=fg, iu=iu, vgg=vgg)


def oracle_training_run(batches, style, params, vggw, sw, cw, tvw):
    """fp64 oracle: the sequence of TF1-Adam steps train_style_network performs on `batches`
    (mean = 0: the reference's graph is built before the .mat mean is assigned)."""
    P = [t64(p).requires_grad_() for p in params]
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    zero = torch.zeros(3, dtype=torch.float64)
    m = [torch.zeros_like(p) for p in P]
    v = [torch.zeros_like(p) for p in P]
    with torch.no_grad():
        targets, _ = O.vgg_taps(O.vgg_preprocess(t64(style), zero), W)
    losses = []
    for t, batch in enumerate(batches, 1):
        loss, s, c, tv, _ = O.train_step_loss(t64(batch), P, W, targets, zero, sw, cw, tvw)
        grads = torch.autograd.grad(loss, P)
        losses.append([loss.item(), s.item(), c.item(), tv.item()])
        with torch.no_grad():
            O.adam_tf1_step(P, grads, m, v, t)
    return losses, [p.detach() for p in P]


def test_train_style_network_runs_the_reference_loop(cuda, sandbox, monkeypatch):
    """5 pictures, batch 4: one full batch and a ragged batch of ONE per epoch, as the reference's
    1001-picture dataset ends every epoch (full_graph_stylizer_network.py:196-203)."""
    fg, iu = sandbox['fg'], sandbox['iu']
    h, w = 32, 24
    monkeypatch.setattr(fg, 'TRAIN_IMG_DIMS', [None, h, w, 3])
    monkeypatch.setenv('STMI_DTYPE', 'f32')
    for i in range(5):
        iu.save_image(smooth_image(sandbox['rng'], h, w), str(sandbox['coco'] / ('img_%d.jpg' % i)))
    test_img_path = str(sandbox['tmp'] / 'antelope.jpg')
    iu.save_image(smooth_image(sandbox['rng'], 40, 28), test_img_path)

    # the order the function will see: same loader, same shuffle stream
    probe = iu.PreprocessedCocoDatasetManager(target_dim=(256, 256), num_images=fg.TOTAL_IMAGES)
    assert len(probe.get_images()) == 5
    np.random.seed(4321)
    probe.shuffle_loaded_images()
    order = [np.asarray(x, np.float32) for x in probe.get_images()]
    batches = [np.stack(order[:4]), np.stack(order[4:])]
    assert batches[1].shape[0] == 1

    np.random.seed(4321)
    logs, sampled = [], []
    real_save = fg.save_image
    monkeypatch.setattr(fg, 'save_image', lambda img, path: (sampled.append((np.array(img), path)), real_save(img, path))[1])
    name = fg.train_style_network('unit_test_style', test_img_path, epochs=1, log=logs.append)
    target_dir = os.path.join(str(sandbox['models']), name)
    assert name.startswith('STYLE_NET_') and name.endswith('_unit_test_style')

    # artefacts (full_graph_stylizer_network.py:162-173, :222-223, :231)
    cfg = json.load(open(os.path.join(target_dir, 'config.json')))
    assert list(cfg) == ['conv_layers', 'residual_layers', 'deconv_layers', 'style_config', 'style_layers',
                         'content_layers', 'batch_size', 'data_size', 'epochs', 'LR', 'config_name']
    assert cfg['batch_size'] == 4 and cfg['LR'] == 1e-3 and cfg['config_name'] == 'unit_test_style'
    assert len(cfg['conv_layers']) == 3 and len(cfg['residual_layers']) == 5 and len(cfg['deconv_layers']) == 2
    assert os.path.exists(os.path.join(target_dir, '0_style_transfer_sample.jpg'))
    assert os.path.exists(os.path.join(target_dir, 'checkpoint'))
    ckpt = fg.latest_checkpoint(target_dir)
    got_params = fg.load_checkpoint(ckpt)
    assert len(got_params) == 48
    assert any('Epoch 1 of 1' in str(x) for x in logs)

    # numbers: both steps' losses and the weights after full batch + ragged batch
    style = iu.load_image(sandbox['style_path'])
    params0 = O.init_stylizer_params(seed=1)
    ref_losses, ref_params = oracle_training_run(batches, style, params0, sandbox['vggw'], 1e-2, 1e4, 1e8)
    logged = [json.loads(x) for x in open(os.path.join(target_dir, 'losses.jsonl'))]
    assert [r['step'] for r in logged] == [0, 1]
    for rec, want in zip(logged, ref_losses):
        for key, wv in zip(('total', 'style', 'content', 'tv'), want):
            assert abs(rec[key] - wv) <= 1e-3 * abs(wv), (rec, want)
    # Adam divides every element by sqrt(v): an element whose gradient is tiny turns a 1e-3 gradient error
    # into a large update error, so the weights are compared by the direction of the two-step update
    for i, (p, r, p0) in enumerate(zip(got_params, ref_params, params0)):
        du, dr = (t64(p) - t64(p0)).reshape(-1), (r - t64(p0)).reshape(-1)
        assert du.norm() > 0
        assert (du @ dr / (du.norm() * dr.norm())).item() >= 0.999, 'update direction of variable %d' % i
        assert (t64(p) - r).abs().max().item() <= 4.1e-3, 'two Adam steps move a weight by at most 2e-3 each way'
    # the per-epoch sample is the stylizer forward of the weights that were saved (the array handed to
    # save_image is compared: the JPEG itself is lossy on a random-init network's noisy output)
    (sample_arr, sample_path), = sampled
    assert sample_path == os.path.join(target_dir, '0_style_transfer_sample.jpg')
    want = O.stylizer_forward(t64(iu.load_image(test_img_path)), [t64(p) for p in got_params])
    assert rel_err(sample_arr, want) <= 1e-3
    assert iu.load_image(sample_path).shape == tuple(want.shape)

    # the saved model is what the service behind run_on_image loads
    src, dst = str(sandbox['tmp'] / 'in.png'), str(sandbox['tmp'] / 'out.png')
    iu.save_image(smooth_image(sandbox['rng'], 24, 36), src)
    fg.run_on_image(src, dst, model_path=name, model_dir=str(sandbox['models']))
    out = iu.load_image(dst)
    ref8 = O.stylizer_forward(t64(iu.load_image(src)), [t64(p) for p in got_params]).numpy()
    assert np.abs(out - ref8).max() <= 1.5 / 255


def test_train_style_network_default_size_bf16(cuda, sandbox, monkeypatch):
    """The function as the reference runs it (256 x 256, TRAIN_IMG_DIMS untouched) in the bf16-MFMA mode:
    9 pictures = two full batches + a ragged one, two epochs; losses finite and decreasing on average."""
    fg, iu = sandbox['fg'], sandbox['iu']
    monkeypatch.setenv('STMI_DTYPE', 'bf16')
    for i in range(9):
        iu.save_image(smooth_image(sandbox['rng'], 256, 256), str(sandbox['coco'] / ('img_%d.jpg' % i)))
    test_img_path = str(sandbox['tmp'] / 'antelope_256.jpg')
    iu.save_image(smooth_image(sandbox['rng'], 256, 256), test_img_path)
    name = fg.train_style_network('unit_test_style', test_img_path, epochs=2, log=lambda *_: None)
    target_dir = os.path.join(str(sandbox['models']), name)
    logged = [json.loads(x) for x in open(os.path.join(target_dir, 'losses.jsonl'))]
    assert [r['step'] for r in logged] == list(range(6))
    assert all(np.isfinite(r['total']) and r['total'] > 0 for r in logged)
    for e in (0, 1):
        img = iu.load_image(os.path.join(target_dir, '%d_style_transfer_sample.jpg' % e))
        assert img.shape == (1, 256, 256, 3)
    assert len(fg.load_checkpoint(fg.latest_checkpoint(target_dir))) == 48


def gatys_oracle(content, style, vggw, mean, sw, cw, tvw, steps):
    W = {k: (t64(w), t64(b)) for k, (w, b) in vggw.items()}
    with torch.no_grad():
        st, _ = O.vgg_taps(O.vgg_preprocess(t64(style), t64(mean)), W)
        _, ct = O.vgg_taps(O.vgg_preprocess(t64(content), t64(mean)), W)
    img = t64(content).clone().requires_grad_()
    m, v = [torch.zeros_like(img)], [torch.zeros_like(img)]
    losses = []
    for t in range(1, steps + 1):
        loss, s, c, tv = O.gatys_loss(img, W, st, ct, t64(mean), sw, cw, tvw)
        (g,) = torch.autograd.grad(loss, img)
        losses.append([loss.item(), s.item(), c.item(), tv.item()])
        with torch.no_grad():
            O.adam_tf1_step([img], [g], m, v, t, lr=0.02, beta1=0.99, beta2=0.999, eps=1e-1)
    return img.detach(), losses


def test_normal_style_transfer_function(cuda, sandbox, monkeypatch):
    fg, iu = sandbox['fg'], sandbox['iu']
    monkeypatch.setenv('STMI_DTYPE', 'f32')
    content_path = str(sandbox['tmp'] / 'antelope_small.jpg')
    iu.save_image(smooth_image(sandbox['rng'], 32, 24), content_path)
    monkeypatch.setattr(fg, 'CONTENT_IMAGE_1', content_path)
    got = fg.normal_style_transfer('unit_test_style', epochs=1, steps_per_epoch=3, log=lambda *_: None)
    content, style = iu.load_image(content_path), iu.load_image(sandbox['style_path'])
    want, _ = gatys_oracle(content, style, sandbox['vggw'], np.zeros(3), 1e-2, 1e4, 1e8, steps=3)
    assert got.shape == content.shape
    assert rel_err(got, want) <= 1e-3
    assert (np.abs(got - content) > 1e-3).any(), 'three Adam steps of 0.02 must have moved the picture'
    saved = iu.load_image(os.path.join(str(sandbox['stylized']), '0_style_transfer_sample_1.jpg'))
    assert saved.shape == content.shape


def test_gatys_steps_bf16(cuda):
    """Gatys optimisation in the bf16-MFMA mode (fp32 master picture, fp32 Adam).

    What the mode is held to: the four loss terms of the first three steps against the fp64 oracle, and a
    30-step loss trajectory that ends within 10 % of the fp32 mode's.  It is NOT pixel-identical to the
    oracle: Adam(eps = 0.1) on gradients of order 1e2..1e4 moves every pixel by about +-lr per step whatever
    the gradient's size, so the ~5 % direction error of a gradient that went back through fourteen bf16
    layers flips the step of the pixels whose gradient is near zero (measured after 3 steps: PSNR 34.9 dB
    against the oracle picture for a movement of 0.03 rms, cosine of the movement 0.82)."""
    from stmi.trainer import GatysOptimizer
    rng = np.random.default_rng(0)
    content = smooth_image(rng, 64, 48)
    style = smooth_image(rng, 60, 44)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean = np.array([103.9, 116.8, 123.7], np.float32)
    sw, cw, tvw = 1e-2, 1e4, 1e8
    want, ref_losses = gatys_oracle(content, style, vggw, mean, sw, cw, tvw, steps=3)
    final = {}
    for dtype, use_graph in ((torch.bfloat16, False), (torch.bfloat16, True), (torch.float32, True)):
        opt = GatysOptimizer(torch.tensor(content), cuda, vggw, mean, sw, cw, tvw, dtype=dtype, use_graph=use_graph)
        opt.set_style_image(torch.tensor(style))
        for t in range(3):
            opt.step()
            got = opt.read_losses()
            # The picture starts AT the content picture, so for the first steps the content term is the
            # squared difference of two almost equal feature maps: in bf16 it sits on the rounding floor
            # of the features (measured: 30.5e3 against 25.5e3 at step 2).  Every term is therefore held
            # to 5 % of itself plus 3 % of the total.
            for a, b in zip(got, ref_losses[t]):
                assert abs(a - b) <= 5e-2 * abs(b) + 3e-2 * abs(ref_losses[t][0]), (dtype, use_graph, t, got, ref_losses[t])
        du = (opt.image.double().cpu() - t64(content)).reshape(-1)
        dr = (want - t64(content)).reshape(-1)
        cos = (du @ dr / (du.norm() * dr.norm())).item()
        p = psnr(opt.image, want)
        print('Gatys %s graph=%s: PSNR vs oracle picture %.2f dB, cosine of the 3-step movement %.4f'
              % (dtype, use_graph, p, cos))
        assert p >= (60.0 if dtype == torch.float32 else 33.0) and cos >= (0.999 if dtype == torch.float32 else 0.75)
        for _ in range(27):
            opt.step()
        final[(dtype, use_graph)] = (opt.read_losses()[0], opt.image.clone())
    f32 = final[(torch.float32, True)]
    print('Gatys fp32: total loss %.1f -> %.1f' % (ref_losses[0][0], f32[0]))
    # (with the reference's lr 0.02 / beta1 0.99 the first tens of steps overshoot: no decrease is asserted)
    for key in ((torch.bfloat16, False), (torch.bfloat16, True)):
        print('Gatys 30 steps %s: total loss %.1f (fp32 %.1f), PSNR vs fp32 picture %.2f dB'
              % (key, final[key][0], f32[0], psnr(final[key][1], f32[1])))
        assert abs(final[key][0] - f32[0]) <= 0.1 * f32[0], (final[key][0], f32[0])
    # graph replay against eager launches: the same kernels in the same order, and no kernel accumulates with
    # float atomics -- 30 steps end on the same picture and the same loss, bit for bit
    assert torch.equal(final[(torch.bfloat16, False)][1], final[(torch.bfloat16, True)][1])
    assert final[(torch.bfloat16, False)][0] == final[(torch.bfloat16, True)][0]


def test_style_swap_after_graph_capture(cuda):
    """A second set_style_image on a trainer whose step is already captured must be seen by the replayed
    graphs (the targets are rewritten in place)."""
    from stmi.trainer import StyleTrainer
    rng = np.random.default_rng(5)
    batch = rng.random((2, 32, 24, 3)).astype(np.float32)
    style_a, style_b = smooth_image(rng, 30, 22), rng.random((1, 30, 22, 3)).astype(np.float32)
    params = O.init_stylizer_params(seed=1)
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    mean = np.zeros(3, np.float32)
    out = {}
    for use_graph in (True, False):
        tr = StyleTrainer(2, 32, 24, cuda, vggw, mean, 1e-2, 1e4, 1e8, dtype=torch.float32, params=params,
                          use_graph=use_graph)
        tr.set_style_image(torch.tensor(style_a))
        tr.step(torch.tensor(batch, device=cuda))
        first = tr.read_losses()
        tr.set_style_image(torch.tensor(style_b))
        tr.step(torch.tensor(batch, device=cuda))
        out[use_graph] = (first, tr.read_losses())
    assert abs(out[True][1][1] - out[False][1][1]) <= 1e-4 * abs(out[False][1][1]), out
    assert abs(out[True][1][1] - out[True][0][1]) > 1e-2 * abs(out[True][0][1]), 'style loss must change with the style'


# ------------------------------------------------------------------ Django view call sequence
def test_django_view_call_sequence_three_services_concurrent_requests(cuda, tmp_path, monkeypatch):
    """What ml_services/style_net_service/views.py does with the drop-in: module imports (:15-19), three
    services constructed and started back to back without waiting (:23-30), then per request
    decode -> flip_BR -> pixel_to_decimal -> image_3d_to_4d -> run_on_image -> save_image (:46-54),
    here from eight request threads at once, two image sizes alternating."""
    import constants  # noqa: F401
    import full_graph_stylizer_network
    import graph_stylizer_network
    import model_utilities  # noqa: F401
    import image_utilities as iu
    from image_utilities import flip_BR, pixel_to_decimal, image_3d_to_4d, save_image
    assert graph_stylizer_network.model_prefix == 'STYLE_NET'
    monkeypatch.setenv('STMI_DTYPE', 'f32')
    monkeypatch.chdir(tmp_path)

    layout = {'rain': 'STYLE_NET_2019-09-24-20-41', 'waves': 'STYLE_NET_2019-09-24-15-54',
              'starry_night': 'STYLE_NET_2019-09-22-22-24'}
    params = {}
    for seed, (sub, name) in enumerate(layout.items(), 3):
        d = tmp_path / 'models' / sub / name
        d.mkdir(parents=True)
        params[sub] = O.init_stylizer_params(seed=seed)
        full_graph_stylizer_network.save_checkpoint(params[sub], str(d), name)
    models_dir = str(tmp_path / 'models')
    rain_service = full_graph_stylizer_network.StyleNetService(layout['rain'], os.path.join(models_dir, 'rain'))
    rain_service.start()
    wave_service = full_graph_stylizer_network.StyleNetService(layout['waves'], os.path.join(models_dir, 'waves'))
    wave_service.start()
    starry_service = full_graph_stylizer_network.StyleNetService(layout['starry_night'],
                                                                 os.path.join(models_dir, 'starry_night'))
    starry_service.start()
    services = {'starry': starry_service, 'wave': wave_service, 'rain': rain_service}
    which = {'starry': 'starry_night', 'wave': 'waves', 'rain': 'rain'}
    for s in services.values():
        s.wait_for_ready()

    rng = np.random.default_rng(0)
    uploads = []
    for i in range(8):
        h, w = ((40, 28), (24, 36))[i % 2]
        path = str(tmp_path / ('upload_%d.png' % i))
        save_image(rng.random((1, h, w, 3)).astype(np.float32), path)
        uploads.append((('starry', 'wave', 'rain')[i % 3], 'upload_%d.png' % i, open(path, 'rb').read()))

    results, errors = {}, []

    def index(style_name, upload_name, data):
        try:
            file_name = style_name + '_' + upload_name
            img = iu.imdecode_bgr(data)      # cv2.imdecode(np.fromstring(data, np.uint8), cv2.IMREAD_COLOR)
            img = flip_BR(img)
            img = pixel_to_decimal(img)
            img = image_3d_to_4d(img)
            omg = services[style_name].run_on_image(img)
            save_image(omg, file_name)
            results[file_name] = (style_name, img, omg)
        except BaseException as exc:  # surfaced in the main thread below
            errors.append(exc)

    for _round in range(2):
        threads = [threading.Thread(target=index, args=u) for u in uploads]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert not errors, errors
    assert len(results) == 8
    for file_name, (style_name, img, omg) in results.items():
        ref = O.stylizer_forward(t64(img), [t64(p) for p in params[which[style_name]]])
        assert omg.dtype == np.float32 and rel_err(omg, ref) <= 1e-3, file_name
        saved = iu.load_image(file_name)
        assert np.abs(saved - ref.numpy()).max() <= 1.5 / 255
    # both image sizes stayed resident in every service that saw them
    assert all(len(s._nets) == 2 for s in services.values())
    for s in services.values():
        s.close()


def test_service_loads_a_tensor_bundle_checkpoint(cuda, tmp_path, monkeypatch):
    """saved_<name>-0.index + .data-00000-of-00001 (the format the reference's Saver writes,
    full_graph_stylizer_network.py:183, :231) served like an .npz."""
    import full_graph_stylizer_network as fg
    import tf_bundle
    monkeypatch.setenv('STMI_DTYPE', 'f32')
    name = 'STYLE_NET_2019-09-22-22-24'
    d = tmp_path / name
    d.mkdir()
    params = O.init_stylizer_params(seed=9)
    tensors = {fg._var_key(i): np.asarray(p, np.float32) for i, p in enumerate(params)}
    # Adam slots ride along in a real checkpoint and must be ignored
    tensors['style_network/Variable/Adam'] = np.zeros_like(params[0])
    tensors['beta1_power'] = np.float32(0.9)
    tf_bundle.write_bundle(str(d / ('saved_%s-0' % name)), tensors)
    (d / 'checkpoint').write_text('model_checkpoint_path: "saved_%s-0"\n' % name)
    svc = fg.StyleNetService(name, str(tmp_path))
    svc.start()
    svc.wait_for_ready()
    x = np.random.default_rng(0).random((1, 24, 20, 3)).astype(np.float32)
    assert rel_err(svc.run_on_image(x), O.stylizer_forward(t64(x), [t64(p) for p in params])) <= 1e-3
    svc.close()


# ------------------------------------------------------------------ full-size inference
def test_inference_1080p_batch16_bf16(cuda):
    """The batched high-resolution configuration at full size: 16 frames of 1920 x 1080, bf16.
    Instance norm is per sample, so a frame stylized inside the batch must equal the same frame stylized
    alone, bit for bit; one frame is compared with the CPU oracle (fp32: fp64 at this size takes minutes)."""
    from stmi.stylizer import StylizerNet
    params = O.init_stylizer_params(seed=1)
    g = torch.Generator(device='cpu').manual_seed(7)
    frames = torch.rand((16, 1080, 1920, 3), generator=g, dtype=torch.float32)
    net = StylizerNet(16, 1080, 1920, cuda, torch.bfloat16, params=params)
    out = net.forward(frames.to(cuda))
    torch.cuda.synchronize()
    assert out.shape == (16, 1080, 1920, 3) and out.dtype == torch.float32
    assert torch.isfinite(out).all()
    assert out.min().item() > 0.0 and out.max().item() < 1.0
    batch_out = out[[0, 7, 15]].cpu()
    del net, out
    torch.cuda.empty_cache()
    single = StylizerNet(1, 1080, 1920, cuda, torch.bfloat16, params=params)
    for j, k in enumerate((0, 7, 15)):
        alone = single.forward(frames[k:k + 1].to(cuda)).cpu()
        assert torch.equal(alone[0], batch_out[j]), 'frame %d differs between batch-16 and batch-1 runs' % k
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        ref = O.stylizer_forward(frames[7:8].float(), [torch.tensor(p, dtype=torch.float32) for p in params])
    assert psnr(batch_out[1:2], ref) >= 40.0


def test_absorbed_norms_equal_the_apply_passes(cuda, monkeypatch):
    """At inference sizes the second convolution of a residual block and the two down-sampling convolutions absorb
    the norm + ReLU in front of them (full_graph_stylizer_network.py:47-57 without the normalised tensors).  Same picture as the route
    with the apply passes, within bf16; a non-positive scale in such a norm switches the block back."""
    from stmi import tuning
    from stmi.stylizer import StylizerNet
    params = O.init_stylizer_params(seed=3)
    g = torch.Generator(device='cpu').manual_seed(11)
    frames = torch.rand((2, 1024, 1536, 3), generator=g, dtype=torch.float32).to(cuda)
    net = StylizerNet(2, 1024, 1536, cuda, torch.bfloat16, params=params)
    absorbing = [i for i, b in enumerate(net.blocks) if b.absorb_ws is not None]
    # the two layers in front of the down-sampling convolutions and the first layer of each residual block
    assert absorbing == [0, 1, 3, 5, 7, 9, 11, 13, 14] and all(net.blocks[i].role == 'a' for i in absorbing[2:7])
    out = net.forward(frames).clone()
    monkeypatch.setattr(tuning, 'ABSORB_NORM', False)
    plain = StylizerNet(2, 1024, 1536, cuda, torch.bfloat16, params=params)
    assert all(b.absorb_ws is None for b in plain.blocks)
    ref = plain.forward(frames).clone()
    assert psnr(out.cpu(), ref.cpu()) >= 45.0
    monkeypatch.setattr(tuning, 'ABSORB_NORM', True)
    flipped = [np.array(p, copy=True) for p in params]
    scale_index = 3 * absorbing[2] + 2                 # W, shift, scale per block, in creation order
    assert flipped[scale_index].shape == (128,)
    flipped[scale_index][5] = -0.5
    net.load_params(flipped)
    net.repack()
    assert net.blocks[absorbing[2]].absorb_ws is None and net.blocks[absorbing[3]].absorb_ws is not None
    plain.load_params(flipped)
    plain.repack()
    assert psnr(net.forward(frames).cpu(), plain.forward(frames).cpu()) >= 45.0


def test_sharded_frames_against_the_oracle(cuda):
    """stylize_frames_sharded (frames split over ranks, no collective; pinned double-buffered staging) against the
    ORACLE, for a frame count that leaves a ragged last chunk: float32 frames in fp32 mode to 1e-3; 8-bit frames
    (as a decoder leaves them) come back as 8-bit frames equal to the oracle's (stylized * 255) truncated
    (image_utilities.py:195-196) up to one count at a truncation boundary; bf16 mode to 40 dB."""
    from stmi import dist as sd
    params = O.init_stylizer_params(seed=2)
    rng = np.random.default_rng(3)
    frames = rng.random((7, 40, 56, 3)).astype(np.float32)
    P = [t64(p) for p in params]
    want = O.stylizer_forward(t64(frames), P)
    idx, got = sd.stylize_frames_sharded(frames, params, dtype=torch.float32, batch=3)
    assert list(idx) == list(range(7)) and got.shape == frames.shape and got.dtype == np.float32
    assert rel_err(got, want) <= 1e-3
    # 8-bit frames: x / 255 happens in the first layer, * 255 and truncation in the last one
    frames_u8 = rng.integers(0, 256, size=(7, 40, 56, 3), dtype=np.uint8)
    want_f = O.stylizer_forward(t64(frames_u8) / 255.0, P)
    want_u8 = (want_f * 255.0).floor().clamp(0, 255)
    idx, got_u8 = sd.stylize_frames_sharded(frames_u8, params, dtype=torch.float32, batch=3)
    assert got_u8.dtype == np.uint8 and got_u8.shape == frames_u8.shape
    diff = (t64(got_u8.astype(np.float64)) - want_u8).abs()
    assert diff.max().item() <= 1.0 and (diff > 0).double().mean().item() <= 2e-3, (diff.max().item(), (diff > 0).double().mean().item())
    idx, got_b16 = sd.stylize_frames_sharded(frames_u8, params, dtype=torch.bfloat16, batch=4)
    assert psnr(t64(got_b16.astype(np.float64)) / 255.0, want_f) >= 40.0
    # the shards of a 3-rank job tile the frame range
    parts = [sd.shard_frames(7, r, 3) for r in range(3)]
    assert [len(p) for p in parts] == [3, 2, 2] and sum((list(p) for p in parts), []) == list(range(7))


def test_eight_bit_frames_equal_the_float_path(cuda):
    """forward_u8 on 8-bit frames = forward on the same frames / 255, then (x * 255) truncated: the two first-layer
    kernels see the same values (k / 255 rounded to fp32 either way), so the stylized pictures agree to a count
    where the float result sits on a truncation boundary.  Odd sizes: tiles clipped at both borders; the stylized
    picture is the next multiple of 4 in size (two stride-2 layers, two exact doublings)."""
    from stmi.stylizer import StylizerNet
    params = O.init_stylizer_params(seed=4)
    rng = np.random.default_rng(9)
    for (n, h, w), dtype in (((2, 37, 45), torch.float32), ((1, 50, 33), torch.bfloat16)):
        x8 = torch.tensor(rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8), device=cuda)
        net = StylizerNet(n, h, w, cuda, dtype, params=params)
        y8 = net.forward_u8(x8).clone()
        assert tuple(y8.shape) == (n, -(-h // 4) * 4, -(-w // 4) * 4, 3)
        yf = net.forward(x8.float() / 255.0)
        want = (yf * 255.0).floor().clamp(0, 255)
        diff = (y8.float() - want).abs()
        assert diff.max().item() <= 1.0 and (diff > 0).float().mean().item() <= 1e-2
