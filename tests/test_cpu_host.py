"""CPU tests of the host side: the C-ABI library loads and exports every declared
symbol, the drop-in modules keep the reference's names/defaults/artefacts, the .mat
loader indexes the file like the reference, and the product path refuses to run
without a GPU instead of falling back."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI
def test_library_exports_every_declared_symbol():
    from stmi import lib
    handle = lib.load()
    header = open(os.path.join(ROOT, 'include', 'stmi.h')).read()
    declared = set(re.findall(r'\b(stmi_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(handle, name), name
        assert name in lib.SIGNATURES, 'no ctypes prototype for ' + name
    assert set(lib.SIGNATURES) <= declared
    assert handle.stmi_version() >= 100


def test_every_kernel_source_is_built_and_product_code_stays_off_the_oracle():
    """The Makefile that build() drives lists every .hip file of csrc/ (a kernel file left out would only show up
    as an unresolved symbol on the GPU box), and nothing under the package imports the oracle."""
    import glob
    import re
    csrc = os.path.join(ROOT, 'style-transfer_amd', 'csrc')
    srcs = re.search(r'^SRCS\s*:=\s*(.*)$', open(os.path.join(csrc, 'Makefile')).read(), re.M).group(1).split()
    assert sorted(srcs) == sorted(os.path.basename(f) for f in glob.glob(os.path.join(csrc, '*.hip')))
    for path in glob.glob(os.path.join(ROOT, 'style-transfer_amd', '**', '*.py'), recursive=True):
        text = open(path).read()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), path


def test_host_only_queries_need_no_gpu():
    from stmi import lib, ops
    h = lib.load()
    # packed sizes: [taps][rows padded to the tile height][K padded to 32]
    assert h.stmi_conv_packed_elems(3, 3, 128, 128, lib.PACK_FWD) == 9 * 128 * 128
    assert h.stmi_conv_packed_elems(3, 3, 64, 128, lib.PACK_DGRAD) == 9 * 64 * 128
    assert h.stmi_conv_packed_elems(9, 9, 3, 32, lib.PACK_FWD) == 9 * 32 * 32        # row taps: 27 -> 32
    assert h.stmi_conv_packed_elems(9, 9, 32, 3, lib.PACK_FWD) == 81 * 16 * 32
    assert h.stmi_conv_packed_elems(9, 9, 32, 3, lib.PACK_DGRAD) == 9 * 32 * 32      # row taps on the 3 outputs
    assert h.stmi_gram_packed_elems(4, 64) == 4 * 64 * 64
    assert h.stmi_instance_norm_ws_floats(4, 256, 256, 32) == 4 * 256 * 2 * 32 + 4 * 32 * 2
    assert ops.last_conv_kernel() == ''     # nothing launched by this thread yet


def test_argument_errors_are_reported_not_crashed():
    from stmi import lib
    h = lib.load()
    assert h.stmi_conv2d_nhwc_fwd(None, None, None, None, None, None, 0, 1, 8, 8, 8, 8, 3, 3, 1, 0, lib.F32, None) == -1
    assert b'null pointer' in h.stmi_last_error()
    assert h.stmi_conv2d_nhwc_fwd(1, 1, None, 1, None, None, 0, 1, 8, 8, 8, 8, 3, 3, 1, 0, 7, None) == -1
    assert b'dtype' in h.stmi_last_error()
    assert h.stmi_adam_tf1(1, 1, 1, 1, 10, 1e-3, .9, .999, 1e-8, 1.0, None, 0, None) == -1
    with pytest.raises(lib.StmiError, match='bad arguments'):
        lib.call('stmi_gram_nhwc_fwd', None, None, None, 1, 4, 4, 8, lib.F32, None)


def test_product_path_refuses_cpu_tensors():
    from stmi import lib, ops
    with pytest.raises(lib.StmiError, match='no CPU fallback'):
        ops.gram_fwd(torch.zeros((1, 4, 4, 8)))
    with pytest.raises(lib.StmiError, match='no CPU fallback'):
        ops.conv2d_fwd(torch.zeros((1, 4, 4, 8)), torch.zeros(9 * 16 * 32), (3, 3, 8, 16), 1)


# ------------------------------------------------------------- drop-in API
def test_constants_and_style_table():
    import constants
    import model_utilities as mu
    for name in ('TENSORBOARD_DIR', 'MODELS_DIR', 'STYLES_DIR', 'CONTENT_DIR', 'TEST_DIR', 'STYLIZED_IMAGES_DIR'):
        assert os.path.isdir(getattr(constants, name)), name     # created at import, like the reference
    d = mu.StyleTransfer.STYLE_CONFIG_DICT
    assert len(d) == 25
    c = d['starry_night_transfer']
    assert (c.style_weight, c.content_weight, c.total_variation_weight) == (1e-2, 1e4, 1e8)
    assert c.style_path.endswith('starry_night.jpg')
    h4 = d['heiro_4']
    assert (h4.style_weight, h4.content_weight, h4.total_variation_weight) == (1.5e-3, 1e4, 1e8)
    assert (d['starry_night_content'].style_weight, d['starry_night_content'].total_variation_weight) == (0, 0)
    assert repr(c).startswith("<StyleConfig {'style_weight': 0.01, 'content_weight': 10000.0, 'var_weight': 100000000.0")
    st = mu.StyleTransfer
    assert (st.STYLE_WEIGHT, st.TOTAL_VARIATION_WEIGHT, st.CONTENT_WEIGHT) == (1e-2, 1e-4, 1e8)
    assert st.VGG_STYLE_TARGET_LAYER_NAMES_ALT == ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
    assert st.VGG_CONTENT_TARGET_LAYER_NAMES_ALT == ['conv5_2']
    assert st.VGG_STYLE_TARGET_LAYER_NAMES_ALT_ACT[0] == 'relu1_1' and st.VGG_CONTENT_TARGET_LAYER_NAMES_ALT_ACT == ['relu5_2']
    assert st.VGG_STYLE_TARGET_LAYER_NAMES[0] == 'block1_conv1' and st.VGG_CONTENT_TARGET_LAYER_NAMES == ['block5_conv2']


def test_layer_config():
    import model_utilities as mu
    c = mu.LayerConfig(64, 3, 2)
    assert (c.padding, c.activation, c.use_batch_norm, c.use_instance_norm) == ('same', mu.nn.relu, False, True)
    d = c.copy()
    d.activation = None
    assert c.activation is mu.nn.relu
    assert repr(c).startswith('<LayerConfig 64, 3x3, 2 stride, same pad, <function') and repr(c).endswith('False bn, True in>')


def test_module_constants_and_cli_dispatch(monkeypatch, tmp_path):
    import full_graph_stylizer_network as fg
    assert (fg.LEARNING_RATE, fg.EPOCHS, fg.BATCH_SIZE, fg.TOTAL_IMAGES) == (1e-3, 80, 4, 1000)
    assert fg.model_prefix == 'STYLE_NET' and fg.TRAIN_IMG_DIMS == [None, 256, 256, 3]
    assert fg.CONTENT_IMAGE_1_2.endswith('antelope_256.jpg')
    calls = []
    monkeypatch.setattr(fg, 'run_on_image', lambda *a, **k: calls.append(('run', a, k)))
    monkeypatch.setattr(fg, 'train_style_network', lambda *a, **k: calls.append(('train', a, k)))
    src = tmp_path / 'in.jpg'
    src.write_bytes(b'x')
    fg.main(['prog', str(src), 'out.jpg'])
    assert calls[-1] == ('run', (str(src), 'out.jpg'), {})
    fg.main(['prog', str(tmp_path / 'missing.jpg'), 'out.jpg'])          # silently does nothing
    assert len(calls) == 1
    base = tmp_path / 'models'
    (base / 'M1').mkdir(parents=True)
    fg.main(['prog', 'M1', str(base), str(src), 'out.jpg'])
    assert calls[-1][0] == 'run' and calls[-1][2] == {'model_path': str(base / 'M1'), 'model_dir': str(base)}
    monkeypatch.setattr(fg.constants, 'MODELS_DIR', str(base))
    fg.main(['prog', 'M1', str(src), 'out.jpg'])
    assert calls[-1] == ('run', (str(src), 'out.jpg'), {'model_path': 'M1'})
    fg.main(['prog'])
    assert calls[-1] == ('train', ('heiro_4',), {})


def test_checkpoint_round_trip_and_service_load(tmp_path, monkeypatch):
    import full_graph_stylizer_network as fg
    from oracle import tf_ops as O
    params = O.init_stylizer_params(seed=3)
    mdir = tmp_path / 'STYLE_NET_2019-09-22-22-24'
    mdir.mkdir()
    path = fg.save_checkpoint(params, str(mdir), 'STYLE_NET_2019-09-22-22-24')
    assert os.path.basename(path) == 'saved_STYLE_NET_2019-09-22-22-24-0.npz'
    assert fg.latest_checkpoint(str(mdir)) == path
    back = fg.load_checkpoint(path)
    assert len(back) == 48 and all(np.array_equal(a, b) for a, b in zip(params, back))
    with np.load(path) as z:
        assert 'style_network/Variable' in z.files and 'style_network/Variable_47' in z.files
    import model_utilities as mu
    (tmp_path / 'STYLE_NET_2019-09-24-20-41').mkdir()
    (tmp_path / 'unrelated').mkdir()
    assert mu.get_most_recent_model_name(str(tmp_path), 'STYLE_NET') == 'STYLE_NET_2019-09-24-20-41'
    # the service thread only reads the weight file; without a GPU the first run_on_image fails loudly
    if not torch.cuda.is_available():
        svc = fg.StyleNetService('STYLE_NET_2019-09-22-22-24', str(tmp_path))
        svc.start()
        svc.join()
        assert svc.loaded is True and svc.error is None
        with pytest.raises(RuntimeError, match='no CPU path'):
            svc.run_on_image(np.zeros((1, 8, 8, 3), np.float32))


def test_image_utilities_round_trip(tmp_path):
    import image_utilities as iu
    rng = np.random.default_rng(0)
    img = rng.random((1, 16, 12, 3)).astype(np.float32)
    p = str(tmp_path / 'a.png')
    iu.save_image(img, p)
    back = iu.load_image(p)
    assert back.shape == (1, 16, 12, 3) and back.dtype == np.float32
    # save truncates to uint8: (x*255).astype(uint8)
    assert np.array_equal((back * 255).round().astype(np.uint8), (img * 255.).astype(np.uint8))
    assert iu.image_4d_to_3d(img).shape == (16, 12, 3) and iu.image_3d_to_4d(img[0]).shape == (1, 16, 12, 3)
    assert np.array_equal(iu.flip_BR(iu.flip_RB(img[0])), img[0])
    assert iu.sort_numerical(['m_10_x', 'm_9_x', 'm_100']) == ['m_9_x', 'm_10_x', 'm_100']
    assert iu.can_clip_image_to_dims(np.zeros((300, 300, 3)), dim=(256, 256))
    assert not iu.can_clip_image_to_dims(np.zeros((256, 300, 3)), dim=(256, 256))
    assert iu.random_clip_image_to_dim(np.zeros((300, 280, 3)), dim=(256, 256)).shape == (256, 256, 3)
    d = tmp_path / 'coco'
    d.mkdir()
    for i in range(5):
        iu.save_image(rng.random((8, 8, 3)).astype(np.float32), str(d / ('%d.jpg' % i)))
    (d / 'notes.txt').write_text('x')
    mgr = iu.PreprocessedCocoDatasetManager(target_dim=(8, 8), num_images=2, directory=str(d))
    assert len(mgr.get_images()) == 3          # num_images + 1, as the reference loads
    mgr.shuffle_loaded_images()
    assert len(mgr.get_images()) == 3 and mgr.get_images()[0].shape == (8, 8, 3)


# -------------------------------------------------------------- .mat loader
def test_vgg_mat_loader_indexes_like_the_reference(tmp_path):
    import scipy.io
    import vgg
    from oracle import tf_ops as O
    w = O.synthetic_vgg_weights(seed=7, width_div=16, bias_scale=0.5)
    mean_img = np.random.default_rng(1).random((224, 224, 3)).astype(np.float32) * 255
    path = str(tmp_path / 'imagenet-vgg-verydeep-19.mat')
    vgg.save_vgg_mat(path, w, mean_img)
    # the reference's own indexing expressions (vgg.py:25-28, :43-45)
    data = scipy.io.loadmat(path)
    assert np.allclose(np.mean(data['normalization'][0][0][0], axis=(0, 1)), mean_img.mean(axis=(0, 1)))
    kernels, bias = data['layers'][0][vgg.VGG_LAYER_NAMES.index('conv3_2')][0][0][0][0]
    assert np.array_equal(np.transpose(kernels, (1, 0, 2, 3)), w['conv3_2'][0])
    assert np.array_equal(bias.reshape(-1), w['conv3_2'][1])
    weights, mean_pixel = vgg.load_vgg_mat(path)
    assert set(weights) == set(w) and len(weights) == 16
    for k in w:
        assert np.array_equal(weights[k][0], w[k][0]) and np.array_equal(weights[k][1], w[k][1])
    assert np.allclose(mean_pixel, mean_img.mean(axis=(0, 1)))


def test_vgg_mean_is_read_at_call_time(monkeypatch):
    """The module global starts at [0,0,0] and is only replaced by a builder (vgg.py:9, :26-27);
    vgg_preprocess_input must read it when called."""
    import vgg
    from stmi import ops
    assert list(vgg.vgg_preprocessing_mean) == [0, 0, 0] or vgg.vgg_preprocessing_mean.shape == (3,)
    seen = []
    monkeypatch.setattr(ops, 'scale_shift', lambda x, mean, scale, dtype, out=None: seen.append((mean.tolist(), scale)) or x)
    monkeypatch.setattr(torch.Tensor, 'cuda', lambda self, *a, **k: self)
    monkeypatch.setattr(torch.Tensor, 'is_cuda', property(lambda self: True))
    x = torch.zeros((1, 2, 2, 3))
    monkeypatch.setattr(vgg, 'vgg_preprocessing_mean', np.array([0, 0, 0]))
    vgg.vgg_preprocess_input(x)
    monkeypatch.setattr(vgg, 'vgg_preprocessing_mean', np.array([103.9, 116.8, 123.7]))
    vgg.vgg_preprocess_input(x)
    assert seen[0] == ([0.0, 0.0, 0.0], 127.5)
    assert np.allclose(seen[1][0], [103.9, 116.8, 123.7]) and seen[1][1] == 127.5


def test_stylizer_executor_layout():
    from stmi import stylizer
    shapes = stylizer.param_shapes()
    assert len(shapes) == 48 and sum(int(np.prod(s)) for s in shapes) == 1677638
    assert shapes[39] == (3, 3, 64, 128) and shapes[42] == (3, 3, 32, 64)      # deconv: [k,k,out,in]
    a, b = stylizer.init_params(seed=1), stylizer.init_params(seed=1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    from oracle import tf_ops as O
    assert all(np.array_equal(x, y) for x, y in zip(a, O.init_stylizer_params(seed=1)))
    from stmi.trainer import BWD_SEGMENTS
    covered = [i for hi, lo in BWD_SEGMENTS for i in range(hi, lo - 1, -1)]
    assert covered == list(range(15, -1, -1))


# ------------------------------------------------------------------ service error path, Django imports
def test_service_reports_a_failed_load(tmp_path):
    import full_graph_stylizer_network as fg
    svc = fg.StyleNetService('STYLE_NET_missing', str(tmp_path))
    svc.start()
    with pytest.raises(RuntimeError, match='failed to load'):
        svc.wait_for_ready()
    assert svc.loaded is False and isinstance(svc.error, FileNotFoundError)
    with pytest.raises(RuntimeError, match='not loaded'):
        svc.run_on_image(np.zeros((1, 8, 8, 3), np.float32))


def test_django_view_imports_resolve_in_the_drop_in():
    """Everything ml_services/style_net_service/views.py imports from src/ (views.py:15-19) and every
    helper it calls after `from image_utilities import *` (:48-54) exists in the drop-in directory.
    When the reference tree is present its view is parsed (as text, never executed) for the list."""
    import ast
    import importlib
    wanted_modules = ['constants', 'full_graph_stylizer_network', 'graph_stylizer_network', 'model_utilities',
                      'image_utilities']
    called = ['flip_BR', 'pixel_to_decimal', 'image_3d_to_4d', 'save_image']
    view = '/root/reference/src/ml_services/style_net_service/views.py'
    if os.path.exists(view):
        tree = ast.parse(open(view).read())
        local = {n.split('.')[0] for n in os.listdir(os.path.join(ROOT, 'style-transfer_amd', 'src'))}
        mods = set()
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                mods |= {a.name for a in node.names}
            elif isinstance(node, ast.ImportFrom) and node.module:
                mods.add(node.module)
        third_party = {m for m in mods if m.split('.')[0] in ('cv2', 'django', 'sys', 'os', 'numpy')}
        assert sorted(mods - third_party) == sorted(wanted_modules)
        assert set(wanted_modules) <= local
    mods = {m: importlib.import_module(m) for m in wanted_modules}
    assert mods['graph_stylizer_network'].model_prefix == mods['full_graph_stylizer_network'].model_prefix == 'STYLE_NET'
    for name in called:
        assert callable(getattr(mods['image_utilities'], name))
    svc_cls = mods['full_graph_stylizer_network'].StyleNetService
    assert mods['graph_stylizer_network'].StyleNetService is svc_cls
    # constructing and starting services needs no GPU (the view does it at import, views.py:23-30)
    assert hasattr(mods['constants'], 'MODELS_DIR')


def test_service_loads_without_a_gpu_and_says_so_at_first_use(tmp_path):
    import full_graph_stylizer_network as fg
    from stmi.stylizer import init_params
    name = 'STYLE_NET_2019-09-24-15-54'
    (tmp_path / name).mkdir()
    fg.save_checkpoint(init_params(seed=1), str(tmp_path / name), name)
    svc = fg.StyleNetService(name, str(tmp_path))
    svc.start()
    svc.wait_for_ready()
    assert svc.loaded and svc.error is None
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match='no MI355X visible'):
            svc.run_on_image(np.zeros((1, 8, 8, 3), np.float32))
    svc.close()


def test_frame_codec_pool_round_trip():
    """Host side of the frame service: JPEG decode / encode on a thread pool, results in input order."""
    from stmi.dist import FrameCodecPool
    yy, xx = np.mgrid[0:48, 0:64].astype(np.float32)
    frames = [np.stack([127 + 100 * np.sin(0.05 * (k + 1) * yy + 0.08 * xx + c) for c in range(3)], -1).astype(np.uint8)
              for k in range(5)]
    pool = FrameCodecPool(workers=3)
    blobs = pool.encode(frames)
    back = pool.decode(blobs)
    pool.close()
    assert len(back) == 5
    for a, b in zip(frames, back):
        assert b.shape == a.shape and b.dtype == np.uint8
        assert np.abs(a.astype(np.int32) - b.astype(np.int32)).mean() <= 3.0     # JPEG, quality 95
