"""Data-parallel training THROUGH THE ENTRY FUNCTION (train_style_network, reference
full_graph_stylizer_network.py:85-232) with two ranks sharing the one GPU of the test box.

The ranks are started the way torchrun starts them (WORLD_SIZE / RANK / LOCAL_RANK / MASTER_* in the
environment); STMI_DIST_BACKEND=gloo replaces RCCL, which needs one device per rank.  Checked: ten pictures, batch
4 per rank -> two steps per epoch, the second one padded from the start of the list; rank 0 alone leaves
config.json / losses.jsonl / the sample / the checkpoint; both ranks end with identical weights; those weights are
what TF1 Adam gives on the MEAN of the two ranks' gradients, step by step."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, N_IMAGES = 32, 24, 10
WEIGHTS = (1e-2, 1e4, 1e8)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class FixedOrder:
    """Dataset manager with a deterministic 'shuffle' (the list rotates by three), so that the parent process can
    replay the epochs."""

    def __init__(self, images):
        self.images = list(images)

    def shuffle_loaded_images(self):
        self.images = self.images[3:] + self.images[:3]

    def get_images(self):
        return self.images


def _images():
    rng = np.random.default_rng(42)
    return [rng.random((H, W, 3)).astype(np.float32) for _ in range(N_IMAGES)]


def _configure(tmp, patch=None):
    """Points the drop-in's module constants at the sandbox the parent prepared (the worker processes are
    spawned: they import everything afresh; the parent passes pytest's monkeypatch so that nothing leaks into
    other tests)."""
    for p in (ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'style-transfer_amd', 'src')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import constants
    import full_graph_stylizer_network as fg
    import model_utilities as mu
    import vgg
    mat = os.path.join(tmp, 'vgg.mat')
    put = setattr if patch is None else patch.setattr
    put(vgg, 'VGG_PATH', mat)
    put(vgg, 'VGG_PATH_REPLICATE', mat)
    put(vgg, 'vgg_preprocessing_mean', np.array([0, 0, 0]))
    put(constants, 'MODELS_DIR', os.path.join(tmp, 'models'))
    put(fg, 'TRAIN_IMG_DIMS', [None, H, W, 3])
    cfg = mu.StyleConfig(os.path.join(tmp, 'style.jpg'), *WEIGHTS)
    if patch is None:
        mu.StyleTransfer.STYLE_CONFIG_DICT['unit_test_style'] = cfg
    else:
        patch.setitem(mu.StyleTransfer.STYLE_CONFIG_DICT, 'unit_test_style', cfg)
    return fg


def _worker(rank, world, port, tmp, q):
    import faulthandler
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ.update(WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), STMI_DIST_BACKEND='gloo', STMI_DTYPE='f32')
    try:
        fg = _configure(tmp)
        name = fg.train_style_network('unit_test_style', os.path.join(tmp, 'content.jpg'),
                                      dataset_manager=FixedOrder(_images()), epochs=2, log=lambda *a: None)
        q.put((rank, name))
    except Exception:
        import traceback
        q.put((rank, 'FAILED ' + traceback.format_exc()))
    finally:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.destroy_process_group()


def test_train_style_network_two_ranks(cuda, tmp_path, monkeypatch):
    tmp = str(tmp_path)
    fg = _configure(tmp, monkeypatch)
    import image_utilities as iu
    import vgg
    from oracle import tf_ops as O
    from stmi.dist import epoch_batches
    os.mkdir(os.path.join(tmp, 'models'))
    vggw = O.synthetic_vgg_weights(seed=7, width_div=8, bias_scale=0.1)
    rng = np.random.default_rng(3)
    vgg.save_vgg_mat(os.path.join(tmp, 'vgg.mat'), vggw, rng.random((224, 224, 3)).astype(np.float32) * 255)
    yy, xx = np.mgrid[0:30, 0:22].astype(np.float32)
    style = (0.5 + 0.4 * np.sin(0.1 * yy[..., None] + 0.07 * xx[..., None] + np.arange(3)))[None].astype(np.float32)
    iu.save_image(style, os.path.join(tmp, 'style.jpg'))
    iu.save_image(style[:, :H, :W], os.path.join(tmp, 'content.jpg'))

    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=170) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    for r in (0, 1):
        assert not got[r].startswith('FAILED'), got[r]
    assert got[0] == got[1]                                   # one model name for all ranks
    model_dir = os.path.join(tmp, 'models', got[0])
    assert sorted(os.listdir(os.path.join(tmp, 'models'))) == [got[0]]          # only rank 0 made a directory
    names = sorted(os.listdir(model_dir))
    assert names == sorted(['config.json', 'losses.jsonl', '0_style_transfer_sample.jpg', '1_style_transfer_sample.jpg',
                            'checkpoint', 'saved_%s-0.npz' % got[0]]), names
    assert json.load(open(os.path.join(model_dir, 'config.json')))['batch_size'] == 4
    assert len(open(os.path.join(model_dir, 'losses.jsonl')).read().splitlines()) == 4   # 2 epochs x 2 steps
    trained = fg.load_checkpoint(fg.latest_checkpoint(model_dir))

    # the same four steps in this process: the two ranks' gradients from single-rank executors, Adam on their mean
    from stmi import ops
    from stmi.stylizer import init_params
    from stmi.trainer import StyleTrainer
    style_loaded = iu.load_image(os.path.join(tmp, 'style.jpg'))
    trainers = [StyleTrainer(4, H, W, cuda, vggw, np.zeros(3, np.float32), *WEIGHTS, dtype=torch.float32,
                             params=init_params(seed=1), use_graph=False) for _ in range(2)]
    for t in trainers:
        t.set_style_image(torch.from_numpy(np.ascontiguousarray(style_loaded)))
    flat = trainers[0].net.flat.clone()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    dm = FixedOrder(_images())
    step_no = 0
    for epoch in range(2):
        dm.shuffle_loaded_images()
        images = dm.get_images()
        plans = [epoch_batches(len(images), 4, 2, r) for r in range(2)]
        assert len(plans[0]) == 2 and plans[1][1] == [2, 3, 4, 5]        # the padded tail
        for s in range(len(plans[0])):
            grads = []
            for r, t in enumerate(trainers):
                t.net.flat.copy_(flat)
                t.net.repack()
                t.batch_in.copy_(torch.from_numpy(np.array([images[i] for i in plans[r][s]], dtype=np.float32)))
                t.net.backward(t._front())
                grads.append(t.net.flat_grad.clone())
            step_no += 1
            ops.adam_tf1(flat, (grads[0] + grads[1]) * 0.5, m, v, 1e-3, 0.9, 0.999, 1e-8, t=step_no)
    torch.cuda.synchronize()
    want = flat.cpu().numpy()
    have = np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in trained])
    start = np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in init_params(seed=1)])
    cos = float(np.dot(want - start, have - start) / (np.linalg.norm(want - start) * np.linalg.norm(have - start)))
    assert cos > 0.9999, cos
    assert np.abs(have - want).max() <= 4.2e-3      # at most the four Adam steps of a sign flip at g ~ 0
