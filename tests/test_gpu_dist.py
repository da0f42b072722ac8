"""Data-parallel training step with two ranks sharing the one GPU of the test box.

RCCL needs one device per rank, so the two processes exchange gradients over gloo (which
stages CUDA tensors through the host); everything else -- bucket ranges, per-segment HIP
graphs, the order of all-reduces against the backward, Adam's 1/world -- is the code the
RCCL job runs.  Checked: both ranks end with the same weights, and those equal one TF1-Adam
step on the MEAN of the two single-rank gradients."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W = 32
STYLE = (1e3, 1.0, 10.0)   # style, content, tv weights


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _setup():
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
    from oracle import tf_ops as O
    from stmi.stylizer import init_params
    vggw = O.synthetic_vgg_weights(seed=5, width_div=8)
    params = init_params(seed=3)
    style = np.random.default_rng(7).random((1, H, W, 3), dtype=np.float32)
    batches = [np.random.default_rng(20 + r).random((1, H, W, 3), dtype=np.float32) for r in range(2)]
    return vggw, params, style, batches


def _worker(rank, world, port, use_graph, q):
    import faulthandler
    import torch.distributed as dist
    faulthandler.dump_traceback_later(90, exit=True)   # a stuck rank ends itself and names where
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        vggw, params, style, batches = _setup()
        from stmi.trainer import StyleTrainer
        dev = torch.device('cuda:0')
        tr = StyleTrainer(1, H, W, dev, vggw, np.zeros(3, np.float32), *STYLE, dtype=torch.float32, params=params,
                          world_size=world, use_graph=use_graph)
        tr.set_style_image(torch.from_numpy(style))
        tr.step(torch.from_numpy(batches[rank]).to(dev))
        tr.step(torch.from_numpy(batches[rank]).to(dev))
        torch.cuda.synchronize()
        q.put((rank, tr.net.flat.cpu().numpy()))
    except Exception:
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('use_graph', [False, True], ids=['eager', 'graph'])
def test_two_rank_step_equals_adam_on_the_mean_gradient(cuda, use_graph):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    for r in (0, 1):
        assert not isinstance(got[r], str), got[r]
    assert np.array_equal(got[0], got[1])                    # replicas stay in lock-step

    # the same two steps in one process: per-rank gradients from single-rank trainers, Adam on their mean
    vggw, params, style, batches = _setup()
    from stmi import ops
    from stmi.trainer import StyleTrainer
    trainers = [StyleTrainer(1, H, W, cuda, vggw, np.zeros(3, np.float32), *STYLE, dtype=torch.float32, params=params,
                             use_graph=False) for _ in range(2)]
    for t in trainers:
        t.set_style_image(torch.from_numpy(style))
    flat = trainers[0].net.flat.clone()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    for step in (1, 2):
        grads = []
        for t, b in zip(trainers, batches):
            t.net.flat.copy_(flat)
            t.net.repack()
            t.batch_in.copy_(torch.from_numpy(b))
            t.losses.zero_()
            t.net.flat_grad.zero_()
            t.net.backward(t._front())
            grads.append(t.net.flat_grad.clone())
        ops.adam_tf1(flat, (grads[0] + grads[1]) / 2, m, v, 1e-3, 0.9, 0.999, 1e-8, t=step)
    torch.cuda.synchronize()
    want = flat.cpu().numpy()
    # Adam normalises the step: compare the parameter CHANGE, where it is not dominated by near-zero gradients
    delta_want, delta_got = want - params_flat(params), got[0] - params_flat(params)
    cos = float(np.dot(delta_want, delta_got) / (np.linalg.norm(delta_want) * np.linalg.norm(delta_got)))
    assert cos > 0.9999, cos
    assert np.abs(got[0] - want).max() <= 2.1e-3          # at most the two Adam steps of a sign flip at g ~ 0


def params_flat(params):
    return np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1) for p in params])
