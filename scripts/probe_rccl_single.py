"""RCCL behind per-segment graph launches, exercised on ONE GPU: a one-rank nccl process group, the trainer
built as if it were one of two ranks (so that the bucketed async all-reduce, the waits and Adam's 1/world all
run), a few dozen graph-replayed steps, finite losses at the end.  Not a numerics test: a hang detector for the
order 'graph launch -> all_reduce(async) -> graph launch' that cannot be tried with two devices here."""
import os, sys, faulthandler
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
faulthandler.dump_traceback_later(100, exit=True)
import bench
from stmi.stylizer import init_params
from stmi.trainer import StyleTrainer

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
dev = torch.device('cuda:0')
tr = StyleTrainer(4, 256, 256, dev, bench.synthetic_vgg(), np.zeros(3, np.float32), *bench.STARRY, dtype=torch.bfloat16,
                  params=init_params(seed=1), world_size=2, use_graph=True)
tr.set_style_image(torch.from_numpy(np.random.default_rng(99).random((1, 300, 400, 3), dtype=np.float32)))
tr.batch_in.copy_(torch.from_numpy(np.random.default_rng(1).random((4, 256, 256, 3), dtype=np.float32)))
import time
for i in range(5):
    tr.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(50):
    tr.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 50
l = tr.read_losses()
print('segmented graphs + RCCL (1 rank): %.3f ms per step, losses %s' % (dt * 1e3, l))
assert np.isfinite(l).all()
dist.destroy_process_group()
print('ok')
