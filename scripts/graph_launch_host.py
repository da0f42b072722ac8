"""Host-side cost of replaying the captured training step: time inside graph.replay() (all nodes handed to
the hardware queues) against the GPU time of the step.  usage: python scripts/graph_launch_host.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
import bench
from stmi.stylizer import init_params
from stmi.trainer import StyleTrainer
dev = torch.device('cuda', 0)
rng = np.random.default_rng(1234)
tr = StyleTrainer(4, 256, 256, dev, bench.synthetic_vgg(), np.zeros(3, np.float32), *bench.STARRY, dtype=torch.bfloat16,
                  params=init_params(seed=1))
tr.set_style_image(torch.from_numpy(np.random.default_rng(99).random((1, 1014, 1280, 3), dtype=np.float32)))
tr.batch_in.copy_(torch.from_numpy(rng.random((4, 256, 256, 3), dtype=np.float32)))
for _ in range(10):
    tr.step()
torch.cuda.synchronize()
# (a) one replay at a time: host time inside replay(), then the time until the GPU is done
host, total = [], []
for _ in range(30):
    t0 = time.perf_counter(); tr.step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
print('single replay: host %.1f us in replay(), %.1f us until done' % (np.median(host) * 1e6, np.median(total) * 1e6))
# (b) back to back
t0 = time.perf_counter()
for _ in range(50):
    tr.step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('50 replays back to back: host returns after %.1f us per step, GPU done after %.1f us per step' % ((t1 - t0) / 50 * 1e6, (t2 - t0) / 50 * 1e6))
