"""Every library call of one eager training step, each replayed alone from a HIP graph and timed.

The recorder wraps stmi.lib.call (the real call still runs); afterwards each recorded call is
captured N times back to back into a graph on the current stream and replayed between two events.
Back-to-back repeats of one call see a warm L2, so the figures are a lower bound of the in-step
cost; their sum against the measured step time shows what overlap and boundaries add or hide.

  python scripts/step_breakdown.py [--dtype bf16] [--full]
"""
import argparse, os, sys, collections
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
import bench
from stmi import lib, ops
from stmi.stylizer import init_params
from stmi.trainer import StyleTrainer

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--full', action='store_true', help='print every call, not only the per-entry totals')
ap.add_argument('--infer', action='store_true', help='16 x 1080p stylizer forward instead of the training step')
args = ap.parse_args()
dev = torch.device('cuda:0')
dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
rng = np.random.default_rng(1234)
if args.infer:
    from stmi.stylizer import StylizerNet
    net = StylizerNet(16, 1080, 1920, dev, dtype, params=init_params(seed=1))
    frames = torch.from_numpy(rng.random((16, 1080, 1920, 3), dtype=np.float32)).to(dev)
    step = lambda: net.forward(frames)
else:
    tr = StyleTrainer(4, 256, 256, dev, bench.synthetic_vgg(), np.zeros(3, np.float32), *bench.STARRY, dtype=dtype,
                      params=init_params(seed=1), world_size=1, use_graph=False)
    tr.set_style_image(torch.from_numpy(np.random.default_rng(99).random((1, 1014, 1280, 3), dtype=np.float32)))
    tr.batch_in.copy_(torch.from_numpy(rng.random((4, 256, 256, 3), dtype=np.float32)))
    step = tr.step
step(); step(); torch.cuda.synchronize()

LOG = []
orig_call = lib.call
main_stream = torch.cuda.current_stream().cuda_stream
def recording_call(name, *a):
    rc = orig_call(name, *a)
    if isinstance(a[-1], int) or a[-1] is None:
        kern = ops.last_conv_kernel() if ('conv2d' in name and 'wgrad' not in name) else ''
        LOG.append((name, a, kern))
    return rc
lib.call = recording_call
step()
torch.cuda.synchronize()
lib.call = orig_call
LOG = [e for e in LOG if e[0] != 'stmi_last_conv_kernel']

def timed(name, a, n=2 if args.infer else 8):
    cur = torch.cuda.current_stream()
    def go():
        orig_call(name, *a[:-1], torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream(); side.wait_stream(cur)
    with torch.cuda.stream(side):
        go()
    cur.wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            go()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3

tot = collections.OrderedDict()
rows = []
for name, a, kern in LOG:
    us = timed(name, a)
    ints = [x for x in a[:-1] if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24)]
    rows.append((name, ints, kern, us, 'main' if a[-1] == main_stream else 'side'))
    t = tot.setdefault(name, [0, 0.0]); t[0] += 1; t[1] += us
if args.full:
    for name, ints, kern, us, where in rows:
        short = kern.replace('void stmi::', '').replace('(stmi::ConvGemmParams)', '')
        print('%-40s %-5s %7.1f us  %s %s' % (name[5:], where, us, ints, short))
print()
total = sum(v[1] for v in tot.values())
for name, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print('%-40s %4d calls %8.1f us  %5.1f %%' % (name[5:], n, us, 100 * us / total))
print('sum of isolated calls: %.1f us over %d calls' % (total, len(rows)))
