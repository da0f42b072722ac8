"""1080p inference: one executor of 16 frames against two executors of 8 frames on two streams
(convolutions are MFMA-bound, instance norms HBM-bound: side by side they can overlap)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi.stylizer import StylizerNet, init_params
dev = torch.device('cuda:0')
params = init_params(seed=1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
frames = torch.rand((B, 1080, 1920, 3), device=dev)


def bench(fn, steps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


one = StylizerNet(B, 1080, 1920, dev, torch.bfloat16, params=params)
t1 = bench(lambda: one.forward(frames))
ref = one.forward(frames).clone()
del one; torch.cuda.empty_cache()
for parts in (2, 4):
    nets = [StylizerNet(B // parts, 1080, 1920, dev, torch.bfloat16, params=params) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    chunks = [frames[i * (B // parts):(i + 1) * (B // parts)] for i in range(parts)]

    def run():
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for net, s, c in zip(nets, streams, chunks):
            with torch.cuda.stream(s):
                net.forward(c)
        for s in streams:
            cur.wait_stream(s)
    t2 = bench(run)
    run(); torch.cuda.synchronize()
    same = all(torch.equal(n.out_f32, ref[i * (B // parts):(i + 1) * (B // parts)]) for i, n in enumerate(nets))
    print('%d frames: one executor %.2f ms; %d executors on %d streams %.2f ms (%.0f -> %.0f frames/s), bit-identical: %s'
          % (B, t1, parts, parts, t2, B / t1 * 1e3, B / t2 * 1e3, same), flush=True)
    del nets; torch.cuda.empty_cache()
