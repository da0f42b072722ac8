"""Gram backward (d tap = tap x dG per sample) at the five style layers of a training batch: us per call alone on the
chip, against the bytes / flops it moves."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'scripts')]
from stmi import ops
from bench_ksplit import run

dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
B = int(os.environ.get('B', '4'))
if __name__ == '__main__':
    for name, hw, c in [('conv1_1', 256, 64), ('conv2_1', 128, 128), ('conv3_1', 64, 256), ('conv4_1', 32, 512), ('conv5_1', 16, 512)]:
        f = torch.randn((B, hw, hw, c), device=dev).to(dtype)
        sp = torch.zeros(ops.gram_packed_elems(B, c), device=dev, dtype=dtype)
        df = torch.zeros_like(f)
        t0 = run(lambda: ops.gram_bwd(f, sp, df, accumulate=False))
        k0 = ops.last_conv_kernel().split('stmi::')[-1].split('(')[0]
        t1 = run(lambda: ops.gram_bwd(f, sp, df, accumulate=True))
        es = 2 if dtype == torch.bfloat16 else 4
        mb = 2 * f.numel() * es / 1e6
        gf = 2.0 * B * hw * hw * c * c / 1e9
        print('%-8s %6.1f us (accumulate %6.1f)  %5.1f MB -> %4.2f TB/s  %4.2f GFLOP -> %5.1f TFLOP/s  %s' % (
            name, t0, t1, mb, mb / t0 / 1e6 * 1e6 / 1e6, gf, gf / t0 / 1e3, k0))
