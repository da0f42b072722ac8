"""One conv3_x-shaped forward convolution (256 -> 256 channels, 4 x 64 x 64, bf16), launched five times:
a fixed point for reading hardware counters against a known MFMA count (256 blocks x 4 chunks x 9 taps x
128 MFMA 16x16x32 = 1,179,648 instructions + 64 per block of pipeline priming = 19.3 GFLOP)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops
dev = torch.device('cuda:0')
x = torch.randn((4, 64, 64, 256), device=dev).bfloat16()
w = torch.randn((3, 3, 256, 256), device=dev) * 0.02
wp = ops.pack_conv_weights(w, ops.PACK_FWD, torch.bfloat16)
y = torch.empty((4, 64, 64, 256), device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.conv2d_fwd(x, wp, (3, 3, 256, 256), 1, out=y)
torch.cuda.synchronize()
print(ops.last_conv_kernel())
