import os, sys, torch, numpy as np
sys.path[:0] = ['/root/repo', '/root/repo/style-transfer_amd']
from oracle import tf_ops as O
from stmi import tuning
from stmi.stylizer import StylizerNet
cuda = torch.device('cuda:0')
def psnr(a, b):
    return float(10 * np.log10(1.0 / ((a.double() - b.double()) ** 2).mean().item()))
for seed in (1, 3):
    params = O.init_stylizer_params(seed=seed)
    g = torch.Generator(device='cpu').manual_seed(11)
    frames = torch.rand((1, 1080, 1920, 3), generator=g, dtype=torch.float32)
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.stylizer_forward(frames.float(), [torch.tensor(p, dtype=torch.float32) for p in params])
    for keep in ([], [0], [1], [0, 1], [3, 5, 7, 9, 11], [0, 1, 3, 5, 7, 9, 11]):
        tuning.ABSORB_NORM = True
        net = StylizerNet(1, 1080, 1920, cuda, torch.bfloat16, params=params)
        for i, b in enumerate(net.blocks):
            if i not in keep:
                b.absorb_ws = None
        out = net.forward(frames.to(cuda)).cpu()
        print('seed', seed, 'absorbing', keep, 'PSNR vs fp32 oracle %.2f dB' % psnr(out, ref), flush=True)
        del net
