"""conv -> ReLU -> pool of the four pooled VGG layers: one launch (pool from the staged tile) against two; us per call."""
import os, sys
os.environ.setdefault('STMI_ENV_LIVE', '1')   # this script switches kernel knobs inside one process
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd'), os.path.join(ROOT, 'scripts')]
from stmi import ops
from bench_ksplit import run

dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
LAYERS = [('vgg1_2', 256, 256, 64, 64), ('vgg2_2', 128, 128, 128, 128), ('vgg3_4', 64, 64, 256, 256), ('vgg4_4', 32, 32, 512, 512)]
if __name__ == '__main__':
    print('%-7s %2s | %9s %9s %9s %9s' % ('layer', 'B', 'conv', 'pool', 'conv+pool', 'fused'))
    for B in [int(b) for b in os.environ.get('B', '1,8').split(',')]:
        for name, h, w, ci, co in LAYERS:
            x = torch.randn((B, h, w, ci), device=dev).to(dtype)
            wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
            wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dtype)
            y = torch.empty((B, h, w, co), device=dev, dtype=dtype)
            yp = torch.empty((B, h // 2, w // 2, co), device=dev, dtype=dtype)
            bias = torch.zeros(co, device=dev)
            ks = (3, 3, ci, co)
            os.environ['STMI_NO_POOL_FUSE'] = '1'
            t_conv = run(lambda: ops.conv2d_fwd(x, wp, ks, 1, bias=bias, relu=True, out=y))
            t_pool = run(lambda: ops.maxpool_fwd(y, out=yp))
            t_two = run(lambda: ops.conv2d_fwd_pool(x, wp, ks, 1, bias=bias, relu=True, out=y, out_pool=yp))
            os.environ.pop('STMI_NO_POOL_FUSE')
            t_one = run(lambda: ops.conv2d_fwd_pool(x, wp, ks, 1, bias=bias, relu=True, out=y, out_pool=yp))
            assert '[+pool]' in ops.last_conv_kernel()
            print('%-7s %2d | %9.1f %9.1f %9.1f %9.1f' % (name, B, t_conv, t_pool, t_two, t_one))
