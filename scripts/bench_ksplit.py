"""Split-K against the unsplit launch on the VGG layers whose tiles do not fill the chip: us per call
(a split call = slice launch + fold launch), replayed from a HIP graph.  B=<batch> selects the batch."""
import os, sys
os.environ.setdefault('STMI_ENV_LIVE', '1')   # this script switches kernel knobs inside one process
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import ops

dev = torch.device('cuda:0')
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
LAYERS = [('vgg3_x', 64, 64, 256, 256), ('vgg4_1', 32, 32, 256, 512), ('vgg4_x', 32, 32, 512, 512), ('vgg5_x', 16, 16, 512, 512)]


def run(fn, iters=20):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(iters):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        graph.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


if __name__ == '__main__':
    splits = [0, 1, 2, 4, 8, 16]
    print('%-7s %2s | %8s | ' % ('layer', 'B', 'auto ks') + ' '.join('%8s' % ('ks=%d' % k if k else 'auto') for k in splits) + '   (us per call)')
    for B in [int(b) for b in os.environ.get('B', '1,4,8').split(',')]:
        for name, h, w, ci, co in LAYERS:
            x = torch.randn((B, h, w, ci), device=dev).to(dtype)
            wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
            wp = ops.pack_conv_weights(wt, ops.PACK_FWD, dtype)
            y = torch.empty((B, h, w, co), device=dev, dtype=dtype)
            bias = torch.zeros(co, device=dev)
            ws = torch.empty(16 * y.numel(), device=dev)
            os.environ.pop('STMI_KSPLIT', None)
            auto = ops.conv_ksplit_ws_floats(x.shape, (3, 3, ci, co), 1, dtype) // y.numel()
            row = []
            for k in splits:
                if k: os.environ['STMI_KSPLIT'] = str(k)
                else: os.environ.pop('STMI_KSPLIT', None)
                t = run(lambda: ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, bias=bias, relu=True, out=y, ws=ws))
                split = '[split K]' in ops.last_conv_kernel()
                row.append('%7.1f%s' % (t, '*' if split else ' '))
            print('%-7s %2d | %8d | ' % (name, B, auto) + ' '.join(row))
