"""Phase times inside conv_patch_kernel blocks (build: make -C style-transfer_amd/csrc stamps).
Per layer: one warm launch, then the stamps of its first 256 blocks: entry -> first step (prologue),
first step -> end of the K loop, K loop end -> kernel end (epilogue); consumer wave 0 and producer wave 4.
Also the shader clock the block saw (cycle counter / 100 MHz wall counter)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
from stmi import lib
lib.LIB_PATH = os.path.join(ROOT, 'style-transfer_amd', 'libstmi355_stamps.so')
from stmi import ops
L = lib.load()
L.stmi_debug_stamps.restype = ctypes.c_int
L.stmi_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device('cuda:0')
LAYERS = [('vgg1_2', 256, 256, 64, 64), ('vgg2_2', 128, 128, 128, 128), ('vgg3_x', 64, 64, 256, 256),
          ('vgg4_x', 32, 32, 512, 512), ('vgg5_x', 16, 16, 512, 512), ('res', 64, 64, 128, 128)]
B = int(os.environ.get('B', '4'))
for name, h, w, ci, co in LAYERS:
    x = torch.randn((B, h, w, ci), device=dev).to(torch.bfloat16)
    wt = torch.randn((3, 3, ci, co), device=dev) * (2.0 / (9 * ci)) ** 0.5
    wp = ops.pack_conv_weights(wt, ops.PACK_FWD, torch.bfloat16)
    y = torch.empty((B, h, w, co), device=dev, dtype=torch.bfloat16)
    tiles = None
    if os.environ.get('TS') == '1':     # with the epilogue's tile statistics
        ts = ops.conv_tile_stats_tiles(x.shape, (3, 3, ci, co), 1, torch.bfloat16)
        tiles = torch.zeros((B, ts, 4, co), dtype=torch.float32, device=dev) if ts > 0 else None
    for _ in range(5):
        ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, out=y, tile_stats=tiles)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d_fwd(x, wp, (3, 3, ci, co), 1, out=y, tile_stats=tiles); e1.record(); torch.cuda.synchronize()
    st = np.zeros(256 * 16, np.uint64)
    assert L.stmi_debug_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(256, 2, 4, 2).astype(np.float64)      # block, role (consumer/producer), stamp, (wall, cycles)
    nblk = min(256, int(np.ceil(B * h * w / 128) * max(1, co // (128 if 'Li4' in ops.last_conv_kernel() else 64))))
    st = st[:nblk]
    wall = st[..., 0] * 10.0      # ns
    cyc = st[..., 1]
    t0 = wall[:, :, 0].min()
    def med(a): return np.median(a) / 1e3
    c, p = wall[:, 0], wall[:, 1]
    mhz = np.median((cyc[:, 0, 2] - cyc[:, 0, 1]) / np.maximum(wall[:, 0, 2] - wall[:, 0, 1], 1.0)) * 1e3
    print('%-7s %s  event %.1f us  blocks %d' % (name, ops.last_conv_kernel()[19:52], e0.elapsed_time(e1) * 1e3, nblk))
    print('   consumer: start spread %.2f  prologue %.2f  K loop %.2f  epilogue %.2f  | block total %.2f  kernel span %.2f us  cycle/wall %.0f MHz'
          % ((c[:, 0].max() - t0) / 1e3, med(c[:, 1] - c[:, 0]), med(c[:, 2] - c[:, 1]), med(c[:, 3] - c[:, 2]),
             med(c[:, 3] - c[:, 0]), (wall[:, :, 3].max() - t0) / 1e3, mhz))
    print('   producer: issue prologue %.2f  K loop %.2f' % (med(p[:, 1] - p[:, 0]), med(p[:, 2] - p[:, 1])))
