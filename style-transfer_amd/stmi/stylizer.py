"""Static executor for the 3-5-3 stylizer (reference full_graph_stylizer_network.py:42-82).

All parameters live in ONE flat fp32 buffer in the reference's variable creation
order (per block: W, IN shift, IN scale; model_utilities.py:236-239, :284-285),
so the optimizer and the data-parallel all-reduce each touch a single tensor.
Activations are NHWC in the compute dtype; every buffer is allocated once, so a
forward/backward pass is a fixed list of kernel launches that can be captured
in a HIP graph.
"""
import numpy as np
import torch

from . import ops, tuning
from .tuning import SERIAL as _SERIAL

# (kind, cout, k, stride, act, residual-role)
#   residual-role: 'a' = first conv of a residual block, 'b' = second (adds the block input)
_BLOCKS = (
    [('conv', 32, 9, 1, ops.ACT_RELU, None), ('conv', 64, 3, 2, ops.ACT_RELU, None), ('conv', 128, 3, 2, ops.ACT_RELU, None)]
    + [('conv', 128, 3, 1, ops.ACT_RELU, 'a'), ('conv', 128, 3, 1, ops.ACT_NONE, 'b')] * 5
    + [('deconv', 64, 3, 2, ops.ACT_RELU, None), ('deconv', 32, 3, 2, ops.ACT_RELU, None),
       ('conv', 3, 9, 1, ops.ACT_SIGMOID, None)]
)


def param_shapes(cin=3):
    """The 48 tensor shapes in creation order."""
    shapes, c = [], cin
    for kind, co, k, _s, _a, _r in _BLOCKS:
        shapes.append((k, k, co, c) if kind == 'deconv' else (k, k, c, co))
        shapes += [(co,), (co,)]
        c = co
    return shapes


def init_params(seed=1, cin=3):
    """Truncated normal (sigma .1, resampled beyond 2 sigma) for W, zeros for shift,
    ones for scale -- the initialisers of model_utilities.py:231-232, :284-285.
    Seeded NumPy stream; not bit-equal to tf.truncated_normal(seed=1)."""
    rng = np.random.default_rng(seed)
    out = []
    for i, shp in enumerate(param_shapes(cin)):
        if len(shp) == 4:
            a = rng.standard_normal(shp)
            bad = np.abs(a) > 2.0
            while bad.any():
                a[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(a) > 2.0
            out.append((a * 0.1).astype(np.float32))
        elif i % 3 == 1:
            out.append(np.zeros(shp, np.float32))
        else:
            out.append(np.ones(shp, np.float32))
    return out


def stylized_size(height, width):
    """Output size of the stylizer for an input of height x width: two stride-2 SAME convolutions (ceil) and two
    transposed convolutions that double exactly (full_graph_stylizer_network.py:62-71) -- the input size when it
    is a multiple of 4, the next multiple of 4 otherwise."""
    f = lambda v: -(-(-(-v // 2)) // 2) * 4
    return f(height), f(width)


class _Block:
    pass


class StylizerNet:
    def __init__(self, batch, height, width, device, dtype=torch.float32, params=None, seed=1, training=False,
                 out_f32=None):
        """out_f32: optional caller-owned fp32 [batch,height,width,3] buffer that receives the stylized batch (a
        trainer hands in the half of its VGG input that the stylized pictures occupy)."""
        self.n, self.h, self.w = batch, height, width
        self.device, self.dtype, self.training = device, dtype, training
        shapes = param_shapes()
        sizes = [int(np.prod(s)) for s in shapes]
        self.num_params = sum(sizes)
        self.flat = torch.empty(self.num_params, dtype=torch.float32, device=device)
        self.flat_grad = torch.zeros(self.num_params, dtype=torch.float32, device=device) if training else None
        offs = np.concatenate([[0], np.cumsum(sizes)])
        self.views = [self.flat[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(len(shapes))]
        self.grad_views = ([self.flat_grad[offs[i]:offs[i + 1]].view(shapes[i]) for i in range(len(shapes))]
                           if training else None)
        self.offsets = offs
        self.load_params(init_params(seed) if params is None else params)

        self.x_in = torch.empty((batch, height, width, 3), dtype=dtype, device=device)
        self.blocks = []
        h, w, c = height, width, 3
        ws_floats = 0
        for i, (kind, co, k, s, act, role) in enumerate(_BLOCKS):
            b = _Block()
            b.kind, b.co, b.k, b.s, b.act, b.role, b.ci = kind, co, k, s, act, role, c
            b.in_h, b.in_w = h, w
            if kind == 'deconv':
                b.kshape = (k, k, co, c)
                b.out_h, b.out_w = h * s, w * s
                fwd_mode, bwd_mode = ops.PACK_DGRAD, ops.PACK_FWD
                pk = (k, k, co, c)
            else:
                b.kshape = (k, k, c, co)
                b.out_h, b.out_w = ops.same_out(h, s), ops.same_out(w, s)
                fwd_mode, bwd_mode = ops.PACK_FWD, ops.PACK_DGRAD
                pk = (k, k, c, co)
            # 9x9 layers with 3 channels on one side run as 9 vertical taps over 32-wide row
            # patches ('in': patches of the input; 'out': the conv emits (kx, co) patches)
            b.rows = None
            if kind == 'conv' and s == 1:
                if c % 8 != 0 and k * c <= 32 and i == 0:
                    b.rows = 'in'
                elif co % 8 != 0 and k * co <= 32 and c % 8 == 0:
                    b.rows = 'out'
                    fwd_mode = ops.PACK_FWD_ROWOUT
            # the 3-channel first layer has a kernel of its own that reads the fp32 image directly; its
            # input row patches are then needed by the weight gradient only
            b.rgb = b.rows == 'in' and (k, co) in ops.RGB_CONFIGS and 'r' not in _SERIAL
            if b.rgb:
                fwd_mode = ops.PACK_FWD_RGB
            # ... and the backward-data of the 3-channel LAST layer is that same kernel on dy
            b.rgb_bwd = b.rows == 'out' and co == 3 and (k, c) in ops.RGB_CONFIGS[:1] and 'r' not in _SERIAL
            if b.rgb_bwd:
                bwd_mode = ops.PACK_DGRAD_RGB
            # ... and the forward of that layer has a one-launch kernel that folds the row patches in LDS
            b.to3 = b.rows == 'out' and co == 3 and ops.to3_supported(k, c, dtype) and 'r' not in _SERIAL
            if b.rows:
                b.pad_left = (k - 1) // 2
                need_rows = (training or not b.rgb) if b.rows == 'in' else not b.to3
                b.rows_buf = torch.empty((batch, h, w, 32), dtype=dtype, device=device) if need_rows else None
                if training:
                    b.dtmp = torch.empty((k, 1, 32, co) if b.rows == 'in' else (k, 1, c, 32), dtype=torch.float32,
                                         device=device)
                    b.wg_call = ((batch, h, w, 32), (k, 1, 32, co)) if b.rows == 'in' else ((batch, h, w, c), (k, 1, c, 32))
                    b.rows_gbuf = torch.empty((batch, h, w, 32), dtype=dtype, device=device) if b.rows == 'out' else None
            # plain convolutions whose kernel can emit per-tile sums of its output: the instance norm behind
            # them skips its statistics pass
            b.tile_stats = None
            if kind == 'conv' and b.rows is None and tuning.TILE_STATS:
                ts = ops.conv_tile_stats_tiles((batch, h, w, c), b.kshape, s, dtype)
                # only where the apply blocks fold the tile rows themselves (<= 32 tiles a sample): with more
                # tiles the finalize launch and the epilogue cost what the statistics pass costs (measured at
                # 128 and at 1020 tiles a sample: 21.2 -> 21.1 us and 1021 -> 1026 us per conv + norm pair)
                # ... or where the persistent kernel leaves the rows for free (1080p planes)
                if 0 < ts <= tuning.TILE_STATS_MAX or (
                        ts > 0 and ops.conv_tile_stats_free((batch, h, w, c), b.kshape, s, dtype)):
                    b.tile_stats = torch.zeros((batch, ts, 4, co), dtype=torch.float32, device=device)
            # ... the 3-channel first layer, whose statistics pass reads the largest plane of the network
            if b.rgb and tuning.TILE_STATS:
                ts = ops.conv_rgb_tile_stats_tiles(h, w, co, k)
                if ts > 0:
                    b.tile_stats = torch.zeros((batch, ts, 4, co), dtype=torch.float32, device=device)
            # ... transposed convolutions: the up-sampling kernel takes the sums from the rows it stores
            if kind == 'deconv' and tuning.TILE_STATS:
                ts = ops.conv_transpose_tile_stats_tiles((batch, h, w, c), b.kshape, s, dtype)
                if ts > 0:
                    b.tile_stats = torch.zeros((batch, ts, 4, co), dtype=torch.float32, device=device)
            # ... and in the backward pass the data-gradient kernel of this block can leave the sums the
            # instance norm of the block BELOW needs (its output gradient is what the kernel writes), so
            # that norm's backward skips its reduction pass
            b.bwd_rows = None
            if training and i > 0 and kind == 'conv' and b.rows is None and tuning.IN_SUMS:
                ts = ops.conv_dgrad_in_sums_tiles((h, w), batch, b.kshape, s, dtype)
                if ts > 0:
                    self.blocks[i - 1].bwd_rows = torch.zeros((batch, ts, 2, c), dtype=torch.float32, device=device)
            b.fwd_mode, b.bwd_mode = fwd_mode, bwd_mode
            b.W, b.shift, b.scale = self.views[3 * i], self.views[3 * i + 1], self.views[3 * i + 2]
            b.pk = pk
            b.n_fwd = ops.packed_elems(*pk, fwd_mode)
            b.n_bwd = ops.packed_elems(*pk, bwd_mode) if training and i > 0 else 0
            oshape = (batch, b.out_h, b.out_w, co)
            b.stats = torch.empty((batch, co, 2), dtype=torch.float32, device=device)
            if training:
                b.dW, b.dshift, b.dscale = (self.grad_views[3 * i], self.grad_views[3 * i + 1],
                                            self.grad_views[3 * i + 2])
                b.conv_out = torch.empty(oshape, dtype=dtype, device=device)
                b.y = torch.empty(oshape, dtype=dtype, device=device)
                # workspace floats of this block's weight-gradient call (partial sums of its pixel slices)
                if kind == 'deconv':
                    b.wg_need = ops.conv2d_transpose_wgrad_ws_floats((batch, h, w, c), b.kshape, s, dtype)
                elif b.rows:
                    b.wg_need = ops.conv2d_wgrad_ws_floats(b.wg_call[0], b.wg_call[1], 1, dtype)
                else:
                    b.wg_need = ops.conv2d_wgrad_ws_floats((batch, h, w, c), b.kshape, s, dtype)
            ws_floats = max(ws_floats, ops._lib.load().stmi_instance_norm_ws_floats(batch, b.out_h, b.out_w, co))
            self.blocks.append(b)
            h, w, c = b.out_h, b.out_w, co
        # all packed weight copies live in one buffer so that one launch refreshes them
        total, table = 0, []
        for i, b in enumerate(self.blocks):
            b.off_fwd, total = total, total + (b.n_fwd + 63) // 64 * 64
            table.append((int(offs[3 * i]), b.off_fwd, b.pk, b.fwd_mode))
            if b.n_bwd:
                b.off_bwd, total = total, total + (b.n_bwd + 63) // 64 * 64
                table.append((int(offs[3 * i]), b.off_bwd, b.pk, b.bwd_mode))
        self.packed = torch.zeros(total, dtype=dtype, device=device)
        self._pack_table = ops.make_pack_table(table)
        for b in self.blocks:
            b.wp_fwd = self.packed[b.off_fwd:b.off_fwd + b.n_fwd]
            b.wp_bwd = self.packed[b.off_bwd:b.off_bwd + b.n_bwd] if b.n_bwd else None
        if out_f32 is None:
            out_f32 = torch.empty((batch, h, w, 3), dtype=torch.float32, device=device)
        assert tuple(out_f32.shape) == (batch, h, w, 3) and out_f32.dtype == torch.float32 and out_f32.is_contiguous()
        self.out_f32 = out_f32
        self.ws = torch.empty(ws_floats, dtype=torch.float32, device=device)
        if not training:
            # inference: blocks share a small pool of buffers sized for the largest activation
            biggest = max(batch * b.out_h * b.out_w * b.co for b in self.blocks)
            self._pool = [torch.empty(biggest, dtype=dtype, device=device) for _ in range(4)]
        else:
            biggest = max(batch * b.out_h * b.out_w * b.co for b in self.blocks)
            # Gradient buffers: g, the skip gradient, dconv, dx -- and two more, so that a weight
            # gradient still reading an older dconv on the side stream does not hold up the chain.
            self._g = [torch.empty(biggest, dtype=dtype, device=device) for _ in range(tuning.GRAD_BUFS)]
            self._side = torch.cuda.Stream(device=device) if device.type == 'cuda' and 'w' not in _SERIAL else None
            # the weight-gradient calls run in stream order on one stream: one workspace serves them all
            self.wgrad_ws = torch.empty(max(4, max(b.wg_need for b in self.blocks)), dtype=torch.float32, device=device)
            self._readers = {}  # data_ptr of a gradient buffer -> (issue number, event) of its last side-stream reader
            self._reader_seq = 0
        self.repack()
        self._absorb_ws = None
        self._plan_absorb()

    # ------------------------------------------------------------ parameters
    def load_params(self, params):
        assert len(params) == len(self.views), 'expected %d tensors in creation order' % len(self.views)
        for v, p in zip(self.views, params):
            t = torch.as_tensor(np.asarray(p), dtype=torch.float32)
            assert tuple(t.shape) == tuple(v.shape), (tuple(t.shape), tuple(v.shape))
            v.copy_(t)
        if hasattr(self, '_absorb_ws'):
            self._plan_absorb()

    def export_params(self):
        return [v.detach().cpu().numpy().copy() for v in self.views]

    def _plan_absorb(self):
        """Inference: where a convolution can absorb the norm + ReLU in front of it (ops.conv2d_fwd_absorb_norm: bf16,
        the second convolution of a residual block on 1080p-sized planes, the down-sampling layers; every scale of
        that norm positive; the normalised tensor needed by nobody else), the norm's apply pass is not run at all.
        b.absorb_ws is set on the block whose norm the NEXT block absorbs.  Decided from the parameters: called again
        after they change."""
        for b in self.blocks:
            b.absorb_ws = None
        if self.training or not tuning.ABSORB_NORM or not tuning.TILE_STATS:
            return
        for i, b in enumerate(self.blocks[:-1]):
            nb = self.blocks[i + 1]
            # the next block must read the tensor for its convolution only (a residual block's first convolution
            # also keeps it as the skip tensor), and leave tile sums for its own norm
            if nb.role == 'a' or b.tile_stats is None or b.act != ops.ACT_RELU:
                continue
            shape = (self.n, b.out_h, b.out_w, b.co)
            if nb.kind == 'conv' and nb.rows == 'out' and nb.to3:      # the 9 x 9 layer into three channels
                need = ops.conv_to3_absorb_norm_ws_bytes(shape, nb.k, self.dtype)
                if need == 0:
                    continue
            elif nb.kind == 'deconv':                                  # the up-sampling kernel
                if nb.tile_stats is None:
                    continue
                need = ops.conv_transpose_absorb_norm_ws_bytes(shape, nb.kshape, nb.s, self.dtype)
                if need == 0:
                    continue
            else:
                if nb.kind != 'conv' or nb.rows is not None or nb.k != 3 or nb.tile_stats is None:
                    continue
                need = ops.conv_absorb_norm_ws_bytes(shape, nb.kshape, nb.s, self.dtype)
                if need == 0 or ops.conv_absorb_norm_tiles(shape, nb.kshape, nb.s, self.dtype) != nb.tile_stats.shape[1]:
                    continue
            if bool((b.scale > 0).all()):
                if self._absorb_ws is None or self._absorb_ws.numel() < need:
                    self._absorb_ws = torch.empty(need, dtype=torch.uint8, device=b.scale.device)
                b.absorb_ws = True
        for b in self.blocks:      # the blocks run in stream order: one workspace serves them all
            if b.absorb_ws:
                b.absorb_ws = self._absorb_ws

    def repack(self):
        """Refreshes the K-contiguous weight copies after the fp32 masters changed."""
        ops.pack_conv_weights_multi(self.flat, self.packed, self._pack_table)

    # --------------------------------------------------------------- forward
    def _view(self, buf, b):
        n = self.n * b.out_h * b.out_w * b.co
        return buf[:n].view(self.n, b.out_h, b.out_w, b.co)

    def forward_u8(self, x_u8, out=None):
        """8-bit frames in, 8-bit frames out (inference executors): x_u8 [n,h,w,3] torch.uint8 as a decoder
        leaves them -> uint8 [n,h,w,3] = (stylized * 255) truncated, what the reference's save_image writes
        (image_utilities.py:195-196, :223-236).  The division by 255 happens while the first layer reads the
        frame, the conversion back in the last layer's normalisation pass: a frame crosses the host link as
        3 bytes a pixel each way."""
        assert not self.training and self.blocks[0].rgb, 'the 8-bit path is built on the 3-channel first-layer kernel'
        if out is None:
            out = torch.empty((self.n,) + stylized_size(self.h, self.w) + (3,), dtype=torch.uint8, device=x_u8.device)
        self._forward(x_u8, out_u8=out)
        return out

    def forward(self, x_f32):
        """x_f32 [n,h,w,3] fp32 in [0,1] -> fp32 [n,h,w,3] (also kept as self.out_f32)."""
        return self._forward(x_f32)

    def _forward(self, x_f32, out_u8=None):
        self._x_f32 = x_f32
        u8_in = x_f32.dtype == torch.uint8
        if not self.blocks[0].rgb:
            ops.cast(x_f32, self.dtype, out=self.x_in) if self.dtype != torch.float32 else self.x_in.copy_(x_f32)
        b0 = self.blocks[0]
        self._rows0_ready = False
        if self.training and b0.rgb and self._side is not None:
            # the first layer's weight gradient wants row patches of the input: they depend on nothing the
            # pass computes, so they are made beside it instead of at the very end of the backward pass
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                ops.cast(x_f32, self.dtype, out=self.x_in) if self.dtype != torch.float32 else self.x_in.copy_(x_f32)
                ops.pixels_to_rows(self.x_in, b0.k, -b0.pad_left, out=b0.rows_buf)
            self._rows0_ready = True
        cur = self.x_in
        skip = None
        absorbed = None      # the block whose norm the next convolution absorbs
        last = len(self.blocks) - 1
        for i, b in enumerate(self.blocks):
            if self.training:
                conv_out, y = b.conv_out, b.y
            else:
                # rotate through the pool, never overwriting the live skip tensor
                free = [p for p in self._pool if p.data_ptr() not in
                        (cur.data_ptr(), None if skip is None else skip.data_ptr())]
                conv_out, y = self._view(free[0], b), self._view(free[1], b)
            if b.kind == 'deconv' and absorbed is not None:
                ops.conv2d_transpose_fwd_absorb_norm(cur, absorbed.tile_stats, absorbed.scale, absorbed.shift, b.wp_fwd,
                                                     b.kshape, b.s, b.tile_stats, absorbed.absorb_ws, in_stats=absorbed.stats,
                                                     out=conv_out)
                absorbed = None
            elif b.kind == 'deconv':
                ops.conv2d_transpose_fwd(cur, b.wp_fwd, b.kshape, b.s, out=conv_out, tile_stats=b.tile_stats)
            elif b.rgb and u8_in:
                ops.conv2d_rgb_u8_fwd(x_f32, b.wp_fwd, b.k, b.co, self.dtype, out=conv_out, tile_stats=b.tile_stats)
            elif b.rgb:
                ops.conv2d_rgb_fwd(x_f32, b.wp_fwd, b.k, b.co, self.dtype, out=conv_out, tile_stats=b.tile_stats)
            elif b.rows == 'in':
                ops.pixels_to_rows(cur, b.k, -b.pad_left, out=b.rows_buf)
                ops.conv2d_fwd(b.rows_buf, b.wp_fwd, (b.k, 1, 32, b.co), 1, out=conv_out)
            elif b.rows == 'out' and b.to3 and absorbed is not None:
                ops.conv2d_to3_absorb_norm(cur, absorbed.tile_stats, absorbed.scale, absorbed.shift, b.wp_fwd, b.k,
                                           absorbed.absorb_ws, in_stats=absorbed.stats, out=conv_out)
                absorbed = None
            elif b.rows == 'out' and b.to3:
                ops.conv2d_to3(cur, b.wp_fwd, b.k, out=conv_out)
            elif b.rows == 'out':
                ops.conv2d_fwd(cur, b.wp_fwd, (b.k, 1, b.ci, 32), 1, out=b.rows_buf)
                ops.rows_to_pixels_sum(b.rows_buf, b.co, b.k, -b.pad_left, out=conv_out)
            elif absorbed is not None:
                # cur is the raw output of the block's first convolution: its norm + ReLU happen on the way in
                ops.conv2d_fwd_absorb_norm(cur, absorbed.tile_stats, absorbed.scale, absorbed.shift, b.wp_fwd, b.kshape,
                                           b.s, b.tile_stats, absorbed.absorb_ws, in_stats=absorbed.stats, out=conv_out)
                absorbed = None
            else:
                ops.conv2d_fwd(cur, b.wp_fwd, b.kshape, b.s, out=conv_out, tile_stats=b.tile_stats)
            if b.role == 'a':
                skip = cur
            if b.absorb_ws is not None:     # the next block's convolution applies this norm + ReLU on its way in
                absorbed = b
                cur = conv_out
                continue
            if i == last and out_u8 is not None:
                ops.instance_norm_stats(conv_out, ws=self.ws, stats=b.stats)
                ops.instance_norm_apply_u8(conv_out, b.stats, b.scale, b.shift, b.act, out=out_u8)
                return out_u8
            ops.instance_norm_fwd(conv_out, b.scale, b.shift, b.act, residual=skip if b.role == 'b' else None,
                                  want_f32=(i == last), ws=self.ws, out=y, stats=b.stats,
                                  out_f32=self.out_f32 if i == last else None, tile_stats=b.tile_stats)
            if b.role == 'b':
                skip = None
            cur = y
        return self.out_f32

    # -------------------------------------------------------------- backward
    def backward(self, d_out, hi=None, lo=0):
        """d_out: gradient w.r.t. the stylized image, compute dtype [n,h,w,3].
        Every parameter gradient is WRITTEN into its slice of self.flat_grad (nothing
        to clear beforehand, no atomics: a step's gradients are bit-reproducible); the
        weight-gradient kernels run on a side stream beside the data-gradient chain
        and are joined before the call returns.

        The pass can be cut into block ranges so that a data-parallel caller can
        start reducing the gradients of finished blocks: backward(d_out, hi=15,
        lo=11) then backward(None, hi=10, lo=7) ... continue where the previous
        call stopped."""
        assert self.training
        if hi is None:
            hi = len(self.blocks) - 1
        if d_out is not None:
            self._bstate = (d_out, None)
        g, pending_skip = self._bstate  # pending_skip: gradient flowing along a residual connection
        bufs = self._g
        main, side = torch.cuda.current_stream(), self._side

        def writable(exclude):
            """Two buffers to overwrite.  Buffers no weight-gradient kernel has read come first, then those
            whose reader was issued LONGEST ago: in a captured graph every wait is a dependency edge, whether
            or not the reader would long be done at run time, and an edge to the weight gradient that has
            only just been issued stalls the chain for that kernel's whole duration."""
            free = [p for p in bufs if p.data_ptr() not in exclude]
            free.sort(key=lambda p: self._readers.get(p.data_ptr(), (-1, None))[0])
            for p in free[:2]:
                seq_ev = self._readers.pop(p.data_ptr(), None)
                if seq_ev is not None:
                    main.wait_event(seq_ev[1])
            return free[0], free[1]

        def fork():
            """Marks the point of the chain a side-stream job depends on."""
            if side is None:
                return None
            ev = torch.cuda.Event()
            ev.record(main)
            return ev

        def on_side(buf, fn, after):
            """The parameter-gradient kernels only feed the optimizer: they run beside the chain.
            They are issued AFTER the chain's next kernel (and wait only for `after`, the fork
            point): a captured graph keeps the first-issued successor of a node on the node's
            own hardware queue, and a hop to another queue costs the chain several microseconds."""
            if side is None:
                fn()
                return
            side.wait_event(after)
            with torch.cuda.stream(side):
                fn()
                ev = torch.cuda.Event()
                ev.record(side)
            self._reader_seq += 1
            self._readers[buf.data_ptr()] = (self._reader_seq, ev)

        have_rows = False   # the previous iteration's data-gradient kernel left the sums of this block's norm
        for i in range(hi, lo - 1, -1):
            b = self.blocks[i]
            x_in = self.blocks[i - 1].y if i > 0 else self.x_in
            w0, w1 = writable((g.data_ptr(), None if pending_skip is None else pending_skip.data_ptr()))
            dconv = self._view(w0, b)
            if b.role == 'b':
                pending_skip = g  # d(block input) receives d(block output) unchanged
            ops.instance_norm_bwd(b.conv_out, g, b.stats, b.scale, b.shift, b.dscale, b.dshift, b.act, ws=self.ws,
                                  out=dconv, rows=b.bwd_rows if have_rows else None, accumulate=False)
            have_rows = False
            rows_g = None
            wws = self.wgrad_ws
            if b.kind == 'deconv':
                wgrad = lambda: ops.conv2d_transpose_wgrad(x_in, dconv, b.kshape, b.s, out=b.dW, ws=wws)
            elif b.rows == 'in':      # rows_buf still holds the forward's input patches, unless the forward
                def wgrad():          # ran the 3-channel kernel: then they are made here, beside the chain
                    if b.rgb and not self._rows0_ready:
                        x32 = self._x_f32
                        ops.cast(x32, self.dtype, out=self.x_in) if self.dtype != torch.float32 else self.x_in.copy_(x32)
                        ops.pixels_to_rows(self.x_in, b.k, -b.pad_left, out=b.rows_buf)
                    ops.conv2d_wgrad(b.rows_buf, dconv, (b.k, 1, 32, b.co), 1, out=b.dtmp, ws=wws)
                    ops.rows_wgrad_to_hwio(b.dtmp, b.kshape, 0, out=b.dW)
            elif b.rows == 'out' and b.rgb_bwd:   # row patches of dconv: for the weight gradient only
                def wgrad():
                    rg = ops.pixels_to_rows(dconv, b.k, b.pad_left - (b.k - 1), out=b.rows_gbuf)
                    ops.conv2d_wgrad(x_in, rg, (b.k, 1, b.ci, 32), 1, out=b.dtmp, ws=wws)
                    ops.rows_wgrad_to_hwio(b.dtmp, b.kshape, 1, out=b.dW)
            elif b.rows == 'out':
                rows_g = ops.pixels_to_rows(dconv, b.k, b.pad_left - (b.k - 1), out=b.rows_gbuf)

                def wgrad():
                    ops.conv2d_wgrad(x_in, rows_g, (b.k, 1, b.ci, 32), 1, out=b.dtmp, ws=wws)
                    ops.rows_wgrad_to_hwio(b.dtmp, b.kshape, 1, out=b.dW)
            else:
                wgrad = lambda: ops.conv2d_wgrad(x_in, dconv, b.kshape, b.s, out=b.dW, ws=wws)
            forked = fork()
            if i == 0:
                on_side(w0, wgrad, forked)
                break
            n_in = self.n * b.in_h * b.in_w * b.ci
            dx = w1[:n_in].view(self.n, b.in_h, b.in_w, b.ci)
            addend = pending_skip if b.role == 'a' else None
            if b.kind == 'deconv':
                ops.conv2d_transpose_dgrad(dconv, b.wp_bwd, b.kshape, b.s, addend=addend, out=dx)
            elif b.rgb_bwd:
                assert addend is None
                ops.conv2d_rgb_dgrad(dconv, b.wp_bwd, b.k, b.ci, out=dx)
            elif rows_g is not None:
                ops.conv2d_dgrad(rows_g, b.wp_bwd, (b.k, 1, b.ci, 32), 1, (b.in_h, b.in_w), addend=addend, out=dx)
            else:
                below = self.blocks[i - 1]
                in_bwd = None
                if below.bwd_rows is not None and i - 1 >= lo:   # the norm below runs in this call
                    in_bwd = (below.conv_out, below.stats, below.scale, below.shift, below.act, below.bwd_rows)
                    have_rows = True
                ops.conv2d_dgrad(dconv, b.wp_bwd, b.kshape, b.s, (b.in_h, b.in_w), addend=addend, out=dx, in_bwd=in_bwd)
            on_side(w0, wgrad, forked)
            if b.role == 'a':
                pending_skip = None
            g = dx
        # the caller reduces / applies the gradients of this range next: join the side stream
        if side is not None:
            main.wait_stream(side)
            self._readers.clear()
        self._bstate = (g, pending_skip)

    def param_range(self, hi, lo):
        """[start, end) of the flat parameter/gradient buffer covered by blocks lo..hi."""
        return int(self.offsets[3 * lo]), int(self.offsets[3 * (hi + 1)])
