"""VGG19 feature extractor and perceptual-loss executor.

Restates what the reference builds with build_vgg_network_replicate_from_mat
(vgg.py:60-108) and StyleTransfer.total_loss (model_utilities.py:391-399) as a
fixed list of HIP kernel launches: preprocessing, 14 frozen 3x3 convolutions
through conv5_2 with fused bias+ReLU, four SAME max-pools, five Gram matrices
of the PRE-ReLU conv1_1..conv5_1 outputs, the content tap at pre-ReLU conv5_2,
and the full backward to the input image (data gradients only; VGG weights are
constants in the reference graph).
"""
import torch

from . import ops
from .tuning import SERIAL as _SERIAL

LAYER_NAMES = [
    'conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
    'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2', 'pool2',
    'conv3_1', 'relu3_1', 'conv3_2', 'relu3_2', 'conv3_3', 'relu3_3', 'conv3_4', 'relu3_4', 'pool3',
    'conv4_1', 'relu4_1', 'conv4_2', 'relu4_2', 'conv4_3', 'relu4_3', 'conv4_4', 'relu4_4', 'pool4',
    'conv5_1', 'relu5_1', 'conv5_2', 'relu5_2', 'conv5_3', 'relu5_3', 'conv5_4', 'relu5_4',
]
STYLE_CONV_TAPS = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CONTENT_CONV_TAP = 'conv5_2'
STYLE_RELU_TAPS = ['relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1']
CONTENT_RELU_TAP = 'relu5_2'


class _Layer:
    pass


class VggFeatures:
    """One weight set, any number of static-shape 'replicas' (see replica())."""

    def __init__(self, weights, mean, device, dtype=torch.float32, with_backward=True, last='conv5_2'):
        """weights: dict conv name -> (W fp32 HWIO [3,3,ci,co], bias [co]) in the orientation
        vgg.py:92-95 hands to conv2d; mean: 3 floats subtracted after the *255/2 scaling."""
        self.device, self.dtype = device, dtype
        self.mean = torch.as_tensor(mean, dtype=torch.float32).reshape(3).to(device).contiguous()
        self.convs = {}
        names = LAYER_NAMES[:LAYER_NAMES.index(last) + 1]
        self.names = names
        for name in names:
            if not name.startswith('conv'):
                continue
            w, b = weights[name]
            w = torch.as_tensor(w, dtype=torch.float32).to(device).contiguous()
            L = _Layer()
            L.kshape = tuple(w.shape)
            L.bias = torch.as_tensor(b, dtype=torch.float32).reshape(-1).to(device).contiguous()
            L.wp_fwd = ops.pack_conv_weights(w, ops.PACK_FWD, dtype)
            L.wp_bwd = ops.pack_conv_weights(w, ops.PACK_DGRAD, dtype) if with_backward else None
            # The RGB layer runs on row patches (3 x 3 x 3 -> three vertical taps over 32-wide rows;
            # PACK_FWD of a < 8 channel kernel already is that layout).  Its backward-data is the
            # forward convolution 64 -> 3 with the flipped, transposed kernel, rows on the output
            # side; d(preprocess)/dx = 255/2 is folded into those weights.
            L.rows = L.kshape[2] < 8 and L.kshape[0] * L.kshape[2] <= 32 and L.kshape[3] % 8 == 0
            # ... unless the 3-channel kernel is built for this layer: it reads the fp32 image and
            # preprocesses it on the way in (forward only; the backward-data stays on row patches)
            L.rgb = (L.kshape[2] == 3 and L.kshape[0] == L.kshape[1]
                     and (L.kshape[0], L.kshape[3]) in ops.RGB_CONFIGS and 'r' not in _SERIAL)
            if L.rgb:
                L.wp_rgb = ops.pack_conv_weights(w, ops.PACK_FWD_RGB, dtype)
            if L.rows and with_backward:
                wt = (w.flip(0, 1).permute(0, 1, 3, 2) * (255.0 / 2.0)).contiguous()
                L.wp_bwd_rows = ops.pack_conv_weights(wt, ops.PACK_FWD_ROWOUT, dtype)
            self.convs[name] = L

    def replica(self, batch, height, width, keep_for_backward=False, relu_taps=False, prefix=0):
        """prefix: leading samples of the batch that only pass forward (a training step's content pictures, whose
        conv5_2 output is the content target of the samples behind them); Gram matrices, loss terms and the
        backward pass cover samples prefix .. batch-1."""
        return VggReplica(self, batch, height, width, keep_for_backward, relu_taps, prefix)


class VggReplica:
    """Static buffers for one input shape."""

    def __init__(self, net, batch, height, width, keep, relu_taps=False, prefix=0):
        self.net, self.n, self.h, self.w, self.keep, self.relu_taps = net, batch, height, width, keep, relu_taps
        assert 0 <= prefix < batch
        self.prefix, self.nl = prefix, batch - prefix     # nl: samples that carry loss terms
        dev, dt = net.device, net.dtype
        self.style_names = STYLE_RELU_TAPS if relu_taps else STYLE_CONV_TAPS
        self.content_name = CONTENT_RELU_TAP if relu_taps else CONTENT_CONV_TAP
        assert not (keep and relu_taps), 'the backward pass is built for the pre-ReLU taps of the training path'
        first = net.convs[net.names[0]]
        if not first.rgb:
            self.x_pre = torch.empty((batch, height, width, 3), dtype=dt, device=dev)
            self.rows_buf = torch.empty((batch, height, width, 32), dtype=dt, device=dev)   # row patches of x_pre
        self.rows_gbuf = torch.empty((batch, height, width, 32), dtype=dt, device=dev) if keep else None
        self.act = {}      # name -> output buffer of that layer (post-ReLU for relu layers)
        self.pre = {}      # conv name -> pre-ReLU output where it is tapped
        self.shape = {}
        h, w, c = height, width, 3
        scratch_need = 0
        for name in net.names:
            if name.startswith('conv'):
                c = net.convs[name].kshape[3]
            elif name.startswith('pool'):
                h, w = (h + 1) // 2, (w + 1) // 2
            self.shape[name] = (batch, h, w, c)
            scratch_need = max(scratch_need, batch * h * w * c)
        tapped_pre = set(STYLE_CONV_TAPS) if not relu_taps else set()
        if keep:
            for name in net.names:
                if name.startswith('conv'):
                    if name in tapped_pre:
                        self.pre[name] = torch.empty(self.shape[name], dtype=dt, device=dev)
                elif name.startswith('relu') or name.startswith('pool'):
                    self.act[name] = torch.empty(self.shape[name], dtype=dt, device=dev)
            self._gbuf = [torch.empty(scratch_need, dtype=dt, device=dev) for _ in range(2)]
        else:
            self._scratch = [torch.empty(scratch_need, dtype=dt, device=dev) for _ in range(3)]
        nl = self.nl
        lshape = {nm: (nl,) + tuple(sh[1:]) for nm, sh in self.shape.items()}
        self.grams = [torch.empty((nl, self.shape[nm][3], self.shape[nm][3]), dtype=torch.float32, device=dev)
                      for nm in self.style_names]
        # partial sums of the Gram kernels' pixel slices (folded in a fixed order: no atomics); the five Gram
        # jobs of a pass run in stream order on one stream, so they share one workspace
        need = max(ops.gram_ws_floats(lshape[nm], dt) for nm in self.style_names)
        self.gram_ws = torch.empty(max(need, 4), dtype=torch.float32, device=dev)
        # raw partial sums of the layers that split their K loop (too few output tiles to fill the chip); the
        # conv chain of a replica, forward and backward, is one stream order, so one workspace serves all layers
        need = 0
        for nm in net.names:
            if nm.startswith('conv') and not net.convs[nm].rgb:
                ks = net.convs[nm].kshape
                hw_ci = (self.shape[nm][1], self.shape[nm][2], ks[2])      # 'same' convolutions of stride 1
                need = max(need, ops.conv_ksplit_ws_floats((batch,) + hw_ci, ks, 1, dt),
                           ops.conv_ksplit_ws_floats((nl,) + hw_ci, ks, 1, dt, dgrad=True) if keep else 0)
        self.ksplit_ws = torch.empty(need, dtype=torch.float32, device=dev) if need else None
        self.content = torch.empty(self.shape[self.content_name], dtype=dt, device=dev)
        if keep:
            self.s_packed = [torch.empty(ops.gram_packed_elems(nl, self.shape[nm][3]), dtype=dt, device=dev)
                             for nm in self.style_names]
            # d(style loss)/d(tap) per style layer, written beside the forward chain; the backward
            # chain's data-gradient kernels take it as an addend
            self.gram_grad = [torch.empty(lshape[nm], dtype=dt, device=dev) for nm in self.style_names]
            self.d_content = torch.empty(lshape[self.content_name], dtype=dt, device=dev)
            self.d_image = torch.empty((nl, height, width, 3), dtype=dt, device=dev)
            self.tv_grad = torch.empty((nl, height, width, 3), dtype=dt, device=dev)
            self._lshape = lshape
        # Gram matrices and their loss terms hang off the conv chain as leaves: with kept
        # (dedicated) tap buffers they run on a side stream beside the following layers.
        self._side = torch.cuda.Stream(device=dev) if keep and dev.type == 'cuda' and 'g' not in _SERIAL else None
        # (A second side stream for alternate taps was measured 3 % SLOWER per training step: every extra
        # stream is another hardware queue for the graph's branches to hop between.)
        self._side2 = None
        self._style_job = None
        self._deferred = []  # (event at the fork point, job) not yet issued on the side stream
        self._style_done = {}  # style layer index (or 'tv') -> event on the side stream after its loss term and gradient

    def _buf(self, pool, shape, avoid):
        n = shape[0] * shape[1] * shape[2] * shape[3]
        for p in pool:
            if all(a is None or p.data_ptr() != a.data_ptr() for a in avoid):
                return p[:n].view(shape)
        raise RuntimeError('no free scratch buffer')

    # --------------------------------------------------------------- forward
    def forward(self, x_f32, want_grams=True, style_job=None):
        """Fills self.grams (5 x fp32 [n,c,c]) and self.content from an fp32 image batch in [0,1].

        style_job(li, tap): optional follow-up of Gram li (its loss term and gradient), launched
        right behind it on the same stream."""
        for _ in self.forward_steps(x_f32, want_grams, style_job):
            pass
        return self.grams, self.content

    def forward_steps(self, x_f32, want_grams=True, style_job=None):
        """forward() as a generator that yields after every layer's launches, so that a caller can
        issue the pass in pieces between the launches of another chain."""
        self._want_grams = want_grams
        self._style_job = style_job
        net = self.net
        names = net.names
        rgb_first = net.convs[names[0]].rgb
        if rgb_first:
            cur = x_f32
        else:
            ops.scale_shift(x_f32, net.mean, 255.0 / 2.0, net.dtype, out=self.x_pre)
            cur = self.x_pre
        i = 0
        while i < len(names):
            name = names[i]
            shape = self.shape[name]
            if name.startswith('conv'):
                L = net.convs[name]
                kshape = L.kshape
                if i == 0 and rgb_first:
                    def conv(src, relu, out, out_pre=None):
                        ops.conv2d_rgb_fwd(src, L.wp_rgb, kshape[0], kshape[3], net.dtype, scale=255.0 / 2.0, mean=net.mean,
                                           bias=L.bias, relu=relu, want_pre=out_pre is not None, out=out, out_pre=out_pre)
                else:
                    if i == 0 and L.rows:
                        k = kshape[0]
                        cur = ops.pixels_to_rows(cur, k, -((k - 1) // 2), out=self.rows_buf)
                        kshape = (k, 1, 32, kshape[3])

                    def conv(src, relu, out, out_pre=None, kshape=kshape, out_pool=None):
                        if out_pool is not None:
                            ops.conv2d_fwd_pool(src, L.wp_fwd, kshape, 1, bias=L.bias, relu=relu, out=out, out_pool=out_pool)
                            return
                        ops.conv2d_fwd(src, L.wp_fwd, kshape, 1, bias=L.bias, relu=relu, want_pre=out_pre is not None,
                                       out=out, out_pre=out_pre, ws=self.ksplit_ws)
                has_relu = i + 1 < len(names) and names[i + 1].startswith('relu')
                tap_pre = (not self.relu_taps) and want_grams and name in STYLE_CONV_TAPS
                if has_relu:
                    rname = names[i + 1]
                    if self.keep:
                        y = self.act[rname]
                        pre = self.pre.get(name)
                    else:
                        y = self._buf(self._scratch, shape, (cur,))
                        pre = self._buf(self._scratch, shape, (cur, y)) if tap_pre else None
                    # conv -> ReLU -> pool: the convolution's launch leaves the pooled tensor as well
                    pname = names[i + 2] if i + 2 < len(names) and names[i + 2].startswith('pool') else None
                    if pname is not None and pre is None and not (i == 0 and rgb_first):
                        yp = self.act[pname] if self.keep else self._buf(self._scratch, self.shape[pname], (cur, y))
                        conv(cur, True, y, out_pool=yp)
                        self._flush_side()
                        self._tap(rname, y)
                        cur = yp
                        i += 3
                        yield
                        continue
                    conv(cur, True, y, pre)
                    self._flush_side()
                    self._tap(name, pre)
                    self._tap(rname, y)
                    cur = y
                    i += 2
                    yield
                    continue
                # last layer of the truncated network: no ReLU follows
                y = self.content if name == self.content_name else self._buf(self._scratch, shape, (cur,))
                conv(cur, False, y)
                self._flush_side()
                self._tap(name, y)
                cur = y
            elif name.startswith('pool'):
                y = self.act[name] if self.keep else self._buf(self._scratch, shape, (cur,))
                ops.maxpool_fwd(cur, out=y)
                self._flush_side()
                cur = y
            i += 1
            yield
        self._flush_side()

    def _flush_side(self):
        """Launches the side-stream jobs whose fork point has been passed.  A captured graph keeps
        the first-issued successor of a node on the node's own hardware queue, so the chain's next
        kernel goes first and the side job second."""
        for ev, job, stream in self._deferred:
            stream.wait_event(ev)
            with torch.cuda.stream(stream):
                job()
        self._deferred.clear()

    def _gram_job(self, li, tensor):
        tensor = tensor[self.prefix:]
        ops.gram_fwd(tensor, out=self.grams[li], ws=self.gram_ws)
        if self._style_job is not None:
            self._style_job(li, tensor)
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._style_done[li] = ev

    def _tap(self, name, tensor):
        if tensor is None:
            return
        if name in self.style_names and self._want_grams:
            li = self.style_names.index(name)
            if self._side is not None:
                # issued after the chain's next kernel (see _flush_side); it waits only for this point
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                stream = self._side2 if (li & 1) and self._side2 is not None else self._side
                self._deferred.append((ev, lambda: self._gram_job(li, tensor), stream))
            else:
                self._gram_job(li, tensor)
        if name == self.content_name and tensor.data_ptr() != self.content.data_ptr():
            self.content.copy_(tensor)

    # -------------------------------------------------------- loss + backward
    LOSS_SLOTS = 7                      # five style terms, content, tv
    LOSS_TERMS = [0, 0, 5, 6, 7]        # loss_slots_sum: out[0] = nothing, out[1] = style, out[2] = content, out[3] = tv

    def loss_and_grad(self, x_f32, style_targets, content_target, style_weight, content_weight, tv_weight, slots,
                      content_stream=None):
        """Runs forward on x_f32, leaves the loss terms in `slots` (ops.loss_slots(LOSS_SLOTS): the caller
        turns them into [_, style, content, tv] with ops.loss_slots_sum(slots, LOSS_TERMS, out4)) and returns
        d(total)/d(x) in the compute dtype, for the samples behind the replica's forward-only prefix (with a prefix,
        content_target must be None: the prefix samples' content tap is the target).

        style_targets: 5 fp32 [c,c] (or [1,c,c]) Grams, content_target: tensor shaped like the content tap;
        content_stream: stream that is still producing content_target (joined before its first use)."""
        assert self.keep
        net, n, p0 = self.net, self.nl, self.prefix
        lead = lambda t: t[p0:]     # the samples that carry loss terms (a contiguous slice of an NHWC batch)
        assert tuple(slots.shape) == (self.LOSS_SLOTS, ops.LOSS_SLOT)
        l_content, l_tv = slots[5], slots[6]
        x_loss = lead(x_f32)

        # style: sum_l mean((G_l - T_l)^2) * w / len(content outputs)   (model_utilities.py:396)
        def style_term(li, tap):
            nm = self.style_names[li]
            c = self.shape[nm][3]
            hw = self.shape[nm][1] * self.shape[nm][2]
            ls = style_weight / float(n * c * c)
            ops.gram_mse_fwd_bwd(self.grams[li], style_targets[li], slots[li], ls, 4.0 * ls / hw, net.dtype,
                                 out=self.s_packed[li])
            # d(style term)/d(tap) = tap x S: everything it needs exists already, so it leaves the
            # backward chain and joins it later as the addend of a data-gradient kernel
            ops.gram_bwd(tap, self.s_packed[li], self.gram_grad[li], accumulate=False)

        # the TV term needs only the image and its gradient joins at the very end: beside the chain
        def tv():
            ops.tv_loss_fwd_bwd(x_loss, l_tv, tv_weight, net.dtype, out=self.tv_grad)
            if self._side is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self._style_done['tv'] = ev

        def join(key):
            """The backward chain meets a side-stream result where it first needs it, not before."""
            ev = self._style_done.pop(key, None)
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)

        self._style_done.clear()
        tv_fork = None
        if self._side is not None:
            tv_fork = torch.cuda.Event()
            tv_fork.record(torch.cuda.current_stream())
        else:
            tv()
        self.forward(x_f32, style_job=style_term)
        self._style_job = None
        if tv_fork is not None:
            # the TV gradient joins the chain at its very end: it goes BEHIND the style jobs, on the stream
            # whose last job is not the one the backward pass needs first
            tv_stream = self._side2 if self._side2 is not None else self._side
            tv_stream.wait_event(tv_fork)
            with torch.cuda.stream(tv_stream):
                tv()
        if content_stream is not None:
            torch.cuda.current_stream().wait_stream(content_stream)
        if p0 > 0:
            # the forward-only samples in front ARE the content pictures of the samples behind them
            assert content_target is None and p0 == n
            content_target = self.content[:p0]
        content = lead(self.content)
        cnt = content.numel()
        ops.mse_fwd_bwd(content, content_target, l_content, content_weight / cnt, 2.0 * content_weight / cnt,
                        out=self.d_content)

        # backward through the truncated VGG; g is always the gradient w.r.t. a conv's pre-ReLU output
        names = net.names
        g = self.d_content
        idx = len(names) - 1
        joined = set()  # style layers whose Gram gradient is already part of g
        while idx >= 0:
            name = names[idx]
            if name.startswith('conv'):
                if name in self.style_names and name not in joined:
                    # no masked data-gradient kernel produced g (the tap feeds a pool): add it here
                    join(self.style_names.index(name))
                    ops.gram_bwd(lead(self.pre[name]), self.s_packed[self.style_names.index(name)], g, accumulate=True)
                L = net.convs[name]
                in_name = names[idx - 1] if idx > 0 else None
                h, w = (self.shape[in_name][1], self.shape[in_name][2]) if in_name else (self.h, self.w)
                if in_name is None and L.rows:
                    # d(preprocess)/dx = 255/2 is in the weights; the TV gradient joins in the row sum
                    k, ci = L.kshape[0], L.kshape[2]
                    join('tv')
                    if ci == 3 and ops.to3_supported(k, L.kshape[3], net.dtype) and 'r' not in _SERIAL:
                        ops.conv2d_to3(g, L.wp_bwd_rows, k, addend=self.tv_grad, out=self.d_image)
                    else:
                        ops.conv2d_fwd(g, L.wp_bwd_rows, (k, 1, L.kshape[3], 32), 1, out=lead(self.rows_gbuf))
                        ops.rows_to_pixels_sum(lead(self.rows_gbuf), ci, k, -((k - 1) // 2), out=self.d_image,
                                               addend=self.tv_grad)
                    g = self.d_image
                elif in_name is None:
                    # d(preprocess)/dx = 255/2; the TV gradient joins here
                    join('tv')
                    ops.conv2d_dgrad(g, L.wp_bwd, L.kshape, 1, (h, w), alpha=255.0 / 2.0, addend=self.tv_grad,
                                     out=self.d_image)
                    g = self.d_image
                elif in_name.startswith('relu'):
                    dx = self._buf(self._gbuf, self._lshape[in_name], (g,))
                    below = names[idx - 2]  # the conv under that ReLU: its Gram gradient joins after the mask
                    addend = None
                    if below in self.style_names and below in self.pre:
                        join(self.style_names.index(below))
                        addend = self.gram_grad[self.style_names.index(below)]
                        joined.add(below)
                    ops.conv2d_dgrad(g, L.wp_bwd, L.kshape, 1, (h, w), relu_mask=lead(self.act[in_name]), addend=addend,
                                     out=dx, ws=self.ksplit_ws)
                    g = dx
                    idx -= 1  # the ReLU is folded into the mask
                else:  # pool output feeds this conv
                    dpool = self._buf(self._gbuf, self._lshape[in_name], (g,))
                    ops.conv2d_dgrad(g, L.wp_bwd, L.kshape, 1, (h, w), out=dpool, ws=self.ksplit_ws)
                    relu_name = names[idx - 2]
                    dx = self._buf(self._gbuf, self._lshape[relu_name], (dpool,))
                    ops.maxpool_bwd(lead(self.act[relu_name]), dpool, out=dx)
                    g = dx
                    idx -= 2
            idx -= 1
        for side in (self._side, self._side2):   # whatever the chain did not meet on its way
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
        self._style_done.clear()
        return self.d_image
