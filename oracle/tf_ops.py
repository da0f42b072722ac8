"""Torch-CPU restatement of the TensorFlow ops the reference calls on its hot path.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned.

All tensors are NHWC like the reference's.  Every function works in the dtype of
its inputs: float64 is the ground truth the HIP kernels are checked against,
float32 is what the CPU baseline is timed in.  Gradients come from torch
autograd applied to these forward definitions, which is what TensorFlow's
``optimizer.minimize`` computes for the same graph.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# TensorFlow padding rule
# --------------------------------------------------------------------------
def same_pads(in_size, k, stride):
    """TF 'SAME': out = ceil(in/s), pad_total = max((out-1)*s + k - in, 0),
    pad_before = pad_total // 2.  Returns (out, pad_before, pad_after)."""
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    before = total // 2
    return out, before, total - before


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------
# convolutions: reference model_utilities.py:236-237, :246-248, :163-164
# --------------------------------------------------------------------------
def conv2d_same(x, w_hwio, stride=1, bias=None):
    """tf.nn.conv2d(x, W, [1,s,s,1], 'SAME') with W [kh,kw,Cin,Cout]
    (model_utilities.py:236-237); optional tf.nn.bias_add (model_utilities.py:163-164)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    _, pt, pb = same_pads(x.shape[1], kh, stride)
    _, pl, pr = same_pads(x.shape[2], kw, stride)
    xp = F.pad(_nchw(x), (pl, pr, pt, pb))
    y = F.conv2d(xp, w_hwio.permute(3, 2, 0, 1), bias=bias, stride=stride)
    return _nhwc(y)


def conv2d_transpose_same(x, w_hwoi, stride=2):
    """tf.nn.conv2d_transpose(x, W, [B, s*H, s*W, Cout], [1,s,s,1], 'SAME') with
    W [kh,kw,Cout,Cin] (model_utilities.py:246-248).  It is the gradient of
    conv2d(SAME, stride s) w.r.t. an input of size s*H x s*W, i.e. the full
    transposed convolution cropped at the forward conv's pad_before."""
    kh, kw = w_hwoi.shape[0], w_hwoi.shape[1]
    ho, wo = x.shape[1] * stride, x.shape[2] * stride
    _, pt, _ = same_pads(ho, kh, stride)
    _, pl, _ = same_pads(wo, kw, stride)
    # conv_transpose2d weight layout is [Cin, Cout, kh, kw]
    full = F.conv_transpose2d(_nchw(x), w_hwoi.permute(3, 2, 0, 1), stride=stride)
    # the full result covers (H-1)*s + k rows; rows past it (never reached by
    # any tap) are zero in TensorFlow's result
    need_h, need_w = pt + ho, pl + wo
    if full.shape[2] < need_h or full.shape[3] < need_w:
        full = F.pad(full, (0, max(need_w - full.shape[3], 0), 0, max(need_h - full.shape[2], 0)))
    return _nhwc(full[:, :, pt:pt + ho, pl:pl + wo])


def max_pool_2x2_same(x):
    """tf.nn.max_pool2d(ksize 2, stride 2, 'SAME') (model_utilities.py:166-167):
    out = ceil(in/2); the padded element on an odd edge never wins the max."""
    return _nhwc(F.max_pool2d(_nchw(x), kernel_size=2, stride=2, ceil_mode=True))


# --------------------------------------------------------------------------
# instance norm: model_utilities.py:282-288
# --------------------------------------------------------------------------
IN_EPS = 1e-3


def instance_norm(x, scale, shift, eps=IN_EPS):
    mu = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(1, 2), keepdim=True)  # tf.nn.moments: biased
    return scale * ((x - mu) / (var + eps) ** 0.5) + shift


# --------------------------------------------------------------------------
# losses: model_utilities.py:364-399
# --------------------------------------------------------------------------
def gram_matrix(f):
    """einsum('bijc,bijd->bcd') / (H*W)  (model_utilities.py:380-384)."""
    b, h, w, c = f.shape
    m = f.reshape(b, h * w, c)
    return torch.einsum('bpc,bpd->bcd', m, m) / float(h * w)


def sum_mse(a_list, b_list):
    """add_n(reduce_mean((a[i]-b[i])**2)) with broadcasting (model_utilities.py:387-388)."""
    total = None
    for a, b in zip(a_list, b_list):
        term = ((a - b) ** 2).mean()
        total = term if total is None else total + term
    return total


def total_variation_loss(img):
    """model_utilities.py:369-377."""
    dx = img[:, :, 1:, :] - img[:, :, :-1, :]
    dy = img[:, 1:, :, :] - img[:, :-1, :, :]
    return (dx ** 2).mean() + (dy ** 2).mean()


def total_loss(net_out, style_targets, content_targets, vgg_style, vgg_content,
               style_weight, content_weight, total_variation_weight):
    """model_utilities.py:391-399; the style term is divided by
    len(vgg_content_outputs), not by the number of style layers."""
    style = sum_mse(vgg_style, style_targets) * style_weight / len(vgg_content)
    content = sum_mse(vgg_content, content_targets) * content_weight
    tv = total_variation_loss(net_out) * total_variation_weight
    return style + content + tv, style, content, tv


# --------------------------------------------------------------------------
# VGG: vgg.py:11-17, :60-108, model_utilities.py:162-167
# --------------------------------------------------------------------------
VGG_LAYER_NAMES = [
    'conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
    'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2', 'pool2',
    'conv3_1', 'relu3_1', 'conv3_2', 'relu3_2', 'conv3_3', 'relu3_3', 'conv3_4', 'relu3_4', 'pool3',
    'conv4_1', 'relu4_1', 'conv4_2', 'relu4_2', 'conv4_3', 'relu4_3', 'conv4_4', 'relu4_4', 'pool4',
    'conv5_1', 'relu5_1', 'conv5_2', 'relu5_2', 'conv5_3', 'relu5_3', 'conv5_4', 'relu5_4',
]
STYLE_TAPS_CONV = ['conv1_1', 'conv2_1', 'conv3_1', 'conv4_1', 'conv5_1']
CONTENT_TAPS_CONV = ['conv5_2']
STYLE_TAPS_RELU = ['relu1_1', 'relu2_1', 'relu3_1', 'relu4_1', 'relu5_1']
CONTENT_TAPS_RELU = ['relu5_2']


def vgg_preprocess(x, mean):
    """x*255/2 - mean (vgg.py:11-17)."""
    return x * 255.0 / 2.0 - mean


def vgg_forward(x, weights, last='conv5_2'):
    """weights: dict conv name -> (W [3,3,Cin,Cout], b [Cout]) already in the
    orientation vgg.py:92-95 feeds to conv2d.  Returns dict name -> activation."""
    acts = {}
    h = x
    for name in VGG_LAYER_NAMES:
        if name.startswith('conv'):
            w, b = weights[name]
            h = conv2d_same(h, w, 1, bias=b)
        elif name.startswith('relu'):
            h = torch.relu(h)
        else:
            h = max_pool_2x2_same(h)
        acts[name] = h
        if name == last:
            break
    return acts


def vgg_taps(x, weights, relu_taps=False):
    """(style Gram list, content list) as vgg.py:104-106 (conv taps, training
    path) or vgg.py:53-55 (relu taps, single-network builder)."""
    if relu_taps:
        acts = vgg_forward(x, weights, last='relu5_4')
        s_names, c_names = STYLE_TAPS_RELU, CONTENT_TAPS_RELU
    else:
        acts = vgg_forward(x, weights, last='conv5_2')
        s_names, c_names = STYLE_TAPS_CONV, CONTENT_TAPS_CONV
    return [gram_matrix(acts[n]) for n in s_names], [acts[n] for n in c_names]


# --------------------------------------------------------------------------
# stylizer: full_graph_stylizer_network.py:42-82, model_utilities.py:235-279
# --------------------------------------------------------------------------
# (kind, Cout, k, stride); order = variable creation order
STYLIZER_LAYERS = (
    [('conv', 32, 9, 1), ('conv', 64, 3, 2), ('conv', 128, 3, 2)]
    + [('res', 128, 3, 1)] * 5
    + [('deconv', 64, 3, 2), ('deconv', 32, 3, 2), ('final', 3, 9, 1)]
)


def stylizer_param_shapes(cin=3):
    """48 shapes in creation order: per conv W, IN shift, IN scale
    (model_utilities.py:236-239, :246-250, :284-285)."""
    shapes = []
    c = cin
    for kind, co, k, _ in STYLIZER_LAYERS:
        if kind == 'res':
            for _ in range(2):
                shapes += [(k, k, c, co), (co,), (co,)]
        elif kind == 'deconv':
            shapes += [(k, k, co, c), (co,), (co,)]
        else:
            shapes += [(k, k, c, co), (co,), (co,)]
        c = co
    return shapes


def truncated_normal(shape, stddev, rng):
    """NumPy stand-in for tf.truncated_normal (model_utilities.py:231-232):
    normal(0, stddev) with samples beyond 2 sigma redrawn.  Not bit-equal to
    TensorFlow's generator; oracle and kernels consume the same arrays."""
    import numpy as np
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * stddev).astype(np.float32)


def init_stylizer_params(seed=1, cin=3):
    import numpy as np
    rng = np.random.default_rng(seed)
    params = []
    for i, shp in enumerate(stylizer_param_shapes(cin)):
        if len(shp) == 4:
            params.append(truncated_normal(shp, 0.1, rng))
        elif i % 3 == 1:
            params.append(np.zeros(shp, np.float32))   # shift
        else:
            params.append(np.ones(shp, np.float32))    # scale
    return params


def stylizer_forward(x, params, collect=None):
    """params: list of 48 tensors in creation order.  ``collect`` (a list) is
    filled with the 16 post-block activations."""
    it = iter(params)

    def block(h, kind, stride, act):
        w, shift, scale = next(it), next(it), next(it)
        if kind == 'deconv':
            h = conv2d_transpose_same(h, w, stride)
        else:
            h = conv2d_same(h, w, stride)
        h = instance_norm(h, scale, shift)
        if act == 'relu':
            h = torch.relu(h)
        elif act == 'sigmoid':
            h = torch.sigmoid(h)
        if collect is not None:
            collect.append(h)
        return h

    h = x
    for kind, _co, _k, s in STYLIZER_LAYERS:
        if kind == 'res':
            r = block(h, 'conv', s, 'relu')
            r = block(r, 'conv', s, None)
            h = h + r
            if collect is not None:
                collect[-1] = h
        elif kind == 'final':
            h = block(h, 'conv', s, 'sigmoid')
        else:
            h = block(h, kind, s, 'relu')
    return h


# --------------------------------------------------------------------------
# TF1 Adam: tf.train.AdamOptimizer (call sites full_graph_stylizer_network.py:125, :278)
# --------------------------------------------------------------------------
def adam_tf1_step(params, grads, ms, vs, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place TF1 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); epsilon is added to
    sqrt(v) without bias correction."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for p, g, m, v in zip(params, grads, ms, vs):
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        p.sub_(lr_t * m / (v.sqrt() + eps))


# --------------------------------------------------------------------------
# whole steps: full_graph_stylizer_network.py:94-129, :185-216 and :257-279, :308-329
# --------------------------------------------------------------------------
def train_step_loss(batch, params, vgg_weights, style_targets, mean,
                    style_weight, content_weight, tv_weight):
    """One training-step loss exactly as the session runs it: content targets
    from the VGG replica fed with the batch, then total_loss on the stylizer
    output's VGG replica.  Returns (loss, style, content, tv, stylized)."""
    with torch.no_grad():
        _, content_targets = vgg_taps(vgg_preprocess(batch, mean), vgg_weights)
    out = stylizer_forward(batch, params)
    style_l, content_l = vgg_taps(vgg_preprocess(out, mean), vgg_weights)
    loss, s, c, tv = total_loss(out, style_targets, content_targets, style_l, content_l,
                                style_weight, content_weight, tv_weight)
    return loss, s, c, tv, out


def gatys_loss(image, vgg_weights, style_targets, content_targets, mean,
               style_weight, content_weight, tv_weight):
    style_l, content_l = vgg_taps(vgg_preprocess(image, mean), vgg_weights)
    return total_loss(image, style_targets, content_targets, style_l, content_l,
                      style_weight, content_weight, tv_weight)


# --------------------------------------------------------------------------
# synthetic VGG weights (the real .mat is not redistributable / not available offline)
# --------------------------------------------------------------------------
VGG_CHANNELS = {
    'conv1_1': (3, 64), 'conv1_2': (64, 64),
    'conv2_1': (64, 128), 'conv2_2': (128, 128),
    'conv3_1': (128, 256), 'conv3_2': (256, 256), 'conv3_3': (256, 256), 'conv3_4': (256, 256),
    'conv4_1': (256, 512), 'conv4_2': (512, 512), 'conv4_3': (512, 512), 'conv4_4': (512, 512),
    'conv5_1': (512, 512), 'conv5_2': (512, 512), 'conv5_3': (512, 512), 'conv5_4': (512, 512),
}


def synthetic_vgg_weights(seed=7, width_div=1, bias_scale=0.0, dtype='float32'):
    """He-normal 3x3 kernels with the VGG19 channel plan divided by ``width_div``
    (minimum 4, conv1_1 keeps 3 inputs).  Returned in the orientation conv2d
    consumes, i.e. after vgg.py:94's transpose."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = {}
    for name in [n for n in VGG_LAYER_NAMES if n.startswith('conv')]:
        ci, co = VGG_CHANNELS[name]
        ci = ci if ci == 3 else max(ci // width_div, 4)
        co = max(co // width_div, 4)
        w = rng.standard_normal((3, 3, ci, co)) * math.sqrt(2.0 / (9 * ci))
        b = rng.standard_normal((co,)) * bias_scale
        out[name] = (w.astype(dtype), b.astype(dtype))
    return out
