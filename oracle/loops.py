"""Loop-level NumPy restatement of the TensorFlow op semantics, for small cases.

Test infrastructure only (see ``oracle/__init__.py``); parity unpinned.

These are written straight from the index formulas, independently of
``oracle/tf_ops.py`` (which leans on torch's convolution routines), so that the
two restatements pin each other: SAME padding with pad_before = total // 2,
conv2d_transpose as the adjoint of the SAME conv, max-pool windows clipped at
an odd edge, biased instance-norm variance.
"""
import numpy as np


def same_pads(in_size, k, stride):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k - in_size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, stride):
    """out[n,oy,ox,co] = sum in[n, oy*s - pt + ky, ox*s - pl + kx, ci] * w[ky,kx,ci,co]
    (tf.nn.conv2d 'SAME'; reference call site model_utilities.py:236-237)."""
    n, h, wd, _ = x.shape
    kh, kw, _, co = w.shape
    ho, pt, _ = same_pads(h, kh, stride)
    wo, pl, _ = same_pads(wd, kw, stride)
    out = np.zeros((n, ho, wo, co), np.float64)
    for oy in range(ho):
        for ox in range(wo):
            for ky in range(kh):
                iy = oy * stride - pt + ky
                if iy < 0 or iy >= h:
                    continue
                for kx in range(kw):
                    ix = ox * stride - pl + kx
                    if ix < 0 or ix >= wd:
                        continue
                    out[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return out


def conv2d_transpose_same(x, w, stride):
    """out[n,y,x,co] = sum in[n,i,j,ci] * w[ky,kx,co,ci] over i*s - pt + ky == y,
    j*s - pl + kx == x, with pt/pl the SAME pads of the forward conv on the
    s*H x s*W output (tf.nn.conv2d_transpose; model_utilities.py:246-248)."""
    n, h, wd, _ = x.shape
    kh, kw, co, _ = w.shape
    ho, wo = h * stride, wd * stride
    _, pt, _ = same_pads(ho, kh, stride)
    _, pl, _ = same_pads(wo, kw, stride)
    out = np.zeros((n, ho, wo, co), np.float64)
    for i in range(h):
        for j in range(wd):
            for ky in range(kh):
                y = i * stride - pt + ky
                if y < 0 or y >= ho:
                    continue
                for kx in range(kw):
                    xx = j * stride - pl + kx
                    if xx < 0 or xx >= wo:
                        continue
                    out[:, y, xx, :] += x[:, i, j, :].astype(np.float64) @ w[ky, kx].astype(np.float64).T
    return out


def max_pool_2x2_same(x):
    """tf.nn.max_pool2d(2, 2, 'SAME') (model_utilities.py:166-167)."""
    n, h, w, c = x.shape
    ho, wo = -(-h // 2), -(-w // 2)
    out = np.empty((n, ho, wo, c), x.dtype)
    for oy in range(ho):
        for ox in range(wo):
            win = x[:, 2 * oy:min(2 * oy + 2, h), 2 * ox:min(2 * ox + 2, w), :]
            out[:, oy, ox, :] = win.max(axis=(1, 2))
    return out


def instance_norm(x, scale, shift, eps=1e-3):
    """model_utilities.py:282-288."""
    x = x.astype(np.float64)
    mu = x.mean(axis=(1, 2), keepdims=True)
    var = x.var(axis=(1, 2), keepdims=True)  # population variance
    return scale * (x - mu) / np.sqrt(var + eps) + shift


def gram_matrix(f):
    """model_utilities.py:380-384."""
    b, h, w, c = f.shape
    out = np.zeros((b, c, c), np.float64)
    for n in range(b):
        m = f[n].reshape(h * w, c).astype(np.float64)
        out[n] = m.T @ m / float(h * w)
    return out


def total_variation_loss(img):
    """model_utilities.py:369-377."""
    img = img.astype(np.float64)
    b, h, w, c = img.shape
    sx = 0.0
    sy = 0.0
    for y in range(h):
        for x in range(w):
            if x + 1 < w:
                sx += ((img[:, y, x + 1, :] - img[:, y, x, :]) ** 2).sum()
            if y + 1 < h:
                sy += ((img[:, y + 1, x, :] - img[:, y, x, :]) ** 2).sum()
    return sx / (b * h * (w - 1) * c) + sy / (b * (h - 1) * w * c)


def adam_tf1(p, g, m, v, t, lr, b1, b2, eps):
    """tf.train.AdamOptimizer's update, epsilon outside the bias correction."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    return p - lr_t * m / (np.sqrt(v) + eps), m, v
