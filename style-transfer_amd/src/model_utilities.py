"""Drop-in for the reference's src/model_utilities.py on the MI355X path.

Same names, argument order and defaults as the reference for everything the
stylizer path uses -- LayerConfig, StyleConfig, NNGraphModelBuilder,
GraphModelBuilder.conv_layer_from_weights / pool_layer, StyleTransfer (layer
name lists, STYLE_CONFIG_DICT, loss functions), get_most_recent_model_name,
create_dir_if_not_exists -- but the functions run eagerly on NHWC torch tensors
that live on the GPU and launch the HIP kernels of libstmi355.so instead of
building a TensorFlow graph.  ``nn.relu`` / ``nn.sigmoid`` stand in for
``tf.nn.relu`` / ``tf.nn.sigmoid``.

Variables: the reference creates unnamed tf.Variables in call order inside
``tf.variable_scope('style_network')``.  Here a VariableStore plays that role:
the first pass through the builder creates the tensors (W, then IN shift, then
IN scale per block, model_utilities.py:236-239, :284-285), later passes inside
``variable_scope(store)`` reuse them by position.
"""
import contextlib
import os

import numpy as np
import torch

import constants
from image_utilities import sort_numerical
from stmi import ops


# --------------------------------------------------------------------------
# tf.nn stand-ins
# --------------------------------------------------------------------------
class nn:
    @staticmethod
    def relu(x):
        return ops.activation(x, ops.ACT_RELU)

    @staticmethod
    def sigmoid(x):
        return ops.activation(x, ops.ACT_SIGMOID)


_FUSABLE = {None: ops.ACT_NONE, nn.relu: ops.ACT_RELU, nn.sigmoid: ops.ACT_SIGMOID}


# --------------------------------------------------------------------------
# configuration records (reference model_utilities.py:11-47)
# --------------------------------------------------------------------------
class LayerConfig:
    def __init__(self, num_filters, filter_size, stride, padding='same', activation=nn.relu, use_batch_norm=False,
                 use_instance_norm=True):
        self.num_filters = num_filters
        self.filter_size = filter_size
        self.stride = stride
        self.padding = padding
        self.activation = activation
        self.use_batch_norm = use_batch_norm
        self.use_instance_norm = use_instance_norm

    def copy(self):
        return LayerConfig(self.num_filters, self.filter_size, self.stride, self.padding, self.activation,
                           self.use_batch_norm, self.use_instance_norm)

    def __repr__(self):
        return '<LayerConfig %d, %dx%d, %d stride, %s pad, %s activation, %r bn, %r in>' % (
            self.num_filters, self.filter_size, self.filter_size, self.stride, self.padding, str(self.activation),
            self.use_batch_norm, self.use_instance_norm)


class StyleConfig:
    """style image path + the three loss weights of one named experiment"""

    def __init__(self, style_path, style_weight, content_weight, var_weight):
        self.style_path = style_path
        self.style_weight = style_weight
        self.content_weight = content_weight
        self.total_variation_weight = var_weight

    def __repr__(self):
        return '<StyleConfig ' + str({'style_weight': self.style_weight, 'content_weight': self.content_weight,
                                      'var_weight': self.total_variation_weight, 'style_path': self.style_path}) + '>'


# --------------------------------------------------------------------------
# model directory helpers (reference model_utilities.py:92-102)
# --------------------------------------------------------------------------
def get_most_recent_model_name(dir, prefix):
    models = [x for x in os.listdir(dir) if prefix in x]
    sort_numerical(models)
    return models[-1]


def create_dir_if_not_exists(dir):
    if not os.path.exists(dir):
        os.makedirs(dir)


# --------------------------------------------------------------------------
# variables
# --------------------------------------------------------------------------
class VariableStore:
    """Ordered fp32 variables of one scope (the reference's 'style_network/Variable[_k]')."""

    def __init__(self, device=None, seed=1, compute_dtype=torch.float32):
        self.device = device
        self.compute_dtype = compute_dtype
        self.variables = []
        self._cursor = 0
        self._rng = np.random.default_rng(seed)

    def rewind(self):
        self._cursor = 0

    def next(self, shape, init):
        if self._cursor < len(self.variables):
            v = self.variables[self._cursor]
            if tuple(v.shape) != tuple(shape):
                raise ValueError('variable %d has shape %s, the graph asks for %s'
                                 % (self._cursor, tuple(v.shape), tuple(shape)))
        else:
            v = torch.from_numpy(init(shape, self._rng)).to(self.device)
            self.variables.append(v)
        self._cursor += 1
        return v

    def load(self, arrays):
        self.variables = [torch.as_tensor(np.asarray(a), dtype=torch.float32).to(self.device).contiguous()
                          for a in arrays]
        self._cursor = 0

    def export(self):
        return [v.detach().cpu().numpy() for v in self.variables]


_default_store = None
_active_store = None


@contextlib.contextmanager
def variable_scope(store):
    """Makes `store` the one var_init draws from and rewinds it (positional reuse)."""
    global _active_store
    prev = _active_store
    _active_store = store
    store.rewind()
    try:
        yield store
    finally:
        _active_store = prev


def _store(device):
    global _default_store
    if _active_store is not None:
        if _active_store.device is None:
            _active_store.device = device
        return _active_store
    if _default_store is None:
        _default_store = VariableStore(device)
    return _default_store


def _truncated_normal(shape, rng):
    """tf.truncated_normal(shape, stddev=.1): samples beyond 2 sigma are redrawn."""
    a = rng.standard_normal(shape)
    bad = np.abs(a) > 2.0
    while bad.any():
        a[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(a) > 2.0
    return (a * 0.1).astype(np.float32)


def _as_compute(x, store):
    """Builder inputs may be fp32 images; activations flow in the store's compute dtype."""
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise ops._lib.StmiError('the stylizer kernels run on an MI355X; no GPU is visible and there is no CPU path')
        x = x.cuda()
    x = x.contiguous()
    if x.dtype != store.compute_dtype:
        x = ops.cast(x.float() if x.dtype not in (torch.float32, store.compute_dtype) else x, store.compute_dtype)
    return x


class ModelBuilder:
    @staticmethod
    def conv_block(layer_input, config: LayerConfig):
        return

    @staticmethod
    def deconv_block(layer_input, config: LayerConfig):
        return

    @staticmethod
    def residual_block(layer_input, config: LayerConfig):
        return


class GraphModelBuilder(ModelBuilder):
    """Only the two VGG helpers of the reference class are on the stylizer path
    (model_utilities.py:162-167); the tf.layers based blocks belong to superseded scripts."""

    @staticmethod
    def conv_layer_from_weights(layer_input, weights, bias, relu=False, want_pre=False):
        """bias_add(conv2d(x, const W, stride 1, SAME), b); W is HWIO."""
        x = layer_input
        w = torch.as_tensor(np.asarray(weights), dtype=torch.float32).to(x.device).contiguous()
        b = torch.as_tensor(np.asarray(bias), dtype=torch.float32).reshape(-1).to(x.device).contiguous()
        wp = ops.pack_conv_weights(w, ops.PACK_FWD, x.dtype)
        return ops.conv2d_fwd(x, wp, tuple(w.shape), 1, bias=b, relu=relu, want_pre=want_pre)

    @staticmethod
    def pool_layer(layer_input):
        return ops.maxpool_fwd(layer_input)


class NNGraphModelBuilder(ModelBuilder):
    @staticmethod
    def var_init(shape, device=None):
        return _store(device).next(tuple(int(s) for s in shape), _truncated_normal)

    @staticmethod
    def _finish(conv_out, config, residual=None):
        """instance norm -> activation (-> skip add), fused into one kernel pair when possible"""
        act = config.activation
        if config.use_instance_norm:
            if act in _FUSABLE:
                return NNGraphModelBuilder.instance_norm(conv_out, _act=_FUSABLE[act], _residual=residual)
            out = act(NNGraphModelBuilder.instance_norm(conv_out))
        else:
            out = conv_out if act is None else act(conv_out)
        if residual is not None:
            raise NotImplementedError('a skip add is only fused behind an instance norm')
        return out

    @staticmethod
    def conv_block(layer_input, config: LayerConfig, _residual=None):
        store = _store(layer_input.device if torch.is_tensor(layer_input) else None)
        x = _as_compute(layer_input, store)
        k, cin = config.filter_size, x.shape[3]
        if config.padding.upper() != 'SAME':
            raise ValueError("only 'same' padding is implemented (the reference uses nothing else)")
        weights = NNGraphModelBuilder.var_init([k, k, cin, config.num_filters], x.device)
        wp = ops.pack_conv_weights(weights, ops.PACK_FWD, x.dtype)
        conv = ops.conv2d_fwd(x, wp, (k, k, cin, config.num_filters), config.stride)
        return NNGraphModelBuilder._finish(conv, config, _residual)

    @staticmethod
    def deconv_block(layer_input, config: LayerConfig):
        store = _store(layer_input.device if torch.is_tensor(layer_input) else None)
        x = _as_compute(layer_input, store)
        k, cin = config.filter_size, x.shape[3]
        weights = NNGraphModelBuilder.var_init([k, k, config.num_filters, cin], x.device)
        wp = ops.pack_conv_weights(weights, ops.PACK_DGRAD, x.dtype)
        deconv = ops.conv2d_transpose_fwd(x, wp, (k, k, config.num_filters, cin), config.stride)
        return NNGraphModelBuilder._finish(deconv, config)

    @staticmethod
    def residual_block(layer_input, config: LayerConfig):
        """conv-IN-act, conv-IN, add; nothing after the add (model_utilities.py:271-279)"""
        conv_config = config.copy()
        conv_config.use_batch_norm = False
        conv_config.use_instance_norm = True
        conv_config.activation = None
        store = _store(layer_input.device if torch.is_tensor(layer_input) else None)
        x = _as_compute(layer_input, store)
        residual_layer = NNGraphModelBuilder.conv_block(x, config)
        return NNGraphModelBuilder.conv_block(residual_layer, conv_config, _residual=x)

    @staticmethod
    def instance_norm(layer_input, _act=ops.ACT_NONE, _residual=None):
        channels = layer_input.shape[3]
        shift = _store(layer_input.device).next((channels,), lambda s, r: np.zeros(s, np.float32))
        scale = _store(layer_input.device).next((channels,), lambda s, r: np.ones(s, np.float32))
        y, _stats = ops.instance_norm_fwd(layer_input, scale, shift, _act, residual=_residual)
        return y


# --------------------------------------------------------------------------
# losses and the named experiments (reference model_utilities.py:291-399)
# --------------------------------------------------------------------------
def _style_table():
    S = constants.STYLES_DIR
    rows = [
        ('starry_night_transfer', 'starry_night.jpg', 1e-2, 1e4, 1e8),
        ('starry_night_style', 'starry_night.jpg', 1, 0, 0),
        ('starry_night_content', 'starry_night.jpg', 0, 1, 0),
        ('honeycomb_transfer', 'honeycomb_squeeze.jpg', 1e-2, 1e4, 1e8),
        ('honeycomb_1', 'honeycomb_squeeze.jpg', 1e-4, 1e4, 1e8),
        ('honeycomb_2', 'honeycomb_squeeze.jpg', 1e-3, 1e4, 1e8),
        ('heiro_transfer', 'heiro.jpg', 1e-2, 1e4, 1e8),
        ('heiro_1', 'heiro.jpg', 1e-3, 1e4, 1e8),
        ('heiro_2', 'heiro.jpg', 5e-3, 1e4, 1e8),
        ('heiro_3', 'heiro.jpg', 2e-3, 1e4, 1e8),
        ('heiro_4', 'heiro.jpg', 1.5e-3, 1e4, 1e8),
        ('heiro_alt_transfer', 'heiro_alt.jpg', 1e-2, 1e4, 1e8),
        ('heiro_alt_1', 'heiro_alt.jpg', 5e-3, 1e4, 1e8),
        ('heiro_alt_2', 'heiro_alt.jpg', 1e-3, 1e4, 1e8),
        ('gears_transfer', 'gears.jpg', 1e-2, 1e4, 1e8),
        ('gears_1', 'gears.jpg', 1e-1, 1e4, 1e8),
        ('rain_transfer', 'rain.jpg', 1e-2, 1e4, 1e8),
        ('rain_1', 'rain.jpg', 1e-3, 1e4, 1e8),
        ('rain_2', 'rain.jpg', 5e-3, 1e4, 1e8),
        ('wave_transfer', 'wave.jpg', 1e-2, 1e4, 1e8),
        ('wave_1', 'wave.jpg', 1e-3, 1e4, 1e8),
        ('wave_2', 'wave.jpg', 5e-3, 1e4, 1e8),
        ('heiro_shadow_1', 'heiro_shadow.jpg', 5e-3, 1e4, 1e8),
        ('heiro_shadow_2', 'heiro_shadow.jpg', 1e-2, 1e4, 1e8),
        ('heiro_shadow_3', 'heiro_shadow.jpg', 5e-2, 1e4, 1e8),
    ]
    return {name: StyleConfig(os.path.join(S, img), sw, cw, tv) for name, img, sw, cw, tv in rows}


class StyleTransfer:
    STYLE_WEIGHT = 1e-2
    TOTAL_VARIATION_WEIGHT = 1e-4
    CONTENT_WEIGHT = 1e8
    VGG_CONTENT_TARGET_LAYER_NAMES = ['block5_conv2']
    VGG_STYLE_TARGET_LAYER_NAMES = ['block%d_conv1' % i for i in range(1, 6)]
    VGG_CONTENT_TARGET_LAYER_NAMES_ALT = ['conv5_2']
    VGG_STYLE_TARGET_LAYER_NAMES_ALT = ['conv%d_1' % i for i in range(1, 6)]
    VGG_CONTENT_TARGET_LAYER_NAMES_ALT_ACT = ['relu5_2']
    VGG_STYLE_TARGET_LAYER_NAMES_ALT_ACT = ['relu%d_1' % i for i in range(1, 6)]

    STYLE_CONFIG_DICT = _style_table()

    @staticmethod
    def clip_0_1(image):
        return ops.clip_(image.clone(), 0.0, 1.0)

    @staticmethod
    def high_pass_x_y(image):
        """Neighbour differences as tensors.  A convenience for callers that want the maps
        themselves; the loss below does not materialise them."""
        return image[:, :, 1:, :] - image[:, :, :-1, :], image[:, 1:, :, :] - image[:, :-1, :, :]

    @staticmethod
    def total_variation_loss(image):
        slot = ops.loss_slots(1, image.device)[0]
        img32 = image if image.dtype == torch.float32 else ops.cast(image, torch.float32)
        ops.tv_loss_fwd_bwd(img32.contiguous(), slot, 1.0, torch.float32)
        return ops.slot_total(slot)[0]

    @staticmethod
    def gram_matrix(input_tensor):
        return ops.gram_fwd(input_tensor.contiguous())

    @staticmethod
    def sum_mse(array_1, array_2):
        """sum_i mean((a_i - b_i)^2); b_i may broadcast over the batch ([1,C,C] style targets)."""
        total = None
        for a, b in zip(array_1, array_2):
            loss = ops.loss_slots(1, a.device)[0]
            if a.dim() == 3 and a.dtype == torch.float32 and b.numel() == a.shape[1] * a.shape[2]:
                n, c, _ = a.shape
                ops.gram_mse_fwd_bwd(a.contiguous(), b.to(a.device, torch.float32).contiguous(), loss,
                                     1.0 / (n * c * c), 0.0, torch.float32)
            else:
                if tuple(a.shape) != tuple(b.shape):
                    b = b.expand_as(a)
                ops.mse_fwd_bwd(a.contiguous(), b.to(a.device, a.dtype).contiguous(), loss, 1.0 / a.numel(), 0.0,
                                want_grad=False)
            term = ops.slot_total(loss)[0]
            total = term if total is None else total + term
        return total

    @staticmethod
    def total_loss(style_network_outputs, style_targets, content_targets, vgg_style_outputs, vgg_content_outputs,
                   style_weight=STYLE_WEIGHT, content_weight=CONTENT_WEIGHT,
                   total_variation_weight=TOTAL_VARIATION_WEIGHT):
        style = StyleTransfer.sum_mse(vgg_style_outputs, style_targets) * style_weight / len(vgg_content_outputs)
        content = StyleTransfer.sum_mse(vgg_content_outputs, content_targets) * content_weight
        tv = StyleTransfer.total_variation_loss(style_network_outputs) * total_variation_weight
        return style + content + tv, style, content, tv
