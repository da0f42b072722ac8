"""Drop-in for the reference's src/vgg.py: imagenet-vgg-verydeep-19.mat loader and the
two VGG19 builders, evaluated eagerly with the HIP kernels.

Kept behaviours (each cited at its function):
  * the preprocessing mean is a MODULE GLOBAL that starts at [0,0,0] and is only set
    when a builder loads the .mat -- and vgg_preprocess_input reads it at call time,
    so a caller that preprocesses BEFORE building (as the reference's training script
    does) gets mean 0;
  * .mat indexing, the (1,0,2,3) kernel transpose and bias flattening;
  * the replicate builder stops at conv5_2 and taps PRE-ReLU conv outputs, the single
    builder runs to relu5_4 and taps ReLU outputs.
"""
import os

import numpy as np
import scipy.io
import torch

from model_utilities import GraphModelBuilder, StyleTransfer
from stmi import ops

HERE = os.path.dirname(os.path.abspath(__file__))

VGG_PATH = os.path.join(HERE, '../res/vgg19/imagenet-vgg-verydeep-19.mat')
# the replicate builder of the reference resolves a different relative path (vgg.py:67)
VGG_PATH_REPLICATE = os.path.join(HERE, '../../../res/vgg19/imagenet-vgg-verydeep-19.mat')

vgg_preprocessing_mean = np.array([0, 0, 0])

# eager tensors flow in this dtype; set STMI_DTYPE=bf16 for the bf16-MFMA mode
COMPUTE_DTYPE = torch.bfloat16 if os.environ.get('STMI_DTYPE', 'f32') == 'bf16' else torch.float32

VGG_LAYER_NAMES = [
    'conv1_1', 'relu1_1', 'conv1_2', 'relu1_2', 'pool1',
    'conv2_1', 'relu2_1', 'conv2_2', 'relu2_2', 'pool2',
    'conv3_1', 'relu3_1', 'conv3_2', 'relu3_2', 'conv3_3', 'relu3_3', 'conv3_4', 'relu3_4', 'pool3',
    'conv4_1', 'relu4_1', 'conv4_2', 'relu4_2', 'conv4_3', 'relu4_3', 'conv4_4', 'relu4_4', 'pool4',
    'conv5_1', 'relu5_1', 'conv5_2', 'relu5_2', 'conv5_3', 'relu5_3', 'conv5_4', 'relu5_4',
]


def vgg_preprocess_input(network_input):
    """x*255/2 - mean, with the mean read from the module global NOW (reference vgg.py:11-17)."""
    x = network_input
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    if not x.is_cuda:
        if not torch.cuda.is_available():
            raise ops._lib.StmiError('VGG preprocessing runs on the GPU; none is visible and there is no CPU path')
        x = x.cuda()
    x32 = x.contiguous() if x.dtype == torch.float32 else ops.cast(x.contiguous(), torch.float32)
    mean = torch.as_tensor(np.asarray(vgg_preprocessing_mean, dtype=np.float32)).to(x32.device)
    return ops.scale_shift(x32, mean, 255. / 2., COMPUTE_DTYPE)


def load_vgg_mat(path):
    """(weights, mean_pixel): weights maps conv name -> (HWIO kernels, bias) exactly as the
    reference hands them to conv2d (vgg.py:43-45, :92-95); mean_pixel as vgg.py:25-27."""
    print('Loading VGG19 data...')
    vgg_data = scipy.io.loadmat(path)
    print('VGG19 data loaded.')
    mean_pixel = np.mean(vgg_data['normalization'][0][0][0], axis=(0, 1))
    layer_constants = vgg_data['layers'][0]
    weights = {}
    for layer, name in enumerate(VGG_LAYER_NAMES):
        if 'conv' in name and layer < len(layer_constants):
            kernels, bias = layer_constants[layer][0][0][0][0]
            weights[name] = (np.ascontiguousarray(np.transpose(kernels, (1, 0, 2, 3)), dtype=np.float32),
                             np.asarray(bias, dtype=np.float32).reshape(-1))
    return weights, mean_pixel


def save_vgg_mat(path, weights, mean_image=None, names=None):
    """Writes a .mat with the nesting load_vgg_mat (and the reference) index: used to make
    synthetic stand-ins for the original file, which is not redistributable."""
    names = VGG_LAYER_NAMES if names is None else names
    layers = np.empty((1, len(names)), dtype=object)
    for i, name in enumerate(names):
        if name in weights:
            w, b = weights[name]
            cell = np.empty((1, 2), dtype=object)
            cell[0, 0] = np.transpose(np.asarray(w, np.float32), (1, 0, 2, 3))   # stored [kw,kh,ci,co]
            cell[0, 1] = np.asarray(b, np.float32).reshape(1, -1)
            entry = np.zeros((1, 1), dtype=[('weights', object), ('type', object), ('name', object)])
            entry[0, 0]['weights'] = cell
            entry[0, 0]['type'] = 'conv'
            entry[0, 0]['name'] = name
        else:
            entry = np.zeros((1, 1), dtype=[('type', object), ('name', object)])
            entry[0, 0]['type'] = 'relu' if 'relu' in name else 'pool'
            entry[0, 0]['name'] = name
        layers[0, i] = entry
    if mean_image is None:
        mean_image = np.zeros((224, 224, 3), np.float32)
    norm = np.zeros((1, 1), dtype=[('averageImage', object)])
    norm[0, 0]['averageImage'] = np.asarray(mean_image, np.float32)
    scipy.io.savemat(path, {'layers': layers, 'normalization': norm})


def _run(network_input, weights, names):
    """One replica: dict layer name -> activation.  Convs that feed a ReLU are launched once
    with the ReLU fused and the pre-ReLU value kept as the conv's own entry."""
    net, x, i = {}, network_input, 0
    while i < len(names):
        name = names[i]
        if 'conv' in name:
            w, b = weights[name]
            if i + 1 < len(names) and 'relu' in names[i + 1]:
                post, pre = GraphModelBuilder.conv_layer_from_weights(x, w, b, relu=True, want_pre=True)
                net[name], net[names[i + 1]] = pre, post
                x = post
                i += 2
                continue
            x = GraphModelBuilder.conv_layer_from_weights(x, w, b)
        elif 'relu' in name:
            x = ops.activation(x, ops.ACT_RELU)
        elif 'pool' in name:
            x = GraphModelBuilder.pool_layer(x)
        net[name] = x
        i += 1
    return net


def build_vgg_network_from_mat(network_input):
    """Single network through relu5_4; style = Gram(relu1_1..relu5_1), content = [relu5_2]
    (reference vgg.py:19-57)."""
    global vgg_preprocessing_mean
    weights, vgg_preprocessing_mean = load_vgg_mat(VGG_PATH)
    net = _run(network_input, weights, VGG_LAYER_NAMES)
    content_layers = [net[x] for x in StyleTransfer.VGG_CONTENT_TARGET_LAYER_NAMES_ALT_ACT]
    style_layers = [StyleTransfer.gram_matrix(net[x]) for x in StyleTransfer.VGG_STYLE_TARGET_LAYER_NAMES_ALT_ACT]
    return style_layers, content_layers


def build_vgg_network_replicate_from_mat(network_inputs, replications=1):
    """`replications` networks sharing one weight set, through conv5_2, tapping the conv
    (pre-ReLU) outputs (reference vgg.py:60-108).  Returns (style_layers, content_layers),
    each a list per replica."""
    global vgg_preprocessing_mean
    weights, vgg_preprocessing_mean = load_vgg_mat(VGG_PATH_REPLICATE if os.path.exists(VGG_PATH_REPLICATE)
                                                   else VGG_PATH)
    names = VGG_LAYER_NAMES[:VGG_LAYER_NAMES.index('conv5_2') + 1]
    nets = [_run(network_inputs[r], weights, names) for r in range(replications)]
    content_layers = [[nets[r][x] for x in StyleTransfer.VGG_CONTENT_TARGET_LAYER_NAMES_ALT]
                      for r in range(replications)]
    style_layers = [[StyleTransfer.gram_matrix(nets[r][x]) for x in StyleTransfer.VGG_STYLE_TARGET_LAYER_NAMES_ALT]
                    for r in range(replications)]
    return style_layers, content_layers
