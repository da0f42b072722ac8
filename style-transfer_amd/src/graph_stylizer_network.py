"""Import target for callers of the reference's src/graph_stylizer_network.py.

The Django view imports this module next to full_graph_stylizer_network
(ml_services/style_net_service/views.py:17) and refers to its ``model_prefix``
(views.py:22); it never builds the network this module defined in the
reference, which was the earlier Keras-VGG variant of the same stylizer (three
transposed convolutions, a VGG19 fetched by keras.applications at import).
That variant is not built here.  The names a caller can reach resolve to the
one stylizer this package implements, so a service started through either
module serves the same model.
"""
import os

import constants
import full_graph_stylizer_network as _full
from full_graph_stylizer_network import StyleNetService  # noqa: F401
from image_utilities import load_image, CocoDatasetManager, save_image, PreprocessedCocoDatasetManager  # noqa: F401
from model_utilities import LayerConfig, StyleTransfer  # noqa: F401

# hyper-parameters of the reference module (graph_stylizer_network.py:18-21, :33)
LEARNING_RATE = 1e-3
EPOCHS = 30
BATCH_SIZE = 4
TOTAL_IMAGES = 500
model_prefix = 'STYLE_NET'

CONTENT_IMAGE_1 = os.path.join(constants.CONTENT_DIR, 'antelope_small.jpg')


def run_on_image(source_image_path, destination_path, model_path=None):
    """Three-argument form of the reference module (graph_stylizer_network.py:357)."""
    return _full.run_on_image(source_image_path, destination_path, model_path=model_path)


def train_style_network(style_type, test_content_image_path=CONTENT_IMAGE_1):
    raise NotImplementedError('the Keras-VGG stylizer variant is not built; '
                              'use full_graph_stylizer_network.train_style_network')


def normal_style_transfer(style_type):
    raise NotImplementedError('the Keras-VGG variant is not built; '
                              'use full_graph_stylizer_network.normal_style_transfer')
