"""Directory layout of the drop-in (same names as the reference's src/constants.py:6-23).

The tree is rooted one level above this file, exactly as in the reference
(``HERE/../models`` ...); set STMI_DATA_ROOT to put it somewhere else.  As in the
reference, importing the module creates every ``*_DIR`` that does not exist yet
(constants.py:31-41).
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.environ.get('STMI_DATA_ROOT', os.path.join(HERE, '..'))

TENSORBOARD_DIR = os.path.join(_ROOT, 'tensorboard_dir/')
TRAINING_IMAGES_DIR = os.path.join(_ROOT, 'res/training_images')
MODELS_DIR = os.path.join(_ROOT, 'models')
TEST_DIR = os.path.join(_ROOT, 'res/test')
STYLES_DIR = os.path.join(_ROOT, 'res/styles')
CONTENT_DIR = os.path.join(_ROOT, 'res/content')

VAE_KERAS_GENERATED_SAMPLES_DIR = os.path.join(TEST_DIR, 'keras_generated_samples')
STYLIZED_IMAGES_DIR = os.path.join(TEST_DIR, 'stylized_images')
STYLE_TRANSFER_IMAGES_DIR = os.path.join(TEST_DIR, 'style_transfer')
VAE_KERAS_GENERATED_TRAINING_IMAGES_DIR = os.path.join(TRAINING_IMAGES_DIR, 'vae_keras_generated')
STYLIZER_NETWORK_INPUT_DIR = os.path.join(TEST_DIR, 'style_network_input')
STYLIZER_NETWORK_OUTPUT_DIR = os.path.join(TEST_DIR, 'style_network_output')


def make_dirs_if_not_exist(dirs):
    for d in dirs:
        os.makedirs(d, exist_ok=True)


def init_dirs():
    make_dirs_if_not_exist([v for k, v in globals().items() if k.endswith('_DIR') and isinstance(v, str)])


init_dirs()
