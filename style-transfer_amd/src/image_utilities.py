"""Host-side image I/O with the reference's function names (src/image_utilities.py).

Images travel as float32 RGB in [0,1], batch dimension added by load_image.
OpenCV is used when it is importable; otherwise Pillow does the decoding and
encoding (the arithmetic around it -- BGR<->RGB, /255, uint8 truncation -- is
the reference's, image_utilities.py:189-236).
"""
import os

import numpy as np

import constants

try:  # the reference's codec
    import cv2 as _cv2
except ImportError:  # pragma: no cover - depends on the image
    _cv2 = None
    from PIL import Image as _Image

HERE = os.path.dirname(os.path.abspath(__file__))
COCO_DATASET = os.path.join(HERE, '../res/coco/coco/train2014/train2014/')
COCO_DATASET_CLIPPED = os.path.join(HERE, '../res/coco/coco_clipped')
DEBUG_MODE = False


def debug(msg):
    if DEBUG_MODE:
        print(msg)


# ---- codec layer: both return / take BGR uint8 like cv2.imread / cv2.imwrite ----
def _imread_bgr(path):
    if _cv2 is not None:
        return _cv2.imread(path)
    with _Image.open(path) as im:
        return np.asarray(im.convert('RGB'))[:, :, ::-1].copy()


def _imwrite_bgr(path, bgr):
    if _cv2 is not None:
        return _cv2.imwrite(path, bgr)
    _Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)
    return True


def imdecode_bgr(data):
    """bytes -> BGR uint8 (what the Django view does with cv2.imdecode, views.py:46-47)."""
    if _cv2 is not None:
        return _cv2.imdecode(np.frombuffer(data, np.uint8), _cv2.IMREAD_COLOR)
    import io
    with _Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert('RGB'))[:, :, ::-1].copy()


def flip_RB(image):
    return np.ascontiguousarray(image[..., ::-1])


def flip_BR(image):
    return np.ascontiguousarray(image[..., ::-1])


def decimal_to_pixel(img):
    return (img * 255.).astype(np.uint8)   # truncation, not rounding (image_utilities.py:195-196)


def pixel_to_decimal(img):
    return np.asarray(img).astype(np.float32) / 255.


def image_3d_to_4d(img):
    image = np.asarray(img)
    if image.ndim == 3 and image.shape[0] != 1:
        image = image.reshape((1,) + image.shape)
    return image


def image_4d_to_3d(img):
    if isinstance(img, np.ndarray) and img.ndim == 4 and img.shape[0] == 1:
        return img.reshape(img.shape[1:])
    return img


def save_image(img, path):
    _imwrite_bgr(path, flip_RB(decimal_to_pixel(image_4d_to_3d(np.asarray(img)))))


def open_image(path):
    return pixel_to_decimal(flip_BR(_imread_bgr(path)))


def load_image(path):
    return image_3d_to_4d(open_image(path))


def write_vector(vector, path):
    with open(path, 'w+') as fl:
        fl.write(str(vector))


def sort_numerical(items):
    items.sort(key=lambda f: int(''.join(filter(str.isdigit, f))))
    return items


# ---- cropping helpers (image_utilities.py:155-185) ----
def clip_image_to_dims(image, width, height, start_x=0, start_y=0):
    if image.shape[0] < height - start_y or image.shape[1] < width - start_x:
        return image
    return image[start_y:height + start_y, start_x:width + start_x]


def _wh(width, height, dim):
    if dim is not None:
        return dim[0], dim[1]
    if width is not None and height is not None:
        return width, height
    return 0, 0


def can_clip_image_to_dims(image, width=None, height=None, dim=None):
    w, h = _wh(width, height, dim)
    return image.shape[0] > h and image.shape[1] > w


def random_clip_image_to_dim(image, width=None, height=None, dim=None):
    w, h = _wh(width, height, dim)
    x0 = np.random.randint(image.shape[1] - w, size=1)[0]
    y0 = np.random.randint(image.shape[0] - h, size=1)[0]
    return clip_image_to_dims(image, w, h, x0, y0)


# ---- dataset managers ----
class _SwapShuffle:
    """len(images) random pair swaps per call (image_utilities.py:142-147)."""

    def shuffle_loaded_images(self):
        for _ in range(len(self.images)):
            i, j = np.random.randint(len(self.images), size=2)
            self.images[i], self.images[j] = self.images[j], self.images[i]

    def get_images(self):
        return self.images


class DatasetManager(_SwapShuffle):
    def __init__(self, directory, target_dim=(224, 224), num_images=None):
        self.images = []
        self.target_dim = target_dim
        self.load_images(directory, num_images)

    def load_images(self, directory, num_images=None):
        for name in os.listdir(directory):
            # the reference keeps loading while len <= num_images, i.e. num_images + 1 images
            if num_images is not None and len(self.images) > num_images:
                continue
            if '.jpg' not in name:
                continue
            img = _imread_bgr(os.path.join(directory, name))
            if can_clip_image_to_dims(img, dim=self.target_dim):
                img = random_clip_image_to_dim(img, dim=self.target_dim)
                self.images.append(pixel_to_decimal(flip_BR(img)))


class CocoDatasetManager(DatasetManager):
    def __init__(self, target_dim=(256, 256), num_images=None):
        super().__init__(COCO_DATASET, target_dim=target_dim, num_images=num_images)


class PreprocessedCocoDatasetManager(_SwapShuffle):
    """Pre-cropped images, loaded whole into host memory (image_utilities.py:119-150)."""

    def __init__(self, target_dim=(256, 256), num_images=None, directory=None):
        self.images = []
        self.target_dim = target_dim
        self.load_images(COCO_DATASET_CLIPPED if directory is None else directory, num_images)

    def load_images(self, directory, num_images=None):
        for name in os.listdir(directory):
            if num_images is not None and len(self.images) > num_images:
                continue
            if '.jpg' not in name:
                continue
            self.images.append(pixel_to_decimal(flip_BR(_imread_bgr(os.path.join(directory, name)))))


def clip_COCO_to_dims(num_images=2000, start=0, target_dim=(256, 256)):
    files = os.listdir(COCO_DATASET)
    if len(files) > start + num_images:
        files = files[start:start + num_images]
    for count, name in enumerate(files):
        if '.jpg' not in name:
            continue
        img = _imread_bgr(os.path.join(COCO_DATASET, name))
        if can_clip_image_to_dims(img, dim=target_dim):
            _imwrite_bgr(os.path.join(COCO_DATASET_CLIPPED, name), random_clip_image_to_dim(img, dim=target_dim))
        print(count)


def generate_gif_from_images_in_dir(dir_path):
    from PIL import Image
    out = os.path.join(dir_path, os.path.basename(os.path.normpath(dir_path) + '.gif'))
    files = sort_numerical([os.path.join(dir_path, x) for x in os.listdir(dir_path) if '.jpg' in x])
    print('Generating gif of ' + str(len(files)) + ' files')
    frames = [Image.open(f).convert('RGB') for f in files]
    if frames:
        frames[0].save(out, save_all=True, append_images=frames[1:], loop=0)
    return out
