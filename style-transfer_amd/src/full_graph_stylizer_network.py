"""Drop-in for the reference's src/full_graph_stylizer_network.py.

Command line (same argv forms as the reference, full_graph_stylizer_network.py:407-433):
    full_graph_stylizer_network.py SRC DST                 newest STYLE_NET model
    full_graph_stylizer_network.py MODEL SRC DST           models/MODEL
    full_graph_stylizer_network.py MODEL BASE_DIR SRC DST  BASE_DIR/MODEL
    anything else                                          train_style_network('heiro_4')

StyleNetService, run_on_image, train_style_network, normal_style_transfer and
build_style_network keep their names, arguments and defaults; the work is done by the
HIP executors in ``stmi`` instead of a TensorFlow session.  Trained weights are stored
as ``saved_<model>-0.npz`` (48 arrays keyed like the reference's checkpoint variables,
``style_network/Variable``, ``style_network/Variable_1`` ...) next to a ``checkpoint``
file naming the newest one.
"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.join(HERE, '..') not in sys.path:
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..')))

import constants
import model_utilities
from image_utilities import load_image, CocoDatasetManager, save_image, PreprocessedCocoDatasetManager  # noqa: F401
from model_utilities import LayerConfig, StyleTransfer, nn
from vgg import *  # noqa: F401,F403  (the reference re-exports the vgg names the same way)
import vgg as _vgg

LEARNING_RATE = 1e-3
EPOCHS = 80
BATCH_SIZE = 4
TOTAL_IMAGES = 1000

STYLE_IMAGE_1 = os.path.join(constants.STYLES_DIR, 'starry_night_small.jpg')
STYLE_IMAGE_1_LARGE = os.path.join(constants.STYLES_DIR, 'starry_night.jpg')
STYLE_IMAGE_2_LARGE = os.path.join(constants.STYLES_DIR, 'honeycomb_large.jpg')
STYLE_IMAGE_2_SQUEEZE = os.path.join(constants.STYLES_DIR, 'honeycomb_squeeze.jpg')

CONTENT_IMAGE_1 = os.path.join(constants.CONTENT_DIR, 'antelope_small.jpg')
CONTENT_IMAGE_1_2 = os.path.join(constants.CONTENT_DIR, 'antelope_256.jpg')
CONTENT_IMAGE_2 = os.path.join(constants.CONTENT_DIR, 'sparrow_small.jpg')
CONTENT_IMAGE_2_LARGE = os.path.join(constants.CONTENT_DIR, 'sparrow.jpg')
CONTENT_IMAGE_3 = os.path.join(constants.CONTENT_DIR, 'museum.jpg')

model_prefix = 'STYLE_NET'

TRAIN_IMG_DIMS = [None, 256, 256, 3]


def _compute_dtype():
    return torch.bfloat16 if os.environ.get('STMI_DTYPE', 'f32') == 'bf16' else torch.float32


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('no MI355X visible: the stylizer runs on HIP kernels only, there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


# --------------------------------------------------------------------------
# network definition (reference :42-82)
# --------------------------------------------------------------------------
def build_style_network(network_input):
    conv_layer_configs = [LayerConfig(32, 9, 1), LayerConfig(64, 3, 2), LayerConfig(128, 3, 2)]
    residual_block_configs = [LayerConfig(128, 3, 1) for _ in range(5)]
    deconv_layer_configs = [LayerConfig(64, 3, 2), LayerConfig(32, 3, 2)]
    last_layer = LayerConfig(3, 9, 1, activation=nn.sigmoid)

    model_builder = model_utilities.NNGraphModelBuilder
    layer = network_input
    for x in conv_layer_configs:
        layer = model_builder.conv_block(layer, x)
    for x in residual_block_configs:
        layer = model_builder.residual_block(layer, x)
    for x in deconv_layer_configs:
        layer = model_builder.deconv_block(layer, x)
    layer = model_builder.conv_block(layer, last_layer)

    layer_configs = {
        'conv_layers': [repr(x) for x in conv_layer_configs],
        'residual_layers': [repr(x) for x in residual_block_configs],
        'deconv_layers': [repr(x) for x in deconv_layer_configs],
    }
    return layer, layer_configs


# --------------------------------------------------------------------------
# weight files
# --------------------------------------------------------------------------
def _var_key(i):
    return 'style_network/Variable' if i == 0 else 'style_network/Variable_%d' % i


def save_checkpoint(params, model_dir, model_name, global_step=0):
    """48 arrays in creation order -> <model_dir>/saved_<model_name>-<step>.npz + 'checkpoint'."""
    base = 'saved_%s-%d' % (model_name, global_step)
    np.savez(os.path.join(model_dir, base + '.npz'), **{_var_key(i): np.asarray(p) for i, p in enumerate(params)})
    with open(os.path.join(model_dir, 'checkpoint'), 'w') as fl:
        fl.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))
    return os.path.join(model_dir, base + '.npz')


def latest_checkpoint(model_dir):
    """Path of the newest weight file in model_dir, or None (tf.train.latest_checkpoint's role):
    the `.npz` this build writes, or the `.index` of a checkpoint the reference's Saver wrote."""
    marker = os.path.join(model_dir, 'checkpoint')
    if os.path.exists(marker):
        with open(marker) as fl:
            for line in fl:
                if line.startswith('model_checkpoint_path:'):
                    base = os.path.basename(line.split(':', 1)[1].strip().strip('"'))
                    for ext in ('.npz', '.index'):
                        path = os.path.join(model_dir, base + ext)
                        if os.path.exists(path):
                            return path
    names = os.listdir(model_dir) if os.path.isdir(model_dir) else []
    for ext in ('.npz', '.index'):
        found = sorted(x for x in names if x.startswith('saved_') and x.endswith(ext))
        if found:
            return os.path.join(model_dir, found[-1])
    return None


def load_checkpoint(path):
    """The 48 stylizer arrays in creation order, from this build's `.npz` or from a TensorFlow
    tensor-bundle checkpoint (`saved_<name>-0.index` + `.data-00000-of-00001`, what
    full_graph_stylizer_network.py:183/:231 of the reference saves and :366-367 restores; its
    Adam slots and beta powers are ignored)."""
    if path.endswith('.index'):
        import tf_bundle
        prefix = path[:-len('.index')]
        have = tf_bundle.list_bundle(prefix)
        count = 0
        while _var_key(count) in have:
            count += 1
        if count == 0:
            raise KeyError('%s holds no style_network/Variable* tensors' % path)
        got = tf_bundle.read_bundle(prefix, [_var_key(i) for i in range(count)])
        return [got[_var_key(i)] for i in range(count)]
    with np.load(path) as z:
        return [z[_var_key(i)] for i in range(len(z.files))]


def _load_vgg_for_training():
    """The VGG weights, and the preprocessing mean the reference's training graph ends up
    with: its vgg_preprocess_input calls are evaluated before the builder assigns the
    module global, so the graph is built with the mean still [0,0,0]
    (full_graph_stylizer_network.py:107-108 with vgg.py:9, :72-74)."""
    mean_at_graph_build = np.asarray(_vgg.vgg_preprocessing_mean, dtype=np.float32).copy()
    path = _vgg.VGG_PATH_REPLICATE if os.path.exists(_vgg.VGG_PATH_REPLICATE) else _vgg.VGG_PATH
    weights, mean_pixel = _vgg.load_vgg_mat(path)
    _vgg.vgg_preprocessing_mean = mean_pixel
    return weights, mean_at_graph_build


# --------------------------------------------------------------------------
# training (reference :85-232)
# --------------------------------------------------------------------------
def train_style_network(style_type, test_content_image_path=CONTENT_IMAGE_1_2, dataset_manager=None,
                        epochs=None, log=print):
    """The reference's training loop (full_graph_stylizer_network.py:85-232).

    Data parallel when started once per GPU with the torchrun environment (WORLD_SIZE, RANK, LOCAL_RANK,
    MASTER_ADDR, MASTER_PORT): every rank loads the same images and shuffles them with the same random stream,
    a step consumes WORLD_SIZE * BATCH_SIZE images (BATCH_SIZE per GPU), the gradients are summed over RCCL and
    averaged, and every rank applies the same TF1-Adam update.  The ragged last batch of an epoch
    (:196-203) is padded with images from the start of the shuffled list, identically on every rank.
    Rank 0 alone writes config.json, losses.jsonl (its own shard's losses), the per-epoch sample and the
    checkpoint; every rank returns the model name."""
    from stmi import dist as stdist
    from stmi.stylizer import StylizerNet
    from stmi.trainer import StyleTrainer
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), 0
    if world > 1:
        world, rank, device = stdist.init_from_env()
    else:
        device = _device()
    lead = rank == 0
    if not lead:
        log = lambda *a, **k: None
    style_config = StyleTransfer.STYLE_CONFIG_DICT[style_type]
    style_image = load_image(style_config.style_path)
    test_content_image = load_image(test_content_image_path)
    dtype = _compute_dtype()
    epochs = EPOCHS if epochs is None else epochs

    log('Building networks...')
    vgg_weights, vgg_mean = _load_vgg_for_training()
    trainer = StyleTrainer(BATCH_SIZE, TRAIN_IMG_DIMS[1], TRAIN_IMG_DIMS[2], device, vgg_weights, vgg_mean,
                           style_config.style_weight, style_config.content_weight,
                           style_config.total_variation_weight, dtype=dtype, lr=LEARNING_RATE, world_size=world)
    # one rank: the last batch of an epoch is ragged (1001 images, batch 4) and gets its own executor
    tail_trainers = {}
    log('Done building networks.')

    log('Loading images...')
    if dataset_manager is None:
        dataset_manager = PreprocessedCocoDatasetManager(target_dim=(256, 256), num_images=TOTAL_IMAGES)
    log('Done loading images.')

    model_name = time.strftime(model_prefix + '_%Y-%m-%d-%H-%M') + '_' + style_type
    if world > 1:
        # one model name and ONE shuffle stream for all ranks (the lists must stay identical epoch after epoch)
        import torch.distributed as tdist
        shared = [model_name, int(np.random.randint(1 << 31))]
        tdist.broadcast_object_list(shared, src=0)
        model_name = shared[0]
        np.random.seed(shared[1])
    target_dir = os.path.join(constants.MODELS_DIR, model_name)
    loss_log = None
    if lead:
        os.mkdir(target_dir)
        _, style_network_configs = _layer_config_strings()
        style_network_configs['style_config'] = repr(style_config)
        style_network_configs['style_layers'] = StyleTransfer.VGG_STYLE_TARGET_LAYER_NAMES
        style_network_configs['content_layers'] = StyleTransfer.VGG_CONTENT_TARGET_LAYER_NAMES
        style_network_configs['batch_size'] = BATCH_SIZE
        style_network_configs['data_size'] = TOTAL_IMAGES
        style_network_configs['epochs'] = epochs
        style_network_configs['LR'] = LEARNING_RATE
        style_network_configs['config_name'] = style_type
        with open(os.path.join(target_dir, 'config.json'), 'w+') as fl:
            json.dump(style_network_configs, fl, indent=4)
        loss_log = open(os.path.join(target_dir, 'losses.jsonl'), 'w')

    trainer.set_style_image(torch.from_numpy(np.ascontiguousarray(style_image)))
    sampler = StylizerNet(1, test_content_image.shape[1], test_content_image.shape[2], device, dtype) if lead else None
    train_start_time = time.time()
    for epoch in range(epochs):
        dataset_manager.shuffle_loaded_images()
        images = dataset_manager.get_images()
        log('Epoch ' + str(epoch + 1) + ' of ' + str(epochs))
        batches = stdist.epoch_batches(len(images), BATCH_SIZE, world, rank)
        num_training_steps = int(len(images) / BATCH_SIZE) + 1 if world == 1 else len(batches)
        for step, index in enumerate(batches):
            batch = torch.from_numpy(np.ascontiguousarray(np.array([images[i] for i in index], dtype=np.float32)))
            if len(index) == BATCH_SIZE:
                active = trainer
            else:
                active = _tail_trainer(tail_trainers, trainer, len(index), vgg_weights, vgg_mean, style_config, dtype)
            active.step(batch.to(device))
            if active is not trainer:
                _sync_back(active, trainer)
            if lead and step % max(int(num_training_steps / 10), 1) == 0:
                total, style_l, content_l, tv_l = active.read_losses()
                loss_log.write(json.dumps({'step': epoch * num_training_steps + step, 'total': total, 'style': style_l,
                                           'content': content_l, 'tv': tv_l}) + '\n')
                loss_log.flush()
        if lead:
            sampler.load_params(trainer.net.export_params())
            sampler.repack()
            sampled_image = sampler.forward(torch.from_numpy(np.ascontiguousarray(test_content_image)).to(device))
            save_image(sampled_image.cpu().numpy(), os.path.join(target_dir, str(epoch) + '_style_transfer_sample.jpg'))
        elapsed = time.time() - train_start_time
        eta = (elapsed / max(1, epoch + 1)) * (epochs - (epoch + 1))
        log('elapsed: ' + str(elapsed / 60.)[:6] + ' min | remaining training time: ' + str(eta / 60.)[:6] + ' min')
    log('Training concluded. Saving model...')
    if lead:
        save_checkpoint(trainer.net.export_params(), target_dir, model_name, global_step=0)
        loss_log.close()
    if world > 1:
        import torch.distributed as tdist
        torch.cuda.synchronize()
        tdist.barrier()   # no rank leaves (and tears the group down) while another still reduces
    log('Model saved.')
    return model_name


def _layer_config_strings():
    convs = [LayerConfig(32, 9, 1), LayerConfig(64, 3, 2), LayerConfig(128, 3, 2)]
    res = [LayerConfig(128, 3, 1) for _ in range(5)]
    dec = [LayerConfig(64, 3, 2), LayerConfig(32, 3, 2)]
    return None, {'conv_layers': [repr(x) for x in convs], 'residual_layers': [repr(x) for x in res],
                  'deconv_layers': [repr(x) for x in dec]}


def _tail_trainer(cache, main, n, vgg_weights, vgg_mean, style_config, dtype):
    """Executor for a ragged batch of n images sharing the main trainer's state."""
    from stmi.trainer import StyleTrainer
    if n not in cache:
        t = StyleTrainer(n, main.h, main.w, main.device, vgg_weights, vgg_mean, style_config.style_weight,
                         style_config.content_weight, style_config.total_variation_weight, dtype=dtype, lr=main.lr,
                         use_graph=False)
        t.style_targets = main.style_targets
        cache[n] = t
    t = cache[n]
    t.net.flat.copy_(main.net.flat); t.m.copy_(main.m); t.v.copy_(main.v); t.step_dev.copy_(main.step_dev)
    t.net.repack()
    return t


def _sync_back(tail, main):
    main.net.flat.copy_(tail.net.flat); main.m.copy_(tail.m); main.v.copy_(tail.v); main.step_dev.copy_(tail.step_dev)
    main.net.repack()


# --------------------------------------------------------------------------
# Gatys optimisation (reference :235-346)
# --------------------------------------------------------------------------
def normal_style_transfer(style_type, epochs=None, steps_per_epoch=100, log=print):
    from stmi.trainer import GatysOptimizer
    style_config = StyleTransfer.STYLE_CONFIG_DICT[style_type]
    style_image_1 = load_image(style_config.style_path)
    content_image_1 = load_image(CONTENT_IMAGE_1)
    device, dtype = _device(), _compute_dtype()
    epochs = EPOCHS if epochs is None else epochs
    vgg_weights, vgg_mean = _load_vgg_for_training()
    opt = GatysOptimizer(torch.from_numpy(np.ascontiguousarray(content_image_1)), device, vgg_weights, vgg_mean,
                         style_config.style_weight, style_config.content_weight, style_config.total_variation_weight,
                         dtype=dtype)
    opt.set_style_image(torch.from_numpy(np.ascontiguousarray(style_image_1)))
    train_start_time = time.time()
    for epoch in range(epochs):
        log('Epoch ' + str(epoch + 1) + ' of ' + str(epochs))
        for _ in range(steps_per_epoch):
            opt.step()
        save_image(opt.image.cpu().numpy(),
                   os.path.join(constants.STYLIZED_IMAGES_DIR, str(epoch) + '_style_transfer_sample_1.jpg'))
        elapsed = time.time() - train_start_time
        eta = (elapsed / max(1, epoch + 1)) * (epochs - (epoch + 1))
        log('elapsed: ' + str(elapsed / 60.)[:6] + ' min | remaining training time: ' + str(eta / 60.)[:6] + ' min')
    return opt.image.cpu().numpy()


# --------------------------------------------------------------------------
# inference service (reference :349-405)
# --------------------------------------------------------------------------
class StyleNetService(threading.Thread):
    """Thread-backed model holder.  start() loads the weights and flips `loaded`; run_on_image
    may then be called from any thread.

    Calls on ONE service are serialised by its lock; different services run side by side, each
    on a HIP stream of its own (the Django view keeps three services and calls them from its
    request threads, ml_services/style_net_service/views.py:23-30, :52).  Executors (static
    buffers for one image shape) are cached per shape, least recently used first out, within
    STMI_SERVICE_CACHE_MB (default 8192) of device memory, so that clients alternating between
    a few image sizes do not rebuild on every request.

    A failed load is kept in `error`: `loaded` stays False as in the reference, but
    wait_for_ready() raises instead of polling forever."""

    def __init__(self, model_name, model_dir=constants.MODELS_DIR):
        super().__init__()
        self.model_name = model_name
        self.model_dir = model_dir
        self.loaded = False
        self.error = None
        self._lock = threading.Lock()
        self._nets = {}          # (n, h, w) -> executor, in least-recently-used order
        self._params = None
        self._device = None
        self._stream = None
        self._cache_bytes = int(os.environ.get('STMI_SERVICE_CACHE_MB', '8192')) << 20

    def run(self):
        print('StyleNetService starting...')
        try:
            model_dir = os.path.join(self.model_dir, self.model_name)
            if not os.path.isdir(model_dir) and os.path.isdir(self.model_name):
                model_dir = self.model_name   # the 4-argument CLI form hands over an already joined path
            path = latest_checkpoint(model_dir)
            if path is None:
                raise FileNotFoundError('no saved_*.npz or saved_*.index weight file in ' + model_dir)
            self._params = load_checkpoint(path)
        except BaseException as exc:   # kept for wait_for_ready(); the thread itself has no caller to raise to
            self.error = exc
            print('StyleNetService failed to load: %r' % (exc,))
            return
        print('StyleNetService running.')
        self.loaded = True

    @staticmethod
    def _executor_bytes(net):
        return sum(t.numel() * t.element_size() for t in net._pool) + net.x_in.numel() * net.x_in.element_size() \
            + net.out_f32.numel() * 4

    def _executor(self, n, h, w):
        from stmi.stylizer import StylizerNet
        net = self._nets.pop((n, h, w), None)
        if net is None:
            net = StylizerNet(n, h, w, self._device, _compute_dtype(), params=self._params)
            need = self._executor_bytes(net)
            while self._nets and need + sum(self._executor_bytes(x) for x in self._nets.values()) > self._cache_bytes:
                self._nets.pop(next(iter(self._nets)))
        self._nets[(n, h, w)] = net      # most recently used last
        return net

    def run_on_image(self, input_image):
        if not self.loaded:
            raise RuntimeError('StyleNetService %s is not loaded%s' % (
                self.model_name, '' if self.error is None else ': %r' % (self.error,)))
        x = np.ascontiguousarray(np.asarray(input_image, dtype=np.float32))
        n, h, w, _ = x.shape
        with self._lock:
            if self._device is None:
                self._device = _device()
                self._stream = torch.cuda.Stream(device=self._device)
            with torch.cuda.device(self._device), torch.cuda.stream(self._stream):
                net = self._executor(n, h, w)
                out = net.forward(torch.from_numpy(x).to(self._device, non_blocking=False))
                host = out.cpu()        # stream-ordered copy; returns once this service's stream is done
            return host.numpy()

    def close(self):
        with self._lock:
            self._nets = {}

    def wait_for_ready(self, increment=.1):
        while not self.loaded:
            if self.error is not None:
                raise RuntimeError('StyleNetService %s failed to load' % self.model_name) from self.error
            if self.ident is not None and not self.is_alive() and not self.loaded:
                raise RuntimeError('StyleNetService %s stopped without loading a model' % self.model_name)
            time.sleep(increment)


def run_on_image(source_image_path, destination_path, model_path=None, model_dir=constants.MODELS_DIR):
    target_image = load_image(source_image_path)
    real_model_path = model_path
    if real_model_path is None:
        real_model_path = model_utilities.get_most_recent_model_name(model_dir, model_prefix)
    service = StyleNetService(real_model_path, model_dir)
    service.start()
    service.wait_for_ready()
    output_image = service.run_on_image(target_image)
    save_image(output_image, destination_path)
    service.close()


def main(argv):
    if len(argv) == 3:
        src, dest = argv[1], argv[2]
        if os.path.exists(src):
            run_on_image(src, dest)
    elif len(argv) == 4:
        model, src, dest = argv[1], argv[2], argv[3]
        model_dir = os.path.join(constants.MODELS_DIR, model)
        if os.path.exists(src) and os.path.exists(model_dir) and os.path.isdir(model_dir):
            run_on_image(src, dest, model_path=model)
    elif len(argv) == 5:
        model, model_base_dir, src, dest = argv[1], argv[2], argv[3], argv[4]
        model_dir = os.path.join(model_base_dir, model)
        print('--' + model_dir)
        if os.path.exists(src) and os.path.exists(model_dir) and os.path.isdir(model_dir):
            # the reference passes the JOINED path as the model name here; os.path.join then
            # ignores model_base_dir only when that path is absolute.  Kept as is.
            run_on_image(src, dest, model_path=model_dir, model_dir=model_base_dir)
    else:
        train_style_network('heiro_4')


if __name__ == '__main__':
    main(sys.argv)
