"""Single-node multi-GPU helpers: one process per GPU (torch.distributed, backend 'nccl' =
RCCL over xGMI).  Training exchanges only the flat gradient buffer (trainer.GradBuckets);
batched inference shards independent frames and exchanges nothing."""
import os

import torch


def shard_frames(num_frames, rank, world):
    """Contiguous, balanced share of frame indices for `rank` (first ranks take the remainder)."""
    base, rem = divmod(num_frames, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def init_from_env(backend=None):
    """(world, rank, device) from the torchrun environment (WORLD_SIZE, RANK, LOCAL_RANK, MASTER_*); no-op for a
    single process.  backend: 'nccl' (RCCL, one GPU per rank; the default) or 'gloo' (rehearsals with several
    ranks on one GPU: set STMI_DIST_BACKEND=gloo and give every rank the LOCAL_RANK of the GPU to share)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    device = torch.device('cuda', local)
    backend = backend or os.environ.get('STMI_DIST_BACKEND', 'nccl')
    if world > 1:
        torch.cuda.set_device(local)
        if not torch.distributed.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if backend == 'nccl':
                torch.distributed.init_process_group(backend, rank=rank, world_size=world, device_id=device)
            else:
                torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, device


def epoch_batches(num_images, batch, world, rank):
    """Index lists of one epoch's batches for `rank`: the reference's loop (full_graph_stylizer_network.py:196-203)
    cut for `world` data-parallel ranks.

    One rank: int(n / batch) + 1 batches of consecutive images, the last one ragged (possibly empty: skipped), as
    the reference runs them.  Several ranks: a step consumes world * batch consecutive images of the (identically
    shuffled) list, rank r the r-th group of `batch`; every rank runs the same number of steps with full batches;
    the ragged tail is PADDED with images from the start of the list, the same way on every rank, so that every
    image is still seen once per epoch and no rank waits for another."""
    if world == 1:
        out = []
        for step in range(int(num_images / batch) + 1):
            start, end = min(num_images, step * batch), min(num_images, (step + 1) * batch)
            if end > start:
                out.append(list(range(start, end)))
        return out
    per_step = world * batch
    steps = -(-num_images // per_step)
    out = []
    for step in range(steps):
        first = step * per_step + rank * batch
        out.append([(first + i) % num_images for i in range(batch)])
    return out


class FrameStager:
    """Pinned, double-buffered host staging for a stream of frame chunks.

    Three HIP streams: H2D copies, compute, D2H copies.  While chunk k is being stylized, chunk
    k+1 is already crossing PCIe into the other device input buffer and chunk k-1 is on its way
    back into pinned memory, so the card never waits for the host link.  Frames travel in the dtype
    the caller hands over: torch.uint8 (what a decoder leaves and an encoder takes: 100 MB each way
    per 16-frame 1080p chunk) or torch.float32 (398 MB each way).  The host-side copy from the
    caller's (pageable) array into pinned memory is the only synchronous step."""

    def __init__(self, chunk_shape, device, dtype=torch.float32, out_shape=None):
        """out_shape: shape of a result chunk when it differs from the input's (the stylizer rounds sizes that are
        not multiples of 4 up)."""
        self.device, self.dtype = device, dtype
        out_shape = chunk_shape if out_shape is None else out_shape
        self.h2d, self.d2h = torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)
        self.pin_in = [torch.empty(chunk_shape, dtype=dtype).pin_memory() for _ in range(2)]
        self.pin_out = [torch.empty(out_shape, dtype=dtype).pin_memory() for _ in range(2)]
        self.dev_in = [torch.empty(chunk_shape, dtype=dtype, device=device) for _ in range(2)]
        self.dev_out = [torch.empty(out_shape, dtype=dtype, device=device) for _ in range(2)]
        self.in_ready = [torch.cuda.Event() for _ in range(2)]     # H2D of slot done
        self.in_free = [torch.cuda.Event() for _ in range(2)]      # compute finished reading slot
        self.out_ready = [torch.cuda.Event() for _ in range(2)]    # compute wrote dev_out[slot]
        self.out_done = [torch.cuda.Event() for _ in range(2)]     # D2H of slot landed in pin_out
        self._used = [False, False]

    def upload(self, slot, chunk_np):
        """Host array -> pinned -> device slot (asynchronous once the pinned copy is made)."""
        m = chunk_np.shape[0]
        if self._used[slot]:
            self.in_ready[slot].synchronize()      # the slot's previous H2D has left the pinned buffer
        self.pin_in[slot][:m].copy_(torch.from_numpy(chunk_np))
        with torch.cuda.stream(self.h2d):
            if self._used[slot]:
                self.h2d.wait_event(self.in_free[slot])
            self.dev_in[slot][:m].copy_(self.pin_in[slot][:m], non_blocking=True)
            self.in_ready[slot].record(self.h2d)

    def run(self, slot, m, forward):
        """forward(dev_in[:m], dev_out[:m]) on the current stream (the result goes into its second argument,
        or is returned and copied there), then device -> pinned."""
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.in_ready[slot])
        if self._used[slot]:
            cur.wait_event(self.out_done[slot])    # the previous occupant of dev_out[slot] has left
        out = forward(self.dev_in[slot][:m], self.dev_out[slot][:m])
        if out is not None and out.data_ptr() != self.dev_out[slot].data_ptr():
            self.dev_out[slot][:m].copy_(out)
        self.in_free[slot].record(cur)
        self.out_ready[slot].record(cur)
        self._used[slot] = True
        with torch.cuda.stream(self.d2h):
            self.d2h.wait_event(self.out_ready[slot])
            self.pin_out[slot][:m].copy_(self.dev_out[slot][:m], non_blocking=True)
            self.out_done[slot].record(self.d2h)

    def collect(self, slot, m, copy=True):
        """The finished chunk of `slot` (waits for its D2H).  copy=False hands out the pinned buffer itself, valid
        until the slot's next run()."""
        self.out_done[slot].synchronize()
        out = self.pin_out[slot][:m].numpy()
        return out.copy() if copy else out


def stylize_frames_sharded(frames, params, dtype=torch.bfloat16, batch=16):
    """Every rank stylizes its own shard of `frames` ([F,H,W,3] host array) and returns (indices, outputs) for
    this rank; no collective is issued.  uint8 frames (as decoded) come back as uint8 frames (as an encoder
    takes them: (stylized * 255) truncated, image_utilities.py:195-196); float32 frames in [0,1] come back as
    float32.  Chunks of `batch` frames move through a FrameStager, so PCIe traffic in both directions overlaps
    the kernels."""
    import numpy as np
    from .stylizer import StylizerNet, stylized_size
    world, rank, device = init_from_env()
    idx = shard_frames(frames.shape[0], rank, world)
    shape = tuple(frames.shape[1:])
    oshape = stylized_size(shape[0], shape[1]) + (3,)
    u8 = frames.dtype == np.uint8
    if not u8 and frames.dtype != np.float32:
        raise TypeError('frames must be uint8 (0..255) or float32 (0..1), got %s' % frames.dtype)
    if len(idx) == 0:
        return idx, np.zeros((0,) + oshape, frames.dtype)
    starts = list(range(idx.start, idx.stop, batch))
    sizes = [min(s + batch, idx.stop) - s for s in starts]
    stager = FrameStager((min(batch, len(idx)),) + shape, device, torch.uint8 if u8 else torch.float32,
                         out_shape=(min(batch, len(idx)),) + oshape)
    nets = {}

    def forward(x, out):
        m = x.shape[0]
        if m not in nets:
            nets[m] = StylizerNet(m, shape[0], shape[1], device, dtype, params=params)
        if u8:
            nets[m].forward_u8(x, out=out)
            return None
        return nets[m].forward(x)

    outs = []
    stager.upload(0, frames[starts[0]:starts[0] + sizes[0]])
    for k, (s, m) in enumerate(zip(starts, sizes)):
        slot = k & 1
        if k + 1 < len(starts):
            stager.upload(slot ^ 1, frames[starts[k + 1]:starts[k + 1] + sizes[k + 1]])
        stager.run(slot, m, forward)
        if k >= 1:
            outs.append(stager.collect(slot ^ 1, sizes[k - 1]))
    outs.append(stager.collect((len(starts) - 1) & 1, sizes[-1]))
    return idx, np.concatenate(outs)


class FrameCodecPool:
    """JPEG decode / encode on a pool of host threads (PIL releases the interpreter lock inside its codecs), the
    host side of a frame service: load_image's decode (image_utilities.py:229-236, without the division by 255,
    which the first layer does) and save_image's encode (:223-227, without the multiplication, which the last
    layer did).  map-style: results come back in input order."""

    def __init__(self, workers=None):
        import os as _os
        from concurrent.futures import ThreadPoolExecutor
        if workers is None:
            workers = int(_os.environ.get('OMP_NUM_THREADS', '0')) or min(16, _os.cpu_count() or 1)
        self.workers = workers
        self._pool = ThreadPoolExecutor(max_workers=workers)

    @staticmethod
    def _decode(data):
        import io
        import numpy as np
        from PIL import Image
        src = io.BytesIO(data) if isinstance(data, (bytes, bytearray, memoryview)) else data
        with Image.open(src) as im:
            return np.asarray(im.convert('RGB'), dtype=np.uint8)

    @staticmethod
    def _encode(frame, quality=95):
        import io
        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(frame, 'RGB').save(buf, format='JPEG', quality=quality)
        return buf.getvalue()

    def decode(self, sources):
        """JPEG byte strings or paths -> list of uint8 [H,W,3] RGB frames."""
        return list(self._pool.map(self._decode, sources))

    def encode(self, frames, quality=95):
        """uint8 [H,W,3] RGB frames -> list of JPEG byte strings."""
        return list(self._pool.map(lambda f: self._encode(f, quality), frames))

    def close(self):
        self._pool.shutdown()
