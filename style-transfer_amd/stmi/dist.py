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


def init_from_env(backend='nccl'):
    """(world, rank, device) from the torchrun environment; no-op for a single process."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    device = torch.device('cuda', local)
    if world > 1 and not torch.distributed.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local)
        torch.distributed.init_process_group(backend, rank=rank, world_size=world, device_id=device)
    return world, rank, device


def stylize_frames_sharded(frames_f32, params, dtype=torch.bfloat16, batch=16):
    """Every rank stylizes its own shard of `frames_f32` ([F,H,W,3] host array); returns
    (indices, outputs) for this rank.  No collective is issued."""
    import numpy as np
    from .stylizer import StylizerNet
    world, rank, device = init_from_env()
    idx = shard_frames(frames_f32.shape[0], rank, world)
    outs = []
    net = None
    for s in range(idx.start, idx.stop, batch):
        chunk = frames_f32[s:min(s + batch, idx.stop)]
        if net is None or net.n != chunk.shape[0]:
            net = StylizerNet(chunk.shape[0], chunk.shape[1], chunk.shape[2], device, dtype, params=params)
        outs.append(net.forward(torch.from_numpy(np.ascontiguousarray(chunk)).to(device)).cpu().numpy())
    return idx, (np.concatenate(outs) if outs else np.zeros((0,) + frames_f32.shape[1:], np.float32))
