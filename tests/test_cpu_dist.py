"""World-size-2 tests of the multi-GPU plumbing on the gloo backend (CPU tensors):
the bucketed gradient all-reduce that data-parallel training uses, and the
collective-free frame sharding of batched inference."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from stmi import stylizer
        from stmi.dist import shard_frames
        from stmi.trainer import BWD_SEGMENTS, GradBuckets
        shapes = stylizer.param_shapes()
        sizes = [int(np.prod(s)) for s in shapes]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        ranges = [(int(offs[3 * lo]), int(offs[3 * (hi + 1)])) for hi, lo in BWD_SEGMENTS]
        total = int(offs[-1])
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(total, generator=g)
        mine = flat.clone()
        buckets = GradBuckets(flat, ranges)
        # buckets become ready in backward order, are reduced asynchronously, then awaited once
        for i in range(len(ranges)):
            buckets.reduce_bucket(i)
        buckets.wait()
        others = [torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        expect = sum(others)
        ok = torch.allclose(flat, expect, atol=1e-6)
        ok = ok and torch.equal(mine, others[rank])
        # Adam then divides by the world size: mean of per-rank means == global-batch mean for equal shards
        ok = ok and torch.allclose(flat / world, torch.stack(others).mean(0), atol=1e-6)
        shard = shard_frames(37, rank, world)
        q.put((rank, bool(ok), ranges, list(shard)))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_frame_sharding_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results)
    ranges = results[0][2]
    # the buckets, in the order backward finishes them, tile all 1,677,638 gradients
    assert len(ranges) >= 2 and ranges[0][1] == 1677638 and ranges[-1][0] == 0
    assert all(a[1] == b[0] for a, b in zip(sorted(ranges), sorted(ranges)[1:]))
    frames = results[0][3] + results[1][3]
    assert sorted(frames) == list(range(37))                    # every frame exactly once, no exchange
    assert abs(len(results[0][3]) - len(results[1][3])) <= 1


def test_shard_frames_edge_cases():
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'style-transfer_amd')]
    from stmi.dist import shard_frames
    assert list(shard_frames(0, 0, 8)) == []
    assert [len(shard_frames(3, r, 8)) for r in range(8)] == [1, 1, 1, 0, 0, 0, 0, 0]
    assert [list(shard_frames(10, r, 4)) for r in range(4)] == [[0, 1, 2], [3, 4, 5], [6, 7], [8, 9]]


def test_epoch_batches_reference_loop_and_data_parallel_cut():
    """One rank: the reference's loop (full_graph_stylizer_network.py:196-203: int(n / 4) + 1 steps, ragged tail,
    empty tail skipped).  Several ranks: equal step counts, full batches, disjoint groups within a step, every
    image seen, the tail padded from the start of the list identically everywhere."""
    from stmi.dist import epoch_batches
    for n in (1001, 1000, 5, 4, 3):
        got = epoch_batches(n, 4, 1, 0)
        want = []
        for step in range(int(n / 4) + 1):           # the reference's index arithmetic
            start, end = min(n, step * 4), min(n, (step + 1) * 4)
            if start != end:
                want.append(list(range(start, end)))
        assert got == want
    for n, world in ((1001, 8), (1001, 2), (10, 2), (64, 4), (7, 4)):
        per_rank = [epoch_batches(n, 4, world, r) for r in range(world)]
        steps = {len(b) for b in per_rank}
        assert len(steps) == 1                          # no rank waits for another
        assert all(len(ix) == 4 for b in per_rank for ix in b)
        seen = set()
        for s in range(steps.pop()):
            group = [i for r in range(world) for i in per_rank[r][s]]
            assert group == [(s * world * 4 + j) % n for j in range(world * 4)]   # consecutive, wrapped at the end
            seen.update(group)
        assert seen == set(range(n))
