"""N>1 path on CPU: world_size-2 gloo process group (no GPU): frame sharding covers every
frame exactly once, the bench's max-over-ranks timing and the rank-0 gather of per-frame
results work."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mv3d_tf_amd import sharding


def test_frame_shard_partition():
    for n in (0, 1, 7, 16):
        for w in (1, 2, 3, 8):
            seen = sorted(f for r in range(w) for f in sharding.frame_shard(n, r, w))
            assert seen == list(range(n))
            sizes = [len(sharding.frame_shard(n, r, w)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1
            assert all(sharding.owner_of(f, w) == r for r in range(w) for f in sharding.frame_shard(n, r, w))
    with pytest.raises(ValueError):
        sharding.frame_shard(4, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7
        mine = sharding.frame_shard(n, rank, world)
        # stand-in for per-frame results: frame f keeps f+1 "rois" with ids f*100 + i
        local = {f: torch.arange(f + 1, dtype=torch.int64) + 100 * f for f in mine}
        res = sharding.gather_frame_results(local, n, dist)
        ok = all(torch.equal(res[f], torch.arange(f + 1, dtype=torch.int64) + 100 * f) for f in range(n))
        tmax = sharding.max_over_ranks(1.0 + rank, dist)
        dist.barrier()
        q.put((rank, ok, tmax, mine))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_and_timing():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert [g[1] for g in got] == [True, True]
    assert [g[2] for g in got] == [2.0, 2.0]                 # max over ranks
    assert got[0][3] == [0, 2, 4, 6] and got[1][3] == [1, 3, 5]


def test_single_process_fallbacks():
    assert sharding.max_over_ranks(3.5) == 3.5
    local = {0: torch.tensor([1, 2]), 1: torch.tensor([3])}
    out = sharding.gather_frame_results(local, 2)
    assert torch.equal(out[0], torch.tensor([1, 2])) and torch.equal(out[1], torch.tensor([3]))
