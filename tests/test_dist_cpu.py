"""CPU tier: the N > 1 path (detector sharding + TOD all-gather) with gloo, world size 2."""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maria_amd import dist as mdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(10000, 8), (217, 2), (217, 8), (50000, 8), (5, 4), (16, 1), (33, 2)])
def test_shard_bounds_cover_and_align(n, world):
    b = [mdist.shard_bounds(n, world, r) for r in range(world)]
    assert b[0][0] == 0 and b[-1][1] == n
    assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    assert all((lo % 16 == 0) or lo == n for lo, _ in b)
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(s for s in sizes if s > 0 or True) <= max(sizes)  # contiguous, last may be short/empty


def _worker(rank, world, port, n_det, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sl = mdist.shard_slice(n_det)
        rows = torch.arange(sl.start, sl.stop, dtype=torch.float32)[:, None]
        local = rows * 1000 + torch.arange(T, dtype=torch.float32)[None]  # value identifies (det, sample)
        full = mdist.all_gather_tod(local, n_det, time_chunk=7)
        expect = torch.arange(n_det, dtype=torch.float32)[:, None] * 1000 + torch.arange(T, dtype=torch.float32)[None]
        # streamed variant: every gathered chunk equals the matching columns of the whole
        seen = []
        nbytes = mdist.stream_gathered_tod(local, n_det, 5, lambda s0, blk: seen.append(bool(torch.equal(blk, expect[:, s0 : s0 + blk.shape[1]]))))
        assert all(seen) and len(seen) == -(-T // 5) and nbytes > 0
        # weak-scaling accounting as bench.py does it: max over ranks of the elapsed time
        tmax = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        q.put((rank, bool(torch.equal(full, expect)), float(tmax.item()), (sl.start, sl.stop)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_det", [217, 40, 3])
def test_all_gather_tod_gloo_world2(n_det):
    world, T = 2, 23
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_det) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_det, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in results), results
    assert all(t == float(world) for _, _, t, _ in results)
    bounds = sorted(b for *_, b in results)
    assert bounds[0][0] == 0 and bounds[-1][1] == n_det and bounds[0][1] == bounds[1][0]


def test_single_process_gather_is_identity():
    x = torch.arange(12.0).reshape(3, 4)
    assert mdist.all_gather_tod(x, 3) is x
    assert mdist.shard_slice(100) == slice(0, 100)
