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


@pytest.mark.parametrize("n_det,world", [(217, 2), (40, 2), (3, 2), (10000, 8), (217, 8)])
def test_all_gather_tod_gloo(n_det, world):
    """World 2, and the eight ranks of the driver's largest run (BASELINE config 4's 10 000 rows in shards of 1 264 and a short
    last one; 217 rows leave ranks without rows)."""
    T = 23
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # a free port of this case's own (a formula of the pid and the row count gave (217, 2) and (217, 8) the same one)
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
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
    assert bounds[0][0] == 0 and bounds[-1][1] == n_det and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))


def test_single_process_gather_is_identity():
    x = torch.arange(12.0).reshape(3, 4)
    assert mdist.all_gather_tod(x, 3) is x
    assert mdist.shard_slice(100) == slice(0, 100)


def _screen_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # five layers of different shapes; "generating" layer l = a function of (seed, l) only
        shapes = [(6, 9), (4, 4), (7, 3), (5, 5), (2, 8)]
        make = lambda l: torch.arange(shapes[l][0] * shapes[l][1], dtype=torch.float32).reshape(shapes[l]) * (l + 1) + 0.5 * l  # noqa: E731
        mine = mdist.layers_of_rank(len(shapes))
        bufs = [make(l) if l in mine else torch.full(shapes[l], float("nan")) for l in range(len(shapes))]
        mdist.exchange_layer_screens(bufs)
        ok = all(torch.equal(b, make(l)) for l, b in enumerate(bufs))
        q.put((rank, ok, mine))
    finally:
        dist.destroy_process_group()


def test_layer_sharded_screens_gloo_world2():
    """Strong-scaling option: each rank generates its round-robin share of the layers and the
    owners broadcast them; every rank ends with all layers, bit-identical."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_screen_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    assert results[0][2] == [0, 2, 4] and results[1][2] == [1, 3]


def _sim_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from maria_amd.instrument import Band, Detectors, Instrument, Site
        from maria_amd.sim import TOD, Plan, Simulation

        bands = [Band(center=93e9, width=27e9, name="f093", gain_error=0.05)]
        inst = Instrument(Detectors.hexagon(40, 0.3, bands, primary_size=6.0))
        plan = Plan.daisy(start_time=1.7e9, duration=10.0, sample_rate=20.0, scan_center=(45.0, 55.0), radius=0.4, speed=0.4)
        sim = Simulation(inst, plan, Site(altitude=1000.0), atmosphere="2d", atmosphere_kwargs={"n_layers": 2}, noise=False,
                         gain_seed=3, shard="auto")
        lo, hi = sim._rows(inst.dets.n)
        obs = sim.obs_list[0]
        # the host geometry is the whole instrument's on every rank (what makes shards bit-identical)
        geom = [float(obs.atmosphere.timestep)] + [float(p["cross_section"][0]) for p in obs.atmosphere.processes.values()]
        # the gather of Simulation.run(gather=True), on a stand-in TOD of this rank's rows
        rows = np.arange(lo, hi, dtype=np.float32)[:, None] + np.zeros((1, len(plan.time)), np.float32)
        tod = TOD(data={"atmosphere": rows}, dets=inst.dets.subset(np.arange(lo, hi)), coords=None, units="pW", metadata={})
        full = sim._gather(obs, tod)
        ok = bool(np.array_equal(full.data["atmosphere"][:, 0], np.arange(inst.dets.n, dtype=np.float32))) and full.dets.n == inst.dets.n
        q.put((rank, sim.shard, (lo, hi), geom, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_simulation_host_logic_gloo_world2():
    """Simulation(shard="auto") under a 2-rank group: disjoint covering row blocks, identical
    geometry on both ranks, and run(gather=True)'s all-gather restores the [ndet, nt] array."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_sim_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, b0, g0, ok0), (r1, s1, b1, g1, ok1) = results
    assert s0 == (0, 2) and s1 == (1, 2)
    assert b0[0] == 0 and b0[1] == b1[0] and b1[1] == 40
    assert g0 == g1 and ok0 and ok1
