#!/usr/bin/env python3
"""Benchmark of the atmosphere -> TOD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config atlast_10k] [--scaling strong|weak]

One step = one full pass of the path over one observation's worth of synthetic
input for this rank's detectors: generate + smooth the turbulent screens
(Philox / Hermitian FFT / fused Gaussian), fused pointing + layer gather + emission
at the coarse rate, not-a-knot spline solve + cubic upsample to the sample rate (one
kernel) -> float32 TOD in HBM.  All inputs are resident in HBM before the timed region.

Launch.  ``python bench.py --gpus N`` from a plain shell starts its own N ranks: the parent
process -- before it touches a GPU -- runs ``python -m torch.distributed.run --nproc-per-node N
bench.py ...`` as a CHILD, lets rank 0's JSON line through and exits with the child's status.
Started by torch.distributed.run itself (WORLD_SIZE set) it is one of the ranks.

Metric (BASELINE.json): detector-samples/s = n_det x n_t x steps / time, summed
over ranks.  N = 1 is the named configuration on one GPU (atlast_50k: its per-GPU share of an
8-GPU run, 1/8 of the detectors -- ``config.weak_unit``).  N > 1:
  * atlast_10k (BASELINE config 4) is STRONG scaling: the configuration's 10 000
    detectors in total, sharded in contiguous row blocks; every rank
    regenerates the (small) screens from the same Philox key, so the data path has no
    collective.  The one RCCL all-gather of the final TOD the north star names runs
    through the C ABI (mrx_allgather_tod, in place in the full [n_det, T] buffer) and is
    timed on its OWN clock (median of the repetitions, max over ranks).  `value` is the
    configuration AS STATED -- synthesis + gather, the two times added --, `value_synthesis_only`
    the synthesis alone; when the gather could not be timed `value` is the synthesis alone
    and `config.parallelism` says so.
  * atlast_50k (config 5) is WEAK scaling (the configuration per GPU; its 288 GB TOD
    cannot be gathered onto one GPU).  --scaling overrides either default.

Prints ONE JSON line on rank 0, including
  roofline      : the dominant kernel -- the one-launch synthesis (atm_tod_kernel: sampler and writer roles, the HBM-bound
                  streaming write sets its time) where DevicePath.run() takes that form, else the writer (spline solve + cubic
                  upsample) --, timed live with events on the launch stream
  second_kernel : the same for the sampler (VALU / vector-memory-latency bound)
  cpu_baseline  : the numpy/scipy oracle on a detector subset, on one host core and on
                  as many processes as the job has cores (physical cores, or the cgroup's CPU quota); parity of the GPU rows against it, on the
                  loading and on the fluctuation alone (per-detector mean removed)
  frontend      : wall time of what a user calls, Simulation(...).run(), set-up and run
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak (spec); ~6300 achievable
TRAFFIC_FILES = ("r06_traffic.json", "r06_traffic_writer.json", "r05_traffic.json", "r05_traffic_writer.json", "r04_traffic.json", "r04_traffic_writer.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")  # newest first


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="atlast_10k")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="N > 1: strong = the named configuration's detectors in total (default for atlast_10k), "
                    "weak = the named configuration per GPU (default for atlast_50k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dets", type=int, default=2048, help="detector rows of the one-core CPU-baseline sample")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes of the all-core CPU baseline (0 = the physical cores, or the cgroup's CPU quota where smaller)")
    ap.add_argument("--no-frontend", action="store_true", help="skip the Simulation(...).run() wall-clock section")
    ap.add_argument("--no-screens-in-step", action="store_true", help="time TOD synthesis only (screens generated once)")
    ap.add_argument("--shard-screens", action="store_true",
                    help="N > 1: each rank generates its round-robin share of the layers and the owners broadcast them "
                    "(mrx_exchange_screens over RCCL; default: every rank regenerates all layers, 0.28 ms, cheaper than the exchange)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the launch on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-allgather", action="store_true", help="skip the separately timed all-gather at N > 1")
    ap.add_argument("--gather-algo", choices=["allgather", "p2p", "both"], default="both",
                    help="the all-gather of the TOD at N > 1: RCCL's ncclAllGather (ring), grouped sends to every peer over its own "
                         "xGMI link (p2p), or both timed in turn -- `value` then takes the FASTER one whose rows arrived bit-identical "
                         "(select_gather); default both")
    ap.add_argument("--nccl-algo", default=None, help="exported as NCCL_ALGO before the communicator is made (e.g. Ring, Tree)")
    ap.add_argument("--gather-reps", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=None, help="detector blocks of the pipelined TOD synthesis (default: DevicePath.default_blocks(); 1 = serial)")
    ap.add_argument("--block-shares", default=None, help="A/B: relative sizes of the detector blocks, e.g. 1,2,2,2 (default: equal blocks)")
    ap.add_argument("--synth-wgs-per-cu", type=int, default=0,
                    help="A/B: resident workgroups per CU of the one-launch synthesis (MRX_OPT_SYNTH_WGS_PER_CU; 0 = as many as fit)")
    ap.add_argument("--lookahead", action="store_true",
                    help="A/B: let successive steps overlap (DevicePath.enable_lookahead: the next step's screens and first samplers run "
                    "beside this step's last writers; the timer brackets all K steps behind a sync either way).  Measured in round 4: "
                    "2.38 against 2.34 ms (atlast_10k), 11.8 against 11.5 (atlast_50k) -- the sampler chain beside the writers becomes the "
                    "critical path -- so the default stays one step after the other")
    ap.add_argument("--shard-of", type=int, default=0,
                    help="development, N = 1 only: run alone on this GPU the rows rank 0 of a run on this many GPUs would take "
                         "(the per-rank step of an N-GPU run, rehearsed on one; the line says so in config.workload)")
    ap.add_argument("--print-launch", action="store_true", help="print the child launch command of --gpus N as JSON and exit (no GPU needed)")
    return ap.parse_args(argv)


# ---- self-launch -----------------------------------------------------------------------

def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_gpus, argv, port):
    """The child command of ``bench.py --gpus N`` started from a plain shell: one rank per GPU on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def needs_launch(args, environ):
    """N > 1 requested and this process is not already a rank of a torch.distributed.run job."""
    return args.gpus > 1 and int(environ.get("WORLD_SIZE", "1")) == 1 and "TORCHELASTIC_RUN_ID" not in environ


def launch(args, argv):
    """Start the ranks as a child job; nothing here touches the GPU.  stdout is inherited, so rank 0's
    JSON line is this process's output; returns the child's exit status."""
    import subprocess

    cmd = launch_command(args.gpus, [a for a in argv if a != "--print-launch"], _free_port())
    if args.print_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


# ---- CPU baseline ----------------------------------------------------------------------

def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        cores = set()
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if cores:
            return len(cores)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def _cpu_quota():
    """CPUs this process may keep busy: the cgroup's quota (cpu.max, or the v1 files) where one is set -- a GPU box leases
    16 CPUs' worth per GPU although it shows the host's 256 --, else the affinity mask."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        if q != "max":
            return max(1, int(-(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = int(f.read())
        if q > 0:
            return max(1, -(-q // p))
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(problem, screens, rows):
    """Time the oracle (kind 'port') on detector rows ``rows`` (a slice), full duration."""
    import numpy as np

    from oracle import hotpath

    sub = dict(problem)
    for key in ("offsets", "band_index", "m00", "gain"):
        sub[key] = None if problem.get(key) is None else problem[key][rows]
    sub["layers"] = [dict(l, values=s) for l, s in zip(problem["layers"], screens)]
    n = len(sub["offsets"])
    # in blocks of 512 rows so the float64 intermediates of scipy stay small
    tods, dt = [], 0.0
    for a in range(0, n, 512):
        blk = dict(sub)
        for key in ("offsets", "band_index", "m00", "gain"):
            blk[key] = None if sub[key] is None else sub[key][a : a + 512]
        t0 = time.perf_counter()
        tods.append(hotpath.run_path(blk))
        dt += time.perf_counter() - t0
    tod = np.concatenate(tods)
    assert tod.shape == (n, len(problem["t"]))
    return tod, dt


def cpu_baseline_all_cores(config, n_total, screens, n_procs, rows_per_proc=512):
    """The same oracle on ``n_procs`` processes at once, ``rows_per_proc`` detector rows each (rows taken
    cyclically from the configuration), every process timing its own share behind a common barrier;
    the aggregate rate is all rows over the slowest process.  The workers never touch the GPU."""
    import multiprocessing as mp
    import tempfile

    import numpy as np

    from oracle import cpu_pool

    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(shm, f"mrx_bench_screens_{os.getpid()}.npy")
    np.save(path, np.stack(screens))
    ctx = mp.get_context("spawn")
    barrier, queue = ctx.Barrier(n_procs), ctx.Queue()
    procs = [ctx.Process(target=cpu_pool.worker, args=(config, n_total, w, rows_per_proc, path, barrier, queue), daemon=True)
             for w in range(n_procs)]
    t0 = time.perf_counter()
    try:
        for p in procs:
            p.start()
        got = [queue.get(timeout=600) for _ in procs]
        for p in procs:
            p.join(30)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        try:
            os.remove(path)
        except OSError:
            pass
    errors = [g for g in got if "error" in g]
    if errors:
        raise RuntimeError(errors[0]["error"])
    slowest = max(g["seconds"] for g in got)
    return {"rows": n_procs * rows_per_proc, "seconds_slowest_process": slowest, "seconds_with_startup": time.perf_counter() - t0,
            "processes": n_procs}


def fluct_err(a, b):
    """max |fluctuation difference| / max |reference fluctuation|, per-detector mean removed."""
    import numpy as np

    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    fa, fb = a - a.mean(axis=-1, keepdims=True), b - b.mean(axis=-1, keepdims=True)
    return float(np.abs(fa - fb).max() / max(np.abs(fb).max(), 1e-300))


def select_gather(variants):
    """Which all-gather variant `value` takes at N > 1: the FASTEST of those that completed and whose probe found the next
    rank's rows bit-identical (the first SCALE record must not print the slow one by default, nor lose the
    line to a variant that fails).  ``variants``: {algo: entry} in the order they ran; an entry is the timing dict of the
    gather section (``ms``, ``rows_of_next_rank_bit_identical``) or ``{"error": ...}``.  Returns (algo or None, reason)."""
    good = [(e["ms"], k) for k, e in variants.items() if "error" not in e and e.get("rows_of_next_rank_bit_identical") and e.get("ms", 0) > 0]
    if not good:
        why = "; ".join(f"{k}: {'error: ' + str(e['error'])[:80] if 'error' in e else 'rows of the next rank did not arrive bit-identical'}"
                        for k, e in variants.items()) or "no variant ran"
        return None, "no verified variant (" + why + "): value stays the synthesis alone"
    ms, algo = min(good)
    others = [k for k in variants if k != algo]
    if not others:
        return algo, "the only variant run"
    notes = []
    for k in others:
        e = variants[k]
        notes.append(f"{k} failed ({str(e['error'])[:60]})" if "error" in e else
                     f"{k} did not verify" if not e.get("rows_of_next_rank_bit_identical") else f"{k} took {e['ms']:.2f} ms")
    return algo, f"fastest verified variant: {ms:.2f} ms; " + ", ".join(notes)


# ---- what a user calls -----------------------------------------------------------------

def frontend_timing(device):
    """Wall clock of Simulation(atlast_10k-shaped instrument, device_output=True): set-up (Atmosphere.initialize:
    hull, rotation search, layer geometry, device upload -- atmosphere/atmosphere.py:81-281) and run()
    (sim/simulation.py:201-272), K_RJ (the reference's default units): atmosphere only, with the detector noise, and
    the north star's literal call -- Simulation(instrument, plans, site, map=...) (sim/simulation.py:76-92) with a
    1024^2 map in the ra/dec frame -- without and with the noise.  ``atmosphere_ms`` of a map row is the span of the
    atmosphere's own launches inside that run (events around _simulate_atmosphere + _compute_atmospheric_loading)."""
    import numpy as np
    import torch

    from maria_amd import map as mmap
    from maria_amd.instrument import Band, Detectors, Instrument, Site
    from maria_amd.sim import Plan, Simulation, sky_transform_stack

    band = Band(center=150e9, width=30e9, shape="top_hat", name="f150")
    inst = Instrument(Detectors.hexagon(10000, 2.0, [band], primary_size=50.0))
    plan = Plan.daisy(start_time=1.7e9, duration=600.0, sample_rate=400.0, scan_center=(45.0, 60.0), radius=0.5, speed=0.5)
    site = Site(altitude=5000.0)
    # the scanned patch's centre in the map's frame: xyz(az, el) @ M[t] = xyz(ra, dec)
    M = sky_transform_stack(plan.time[::4000], site.latitude, site.longitude)
    az, el = plan.phi[::4000], plan.theta[::4000]
    xyz = np.einsum("ti,tij->tj", np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1), M).mean(axis=0)
    centre = (float(np.degrees(np.arctan2(xyz[1], xyz[0]) % (2 * np.pi))), float(np.degrees(np.arcsin(xyz[2] / np.linalg.norm(xyz)))))
    n = 1024
    X, Y = np.meshgrid(np.linspace(-1, 1, n), np.linspace(-1, 1, n))
    sky = mmap.ProjectionMap((1e-3 * (np.exp(-((X - 0.2) ** 2 + (Y + 0.1) ** 2) / 0.05) + 0.3 * np.sin(9 * X) * np.cos(7 * Y))).astype(np.float32),
                             nu=150e9, width=4.0, center=centre, frame="ra/dec")
    out = {"instrument": f"{inst.dets.n} det hexagon, 2 deg field, 1 band; daisy 600 s @ 400 Hz; units K_RJ; device_output; "
                         f"map rows: {n}^2 K_RJ map, 4 deg wide, ra/dec frame"}
    for name, noise, with_map in (("atmosphere", False, False), ("atmosphere_noise", True, False),
                                  ("atmosphere_map", False, True), ("atmosphere_map_noise", True, True)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim = Simulation(inst, plan, site, atmosphere="2d", map=sky if with_map else None, noise=noise, device_output=True,
                         noise_seed=1, progress_bars=False)
        torch.cuda.synchronize()
        init_s = time.perf_counter() - t0
        atm_spans = []
        if with_map:  # the atmosphere's share of the run: events around its two stages (screens; sampling + TOD)
            first, second = sim._simulate_atmosphere, sim._compute_atmospheric_loading

            def timed_first(*a, **k):
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                atm_spans.append([e0, None])
                return first(*a, **k)

            def timed_second(*a, **k):
                r = second(*a, **k)
                atm_spans[-1][1] = torch.cuda.Event(enable_timing=True)
                atm_spans[-1][1].record()
                return r

            sim._simulate_atmosphere, sim._compute_atmospheric_loading = timed_first, timed_second
        walls, spans = [], []
        for _ in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record()
            (tod,) = sim.run()
            e1.record()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            spans.append(e0.elapsed_time(e1))
            del tod
        # (median of 8 runs after the first; single runs show 50-80 ms host stalls on a shared box)
        out[name] = {"init_s": init_s, "run_ms": 1e3 * float(np.median(walls[1:])), "run_ms_min": 1e3 * float(np.min(walls[1:])),
                     "first_run_ms": 1e3 * walls[0], "gpu_span_ms": float(np.median(spans[1:]))}
        if atm_spans:
            out[name]["atmosphere_ms"] = float(np.median([a.elapsed_time(b) for a, b in atm_spans[1:]]))
        del sim
        torch.cuda.empty_cache()
    return out


# ---- the benchmark ---------------------------------------------------------------------

def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start with `python bench.py --gpus N` or torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if args.nccl_algo:
        os.environ["NCCL_ALGO"] = args.nccl_algo
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)
    red_device = device if args.backend == "nccl" else "cpu"

    from maria_amd import synthetic
    from maria_amd._lib import Context
    from maria_amd.dist import TodGather, exchange_layer_screens, layers_of_rank, shard_bounds
    from maria_amd.pipeline import DevicePath

    scaling = args.scaling or ("weak" if args.config == "atlast_50k" else "strong")
    n_config = synthetic.CONFIGS[args.config]["n_det"]
    weak_unit = None
    if args.config == "atlast_50k" and scaling == "weak":
        # BASELINE config 5 is stated on 8 GPUs (6 250 detectors and a 36 GB TOD each; the whole 288 GB
        # TOD fits no single GPU): the weak-scaling unit is that per-GPU share, on any number of GPUs
        n_total = (n_config // 8) * world
        weak_unit = "atlast_50k/8: 6250 detectors per GPU"
    else:
        n_total = n_config if (scaling == "strong" or world == 1) else n_config * world
    problem = synthetic.config_problem(args.config, n_det=n_total)
    lo, hi = shard_bounds(n_total, world, rank)
    if args.shard_of > 1:
        if world != 1:
            raise SystemExit("--shard-of rehearses one rank's rows on one GPU: use it with --gpus 1")
        lo, hi = shard_bounds(n_total, args.shard_of, 0)
    path = DevicePath(problem, device=device, det_slice=slice(lo, hi))
    D, T, Ta = path.D, path.T, path.Ta
    L = len(problem["layers"])

    # the output buffer: with the all-gather, the whole [n_det, T] TOD with this rank's shard
    # written straight into its rows; otherwise the shard alone
    gatherer, full = None, None
    want_comm = world > 1 and args.backend == "nccl" and ((not args.no_allgather and scaling == "strong") or args.shard_screens)
    gather_note = None
    if want_comm:
        # The communicator comes up in a helper thread with a deadline and a context of its own: a
        # rendezvous that never completes must cost the all-gather figure, not the benchmark line.
        # The ranks then agree (a torch.distributed reduction) on whether every one of them has it.
        import threading

        box = {}
        comm_ctx = Context(local_rank)
        try:  # the id travels over torch.distributed, from the thread that owns the group
            unique_id = TodGather.exchange_unique_id(comm_ctx, world, rank)
        except Exception as exc:  # pragma: no cover - depends on the node
            unique_id, box["note"] = None, f"{type(exc).__name__}: {exc}"[:300]

        def make():  # ncclCommInitRank inside libmrx: the only part that can hang
            try:
                torch.cuda.set_device(local_rank)  # the current device is per thread
                box["gatherer"] = TodGather(comm_ctx, n_total, world, rank, unique_id=unique_id)
            except Exception as exc:  # pragma: no cover - depends on the node
                box["note"] = f"{type(exc).__name__}: {exc}"[:300]

        th = None
        if unique_id is not None:
            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(90.0)
            if th.is_alive():
                box["note"] = "the RCCL communicator did not come up within 90 s"
        ok = torch.tensor([1 if ("gatherer" in box and not (th and th.is_alive())) else 0], dtype=torch.int32, device=red_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gatherer = box["gatherer"]
            if not args.no_allgather and scaling == "strong":
                full = gatherer.full_buffer(T, device)
                tod = gatherer.my_rows(full)
        else:
            gather_note = box.get("note", "another rank could not create the communicator")
            if "gatherer" in box and not (th and th.is_alive()):  # this rank has one the job cannot use
                try:
                    box["gatherer"].close()
                except Exception:
                    pass
            gatherer, full = None, None
    if full is None:
        tod = torch.empty((D, T), dtype=torch.float32, device=device)
    own_layers = layers_of_rank(L, world, rank) if (args.shard_screens and world > 1) else None
    if own_layers is not None and args.backend == "nccl" and gatherer is None:
        own_layers = None  # no communicator: fall back to regenerating every layer on every rank

    def exchange(scr):
        """the owners broadcast their layers: on the stream the screens were generated on (the current one)"""
        if gatherer is not None:  # RCCL through the C ABI
            gatherer.ctx.set_stream(torch.cuda.current_stream())
            gatherer.exchange_screens(scr)
        else:  # gloo rehearsal
            exchange_layer_screens(scr)

    def screens():
        path.generate_screens(only=own_layers, exchange=exchange if own_layers is not None else None)

    writer_events = []
    # the form DevicePath.run() takes by itself: ONE launch (mrx_atm_synthesize: sampler and writer as two roles of one
    # grid) where it applies, else detector blocks pipelined on two streams, else the stages back to back
    one_launch = args.blocks is None and not args.block_shares and path.synthesize_applies()
    n_blocks = 1 if one_launch else args.blocks if args.blocks is not None else path.default_blocks()
    if args.block_shares:
        path.block_shares = [int(x) for x in args.block_shares.split(",")]
        n_blocks = len(path.block_shares)

    def step(ev=None):
        """One pass: screens, then the TOD synthesis (DevicePath.run: one launch, or the sampler of block b+1 beside the
        writer of block b on two streams)."""
        if ev: ev[0].record()
        if not args.no_screens_in_step:
            screens()
        if ev: ev[1].record()
        if one_launch:
            path.run(tod, writer_events=writer_events if ev else None)
        else:
            path.run(tod, blocks=n_blocks, writer_events=writer_events if (ev and n_blocks > 1) else None)
        if ev: ev[2].record()

    def serial_step(sev):
        """The same launches back to back on one stream, for the per-stage breakdown only."""
        if n_blocks > 1:
            path._run_pipelined(tod, n_blocks, serial_events=sev)
        else:
            tev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tev[0].record()
            path.sample()
            tev[1].record()
            path.upsample_fused(tod)
            tev[2].record()
            sev.append(tev)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    screens()
    # successive steps overlap like the observations of one Simulation.run() (sim/simulation.py:201-211): the same
    # launches, ordered by events instead of by one stream
    lookahead = args.lookahead and path.enable_lookahead()
    if args.synth_wgs_per_cu:
        from maria_amd import _lib as _mlib
        path.ctx.set_option(_mlib.OPT_SYNTH_WGS_PER_CU, args.synth_wgs_per_cu)
    for _ in range(args.warmup):
        step()
    barrier()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    gc.collect()
    gc.disable()  # (a collection inside a 25-ms timed region would be a tenth of it)
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    flags = path.check_flags()

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # outside the timed region: the screens alone (on the stream they run on), then the stages back to back on one
    # stream, for the breakdown
    scr_ms = []
    for k in range(4):
        gen_stream = path._la["stream"] if lookahead else torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(gen_stream)
        screens()
        e1.record(gen_stream)
        torch.cuda.synchronize()
        scr_ms.append(e0.elapsed_time(e1))
    screens_alone_ms = float(np.median(scr_ms[1:]))
    # (round 6: one untimed pass, then the MEDIAN of seven -- the mean of three passes, the first of them cold, scattered by
    # +-4 % from run to run on one box, more than the differences it was read for: profiles/r06_writer_bisect.txt)
    n_serial = 7
    serial_step([])
    torch.cuda.synchronize()
    sev = []
    for k in range(n_serial):
        serial_step(sev)
    torch.cuda.synchronize()
    # per stage: the sum over a step's block launches in a pass, the median over the passes
    per_pass = np.array([[t[i].elapsed_time(t[i + 1]) for i in range(2)] for t in sev]).reshape(n_serial, -1, 2).sum(axis=1)
    serial_ms = np.median(per_pass, axis=0)
    serial_up_launch_ms = float(np.median([t[1].elapsed_time(t[2]) for t in sev]))
    step_ms = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(2)] for k in range(args.steps)]).mean(axis=0)
    # the dominant kernel, timed live in the timed region on the stream it runs on: one launch
    # per detector block when the step is pipelined (its rows x T samples each)
    n_launch = max(1, n_blocks if n_blocks > 1 else 1)
    if writer_events:
        up_ms = float(np.mean([a.elapsed_time(b) for a, b in writer_events]))
        rows_per_launch = D / n_launch
    else:
        up_ms, rows_per_launch = serial_up_launch_ms, D / n_launch
    writer_bytes = 4.0 * rows_per_launch * T + 4.0 * rows_per_launch * Ta + 8.0 * T  # TOD write + coarse loading read + sample times read
    # one launch: the kernel IS the atmosphere -> TOD path (SURVEY 8(d)'s B_alg: TOD + coarse loading written and read +
    # every screen once + inputs); otherwise the writer's launch
    up_bytes = float(path.algorithmic_bytes()) if one_launch else writer_bytes
    achieved = up_bytes / (up_ms * 1e-3) / 1e9
    alone = writer_bytes / (serial_up_launch_ms * 1e-3) / 1e9
    # the sampler: 4 B/det-step written + each screen read once + inputs (it is not HBM-bound)
    sm_ms = float(serial_ms[0])
    sm_bytes = 4.0 * D * Ta + 4.0 * sum(len(l["extrusion"]) * len(l["cross_section"]) for l in problem["layers"]) + 8.0 * Ta + 8.0 * D

    n_step = n_total if world > 1 and scaling == "strong" else D * world
    result = {
        "metric": "detector-samples/sec (ndet x nt), atmosphere -> TOD synthesis",
        "value": n_step * T * args.steps / elapsed,
        "unit": "detector-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": scaling if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {n_step} det x {T} samples ({problem['fs']:.0f} Hz), Ta={Ta}, "
            f"{L} layers of {len(problem['layers'][0]['extrusion'])}^2 screens, "
            f"{len(problem['tables'])} band(s), in total over {world} GPU(s)"
            + (f" -- the rows of rank 0 of a {args.shard_of}-GPU run of {n_total} detectors, run ALONE on one GPU (--shard-of)" if args.shard_of > 1 else ""),
            "n_det_total": n_step,
            "n_det_per_gpu": D,
            "n_samples": T,
            "screens_in_step": not args.no_screens_in_step,
            "screens": "sharded by layer + broadcast by the owners" if own_layers is not None else "regenerated on every rank from the Philox key",
            "parallelism": (f"detector-sharded x{world}, no data-path collective" if world == 1 or scaling == "weak" else
                            f"detector-sharded x{world}; value is the SYNTHESIS ONLY (the all-gather of the TOD was not timed: see allgather)"),
            "steps_overlap": bool(lookahead),
            "steps_overlap_note": "successive steps overlap as the observations of one Simulation.run() do: the next step's screens (a stream "
            "and a buffer set of their own) and samplers start while this step's writers stream; every step makes the same launches, "
            "the K steps are timed as a whole behind a device synchronisation (bench.py --lookahead; off by default)",
        },
        # the same K steps on the GPU's own clock (events around every step, on the stream the step runs on): the
        # wall-clock figure above should sit within a few per cent of it; a host hiccup inside the timed region shows here
        "gpu_ms_per_step": float(ev[0][0].elapsed_time(ev[-1][2])) / args.steps,
        "stage_ms": {
            "screens": screens_alone_ms if lookahead else float(step_ms[0]),
            "tod_synthesis_pipelined": float(step_ms[1]),
            "note": ("screens: alone on their stream, outside the timed region; tod_synthesis_pipelined: between the events around "
                     "DevicePath.run() on the caller's stream -- with overlapping steps these intervals overlap too and do not add up to ms_per_step")
            if lookahead else "events on the caller's stream around the two stages of every timed step",
            "serial_breakdown": {"sample": sm_ms, "upsample_with_spline_solve": float(serial_ms[1]),
                                 "note": "the same block launches back to back on one stream, outside the timed region; sums over the blocks, "
                                         "median of seven passes after an untimed one"},
            "detector_blocks": n_launch,
            "form": ("one launch (mrx_atm_synthesize): sampler and writer as two roles of one grid, hand-over on the device"
                     if one_launch else "detector blocks pipelined on two streams" if n_blocks > 1 else "stages back to back on one stream"),
        },
        "path_hbm_gbps": path.algorithmic_bytes() / (float(ev[0][0].elapsed_time(ev[-1][2])) / args.steps * 1e-3) / 1e9,
        "roofline": {
            "kernel": "atm_tod_kernel" if one_launch else "spline_upsample_fused_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "bytes_per_launch": up_bytes,
            "ms_per_launch": up_ms,
            "launches_per_step": n_launch,
            "note": ("the whole atmosphere -> TOD synthesis in one launch, timed in the timed region: its algorithmic bytes are the "
                     "path's (TOD + coarse loading + screens + inputs); the stand-alone writer (spline_upsample_fused_kernel, "
                     "serial breakdown) reaches frac_alone on its own bytes") if one_launch else
                    "timed in the timed region, where each launch shares the chip with the next block's sampler; "
                    "the same launches back to back on one stream (serial breakdown) reach frac_alone",
            "frac_alone": alone / HBM_PEAK_GBPS,
        },
        "second_kernel": {
            "kernel": "atm_sample_px_kernel",
            "bound": "valu + vector-memory latency (not hbm): see DESIGN 3.2 and profiles/r04_50k_kernel_pmc.txt",
            "ms_per_step": sm_ms,
            "launches_per_step": n_launch,
            "note": "sum over the step's block launches run back to back on one stream (serial breakdown)",
            "bytes_per_step": sm_bytes,
            "achieved_GBps": sm_bytes / (sm_ms * 1e-3) / 1e9,
            "layer_samples_per_s": D * Ta * L / (sm_ms * 1e-3),
        },
        "flags": int(flags),
    }
    if weak_unit:
        result["config"]["weak_unit"] = weak_unit

    # HBM bytes per launch of the dominant kernel from the PMC counters: collected in
    # separate rocprofv3 --pmc passes (gpurun refuses --pmc inside an ordinary run), stored
    # with their method under profiles/, and quoted here for the matching configuration
    for name in TRAFFIC_FILES:
        traffic_file = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(traffic_file):
            continue
        with open(traffic_file) as f:
            tr = json.load(f)
        if tr.get("config") == args.config and tr.get("kernel") == result["roofline"]["kernel"] and world == 1:
            # per launch of the profiled run; scaled by rows when this run cuts the shard differently
            result["roofline"]["traffic"] = tr["hbm_bytes_per_launch"] * tr.get("launches_per_step", 1) / n_launch
            result["roofline"]["traffic_source"] = tr["source"]
            break

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import threadpoolctl

        n_sub = min(args.cpu_dets, D)
        scr = [b[0].cpu().numpy() for b in path._layer_bufs]
        with threadpoolctl.threadpool_limits(limits=1):
            ref, cpu_s1 = cpu_baseline(problem, scr, slice(0, n_sub))
        got = tod[:n_sub].cpu().numpy()
        err = float(np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max())
        result["cpu_baseline"] = {
            "value": n_sub * T / cpu_s1,
            "unit": "detector-samples/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {n_sub} of {D} detector rows, full {T} samples, screens given (sampling + emission + cubic "
            f"upsample; numpy/scipy with BLAS/OpenMP pools limited to 1 thread): {cpu_s1:.2f} s",
            "host_cpus": os.cpu_count(), "cpu_quota": _cpu_quota(),
            "physical_cores": _physical_cores(),
            "cpu_model": _cpu_model(),
            "parity_max_rel_err_vs_gpu": err,
            # the turbulent signal alone: per-detector mean removed, relative to the largest fluctuation
            "parity_fluct_rel_err": fluct_err(got, ref),
        }
        del ref, got
        try:
            # all the cores this job may use: the physical cores, or the cgroup's CPU quota where that is smaller (one
            # process per core; more processes than the quota only take turns)
            n_procs = args.cpu_procs or max(1, min(_physical_cores(), _cpu_quota()))
            allc = cpu_baseline_all_cores(args.config, n_total, scr, n_procs)
            result["cpu_baseline"]["all_cores"] = {
                "value": allc["rows"] * T / allc["seconds_slowest_process"], "unit": "detector-samples/s", "cores": n_procs,
                "cpu_quota": _cpu_quota(), "seconds": allc["seconds_slowest_process"],
                "sample": f"{n_procs} processes x 512 detector rows each (rows taken cyclically), full {T} samples, one "
                "numpy/scipy thread per process, timed behind a common barrier; rate = all rows / slowest process",
            }
        except Exception as exc:  # pragma: no cover - depends on the host
            result["cpu_baseline"]["all_cores"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    if rank == 0 and world == 1 and not args.no_frontend and args.config == "atlast_10k":
        del tod
        path._pipe = None
        torch.cuda.empty_cache()
        try:
            result["frontend"] = frontend_timing(device)
        except Exception as exc:  # pragma: no cover
            result["frontend"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}

    # The one all-gather of the final TOD over xGMI (north star / BASELINE config 4), through
    # the C ABI, on its own clock: reported beside `value`, never part of it.  A failure here
    # must not lose the benchmark line, and a collective that never completes must not look
    # like success: after 180 s the watchdog prints the line (rank 0) and exits non-zero.
    if world > 1 and not args.no_allgather:
        import threading

        finished = threading.Event()

        def watchdog():
            if not finished.wait(180.0):
                if rank == 0:
                    result.setdefault("allgather", {})["error"] = "no completion within 180 s"
                    print(json.dumps(result), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if gatherer is not None and full is not None:
                gatherer.ctx.set_stream(torch.cuda.current_stream())
                algos = ["allgather", "p2p"] if args.gather_algo == "both" else [args.gather_algo]
                transports = {"allgather": "RCCL ncclAllGather via libmrx mrx_allgather_tod",
                              "p2p": "RCCL grouped ncclSend/ncclRecv to every peer via libmrx mrx_allgather_tod_p2p"}
                variants = {}
                for n_algo, algo in enumerate(algos):
                    # every variant in a try of its own, agreed on by all ranks: one that throws on some rank is recorded and
                    # the next still runs -- the line is never lost to a variant
                    entry, note = None, None
                    try:
                        if n_algo > 0:  # the variant before has filled the buffer: move the rows again
                            full.zero_()
                            path.run(tod, blocks=None if one_launch else n_blocks)
                        barrier()
                        gatherer.gather(full, algo=algo)  # warm-up outside the timed repetitions: RCCL's first call sets up its channels
                        barrier()
                        times = []
                        for _ in range(max(1, args.gather_reps)):
                            barrier()
                            t0 = time.perf_counter()
                            gathered = gatherer.gather(full, algo=algo)
                            barrier()
                            times.append(time.perf_counter() - t0)
                        # rows of another rank must have arrived: compare with what this rank would have produced there
                        other = (rank + 1) % world
                        olo, ohi = shard_bounds(n_total, world, other)
                        probe = DevicePath(problem, device=device, det_slice=slice(olo, min(olo + 16, ohi)))
                        probe.set_screens(path._gen_screens)
                        check = probe.run()
                        same = bool(torch.equal(check, gathered[olo : olo + check.shape[0]]))
                        dt = float(np.median(times))
                        nbytes = gatherer.bytes_received(T)
                        entry = {
                            "ms": 1e3 * dt, "received_GB_per_rank": nbytes / 1e9,
                            "in_timed_region": False, "in_place": True, "reps": len(times), "warmed_up": True,
                            "transport": transports[algo], "NCCL_ALGO": os.environ.get("NCCL_ALGO"),
                            "rows_of_next_rank_bit_identical": same,
                        }
                    except Exception as exc:  # pragma: no cover - depends on the node
                        note = f"{type(exc).__name__}: {exc}"[:300]
                    # the ranks agree: slowest time, every probe identical, nobody failed
                    agree = torch.tensor([entry["ms"] if entry else -1.0, 1.0 if (entry and entry["rows_of_next_rank_bit_identical"]) else 0.0,
                                          0.0 if entry else 1.0], dtype=torch.float64, device=red_device)
                    hi3, lo3 = agree.clone(), agree.clone()
                    dist.all_reduce(hi3, op=dist.ReduceOp.MAX)
                    dist.all_reduce(lo3, op=dist.ReduceOp.MIN)
                    if float(hi3[2].item()) > 0.0 or entry is None:
                        variants[algo] = {"error": note or "another rank failed in this variant", "transport": transports[algo]}
                        continue
                    dt = float(hi3[0].item()) * 1e-3
                    entry.update(ms=1e3 * dt, GBps_per_rank=entry["received_GB_per_rank"] / dt,
                                 rows_of_next_rank_bit_identical=bool(float(lo3[1].item()) > 0.0),
                                 step_plus_gather_ms=1e3 * elapsed / args.steps + 1e3 * dt,
                                 value_with_gather=n_step * T / (elapsed / args.steps + dt))
                    variants[algo] = entry
                chosen, reason = select_gather(variants)
                for algo, entry in variants.items():
                    entry["in_value"] = algo == chosen
                result["allgather_variants"] = variants
                result["allgather"] = dict(variants[chosen], algo=chosen, chosen_because=reason) if chosen else {"skipped": reason}
                if chosen:
                    # BASELINE config 4 as stated ends with every GPU holding the whole TOD: `value` is the
                    # synthesis AND the gather (each on its own clock, added); the synthesis alone stays beside it
                    entry = variants[chosen]
                    result["value_synthesis_only"] = result["value"]
                    result["ms_per_step_synthesis_only"] = result["ms_per_step"]
                    result["value"] = entry["value_with_gather"]
                    result["ms_per_step"] = entry["step_plus_gather_ms"]
                    result["config"]["parallelism"] = (
                        f"detector-sharded x{world}; synthesis has no collective; ONE all-gather of the TOD per step over xGMI "
                        f"({entry['transport']}: {reason}; in place, {entry['received_GB_per_rank']:.2f} GB received per rank), timed on its "
                        "own clock (median of the repetitions after a warm-up, max over ranks) and ADDED to the step: "
                        "value = detector-samples / (synthesis + gather)")
            elif args.backend != "nccl":
                # rehearsal on one device: the torch.distributed fallback on a small CPU slice
                from maria_amd.dist import all_gather_tod

                barrier()
                t0 = time.perf_counter()
                small = all_gather_tod(tod[:, :4096].cpu(), n_total)
                dt = time.perf_counter() - t0
                result["allgather"] = {"rehearsal": True, "ms": 1e3 * dt, "shape": list(small.shape), "in_timed_region": False,
                                       "transport": "torch.distributed gloo on a 4096-sample CPU slice (rehearsal only)"}
            else:
                result["allgather"] = {"skipped": gather_note or ("weak scaling: the gathered TOD does not fit one GPU" if scaling == "weak" else "disabled")}
        except Exception as exc:  # pragma: no cover - depends on the node
            result["allgather"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        finished.set()

    if rank == 0:
        print(json.dumps(result), flush=True)
    if gatherer is not None:
        try:
            gatherer.close()
        except Exception:
            pass
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.print_launch or needs_launch(args, os.environ):
        return launch(args, argv)
    run(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
