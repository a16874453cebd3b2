#!/usr/bin/env python3
"""Benchmark of the atmosphere -> TOD hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config atlast_10k] [--scaling strong|weak]

One step = one full pass of the path over one observation's worth of synthetic
input for this rank's detectors: generate + smooth the turbulent screens
(Philox / Hermitian FFT / fused Gaussian), fused pointing + layer gather + emission
at the coarse rate, not-a-knot spline solve + cubic upsample to the sample rate (one
kernel) -> float32 TOD in HBM.  All inputs are resident in HBM before the timed region.

Metric (BASELINE.json): detector-samples/s = n_det x n_t x steps / time, summed
over ranks.  N = 1 is the named configuration on one GPU.  N > 1:
  * atlast_10k (BASELINE config 4) is STRONG scaling: the configuration's 10 000
    detectors in total, sharded in contiguous row blocks; every rank
    regenerates the (small) screens from the same Philox key, so the data path has no
    collective.  The one RCCL all-gather of the final TOD the north star names runs
    through the C ABI (mrx_allgather_tod, in place in the full [n_det, T] buffer) and is
    timed on its OWN clock, reported beside `value`, never folded into it.
  * atlast_50k (config 5) is WEAK scaling (the configuration per GPU; its 288 GB TOD
    cannot be gathered onto one GPU).  --scaling overrides either default.

Prints ONE JSON line on rank 0, including
  roofline      : the dominant kernel (spline solve + cubic upsample, HBM-bound streaming write),
                  timed live with events on the launch stream
  second_kernel : the same for atm_sample_kernel (VALU / vector-memory-issue bound)
  cpu_baseline  : the numpy/scipy oracle on a detector subset, on this host's cores,
                  with one BLAS/OpenMP thread and with all of them
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak (spec); ~6300 achievable
TRAFFIC_FILES = ("r02_traffic.json", "r01_traffic.json")  # newest first


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="atlast_10k")
    ap.add_argument("--scaling", choices=["strong", "weak"], default=None,
                    help="N > 1: strong = the named configuration's detectors in total (default for atlast_10k), "
                    "weak = the named configuration per GPU (default for atlast_50k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-dets", type=int, default=2048, help="detector rows of each CPU-baseline sample")
    ap.add_argument("--no-screens-in-step", action="store_true", help="time TOD synthesis only (screens generated once)")
    ap.add_argument("--shard-screens", action="store_true",
                    help="N > 1: each rank generates its round-robin share of the layers and the owners broadcast them "
                    "(default: every rank regenerates all layers, 0.28 ms, cheaper than any collective)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the launch on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-allgather", action="store_true", help="skip the separately timed all-gather at N > 1")
    ap.add_argument("--gather-reps", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=None, help="detector blocks of the pipelined TOD synthesis (default: DevicePath.default_blocks(); 1 = serial)")
    return ap.parse_args()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)
    red_device = device if args.backend == "nccl" else "cpu"

    from maria_amd import synthetic
    from maria_amd.dist import TodGather, exchange_layer_screens, layers_of_rank, shard_bounds
    from maria_amd.pipeline import DevicePath

    scaling = args.scaling or ("weak" if args.config == "atlast_50k" else "strong")
    n_config = synthetic.CONFIGS[args.config]["n_det"]
    if args.config == "atlast_50k" and scaling == "weak":
        # BASELINE config 5 is stated on 8 GPUs (6 250 detectors and a 36 GB TOD each; the whole 288 GB
        # TOD fits no single GPU): the weak-scaling unit is that per-GPU share, on any number of GPUs
        n_total = (n_config // 8) * world
    else:
        n_total = n_config if (scaling == "strong" or world == 1) else n_config * world
    problem = synthetic.config_problem(args.config, n_det=n_total)
    lo, hi = shard_bounds(n_total, world, rank)
    path = DevicePath(problem, device=device, det_slice=slice(lo, hi))
    D, T, Ta = path.D, path.T, path.Ta
    L = len(problem["layers"])

    # the output buffer: with the all-gather, the whole [n_det, T] TOD with this rank's shard
    # written straight into its rows; otherwise the shard alone
    gatherer, full = None, None
    want_gather = world > 1 and not args.no_allgather and scaling == "strong" and args.backend == "nccl"
    gather_note = None
    if want_gather:
        # The communicator comes up in a helper thread with a deadline: a rendezvous that never
        # completes must cost the all-gather figure, not the benchmark line.  The ranks then agree
        # (a torch.distributed reduction) on whether every one of them has it.
        import threading

        box = {}
        try:  # the id travels over torch.distributed, from the thread that owns the group
            unique_id = TodGather.exchange_unique_id(path.ctx, world, rank)
        except Exception as exc:  # pragma: no cover - depends on the node
            unique_id, box["note"] = None, f"{type(exc).__name__}: {exc}"[:300]

        def make():  # ncclCommInitRank inside libmrx: the only part that can hang
            try:
                torch.cuda.set_device(local_rank)  # the current device is per thread
                box["gatherer"] = TodGather(path.ctx, n_total, world, rank, unique_id=unique_id)
            except Exception as exc:  # pragma: no cover - depends on the node
                box["note"] = f"{type(exc).__name__}: {exc}"[:300]

        if unique_id is not None:
            th = threading.Thread(target=make, daemon=True)
            th.start()
            th.join(90.0)
            if th.is_alive():
                box["note"] = "the RCCL communicator did not come up within 90 s"
        ok = torch.tensor([1 if "gatherer" in box else 0], dtype=torch.int32, device=red_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            gatherer = box["gatherer"]
            full = gatherer.full_buffer(T, device)
            tod = gatherer.my_rows(full)
        else:
            gather_note = box.get("note", "another rank could not create the communicator")
            gatherer, full = None, None
    if full is None:
        tod = torch.empty((D, T), dtype=torch.float32, device=device)
    own_layers = layers_of_rank(L, world, rank) if (args.shard_screens and world > 1) else None

    def screens():
        path.generate_screens(only=own_layers)
        if own_layers is not None:
            exchange_layer_screens(path._gen_screens)

    writer_events = []
    n_blocks = args.blocks if args.blocks is not None else path.default_blocks()

    def step(ev=None):
        """One pass: screens, then the TOD synthesis -- detector blocks pipelined on two streams
        (DevicePath.run: the sampler of block b+1 beside the writer of block b)."""
        if ev: ev[0].record()
        if not args.no_screens_in_step:
            screens()
        if ev: ev[1].record()
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
    for _ in range(args.warmup):
        step()
    barrier()

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    t_start = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    barrier()
    elapsed = time.perf_counter() - t_start
    flags = path.check_flags()

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # outside the timed region: the stages back to back on one stream, for the breakdown
    sev = []
    for k in range(3):
        serial_step(sev)
    torch.cuda.synchronize()
    # per stage: the sum over a step's block launches, averaged over the 3 passes
    serial_ms = np.array([[t[i].elapsed_time(t[i + 1]) for i in range(2)] for t in sev]).sum(axis=0) / 3.0
    serial_up_launch_ms = float(np.mean([t[1].elapsed_time(t[2]) for t in sev]))
    step_ms = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(2)] for k in range(args.steps)]).mean(axis=0)
    # the dominant kernel, timed live in the timed region on the stream it runs on: one launch
    # per detector block when the step is pipelined (its rows x T samples each)
    n_launch = max(1, n_blocks if n_blocks > 1 else 1)
    if writer_events:
        up_ms = float(np.mean([a.elapsed_time(b) for a, b in writer_events]))
        rows_per_launch = D / n_launch
    else:
        up_ms, rows_per_launch = serial_up_launch_ms, D / n_launch
    up_bytes = 4.0 * rows_per_launch * T + 4.0 * rows_per_launch * Ta + 8.0 * T  # TOD write + coarse loading read + sample times read
    achieved = up_bytes / (up_ms * 1e-3) / 1e9
    alone = up_bytes / (serial_up_launch_ms * 1e-3) / 1e9
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
            f"{len(problem['tables'])} band(s), in total over {world} GPU(s)",
            "n_det_total": n_step,
            "n_det_per_gpu": D,
            "n_samples": T,
            "screens_in_step": not args.no_screens_in_step,
            "screens": "sharded by layer + broadcast" if own_layers is not None else "regenerated on every rank from the Philox key",
            "parallelism": f"detector-sharded x{world}, no data-path collective",
        },
        "stage_ms": {
            "screens": float(step_ms[0]),
            "tod_synthesis_pipelined": float(step_ms[1]),
            "serial_breakdown": {"sample": sm_ms, "upsample_with_spline_solve": float(serial_ms[1]),
                                 "note": "the same block launches back to back on one stream, outside the timed region; sums over the blocks"},
            "detector_blocks": n_launch,
        },
        "path_hbm_gbps": path.algorithmic_bytes() / (1e-3 * float(step_ms.sum())) / 1e9,
        "roofline": {
            "kernel": "spline_upsample_fused_kernel",
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": None,
            "bytes_per_launch": up_bytes,
            "ms_per_launch": up_ms,
            "launches_per_step": n_launch,
            "note": "timed in the timed region, where each launch shares the chip with the next block's sampler; "
                    "the same launches back to back on one stream (serial breakdown) reach frac_alone",
            "frac_alone": alone / HBM_PEAK_GBPS,
        },
        "second_kernel": {
            "kernel": "atm_sample_kernel",
            "bound": "valu + vector-memory issue (not hbm): see DESIGN 3.2 and profiles/r02_kernel_pmc.txt",
            "ms_per_step": sm_ms,
            "launches_per_step": n_launch,
            "note": "sum over the step's block launches run back to back on one stream (serial breakdown)",
            "bytes_per_step": sm_bytes,
            "achieved_GBps": sm_bytes / (sm_ms * 1e-3) / 1e9,
            "layer_samples_per_s": D * Ta * L / (sm_ms * 1e-3),
        },
        "flags": int(flags),
    }

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
        threads = max([p["num_threads"] for p in threadpoolctl.threadpool_info()] or [1])
        ref2, cpu_sn = cpu_baseline(problem, scr, slice(D - n_sub, D))  # library defaults: all cores
        err2 = float(np.abs(tod[D - n_sub :].cpu().numpy().astype(np.float64) - ref2).max() / np.abs(ref2).max())
        result["cpu_baseline"] = {
            "value": n_sub * T / cpu_s1,
            "unit": "detector-samples/s",
            "cores": 1,
            "kind": "port",
            "sample": f"first {n_sub} of {D} detector rows, full {T} samples, screens given (sampling + emission + cubic "
            f"upsample; numpy/scipy with BLAS/OpenMP pools limited to 1 thread): {cpu_s1:.2f} s",
            "all_cores": {
                "value": n_sub * T / cpu_sn, "threads": threads, "seconds": cpu_sn,
                "sample": f"last {n_sub} rows, thread pools at the library default ({threads}); numpy fancy indexing and "
                "scipy interp1d do not use them, so both timings agree",
            },
            "host_cpus": os.cpu_count(),
            "cpu_model": _cpu_model(),
            "parity_max_rel_err_vs_gpu": max(err, err2),
        }

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
                    result["allgather"] = {"error": "no completion within 180 s"}
                    print(json.dumps(result), flush=True)
                os._exit(3)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if gatherer is not None:
                times = []
                for _ in range(max(1, args.gather_reps)):
                    barrier()
                    t0 = time.perf_counter()
                    gathered = gatherer.gather(full)
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
                tmax = torch.tensor([dt], dtype=torch.float64, device=red_device)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
                result["allgather"] = {
                    "ms": 1e3 * dt, "received_GB_per_rank": nbytes / 1e9, "GBps_per_rank": nbytes / dt / 1e9,
                    "in_timed_region": False, "in_place": True, "reps": len(times),
                    "transport": "RCCL ncclAllGather via libmrx mrx_allgather_tod",
                    "rows_of_next_rank_bit_identical": same,
                    "step_plus_gather_ms": 1e3 * elapsed / args.steps + 1e3 * dt,
                    "value_with_gather": n_step * T / (elapsed / args.steps + dt),
                }
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


if __name__ == "__main__":
    main()
