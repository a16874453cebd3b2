"""bench.py's self-launch (CPU tier): `python bench.py --gpus N` from a plain shell must start N ranks as a
CHILD job before touching a GPU, pass its arguments on and return the child's status."""

from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TORCHELASTIC_RUN_ID")}
    env["MRX_TEST"] = "1"
    return env


def test_launch_command_carries_the_arguments():
    argv = ["--gpus", "4", "--steps", "7", "--warmup", "2", "--backend", "gloo", "--single-device"]
    cmd = bench.launch_command(4, argv, 29123)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1 :] == argv


def test_needs_launch_only_from_a_plain_shell():
    args = bench.parse_args(["--gpus", "8"])
    assert bench.needs_launch(args, {})
    assert bench.needs_launch(args, {"WORLD_SIZE": "1"})
    assert not bench.needs_launch(args, {"WORLD_SIZE": "8", "RANK": "3"})  # already a rank of torch.distributed.run
    assert not bench.needs_launch(args, {"WORLD_SIZE": "1", "TORCHELASTIC_RUN_ID": "x"})
    assert not bench.needs_launch(bench.parse_args([]), {})  # N = 1 runs in this process


def test_print_launch_from_the_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--print-launch"],
                         env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert "--nproc-per-node=2" in cmd and cmd[-4:] == ["--gpus", "2", "--steps", "3"]


@pytest.mark.timeout(600)
def test_the_parent_returns_the_childs_status():
    """Without a GPU every rank stops at bench.py's own "needs a GPU" assertion: the launcher must have started
    them (two ranks, gloo) and must hand their failure on as a non-zero status, not swallow it."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run the benchmark")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--single-device",
                          "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         env=_clean_env(), capture_output=True, text=True, timeout=560)
    assert out.returncode != 0
    assert "bench.py needs a GPU" in out.stderr


def test_recorded_bench_lines_carry_the_contract_fields():
    """The lines bench.py printed on the GPU box (profiles/: N = 1 under the defaults, and the two-rank rehearsal) hold
    every field of the driver's contract, the roofline and CPU-baseline objects, and consistent arithmetic."""
    for name in ("r03_bench.json", "r03_bench_gloo2.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            line = json.loads(f.read().strip().splitlines()[-1])
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                    "dtype", "data", "config", "roofline"):
            assert key in line, (name, key)
        assert line["unit"] == "detector-samples/s" and line["higher_is_better"] is True and line["vs_baseline"] is None
        assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
        assert line["scaling"] in ("weak", "strong")
        # value = whole-job samples per second: all detectors x samples x steps over the timed region
        per_step = line["config"]["n_det_total"] * line["config"]["n_samples"]
        assert abs(line["value"] / (per_step / (line["ms_per_step"] * 1e-3)) - 1) < 1e-6
        roof = line["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in roof, (name, key)
        assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0.3 < roof["frac"] < 1.0
        if line["n_gpus"] == 1:
            cpu = line["cpu_baseline"]
            for key in ("value", "unit", "cores", "kind", "sample"):
                assert key in cpu, key
            assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["unit"] == line["unit"]
            assert cpu["parity_max_rel_err_vs_gpu"] <= 1e-5 and cpu["parity_fluct_rel_err"] <= 5e-4
            # the kernel's measured HBM traffic stays near its algorithmic bytes (re-reads would show here first)
            assert 1.0 <= roof["traffic"] / roof["bytes_per_launch"] < 1.15
            # the wall-clock figure against the same steps on the GPU's own clock: no host stall inside the timed region
            assert 0.97 < line["gpu_ms_per_step"] / line["ms_per_step"] <= 1.0


def test_the_gather_that_value_takes_is_the_fastest_verified_one():
    """bench.py at N > 1 times both all-gathers (default --gather-algo both) and puts the FASTER one whose probe found the next
    rank's rows bit-identical into `value` (select_gather); a variant that errors or fails its probe is named and the other
    taken; with none left `value` stays the synthesis alone -- the line is never lost.  Entries shaped as
    the gather section records them."""
    ok = lambda ms: {"ms": ms, "rows_of_next_rank_bit_identical": True}  # noqa: E731
    assert bench.parse_args([]).gather_algo == "both"
    # both verified: the faster one, whichever ran first
    algo, why = bench.select_gather({"allgather": ok(28.0), "p2p": ok(8.1)})
    assert algo == "p2p" and "28.00 ms" in why
    algo, _ = bench.select_gather({"allgather": ok(7.9), "p2p": ok(8.1)})
    assert algo == "allgather"
    # the faster one failed its probe: the slower verified one
    algo, why = bench.select_gather({"allgather": ok(28.0), "p2p": {"ms": 8.0, "rows_of_next_rank_bit_identical": False}})
    assert algo == "allgather" and "did not verify" in why
    # one raised: the other; and said so
    algo, why = bench.select_gather({"allgather": {"error": "RuntimeError: ncclAllGather failed"}, "p2p": ok(9.0)})
    assert algo == "p2p" and "failed" in why
    # nothing usable: no gather in `value`, with the reasons
    algo, why = bench.select_gather({"allgather": {"error": "timeout"}, "p2p": {"ms": 8.0, "rows_of_next_rank_bit_identical": False}})
    assert algo is None and "synthesis alone" in why and "timeout" in why
    assert bench.select_gather({}) == (None, "no verified variant (no variant ran): value stays the synthesis alone")
    # a single variant (--gather-algo p2p)
    assert bench.select_gather({"p2p": ok(8.0)}) == ("p2p", "the only variant run")
