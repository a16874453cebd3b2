"""Timeline of the one-launch synthesis (atm_tod_kernel) from a -DMRX_SYNTH_TRACE build: what every workgroup did and when.

    make -C maria_amd/csrc OUT=$PWD/scripts/ab_trace/libmrx_trace.so OBJDIR=$PWD/build/trace_obj EXTRA_mrx_synth=-DMRX_SYNTH_TRACE
    MRX_LIB_PATH=scripts/ab_trace/libmrx_trace.so python3 scripts/exp/synth_timeline.py [config] [bin_us]

Prints, per time bin of the launch: workgroups in a writer tile, in a sampler item, waiting; tiles finished (=> bytes
stored) in the bin; then the distributions of tile / item durations and of the gaps between a workgroup's events.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from maria_amd import synthetic  # noqa: E402
from maria_amd import _lib  # noqa: E402
from maria_amd.pipeline import DevicePath  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "atlast_10k"
bin_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
n_det = synthetic.CONFIGS[config]["n_det"]
if config == "atlast_50k":
    n_det //= 8
if os.environ.get("MRX_TL_ROWS"):  # a shard: the first rows of the configuration
    n_det = int(os.environ["MRX_TL_ROWS"])
problem = synthetic.config_problem(config, n_det=n_det)
path = DevicePath(problem, device="cuda:0")
tod = torch.empty((path.D, path.T), dtype=torch.float32, device="cuda:0")
if os.environ.get("MRX_TL_CHUNK"):
    path.ctx.set_option(_lib.OPT_SAMPLE_CHUNK, int(os.environ["MRX_TL_CHUNK"]))
if os.environ.get("MRX_TL_SAMPLERS"):  # dedicated sampler workgroups per CU
    path.ctx.set_option(_lib.OPT_SAMPLE_WGS_PER_CU, int(os.environ["MRX_TL_SAMPLERS"]))
path.generate_screens()
for _ in range(3):
    path.run(tod)
torch.cuda.synchronize()
lib = _lib.load()
W, E = 2048, 384
ev = np.zeros((W, E, 4), dtype=np.uint32)
cnt = np.zeros(W, dtype=np.int32)
rc = lib.mrx_debug_synth_trace(ev.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p), 1)
assert rc == 0
path.run(tod)
torch.cuda.synchronize()
rc = lib.mrx_debug_synth_trace(ev.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p), 1)
assert rc == 0
rows = []
for w in range(W):
    for k in range(min(cnt[w], E)):
        rows.append((w, *ev[w, k]))
a = np.array(rows, dtype=np.int64)
turns = a[a[:, 1] == 4]
parts = a[a[:, 1] == 5]
a = a[(a[:, 1] != 4) & (a[:, 1] != 5)]
if len(parts):  # inside a sampler item (thread 0's wave): the steps' loop and the drain of its stores, in 10-ns ticks
    for name, col in (("steps' loop (behind the prologue)", 2), ("drain of the write-through stores", 3)):
        d = parts[:, col] / 100.0
        print(f"item {name}: n={len(d)} median {np.median(d):.2f} us p90 {np.percentile(d, 90):.2f} mean {d.mean():.2f}")
if len(turns):  # what the first wave spent before a tile: (ticket, decode + the tile's knots, poll) in 10-ns ticks
    for name, col in (("ticket (queue atomic)", 2), ("decode + t[] of the tile", 3), ("poll", 4)):
        d = turns[:, col] / 100.0
        print(f"turn {name}: n={len(d)} median {np.median(d):.2f} us p90 {np.percentile(d, 90):.2f} mean {d.mean():.2f}")
print(f"{config}: D={path.D} T={path.T} Ta={path.Ta}; workgroups with events {int((cnt > 0).sum())}, events {len(a)}, truncated {int((cnt >= E).sum())}")
t_origin = a[:, 3].min()
t0 = (a[:, 3] - t_origin) / 100.0  # us (100 MHz)
t1 = (a[:, 4] - t_origin) / 100.0
kind = a[:, 1]
end = t1.max()
print(f"launch span (first event start -> last event end): {end:.1f} us")
tile_bytes = 32 * 1024 * 4
nb = int(end / bin_us) + 1
print(f"{'bin_us':>8} {'writing':>8} {'sampling':>9} {'waiting':>8} {'tiles_done':>10} {'GB/s':>8}")
for b in range(nb):
    lo, hi = b * bin_us, (b + 1) * bin_us
    def occ(k):
        m = kind == k
        return float(np.clip(np.minimum(t1[m], hi) - np.maximum(t0[m], lo), 0, None).sum() / bin_us)
    done = int(((kind == 1) & (t1 >= lo) & (t1 < hi)).sum())
    print(f"{lo:8.0f} {occ(1):8.1f} {occ(2):9.1f} {occ(3):8.1f} {done:10d} {done * tile_bytes / (bin_us * 1e-6) / 1e9:8.0f}")
for k, name in ((1, "tile"), (2, "item"), (3, "wait")):
    m = kind == k
    if m.any():
        d = t1[m] - t0[m]
        print(f"{name}: n={int(m.sum())} duration us: median {np.median(d):.1f} p10 {np.percentile(d, 10):.1f} p90 {np.percentile(d, 90):.1f} sum/wg {d.sum() / (cnt > 0).sum():.0f}")
# gaps between a workgroup's consecutive events (queue, poll, decode, barriers)
gaps, gap_after_tile, gap_after_item = [], [], []
for w in np.unique(a[:, 0]):
    m = a[:, 0] == w
    s0, s1, kk = t0[m], t1[m], kind[m]
    for i in range(1, len(s0)):
        g = s0[i] - s1[i - 1]
        (gap_after_tile if kk[i - 1] == 1 else gap_after_item).append(g)
for name, g in (("after a tile", gap_after_tile), ("after an item / wait", gap_after_item)):
    g = np.array(g)
    if len(g):
        print(f"gap {name}: n={len(g)} median {np.median(g):.2f} us p90 {np.percentile(g, 90):.2f} sum/wg {g.sum() / (cnt > 0).sum():.0f}")
first_tile_end = t1[kind == 1].min()
print(f"first tile finished at {first_tile_end:.1f} us; last item finished at {t1[kind == 2].max():.1f} us")
per_wg_first = [t0[(a[:, 0] == w)].min() for w in np.unique(a[:, 0])]
print(f"workgroups' first event starts: min {min(per_wg_first):.1f} max {max(per_wg_first):.1f} us")
