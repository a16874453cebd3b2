#!/usr/bin/env python3
"""Print the headline figures of bench.py JSON lines: python scripts/show_bench.py <file>..."""
import json, sys
for f in sys.argv[1:]:
    r = json.load(open(f))
    st = r["stage_ms"]
    extra = ""
    if "cpu_baseline" in r:
        c = r["cpu_baseline"]
        extra = f' parity {c.get("parity_max_rel_err_vs_gpu"):.2e} fluct {c.get("parity_fluct_rel_err"):.2e}'
    if "frontend" in r and "atmosphere" in r["frontend"]:
        extra += f' frontend {r["frontend"]["atmosphere"]["run_ms"]:.2f} / {r["frontend"]["atmosphere_noise"]["run_ms"]:.2f} ms'
    print(f'{f}: {r["ms_per_step"]:.3f} ms/step (gpu {r["gpu_ms_per_step"]:.3f}), screens {st["screens"]:.3f} + TOD {st["tod_synthesis_pipelined"]:.3f} '
          f'({st["detector_blocks"]} blocks), writer {r["roofline"]["ms_per_launch"]:.4f} ms frac {r["roofline"]["frac"]:.3f} (alone {r["roofline"]["frac_alone"]:.3f}), '
          f'path {r["path_hbm_gbps"] / 8000:.3f} of 8 TB/s, sampler serial {st["serial_breakdown"]["sample"]:.3f} ms{extra}')
