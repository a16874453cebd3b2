#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes of scripts/profile_round.sh into HBM bytes per launch
per kernel, with the corrections MI355X_MICROARCH.md prescribes for gfx950: both counters are
in KB (x 1024), FETCH_SIZE counts a 128-byte read request as 64 bytes (x 2).
Usage: python scripts/pmc_summary.py gpurun_out/<tag> profiles/<prefix>"""

import collections
import csv
import json
import re
import sys


def per_launch(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            name = re.split(r"[<(]", name)[0].strip()
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, dict(cnt)


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    fetch, n = per_launch(f"{src}/pmc_fetch/run_counter_collection.csv", "FETCH_SIZE")
    write, _ = per_launch(f"{src}/pmc_write/run_counter_collection.csv", "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * 1024.0 * fetch.get(k, 0.0), 1024.0 * write.get(k, 0.0)
        rows.append((k, n.get(k, 0), fetch.get(k, 0.0), write.get(k, 0.0), fb, wb, fb + wb))
    with open(f"{prefix}_pmc_hbm_traffic.csv", "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,fetch_bytes_corrected,write_bytes,hbm_bytes_per_launch\n")
        for r in rows:
            f.write(",".join(str(x) for x in r) + "\n")
    method = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace only (KB units x1024); "
              "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests as 64 B)")
    source = "profiles/" + f"{prefix}_pmc_hbm_traffic.csv".split("/")[-1]

    def dump(row, path, launches):
        with open(path, "w") as f:
            json.dump({"config": "atlast_10k", "kernel": row[0], "hbm_bytes_per_launch": row[6], "fetch_bytes_corrected": row[4],
                       "launches_per_step": launches, "write_bytes": row[5], "method": method, "source": source}, f, indent=1)

    # the step's dominant kernel: the one-launch synthesis where bench.py ran it (one launch per step), else the writer
    # (one launch per detector block); the stand-alone writer of the serial breakdown gets a file of its own beside it
    synth = [r for r in rows if r[0] == "atm_tod_kernel"]
    writer = [r for r in rows if r[0] in ("spline_upsample_fused_kernel", "spline_upsample_kernel")]
    if synth:
        dump(synth[0], f"{prefix}_traffic.json", 1)
        if writer:
            dump(writer[0], f"{prefix}_traffic_writer.json", 1)
    else:
        dump(writer[0], f"{prefix}_traffic.json", int(sys.argv[3]) if len(sys.argv) > 3 else 4)
    for r in rows:
        print(f"{r[0]:32s} launches {r[1]:4d}  read {r[4]/1e6:10.2f} MB  written {r[5]/1e6:10.2f} MB")
    # the SQ / TCP / LDS counter passes: per-launch averages of every kernel of the step
    lines = []
    for sub in ("pmc_sq", "pmc_tcp", "pmc_lds"):
        path = f"{src}/{sub}/run_counter_collection.csv"
        try:
            tot, cnt = collections.defaultdict(collections.Counter), collections.Counter()
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                    name = re.split(r"[(]", name)[0].strip()
                    tot[name][row["Counter_Name"]] += float(row["Counter_Value"])
                    cnt[(name, row["Counter_Name"])] += 1
        except OSError:
            continue
        lines.append(f"# pass {sub}: rocprofv3 --kernel-trace --pmc <counters below> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-frontend")
        for k in sorted(tot):
            if k.startswith("__amd") or k.startswith("at::") or k.startswith("plan_"):
                continue
            lines.append(f"{k:44s} launches {max(cnt[(k, c)] for c in tot[k]):3d}  " + " ".join(f"{c}={v / cnt[(k, c)]:.4g}" for c, v in sorted(tot[k].items())))
    if lines:
        with open(f"{prefix}_kernel_pmc.txt", "w") as f:
            f.write("# per-launch averages; SQ_* cycle counters are in quad-cycles summed over waves (MI355X_MICROARCH.md)\n")
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
