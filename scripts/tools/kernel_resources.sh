#!/bin/bash
# Registers, spills, occupancy and LDS of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage),
# one line per kernel.   scripts/tools/kernel_resources.sh mrx_map [filter-regex] [extra hipcc flags...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TU=$1; FILTER=${2:-.}; shift; shift
EXTRA=$(grep "^EXTRA_$TU " $ROOT/maria_amd/csrc/Makefile | sed 's/^[^=]*= *//')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -fno-slp-vectorize $EXTRA "$@" \
  --cuda-device-only -c -o /dev/null $ROOT/maria_amd/csrc/$TU.hip -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
cur = {}
def flush():
    if cur: print("{name:100s} vgpr {v:>3s} sgpr {s:>3s} spill v{vs} s{ss} occ {o} lds {l}".format(**cur))
for line in sys.stdin:
    m = re.search(r"remark: +(.*?): (\S+) \[-Rpass", line)
    m0 = re.search(r"Function Name: (\S+)", line)
    if m0:
        flush(); cur.clear()
        n = subprocess.run(["c++filt", m0.group(1)], capture_output=True, text=True).stdout.strip()
        cur["name"] = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:100]
        continue
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    key = {"VGPRs": "v", "TotalSGPRs": "s", "VGPRs Spill": "vs", "SGPRs Spill": "ss", "Occupancy [waves/SIMD]": "o", "LDS Size [bytes/block]": "l"}.get(k)
    if key: cur[key] = v
flush()
' | grep -E "$FILTER"
