#!/bin/bash
# Register / LDS / scratch use of the kernels of one TU's object (build/csrc/<tu>.o) whose names match $2
#   scripts/kernel_regs.sh mrx_sample 'atm_sample_px'
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=${3:-$ROOT/build/csrc}/$1.o
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $OBJ $TMP/fatbin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$TMP/co.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/co.o | awk -v pat="$2" '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.agpr_count:/ {a=$2}
  /\.group_segment_fixed_size:/ {l=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.vgpr_spill_count:/ {sp=$2}
  /\.wavefront_size:/ {if (name ~ pat) printf "%-100s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s spill %s\n", substr(name,1,100), v, a, s, l, p, sp}'
rm -rf $TMP
