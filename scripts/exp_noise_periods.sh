#!/bin/bash
# noise throughput by period: bash scripts/exp_noise_periods.sh <lib>...
for v in "$@"; do for T in 400000 800000 1440000 3000000; do
  D=$((2048 * 1440000 / T)); MRX_LIB_PATH=$PWD/$v MRX_NOISE_BATCH=512 python3 scripts/noise_bench.py $D $T 2 2>&1 | grep "modes=5" | sed "s|^|$v |" | cut -c1-150
done; done
