#!/bin/bash
for n in "$@"; do python3 scripts/bin_bench.py $n 1 2 2>&1 | grep "^bin"; done
