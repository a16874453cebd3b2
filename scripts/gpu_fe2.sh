#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for MODE in atm noise; do
  echo "== $MODE"
  timeout -k 10 300 python3 scripts/frontend_trace.py $MODE 2>&1 | grep "^run"
done
