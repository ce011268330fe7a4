#!/bin/bash
set -e
cd "$(dirname "$0")/.."
for n in 64 128 256; do
  for ra in 1 0 1 0; do
    echo "n $n NG_SGS_RECOVER_ALL=$ra"
    NG_SGS_RECOVER_ALL=$ra timeout -k 10 200 python3 scripts/sgsflow_bench.py $n flow:6,6 | grep -o '^flow:[0-9,]* {"cg_ms_per_it": [0-9.]*, "sgs_ms_per_apply": [0-9.]*'
  done
done
