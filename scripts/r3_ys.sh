#!/bin/bash
set -e
cd "$(dirname "$0")/.."
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_bench_path.py -m gpu -x -q --timeout 300 --timeout-method thread -k "slot_major" > gpurun_out/ys.log 2>&1
for v in 0 1 0 1; do
  NG_ASM_LAG_YSHARE=$v timeout -k 10 120 python3 scripts/asmtime.py 256 "YSHARE=$v" >> gpurun_out/ys.log 2>&1
done
