#!/bin/bash
# mt_sgs after the colour-major copy: its parity tests, then time per iteration at 256^3
set -e
cd "$(dirname "$0")/.."
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_distributed.py tests/test_gpu_bench_path.py -m gpu -x -q --timeout 300 --timeout-method thread -k "solver or mt_sgs or precond or krylov or 27 or element or dist" > gpurun_out/mc.log 2>&1
timeout -k 10 300 python3 scripts/sgsbench.py 256 >> gpurun_out/mc.log 2>&1
