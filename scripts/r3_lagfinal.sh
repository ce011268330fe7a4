set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_bench_path.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_lag_tests.log 2>&1 || { tail -30 gpurun_out/r3_lag_tests.log; exit 1; }
tail -2 gpurun_out/r3_lag_tests.log
timeout -k 10 120 python scripts/asmtime.py 256 grouped 2>&1 | grep ASMTIME
NG_ASM_LAG_GROUPS=0 timeout -k 10 120 python scripts/asmtime.py 256 ungrouped 2>&1 | grep ASMTIME
timeout -k 10 200 python scripts/asmtime.py 512 grouped512 2>&1 | grep ASMTIME
NG_ASM_LAG_GROUPS=0 timeout -k 10 200 python scripts/asmtime.py 512 ungrouped512 2>&1 | grep ASMTIME
