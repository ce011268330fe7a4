# round 3: the default assembly kernel (prefetching lagged kernel, groups of four lines) against its variants
set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_bench_path.py -x -q --timeout 300 --timeout-method thread -k "slot_major" > gpurun_out/r3_gy_tests.log 2>&1 || { tail -30 gpurun_out/r3_gy_tests.log; exit 1; }
tail -2 gpurun_out/r3_gy_tests.log
timeout -k 10 120 python scripts/asmtime.py 256 default 2>&1 | grep ASMTIME
NG_MARCH_LEN=32 timeout -k 10 120 python scripts/asmtime.py 256 len32 2>&1 | grep ASMTIME
NG_ASM_LAG_GY=8 timeout -k 10 120 python scripts/asmtime.py 256 gy8 2>&1 | grep ASMTIME
NG_ASM_LAG_PF=0 NG_ASM_LAG_GROUPS=0 timeout -k 10 120 python scripts/asmtime.py 256 nopf 2>&1 | grep ASMTIME
timeout -k 10 200 python scripts/asmtime.py 512 default512 2>&1 | grep ASMTIME
