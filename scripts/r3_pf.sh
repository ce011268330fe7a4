# round 3: the lagged kernel with / without the one-step-ahead LDS prefetch (NG_ASM_LAG_PF), alone and with groups
set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 400 python -u -m pytest tests/test_gpu_bench_path.py -x -q --timeout 300 --timeout-method thread -k "lag_pf or default" > gpurun_out/r3_pf_tests.log 2>&1 || { tail -30 gpurun_out/r3_pf_tests.log; exit 1; }
tail -2 gpurun_out/r3_pf_tests.log
timeout -k 10 120 python scripts/asmtime.py 256 base 2>&1 | grep ASMTIME
NG_ASM_LAG_PF=1 timeout -k 10 120 python scripts/asmtime.py 256 pf 2>&1 | grep ASMTIME
NG_ASM_LAG_PF=1 NG_ASM_LAG_GROUPS=1 timeout -k 10 120 python scripts/asmtime.py 256 pf_groups 2>&1 | grep ASMTIME
timeout -k 10 200 python scripts/asmtime.py 512 base512 2>&1 | grep ASMTIME
NG_ASM_LAG_PF=1 NG_ASM_LAG_GROUPS=1 timeout -k 10 200 python scripts/asmtime.py 512 pf_groups512 2>&1 | grep ASMTIME
