set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_bench_path.py -x -q --timeout 300 --timeout-method thread -k "slot_major and default" 2>&1 | tail -1
NG_ASM_LAG_TAIL=2 timeout -k 10 300 python -u -m pytest tests/test_gpu_bench_path.py -x -q --timeout 300 --timeout-method thread -k "slot_major and default" 2>&1 | tail -1
for i in 1 2; do
timeout -k 10 120 python scripts/asmtime.py 256 default 2>&1 | grep ASMTIME
NG_ASM_LAG_TAIL=2 timeout -k 10 120 python scripts/asmtime.py 256 tailfirst 2>&1 | grep ASMTIME
NALU_GPU_LIB=$R/scripts/ablate/libnalu_gpu_latebar.so timeout -k 10 120 python scripts/asmtime.py 256 latebar 2>&1 | grep ASMTIME
NG_ASM_LAG_TAIL=2 NALU_GPU_LIB=$R/scripts/ablate/libnalu_gpu_latebar.so timeout -k 10 120 python scripts/asmtime.py 256 latebar_tailfirst 2>&1 | grep ASMTIME
done
