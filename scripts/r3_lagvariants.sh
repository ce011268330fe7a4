set -e
R=$GRAFT_REPO_ROOT; cd $R
run() { tag=$1; shift; env "$@" timeout -k 10 120 python scripts/asmtime.py 256 "$tag" 2>&1 | grep ASMTIME >> gpurun_out/r3_lagvariants.txt; }
: > gpurun_out/r3_lagvariants.txt
run default X=1
run lag_off NG_ASM_LAG=0
run dpparea NALU_GPU_LIB=$R/nalu_amd/variants/libnalu_gpu_dpparea.so
run nt NALU_GPU_LIB=$R/nalu_amd/variants/libnalu_gpu_nt.so
run both NALU_GPU_LIB=$R/nalu_amd/variants/libnalu_gpu_both.so
run len8 NG_MARCH_LEN=8
run len32 NG_MARCH_LEN=32
run len64 NG_MARCH_LEN=64
run xcd2 NG_MARCH_XCD=2
run xcd4 NG_MARCH_XCD=4
run xcd16 NG_MARCH_XCD=16
run xcd32 NG_MARCH_XCD=32
cat gpurun_out/r3_lagvariants.txt
