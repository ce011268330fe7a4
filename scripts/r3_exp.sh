set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 120 python scripts/asmtime.py 256 default 2>&1 | grep ASMTIME
NALU_GPU_LIB=$R/scripts/ablate/libnalu_gpu_latebar.so timeout -k 10 120 python scripts/asmtime.py 256 latebar 2>&1 | grep ASMTIME
NG_MARCH_LEN=32 timeout -k 10 120 python scripts/asmtime.py 256 len32 2>&1 | grep ASMTIME
NG_MARCH_LEN=64 timeout -k 10 120 python scripts/asmtime.py 256 len64 2>&1 | grep ASMTIME
NG_MARCH_LEN=128 timeout -k 10 120 python scripts/asmtime.py 256 len128 2>&1 | grep ASMTIME
NG_MARCH_LEN=32 NG_MARCH_XCD=4 timeout -k 10 120 python scripts/asmtime.py 256 len32_xcd4 2>&1 | grep ASMTIME
NG_MARCH_LEN=32 timeout -k 10 200 python scripts/asmtime.py 512 len32_512 2>&1 | grep ASMTIME
NG_MARCH_LEN=64 timeout -k 10 200 python scripts/asmtime.py 512 len64_512 2>&1 | grep ASMTIME
