set -e
R=$GRAFT_REPO_ROOT; cd $R
run() { tag=$1; shift; env "$@" timeout -k 10 120 python scripts/asmtime.py 256 "$tag" 2>&1 | grep ASMTIME >> gpurun_out/r3_lagablate.txt; }
: > gpurun_out/r3_lagablate.txt
for t in noflux noy nostore noflux_nostore; do run $t NALU_GPU_LIB=$R/nalu_amd/variants/libnalu_gpu_$t.so; done
cat gpurun_out/r3_lagablate.txt
