# round 3: timing of the ablation builds of the prefetching lagged kernel (scripts/r3_ablate_build.sh); results of these builds are wrong by construction
set -e
R=$GRAFT_REPO_ROOT; cd $R
export NG_ASM_LAG_PF=1 NG_ASM_LAG_GROUPS=1 NG_ASM_LAG_GY=${GY:-4}
timeout -k 10 120 python scripts/asmtime.py 256 base_gy$NG_ASM_LAG_GY 2>&1 | grep ASMTIME
for tag in "$@"; do
  NALU_GPU_LIB=$R/scripts/ablate/libnalu_gpu_$tag.so timeout -k 10 120 python scripts/asmtime.py 256 $tag 2>&1 | grep ASMTIME
done
