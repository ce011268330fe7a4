# round 3: counters of the lagged kernel variants (three --pmc passes each, each a run of its own)
# usage: bash scripts/r3_pf_pmc.sh <n> <tag> [NG_*=value ...]
R=$GRAFT_REPO_ROOT
n=$1; tag=$2; shift 2
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp; cd /tmp
i=0
for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -o p$i -- python3 $R/scripts/asmtime.py $n $tag > $R/gpurun_out/pmc_$tag.p$i.log 2>&1 || { echo "pass $i $tag failed"; tail -3 $R/gpurun_out/pmc_$tag.p$i.log; exit 1; }
done
cd $R
python3 scripts/pmc_sum.py gpurun_out/pmc_$tag
