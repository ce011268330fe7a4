#!/bin/bash
# usage: scripts/pmc_fetch.sh <tag> <script> [args]; FETCH_SIZE/TCC hit-miss only (2 passes)
R=$GRAFT_REPO_ROOT; tag=$1; shift
export TMPDIR=/tmp; cd /tmp
i=0
for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -o p$i -- python3 $R/"$@" > $R/gpurun_out/pmc_$tag.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_$tag.p$i.log; exit 1; }
done
