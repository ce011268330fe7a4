#!/bin/bash
# usage: scripts/pmc.sh <tag> <script> [args]; one rocprofv3 --pmc pass per counter group
R=$GRAFT_REPO_ROOT; tag=$1; shift
export TMPDIR=/tmp; cd /tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_$tag -o p$i -- python3 $R/"$@" > $R/gpurun_out/pmc_$tag.p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_$tag.p$i.log; exit 1; }
done
ls $R/gpurun_out/pmc_$tag
