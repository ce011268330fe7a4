#!/bin/bash
# usage: scripts/pmc_insts.sh <tag> <program> [args]: instruction-mix counters, one pass
R=$GRAFT_REPO_ROOT; tag=$1; shift
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmci_$tag -o p1 -- "$@" > $R/gpurun_out/pmci_$tag.log 2>&1 || { echo "pass failed"; tail -n 3 $R/gpurun_out/pmci_$tag.log; exit 1; }
