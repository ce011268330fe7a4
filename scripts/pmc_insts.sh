#!/bin/bash
# usage: scripts/pmc_insts.sh <tag> <script.py> [args]: instruction-mix counters, one pass.
# <script.py> is a Python file relative to the repository root and runs as `python3 <root>/<script.py>`:
# with --pmc the profiler's preloaded library initialises the GPU BEFORE the program starts, so the
# program after `--` must be the interpreter itself -- never a wrapper, a `#!/usr/bin/env` script, `env`,
# `bash -c` or anything else that re-execs (an exec after GPU initialisation takes the box down).
R=$GRAFT_REPO_ROOT; tag=$1; shift
case "$1" in
  *.py) [ -f "$R/$1" ] || { echo "pmc_insts.sh: $R/$1 not found"; exit 2; } ;;
  *) echo "pmc_insts.sh: the program must be a .py file of this repository (it is run by python3 directly)"; exit 2 ;;
esac
export TMPDIR=/tmp; cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmci_$tag -o p1 -- python3 $R/"$@" > $R/gpurun_out/pmci_$tag.log 2>&1 || { echo "pass failed"; tail -n 3 $R/gpurun_out/pmci_$tag.log; exit 1; }
