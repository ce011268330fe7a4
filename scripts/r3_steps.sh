#!/bin/bash
# step timelines of the dataflow SGS sweep: a 2 x 2 tile lattice and the 256^3 lattice
set -e
# (needs a library built with -DNG_FLOW_STEP_TRACE: make -C nalu_amd/csrc CXXFLAGS+=-DNG_FLOW_STEP_TRACE precond.o all)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export NG_SGS_TRACE_STEPS=gpurun_out/steps_small
timeout -k 10 120 python3 scripts/sgsflow_probe.py 256 12 12 4 > gpurun_out/steps_small.log 2>&1
python3 scripts/flowsteps.py gpurun_out/steps_small.fwd 2 2 5 1,1 > gpurun_out/steps_small.txt 2>&1
export NG_SGS_TRACE_STEPS=gpurun_out/steps_full
timeout -k 10 200 python3 scripts/sgsflow_probe.py 256 256 256 3 > gpurun_out/steps_full.log 2>&1
python3 scripts/flowsteps.py gpurun_out/steps_full.fwd 43 43 5 20,20 > gpurun_out/steps_full.txt 2>&1
unset NG_SGS_TRACE_STEPS
timeout -k 10 200 python3 scripts/sgsflow_probe.py 256 256 256 10 > gpurun_out/steps_notrace.log 2>&1
tail -1 gpurun_out/steps_notrace.log
