# round-2 evidence run (one gpurun call): default bench line, the same bench under
# rocprofv3 --kernel-trace --stats, the traffic counters of the assembly kernel and SpMV
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_stats -o r2 -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-n 0 --no-ttt > $R/gpurun_out/r2_bench_under_rocprof.json 2> $R/gpurun_out/r2_rocprof.err
cd $R
bash scripts/pmc_short.sh r2asm scripts/pmc_asm.py 256
python scripts/pmc_summary.py gpurun_out/pmc_r2asm > gpurun_out/r2_pmc_summary.txt
python scripts/r2_traffic.py gpurun_out/pmc_r2asm > gpurun_out/r2_traffic.json
# the plain row kernel for comparison
NG_ASM_MARCH=0 bash scripts/pmc_short.sh r2asm_row scripts/pmc_asm.py 256
python scripts/pmc_summary.py gpurun_out/pmc_r2asm_row > gpurun_out/r2_pmc_summary_row_kernel.txt
tail -n 3 gpurun_out/r2_bench.json
