# round-3 evidence runs (one phase per gpurun call):
#   bash scripts/r3_profile.sh pmc    -> the traffic counters (two --pmc passes, each a run of its own) of the hot
#                                        kernels at 256^3 and 512^3 -> gpurun_out/r3b_traffic.json (copy to profiles/)
#   bash scripts/r3_profile.sh bench  -> the default bench line, the same bench under rocprofv3 --kernel-trace --stats,
#                                        and the scalar / picard / 512^3 / --variants lines (they read profiles/r3b_traffic.json)
# The program after `--` is python3 itself (with --pmc the profiler initialises the GPU before the program starts:
# no wrapper, no exec hop).
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
case "$1" in
pmc)
  cd /tmp
  for n in 256 512; do
    i=0
    for grp in "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
      i=$((i+1))
      timeout -s KILL 400 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_r3_$n -o p$i -- python3 $R/scripts/pmc_r3.py $n > $R/gpurun_out/pmc_r3_$n.p$i.log 2>&1 || { echo "pass $i at $n failed"; tail -3 $R/gpurun_out/pmc_r3_$n.p$i.log; exit 1; }
      echo "pmc pass $i at $n done"
    done
  done
  cd $R
  python scripts/r3_traffic.py 256=gpurun_out/pmc_r3_256 512=gpurun_out/pmc_r3_512 > gpurun_out/r3b_traffic.json
  rm -rf gpurun_out/pmc_r3_256 gpurun_out/pmc_r3_512  # (per-dispatch CSVs: tens of MiB; the summary is what is kept)
  tail -c 400 gpurun_out/r3b_traffic.json
  ;;
bench)
  cd $R
  timeout -k 10 500 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err
  echo "default line done"
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3b_stats -o r3b -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-n 0 --no-ttt > $R/gpurun_out/r3b_bench_under_rocprof.json 2> $R/gpurun_out/r3b_rocprof.err
  echo "kernel trace done"
  cd $R
  timeout -k 10 300 python bench.py --workload scalar > gpurun_out/r3b_bench_scalar.json 2> gpurun_out/r3b_bench_scalar.err
  echo "scalar line done"
  timeout -k 10 300 python bench.py --workload scalar --precond sgs --cpu-n 0 > gpurun_out/r3b_bench_scalar_sgs.json 2> gpurun_out/r3b_bench_scalar_sgs.err
  timeout -k 10 300 python bench.py --workload scalar --precond mt_sgs --cpu-n 0 > gpurun_out/r3b_bench_scalar_mt_sgs.json 2> gpurun_out/r3b_bench_scalar_mt_sgs.err
  echo "scalar sgs / mt_sgs lines done"
  timeout -k 10 200 python bench.py --workload picard --cpu-n 0 > gpurun_out/r3b_bench_picard.json 2> gpurun_out/r3b_bench_picard.err
  echo "picard line done"
  timeout -k 10 300 python bench.py --n 512 --cpu-n 0 --no-ttt > gpurun_out/r3b_bench_512.json 2> gpurun_out/r3b_bench_512.err
  echo "512 line done"
  timeout -k 10 300 python bench.py --n 512 --workload scalar --cpu-n 0 > gpurun_out/r3b_bench_512_scalar.json 2> gpurun_out/r3b_bench_512_scalar.err
  echo "512 scalar line done"
  timeout -k 10 300 python bench.py --variants --cpu-n 0 --no-ttt > gpurun_out/r3b_bench_variants.json 2> gpurun_out/r3b_bench_variants.err
  tail -c 600 gpurun_out/r3b_bench.json
  ;;
*) echo "usage: r3_profile.sh pmc|bench"; exit 2;;
esac
