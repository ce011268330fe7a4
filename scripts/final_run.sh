set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 200 --timeout-method thread > gpurun_out/pytest_final2.log 2>&1 || { tail -30 gpurun_out/pytest_final2.log; exit 1; }
tail -2 gpurun_out/pytest_final2.log
timeout -k 10 300 python bench.py > gpurun_out/bench_final2.json 2> gpurun_out/bench_final2.err
export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final2 -o r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-n 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof2.json 2> $GRAFT_REPO_ROOT/gpurun_out/rocprof2.err
ls $GRAFT_REPO_ROOT/gpurun_out/final2 | head
