# the whole GPU suite, then the default bench line
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/pytest_full.log 2>&1 || { tail -30 gpurun_out/pytest_full.log; exit 1; }
tail -3 gpurun_out/pytest_full.log
timeout -k 10 400 python bench.py > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err
tail -c 1500 gpurun_out/bench_now.json
