set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_bench_path.py tests/test_gpu_parity.py tests/test_gpu_picard.py -x -q --timeout 600 --timeout-method thread > gpurun_out/r3_exp_tests.log 2>&1 || { tail -30 gpurun_out/r3_exp_tests.log; exit 1; }
tail -2 gpurun_out/r3_exp_tests.log
for i in 1 2; do
timeout -k 10 120 python scripts/asmtime.py 256 hoist 2>&1 | grep ASMTIME
done
timeout -k 10 300 python bench.py --cpu-n 0 --no-ttt --workload scalar > gpurun_out/bench_exp_scalar.json 2> gpurun_out/bench_exp.err
python - <<'PY'
import json
for f in ("bench_exp_scalar",):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print("BENCH %s asm %.4f ms frac %.4f step %.3f" % (f, d["roofline_assembly"]["avg_ms"], d["roofline_assembly"]["frac"], d["ms_per_step"]))
PY
