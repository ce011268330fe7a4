set -e
R=$GRAFT_REPO_ROOT; cd $R
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 600 --timeout-method thread -k "skewed or spmv or solve_history" 2>&1 | tail -15
timeout -k 10 300 python bench.py --cpu-n 0 --no-ttt > gpurun_out/bench_exp.json 2> gpurun_out/bench_exp.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_exp.json").read().strip().splitlines()[-1])
print("BENCH asm %.4f ms frac %.4f step %.3f spmv %.4f" % (d["roofline_assembly"]["avg_ms"], d["roofline_assembly"]["frac"], d["ms_per_step"], d["roofline"]["avg_ms"]))
PY
