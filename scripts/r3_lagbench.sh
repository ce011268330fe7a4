# round 3: parity of the lagged marching kernel, then bench lines with it and with round 2's marching kernel
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 500 python -u -m pytest tests/test_gpu_bench_path.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_lag_tests.log 2>&1 || { tail -30 gpurun_out/r3_lag_tests.log; exit 1; }
tail -2 gpurun_out/r3_lag_tests.log
for wl in continuity scalar; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-n 0 --no-ttt --workload $wl > gpurun_out/r3_lag_$wl.json 2> gpurun_out/r3_lag_$wl.err
  NG_ASM_LAG=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-n 0 --no-ttt --workload $wl > gpurun_out/r3_lagoff_$wl.json 2>> gpurun_out/r3_lag_$wl.err
done
python - <<'PY'
import json
for wl in ("continuity", "scalar"):
    for k in ("lag", "lagoff"):
        d = json.load(open(f"gpurun_out/r3_{k}_{wl}.json"))
        print(wl, k, "step %.3f ms" % d["ms_per_step"], "asm %.4f ms frac %.3f" % (d["roofline_assembly"]["avg_ms"], d["roofline_assembly"]["frac"]),
              "variant", d["config"]["assembly_variant"], "spmv %.4f" % d["roofline"]["avg_ms"])
PY
