set -e
R=$GRAFT_REPO_ROOT; cd $R
NALU_GPU_LIB=$R/scripts/ablate/libnalu_gpu_hostkry.so timeout -k 10 200 python scripts/krycheck.py host 2>&1 | grep KRYCHECK
timeout -k 10 200 python scripts/krycheck.py dev 2>&1 | grep KRYCHECK
python - <<'PY'
import numpy as np
a=np.load("gpurun_out/krycheck_host.npz"); b=np.load("gpurun_out/krycheck_dev.npz")
bad=[k for k in a.files if not (a[k].shape==b[k].shape and np.array_equal(a[k],b[k],equal_nan=True))]
print("KRYCHECK keys", len(a.files), "different:", bad)
for k in bad[:6]:
    print(k, a[k].shape, b[k].shape, np.abs(a[k][:min(len(a[k]),len(b[k]))]-b[k][:min(len(a[k]),len(b[k]))]).max())
PY
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_sgs_flow.py tests/test_distributed.py -x -q -m gpu --timeout 300 --timeout-method thread -k "solve or sgs or tfqmr or biCg or distributed or rank" > gpurun_out/kry_tests.log 2>&1 || { tail -30 gpurun_out/kry_tests.log; exit 1; }
tail -2 gpurun_out/kry_tests.log
