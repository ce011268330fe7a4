"""SpMV micro-benchmark (development aid)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import build_rank

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
g.zeroSystem()
g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
g.loadComplete()
x = torch.rand(S.n_local, dtype=torch.float64, device="cuda:0"); y = torch.empty(S.n_owned, dtype=torch.float64, device="cuda:0")
g.set_profiling(True)
for _ in range(3): g.spmv(x, y)
g.reset_timers()
for _ in range(20): g.spmv(x, y)
ms, cnt = g.timer("spmv")
print(json.dumps({"abl": os.environ.get("NG_SPMV_ABL"), "spb": os.environ.get("NG_SPMV_SPB"), "xcd": os.environ.get("NG_XCD"), "spmv_ms": ms / cnt}))
