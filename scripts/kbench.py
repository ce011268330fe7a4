"""Kernel micro-benchmarks (development aid): times the assembly variants and SpMV at n^3."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import build_rank

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
res = {"lib": os.path.basename(nalu_amd.LIB_PATH), "n": n}
skew = int(os.environ.get("NALU_FIELD_SKEW", "0"))  # doubles of padding between component arrays
if skew:
    comps = sum(v.shape[0] if v.dim() == 2 else 1 for v in F.values())
    longest = max(v.shape[-1] for v in F.values())
    big = torch.empty(comps * (longest + skew), dtype=torch.float64, device="cuda:0")
    off = 0
    for k in list(F):
        v = F[k]
        r = v.shape[0] if v.dim() == 2 else 1
        view = big[off:off + r * (longest + skew)].view(r, longest + skew)[:, :v.shape[-1]]
        view.copy_(v.view(r, -1))
        F[k] = view if v.dim() == 2 else view[0]
        off += r * (longest + skew)
    res["skew"] = skew
for mode in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["rows"]):
    g.set_profiling(True)
    for rep in range(7):
        if rep == 2: g.reset_timers()  # two warm-up passes
        g.zeroSystem()
        g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, mode)
        g.loadComplete()
    ms, cnt = g.timer("assemble_edges")
    res[mode] = ms / cnt
x = torch.rand(S.n_local, dtype=torch.float64, device="cuda:0"); y = torch.empty(S.n_owned, dtype=torch.float64, device="cuda:0")
g.reset_timers()
for _ in range(20): g.spmv(x, y)
ms, cnt = g.timer("spmv"); res["spmv"] = ms / cnt
print(json.dumps(res))
