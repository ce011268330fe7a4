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
