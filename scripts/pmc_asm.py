"""Runs 3 continuity assemblies (rows) and 5 SpMVs at n^3 for PMC collection."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_rank
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
for _ in range(3):
    g.zeroSystem()
    g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
x = torch.rand(S.n_local, dtype=torch.float64, device="cuda:0"); y = torch.empty(S.n_owned, dtype=torch.float64, device="cuda:0")
for _ in range(5): g.spmv(x, y)
g.synchronize()
