"""Runs 3 continuity assemblies (rows) and 5 SpMVs at n^3 for PMC collection."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from bench import build_rank
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
skew = int(os.environ.get("NALU_FIELD_SKEW", "0"))
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
for _ in range(3):
    g.zeroSystem()
    g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
x = torch.rand(S.n_local, dtype=torch.float64, device="cuda:0"); y = torch.empty(S.n_owned, dtype=torch.float64, device="cuda:0")
for _ in range(5): g.spmv(x, y)
g.synchronize()
