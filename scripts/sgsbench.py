"""Time CG/GMRES iterations with Jacobi / Chebyshev / SGS at n^3 (development aid)."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import build_rank

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
p0 = float(F["p"][0].item())
res = {"n": n}
for method, pc, its in (("cg", "jacobi", 50), ("cg", "chebyshev", 50), ("cg", "mt_sgs", 50), ("gmres", "mt_sgs", 20), ("cg", "sgs", 10), ("gmres", "jacobi", 20), ("gmres", "chebyshev", 20), ("gmres", "sgs", 10)):
    cfg = nalu_amd.LinearSolverConfig(method=method, preconditioner=pc, tolerance=1e-12, max_iterations=its, kspace=50)
    for rep in range(2):
        g.zeroSystem()
        g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
        g.resetRows([0]); g.sumInto([0], [0.0 - p0], [1.0]); g.loadComplete()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.solve(cfg)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[f"{method}+{pc}"] = dict(ms_per_it=dt * 1e3 / max(g.linearSolveIterations, 1), its=g.linearSolveIterations, rel=float(g.history[-1]))
print(json.dumps(res))
