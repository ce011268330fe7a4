"""One GMRES+SGS solve of 3 iterations at n^3 for rocprof (development aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, nalu_amd
from bench import build_rank
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
g.zeroSystem()
g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
g.resetRows([0]); g.sumInto([0], [0.0 - float(F["p"][0].item())], [1.0]); g.loadComplete()
g.solve(nalu_amd.LinearSolverConfig(method="cg", preconditioner="sgs", tolerance=1e-12, max_iterations=3))
g.synchronize()
