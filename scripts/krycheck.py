"""BiCGStab / TFQMR histories and solutions of a small pinned-Poisson and a scalar-transport system, written to
gpurun_out/krycheck_<tag>.npz (to compare two builds of the library bit for bit: NALU_GPU_LIB selects the build).
usage: python scripts/krycheck.py <tag>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import nalu_amd
from bench import RankSystem
tag = sys.argv[1]
R = RankSystem((48, 40, 36), 0, 1, "cuda:0")
g, F = R.g, R.F
cfg = nalu_amd.LinearSolverConfig
out = {}
eqc = R.continuity_eqsys(cfg(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=2))
eqc.assemble_and_solve()
for meth in ("biCgStab", "tfqmr"):
    for pre in ("jacobi", "sgs", "chebyshev", "none"):
        for ce in (1, 7):
            g.solve(cfg(method=meth, preconditioner=pre, tolerance=1e-9, max_iterations=60, check_every=ce))
            out[f"c_{meth}_{pre}_{ce}_hist"] = np.array(g.history)
            out[f"c_{meth}_{pre}_{ce}_x"] = g.solution()
            out[f"c_{meth}_{pre}_{ce}_it"] = np.array([g.linearSolveIterations, g.linearResidual])
eqs = R.scalar_eqsys(cfg(method="gmres", preconditioner="jacobi", tolerance=1e-5, max_iterations=2, kspace=5), "cuda:0")
eqs.assemble_and_solve()
for meth in ("biCgStab", "tfqmr"):
    for pre in ("jacobi", "sgs"):
        g.solve(cfg(method=meth, preconditioner=pre, tolerance=1e-12, max_iterations=25))
        out[f"s_{meth}_{pre}_hist"] = np.array(g.history)
        out[f"s_{meth}_{pre}_x"] = g.solution()
        out[f"s_{meth}_{pre}_it"] = np.array([g.linearSolveIterations, g.linearResidual])
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", f"krycheck_{tag}.npz"), **out)
print("KRYCHECK", tag, {k: (int(v[0]), float(v[1])) for k, v in out.items() if k.endswith("_it")}, flush=True)
