"""Times the continuity and the scalar edge assembly at n^3 (HIP events of the library's own timer).
usage: python scripts/asmtime.py [n] [tag]   (knobs through NG_* / NALU_GPU_LIB in the environment)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from bench import RankSystem
import nalu_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tag = sys.argv[2] if len(sys.argv) > 2 else ""
R = RankSystem((n, n, n), 0, 1, "cuda:0")
g, F, S = R.g, R.F, R.S
cfg = nalu_amd.LinearSolverConfig(method="gmres", preconditioner="jacobi", tolerance=1e-5, max_iterations=1, kspace=2)
R.scalar_eqsys(cfg, "cuda:0")  # fills q, dqdx, gamma, mdot
out = {}
for wl in ("continuity", "scalar"):
    for rep in range(2):
        g.set_profiling(True)
        g.reset_timers()
        for _ in range(10):
            g.zeroSystem()
            if wl == "continuity":
                g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1)
            else:
                g.assemble_scalar_edges(F["coords"], F["dqdx"], F["vel"], F["q"], F["rho"], F["gamma"], F["area"], F["mdot"],
                                        hybrid_factor=1.0, alpha_upw=1.0, upw_factor=1.0, limiter=True)
        ms, cnt = g.timer("assemble_edges")
        g.set_profiling(False)
    out[wl] = ms / cnt
li = g.layout_info()
print("ASMTIME %-28s variant %d  continuity %.4f ms  scalar %.4f ms" % (tag, li["assembly_variant"], out["continuity"], out["scalar"]), flush=True)
