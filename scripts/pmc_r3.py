"""The kernels of the hot path, a few launches each, for rocprofv3 --pmc passes (round 3).
usage (through scripts/r3_profile.sh): python3 scripts/pmc_r3.py <n>
continuity + scalar edge assembly, Jacobi-CG (k_spmv<DOT>, k_cg_update, k_cg_p, k_cg_flush_x), GMRES(50)/Jacobi
(k_spmv<DINV>, k_gm_dots_all, k_gm_upd_dots, k_gm_update) and CG + exact SGS (k_flow_sweep) at n^3."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import RankSystem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = RankSystem((n, n, n), 0, 1, "cuda:0")
g, F = R.g, R.F
cfg = nalu_amd.LinearSolverConfig
eqs = R.scalar_eqsys(cfg(method="gmres", preconditioner="jacobi", tolerance=1e-30, max_iterations=30, kspace=50), "cuda:0")
eqc = R.continuity_eqsys(cfg(method="cg", preconditioner="jacobi", tolerance=1e-30, max_iterations=20))
for _ in range(3):
    g.zeroSystem()
    g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1)
for _ in range(3):
    g.zeroSystem()
    g.assemble_scalar_edges(F["coords"], F["dqdx"], F["vel"], F["q"], F["rho"], F["gamma"], F["area"], F["mdot"],
                            hybrid_factor=1.0, alpha_upw=1.0, upw_factor=1.0, limiter=True)
eqs.assemble_and_solve()   # 30 GMRES steps: the ICGS kernels up to 30 basis vectors
eqc.assemble_and_solve()   # 20 CG iterations
if n <= 256 and os.environ.get("PMC_SGS", "1") == "1":
    g.solve(cfg(method="cg", preconditioner="sgs", tolerance=1e-30, max_iterations=3))
g.synchronize()
