"""SpMV (+ fused dot, inside CG) time per row for boxes of about the same size but different plane
strides (development aid): is the 256^3 cube special (power-of-two strides between the gathered lines)?"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import RankSystem
out = {}
cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-30, max_iterations=30, kspace=50, check_every=30)
for dims in ((256, 256, 256), (256, 255, 257), (256, 264, 248), (256, 272, 241), (512, 512, 64), (512, 128, 256), (256, 512, 128), (256, 128, 512)):
    R = RankSystem(dims, 0, 1, "cuda:0")
    eq = R.continuity_eqsys(cfg)
    eq.assemble_and_solve()
    R.g.set_profiling(True); R.g.reset_timers()
    for _ in range(2): R.g.solve(cfg)
    sp, ns = R.g.timer("spmv"); so, _ = R.g.timer("solve")
    n = R.S.n_owned
    out["x".join(map(str, dims))] = dict(spmv_ms=sp / ns, ns_per_row=sp / ns * 1e6 / n, cg_iter_ms=so / 2 / 30,
                                         cg_ns_per_row=so / 2 / 30 * 1e6 / n)
    R.g.close()
print(json.dumps(out, indent=1))
