"""Per-node cost of the bench step on the 256^3 cube (N = 1 workload) against one 512 x 512 x 64 slab
(the per-GPU share of the weak-scaling workload) on ONE GPU without a communicator (development aid)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import RankSystem, timed_steps
out = {}
for dims in ((256, 256, 256), (512, 512, 64)):
    R = RankSystem(dims, 0, 1, "cuda:0")
    cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=50, kspace=50)
    eq = R.continuity_eqsys(cfg)
    dt = timed_steps(eq, R.g, 5, 2, torch.cuda.synchronize)
    asm, na = R.g.timer("assemble_edges"); sp, ns = R.g.timer("spmv")
    out["x".join(map(str, dims))] = dict(ms_per_step=dt / 5 * 1e3, assemble_ms=asm / na, spmv_ms=sp / ns,
                                         variant=R.g.layout_info()["assembly_variant"])
    R.g.close()
print(json.dumps(out))
