"""Cost of the per-kernel HIP-event timers inside the bench's timed region (development aid)."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import RankSystem
R = RankSystem((256, 256, 256), 0, 1, "cuda:0")
cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=50, kspace=50)
eq = R.continuity_eqsys(cfg)
for _ in range(3): eq.assemble_and_solve()
out = {}
for rep in range(2):
    for prof in (False, True):
        R.g.set_profiling(prof); R.g.reset_timers()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): eq.assemble_and_solve()
        torch.cuda.synchronize()
        out[f"profiling_{prof}_{rep}"] = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps(out))
