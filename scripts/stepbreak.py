"""Where the bench step's time goes besides the two big kernels (development aid): wall time per
phase of EquationSystem.assemble_and_solve with a device sync after each, and the library timers."""
import os, sys, time, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from bench import RankSystem
R = RankSystem((256, 256, 256), 0, 1, "cuda:0")
g = R.g
cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=50, kspace=50)
eq = R.continuity_eqsys(cfg)
for _ in range(2): eq.assemble_and_solve()
sync = torch.cuda.synchronize
res = {}
def phase(name, fn):
    sync(); t0 = time.perf_counter(); fn(); sync(); res[name] = res.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
N = 5
g.set_profiling(True); g.reset_timers()
for _ in range(N):
    phase("zero", g.zeroSystem)
    phase("algs", lambda: eq.solverAlgDriver.execute(g))
    phase("loadComplete", g.loadComplete)
    phase("solve", lambda: g.solve(cfg))
sync(); t0 = time.perf_counter()
for _ in range(N): eq.assemble_and_solve()
sync(); res["whole_step_no_sync"] = (time.perf_counter() - t0) * 1e3
out = {k: v / N for k, v in res.items()}
for name in ("zero", "assemble_edges", "spmv", "solve", "precond", "bc"):
    ms, cnt = g.timer(name); out["timer_" + name] = [ms / (2 * N), cnt / (2 * N)]
print(json.dumps(out))
