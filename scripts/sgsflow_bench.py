"""Exact SGS at n^3: ms per application for the dataflow sweeps (tile shapes) and the level sweeps.

usage: sgsflow_bench.py n variant [variant ...]   variant = levels | flow:By,Bz[:CHUNK[:DEPTH]]
The application time is (CG+sgs iteration) - (CG+jacobi iteration) + the Jacobi scaling it replaces,
and, with profiling on, the device time of the preconditioner kernels themselves (rocprofv3 is the
authority: profiles/r2_sgsflow_stats.csv)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import nalu_amd  # noqa: E402
from bench import build_rank  # noqa: E402

n = int(sys.argv[1])
variants = sys.argv[2:] or ["flow:8,8"]
its = int(os.environ.get("SGS_ITS", "10"))
out = {"n": n, "its": its}


def run(g, F, p0, method, pc):
    cfg = nalu_amd.LinearSolverConfig(method=method, preconditioner=pc, tolerance=1e-30, max_iterations=its,
                                      kspace=50)
    best = None
    for rep in range(3):
        g.zeroSystem()
        g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0,
                                    1.0, "rows")
        g.resetRows([0]); g.sumInto([0], [0.0 - p0], [1.0]); g.loadComplete()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.solve(cfg)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if rep == 0:
            first = dt
        best = dt if best is None else min(best, dt)
    return best * 1e3 / max(g.linearSolveIterations, 1), first, float(g.history[-1]), g.linearSolveIterations


for v in variants:
    for k in ("NG_SGS_MODE", "NG_SGS_TILE", "NG_SGS_CHUNK", "NG_SGS_DEPTH"):
        os.environ.pop(k, None)
    if v == "levels":
        os.environ["NG_SGS_MODE"] = "levels"
    else:
        parts = v.split(":")
        os.environ["NG_SGS_TILE"] = parts[1]
        if len(parts) > 2 and parts[2]:
            os.environ["NG_SGS_CHUNK"] = parts[2]
        if len(parts) > 3:
            os.environ["NG_SGS_DEPTH"] = parts[3]
    S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
    p0 = float(F["p"][0].item())
    if "jacobi_ms_per_it" not in out:
        out["jacobi_ms_per_it"] = run(g, F, p0, "cg", "jacobi")[0]
    ms, first, rel, it = run(g, F, p0, "cg", "sgs")
    out[v] = dict(cg_ms_per_it=round(ms, 4), sgs_ms_per_apply=round(ms - out["jacobi_ms_per_it"], 4),
                  first_solve_s=round(first, 3), rel=rel, its=it, schedule=g.sgs_schedule())
    print(v, json.dumps(out[v]), flush=True)
    del g, S, F
    torch.cuda.empty_cache()
print(json.dumps(out))
