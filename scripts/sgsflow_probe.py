"""CG + exact SGS on an nx x ny x nz lattice with random diagonally dominant values (development aid:
run under `rocprofv3 --kernel-trace --stats` to read the duration of one dataflow sweep).
usage: sgsflow_probe.py nx ny nz [its]      (NG_SGS_TILE / NG_SGS_MODE as usual)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import nalu_amd  # noqa: E402
from nalu_amd.box import box_edges  # noqa: E402

nx, ny, nz = (int(a) for a in sys.argv[1:4])
its = int(sys.argv[4]) if len(sys.argv) > 4 else 10
n = nx * ny * nz
e, _ = box_edges(nx, ny, nz)
g = nalu_amd.GpuLinearSystem()
g.beginLinearSystemConstruction(n)
g.buildEdgeToNodeGraph(np.ascontiguousarray(e, np.int32))
g.finalizeLinearSystem()
rp, ci = g.graph()
rs = np.random.RandomState(1)
v = rs.uniform(-1, 1, len(ci))
rows = np.repeat(np.arange(n), np.diff(rp))
v[ci == rows] = np.bincount(rows, np.abs(v), n) + 1.0
g.set_matrix_values(v)
g.set_rhs(rs.uniform(-1, 1, n))
cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="sgs", tolerance=1e-30, max_iterations=its)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g.solve(cfg)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{nx}x{ny}x{nz} tile {os.environ.get('NG_SGS_TILE', 'default')}: {dt * 1e3 / its:.3f} ms per CG+sgs iteration, "
      f"rel {g.history[-1]:.3e}, schedule {g.sgs_schedule()}")
