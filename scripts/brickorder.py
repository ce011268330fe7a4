"""Experiment (development aid): the same 256^3 box with brick-major NODE NUMBERING (so that a
SELL slice of 256 consecutive rows is a BX x BY x BZ brick and every array stays contiguous);
times the row-gather assembly and SpMV.  usage: brickorder.py n BX BY BZ"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import nalu_amd
from nalu_amd import synth
from nalu_amd.box import BoxSlab

n = int(sys.argv[1]); BX, BY, BZ = (int(v) for v in sys.argv[2:5])
S = BoxSlab(n, n, n, edge_order="sorted")
i, j, k = S.i.astype(np.int64), S.j.astype(np.int64), S.k.astype(np.int64)
nbx, nby = n // BX, n // BY
key = (((k // BZ) * nby + j // BY) * nbx + i // BX) * (BX * BY * BZ) + ((k % BZ) * BY + j % BY) * BX + i % BX
new_of_old = np.empty(S.n_local, np.int64); new_of_old[np.argsort(key, kind="stable")] = np.arange(S.n_local)
if BX == n and BY == 1:  # identity check
    assert np.array_equal(new_of_old, np.arange(S.n_local))
old_of_new = np.argsort(new_of_old)
en = new_of_old[S.edge_nodes.astype(np.int64)]
ed = S.edge_dir
o = np.lexsort((en[:, 0], ed))  # slot-major: direction, then new L id
en, ed = en[o], ed[o]
vol, area = S.uniform_geometry()
area = area[:, o]
xyz = S.coords()[:, old_of_new]
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
F = dict(coords=t(xyz), dpdx=t(synth.quadratic_grad(xyz.T, 0).T), vel=t(synth.velocity(xyz.T).T),
         p=t(synth.quadratic(xyz.T, 0)), rho=t(synth.density(xyz.T)), area=t(area))
g = nalu_amd.GpuLinearSystem(1, 0, torch.cuda.current_stream().cuda_stream)
g.beginLinearSystemConstruction(S.n_owned)
g.buildEdgeToNodeGraph(en.astype(np.int32))
g.finalizeLinearSystem()
res = {"n": n, "brick": [BX, BY, BZ]}
g.set_profiling(True)
for rep in range(7):
    if rep == 2: g.reset_timers()
    g.zeroSystem()
    g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
    g.loadComplete()
ms, cnt = g.timer("assemble_edges"); res["rows"] = ms / cnt
x = torch.rand(S.n_local, dtype=torch.float64, device="cuda:0"); y = torch.empty(S.n_owned, dtype=torch.float64, device="cuda:0")
g.reset_timers()
for _ in range(20): g.spmv(x, y)
ms, cnt = g.timer("spmv"); res["spmv"] = ms / cnt
# sanity: row sums of the Poisson matrix vanish away from nothing pinned
v = g.matrix_values(); rp, _ = g.graph()
res["max_rowsum"] = float(np.abs(np.add.reduceat(v, rp[:-1])).max())
print(json.dumps(res))
