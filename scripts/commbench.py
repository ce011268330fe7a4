"""CG iteration time with the comm callbacks active (RCCL, world size 1) vs without:
measures the host/stream overhead the multi-GPU path adds per iteration (development aid)."""
import os, sys, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
import nalu_amd
from nalu_amd.comm import TorchComm
from bench import build_rank

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
S, F, g, _ = build_rank(n, 0, 1, "cuda:0")
p0 = float(F["p"][0].item())
cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=50)
res = {}
for label in ("no_comm", "rccl_world1"):
    if label == "rccl_world1":
        e = np.zeros(0, np.int32)
        g.set_halo_plan(e, np.zeros(1, np.int32), e, np.zeros(1, np.int32))
        comm = TorchComm(g, [], [0], [0], "cuda:0")
    for rep in range(3):
        g.zeroSystem()
        g.assemble_continuity_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], 0.1, 1.0, 1.0, 1.0, "rows")
        g.resetRows([0]); g.sumInto([0], [0.0 - p0], [1.0]); g.loadComplete()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g.solve(cfg)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[label] = dt * 1e3 / g.linearSolveIterations
print(json.dumps(res))
dist.destroy_process_group()
