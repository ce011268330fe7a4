"""Latency budget of one CG iteration on the per-GPU share of the 8-GPU strong-scaling case
(development aid): a 256 x 256 x 32 slab (2.1 M rows) on ONE GPU, with the
comm callbacks installed at world size 1 (native RCCL through libnalu_gpu_rccl.so: every
reduction goes through ncclAllReduce and the one-thread scalar kernels of the multi-rank path)
and without.  80 % parallel efficiency at 8 GPUs needs <= 0.489 ms / 8 / 0.8 = 76 us per
iteration including the halo exchange, which a one-rank run cannot include."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.distributed as dist
import nalu_amd
from bench import RankSystem

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from nalu_amd.rccl import RcclWorld
res = {}
for dims in ((256, 256, 32), (256, 256, 64), (256, 256, 128), (256, 256, 256)):
    R = RankSystem(dims, 0, 1, "cuda:0")
    g = R.g
    for its in (50,):
        cfg = nalu_amd.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-30, max_iterations=its, check_every=its)
        eq = R.continuity_eqsys(cfg)
        eq.assemble_and_solve()
        out = {}
        for label in ("no_comm", "rccl_world1"):
            if label == "rccl_world1":
                w = RcclWorld(0, 1, 0)
                g.set_halo_plan([], [0], [], [0])
                w.attach(g)
            g.solve(cfg)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): g.solve(cfg)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            out[label] = dict(us_per_iteration=dt / its * 1e6, solve_ms=dt * 1e3)
        w.close()
        res["x".join(map(str, dims))] = out
    g.close()
print(json.dumps(res))
dist.destroy_process_group()
