"""Worker of the multi-rank tests (launched by torch.distributed.run, backend gloo).

mode "plan": CPU only.  Checks the z-slab partition, the halo plan and the all-reduce
             plumbing of nalu_amd.comm against the global mesh.
mode "gpu" : every rank drives the HIP library on cuda:0 (several ranks share the one GPU of
             the test box; on the 8-GPU node the backend is "nccl" and nothing else changes).
             Assembled owned rows, residual history and solution are compared with the CPU
             oracle's single-rank run of the global mesh.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)

import nalu_amd  # noqa: E402
from nalu_amd.comm import exchange_halo  # noqa: E402


def plan_test(rank, world):
    dims = (5, 4, 11)
    S = nalu_amd.BoxSlab(*dims, rank, world)
    # partition invariants
    tot = torch.tensor([S.n_owned], dtype=torch.int64)
    dist.all_reduce(tot)
    assert tot.item() == dims[0] * dims[1] * dims[2]
    assert np.all(np.diff(S.gids[:S.n_owned]) > 0)
    assert np.all(np.diff(S.gids[S.n_owned:S.n_owned + S.n_shared]) > 0)
    assert np.all(S.owners[:S.n_owned] == rank) and np.all(S.owners[S.n_owned:] != rank)
    # every local edge touches an owned node, oriented low gid -> high gid
    L, R = S.edge_nodes[:, 0], S.edge_nodes[:, 1]
    assert np.all((L < S.n_owned) | (R < S.n_owned)) and np.all(S.gids[L] < S.gids[R])
    # halo: owned values f(gid) must arrive in the shared/ghost slots
    f = lambda g: np.sin(0.37 * g) + g
    vec = torch.zeros(S.n_local, dtype=torch.float64)
    vec[:S.n_owned] = torch.as_tensor(f(S.gids[:S.n_owned]))
    send = vec[torch.as_tensor(S.send_lids.astype(np.int64))].contiguous()
    recv = torch.zeros(S.n_local - S.n_owned, dtype=torch.float64)
    for w in exchange_halo(send, recv, S.neighbors, S.send_offsets, S.recv_offsets):
        w.wait()
    vec[S.n_owned:] = recv
    assert np.array_equal(vec.numpy(), f(S.gids))
    # all-reduce of owned partial dots == global dot
    part = torch.tensor([float((vec[:S.n_owned] ** 2).sum())], dtype=torch.float64)
    dist.all_reduce(part)
    g_all = np.arange(1, dims[0] * dims[1] * dims[2] + 1)
    assert abs(part.item() - (f(g_all) ** 2).sum()) <= 1e-9 * part.item()
    # edges: every global edge is assembled for each owned endpoint exactly once
    cnt = torch.zeros(dims[0] * dims[1] * dims[2], dtype=torch.float64)
    for e in range(S.n_edges):
        for a in (L[e], R[e]):
            if a < S.n_owned:
                cnt[S.gids[a] - 1] += 1
    dist.all_reduce(cnt)
    i = np.arange(dims[0])[None, None, :]
    j = np.arange(dims[1])[None, :, None]
    k = np.arange(dims[2])[:, None, None]
    deg = (i > 0) * 1 + (i < dims[0] - 1) + (j > 0) + (j < dims[1] - 1) + (k > 0) + (k < dims[2] - 1)
    assert np.array_equal(cnt.numpy(), deg.reshape(-1).astype(float))


def gpu_test(rank, world, method):
    from oracle import pyoracle as po
    from common import Case, rel_err
    from nalu_amd.comm import TorchComm
    dims = (9, 8, 23)
    c = Case(po, 0, 0.125, dims=dims)
    S = nalu_amd.BoxSlab(*dims, rank, world)
    gi = S.gids - 1  # local -> global node index
    # local edge -> global edge index through the (L,R)-sorted global edge list
    key = c.m.edge_nodes[:, 0].astype(np.int64) * c.m.nnodes + c.m.edge_nodes[:, 1]
    lk = gi[S.edge_nodes[:, 0]] * c.m.nnodes + gi[S.edge_nodes[:, 1]]
    ge = np.searchsorted(key, lk)
    assert np.array_equal(key[ge], lk)
    dev = "cuda:0"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=dev)
    # NALU_TEST_SIDE_STREAM=1: the system lives on a non-default stream and every library call is
    # made with torch's DEFAULT stream current -- the comm callbacks must order their collectives
    # against the library's stream, not against the current one (nalu_amd/comm.py)
    side = torch.cuda.Stream() if os.environ.get("NALU_TEST_SIDE_STREAM") == "1" else None
    stream = side.cuda_stream if side is not None else torch.cuda.current_stream().cuda_stream
    assert side is None or stream != torch.cuda.current_stream().cuda_stream
    g = nalu_amd.GpuLinearSystem(1, 0, stream)
    g.beginLinearSystemConstruction(S.n_owned, S.n_shared, S.n_ghost, S.gids, S.owners, rank)
    g.buildEdgeToNodeGraph(S.edge_nodes)
    g.finalizeLinearSystem()
    g.set_halo_plan(S.neighbors, S.send_offsets, S.send_lids, S.recv_offsets)
    # NALU_TEST_NATIVE_RCCL=1: the callbacks are libnalu_gpu_rccl.so's (ncclAllReduce / grouped
    # ncclSend+ncclRecv called straight from the library) instead of torch.distributed's
    native = os.environ.get("NALU_TEST_NATIVE_RCCL") == "1"
    if native:
        from nalu_amd.rccl import RcclWorld
        rworld = RcclWorld(rank, world, 0)
        rworld.attach(g)
        comm = type("NoErrors", (), {"errors": []})()
    else:
        comm = TorchComm(g, S.neighbors, S.send_offsets, S.recv_offsets, dev)
    # nodal inputs: owned values only, shared/ghost copies through the library's halo exchange
    def nodal(a):
        v = t(a[gi])
        v[S.n_owned:] = 0.0
        torch.cuda.synchronize()  # torch wrote v on its own stream
        g.halo_exchange(v)
        g.synchronize()
        return v
    coords = torch.stack([nodal(c.m.xyz[:, d]) for d in range(3)])
    area = t(c.area[ge].T)
    torch.cuda.synchronize()
    g.zeroSystem()
    if method == "cg":
        dq = torch.stack([nodal(c.dqdx[:, d]) for d in range(3)])
        vel = torch.stack([nodal(c.vel[:, d]) for d in range(3)])
        q, rho, gam, vol = nodal(c.q), nodal(c.rho), nodal(c.gamma), nodal(c.vol)
        mdot, q9 = t(c.mdot[ge]), 0.9 * q
        torch.cuda.synchronize()
        g.assemble_scalar_edges(coords, dq, vel, q, rho, gam, area, mdot)
        g.assemble_scalar_mass_be(q9, q, rho, rho, vol, c.dt)
        ls = c.oracle_scalar()
        ls.assemble_scalar_mass_be(0.9 * c.q, c.q, c.rho, c.rho, c.vol, c.dt)
    else:
        dp = torch.stack([nodal(c.dpdx[:, d]) for d in range(3)])
        vel = torch.stack([nodal(c.vel[:, d]) for d in range(3)])
        pn, rn = nodal(c.p), nodal(c.rho)
        torch.cuda.synchronize()
        g.assemble_continuity_edges(coords, dp, vel, pn, rn, area, c.dt)
        ls = c.oracle_continuity()
        ls.reset_rows([0])
        ls.sum_into([0], [0.0 - c.p[0]], [1.0])
        if rank == 0:
            g.resetRows([0])
            g.sumInto([0], [0.0 - c.p[0]], [1.0])
    g.loadComplete()
    # owned rows == the rows of the global (np = 1) matrix, compared through global ids
    # (as unit_tests/UnitTestTpetra.C:189-202 does for np = 2)
    rp, ci, v = ls.owned_csr()
    grp, gci = g.graph()
    gv, gb = g.matrix_values(), g.rhs()
    for r in range(S.n_owned):
        G = gi[r]
        cols = gi[gci[grp[r]:grp[r + 1]]]
        o = np.argsort(cols)
        assert np.array_equal(cols[o], ci[rp[G]:rp[G + 1]])
        ref = v[rp[G]:rp[G + 1]]
        assert np.abs(gv[grp[r]:grp[r + 1]][o] - ref).max() <= 1e-12 * np.abs(ref).max()
    assert rel_err(gb, ls.owned_rhs()[gi[:S.n_owned]]) <= 1e-12
    kw = dict(tolerance=1e-9, max_iterations=200, kspace=30)
    if world == 1:
        # SGS is processor-local (Ifpack2): only at np = 1 does it equal the oracle's sweep
        ref = ls.solve(method, "sgs", **kw)
        g.solve(nalu_amd.LinearSolverConfig(method=method, preconditioner="sgs", **kw))
        assert g.linearSolveIterations == ref["iters"]
        assert np.abs(g.history - ref["history"]).max() <= 1e-10
    # Chebyshev applies the distributed operator (halo exchange inside every application), so
    # unlike SGS it is the same preconditioner on any rank count
    ref = ls.solve(method, "chebyshev", **kw)
    g.solve(nalu_amd.LinearSolverConfig(method=method, preconditioner="chebyshev", **kw))
    assert g.linearSolveIterations == ref["iters"], (g.linearSolveIterations, ref["iters"])
    assert np.abs(g.history - ref["history"]).max() <= 1e-10
    ref = ls.solve(method, "jacobi", **kw)
    x = torch.zeros(S.n_local, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    g.solve(nalu_amd.LinearSolverConfig(method=method, preconditioner="jacobi", **kw), x)
    assert not comm.errors, comm.errors
    assert g.linearSolveIterations == ref["iters"], (g.linearSolveIterations, ref["iters"])
    assert np.abs(g.history - ref["history"]).max() <= 1e-10
    # solution incl. shared/ghost copies (sync_field)
    assert rel_err(x.cpu().numpy(), ref["x"][gi]) <= 1e-9
    assert abs(g.linearResidual - ref["final_resid"]) <= 1e-10
    assert abs(g.nonLinearResidual - np.linalg.norm(ls.owned_rhs())) <= 1e-12 * np.linalg.norm(ls.owned_rhs())
    if native:
        rworld.close()
    g.close()


def main():
    mode = sys.argv[1]
    import datetime
    backend = os.environ.get("NALU_TEST_BACKEND", "gloo")
    kw = {}
    if backend == "nccl":  # RCCL: one rank per GPU, so only world size 1 fits the test box
        torch.cuda.set_device(0)
        kw["device_id"] = torch.device("cuda:0")
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=90), **kw)
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode == "plan":
        plan_test(rank, world)
    else:
        gpu_test(rank, world, sys.argv[2])
    dist.barrier()
    if rank == 0:
        print("DIST-OK", mode, world)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
