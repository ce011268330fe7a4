"""Dataflow SGS sweeps (csrc/precond.hip k_flow_sweep, csrc/sgs_flow.hpp) against the sweeps with one
launch per level and against the oracle's sequential loops.

The preconditioner is Ifpack2's sequential symmetric Gauss-Seidel (src/LinearSolverConfig.C:71-75);
every schedule performs the same operations in the same order per row, so the solves must agree
BIT FOR BIT between schedules, and to 1e-10 with the oracle (TOL_SOLVE, as for every other solve)."""
import os

import numpy as np
import pytest

from common import rel_err

pytestmark = pytest.mark.gpu
TOL_SOLVE = 1e-10


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def G():
    import nalu_amd
    nalu_amd.load()  # raises if libnalu_gpu.so is missing: no fallback
    return nalu_amd


class _Env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _system(G, oracle, m, elem, seed):
    """graph of a box mesh (edge graph: 7-point rows; element graph: 27-point rows) with random
    strictly diagonally dominant values, the same on both sides"""
    n = m.nnodes
    ls = oracle.LinSys(1, n)
    ls.add_connections(m.hex_nodes if elem else m.edge_nodes)
    ls.finalize()
    ls.zero()
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(n)
    (g.buildElemToNodeGraph if elem else g.buildEdgeToNodeGraph)(m.hex_nodes if elem else m.edge_nodes)
    g.finalizeLinearSystem()
    rp, ci = g.graph()
    rs = np.random.RandomState(seed)
    v = rs.uniform(-1, 1, len(ci))
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows] = np.bincount(rows, np.abs(v), n) + 1.0
    b = rs.uniform(-1, 1, n)
    oracle.set_values(ls, v)
    oracle.set_rhs(ls, b)
    g.set_matrix_values(v)
    g.set_rhs(b)
    return g, ls


def _solve(G, g, method, **kw):
    g.solve(G.LinearSolverConfig(method=method, preconditioner="sgs", **kw))
    return g.linearSolveIterations, np.array(g.history), g.solution().copy()


CASES = [
    # dims, element graph, environment of the dataflow build, expected (tiled, chunks > 1)
    ((20, 20, 20), False, {}, True),
    ((20, 20, 20), False, {"NG_SGS_TILE": "4,4"}, True),
    ((20, 20, 20), False, {"NG_SGS_TILE": "0", "NG_SGS_CHUNK": "256"}, False),
    ((256, 5, 4), False, {"NG_SGS_TILE": "2,2"}, True),
    ((33, 7, 19), False, {"NG_SGS_TILE": "8,16"}, True),
    ((64, 48, 40), False, {}, True),
    # one consumer wave per chunk instead of the two that take the steps in turn (7-point rows, the default)
    ((64, 48, 40), False, {"NG_SGS_CONSUMERS": "1"}, True),
    ((33, 7, 19), False, {"NG_SGS_TILE": "8,16", "NG_SGS_CONSUMERS": "1"}, True),
    ((256, 5, 4), False, {"NG_SGS_TILE": "2,2", "NG_SGS_DEPTH": "4"}, True),   # look-ahead of four steps: one consumer
    ((7, 6, 5), True, {}, False),
    ((12, 11, 10), True, {"NG_SGS_CHUNK": "128"}, False),
]


@pytest.mark.parametrize("dims,elem,env,tiled", CASES)
def test_dataflow_sweeps_equal_level_sweeps_bit_for_bit(G, torch, oracle, dims, elem, env, tiled):
    m = oracle.BoxMesh(*dims)
    kw = dict(tolerance=1e-11, max_iterations=60, kspace=30)
    with _Env(NG_SGS_MODE=None, NG_SGS_INPLACE=None, NG_SGS_TILE=None, NG_SGS_CHUNK=None, NG_SGS_CONSUMERS=None,
              NG_SGS_DEPTH=None):
        with _Env(**env):
            g, ls = _system(G, oracle, m, elem, seed=3)
            res = {meth: _solve(G, g, meth, **kw) for meth in ("cg", "gmres")}
            sched = g.sgs_schedule()
        assert sched["mode"] == "flow" and sched["tiled"] == tiled, sched
        assert sched["chunks"] >= 1 and sched["max_steps"] >= 1
        if not elem and np.prod(dims) > 2048:
            assert sched["chunks"] > 1  # values really travel between workgroups
        with _Env(NG_SGS_MODE="levels"):
            g2, _ = _system(G, oracle, m, elem, seed=3)
            for meth in ("cg", "gmres"):
                it, hist, x = _solve(G, g2, meth, **kw)
                assert g2.sgs_schedule()["mode"] == "levels"
                assert it == res[meth][0], meth
                assert np.array_equal(hist, res[meth][1]), meth
                assert np.array_equal(x, res[meth][2]), meth
    for meth in ("cg", "gmres"):
        ref = ls.solve(meth, "sgs", **kw)
        assert res[meth][0] == ref["iters"]
        assert np.abs(res[meth][1] - ref["history"]).max() <= TOL_SOLVE
        assert rel_err(res[meth][2], ref["x"]) <= 1e-9


def test_dataflow_sweeps_repeat_and_survive_a_converged_solve(G, torch, oracle):
    """The exchange buffers double as 'not computed yet' flags and are cleared before every sweep:
    many applications in one solve, a solve that converges (later kernels return early on the device's
    done flag), host-driven methods (no done flag) and a second solve on new values must all agree with
    the level sweeps."""
    m = oracle.BoxMesh(24, 18, 16)
    out = {}
    for mode in (None, "levels"):
        with _Env(NG_SGS_MODE=mode, NG_SGS_TILE="4,4"):
            g, ls = _system(G, oracle, m, False, seed=11)
            r = []
            r.append(_solve(G, g, "cg", tolerance=1e-3, max_iterations=200))     # converges early
            r.append(_solve(G, g, "gmres", tolerance=1e-12, max_iterations=37, kspace=7))
            r.append(_solve(G, g, "biCgStab", tolerance=1e-10, max_iterations=50))
            r.append(_solve(G, g, "tfqmr", tolerance=1e-10, max_iterations=50))
            rs = np.random.RandomState(5)
            g.set_rhs(rs.uniform(-1, 1, m.nnodes))
            r.append(_solve(G, g, "cg", tolerance=1e-12, max_iterations=25))
            assert g.sgs_schedule()["mode"] == ("flow" if mode is None else "levels")
            out[mode] = r
    for a, b in zip(out[None], out["levels"]):
        assert a[0] == b[0]
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_wide_rows_fall_back_to_level_sweeps(G, torch, oracle):
    """rows with more than 16 entries below the diagonal are not taken by the dataflow kernel: the level
    sweeps run instead, and say so"""
    n = 600
    rs = np.random.RandomState(9)
    L = rs.randint(0, n, n * 20)
    R = (L + 1 + rs.randint(0, n - 1, len(L))) % n
    e = np.concatenate([np.stack([L, R], 1), np.stack([np.arange(n - 1), np.arange(1, n)], 1)])
    e = np.unique(np.sort(e, axis=1), axis=0).astype(np.int32)
    ls = oracle.LinSys(1, n)
    ls.add_connections(e)
    ls.finalize()
    ls.zero()
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(n)
    g.buildEdgeToNodeGraph(e)
    g.finalizeLinearSystem()
    rp, ci = g.graph()
    v = rs.uniform(-1, 1, len(ci))
    rows = np.repeat(np.arange(n), np.diff(rp))
    v[ci == rows] = np.bincount(rows, np.abs(v), n) + 1.0
    b = rs.uniform(-1, 1, n)
    oracle.set_values(ls, v)
    oracle.set_rhs(ls, b)
    g.set_matrix_values(v)
    g.set_rhs(b)
    kw = dict(tolerance=1e-11, max_iterations=40, kspace=20)
    it, hist, x = _solve(G, g, "gmres", **kw)
    assert g.sgs_schedule()["mode"] == "levels"
    ref = ls.solve("gmres", "sgs", **kw)
    assert it == ref["iters"] and np.abs(hist - ref["history"]).max() <= TOL_SOLVE


def test_a_sweep_that_gives_up_is_redone_on_the_level_schedule(G, torch, oracle):
    """NG_SGS_MAX_POLL=1: a consumer gives up the first time a value of another chunk is not there yet
    (what a workgroup that never got its turn would cause after the real bound).  nalu_gpu_solve must
    not return garbage or NALU_GPU_ESTATE: it redoes the solve on the level schedule in the same call
    (same bits as NG_SGS_MODE=levels), reports the fall-back, and the handle stays on that schedule."""
    m = oracle.BoxMesh(20, 20, 20)
    kw = dict(tolerance=1e-11, max_iterations=60, kspace=30)
    with _Env(NG_SGS_MODE=None, NG_SGS_INPLACE=None, NG_SGS_TILE="4,4", NG_SGS_CHUNK=None):
        with _Env(NG_SGS_MODE="levels"):
            g2, _ = _system(G, oracle, m, False, seed=3)
            ref = {meth: _solve(G, g2, meth, **kw) for meth in ("cg", "gmres", "tfqmr")}
        g, ls = _system(G, oracle, m, False, seed=3)
        with _Env(NG_SGS_MAX_POLL="1"):
            got = _solve(G, g, "cg", **kw)
        sched = g.sgs_schedule()
        assert sched["mode"] == "levels" and sched["fell_back"], sched
        assert got[0] == ref["cg"][0] and np.array_equal(got[1], ref["cg"][1]) and np.array_equal(got[2], ref["cg"][2])
        # later solves on this handle: level schedule, no second fall-back needed
        for meth in ("gmres", "tfqmr"):
            got = _solve(G, g, meth, **kw)
            assert got[0] == ref[meth][0] and np.array_equal(got[1], ref[meth][1]), meth
        assert g.sgs_schedule()["fell_back"]
        # a fresh handle is on the dataflow schedule again, and with the real bound it does not give up
        g3, _ = _system(G, oracle, m, False, seed=3)
        got = _solve(G, g3, "cg", **kw)
        assert g3.sgs_schedule()["mode"] == "flow" and not g3.sgs_schedule()["fell_back"]
        assert np.array_equal(got[1], ref["cg"][1])
    o = ls.solve("cg", "sgs", **kw)
    assert ref["cg"][0] == o["iters"] and np.abs(ref["cg"][1] - o["history"]).max() <= TOL_SOLVE


def test_an_all_ones_nan_in_the_rhs_propagates_instead_of_stalling(G, torch, oracle):
    """the exchange buffer's 'not computed yet' pattern is the all-ones NaN: a computed value with those
    very bits (a 0xFF-filled rhs) is published as another NaN, so the sweep finishes at once -- the
    solve ends with a NaN residual, not with a time-out and NALU_GPU_ESTATE"""
    import time
    m = oracle.BoxMesh(20, 20, 20)
    with _Env(NG_SGS_MODE=None, NG_SGS_INPLACE=None, NG_SGS_TILE="4,4", NG_SGS_CHUNK=None):
        g, _ = _system(G, oracle, m, False, seed=3)
        b = np.random.RandomState(1).uniform(-1, 1, m.nnodes)
        b.view(np.uint64)[::7] = np.uint64(0xFFFFFFFFFFFFFFFF)
        g.set_rhs(b)
        t0 = time.perf_counter()
        g.solve(G.LinearSolverConfig(method="cg", preconditioner="sgs", tolerance=1e-8, max_iterations=3))
        assert time.perf_counter() - t0 < 5.0
        sched = g.sgs_schedule()
        assert sched["mode"] == "flow" and not sched["fell_back"], sched
        assert np.isnan(g.linearResidual)
