"""The composed loops the hot path exists for: every piece device resident, fields
that CHANGE between the iterations, against the oracle's composition of the same functions.
  continuity / projection : src/LowMachEquationSystem.C:621-717, 772-855
  scalar transport        : src/MixtureFractionEquationSystem.C:905-1013 (update_and_clip)"""
import numpy as np
import pytest

from common import Case, oracle_continuity_loop, oracle_scalar_loop, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-10  # north_star: final fields within 1e-10 relative


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def G():
    import nalu_amd
    nalu_amd.load()
    return nalu_amd


def _fields(torch, c):
    d = c.dev(torch)
    d["mdot"] = d["mdot"].clone()
    return d


@pytest.mark.parametrize("precond", ["jacobi", "sgs"])
def test_three_picard_iterations_of_the_pressure_projection(G, torch, oracle, precond):
    from nalu_amd import eqsys
    c = Case(oracle, 0, 0.125, dims=(12, 10, 14))
    kw = dict(method="cg", preconditioner=precond, tolerance=1e-12, max_iterations=600, kspace=50)
    ref = oracle_continuity_loop(oracle, c, 3, **{"precond" if k == "preconditioner" else k: v for k, v in kw.items()})
    F = _fields(torch, c)
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(c.m.nnodes)
    eq = eqsys.EquationSystem(g, G.LinearSolverConfig(**kw), "ContinuityEQS")
    eq.solverAlgDriver.solverAlgMap["INTERIOR"] = eqsys.AssembleContinuityEdgeSolverAlgorithm(c.m.edge_nodes, F, c.dt)
    eq.solverAlgDriver.solverAlgMap["REF_PRESSURE"] = eqsys.FixPressureAtNodeAlgorithm(0, 0.0, F["p"])
    eq.initialize()
    loop = eqsys.ContinuityProjectionLoop(eq, F, c.dt)
    its = []
    for _ in range(3):
        loop.solve_and_update(1, sync=g.synchronize)
        its.append(g.linearSolveIterations)
    assert its == ref["iters"] and min(its) > 20, (its, ref["iters"])
    for name, got in (("p", F["p"]), ("vel", F["vel"].T), ("dpdx", F["dpdx"].T), ("mdot", F["mdot"])):
        assert rel_err(got.cpu().numpy(), ref[name]) <= TOL, name
    # the fields did change: the loop is not three times the same system
    assert rel_err(ref["p"], c.p) > 1e-3 and rel_err(ref["vel"], c.vel) > 1e-6
    assert all(v > 0 for v in loop.timer.values())


@pytest.mark.parametrize("limiter", [False, True])
def test_three_iterations_of_scalar_transport_with_update_and_clip(G, torch, oracle, limiter):
    from nalu_amd import eqsys
    c = Case(oracle, 0, 0.125, dims=(12, 10, 14))
    opts = dict(alpha_upw=1.0, upw_factor=1.0, hybrid_factor=1.0, limiter=limiter)
    kw = dict(method="gmres", preconditioner="jacobi", tolerance=1e-12, max_iterations=300, kspace=50)
    okw = {"precond" if k == "preconditioner" else k: v for k, v in kw.items()}
    ref = oracle_scalar_loop(oracle, c, 3, opts, delta_clip=0.05, **okw)
    F = _fields(torch, c)
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(c.m.nnodes)
    eq = eqsys.EquationSystem(g, G.LinearSolverConfig(**kw), "MixtureFractionEQS")
    eq.solverAlgDriver.solverAlgMap["INTERIOR"] = eqsys.AssembleScalarEdgeSolverAlgorithm(c.m.edge_nodes, F, **opts)
    eq.solverAlgDriver.solverAlgMap["MASS"] = eqsys.ScalarMassBackwardEulerAlgorithm(
        np.arange(c.m.nnodes, dtype=np.int32), 0.9 * F["q"], F["q"], F["rho"], F["rho"], F["vol"], c.dt)
    eq.initialize()
    loop = eqsys.ScalarTransportLoop(eq, F, delta_clip=0.05)
    its, clips = [], []
    for _ in range(3):
        loop.solve_and_update(1, diagnostics=True)
        its.append(g.linearSolveIterations)
        clips.append(loop.clip_info.copy())
    assert its == ref["iters"], (its, ref["iters"])
    for a, b in zip(clips, ref["clips"]):
        assert a[0] == b[0] and a[1] == b[1] and a[0] + a[1] > 0, (a, b)  # some nodes are clipped, the same ones
        assert abs(a[2] - b[2]) <= TOL * max(1.0, abs(b[2])) and abs(a[3] - b[3]) <= TOL * max(1.0, abs(b[3]))
    assert rel_err(F["q"].cpu().numpy(), ref["q"]) <= TOL
    assert rel_err(loop.qUF.cpu().numpy(), ref["q_unfiltered"]) <= TOL
    assert rel_err(F["dqdx"].T.cpu().numpy(), ref["dqdx"]) <= TOL
    assert F["q"].min().item() >= -0.05 and F["q"].max().item() <= 1.05
    # without diagnostics: no host round trip, same field
    q_before = F["q"].clone()
    assert g.update_and_clip(torch.zeros_like(q_before), F["q"]) is None
    assert torch.equal(F["q"], q_before.clamp(0.0, 1.0))
