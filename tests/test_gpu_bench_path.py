"""Parity of the exact code path bench.py times ("What's weak" 1-5): edges numbered
slot-major (BoxSlab(edge_order="slot")) so that the row kernel computes its edge tags instead of
loading them, slices that are exactly one x-line (nx = 256), the pinned-Poisson system under
Jacobi-CG with the regression's limits, BASELINE config 2 at its full size (64^3) and the
scalar-transport + GMRES(50) step of config 4 -- each against the CPU oracle on the same inputs.

The oracle always runs its edge loop over (L,R)-sorted edges; the device gets the same edges in
slot-major order with the per-edge fields permuted accordingly.  The row gather adds the
shares of a row in CSR entry order whatever the edge numbering, so matrix and rhs must be equal
BIT FOR BIT (src/AssembleContinuityEdgeSolverAlgorithm.C:119-208,
src/AssembleScalarEdgeSolverAlgorithm.C:139-314, src/TpetraLinearSystem.C:1454-1507)."""
import numpy as np
import pytest

from common import rel_err
from nalu_amd.synth import density, quadratic, quadratic_grad, velocity

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
    nalu_amd.load()
    return nalu_amd


class BenchCase:
    """bench.py's build_rank() inputs on one rank: BoxSlab numbering (slot-major edges by
    default), lattice geometry (uniform) or the oracle's Hex8 SCS/SCV geometry (perturbed), the
    synthetic fields of nalu_amd.synth -- and the same data in the oracle's (L,R)-sorted edge
    order."""

    def __init__(self, G, oracle, dims, perturb=0.0, edge_order="slot"):
        if edge_order == "suggested":  # (L,R)-sorted edges renumbered by nalu_gpu_suggest_edge_order
            self.S = S = G.BoxSlab(*dims, edge_order="sorted")
            perm = G.suggest_edge_order(S.edge_nodes)
            assert sorted(perm.tolist()) == list(range(S.n_edges))
            S.edge_nodes = np.ascontiguousarray(S.edge_nodes[perm])
            S.edge_dir = S.edge_dir[perm]
        else:
            self.S = S = G.BoxSlab(*dims, edge_order=edge_order)
        self.dims = dims
        en = S.edge_nodes
        # oracle edge order: sorted by (L, R); to_sorted[i] = device edge index of oracle edge i
        self.to_sorted = np.lexsort((en[:, 1], en[:, 0]))
        self.en_sorted = np.ascontiguousarray(en[self.to_sorted])
        if perturb:
            m = oracle.BoxMesh(*dims, perturb=perturb)
            assert np.array_equal(m.edge_nodes, self.en_sorted)
            self.xyz = m.xyz
            self.vol, area_sorted = m.geometry()
            self.area = np.empty_like(area_sorted)
            self.area[self.to_sorted] = area_sorted  # device (slot-major) order, (E, 3)
        else:
            self.xyz = np.ascontiguousarray(S.coords().T)
            self.vol, a = S.uniform_geometry()
            self.area = np.ascontiguousarray(a.T)
        self.area_sorted = np.ascontiguousarray(self.area[self.to_sorted])
        x = self.xyz
        self.p, self.q = quadratic(x, 0), quadratic(x, 1)
        self.dpdx, self.dqdx = quadratic_grad(x, 0), quadratic_grad(x, 1)
        self.vel, self.rho = velocity(x), density(x)
        self.gamma = np.full(S.n_local, 0.2)
        self.dt = 0.1
        self.o = oracle
        self.mdot_sorted = oracle.compute_mdot(self.en_sorted, self.area_sorted, x, self.dpdx, self.vel,
                                               self.p, self.rho, self.dt)
        self.mdot = np.empty_like(self.mdot_sorted)
        self.mdot[self.to_sorted] = self.mdot_sorted

    def dev(self, torch):
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device="cuda:0")
        soa = lambda a: t(np.ascontiguousarray(a.T))
        return dict(coords=soa(self.xyz), dpdx=soa(self.dpdx), dqdx=soa(self.dqdx), vel=soa(self.vel),
                    p=t(self.p), q=t(self.q), rho=t(self.rho), gamma=t(self.gamma), area=soa(self.area),
                    mdot=t(self.mdot), vol=t(self.vol))

    def gpu_system(self, G, extra_node_graph=False):
        S = self.S
        g = G.GpuLinearSystem()
        g.beginLinearSystemConstruction(S.n_owned, gids=S.gids)
        g.buildEdgeToNodeGraph(S.edge_nodes)
        if extra_node_graph:
            g.buildNodeGraph(np.arange(S.n_owned, dtype=np.int32))
        g.finalizeLinearSystem()
        g.zeroSystem()
        return g

    def oracle_system(self):
        ls = self.o.LinSys(1, self.S.n_owned)
        ls.add_connections(self.en_sorted)
        ls.finalize()
        ls.zero()
        return ls

    def oracle_continuity(self, noc=1.0, interp=1.0, gamma1=1.0, pin=False):
        ls = self.oracle_system()
        ls.assemble_continuity(self.en_sorted, self.area_sorted, self.xyz, self.dpdx, self.vel, self.p,
                               self.rho, self.dt, gamma1, noc, interp)
        if pin:
            ls.reset_rows([0])
            ls.sum_into([0], [0.0 - self.p[0]], [1.0])
        return ls

    def oracle_scalar(self, mass=False, **kw):
        ls = self.oracle_system()
        ls.assemble_scalar(self.en_sorted, self.area_sorted, self.mdot_sorted, self.xyz, self.dqdx,
                           self.vel, self.q, self.rho, self.gamma, self.o.scalar_opts(**kw))
        if mass:
            ls.assemble_scalar_mass_be(0.9 * self.q, self.q, self.rho, self.rho, self.vol, self.dt)
        return ls


def _equal_system(g, ls):
    rp, ci, v = ls.owned_csr()
    grp, gci = g.graph()
    assert np.array_equal(grp, rp) and np.array_equal(gci, ci)
    assert np.array_equal(g.matrix_values(), v)
    assert np.array_equal(g.rhs(), ls.owned_rhs())


# (dims, perturbation, minimum share of rows that must take the computed-tag branch)
SLOT_CASES = [
    ((20, 20, 20), 0.125, 0.0),   # 12.8 x-lines per slice: tags are loaded (flagged slices)
    ((256, 5, 4), 0.0, 0.9),      # one x-line per slice, as at 256^3: tags computed
    ((256, 5, 4), 0.125, 0.9),    # ... on a distorted mesh (non-orthogonal correction active)
    ((512, 3, 3), 0.125, 0.9),    # two slices per x-line (the 512^3 geometry)
    ((300, 4, 3), 0.0, 0.0),      # x-lines straddle slices: mixed flagged / computed rows
    ((256, 6, 40), 0.125, 0.9),   # long z-runs for the marching kernels: 38 interior planes = a run of 32 slices + one of 6
    ((512, 6, 40), 0.125, 0.9),   # ... with two slices per x-line: the node across the slice boundary is staged too
    ((256, 13, 7), 0.125, 0.9),   # 11 interior lines: groups of 4 + 4 + 3 (or 8 + 3) adjacent lines
    ((256, 7, 5), 0.125, 0.9),    # 5 interior lines: a group of four and ONE line left over (the kernel's single-run form)
    ((256, 3, 6), 0.125, 0.9),    # 1 interior line per plane: no groups at all
    ((256, 8, 6), 0.125, 0.9),    # 6 interior lines: a group of four and a group of TWO (each wave an end of its group)
]


@pytest.mark.parametrize("variant", ["default", "row", "NG_ASM_BURST", "NG_ASM_MARCH", "lag_off", "lag_nopf", "lag_groups", "lag_pf", "lag_pf_groups8", "lag_pf_noyshare"])
@pytest.mark.parametrize("dims,perturb,min_computed", SLOT_CASES)
def test_slot_major_assembly_bit_exact(G, torch, oracle, monkeypatch, dims, perturb, min_computed, variant):
    """default: the automatic choice (the marching kernel where at least half of the slices are in
    pattern runs, else the row kernel); row: the plain row kernel; NG_ASM_MARCH: marching + burst
    kernels forced.  variant NG_ASM_BURST: the opt-in burst row kernel (a row's loads issued together, edge lists
    from the per-slice descriptor; the two boundary rows of an x-line are a wave's one irregular
    row, x-lines that straddle slices give waves with several) -- same bits."""
    if variant == "row":  # the plain row kernel, which the automatic choice replaces on lattices
        monkeypatch.setenv("NG_ASM_MARCH", "0")
    elif variant == "lag_off":  # round 2's marching kernel (LDS + one barrier per step) instead of the lagged one
        monkeypatch.setenv("NG_ASM_LAG", "0")
    elif variant == "lag_nopf":  # round 3's first lagged kernel: no LDS prefetch, single runs
        monkeypatch.setenv("NG_ASM_LAG_PF", "0")
        monkeypatch.setenv("NG_ASM_LAG_GROUPS", "0")
    elif variant == "lag_groups":  # ... four runs of adjacent lines per workgroup, their rows shared through LDS
        monkeypatch.setenv("NG_ASM_LAG_PF", "0")
    elif variant == "lag_pf":  # a step's streamed operands prefetched into LDS one step ahead, single runs
        monkeypatch.setenv("NG_ASM_LAG_GROUPS", "0")
    elif variant == "lag_pf_groups8":  # the default kernel on groups of eight lines (512-thread workgroups) instead of four
        monkeypatch.setenv("NG_ASM_LAG_GY", "8")
    elif variant == "lag_pf_noyshare":  # every wave evaluates its own slot 4 (the default hands the shares of an edge
        monkeypatch.setenv("NG_ASM_LAG_YSHARE", "0")  # between two lines of a group down through LDS: one evaluation)
    elif variant != "default":
        monkeypatch.setenv(variant, "1")
    c = BenchCase(G, oracle, dims, perturb)
    d = c.dev(torch)
    g = c.gpu_system(G)
    li = g.layout_info()
    n = c.S.n_owned
    # what makes the marching kernel the automatic choice
    marching = li["march_slices"] * 2 >= li["n_slices"] and li["march_slices"] >= 4 * li["march_runs"]
    # (5: the lagged, barrier-free marching kernel -- every run of these lattices qualifies for it; 6: its
    # prefetching form, the default)
    lag, lagpf = (5, 6) if li["march_runs"] > 0 else (4, 4)
    assert li["assembly_variant"] == {"default": lagpf if marching else 0, "row": 0, "NG_ASM_BURST": 3,
                                      "NG_ASM_MARCH": lagpf, "lag_off": 4 if marching else 0,
                                      "lag_nopf": lag if marching else 0, "lag_groups": lag if marching else 0,
                                      "lag_pf": lagpf if marching else 0,
                                      "lag_pf_groups8": lagpf if marching else 0,
                                      "lag_pf_noyshare": lagpf if marching else 0}[variant], li
    if dims == (256, 6, 40):  # 4 interior lines x 38 interior planes: runs of 32 + 6 slices
        assert marching and li["march_longest"] == 32 and li["march_slices"] == 152 and li["march_runs"] == 8, li
    if dims == (512, 6, 40):  # ... two slices per line
        assert marching and li["march_longest"] == 32 and li["march_slices"] == 304 and li["march_runs"] == 16, li
    assert li["rows_computed_tags"] >= min_computed * n, li
    assert li["rows_computed_columns"] >= min_computed * n, li
    if min_computed > 0:  # and the boundary rows of the x-lines are the flagged exceptions
        assert li["rows_computed_tags"] < n
    for noc, interp, gamma1 in ((1.0, 1.0, 1.0), (0.0, 0.0, 1.5)):
        g.zeroSystem()
        g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt,
                                    gamma1, noc, interp)
        g.loadComplete()
        _equal_system(g, c.oracle_continuity(noc, interp, gamma1))
    for opts in (dict(hybrid_factor=1.0), dict(hybrid_factor=1.0, limiter=True, alpha=0.3, alpha_upw=0.8,
                                               upw_factor=0.9), dict()):
        g.zeroSystem()
        g.assemble_scalar_edges(d["coords"], d["dqdx"], d["vel"], d["q"], d["rho"], d["gamma"], d["area"],
                                d["mdot"], **opts)
        g.loadComplete()
        _equal_system(g, c.oracle_scalar(**opts))
    # a second assembly on top of the first one (the kernels' accumulate path): exactly twice the values
    g.zeroSystem()
    for _ in range(2):
        g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt)
    g.loadComplete()
    ls = c.oracle_continuity()
    assert np.array_equal(g.matrix_values(), 2.0 * ls.owned_csr()[2]) and np.array_equal(g.rhs(), 2.0 * ls.owned_rhs())
    # the adjacent edge algorithms on slot-major edges
    mdot = torch.empty(c.S.n_edges, dtype=torch.float64, device="cuda:0")
    g.compute_mdot_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt, mdot)
    grad = torch.empty(3, c.S.n_local, dtype=torch.float64, device="cuda:0")
    g.nodal_grad_edges(d["area"], d["vol"], d["q"], grad)
    g.synchronize()
    assert np.array_equal(mdot.cpu().numpy(), c.mdot)
    assert np.array_equal(grad.cpu().numpy().T,
                          oracle.nodal_grad(c.en_sorted, c.area_sorted, c.vol, c.q))


@pytest.mark.parametrize("dims,perturb", [((256, 6, 20), 0.125), ((512, 5, 9), 0.0), ((20, 20, 20), 0.125)])
def test_suggested_edge_order_reaches_the_fast_paths(G, torch, oracle, dims, perturb):
    """nalu_gpu_suggest_edge_order on an arbitrarily ((L,R)-sorted) numbered edge list: the general
    "j-th upper edge of a node" rule -- no knowledge of the lattice -- gives computed edge ids and
    marching runs like BoxSlab's own direction-major numbering, and the oracle's values bit for bit."""
    c = BenchCase(G, oracle, dims, perturb, edge_order="suggested")
    ref = BenchCase(G, oracle, dims, perturb, edge_order="slot")
    d = c.dev(torch)
    g = c.gpu_system(G)
    li, lr = g.layout_info(), ref.gpu_system(G).layout_info()
    assert li["assembly_variant"] == lr["assembly_variant"] and (dims != (256, 6, 20) or lr["assembly_variant"] == 6)
    if dims[0] % 256 == 0:  # one x-line (or half of one) per slice: the same fast paths, row for row
        for k in ("rows_computed_tags", "rows_computed_columns", "march_slices", "march_runs"):
            assert li[k] == lr[k], (k, li, lr)
    g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt)
    g.loadComplete()
    _equal_system(g, c.oracle_continuity())
    g.zeroSystem()
    g.assemble_scalar_edges(d["coords"], d["dqdx"], d["vel"], d["q"], d["rho"], d["gamma"], d["area"], d["mdot"],
                            hybrid_factor=1.0, limiter=True)
    g.loadComplete()
    _equal_system(g, c.oracle_scalar(hybrid_factor=1.0, limiter=True))


def test_sorted_and_slot_major_numbering_agree(G, torch, oracle):
    """The same mesh with (L,R)-sorted and with slot-major edge ids: identical device results."""
    res = []
    for order in ("sorted", "slot"):
        c = BenchCase(G, oracle, (256, 4, 3), 0.125, edge_order=order)
        d = c.dev(torch)
        g = c.gpu_system(G)
        g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt)
        res.append((g.matrix_values(), g.rhs(), g.layout_info()["rows_computed_tags"]))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert res[0][2] == 0 and res[1][2] > 0  # only slot-major numbering gets computed tags


def _check_solve(g, ref, ls, kw, xout):
    assert g.linearSolveIterations == ref["iters"]
    assert len(g.history) == len(ref["history"])
    assert np.abs(g.history - ref["history"]).max() <= TOL_SOLVE
    assert rel_err(xout.cpu().numpy()[:len(ref["x"])], ref["x"]) <= TOL_SOLVE
    assert abs(g.linearResidual - ref["final_resid"]) <= TOL_SOLVE * max(1.0, np.linalg.norm(ls.owned_rhs()))


def _bench_step(G, torch, c, d, cfg, xout):
    """one bench.py step: the continuity equation system through EquationSystem::assemble_and_solve"""
    from nalu_amd import eqsys
    S = c.S
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(S.n_owned, gids=S.gids)
    eq = eqsys.EquationSystem(g, cfg, "ContinuityEQS")
    eq.solverAlgDriver.solverAlgMap["INTERIOR"] = eqsys.AssembleContinuityEdgeSolverAlgorithm(
        S.edge_nodes, d, dt=c.dt, gamma1=1.0, noc=1.0, interp_together=1.0)
    eq.solverAlgDriver.solverAlgMap["REF_PRESSURE"] = eqsys.FixPressureAtNodeAlgorithm(0, 0.0, float(c.p[0]))
    eq.initialize()
    assert eq.assemble_and_solve(xout) == 0
    return g


@pytest.mark.parametrize("dims,perturb", [((20, 20, 20), 0.125), ((64, 64, 64), 0.0)])
@pytest.mark.parametrize("kw", [dict(tolerance=1e-5, max_iterations=50),      # the regression's limits = bench.py
                                dict(tolerance=1e-10, max_iterations=600)])   # the residual-history criterion
def test_pinned_poisson_jacobi_cg(G, torch, oracle, dims, perturb, kw):
    """The benchmarked solve: continuity assembly (slot-major edges) + FixPressureAtNode on node 0 +
    Jacobi-CG, through EquationSystem::assemble_and_solve; 64^3 is BASELINE config 2 at full size."""
    c = BenchCase(G, oracle, dims, perturb)
    d = c.dev(torch)
    cfg = G.LinearSolverConfig(method="cg", preconditioner="jacobi", **kw)
    xout = torch.zeros(c.S.n_local, dtype=torch.float64, device="cuda:0")
    g = _bench_step(G, torch, c, d, cfg, xout)
    ls = c.oracle_continuity(pin=True)
    _equal_system(g, ls)
    ref = ls.solve("cg", "jacobi", **kw)
    _check_solve(g, ref, ls, kw, xout)
    if kw["max_iterations"] == 50 and max(dims) >= 64:
        assert ref["iters"] == 50  # the iteration cap binds, as in the benchmark


@pytest.mark.parametrize("dims,perturb", [((256, 6, 5), 0.0), ((256, 6, 5), 0.125), ((512, 4, 3), 0.0)])
def test_one_line_slices_spmv_and_cg(G, torch, oracle, dims, perturb):
    """nx = 256: every slice is exactly one x-line (the 256^3 geometry), so SpMV computes its
    column indices and the fused p.Ap partials are per x-line.  SpMV bit for bit; Jacobi-CG on the
    SPD scalar diffusion + mass system (the thin pinned-Poisson box diverges under CG, in the oracle
    too, and a diverging recurrence amplifies rounding: not a parity case)."""
    c = BenchCase(G, oracle, dims, perturb)
    d = c.dev(torch)
    g = c.gpu_system(G, extra_node_graph=True)
    assert g.layout_info()["rows_computed_columns"] >= 0.9 * c.S.n_owned
    g.assemble_scalar_edges(d["coords"], d["dqdx"], d["vel"], d["q"], d["rho"], d["gamma"], d["area"], d["mdot"])
    g.assemble_scalar_mass_be(0.9 * d["q"], d["q"], d["rho"], d["rho"], d["vol"], c.dt)
    g.loadComplete()
    ls = c.oracle_scalar(mass=True)
    _equal_system(g, ls)
    x = np.random.RandomState(9).uniform(-1, 1, c.S.n_local)
    y = torch.empty(c.S.n_owned, dtype=torch.float64, device="cuda:0")
    g.spmv(torch.as_tensor(x, device="cuda:0"), y)
    g.synchronize()
    assert np.array_equal(y.cpu().numpy(), ls.spmv(x))
    for kw in (dict(tolerance=1e-5, max_iterations=50), dict(tolerance=1e-10, max_iterations=200)):
        xout = torch.zeros(c.S.n_local, dtype=torch.float64, device="cuda:0")
        g.solve(G.LinearSolverConfig(method="cg", preconditioner="jacobi", **kw), xout)
        ref = ls.solve("cg", "jacobi", **kw)
        _check_solve(g, ref, ls, kw, xout)


@pytest.mark.parametrize("dims,perturb", [((20, 20, 20), 0.125), ((256, 5, 4), 0.125), ((40, 40, 40), 0.0)])
def test_scalar_upwind_gmres_step(G, torch, oracle, dims, perturb):
    """BASELINE config 4 in small (bench.py --workload scalar): hybrid_factor 1, alpha_upw 1,
    upw_factor 1, van Leer limiter, backward-Euler mass term, GMRES(50)/Jacobi -- slot-major edges."""
    from nalu_amd import eqsys
    c = BenchCase(G, oracle, dims, perturb)
    d = c.dev(torch)
    S = c.S
    opts = dict(hybrid_factor=1.0, alpha_upw=1.0, upw_factor=1.0, limiter=True)
    for kw in (dict(tolerance=1e-5, max_iterations=50, kspace=50), dict(tolerance=1e-10, max_iterations=120, kspace=50)):
        cfg = G.LinearSolverConfig(method="gmres", preconditioner="jacobi", **kw)
        g = G.GpuLinearSystem()
        g.beginLinearSystemConstruction(S.n_owned, gids=S.gids)
        eq = eqsys.EquationSystem(g, cfg, "MixtureFractionEQS")
        eq.solverAlgDriver.solverAlgMap["INTERIOR"] = eqsys.AssembleScalarEdgeSolverAlgorithm(
            S.edge_nodes, d, **opts)
        eq.solverAlgDriver.solverAlgMap["MASS"] = eqsys.ScalarMassBackwardEulerAlgorithm(
            np.arange(S.n_owned, dtype=np.int32), 0.9 * d["q"], d["q"], d["rho"], d["rho"], d["vol"], c.dt)
        eq.initialize()
        xout = torch.zeros(S.n_local, dtype=torch.float64, device="cuda:0")
        assert eq.assemble_and_solve(xout) == 0
        ls = c.oracle_scalar(mass=True, **opts)
        _equal_system(g, ls)
        ref = ls.solve("gmres", "jacobi", **kw)
        _check_solve(g, ref, ls, kw, xout)


def test_config2_scale_vs_oracle_96(G, torch, oracle):
    """96^3 (885k nodes, 2.6 M edges; above BASELINE config 2's 64^3): the oracle does this in
    seconds, so the comparison is direct -- matrix / rhs bit for bit, the three scatter variants to
    1e-12, SpMV bit for bit, 50 Jacobi-CG iterations to 1e-10."""
    c = BenchCase(G, oracle, (96, 96, 96))
    d = c.dev(torch)
    g = c.gpu_system(G)
    ls = c.oracle_continuity()
    li = g.layout_info()
    res = {}
    for mode in ("rows", "colored", "atomic"):
        g.zeroSystem()
        g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt, mode=mode)
        g.loadComplete()
        res[mode] = (g.matrix_values(), g.rhs())
    rp, ci, v = ls.owned_csr()
    assert np.array_equal(res["rows"][0], v) and np.array_equal(res["rows"][1], ls.owned_rhs())
    for mode in ("colored", "atomic"):
        assert rel_err(res[mode][0], v) <= 1e-12 and rel_err(res[mode][1], ls.owned_rhs()) <= 1e-12
    g.zeroSystem()
    g.assemble_continuity_edges(d["coords"], d["dpdx"], d["vel"], d["p"], d["rho"], d["area"], c.dt)
    x = np.random.RandomState(3).uniform(-1, 1, c.S.n_local)
    y = torch.empty(c.S.n_owned, dtype=torch.float64, device="cuda:0")
    g.spmv(torch.as_tensor(x, device="cuda:0"), y)
    g.synchronize()
    assert np.array_equal(y.cpu().numpy(), ls.spmv(x))
    g.resetRows([0])
    g.sumInto([0], [0.0 - c.p[0]], [1.0])
    ls.reset_rows([0])
    ls.sum_into([0], [0.0 - c.p[0]], [1.0])
    kw = dict(tolerance=1e-5, max_iterations=50)
    xout = torch.zeros(c.S.n_local, dtype=torch.float64, device="cuda:0")
    g.solve(G.LinearSolverConfig(method="cg", preconditioner="jacobi", **kw), xout)
    ref = ls.solve("cg", "jacobi", **kw)
    assert ref["iters"] == 50
    _check_solve(g, ref, ls, kw, xout)
    assert li["rows_computed_columns"] > 0


@pytest.mark.timeout(900)
def test_baseline_configs_3_and_4_full_size_against_the_oracle(G, torch, oracle):
    """BASELINE configs 3 and 4 at their own size, 256^3 nodes (16.8 M nodes, 50.1 M edges), against the ORACLE
    (until now only GPU np = 2 against GPU np = 1 at this size).  The oracle runs its
    OpenMP flavour (same source, same per-row summation order: its assembled system equals the serial one bit
    for bit).  Its dot products are summed in the fixed reduction tree (`dot_tree`): over 16.8 M entries a
    sequential fp64 sum carries ~1e-12 of rounding error of its own (and the OpenMP reduction's partition
    changes with the thread count), which 15 unconverged GMRES steps amplify to 6e-10 in the solution -- the
    histories agree to 1e-10 either way.
    Config 3: continuity assembly + FixPressureAtNode + Jacobi-CG with the regression's limits (50 iterations)
    through EquationSystem::assemble_and_solve -- matrix and rhs bit for bit, iterations equal, history and
    solution <= 1e-10.  Config 4: scalar transport (upwind, van Leer limiter) + backward-Euler mass term,
    matrix and rhs bit for bit, then GMRES(50)/Jacobi; the oracle's ICGS costs ~2.5 s per iteration at this
    size, so the solve is compared over its first 15 iterations (the 50-iteration comparison is made at the
    smaller sizes above)."""
    from nalu_amd import eqsys
    c = BenchCase(G, oracle, (256, 256, 256))
    d = c.dev(torch)
    S = c.S
    # ---- config 3
    cfg = G.LinearSolverConfig(method="cg", preconditioner="jacobi", tolerance=1e-5, max_iterations=50)
    xout = torch.zeros(S.n_local, dtype=torch.float64, device="cuda:0")
    g = _bench_step(G, torch, c, d, cfg, xout)
    assert g.layout_info()["assembly_variant"] == 6  # the kernel bench.py times
    ls = oracle.LinSys(1, S.n_owned, fast="omp")
    ls.add_connections(c.en_sorted)
    ls.finalize()
    ls.prepare_edge_buckets(c.en_sorted)
    ls.zero()
    ls.assemble_continuity(c.en_sorted, c.area_sorted, c.xyz, c.dpdx, c.vel, c.p, c.rho, c.dt, 1.0, 1.0, 1.0)
    ls.reset_rows([0])
    ls.sum_into([0], [0.0 - c.p[0]], [1.0])
    _equal_system(g, ls)
    kw = dict(tolerance=1e-5, max_iterations=50)
    ref = ls.solve("cg", "jacobi", dot_tree=True, **kw)
    assert ref["iters"] == 50
    _check_solve(g, ref, ls, kw, xout)
    g.close()
    del ls, g
    # ---- config 4
    opts = dict(hybrid_factor=1.0, alpha_upw=1.0, upw_factor=1.0, limiter=True)
    kw = dict(tolerance=1e-5, max_iterations=15, kspace=50)
    g = G.GpuLinearSystem()
    g.beginLinearSystemConstruction(S.n_owned, gids=S.gids)
    eq = eqsys.EquationSystem(g, G.LinearSolverConfig(method="gmres", preconditioner="jacobi", **kw), "MixtureFractionEQS")
    eq.solverAlgDriver.solverAlgMap["INTERIOR"] = eqsys.AssembleScalarEdgeSolverAlgorithm(S.edge_nodes, d, **opts)
    eq.solverAlgDriver.solverAlgMap["MASS"] = eqsys.ScalarMassBackwardEulerAlgorithm(
        np.arange(S.n_owned, dtype=np.int32), 0.9 * d["q"], d["q"], d["rho"], d["rho"], d["vol"], c.dt)
    eq.initialize()
    xout.zero_()
    assert eq.assemble_and_solve(xout) == 0
    ls = oracle.LinSys(1, S.n_owned, fast="omp")
    ls.add_connections(c.en_sorted)
    ls.add_connections(np.arange(S.n_owned, dtype=np.int32).reshape(-1, 1))
    ls.finalize()
    ls.prepare_edge_buckets(c.en_sorted)
    ls.zero()
    ls.assemble_scalar(c.en_sorted, c.area_sorted, c.mdot_sorted, c.xyz, c.dqdx, c.vel, c.q, c.rho, c.gamma,
                       oracle.scalar_opts(**opts))
    ls.assemble_scalar_mass_be(0.9 * c.q, c.q, c.rho, c.rho, c.vol, c.dt)
    _equal_system(g, ls)
    ref = ls.solve("gmres", "jacobi", dot_tree=True, **kw)
    assert ref["iters"] == 15
    _check_solve(g, ref, ls, kw, xout)


@pytest.mark.timeout(900)
def test_baseline_config_5_per_gpu_slab_against_the_oracle(G, torch, oracle):
    """BASELINE config 5's per-GPU share -- the 512 x 512 x 64-node slab of the weak-scaling run (16.8 M nodes,
    two slices per x-line) -- as a box of its own against the ORACLE: continuity assembly + pinned node bit for
    bit, 50 Jacobi-CG iterations to 1e-10.  (The full 512^3 box: eight ranks against one rank on the GPU,
    tests/test_distributed.py; the oracle would need ~8 x the memory and time of this test for it.)"""
    c = BenchCase(G, oracle, (512, 512, 64))
    d = c.dev(torch)
    S = c.S
    kw = dict(tolerance=1e-5, max_iterations=50)
    xout = torch.zeros(S.n_local, dtype=torch.float64, device="cuda:0")
    g = _bench_step(G, torch, c, d, G.LinearSolverConfig(method="cg", preconditioner="jacobi", **kw), xout)
    li = g.layout_info()
    assert li["assembly_variant"] == 6 and li["march_slices"] > 0.9 * li["n_slices"], li
    ls = oracle.LinSys(1, S.n_owned, fast="omp")
    ls.add_connections(c.en_sorted)
    ls.finalize()
    ls.prepare_edge_buckets(c.en_sorted)
    ls.zero()
    ls.assemble_continuity(c.en_sorted, c.area_sorted, c.xyz, c.dpdx, c.vel, c.p, c.rho, c.dt, 1.0, 1.0, 1.0)
    ls.reset_rows([0])
    ls.sum_into([0], [0.0 - c.p[0]], [1.0])
    _equal_system(g, ls)
    ref = ls.solve("cg", "jacobi", dot_tree=True, **kw)
    assert ref["iters"] == 50
    _check_solve(g, ref, ls, kw, xout)
