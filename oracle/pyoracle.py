"""ctypes front end of the CPU oracle (oracle/nalu_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under nalu_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


# flavours: False = the checker (-O2), True = -O3 -march=native (1-core CPU baseline),
# "omp" = the same with OpenMP (multi-core CPU baseline; never the checker)
_NAMES = {False: "liboracle.so", True: "liboracle_fast.so", "omp": "liboracle_omp.so"}


def build(fast=False):
    """Compile the oracle with gcc (make).  Building the checker is not using it."""
    target = _NAMES[fast]
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


def _load(fast=False):
    name = _NAMES[fast]
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build(fast)
    return C.CDLL(path)


_dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


class _Graph(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("rowPointers", C.POINTER(C.c_size_t)),
                ("colIndices", C.POINTER(C.c_int))]


class _LinSys(C.Structure):
    _fields_ = [("numDof", C.c_int), ("nOwnedNodes", C.c_int), ("nSharedNodes", C.c_int),
                ("maxOwnedRowId", C.c_int), ("maxSharedNotOwnedRowId", C.c_int),
                ("ownedGraph", C.POINTER(_Graph)), ("sharedGraph", C.POINTER(_Graph)),
                ("ownedVals", C.POINTER(C.c_double)), ("sharedVals", C.POINTER(C.c_double)),
                ("ownedRhs", C.POINTER(C.c_double)), ("sharedRhs", C.POINTER(C.c_double)),
                ("sln", C.POINTER(C.c_double)),
                ("conn", C.c_void_p), ("connLen", C.c_void_p), ("connCap", C.c_void_p)]


class ScalarOpts(C.Structure):
    _fields_ = [("alpha", C.c_double), ("alphaUpw", C.c_double), ("hoUpwind", C.c_double),
                ("useLimiter", C.c_int), ("pecletKind", C.c_int), ("pecA", C.c_double),
                ("pecHf", C.c_double), ("pecC1", C.c_double), ("pecC2", C.c_double)]


class SolverConfig(C.Structure):
    _fields_ = [("method", C.c_int), ("precond", C.c_int), ("tolerance", C.c_double),
                ("max_iterations", C.c_int), ("kspace", C.c_int)]


METHODS = {"cg": 0, "gmres": 1, "biCgStab": 2, "bicgstab": 2, "tfqmr": 3}
PRECONDS = {"none": 0, "jacobi": 1, "default": 1, "sgs": 2, "chebyshev": 3, "mt_sgs": 4}


def _bind(lib):
    lib.orc_box_coords.argtypes = [C.c_int] * 3 + [C.c_double] * 3 + [_dp]
    lib.orc_box_perturb.argtypes = [C.c_int] * 3 + [_dp, C.c_uint32, C.c_double]
    lib.orc_box_hexes.argtypes = [C.c_int] * 3 + [_ip]
    lib.orc_box_edges.argtypes = [C.c_int] * 3 + [_ip]
    lib.orc_box_edges.restype = C.c_int64
    lib.orc_quad_area_by_triangle_facets.argtypes = [_dp, _dp]
    lib.orc_hex_volume_by_triangle_facets.argtypes = [_dp]
    lib.orc_hex_volume_by_triangle_facets.restype = C.c_double
    lib.orc_hex_scs_det.argtypes = [_dp, _dp]
    lib.orc_hex_scv_det.argtypes = [_dp, _dp]
    lib.orc_compute_geometry.argtypes = [C.c_int64, _ip, _dp, C.c_int64, C.c_int64, _ip, _dp, _dp]
    lib.orc_compute_geometry.restype = C.c_int
    lib.orc_graph_create.argtypes = [C.c_int64, np.ctypeslib.ndpointer(np.uintp)]
    lib.orc_graph_create.restype = C.POINTER(_Graph)
    lib.orc_graph_destroy.argtypes = [C.POINTER(_Graph)]
    lib.orc_graph_row_length.argtypes = [C.POINTER(_Graph), C.c_int64]
    lib.orc_graph_row_length.restype = C.c_size_t
    lib.orc_graph_insert_indices.argtypes = [C.POINTER(_Graph), C.c_int64, C.c_size_t, _ip, C.c_int]
    lib.orc_graph_remove_invalid.argtypes = [C.POINTER(_Graph)]
    lib.orc_linsys_create.argtypes = [C.c_int] * 3
    lib.orc_linsys_create.restype = C.POINTER(_LinSys)
    lib.orc_linsys_destroy.argtypes = [C.POINTER(_LinSys)]
    lib.orc_set_gids.argtypes = [C.c_void_p]
    lib.orc_add_connections.argtypes = [C.POINTER(_LinSys), C.c_int64, C.c_int, _ip]
    lib.orc_finalize.argtypes = [C.POINTER(_LinSys)]
    lib.orc_zero_system.argtypes = [C.POINTER(_LinSys)]
    lib.orc_sum_into.argtypes = [C.POINTER(_LinSys), C.c_int, _ip, _dp, _dp]
    lib.orc_apply_dirichlet.argtypes = [C.POINTER(_LinSys), C.c_int64, _ip, _dp, _dp, C.c_uint, C.c_uint]
    lib.orc_reset_rows.argtypes = [C.POINTER(_LinSys), C.c_int64, _ip, C.c_uint, C.c_uint]
    lib.orc_peclet_classic.argtypes = [C.c_double] * 3
    lib.orc_peclet_classic.restype = C.c_double
    lib.orc_peclet_tanh.argtypes = [C.c_double] * 3
    lib.orc_peclet_tanh.restype = C.c_double
    lib.orc_van_leer.argtypes = [C.c_double] * 3
    lib.orc_van_leer.restype = C.c_double
    cont = [C.c_int64, _ip, _dp, _dp, _dp, _dp, _dp, _dp] + [C.c_double] * 4
    lib.orc_assemble_continuity_edges.argtypes = [C.POINTER(_LinSys)] + cont
    lib.orc_omp_prepare_edge_buckets.argtypes = [C.POINTER(_LinSys), C.c_int64, _ip]
    lib.orc_continuity_edge_coeffs.argtypes = cont + [_dp, _dp]
    lib.orc_compute_mdot_edges.argtypes = cont + [_dp]
    scal = [C.c_int64, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.POINTER(ScalarOpts)]
    lib.orc_assemble_scalar_edges.argtypes = [C.POINTER(_LinSys)] + scal
    lib.orc_scalar_edge_coeffs.argtypes = scal + [_dp, _dp]
    lib.orc_nodal_grad_edges.argtypes = [C.c_int64, _ip, _dp, _dp, _dp, _dp]
    lib.orc_assemble_scalar_mass_be.argtypes = [C.POINTER(_LinSys), C.c_int64, _dp, _dp, _dp, _dp, _dp, C.c_double]
    lib.orc_field_axpby.argtypes = [C.c_int64, C.c_double, _dp, C.c_double, _dp]
    lib.orc_project_nodal_velocity.argtypes = [C.c_int64, C.c_double, C.c_double, _dp, _dp, _dp,
                                               C.c_void_p, _dp]
    lib.orc_update_and_clip.argtypes = [C.c_int64, _dp, _dp, _dp, C.c_double, _dp]
    lib.orc_spmv.argtypes = [C.POINTER(_LinSys), _dp, _dp]
    lib.orc_solve.argtypes = [C.POINTER(_LinSys), C.POINTER(SolverConfig), C.POINTER(C.c_double),
                              _dp, C.c_int, C.POINTER(C.c_int)]
    lib.orc_solve.restype = C.c_int
    lib.orc_set_dot_tree.argtypes = [C.c_int]
    lib.orc_set_dot_tree.restype = None
    lib.orc_dot.argtypes = [_dp, _dp, C.c_int64]
    lib.orc_dot.restype = C.c_double
    lib.orc_norm2.argtypes = [_dp, C.c_int64]
    lib.orc_greedy_colors.argtypes = [C.POINTER(_LinSys), _ip]
    lib.orc_greedy_colors.restype = C.c_int
    lib.orc_norm2.restype = C.c_double
    return lib


_LIBS = {}


def lib(fast=False):
    if fast not in _LIBS:
        _LIBS[fast] = _bind(_load(fast))
    return _LIBS[fast]


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------- mesh
class BoxMesh:
    """STK `generated:` style Hex8 box with n nodes per side (nxn, nyn, nzn)."""

    def __init__(self, nxn, nyn=None, nzn=None, lengths=(1.0, 1.0, 1.0), perturb=0.0, seed=1,
                 fast=False):
        nyn = nxn if nyn is None else nyn
        nzn = nxn if nzn is None else nzn
        L = lib(fast)
        self.dims = (nxn, nyn, nzn)
        self.nnodes = nxn * nyn * nzn
        self.xyz = np.zeros((self.nnodes, 3))
        L.orc_box_coords(nxn, nyn, nzn, *[float(v) for v in lengths], self.xyz)
        if perturb:
            h = min(lengths[0] / max(nxn - 1, 1), lengths[1] / max(nyn - 1, 1),
                    lengths[2] / max(nzn - 1, 1))
            L.orc_box_perturb(nxn, nyn, nzn, self.xyz, seed, perturb * h)
        self.nelem = (nxn - 1) * (nyn - 1) * (nzn - 1)
        self.hex_nodes = np.zeros((max(self.nelem, 1), 8), np.int32)
        L.orc_box_hexes(nxn, nyn, nzn, self.hex_nodes)
        self.hex_nodes = self.hex_nodes[:self.nelem]
        ne = (nxn - 1) * nyn * nzn + nxn * (nyn - 1) * nzn + nxn * nyn * (nzn - 1)
        self.edge_nodes = np.zeros((max(ne, 1), 2), np.int32)
        got = L.orc_box_edges(nxn, nyn, nzn, self.edge_nodes)
        assert got == ne
        self.edge_nodes = self.edge_nodes[:ne]
        self.nedges = ne
        self.gids = np.arange(1, self.nnodes + 1, dtype=np.int64)

    def geometry(self, fast=False):
        vol = np.zeros(self.nnodes)
        area = np.zeros((self.nedges, 3))
        rc = lib(fast).orc_compute_geometry(self.nelem, self.hex_nodes, self.xyz, self.nnodes,
                                            self.nedges, self.edge_nodes, vol, area)
        assert rc == 0
        return vol, area


# ------------------------------------------------------------ linear system
class LinSys:
    """Single-rank (or one rank's local) Tpetra-like linear system of the oracle."""

    def __init__(self, num_dof, n_owned, n_shared=0, fast=False):
        self.L = lib(fast)
        self.p = self.L.orc_linsys_create(num_dof, n_owned, n_shared)
        self.num_dof, self.n_owned, self.n_shared = num_dof, n_owned, n_shared

    def __del__(self):
        if getattr(self, "p", None):
            self.L.orc_linsys_destroy(self.p)
            self.p = None

    def add_connections(self, ent_nodes, gids=None):
        ent_nodes = i32(ent_nodes)
        if gids is not None:
            g = np.ascontiguousarray(gids, np.int64)
            self.L.orc_set_gids(g.ctypes.data)
        self.L.orc_add_connections(self.p, ent_nodes.shape[0], ent_nodes.shape[1], ent_nodes)
        self.L.orc_set_gids(None)

    def finalize(self):
        self.L.orc_finalize(self.p)

    def zero(self):
        self.L.orc_zero_system(self.p)

    def sum_into(self, node_lids, rhs, lhs):
        node_lids = i32(node_lids)
        self.L.orc_sum_into(self.p, len(node_lids), node_lids, f64(rhs), f64(lhs))

    def apply_dirichlet(self, node_lids, solution, bc_values, begin=0, end=None):
        end = self.num_dof if end is None else end
        node_lids = i32(node_lids)
        self.L.orc_apply_dirichlet(self.p, len(node_lids), node_lids, f64(solution),
                                   f64(bc_values), begin, end)

    def reset_rows(self, node_lids, begin=0, end=None):
        end = self.num_dof if end is None else end
        node_lids = i32(node_lids)
        self.L.orc_reset_rows(self.p, len(node_lids), node_lids, begin, end)

    def _csr(self, graph, vals):
        g = graph.contents
        n = g.nrows
        rp = np.ctypeslib.as_array(g.rowPointers, shape=(n + 1,)).astype(np.int64)
        nnz = int(rp[-1])
        if nnz == 0:
            return rp, np.zeros(0, np.int32), np.zeros(0)
        ci = np.ctypeslib.as_array(g.colIndices, shape=(nnz,)).copy()
        v = np.ctypeslib.as_array(vals, shape=(nnz,)).copy()
        return rp, ci, v

    def owned_csr(self):
        return self._csr(self.p.contents.ownedGraph, self.p.contents.ownedVals)

    def shared_csr(self):
        return self._csr(self.p.contents.sharedGraph, self.p.contents.sharedVals)

    def owned_rhs(self):
        n = self.n_owned * self.num_dof
        return np.ctypeslib.as_array(self.p.contents.ownedRhs, shape=(max(n, 1),))[:n].copy()

    def shared_rhs(self):
        n = self.n_shared * self.num_dof
        return np.ctypeslib.as_array(self.p.contents.sharedRhs, shape=(max(n, 1),))[:n].copy()

    def solution(self):
        n = self.n_owned * self.num_dof
        return np.ctypeslib.as_array(self.p.contents.sln, shape=(max(n, 1),))[:n].copy()

    def prepare_edge_buckets(self, edge_nodes):
        """OpenMP flavour only: per-thread edge lists (set-up); edge_nodes must be the very array
        (same memory) that the assemble_* calls get"""
        assert edge_nodes.dtype == np.int32 and edge_nodes.flags["C_CONTIGUOUS"]
        self.L.orc_omp_prepare_edge_buckets(self.p, edge_nodes.shape[0], edge_nodes)

    def assemble_continuity(self, edge_nodes, area, xyz, dpdx, vel, p, rho, dt, gamma1=1.0,
                            noc=1.0, interp=1.0):
        edge_nodes = i32(edge_nodes)
        self.L.orc_assemble_continuity_edges(self.p, edge_nodes.shape[0], edge_nodes, f64(area),
                                             f64(xyz), f64(dpdx), f64(vel), f64(p), f64(rho),
                                             dt, gamma1, noc, interp)

    def assemble_scalar(self, edge_nodes, area, mdot, xyz, dqdx, vel, q, rho, gamma, opts):
        edge_nodes = i32(edge_nodes)
        self.L.orc_assemble_scalar_edges(self.p, edge_nodes.shape[0], edge_nodes, f64(area),
                                         f64(mdot), f64(xyz), f64(dqdx), f64(vel), f64(q),
                                         f64(rho), f64(gamma), C.byref(opts))

    def assemble_scalar_mass_be(self, qN, qNp1, rhoN, rhoNp1, vol, dt):
        self.L.orc_assemble_scalar_mass_be(self.p, len(qN), f64(qN), f64(qNp1), f64(rhoN),
                                           f64(rhoNp1), f64(vol), dt)

    def greedy_colors(self):
        """the row-order greedy colouring of the mt_sgs preconditioner: (colours, count)"""
        col = np.zeros(max(self.n_owned * self.num_dof, 1), np.int32)
        nc = self.L.orc_greedy_colors(self.p, col)
        return col[:self.n_owned * self.num_dof], nc

    def spmv(self, x):
        y = np.zeros(self.n_owned * self.num_dof)
        self.L.orc_spmv(self.p, f64(x), y)
        return y

    def solve(self, method="cg", precond="jacobi", tolerance=1e-4, max_iterations=50, kspace=50, dot_tree=False):
        """dot_tree: the dot products and norms summed in the device library's fixed reduction tree instead of
        sequentially (orc_set_dot_tree; what BiCGStab / TFQMR need for a 1e-10 comparison)"""
        cfg = SolverConfig(METHODS[method], PRECONDS[precond], tolerance, max_iterations, kspace)
        hist = np.zeros(max_iterations + 2)
        hl = C.c_int(0)
        fr = C.c_double(0.0)
        self.L.orc_set_dot_tree(1 if dot_tree else 0)
        try:
            iters = self.L.orc_solve(self.p, C.byref(cfg), C.byref(fr), hist, len(hist), C.byref(hl))
        finally:
            self.L.orc_set_dot_tree(0)
        return dict(iters=iters, final_resid=fr.value, history=hist[:hl.value].copy(),
                    x=self.solution())


def dot(a, b, tree=False):
    """a . b summed sequentially, or (tree) in the device library's fixed reduction tree"""
    a, b = f64(a), f64(b)
    L = lib()
    L.orc_set_dot_tree(1 if tree else 0)
    try:
        return L.orc_dot(a, b, a.size)
    finally:
        L.orc_set_dot_tree(0)


def scalar_opts(alpha=0.0, alpha_upw=1.0, upw_factor=1.0, limiter=False, peclet="classic",
                hybrid_factor=0.0, A=5.0, c1=2.0, c2=4.0):
    """Defaults are the reference's (src/SolutionOptions.C:32-51)."""
    return ScalarOpts(alpha, alpha_upw, upw_factor, int(limiter), 0 if peclet == "classic" else 1,
                      A, hybrid_factor, c1, c2)


def continuity_edge_coeffs(edge_nodes, area, xyz, dpdx, vel, p, rho, dt, gamma1=1.0, noc=1.0,
                           interp=1.0):
    edge_nodes = i32(edge_nodes)
    ne = edge_nodes.shape[0]
    lhs, rhs = np.zeros((ne, 4)), np.zeros((ne, 2))
    lib().orc_continuity_edge_coeffs(ne, edge_nodes, f64(area), f64(xyz), f64(dpdx), f64(vel),
                                     f64(p), f64(rho), dt, gamma1, noc, interp, lhs, rhs)
    return lhs, rhs


def scalar_edge_coeffs(edge_nodes, area, mdot, xyz, dqdx, vel, q, rho, gamma, opts):
    edge_nodes = i32(edge_nodes)
    ne = edge_nodes.shape[0]
    lhs, rhs = np.zeros((ne, 4)), np.zeros((ne, 2))
    lib().orc_scalar_edge_coeffs(ne, edge_nodes, f64(area), f64(mdot), f64(xyz), f64(dqdx),
                                 f64(vel), f64(q), f64(rho), f64(gamma), C.byref(opts), lhs, rhs)
    return lhs, rhs


def compute_mdot(edge_nodes, area, xyz, dpdx, vel, p, rho, dt, gamma1=1.0, noc=1.0, interp=1.0,
                 fast=False):
    edge_nodes = i32(edge_nodes)
    mdot = np.zeros(edge_nodes.shape[0])
    lib(fast).orc_compute_mdot_edges(edge_nodes.shape[0], edge_nodes, f64(area), f64(xyz),
                                     f64(dpdx), f64(vel), f64(p), f64(rho), dt, gamma1, noc,
                                     interp, mdot)
    return mdot


def nodal_grad(edge_nodes, area, vol, q, fast=False):
    edge_nodes = i32(edge_nodes)
    dqdx = np.zeros((len(q), 3))
    lib(fast).orc_nodal_grad_edges(edge_nodes.shape[0], edge_nodes, f64(area), f64(vol), f64(q),
                                   dqdx)
    return dqdx


def field_axpby(alpha, x, beta, y):
    y = f64(y).copy()
    lib().orc_field_axpby(y.size, alpha, f64(x), beta, y)
    return y


def project_nodal_velocity(dt, gamma1, dpdx_new, dpdx_old, rho, vel, projected=None):
    """AoS (n, 3) fields; returns the projected velocity"""
    u = f64(vel).copy()
    m = None if projected is None else np.ascontiguousarray(projected, dtype=np.uint8)
    lib().orc_project_nodal_velocity(len(rho), dt, gamma1, f64(dpdx_new), f64(dpdx_old), f64(rho),
                                     None if m is None else m.ctypes.data, u)
    return u


def update_and_clip(z, dz, delta_clip=0.0):
    """returns (z_new, z_unfiltered, [n_clip_low, n_clip_high, min, max])"""
    zn, uf, info = f64(z).copy(), np.zeros(len(z)), np.zeros(4)
    lib().orc_update_and_clip(len(zn), f64(dz), zn, uf, delta_clip, info)
    return zn, uf, info


def set_values(ls, vals):
    """overwrite the owned CSR values (test set-up: same numbers on both sides)"""
    nnz = len(ls.owned_csr()[2])
    v = f64(vals)
    assert len(v) == nnz
    C.memmove(ls.p.contents.ownedVals, v.ctypes.data, 8 * nnz)


def set_rhs(ls, b):
    b = f64(b)
    assert len(b) == ls.n_owned * ls.num_dof
    C.memmove(ls.p.contents.ownedRhs, b.ctypes.data, 8 * len(b))


# --------------------------------------------- reference F77 kernels (_ref)
def ref_geom():
    """The reference's own hex_scs_det / hex_scv_det, compiled by oracle/Makefile from
    /root/reference (this container only).  Returns None when the .so is absent."""
    path = os.path.join(_HERE, "_ref", "libnalu_ref_geom.so")
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    ip = C.POINTER(C.c_int)
    r.hex_scs_det_.argtypes = [ip, ip, ip, _dp, _dp]
    r.hex_scv_det_.argtypes = [ip, ip, ip, _dp, _dp, _dp, ip]
    return r


def ref_hex_scs_det(r, coords8x3):
    one, npe, nscs = C.c_int(1), C.c_int(8), C.c_int(12)
    out = np.zeros((12, 3))  # area_vec(3, nelem=1, nscs) column-major == [nscs][3] C order
    r.hex_scs_det_(C.byref(one), C.byref(npe), C.byref(nscs), f64(coords8x3), out)
    return out


def ref_hex_scv_det(r, coords8x3):
    one, npe, nscv, nerr = C.c_int(1), C.c_int(8), C.c_int(8), C.c_int(0)
    vol, err = np.zeros(8), np.zeros(1)
    r.hex_scv_det_(C.byref(one), C.byref(npe), C.byref(nscv), f64(coords8x3), vol, err,
                   C.byref(nerr))
    return vol
