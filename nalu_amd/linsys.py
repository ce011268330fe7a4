"""Host-side mirror of the reference's LinearSystem / TpetraLinearSolver surface
(include/LinearSystem.h:40-168, include/LinearSolver.h:188-257) over the C ABI.

Python is only the test/bench harness language here; every method forwards to one
`nalu_gpu_*` entry point.  PyTorch supplies device memory and the stream handle."""
import ctypes as C

import numpy as np

from . import _lib as L


def _ptr(t):
    """device pointer of a torch tensor (or None)"""
    return None if t is None else C.c_void_p(t.data_ptr())


def _vec3(t):
    """(3, n) fp64 device tensor with unit-stride rows -> nalu_gpu_vec3 of its rows (SoA);
    the rows need not be adjacent (a padded row pitch is fine)"""
    assert t.dim() == 2 and t.shape[0] == 3 and t.stride(1) == 1 and t.dtype.is_floating_point
    assert t.element_size() == 8
    n = t.stride(0) * 8
    p = t.data_ptr()
    return L.Vec3(p, p + n, p + 2 * n)


def _np(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class LinearSolverConfig:
    """TpetraLinearSolverConfig (src/LinearSolverConfig.C:34-107); same keys and defaults."""

    def __init__(self, name="solver", method="gmres", preconditioner="default", tolerance=1e-4,
                 final_tolerance=None, max_iterations=50, kspace=50, output_level=0,
                 write_matrix_files=False, check_every=0):
        if preconditioner not in ("sgs", "mt_sgs", "jacobi", "default", "ilut", "riluk", "muelu",
                                  "none", "chebyshev"):
            # src/LinearSolverConfig.C:97-99
            raise RuntimeError("invalid linear solver preconditioner specified ")
        self.name, self.method, self.preconditioner = name, method, preconditioner
        self.tolerance = tolerance
        self.final_tolerance = tolerance if final_tolerance is None else final_tolerance
        self.max_iterations, self.kspace = max_iterations, kspace
        self.output_level, self.write_matrix_files = output_level, write_matrix_files
        self.check_every = check_every

    @classmethod
    def from_dict(cls, node):
        """One entry of the YAML `linear_solvers:` list, already parsed into a dict."""
        keys = ("name", "method", "preconditioner", "tolerance", "final_tolerance",
                "max_iterations", "kspace", "output_level", "write_matrix_files")
        return cls(**{k: node[k] for k in keys if k in node})

    def c_struct(self, is_final_outer_iter=False):
        if self.method not in L.METHOD:
            raise L.NaluGpuError(f"solver method '{self.method}' not supported (cg, gmres, biCgStab, tfqmr)")
        if self.preconditioner not in L.PRECOND:
            raise L.NaluGpuError(f"preconditioner '{self.preconditioner}' not supported on GPU")
        tol = self.final_tolerance if is_final_outer_iter else self.tolerance
        return L.SolverConfig(L.METHOD[self.method], L.PRECOND[self.preconditioner], tol,
                              self.max_iterations, self.kspace, self.check_every)


class ScatterPlan:
    """nalu_gpu_scatter_plan: entity -> matrix-slot map of one connectivity."""

    def __init__(self, owner, ent_lids):
        a = _np(ent_lids, np.int32)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        self.owner, self.n_ent, self.n_obj = owner, a.shape[0], a.shape[1]
        self.p = C.c_void_p()
        L.check(owner.lib.nalu_gpu_scatter_plan_create(owner.h, self.n_ent, self.n_obj,
                                                       a.ctypes.data, C.byref(self.p)))

    def n_colors(self):
        n = C.c_int()
        L.check(self.owner.lib.nalu_gpu_scatter_plan_colors(self.p, C.byref(n)))
        return n.value

    def close(self):
        # safe in either order: a system that is destroyed first releases its plans' device memory
        if self.p:
            self.owner.lib.nalu_gpu_scatter_plan_destroy(self.p)
        self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SharePlan:
    """nalu_gpu_share_plan: the shared entities of every halo neighbour (stk::mesh::parallel_sum)."""

    def __init__(self, owner, send_offsets, send_ids, recv_offsets, recv_ids, n_entities):
        so, si = _np(send_offsets, np.int32), _np(send_ids, np.int32)
        ro, ri = _np(recv_offsets, np.int32), _np(recv_ids, np.int32)
        self.owner, self.n_entities = owner, int(n_entities)
        self.p = C.c_void_p()
        L.check(owner.lib.nalu_gpu_share_plan_create(owner.h, so.ctypes.data, si.ctypes.data, ro.ctypes.data,
                                                     ri.ctypes.data, self.n_entities, C.byref(self.p)))

    def close(self):
        if self.p and self.owner.h:  # (before its linear system goes away)
            self.owner.lib.nalu_gpu_share_plan_destroy(self.p)
        self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def suggest_edge_order(edge_nodes):
    """nalu_gpu_suggest_edge_order: the slot-major permutation (perm[new] = old) of an edge list of local
    node ids -- apply it to the edge list AND to every edge field before buildEdgeToNodeGraph."""
    a = _np(edge_nodes, np.int32)
    assert a.ndim == 2 and a.shape[1] == 2
    perm = np.empty(a.shape[0], np.int64)
    L.check(L.load().nalu_gpu_suggest_edge_order(a.shape[0], a.ctypes.data, perm.ctypes.data))
    return perm


class GpuLinearSystem:
    """One sierra::nalu::LinearSystem on one GPU."""

    def __init__(self, num_dof=1, device=0, stream=None):
        self.lib = L.load()
        self.h = C.c_void_p()
        L.check(self.lib.nalu_gpu_create(C.byref(self.h), device, num_dof,
                                         C.c_void_p(stream) if stream else None))
        self.num_dof = num_dof
        self.device = device
        self._comm_keepalive = None
        # accessors of include/LinearSystem.h:129-139
        self.linearSolveIterations = 0
        self.linearResidual = 0.0
        self.nonLinearResidual = 0.0
        self.scaledNonLinearResidual = 0.0
        self.firstNonLinearResidual = None
        self.timer_precond = 0.0
        self.history = np.zeros(0)
        self.writeCounter = 0  # include/LinearSystem.h: writeCounter_, bumped by the caller

    def close(self):
        if self.h:
            self.lib.nalu_gpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_assembly_mode(self, mode):
        """"owner_computes" (default) or "export" (the reference's shared-not-owned rows + Export ADD)"""
        L.check(self.lib.nalu_gpu_set_assembly_mode(self.h, {"owner_computes": 0, "export": 1}[mode]))

    def set_edge_ownership(self, locally_owned):
        a = _np(locally_owned, np.uint8)
        L.check(self.lib.nalu_gpu_set_edge_ownership(self.h, len(a), a.ctypes.data))

    # ---- graph ----------------------------------------------------------------
    def beginLinearSystemConstruction(self, n_owned, n_shared=0, n_ghost=0, gids=None,
                                      owners=None, my_rank=0):
        n = n_owned + n_shared + n_ghost
        gids = np.arange(1, n + 1, dtype=np.int64) if gids is None else _np(gids, np.int64)
        assert len(gids) == n
        own = None if owners is None else _np(owners, np.int32)
        L.check(self.lib.nalu_gpu_begin_construction(
            self.h, n_owned, n_shared, n_ghost, gids.ctypes.data,
            None if own is None else own.ctypes.data, my_rank))
        self.n_owned, self.n_shared, self.n_ghost, self.n_cols = n_owned, n_shared, n_ghost, n
        self.gids, self.my_rank = gids.copy(), my_rank

    def getDofStatus(self, node_lids):
        """DofStatus of local nodes (src/TpetraLinearSystem.C:157-160, 2111-2157)."""
        a = _np(np.atleast_1d(node_lids), np.int32)
        st = np.zeros(len(a), np.int32)
        L.check(self.lib.nalu_gpu_get_dof_status(self.h, len(a), a.ctypes.data, st.ctypes.data))
        return st

    def buildNodeGraph(self, node_lids):
        a = _np(node_lids, np.int32)
        L.check(self.lib.nalu_gpu_build_node_graph(self.h, len(a), a.ctypes.data))

    def buildEdgeToNodeGraph(self, edge_nodes):
        a = _np(edge_nodes, np.int32)
        assert a.ndim == 2 and a.shape[1] == 2
        L.check(self.lib.nalu_gpu_build_edge_to_node_graph(self.h, a.shape[0], a.ctypes.data))
        self.n_edges = getattr(self, "n_edges", 0) + a.shape[0]

    def buildElemToNodeGraph(self, elem_nodes):
        a = _np(elem_nodes, np.int32)
        L.check(self.lib.nalu_gpu_build_elem_to_node_graph(self.h, a.shape[0], a.shape[1],
                                                           a.ctypes.data))

    buildFaceToNodeGraph = buildElemToNodeGraph

    def buildDirichletNodeGraph(self, node_lids):
        a = _np(node_lids, np.int32)
        L.check(self.lib.nalu_gpu_build_dirichlet_node_graph(self.h, len(a), a.ctypes.data))

    def finalizeLinearSystem(self):
        L.check(self.lib.nalu_gpu_finalize(self.h))
        nr, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.lib.nalu_gpu_get_sizes(self.h, C.byref(nr), C.byref(nc), C.byref(nnz)))
        self.n_rows, self.n_cols, self.nnz = nr.value, nc.value, nnz.value

    # ---- views ----------------------------------------------------------------
    def graph(self):
        rp = np.zeros(self.n_rows + 1, np.int64)
        ci = np.zeros(max(self.nnz, 1), np.int32)
        L.check(self.lib.nalu_gpu_get_graph(self.h, rp.ctypes.data, ci.ctypes.data))
        return rp, ci[:self.nnz]

    def matrix_values(self):
        v = np.zeros(max(self.nnz, 1))
        L.check(self.lib.nalu_gpu_get_matrix_values(self.h, v.ctypes.data))
        return v[:self.nnz]

    def set_matrix_values(self, v):
        v = _np(v, np.float64)
        assert len(v) == self.nnz
        L.check(self.lib.nalu_gpu_set_matrix_values(self.h, v.ctypes.data))

    def rhs(self):
        b = np.zeros(max(self.n_rows, 1))
        L.check(self.lib.nalu_gpu_get_rhs(self.h, b.ctypes.data))
        return b[:self.n_rows]

    def set_rhs(self, b):
        b = _np(b, np.float64)
        assert len(b) == self.n_rows
        L.check(self.lib.nalu_gpu_set_rhs(self.h, b.ctypes.data))

    def solution(self):
        x = np.zeros(max(self.n_rows, 1))
        L.check(self.lib.nalu_gpu_get_solution(self.h, x.ctypes.data))
        return x[:self.n_rows]

    def layout_info(self):
        """nalu_gpu_get_layout_info as a dict (which fast paths the mesh numbering gets)"""
        li = L.LayoutInfo()
        L.check(self.lib.nalu_gpu_get_layout_info(self.h, C.byref(li)))
        return {k: getattr(li, k) for k, _ in li._fields_}

    def check_system(self):
        """checkForNaN / checkForZeroRow (src/TpetraLinearSystem.C:1782-1899) as a dict"""
        ci = L.CheckInfo()
        L.check(self.lib.nalu_gpu_check_system(self.h, C.byref(ci)))
        return {k: getattr(ci, k) for k, _ in ci._fields_}

    def set_debug(self, on=True):
        """realm_.debug(): run the checks before every solve and fail instead of solving"""
        L.check(self.lib.nalu_gpu_set_debug(self.h, int(on)))

    # ---- MatrixMarket dumps (src/TpetraLinearSystem.C:1901-1924, 2002-2019) -----
    def writeToFile(self, base_filename, useOwned=True, eq_sys_name="", n_ranks=1,
                    n_global=None):
        """`<base>-O-<count>.mm.<p_size>` + `.rhs.<p_size>` of the owned matrix/rhs in global
        ids.  The shared-not-owned matrix does not exist here (its rows are assembled by
        their owner), so useOwned=False raises."""
        if not useOwned:
            raise RuntimeError("writeToFile: there is no shared-not-owned matrix in this "
                               "linear system (cut edges are assembled by the row owner)")
        from . import mmio
        rp, ci = self.graph()
        ng = int(n_global if n_global is not None else
                 (self.n_owned if n_ranks == 1 else self.gids.max()))
        names = mmio.file_names(base_filename, self.writeCounter, n_ranks, self.my_rank)
        mmio.write_sparse(names["mm"], ng * self.num_dof, ng * self.num_dof,
                          self.gids[:self.n_owned], rp, self.gids[ci], self.matrix_values(),
                          eq_sys_name, "Tpetra matrix for: " + eq_sys_name)
        mmio.write_dense(names["rhs"], self.rhs())
        return names

    def writeSolutionToFile(self, base_filename, useOwned=True, n_ranks=1):
        from . import mmio
        names = mmio.file_names(base_filename, self.writeCounter, n_ranks, self.my_rank)
        mmio.write_dense(names["sln"], self.solution())
        return names

    def readFromFile(self, base_filename, n_ranks=1, read_rhs=True):
        """Load the matrix values (and rhs) of a `writeToFile` dump -- this library's or a
        Trilinos run's -- into the assembled system.  The entries are matched to the graph by
        global (row, column) id, so their order in the file is free; an entry outside the
        graph, or a graph entry the file lacks, is an error."""
        from . import mmio
        names = mmio.file_names(base_filename, self.writeCounter, n_ranks, self.my_rank)
        _, _, r, c, v = mmio.read_sparse(names["mm"])
        rp, ci = self.graph()
        rows = np.repeat(np.arange(self.n_rows), np.diff(rp))
        span = int(self.gids.max()) + 1
        have = self.gids[rows].astype(np.int64) * span + self.gids[ci]
        order = np.argsort(have, kind="stable")
        want = r * span + c
        pos = np.searchsorted(have[order], want)
        ok = (pos < len(have)) & (have[order][np.minimum(pos, len(have) - 1)] == want)
        if not ok.all() or len(np.unique(want)) != len(have):
            raise RuntimeError(f"{names['mm']} does not match the graph of this linear system")
        vals = np.empty(self.nnz)
        vals[order[pos]] = v
        self.set_matrix_values(vals)
        if read_rhs:
            self.set_rhs(mmio.read_dense(names["rhs"]))
        return names

    # ---- assembly -------------------------------------------------------------
    def zeroSystem(self):
        L.check(self.lib.nalu_gpu_zero_system(self.h))

    def sumInto(self, node_lids, rhs, lhs):
        ids = _np(node_lids, np.int32)
        r, m = _np(rhs, np.float64), _np(lhs, np.float64)
        assert r.size == ids.size and m.size == ids.size ** 2
        L.check(self.lib.nalu_gpu_sum_into(self.h, len(ids), ids.ctypes.data, r.ctypes.data,
                                           m.ctypes.data))

    def createScatterPlan(self, ent_lids):
        """Plan of the bulk sumInto for one connectivity: ent_lids (n_ent, n_obj) local ids."""
        return ScatterPlan(self, ent_lids)

    def sumIntoBatch(self, plan, rhs, lhs, mode="rows"):
        """sumInto (view overload, src/TpetraLinearSystem.C:1387-1452) of all entities of the
        plan: device tensors rhs (n_ent, n_obj) and lhs (n_ent, n_obj, n_obj)."""
        assert plan.owner is self
        assert rhs.numel() == plan.n_ent * plan.n_obj and rhs.is_contiguous()
        assert lhs.numel() == plan.n_ent * plan.n_obj ** 2 and lhs.is_contiguous()
        assert rhs.element_size() == 8 and lhs.element_size() == 8
        L.check(self.lib.nalu_gpu_sum_into_batch(self.h, plan.p, _ptr(rhs), _ptr(lhs),
                                                 L.SCATTER[mode]))

    def assemble_continuity_edges(self, coords, dpdx, vel, pressure, density, edge_area, dt,
                                  gamma1=1.0, noc=1.0, interp_together=1.0, mode="rows"):
        f = L.ContinuityFields(_vec3(coords), _vec3(dpdx), _vec3(vel), pressure.data_ptr(),
                               density.data_ptr(), _vec3(edge_area))
        self._check_fields(coords, dpdx, vel, pressure, density, edge_area=edge_area)
        L.check(self.lib.nalu_gpu_assemble_continuity_edges(
            self.h, C.byref(f), dt, gamma1, noc, interp_together, L.SCATTER[mode]))

    def assemble_scalar_edges(self, coords, dqdx, vel, q, density, diff_flux_coeff, edge_area,
                              mdot, alpha=0.0, alpha_upw=1.0, upw_factor=1.0, limiter=False,
                              peclet="classic", hybrid_factor=0.0, tanh_c1=2.0, tanh_c2=4.0,
                              mode="rows"):
        """Option defaults are the reference's (src/SolutionOptions.C:32-51)."""
        self._check_fields(coords, dqdx, vel, q, density, diff_flux_coeff, edge_area=edge_area,
                           edge_scalar=mdot)
        f = L.ScalarFields(_vec3(coords), _vec3(dqdx), _vec3(vel), q.data_ptr(),
                           density.data_ptr(), diff_flux_coeff.data_ptr(), _vec3(edge_area),
                           mdot.data_ptr())
        o = L.ScalarOptions(alpha, alpha_upw, upw_factor, int(limiter), L.PECLET[peclet],
                            hybrid_factor, tanh_c1, tanh_c2)
        L.check(self.lib.nalu_gpu_assemble_scalar_edges(self.h, C.byref(f), C.byref(o),
                                                        L.SCATTER[mode]))

    def _check_fields(self, *node_fields, edge_area=None, edge_scalar=None):
        # host-side shape checks: the kernels index n_cols nodes and n_edges edges
        for t in node_fields:
            assert t.shape[-1] >= self.n_cols and t.stride(-1) == 1 and t.element_size() == 8, \
                "nodal field smaller than the local node count"
        ne = getattr(self, "n_edges", 0)
        if edge_area is not None:
            assert edge_area.shape[-1] >= ne
        if edge_scalar is not None:
            assert edge_scalar.shape[-1] >= ne and edge_scalar.is_contiguous()

    def assemble_scalar_mass_be(self, q_n, q_np1, rho_n, rho_np1, dual_volume, dt):
        self._check_fields(q_n, q_np1, rho_n, rho_np1, dual_volume)
        L.check(self.lib.nalu_gpu_assemble_scalar_mass_be(
            self.h, _ptr(q_n), _ptr(q_np1), _ptr(rho_n), _ptr(rho_np1), _ptr(dual_volume), dt))

    def applyDirichletBCs(self, solution_field, bc_values_field, node_lids):
        ids = _np(node_lids, np.int32)
        self._check_fields(solution_field, bc_values_field)
        L.check(self.lib.nalu_gpu_apply_dirichlet(self.h, len(ids), ids.ctypes.data,
                                                  _ptr(solution_field), _ptr(bc_values_field)))

    def resetRows(self, node_lids):
        ids = _np(node_lids, np.int32)
        L.check(self.lib.nalu_gpu_reset_rows(self.h, len(ids), ids.ctypes.data))

    def loadComplete(self):
        L.check(self.lib.nalu_gpu_load_complete(self.h))

    # ---- solve ----------------------------------------------------------------
    def solve(self, config, x_out=None, is_final_outer_iter=False, l2_scaling=1.0,
              first_time_step_solve=True):
        """TpetraLinearSystem::solve (src/TpetraLinearSystem.C:1709-1780)."""
        cfg = config.c_struct(is_final_outer_iter)
        info = L.SolveInfo()
        hist = np.zeros(config.max_iterations + 2)
        if x_out is not None:
            assert x_out.numel() >= self.n_cols and x_out.is_contiguous()
        L.check(self.lib.nalu_gpu_solve(self.h, C.byref(cfg), _ptr(x_out), C.byref(info),
                                        hist.ctypes.data, len(hist)))
        self.history = hist[:info.history_len].copy()
        self.linearSolveIterations = info.iters
        self.linearResidual = info.final_resid
        self.nonLinearResidual = l2_scaling * info.rhs_norm2
        if first_time_step_solve or self.firstNonLinearResidual is None:
            self.firstNonLinearResidual = self.nonLinearResidual
        self.scaledNonLinearResidual = self.nonLinearResidual / max(
            np.finfo(float).eps, self.firstNonLinearResidual)
        self.timer_precond = info.precond_seconds
        self.solve_seconds = info.solve_seconds
        return 0  # status is always 0 in the reference (src/LinearSolver.C:191)

    # ---- building blocks ------------------------------------------------------
    def spmv(self, x, y):
        assert x.numel() >= self.n_cols and y.numel() >= self.n_rows
        L.check(self.lib.nalu_gpu_spmv(self.h, _ptr(x), _ptr(y)))

    def dot(self, a, b):
        out = C.c_double()
        n = min(a.numel(), b.numel())
        L.check(self.lib.nalu_gpu_dot(self.h, _ptr(a), _ptr(b), n, C.byref(out)))
        return out.value

    def norm2(self, a):
        out = C.c_double()
        L.check(self.lib.nalu_gpu_norm2(self.h, _ptr(a), a.numel(), C.byref(out)))
        return out.value

    def axpy(self, alpha, x, y):
        assert x.numel() == y.numel()
        L.check(self.lib.nalu_gpu_axpy(self.h, alpha, _ptr(x), _ptr(y), x.numel()))

    def compute_mdot_edges(self, coords, dpdx, vel, pressure, density, edge_area, dt, mdot_out,
                           gamma1=1.0, noc=1.0, interp_together=1.0):
        self._check_fields(coords, dpdx, vel, pressure, density, edge_area=edge_area,
                           edge_scalar=mdot_out)
        f = L.ContinuityFields(_vec3(coords), _vec3(dpdx), _vec3(vel), pressure.data_ptr(),
                               density.data_ptr(), _vec3(edge_area))
        L.check(self.lib.nalu_gpu_compute_mdot_edges(self.h, C.byref(f), dt, gamma1, noc,
                                                     interp_together, _ptr(mdot_out)))

    def nodal_grad_edges(self, edge_area, dual_volume, q, grad_out):
        self._check_fields(dual_volume, q, edge_area=edge_area)
        g = _vec3(grad_out)
        assert grad_out.shape[1] >= self.n_cols  # (export mode also writes the shared-not-owned entries)
        L.check(self.lib.nalu_gpu_nodal_grad_edges(self.h, _vec3(edge_area), _ptr(dual_volume),
                                                   _ptr(q), g.x, g.y, g.z))

    # ---- multi-GPU --------------------------------------------------------------
    def compute_geometry_hex8(self, hex_nodes, coords, dual_volume_out, edge_area_out):
        """dual_nodal_volume (n_cols) and edge_area_vector (3, n_edges) of a Hex8 mesh
        (src/ComputeGeometryInteriorAlgorithm.C:78-216); hex_nodes: int32 device tensor
        (n_elem, 8) of local node ids."""
        ne = getattr(self, "n_edges", 0)
        assert hex_nodes.dtype.itemsize == 4 and hex_nodes.is_contiguous() and hex_nodes.shape[-1] == 8
        self._check_fields(coords, dual_volume_out)
        assert edge_area_out.shape[0] == 3 and edge_area_out.shape[-1] >= ne and edge_area_out.is_contiguous()
        n = edge_area_out.shape[-1]
        p = edge_area_out.data_ptr()
        L.check(self.lib.nalu_gpu_compute_geometry_hex8(
            self.h, hex_nodes.shape[0], _ptr(hex_nodes), _vec3(coords), _ptr(dual_volume_out),
            C.c_void_p(p), C.c_void_p(p + 8 * n), C.c_void_p(p + 16 * n)))

    def field_axpby(self, alpha, x, beta, y):
        """y = alpha x + beta y (src/FieldFunctions.C:24-58)"""
        assert x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous()
        L.check(self.lib.nalu_gpu_field_axpby(self.h, y.numel(), alpha, _ptr(x), beta, _ptr(y)))

    def project_nodal_velocity(self, dt, gamma1, dpdx_new, dpdx_old, density, velocity,
                               projected=None):
        """u -= (dt/gamma1)/rho (dpdx_new - dpdx_old), SoA (3, n) fields
        (src/LowMachEquationSystem.C:772-855); `projected`: uint8 mask or None."""
        n = density.numel()
        for t in (dpdx_new, dpdx_old, velocity):
            assert t.shape[0] == 3 and t.shape[-1] >= n and t.is_contiguous()
        assert projected is None or (projected.numel() >= n and projected.element_size() == 1)
        L.check(self.lib.nalu_gpu_project_nodal_velocity(
            self.h, n, dt, gamma1, _vec3(dpdx_new), _vec3(dpdx_old), _ptr(density),
            _ptr(projected), _vec3(velocity)))

    def update_and_clip(self, dz, z, z_unfiltered=None, delta_clip=0.0, diagnostics=False):
        """z += dz, clipped to [0 - delta_clip, 1 + delta_clip] (src/MixtureFractionEquationSystem.C:942-1013);
        diagnostics: returns [n_clipped_low, n_clipped_high, min, max] of this rank"""
        n = z.numel()
        assert dz.numel() >= n and z.is_contiguous() and dz.is_contiguous() and z.element_size() == 8
        assert z_unfiltered is None or (z_unfiltered.numel() >= n and z_unfiltered.is_contiguous())
        info = np.zeros(4) if diagnostics else None
        L.check(self.lib.nalu_gpu_update_and_clip(self.h, n, _ptr(dz), _ptr(z), _ptr(z_unfiltered), delta_clip,
                                                  None if info is None else info.ctypes.data))
        return info

    def createSharePlan(self, send_offsets, send_ids, recv_offsets, recv_ids, n_entities):
        """share plan of stk::mesh::parallel_sum for non-nodal entities (edges): per halo neighbour the owned
        entities it shares (send) and the entities it owns that this rank shares (recv)"""
        return SharePlan(self, send_offsets, send_ids, recv_offsets, recv_ids, n_entities)

    def parallel_sum(self, fields, plan=None):
        """stk::mesh::parallel_sum of device arrays: nodal fields (plan None, n_cols entries each) or the
        entities of a share plan"""
        assert (plan is None or plan.owner is self) and 1 <= len(fields) <= 16
        n = self.n_cols if plan is None else plan.n_entities
        for t in fields:
            assert t.dim() == 1 and t.stride(0) == 1 and t.element_size() == 8 and t.numel() >= n
        ptrs = (C.c_void_p * len(fields))(*[t.data_ptr() for t in fields])
        L.check(self.lib.nalu_gpu_parallel_sum(self.h, None if plan is None else plan.p, len(fields), ptrs))

    def set_halo_plan(self, neighbor_ranks, send_offsets, send_lids, recv_offsets, recv_lids=None):
        nb = _np(neighbor_ranks, np.int32)
        so, sl, ro = _np(send_offsets, np.int32), _np(send_lids, np.int32), _np(recv_offsets, np.int32)
        if recv_lids is None:
            L.check(self.lib.nalu_gpu_set_halo_plan(self.h, len(nb), nb.ctypes.data, so.ctypes.data,
                                                    sl.ctypes.data, ro.ctypes.data))
        else:  # the general (indexed) form: several owners among the shared-not-owned nodes
            rl = _np(recv_lids, np.int32)
            L.check(self.lib.nalu_gpu_set_halo_plan_indexed(self.h, len(nb), nb.ctypes.data, so.ctypes.data,
                                                            sl.ctypes.data, ro.ctypes.data, rl.ctypes.data))

    def comm_buffers(self):
        sb, rb, red = C.c_void_p(), C.c_void_p(), C.c_void_p()
        sc, rc, cap = C.c_int64(), C.c_int64(), C.c_int()
        L.check(self.lib.nalu_gpu_get_comm_buffers(self.h, C.byref(sb), C.byref(sc), C.byref(rb),
                                                   C.byref(rc), C.byref(red), C.byref(cap)))
        return (sb.value, sc.value), (rb.value, rc.value), (red.value, cap.value)

    def set_comm(self, comm_struct):
        self._comm_keepalive = comm_struct
        L.check(self.lib.nalu_gpu_set_comm(self.h, C.byref(comm_struct) if comm_struct else None))

    def halo_exchange(self, vec):
        assert vec.numel() >= self.n_cols
        L.check(self.lib.nalu_gpu_halo_exchange(self.h, _ptr(vec)))

    # ---- timers ---------------------------------------------------------------
    def set_profiling(self, on=True):
        """on: False / True, or n > 1: the SpMV timer times every n-th launch (an event pair costs ~4 us)"""
        L.check(self.lib.nalu_gpu_set_profiling(self.h, int(on)))

    def sgs_schedule(self):
        """how the exact SGS sweeps run (nalu_gpu_get_sgs_schedule): dict with mode 'flow' / 'levels' / None"""
        out = (C.c_int * 8)()
        L.check(self.lib.nalu_gpu_get_sgs_schedule(self.h, out))
        mode = {1: "flow", -1: "levels", -2: "levels", 0: None}[out[0]]
        return dict(mode=mode, fell_back=out[0] == -2, tiled=bool(out[1]), chunks=out[2], max_steps=out[3], widest_row=out[4],
                    ring=out[5], tile=(out[6], out[7]))

    def timer_launches(self, name):
        """launches a timer saw while profiling was on (timer()'s count: the launches timed)"""
        n = C.c_int64()
        L.check(self.lib.nalu_gpu_get_timer_launches(self.h, L.TIMERS[name], C.byref(n)))
        return n.value

    def timer(self, name):
        ms, cnt = C.c_double(), C.c_int64()
        L.check(self.lib.nalu_gpu_get_timer(self.h, L.TIMERS[name], C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def reset_timers(self):
        L.check(self.lib.nalu_gpu_reset_timers(self.h))

    def synchronize(self):
        L.check(self.lib.nalu_gpu_synchronize(self.h))
