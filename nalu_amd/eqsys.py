"""Host-side mirror of the callers of the hot path: SolverAlgorithmDriver::execute
(src/SolverAlgorithmDriver.C:109-138) and EquationSystem::assemble_and_solve
(src/EquationSystem.C:263-307), with the reference's timers
(timerAssemble_/timerLoadComplete_/timerSolve_/timerPrecond_).  The algorithms are the GPU
replacements of the reference's SolverAlgorithm::execute() bodies: each one is a single bulk
call on the GpuLinearSystem instead of a per-edge apply_coeff loop."""
import time

# include/Enums.h:18-43 -- the std::map<AlgorithmType, ...> iteration order of the driver
ALGORITHM_TYPE = {"INTERIOR": 0, "INFLOW": 1, "WALL": 2, "OPEN": 3, "MASS": 4, "SRC": 5,
                  "SYMMETRY": 6, "WALL_HF": 7, "WALL_CHT": 8, "WALL_RAD": 9, "NON_CONFORMAL": 10,
                  "ELEM_SOURCE": 11, "OVERSET": 12, "WALL_ABL": 13, "REF_PRESSURE": 14}


class SolverAlgorithm:
    """include/SolverAlgorithm.h:34-55: initialize_connectivity + execute"""

    def initialize_connectivity(self, linsys):
        pass

    def execute(self, linsys):
        raise NotImplementedError


class AssembleContinuityEdgeSolverAlgorithm(SolverAlgorithm):
    """src/AssembleContinuityEdgeSolverAlgorithm.C:62-209; `fields` holds the SoA device
    tensors coords, dpdx, vel, p, rho, area; `edge_nodes` the local edge connectivity."""

    def __init__(self, edge_nodes, fields, dt, gamma1=1.0, noc=1.0, interp_together=1.0,
                 mode="rows"):
        self.edge_nodes, self.f = edge_nodes, fields
        self.dt, self.gamma1, self.noc, self.interp, self.mode = dt, gamma1, noc, interp_together, mode

    def initialize_connectivity(self, linsys):  # :62-66
        linsys.buildEdgeToNodeGraph(self.edge_nodes)

    def execute(self, linsys):
        f = self.f
        linsys.assemble_continuity_edges(f["coords"], f["dpdx"], f["vel"], f["p"], f["rho"],
                                         f["area"], self.dt, self.gamma1, self.noc, self.interp,
                                         self.mode)


class AssembleScalarEdgeSolverAlgorithm(SolverAlgorithm):
    """src/AssembleScalarEdgeSolverAlgorithm.C:78-315; fields: coords, dqdx, vel, q, rho,
    gamma, area, mdot; `opts` are the keyword options of assemble_scalar_edges."""

    def __init__(self, edge_nodes, fields, mode="rows", **opts):
        self.edge_nodes, self.f, self.mode, self.opts = edge_nodes, fields, mode, opts

    def initialize_connectivity(self, linsys):
        linsys.buildEdgeToNodeGraph(self.edge_nodes)

    def execute(self, linsys):
        f = self.f
        linsys.assemble_scalar_edges(f["coords"], f["dqdx"], f["vel"], f["q"], f["rho"], f["gamma"],
                                     f["area"], f["mdot"], mode=self.mode, **self.opts)


class ScalarMassBackwardEulerAlgorithm(SolverAlgorithm):
    """AssembleNodeSolverAlgorithm + ScalarMassBackwardEulerNodeSuppAlg
    (src/AssembleNodeSolverAlgorithm.C:57-112, src/ScalarMassBackwardEulerNodeSuppAlg.C:65-80)"""

    def __init__(self, node_lids, q_n, q_np1, rho_n, rho_np1, dual_volume, dt):
        self.node_lids = node_lids
        self.args = (q_n, q_np1, rho_n, rho_np1, dual_volume, dt)

    def initialize_connectivity(self, linsys):  # :50-54 buildNodeGraph
        linsys.buildNodeGraph(self.node_lids)

    def execute(self, linsys):
        linsys.assemble_scalar_mass_be(*self.args)


class FixPressureAtNodeAlgorithm(SolverAlgorithm):
    """src/FixPressureAtNodeAlgorithm.C:59-79: reset the row, then sum 1.0 / (pRef - p)."""

    def __init__(self, node_lid, ref_pressure, pressure_at_node):
        """pressure_at_node: a number, or the (device) pressure field itself -- read at execute() like
        the reference's stk::mesh::field_data(*pressure_, node) (:72)"""
        self.node, self.pref, self.p = node_lid, ref_pressure, pressure_at_node

    def execute(self, linsys):
        if self.node is None:  # the node lives on another rank
            return
        p = self.p if isinstance(self.p, (int, float)) else float(self.p[self.node])
        linsys.resetRows([self.node])
        linsys.sumInto([self.node], [self.pref - p], [1.0])


class DirichletBC(SolverAlgorithm):
    """src/DirichletBC.C:39-49"""

    def __init__(self, node_lids, solution_field, bc_values_field):
        self.nodes, self.sol, self.bc = node_lids, solution_field, bc_values_field

    def initialize_connectivity(self, linsys):  # src/DirichletBC.C:32-36
        linsys.buildDirichletNodeGraph(self.nodes)

    def execute(self, linsys):
        linsys.applyDirichletBCs(self.sol, self.bc, self.nodes)


class SolverAlgorithmDriver:
    def __init__(self):
        self.solverAlgMap = {}            # AlgorithmType -> SolverAlgorithm
        self.solverConstraintAlgMap = {}
        self.solverDirichAlgMap = {}

    @staticmethod
    def _ordered(m):
        return [m[k] for k in sorted(m, key=lambda t: ALGORITHM_TYPE[t] if isinstance(t, str) else t)]

    def initialize_connectivity(self, linsys):  # src/SolverAlgorithmDriver.C:66-85
        for m in (self.solverAlgMap, self.solverConstraintAlgMap, self.solverDirichAlgMap):
            for alg in self._ordered(m):
                alg.initialize_connectivity(linsys)

    def execute(self, linsys):  # :109-138: interior/boundary, then constraints, Dirichlet last
        for m in (self.solverAlgMap, self.solverConstraintAlgMap, self.solverDirichAlgMap):
            for alg in self._ordered(m):
                alg.execute(linsys)


class EquationSystem:
    """The slice of sierra::nalu::EquationSystem the hot path needs."""

    def __init__(self, linsys, solver_config, name="EqSys", sync=None):
        self.linsys, self.config, self.name = linsys, solver_config, name
        self.solverAlgDriver = SolverAlgorithmDriver()
        self.timerAssemble = self.timerLoadComplete = self.timerSolve = self.timerPrecond = 0.0
        self.timerInit = 0.0
        self.numberLinearIterations = 0
        # the reference's timers bracket synchronous host code; GPU work is asynchronous, so a
        # caller who wants the buckets to mean device time passes sync=linsys.synchronize
        self._sync = sync or (lambda: None)

    def _now(self):
        self._sync()
        return time.perf_counter()

    def initialize(self):
        """<Eq>::initialize (src/LowMachEquationSystem.C:2939-2940): connectivity + finalize"""
        t = self._now()
        self.solverAlgDriver.initialize_connectivity(self.linsys)
        self.linsys.finalizeLinearSystem()
        self.timerInit += self._now() - t

    def assemble_and_solve(self, delta_solution=None, is_final_outer_iter=False, **solve_kw):
        """src/EquationSystem.C:263-307"""
        ls = self.linsys
        t = self._now()
        ls.zeroSystem()
        self.solverAlgDriver.execute(ls)
        t1 = self._now()
        self.timerAssemble += t1 - t
        ls.loadComplete()
        t2 = self._now()
        self.timerLoadComplete += t2 - t1
        if self.config.write_matrix_files:  # src/TpetraLinearSystem.C:1723-1726
            ls.writeToFile(self.name, eq_sys_name=self.name)
        error = ls.solve(self.config, delta_solution, is_final_outer_iter, **solve_kw)
        if self.config.write_matrix_files:  # :1747-1750
            ls.writeSolutionToFile(self.name)
            ls.writeCounter += 1
        self.timerSolve += self._now() - t2
        self.timerPrecond += ls.timer_precond
        self.numberLinearIterations += ls.linearSolveIterations  # update_iteration_statistics
        if error > 0:
            print(f"Error in {self.name}::solve_and_update()  ")
        return error


class ContinuityProjectionLoop:
    """The continuity half of LowMachEquationSystem::solve_and_update (src/LowMachEquationSystem.C:621-717)
    with everything device resident: per Picard iteration
        continuity assemble_and_solve(pTmp)   (:674)
        pressure += pTmp                      (:678-683, field_axpby)
        mass_flow_rate = ComputeMdotEdge      (:689)
        project_nodal_velocity                (:695 -> :772-855: uTmp = dpdx, dpdx = G p, u -= tau/rho (dpdx - uTmp))
    (the momentum solve between the iterations is out of scope; compute_vrtm is the identity on a
    fixed mesh).  `eq` is the continuity EquationSystem whose INTERIOR algorithm reads F; F holds the
    SoA device tensors coords, dpdx, vel, p, rho, area, vol, mdot (nodal ones with n_cols entries: on
    several ranks the copies are refreshed by the halo exchange after every nodal gradient)."""

    def __init__(self, eq, F, dt, gamma1=1.0, noc=1.0, interp_together=1.0):
        import torch
        self.eq, self.g, self.F = eq, eq.linsys, F
        self.dt, self.gamma1, self.noc, self.interp = dt, gamma1, noc, interp_together
        self.pTmp = torch.zeros_like(F["p"])
        self.uTmp = torch.zeros_like(F["dpdx"])
        self.isInit = True
        self.timer = dict(assemble_and_solve=0.0, update=0.0, mdot=0.0, project=0.0)

    def compute_projected_nodal_gradient(self):
        """AssembleNodalGradAlgorithmDriver (src/AssembleNodalGradAlgorithmDriver.C:59-123): dpdx = G p"""
        g, F = self.g, self.F
        g.nodal_grad_edges(F["area"], F["vol"], F["p"], F["dpdx"])
        if g.n_cols > g.n_owned:
            for d in range(3):
                g.halo_exchange(F["dpdx"][d])

    def compute_mdot(self):
        F = self.F
        self.g.compute_mdot_edges(F["coords"], F["dpdx"], F["vel"], F["p"], F["rho"], F["area"], self.dt, F["mdot"],
                                  self.gamma1, self.noc, self.interp)

    def project_nodal_velocity(self):
        """:772-855"""
        F = self.F
        self.uTmp.copy_(F["dpdx"])  # (torch copy on the current stream = the library's stream)
        self.compute_projected_nodal_gradient()
        self.g.project_nodal_velocity(self.dt, self.gamma1, F["dpdx"], self.uTmp, F["rho"], F["vel"])

    def solve_and_update(self, max_iterations=1, sync=None):
        import time
        sync = sync or (lambda: None)

        def lap(name, t0):
            sync()
            t1 = time.perf_counter()
            self.timer[name] += t1 - t0
            return t1
        if self.isInit:  # :626-634
            self.compute_projected_nodal_gradient()
            self.compute_mdot()
            self.isInit = False
        for _ in range(max_iterations):
            sync()
            t = time.perf_counter()
            self.eq.assemble_and_solve(self.pTmp)
            t = lap("assemble_and_solve", t)
            self.g.field_axpby(1.0, self.pTmp, 1.0, self.F["p"])
            t = lap("update", t)
            self.compute_mdot()
            t = lap("mdot", t)
            self.project_nodal_velocity()
            lap("project", t)


class ScalarTransportLoop:
    """MixtureFractionEquationSystem::solve_and_update (src/MixtureFractionEquationSystem.C:905-937): per
    iteration assemble_and_solve(zTmp), update_and_clip (:942-1013), compute_projected_nodal_gradient.
    F: coords, dqdx, vel, q, rho, gamma, area, mdot, vol (SoA device tensors, nodal ones n_cols long)."""

    def __init__(self, eq, F, delta_clip=0.0):
        import torch
        self.eq, self.g, self.F, self.delta_clip = eq, eq.linsys, F, delta_clip
        self.zTmp = torch.zeros_like(F["q"])
        self.qUF = torch.zeros_like(F["q"])
        self.isInit = True
        self.clip_info = None

    def compute_projected_nodal_gradient(self):
        g, F = self.g, self.F
        g.nodal_grad_edges(F["area"], F["vol"], F["q"], F["dqdx"])
        if g.n_cols > g.n_owned:
            for d in range(3):
                g.halo_exchange(F["dqdx"][d])

    def solve_and_update(self, max_iterations=1, diagnostics=False):
        if self.isInit:  # :910-913
            self.compute_projected_nodal_gradient()
            self.isInit = False
        for _ in range(max_iterations):
            self.eq.assemble_and_solve(self.zTmp)
            self.clip_info = self.g.update_and_clip(self.zTmp, self.F["q"], self.qUF, self.delta_clip, diagnostics)
            self.compute_projected_nodal_gradient()
