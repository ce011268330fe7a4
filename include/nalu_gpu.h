/*
 * nalu_gpu.h -- C ABI of the MI355X-native drop-in for gantech/Nalu's edge-based
 * assembly + Krylov solve hot path (libnalu_gpu.so, gfx950).
 *
 * One opaque handle == one sierra::nalu::LinearSystem (include/LinearSystem.h:40-168 of
 * the reference) plus the TpetraLinearSolver it owns.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference tree).  All functions
 * return NALU_GPU_OK (0) or an error code; nalu_gpu_last_error() gives the message of the
 * calling thread's last failure.  Pointers named *_dev are device (HBM) pointers, all
 * others are host pointers.  numDof == 1 only (continuity and scalar transport systems).
 *
 * Local node numbering follows TpetraLinearSystem (src/TpetraLinearSystem.C:217-285,
 * 1046-1074): owned nodes sorted by global id, then shared-not-owned sorted by global id,
 * then ghosts ordered by (owner rank, global id).  Row i == owned node i; column j == local
 * node j.
 */
#ifndef NALU_GPU_H
#define NALU_GPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nalu_gpu_linsys nalu_gpu_linsys;

enum {
  NALU_GPU_OK = 0,
  NALU_GPU_EINVAL = 1,      /* bad argument / shapes that do not match the graph */
  NALU_GPU_ESTATE = 2,      /* call out of order (e.g. sumInto before finalize) */
  NALU_GPU_EHIP = 3,        /* a HIP runtime call failed */
  NALU_GPU_EUNSUPPORTED = 4 /* e.g. numDof != 1; mirrors "not supported" throws */
};

/* how the bulk edge kernels resolve the row conflicts of sumInto */
enum {
  NALU_GPU_SCATTER_ROWS = 0,    /* owner-computes row gather: deterministic, no atomics (default) */
  NALU_GPU_SCATTER_COLORED = 1, /* edge-parallel, one launch per edge colour: deterministic */
  NALU_GPU_SCATTER_ATOMIC = 2   /* edge-parallel, fp64 atomics on diag/rhs: NOT bit-reproducible */
};

/* Belos solver names of src/LinearSolverConfig.C:34-55 as the regression inputs use them.  All four run as
 * device-resident loops (scalar recurrences on the device, the host polls a done flag every check_every
 * iterations).  Residual histories against the CPU oracle: cg / gmres <= 1e-10 whatever the summation order
 * of the dot products; biCgStab / tfqmr amplify last-bit differences of the dot products by ~1e6, so they
 * hold 1e-10 against an oracle that sums in the library's own fixed reduction tree (one rank) and 1e-7,
 * with a stop that may move by one iteration, between different rank counts (the all-reduce changes the
 * summation order, as it does between MPI rank counts in the reference: separate golds per np,
 * reg_tests/CTestList.cmake:133-135). */
enum { NALU_GPU_METHOD_CG = 0, NALU_GPU_METHOD_GMRES = 1, NALU_GPU_METHOD_BICGSTAB = 2, NALU_GPU_METHOD_TFQMR = 3 };
enum { NALU_GPU_PRECOND_NONE = 0, NALU_GPU_PRECOND_JACOBI = 1, NALU_GPU_PRECOND_SGS = 2,
       NALU_GPU_PRECOND_CHEBYSHEV = 3,
       NALU_GPU_PRECOND_MT_SGS = 4 /* "mt_sgs" (src/LinearSolverConfig.C:76-80): multicolour SGS */ };
enum { NALU_GPU_PECLET_CLASSIC = 0, NALU_GPU_PECLET_TANH = 1 };

const char *nalu_gpu_last_error(void);
int nalu_gpu_version(void);

/* ---- lifetime: LinearSystem::create (src/LinearSystem.C:90-109), ~EquationSystem
 * (src/EquationSystem.C:76-77).  hip_stream: hipStream_t all work is enqueued on
 * (NULL = the device's default stream, which is also PyTorch's default). */
int nalu_gpu_create(nalu_gpu_linsys **out, int device, int num_dof, void *hip_stream);
int nalu_gpu_destroy(nalu_gpu_linsys *h);

/* How rows of shared-not-owned nodes are handled.
 * OWNER_COMPUTES (default): they are not stored; every rank assembles all entities that touch its
 *   owned rows (cut edges are passed on both sides), loadComplete moves nothing, results do not
 *   depend on the rank count.  A host sumInto that names a shared-not-owned row is an error
 *   (NALU_GPU_EUNSUPPORTED): nothing is dropped silently.
 * EXPORT_SHARED_ROWS: the reference's scheme (src/TpetraLinearSystem.C:1495-1505, 1686-1707): every
 *   entity is assembled on ONE rank (its owner); contributions to rows of shared-not-owned nodes
 *   go to local copies of those rows (sharedNotOwnedMatrix_) and nalu_gpu_load_complete sends
 *   them to the owners, which add them (Tpetra Export, ADD).  The GRAPH still needs every edge
 *   that touches an owned node (the reference exchanges the graph rows at finalize, :700-738);
 *   nalu_gpu_set_edge_ownership then says which edges this rank assembles.  Needs a comm with
 *   exchange_rows.  Call before nalu_gpu_begin_construction. */
enum { NALU_GPU_ASSEMBLY_OWNER_COMPUTES = 0, NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS = 1 };
int nalu_gpu_set_assembly_mode(nalu_gpu_linsys *h, int mode);
/* export mode: locally_owned[e] != 0 for the edges (in graph edge order, all edge blocks
 * registered so far) that this rank assembles (the reference's locally_owned selector,
 * src/AssembleContinuityEdgeSolverAlgorithm.C:115-117); the others only shape the graph.
 * Call after the edge graph builders and before nalu_gpu_finalize; default: all owned. */
int nalu_gpu_set_edge_ownership(nalu_gpu_linsys *h, int64_t n_edges, const unsigned char *locally_owned);

/* ---- graph construction -------------------------------------------------------------
 * beginLinearSystemConstruction (src/TpetraLinearSystem.C:162-302). gids/owner_ranks have
 * n_owned+n_shared+n_ghost entries in local order; the ordering rules above are checked. */
int nalu_gpu_begin_construction(nalu_gpu_linsys *h, int n_owned, int n_shared, int n_ghost,
                                const int64_t *gids, const int *owner_ranks, int my_rank);
/* getDofStatus (src/TpetraLinearSystem.C:157-160 -> getDofStatus_impl :2111-2157, the "simple
 * case": no periodic / non-conformal parts): DofStatus of include/TpetraLinearSystem.h:49-55 for
 * n local nodes.  Valid once begin_construction has been called. */
enum {
  NALU_GPU_DS_NotSet = 0, NALU_GPU_DS_SkippedDOF = 1 << 1, NALU_GPU_DS_OwnedDOF = 1 << 2,
  NALU_GPU_DS_SharedNotOwnedDOF = 1 << 3, NALU_GPU_DS_GhostedDOF = 1 << 4
};
int nalu_gpu_get_dof_status(nalu_gpu_linsys *h, int64_t n_nodes, const int *node_lids, int *status);
/* buildNodeGraph (src/TpetraLinearSystem.C:373-395) */
int nalu_gpu_build_node_graph(nalu_gpu_linsys *h, int64_t n_nodes, const int *node_lids);
/* buildEdgeToNodeGraph (src/TpetraLinearSystem.C:420-425 -> :397-418 -> addConnections
 * :355-371).  edge_nodes[2*e+{0,1}] = local ids of the edge's L and R node.  The edge
 * index e defines the layout of every edge field passed later.  Pass every edge that
 * touches an owned node (the locally owned ones plus the cut edges owned by neighbour
 * ranks): owned rows are computed by their owner, so loadComplete needs no matrix export. */
int nalu_gpu_build_edge_to_node_graph(nalu_gpu_linsys *h, int64_t n_edges, const int *edge_nodes);
/* No counterpart in the reference (STK numbers edges in creation order, and a CPU edge loop does not
 * care).  The edge order a caller should give its edge list -- and with it every edge field --
 * before nalu_gpu_build_edge_to_node_graph so that the row-gather kernels can compute edge ids
 * instead of loading them and the marching kernel applies: "slot-major", i.e. all first upper
 * edges of the nodes (in node order), then all second ones, ...; the upper edges of a node are its
 * edges to higher local ids, in ascending order of the other node.  perm_out[new] = old index.
 * Any order is correct; this one is the fast one on meshes whose node numbering has some
 * regularity (on a lattice: x-, y-, z-edges).  Host-only, no handle needed. */
int nalu_gpu_suggest_edge_order(int64_t n_edges, const int *edge_nodes, int64_t *perm_out);
/* buildElemToNodeGraph (src/TpetraLinearSystem.C:435-440); also serves face/reduced-elem
 * graphs, which differ only in the entity rank the caller iterates. */
int nalu_gpu_build_elem_to_node_graph(nalu_gpu_linsys *h, int64_t n_elems, int nodes_per_elem,
                                      const int *elem_nodes);
/* buildDirichletNodeGraph (include/LinearSystem.h:68-75): default no-op in the reference */
int nalu_gpu_build_dirichlet_node_graph(nalu_gpu_linsys *h, int64_t n_nodes, const int *node_lids);
/* finalizeLinearSystem (src/TpetraLinearSystem.C:1232-1329): sorted CSR, device matrix,
 * vectors, solver workspace. */
int nalu_gpu_finalize(nalu_gpu_linsys *h);

/* ---- views for parity dumps (Tpetra CRS layout, getLocalRowView semantics used by
 * unit_tests/UnitTestTpetra.C:192-201 and writeToFile src/TpetraLinearSystem.C:1901-1975) */
int nalu_gpu_get_sizes(nalu_gpu_linsys *h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
int nalu_gpu_get_graph(nalu_gpu_linsys *h, int64_t *row_ptr /* n_rows+1 */, int *col_idx /* nnz */);
int nalu_gpu_get_matrix_values(nalu_gpu_linsys *h, double *values /* nnz */);
int nalu_gpu_get_rhs(nalu_gpu_linsys *h, double *rhs /* n_rows */);
int nalu_gpu_get_solution(nalu_gpu_linsys *h, double *x /* n_rows */);
/* overwrite matrix values / rhs from host CSR arrays (reads a MatrixMarket dump back in) */
int nalu_gpu_set_matrix_values(nalu_gpu_linsys *h, const double *values /* nnz */);
int nalu_gpu_set_rhs(nalu_gpu_linsys *h, const double *rhs /* n_rows */);

/* Layout diagnostics (no reference counterpart): the device matrix is SELL-256; rows whose
 * k-th column / edge id is `row + constant(slice, k)` take a fast path of the row kernels that
 * computes them instead of loading them.  Tests use this to assert which path a mesh numbering
 * exercises; bench.py reports it. */
typedef struct {
  int64_t n_slices, sell_slots;
  int max_slice_width;
  int64_t rows_computed_columns; /* owned rows whose column indices are computed (SpMV, assembly) */
  int64_t rows_computed_tags;    /* owned rows whose edge ids are computed (row-gather assembly) */
  int assembly_variant;          /* row-gather kernel in force for the continuity operator: 0 row kernel,
                                    1 entries, 2 ordered, 3 burst, 4 marching, 5 lagged marching without
                                    barriers, 6 the lagged kernel with its streamed operands prefetched into
                                    LDS on groups of adjacent lines (default where it applies) */
  int xcd_run;                   /* slices per XCD run of the row-kernel schedule */
  int64_t march_runs;            /* runs of slices the marching assembly kernel walks (0: none) */
  int64_t march_slices;          /* slices in those runs */
  int march_longest;             /* slices in the longest run */
} nalu_gpu_layout_info;
int nalu_gpu_get_layout_info(nalu_gpu_linsys *h, nalu_gpu_layout_info *info);

/* ---- assembly ------------------------------------------------------------------------
 * zeroSystem (src/TpetraLinearSystem.C:1331-1348) */
int nalu_gpu_zero_system(nalu_gpu_linsys *h);
/* sumInto, vector overload (src/TpetraLinearSystem.C:1454-1507 + sum_into_row :1350-1385):
 * host slow path for nodal / BC algorithms. lhs is n_obj x n_obj row-major.  Rows of
 * non-owned nodes are skipped (their owner assembles them).  A column that is not in a row is
 * skipped silently together with every larger column of the call, exactly as the single
 * ascending pass of sum_into_row (:1363-1383) does. */
int nalu_gpu_sum_into(nalu_gpu_linsys *h, int n_obj, const int *node_lids, const double *rhs,
                      const double *lhs);

/* sumInto, Kokkos-view overload (src/TpetraLinearSystem.C:1387-1452), which the reference
 * calls from inside a team loop over entities: here the loop is the launch.  A plan holds the
 * entity -> matrix-slot search of sum_into_row (:1350-1385) for one connectivity
 * (ent_lids[n_ent][n_obj], host, local node ids; edges n_obj = 2, Hex8 elements 8, nodes 1);
 * nalu_gpu_sum_into_batch then adds lhs_dev[n_ent][n_obj][n_obj] (row-major blocks) and
 * rhs_dev[n_ent][n_obj] of every entity.  NALU_GPU_SCATTER_ROWS adds in ascending entity order
 * per value (bit-identical to the sequential entity loop), _COLORED in entity-colour order,
 * _ATOMIC in any order.  Rows of non-owned nodes and columns outside the graph are skipped as
 * in nalu_gpu_sum_into; an entity that lists a node twice is rejected. */
typedef struct nalu_gpu_scatter_plan nalu_gpu_scatter_plan;
int nalu_gpu_scatter_plan_create(nalu_gpu_linsys *h, int64_t n_ent, int n_obj, const int *ent_lids,
                                 nalu_gpu_scatter_plan **out);
int nalu_gpu_scatter_plan_destroy(nalu_gpu_scatter_plan *p);
/* number of entity colours of the COLORED mode (computed on first use) */
int nalu_gpu_scatter_plan_colors(nalu_gpu_scatter_plan *p, int *n_colors);
int nalu_gpu_sum_into_batch(nalu_gpu_linsys *h, nalu_gpu_scatter_plan *p, const double *rhs_dev,
                            const double *lhs_dev, int scatter_mode);

typedef struct { const double *x, *y, *z; } nalu_gpu_vec3; /* SoA device arrays */

/* AssembleContinuityEdgeSolverAlgorithm (src/AssembleContinuityEdgeSolverAlgorithm.C:52-59) */
typedef struct {
  nalu_gpu_vec3 coordinates; /* node */
  nalu_gpu_vec3 dpdx;        /* node, "dpdx" */
  nalu_gpu_vec3 velocity;    /* node, "velocity" or "velocity_rtm" */
  const double *pressure;    /* node */
  const double *density;     /* node, state NP1 */
  nalu_gpu_vec3 edge_area;   /* edge, "edge_area_vector" */
} nalu_gpu_continuity_fields;
/* AssembleContinuityEdgeSolverAlgorithm::execute (:74-209) for every edge of the graph */
int nalu_gpu_assemble_continuity_edges(nalu_gpu_linsys *h, const nalu_gpu_continuity_fields *f_dev,
                                       double dt, double gamma1, double noc_fac,
                                       double interp_together, int scatter_mode);

/* AssembleScalarEdgeSolverAlgorithm (src/AssembleScalarEdgeSolverAlgorithm.C:56-67) */
typedef struct {
  nalu_gpu_vec3 coordinates;
  nalu_gpu_vec3 dqdx;
  nalu_gpu_vec3 velocity;
  const double *scalar_q;        /* state NP1 */
  const double *density;         /* state NP1 */
  const double *diff_flux_coeff;
  nalu_gpu_vec3 edge_area;
  const double *mass_flow_rate;  /* edge */
} nalu_gpu_scalar_fields;
/* realm_.get_alpha_factor / get_alpha_upw_factor / get_upw_factor / primitive_uses_limiter
 * (:102-106) and EquationSystem::create_peclet_function (include/EquationSystem.h:305-322) */
typedef struct {
  double alpha, alpha_upw, ho_upwind;
  int use_limiter;
  int peclet_kind;       /* NALU_GPU_PECLET_* */
  double hybrid_factor;  /* classic: hf */
  double tanh_c1, tanh_c2;
} nalu_gpu_scalar_options;
/* AssembleScalarEdgeSolverAlgorithm::execute (:90-315) */
int nalu_gpu_assemble_scalar_edges(nalu_gpu_linsys *h, const nalu_gpu_scalar_fields *f_dev,
                                   const nalu_gpu_scalar_options *opts, int scatter_mode);
/* AssembleNodeSolverAlgorithm::execute (src/AssembleNodeSolverAlgorithm.C:57-112) with
 * ScalarMassBackwardEulerNodeSuppAlg::node_execute (src/ScalarMassBackwardEulerNodeSuppAlg.C:65-80)
 * over all owned nodes, in bulk. */
int nalu_gpu_assemble_scalar_mass_be(nalu_gpu_linsys *h, const double *q_n_dev,
                                     const double *q_np1_dev, const double *rho_n_dev,
                                     const double *rho_np1_dev, const double *dual_volume_dev,
                                     double dt);
/* applyDirichletBCs (src/TpetraLinearSystem.C:1509-1586): node list on the host, fields on
 * the device indexed by local node id. */
int nalu_gpu_apply_dirichlet(nalu_gpu_linsys *h, int64_t n_nodes, const int *node_lids,
                             const double *solution_dev, const double *bc_values_dev);
/* resetRows (src/TpetraLinearSystem.C:1638-1684) */
int nalu_gpu_reset_rows(nalu_gpu_linsys *h, int64_t n_nodes, const int *node_lids);
/* loadComplete (src/TpetraLinearSystem.C:1686-1707): owner-computes mode has nothing to move;
 * export mode packs the shared-not-owned rows (values + rhs), exchanges them and adds them to
 * the owners' rows in neighbour order (deterministic).  Collective in export mode. */
int nalu_gpu_load_complete(nalu_gpu_linsys *h);

/* ---- solve ---------------------------------------------------------------------------
 * TpetraLinearSolverConfig::load (src/LinearSolverConfig.C:34-107; defaults
 * include/LinearSolverConfig.h:70-80): tolerance 1e-4, max_iterations 50, kspace 50. */
typedef struct {
  int method;         /* NALU_GPU_METHOD_* */
  int precond;        /* NALU_GPU_PRECOND_* */
  double tolerance;   /* "Convergence Tolerance" (final_tolerance on the last outer iter) */
  int max_iterations; /* "Maximum Iterations" */
  int kspace;         /* "Num Blocks"; Maximum Restarts = max(1, max_iterations/kspace) */
  int check_every;    /* host polls the device convergence flag every this many iterations
                         (0 = default 10); does not change results */
} nalu_gpu_solver_config;
typedef struct {
  int iters;              /* solver_->getNumIters() (src/LinearSolver.C:226) */
  double final_resid;     /* ||A x - b||_2 (src/LinearSolver.C:153-178,227) */
  double rhs_norm2;       /* ownedRhs_->norm2() (src/TpetraLinearSystem.C:1756), global */
  double precond_seconds; /* timerPrecond_ (src/LinearSolver.C:195-211) */
  double solve_seconds;
  int history_len;
} nalu_gpu_solve_info;
/* TpetraLinearSystem::solve (src/TpetraLinearSystem.C:1709-1780) -> TpetraLinearSolver::solve
 * (src/LinearSolver.C:181-230).  x0 = 0.  x_out_dev (n_owned+n_shared+n_ghost doubles, may be
 * NULL) receives the solution with shared/ghost entries synchronised (copy_tpetra_to_stk +
 * sync_field).  history[k] = relative residual after iteration k (history[0] = 1): what
 * Belos prints with output_level > 0 (src/LinearSolverConfig.C:56-64). */
int nalu_gpu_solve(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, double *x_out_dev,
                   nalu_gpu_solve_info *info, double *history, int history_cap);

/* checkForNaN + checkForZeroRow (src/TpetraLinearSystem.C:1782-1899) on the owned rows of this
 * rank: rows with a NaN entry, NaN entries of the rhs, rows with sum |a_ij| < 1e-10 (:1862), and
 * the smallest local row of each kind (-1: none). */
typedef struct {
  int64_t lhs_nan_rows, rhs_nan_rows, zero_rows;
  int64_t first_lhs_nan_row, first_rhs_nan_row, first_zero_row;
} nalu_gpu_check_info;
int nalu_gpu_check_system(nalu_gpu_linsys *h, nalu_gpu_check_info *info);
/* realm_.debug() (src/TpetraLinearSystem.C:1716-1721): when on, nalu_gpu_solve runs the two checks
 * first and fails with NALU_GPU_EINVAL ("bad LHS" / "ERROR checkForZeroRow in solve()") instead of
 * solving.  Independently of this switch a zero, missing or non-finite diagonal under a
 * relaxation preconditioner is always an error, never a silent inf.  Either verdict is summed over the
 * ranks of the comm first, so that all ranks of a multi-rank solve return the error together. */
int nalu_gpu_set_debug(nalu_gpu_linsys *h, int enabled);

/* ---- building blocks exposed for measurement and parity (Tpetra call sites:
 * matrix_->apply src/LinearSolver.C:164, update :166, norm2 :172) */
int nalu_gpu_spmv(nalu_gpu_linsys *h, const double *x_dev /* n_cols */, double *y_dev /* n_rows */);
int nalu_gpu_dot(nalu_gpu_linsys *h, const double *a_dev, const double *b_dev, int64_t n, double *out);
int nalu_gpu_axpy(nalu_gpu_linsys *h, double alpha, const double *x_dev, double *y_dev, int64_t n);
int nalu_gpu_norm2(nalu_gpu_linsys *h, const double *a_dev, int64_t n, double *out);

/* The ONE entry point that allocates nothing.  It has no counterpart in the reference: it is for
 * callers that cannot call hipMalloc -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same result, bit for bit, as nalu_gpu_dot on one rank (same
 * kernels, grid and summation order), but it needs no handle: the per-workgroup partial sums
 * live in a static device array of 16,384 doubles (128 KiB, held by every GPU the library runs
 * on until the process ends) and the result is written to out_dev (device memory, no host
 * synchronisation).  n above NALU_GPU_DOT_STATIC_MAX_N is rejected with NALU_GPU_EINVAL.  No
 * all-reduce over ranks.  Calls that overlap on one GPU (different streams or threads) share
 * the array and race: serialise them. */
#define NALU_GPU_DOT_STATIC_MAX_N (INT64_C(1) << 32)
int nalu_gpu_dot_static(int device, void *hip_stream, const double *a_dev, const double *b_dev,
                        int64_t n, double *out_dev);

/* ---- adjacent edge algorithms ---------------------------------
 * ComputeMdotEdgeAlgorithm::execute (src/ComputeMdotEdgeAlgorithm.C:67-174); export mode: only the
 * edges this rank owns are written (the locally_owned selector, :104-106) */
int nalu_gpu_compute_mdot_edges(nalu_gpu_linsys *h, const nalu_gpu_continuity_fields *f_dev,
                                double dt, double gamma1, double noc_fac, double interp_together,
                                double *mdot_dev /* n_edges */);
/* AssembleNodalGradEdgeAlgorithm::execute + driver zeroing
 * (src/AssembleNodalGradEdgeAlgorithm.C:54-118, src/AssembleNodalGradAlgorithmDriver.C:59-123).
 * Owner-computes mode: the complete gradient of the owned nodes (all their edges are local); the
 * shared / ghost entries are left for nalu_gpu_halo_exchange.  Export mode: the reference's edge
 * loop over the edges this rank owns, into owned and shared-not-owned nodes (n_owned + n_shared
 * entries written); the driver's stk::mesh::parallel_sum (:111) is nalu_gpu_parallel_sum(h, NULL, 3, ..). */
int nalu_gpu_nodal_grad_edges(nalu_gpu_linsys *h, nalu_gpu_vec3 edge_area_dev,
                              const double *dual_volume_dev, const double *q_dev,
                              double *gx_dev, double *gy_dev, double *gz_dev);

/* ---- Hex8 geometry preprocessing --------------------------------
 * ComputeGeometryInteriorAlgorithm::execute (src/ComputeGeometryInteriorAlgorithm.C:78-216)
 * for Hex8: dual_nodal_volume (n_cols) from hex_scv_det and edge_area_vector (n_edges, SoA)
 * from hex_scs_det (src/master_element/MasterElementWork.F:1-231, 233-452) with the sign
 * convention of :196-212 (relative to the edge's node 0).  hex_nodes_dev is [n_elem][8]
 * local node ids in the reference's Hex8 node order.  Sums run in ascending element order:
 * deterministic and bit-identical to the element loop.  Element edges that are not
 * registered edges of this system are skipped (owner-computes mode: an edge between two
 * shared-not-owned nodes has no row to be found through; export mode stores those rows).  Two ways
 * to complete the sums over ranks: owner-computes -- pass the elements around every owned node (own
 * + one layer of the neighbours'): owned volumes and all registered edges are complete, the copies'
 * volumes come with nalu_gpu_halo_exchange; reference style (export mode) -- pass the rank's own
 * elements and let nalu_gpu_parallel_sum add the sharers' parts (ComputeGeometryAlgorithmDriver.C:126-133). */
int nalu_gpu_compute_geometry_hex8(nalu_gpu_linsys *h, int64_t n_elem, const int *hex_nodes_dev,
                                   nalu_gpu_vec3 coordinates_dev, double *dual_volume_dev,
                                   double *area_x_dev, double *area_y_dev, double *area_z_dev);

/* ---- nodal updates of the pressure-projection loop --------------
 * field_axpby (src/FieldFunctions.C:24-58): y = alpha x + beta y over n doubles; the
 * pressure update p += dp of src/LowMachEquationSystem.C:676-683 is (1, pTmp, 1, pressure). */
int nalu_gpu_field_axpby(nalu_gpu_linsys *h, int64_t n, double alpha, const double *x_dev,
                         double beta, double *y_dev);
/* LowMachEquationSystem::project_nodal_velocity (src/LowMachEquationSystem.C:772-855):
 * u -= (dt/gamma1)/rho (dpdx_new - dpdx_old) on n nodes; projected_dev (one byte per node,
 * 0 = the node is in notProjectedPart_, :826-828) may be NULL = project every node. */
int nalu_gpu_project_nodal_velocity(nalu_gpu_linsys *h, int64_t n, double dt, double gamma1,
                                    nalu_gpu_vec3 dpdx_new_dev, nalu_gpu_vec3 dpdx_old_dev,
                                    const double *density_dev,
                                    const unsigned char *projected_dev,
                                    nalu_gpu_vec3 velocity_dev);

/* MixtureFractionEquationSystem::update_and_clip (src/MixtureFractionEquationSystem.C:942-1013):
 * z += dz on n nodes (the reference: owned and shared), the un-clipped sum kept in
 * z_unfiltered_dev (mixFracUF_, may be NULL), z clipped to [0 - delta_clip, 1 + delta_clip].
 * clip_info (host, may be NULL = no diagnostics, no host synchronisation): {nodes clipped below,
 * nodes clipped above, smallest and largest un-clipped value (+1e16 / -1e16 when nothing was
 * clipped)} of this rank; the all-reduce of :991-1007 (outputClippingDiag_) is the caller's. */
int nalu_gpu_update_and_clip(nalu_gpu_linsys *h, int64_t n, const double *dz_dev, double *z_dev,
                             double *z_unfiltered_dev, double delta_clip, double clip_info[4]);

/* ---- multi-GPU plumbing (Tpetra Import/Export + all-reduce equivalents) ----------------
 * The library packs/unpacks and reduces on the device; moving bytes between GPUs is the
 * host's job (RCCL through torch.distributed, or ncclSend/Recv from a C++ host).
 * Halo plan: for neighbour p, send_lids[send_offsets[p]..send_offsets[p+1]) are owned rows
 * whose values go to p; received values land in columns n_owned+recv_offsets[p].. in
 * order (shared-not-owned and ghost columns are contiguous per owner by construction). */
int nalu_gpu_set_halo_plan(nalu_gpu_linsys *h, int n_neighbors, const int *neighbor_ranks,
                           const int *send_offsets, const int *send_lids,
                           const int *recv_offsets);
/* The general form (Tpetra Import's permute-to list): recv_lids[i] is the shared / ghost column that
 * the i-th received value lands in.  Needed when the shared-not-owned nodes (sorted by global id
 * by the numbering rules) have several owners, e.g. a 2 x 2 x 2 block partition where a rank has
 * up to seven neighbours.  recv_lids == NULL is nalu_gpu_set_halo_plan. */
int nalu_gpu_set_halo_plan_indexed(nalu_gpu_linsys *h, int n_neighbors, const int *neighbor_ranks,
                                   const int *send_offsets, const int *send_lids,
                                   const int *recv_offsets, const int *recv_lids);
/* the plan as set: pointers into the handle (valid until the next set_halo_plan / destroy) */
int nalu_gpu_get_halo_plan(nalu_gpu_linsys *h, int *n_neighbors, const int **neighbor_ranks,
                           const int **send_offsets, const int **recv_offsets);
typedef struct {
  void *ctx;
  /* in-place sum over ranks of count doubles at the handle's reduce buffer */
  int (*allreduce_sum)(void *ctx, int count, void *hip_stream);
  /* move the packed send buffer to the neighbours' receive buffers (begin may return
   * before completion; end makes hip_stream wait for the received data) */
  int (*exchange_begin)(void *ctx, void *hip_stream);
  int (*exchange_end)(void *ctx, void *hip_stream);
  /* optional (NULL: export mode is refused): neighbour exchange of other buffers than the halo's
   * -- the row export of loadComplete.  For neighbour p (order of the halo plan) send the doubles
   * send_dev[send_off[p] .. send_off[p+1]) and receive recv_dev[recv_off[p] .. recv_off[p+1]);
   * work enqueued on hip_stream afterwards sees the received data. */
  int (*exchange_rows)(void *ctx, const double *send_dev, const int64_t *send_off, double *recv_dev,
                       const int64_t *recv_off, void *hip_stream);
} nalu_gpu_comm;
int nalu_gpu_set_comm(nalu_gpu_linsys *h, const nalu_gpu_comm *comm);
/* A callback that fails (returns non-zero) may first leave a message here; the library appends
 * it to the NALU_GPU_EHIP error it raises, so nalu_gpu_last_error() names the cause (the
 * reference's MPI errors abort with their own text). */
int nalu_gpu_set_callback_error(const char *message);
int nalu_gpu_get_comm_buffers(nalu_gpu_linsys *h, double **send_buf_dev, int64_t *send_count,
                              double **recv_buf_dev, int64_t *recv_count,
                              double **reduce_buf_dev, int *reduce_capacity);
/* owner -> sharer/ghost update of one column-space vector (stk::mesh::copy_owned_to_shared,
 * src/LinearSystem.C:111-116; Tpetra Import inside CrsMatrix::apply) */
int nalu_gpu_halo_exchange(nalu_gpu_linsys *h, double *vec_dev /* n_cols */);

/* stk::mesh::parallel_sum (src/AssembleNodalGradAlgorithmDriver.C:111 for the nodal gradient,
 * src/ComputeGeometryAlgorithmDriver.C:126-133 for dual_nodal_volume and edge_area_vector): every
 * rank that shares an entity -- holds it through elements of its own -- gets the sum of all
 * sharers' values.  As in STK the sum is formed on the entity's owner (own value first, then the
 * sharers' in ascending rank order: deterministic) and sent back, two neighbour exchanges through
 * the comm's exchange_rows; every sharer holds the owner's bits afterwards.
 * plan == NULL: nodal fields (n_owned + n_shared + n_ghost entries each) over the halo plan's own
 * lists; ghost copies contribute nothing and receive the totals like the shared ones.
 * A share plan serves any other entity rank (edges, in the graph's edge order): for every neighbour
 * of the halo plan, in its order (lists may be empty), send_ids = the entities this rank OWNS that
 * the neighbour shares, recv_ids = the entities the neighbour owns that this rank shares, each list
 * in the same order as its counterpart on the neighbour (ascending global id, as STK's comm lists).
 * fields_dev: n_fields (<= 16) device arrays indexed by entity. */
typedef struct nalu_gpu_share_plan nalu_gpu_share_plan;
int nalu_gpu_share_plan_create(nalu_gpu_linsys *h, const int *send_offsets, const int *send_ids,
                               const int *recv_offsets, const int *recv_ids, int64_t n_entities,
                               nalu_gpu_share_plan **out);
int nalu_gpu_share_plan_destroy(nalu_gpu_share_plan *p);
int nalu_gpu_parallel_sum(nalu_gpu_linsys *h, nalu_gpu_share_plan *plan, int n_fields,
                          double *const *fields_dev);

/* ---- timers: EquationSystem's buckets (src/EquationSystem.C:270-292) measured with HIP
 * events on the handle's stream. */
enum {
  NALU_GPU_T_ZERO = 0, NALU_GPU_T_ASSEMBLE_EDGES = 1, NALU_GPU_T_SPMV = 2,
  NALU_GPU_T_SOLVE = 3, NALU_GPU_T_PRECOND = 4, NALU_GPU_T_BC = 5, NALU_GPU_T_COUNT = 6
};
/* enabled: 0 off, 1 every launch is timed, n > 1: the SpMV timer (one launch per Krylov iteration; an event
 * pair costs ~4 us of stream time) times every n-th launch, the others every launch */
int nalu_gpu_set_profiling(nalu_gpu_linsys *h, int enabled);
/* launches seen by a timer while profiling was on (nalu_gpu_get_timer's count = the launches timed) */
int nalu_gpu_get_timer_launches(nalu_gpu_linsys *h, int which, int64_t *launched);
/* Schedule of the exact SGS sweeps (preconditioner "sgs"), valid after the first solve with it.  No counterpart
 * in the reference (Ifpack2 sweeps rows sequentially on the host).  out[0]: 1 = dataflow sweeps (one launch per
 * sweep), -1 = one launch per level (rows wider than the dataflow kernel takes, or NG_SGS_MODE=levels|inplace),
 * -2 = one launch per level because a dataflow sweep gave up waiting in an earlier solve (that solve was redone
 * on the level schedule inside nalu_gpu_solve: same bits, nothing to wait for), 0 = not built yet; out[1]: 1 = lattice tiles, 0 = contiguous row chunks; out[2]: chunks; out[3]: most steps of
 * a chunk; out[4]: widest row (entries one side of the diagonal); out[5]: LDS ring (rows); out[6], out[7]: lines
 * and planes per tile */
int nalu_gpu_get_sgs_schedule(nalu_gpu_linsys *h, int out[8]);
/* synchronises the stream, then returns accumulated device ms and launch count */
int nalu_gpu_get_timer(nalu_gpu_linsys *h, int which, double *total_ms, int64_t *count);
int nalu_gpu_reset_timers(nalu_gpu_linsys *h);
int nalu_gpu_synchronize(nalu_gpu_linsys *h);

#ifdef __cplusplus
}
#endif
#endif
