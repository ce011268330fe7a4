/*
 * nalu_oracle.h -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A deliberately boring, single-threaded, fp64 restatement in plain C of the
 * reference's (gantech/Nalu) edge-based assembly + Krylov hot path.  Every
 * function cites the reference file:line it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (nalu_amd/, include/nalu_gpu.h) never links or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - graph + sumInto + loadComplete : pinned by unit_tests/UnitTestTpetra.C:86-158 gold,
 *     unit_tests/UnitTestLocalGraphArrays.C:9-105.
 *   - Peclet functions               : pinned by unit_tests/UnitTestPecletFunction.C:21-87.
 *   - Hex8 SCS/SCV geometry          : pinned by oracle/_ref (the reference's own F77
 *     kernels compiled in place) and unit_tests/UnitTestHexSCVDeterminant.C:67-69,125-129.
 *   - the two edge kernels           : PARITY UNPINNED at unit level (no reference test
 *     or fixture covers them); line-by-line restatement + analytic properties.
 *   - Belos CG / GMRES / Ifpack2 Jacobi: PARITY UNPINNED (Trilinos is not in the tree);
 *     restated from the published algorithms, cross-checked against SciPy in tests.
 */
#ifndef NALU_ORACLE_H
#define NALU_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ mesh */
/* STK/Ioss "generated:" numbering: id = 1+i+nxn*j+nxn*nyn*k */
void orc_box_coords(int nxn, int nyn, int nzn, double lx, double ly, double lz,
                    double *xyz /* [N][3] */);
/* LCG of unit_tests/UnitTestUtils.C:41-75 applied to interior nodes in id order */
void orc_box_perturb(int nxn, int nyn, int nzn, double *xyz, uint32_t seed, double amp);
/* hexes in id order, node order 000,100,110,010,001,101,111,011 (0-based node idx) */
void orc_box_hexes(int nxn, int nyn, int nzn, int *hex_nodes /* [nelem][8] */);
/* lattice edges oriented low id -> high id, sorted by (L,R); returns count */
int64_t orc_box_edges(int nxn, int nyn, int nzn, int *edge_nodes /* [E][2] */);

/* -------------------------------------------------------------- geometry */
/* src/master_element/CalculateQuadArea.F:1-72 */
void orc_quad_area_by_triangle_facets(const double ac[4][3], double area[3]);
/* src/master_element/CalculatePolyhedralVolume.F:1-75 + CalculateHexVolume.F:1-100 */
double orc_hex_volume_by_triangle_facets(const double vc[8][3]);
/* src/master_element/MasterElementWork.F:1-231 (hex_scs_det), nelem = 1 */
void orc_hex_scs_det(const double cordel[8][3], double area_vec[12][3]);
/* src/master_element/MasterElementWork.F:233-457 (hex_scv_det), nelem = 1 */
void orc_hex_scv_det(const double cordel[8][3], double vol[8]);
/* src/ComputeGeometryInteriorAlgorithm.C:78-216: dual_nodal_volume, edge_area_vector.
 * edge_nodes must be sorted by (L,R) with L<R (as orc_box_edges gives). */
int orc_compute_geometry(int64_t nelem, const int *hex_nodes, const double *xyz,
                         int64_t nnodes, int64_t nedges, const int *edge_nodes,
                         double *dual_volume /* [N] */, double *edge_area /* [E][3] */);

/* ------------------------------------------------- LocalGraphArrays (CSR) */
/* include/LinearSolver.h:66-124 */
typedef struct {
  int64_t nrows;
  size_t *rowPointers; /* nrows+1 */
  int *colIndices;     /* nnz, INVALID-filled then sorted-inserted */
} orc_graph;
#define ORC_INVALID 2147483647
orc_graph *orc_graph_create(int64_t nrows, const size_t *rowLengths);
void orc_graph_destroy(orc_graph *g);
size_t orc_graph_row_length(const orc_graph *g, int64_t row);
void orc_graph_insert_indices(orc_graph *g, int64_t localRow, size_t numInds,
                              const int *inds, int numDof);
/* src/TpetraLinearSystem.C:1169-1201 */
void orc_graph_remove_invalid(orc_graph *g);

/* ------------------------------------------------------- linear system */
typedef struct {
  int numDof;
  int nOwnedNodes, nSharedNodes; /* local node numbering: owned, then shared-not-owned */
  int maxOwnedRowId, maxSharedNotOwnedRowId;
  orc_graph *ownedGraph, *sharedGraph;
  double *ownedVals, *sharedVals;
  double *ownedRhs, *sharedRhs;
  double *sln;
  /* connections_ of src/TpetraLinearSystem.C:333-371 */
  int **conn;
  int *connLen, *connCap;
} orc_linsys;

/* beginLinearSystemConstruction (src/TpetraLinearSystem.C:162-302); nodes are already
 * numbered owned-sorted-by-gid then shared-sorted-by-gid by the caller. */
orc_linsys *orc_linsys_create(int numDof, int nOwnedNodes, int nSharedNodes);
void orc_linsys_destroy(orc_linsys *ls);
/* optional global ids per local node for the id comparison in addConnections (:365);
 * NULL (default) means local-id order == global-id order (single rank). */
void orc_set_gids(const int64_t *gids);
/* addConnections (:355-371) for n_ent entities of npe nodes each (local node ids;
 * ids >= nOwned+nShared are ghosts and are skipped as rows). */
void orc_add_connections(orc_linsys *ls, int64_t n_ent, int npe, const int *ent_nodes);
/* finalizeLinearSystem (:1232-1329), single-rank part (no neighbour exchange). */
void orc_finalize(orc_linsys *ls);
/* zeroSystem (:1331-1348) */
void orc_zero_system(orc_linsys *ls);
/* sumInto vector overload (:1454-1507) + sum_into_row (:1350-1385).
 * node_lids are local node ids; col id = row id = numDof*node_lid (+d). */
void orc_sum_into(orc_linsys *ls, int n_obj, const int *node_lids, const double *rhs,
                  const double *lhs);
/* applyDirichletBCs (:1509-1586) for a node list; solution/bcValues indexed by node */
void orc_apply_dirichlet(orc_linsys *ls, int64_t n, const int *node_lids,
                         const double *solution, const double *bcValues,
                         unsigned beginPos, unsigned endPos);
/* resetRows (:1638-1684) */
void orc_reset_rows(orc_linsys *ls, int64_t n, const int *node_lids, unsigned beginPos,
                    unsigned endPos);

/* --------------------------------------------------------- edge kernels */
/* src/PecletFunction.C:70-75 and :105-109 */
double orc_peclet_classic(double A, double hf, double pe);
double orc_peclet_tanh(double c1, double c2, double x);
/* src/AssembleScalarEdgeSolverAlgorithm.C:320-329 */
double orc_van_leer(double dqm, double dqp, double small);

/* src/AssembleContinuityEdgeSolverAlgorithm.C:74-209. All nodal arrays AoS as STK. */
/* CPU-baseline (OpenMP) build: per-thread owner-computes edge lists, built once per edge list
 * (set-up, like the mesh partition of an MPI run); a no-op in the single-thread builds */
void orc_omp_prepare_edge_buckets(const orc_linsys *ls, int64_t nedges, const int *edge_nodes);
void orc_assemble_continuity_edges(orc_linsys *ls, int64_t nedges, const int *edge_nodes,
                                   const double *edge_area, const double *xyz,
                                   const double *dpdx, const double *vel,
                                   const double *pressure, const double *density,
                                   double dt, double gamma1, double nocFac,
                                   double interpTogether);
/* per-edge outputs only (lhs[4], rhs[2]) -- for kernel-level comparison */
void orc_continuity_edge_coeffs(int64_t nedges, const int *edge_nodes,
                                const double *edge_area, const double *xyz,
                                const double *dpdx, const double *vel,
                                const double *pressure, const double *density, double dt,
                                double gamma1, double nocFac, double interpTogether,
                                double *lhs /* [E][4] */, double *rhs /* [E][2] */);

typedef struct {
  double alpha, alphaUpw, hoUpwind;
  int useLimiter;
  int pecletKind; /* 0 classic, 1 tanh */
  double pecA, pecHf, pecC1, pecC2;
} orc_scalar_opts;
/* src/AssembleScalarEdgeSolverAlgorithm.C:90-315 */
void orc_assemble_scalar_edges(orc_linsys *ls, int64_t nedges, const int *edge_nodes,
                               const double *edge_area, const double *mdot,
                               const double *xyz, const double *dqdx, const double *vel,
                               const double *q, const double *density,
                               const double *diffFluxCoeff, const orc_scalar_opts *o);
void orc_scalar_edge_coeffs(int64_t nedges, const int *edge_nodes, const double *edge_area,
                            const double *mdot, const double *xyz, const double *dqdx,
                            const double *vel, const double *q, const double *density,
                            const double *diffFluxCoeff, const orc_scalar_opts *o,
                            double *lhs, double *rhs);
/* src/ComputeMdotEdgeAlgorithm.C:67-174 */
void orc_compute_mdot_edges(int64_t nedges, const int *edge_nodes, const double *edge_area,
                            const double *xyz, const double *dpdx, const double *vel,
                            const double *pressure, const double *density, double dt,
                            double gamma1, double nocFac, double interpTogether,
                            double *mdot);
/* src/AssembleNodalGradEdgeAlgorithm.C:54-118 (dqdx must be zeroed by the caller, as the
 * driver does, src/AssembleNodalGradAlgorithmDriver.C:59-123) */
void orc_nodal_grad_edges(int64_t nedges, const int *edge_nodes, const double *edge_area,
                          const double *dual_volume, const double *q, double *dqdx);
/* src/AssembleNodeSolverAlgorithm.C:57-112 + src/ScalarMassBackwardEulerNodeSuppAlg.C:65-80 */
void orc_assemble_scalar_mass_be(orc_linsys *ls, int64_t nnodes, const double *qN,
                                 const double *qNp1, const double *rhoN,
                                 const double *rhoNp1, const double *dualVolume, double dt);

/* src/FieldFunctions.C:24-58 */
void orc_field_axpby(int64_t n, double alpha, const double *x, double beta, double *y);
/* src/LowMachEquationSystem.C:772-855 (the projection loop :842-853) */
void orc_project_nodal_velocity(int64_t nnodes, double dt, double gamma1, const double *dpdx_new,
                                const double *dpdx_old, const double *density,
                                const unsigned char *projected, double *velocity);
void orc_update_and_clip(int64_t nnodes, const double *zTmp, double *mixFrac, double *mixFracUF,
                         double deltaZ, double *info);

/* -------------------------------------------------------------- solver */
/* y = A x on the owned CSR (Tpetra CrsMatrix::apply call site src/LinearSolver.C:164) */
void orc_spmv(const orc_linsys *ls, const double *x, double *y);
typedef struct {
  int method;  /* 0 = cg, 1 = gmres */
  int precond; /* 0 = none, 1 = jacobi (src/LinearSolverConfig.C:81-85), 2 = sgs (:71-75),
                  3 = chebyshev (not a reference option: see cheb_apply in nalu_oracle.c),
                  4 = mt_sgs (:76-80), multicolour SGS: see mc_sgs_apply in nalu_oracle.c */
  double tolerance;
  int max_iterations;
  int kspace;
} orc_solver_config;
/* TpetraLinearSolver::solve (src/LinearSolver.C:181-230): x0 = 0, right preconditioning,
 * relative-to-initial-residual test, ICGS (2 classical GS passes), restart = kspace,
 * Maximum Restarts = max(1,max_it/kspace) (src/LinearSolverConfig.C:54-69).
 * history[k] = relative residual after iteration k (history[0] = 1), up to history_cap.
 * Returns iterations; *finalResidNrm = ||A x - b||_2 (:153-178,227). */
int orc_solve(orc_linsys *ls, const orc_solver_config *cfg, double *finalResidNrm,
              double *history, int history_cap, int *history_len);
double orc_norm2(const double *v, int64_t n);
/* summation order of the solver's dot products and norms: 0 (default) sequential, 1 the fixed reduction
 * tree the device library uses (see nalu_oracle.c) -- both are restatements of "a deterministic dot product" */
void orc_set_dot_tree(int on);
double orc_dot(const double *a, const double *b, int64_t n);
/* greedy colouring in row order used by precond 4; color has nrows entries; returns #colours */
int orc_greedy_colors(const orc_linsys *ls, int *color);

#ifdef __cplusplus
}
#endif
#endif
