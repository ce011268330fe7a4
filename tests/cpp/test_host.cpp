// test_host.cpp -- parity tests of the C++ host layer (include/GpuLinearSystem.h) over the
// C ABI, written the way the reference's unit tests are (unit_tests/UnitTestTpetra.C:86-251)
// and checked against the CPU oracle (oracle/nalu_oracle.h -- test infrastructure; linked
// here, never by the product).  Run by tests/test_cpp_host.py:
//   test_host.bin             all tests (needs a GPU)
//   test_host.bin --host-only the tests that touch no device
#include "GpuLinearSystem.h"
#include "RcclComm.h"
#include "nalu_oracle.h"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>

using namespace sierra::nalu::gpu;

static int g_fail = 0, g_checks = 0;
#define EXPECT_TRUE(c)                                                                      \
  do {                                                                                      \
    ++g_checks;                                                                             \
    if (!(c)) { ++g_fail; std::printf("  FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); }      \
  } while (0)
#define EXPECT_EQ(a, b)                                                                     \
  do {                                                                                      \
    ++g_checks;                                                                             \
    if (!((a) == (b))) {                                                                    \
      ++g_fail;                                                                             \
      std::ostringstream os_;                                                               \
      os_ << (a) << " vs " << (b);                                                          \
      std::printf("  FAIL %s:%d  %s == %s (%s)\n", __FILE__, __LINE__, #a, #b, os_.str().c_str()); \
    }                                                                                       \
  } while (0)
#define EXPECT_NEAR(a, b, tol)                                                              \
  do {                                                                                      \
    ++g_checks;                                                                             \
    if (!(std::fabs((a) - (b)) <= (tol))) {                                                 \
      ++g_fail;                                                                             \
      std::printf("  FAIL %s:%d  |%s - %s| = %.3e > %.1e\n", __FILE__, __LINE__, #a, #b,    \
                  std::fabs((a) - (b)), (double)(tol));                                     \
    }                                                                                       \
  } while (0)
#define EXPECT_THROW(stmt)                                                                  \
  do {                                                                                      \
    ++g_checks;                                                                             \
    bool thrown_ = false;                                                                   \
    try { stmt; } catch (const std::runtime_error &) { thrown_ = true; }                    \
    if (!thrown_) { ++g_fail; std::printf("  FAIL %s:%d  no std::runtime_error from %s\n", __FILE__, __LINE__, #stmt); } \
  } while (0)
#define HIP(x)                                                                              \
  do {                                                                                      \
    hipError_t e_ = (x);                                                                    \
    if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } \
  } while (0)

// device copy of a host array (the caller of the library owns field memory)
struct DevBuf {
  double *p = nullptr;
  size_t n = 0;
  DevBuf() {}
  explicit DevBuf(size_t n_) : n(n_) { HIP(hipMalloc((void **)&p, std::max<size_t>(n, 1) * 8)); HIP(hipMemset(p, 0, std::max<size_t>(n, 1) * 8)); }
  explicit DevBuf(const std::vector<double> &h) : n(h.size())
  {
    HIP(hipMalloc((void **)&p, std::max<size_t>(n, 1) * 8));
    if (n) HIP(hipMemcpy(p, h.data(), n * 8, hipMemcpyHostToDevice));
  }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  std::vector<double> host() const
  {
    std::vector<double> h(n);
    if (n) HIP(hipMemcpy(h.data(), p, n * 8, hipMemcpyDeviceToHost));
    return h;
  }
};

// AoS [n][3] -> SoA 3 x n on the device
struct DevVec3 {
  DevBuf buf;
  size_t n;
  explicit DevVec3(const std::vector<double> &aos) : buf(soa(aos)), n(aos.size() / 3) {}
  static std::vector<double> soa(const std::vector<double> &aos)
  {
    const size_t n = aos.size() / 3;
    std::vector<double> s(3 * n);
    for (size_t i = 0; i < n; ++i)
      for (int j = 0; j < 3; ++j) s[j * n + i] = aos[3 * i + j];
    return s;
  }
  nalu_gpu_vec3 v() const { return nalu_gpu_vec3{buf.p, buf.p + n, buf.p + 2 * n}; }
};

// ---------------------------------------------------------------------------------------
// synthetic box in the oracle's (reference's) AoS layout: SURVEY.md 8(d)
struct Box {
  int n;
  int64_t N, E, nelem;
  std::vector<double> xyz, area, dualVol, vel, p, rho, dpdx, gamma, q, dqdx, mdot;
  std::vector<int> edges, hexes;
  explicit Box(int n_, bool perturbed = true) : n(n_)
  {
    N = (int64_t)n * n * n;
    nelem = (int64_t)(n - 1) * (n - 1) * (n - 1);
    xyz.resize(3 * N);
    orc_box_coords(n, n, n, 1.0, 1.0, 1.0, xyz.data());
    if (perturbed) orc_box_perturb(n, n, n, xyz.data(), 1u, 0.125 / (n - 1));
    hexes.resize(8 * nelem);
    orc_box_hexes(n, n, n, hexes.data());
    edges.resize(2 * 3 * (size_t)N);
    E = orc_box_edges(n, n, n, edges.data());
    edges.resize(2 * E);
    area.resize(3 * E);
    dualVol.resize(N);
    orc_compute_geometry(nelem, hexes.data(), xyz.data(), N, E, edges.data(), dualVol.data(), area.data());
    vel.resize(3 * N); p.resize(N); rho.resize(N); gamma.assign(N, 0.2); q.resize(N);
    const double pi = 3.14159265358979323846;
    for (int64_t i = 0; i < N; ++i) {
      const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
      vel[3 * i] = std::sin(pi * x) * std::cos(pi * y);
      vel[3 * i + 1] = -std::cos(pi * x) * std::sin(pi * y);
      vel[3 * i + 2] = 0.0;
      rho[i] = 1.0 + 0.1 * std::sin(2 * pi * x);
      p[i] = 0.3 + 0.7 * x - 0.4 * y + 0.2 * z + 0.5 * (0.9 * x * x - 0.6 * y * y + 0.35 * z * z) + 0.25 * x * y - 0.15 * y * z;
      q[i] = 0.1 + 0.5 * x * y + 0.25 * z * z - 0.3 * x;
    }
    dpdx.assign(3 * N, 0.0);
    orc_nodal_grad_edges(E, edges.data(), area.data(), dualVol.data(), p.data(), dpdx.data());
    dqdx.assign(3 * N, 0.0);
    orc_nodal_grad_edges(E, edges.data(), area.data(), dualVol.data(), q.data(), dqdx.data());
    mdot.resize(E);
    orc_compute_mdot_edges(E, edges.data(), area.data(), xyz.data(), dpdx.data(), vel.data(), p.data(), rho.data(),
                           0.1, 1.0, 1.0, 1.0, mdot.data());
  }
  Realm realm() const
  {
    Realm r;
    r.nOwned = (int)N;
    r.naluGlobalId.resize(N);
    for (int64_t i = 0; i < N; ++i) r.naluGlobalId[i] = i + 1;
    r.l2Scaling_ = 1.0 / std::sqrt((double)N);  // src/Realm.C:2965-2969
    return r;
  }
};

static double maxRelDiff(const std::vector<double> &a, const double *b, size_t n)
{
  double scale = 0.0, d = 0.0;
  for (size_t i = 0; i < n; ++i) {
    scale = std::max(scale, std::fabs(b[i]));
    d = std::max(d, std::fabs(a[i] - b[i]));
  }
  return d / std::max(scale, 1e-300);
}

static void expectSameSystem(GpuLinearSystem &ls, const orc_linsys *o, double tol, bool bitExact)
{
  std::vector<int64_t> rp;
  std::vector<int> ci;
  std::vector<double> v, b;
  ls.getGraph(rp, ci);
  ls.getMatrixValues(v);
  ls.getRhs(b);
  const int64_t n = o->ownedGraph->nrows;
  EXPECT_EQ(ls.numRows(), n);
  EXPECT_EQ((size_t)ls.numNonZeros(), o->ownedGraph->rowPointers[n]);
  bool sameGraph = (size_t)ls.numNonZeros() == o->ownedGraph->rowPointers[n];
  for (int64_t i = 0; i <= n && sameGraph; ++i) sameGraph = (size_t)rp[i] == o->ownedGraph->rowPointers[i];
  for (size_t k = 0; k < ci.size() && sameGraph; ++k) sameGraph = ci[k] == o->ownedGraph->colIndices[k];
  EXPECT_TRUE(sameGraph);
  if (!sameGraph) return;
  if (bitExact) {
    EXPECT_TRUE(std::memcmp(v.data(), o->ownedVals, v.size() * 8) == 0);
    EXPECT_TRUE(std::memcmp(b.data(), o->ownedRhs, b.size() * 8) == 0);
  }
  EXPECT_TRUE(maxRelDiff(v, o->ownedVals, v.size()) <= tol);  // 1e-12: north_star's bar for matrix/rhs
  EXPECT_TRUE(maxRelDiff(b, o->ownedRhs, b.size()) <= tol);
}

// ---------------------------------------------------------------------------------------
// unit_tests/UnitTestTpetra.C:86-114 (data)
static const double elemVals[8][8] = {
  {2.0, -0.8, -0.7, -0.6, -0.5, -0.4, -0.3, -0.2}, {-0.8, 2.0, -0.8, -0.7, -0.6, -0.5, -0.4, -0.3},
  {-0.7, -0.8, 2.0, -0.8, -0.7, -0.6, -0.5, -0.4}, {-0.6, -0.7, -0.8, 2.0, -0.8, -0.7, -0.6, -0.5},
  {-0.5, -0.6, -0.7, -0.8, 2.0, -0.8, -0.7, -0.6}, {-0.4, -0.5, -0.6, -0.7, -0.8, 2.0, -0.8, -0.7},
  {-0.3, -0.4, -0.5, -0.6, -0.7, -0.8, 2.0, -0.8}, {-0.2, -0.3, -0.4, -0.5, -0.6, -0.7, -0.8, 2.0}};
static const double lhsVals[12][12] = {
  {2.0, -0.8, -0.6, -0.7, -0.5, -0.4, -0.2, -0.3, 0.0, 0.0, 0.0, 0.0},
  {-0.8, 2.0, -0.7, -0.8, -0.6, -0.5, -0.3, -0.4, 0.0, 0.0, 0.0, 0.0},
  {-0.6, -0.7, 2.0, -0.8, -0.8, -0.7, -0.5, -0.6, 0.0, 0.0, 0.0, 0.0},
  {-0.7, -0.8, -0.8, 2.0, -0.7, -0.6, -0.4, -0.5, 0.0, 0.0, 0.0, 0.0},
  {-0.5, -0.6, -0.8, -0.7, 4.0, -1.6, -1.2, -1.4, -0.5, -0.4, -0.2, -0.3},
  {-0.4, -0.5, -0.7, -0.6, -1.6, 4.0, -1.4, -1.6, -0.6, -0.5, -0.3, -0.4},
  {-0.2, -0.3, -0.5, -0.4, -1.2, -1.4, 4.0, -1.6, -0.8, -0.7, -0.5, -0.6},
  {-0.3, -0.4, -0.6, -0.5, -1.4, -1.6, -1.6, 4.0, -0.7, -0.6, -0.4, -0.5},
  {0.0, 0.0, 0.0, 0.0, -0.5, -0.6, -0.8, -0.7, 2.0, -0.8, -0.6, -0.7},
  {0.0, 0.0, 0.0, 0.0, -0.4, -0.5, -0.7, -0.6, -0.8, 2.0, -0.7, -0.8},
  {0.0, 0.0, 0.0, 0.0, -0.2, -0.3, -0.5, -0.4, -0.6, -0.7, 2.0, -0.8},
  {0.0, 0.0, 0.0, 0.0, -0.3, -0.4, -0.6, -0.5, -0.7, -0.8, -0.8, 2.0}};

// the TestKernel + AssembleElemSolverAlgorithm pair of that test: one 8x8 block per element
// through apply_coeff
class TestElemAlgorithm : public SolverAlgorithm
{
public:
  TestElemAlgorithm(GpuLinearSystem &ls, const std::vector<int> &elemNodes) : SolverAlgorithm(ls), elemNodes_(elemNodes), numCallsToExecute(0) {}
  void initialize_connectivity() { linsys_.buildElemToNodeGraph(elemNodes_, 8); }
  void execute()
  {
    std::vector<double> lhs(64), rhs(8, 0.0), scratchVals;
    std::vector<int> scratchIds;
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j) lhs[8 * i + j] = elemVals[i][j];
    for (size_t e = 0; e < elemNodes_.size() / 8; ++e) {
      std::vector<Entity> nodes(elemNodes_.begin() + 8 * e, elemNodes_.begin() + 8 * e + 8);
      apply_coeff(nodes, scratchIds, scratchVals, rhs, lhs, "TestKernel");
      ++numCallsToExecute;
    }
  }
  std::vector<int> elemNodes_;
  unsigned numCallsToExecute;
};

static void test_Tpetra_basic()
{
  // generated:1x1x2 (np = 1): elem 1 nodes 1 2 4 3 5 6 8 7, elem 2 nodes 5 6 8 7 9 10 12 11
  std::vector<int> hexes(16);
  orc_box_hexes(2, 2, 3, hexes.data());
  const int gold1[8] = {1, 2, 4, 3, 5, 6, 8, 7};
  for (int i = 0; i < 8; ++i) EXPECT_EQ(hexes[i] + 1, gold1[i]);
  Realm realm;
  realm.nOwned = 12;
  for (int i = 0; i < 12; ++i) realm.naluGlobalId.push_back(i + 1);
  LinearSolverConfig cfg;
  EquationSystem eqsys(realm, "TestEqSys", &cfg);
  TestElemAlgorithm *alg = new TestElemAlgorithm(*eqsys.linsys_, hexes);
  eqsys.solverAlgDriver_->solverAlgorithmMap_["test"] = alg;

  eqsys.linsys_->buildElemToNodeGraph(alg->elemNodes_, 8);
  eqsys.linsys_->finalizeLinearSystem();

  // verify_graph_for_2_hex8_mesh
  const unsigned goldRowLengths[12] = {8, 8, 8, 8, 12, 12, 12, 12, 8, 8, 8, 8};
  std::vector<int64_t> rp;
  std::vector<int> ci;
  eqsys.linsys_->getGraph(rp, ci);
  EXPECT_EQ(eqsys.linsys_->numRows(), 12);
  for (int r = 0; r < 12; ++r) EXPECT_EQ((unsigned)(rp[r + 1] - rp[r]), goldRowLengths[r]);

  eqsys.linsys_->zeroSystem();
  eqsys.solverAlgDriver_->execute();
  eqsys.linsys_->loadComplete();
  EXPECT_EQ(alg->numCallsToExecute, 2u);

  // verify_matrix_for_2_hex8_mesh
  std::vector<double> vals;
  eqsys.linsys_->getMatrixValues(vals);
  for (int r = 0; r < 12; ++r)
    for (int64_t k = rp[r]; k < rp[r + 1]; ++k) EXPECT_NEAR(lhsVals[r][ci[k]], vals[k], 1.e-9);

  // the same gold matrix through the bulk device sumInto (view overload, one call for both
  // elements), in the three scatter modes
  std::vector<double> blocks(2 * 64), zeros(16, 0.0);
  for (int e = 0; e < 2; ++e)
    for (int i = 0; i < 64; ++i) blocks[e * 64 + i] = elemVals[i / 8][i % 8];
  DevBuf dBlocks(blocks), dZeros(zeros);
  nalu_gpu_scatter_plan *plan = eqsys.linsys_->createScatterPlan(alg->elemNodes_, 8);
  for (int mode = 0; mode < 3; ++mode) {
    eqsys.linsys_->zeroSystem();
    eqsys.linsys_->sumIntoBatch(plan, dZeros.p, dBlocks.p, mode);
    eqsys.linsys_->loadComplete();
    eqsys.linsys_->getMatrixValues(vals);
    for (int r = 0; r < 12; ++r)
      for (int64_t k = rp[r]; k < rp[r + 1]; ++k) EXPECT_NEAR(lhsVals[r][ci[k]], vals[k], 1.e-9);
  }
  eqsys.linsys_->destroyScatterPlan(plan);
}

// ---------------------------------------------------------------------------------------
static void test_continuity_edge_assemble_and_solve(int n, const char *precond, const char *method, int maxIt)
{
  Box box(n);
  const double dt = 0.1, gamma1 = 1.0, noc = 1.0, interp = 1.0, pRef = 0.25;
  const int pinNode = 0;

  // oracle: the reference's loop order
  orc_linsys *o = orc_linsys_create(1, (int)box.N, 0);
  orc_add_connections(o, box.E, 2, box.edges.data());
  orc_finalize(o);
  orc_zero_system(o);
  orc_assemble_continuity_edges(o, box.E, box.edges.data(), box.area.data(), box.xyz.data(), box.dpdx.data(),
                                box.vel.data(), box.p.data(), box.rho.data(), dt, gamma1, noc, interp);
  orc_reset_rows(o, 1, &pinNode, 0, 1);
  const double one = 1.0, pinRhs = pRef - box.p[pinNode];
  orc_sum_into(o, 1, &pinNode, &pinRhs, &one);

  // GPU, through the reference-shaped host layer
  Realm realm = box.realm();
  LinearSolverConfig cfg;
  std::map<std::string, std::string> node;
  node["name"] = "solve_cont"; node["method"] = method; node["preconditioner"] = precond;
  node["tolerance"] = "1e-10"; node["max_iterations"] = std::to_string(maxIt); node["kspace"] = "30";
  cfg.load(node);
  DevVec3 xyz(box.xyz), dpdx(box.dpdx), vel(box.vel), area(box.area);
  DevBuf p(box.p), rho(box.rho), dp((size_t)box.N);
  nalu_gpu_continuity_fields f = {xyz.v(), dpdx.v(), vel.v(), p.p, rho.p, area.v()};

  EquationSystem eqsys(realm, "ContinuityEQS", &cfg);
  AssembleContinuityEdgeSolverAlgorithm *alg = new AssembleContinuityEdgeSolverAlgorithm(*eqsys.linsys_, box.edges, f);
  alg->dt_ = dt; alg->gamma1_ = gamma1; alg->nocFac_ = noc; alg->interpTogether_ = interp;
  eqsys.solverAlgDriver_->solverAlgMap_[INTERIOR] = alg;
  eqsys.solverAlgDriver_->solverConstraintAlgMap_[REF_PRESSURE] =
    new FixPressureAtNodeAlgorithm(*eqsys.linsys_, true, pinNode, pRef, box.p[pinNode]);
  eqsys.initialize();
  eqsys.deviceTimers_ = true;
  eqsys.assemble_and_solve(dp.p);

  // edges are (L,R)-sorted: the row sums run in the oracle's order -> bit-identical
  expectSameSystem(*eqsys.linsys_, o, 1e-12, true);

  orc_solver_config oc = {std::strcmp(method, "cg") == 0 ? 0 : 1,
                          std::strcmp(precond, "jacobi") == 0 ? 1 : std::strcmp(precond, "sgs") == 0 ? 2 : 3,
                          1e-10, maxIt, 30};
  std::vector<double> hist((size_t)maxIt + 2);
  int histLen = 0;
  double finalResid = 0.0;
  const int iters = orc_solve(o, &oc, &finalResid, hist.data(), (int)hist.size(), &histLen);
  EXPECT_EQ(eqsys.linsys_->linearSolveIterations(), iters);
  const std::vector<double> &gh = eqsys.linsys_->residualHistory();
  EXPECT_EQ((int)gh.size(), histLen);
  double worst = 0.0;
  for (int k = 0; k < histLen && k < (int)gh.size(); ++k) worst = std::max(worst, std::fabs(gh[k] - hist[k]));
  EXPECT_TRUE(worst <= 1e-10);  // north_star: residual history (relative to r0, history[0] = 1) within 1e-10
  const std::vector<double> x = dp.host();
  EXPECT_TRUE(maxRelDiff(x, o->sln, (size_t)box.N) <= 1e-10);
  EXPECT_NEAR(eqsys.linsys_->linearResidual(), finalResid, 1e-10 * std::max(1.0, finalResid));
  EXPECT_NEAR(eqsys.linsys_->nonLinearResidual(), realm.l2Scaling_ * orc_norm2(o->ownedRhs, box.N), 1e-12);
  EXPECT_NEAR(eqsys.linsys_->scaledNonLinearResidual(), 1.0, 1e-15);  // first solve of the time step
  EXPECT_TRUE(eqsys.timerAssemble_ > 0.0 && eqsys.timerSolve_ > 0.0);
  EXPECT_EQ(eqsys.numberLinearIterations_, (long long)iters);
  std::printf("    n=%d %s/%s: %d iterations, history max abs diff %.2e\n", n, method, precond, iters, worst);
  orc_linsys_destroy(o);
}

// scalar transport: edge advection-diffusion + backward-Euler mass + Dirichlet rows, GMRES
static void test_scalar_edge_assemble_and_solve(int n)
{
  Box box(n);
  const double dt = 0.05;
  std::vector<double> qN(box.q), rhoN(box.rho), bc((size_t)box.N, 0.75);
  for (int64_t i = 0; i < box.N; ++i) qN[i] = box.q[i] - 0.01 * std::sin(3.0 * box.xyz[3 * i]);
  EntityVector wall, all((size_t)box.N);
  for (int i = 0; i < n * n; ++i) wall.push_back(i);  // the z = 0 face
  for (int64_t i = 0; i < box.N; ++i) all[(size_t)i] = (int)i;
  orc_scalar_opts oo = {0.0, 1.0, 1.0, 1, 0, 5.0, 1.0, 2.0, 4.0};

  orc_linsys *o = orc_linsys_create(1, (int)box.N, 0);
  orc_add_connections(o, box.E, 2, box.edges.data());
  orc_add_connections(o, box.N, 1, all.data());
  orc_finalize(o);
  orc_zero_system(o);
  orc_assemble_scalar_edges(o, box.E, box.edges.data(), box.area.data(), box.mdot.data(), box.xyz.data(),
                            box.dqdx.data(), box.vel.data(), box.q.data(), box.rho.data(), box.gamma.data(), &oo);
  orc_assemble_scalar_mass_be(o, box.N, qN.data(), box.q.data(), rhoN.data(), box.rho.data(), box.dualVol.data(), dt);
  orc_apply_dirichlet(o, (int64_t)wall.size(), wall.data(), box.q.data(), bc.data(), 0, 1);

  Realm realm = box.realm();
  LinearSolverConfig cfg;
  cfg.set("gmres", "jacobi", 1e-10, 60, 20);
  DevVec3 xyz(box.xyz), dqdx(box.dqdx), vel(box.vel), area(box.area);
  DevBuf q(box.q), rho(box.rho), gam(box.gamma), mdot(box.mdot), dQn(qN), dRhoN(rhoN), vol(box.dualVol), dBc(bc),
    dq((size_t)box.N);
  nalu_gpu_scalar_fields f = {xyz.v(), dqdx.v(), vel.v(), q.p, rho.p, gam.p, area.v(), mdot.p};

  EquationSystem eqsys(realm, "MixtureFractionEQS", &cfg);
  AssembleScalarEdgeSolverAlgorithm *alg = new AssembleScalarEdgeSolverAlgorithm(*eqsys.linsys_, box.edges, f);
  alg->opts_.use_limiter = 1; alg->opts_.hybrid_factor = 1.0;
  eqsys.solverAlgDriver_->solverAlgMap_[INTERIOR] = alg;
  eqsys.solverAlgDriver_->solverAlgMap_[MASS] =
    new ScalarMassBackwardEulerAlgorithm(*eqsys.linsys_, all, dQn.p, q.p, dRhoN.p, rho.p, vol.p, dt);
  eqsys.solverAlgDriver_->solverDirichAlgMap_[WALL] = new DirichletBC(*eqsys.linsys_, wall, q.p, dBc.p, 0, 1);
  eqsys.initialize();
  eqsys.assemble_and_solve(dq.p);
  expectSameSystem(*eqsys.linsys_, o, 1e-12, true);

  orc_solver_config oc = {1, 1, 1e-10, 60, 20};
  std::vector<double> hist(62);
  int histLen = 0;
  double finalResid = 0.0;
  const int iters = orc_solve(o, &oc, &finalResid, hist.data(), 62, &histLen);
  EXPECT_EQ(eqsys.linsys_->linearSolveIterations(), iters);
  const std::vector<double> &gh = eqsys.linsys_->residualHistory();
  double worst = 0.0;
  for (int k = 0; k < histLen && k < (int)gh.size(); ++k) worst = std::max(worst, std::fabs(gh[k] - hist[k]));
  EXPECT_TRUE(worst <= 1e-10);
  EXPECT_TRUE(maxRelDiff(dq.host(), o->sln, (size_t)box.N) <= 1e-10);
  std::printf("    n=%d scalar gmres/jacobi: %d iterations, history max abs diff %.2e\n", n, iters, worst);

  // a second Picard iteration of the same time step: scaled residual relative to the first
  const double first = eqsys.linsys_->nonLinearResidual();
  eqsys.assemble_and_solve(dq.p);
  EXPECT_NEAR(eqsys.linsys_->scaledNonLinearResidual(), eqsys.linsys_->nonLinearResidual() / first, 1e-15);
  orc_linsys_destroy(o);
}

// the adjacent algorithms of the projection loop through the host layer: bit-exact
static void test_adjacent_algorithms()
{
  Box box(9);
  Realm realm = box.realm();
  LinearSolverConfig cfg;
  GpuLinearSystem ls(realm, 1, "ContinuityEQS", &cfg);
  ls.buildEdgeToNodeGraph(box.edges);
  ls.finalizeLinearSystem();
  DevVec3 xyz(box.xyz), dpdx(box.dpdx), vel(box.vel), area(box.area);
  DevBuf p(box.p), rho(box.rho), vol(box.dualVol), mdot((size_t)box.E), g((size_t)3 * box.N);
  nalu_gpu_continuity_fields f = {xyz.v(), dpdx.v(), vel.v(), p.p, rho.p, area.v()};
  ls.computeMdotEdges(f, 0.1, 1.0, 1.0, 1.0, mdot.p);
  ls.assembleNodalGradEdges(area.v(), vol.p, p.p, g.p, g.p + box.N, g.p + 2 * box.N);
  ls.synchronize();
  const std::vector<double> m = mdot.host(), gh = g.host();
  EXPECT_TRUE(std::memcmp(m.data(), box.mdot.data(), m.size() * 8) == 0);
  bool same = true;
  for (int64_t i = 0; i < box.N && same; ++i)
    for (int j = 0; j < 3; ++j) same = same && gh[(size_t)(j * box.N + i)] == box.dpdx[(size_t)(3 * i + j)];
  EXPECT_TRUE(same);

  // p += dp, then u -= (dt/gamma1)/rho (G_new - G_old)
  std::vector<double> dp((size_t)box.N), gOld(box.dpdx), u(box.vel);
  for (int64_t i = 0; i < box.N; ++i) dp[(size_t)i] = 0.01 * std::cos(3.0 * box.xyz[3 * i + 1]);
  for (size_t i = 0; i < gOld.size(); ++i) gOld[i] *= 0.9;
  std::vector<unsigned char> mask((size_t)box.N, 1);
  for (int64_t i = 0; i < box.N; i += 7) mask[(size_t)i] = 0;
  DevBuf dDp(dp), dP2(box.p);
  ls.fieldAxpby(box.N, 1.0, dDp.p, 1.0, dP2.p);
  std::vector<double> pRef(box.p);
  orc_field_axpby(box.N, 1.0, dp.data(), 1.0, pRef.data());
  unsigned char *dMask = nullptr;
  HIP(hipMalloc((void **)&dMask, (size_t)box.N));
  HIP(hipMemcpy(dMask, mask.data(), (size_t)box.N, hipMemcpyHostToDevice));
  DevVec3 dOld(gOld), dU(box.vel);
  ls.projectNodalVelocity(box.N, 0.1, 1.0, dpdx.v(), dOld.v(), rho.p, dMask, dU.v());
  ls.synchronize();
  orc_project_nodal_velocity(box.N, 0.1, 1.0, box.dpdx.data(), gOld.data(), box.rho.data(), mask.data(), u.data());
  const std::vector<double> p2 = dP2.host(), uh = dU.buf.host();
  EXPECT_TRUE(std::memcmp(p2.data(), pRef.data(), p2.size() * 8) == 0);
  same = true;
  for (int64_t i = 0; i < box.N && same; ++i)
    for (int j = 0; j < 3; ++j) same = same && uh[(size_t)(j * box.N + i)] == u[(size_t)(3 * i + j)];
  EXPECT_TRUE(same);
  (void)hipFree(dMask);
}

// RcclComm on a one-rank communicator: every dot product of the solve goes through
// ncclAllReduce on the library's reduce buffer; results equal the solve without a communicator
static void test_rccl_comm_world_1(const char *method)
{
  Box box(10);
  Realm realm = box.realm();
  LinearSolverConfig cfg;
  cfg.set(method, "jacobi", 1e-10, 150, 25);
  DevVec3 xyz(box.xyz), dpdx(box.dpdx), vel(box.vel), area(box.area);
  DevBuf p(box.p), rho(box.rho), x0((size_t)box.N), x1((size_t)box.N);
  nalu_gpu_continuity_fields f = {xyz.v(), dpdx.v(), vel.v(), p.p, rho.p, area.v()};
  std::vector<double> hist[2];
  int iters[2] = {0, 0};
  ncclComm_t comm;
  int dev = 0;
  EXPECT_TRUE(ncclCommInitAll(&comm, 1, &dev) == ncclSuccess);
  for (int withComm = 0; withComm < 2; ++withComm) {
    EquationSystem eqsys(realm, "ContinuityEQS", &cfg);
    eqsys.solverAlgDriver_->solverAlgMap_[INTERIOR] = new AssembleContinuityEdgeSolverAlgorithm(*eqsys.linsys_, box.edges, f);
    eqsys.solverAlgDriver_->solverConstraintAlgMap_[REF_PRESSURE] =
      new FixPressureAtNodeAlgorithm(*eqsys.linsys_, true, 0, 0.0, box.p[0]);
    eqsys.initialize();
    std::vector<int> none, zero(1, 0);
    eqsys.linsys_->setHaloPlan(none, zero, none, zero);
    RcclComm *rc = withComm ? new RcclComm(eqsys.linsys_->handle(), comm, none, zero, zero) : nullptr;
    eqsys.assemble_and_solve(withComm ? x1.p : x0.p);
    hist[withComm] = eqsys.linsys_->residualHistory();
    iters[withComm] = eqsys.linsys_->linearSolveIterations();
    eqsys.linsys_->setComm(nullptr);
    delete rc;
  }
  EXPECT_TRUE(iters[0] > 5);
  EXPECT_EQ(iters[0], iters[1]);
  EXPECT_TRUE(hist[0] == hist[1]);
  const std::vector<double> a = x0.host(), b = x1.host();
  EXPECT_TRUE(std::memcmp(a.data(), b.data(), a.size() * 8) == 0);
  ncclCommDestroy(comm);
}

static void test_error_behaviour()
{
  Realm realm;
  realm.nOwned = 4;
  for (int i = 0; i < 4; ++i) realm.naluGlobalId.push_back(i + 1);
  LinearSolverConfig cfg;
  EXPECT_THROW(GpuLinearSystem(realm, 3, "MomentumEQS", &cfg));  // numDof == 1 only
  GpuLinearSystem ls(realm, 1, "Test", &cfg);
  EXPECT_THROW(ls.finalizeLinearSystem());                       // nothing built yet
  std::vector<int> tri = {0, 1, 2, 1, 2, 3};
  EXPECT_THROW(ls.buildFaceElemToNodeGraph(tri, 3));
  EXPECT_THROW(ls.buildNonConformalNodeGraph(tri));
  EXPECT_THROW(ls.buildOversetNodeGraph(tri));
  EXPECT_THROW(ls.prepareConstraints(0, 1));
  std::vector<int> edges = {0, 1, 1, 2, 2, 3, 0, 7};
  EXPECT_THROW(ls.buildEdgeToNodeGraph(edges));                  // node 7 does not exist
  edges.resize(6);
  ls.buildEdgeToNodeGraph(edges);
  ls.finalizeLinearSystem();
  ls.zeroSystem();
  std::vector<int> ids;
  std::vector<double> sv, rhs(2, 1.0), lhs(3, 1.0);
  EXPECT_THROW(ls.sumInto(std::vector<Entity>{0, 1}, ids, sv, rhs, lhs));  // lhs is not 2 x 2
  lhs.assign(4, 1.0);
  ls.sumInto(std::vector<Entity>{0, 3}, ids, sv, rhs, lhs);  // (0,3) is not in the graph: skipped silently
  std::vector<double> v;
  ls.getMatrixValues(v);
  double sum = 0.0;
  for (double a : v) sum += a;
  // sum_into_row's single ascending pass (src/TpetraLinearSystem.C:1363-1383): row 0 gets its
  // diagonal; in row 3 the missing column 0 runs the scan to the end, so (3,3) is dropped too
  EXPECT_NEAR(sum, 1.0, 0.0);
  EXPECT_THROW(ls.resetRows(std::vector<Entity>{0}, 0, 3));
  EXPECT_THROW(ls.writeToFile("x", false));
  cfg.set("biCgStab", "jacobi", 1e-6, 10);
  DevBuf x(4);
  EXPECT_THROW(ls.solve(x.p));
  cfg.set("cg", "ilut", 1e-6, 10);
  EXPECT_THROW(ls.solve(x.p));
  // non-convergence is not an error (src/LinearSolver.C:224): status 0, iterations = max
  std::vector<double> diag(1, 2.0), one(1, 1.0);
  ls.zeroSystem();
  for (int i = 0; i < 4; ++i) ls.sumInto(std::vector<Entity>{i}, ids, sv, one, diag);
  std::vector<double> off = {0.0, -1.0, -1.0, 0.0}, zero2(2, 0.0);
  for (int i = 0; i < 3; ++i) ls.sumInto(std::vector<Entity>{i, i + 1}, ids, sv, zero2, off);
  ls.loadComplete();
  cfg.set("cg", "jacobi", 1e-14, 1);
  EXPECT_EQ(ls.solve(x.p), 0);
  EXPECT_EQ(ls.linearSolveIterations(), 1);
}

static void test_write_matrix_files(const std::string &dir)
{
  Box box(4, false);
  Realm realm = box.realm();
  LinearSolverConfig cfg;
  cfg.set("cg", "jacobi", 1e-8, 50);
  cfg.setWriteMatrixFiles(true);
  DevVec3 xyz(box.xyz), dpdx(box.dpdx), vel(box.vel), area(box.area);
  DevBuf p(box.p), rho(box.rho), dp((size_t)box.N);
  nalu_gpu_continuity_fields f = {xyz.v(), dpdx.v(), vel.v(), p.p, rho.p, area.v()};
  const std::string base = dir + "/ContinuityEQS";
  EquationSystem eqsys(realm, base, &cfg);
  eqsys.solverAlgDriver_->solverAlgMap_[INTERIOR] = new AssembleContinuityEdgeSolverAlgorithm(*eqsys.linsys_, box.edges, f);
  eqsys.solverAlgDriver_->solverConstraintAlgMap_[REF_PRESSURE] =
    new FixPressureAtNodeAlgorithm(*eqsys.linsys_, true, 0, 0.0, box.p[0]);
  eqsys.initialize();
  eqsys.assemble_and_solve(dp.p);
  const double precond1 = eqsys.linsys_->get_timer_precond();
  eqsys.assemble_and_solve(dp.p);  // writeCounter_ advances: -O-0 and -O-1
  // the linear system holds the LAST solve's preconditioner time (src/LinearSolver.C:211); the
  // equation system is what sums it (src/EquationSystem.C:292)
  const double precond2 = eqsys.linsys_->get_timer_precond();
  EXPECT_TRUE(precond1 > 0.0 && precond2 > 0.0);
  EXPECT_NEAR(eqsys.timerPrecond_, precond1 + precond2, 1e-12);
  for (int c = 0; c < 2; ++c) {
    std::ifstream mm(base + "-O-" + std::to_string(c) + ".mm.1");
    EXPECT_TRUE(mm.good());
    std::string line;
    std::getline(mm, line);
    EXPECT_EQ(line, std::string("%%MatrixMarket matrix coordinate real general"));
    while (std::getline(mm, line) && line[0] == '%') {}
    long long nr, nc, nnz;
    std::istringstream(line) >> nr >> nc >> nnz;
    EXPECT_EQ(nr, (long long)box.N);
    EXPECT_EQ(nnz, (long long)eqsys.linsys_->numNonZeros());
    long long count = 0, r, cc;
    double v, trace = 0.0;
    while (mm >> r >> cc >> v) { ++count; if (r == cc) trace += v; }
    EXPECT_EQ(count, nnz);
    EXPECT_TRUE(trace > 0.0);
    EXPECT_TRUE(std::ifstream(base + "-O-" + std::to_string(c) + ".rhs.1").good());
    EXPECT_TRUE(std::ifstream(base + "-O-" + std::to_string(c) + ".sln.1").good());
  }
}

// ---------------------------------------------------------------------------------------
static void test_solver_config()
{
  LinearSolverConfig cfg;
  std::map<std::string, std::string> node;
  EXPECT_THROW(cfg.load(node));  // name and method are mandatory (node["name"].as<>)
  node["name"] = "solve_scalar"; node["method"] = "gmres";
  cfg.load(node);  // defaults of src/LinearSolverConfig.C:39-49
  EXPECT_EQ(cfg.preconditioner_type(), std::string("default"));
  EXPECT_NEAR(cfg.tolerance(), 1.e-4, 0.0);
  EXPECT_NEAR(cfg.finalTolerance(), 1.e-4, 0.0);
  EXPECT_EQ(cfg.maxIterations(), 50);
  EXPECT_EQ(cfg.kspace(), 50);
  EXPECT_TRUE(!cfg.getWriteMatrixFiles());
  nalu_gpu_solver_config c = cfg.c_struct(false);
  EXPECT_EQ(c.method, (int)NALU_GPU_METHOD_GMRES);
  EXPECT_EQ(c.precond, (int)NALU_GPU_PRECOND_JACOBI);
  node["tolerance"] = "1e-5"; node["final_tolerance"] = "1e-8"; node["preconditioner"] = "sgs";
  node["write_matrix_files"] = "yes";
  cfg.load(node);
  EXPECT_NEAR(cfg.c_struct(false).tolerance, 1e-5, 0.0);
  EXPECT_NEAR(cfg.c_struct(true).tolerance, 1e-8, 0.0);  // src/LinearSolver.C:215-219
  EXPECT_EQ(cfg.c_struct(false).precond, (int)NALU_GPU_PRECOND_SGS);
  EXPECT_TRUE(cfg.getWriteMatrixFiles());
  node["preconditioner"] = "amg";
  EXPECT_THROW(cfg.load(node));  // "invalid linear solver preconditioner specified"
  node["preconditioner"] = "muelu";
  cfg.load(node);                // a valid reference option ...
  EXPECT_THROW(cfg.c_struct(false));  // ... that this backend rejects loudly at solve time
}

int main(int argc, char **argv)
{
  bool hostOnly = false;
  std::string tmp = ".";
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--host-only")) hostOnly = true;
    else if (!std::strcmp(argv[i], "--tmp") && i + 1 < argc) tmp = argv[++i];
  }
  struct T { const char *name; std::function<void()> fn; bool gpu; };
  const std::vector<T> tests = {
    {"LinearSolverConfig.load", test_solver_config, false},
    {"Tpetra.basic", test_Tpetra_basic, true},
    {"ContinuityEdge.cg_jacobi", [] { test_continuity_edge_assemble_and_solve(12, "jacobi", "cg", 200); }, true},
    {"ContinuityEdge.gmres_sgs", [] { test_continuity_edge_assemble_and_solve(9, "sgs", "gmres", 90); }, true},
    {"ScalarEdge.gmres_jacobi", [] { test_scalar_edge_assemble_and_solve(10); }, true},
    {"AdjacentAlgorithms.bit_exact", test_adjacent_algorithms, true},
    {"RcclComm.world_1_cg", [] { test_rccl_comm_world_1("cg"); }, true},
    {"RcclComm.world_1_gmres", [] { test_rccl_comm_world_1("gmres"); }, true},
    {"LinearSystem.errors", test_error_behaviour, true},
    {"LinearSystem.writeToFile", [&] { test_write_matrix_files(tmp); }, true},
  };
  int ran = 0;
  for (const T &t : tests) {
    if (t.gpu && hostOnly) continue;
    const int before = g_fail;
    std::printf("[ RUN      ] %s\n", t.name);
    std::fflush(stdout);
    try {
      t.fn();
    } catch (const std::exception &e) {
      ++g_fail;
      std::printf("  FAIL unexpected exception: %s\n", e.what());
    }
    std::printf("[ %s ] %s\n", g_fail == before ? "      OK" : " FAILED ", t.name);
    std::fflush(stdout);
    ++ran;
  }
  std::printf("%d tests, %d checks, %d failures\n", ran, g_checks, g_fail);
  return g_fail ? 1 : 0;
}
