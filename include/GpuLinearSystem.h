// GpuLinearSystem.h -- C++11 host side above the C ABI of nalu_gpu.h (header only).
//
// The reference is compiled C++, so this is the host layer in the reference's language: the
// sierra::nalu::LinearSystem surface (include/LinearSystem.h:40-168 of the reference), the
// TpetraLinearSolverConfig keys (src/LinearSolverConfig.C:34-107), the SolverAlgorithm /
// SolverAlgorithmDriver pair (include/SolverAlgorithm.h:34-55, src/SolverAlgorithmDriver.C:66-138)
// and EquationSystem::assemble_and_solve (src/EquationSystem.C:263-307) with the same method
// names, argument meaning and error behaviour (std::runtime_error, status 0 from solve,
// non-convergence is not an error).  STK/Trilinos are not available where this is built, so
// the three STK types the interface mentions are replaced by plain stand-ins:
//
//   stk::mesh::Entity       -> Entity     = local node id (TpetraLinearSystem::entityToLID_,
//                                           src/TpetraLinearSystem.C:950-974, already applied)
//   stk::mesh::PartVector   -> EntityVector / connectivity arrays of the entities in the parts
//   stk::mesh::FieldBase*   -> double*      device-resident array indexed by local node id
//
// Inside Nalu the same class derives from sierra::nalu::LinearSystem and resolves the three
// types from STK.  Nothing here touches the CPU oracle.
#ifndef GpuLinearSystem_h
#define GpuLinearSystem_h

#include "nalu_gpu.h"

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace sierra {
namespace nalu {
namespace gpu {

typedef int Entity;
typedef std::vector<Entity> EntityVector;

// include/Enums.h:18-43 -- the key order of SolverAlgorithmDriver's std::map
enum AlgorithmType {
  INTERIOR = 0, INFLOW = 1, WALL = 2, OPEN = 3, MASS = 4, SRC = 5, SYMMETRY = 6, WALL_HF = 7,
  WALL_CHT = 8, WALL_RAD = 9, NON_CONFORMAL = 10, ELEM_SOURCE = 11, OVERSET = 12, WALL_ABL = 13,
  REF_PRESSURE = 14
};

inline void checkError(const int err_code, const char *msg)
{
  if (err_code != NALU_GPU_OK)
    throw std::runtime_error(std::string(msg) + ": " + nalu_gpu_last_error());
}

// ---------------------------------------------------------------------------------------
// TpetraLinearSolverConfig (include/LinearSolverConfig.h:28-89, src/LinearSolverConfig.C:34-107)
class LinearSolverConfig
{
public:
  LinearSolverConfig()
    : method_("gmres"), precond_("default"), tolerance_(1.e-4), finalTolerance_(1.e-4),
      maxIterations_(50), kspace_(50), outputLevel_(0), writeMatrixFiles_(false)
  {
  }

  // One entry of the YAML `linear_solvers:` list as key -> scalar text (yaml-cpp is not
  // available here); same keys, defaults and failure modes as TpetraLinearSolverConfig::load.
  void load(const std::map<std::string, std::string> &node)
  {
    name_ = require(node, "name");
    method_ = require(node, "method");
    precond_ = get(node, "preconditioner", "default");
    tolerance_ = std::stod(get(node, "tolerance", "1.e-4"));
    finalTolerance_ = node.count("final_tolerance") ? std::stod(node.at("final_tolerance")) : tolerance_;
    maxIterations_ = std::stoi(get(node, "max_iterations", "50"));
    kspace_ = std::stoi(get(node, "kspace", "50"));
    outputLevel_ = std::stoi(get(node, "output_level", "0"));
    static const char *known[] = {"sgs", "mt_sgs", "jacobi", "default", "ilut", "riluk", "muelu",
                                  "none", "chebyshev"};
    if (std::find(known, known + 9, precond_) == known + 9)
      throw std::runtime_error("invalid linear solver preconditioner specified ");  // :97-99
    const std::string w = get(node, "write_matrix_files", "false");
    writeMatrixFiles_ = (w == "true" || w == "yes" || w == "on" || w == "1");
  }

  const std::string &name() const { return name_; }
  const std::string &get_method() const { return method_; }
  const std::string &preconditioner_type() const { return precond_; }
  double &tolerance() { return tolerance_; }
  double &finalTolerance() { return finalTolerance_; }
  int &maxIterations() { return maxIterations_; }
  int &kspace() { return kspace_; }
  bool getWriteMatrixFiles() const { return writeMatrixFiles_; }
  void setWriteMatrixFiles(bool v) { writeMatrixFiles_ = v; }
  void set(const std::string &method, const std::string &precond, double tol, int maxIt, int kspace = 50)
  {
    method_ = method; precond_ = precond; tolerance_ = finalTolerance_ = tol;
    maxIterations_ = maxIt; kspace_ = kspace;
  }

  // what TpetraLinearSolver::solve hands to Belos (src/LinearSolver.C:181-230)
  nalu_gpu_solver_config c_struct(bool isFinalOuterIter) const
  {
    nalu_gpu_solver_config c;
    if (method_ == "cg") c.method = NALU_GPU_METHOD_CG;
    else if (method_ == "gmres") c.method = NALU_GPU_METHOD_GMRES;
    else if (method_ == "biCgStab" || method_ == "bicgstab") c.method = NALU_GPU_METHOD_BICGSTAB;
    else if (method_ == "tfqmr") c.method = NALU_GPU_METHOD_TFQMR;
    else throw std::runtime_error("solver method '" + method_ + "' not supported on the GPU (cg, gmres, biCgStab, tfqmr)");
    if (precond_ == "jacobi" || precond_ == "default") c.precond = NALU_GPU_PRECOND_JACOBI;
    else if (precond_ == "sgs") c.precond = NALU_GPU_PRECOND_SGS;
    else if (precond_ == "mt_sgs") c.precond = NALU_GPU_PRECOND_MT_SGS;
    else if (precond_ == "chebyshev") c.precond = NALU_GPU_PRECOND_CHEBYSHEV;
    else if (precond_ == "none") c.precond = NALU_GPU_PRECOND_NONE;
    else throw std::runtime_error("preconditioner '" + precond_ + "' not supported on the GPU");
    c.tolerance = isFinalOuterIter ? finalTolerance_ : tolerance_;  // src/LinearSolver.C:215-219
    c.max_iterations = maxIterations_;
    c.kspace = kspace_;
    c.check_every = 0;
    return c;
  }

private:
  static std::string require(const std::map<std::string, std::string> &n, const char *k)
  {
    std::map<std::string, std::string>::const_iterator it = n.find(k);
    if (it == n.end()) throw std::runtime_error(std::string("linear solver: missing key ") + k);
    return it->second;
  }
  static std::string get(const std::map<std::string, std::string> &n, const char *k, const char *dflt)
  {
    std::map<std::string, std::string>::const_iterator it = n.find(k);
    return it == n.end() ? std::string(dflt) : it->second;
  }
  std::string name_, method_, precond_;
  double tolerance_, finalTolerance_;
  int maxIterations_, kspace_, outputLevel_;
  bool writeMatrixFiles_;
};

// ---------------------------------------------------------------------------------------
// What the linear system reads from Realm / BulkData: the local numbering of
// beginLinearSystemConstruction (src/TpetraLinearSystem.C:162-302) and the norm scaling.
struct Realm {
  int nOwned, nSharedNotOwned, nGhost;
  std::vector<int64_t> naluGlobalId;  // local order: owned, shared-not-owned, ghosts
  std::vector<int> ownerRank;         // same order; may be empty on one rank
  int pRank, pSize;
  int device;
  void *hipStream;
  double l2Scaling_;       // src/Realm.C:2965-2969: 1/sqrt(global owned node count)
  bool isFinalOuterIter_;
  Realm()
    : nOwned(0), nSharedNotOwned(0), nGhost(0), pRank(0), pSize(1), device(0), hipStream(nullptr),
      l2Scaling_(1.0), isFinalOuterIter_(false)
  {
  }
};

// ---------------------------------------------------------------------------------------
class GpuLinearSystem
{
public:
  // LinearSystem::create(realm, numDof, eqSys, linearSolver) (src/LinearSystem.C:90-109)
  GpuLinearSystem(Realm &realm, const unsigned numDof, const std::string &eqSysName,
                  LinearSolverConfig *linearSolver)
    : realm_(realm), h_(nullptr), inConstruction_(false), writeCounter_(0), numDof_(numDof),
      eqSysName_(eqSysName), linearSolver_(linearSolver), linearSolveIterations_(0),
      nonLinearResidual_(0.0), linearResidual_(0.0), firstNonLinearResidual_(1.0e8),
      scaledNonLinearResidual_(1.0e8), timerPrecond_(0.0), numEdges_(0), firstTimeStepSolve_(true),
      provideOutput_(false)
  {
    checkError(nalu_gpu_create(&h_, realm.device, (int)numDof, realm.hipStream), "GpuLinearSystem");
  }
  ~GpuLinearSystem()
  {
    if (h_) nalu_gpu_destroy(h_);
  }
  GpuLinearSystem(const GpuLinearSystem &) = delete;
  GpuLinearSystem &operator=(const GpuLinearSystem &) = delete;

  nalu_gpu_linsys *handle() { return h_; }

  // ---- Graph/Matrix Construction (include/LinearSystem.h:55-75)
  void buildNodeGraph(const EntityVector &nodes)
  {
    beginLinearSystemConstruction();
    checkError(nalu_gpu_build_node_graph(h_, (int64_t)nodes.size(), nodes.data()), "buildNodeGraph");
  }
  // edgeNodes: 2 local node ids per edge; the edge order defines the layout of edge fields
  // nalu_gpu_suggest_edge_order: the slot-major permutation (perm[new] = old) of an edge list, to be applied
  // to the list and to every edge field before buildEdgeToNodeGraph (no counterpart in the reference)
  static std::vector<int64_t> suggestEdgeOrder(const std::vector<int> &edgeNodes)
  {
    std::vector<int64_t> perm(edgeNodes.size() / 2);
    if (nalu_gpu_suggest_edge_order((int64_t)perm.size(), edgeNodes.data(), perm.data()) != NALU_GPU_OK)
      throw std::runtime_error(nalu_gpu_last_error());
    return perm;
  }
  void buildEdgeToNodeGraph(const std::vector<int> &edgeNodes)
  {
    beginLinearSystemConstruction();
    checkError(nalu_gpu_build_edge_to_node_graph(h_, (int64_t)(edgeNodes.size() / 2), edgeNodes.data()),
               "buildEdgeToNodeGraph");
    numEdges_ += (int64_t)(edgeNodes.size() / 2);
  }
  void buildElemToNodeGraph(const std::vector<int> &elemNodes, int nodesPerElem)
  {
    beginLinearSystemConstruction();
    if (nodesPerElem <= 0 || elemNodes.size() % (size_t)nodesPerElem)
      throw std::runtime_error("buildElemToNodeGraph: connectivity is not a multiple of nodesPerElem");
    checkError(nalu_gpu_build_elem_to_node_graph(h_, (int64_t)(elemNodes.size() / nodesPerElem), nodesPerElem,
                                                 elemNodes.data()),
               "buildElemToNodeGraph");
  }
  void buildFaceToNodeGraph(const std::vector<int> &faceNodes, int nodesPerFace)
  {
    buildElemToNodeGraph(faceNodes, nodesPerFace);  // same loop over another entity rank (:427-433)
  }
  void buildReducedElemToNodeGraph(const std::vector<int> &, int) { notSupported("buildReducedElemToNodeGraph"); }
  void buildFaceElemToNodeGraph(const std::vector<int> &, int) { notSupported("buildFaceElemToNodeGraph"); }
  void buildNonConformalNodeGraph(const std::vector<int> &) { notSupported("buildNonConformalNodeGraph"); }
  void buildOversetNodeGraph(const std::vector<int> &) { notSupported("buildOversetNodeGraph"); }
  void buildDirichletNodeGraph(const EntityVector &nodes)
  {
    beginLinearSystemConstruction();
    checkError(nalu_gpu_build_dirichlet_node_graph(h_, (int64_t)nodes.size(), nodes.data()),
               "buildDirichletNodeGraph");
  }
  // src/TpetraLinearSystem.C:157-160 (simple case of getDofStatus_impl, :2148-2157)
  int getDofStatus(Entity node)
  {
    int st = 0;
    int rc = nalu_gpu_get_dof_status(h_, 1, &node, &st);
    if (rc == NALU_GPU_ESTATE) {  // nothing built yet: the node counts are not registered
      beginLinearSystemConstruction();
      rc = nalu_gpu_get_dof_status(h_, 1, &node, &st);
    }
    checkError(rc, "getDofStatus");
    return st;
  }
  void finalizeLinearSystem()
  {
    if (!inConstruction_) throw std::runtime_error("finalizeLinearSystem: no graph was built");
    inConstruction_ = false;
    checkError(nalu_gpu_finalize(h_), "finalizeLinearSystem");
    int64_t nr, nc, nnz;
    checkError(nalu_gpu_get_sizes(h_, &nr, &nc, &nnz), "finalizeLinearSystem");
    numRows_ = nr; numCols_ = nc; nnz_ = nnz;
  }

  // ---- Matrix Assembly
  void zeroSystem() { checkError(nalu_gpu_zero_system(h_), "zeroSystem"); }

  // include/LinearSystem.h:92-99; scratchIds/scratchVals are resized as the reference does
  // (src/TpetraLinearSystem.C:1470), scratchIds ends up holding the local ids that were used.
  void sumInto(const std::vector<Entity> &sym_meshobj, std::vector<int> &scratchIds,
               std::vector<double> &scratchVals, const std::vector<double> &rhs,
               const std::vector<double> &lhs, const char *trace_tag = 0)
  {
    (void)trace_tag;
    const size_t n_obj = sym_meshobj.size();
    const size_t numRows = n_obj * numDof_;
    if (rhs.size() != numRows || lhs.size() != numRows * numRows)
      throw std::runtime_error("sumInto: rhs/lhs sizes do not match the entity count");
    scratchIds.assign(sym_meshobj.begin(), sym_meshobj.end());
    scratchVals.resize(numRows);
    checkError(nalu_gpu_sum_into(h_, (int)n_obj, scratchIds.data(), rhs.data(), lhs.data()), "sumInto");
  }

  // include/LinearSystem.h:80-90, the Kokkos-view overload that the reference calls per entity
  // from inside a team loop: on the GPU the loop is the launch.  `plan` = createScatterPlan of the
  // entities' connectivity; rhs / lhs are device arrays [n_ent][n_obj] / [n_ent][n_obj][n_obj].
  nalu_gpu_scatter_plan *createScatterPlan(const std::vector<Entity> &entityNodes, int nodesPerEntity)
  {
    if (nodesPerEntity < 1 || entityNodes.size() % (size_t)nodesPerEntity)
      throw std::runtime_error("createScatterPlan: entity list is not a multiple of nodesPerEntity");
    nalu_gpu_scatter_plan *p = 0;
    checkError(nalu_gpu_scatter_plan_create(h_, (int64_t)(entityNodes.size() / nodesPerEntity), nodesPerEntity,
                                            entityNodes.data(), &p), "createScatterPlan");
    return p;
  }
  void destroyScatterPlan(nalu_gpu_scatter_plan *p) { nalu_gpu_scatter_plan_destroy(p); }
  void sumIntoBatch(nalu_gpu_scatter_plan *plan, const double *rhsDev, const double *lhsDev,
                    int scatterMode = NALU_GPU_SCATTER_ROWS)
  {
    checkError(nalu_gpu_sum_into_batch(h_, plan, rhsDev, lhsDev, scatterMode), "sumInto (batch)");
  }

  // include/LinearSystem.h:101-106: the nodes of `parts`, fields on the device
  void applyDirichletBCs(const double *solutionField, const double *bcValuesField, const EntityVector &nodes,
                         const unsigned beginPos, const unsigned endPos)
  {
    requireScalar(beginPos, endPos, "applyDirichletBCs");
    checkError(nalu_gpu_apply_dirichlet(h_, (int64_t)nodes.size(), nodes.data(), solutionField, bcValuesField),
               "applyDirichletBCs");
  }
  void prepareConstraints(const unsigned, const unsigned) { notSupported("prepareConstraints"); }
  void resetRows(std::vector<Entity> nodeList, const unsigned beginPos, const unsigned endPos)
  {
    requireScalar(beginPos, endPos, "resetRows");
    checkError(nalu_gpu_reset_rows(h_, (int64_t)nodeList.size(), nodeList.data()), "resetRows");
  }

  // ---- bulk entries that replace the per-edge apply_coeff loops (SolverAlgorithm::execute)
  void assembleContinuityEdges(const nalu_gpu_continuity_fields &f, double dt, double gamma1, double nocFac,
                               double interpTogether, int scatterMode = NALU_GPU_SCATTER_ROWS)
  {
    checkError(nalu_gpu_assemble_continuity_edges(h_, &f, dt, gamma1, nocFac, interpTogether, scatterMode),
               "AssembleContinuityEdgeSolverAlgorithm");
  }
  void assembleScalarEdges(const nalu_gpu_scalar_fields &f, const nalu_gpu_scalar_options &o,
                           int scatterMode = NALU_GPU_SCATTER_ROWS)
  {
    checkError(nalu_gpu_assemble_scalar_edges(h_, &f, &o, scatterMode), "AssembleScalarEdgeSolverAlgorithm");
  }
  void assembleScalarMassBackwardEuler(const double *qN, const double *qNp1, const double *rhoN,
                                       const double *rhoNp1, const double *dualVolume, double dt)
  {
    checkError(nalu_gpu_assemble_scalar_mass_be(h_, qN, qNp1, rhoN, rhoNp1, dualVolume, dt),
               "ScalarMassBackwardEulerNodeSuppAlg");
  }

  // ---- the adjacent algorithms of the pressure-projection loop; each is one call
  // ComputeMdotEdgeAlgorithm::execute (src/ComputeMdotEdgeAlgorithm.C:67-174)
  void computeMdotEdges(const nalu_gpu_continuity_fields &f, double dt, double gamma1, double nocFac,
                        double interpTogether, double *massFlowRate)
  {
    checkError(nalu_gpu_compute_mdot_edges(h_, &f, dt, gamma1, nocFac, interpTogether, massFlowRate),
               "ComputeMdotEdgeAlgorithm");
  }
  // AssembleNodalGradEdgeAlgorithm::execute + driver zeroing (src/AssembleNodalGradEdgeAlgorithm.C:54-118)
  void assembleNodalGradEdges(nalu_gpu_vec3 edgeAreaVec, const double *dualNodalVolume, const double *q,
                              double *gx, double *gy, double *gz)
  {
    checkError(nalu_gpu_nodal_grad_edges(h_, edgeAreaVec, dualNodalVolume, q, gx, gy, gz),
               "AssembleNodalGradEdgeAlgorithm");
  }
  // ComputeGeometryInteriorAlgorithm::execute for Hex8 (src/ComputeGeometryInteriorAlgorithm.C:78-216)
  void computeGeometryHex8(int64_t nElem, const int *hexNodesDev, nalu_gpu_vec3 coordinates,
                           double *dualNodalVolume, double *areaX, double *areaY, double *areaZ)
  {
    checkError(nalu_gpu_compute_geometry_hex8(h_, nElem, hexNodesDev, coordinates, dualNodalVolume, areaX,
                                              areaY, areaZ),
               "ComputeGeometryInteriorAlgorithm");
  }
  // field_axpby (src/FieldFunctions.C:24-58): y = alpha x + beta y
  void fieldAxpby(int64_t n, double alpha, const double *x, double beta, double *y)
  {
    checkError(nalu_gpu_field_axpby(h_, n, alpha, x, beta, y), "field_axpby");
  }
  // LowMachEquationSystem::project_nodal_velocity (src/LowMachEquationSystem.C:772-855)
  void projectNodalVelocity(int64_t n, double dt, double gamma1, nalu_gpu_vec3 dpdxNew, nalu_gpu_vec3 dpdxOld,
                            const double *density, const unsigned char *projected, nalu_gpu_vec3 velocity)
  {
    checkError(nalu_gpu_project_nodal_velocity(h_, n, dt, gamma1, dpdxNew, dpdxOld, density, projected, velocity),
               "project_nodal_velocity");
  }

  // ---- multi-GPU plumbing: the Tpetra Import / all-reduce equivalents are registered once
  // after finalizeLinearSystem (see nalu_gpu.h; RcclComm.h gives the callbacks for RCCL)
  void setHaloPlan(const std::vector<int> &neighborRanks, const std::vector<int> &sendOffsets,
                   const std::vector<int> &sendLids, const std::vector<int> &recvOffsets)
  {
    if (sendOffsets.size() != neighborRanks.size() + 1 || recvOffsets.size() != neighborRanks.size() + 1)
      throw std::runtime_error("setHaloPlan: offsets need one entry per neighbour plus one");
    checkError(nalu_gpu_set_halo_plan(h_, (int)neighborRanks.size(), neighborRanks.data(), sendOffsets.data(),
                                      sendLids.data(), recvOffsets.data()),
               "setHaloPlan");
  }
  void setComm(const nalu_gpu_comm *comm) { checkError(nalu_gpu_set_comm(h_, comm), "setComm"); }
  // stk::mesh::copy_owned_to_shared of one column-space field (src/LinearSystem.C:111-116)
  void syncField(double *field) { checkError(nalu_gpu_halo_exchange(h_, field), "sync_field"); }

  // ---- Solve (src/TpetraLinearSystem.C:1709-1780)
  int solve(double *linearSolutionField)
  {
    if (linearSolver_->getWriteMatrixFiles()) writeToFile(eqSysName_.c_str());
    const nalu_gpu_solver_config c = linearSolver_->c_struct(realm_.isFinalOuterIter_);
    nalu_gpu_solve_info info;
    residualHistory_.assign((size_t)c.max_iterations + 2, 0.0);
    checkError(nalu_gpu_solve(h_, &c, linearSolutionField, &info, residualHistory_.data(),
                              (int)residualHistory_.size()),
               "solve");
    residualHistory_.resize((size_t)info.history_len);
    if (linearSolver_->getWriteMatrixFiles()) {
      writeSolutionToFile(eqSysName_.c_str());
      ++writeCounter_;
    }
    linearSolveIterations_ = info.iters;
    nonLinearResidual_ = realm_.l2Scaling_ * info.rhs_norm2;
    linearResidual_ = info.final_resid;
    timerPrecond_ = info.precond_seconds;  // the last solve's (src/LinearSolver.C:211); the caller sums

    if (firstTimeStepSolve_) firstNonLinearResidual_ = nonLinearResidual_;
    scaledNonLinearResidual_ =
      nonLinearResidual_ / std::max(std::numeric_limits<double>::epsilon(), firstNonLinearResidual_);
    if (provideOutput_ && realm_.pRank == 0)
      std::cout << eqSysName_ << " " << info.iters << " " << info.final_resid << " " << nonLinearResidual_
                << " " << scaledNonLinearResidual_ << std::endl;
    firstTimeStepSolve_ = false;
    return 0;  // src/LinearSolver.C:191: the status is always 0; Belos' verdict is ignored (:224)
  }
  void loadComplete() { checkError(nalu_gpu_load_complete(h_), "loadComplete"); }

  // ---- MatrixMarket dumps (src/TpetraLinearSystem.C:1901-1924, 2002-2019): owned matrix and rhs
  // in 1-based global ids, "<name>-O-<count>.mm.<p_size>[.<p_rank>]", ".rhs", ".sln"
  void writeToFile(const char *filename, bool useOwned = true)
  {
    if (!useOwned)
      throw std::runtime_error("writeToFile: there is no shared-not-owned matrix (cut edges are assembled "
                               "by the row owner)");
    std::vector<int64_t> rp((size_t)numRows_ + 1);
    std::vector<int> ci((size_t)std::max<int64_t>(nnz_, 1));
    std::vector<double> v((size_t)std::max<int64_t>(nnz_, 1)), b((size_t)std::max<int64_t>(numRows_, 1));
    checkError(nalu_gpu_get_graph(h_, rp.data(), ci.data()), "writeToFile");
    checkError(nalu_gpu_get_matrix_values(h_, v.data()), "writeToFile");
    checkError(nalu_gpu_get_rhs(h_, b.data()), "writeToFile");
    int64_t nGlobal = realm_.nOwned;
    if (realm_.pSize > 1)
      nGlobal = *std::max_element(realm_.naluGlobalId.begin(), realm_.naluGlobalId.end());
    FILE *f = openFile(fileName(filename, "mm"));
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%% %s\n%% Tpetra matrix for: %s\n",
                 eqSysName_.c_str(), eqSysName_.c_str());
    std::fprintf(f, "%lld %lld %lld\n", (long long)nGlobal * numDof_, (long long)nGlobal * numDof_, (long long)nnz_);
    for (int64_t i = 0; i < numRows_; ++i)
      for (int64_t k = rp[(size_t)i]; k < rp[(size_t)i + 1]; ++k)
        std::fprintf(f, "%lld %lld %.17e\n", (long long)realm_.naluGlobalId[(size_t)i],
                     (long long)realm_.naluGlobalId[(size_t)ci[(size_t)k]], v[(size_t)k]);
    std::fclose(f);
    writeDense(fileName(filename, "rhs"), b.data(), numRows_);
  }
  void writeSolutionToFile(const char *filename, bool useOwned = true)
  {
    (void)useOwned;
    std::vector<double> x((size_t)std::max<int64_t>(numRows_, 1));
    checkError(nalu_gpu_get_solution(h_, x.data()), "writeSolutionToFile");
    writeDense(fileName(filename, "sln"), x.data(), numRows_);
  }
  std::string fileName(const char *base, const char *kind) const
  {
    std::string s = std::string(base) + "-O-" + std::to_string(writeCounter_) + "." + kind + "." +
                    std::to_string(realm_.pSize);
    if (realm_.pSize > 1) s += "." + std::to_string(realm_.pRank);
    return s;
  }

  // ---- accessors (include/LinearSystem.h:129-139)
  unsigned numDof() const { return numDof_; }
  const int &linearSolveIterations() { return linearSolveIterations_; }
  const double &linearResidual() { return linearResidual_; }
  const double &nonLinearResidual() { return nonLinearResidual_; }
  const double &scaledNonLinearResidual() { return scaledNonLinearResidual_; }
  void setNonLinearResidual(const double nlr) { nonLinearResidual_ = nlr; }
  const std::string name() { return eqSysName_; }
  double get_timer_precond() { return timerPrecond_; }
  void zero_timer_precond() { timerPrecond_ = 0.0; }
  const std::vector<double> &residualHistory() const { return residualHistory_; }
  int64_t numRows() const { return numRows_; }
  int64_t numCols() const { return numCols_; }
  int64_t numNonZeros() const { return nnz_; }
  int64_t numEdges() const { return numEdges_; }
  void synchronize() { checkError(nalu_gpu_synchronize(h_), "synchronize"); }

  // Tpetra-style row views for tests (getLocalRowView, unit_tests/UnitTestTpetra.C:192-201)
  void getGraph(std::vector<int64_t> &rowPtr, std::vector<int> &colIdx)
  {
    rowPtr.assign((size_t)numRows_ + 1, 0);
    colIdx.assign((size_t)std::max<int64_t>(nnz_, 1), 0);
    checkError(nalu_gpu_get_graph(h_, rowPtr.data(), colIdx.data()), "getGraph");
    colIdx.resize((size_t)nnz_);
  }
  void getMatrixValues(std::vector<double> &vals)
  {
    vals.assign((size_t)std::max<int64_t>(nnz_, 1), 0.0);
    checkError(nalu_gpu_get_matrix_values(h_, vals.data()), "getMatrixValues");
    vals.resize((size_t)nnz_);
  }
  void getRhs(std::vector<double> &b)
  {
    b.assign((size_t)std::max<int64_t>(numRows_, 1), 0.0);
    checkError(nalu_gpu_get_rhs(h_, b.data()), "getRhs");
    b.resize((size_t)numRows_);
  }

  bool firstTimeStepSolve() const { return firstTimeStepSolve_; }
  void setFirstTimeStepSolve(bool v) { firstTimeStepSolve_ = v; }

protected:
  // src/TpetraLinearSystem.C:162-302 (idempotent, as there: `if (inConstruction_) return;`)
  void beginLinearSystemConstruction()
  {
    if (inConstruction_) return;
    inConstruction_ = true;
    const size_t n = (size_t)realm_.nOwned + realm_.nSharedNotOwned + realm_.nGhost;
    if (realm_.naluGlobalId.size() != n)
      throw std::runtime_error("beginLinearSystemConstruction: naluGlobalId does not cover the local nodes");
    if (!realm_.ownerRank.empty() && realm_.ownerRank.size() != n)
      throw std::runtime_error("beginLinearSystemConstruction: ownerRank does not cover the local nodes");
    checkError(nalu_gpu_begin_construction(h_, realm_.nOwned, realm_.nSharedNotOwned, realm_.nGhost,
                                           realm_.naluGlobalId.data(),
                                           realm_.ownerRank.empty() ? nullptr : realm_.ownerRank.data(),
                                           realm_.pRank),
               "beginLinearSystemConstruction");
  }

private:
  static void notSupported(const char *what)
  {
    throw std::runtime_error(std::string(what) + " not supported in GpuLinearSystem");
  }
  void requireScalar(unsigned beginPos, unsigned endPos, const char *what) const
  {
    if (beginPos != 0 || endPos != numDof_)
      throw std::runtime_error(std::string(what) + ": only beginPos = 0, endPos = numDof is supported");
  }
  static FILE *openFile(const std::string &path)
  {
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw std::runtime_error("cannot open " + path);
    return f;
  }
  static void writeDense(const std::string &path, const double *v, int64_t n)
  {
    FILE *f = openFile(path);
    std::fprintf(f, "%%%%MatrixMarket matrix array real general\n%lld 1\n", (long long)n);
    for (int64_t i = 0; i < n; ++i) std::fprintf(f, "%.17e\n", v[i]);
    std::fclose(f);
  }

  Realm &realm_;
  nalu_gpu_linsys *h_;
  bool inConstruction_;
  int writeCounter_;
  const unsigned numDof_;
  const std::string eqSysName_;
  LinearSolverConfig *linearSolver_;
  int linearSolveIterations_;
  double nonLinearResidual_, linearResidual_, firstNonLinearResidual_, scaledNonLinearResidual_;
  double timerPrecond_;
  int64_t numRows_ = 0, numCols_ = 0, nnz_ = 0, numEdges_;
  bool firstTimeStepSolve_;
  std::vector<double> residualHistory_;

public:
  bool provideOutput_;
};

// ---------------------------------------------------------------------------------------
// include/SolverAlgorithm.h:34-55.  execute() is one bulk call; apply_coeff is the host slow
// path for the algorithms that still produce one small dense block at a time.
class SolverAlgorithm
{
public:
  explicit SolverAlgorithm(GpuLinearSystem &linsys) : linsys_(linsys) {}
  virtual ~SolverAlgorithm() {}
  virtual void execute() = 0;
  virtual void initialize_connectivity() = 0;

protected:
  void apply_coeff(const std::vector<Entity> &sym_meshobj, std::vector<int> &scratchIds,
                   std::vector<double> &scratchVals, const std::vector<double> &rhs,
                   const std::vector<double> &lhs, const char *trace_tag = 0)
  {
    linsys_.sumInto(sym_meshobj, scratchIds, scratchVals, rhs, lhs, trace_tag);
  }
  GpuLinearSystem &linsys_;
};

// src/AssembleContinuityEdgeSolverAlgorithm.C:34-209
class AssembleContinuityEdgeSolverAlgorithm : public SolverAlgorithm
{
public:
  AssembleContinuityEdgeSolverAlgorithm(GpuLinearSystem &linsys, const std::vector<int> &edgeNodes,
                                        const nalu_gpu_continuity_fields &fields)
    : SolverAlgorithm(linsys), edgeNodes_(edgeNodes), fields_(fields), dt_(1.0), gamma1_(1.0), nocFac_(1.0),
      interpTogether_(1.0), scatterMode_(NALU_GPU_SCATTER_ROWS)
  {
  }
  void initialize_connectivity() { linsys_.buildEdgeToNodeGraph(edgeNodes_); }  // :62-66
  void execute() { linsys_.assembleContinuityEdges(fields_, dt_, gamma1_, nocFac_, interpTogether_, scatterMode_); }
  std::vector<int> edgeNodes_;
  nalu_gpu_continuity_fields fields_;
  double dt_, gamma1_, nocFac_, interpTogether_;  // :84-94
  int scatterMode_;
};

// src/AssembleScalarEdgeSolverAlgorithm.C:40-315
class AssembleScalarEdgeSolverAlgorithm : public SolverAlgorithm
{
public:
  AssembleScalarEdgeSolverAlgorithm(GpuLinearSystem &linsys, const std::vector<int> &edgeNodes,
                                    const nalu_gpu_scalar_fields &fields)
    : SolverAlgorithm(linsys), edgeNodes_(edgeNodes), fields_(fields), scatterMode_(NALU_GPU_SCATTER_ROWS)
  {
    // src/SolutionOptions.C:32-51 defaults
    opts_.alpha = 0.0; opts_.alpha_upw = 1.0; opts_.ho_upwind = 1.0; opts_.use_limiter = 0;
    opts_.peclet_kind = NALU_GPU_PECLET_CLASSIC; opts_.hybrid_factor = 0.0;
    opts_.tanh_c1 = 2.0; opts_.tanh_c2 = 4.0;
  }
  void initialize_connectivity() { linsys_.buildEdgeToNodeGraph(edgeNodes_); }
  void execute() { linsys_.assembleScalarEdges(fields_, opts_, scatterMode_); }
  std::vector<int> edgeNodes_;
  nalu_gpu_scalar_fields fields_;
  nalu_gpu_scalar_options opts_;
  int scatterMode_;
};

// AssembleNodeSolverAlgorithm + ScalarMassBackwardEulerNodeSuppAlg
// (src/AssembleNodeSolverAlgorithm.C:50-112, src/ScalarMassBackwardEulerNodeSuppAlg.C:65-80)
class ScalarMassBackwardEulerAlgorithm : public SolverAlgorithm
{
public:
  ScalarMassBackwardEulerAlgorithm(GpuLinearSystem &linsys, const EntityVector &nodes, const double *qN,
                                   const double *qNp1, const double *rhoN, const double *rhoNp1,
                                   const double *dualVolume, double dt)
    : SolverAlgorithm(linsys), nodes_(nodes), qN_(qN), qNp1_(qNp1), rhoN_(rhoN), rhoNp1_(rhoNp1),
      dualVolume_(dualVolume), dt_(dt)
  {
  }
  void initialize_connectivity() { linsys_.buildNodeGraph(nodes_); }
  void execute() { linsys_.assembleScalarMassBackwardEuler(qN_, qNp1_, rhoN_, rhoNp1_, dualVolume_, dt_); }
  EntityVector nodes_;
  const double *qN_, *qNp1_, *rhoN_, *rhoNp1_, *dualVolume_;
  double dt_;
};

// src/FixPressureAtNodeAlgorithm.C:59-79 (the node's current pressure is read by the caller)
class FixPressureAtNodeAlgorithm : public SolverAlgorithm
{
public:
  FixPressureAtNodeAlgorithm(GpuLinearSystem &linsys, bool nodeIsLocal, Entity node, double refPressure,
                             double pressureAtNode)
    : SolverAlgorithm(linsys), local_(nodeIsLocal), refNodeList_(1, node), refPressure_(refPressure),
      pressureAtNode_(pressureAtNode)
  {
  }
  void initialize_connectivity()
  {
    if (local_) linsys_.buildDirichletNodeGraph(refNodeList_);
  }
  void execute()
  {
    if (!local_) return;  // the node lives on another rank
    linsys_.resetRows(refNodeList_, 0, 1);
    std::vector<int> scratchIds;
    std::vector<double> scratchVals;
    std::vector<double> rhs(1, refPressure_ - pressureAtNode_), lhs(1, 1.0);
    apply_coeff(refNodeList_, scratchIds, scratchVals, rhs, lhs, "FixPressureAtNodeAlgorithm");
  }
  bool local_;
  std::vector<Entity> refNodeList_;
  double refPressure_, pressureAtNode_;
};

// src/DirichletBC.C:32-49
class DirichletBC : public SolverAlgorithm
{
public:
  DirichletBC(GpuLinearSystem &linsys, const EntityVector &nodes, const double *field, const double *bcValues,
              unsigned beginPos, unsigned endPos)
    : SolverAlgorithm(linsys), nodes_(nodes), field_(field), bcValues_(bcValues), beginPos_(beginPos),
      endPos_(endPos)
  {
  }
  void initialize_connectivity() { linsys_.buildDirichletNodeGraph(nodes_); }
  void execute() { linsys_.applyDirichletBCs(field_, bcValues_, nodes_, beginPos_, endPos_); }
  EntityVector nodes_;
  const double *field_, *bcValues_;
  unsigned beginPos_, endPos_;
};

// src/SolverAlgorithmDriver.C:28-138 (owns its algorithms, as there)
class SolverAlgorithmDriver
{
public:
  ~SolverAlgorithmDriver()
  {
    std::map<std::string, SolverAlgorithm *>::iterator is;
    for (is = solverAlgorithmMap_.begin(); is != solverAlgorithmMap_.end(); ++is) delete is->second;
    std::map<AlgorithmType, SolverAlgorithm *>::iterator ii;
    for (ii = solverAlgMap_.begin(); ii != solverAlgMap_.end(); ++ii) delete ii->second;
    for (ii = solverConstraintAlgMap_.begin(); ii != solverConstraintAlgMap_.end(); ++ii) delete ii->second;
    for (ii = solverDirichAlgMap_.begin(); ii != solverDirichAlgMap_.end(); ++ii) delete ii->second;
  }
  void initialize_connectivity()
  {
    std::map<std::string, SolverAlgorithm *>::iterator itc;
    for (itc = solverAlgorithmMap_.begin(); itc != solverAlgorithmMap_.end(); ++itc)
      itc->second->initialize_connectivity();
    std::map<AlgorithmType, SolverAlgorithm *>::iterator it;
    for (it = solverAlgMap_.begin(); it != solverAlgMap_.end(); ++it) it->second->initialize_connectivity();
    for (it = solverConstraintAlgMap_.begin(); it != solverConstraintAlgMap_.end(); ++it)
      it->second->initialize_connectivity();
    for (it = solverDirichAlgMap_.begin(); it != solverDirichAlgMap_.end(); ++it)
      it->second->initialize_connectivity();
  }
  void execute()
  {
    // interior and boundary contributions, then constraints (they reset rows), Dirichlet last
    std::map<std::string, SolverAlgorithm *>::iterator itc;
    for (itc = solverAlgorithmMap_.begin(); itc != solverAlgorithmMap_.end(); ++itc) itc->second->execute();
    std::map<AlgorithmType, SolverAlgorithm *>::iterator it;
    for (it = solverAlgMap_.begin(); it != solverAlgMap_.end(); ++it) it->second->execute();
    for (it = solverConstraintAlgMap_.begin(); it != solverConstraintAlgMap_.end(); ++it) it->second->execute();
    for (it = solverDirichAlgMap_.begin(); it != solverDirichAlgMap_.end(); ++it) it->second->execute();
  }
  std::map<std::string, SolverAlgorithm *> solverAlgorithmMap_;
  std::map<AlgorithmType, SolverAlgorithm *> solverAlgMap_;
  std::map<AlgorithmType, SolverAlgorithm *> solverConstraintAlgMap_;
  std::map<AlgorithmType, SolverAlgorithm *> solverDirichAlgMap_;
};

// The slice of EquationSystem the path runs through (src/EquationSystem.C:52-77, 263-307)
class EquationSystem
{
public:
  EquationSystem(Realm &realm, const std::string &name, LinearSolverConfig *solver, unsigned numDof = 1)
    : realm_(realm), name_(name), linsys_(new GpuLinearSystem(realm, numDof, name, solver)),
      solverAlgDriver_(new SolverAlgorithmDriver()), timerAssemble_(0.0), timerLoadComplete_(0.0),
      timerSolve_(0.0), timerInit_(0.0), timerPrecond_(0.0), numberLinearIterations_(0), deviceTimers_(false)
  {
  }
  ~EquationSystem()
  {
    delete solverAlgDriver_;
    delete linsys_;
  }
  EquationSystem(const EquationSystem &) = delete;
  EquationSystem &operator=(const EquationSystem &) = delete;

  // <Eq>::initialize (src/LowMachEquationSystem.C:2939-2940)
  void initialize()
  {
    const double t0 = now();
    solverAlgDriver_->initialize_connectivity();
    linsys_->finalizeLinearSystem();
    timerInit_ += now() - t0;
  }

  // src/EquationSystem.C:263-307
  void assemble_and_solve(double *deltaSolution)
  {
    int error = 0;
    double timeA = now();
    linsys_->zeroSystem();
    solverAlgDriver_->execute();
    double timeB = now();
    timerAssemble_ += (timeB - timeA);

    timeA = now();
    linsys_->loadComplete();
    timeB = now();
    timerLoadComplete_ += (timeB - timeA);

    timeA = now();
    error = linsys_->solve(deltaSolution);
    timeB = now();
    timerSolve_ += (timeB - timeA);
    timerPrecond_ += linsys_->get_timer_precond();  // src/EquationSystem.C:292
    numberLinearIterations_ += linsys_->linearSolveIterations();  // update_iteration_statistics

    if (error > 0) std::cout << "Error in " << name_ << "::solve_and_update()  " << std::endl;
  }

  Realm &realm_;
  std::string name_;
  GpuLinearSystem *linsys_;
  SolverAlgorithmDriver *solverAlgDriver_;
  double timerAssemble_, timerLoadComplete_, timerSolve_, timerInit_, timerPrecond_;
  long long numberLinearIterations_;
  // the reference's timers bracket synchronous host code; GPU work is asynchronous, so the
  // buckets only mean device time when each reading synchronises the stream first
  bool deviceTimers_;

private:
  double now()
  {
    if (deviceTimers_) linsys_->synchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }
};

}  // namespace gpu
}  // namespace nalu
}  // namespace sierra

#endif
