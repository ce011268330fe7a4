// internal.hpp -- shared declarations of libnalu_gpu.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "nalu_gpu.h"

namespace nalu_gpu {

// ---- matrix layout in HBM ---------------------------------------------------------
// SELL-C (sliced ELLPACK), C = 256 rows per slice = one 256-thread workgroup (4 waves).
// Entry k of row r lives at  sliceOff[r / C] + k * C + (r % C):
// consecutive lanes touch consecutive addresses for every k (coalesced 512 B per wave
// instruction for values, 256 B for columns), and the k-th entry of row r in the
// Tpetra-ordered CSR (columns ascending) is the k-th entry here, so no CSR<->SELL
// permutation table exists.  Slice width = longest row of the slice; padding entries
// carry value 0 and column = own row.
constexpr int SELL_C = 256;

// per-entry edge tag (same layout as the values): edge id in bits 0..30, bit 31 set when
// the row's node is the edge's R node.
constexpr unsigned TAG_PAD = 0xFFFFFFFFu;   // padding
constexpr unsigned TAG_DIAG = 0xFFFFFFFEu;  // the diagonal entry
constexpr unsigned TAG_OTHER = 0xFFFFFFFDu; // graph entry that no edge maps to
constexpr unsigned TAG_MAX_EDGE = 0x7FFFFFF0u;
constexpr unsigned ORD_IDENTITY = 0xFAC688u;  // entries 0..7 in order, none early

// per-slice descriptor of the burst assembly kernel (finalize, k_burst_desc): everything a wave
// needs to form the addresses of its rows' loads, in three 64-byte scalar loads issued together
constexpr int BURST_W = 8;   // widest slice the burst kernel takes
constexpr int BURST_NE = 6;  // edges per row its bursts take (further edges: tail loop)
constexpr unsigned BURST_IRR_NONE = 0xFFFFFFFFu, BURST_IRR_MANY = 0xFFFFFFFEu;
struct BurstRow {           // the edge list of a row (64 bytes)
  unsigned lane;            // BurstSlice::reg: unused; BurstSlice::irr: lane of the wave's one irregular row,
                            // BURST_IRR_NONE, or BURST_IRR_MANY (several: they read tags / columns themselves)
  unsigned kl;              // 4 bits per edge slot e < BURST_NE: the entry index k of the e-th edge entry
  unsigned info;            // bits 0-3 edge entries of the row, 4-7 entry of the diagonal + 1 (0: none),
                            // 8-15 entries a SET pass zeroes, 16-23 edge entries beyond the BURST_NE-th,
                            // bit 24: the row has graph entries that are neither edges nor the diagonal nor
                            // padding (their columns are not in this list)
  unsigned tb[BURST_NE];    // reg: tBase of slot e (tag = tb[e] + row); irr: the tag itself
  int ko[BURST_NE];         // reg: kOff of slot e (column = row + ko[e]); irr: the column itself
  unsigned pad;
};
struct alignas(64) BurstSlice {
  BurstRow reg;                  // the slice's regular rows
  long long off;                 // = sliceOff
  int w;                         // = sliceWidth
  unsigned pad0;
  unsigned long long irrMask[4]; // exMask | exMaskT per wave
  unsigned pad1[4];
  BurstRow irr[4];               // per wave
};
static_assert(sizeof(BurstRow) == 64 && sizeof(BurstSlice) == 384, "burst descriptor layout");

// a run of slices the marching assembly kernel walks with one workgroup: slice start + t * step,
// t < len; every slice of a run has the same structure (k_march_links), its rows' first edge is the
// last edge of the rows `rowStep` below and its SELL storage lies `offStep` entries further on
struct MarchSeg {
  int start, len, step, rowStep;
  long long offStep;
  int halo, lag;  // halo bit 0 / 1: some row reads the node just below / above the slice through a near slot;
                  // lag: the run qualifies for k_assemble_rows_lag (near slots = the lanes next door, eStep valid)
  unsigned eStep[6];  // edge id (tag) of slot q one slice further on = this slice's + eStep[q] (regular rows)
  int pad2[2];
  unsigned xStep[4][6];  // the same for every wave's one irregular row (its edges in the slots they sit in)
  int pad3[8];
};
static_assert(sizeof(MarchSeg) == 192, "march run layout");
constexpr int MARCH_MAX_LEN = 32;  // (16 until the second half of round 3: a run's first step pays an exposed round trip; 64 measures the same, 128 worse)
// the slot pattern the marching kernel is specialised for (a slice qualifies when, for every row,
// the neighbours of the slots in MARCH_NEAR are rows of the slice and those of the other slots are
// not; an irregular row's edges are placed in the slots of the regular rows' offsets they match):
// first / last slot = the rows below / above, slots 2 and 3 = neighbours inside the slice (on a
// lattice: the line's own x neighbours; "inside" includes the node just below and just above the
// slice, which the kernel stages next to the slice's own data -- a 512-node line is two slices),
// slots 1 and 4 = anything
constexpr unsigned MARCH_NEAR = 0x0Cu;

struct SellView {
  const long long *sliceOff;
  const int *sliceWidth;
  const int *cols;
  double *vals;
  const unsigned *tags;
  int xcdRun;  // slice schedule of the row kernels (xcd_run): slices per XCD run, 0 = identity
  // column-offset compression (see finalize): entry k of every row of slice s that is not
  // flagged in exMask has column row + kOff[sliceOff[s] / C + k], so SpMV skips its cols load
  const int *kOff;
  const unsigned long long *exMask;  // 4 words per slice, bit r%64 of word r/64: row r reads cols
  // the same for the edge tags: tag = tBase + tMul * row (tMul 0 for DIAG/PAD/OTHER, 1 for edges
  // numbered along the rows) unless the row is flagged in exMaskT
  const unsigned *tBase, *tMul;
  const unsigned long long *exMaskT;
  // per-slice entry order of the assembly row kernel (finalize, k_slice_order): entry visited at
  // position i < 8 is (ordWord >> 3 i) & 7, bit 24 + i says that a smaller entry follows later
  // ("early": its diagonal / rhs shares wait in a stash so that they are added in CSR order)
  const unsigned *ordWord;
  const BurstSlice *burst;  // null when some slice is wider than BURST_W
  const MarchSeg *march;    // runs of slices for the marching kernel, or null: the first nMarchPat of
  int nMarch, nMarchPat;    // the nMarch runs have the MARCH_NEAR pattern, the others are single slices
  int marchXcdRun;          // consecutive runs per XCD in the marching kernel's schedule
  // export mode only (else null): edgeOwned[e] == 0 marks an edge another rank assembles; the
  // assembly kernels treat its entries as plain graph entries
  const unsigned char *edgeOwned;
  int nSlices;
  int nRows;
  // entries of every stored row before its padding; null unless some slice is wider than LONG_ROW_W (SpMV then
  // takes the slices wider than that through its long-row path)
  const int *rowLen;
};

// SELL-256 pads a slice to its longest row.  Hex8 edge / element graphs have at most 7 / 27 entries per row; a
// slice wider than this has a skewed row (a hub node, a constraint row) and SpMV must not walk the padding of
// its other 255 rows (solve.hip: long_slice_sum)
constexpr int LONG_ROW_W = 32;

struct EventPair {
  hipEvent_t a, b;
};

struct Timer {
  std::vector<EventPair> pending;
  double totalMs = 0.0;
  long long count = 0;     // launches timed
  long long launched = 0;  // launches seen while profiling (timed or not, see nalu_gpu_set_profiling)
  bool skip = false;       // the launch between the current timer_begin / timer_end is not timed
};

constexpr int MAX_BLOCKS = 16384;  // cap of the reduction kernels' grids (partials per slot)
constexpr int REDUCE_CAP = 256;   // doubles in the reduce buffer
constexpr int PARTIAL_SLOTS = 8;  // simultaneous reductions per kernel

// device-resident solver scalars (one allocation, read by every solver kernel)
struct SolverScalars {
  double rz, rr0, r0norm, alpha, beta, tol, hn;
  int iter, done, j, maxIter, histCap, pad;
  double *alphaHist;  // CG: alpha of every iteration (the x update is applied in batches)
  double aux[8];      // BiCGStab / TFQMR recurrences (solve.hip)
};

}  // namespace nalu_gpu

struct EntBlock {
  int npe;
  bool isEdge;
  std::vector<int> nodes;
};

// bulk sumInto plan (scatter.hip)
struct nalu_gpu_scatter_plan {
  nalu_gpu_linsys *h = nullptr;
  int64_t nEnt = 0;
  int nObj = 0;
  std::vector<int> lids;        // host copy of the connectivity (colouring is lazy)
  int *dRow = nullptr;          // [nEnt * nObj] owned row of local row i of entity e, or -1
  short *dK = nullptr;          // [nEnt * nObj * nObj] entry index of column j in that row, or -1
  long long *dIncPtr = nullptr; // [nOwned + 1] (entity, i) pairs per matrix row ...
  int *dInc = nullptr;          // ... as e * nObj + i, ascending
  int nColors = 0;
  std::vector<int64_t> colorOff;
  int *dColorPerm = nullptr;
};

struct nalu_gpu_linsys {
  int device = 0;
  hipStream_t stream = nullptr;
  bool ownStream = false;
  int numDof = 1;
  int state = 0;  // 0 created, 1 under construction, 2 finalized
  int nOwned = 0, nShared = 0, nGhost = 0, nCols = 0, myRank = 0;
  // NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS: the matrix also stores the rows of the shared-not-owned
  // nodes (rows nOwned .. nOwned + nShared: the reference's sharedNotOwnedMatrix_), assembled from
  // the locally owned edges and added to their owners' rows by loadComplete
  int asmMode = 0;
  int nRowsAll = 0;       // rows stored: nOwned (+ nShared in export mode)
  int nSlicesOwned = 0;   // slices that hold owned rows (what the solver kernels sweep)
  std::vector<unsigned char> edgeOwnedHost;
  unsigned char *dEdgeOwned = nullptr;
  // export plan (built on the first loadComplete of export mode)
  bool exportReady = false;
  bool exportedSinceZero = false;  // loadComplete has exported the shared rows since the last zeroSystem
  std::vector<int64_t> expSendOff, expRecvOff;  // per neighbour, in doubles (value messages)
  std::vector<int> expSendRows, expRecvRows;    // rows per neighbour
  std::vector<int> expSendW, expRecvW;          // message width (entries per row) per neighbour
  int *dExpSendRow = nullptr;                   // shared rows in send order
  int *dExpRecvRow = nullptr;                   // owned row of every received row
  long long *dExpRecvPos = nullptr;             // SELL position of every received entry (-1: padding)
  std::vector<int64_t> expRecvPosOff;           // per neighbour offsets into dExpRecvPos
  double *dExpSend = nullptr, *dExpRecv = nullptr;
  double *dPsA = nullptr, *dPsB = nullptr;  // staging of nalu_gpu_parallel_sum
  size_t psBytes = 0;
  std::vector<int64_t> gids;
  std::vector<int> owners;
  std::vector<EntBlock> blocks;
  int64_t nEdges = 0;

  // host CSR (Tpetra layout)
  std::vector<int64_t> rowPtr;
  std::vector<int> colIdx;
  int64_t nnz = 0;

  // device SELL
  int nSlices = 0;
  std::vector<long long> hSliceOff;
  std::vector<int> hSliceWidth;
  long long sellSize = 0;
  int maxSliceWidth = 0;
  long long *dSliceOff = nullptr;
  int *dSliceWidth = nullptr;
  int *dCols = nullptr;
  double *dVals = nullptr;
  unsigned *dTags = nullptr;
  long long *dDiagPos = nullptr;  // SELL position of each row's diagonal entry (-1: none)
  int xcdRun = 8;  // slices per XCD run of the row-kernel schedule (NG_XCD_RUN)
  int asmVariant = 0;  // 0 row kernel (default), 1 entry-parallel (NG_ASM_ENTRIES), 2 ordered (NG_ASM_ORDER), 3 burst (NG_ASM_BURST), 4 marching (NG_ASM_MARCH; also the automatic choice, see assembly_variant_for)
  int *dKOff = nullptr;
  unsigned long long *dExMask = nullptr, *dExMaskT = nullptr;
  unsigned *dTBase = nullptr, *dTMul = nullptr;
  unsigned *dOrdWord = nullptr;  // nSlices
  nalu_gpu::BurstSlice *dBurst = nullptr;
  nalu_gpu::MarchSeg *dMarch = nullptr;  // marching runs that together cover every slice
  int nMarch = 0, nMarchPat = 0;
  long long marchSlices = 0;             // slices in the nMarchPat pattern runs
  int marchLongest = 0;
  int marchXcdRun = 8;
  bool marchLag = false;  // every pattern run qualifies for the lagged, barrier-free marching kernel
  // groups of four runs of adjacent lines (k_assemble_rows_lag4: the lines' rows next to each other in LDS)
  int *dMarchGroups = nullptr, *dMarchSingles = nullptr;  // 4 run indices per group; the runs left over
  int nMarchGroups = 0, nMarchSingles = 0;
  int *dMarchGroupsP = nullptr;  // every pattern run in groups of up to four adjacent lines, -1 = no member (lagpf kernel)
  int nMarchGroupsP = 0;
  int *dRowLen = nullptr;         // SellView::rowLen
  int *dMarchSinglesP = nullptr;  // the pattern runs without a neighbouring line to group with
  int nMarchSinglesP = 0;
  int marchGY = 4;               // lines per group of dMarchGroupsP (4 or 8; NG_ASM_LAG_GY)
  bool asmExplicit = false;              // an NG_ASM_* variable chose the assembly kernel   // nSlices (null: maxSliceWidth > BURST_W)

  // edges
  int *dEdgeL = nullptr, *dEdgeR = nullptr;
  int *dSlots = nullptr;  // [4][nEdges] SELL positions LL, LR, RL, RR (-1: row not owned)
  int nColors = 0;
  std::vector<int64_t> colorOff;
  int *dColorPerm = nullptr;

  // vectors
  double *dRhs = nullptr;   // nOwned
  double *dX = nullptr;     // nCols
  double *dDinv = nullptr;  // nCols
  bool lazyZero = false;
  bool debug = false;  // realm_.debug(): checkForNaN / checkForZeroRow before every solve

  // solver workspace
  double *dR = nullptr, *dP = nullptr, *dQ = nullptr, *dW = nullptr, *dT = nullptr;
  double *dV = nullptr;  // GMRES basis, (kspace+1) x nCols
  double *dKry = nullptr;  // work vectors of BiCGStab / TFQMR, kryCount x nCols
  int kryCount = 0;
  int vCap = 0;
  double *dPartials = nullptr;  // PARTIAL_SLOTS x MAX_BLOCKS
  double *dSlicePartials = nullptr;  // nSlices (+64): per-slice partials of the SpMV-fused dot
  int *dSliceLists = nullptr;        // interior slices, then boundary slices (multi-rank only)
  int nInterior = 0, nBoundary = 0;
  double *dReduce = nullptr;    // REDUCE_CAP
  nalu_gpu::SolverScalars *dScal = nullptr;
  double *dHist = nullptr;
  double *dAlphaHist = nullptr;  // histCap
  double *dPRing = nullptr;      // CG direction ring: pRingM vectors of nCols
  int pRingM = 0;
  int histCap = 0;
  double *dH = nullptr, *dCs = nullptr, *dSn = nullptr, *dG = nullptr, *dY = nullptr;
  int hCap = 0;
  void *hPinned = nullptr;  // pinned host staging (4 KiB)
  int *dList = nullptr;     // node-list staging for BC kernels
  int64_t listCap = 0;

  // SGS level schedule (precond.hip)
  bool sgsReady = false;
  std::vector<int> sgsFwdOff, sgsBwdOff;
  int *dSgsFwdRows = nullptr, *dSgsBwdRows = nullptr;
  // level-major copies of the strictly lower / upper triangle (precond.hip)
  std::vector<long long> lmFwdEntOff, lmBwdEntOff;
  std::vector<int> lmFwdW, lmBwdW;
  long long lmFwdN = 0, lmBwdN = 0;
  long long *dLmFwdSrc = nullptr, *dLmBwdSrc = nullptr;
  int *dLmFwdCol = nullptr, *dLmBwdCol = nullptr, *dLmTpos = nullptr;
  double *dLmFwdVal = nullptr, *dLmBwdVal = nullptr, *dLmDinvF = nullptr, *dLmDinvB = nullptr;
  double *dLmR = nullptr, *dLmT = nullptr, *dLmZ = nullptr;

  // dataflow SGS sweeps (precond.hip, sgs_flow.hpp): one launch per sweep, chunks of rows wait
  // for each other through the exchange buffers (data-as-flag: all-ones NaN = "not computed yet")
  int flowState = 0;  // 0 not built, 1 ready, -1 not available for this matrix (level launches instead)
  bool flowGaveUp = false;  // a sweep gave up waiting in some solve: level launches from then on
  struct FlowDir {
    void *dChunks = nullptr;  // nalu_gpu::FlowChunk[nChunks], processing order
    int *dRowOf = nullptr, *dPubOf = nullptr, *dCode = nullptr, *dSteps = nullptr;
    unsigned long long *dTicket = nullptr;  // chunks drawn so far (k_flow_sweep: the chunk of a workgroup is its ticket)
    double *dXch = nullptr;  // exchange buffer: the rows other chunks read, in schedule order (nVisible)
    int nVisible = 0;
    unsigned char *dKFirst = nullptr;  // backward only: entry k of compact row p is the (kFirst[p] + k)-th of its matrix row
    double *dVal = nullptr;
    int nChunks = 0, wMax = 0, ring = 0, depth = 0, maxSteps = 0;
    long long nEnt = 0;
    size_t ldsBytes = 0;
  } flow[2];  // forward, backward
  double *dFlowT = nullptr;  // nOwned: t = (D + L)^-1 r between the two sweeps
  bool flowTiled = false;
  int flowTile[4] = {0, 0, 0, 0};  // Q, P, By, Bz of the lattice tiling

  // multicolour SGS (precond.hip): rows bucketed by greedy colour
  bool mcReady = false;
  std::vector<int> mcOff;
  int *dMcRows = nullptr;
  // colour-major copy of the entries a colour's rows read in the forward ([0]: lower colours) and in the backward
  // sweep ([1]: higher colours): per colour w x cnt values / columns (-1: none) / entry numbers inside the SELL row
  struct McDir {
    std::vector<long long> entOff;  // per colour
    std::vector<int> width;
    long long nEnt = 0;
    int *dCol = nullptr;
    unsigned short *dK = nullptr;
    double *dVal = nullptr;
  } mc[2];

  // Chebyshev preconditioner (solve.hip)
  double *dChebV = nullptr;      // nOwned
  double *dChebStart = nullptr;  // nCols: power-method start vector (hash of the global ids)
  double chebLambdaMax = 0.0;

  // halo
  int nNbr = 0;
  std::vector<int> nbrRanks, sendOff, recvOff;
  int *dSendLids = nullptr;
  int *dRecvLids = nullptr;  // indexed halo plan: column of every received value (null: contiguous)
  double *dSendBuf = nullptr, *dRecvBuf = nullptr;
  int64_t sendCount = 0, recvCount = 0;
  nalu_gpu_comm comm{};
  bool hasComm = false;

  // live bulk-sumInto plans (released with the system if the caller has not destroyed them)
  std::vector<nalu_gpu_scatter_plan *> plans;

  // timers
  int profiling = 0;  // 0 off; n >= 1: on, the SpMV timer samples every n-th launch
  nalu_gpu::Timer timers[NALU_GPU_T_COUNT];
  std::vector<hipEvent_t> eventPool;
};

namespace nalu_gpu {

int set_error(int code, const char *fmt, ...);
const char *callback_error();

#define NG_HIP(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return nalu_gpu::set_error(NALU_GPU_EHIP, "%s failed: %s (%s:%d)", #call,        \
                                 hipGetErrorString(e_), __FILE__, __LINE__);           \
  } while (0)

#define NG_REQUIRE(cond, code, ...)                                   \
  do {                                                                \
    if (!(cond)) return nalu_gpu::set_error(code, __VA_ARGS__);       \
  } while (0)

// Makes the handle's device current for the duration of an entry point and restores the
// caller's device on the way out (a host that drives several GPUs, or a PyTorch process whose
// current device differs, keeps its own current device).
struct DeviceGuard {
  int prev = -1, dev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int d) : dev(d)
  {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev);
  }
  ~DeviceGuard()
  {
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// (statements, not a do-while: the guard must live until the entry point returns)
#define NG_DEVICE(devArg_)                                                                      \
  nalu_gpu::DeviceGuard ngGuard_(devArg_);                                                     \
  if (ngGuard_.err != hipSuccess)                                                           \
    return nalu_gpu::set_error(NALU_GPU_EHIP, "hipSetDevice(%d) failed: %s", ngGuard_.dev, \
                               hipGetErrorString(ngGuard_.err))

#define NG_FINALIZED(h)                                                                     \
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");                                            \
  NG_REQUIRE((h)->state == 2, NALU_GPU_ESTATE, "linear system is not finalized");           \
  NG_DEVICE((h)->device)

#define NG_TRY(call)            \
  do {                          \
    int rc_ = (call);           \
    if (rc_ != NALU_GPU_OK) return rc_; \
  } while (0)

inline SellView sell_view(const nalu_gpu_linsys *h)
{
  SellView v;
  v.sliceOff = h->dSliceOff;
  v.sliceWidth = h->dSliceWidth;
  v.cols = h->dCols;
  v.vals = h->dVals;
  v.tags = h->dTags;
  v.xcdRun = h->xcdRun;
  v.kOff = h->dKOff;
  v.exMask = h->dExMask;
  v.tBase = h->dTBase;
  v.tMul = h->dTMul;
  v.exMaskT = h->dExMaskT;
  v.ordWord = h->dOrdWord;
  v.burst = h->dBurst;
  v.march = h->dMarch;
  v.nMarch = h->nMarch;
  v.nMarchPat = h->nMarchPat;
  v.marchXcdRun = h->marchXcdRun;
  v.edgeOwned = h->dEdgeOwned;
  v.rowLen = h->dRowLen;
  v.nSlices = h->nSlicesOwned;  // the solver's view: owned rows only
  v.nRows = h->nOwned;
  return v;
}

// the assembly's view: every stored row (owned, and in export mode the shared-not-owned ones)
inline SellView sell_view_all(const nalu_gpu_linsys *h)
{
  SellView v = sell_view(h);
  v.nSlices = h->nSlices;
  v.nRows = h->nRowsAll;
  return v;
}

// XCD-aware, bijective blockIdx -> item map with a single streaming front.  Workgroups are
// dealt round-robin over the 8 XCDs; runs of G consecutive items go to one XCD (so
// neighbouring rows share an L2) while the eight XCDs advance through memory together.
// (Giving each XCD one contiguous eighth of the matrix instead costs 15-25 % of the HBM
// rate on MI355X: eight separate fronts.)  G <= 0: identity.
__device__ __forceinline__ int xcd_run(int b, int n, int G)
{
  if (G <= 0) return b;
  const int full = n / (8 * G) * (8 * G);
  if (b >= full) return b;
  const int xcd = b & 7, idx = b >> 3;
  return ((idx / G) * 8 + xcd) * G + idx % G;
}

// workgroup -> slice of the row-parallel kernels
__device__ __forceinline__ int sched_slice(const SellView &A, int b)
{
  return xcd_run(b, A.nSlices, A.xcdRun);
}

// the marching kernel is the row-gather kernel of choice when no NG_ASM_* variable says otherwise,
// at least half of the slices sit in pattern runs (lattice-numbered meshes) and the runs average
// four slices or more (a run's first and last rows pay an extra round trip each)
inline bool march_by_default(const nalu_gpu_linsys *h)
{
  return !h->asmExplicit && h->asmVariant == 0 && h->dBurst && h->dMarch && !h->dEdgeOwned &&
         2 * h->marchSlices >= (long long)h->nSlices && h->marchSlices >= 4LL * h->nMarchPat;
}

// the lagged, barrier-free marching kernel runs where every pattern run qualifies (NG_ASM_LAG=0 / 1 forces it
// off / on -- on only where it applies)
inline bool lag_kernel(const nalu_gpu_linsys *h)
{
  if (!h->marchLag) return false;
  const char *e = getenv("NG_ASM_LAG");
  return !(e && atoi(e) == 0);
}

// ... by default in its prefetching form (k_assemble_rows_lagpf: a step's streamed operands requested one step
// ahead by direct-to-LDS loads; NG_ASM_LAG_PF=0 turns it off) on groups of adjacent lines (NG_ASM_LAG_GROUPS=0:
// single runs)
inline bool env_not_zero(const char *name)
{
  const char *e = getenv(name);
  return !e || atoi(e) != 0;
}
inline bool lag_prefetch(const nalu_gpu_linsys *h) { return env_not_zero("NG_ASM_LAG_PF") && h->nEdges >= 64; }
inline bool lag_grouped() { return env_not_zero("NG_ASM_LAG_GROUPS"); }

void timer_begin(nalu_gpu_linsys *h, int which);
void timer_end(nalu_gpu_linsys *h, int which);
int materialize_zero(nalu_gpu_linsys *h);
int ensure_edge_slots(nalu_gpu_linsys *h);
int ensure_colors(nalu_gpu_linsys *h);
int ensure_list(nalu_gpu_linsys *h, int64_t n, const int *lids);
int halo_exchange(nalu_gpu_linsys *h, double *vec);
int halo_begin(nalu_gpu_linsys *h, const double *vec);
int halo_end(nalu_gpu_linsys *h, double *vec);
int export_shared_rows(nalu_gpu_linsys *h);
int agree_counts(nalu_gpu_linsys *h, int n, long long *vals);
// reductions: kernels leave per-block partials in dPartials[slot][block]; this sums them
// in a fixed order into dReduce[0..nslots) and (if a comm is set) all-reduces them.
int finish_reduction(nalu_gpu_linsys *h, int nslots, int nblocks, const double *src = nullptr);
int vec_grid(int64_t n);
void release_plan(nalu_gpu_scatter_plan *p);
int ensure_sgs_levels(nalu_gpu_linsys *h);
int sgs_prepare(nalu_gpu_linsys *h);  // whichever schedule NG_SGS_MODE selects
void sgs_flow_disable(nalu_gpu_linsys *h);
int apply_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone);
int sgs_compute(nalu_gpu_linsys *h);
int ensure_mc_colors(nalu_gpu_linsys *h);
int mc_sgs_compute(nalu_gpu_linsys *h);
int apply_mc_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone);

}  // namespace nalu_gpu
