// core.hip -- handle lifetime, graph construction, CSR/SELL set-up, zero / sumInto /
// Dirichlet rows, halo plumbing and timers of libnalu_gpu.so (gfx950).
#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <numeric>
#include <thread>
#include <unordered_map>

#include "internal.hpp"

using namespace nalu_gpu;

// ------------------------------------------------------------------------- errors
static thread_local std::string g_err;

int nalu_gpu::set_error(int code, const char *fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

extern "C" const char *nalu_gpu_last_error(void) { return g_err.c_str(); }

// what a failing comm callback wants nalu_gpu_last_error() to say (callbacks cannot throw
// through the C ABI); consumed by the next "callback failed" error of this thread
static thread_local std::string g_cbErr;
extern "C" int nalu_gpu_set_callback_error(const char *msg)
{
  g_cbErr = msg ? msg : "";
  return NALU_GPU_OK;
}
const char *nalu_gpu::callback_error()
{
  static thread_local std::string last;
  last = g_cbErr.empty() ? std::string("(the callback left no message)") : g_cbErr;
  g_cbErr.clear();
  return last.c_str();
}
extern "C" int nalu_gpu_version(void) { return 100; }

// ------------------------------------------------------------------------- timers
static hipEvent_t get_event(nalu_gpu_linsys *h)
{
  if (!h->eventPool.empty()) {
    hipEvent_t e = h->eventPool.back();
    h->eventPool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void nalu_gpu::timer_begin(nalu_gpu_linsys *h, int which)
{
  if (!h->profiling) return;
  Timer &tm = h->timers[which];
  // an event pair costs ~4 us of stream time: the one high-frequency timer (SpMV, once per Krylov
  // iteration) may sample every n-th launch
  tm.skip = which == NALU_GPU_T_SPMV && h->profiling > 1 && tm.launched % h->profiling != 0;
  ++tm.launched;
  if (tm.skip) return;
  EventPair p;
  p.a = get_event(h);
  p.b = get_event(h);
  (void)hipEventRecord(p.a, h->stream);
  tm.pending.push_back(p);
}

void nalu_gpu::timer_end(nalu_gpu_linsys *h, int which)
{
  if (!h->profiling || h->timers[which].skip) return;
  (void)hipEventRecord(h->timers[which].pending.back().b, h->stream);
}

static void drain_timers(nalu_gpu_linsys *h)
{
  for (int t = 0; t < NALU_GPU_T_COUNT; ++t) {
    Timer &tm = h->timers[t];
    for (EventPair &p : tm.pending) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
        tm.totalMs += ms;
        tm.count += 1;
      }
      h->eventPool.push_back(p.a);
      h->eventPool.push_back(p.b);
    }
    tm.pending.clear();
  }
}

extern "C" int nalu_gpu_set_profiling(nalu_gpu_linsys *h, int enabled)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  h->profiling = enabled > 0 ? enabled : 0;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_timer_launches(nalu_gpu_linsys *h, int which, int64_t *launched)
{
  NG_REQUIRE(h && launched && which >= 0 && which < NALU_GPU_T_COUNT, NALU_GPU_EINVAL, "bad timer id");
  *launched = h->timers[which].launched;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_synchronize(nalu_gpu_linsys *h)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_DEVICE(h->device);
  NG_HIP(hipStreamSynchronize(h->stream));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_timer(nalu_gpu_linsys *h, int which, double *total_ms, int64_t *count)
{
  NG_REQUIRE(h && which >= 0 && which < NALU_GPU_T_COUNT, NALU_GPU_EINVAL, "bad timer id");
  NG_TRY(nalu_gpu_synchronize(h));
  drain_timers(h);
  if (total_ms) *total_ms = h->timers[which].totalMs;
  if (count) *count = h->timers[which].count;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_reset_timers(nalu_gpu_linsys *h)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_TRY(nalu_gpu_synchronize(h));
  drain_timers(h);
  for (int t = 0; t < NALU_GPU_T_COUNT; ++t) {
    h->timers[t].totalMs = 0.0;
    h->timers[t].count = 0;
    h->timers[t].launched = 0;
  }
  return NALU_GPU_OK;
}

// ----------------------------------------------------------------------- lifetime
extern "C" int nalu_gpu_create(nalu_gpu_linsys **out, int device, int num_dof, void *hip_stream)
{
  NG_REQUIRE(out, NALU_GPU_EINVAL, "out is null");
  NG_REQUIRE(num_dof == 1, NALU_GPU_EUNSUPPORTED,
             "numDof = %d: only the numDof = 1 systems (continuity, scalar transport) are "
             "supported", num_dof);
  int ndev = 0;
  NG_HIP(hipGetDeviceCount(&ndev));
  NG_REQUIRE(device >= 0 && device < ndev, NALU_GPU_EINVAL, "device %d of %d", device, ndev);
  NG_DEVICE(device);
  nalu_gpu_linsys *h = new nalu_gpu_linsys();
  h->device = device;
  h->numDof = num_dof;
  h->stream = (hipStream_t)hip_stream;  // NULL = the device's default stream
  *out = h;
  return NALU_GPU_OK;
}

#define NG_FREE(p)            \
  do {                        \
    if (p) (void)hipFree(p);  \
    p = nullptr;              \
  } while (0)

extern "C" int nalu_gpu_destroy(nalu_gpu_linsys *h)
{
  if (!h) return NALU_GPU_OK;
  nalu_gpu::DeviceGuard ngGuard_(h->device);
  (void)hipStreamSynchronize(h->stream);
  drain_timers(h);
  for (nalu_gpu_scatter_plan *p : h->plans) release_plan(p);  // the structs stay the caller's
  h->plans.clear();
  for (hipEvent_t e : h->eventPool) (void)hipEventDestroy(e);
  NG_FREE(h->dSliceOff); NG_FREE(h->dSliceWidth); NG_FREE(h->dCols); NG_FREE(h->dVals);
  NG_FREE(h->dTags); NG_FREE(h->dDiagPos); NG_FREE(h->dEdgeL); NG_FREE(h->dEdgeR); NG_FREE(h->dSlots);
  NG_FREE(h->dColorPerm); NG_FREE(h->dRhs); NG_FREE(h->dX); NG_FREE(h->dDinv);
  NG_FREE(h->dR); NG_FREE(h->dP); NG_FREE(h->dQ); NG_FREE(h->dW); NG_FREE(h->dT);
  NG_FREE(h->dV); NG_FREE(h->dKry); NG_FREE(h->dPartials); NG_FREE(h->dSlicePartials); NG_FREE(h->dSliceLists); NG_FREE(h->dKOff); NG_FREE(h->dExMask); NG_FREE(h->dTBase); NG_FREE(h->dTMul); NG_FREE(h->dOrdWord); NG_FREE(h->dBurst); NG_FREE(h->dMarch); NG_FREE(h->dMarchGroups); NG_FREE(h->dMarchSingles); NG_FREE(h->dMarchGroupsP); NG_FREE(h->dMarchSinglesP); NG_FREE(h->dRowLen); NG_FREE(h->dExMaskT); NG_FREE(h->dSgsFwdRows); NG_FREE(h->dSgsBwdRows); NG_FREE(h->dLmFwdSrc); NG_FREE(h->dLmBwdSrc); NG_FREE(h->dLmFwdCol); NG_FREE(h->dLmBwdCol); NG_FREE(h->dLmTpos); NG_FREE(h->dLmFwdVal); NG_FREE(h->dLmBwdVal); NG_FREE(h->dLmDinvF); NG_FREE(h->dLmDinvB); NG_FREE(h->dLmR); NG_FREE(h->dLmT); NG_FREE(h->dLmZ); NG_FREE(h->dMcRows); for (auto &m : h->mc) { NG_FREE(m.dCol); NG_FREE(m.dK); NG_FREE(m.dVal); } NG_FREE(h->dAlphaHist); NG_FREE(h->dPRing); NG_FREE(h->dChebV); NG_FREE(h->dChebStart); NG_FREE(h->dReduce); NG_FREE(h->dScal);
  NG_FREE(h->dHist); NG_FREE(h->dH); NG_FREE(h->dCs); NG_FREE(h->dSn); NG_FREE(h->dG);
  for (auto &fd : h->flow) {
    NG_FREE(fd.dTicket); NG_FREE(fd.dChunks); NG_FREE(fd.dRowOf); NG_FREE(fd.dPubOf); NG_FREE(fd.dXch); NG_FREE(fd.dCode); NG_FREE(fd.dSteps); NG_FREE(fd.dKFirst); NG_FREE(fd.dVal);
  }
  NG_FREE(h->dFlowT);
  NG_FREE(h->dY); NG_FREE(h->dList); NG_FREE(h->dSendLids); NG_FREE(h->dSendBuf);
  NG_FREE(h->dRecvBuf); NG_FREE(h->dRecvLids); NG_FREE(h->dEdgeOwned); NG_FREE(h->dExpSendRow); NG_FREE(h->dExpRecvRow);
  NG_FREE(h->dExpRecvPos); NG_FREE(h->dExpSend); NG_FREE(h->dExpRecv); NG_FREE(h->dPsA); NG_FREE(h->dPsB);
  if (h->hPinned) (void)hipHostFree(h->hPinned);
  if (h->ownStream) (void)hipStreamDestroy(h->stream);
  delete h;
  return NALU_GPU_OK;
}

// -------------------------------------------------------------- graph construction
extern "C" int nalu_gpu_set_assembly_mode(nalu_gpu_linsys *h, int mode)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state == 0, NALU_GPU_ESTATE, "set_assembly_mode comes before begin_construction");
  NG_REQUIRE(mode == NALU_GPU_ASSEMBLY_OWNER_COMPUTES || mode == NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS,
             NALU_GPU_EINVAL, "unknown assembly mode %d", mode);
  h->asmMode = mode;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_set_edge_ownership(nalu_gpu_linsys *h, int64_t n_edges, const unsigned char *owned)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state == 1, NALU_GPU_ESTATE, "set_edge_ownership comes between the graph builders and finalize");
  NG_REQUIRE(h->asmMode == NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS, NALU_GPU_ESTATE,
             "edge ownership only matters in export mode (nalu_gpu_set_assembly_mode)");
  NG_REQUIRE(n_edges == h->nEdges && (owned || n_edges == 0), NALU_GPU_EINVAL,
             "edge ownership for %lld edges, the graph has %lld", (long long)n_edges, (long long)h->nEdges);
  h->edgeOwnedHost.assign(owned, owned + n_edges);
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_begin_construction(nalu_gpu_linsys *h, int n_owned, int n_shared,
                                           int n_ghost, const int64_t *gids,
                                           const int *owner_ranks, int my_rank)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  // src/TpetraLinearSystem.C:165: a second begin while under construction is a no-op
  if (h->state == 1) return NALU_GPU_OK;
  NG_REQUIRE(h->state == 0, NALU_GPU_ESTATE,
             "linear system already finalized (the reference deletes and recreates it, "
             "src/LowMachEquationSystem.C:2947-2971)");
  NG_REQUIRE(n_owned >= 0 && n_shared >= 0 && n_ghost >= 0 && gids, NALU_GPU_EINVAL,
             "bad node counts");
  const int64_t n = (int64_t)n_owned + n_shared + n_ghost;
  NG_REQUIRE(n < (int64_t)TAG_MAX_EDGE, NALU_GPU_EINVAL, "too many local nodes");
  // ordering rules of src/TpetraLinearSystem.C:247-285 and :1046-1074
  for (int64_t i = 1; i < n_owned; ++i)
    NG_REQUIRE(gids[i - 1] < gids[i], NALU_GPU_EINVAL,
               "owned nodes must be sorted by global id (local %lld)", (long long)i);
  for (int64_t i = n_owned + 1; i < n_owned + n_shared; ++i)
    NG_REQUIRE(gids[i - 1] < gids[i], NALU_GPU_EINVAL,
               "shared-not-owned nodes must be sorted by global id (local %lld)", (long long)i);
  if (owner_ranks) {
    for (int64_t i = 0; i < n_owned; ++i)
      NG_REQUIRE(owner_ranks[i] == my_rank, NALU_GPU_EINVAL, "owned node %lld has owner %d",
                 (long long)i, owner_ranks[i]);
    for (int64_t i = n_owned; i < n; ++i)
      NG_REQUIRE(owner_ranks[i] != my_rank, NALU_GPU_EINVAL,
                 "non-owned node %lld is owned by this rank", (long long)i);
    for (int64_t i = n_owned + n_shared + 1; i < n; ++i)
      NG_REQUIRE(owner_ranks[i - 1] < owner_ranks[i] ||
                     (owner_ranks[i - 1] == owner_ranks[i] && gids[i - 1] < gids[i]),
                 NALU_GPU_EINVAL, "ghost nodes must be ordered by (owner, global id)");
  }
  h->nOwned = n_owned;
  h->nShared = n_shared;
  h->nGhost = n_ghost;
  h->nCols = (int)n;
  h->nRowsAll = n_owned + (h->asmMode == NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS ? n_shared : 0);
  h->myRank = my_rank;
  h->gids.assign(gids, gids + n);
  if (owner_ranks) h->owners.assign(owner_ranks, owner_ranks + n);
  else h->owners.assign((size_t)n, my_rank);
  h->state = 1;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_dof_status(nalu_gpu_linsys *h, int64_t n, const int *lids, int *status)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state >= 1, NALU_GPU_ESTATE, "getDofStatus needs begin_construction first");
  NG_REQUIRE(n >= 0 && (n == 0 || (lids && status)), NALU_GPU_EINVAL, "bad node list");
  for (int64_t i = 0; i < n; ++i) {
    NG_REQUIRE(lids[i] >= 0 && lids[i] < h->nCols, NALU_GPU_EINVAL,
               "local node %d outside [0,%d)", lids[i], h->nCols);
    // the local numbering is owned | shared-not-owned | ghosted (begin_construction)
    status[i] = lids[i] < h->nOwned ? NALU_GPU_DS_OwnedDOF
              : lids[i] < h->nOwned + h->nShared ? NALU_GPU_DS_SharedNotOwnedDOF
                                                 : NALU_GPU_DS_GhostedDOF;
  }
  return NALU_GPU_OK;
}

static int add_block(nalu_gpu_linsys *h, int64_t n_ent, int npe, const int *nodes, bool isEdge)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state == 1, NALU_GPU_ESTATE, "graph builders need begin_construction first");
  NG_REQUIRE(n_ent >= 0 && npe >= 1 && (nodes || n_ent == 0), NALU_GPU_EINVAL, "bad entity list");
  const int64_t tot = n_ent * npe;
  for (int64_t i = 0; i < tot; ++i)
    NG_REQUIRE(nodes[i] >= 0 && nodes[i] < h->nCols, NALU_GPU_EINVAL,
               "entity %lld references local node %d outside [0,%d)", (long long)(i / npe),
               nodes[i], h->nCols);
  if (isEdge) {
    NG_REQUIRE(h->nEdges + n_ent < (int64_t)TAG_MAX_EDGE, NALU_GPU_EINVAL, "too many edges");
    for (int64_t e = 0; e < n_ent; ++e)
      NG_REQUIRE(nodes[2 * e] != nodes[2 * e + 1], NALU_GPU_EINVAL, "degenerate edge %lld",
                 (long long)e);
    h->nEdges += n_ent;
  }
  EntBlock b;
  b.npe = npe;
  b.isEdge = isEdge;
  b.nodes.assign(nodes, nodes + tot);
  h->blocks.push_back(std::move(b));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_build_node_graph(nalu_gpu_linsys *h, int64_t n, const int *lids)
{
  return add_block(h, n, 1, lids, false);
}
extern "C" int nalu_gpu_suggest_edge_order(int64_t n_edges, const int *edge_nodes, int64_t *perm_out)
{
  NG_REQUIRE(n_edges >= 0 && (n_edges == 0 || (edge_nodes && perm_out)), NALU_GPU_EINVAL, "bad edge list");
  std::vector<int64_t> idx((size_t)n_edges);
  for (int64_t e = 0; e < n_edges; ++e) idx[(size_t)e] = e;
  auto lo = [&](int64_t e) { return std::min(edge_nodes[2 * e], edge_nodes[2 * e + 1]); };
  auto hi = [&](int64_t e) { return std::max(edge_nodes[2 * e], edge_nodes[2 * e + 1]); };
  std::sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) {
    return lo(a) != lo(b) ? lo(a) < lo(b) : (hi(a) != hi(b) ? hi(a) < hi(b) : a < b);
  });
  // slot of an edge = its position among its lower node's upper edges; counting sort by slot
  std::vector<int> slot((size_t)n_edges);
  std::vector<int64_t> count;
  for (int64_t q = 0; q < n_edges; ++q) {
    const int sl = (q > 0 && lo(idx[(size_t)q]) == lo(idx[(size_t)q - 1])) ? slot[(size_t)q - 1] + 1 : 0;
    slot[(size_t)q] = sl;
    if ((size_t)sl >= count.size()) count.resize((size_t)sl + 1, 0);
    ++count[(size_t)sl];
  }
  std::vector<int64_t> start(count.size() + 1, 0);
  for (size_t c = 0; c < count.size(); ++c) start[c + 1] = start[c] + count[c];
  for (int64_t q = 0; q < n_edges; ++q) perm_out[start[(size_t)slot[(size_t)q]]++] = idx[(size_t)q];
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_build_edge_to_node_graph(nalu_gpu_linsys *h, int64_t n_edges,
                                                 const int *edge_nodes)
{
  return add_block(h, n_edges, 2, edge_nodes, true);
}
extern "C" int nalu_gpu_build_elem_to_node_graph(nalu_gpu_linsys *h, int64_t n_elems, int npe,
                                                 const int *elem_nodes)
{
  return add_block(h, n_elems, npe, elem_nodes, false);
}
extern "C" int nalu_gpu_build_dirichlet_node_graph(nalu_gpu_linsys *h, int64_t, const int *)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  return NALU_GPU_OK;  // include/LinearSystem.h:68-75: default no-op
}

template <typename F>
static void parallel_for(int64_t n, F f)
{
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(nt, 16u));
  if (n < 200000) nt = 1;
  if (nt == 1) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  const int64_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const int64_t a = t * chunk, b = std::min(n, a + chunk);
    if (a >= b) break;
    th.emplace_back([=]() { f(a, b); });
  }
  for (auto &t : th) t.join();
}

// CSR (host) -> SELL-C (device): one thread per row
__global__ void k_csr_to_sell(int nRows, const long long *rowPtr, const int *colIdx,
                              const long long *sliceOff, const int *sliceWidth, int *cols,
                              unsigned *tags)
{
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const int slice = row / SELL_C, r = row % SELL_C;
  const long long off = sliceOff[slice];
  const int w = sliceWidth[slice];
  if (row < nRows) {
    const long long b = rowPtr[row];
    const int len = (int)(rowPtr[row + 1] - b);
    for (int k = 0; k < w; ++k) {
      const long long pos = off + (long long)k * SELL_C + r;
      if (k < len) {
        const int c = colIdx[b + k];
        cols[pos] = c;
        tags[pos] = (c == row) ? TAG_DIAG : TAG_OTHER;
      } else {
        cols[pos] = row;
        tags[pos] = TAG_PAD;
      }
    }
  } else {  // rows beyond nRows in the last slice
    for (int k = 0; k < w; ++k) {
      const long long pos = off + (long long)k * SELL_C + r;
      cols[pos] = 0;
      tags[pos] = TAG_PAD;
    }
  }
}

__device__ __forceinline__ long long find_in_row(const long long *sliceOff, const int *sliceWidth,
                                                 const int *cols, int row, int col)
{
  const int slice = row / SELL_C, r = row % SELL_C;
  const long long off = sliceOff[slice];
  const int w = sliceWidth[slice];
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    if (cols[pos] == col) return pos;
  }
  return -1;
}

// tag the two off-diagonal entries of every edge
__global__ void k_tag_edges(long long nEdges, const int *eL, const int *eR, int nOwned,
                            const long long *sliceOff, const int *sliceWidth, const int *cols,
                            unsigned *tags, int *err)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nEdges) return;
  const int L = eL[e], R = eR[e];
  if (L < nOwned) {
    const long long pos = find_in_row(sliceOff, sliceWidth, cols, L, R);
    if (pos < 0) atomicAdd(err, 1);
    else {
      if (tags[pos] != TAG_OTHER) atomicAdd(err + 1, 1);  // duplicate edge
      tags[pos] = (unsigned)e;
    }
  }
  if (R < nOwned) {
    const long long pos = find_in_row(sliceOff, sliceWidth, cols, R, L);
    if (pos < 0) atomicAdd(err, 1);
    else {
      if (tags[pos] != TAG_OTHER) atomicAdd(err + 1, 1);
      tags[pos] = (unsigned)e | 0x80000000u;
    }
  }
}

// burst-assembly descriptor of every slice (BurstSlice): which entries of a regular row are edges,
// in CSR order, with their tag and column offsets; the same list for a wave's irregular row when it
// has exactly one (read from the tag / column arrays here, so that the kernel need not)
__device__ void burst_row_finish(BurstRow &R, unsigned ne, unsigned kd, unsigned zm, unsigned lm)
{
  R.info = ne | (kd << 4) | (zm << 8) | (lm << 16);
}

__global__ void k_burst_desc(int nSlices, int nRows, const long long *sliceOff, const int *sliceWidth, const int *kOff,
                             const unsigned *tBase, const unsigned *tMul, const unsigned long long *exMask,
                             const unsigned long long *exMaskT, const unsigned *tags, const int *cols,
                             BurstSlice *out)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nSlices) return;
  const long long off = sliceOff[s], o = off / SELL_C;
  const int w = sliceWidth[s];
  BurstSlice D;
  memset(&D, 0, sizeof(D));
  D.off = off;
  D.w = w;
  {
    unsigned ne = 0, kd = 0, zm = 0, lm = 0;
    for (int k = 0; k < w && k < BURST_W; ++k) {
      if (tMul[o + k] == 1u) {
        if (ne < (unsigned)BURST_NE) {
          D.reg.kl |= (unsigned)k << (4 * ne);
          D.reg.tb[ne] = tBase[o + k];
          D.reg.ko[ne] = kOff[o + k];
        } else {
          lm |= 1u << k;
        }
        ++ne;
      } else if (tBase[o + k] == TAG_DIAG) {
        kd = (unsigned)k + 1;
      } else {
        zm |= 1u << k;
      }
    }
    burst_row_finish(D.reg, ne, kd, zm, lm);
  }
  for (int wv = 0; wv < 4; ++wv) {
    unsigned long long m = exMask[(size_t)s * 4 + wv] | exMaskT[(size_t)s * 4 + wv];
    // (rows past the end of the matrix are not rows)
    const long long row0 = (long long)s * SELL_C + wv * 64;
    if (row0 + 64 > nRows) m &= row0 >= nRows ? 0ULL : ((1ULL << (nRows - row0)) - 1ULL);
    D.irrMask[wv] = m;
    BurstRow &R = D.irr[wv];
    R.lane = m == 0ULL ? BURST_IRR_NONE : BURST_IRR_MANY;
    if (m != 0ULL && (m & (m - 1ULL)) == 0ULL) {
      int lane = 0;
      while (!((m >> lane) & 1ULL)) ++lane;
      R.lane = (unsigned)lane;
      const long long p0 = off + wv * 64 + lane;
      unsigned ne = 0, kd = 0, zm = 0, lm = 0, other = 0;
      for (int k = 0; k < w && k < BURST_W; ++k) {
        const unsigned tag = tags[p0 + (long long)k * SELL_C];
        if (tag < TAG_OTHER) {
          if (ne < (unsigned)BURST_NE) {
            R.kl |= (unsigned)k << (4 * ne);
            R.tb[ne] = tag;
            R.ko[ne] = cols[p0 + (long long)k * SELL_C];
          } else {
            lm |= 1u << k;
          }
          ++ne;
        } else if (tag == TAG_DIAG) {
          kd = (unsigned)k + 1;
        } else {
          zm |= 1u << k;
          if (tag == TAG_OTHER) other = 1u << 24;
        }
      }
      burst_row_finish(R, ne, kd, zm, lm);
      R.info |= other;
    }
  }
  out[s] = D;
}

// marching links: next[s] = the slice whose rows continue the rows of slice s (row + D, D the
// column offset of the rows' LAST edge, a whole number of slices) when both slices are full, have
// the same structure (regular rows and every wave's one irregular row), the rows' last edge in s
// leads to row + D and their first edge in next[s] back to row; -1 otherwise.
__device__ bool burst_rows_alike(const BurstRow &a, const BurstRow &b, bool regular, int rowA, int rowB)
{
  if (a.lane != b.lane || a.kl != b.kl || a.info != b.info) return false;
  const int ne = (int)(a.info & 15u);
  for (int e = 0; e < ne && e < BURST_NE; ++e) {
    const int da = regular ? a.ko[e] : a.ko[e] - rowA, db = regular ? b.ko[e] : b.ko[e] - rowB;
    if (da != db) return false;
  }
  return true;
}

// does every row of slice s follow the MARCH_NEAR pattern?  (slot e of a regular row: entry e of
// the slice's list; an irregular row's edge goes to the slot whose regular offset it has.)  The
// neighbours of the near slots must be rows of the slice or the node just below / above it.
// Returns 0 (no) or 1 | 2 (the node below is read) | 4 (the node above is read).
__device__ int march_pattern(const BurstSlice &a, int s, int nRows)
{
  const int ne = (int)(a.reg.info & 15u);
  if (ne != BURST_NE || ((a.reg.info >> 16) & 255u) || (s + 1) * SELL_C > nRows) return 0;
  int code = 1;
  for (int wv = 0; wv < 4; ++wv) {
    const BurstRow &x = a.irr[wv];
    if (x.lane == BURST_IRR_MANY) return 0;
    for (int l = 0; l < 64; ++l) {
      const int r = wv * 64 + l;
      if ((a.irrMask[wv] >> l) & 1ULL) {
        // (the wave's one irregular row) every edge must sit on a regular offset
        const int row = s * SELL_C + r, nx = (int)(x.info & 15u);
        if (nx > BURST_NE || ((x.info >> 16) & 255u)) return 0;
        for (int e = 0; e < nx; ++e) {
          int slot = -1;
          for (int q = 0; q < BURST_NE; ++q)
            if (x.ko[e] - row == a.reg.ko[q]) slot = q;
          if (slot < 0) return 0;
          if ((MARCH_NEAR >> slot) & 1u) {
            const int idx = r + a.reg.ko[slot];
            if (idx < -1 || idx > SELL_C) return 0;
            if (idx == -1) code |= 2;
            if (idx == SELL_C) code |= 4;
          }
        }
      } else {
        for (int e = 0; e < BURST_NE; ++e) {
          if (!((MARCH_NEAR >> e) & 1u)) continue;
          const int idx = r + a.reg.ko[e];
          if (idx < -1 || idx > SELL_C) return 0;
          if (idx == -1) code |= 2;
          if (idx == SELL_C) code |= 4;
        }
      }
    }
  }
  return code;
}

__global__ void k_march_links(int nSlices, int nRows, const BurstSlice *B, int *next, long long *offStep, int *pat,
                              unsigned *eStep, int *lagOk, int *yOff)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nSlices) return;
  next[s] = -1;
  offStep[s] = 0;
  lagOk[s] = 0;
  yOff[2 * s] = yOff[2 * s + 1] = 0;
  for (int q = 0; q < 5 * BURST_NE; ++q) eStep[(size_t)s * 5 * BURST_NE + q] = 0u;
  const BurstSlice &a = B[s];
  pat[s] = march_pattern(a, s, nRows);
  if (!pat[s]) return;
  // the lagged kernel takes its in-slice neighbours from the lanes next door
  lagOk[s] = a.reg.ko[2] == -1 && a.reg.ko[3] == 1;
  yOff[2 * s] = a.reg.ko[1];  // the far slots: on a lattice the lines below / above in the plane
  yOff[2 * s + 1] = a.reg.ko[4];
  const int ne = BURST_NE;
  const int D = a.reg.ko[ne - 1];
  if (D <= 0 || D % SELL_C) return;
  const long long sn = (long long)s + D / SELL_C;
  if (sn >= nSlices || march_pattern(B[sn], (int)sn, nRows) != pat[s]) return;
  const BurstSlice &b = B[sn];
  if (a.w != b.w || a.reg.ko[0] != -D || !burst_rows_alike(a.reg, b.reg, true, 0, 0)) return;
  for (int wv = 0; wv < 4; ++wv) {
    if (a.irrMask[wv] != b.irrMask[wv]) return;
    const BurstRow &x = a.irr[wv], &y = b.irr[wv];
    if (x.lane == BURST_IRR_NONE && y.lane == BURST_IRR_NONE) continue;
    if (x.lane >= 64u) return;
    const int rowA = s * SELL_C + wv * 64 + (int)x.lane, rowB = rowA + D;
    const int nx = (int)(x.info & 15u);
    if (nx < 2) return;
    if (!burst_rows_alike(x, y, false, rowA, rowB)) return;
    if (x.ko[nx - 1] != rowA + D || x.ko[0] != rowA - D) return;
  }
  next[s] = (int)sn;
  offStep[s] = b.off - a.off;
  // edge ids one slice further on (the lagged kernel steps them instead of loading tags): regular rows
  // have tag = tb + row, so the step is tb' + (row + D) - (tb + row); a wave's irregular row steps by the
  // difference of its listed tags, kept per slot its edges sit in.  [s][0][q]: regular, [s][1 + wave][q]
  unsigned *es = eStep + (size_t)s * 5 * BURST_NE;
  for (int q = 0; q < BURST_NE; ++q) es[q] = b.reg.tb[q] - a.reg.tb[q] + (unsigned)D;
  for (int wv = 0; wv < 4; ++wv) {
    const BurstRow &x = a.irr[wv], &y = b.irr[wv];
    if (x.lane >= 64u) continue;
    const int rowA = s * SELL_C + wv * 64 + (int)x.lane, nx = (int)(x.info & 15u);
    for (int e = 0; e < nx && e < BURST_NE; ++e)
      for (int q = 0; q < BURST_NE; ++q)
        if (x.ko[e] - rowA == a.reg.ko[q]) es[(1 + wv) * BURST_NE + q] = y.tb[e] - x.tb[e];
  }
}

// Tag-offset compression (the assembly kernels' counterpart of the column offsets, see
// finalize): per (slice, k) find (mul, base) with tag = base + mul * row for most rows --
// (0, DIAG/PAD/OTHER) for the special tags, (1, tag - row) when the caller numbered the edges
// along the rows (slot-major) -- and flag the rows that deviate.  Candidates are the keys of
// four sample lanes; the one most rows agree with wins.  A slice with more than a quarter of
// its rows flagged is flagged whole.
__global__ __launch_bounds__(SELL_C) void k_tag_offsets(int nRows, const long long *sliceOff,
                                                         const int *sliceWidth, const unsigned *tags,
                                                         unsigned *tBase, unsigned *tMul,
                                                         unsigned long long *exMaskT, int enabled)
{
  __shared__ unsigned sMul[4], sBase[4];
  const int slice = blockIdx.x, r = threadIdx.x;
  const int row = slice * SELL_C + r;
  const bool live = row < nRows;
  const long long off = sliceOff[slice];
  const int w = sliceWidth[slice];
  bool ex = !live;
  for (int k = 0; k < w && enabled; ++k) {
    const unsigned tag = tags[off + (long long)k * SELL_C + r];
    const unsigned mul = tag >= TAG_OTHER ? 0u : 1u;
    const unsigned base = tag - mul * (unsigned)row;
    if ((r & 63) == 32) { sMul[r >> 6] = mul; sBase[r >> 6] = base; }
    __syncthreads();
    int best = 0, bestCnt = -1;
    for (int c = 0; c < 4; ++c) {
      const int cnt = __syncthreads_count(live && mul == sMul[c] && base == sBase[c]);
      if (cnt > bestCnt) { bestCnt = cnt; best = c; }
    }
    if (mul != sMul[best] || base != sBase[best]) ex = true;
    if (r == 0) {
      tBase[off / SELL_C + k] = sBase[best];
      tMul[off / SELL_C + k] = sMul[best];
    }
    __syncthreads();
  }
  const int nex = __syncthreads_count(ex && live);
  if (!enabled || nex > SELL_C / 4) ex = true;
  const unsigned long long m = __ballot(ex);
  if ((r & 63) == 0) exMaskT[(size_t)slice * 4 + (r >> 6)] = m;
}

// Entry order of the assembly row kernel, one thread per slice.  Entries whose neighbour lies
// outside the slice for (nearly) every row -- column offset of at least half a slice -- are
// visited first, in ascending k, then the others: on a lattice the gathers z-, y-, y+, z+ go out
// back to back and x-, x+ (served from LDS) follow, so that all users of a nodal line touch it
// within a few microseconds of each other and it is still in L2.  The
// shares are added in CSR order all the same: an entry visited before a smaller one is flagged
// "early" and the kernel stashes its shares (two slots; orders that would need more fall back
// to the identity, as do slices wider than 8 entries).
__global__ void k_slice_order(int nSlices, const long long *sliceOff, const int *sliceWidth,
                              const int *kOff, const unsigned *tBase, const unsigned *tMul,
                              unsigned *ordWord, int enabled)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nSlices) return;
  const int w = sliceWidth[s];
  const long long o = sliceOff[s] / SELL_C;
  unsigned word = ORD_IDENTITY;
  if (enabled && w <= 8) {
    int ord[8], n = 0;
    bool ext[8];
    for (int k = 0; k < w; ++k) {
      const bool edge = tMul[o + k] != 0 || tBase[o + k] < TAG_OTHER;
      const int d = kOff[o + k];
      ext[k] = edge && (d >= SELL_C / 2 || d <= -SELL_C / 2);
    }
    for (int k = 0; k < w; ++k)
      if (ext[k]) ord[n++] = k;
    for (int k = 0; k < w; ++k)
      if (!ext[k]) ord[n++] = k;
    for (int k = w; k < 8; ++k) ord[n++] = k;
    unsigned early = 0;
    int nEarly = 0;
    for (int i = 0; i < 8; ++i)
      for (int j = i + 1; j < 8; ++j)
        if (ord[j] < ord[i]) {
          early |= 1u << i;
          ++nEarly;
          break;
        }
    if (nEarly <= 2) {
      word = early << 24;
      for (int i = 0; i < 8; ++i) word |= (unsigned)ord[i] << (3 * i);
    }
  }
  ordWord[s] = word;
}

// entries of every row before its padding (the k-th entry of a row is its k-th CSR entry: padding comes last)
__global__ __launch_bounds__(SELL_C) void k_row_len(const long long *__restrict__ sliceOff,
                                                     const int *__restrict__ sliceWidth,
                                                     const unsigned *__restrict__ tags, int *__restrict__ rowLen)
{
  const int s = blockIdx.x, w = sliceWidth[s];
  const long long pos0 = sliceOff[s] + threadIdx.x;
  int len = 0;
  for (int k = 0; k < w; ++k) len += tags[pos0 + (long long)k * SELL_C] != TAG_PAD;
  rowLen[(size_t)s * SELL_C + threadIdx.x] = len;
}

__global__ __launch_bounds__(SELL_C) void k_find_diag(int nRows, const long long *sliceOff,
                                                       const int *sliceWidth, const unsigned *tags,
                                                       long long *diagPos)
{
  const int slice = blockIdx.x, r = threadIdx.x;
  const int row = slice * SELL_C + r;
  if (row >= nRows) return;
  const long long off = sliceOff[slice];
  const int w = sliceWidth[slice];
  long long p = -1;
  for (int k = 0; k < w; ++k)
    if (tags[off + (long long)k * SELL_C + r] == TAG_DIAG) p = off + (long long)k * SELL_C + r;
  diagPos[row] = p;
}

extern "C" int nalu_gpu_finalize(nalu_gpu_linsys *h)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  NG_REQUIRE(h->state == 1, NALU_GPU_ESTATE, "finalize needs a system under construction");
  NG_DEVICE(h->device);
  // rows stored: the owned ones and, in export mode, the shared-not-owned ones behind them
  // (row == local node id for both); everything below that says nO means "stored rows"
  const int nO = h->nRowsAll;

  // addConnections + insert_graph_connections (src/TpetraLinearSystem.C:355-371,877-917):
  // the pattern of owned row a is {a} U {b : a and b share an entity}, columns ascending
  // (LocalGraphArrays::insertIndices, include/LinearSolver.h:81-93).
  std::vector<int64_t> cnt((size_t)nO + 1, 0);
  for (const EntBlock &b : h->blocks) {
    const int64_t ne = (int64_t)b.nodes.size() / b.npe;
    for (int64_t e = 0; e < ne; ++e)
      for (int i = 0; i < b.npe; ++i) {
        const int a = b.nodes[e * b.npe + i];
        if (a < nO) cnt[a + 1] += b.npe - 1;
      }
  }
  for (int i = 0; i < nO; ++i) cnt[i + 1] += cnt[i] + 1;  // +1: the diagonal
  // (cnt[i] is now the start of row i's scratch list)
  std::vector<int> scratch((size_t)cnt[nO]);
  std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
  for (int i = 0; i < nO; ++i) scratch[fill[i]++] = i;
  for (const EntBlock &b : h->blocks) {
    const int64_t ne = (int64_t)b.nodes.size() / b.npe;
    for (int64_t e = 0; e < ne; ++e) {
      const int *en = &b.nodes[e * b.npe];
      for (int i = 0; i < b.npe; ++i) {
        const int a = en[i];
        if (a >= nO) continue;
        for (int j = 0; j < b.npe; ++j)
          if (j != i) scratch[fill[a]++] = en[j];
      }
    }
  }
  std::vector<int> len((size_t)nO);
  parallel_for(nO, [&](int64_t r0, int64_t r1) {
    for (int64_t i = r0; i < r1; ++i) {
      int *b = &scratch[cnt[i]], *e = &scratch[cnt[i + 1]];
      std::sort(b, e);
      len[i] = (int)(std::unique(b, e) - b);
    }
  });
  h->rowPtr.assign((size_t)nO + 1, 0);
  for (int i = 0; i < nO; ++i) h->rowPtr[i + 1] = h->rowPtr[i] + len[i];
  h->nnz = h->rowPtr[h->nOwned];  // of the owned matrix (the views the callers get)
  h->colIdx.resize((size_t)h->rowPtr[nO]);
  parallel_for(nO, [&](int64_t r0, int64_t r1) {
    for (int64_t i = r0; i < r1; ++i)
      std::copy(&scratch[cnt[i]], &scratch[cnt[i]] + len[i], &h->colIdx[h->rowPtr[i]]);
  });
  std::vector<int>().swap(scratch);

  // SELL slices
  h->nSlices = (nO + SELL_C - 1) / SELL_C;
  h->nSlicesOwned = (h->nOwned + SELL_C - 1) / SELL_C;
  h->hSliceOff.assign((size_t)h->nSlices + 1, 0);
  h->hSliceWidth.assign((size_t)std::max(h->nSlices, 1), 0);
  for (int s = 0; s < h->nSlices; ++s) {
    int w = 0;
    const int r1 = std::min(nO, (s + 1) * SELL_C);
    for (int r = s * SELL_C; r < r1; ++r) w = std::max(w, len[r]);
    h->hSliceWidth[s] = w;
    h->maxSliceWidth = std::max(h->maxSliceWidth, w);
    h->hSliceOff[s + 1] = h->hSliceOff[s] + (long long)w * SELL_C;
  }
  h->sellSize = h->hSliceOff[h->nSlices];

  const size_t nS = (size_t)std::max(h->nSlices, 1), sz = (size_t)std::max<long long>(h->sellSize, 1);
  NG_HIP(hipMalloc(&h->dSliceOff, sizeof(long long) * (nS + 1)));
  NG_HIP(hipMalloc(&h->dSliceWidth, sizeof(int) * nS));
  NG_HIP(hipMalloc(&h->dCols, sizeof(int) * sz));
  NG_HIP(hipMalloc(&h->dTags, sizeof(unsigned) * sz));
  NG_HIP(hipMalloc(&h->dVals, sizeof(double) * sz));
  NG_HIP(hipMemcpyAsync(h->dSliceOff, h->hSliceOff.data(), sizeof(long long) * (h->nSlices + 1),
                        hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipMemcpyAsync(h->dSliceWidth, h->hSliceWidth.data(), sizeof(int) * nS,
                        hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipMemsetAsync(h->dVals, 0, sizeof(double) * sz, h->stream));
  {
    long long *dRowPtr = nullptr;
    int *dColIdx = nullptr;
    NG_HIP(hipMalloc(&dRowPtr, sizeof(long long) * ((size_t)nO + 1)));
    NG_HIP(hipMalloc(&dColIdx, sizeof(int) * (size_t)std::max<int64_t>(h->rowPtr[nO], 1)));
    NG_HIP(hipMemcpyAsync(dRowPtr, h->rowPtr.data(), sizeof(long long) * ((size_t)nO + 1),
                          hipMemcpyHostToDevice, h->stream));
    NG_HIP(hipMemcpyAsync(dColIdx, h->colIdx.data(), sizeof(int) * (size_t)h->rowPtr[nO],
                          hipMemcpyHostToDevice, h->stream));
    if (h->nSlices > 0)
      k_csr_to_sell<<<h->nSlices, SELL_C, 0, h->stream>>>(nO, dRowPtr, dColIdx, h->dSliceOff,
                                                           h->dSliceWidth, h->dCols, h->dTags);
    NG_HIP(hipGetLastError());
    NG_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(dRowPtr);
    (void)hipFree(dColIdx);
  }

  // Column-offset compression for SpMV.  On meshes with any regularity in the numbering the
  // k-th column of most rows of a slice is row + (one constant per slice and k): the stencil
  // offset.  Store that constant (the most frequent c - row over the slice's rows) per
  // (slice, k) and flag the rows that deviate anywhere ("exceptional": e.g. the two lattice
  // boundary rows of an x-line, whose entries are shifted by one); SpMV computes the column
  // of the other rows instead of loading it, which removes most of the 4 B/entry index
  // stream.  Slices with more than a quarter of their rows exceptional are flagged whole and
  // behave as before.  NG_SPMV_OFFSETS=0 turns it off.
  {
    const bool on = !(getenv("NG_SPMV_OFFSETS") && atoi(getenv("NG_SPMV_OFFSETS")) == 0);
    std::vector<int> kOff((size_t)(h->sellSize / SELL_C) + 1, 0);
    std::vector<unsigned long long> ex((size_t)std::max(h->nSlices, 1) * 4, ~0ULL);
    if (on)
      parallel_for(h->nSlices, [&](int64_t a, int64_t b) {
        std::vector<std::pair<int, int>> votes;
        for (int64_t sl = a; sl < b; ++sl) {
          const int r0 = (int)sl * SELL_C, r1 = std::min(nO, r0 + SELL_C), w = h->hSliceWidth[sl];
          int *off = kOff.data() + h->hSliceOff[sl] / SELL_C;
          unsigned long long m[4] = {0, 0, 0, 0};
          for (int r = r1; r < r0 + SELL_C; ++r) m[(r - r0) >> 6] |= 1ULL << ((r - r0) & 63);
          for (int k = 0; k < w; ++k) {
            votes.clear();
            for (int r = r0; r < r1; ++r) {
              const int64_t len = h->rowPtr[r + 1] - h->rowPtr[r];
              const int d = k < len ? h->colIdx[h->rowPtr[r] + k] - r : 0;  // padding: col = row
              bool found = false;
              for (auto &v : votes)
                if (v.first == d) { ++v.second; found = true; break; }
              if (!found && votes.size() < 16) votes.emplace_back(d, 1);
            }
            int best = 0, bv = -1;
            for (auto &v : votes)
              if (v.second > bv) { best = v.first; bv = v.second; }
            off[k] = best;
            for (int r = r0; r < r1; ++r) {
              const int64_t len = h->rowPtr[r + 1] - h->rowPtr[r];
              const int d = k < len ? h->colIdx[h->rowPtr[r] + k] - r : 0;
              if (d != best) m[(r - r0) >> 6] |= 1ULL << ((r - r0) & 63);
            }
          }
          const int nex = __builtin_popcountll(m[0]) + __builtin_popcountll(m[1]) +
                          __builtin_popcountll(m[2]) + __builtin_popcountll(m[3]);
          if (nex - (r0 + SELL_C - r1) <= SELL_C / 4)
            for (int i = 0; i < 4; ++i) ex[(size_t)sl * 4 + i] = m[i];
        }
      });
    NG_HIP(hipMalloc(&h->dKOff, sizeof(int) * kOff.size()));
    NG_HIP(hipMalloc(&h->dExMask, sizeof(unsigned long long) * ex.size()));
    NG_HIP(hipMemcpy(h->dKOff, kOff.data(), sizeof(int) * kOff.size(), hipMemcpyHostToDevice));
    NG_HIP(hipMemcpy(h->dExMask, ex.data(), sizeof(unsigned long long) * ex.size(), hipMemcpyHostToDevice));
  }

  // slice schedule of the row kernels (tuning knob)
  if (const char *e = getenv("NG_XCD_RUN")) h->xcdRun = atoi(e);
  // experimental variants of the row-gather assembly, read once here
  if (const char *e = getenv("NG_ASM_ENTRIES")) h->asmVariant = atoi(e) ? 1 : 0;
  if (const char *e = getenv("NG_ASM_ORDER")) h->asmVariant = atoi(e) ? 2 : h->asmVariant;
  if (const char *e = getenv("NG_ASM_BURST")) h->asmVariant = atoi(e) ? 3 : h->asmVariant;
  if (const char *e = getenv("NG_ASM_MARCH")) h->asmVariant = atoi(e) ? 4 : h->asmVariant;
  h->asmExplicit = getenv("NG_ASM_ENTRIES") || getenv("NG_ASM_ORDER") || getenv("NG_ASM_BURST") || getenv("NG_ASM_MARCH");

  // edges: SoA copy on the device + per-entry edge tags
  if (h->nEdges > 0) {
    std::vector<int> eL((size_t)h->nEdges), eR((size_t)h->nEdges);
    int64_t e0 = 0;
    for (const EntBlock &b : h->blocks) {
      if (!b.isEdge) continue;
      const int64_t ne = (int64_t)b.nodes.size() / 2;
      for (int64_t e = 0; e < ne; ++e) {
        eL[e0 + e] = b.nodes[2 * e];
        eR[e0 + e] = b.nodes[2 * e + 1];
      }
      e0 += ne;
    }
    NG_HIP(hipMalloc(&h->dEdgeL, sizeof(int) * (size_t)h->nEdges));
    NG_HIP(hipMalloc(&h->dEdgeR, sizeof(int) * (size_t)h->nEdges));
    NG_HIP(hipMemcpy(h->dEdgeL, eL.data(), sizeof(int) * (size_t)h->nEdges, hipMemcpyHostToDevice));
    NG_HIP(hipMemcpy(h->dEdgeR, eR.data(), sizeof(int) * (size_t)h->nEdges, hipMemcpyHostToDevice));
    if (h->asmMode == NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS) {
      if (h->edgeOwnedHost.empty()) h->edgeOwnedHost.assign((size_t)h->nEdges, 1);
      // (an edge block registered after nalu_gpu_set_edge_ownership has no entry in the list)
      NG_REQUIRE((int64_t)h->edgeOwnedHost.size() == h->nEdges, NALU_GPU_ESTATE,
                 "edge ownership was given for %lld edges but the graph has %lld: call nalu_gpu_set_edge_ownership "
                 "after the last edge block", (long long)h->edgeOwnedHost.size(), (long long)h->nEdges);
      NG_HIP(hipMalloc(&h->dEdgeOwned, (size_t)h->nEdges));
      NG_HIP(hipMemcpy(h->dEdgeOwned, h->edgeOwnedHost.data(), (size_t)h->nEdges, hipMemcpyHostToDevice));
    }
    int *dErr = nullptr;
    NG_HIP(hipMalloc(&dErr, 2 * sizeof(int)));
    NG_HIP(hipMemsetAsync(dErr, 0, 2 * sizeof(int), h->stream));
    const long long nb = (h->nEdges + 255) / 256;
    k_tag_edges<<<(unsigned)nb, 256, 0, h->stream>>>(h->nEdges, h->dEdgeL, h->dEdgeR, nO,
                                                      h->dSliceOff, h->dSliceWidth, h->dCols,
                                                      h->dTags, dErr);
    NG_HIP(hipGetLastError());
    int err[2] = {0, 0};
    // the null-stream copy below does not wait for a non-blocking h->stream by itself
    NG_HIP(hipStreamSynchronize(h->stream));
    NG_HIP(hipMemcpy(err, dErr, 2 * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dErr);
    NG_REQUIRE(err[0] == 0, NALU_GPU_EINVAL, "internal: %d edge entries missing from the graph", err[0]);
    NG_REQUIRE(err[1] == 0, NALU_GPU_EINVAL, "%d duplicate edges (same node pair twice)", err[1]);
  }
  {
    const size_t nk = (size_t)(h->sellSize / SELL_C) + 1;
    NG_HIP(hipMalloc(&h->dTBase, sizeof(unsigned) * nk));
    NG_HIP(hipMalloc(&h->dTMul, sizeof(unsigned) * nk));
    NG_HIP(hipMalloc(&h->dExMaskT, sizeof(unsigned long long) * (size_t)std::max(h->nSlices, 1) * 4));
    NG_HIP(hipMemsetAsync(h->dExMaskT, 0xFF, sizeof(unsigned long long) * (size_t)std::max(h->nSlices, 1) * 4, h->stream));
    const int on = !(getenv("NG_TAG_OFFSETS") && atoi(getenv("NG_TAG_OFFSETS")) == 0);
    if (h->nSlices > 0)
      k_tag_offsets<<<h->nSlices, SELL_C, 0, h->stream>>>(nO, h->dSliceOff, h->dSliceWidth, h->dTags,
                                                          h->dTBase, h->dTMul, h->dExMaskT, on);
    NG_HIP(hipGetLastError());
  }
  {
    NG_HIP(hipMalloc(&h->dOrdWord, sizeof(unsigned) * (size_t)std::max(h->nSlices, 1)));
    const int on = 1;
    if (h->nSlices > 0)
      k_slice_order<<<(h->nSlices + 255) / 256, 256, 0, h->stream>>>(h->nSlices, h->dSliceOff, h->dSliceWidth,
                                                                      h->dKOff, h->dTBase, h->dTMul,
                                                                      h->dOrdWord, on);
    NG_HIP(hipGetLastError());
  }
  if (h->nSlices > 0 && h->maxSliceWidth > LONG_ROW_W) {  // skewed rows: SpMV's long-row path needs the row lengths
    NG_HIP(hipMalloc(&h->dRowLen, sizeof(int) * (size_t)h->nSlices * SELL_C));
    k_row_len<<<h->nSlices, SELL_C, 0, h->stream>>>(h->dSliceOff, h->dSliceWidth, h->dTags, h->dRowLen);
    NG_HIP(hipGetLastError());
  }
  if (h->nSlices > 0 && h->maxSliceWidth <= BURST_W) {
    NG_HIP(hipMalloc(&h->dBurst, sizeof(BurstSlice) * (size_t)h->nSlices));
    k_burst_desc<<<(h->nSlices + 255) / 256, 256, 0, h->stream>>>(h->nSlices, nO, h->dSliceOff, h->dSliceWidth, h->dKOff,
                                                                   h->dTBase, h->dTMul, h->dExMask, h->dExMaskT,
                                                                   h->dTags, h->dCols, h->dBurst);
    NG_HIP(hipGetLastError());
    // runs of slices for the marching kernel: follow the links while step and storage stride stay
    // the same, at most MARCH_MAX_LEN slices per run; every slice is in exactly one run
    {
      int *dNext = nullptr, *dPat = nullptr, *dLag = nullptr, *dYOff = nullptr;
      long long *dOffStep = nullptr;
      unsigned *dEStep = nullptr;
      NG_HIP(hipMalloc(&dNext, sizeof(int) * (size_t)h->nSlices));
      NG_HIP(hipMalloc(&dPat, sizeof(int) * (size_t)h->nSlices));
      NG_HIP(hipMalloc(&dLag, sizeof(int) * (size_t)h->nSlices));
      NG_HIP(hipMalloc(&dYOff, sizeof(int) * 2 * (size_t)h->nSlices));
      NG_HIP(hipMalloc(&dOffStep, sizeof(long long) * (size_t)h->nSlices));
      NG_HIP(hipMalloc(&dEStep, sizeof(unsigned) * (size_t)h->nSlices * 5 * BURST_NE));
      k_march_links<<<(h->nSlices + 255) / 256, 256, 0, h->stream>>>(h->nSlices, nO, h->dBurst, dNext, dOffStep, dPat,
                                                                      dEStep, dLag, dYOff);
      NG_HIP(hipGetLastError());
      std::vector<int> nxt((size_t)h->nSlices), pat((size_t)h->nSlices), lagv((size_t)h->nSlices), yoff(2 * (size_t)h->nSlices);
      std::vector<long long> ost((size_t)h->nSlices);
      std::vector<unsigned> est((size_t)h->nSlices * 5 * BURST_NE);
      NG_HIP(hipStreamSynchronize(h->stream));
      NG_HIP(hipMemcpy(nxt.data(), dNext, sizeof(int) * nxt.size(), hipMemcpyDeviceToHost));
      NG_HIP(hipMemcpy(pat.data(), dPat, sizeof(int) * pat.size(), hipMemcpyDeviceToHost));
      NG_HIP(hipMemcpy(lagv.data(), dLag, sizeof(int) * lagv.size(), hipMemcpyDeviceToHost));
      NG_HIP(hipMemcpy(yoff.data(), dYOff, sizeof(int) * yoff.size(), hipMemcpyDeviceToHost));
      NG_HIP(hipMemcpy(ost.data(), dOffStep, sizeof(long long) * ost.size(), hipMemcpyDeviceToHost));
      NG_HIP(hipMemcpy(est.data(), dEStep, sizeof(unsigned) * est.size(), hipMemcpyDeviceToHost));
      (void)hipFree(dNext);
      (void)hipFree(dPat);
      (void)hipFree(dLag);
      (void)hipFree(dYOff);
      (void)hipFree(dOffStep);
      (void)hipFree(dEStep);
      const int maxLen = getenv("NG_MARCH_LEN") ? std::max(1, atoi(getenv("NG_MARCH_LEN"))) : MARCH_MAX_LEN;
      std::vector<MarchSeg> segs, others;
      std::vector<char> seen((size_t)h->nSlices, 0);
      for (int s0 = 0; s0 < h->nSlices; ++s0) {
        if (seen[(size_t)s0]) continue;
        MarchSeg g{};
        g.start = s0;
        g.len = 1;
        g.halo = pat[(size_t)s0] >> 1;
        g.lag = lagv[(size_t)s0];
        seen[(size_t)s0] = 1;
        if (!pat[(size_t)s0]) {  // not the marching pattern: a run of one slice for the general kernel
          others.push_back(g);
          continue;
        }
        int cur = s0;
        while (g.len < maxLen) {
          const int n1 = nxt[(size_t)cur];
          if (n1 < 0 || seen[(size_t)n1]) break;
          const int step = n1 - cur;
          const unsigned *es = &est[(size_t)cur * 5 * BURST_NE];
          if (g.len > 1 && (step != g.step || ost[(size_t)cur] != g.offStep)) break;
          // (a change of the edge-id steps only ends a run that is still fit for the lagged kernel)
          if (g.len > 1 && g.lag && lagv[(size_t)cur] &&
              (memcmp(es, g.eStep, sizeof(g.eStep)) != 0 || memcmp(es + BURST_NE, g.xStep, sizeof(g.xStep)) != 0))
            break;
          if (g.len == 1) {
            memcpy(g.eStep, es, sizeof(g.eStep));
            memcpy(g.xStep, es + BURST_NE, sizeof(g.xStep));
          }
          g.lag = g.lag && lagv[(size_t)cur] && lagv[(size_t)n1];
          g.step = step;
          g.rowStep = step * SELL_C;
          g.offStep = ost[(size_t)cur];
          ++g.len;
          seen[(size_t)n1] = 1;
          cur = n1;
        }
        segs.push_back(g);
      }
      h->nMarchPat = (int)segs.size();
      if (const char *e = getenv("NG_MARCH_XCD")) h->marchXcdRun = std::max(1, atoi(e));
      if (const char *e = getenv("NG_ASM_LAG_GY")) h->marchGY = atoi(e) == 8 ? 8 : 4;
      h->marchSlices = 0;
      h->marchLongest = 0;
      h->marchLag = !segs.empty();
      for (const MarchSeg &g : segs) {
        h->marchSlices += g.len;
        h->marchLongest = std::max(h->marchLongest, g.len);
        h->marchLag = h->marchLag && g.lag;
      }
      // groups of four pattern runs of adjacent lines for k_assemble_rows_lag4: run r with the runs whose
      // first slices lie one, two and three "far offsets" further on (ko[4] = -ko[1] = a whole number of
      // slices: on a lattice the next line of the plane), same length and strides
      {
        std::vector<int> groups, singles, groupsP, singlesP;  // groupsP: the same, but incomplete groups of two or more are kept and padded with -1
        if (h->marchLag) {
          std::unordered_map<int, int> runAt;
          for (int r = 0; r < h->nMarchPat; ++r) runAt[segs[(size_t)r].start] = r;
          std::vector<char> used((size_t)h->nMarchPat, 0);
          for (int r = 0; r < h->nMarchPat; ++r) {
            if (used[(size_t)r]) continue;
            const MarchSeg &g0 = segs[(size_t)r];
            const int up = yoff[2 * (size_t)g0.start + 1], dn = yoff[2 * (size_t)g0.start];
            int member[4] = {r, -1, -1, -1};
            bool ok = up > 0 && up % SELL_C == 0 && dn == -up;
            for (int q = 1; q < 4 && ok; ++q) {
              auto it = runAt.find(g0.start + q * (up / SELL_C));
              ok = it != runAt.end() && !used[(size_t)it->second];
              if (!ok) break;
              const MarchSeg &g = segs[(size_t)it->second];
              ok = g.len == g0.len && g.step == g0.step && g.offStep == g0.offStep && g.rowStep == g0.rowStep &&
                   yoff[2 * (size_t)g.start + 1] == up && yoff[2 * (size_t)g.start] == dn;
              member[q] = it->second;
            }
            if (ok) {
              for (int q = 0; q < 4; ++q) { used[(size_t)member[q]] = 1; groups.push_back(member[q]); }
            } else {
              used[(size_t)r] = 1;
              singles.push_back(r);
            }
          }
          // (the prefetching kernel's groups: GP = 4 lines, one 256-thread workgroup per quarter of the lines;
          // NG_ASM_LAG_GY=8: eight lines, 512-thread workgroups)
          const int GP = h->marchGY;
          std::fill(used.begin(), used.end(), 0);
          for (int r = 0; r < h->nMarchPat; ++r) {
            if (used[(size_t)r]) continue;
            const MarchSeg &g0 = segs[(size_t)r];
            const int up = yoff[2 * (size_t)g0.start + 1], dn = yoff[2 * (size_t)g0.start];
            int member[8] = {r, -1, -1, -1, -1, -1, -1, -1};
            used[(size_t)r] = 1;
            const bool chain = up > 0 && up % SELL_C == 0 && dn == -up;
            for (int q = 1; q < GP && chain; ++q) {
              auto it = runAt.find(g0.start + q * (up / SELL_C));
              if (it == runAt.end() || used[(size_t)it->second]) break;
              const MarchSeg &g = segs[(size_t)it->second];
              if (!(g.len == g0.len && g.step == g0.step && g.offStep == g0.offStep && g.rowStep == g0.rowStep &&
                    yoff[2 * (size_t)g.start + 1] == up && yoff[2 * (size_t)g.start] == dn))
                break;
              member[q] = it->second;
              used[(size_t)it->second] = 1;
            }
            // (a group of ONE line goes to the kernel's single-run form: a wave with both far rows would need
            // a second far load in the middle of its step)
            if (member[1] < 0) singlesP.push_back(r);
            else
              for (int q = 0; q < GP; ++q) groupsP.push_back(member[q]);
          }
        }
        h->nMarchGroups = (int)groups.size() / 4;
        h->nMarchSingles = (int)singles.size();
        h->nMarchGroupsP = (int)groupsP.size() / h->marchGY;
        h->nMarchSinglesP = (int)singlesP.size();
        NG_HIP(hipMalloc(&h->dMarchSinglesP, sizeof(int) * std::max<size_t>(singlesP.size(), 1)));
        if (!singlesP.empty())
          NG_HIP(hipMemcpy(h->dMarchSinglesP, singlesP.data(), sizeof(int) * singlesP.size(), hipMemcpyHostToDevice));
        NG_HIP(hipMalloc(&h->dMarchGroupsP, sizeof(int) * std::max<size_t>(groupsP.size(), 1)));
        if (!groupsP.empty())
          NG_HIP(hipMemcpy(h->dMarchGroupsP, groupsP.data(), sizeof(int) * groupsP.size(), hipMemcpyHostToDevice));
        NG_HIP(hipMalloc(&h->dMarchGroups, sizeof(int) * std::max<size_t>(groups.size(), 1)));
        NG_HIP(hipMalloc(&h->dMarchSingles, sizeof(int) * std::max<size_t>(singles.size(), 1)));
        if (!groups.empty())
          NG_HIP(hipMemcpy(h->dMarchGroups, groups.data(), sizeof(int) * groups.size(), hipMemcpyHostToDevice));
        if (!singles.empty())
          NG_HIP(hipMemcpy(h->dMarchSingles, singles.data(), sizeof(int) * singles.size(), hipMemcpyHostToDevice));
      }
      segs.insert(segs.end(), others.begin(), others.end());
      h->nMarch = (int)segs.size();
      NG_HIP(hipMalloc(&h->dMarch, sizeof(MarchSeg) * segs.size()));
      NG_HIP(hipMemcpy(h->dMarch, segs.data(), sizeof(MarchSeg) * segs.size(), hipMemcpyHostToDevice));
    }
    NG_HIP(hipGetLastError());
  }
  NG_HIP(hipMalloc(&h->dDiagPos, sizeof(long long) * (size_t)std::max(nO, 1)));
  if (h->nSlices > 0)
    k_find_diag<<<h->nSlices, SELL_C, 0, h->stream>>>(nO, h->dSliceOff, h->dSliceWidth, h->dTags, h->dDiagPos);
  NG_HIP(hipGetLastError());
  // entity lists are no longer needed
  std::vector<EntBlock>().swap(h->blocks);

  // vectors + solver workspace that does not depend on the solver configuration
  const size_t nC = (size_t)std::max(h->nCols, 1), nR = (size_t)std::max(nO, 1);
  NG_HIP(hipMalloc(&h->dRhs, sizeof(double) * nR));
  NG_HIP(hipMalloc(&h->dX, sizeof(double) * nC));
  NG_HIP(hipMalloc(&h->dDinv, sizeof(double) * nC));
  NG_HIP(hipMalloc(&h->dR, sizeof(double) * nR));
  NG_HIP(hipMalloc(&h->dP, sizeof(double) * nC));
  NG_HIP(hipMalloc(&h->dQ, sizeof(double) * nR));
  NG_HIP(hipMalloc(&h->dW, sizeof(double) * nC));
  NG_HIP(hipMalloc(&h->dT, sizeof(double) * nC));
  NG_HIP(hipMalloc(&h->dPartials, sizeof(double) * PARTIAL_SLOTS * MAX_BLOCKS));
  NG_HIP(hipMalloc(&h->dSlicePartials, sizeof(double) * (std::max(h->nSlices, 1) + 64)));  // + pre-reduction

  // slices whose rows read shared/ghost columns ("boundary") and the rest ("interior"): the
  // SpMV of the interior slices overlaps the halo exchange (Tpetra's Import in apply)
  h->nInterior = h->nBoundary = 0;
  if (h->nCols > h->nOwned && h->nSlicesOwned > 0) {
    const int nOw = h->nOwned, nSo = h->nSlicesOwned;
    std::vector<char> halo((size_t)nSo, 0);
    parallel_for(nSo, [&](int64_t a, int64_t b) {
      for (int64_t sl = a; sl < b; ++sl) {
        const int r1 = std::min(nOw, (int)(sl + 1) * SELL_C);
        char any = 0;
        for (int64_t k = h->rowPtr[sl * SELL_C]; k < h->rowPtr[r1] && !any; ++k)
          any = h->colIdx[k] >= nOw;
        halo[sl] = any;
      }
    });
    std::vector<int> lists((size_t)nSo);
    int ni = 0;
    for (int sl = 0; sl < nSo; ++sl)
      if (!halo[sl]) lists[ni++] = sl;
    h->nInterior = ni;
    for (int sl = 0; sl < nSo; ++sl)
      if (halo[sl]) lists[ni++] = sl;
    h->nBoundary = nSo - h->nInterior;
    NG_HIP(hipMalloc(&h->dSliceLists, sizeof(int) * (size_t)nSo));
    NG_HIP(hipMemcpy(h->dSliceLists, lists.data(), sizeof(int) * (size_t)nSo, hipMemcpyHostToDevice));
  }
  NG_HIP(hipMalloc(&h->dReduce, sizeof(double) * (REDUCE_CAP + 8)));  // + 8 ints of error counters
  NG_HIP(hipMalloc(&h->dScal, sizeof(SolverScalars)));
  NG_HIP(hipHostMalloc(&h->hPinned, 4096, hipHostMallocDefault));
  NG_HIP(hipMemsetAsync(h->dRhs, 0, sizeof(double) * nR, h->stream));
  NG_HIP(hipMemsetAsync(h->dX, 0, sizeof(double) * nC, h->stream));
  NG_HIP(hipMemsetAsync(h->dP, 0, sizeof(double) * nC, h->stream));
  NG_HIP(hipMemsetAsync(h->dW, 0, sizeof(double) * nC, h->stream));
  NG_HIP(hipMemsetAsync(h->dT, 0, sizeof(double) * nC, h->stream));
  NG_HIP(hipMemsetAsync(h->dDinv, 0, sizeof(double) * nC, h->stream));
  NG_HIP(hipMemsetAsync(h->dReduce, 0, sizeof(double) * REDUCE_CAP, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  h->state = 2;
  return NALU_GPU_OK;
}

// --------------------------------------------------------------------------- views

extern "C" int nalu_gpu_get_sizes(nalu_gpu_linsys *h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz)
{
  NG_FINALIZED(h);
  if (n_rows) *n_rows = h->nOwned;
  if (n_cols) *n_cols = h->nCols;
  if (nnz) *nnz = h->nnz;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_graph(nalu_gpu_linsys *h, int64_t *row_ptr, int *col_idx)
{
  NG_FINALIZED(h);
  if (row_ptr) std::copy(h->rowPtr.begin(), h->rowPtr.begin() + h->nOwned + 1, row_ptr);
  if (col_idx) std::copy(h->colIdx.begin(), h->colIdx.begin() + h->nnz, col_idx);
  return NALU_GPU_OK;
}

// rows of the matrix that take the computed-column / computed-tag fast paths of the row kernels
extern "C" int nalu_gpu_get_layout_info(nalu_gpu_linsys *h, nalu_gpu_layout_info *info)
{
  NG_FINALIZED(h);
  NG_REQUIRE(info, NALU_GPU_EINVAL, "info is null");
  const size_t nw = (size_t)std::max(h->nSlices, 1) * 4;
  std::vector<unsigned long long> ex(nw, ~0ULL), exT(nw, ~0ULL);
  NG_HIP(hipStreamSynchronize(h->stream));
  if (h->nSlices > 0) {
    NG_HIP(hipMemcpy(ex.data(), h->dExMask, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
    NG_HIP(hipMemcpy(exT.data(), h->dExMaskT, sizeof(unsigned long long) * nw, hipMemcpyDeviceToHost));
  }
  int64_t cc = 0, ct = 0;
  for (int r = 0; r < h->nOwned; ++r) {
    const size_t wd = (size_t)(r / SELL_C) * 4 + ((r % SELL_C) >> 6);
    cc += !((ex[wd] >> (r & 63)) & 1ULL);
    ct += !((exT[wd] >> (r & 63)) & 1ULL);
  }
  info->n_slices = h->nSlices;
  info->sell_slots = h->sellSize;
  info->max_slice_width = h->maxSliceWidth;
  info->rows_computed_columns = cc;
  info->rows_computed_tags = ct;
  info->assembly_variant = (march_by_default(h) || h->asmVariant == 4) ? (lag_kernel(h) ? (lag_prefetch(h) ? 6 : 5) : 4) : h->asmVariant;
  info->march_runs = h->nMarchPat;
  info->march_slices = h->marchSlices;
  info->march_longest = h->marchLongest;
  info->xcd_run = h->xcdRun;
  return NALU_GPU_OK;
}

// gather / scatter between the CSR value order and the SELL layout
template <bool TO_CSR>
__global__ void k_sell_csr_values(int nRows, const long long *rowPtr, const long long *sliceOff,
                                  double *sellVals, double *csrVals)
{
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nRows) return;
  const long long off = sliceOff[row / SELL_C];
  const int r = row % SELL_C;
  const long long b = rowPtr[row];
  const int len = (int)(rowPtr[row + 1] - b);
  for (int k = 0; k < len; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    if (TO_CSR) csrVals[b + k] = sellVals[pos];
    else sellVals[pos] = csrVals[b + k];
  }
}

static int csr_values_io(nalu_gpu_linsys *h, double *host, bool toHost)
{
  NG_TRY(materialize_zero(h));
  long long *dRowPtr = nullptr;
  double *dCsr = nullptr;
  const size_t nnz = (size_t)std::max<int64_t>(h->nnz, 1);
  NG_HIP(hipMalloc(&dRowPtr, sizeof(long long) * ((size_t)h->nOwned + 1)));
  NG_HIP(hipMalloc(&dCsr, sizeof(double) * nnz));
  NG_HIP(hipMemcpyAsync(dRowPtr, h->rowPtr.data(), sizeof(long long) * ((size_t)h->nOwned + 1),
                        hipMemcpyHostToDevice, h->stream));
  const int nb = (h->nOwned + 255) / 256;
  if (toHost) {
    if (nb) k_sell_csr_values<true><<<nb, 256, 0, h->stream>>>(h->nOwned, dRowPtr, h->dSliceOff, h->dVals, dCsr);
    NG_HIP(hipGetLastError());
    NG_HIP(hipStreamSynchronize(h->stream));
    NG_HIP(hipMemcpy(host, dCsr, sizeof(double) * (size_t)h->nnz, hipMemcpyDeviceToHost));
  } else {
    NG_HIP(hipMemcpyAsync(dCsr, host, sizeof(double) * (size_t)h->nnz, hipMemcpyHostToDevice, h->stream));
    if (nb) k_sell_csr_values<false><<<nb, 256, 0, h->stream>>>(h->nOwned, dRowPtr, h->dSliceOff, h->dVals, dCsr);
    NG_HIP(hipGetLastError());
    NG_HIP(hipStreamSynchronize(h->stream));
  }
  (void)hipFree(dRowPtr);
  (void)hipFree(dCsr);
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_matrix_values(nalu_gpu_linsys *h, double *values)
{
  NG_FINALIZED(h);
  NG_REQUIRE(values, NALU_GPU_EINVAL, "values is null");
  return csr_values_io(h, values, true);
}

extern "C" int nalu_gpu_set_matrix_values(nalu_gpu_linsys *h, const double *values)
{
  NG_FINALIZED(h);
  NG_REQUIRE(values, NALU_GPU_EINVAL, "values is null");
  return csr_values_io(h, const_cast<double *>(values), false);
}

extern "C" int nalu_gpu_get_rhs(nalu_gpu_linsys *h, double *rhs)
{
  NG_FINALIZED(h);
  NG_TRY(materialize_zero(h));
  NG_HIP(hipStreamSynchronize(h->stream));
  NG_HIP(hipMemcpy(rhs, h->dRhs, sizeof(double) * (size_t)h->nOwned, hipMemcpyDeviceToHost));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_set_rhs(nalu_gpu_linsys *h, const double *rhs)
{
  NG_FINALIZED(h);
  NG_TRY(materialize_zero(h));
  NG_HIP(hipStreamSynchronize(h->stream));
  NG_HIP(hipMemcpy(h->dRhs, rhs, sizeof(double) * (size_t)h->nOwned, hipMemcpyHostToDevice));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_solution(nalu_gpu_linsys *h, double *x)
{
  NG_FINALIZED(h);
  NG_HIP(hipStreamSynchronize(h->stream));
  NG_HIP(hipMemcpy(x, h->dX, sizeof(double) * (size_t)h->nOwned, hipMemcpyDeviceToHost));
  return NALU_GPU_OK;
}

// ------------------------------------------------------- collective error verdicts
// Collective entry points (solve, loadComplete in export mode) must leave on every rank together: a
// rank that returns an error by itself leaves the others inside the next all-reduce / exchange for
// good.  Local error counts therefore go through the comm's all-reduce before anyone decides:
// vals[0..n) (non-negative counts, n <= 8) become their sums over the ranks.  Without a comm this
// is a no-op.  Uses the head of the reduce buffer: call it between, not inside, the solver's
// reductions.
int nalu_gpu::agree_counts(nalu_gpu_linsys *h, int n, long long *vals)
{
  if (!h->hasComm) return NALU_GPU_OK;
  NG_REQUIRE(n >= 1 && n <= 8, NALU_GPU_EINVAL, "agree_counts: n = %d", n);
  double *pin = (double *)((char *)h->hPinned + 3584);
  for (int i = 0; i < n; ++i) pin[i] = (double)vals[i];
  NG_HIP(hipMemcpyAsync(h->dReduce, pin, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  NG_REQUIRE(h->comm.allreduce_sum(h->comm.ctx, n, h->stream) == 0, NALU_GPU_EHIP,
             "allreduce_sum callback failed: %s", callback_error());
  NG_HIP(hipMemcpyAsync(pin, h->dReduce, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) vals[i] = (long long)(pin[i] + 0.5);
  return NALU_GPU_OK;
}

// ------------------------------------------------------------- zero / sumInto / BC
// zeroSystem is lazy: it only marks matrix and rhs as logically zero.  The first bulk
// row-gather assembly then *stores* its rows (no memset, no read-modify-write); any other
// consumer materialises the zeros first.
extern "C" int nalu_gpu_zero_system(nalu_gpu_linsys *h)
{
  NG_FINALIZED(h);
  h->lazyZero = true;
  h->exportedSinceZero = false;
  // sln_->putScalar(0) (src/TpetraLinearSystem.C:1347): the solve starts from x0 = 0 itself
  return NALU_GPU_OK;
}

int nalu_gpu::materialize_zero(nalu_gpu_linsys *h)
{
  if (!h->lazyZero) return NALU_GPU_OK;
  timer_begin(h, NALU_GPU_T_ZERO);
  NG_HIP(hipMemsetAsync(h->dVals, 0, sizeof(double) * (size_t)std::max<long long>(h->sellSize, 1), h->stream));
  NG_HIP(hipMemsetAsync(h->dRhs, 0, sizeof(double) * (size_t)std::max(h->nRowsAll, 1), h->stream));
  timer_end(h, NALU_GPU_T_ZERO);
  h->lazyZero = false;
  return NALU_GPU_OK;
}

struct SumIntoArgs {
  int nPos, nRhs;
  long long pos[64];
  double val[64];
  int row[8];
  double rhs[8];
};

__global__ void k_sum_into(SumIntoArgs a, double *vals, double *rhs)
{
  const int t = threadIdx.x;
  if (t < a.nPos) vals[a.pos[t]] += a.val[t];
  if (t < a.nRhs) rhs[a.row[t]] += a.rhs[t];
}

extern "C" int nalu_gpu_sum_into(nalu_gpu_linsys *h, int n_obj, const int *node_lids,
                                 const double *rhs, const double *lhs)
{
  NG_FINALIZED(h);
  NG_REQUIRE(n_obj >= 1 && n_obj <= 8 && node_lids && rhs && lhs, NALU_GPU_EINVAL,
             "sum_into supports 1..8 entities per call (got %d)", n_obj);
  for (int i = 0; i < n_obj; ++i)
    NG_REQUIRE(node_lids[i] >= 0 && node_lids[i] < h->nCols, NALU_GPU_EINVAL,
               "local node %d outside [0,%d)", node_lids[i], h->nCols);
  NG_TRY(materialize_zero(h));
  // src/TpetraLinearSystem.C:1486-1506: rows beyond maxOwnedRowId_ go to the shared-not-
  // owned matrix there; here those rows are assembled by their owner, so they are skipped.
  // Columns missing from the row are skipped silently (:1370-1382).
  SumIntoArgs a;
  a.nPos = a.nRhs = 0;
  for (int i = 0; i < n_obj; ++i) {
    const int row = node_lids[i];
    if (row >= h->nRowsAll) {
      // :1495-1505: a shared-not-owned row goes to the shared matrix (export mode: handled like an
      // owned row, it is stored); rows of ghosts are dropped silently there and here.  In
      // owner-computes mode the shared rows do not exist: refusing beats losing the contribution.
      NG_REQUIRE(row >= h->nOwned + h->nShared, NALU_GPU_EUNSUPPORTED,
                 "sum_into: local node %d is shared-not-owned; in owner-computes mode its row is assembled by its "
                 "owner (pass the entity there), or use NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS "
                 "(src/TpetraLinearSystem.C:1495-1505, 1686-1707)", row);
      continue;
    }
    const int64_t b = h->rowPtr[row], e = h->rowPtr[row + 1];
    const long long off = h->hSliceOff[row / SELL_C];
    // sum_into_row (:1363-1383) passes once through the row with the columns in ascending
    // order: a column that is not in the row runs the scan to the row's end, so that column
    // AND every larger one of this call are dropped -- reproduced here
    int order[8];
    for (int j = 0; j < n_obj; ++j) order[j] = j;
    std::sort(order, order + n_obj, [&](int x, int y) { return node_lids[x] < node_lids[y]; });
    for (int jj = 0; jj < n_obj; ++jj) {
      const int j = order[jj];
      const int *it = std::lower_bound(&h->colIdx[b], &h->colIdx[b] + (e - b), node_lids[j]);
      if (it == &h->colIdx[b] + (e - b) || *it != node_lids[j]) break;
      const int k = (int)(it - &h->colIdx[b]);
      a.pos[a.nPos] = off + (long long)k * SELL_C + row % SELL_C;
      a.val[a.nPos] = lhs[i * n_obj + j];
      ++a.nPos;
    }
    a.row[a.nRhs] = row;
    a.rhs[a.nRhs] = rhs[i];
    ++a.nRhs;
  }
  // one thread per entry; duplicate positions (same node twice) need serial adds
  bool dup = false;
  for (int i = 0; i < a.nPos && !dup; ++i)
    for (int j = i + 1; j < a.nPos; ++j)
      if (a.pos[i] == a.pos[j]) { dup = true; break; }
  NG_REQUIRE(!dup, NALU_GPU_EINVAL, "sum_into: the same node appears twice in one call");
  if (a.nPos || a.nRhs) k_sum_into<<<1, 64, 0, h->stream>>>(a, h->dVals, h->dRhs);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

int nalu_gpu::ensure_list(nalu_gpu_linsys *h, int64_t n, const int *lids)
{
  if (n > h->listCap) {
    // the stream may still read the old list
    NG_HIP(hipStreamSynchronize(h->stream));
    if (h->dList) (void)hipFree(h->dList);
    h->listCap = std::max<int64_t>(n, 1024);
    NG_HIP(hipMalloc(&h->dList, sizeof(int) * (size_t)h->listCap));
  }
  // pageable source: hipMemcpyAsync stages it before returning
  NG_HIP(hipMemcpyAsync(h->dList, lids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  return NALU_GPU_OK;
}

// applyDirichletBCs / resetRows (src/TpetraLinearSystem.C:1546-1583, 1650-1683)
__global__ void k_replace_rows(SellView A, int nOwned, int n, const int *lids, double diagVal,
                               const double *solution, const double *bc, double *rhs)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int row = lids[t];
  if (row >= A.nRows) return;  // shared copy: its row lives (and is replaced) on the owner
  // export mode stores the shared copy: all zero, diagonal included, rhs 0 (:1565-1580), so that
  // the export adds nothing to the owner's replaced row
  const bool sharedCopy = row >= nOwned;
  if (sharedCopy) diagVal = 0.0;
  const int slice = row / SELL_C, r = row % SELL_C;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    A.vals[pos] = (A.tags[pos] == TAG_DIAG) ? diagVal : 0.0;
  }
  rhs[row] = (bc && !sharedCopy) ? (bc[row] - solution[row]) : 0.0;
}

static int replace_rows(nalu_gpu_linsys *h, int64_t n, const int *lids, double diagVal,
                        const double *sol, const double *bc)
{
  NG_REQUIRE(n >= 0 && (lids || n == 0), NALU_GPU_EINVAL, "bad node list");
  for (int64_t i = 0; i < n; ++i)
    NG_REQUIRE(lids[i] >= 0 && lids[i] < h->nOwned + h->nShared, NALU_GPU_EINVAL,
               "node %d is neither owned nor shared (src/TpetraLinearSystem.C:1558-1561)", lids[i]);
  NG_TRY(materialize_zero(h));
  if (n == 0) return NALU_GPU_OK;
  NG_TRY(ensure_list(h, n, lids));
  timer_begin(h, NALU_GPU_T_BC);
  k_replace_rows<<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(sell_view_all(h), h->nOwned, (int)n,
                                                                     h->dList, diagVal, sol, bc, h->dRhs);
  timer_end(h, NALU_GPU_T_BC);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_apply_dirichlet(nalu_gpu_linsys *h, int64_t n, const int *lids,
                                        const double *solution_dev, const double *bc_dev)
{
  NG_FINALIZED(h);
  NG_REQUIRE(solution_dev && bc_dev, NALU_GPU_EINVAL, "null field");
  return replace_rows(h, n, lids, 1.0, solution_dev, bc_dev);
}

extern "C" int nalu_gpu_reset_rows(nalu_gpu_linsys *h, int64_t n, const int *lids)
{
  NG_FINALIZED(h);
  return replace_rows(h, n, lids, 0.0, nullptr, nullptr);
}

extern "C" int nalu_gpu_load_complete(nalu_gpu_linsys *h)
{
  NG_FINALIZED(h);
  // src/TpetraLinearSystem.C:1699,1706 export shared-not-owned rows to their owner.  In
  // owner-computes mode every owned row was assembled completely by its owner (cut edges are
  // duplicated on both sides), so there is nothing to move; only pending zeros are made real.
  NG_TRY(materialize_zero(h));
  if (h->asmMode == NALU_GPU_ASSEMBLY_EXPORT_SHARED_ROWS) {
    // a second loadComplete without a zeroSystem in between would export -- and add -- the shared rows
    // twice (Tpetra: fillComplete on a matrix that is already fill-complete throws)
    NG_REQUIRE(!h->exportedSinceZero, NALU_GPU_ESTATE,
               "load_complete: the shared-not-owned rows were already exported since the last zeroSystem "
               "(src/TpetraLinearSystem.C:1686-1707 runs once per assembly)");
    NG_TRY(export_shared_rows(h));
    h->exportedSinceZero = true;
  }
  return NALU_GPU_OK;
}

// ------------------------------------------------ export of the shared-not-owned rows
// Tpetra's doExport(sharedNotOwned -> owned, ADD) for matrix and rhs (src/TpetraLinearSystem.C:
// 1699, 1706).  A value message carries, per shared row, [rhs, v_0 .. v_{W-1}] (W = the sender's
// longest shared row, entries in the sender's column order).  The plan -- which owned row and which
// matrix slot every received number is added to -- is built once from a set-up exchange of the
// rows' global ids (as doubles: ids below 2^53 are exact), so the hot part is pack -> exchange ->
// add, one launch per neighbour in neighbour order: a fixed summation order.
namespace {

__global__ void k_export_pack(int n, const int *__restrict__ rows, int W, SellView A,
                              const double *__restrict__ rhs, double *__restrict__ out)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int row = rows[q];
  const int slice = row / SELL_C, r = row % SELL_C;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  double *o = out + (size_t)q * (W + 1);
  o[0] = rhs[row];
  for (int k = 0; k < W; ++k) o[1 + k] = k < w ? A.vals[off + (long long)k * SELL_C + r] : 0.0;
}

__global__ void k_export_add(int n, int W, const double *__restrict__ in, const int *__restrict__ rowOf,
                             const long long *__restrict__ pos, double *vals, double *rhs)
{
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const double *m = in + (size_t)q * (W + 1);
  rhs[rowOf[q]] += m[0];
  for (int k = 0; k < W; ++k) {
    const long long p = pos[(size_t)q * W + k];
    if (p >= 0) vals[p] += m[1 + k];
  }
}

// one neighbour exchange of host data through the comm's exchange_rows (set-up only)
int exchange_host(nalu_gpu_linsys *h, const std::vector<double> &send, const std::vector<int64_t> &so,
                  std::vector<double> &recv, const std::vector<int64_t> &ro)
{
  double *dS = nullptr, *dR = nullptr;
  NG_HIP(hipMalloc(&dS, sizeof(double) * std::max<size_t>(send.size(), 1)));
  NG_HIP(hipMalloc(&dR, sizeof(double) * std::max<size_t>(recv.size(), 1)));
  if (!send.empty())
    NG_HIP(hipMemcpyAsync(dS, send.data(), sizeof(double) * send.size(), hipMemcpyHostToDevice, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  const int rc = h->comm.exchange_rows(h->comm.ctx, dS, so.data(), dR, ro.data(), h->stream);
  if (rc == 0 && !recv.empty())
    (void)hipMemcpyAsync(recv.data(), dR, sizeof(double) * recv.size(), hipMemcpyDeviceToHost, h->stream);
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(dS);
  (void)hipFree(dR);
  NG_REQUIRE(rc == 0, NALU_GPU_EHIP, "exchange_rows callback failed: %s", callback_error());
  return NALU_GPU_OK;
}

int ensure_export_plan(nalu_gpu_linsys *h)
{
  if (h->exportReady) return NALU_GPU_OK;
  NG_REQUIRE(h->hasComm && h->comm.exchange_rows, NALU_GPU_ESTATE,
             "export mode needs a comm with exchange_rows (nalu_gpu_set_comm) and the halo plan");
  NG_REQUIRE(h->recvCount == h->nCols - h->nOwned, NALU_GPU_ESTATE, "export mode: set the halo plan first");
  const int nN = h->nNbr, nO = h->nOwned;
  // What one rank finds wrong with its own lists must not make it leave while the others are inside
  // the exchanges below: the first such finding is kept, the exchanges run all the same (without the
  // offending rows), and the verdict is all-reduced at the end so that every rank returns together.
  char localErr[512] = "";
  auto complain = [&](const char *fmt, auto... args) {
    if (!localErr[0]) snprintf(localErr, sizeof(localErr), fmt, args...);
  };
  // shared rows by owner, in neighbour order (ascending local id = ascending global id inside)
  std::vector<std::vector<int>> rowsTo((size_t)nN);
  int Wloc = 0;
  for (int s = 0; s < h->nShared; ++s) {
    const int lid = nO + s;
    int p = 0;
    while (p < nN && h->nbrRanks[p] != h->owners[lid]) ++p;
    if (p == nN) {
      complain("shared node %d is owned by rank %d, which is no halo neighbour", lid, h->owners[lid]);
      continue;
    }
    rowsTo[p].push_back(lid);
    Wloc = std::max(Wloc, (int)(h->rowPtr[lid + 1] - h->rowPtr[lid]));
  }
  // 1. how many rows of which width come from whom
  std::vector<double> cs((size_t)2 * nN), cr((size_t)2 * nN);
  std::vector<int64_t> two((size_t)nN + 1);
  for (int p = 0; p <= nN; ++p) two[p] = 2 * p;
  for (int p = 0; p < nN; ++p) { cs[2 * p] = (double)rowsTo[p].size(); cs[2 * p + 1] = (double)Wloc; }
  NG_TRY(exchange_host(h, cs, two, cr, two));
  h->expSendRows.assign(nN, 0); h->expRecvRows.assign(nN, 0);
  h->expSendW.assign(nN, Wloc); h->expRecvW.assign(nN, 0);
  std::vector<int64_t> mso((size_t)nN + 1, 0), mro((size_t)nN + 1, 0);
  h->expSendOff.assign((size_t)nN + 1, 0); h->expRecvOff.assign((size_t)nN + 1, 0);
  h->expRecvPosOff.assign((size_t)nN + 1, 0);
  for (int p = 0; p < nN; ++p) {
    h->expSendRows[p] = (int)rowsTo[p].size();
    h->expRecvRows[p] = (int)cr[2 * p];
    h->expRecvW[p] = (int)cr[2 * p + 1];
    mso[p + 1] = mso[p] + (int64_t)h->expSendRows[p] * (Wloc + 2);
    mro[p + 1] = mro[p] + (int64_t)h->expRecvRows[p] * (h->expRecvW[p] + 2);
    h->expSendOff[p + 1] = h->expSendOff[p] + (int64_t)h->expSendRows[p] * (Wloc + 1);
    h->expRecvOff[p + 1] = h->expRecvOff[p] + (int64_t)h->expRecvRows[p] * (h->expRecvW[p] + 1);
    h->expRecvPosOff[p + 1] = h->expRecvPosOff[p] + (int64_t)h->expRecvRows[p] * h->expRecvW[p];
  }
  // 2. the rows' global ids: [row gid, length, column gids ..., -1 padding]
  std::vector<double> ms((size_t)mso[nN], -1.0), mr((size_t)mro[nN], -1.0);
  std::vector<int> sendRow;
  for (int p = 0; p < nN; ++p)
    for (size_t q = 0; q < rowsTo[p].size(); ++q) {
      const int lid = rowsTo[p][q];
      sendRow.push_back(lid);
      double *m = &ms[(size_t)mso[p] + q * (Wloc + 2)];
      const int64_t b = h->rowPtr[lid], len = h->rowPtr[lid + 1] - b;
      m[0] = (double)h->gids[lid];
      m[1] = (double)len;
      for (int64_t k = 0; k < len; ++k) m[2 + k] = (double)h->gids[h->colIdx[b + k]];
    }
  NG_TRY(exchange_host(h, ms, mso, mr, mro));
  // 3. where every received number goes.  (Global id -> local node, an owned node first: a mesh
  // that is periodic across the rank's own boundary holds copies of its own nodes.)
  std::unordered_map<int64_t, int> lidOf;
  lidOf.reserve((size_t)h->nCols * 2);
  for (int i = 0; i < h->nCols; ++i) lidOf.emplace(h->gids[i], i);
  std::vector<int> recvRow;
  std::vector<long long> recvPos((size_t)h->expRecvPosOff[nN], -1);
  long long missing = 0;
  for (int p = 0; p < nN; ++p) {
    const int W = h->expRecvW[p];
    for (int q = 0; q < h->expRecvRows[p]; ++q) {
      const double *m = &mr[(size_t)mro[p] + (size_t)q * (W + 2)];
      auto it = lidOf.find((int64_t)m[0]);
      if (it == lidOf.end() || it->second >= nO) {
        complain("export: rank %d sent a row of global id %lld, which this rank does not own", h->nbrRanks[p],
                 (long long)m[0]);
        recvRow.push_back(0);
        continue;
      }
      const int row = it->second;
      recvRow.push_back(row);
      const int *cb = &h->colIdx[h->rowPtr[row]], *ce = &h->colIdx[h->rowPtr[row + 1]];
      const long long off = h->hSliceOff[row / SELL_C];
      for (int k = 0; k < (int)m[1]; ++k) {
        // (a column's global id may name an owned node and a copy of it: any of them that is in the row)
        const int64_t cg = (int64_t)m[2 + k];
        const int *f = ce;
        auto ic = lidOf.find(cg);
        if (ic != lidOf.end()) {
          f = std::lower_bound(cb, ce, ic->second);
          if (f == ce || *f != ic->second) {
            f = ce;
            for (const int *c = cb; c < ce; ++c)
              if (h->gids[*c] == cg) { f = c; break; }
          }
        }
        if (f == ce) { ++missing; continue; }
        recvPos[(size_t)h->expRecvPosOff[p] + (size_t)q * W + k] = off + (long long)(f - cb) * SELL_C + row % SELL_C;
      }
    }
  }
  if (missing)
    complain("export: %lld entries of received rows have no slot in the owner's graph: pass every edge that "
             "touches an owned node to the graph builders (the reference exchanges graph rows at finalize)",
             missing);
  long long bad[1] = {localErr[0] ? 1 : 0};
  NG_TRY(agree_counts(h, 1, bad));
  if (bad[0])
    return set_error(NALU_GPU_EINVAL, "%s",
                     localErr[0] ? localErr : "export: another rank found its shared rows inconsistent with the halo plan");
  NG_FREE(h->dExpSendRow); NG_FREE(h->dExpRecvRow); NG_FREE(h->dExpRecvPos); NG_FREE(h->dExpSend); NG_FREE(h->dExpRecv);
  NG_HIP(hipMalloc(&h->dExpSendRow, sizeof(int) * std::max<size_t>(sendRow.size(), 1)));
  NG_HIP(hipMalloc(&h->dExpRecvRow, sizeof(int) * std::max<size_t>(recvRow.size(), 1)));
  NG_HIP(hipMalloc(&h->dExpRecvPos, sizeof(long long) * std::max<size_t>(recvPos.size(), 1)));
  NG_HIP(hipMalloc(&h->dExpSend, sizeof(double) * (size_t)std::max<int64_t>(h->expSendOff[nN], 1)));
  NG_HIP(hipMalloc(&h->dExpRecv, sizeof(double) * (size_t)std::max<int64_t>(h->expRecvOff[nN], 1)));
  if (!sendRow.empty())
    NG_HIP(hipMemcpy(h->dExpSendRow, sendRow.data(), sizeof(int) * sendRow.size(), hipMemcpyHostToDevice));
  if (!recvRow.empty())
    NG_HIP(hipMemcpy(h->dExpRecvRow, recvRow.data(), sizeof(int) * recvRow.size(), hipMemcpyHostToDevice));
  if (!recvPos.empty())
    NG_HIP(hipMemcpy(h->dExpRecvPos, recvPos.data(), sizeof(long long) * recvPos.size(), hipMemcpyHostToDevice));
  h->exportReady = true;
  return NALU_GPU_OK;
}

}  // namespace

int nalu_gpu::export_shared_rows(nalu_gpu_linsys *h)
{
  if (h->nCols == h->nOwned && h->nNbr == 0) return NALU_GPU_OK;  // one rank: nothing is shared
  NG_TRY(ensure_export_plan(h));
  const int nN = h->nNbr;
  const SellView A = sell_view_all(h);
  int rowBase = 0;
  for (int p = 0; p < nN; ++p) {
    const int n = h->expSendRows[p];
    if (n)
      k_export_pack<<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dExpSendRow + rowBase, h->expSendW[p], A, h->dRhs,
                                                            h->dExpSend + h->expSendOff[p]);
    rowBase += n;
  }
  NG_HIP(hipGetLastError());
  NG_REQUIRE(h->comm.exchange_rows(h->comm.ctx, h->dExpSend, h->expSendOff.data(), h->dExpRecv,
                                   h->expRecvOff.data(), h->stream) == 0,
             NALU_GPU_EHIP, "exchange_rows callback failed: %s", callback_error());
  rowBase = 0;
  for (int p = 0; p < nN; ++p) {  // neighbour order: a fixed summation order
    const int n = h->expRecvRows[p];
    if (n)
      k_export_add<<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->expRecvW[p], h->dExpRecv + h->expRecvOff[p],
                                                           h->dExpRecvRow + rowBase,
                                                           h->dExpRecvPos + h->expRecvPosOff[p], h->dVals, h->dRhs);
    rowBase += n;
  }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// ---------------------------------------------------------- edge slots and colours
__global__ void k_edge_slots(long long nEdges, const int *eL, const int *eR, int nOwned,
                             const long long *sliceOff, const int *sliceWidth, const int *cols,
                             int *slots)
{
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nEdges) return;
  const int L = eL[e], R = eR[e];
  long long ll = -1, lr = -1, rl = -1, rr = -1;
  if (L < nOwned) {
    ll = find_in_row(sliceOff, sliceWidth, cols, L, L);
    lr = find_in_row(sliceOff, sliceWidth, cols, L, R);
  }
  if (R < nOwned) {
    rl = find_in_row(sliceOff, sliceWidth, cols, R, L);
    rr = find_in_row(sliceOff, sliceWidth, cols, R, R);
  }
  slots[e] = (int)ll;
  slots[nEdges + e] = (int)lr;
  slots[2 * nEdges + e] = (int)rl;
  slots[3 * nEdges + e] = (int)rr;
}

int nalu_gpu::ensure_edge_slots(nalu_gpu_linsys *h)
{
  if (h->dSlots || h->nEdges == 0) return NALU_GPU_OK;
  NG_REQUIRE(h->sellSize < 2147483647LL, NALU_GPU_EUNSUPPORTED,
             "edge-parallel scatter modes need < 2^31 matrix slots");
  NG_HIP(hipMalloc(&h->dSlots, sizeof(int) * 4 * (size_t)h->nEdges));
  const long long nb = (h->nEdges + 255) / 256;
  k_edge_slots<<<(unsigned)nb, 256, 0, h->stream>>>(h->nEdges, h->dEdgeL, h->dEdgeR, h->nRowsAll,
                                                     h->dSliceOff, h->dSliceWidth, h->dCols,
                                                     h->dSlots);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// greedy edge colouring on the host (set-up only): no two edges of one colour share a node
int nalu_gpu::ensure_colors(nalu_gpu_linsys *h)
{
  if (h->dColorPerm || h->nEdges == 0) return NALU_GPU_OK;
  const int64_t nE = h->nEdges;
  std::vector<int> eL((size_t)nE), eR((size_t)nE);
  NG_HIP(hipMemcpy(eL.data(), h->dEdgeL, sizeof(int) * (size_t)nE, hipMemcpyDeviceToHost));
  NG_HIP(hipMemcpy(eR.data(), h->dEdgeR, sizeof(int) * (size_t)nE, hipMemcpyDeviceToHost));
  std::vector<uint64_t> used((size_t)h->nCols, 0);
  std::vector<unsigned char> color((size_t)nE);
  int nColors = 0;
  for (int64_t e = 0; e < nE; ++e) {
    const uint64_t m = used[eL[e]] | used[eR[e]];
    NG_REQUIRE(m != ~0ull, NALU_GPU_EUNSUPPORTED, "more than 64 edge colours needed");
    const int c = __builtin_ctzll(~m);
    color[e] = (unsigned char)c;
    used[eL[e]] |= 1ull << c;
    used[eR[e]] |= 1ull << c;
    nColors = std::max(nColors, c + 1);
  }
  h->nColors = nColors;
  h->colorOff.assign((size_t)nColors + 1, 0);
  for (int64_t e = 0; e < nE; ++e) h->colorOff[color[e] + 1]++;
  for (int c = 0; c < nColors; ++c) h->colorOff[c + 1] += h->colorOff[c];
  std::vector<int> perm((size_t)nE);
  std::vector<int64_t> fill(h->colorOff.begin(), h->colorOff.end() - 1);
  for (int64_t e = 0; e < nE; ++e) perm[fill[color[e]]++] = (int)e;
  NG_HIP(hipMalloc(&h->dColorPerm, sizeof(int) * (size_t)nE));
  NG_HIP(hipMemcpy(h->dColorPerm, perm.data(), sizeof(int) * (size_t)nE, hipMemcpyHostToDevice));
  return NALU_GPU_OK;
}

// ------------------------------------------------------------------------- halo
extern "C" int nalu_gpu_set_halo_plan(nalu_gpu_linsys *h, int n_nbr, const int *nbr_ranks,
                                      const int *send_offsets, const int *send_lids,
                                      const int *recv_offsets)
{
  return nalu_gpu_set_halo_plan_indexed(h, n_nbr, nbr_ranks, send_offsets, send_lids, recv_offsets, nullptr);
}

extern "C" int nalu_gpu_set_halo_plan_indexed(nalu_gpu_linsys *h, int n_nbr, const int *nbr_ranks,
                                              const int *send_offsets, const int *send_lids,
                                              const int *recv_offsets, const int *recv_lids)
{
  NG_FINALIZED(h);
  NG_REQUIRE(n_nbr >= 0, NALU_GPU_EINVAL, "bad neighbour count");
  NG_REQUIRE(n_nbr == 0 || (nbr_ranks && send_offsets && recv_offsets), NALU_GPU_EINVAL, "null plan");
  const int64_t ns = n_nbr ? send_offsets[n_nbr] : 0, nr = n_nbr ? recv_offsets[n_nbr] : 0;
  NG_REQUIRE(nr == h->nShared + h->nGhost, NALU_GPU_EINVAL,
             "halo plan receives %lld values but there are %d shared+ghost columns",
             (long long)nr, h->nShared + h->nGhost);
  for (int64_t i = 0; i < ns; ++i)
    NG_REQUIRE(send_lids[i] >= 0 && send_lids[i] < h->nOwned, NALU_GPU_EINVAL,
               "send list entry %lld is not an owned row", (long long)i);
  for (int p = 0; p < n_nbr; ++p)
    NG_REQUIRE(send_offsets[p] <= send_offsets[p + 1] && recv_offsets[p] <= recv_offsets[p + 1],
               NALU_GPU_EINVAL, "offsets must be non-decreasing");
  if (recv_lids) {  // every shared / ghost column receives exactly one value
    std::vector<char> seen((size_t)std::max<int64_t>(nr, 1), 0);
    for (int64_t i = 0; i < nr; ++i) {
      const int c = recv_lids[i];
      NG_REQUIRE(c >= h->nOwned && c < h->nCols && !seen[c - h->nOwned], NALU_GPU_EINVAL,
                 "receive list entry %lld (column %d) is not a shared / ghost column, or is named twice",
                 (long long)i, c);
      seen[c - h->nOwned] = 1;
    }
  }
  NG_HIP(hipStreamSynchronize(h->stream));
  NG_FREE(h->dSendLids); NG_FREE(h->dSendBuf); NG_FREE(h->dRecvBuf); NG_FREE(h->dRecvLids);
  h->nNbr = n_nbr;
  h->nbrRanks.assign(nbr_ranks, nbr_ranks + n_nbr);
  h->sendOff.assign(send_offsets, send_offsets + n_nbr + (n_nbr ? 1 : 0));
  h->recvOff.assign(recv_offsets, recv_offsets + n_nbr + (n_nbr ? 1 : 0));
  h->sendCount = ns;
  h->recvCount = nr;
  NG_HIP(hipMalloc(&h->dSendLids, sizeof(int) * (size_t)std::max<int64_t>(ns, 1)));
  NG_HIP(hipMalloc(&h->dSendBuf, sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
  NG_HIP(hipMalloc(&h->dRecvBuf, sizeof(double) * (size_t)std::max<int64_t>(nr, 1)));
  if (ns) NG_HIP(hipMemcpy(h->dSendLids, send_lids, sizeof(int) * (size_t)ns, hipMemcpyHostToDevice));
  if (recv_lids && nr) {
    NG_HIP(hipMalloc(&h->dRecvLids, sizeof(int) * (size_t)nr));
    NG_HIP(hipMemcpy(h->dRecvLids, recv_lids, sizeof(int) * (size_t)nr, hipMemcpyHostToDevice));
  }
  h->exportReady = false;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_halo_plan(nalu_gpu_linsys *h, int *n_nbr, const int **nbr_ranks,
                                      const int **send_offsets, const int **recv_offsets)
{
  NG_FINALIZED(h);
  NG_REQUIRE(h->nCols == h->nOwned || h->recvCount == h->nCols - h->nOwned, NALU_GPU_ESTATE,
             "this system has %d shared/ghost columns and no halo plan yet", h->nCols - h->nOwned);
  if (n_nbr) *n_nbr = h->nNbr;
  if (nbr_ranks) *nbr_ranks = h->nbrRanks.data();
  if (send_offsets) *send_offsets = h->sendOff.data();
  if (recv_offsets) *recv_offsets = h->recvOff.data();
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_set_comm(nalu_gpu_linsys *h, const nalu_gpu_comm *comm)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  if (!comm) {
    h->hasComm = false;
    return NALU_GPU_OK;
  }
  NG_REQUIRE(comm->allreduce_sum && comm->exchange_begin && comm->exchange_end, NALU_GPU_EINVAL,
             "all three callbacks are required");
  h->comm = *comm;
  h->hasComm = true;
  h->exportReady = false;  // (the plan's set-up exchange runs through the new comm)
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_get_comm_buffers(nalu_gpu_linsys *h, double **send_buf, int64_t *send_count,
                                         double **recv_buf, int64_t *recv_count,
                                         double **reduce_buf, int *reduce_cap)
{
  NG_FINALIZED(h);
  if (send_buf) *send_buf = h->dSendBuf;
  if (send_count) *send_count = h->sendCount;
  if (recv_buf) *recv_buf = h->dRecvBuf;
  if (recv_count) *recv_count = h->recvCount;
  if (reduce_buf) *reduce_buf = h->dReduce;
  if (reduce_cap) *reduce_cap = REDUCE_CAP;
  return NALU_GPU_OK;
}

__global__ void k_pack(long long n, const int *lids, const double *vec, double *buf)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = vec[lids[i]];
}

__global__ void k_unpack(long long n, const int *lids, const double *buf, double *vec)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vec[lids[i]] = buf[i];
}

int nalu_gpu::halo_begin(nalu_gpu_linsys *h, const double *vec)
{
  if (h->nCols == h->nOwned) return NALU_GPU_OK;
  NG_REQUIRE(h->hasComm && h->recvCount == h->nCols - h->nOwned, NALU_GPU_ESTATE,
             "this system has %d shared/ghost columns: set_halo_plan and set_comm first",
             h->nCols - h->nOwned);
  if (h->sendCount)
    k_pack<<<(unsigned)((h->sendCount + 255) / 256), 256, 0, h->stream>>>(h->sendCount, h->dSendLids,
                                                                          vec, h->dSendBuf);
  NG_HIP(hipGetLastError());
  NG_REQUIRE(h->comm.exchange_begin(h->comm.ctx, h->stream) == 0, NALU_GPU_EHIP,
             "exchange_begin callback failed: %s", callback_error());
  return NALU_GPU_OK;
}

int nalu_gpu::halo_end(nalu_gpu_linsys *h, double *vec)
{
  if (h->nCols == h->nOwned) return NALU_GPU_OK;
  NG_REQUIRE(h->comm.exchange_end(h->comm.ctx, h->stream) == 0, NALU_GPU_EHIP,
             "exchange_end callback failed: %s", callback_error());
  if (h->dRecvLids) {  // indexed plan: the shared / ghost columns are not grouped by sender
    k_unpack<<<(unsigned)((h->recvCount + 255) / 256), 256, 0, h->stream>>>(h->recvCount, h->dRecvLids, h->dRecvBuf,
                                                                            vec);
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  NG_HIP(hipMemcpyAsync(vec + h->nOwned, h->dRecvBuf, sizeof(double) * (size_t)h->recvCount,
                        hipMemcpyDeviceToDevice, h->stream));
  return NALU_GPU_OK;
}

int nalu_gpu::halo_exchange(nalu_gpu_linsys *h, double *vec)
{
  NG_TRY(halo_begin(h, vec));
  return halo_end(h, vec);
}

extern "C" int nalu_gpu_halo_exchange(nalu_gpu_linsys *h, double *vec_dev)
{
  NG_FINALIZED(h);
  NG_REQUIRE(vec_dev, NALU_GPU_EINVAL, "null vector");
  return halo_exchange(h, vec_dev);
}

// ---------------------------------------------------------------- parallel_sum
// stk::mesh::parallel_sum (src/AssembleNodalGradAlgorithmDriver.C:111, src/ComputeGeometryAlgorithmDriver.C:
// 126-133): every sharer of an entity ends up with the sum of all sharers' values.  As in STK the sum
// is formed on the entity's owner and sent back: (1) every rank sends the values of the entities it
// shares but does not own to their owners, (2) the owner adds them to its own value, neighbour after
// neighbour in ascending rank order (a fixed order: deterministic), (3) the totals travel back the
// same way.  Two neighbour exchanges through the comm's exchange_rows; every sharer holds the
// owner's bits afterwards.
struct nalu_gpu_share_plan {
  nalu_gpu_linsys *h = nullptr;
  int64_t nEntities = 0;
  std::vector<int> so, ro;  // per halo neighbour: owned entities it shares / entities it owns that this rank shares
  int *dSendIds = nullptr, *dRecvIds = nullptr;
};

namespace {
constexpr int PS_MAX_FIELDS = 16;
struct PsFields { double *f[PS_MAX_FIELDS]; };

// ids == nullptr: entity base + i.  Entities from zeroFrom on contribute nothing (ghost copies)
__global__ void k_ps_pack(long long n, const int *__restrict__ ids, int base, int zeroFrom, PsFields F, int nF,
                          double *__restrict__ out)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = ids ? ids[i] : base + (int)i;
  for (int f = 0; f < nF; ++f) out[i * nF + f] = id >= zeroFrom ? 0.0 : F.f[f][id];
}
__global__ void k_ps_add(long long n, const int *__restrict__ ids, PsFields F, int nF, const double *__restrict__ in)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = ids[i];
  for (int f = 0; f < nF; ++f) F.f[f][id] += in[i * nF + f];
}
__global__ void k_ps_unpack(long long n, const int *__restrict__ ids, int base, PsFields F, int nF,
                            const double *__restrict__ in)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = ids ? ids[i] : base + (int)i;
  for (int f = 0; f < nF; ++f) F.f[f][id] = in[i * nF + f];
}

int check_id_lists(const nalu_gpu_linsys *h, const int *off, const int *ids, int64_t nEnt, const char *what)
{
  NG_REQUIRE(off[0] == 0, NALU_GPU_EINVAL, "share plan: %s offsets must start at 0", what);
  for (int p = 0; p < h->nNbr; ++p) {
    NG_REQUIRE(off[p] <= off[p + 1], NALU_GPU_EINVAL, "share plan: %s offsets must not decrease", what);
    std::vector<int> one(ids + off[p], ids + off[p + 1]);
    for (int id : one)
      NG_REQUIRE(id >= 0 && id < nEnt, NALU_GPU_EINVAL, "share plan: %s entity %d outside [0,%lld)", what, id, (long long)nEnt);
    std::sort(one.begin(), one.end());
    NG_REQUIRE(std::adjacent_find(one.begin(), one.end()) == one.end(), NALU_GPU_EINVAL,
               "share plan: an entity is listed twice in the %s list of neighbour rank %d", what, h->nbrRanks[p]);
  }
  return NALU_GPU_OK;
}
}  // namespace

extern "C" int nalu_gpu_share_plan_create(nalu_gpu_linsys *h, const int *send_offsets, const int *send_ids,
                                          const int *recv_offsets, const int *recv_ids, int64_t n_entities,
                                          nalu_gpu_share_plan **out)
{
  NG_FINALIZED(h);
  NG_REQUIRE(out && n_entities >= 0, NALU_GPU_EINVAL, "share plan: bad arguments");
  const int nN = h->nNbr;
  NG_REQUIRE(nN == 0 || (send_offsets && recv_offsets), NALU_GPU_EINVAL,
             "share plan: one pair of lists per halo neighbour (set the halo plan first)");
  const int64_t ns = nN ? send_offsets[nN] : 0, nr = nN ? recv_offsets[nN] : 0;
  NG_REQUIRE((ns == 0 || send_ids) && (nr == 0 || recv_ids), NALU_GPU_EINVAL, "share plan: null list");
  if (nN) {
    NG_TRY(check_id_lists(h, send_offsets, send_ids, n_entities, "owned"));
    NG_TRY(check_id_lists(h, recv_offsets, recv_ids, n_entities, "shared"));
  }
  nalu_gpu_share_plan *P = new nalu_gpu_share_plan();
  P->h = h;
  P->nEntities = n_entities;
  if (nN) {
    P->so.assign(send_offsets, send_offsets + nN + 1);
    P->ro.assign(recv_offsets, recv_offsets + nN + 1);
  } else {
    P->so.assign(1, 0);
    P->ro.assign(1, 0);
  }
  *out = P;  // (the caller destroys it also when a copy below fails)
  NG_HIP(hipMalloc(&P->dSendIds, sizeof(int) * (size_t)std::max<int64_t>(ns, 1)));
  NG_HIP(hipMalloc(&P->dRecvIds, sizeof(int) * (size_t)std::max<int64_t>(nr, 1)));
  if (ns) NG_HIP(hipMemcpy(P->dSendIds, send_ids, sizeof(int) * (size_t)ns, hipMemcpyHostToDevice));
  if (nr) NG_HIP(hipMemcpy(P->dRecvIds, recv_ids, sizeof(int) * (size_t)nr, hipMemcpyHostToDevice));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_share_plan_destroy(nalu_gpu_share_plan *P)
{
  if (!P) return NALU_GPU_OK;
  nalu_gpu::DeviceGuard ngGuard_(P->h->device);
  (void)hipStreamSynchronize(P->h->stream);
  NG_FREE(P->dSendIds); NG_FREE(P->dRecvIds);
  delete P;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_parallel_sum(nalu_gpu_linsys *h, nalu_gpu_share_plan *P, int n_fields,
                                     double *const *fields_dev)
{
  NG_FINALIZED(h);
  NG_REQUIRE(!P || P->h == h, NALU_GPU_EINVAL, "parallel_sum: the plan belongs to another linear system");
  NG_REQUIRE(n_fields >= 1 && n_fields <= PS_MAX_FIELDS && fields_dev, NALU_GPU_EINVAL,
             "parallel_sum: 1..%d fields per call (got %d)", PS_MAX_FIELDS, n_fields);
  PsFields F;
  for (int f = 0; f < n_fields; ++f) {
    NG_REQUIRE(fields_dev[f], NALU_GPU_EINVAL, "parallel_sum: field %d is null", f);
    F.f[f] = fields_dev[f];
  }
  const int nN = h->nNbr;
  if (nN == 0) return NALU_GPU_OK;  // one rank: nothing is shared
  NG_REQUIRE(h->hasComm && h->comm.exchange_rows, NALU_GPU_ESTATE,
             "parallel_sum needs a comm with exchange_rows (nalu_gpu_set_comm)");
  // plan == null: nodal fields through the halo plan's own lists; ghost copies send zeros and
  // receive the totals like the shared ones
  const std::vector<int> &so = P ? P->so : h->sendOff, &ro = P ? P->ro : h->recvOff;
  NG_REQUIRE((int)so.size() == nN + 1 && (int)ro.size() == nN + 1, NALU_GPU_ESTATE,
             "parallel_sum: the halo plan changed since the share plan was made");
  const int *sendIds = P ? P->dSendIds : h->dSendLids;
  const int *recvIds = P ? P->dRecvIds : h->dRecvLids;  // (null: columns nOwned + i)
  const int zeroFrom = P ? 0x7FFFFFFF : h->nOwned + h->nShared;
  const long long ns = so[nN], nr = ro[nN];
  // staging: the scatter-plan style "grow on demand" pair, kept on the handle
  const size_t need = sizeof(double) * (size_t)std::max<long long>(std::max(ns, nr), 1) * n_fields;
  if (need > h->psBytes) {
    NG_HIP(hipStreamSynchronize(h->stream));
    NG_FREE(h->dPsA); NG_FREE(h->dPsB);
    NG_HIP(hipMalloc(&h->dPsA, need));
    NG_HIP(hipMalloc(&h->dPsB, need));
    h->psBytes = need;
  }
  auto grid = [](long long n) { return (unsigned)((n + 255) / 256); };
  std::vector<int64_t> so64((size_t)nN + 1), ro64((size_t)nN + 1);
  for (int p = 0; p <= nN; ++p) { so64[p] = (int64_t)so[p] * n_fields; ro64[p] = (int64_t)ro[p] * n_fields; }
  // (1) the sharers' values to the owners
  if (nr) k_ps_pack<<<grid(nr), 256, 0, h->stream>>>(nr, recvIds, h->nOwned, zeroFrom, F, n_fields, h->dPsA);
  NG_HIP(hipGetLastError());
  NG_REQUIRE(h->comm.exchange_rows(h->comm.ctx, h->dPsA, ro64.data(), h->dPsB, so64.data(), h->stream) == 0,
             NALU_GPU_EHIP, "exchange_rows callback failed: %s", callback_error());
  // (2) added on the owner, neighbours in ascending rank order
  std::vector<int> order((size_t)nN);
  for (int p = 0; p < nN; ++p) order[p] = p;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h->nbrRanks[a] < h->nbrRanks[b]; });
  for (int q = 0; q < nN; ++q) {
    const int p = order[q];
    const long long n = so[p + 1] - so[p];
    if (n) k_ps_add<<<grid(n), 256, 0, h->stream>>>(n, sendIds + so[p], F, n_fields, h->dPsB + so64[p]);
  }
  // (3) the totals back to the sharers
  if (ns) k_ps_pack<<<grid(ns), 256, 0, h->stream>>>(ns, sendIds, 0, 0x7FFFFFFF, F, n_fields, h->dPsB);
  NG_HIP(hipGetLastError());
  NG_REQUIRE(h->comm.exchange_rows(h->comm.ctx, h->dPsB, so64.data(), h->dPsA, ro64.data(), h->stream) == 0,
             NALU_GPU_EHIP, "exchange_rows callback failed: %s", callback_error());
  if (nr) k_ps_unpack<<<grid(nr), 256, 0, h->stream>>>(nr, recvIds, h->nOwned, F, n_fields, h->dPsA);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}
