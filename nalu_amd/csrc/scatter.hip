// scatter.hip -- bulk sumInto on the device (gfx950).
//
// The reference's second sumInto overload takes Kokkos views and is meant to be called from
// inside a team loop over entities (src/TpetraLinearSystem.C:1387-1452); on the GPU the
// loop over entities is the launch, so the counterpart is ONE call that scatters the dense
// n_obj x n_obj blocks of a whole batch of entities (edges, elements, faces, nodes).  The
// entity -> matrix-slot search of sum_into_row (:1350-1385) is done once per connectivity
// ("plan"), the per-call work is the adds alone.  Three ways to resolve the row conflicts:
//   ROWS    one thread per matrix row walks the (entity, local row) pairs that touch it in
//           ascending entity order: single writer per value, no atomics, and the same order
//           of additions as the reference's sequential entity loop -> bit-identical.
//   COLORED greedy entity colouring (no two entities of a colour share a node), one launch
//           per colour, plain adds: deterministic, order of additions = colour order.
//   ATOMIC  one thread per (entity, local row), fp64 atomics: not bit-reproducible.
#include <algorithm>
#include <atomic>
#include <thread>

#include "internal.hpp"

using namespace nalu_gpu;

namespace {

template <typename F>
void parallel_for(int64_t n, F f)
{
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::max(1u, std::min(nt, 16u));
  if (n < 100000) nt = 1;
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

// ROWS: thread r of workgroup s owns row s * C + r.  Every add goes to memory in entity order
// (the values of a row may be hit several times), exactly as the sequential loop would.
__global__ __launch_bounds__(SELL_C) void k_batch_rows(SellView A, int nObj,
                                                        const long long *__restrict__ incPtr,
                                                        const int *__restrict__ inc,
                                                        const short *__restrict__ kIdx,
                                                        const double *__restrict__ lhs,
                                                        const double *__restrict__ rhsIn,
                                                        double *__restrict__ rhs)
{
  const int slice = blockIdx.x, r = threadIdx.x;
  const int row = slice * SELL_C + r;
  if (row >= A.nRows) return;
  const long long off = A.sliceOff[slice];
  double b = rhs[row];
  for (long long t = incPtr[row]; t < incPtr[row + 1]; ++t) {
    const long long ei = inc[t];
    const double *l = lhs + ei * nObj;
    const short *k = kIdx + ei * nObj;
    for (int j = 0; j < nObj; ++j)
      if (k[j] >= 0) A.vals[off + (long long)k[j] * SELL_C + r] += l[j];
    b += rhsIn[ei];
  }
  rhs[row] = b;
}

// COLORED / ATOMIC: thread t handles local row t % nObj of entity perm[first + t / nObj]
template <bool ATOMIC>
__global__ __launch_bounds__(256) void k_batch_entities(long long n, long long first,
                                                         const int *__restrict__ perm, int nObj,
                                                         const int *__restrict__ rowOf,
                                                         const short *__restrict__ kIdx,
                                                         const long long *__restrict__ sliceOff,
                                                         const double *__restrict__ lhs,
                                                         const double *__restrict__ rhsIn,
                                                         double *vals, double *rhs)
{
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const long long e = perm ? perm[first + t / nObj] : first + t / nObj;
  const long long ei = e * nObj + t % nObj;
  const int row = rowOf[ei];
  if (row < 0) return;
  const long long base = sliceOff[row / SELL_C] + row % SELL_C;
  const double *l = lhs + ei * nObj;
  const short *k = kIdx + ei * nObj;
  for (int j = 0; j < nObj; ++j) {
    if (k[j] < 0) continue;
    double *p = vals + base + (long long)k[j] * SELL_C;
    if (ATOMIC) unsafeAtomicAdd(p, l[j]);
    else *p += l[j];
  }
  if (ATOMIC) unsafeAtomicAdd(rhs + row, rhsIn[ei]);
  else rhs[row] += rhsIn[ei];
}

int ensure_plan_colors(nalu_gpu_scatter_plan *p)
{
  if (p->dColorPerm || p->nEnt == 0) return NALU_GPU_OK;
  const nalu_gpu_linsys *h = p->h;
  std::vector<uint64_t> used((size_t)h->nCols, 0);
  std::vector<unsigned char> color((size_t)p->nEnt);
  int nColors = 0;
  for (int64_t e = 0; e < p->nEnt; ++e) {
    const int *en = &p->lids[e * p->nObj];
    uint64_t m = 0;
    for (int i = 0; i < p->nObj; ++i) m |= used[en[i]];
    NG_REQUIRE(m != ~0ull, NALU_GPU_EUNSUPPORTED, "more than 64 entity colours needed");
    const int c = __builtin_ctzll(~m);
    color[e] = (unsigned char)c;
    for (int i = 0; i < p->nObj; ++i) used[en[i]] |= 1ull << c;
    nColors = std::max(nColors, c + 1);
  }
  p->nColors = nColors;
  p->colorOff.assign((size_t)nColors + 1, 0);
  for (int64_t e = 0; e < p->nEnt; ++e) p->colorOff[color[e] + 1]++;
  for (int c = 0; c < nColors; ++c) p->colorOff[c + 1] += p->colorOff[c];
  std::vector<int> perm((size_t)p->nEnt);
  std::vector<int64_t> fill(p->colorOff.begin(), p->colorOff.end() - 1);
  for (int64_t e = 0; e < p->nEnt; ++e) perm[fill[color[e]]++] = (int)e;
  NG_HIP(hipMalloc(&p->dColorPerm, sizeof(int) * (size_t)p->nEnt));
  NG_HIP(hipMemcpy(p->dColorPerm, perm.data(), sizeof(int) * (size_t)p->nEnt, hipMemcpyHostToDevice));
  return NALU_GPU_OK;
}

}  // namespace

extern "C" int nalu_gpu_scatter_plan_create(nalu_gpu_linsys *h, int64_t n_ent, int n_obj,
                                            const int *ent_lids, nalu_gpu_scatter_plan **out)
{
  NG_REQUIRE(h && out, NALU_GPU_EINVAL, "null handle / out");
  NG_REQUIRE(h->state == 2, NALU_GPU_ESTATE, "linear system is not finalized");
  NG_REQUIRE(n_ent >= 0 && n_obj >= 1 && n_obj <= 64 && (ent_lids || n_ent == 0), NALU_GPU_EINVAL,
             "scatter plan: 1..64 nodes per entity (got %d)", n_obj);
  NG_REQUIRE(n_ent * n_obj < 2147483647LL, NALU_GPU_EUNSUPPORTED, "scatter plan: too many entities");
  NG_DEVICE(h->device);
  const int64_t tot = n_ent * n_obj;
  for (int64_t t = 0; t < tot; ++t)
    NG_REQUIRE(ent_lids[t] >= 0 && ent_lids[t] < h->nCols, NALU_GPU_EINVAL,
               "entity %lld references local node %d outside [0,%d)", (long long)(t / n_obj),
               ent_lids[t], h->nCols);
  for (int64_t e = 0; e < n_ent; ++e)
    for (int i = 1; i < n_obj; ++i)
      for (int j = 0; j < i; ++j)
        NG_REQUIRE(ent_lids[e * n_obj + i] != ent_lids[e * n_obj + j], NALU_GPU_EINVAL,
                   "entity %lld lists node %d twice", (long long)e, ent_lids[e * n_obj + i]);
  nalu_gpu_scatter_plan *p = new nalu_gpu_scatter_plan();
  p->h = h;
  p->nEnt = n_ent;
  p->nObj = n_obj;
  p->lids.assign(ent_lids, ent_lids + tot);
  const int nO = h->nRowsAll;  // rows stored (export mode: the shared-not-owned copies too)
  // rows beyond the stored ones are assembled by their owner (see nalu_gpu_sum_into)
  std::vector<int> rowOf((size_t)std::max<int64_t>(tot, 1));
  std::vector<short> kIdx((size_t)std::max<int64_t>(tot * n_obj, 1));
  std::atomic<bool> tooWide(false);
  parallel_for(n_ent, [&](int64_t e0, int64_t e1) {
    for (int64_t e = e0; e < e1; ++e) {
      const int *en = &p->lids[e * n_obj];
      for (int i = 0; i < n_obj; ++i) {
        const int row = en[i];
        short *k = &kIdx[(e * n_obj + i) * n_obj];
        if (row >= nO) {
          rowOf[e * n_obj + i] = -1;
          for (int j = 0; j < n_obj; ++j) k[j] = -1;
          continue;
        }
        rowOf[e * n_obj + i] = row;
        const int *b = &h->colIdx[h->rowPtr[row]], *end = &h->colIdx[h->rowPtr[row + 1]];
        if (end - b > 32767) tooWide = true;
        // sum_into_row (:1363-1383) scans the row once with the columns ascending: a column
        // that is missing runs the scan to the row's end, so it and every larger column of the
        // entity are dropped
        int order[64];
        for (int j = 0; j < n_obj; ++j) {
          order[j] = j;
          k[j] = -1;
        }
        std::sort(order, order + n_obj, [&](int x, int y) { return en[x] < en[y]; });
        for (int jj = 0; jj < n_obj; ++jj) {
          const int j = order[jj];
          const int *it = std::lower_bound(b, end, en[j]);
          if (it == end || *it != en[j]) break;
          k[j] = (short)(it - b);
        }
      }
    }
  });
  if (tooWide) {
    delete p;
    return set_error(NALU_GPU_EUNSUPPORTED, "scatter plan: matrix rows longer than 32767 entries");
  }
  // (entity, i) pairs per owned row, ascending in e * n_obj + i
  std::vector<long long> incPtr((size_t)nO + 1, 0);
  for (int64_t t = 0; t < tot; ++t)
    if (rowOf[t] >= 0) incPtr[rowOf[t] + 1]++;
  for (int r = 0; r < nO; ++r) incPtr[r + 1] += incPtr[r];
  std::vector<int> inc((size_t)std::max<long long>(incPtr[nO], 1));
  {
    std::vector<long long> fill(incPtr.begin(), incPtr.end() - 1);
    for (int64_t t = 0; t < tot; ++t)
      if (rowOf[t] >= 0) inc[fill[rowOf[t]]++] = (int)t;
  }
#define NG_PLAN_HIP(call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      release_plan(p);                                                                         \
      delete p;                                                                                \
      return set_error(NALU_GPU_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));          \
    }                                                                                          \
  } while (0)
  NG_PLAN_HIP(hipMalloc(&p->dRow, sizeof(int) * rowOf.size()));
  NG_PLAN_HIP(hipMalloc(&p->dK, sizeof(short) * kIdx.size()));
  NG_PLAN_HIP(hipMalloc(&p->dIncPtr, sizeof(long long) * incPtr.size()));
  NG_PLAN_HIP(hipMalloc(&p->dInc, sizeof(int) * inc.size()));
  NG_PLAN_HIP(hipMemcpy(p->dRow, rowOf.data(), sizeof(int) * rowOf.size(), hipMemcpyHostToDevice));
  NG_PLAN_HIP(hipMemcpy(p->dK, kIdx.data(), sizeof(short) * kIdx.size(), hipMemcpyHostToDevice));
  NG_PLAN_HIP(hipMemcpy(p->dIncPtr, incPtr.data(), sizeof(long long) * incPtr.size(), hipMemcpyHostToDevice));
  NG_PLAN_HIP(hipMemcpy(p->dInc, inc.data(), sizeof(int) * inc.size(), hipMemcpyHostToDevice));
#undef NG_PLAN_HIP
  h->plans.push_back(p);
  *out = p;
  return NALU_GPU_OK;
}

// device memory of a plan; also called by nalu_gpu_destroy for plans that outlive their system
void nalu_gpu::release_plan(nalu_gpu_scatter_plan *p)
{
  if (p->dRow) (void)hipFree(p->dRow);
  if (p->dK) (void)hipFree(p->dK);
  if (p->dIncPtr) (void)hipFree(p->dIncPtr);
  if (p->dInc) (void)hipFree(p->dInc);
  if (p->dColorPerm) (void)hipFree(p->dColorPerm);
  p->dRow = nullptr; p->dK = nullptr; p->dIncPtr = nullptr; p->dInc = nullptr; p->dColorPerm = nullptr;
  p->h = nullptr;
}

extern "C" int nalu_gpu_scatter_plan_destroy(nalu_gpu_scatter_plan *p)
{
  if (!p) return NALU_GPU_OK;
  if (nalu_gpu_linsys *h = p->h) {  // null: the system was destroyed first and released the plan
    nalu_gpu::DeviceGuard ngGuard_(h->device);
    (void)hipStreamSynchronize(h->stream);
    h->plans.erase(std::remove(h->plans.begin(), h->plans.end(), p), h->plans.end());
    release_plan(p);
  }
  delete p;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_scatter_plan_colors(nalu_gpu_scatter_plan *p, int *n_colors)
{
  NG_REQUIRE(p && n_colors, NALU_GPU_EINVAL, "null plan / out");
  NG_REQUIRE(p->h, NALU_GPU_ESTATE, "the plan's linear system was destroyed");
  NG_DEVICE(p->h->device);
  NG_TRY(ensure_plan_colors(p));
  *n_colors = p->nColors;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_sum_into_batch(nalu_gpu_linsys *h, nalu_gpu_scatter_plan *p,
                                       const double *rhs_dev, const double *lhs_dev,
                                       int scatter_mode)
{
  NG_REQUIRE(h && p, NALU_GPU_EINVAL, "null handle / plan");
  NG_REQUIRE(p->h == h && h->state == 2, NALU_GPU_ESTATE,
             "the plan belongs to another (or a not finalized) linear system");
  NG_REQUIRE(p->nEnt == 0 || (rhs_dev && lhs_dev), NALU_GPU_EINVAL, "null rhs / lhs");
  NG_DEVICE(h->device);
  NG_TRY(materialize_zero(h));
  if (p->nEnt == 0 || h->nOwned == 0) return NALU_GPU_OK;
  const long long nThreads = (long long)p->nEnt * p->nObj;
  if (scatter_mode == NALU_GPU_SCATTER_ROWS) {
    timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
    k_batch_rows<<<h->nSlices, SELL_C, 0, h->stream>>>(sell_view_all(h), p->nObj, p->dIncPtr, p->dInc,
                                                       p->dK, lhs_dev, rhs_dev, h->dRhs);
    timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
  } else if (scatter_mode == NALU_GPU_SCATTER_ATOMIC) {
    timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
    k_batch_entities<true><<<(unsigned)((nThreads + 255) / 256), 256, 0, h->stream>>>(
        nThreads, 0, nullptr, p->nObj, p->dRow, p->dK, h->dSliceOff, lhs_dev, rhs_dev, h->dVals,
        h->dRhs);
    timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
  } else {
    NG_REQUIRE(scatter_mode == NALU_GPU_SCATTER_COLORED, NALU_GPU_EINVAL,
               "unknown scatter mode %d", scatter_mode);
    NG_TRY(ensure_plan_colors(p));
    timer_begin(h, NALU_GPU_T_ASSEMBLE_EDGES);
    for (int c = 0; c < p->nColors; ++c) {
      const long long n = (p->colorOff[c + 1] - p->colorOff[c]) * p->nObj;
      if (n == 0) continue;
      k_batch_entities<false><<<(unsigned)((n + 255) / 256), 256, 0, h->stream>>>(
          n, p->colorOff[c], p->dColorPerm, p->nObj, p->dRow, p->dK, h->dSliceOff, lhs_dev,
          rhs_dev, h->dVals, h->dRhs);
    }
    timer_end(h, NALU_GPU_T_ASSEMBLE_EDGES);
  }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}
