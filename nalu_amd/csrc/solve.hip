// solve.hip -- SpMV, vector kernels, deterministic reductions and the device-resident
// Jacobi-preconditioned CG / restarted right-preconditioned GMRES(m) (ICGS) of
// libnalu_gpu.so (gfx950).  All kernels are HBM-bound fp64 streaming work: no MFMA.
//
// The Krylov loops never bring a scalar to the host: dot products end in a fixed-order
// reduction of per-workgroup partials into the handle's reduce buffer, the scalar
// recurrences (alpha, beta, Givens rotations, convergence test, residual history) run in
// one-thread kernels on the device, and every kernel checks the device "done" flag, so the
// host only polls that flag every few iterations and the iteration count is exact.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>

#include "internal.hpp"

using namespace nalu_gpu;

namespace {

// ------------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// fixed-order block sums (blockDim = 256): lane tree inside each wave, then the four
// wave sums are added in order.
template <int NS>
__device__ __forceinline__ void block_sums(double (&v)[NS], double *partials, int nvalid,
                                           int block = -1)
{
  if (block < 0) block = blockIdx.x;
  __shared__ double s[NS][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double t = wave_sum(v[i]);
    if (lane == 0) s[i][wave] = t;
  }
  __syncthreads();
  if (threadIdx.x < NS && (int)threadIdx.x < nvalid) {
    const int i = threadIdx.x;
    partials[(size_t)i * MAX_BLOCKS + block] = ((s[i][0] + s[i][1]) + s[i][2]) + s[i][3];
  }
}

// CG scalar recurrences (one thread).  They run either as the epilogue of the partial-sum
// reduction (single rank) or as one-thread kernels behind the all-reduce (multi-rank).
__device__ __forceinline__ void cg_scal_a(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  sc->alpha = sc->rz / red[0];
  if (sc->alphaHist && sc->iter < sc->histCap) sc->alphaHist[sc->iter] = sc->alpha;
}

__device__ __forceinline__ void cg_scal_b(SolverScalars *sc, const double *red, double *hist)
{
  if (sc->done) return;
  sc->iter += 1;
  const double rel = sqrt(red[0]) / sc->r0norm;
  if (sc->iter < sc->histCap) hist[sc->iter] = rel;
  if (rel <= sc->tol || sc->iter >= sc->maxIter) {
    sc->done = 1;
    return;
  }
  sc->beta = red[1] / sc->rz;
  sc->rz = red[1];
}

enum { EPI_NONE = 0, EPI_CG_A = 1, EPI_CG_B = 2 };

// fixed-order sum of the per-workgroup partials of every slot (one workgroup per NS slots, or one
// workgroup for all slots followed by the scalar recurrence that consumes the sums).  The NS
// slots of a workgroup are summed together -- their loads in flight at once, one barrier -- by
// shuffle trees inside the 16 waves and one more over the 16 wave sums: two dependent steps
// instead of a ten-level LDS tree per slot (this kernel is pure latency: it sits between the
// kernels of every Krylov iteration).
constexpr int RED_T = 1024;
template <int EPI, int NS>
__global__ __launch_bounds__(RED_T) void k_reduce_partials(const double *__restrict__ partials,
                                                          int nblocks, int nslots,
                                                          double *__restrict__ out,
                                                          SolverScalars *sc, double *hist)
{
  __shared__ double s[NS][RED_T / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int slot0 = blockIdx.x * NS; slot0 < nslots; slot0 += gridDim.x * NS) {  // epilogues need a grid of 1
    double v[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) v[q] = 0.0;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += RED_T * 8) {  // 8 NS loads in flight, fixed add order
      double t[NS][8];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double *p = partials + (size_t)min(slot0 + q, nslots - 1) * MAX_BLOCKS;
#pragma unroll
        for (int u = 0; u < 8; ++u) t[q][u] = (i0 + u * RED_T < nblocks) ? p[i0 + u * RED_T] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[q] += t[q][u];
    }
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const double t = wave_sum(v[q]);
      if (lane == 0) s[q][wave] = t;
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        double t = lane < RED_T / 64 ? s[q][lane] : 0.0;
#pragma unroll
        for (int o = RED_T / 128; o > 0; o >>= 1) t += __shfl_down(t, o, 64);
        if (lane == 0 && slot0 + q < nslots) out[slot0 + q] = t;
      }
    }
    __syncthreads();
  }
  if (EPI != EPI_NONE && threadIdx.x == 0) {  // thread 0 wrote out[] itself
    if (EPI == EPI_CG_A) cg_scal_a(sc, out);
    if (EPI == EPI_CG_B) cg_scal_b(sc, out, hist);
  }
}

// first stage for long partial lists (the SpMV-fused dot leaves one partial per slice, 65,536
// at 256^3, and a single workgroup reads them at single-CU bandwidth): PRE_WG workgroups sum
// contiguous chunks in a fixed order
constexpr int PRE_WG = 64;
__global__ __launch_bounds__(256) void k_prereduce(const double *__restrict__ src, int n,
                                                    double *__restrict__ dst,
                                                    const SolverScalars *__restrict__ sc)
{
  if (sc && sc->done) return;
  __shared__ double s[4];
  const int chunk = (n + PRE_WG - 1) / PRE_WG;
  const int i0 = blockIdx.x * chunk, i1 = min(n, i0 + chunk);
  double v = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += 1024) {  // 4 loads in flight
    double t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = (i + u * 256 < i1) ? src[i + u * 256] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) v += t[u];
  }
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) dst[blockIdx.x] = ((s[0] + s[1]) + s[2]) + s[3];
}

// slice partials -> at most PRE_WG partials (in place behind the slice partials)
int prereduce_slices(nalu_gpu_linsys *h, int *count, const double **src, bool useDone)
{
  *src = h->dSlicePartials;
  if (*count <= 16384) return NALU_GPU_OK;  // one 1024-thread workgroup sums these in ~1 us
  double *dst = h->dSlicePartials + h->nSlices;
  k_prereduce<<<PRE_WG, 256, 0, h->stream>>>(h->dSlicePartials, *count, dst, useDone ? h->dScal : nullptr);
  NG_HIP(hipGetLastError());
  *src = dst;
  *count = PRE_WG;
  return NALU_GPU_OK;
}

int reduce_partials(nalu_gpu_linsys *h, int nslots, int nblocks, int outOffset,
                    const double *src = nullptr)
{
  k_reduce_partials<EPI_NONE, 1><<<nslots, RED_T, 0, h->stream>>>(src ? src : h->dPartials, nblocks, nslots,
                                                        h->dReduce + outOffset, nullptr, nullptr);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

int allreduce(nalu_gpu_linsys *h, int count)
{
  if (!h->hasComm) return NALU_GPU_OK;
  NG_REQUIRE(h->comm.allreduce_sum(h->comm.ctx, count, h->stream) == 0, NALU_GPU_EHIP,
             "allreduce_sum callback failed: %s", callback_error());
  return NALU_GPU_OK;
}

}  // namespace

int nalu_gpu::vec_grid(int64_t n)
{
  static const int per = getenv("NG_VEC_PER") ? std::max(1, atoi(getenv("NG_VEC_PER"))) : 4;
  const int64_t nb = (n + 256 * per - 1) / (256 * per);  // elements per thread (at least)
  return (int)std::max<int64_t>(1, std::min<int64_t>(nb, MAX_BLOCKS));
}

int nalu_gpu::finish_reduction(nalu_gpu_linsys *h, int nslots, int nblocks, const double *src)
{
  NG_TRY(reduce_partials(h, nslots, nblocks, 0, src));
  return allreduce(h, nslots);
}

namespace {

// ------------------------------------------------------------------------- SpMV
// y = A x on SELL-256: thread per row, lanes contiguous for every k.  Each workgroup owns a
// contiguous chunk of slices (XCD-aware), optionally fusing the p.Ap partial sum (CG) or
// the right-preconditioner column scaling x_c -> dinv_c x_c (GMRES).
// W entries of one row starting at entry k0: all column and value loads are issued before
// the first x gather is consumed (the slice width is only known at run time, so without
// this the k loop is a chain of dependent round trips), then the products are added in
// entry order -- the order of the oracle's CSR loop.
// W entries of one row starting at entry k0: all column and value loads are issued before
// the first x gather is consumed, then the products are added in entry order -- the order of
// the oracle's CSR loop.
template <int W, bool DINV>
__device__ __forceinline__ double row_chunk(const int *__restrict__ cols,
                                            const double *__restrict__ vals, long long pos,
                                            const int *__restrict__ off, bool ex, int row,
                                            const double *__restrict__ x,
                                            const double *__restrict__ dinv, double sum)
{
  int c[W];
  double v[W], xv[W];
#pragma unroll
  for (int j = 0; j < W; ++j) c[j] = ex ? cols[pos + (long long)j * SELL_C] : row + off[j];
#pragma unroll
  // streamed once: nontemporal, so the matrix does not displace x in L2 (-2.5 % measured)
  for (int j = 0; j < W; ++j) v[j] = __builtin_nontemporal_load(vals + pos + (long long)j * SELL_C);
#pragma unroll
  for (int j = 0; j < W; ++j) xv[j] = DINV ? dinv[c[j]] * x[c[j]] : x[c[j]];
#pragma unroll
  for (int j = 0; j < W; ++j) sum += v[j] * xv[j];
  return sum;
}

template <bool DINV>
__device__ __forceinline__ double row_sum(const SellView &A, long long pos, int w,
                                          const int *__restrict__ off, bool ex, int row,
                                          const double *__restrict__ x,
                                          const double *__restrict__ dinv)
{
  double sum = 0.0;
  int k = 0;
  for (; k + 8 <= w; k += 8, pos += 8LL * SELL_C, off += 8)
    sum = row_chunk<8, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum);
  switch (w - k) {  // wave-uniform
    case 7: sum = row_chunk<7, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 6: sum = row_chunk<6, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 5: sum = row_chunk<5, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 4: sum = row_chunk<4, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 3: sum = row_chunk<3, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 2: sum = row_chunk<2, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    case 1: sum = row_chunk<1, DINV>(A.cols, A.vals, pos, off, ex, row, x, dinv, sum); break;
    default: break;
  }
  return sum;
}

// The long-row path: a slice wider than LONG_ROW_W has a skewed row (SELL-256 pads a slice to its longest row,
// and the plain loop would walk that padding for all 256 rows).  Every row's first LONG_ROW_W entries are summed
// by its own thread in entry order, up to the row's length; the rest of a longer row is summed by the whole
// workgroup, thread t taking entries LONG_ROW_W + t, + 256, ..., and added in a fixed tree (wave shuffle tree,
// the four wave sums in order).  Deterministic; for the rows longer than LONG_ROW_W the summation order is
// not the CSR loop's (<= 1e-12 against the oracle instead of bit for bit).  North star's "general fallback" for
// skewed rows: time is bounded by the entries that exist; the storage of such a slice stays padded.
template <bool DINV>
__device__ __forceinline__ double long_slice_sum(const SellView &A, long long so, int row,
                                                 const double *__restrict__ x, const double *__restrict__ dinv)
{
  __shared__ int sLen[SELL_C];
  __shared__ double sPart[SELL_C / 64];
  const int t = threadIdx.x;
  const int len = row < A.nRows ? A.rowLen[row] : 0;
  sLen[t] = len;
  double sum = 0.0;
  const int n1 = min(len, LONG_ROW_W);
  for (int k = 0; k < n1; ++k) {
    const long long pos = so + (long long)k * SELL_C + t;
    const int c = A.cols[pos];
    sum += A.vals[pos] * (DINV ? dinv[c] * x[c] : x[c]);
  }
  __syncthreads();
  for (int r = 0; r < SELL_C; ++r) {  // (workgroup-uniform: the lengths are in LDS)
    const int lr = sLen[r];
    if (lr <= LONG_ROW_W) continue;
    double part = 0.0;
    for (int k = LONG_ROW_W + t; k < lr; k += SELL_C) {
      const long long pos = so + (long long)k * SELL_C + r;
      const int c = A.cols[pos];
      part += A.vals[pos] * (DINV ? dinv[c] * x[c] : x[c]);
    }
    part = wave_sum(part);
    if ((t & 63) == 0) sPart[t >> 6] = part;
    __syncthreads();
    if (t == r) sum += ((sPart[0] + sPart[1]) + sPart[2]) + sPart[3];
    __syncthreads();
  }
  return sum;
}

// One workgroup per slice, dispatched so that the eight XCDs sweep the matrix as ONE
// streaming front (xcd_run): measured on MI355X at 256^3 this reads 5.7 TB/s against
// 4.4 TB/s for per-XCD contiguous chunks marched by a capped grid (profiles/).  With DOT the
// workgroup leaves its x.Ax partial in partials[slice] (summed in slice order afterwards).
template <bool DOT, bool DINV>
__global__ __launch_bounds__(SELL_C) void k_spmv(SellView A, const double *__restrict__ x,
                                                  const double *__restrict__ dinv,
                                                  double *__restrict__ y,
                                                  double *__restrict__ partials,
                                                  const SolverScalars *__restrict__ sc, int run,
                                                  const int *__restrict__ list, int nList)
{
  // the done flag is only needed before the stores: let its load overlap the matrix loads
  // (a workgroup lives a few microseconds; a dependent L2 round trip up front is not free)
  const int stop = sc ? sc->done : 0;
  __shared__ double sWave[SELL_C / 64];
  __shared__ int sTicket;
  if (DOT) {  // all waves are here at launch: this barrier costs nothing
    if (threadIdx.x == 0) sTicket = 0;
    __syncthreads();
  }
  const int s = list ? list[xcd_run(blockIdx.x, nList, run)] : xcd_run(blockIdx.x, A.nSlices, run);
  const int row = s * SELL_C + threadIdx.x;
  const long long so = A.sliceOff[s];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long mask = A.exMask[(size_t)s * 4 + wave];  // wave-uniform
  const bool ex = (mask >> (threadIdx.x & 63)) & 1ULL;
  double sum;
  // A wave whose only row with irregular columns is described by the assembly's per-slice descriptor
  // (BurstSlice::irr: on a lattice the first / last node of an x-line) takes that row's columns from
  // the descriptor -- a scalar load issued with the slice offset -- instead of reading them from the
  // column array, which is a dependent round trip that the whole wave waits for (two of the four
  // waves of every slice at 256^3: 0.270 -> 0.2xx ms).
  const BurstRow *X = (A.burst && mask != 0ULL) ? &A.burst[s].irr[wave] : nullptr;
  // (not for a row with more than BURST_NE edges or with other graph entries: bits 16-24 of info)
  if (A.rowLen && A.sliceWidth[s] > LONG_ROW_W) {  // (workgroup-uniform) a slice with a skewed row
    sum = long_slice_sum<DINV>(A, so, row, x, dinv);
  } else if (X && X->lane < 64u && (X->info >> 16) == 0u && mask == (1ULL << X->lane)) {
    constexpr int WMAX = BURST_W;
    const int w = A.sliceWidth[s];
    const int *off = A.kOff + so / SELL_C;
    const long long pos = so + threadIdx.x;
    const unsigned kl = X->kl, nx = X->info & 15u;
    int c[WMAX];
    double v[WMAX], xv[WMAX];
#pragma unroll
    for (int j = 0; j < WMAX; ++j) c[j] = j < w ? row + off[j] : row;
    if (ex) {  // (one lane) the diagonal and the padding have the row's own column, the edges are listed
#pragma unroll
      for (int j = 0; j < WMAX; ++j) c[j] = row;
#pragma unroll
      for (int e = 0; e < BURST_NE; ++e) {
        const unsigned k = (kl >> (4 * e)) & 15u;
#pragma unroll
        for (int j = 0; j < WMAX; ++j)
          if ((unsigned)e < nx && k == (unsigned)j) c[j] = X->ko[e];
      }
    }
#pragma unroll
    for (int j = 0; j < WMAX; ++j)
      if (j < w) v[j] = __builtin_nontemporal_load(A.vals + pos + (long long)j * SELL_C);
#pragma unroll
    for (int j = 0; j < WMAX; ++j)
      if (j < w) xv[j] = DINV ? dinv[c[j]] * x[c[j]] : x[c[j]];
    sum = 0.0;
#pragma unroll
    for (int j = 0; j < WMAX; ++j)
      if (j < w) sum += v[j] * xv[j];
  } else {
    sum = row_sum<DINV>(A, so + threadIdx.x, A.sliceWidth[s], A.kOff + so / SELL_C, ex, row, x, dinv);
  }
  if (stop) return;
  double prod = 0.0;
  if (row < A.nRows) {
    __builtin_nontemporal_store(sum, y + row);  // -4 % measured against a plain store
    if (DOT) prod = x[row] * sum;
  }
  if (DOT) {
    // barrier-free epilogue: every wave leaves its sum in LDS and takes a ticket; the wave
    // that draws the last ticket adds the four sums in wave order (the others are gone by
    // then, they do not wait for the slowest wave of the workgroup)
    const double t = wave_sum(prod);
    if ((threadIdx.x & 63) == 0) {
      sWave[threadIdx.x >> 6] = t;
      __threadfence_block();
      if (atomicAdd(&sTicket, 1) == SELL_C / 64 - 1) {
        __threadfence_block();  // acquire: the other waves' sums are in LDS
        const volatile double *w = sWave;
        partials[s] = ((w[0] + w[1]) + w[2]) + w[3];
      }
    }
  }
}

// DOT: the partials land in h->dSlicePartials[0 .. *nblocks).  list != nullptr: only the
// nList slices of the list
template <bool DOT, bool DINV>
int launch_spmv(nalu_gpu_linsys *h, const double *x, double *y, bool useDone, int *nblocks,
                const int *list = nullptr, int nList = 0)
{
  if (nblocks) *nblocks = h->nSlicesOwned;
  const int nb = list ? nList : h->nSlicesOwned;
  if (nb == 0) return NALU_GPU_OK;
  timer_begin(h, NALU_GPU_T_SPMV);
  k_spmv<DOT, DINV><<<nb, SELL_C, 0, h->stream>>>(sell_view(h), x, h->dDinv, y, h->dSlicePartials,
                                                   useDone ? h->dScal : nullptr, h->xcdRun, list,
                                                   nList);
  timer_end(h, NALU_GPU_T_SPMV);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// y = A x with the halo of x (Tpetra Import inside CrsMatrix::apply): pack + post the
// exchange, multiply the slices that read owned columns only while it is in flight, then
// finish the exchange and multiply the rest.
template <bool DOT, bool DINV>
int spmv_with_halo(nalu_gpu_linsys *h, double *x, double *y, bool useDone, int *nblocks)
{
  if (h->nCols == h->nOwned) return launch_spmv<DOT, DINV>(h, x, y, useDone, nblocks);
  NG_TRY(halo_begin(h, x));
  NG_TRY((launch_spmv<DOT, DINV>(h, x, y, useDone, nblocks, h->dSliceLists, h->nInterior)));
  NG_TRY(halo_end(h, x));
  return launch_spmv<DOT, DINV>(h, x, y, useDone, nblocks, h->dSliceLists + h->nInterior,
                                h->nBoundary);
}

// ------------------------------------------------------------------ vector kernels
__device__ __forceinline__ double ldnt(const double *p) { return __builtin_nontemporal_load(p); }

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

__global__ __launch_bounds__(256) void k_dot(long long n, const double *__restrict__ a,
                                              const double *__restrict__ b,
                                              double *__restrict__ partials)
{
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) acc[0] += a[i] * b[i];
  block_sums<1>(acc, partials, 1);
}

__global__ __launch_bounds__(256) void k_axpy(long long n, double alpha, const double *__restrict__ x,
                                               double *__restrict__ y)
{
  GRID_STRIDE(i, n) y[i] += alpha * x[i];
}

// dinv = 1/diag(A) (Ifpack2 Jacobi compute(): diagonal extraction), or 1 without precond.
// diagPos holds the SELL position of every row's diagonal entry (-1: none), found once at
// finalize, so this reads one value per row instead of scanning the row's tags.
// bad[0] counts the rows whose diagonal is missing, zero or not finite (dinv would be inf / NaN
// and poison the whole solve without a word), bad[1] holds the smallest such row + 1.
__global__ __launch_bounds__(256) void k_extract_dinv(int nRows, const long long *__restrict__ diagPos,
                                                       const double *__restrict__ vals, int jacobi,
                                                       double *__restrict__ dinv, int *__restrict__ bad)
{
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= nRows) return;
  double d = 1.0;
  if (jacobi) {
    const long long pos = diagPos[row];
    const double a = pos >= 0 ? vals[pos] : 0.0;
    d = 1.0 / a;
    if (!(a != 0.0) || !(fabs(a) <= 1.79769313486231570e308)) {  // zero, NaN or inf
      atomicAdd(bad, 1);
      atomicMax(bad + 1, nRows - row);  // (max of n - row = the smallest row)
    }
  }
  dinv[row] = d;
}

// checkForNaN + checkForZeroRow (src/TpetraLinearSystem.C:1782-1899), one thread per owned row:
// out[0] rows with a NaN in the matrix, out[1] NaN entries of the rhs, out[2] rows with
// sum |a_ij| < 1e-10 (:1862); out[3..5] the smallest row of each kind, stored as n - row
__global__ __launch_bounds__(SELL_C) void k_check_rows(SellView A, const double *__restrict__ rhs, int *out)
{
  const int slice = blockIdx.x, r = threadIdx.x;
  const int row = slice * SELL_C + r;
  if (row >= A.nRows) return;
  const long long off = A.sliceOff[slice];
  const int w = A.sliceWidth[slice];
  bool nan = false;
  double sum = 0.0;
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C + r;
    if (A.tags[pos] == TAG_PAD) continue;
    const double v = A.vals[pos];
    nan |= v != v;
    sum += fabs(v);
  }
  if (nan) { atomicAdd(out, 1); atomicMax(out + 3, A.nRows - row); }
  if (rhs[row] != rhs[row]) { atomicAdd(out + 1, 1); atomicMax(out + 4, A.nRows - row); }
  if (!nan && sum < 1.e-10) { atomicAdd(out + 2, 1); atomicMax(out + 5, A.nRows - row); }
}

// ------------------------------------------------------------------- Chebyshev
// Polynomial preconditioner on D^-1 A: Ifpack2::Details::Chebyshev (ifpackApplyImpl and
// powerMethod; Trilinos, not in the reference tree: parity unpinned) with the parameters of
// the reference's MueLu smoother block reg_tests/xml/milestone.xml:5-12 -- degree 2,
// eigenvalue ratio 20, 15 power iterations, zero starting solution -- and Ifpack2's boost
// factor 1.1.  The reference has no Ifpack2-level Chebyshev option; this is the SpMV-only
// alternative to SGS that BASELINE's north star asks for.  The power method starts from a
// fixed hash of the global node ids (Ifpack2: a random vector) so runs are reproducible.
constexpr int CHEB_DEGREE = 2, CHEB_POWER_ITS = 15;
constexpr double CHEB_RATIO = 20.0, CHEB_BOOST = 1.1;

double cheb_start_value(int64_t gid)
{
  uint32_t v = (uint32_t)gid * 2654435761u;
  v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
  return (double)v / 4294967296.0 * 2.0 - 1.0;
}

// y = dinv .* y and the partial sums of y.x, x.x, y.y
__global__ __launch_bounds__(256) void k_cheb_power(long long n, const double *__restrict__ dinv,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     double *__restrict__ partials)
{
  double acc[3] = {0.0, 0.0, 0.0};
  GRID_STRIDE(i, n) {
    const double yi = dinv[i] * y[i], xi = x[i];
    y[i] = yi;
    acc[0] += yi * xi;
    acc[1] += xi * xi;
    acc[2] += yi * yi;
  }
  block_sums<3>(acc, partials, 3);
}

__global__ __launch_bounds__(256) void k_div_copy(long long n, const double *__restrict__ y, double d,
                                                   double *__restrict__ x)
{
  GRID_STRIDE(i, n) x[i] = y[i] / d;
}

// w = (dinv .* r) / theta ; x = w
__global__ __launch_bounds__(256) void k_cheb_first(long long n, const SolverScalars *__restrict__ sc,
                                                     double oneOverTheta, const double *__restrict__ dinv,
                                                     const double *__restrict__ r, double *__restrict__ w,
                                                     double *__restrict__ x)
{
  if (sc && sc->done) return;
  GRID_STRIDE(i, n) {
    const double wi = oneOverTheta * (dinv[i] * r[i]);
    w[i] = wi;
    x[i] = wi;
  }
}

// w = d1 w + d2 dinv .* (r - v) ; x += w
__global__ __launch_bounds__(256) void k_cheb_step(long long n, const SolverScalars *__restrict__ sc,
                                                    double d1, double d2, const double *__restrict__ dinv,
                                                    const double *__restrict__ r,
                                                    const double *__restrict__ v, double *__restrict__ w,
                                                    double *__restrict__ x)
{
  if (sc && sc->done) return;
  GRID_STRIDE(i, n) {
    const double wi = d1 * w[i] + d2 * (dinv[i] * (r[i] - v[i]));
    w[i] = wi;
    x[i] = x[i] + wi;
  }
}

// Ifpack2 compute(): lambda_max(D^-1 A) by the power method; dDinv is ready
int cheb_compute(nalu_gpu_linsys *h)
{
  const long long n = h->nOwned;
  const int vb = vec_grid(n);
  if (!h->dChebV) NG_HIP(hipMalloc(&h->dChebV, sizeof(double) * (size_t)std::max<long long>(n, 1)));
  if (!h->dChebStart) {
    std::vector<double> st((size_t)std::max(h->nCols, 1), 0.0);
    for (int i = 0; i < h->nOwned; ++i) st[i] = cheb_start_value(h->gids[i]);
    NG_HIP(hipMalloc(&h->dChebStart, sizeof(double) * st.size()));
    NG_HIP(hipMemcpy(h->dChebStart, st.data(), sizeof(double) * st.size(), hipMemcpyHostToDevice));
  }
  double *x = h->dP, *y = h->dChebV;
  double *pin = (double *)((char *)h->hPinned + 2048);
  auto fetch = [&](int cnt) -> int {
    NG_HIP(hipMemcpyAsync(pin, h->dReduce, sizeof(double) * cnt, hipMemcpyDeviceToHost, h->stream));
    NG_HIP(hipStreamSynchronize(h->stream));
    return NALU_GPU_OK;
  };
  k_dot<<<vb, 256, 0, h->stream>>>(n, h->dChebStart, h->dChebStart, h->dPartials);
  NG_TRY(finish_reduction(h, 1, vb));
  NG_TRY(fetch(1));
  double norm = std::sqrt(pin[0]), lambda = 0.0;
  if (norm != 0.0) {
    k_div_copy<<<vb, 256, 0, h->stream>>>(n, h->dChebStart, norm, x);
    for (int it = 0; it < CHEB_POWER_ITS; ++it) {
      int sb = 0;
      NG_TRY((spmv_with_halo<false, false>(h, x, y, false, &sb)));
      k_cheb_power<<<vb, 256, 0, h->stream>>>(n, h->dDinv, x, y, h->dPartials);
      NG_TRY(finish_reduction(h, 3, vb));
      NG_TRY(fetch(3));
      lambda = pin[0] / pin[1];
      norm = std::sqrt(pin[2]);
      if (norm == 0.0) { lambda = 0.0; break; }
      k_div_copy<<<vb, 256, 0, h->stream>>>(n, y, norm, x);
    }
  }
  NG_HIP(hipGetLastError());
  h->chebLambdaMax = lambda;
  return NALU_GPU_OK;
}

// x = Cheb(r); x must be a column-space vector (it is multiplied by A); uses dT and dChebV
int apply_cheb(nalu_gpu_linsys *h, const double *r, double *x, bool useDone)
{
  const long long n = h->nOwned;
  const int vb = vec_grid(n);
  const SolverScalars *sc = useDone ? h->dScal : nullptr;
  const double lambdaMax = h->chebLambdaMax;
  const double alpha = lambdaMax / CHEB_RATIO, beta = CHEB_BOOST * lambdaMax;
  const double delta = 2.0 / (beta - alpha), theta = (beta + alpha) / 2.0, s1 = theta * delta;
  double *w = h->dT, *v = h->dChebV;
  k_cheb_first<<<vb, 256, 0, h->stream>>>(n, sc, 1.0 / theta, h->dDinv, r, w, x);
  double rhok = 1.0 / s1;
  for (int deg = 1; deg < CHEB_DEGREE; ++deg) {
    int sb = 0;
    NG_TRY((spmv_with_halo<false, false>(h, x, v, useDone, &sb)));
    const double rhokp1 = 1.0 / (2.0 * s1 - rhok);
    const double d1 = rhokp1 * rhok, d2 = 2.0 * rhokp1 * delta;
    rhok = rhokp1;
    k_cheb_step<<<vb, 256, 0, h->stream>>>(n, sc, d1, d2, h->dDinv, r, v, w, x);
  }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// z = M^-1 r for the preconditioners that are not a diagonal scaling
int apply_general(nalu_gpu_linsys *h, int precond, const double *r, double *z, bool useDone)
{
  if (precond == NALU_GPU_PRECOND_CHEBYSHEV) return apply_cheb(h, r, z, useDone);
  if (precond == NALU_GPU_PRECOND_MT_SGS) return apply_mc_sgs(h, r, z, useDone);
  return apply_sgs(h, r, z, useDone);
}

// ------------------------------------------------------------------------- CG
__global__ __launch_bounds__(256) void k_cg_init(long long n, const double *__restrict__ b,
                                                  const double *__restrict__ dinv,
                                                  double *__restrict__ x, double *__restrict__ r,
                                                  double *__restrict__ p,
                                                  double *__restrict__ partials)
{
  double acc[2] = {0.0, 0.0};
  GRID_STRIDE(i, n) {
    const double ri = b[i];
    const double zi = dinv[i] * ri;
    x[i] = 0.0;
    r[i] = ri;
    p[i] = zi;
    acc[0] += ri * zi;
    acc[1] += ri * ri;
  }
  block_sums<2>(acc, partials, 2);
}

__global__ void k_cg_scal_init(SolverScalars *sc, const double *red, double *hist, double tol,
                               int maxIter, int histCap, double *alphaHist = nullptr)
{
  sc->alphaHist = alphaHist;
  sc->rz = red[0];
  sc->rr0 = red[1];
  sc->r0norm = sqrt(red[1]);
  sc->alpha = sc->beta = 0.0;
  sc->tol = tol;
  sc->iter = 0;
  sc->j = 0;
  sc->maxIter = maxIter;
  sc->histCap = histCap;
  sc->done = (sc->r0norm == 0.0) ? 1 : 0;
  if (histCap > 0) hist[0] = 1.0;
}

__global__ void k_cg_scal_a(SolverScalars *sc, const double *red) { cg_scal_a(sc, red); }

// r -= alpha q with the partial sums of r.r and r.z; the x update waits for k_cg_p, which
// reads p anyway (one vector read less per iteration)
// red != nullptr (multi-rank): alpha = rz / (p.Ap) is formed here from the all-reduced dot product
// by every thread, and recorded by one -- the one-thread scalar kernel between the all-reduce and
// this kernel (one more dependent launch per iteration) is folded away.  sc->rz is not written
// by this kernel, so every workgroup reads the same operands.
__global__ __launch_bounds__(256) void k_cg_update(long long n, SolverScalars *__restrict__ sc,
                                                    const double *__restrict__ dinv,
                                                    const double *__restrict__ q,
                                                    double *__restrict__ r,
                                                    double *__restrict__ partials,
                                                    const double *__restrict__ red)
{
  if (sc->done) return;
  double alpha;
  if (red) {
    alpha = sc->rz / red[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      sc->alpha = alpha;
      if (sc->alphaHist && sc->iter < sc->histCap) sc->alphaHist[sc->iter] = alpha;
    }
  } else {
    alpha = sc->alpha;
  }
  double acc[2] = {0.0, 0.0};
  GRID_STRIDE(i, n) {
    // streamed operands are loaded nontemporally: they do not displace the vector the next
    // SpMV gathers from (measured: -4 % per CG iteration, the SpMV itself gets 3 % faster)
    const double ri = ldnt(r + i) - alpha * ldnt(q + i);
    r[i] = ri;
    acc[0] += ri * ri;
    acc[1] += ri * (ldnt(dinv + i) * ri);
  }
  block_sums<2>(acc, partials, 2);
}

__global__ void k_cg_scal_b(SolverScalars *sc, const double *red, double *hist)
{
  cg_scal_b(sc, red, hist);
}

// p_new = z + beta p_old, written to the next slot of the direction ring
__global__ __launch_bounds__(256) void k_cg_p(long long n, const SolverScalars *__restrict__ sc,
                                               const double *__restrict__ dinv,
                                               const double *__restrict__ r,
                                               const double *__restrict__ pOld, double *__restrict__ pNew)
{
  if (sc->done) return;
  const double beta = sc->beta;
  GRID_STRIDE(i, n) pNew[i] = ldnt(dinv + i) * ldnt(r + i) + beta * ldnt(pOld + i);
}

// The solution update x += alpha_k p_k is applied in batches: the directions of up to M
// iterations stay in a ring and x is read and written once per batch instead of once per
// iteration.  The terms are added in iteration order, so x is bit-identical to the
// per-iteration update.  k0 = first iteration of the batch; the count comes from the device
// iteration counter (iterations that completed, including the one that set the done flag).
__global__ __launch_bounds__(256) void k_cg_flush_x(long long n, const SolverScalars *__restrict__ sc,
                                                     const double *__restrict__ ring, long long ld,
                                                     int M, int k0, double *__restrict__ x)
{
  const int cnt = min(M, sc->iter - k0);
  if (cnt <= 0) return;
  const double *ah = sc->alphaHist;
  GRID_STRIDE(i, n) {
    double xi = ldnt(x + i);
    for (int k = k0; k < k0 + cnt; ++k) xi += ah[k] * ldnt(ring + (long long)(k % M) * ld + i);
    x[i] = xi;
  }
}

// final true residual ||A x - b||^2 and ||b||^2 partials (src/LinearSolver.C:153-178;
// src/TpetraLinearSystem.C:1756)
__global__ __launch_bounds__(256) void k_resid_norms(long long n, const double *__restrict__ ax,
                                                      const double *__restrict__ b,
                                                      double *__restrict__ partials)
{
  double acc[2] = {0.0, 0.0};
  GRID_STRIDE(i, n) {
    const double d = ax[i] - b[i];
    acc[0] += d * d;
    acc[1] += b[i] * b[i];
  }
  block_sums<2>(acc, partials, 2);
}

// ------------------------------------------------------------------------ GMRES
// w = b - q (q = A x), partial ||w||^2 ; cycle 0 passes q = nullptr (x0 = 0 => w = b)
__global__ __launch_bounds__(256) void k_gm_resid(long long n, const double *__restrict__ b,
                                                   const double *__restrict__ q,
                                                   double *__restrict__ w, double *__restrict__ x,
                                                   double *__restrict__ partials)
{
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) {
    double wi;
    if (q) wi = b[i] - q[i];
    else { wi = b[i]; x[i] = 0.0; }
    w[i] = wi;
    acc[0] += wi * wi;
  }
  block_sums<1>(acc, partials, 1);
}

__global__ void k_gm_scal_begin(SolverScalars *sc, const double *red, double *hist, double *g,
                                int m, int cycle, double tol, int maxIter, int histCap)
{
  const double beta = sqrt(red[0]);
  if (cycle == 0) {
    sc->r0norm = beta;
    sc->rr0 = red[0];
    sc->tol = tol;
    sc->iter = 0;
    sc->maxIter = maxIter;
    sc->histCap = histCap;
    sc->done = (beta == 0.0) ? 1 : 0;
    if (histCap > 0) hist[0] = 1.0;
  } else if (beta / sc->r0norm <= sc->tol) {
    sc->done = 1;
  }
  sc->hn = beta;
  sc->j = 0;
  for (int i = 0; i <= m; ++i) g[i] = 0.0;
  g[0] = beta;
}

// v = w / hn
__global__ __launch_bounds__(256) void k_gm_scale(long long n, const SolverScalars *__restrict__ sc,
                                                   const double *__restrict__ w,
                                                   double *__restrict__ v)
{
  if (sc->done) return;
  const double hn = sc->hn;
  GRID_STRIDE(i, n) v[i] = w[i] / hn;
}

// up to 8 dot products V_{i0+t} . w in one pass over w
__global__ __launch_bounds__(256) void k_gm_dots(long long n, const SolverScalars *__restrict__ sc,
                                                  const double *__restrict__ V, long long ld, int i0,
                                                  int nv, const double *__restrict__ w,
                                                  double *__restrict__ partials)
{
  if (sc->done) return;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  GRID_STRIDE(i, n) {
    const double wv = w[i];
#pragma unroll
    for (int t = 0; t < 8; ++t)
      if (t < nv) acc[t] += ldnt(V + (long long)(i0 + t) * ld + i) * wv;
  }
  block_sums<8>(acc, partials, nv);
}

// w -= sum_i h_i V_i (i ascending, as the classical Gram-Schmidt update loop);
// workgroup 0 also accumulates the Hessenberg column: H(:,j) (+)= h
__global__ __launch_bounds__(256) void k_gm_update(long long n, const SolverScalars *__restrict__ sc,
                                                    const double *__restrict__ V, long long ld,
                                                    int nvec, const double *__restrict__ hred,
                                                    double *__restrict__ w, double *__restrict__ Hcol,
                                                    int first, double *__restrict__ partials,
                                                    int wantNorm)
{
  if (sc->done) return;
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) {
    double wv = w[i];
    for (int t = 0; t < nvec; ++t) wv -= hred[t] * ldnt(V + (long long)t * ld + i);
    w[i] = wv;
    acc[0] += wv * wv;
  }
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < nvec; t += blockDim.x) Hcol[t] = first ? hred[t] : Hcol[t] + hred[t];
  if (wantNorm) block_sums<1>(acc, partials, 1);
}

// out[t] = fixed-order sum of the nb partials of vector t (one workgroup per vector)
__global__ __launch_bounds__(256) void k_gm_reduce_cols(const double *__restrict__ partials, int nb,
                                                         double *__restrict__ out,
                                                         const SolverScalars *__restrict__ sc)
{
  if (sc->done) return;
  __shared__ double s[256];
  const double *p = partials + (size_t)blockIdx.x * nb;
  double v = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) v += p[i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

// Iterated classical Gram-Schmidt with three sweeps over the basis instead of four:
//   k_gm_dots_all : h = V^T w for all J vectors, w read once (not once per 8 vectors);
//   k_gm_upd_dots : w -= V h, then h2 = V^T w of the UPDATED w, with V read once -- a tile of V is
//                   staged in LDS between the two uses;
//   k_gm_update   : w -= V h2 and ||w||^2 (above).
// In both kernels wave v of a workgroup owns the vectors t = v, v+4, ... and keeps one
// accumulator per vector in registers over the whole grid-stride loop: one partial per
// (vector, workgroup), no cross-wave reduction; fixed grid and fixed order: deterministic.
constexpr int GM_TILE = 128;
constexpr int GM_MAXV = 52;   // vectors per pass (kspace 50 is the reference's default)
constexpr int GM_GRID = 768;  // persistent grid: three workgroups per CU
__global__ __launch_bounds__(256, 3) void k_gm_dots_all(long long n, const SolverScalars *__restrict__ sc,
                                                         const double *__restrict__ V, long long ld, int J,
                                                         const double *__restrict__ w,
                                                         double *__restrict__ partials)
{
  if (sc->done) return;
  constexpr int NA = (GM_MAXV + 3) / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) acc[q] = 0.0;
  const long long ntiles = (n + GM_TILE - 1) / GM_TILE;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long e0 = tile * GM_TILE;
    const bool in0 = e0 + lane < n, in1 = e0 + 64 + lane < n;
    const double w0 = in0 ? w[e0 + lane] : 0.0, w1 = in1 ? w[e0 + 64 + lane] : 0.0;
    double a[NA], b[NA];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int t = wv + 4 * q;
      a[q] = (t < J && in0) ? ldnt(V + (long long)t * ld + e0 + lane) : 0.0;
      b[q] = (t < J && in1) ? ldnt(V + (long long)t * ld + e0 + 64 + lane) : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      acc[q] += a[q] * w0;
      acc[q] += b[q] * w1;
    }
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int t = wv + 4 * q;
    if (t < J) {  // uniform per wave
      const double r = wave_sum(acc[q]);
      if (lane == 0) partials[(size_t)t * gridDim.x + blockIdx.x] = r;
    }
  }
}

// Tile = 128 * MULT elements, J <= GM_MAXV / MULT vectors (small J takes large tiles so that
// enough bytes are in flight).  Thread (half, e) carries the vectors t = half, half+2, ... of
// the elements e + 128 m: it fetches the NEXT tile into registers while the current one is
// worked on, forms its half of the update sum_t h_t V_t from those registers (t ascending; the
// two halves are added through LDS: w - (even + odd)) and stages the values in LDS for the
// dot products with the updated w.  LDS: (J + 3) * 128 * MULT doubles.
template <int MULT>
__global__ __launch_bounds__(256, 3) void k_gm_upd_dots(long long n, const SolverScalars *__restrict__ sc,
                                                         const double *__restrict__ V, long long ld, int J,
                                                         const double *__restrict__ hred,
                                                         double *__restrict__ w, double *__restrict__ Hcol,
                                                         int first, double *__restrict__ partials)
{
  if (sc->done) return;
  constexpr int TE = GM_TILE * MULT;
  constexpr int JM = GM_MAXV / MULT;
  constexpr int NQ = (JM + 1) / 2;  // vectors per half
  constexpr int NA = (JM + 3) / 4;  // vectors per wave in the dot phase
  extern __shared__ double lds[];   // [J][TE] tile of V | [2][TE] half sums | [TE] updated w
  double *ldsP = lds + (size_t)J * TE, *ldsW = ldsP + 2 * TE;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, e = tid & (GM_TILE - 1);
  const int half = __builtin_amdgcn_readfirstlane(tid >> 7);  // wave-uniform: scalar addresses
  double acc[NA], hq[NQ];
#pragma unroll
  for (int q = 0; q < NA; ++q) acc[q] = 0.0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) hq[q] = (half + 2 * q < J) ? hred[half + 2 * q] : 0.0;
  const long long ntiles = (n + TE - 1) / TE;
  double pre[NQ][MULT], wpre[MULT];
  auto prefetch = [&](long long tile) {
    // uniform 64-bit base per vector + one 32-bit lane offset (n < 2^29, checked by the host)
    const double *vb = V + (long long)half * ld;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
#pragma unroll
      for (int m = 0; m < MULT; ++m) {
        const unsigned off = (unsigned)(tile * TE) + (unsigned)(m * GM_TILE + e);
        pre[q][m] = (tile < ntiles && off < (unsigned)n && half + 2 * q < J) ? ldnt(vb + off) : 0.0;
      }
      vb += 2 * ld;
    }
#pragma unroll
    for (int m = 0; m < MULT; ++m) {
      const unsigned off = (unsigned)(tile * TE) + (unsigned)(m * GM_TILE + e);
      wpre[m] = (tile < ntiles && off < (unsigned)n) ? w[off] : 0.0;
    }
  };
  prefetch(blockIdx.x);
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    double wn[MULT];
#pragma unroll
    for (int m = 0; m < MULT; ++m) {
      double sum = 0.0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (half + 2 * q < J) lds[(half + 2 * q) * TE + m * GM_TILE + e] = pre[q][m];
        sum += hq[q] * pre[q][m];
      }
      ldsP[half * TE + m * GM_TILE + e] = sum;
      wn[m] = wpre[m];
    }
    __syncthreads();
    const long long e0 = tile * TE;
    prefetch(tile + gridDim.x);
    if (half == 0) {
#pragma unroll
      for (int m = 0; m < MULT; ++m) {
        const int el = m * GM_TILE + e;
        wn[m] -= ldsP[el] + ldsP[TE + el];
        if (e0 + el < n) w[e0 + el] = wn[m];
        ldsW[el] = wn[m];
      }
    }
    __syncthreads();
    double wl[2 * MULT];
#pragma unroll
    for (int m = 0; m < 2 * MULT; ++m) wl[m] = ldsW[m * 64 + lane];
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int t = wv + 4 * q;
      if (t < J) {
#pragma unroll
        for (int m = 0; m < 2 * MULT; ++m) acc[q] += lds[t * TE + m * 64 + lane] * wl[m];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int t = wv + 4 * q;
    if (t < J) {  // uniform per wave
      const double r = wave_sum(acc[q]);
      if (lane == 0) partials[(size_t)t * gridDim.x + blockIdx.x] = r;
    }
  }
  if (blockIdx.x == 0)
    for (int t = tid; t < J; t += blockDim.x) Hcol[t] = first ? hred[t] : Hcol[t] + hred[t];
}

template <int MULT>
void launch_upd_dots(nalu_gpu_linsys *h, long long n, long long ld, int J, double *Hcol)
{
  static bool ldsOptIn[64] = {};  // more than 48 KiB of dynamic LDS: once per device
  if (h->device >= 0 && h->device < 64 && !ldsOptIn[h->device]) {
    (void)hipFuncSetAttribute((const void *)k_gm_upd_dots<MULT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              65536);
    (void)hipGetLastError();
    ldsOptIn[h->device] = true;
  }
  constexpr int TE = GM_TILE * MULT;
  const int gb = (int)std::min<long long>((n + TE - 1) / TE, GM_GRID);
  const size_t ldsBytes = sizeof(double) * (size_t)(J + 3) * TE;
  k_gm_upd_dots<MULT><<<gb, 256, ldsBytes, h->stream>>>(n, h->dScal, h->dV, ld, J, h->dReduce, h->dW, Hcol, 1,
                                                        h->dPartials);
  k_gm_reduce_cols<<<J, 256, 0, h->stream>>>(h->dPartials, gb, h->dReduce, h->dScal);
}

// Arnoldi step j bookkeeping: H(j+1,j), Givens rotations, implicit residual, history
__global__ void k_gm_scal_step(SolverScalars *sc, const double *red, double *H, double *cs, double *sn,
                               double *g, double *hist, int m, int j)
{
  if (sc->done) return;
  const double hn = sqrt(red[0]);
  double *hj = H + (size_t)j * (m + 1);
  hj[j + 1] = hn;
  sc->hn = hn;
  for (int i = 0; i < j; ++i) {
    const double t = cs[i] * hj[i] + sn[i] * hj[i + 1];
    hj[i + 1] = -sn[i] * hj[i] + cs[i] * hj[i + 1];
    hj[i] = t;
  }
  const double denom = sqrt(hj[j] * hj[j] + hj[j + 1] * hj[j + 1]);
  cs[j] = hj[j] / denom;
  sn[j] = hj[j + 1] / denom;
  hj[j] = denom;
  hj[j + 1] = 0.0;
  g[j + 1] = -sn[j] * g[j];
  g[j] = cs[j] * g[j];
  sc->iter += 1;
  sc->j = j + 1;
  const double rel = fabs(g[j + 1]) / sc->r0norm;
  if (sc->iter < sc->histCap) hist[sc->iter] = rel;
  if (rel <= sc->tol || sc->iter >= sc->maxIter) sc->done = 1;
}

__global__ void k_gm_solve_y(const SolverScalars *sc, const double *H, const double *g, double *y, int m)
{
  const int jn = sc->j;
  for (int i = jn - 1; i >= 0; --i) {
    double s = g[i];
    for (int k = i + 1; k < jn; ++k) s -= H[i + (size_t)k * (m + 1)] * y[k];
    y[i] = s / H[i + (size_t)i * (m + 1)];
  }
}

// x += dinv .* (sum_i y_i V_i)
__global__ __launch_bounds__(256) void k_gm_update_x(long long n, const SolverScalars *__restrict__ sc,
                                                      const double *__restrict__ V, long long ld,
                                                      const double *__restrict__ y,
                                                      const double *__restrict__ dinv,
                                                      double *__restrict__ x)
{
  const int jn = sc->j;
  GRID_STRIDE(i, n) {
    double s = 0.0;
    for (int t = 0; t < jn; ++t) s += y[t] * V[(long long)t * ld + i];
    x[i] += dinv[i] * s;
  }
}

int read_scalars(nalu_gpu_linsys *h, SolverScalars *out)
{
  NG_HIP(hipMemcpyAsync(h->hPinned, h->dScal, sizeof(SolverScalars), hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  *out = *(SolverScalars *)h->hPinned;
  return NALU_GPU_OK;
}

int ensure_hist(nalu_gpu_linsys *h, int cap)
{
  if (cap <= h->histCap) return NALU_GPU_OK;
  NG_HIP(hipStreamSynchronize(h->stream));
  if (h->dHist) (void)hipFree(h->dHist);
  if (h->dAlphaHist) (void)hipFree(h->dAlphaHist);
  NG_HIP(hipMalloc(&h->dHist, sizeof(double) * (size_t)cap));
  NG_HIP(hipMalloc(&h->dAlphaHist, sizeof(double) * (size_t)cap));
  h->histCap = cap;
  return NALU_GPU_OK;
}

int solve_cg(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, int checkEvery)
{
  const long long n = h->nOwned;
  const int vb = vec_grid(n);
  // kernels without a reduction are not bound to the partial-sum layout: one element per
  // thread streams a few % faster than a 2048-workgroup grid-stride loop (scripts/membench)
  static const int pbDiv = getenv("NG_P_DIV") ? atoi(getenv("NG_P_DIV")) : 1;
  const int pb = (int)std::max<long long>(1, std::min<long long>((n + 256LL * pbDiv - 1) / (256LL * pbDiv), 1 << 22));
  // direction ring (see k_cg_flush_x): as many vectors as fit a modest share of the HBM
  const long long ld = std::max(h->nCols, 1);
  int M = 16;
  while (M > 2 && (double)M * ld * sizeof(double) > 24e9) M /= 2;
  if (h->pRingM != M) {
    NG_HIP(hipStreamSynchronize(h->stream));
    if (h->dPRing) (void)hipFree(h->dPRing);
    h->dPRing = nullptr;
    NG_HIP(hipMalloc(&h->dPRing, sizeof(double) * (size_t)M * (size_t)ld));
    NG_HIP(hipMemsetAsync(h->dPRing, 0, sizeof(double) * (size_t)M * (size_t)ld, h->stream));
    h->pRingM = M;
  }
  auto slot = [&](int it) { return h->dPRing + (size_t)(it % M) * (size_t)ld; };
  k_cg_init<<<vb, 256, 0, h->stream>>>(n, h->dRhs, h->dDinv, h->dX, h->dR, slot(0), h->dPartials);
  NG_HIP(hipGetLastError());
  NG_TRY(finish_reduction(h, 2, vb));
  k_cg_scal_init<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist, cfg->tolerance,
                                         cfg->max_iterations, h->histCap, h->dAlphaHist);
  SolverScalars sc;
  int flushed = 0;  // x holds the updates of the iterations before this one
  for (int it = 0; it < cfg->max_iterations; ++it) {
    int sb = 0;
    NG_TRY((spmv_with_halo<true, false>(h, slot(it), h->dQ, true, &sb)));
    const double *pp = nullptr;
    NG_TRY(prereduce_slices(h, &sb, &pp, true));
    // multi-rank: the update kernel forms alpha itself from the all-reduced p.Ap in the reduce
    // buffer (which the next reduction only overwrites after that kernel has finished)
    if (h->hasComm) {
      NG_TRY(finish_reduction(h, 1, sb, pp));
    } else {
      k_reduce_partials<EPI_CG_A, 1><<<1, RED_T, 0, h->stream>>>(pp, sb, 1, h->dReduce, h->dScal, nullptr);
    }
    k_cg_update<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dDinv, h->dQ, h->dR, h->dPartials,
                                           h->hasComm ? h->dReduce : nullptr);
    if (h->hasComm) {
      NG_TRY(finish_reduction(h, 2, vb));
      k_cg_scal_b<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist);
    } else {
      k_reduce_partials<EPI_CG_B, 2><<<1, RED_T, 0, h->stream>>>(h->dPartials, vb, 2, h->dReduce,
                                                            h->dScal, h->dHist);
    }
    if ((it + 1) % M == 0) {  // the next direction overwrites the oldest one of this batch
      k_cg_flush_x<<<pb, 256, 0, h->stream>>>(n, h->dScal, h->dPRing, ld, M, flushed, h->dX);
      flushed = it + 1;
    }
    k_cg_p<<<pb, 256, 0, h->stream>>>(n, h->dScal, h->dDinv, h->dR, slot(it), slot(it + 1));
    NG_HIP(hipGetLastError());
    if ((it + 1) % checkEvery == 0 && it + 1 < cfg->max_iterations) {
      NG_TRY(read_scalars(h, &sc));
      if (sc.done) break;
    }
  }
  k_cg_flush_x<<<pb, 256, 0, h->stream>>>(n, h->dScal, h->dPRing, ld, M, flushed, h->dX);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// ---- CG with a preconditioner that is not a diagonal scaling (SGS): z is a vector of its
// own, the residual test comes before the preconditioner application as in the oracle
__global__ __launch_bounds__(256) void k_pcg_init(long long n, const double *__restrict__ b,
                                                   double *__restrict__ x, double *__restrict__ r,
                                                   double *__restrict__ partials)
{
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) {
    const double ri = b[i];
    x[i] = 0.0;
    r[i] = ri;
    acc[0] += ri * ri;
  }
  block_sums<1>(acc, partials, 1);
}

__global__ __launch_bounds__(256) void k_dot_done(long long n, const SolverScalars *__restrict__ sc,
                                                   const double *__restrict__ a,
                                                   const double *__restrict__ b,
                                                   double *__restrict__ partials)
{
  if (sc->done) return;
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) acc[0] += a[i] * b[i];
  block_sums<1>(acc, partials, 1);
}

__global__ __launch_bounds__(256) void k_pcg_update(long long n, const SolverScalars *__restrict__ sc,
                                                     const double *__restrict__ p,
                                                     const double *__restrict__ q,
                                                     double *__restrict__ x, double *__restrict__ r,
                                                     double *__restrict__ partials)
{
  if (sc->done) return;
  const double alpha = sc->alpha;
  double acc[1] = {0.0};
  GRID_STRIDE(i, n) {
    x[i] += alpha * p[i];
    const double ri = r[i] - alpha * q[i];
    r[i] = ri;
    acc[0] += ri * ri;
  }
  block_sums<1>(acc, partials, 1);
}

__global__ void k_pcg_scal_test(SolverScalars *sc, const double *red, double *hist)
{
  if (sc->done) return;
  sc->iter += 1;
  const double rel = sqrt(red[0]) / sc->r0norm;
  if (sc->iter < sc->histCap) hist[sc->iter] = rel;
  if (rel <= sc->tol || sc->iter >= sc->maxIter) sc->done = 1;
}

__global__ void k_pcg_scal_beta(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  sc->beta = red[0] / sc->rz;
  sc->rz = red[0];
}

__global__ __launch_bounds__(256) void k_pcg_p(long long n, const SolverScalars *__restrict__ sc,
                                                const double *__restrict__ z, double *__restrict__ p,
                                                int first)
{
  if (sc->done) return;
  const double beta = sc->beta;
  GRID_STRIDE(i, n) p[i] = first ? z[i] : z[i] + beta * p[i];
}

int solve_cg_general(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, int checkEvery)
{
  const long long n = h->nOwned;
  const int vb = vec_grid(n);
  double *z = h->dW;
  k_pcg_init<<<vb, 256, 0, h->stream>>>(n, h->dRhs, h->dX, h->dR, h->dPartials);
  NG_TRY(reduce_partials(h, 1, vb, 1));  // r.r -> reduce[1]
  NG_TRY(apply_general(h, cfg->precond, h->dR, z, false));
  k_dot<<<vb, 256, 0, h->stream>>>(n, h->dR, z, h->dPartials);
  NG_TRY(reduce_partials(h, 1, vb, 0));  // r.z -> reduce[0]
  NG_TRY(allreduce(h, 2));
  k_cg_scal_init<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist, cfg->tolerance,
                                         cfg->max_iterations, h->histCap);
  k_pcg_p<<<vb, 256, 0, h->stream>>>(n, h->dScal, z, h->dP, 1);
  NG_HIP(hipGetLastError());
  SolverScalars sc;
  for (int it = 0; it < cfg->max_iterations; ++it) {
    int sb = 0;
    NG_TRY((spmv_with_halo<true, false>(h, h->dP, h->dQ, true, &sb)));
    const double *pp = nullptr;
    NG_TRY(prereduce_slices(h, &sb, &pp, true));
    NG_TRY(finish_reduction(h, 1, sb, pp));
    k_cg_scal_a<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
    k_pcg_update<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dP, h->dQ, h->dX, h->dR, h->dPartials);
    NG_TRY(finish_reduction(h, 1, vb));
    k_pcg_scal_test<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist);
    NG_TRY(apply_general(h, cfg->precond, h->dR, z, true));
    k_dot_done<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dR, z, h->dPartials);
    NG_TRY(finish_reduction(h, 1, vb));
    k_pcg_scal_beta<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
    k_pcg_p<<<vb, 256, 0, h->stream>>>(n, h->dScal, z, h->dP, 0);
    NG_HIP(hipGetLastError());
    if ((it + 1) % checkEvery == 0 && it + 1 < cfg->max_iterations) {
      NG_TRY(read_scalars(h, &sc));
      if (sc.done) break;
    }
  }
  return NALU_GPU_OK;
}

// s = sum_i y_i V_i (GMRES update with a non-diagonal right preconditioner)
__global__ __launch_bounds__(256) void k_gm_combine(long long n, const SolverScalars *__restrict__ sc,
                                                     const double *__restrict__ V, long long ld,
                                                     const double *__restrict__ y,
                                                     double *__restrict__ out)
{
  const int jn = sc->j;
  GRID_STRIDE(i, n) {
    double s = 0.0;
    for (int t = 0; t < jn; ++t) s += y[t] * V[(long long)t * ld + i];
    out[i] = s;
  }
}

int solve_gmres(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, int checkEvery)
{
  const bool sgs = cfg->precond == NALU_GPU_PRECOND_SGS || cfg->precond == NALU_GPU_PRECOND_CHEBYSHEV ||
                   cfg->precond == NALU_GPU_PRECOND_MT_SGS;
  const long long n = h->nOwned, ld = h->nCols;
  const int m = cfg->kspace;
  NG_REQUIRE(m >= 1 && m + 1 <= REDUCE_CAP, NALU_GPU_EINVAL, "kspace must be in [1,%d]", REDUCE_CAP - 1);
  if (m + 1 > h->vCap) {
    NG_HIP(hipStreamSynchronize(h->stream));
    if (h->dV) (void)hipFree(h->dV);
    h->dV = nullptr;
    NG_HIP(hipMalloc(&h->dV, sizeof(double) * (size_t)(m + 1) * (size_t)std::max<long long>(ld, 1)));
    NG_HIP(hipMemsetAsync(h->dV, 0, sizeof(double) * (size_t)(m + 1) * (size_t)std::max<long long>(ld, 1), h->stream));
    h->vCap = m + 1;
  }
  if (m > h->hCap) {
    NG_HIP(hipStreamSynchronize(h->stream));
    if (h->dH) { (void)hipFree(h->dH); (void)hipFree(h->dCs); (void)hipFree(h->dSn); (void)hipFree(h->dG); (void)hipFree(h->dY); }
    NG_HIP(hipMalloc(&h->dH, sizeof(double) * (size_t)(m + 1) * m));
    NG_HIP(hipMalloc(&h->dCs, sizeof(double) * m));
    NG_HIP(hipMalloc(&h->dSn, sizeof(double) * m));
    NG_HIP(hipMalloc(&h->dG, sizeof(double) * (m + 1)));
    NG_HIP(hipMalloc(&h->dY, sizeof(double) * m));
    h->hCap = m;
  }
  const int vb = vec_grid(n);
  static const bool fusedIcgs = !(getenv("NG_GMRES_FUSED") && atoi(getenv("NG_GMRES_FUSED")) == 0);
  SolverScalars sc;
  int enq = 0;
  for (int cycle = 0;; ++cycle) {
    if (cycle == 0) {
      k_gm_resid<<<vb, 256, 0, h->stream>>>(n, h->dRhs, nullptr, h->dW, h->dX, h->dPartials);
    } else {
      int sb = 0;
      NG_TRY((spmv_with_halo<false, false>(h, h->dX, h->dQ, false, &sb)));
      k_gm_resid<<<vb, 256, 0, h->stream>>>(n, h->dRhs, h->dQ, h->dW, h->dX, h->dPartials);
    }
    NG_HIP(hipGetLastError());
    NG_TRY(finish_reduction(h, 1, vb));
    k_gm_scal_begin<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist, h->dG, m, cycle,
                                            cfg->tolerance, cfg->max_iterations, h->histCap);
    k_gm_scale<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dW, h->dV);
    NG_HIP(hipGetLastError());
    NG_TRY(read_scalars(h, &sc));
    if (sc.done) break;
    bool stop = false;
    for (int j = 0; j < m && enq < cfg->max_iterations; ++j, ++enq) {
      double *vj = h->dV + (size_t)j * ld;
      int sb = 0;
      if (sgs) {  // w = A (M^-1 v_j)
        NG_TRY(apply_general(h, cfg->precond, vj, h->dP, true));
        NG_TRY((spmv_with_halo<false, false>(h, h->dP, h->dW, true, &sb)));
      } else {
        NG_TRY((spmv_with_halo<false, true>(h, vj, h->dW, true, &sb)));
      }
      double *Hcol = h->dH + (size_t)j * (m + 1);
      if (fusedIcgs && j + 1 <= GM_MAXV && n < (1LL << 29)) {
        // ICGS with three sweeps over V instead of four: dots | update + dots | update + norm
        const int J = j + 1;
        if (J <= 8) {
          k_gm_dots<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, 0, J, h->dW, h->dPartials);
          NG_TRY(reduce_partials(h, J, vb, 0));
        } else {
          const int gb = (int)std::min<long long>((n + GM_TILE - 1) / GM_TILE, GM_GRID);
          k_gm_dots_all<<<gb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, J, h->dW, h->dPartials);
          k_gm_reduce_cols<<<J, 256, 0, h->stream>>>(h->dPartials, gb, h->dReduce, h->dScal);
        }
        NG_TRY(allreduce(h, J));
        if (J <= 4) launch_upd_dots<8>(h, n, ld, J, Hcol);  // (J + 3) * 8 KiB of LDS
        else if (J <= GM_MAXV / 4) launch_upd_dots<4>(h, n, ld, J, Hcol);
        else if (J <= GM_MAXV / 2) launch_upd_dots<2>(h, n, ld, J, Hcol);
        else launch_upd_dots<1>(h, n, ld, J, Hcol);
        NG_TRY(allreduce(h, J));
        k_gm_update<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, J, h->dReduce, h->dW, Hcol, 0,
                                               h->dPartials, 1);
      } else
      for (int pass = 0; pass < 2; ++pass) {  // ICGS: two classical Gram-Schmidt passes
        for (int i0 = 0; i0 <= j; i0 += 8) {
          const int nv = std::min(8, j + 1 - i0);
          k_gm_dots<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, i0, nv, h->dW, h->dPartials);
          NG_TRY(reduce_partials(h, nv, vb, i0));
        }
        NG_TRY(allreduce(h, j + 1));
        k_gm_update<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, j + 1, h->dReduce, h->dW, Hcol,
                                               pass == 0, h->dPartials, pass == 1);
      }
      NG_HIP(hipGetLastError());
      NG_TRY(finish_reduction(h, 1, vb));
      k_gm_scal_step<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dH, h->dCs, h->dSn, h->dG,
                                             h->dHist, m, j);
      k_gm_scale<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dW, h->dV + (size_t)(j + 1) * ld);
      NG_HIP(hipGetLastError());
      if ((j + 1) % checkEvery == 0) {
        NG_TRY(read_scalars(h, &sc));
        if (sc.done) { stop = true; ++enq; break; }
      }
    }
    k_gm_solve_y<<<1, 1, 0, h->stream>>>(h->dScal, h->dH, h->dG, h->dY, m);
    if (sgs) {  // x += M^-1 (V y)
      k_gm_combine<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, h->dY, h->dR);
      NG_TRY(apply_general(h, cfg->precond, h->dR, h->dP, false));
      k_axpy<<<vb, 256, 0, h->stream>>>(n, 1.0, h->dP, h->dX);
    } else {
      k_gm_update_x<<<vb, 256, 0, h->stream>>>(n, h->dScal, h->dV, ld, h->dY, h->dDinv, h->dX);
    }
    NG_HIP(hipGetLastError());
    NG_TRY(read_scalars(h, &sc));
    if (stop || sc.done || sc.iter >= cfg->max_iterations) break;
  }
  return NALU_GPU_OK;
}

// ------------------------------------------------------------ BiCGStab / TFQMR
// The two other Belos methods the regression inputs name (biCgStab: 1 input, tfqmr: 3, e.g.
// reg_tests/test_files/edgeHybridFluids/edgeHybridFluids.i:10).  Right-preconditioned like the
// rest (src/LinearSolver.C:105); Belos' iteration classes are Trilinos code, so the published
// algorithms are restated (van der Vorst; Freund's TFQMR as in Saad, Alg. 7.8): parity
// unpinned.  Device resident since round 3, like cg and gmres: the scalar recurrences run in
// one-thread kernels behind the (all-)reduced dot products, the vector kernels read their
// coefficients from the device scalars, every kernel checks the device `done` flag and the host
// polls it every check_every iterations -- no scalar read-back per dot product any more.  The
// arithmetic (operations and their order, the partial-sum layout of the dot products) is the
// host-driven loop's of round 2, so the histories are the same bit for bit.

// a coefficient of the vector kernels below: sign * (*ptr), or sign alone (ptr == nullptr)
struct Coef {
  const double *ptr;
  double sign;
  __device__ __forceinline__ double get() const { return ptr ? sign * *ptr : sign; }
};

// out = a x + b y + c z (null y / z are skipped; out may alias any of them)
__global__ __launch_bounds__(256) void k_lin3(long long n, const SolverScalars *__restrict__ sc, Coef ca,
                                               const double *x, Coef cb, const double *y, Coef cc, const double *z,
                                               double *out)
{
  if (sc->done) return;
  const double a = ca.get(), b = y ? cb.get() : 0.0, c = z ? cc.get() : 0.0;
  GRID_STRIDE(i, n) {
    double v = a * x[i];
    if (y) v = v + b * y[i];
    if (z) v = v + c * z[i];
    out[i] = v;
  }
}

// p = r + beta (p - omega v)   (beta = sc->beta, omega = sc->hn)
__global__ __launch_bounds__(256) void k_bicg_p(long long n, const SolverScalars *__restrict__ sc, const double *r,
                                                 const double *v, double *p)
{
  if (sc->done) return;
  const double beta = sc->beta, omega = sc->hn;
  GRID_STRIDE(i, n) p[i] = r[i] + beta * (p[i] - omega * v[i]);
}

// v = beta (au + beta v)
__global__ __launch_bounds__(256) void k_tfqmr_v(long long n, const SolverScalars *__restrict__ sc, const double *au,
                                                  double *v)
{
  if (sc->done) return;
  const double beta = sc->beta;
  GRID_STRIDE(i, n) v[i] = beta * (au[i] + beta * v[i]);
}

__global__ __launch_bounds__(256) void k_scale_by(long long n, const SolverScalars *__restrict__ sc,
                                                   const double *dinv, const double *r, double *z)
{
  if (sc && sc->done) return;
  GRID_STRIDE(i, n) z[i] = dinv[i] * r[i];
}

// ---- scalar recurrences (one thread each).  SolverScalars: rz = rho of the previous iteration,
// alpha, beta, hn = omega (BiCGStab); aux[0] = rho of this iteration (BiCGStab) / rho (TFQMR),
// aux[1] = tau, aux[2] = theta, aux[3] = eta, aux[4] = theta^2 eta / alpha (TFQMR)
__global__ void k_kry_scal_init(SolverScalars *sc, const double *red, double *hist, double tol, int maxIter,
                                int histCap)
{
  sc->alphaHist = nullptr;
  sc->rr0 = red[0];
  sc->r0norm = sqrt(red[0]);
  sc->rz = 1.0;
  sc->alpha = 1.0;
  sc->hn = 1.0;
  sc->beta = 0.0;
  sc->tol = tol;
  sc->iter = 0;
  sc->j = 0;
  sc->maxIter = maxIter;
  sc->histCap = histCap;
  sc->done = (sc->r0norm == 0.0 || maxIter <= 0) ? 1 : 0;
  for (int i = 0; i < 8; ++i) sc->aux[i] = 0.0;
  sc->aux[0] = red[0];        // TFQMR: rho = (r*, w) = (b, b)
  sc->aux[1] = sc->r0norm;    // TFQMR: tau
  if (histCap > 0) hist[0] = 1.0;
}

__global__ void k_bi_scal_beta(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  const double rhoNew = red[0];
  sc->beta = (rhoNew / sc->rz) * (sc->alpha / sc->hn);
  sc->aux[0] = rhoNew;
}

__global__ void k_bi_scal_alpha(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  sc->alpha = sc->aux[0] / red[0];
}

__global__ void k_bi_scal_omega(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  sc->hn = red[0] / red[1];
}

__global__ void k_bi_scal_test(SolverScalars *sc, const double *red, double *hist)
{
  if (sc->done) return;
  sc->rz = sc->aux[0];
  sc->iter += 1;
  const double rel = sqrt(red[0]) / sc->r0norm;
  if (sc->iter < sc->histCap) hist[sc->iter] = rel;
  if (rel <= sc->tol || sc->iter >= sc->maxIter) sc->done = 1;
}

__global__ void k_tf_scal_alpha(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  sc->alpha = sc->aux[0] / red[0];
  sc->aux[4] = sc->aux[2] * sc->aux[2] * sc->aux[3] / sc->alpha;
}

__global__ void k_tf_scal_theta(SolverScalars *sc, const double *red)
{
  if (sc->done) return;
  const double theta = sqrt(red[0]) / sc->aux[1];
  const double c = 1.0 / sqrt(1.0 + theta * theta);
  sc->aux[2] = theta;
  sc->aux[1] = sc->aux[1] * theta * c;
  sc->aux[3] = c * c * sc->alpha;
  sc->aux[4] = sc->aux[2] * sc->aux[2] * sc->aux[3] / sc->alpha;  // (the next half step's, same alpha)
}

// end of a TFQMR iteration: beta and the new rho, then the residual estimate and the test.  The vector
// work behind it only prepares the next iteration, so a stop here skips it.
__global__ void k_tf_scal_beta_test(SolverScalars *sc, const double *red, double *hist)
{
  if (sc->done) return;
  const double rhoNew = red[0];
  sc->beta = rhoNew / sc->aux[0];
  sc->aux[0] = rhoNew;
  sc->iter += 1;
  const double rel = sqrt(2.0 * sc->iter + 1.0) * sc->aux[1] / sc->r0norm;
  if (sc->iter < sc->histCap) hist[sc->iter] = rel;
  if (rel <= sc->tol || sc->iter >= sc->maxIter) sc->done = 1;
}

// the vector side of both loops: kernels behind the done flag, dot products into the reduce buffer
struct DevKrylov {
  nalu_gpu_linsys *h;
  long long n;
  int vb, precond;
  const SolverScalars *sc() const { return h->dScal; }
  Coef one() const { return Coef{nullptr, 1.0}; }
  Coef coef(size_t off, double sign) const { return Coef{(const double *)((const char *)h->dScal + off), sign}; }
  Coef alpha(double sign = 1.0) const { return coef(offsetof(SolverScalars, alpha), sign); }
  Coef beta(double sign = 1.0) const { return coef(offsetof(SolverScalars, beta), sign); }
  Coef omega(double sign = 1.0) const { return coef(offsetof(SolverScalars, hn), sign); }
  Coef aux(int i, double sign = 1.0) const { return coef(offsetof(SolverScalars, aux) + sizeof(double) * i, sign); }
  // k dot products -> h->dReduce[0..k) (all-reduced); the first may run before the done flag exists
  int dots(int k, const double *const *a, const double *const *b, bool useDone = true)
  {
    for (int i = 0; i < k; ++i) {
      if (useDone) k_dot_done<<<vb, 256, 0, h->stream>>>(n, h->dScal, a[i], b[i], h->dPartials + (size_t)i * MAX_BLOCKS);
      else k_dot<<<vb, 256, 0, h->stream>>>(n, a[i], b[i], h->dPartials + (size_t)i * MAX_BLOCKS);
    }
    NG_HIP(hipGetLastError());
    return finish_reduction(h, k, vb);
  }
  int dot1(const double *a, const double *b, bool useDone = true) { return dots(1, &a, &b, useDone); }
  // z = M^-1 r (z is a column-space vector: it is multiplied by A next)
  int prec(const double *r, double *z, bool useDone = true)
  {
    if (precond == NALU_GPU_PRECOND_NONE || precond == NALU_GPU_PRECOND_JACOBI) {
      k_scale_by<<<vb, 256, 0, h->stream>>>(n, useDone ? h->dScal : nullptr, h->dDinv, r, z);  // (dinv = 1 without a preconditioner)
      NG_HIP(hipGetLastError());
      return NALU_GPU_OK;
    }
    return apply_general(h, precond, r, z, useDone);
  }
  int amul(double *x, double *y, bool useDone = true)
  {
    int sb = 0;
    return spmv_with_halo<false, false>(h, x, y, useDone, &sb);
  }
  void lin3(Coef a, const double *x, Coef b, const double *y, Coef c, const double *z, double *out)
  {
    k_lin3<<<vb, 256, 0, h->stream>>>(n, h->dScal, a, x, b, y, c, z, out);
  }
  void lin2(Coef a, const double *x, Coef b, const double *y, double *out) { lin3(a, x, b, y, one(), nullptr, out); }
};

int ensure_kry(nalu_gpu_linsys *h, int count)
{
  if (h->kryCount >= count) return NALU_GPU_OK;
  NG_HIP(hipStreamSynchronize(h->stream));
  if (h->dKry) (void)hipFree(h->dKry);
  h->dKry = nullptr;
  const size_t ld = (size_t)std::max(h->nCols, 1);
  NG_HIP(hipMalloc(&h->dKry, sizeof(double) * ld * (size_t)count));
  NG_HIP(hipMemsetAsync(h->dKry, 0, sizeof(double) * ld * (size_t)count, h->stream));
  h->kryCount = count;
  return NALU_GPU_OK;
}

int solve_bicgstab(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, int checkEvery)
{
  const long long n = h->nOwned;
  const size_t ld = (size_t)std::max(h->nCols, 1);
  NG_TRY(ensure_kry(h, 8));
  DevKrylov K{h, n, vec_grid(n), cfg->precond};
  double *r = h->dKry, *rh = r + ld, *p = rh + ld, *v = p + ld, *sv = v + ld, *t = sv + ld, *y = t + ld, *z = y + ld;
  double *x = h->dX;
  const size_t bytes = sizeof(double) * (size_t)n;
  NG_HIP(hipMemsetAsync(x, 0, sizeof(double) * ld, h->stream));
  NG_HIP(hipMemcpyAsync(r, h->dRhs, bytes, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipMemcpyAsync(rh, h->dRhs, bytes, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipMemsetAsync(p, 0, bytes, h->stream));
  NG_HIP(hipMemsetAsync(v, 0, bytes, h->stream));
  NG_TRY(K.dot1(r, r, false));
  k_kry_scal_init<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist, cfg->tolerance, cfg->max_iterations,
                                          h->histCap);
  SolverScalars sc;
  for (int it = 0; it < cfg->max_iterations; ++it) {
    NG_TRY(K.dot1(rh, r));
    k_bi_scal_beta<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
    k_bicg_p<<<K.vb, 256, 0, h->stream>>>(n, h->dScal, r, v, p);
    NG_TRY(K.prec(p, y));
    NG_TRY(K.amul(y, v));
    NG_TRY(K.dot1(rh, v));
    k_bi_scal_alpha<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
    K.lin2(K.one(), r, K.alpha(-1.0), v, sv);  // s = r - alpha v
    NG_TRY(K.prec(sv, z));
    NG_TRY(K.amul(z, t));
    const double *a2[2] = {t, t}, *b2[2] = {sv, t};
    NG_TRY(K.dots(2, a2, b2));
    k_bi_scal_omega<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
    K.lin3(K.one(), x, K.alpha(), y, K.omega(), z, x);  // x = x + alpha y + omega z
    K.lin2(K.one(), sv, K.omega(-1.0), t, r);            // r = s - omega t
    NG_TRY(K.dot1(r, r));
    k_bi_scal_test<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist);
    NG_HIP(hipGetLastError());
    if ((it + 1) % checkEvery == 0 && it + 1 < cfg->max_iterations) {
      NG_TRY(read_scalars(h, &sc));
      if (sc.done) break;
    }
  }
  return NALU_GPU_OK;
}

int solve_tfqmr(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, int checkEvery)
{
  const long long n = h->nOwned;
  const size_t ld = (size_t)std::max(h->nCols, 1);
  NG_TRY(ensure_kry(h, 8));
  DevKrylov K{h, n, vec_grid(n), cfg->precond};
  double *w = h->dKry, *u = w + ld, *v = u + ld, *d = v + ld, *au = d + ld, *rs = au + ld, *ys = rs + ld, *z = ys + ld;
  double *x = h->dX;
  const size_t bytes = sizeof(double) * (size_t)n;
  NG_HIP(hipMemsetAsync(x, 0, sizeof(double) * ld, h->stream));
  NG_HIP(hipMemcpyAsync(w, h->dRhs, bytes, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipMemcpyAsync(u, h->dRhs, bytes, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipMemcpyAsync(rs, h->dRhs, bytes, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipMemsetAsync(d, 0, bytes, h->stream));
  NG_HIP(hipMemsetAsync(ys, 0, bytes, h->stream));
  NG_TRY(K.dot1(rs, w, false));
  k_kry_scal_init<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist, cfg->tolerance, cfg->max_iterations,
                                          h->histCap);
  NG_TRY(K.prec(u, z));
  NG_TRY(K.amul(z, v));  // v = A M^-1 u
  K.lin2(K.one(), v, K.one(), nullptr, au);
  SolverScalars sc;
  for (int it = 0; it < cfg->max_iterations; ++it) {
    for (int half = 0; half < 2; ++half) {
      if (half == 0) {
        NG_TRY(K.dot1(v, rs));
        k_tf_scal_alpha<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
      } else {
        K.lin2(K.one(), u, K.alpha(-1.0), v, u);
        NG_TRY(K.prec(u, z));
        NG_TRY(K.amul(z, au));
      }
      K.lin2(K.one(), w, K.alpha(-1.0), au, w);
      K.lin2(K.one(), u, K.aux(4), d, d);  // d = u + (theta^2 eta / alpha) d
      NG_TRY(K.dot1(w, w));
      k_tf_scal_theta<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce);
      K.lin2(K.one(), ys, K.aux(3), d, ys);  // y += eta d
      if (half == 1) {
        NG_TRY(K.dot1(rs, w));
        k_tf_scal_beta_test<<<1, 1, 0, h->stream>>>(h->dScal, h->dReduce, h->dHist);
        K.lin2(K.one(), w, K.beta(), u, u);
        k_tfqmr_v<<<K.vb, 256, 0, h->stream>>>(n, h->dScal, au, v);
        NG_TRY(K.prec(u, z));
        NG_TRY(K.amul(z, au));
        K.lin2(K.one(), au, K.one(), v, v);
      }
    }
    NG_HIP(hipGetLastError());
    if ((it + 1) % checkEvery == 0 && it + 1 < cfg->max_iterations) {
      NG_TRY(read_scalars(h, &sc));
      if (sc.done) break;
    }
  }
  // x = M^-1 y (the loop has stopped: no done check)
  NG_TRY(K.prec(ys, z, false));
  NG_HIP(hipMemcpyAsync(x, z, bytes, hipMemcpyDeviceToDevice, h->stream));
  return NALU_GPU_OK;
}

}  // namespace

extern "C" int nalu_gpu_spmv(nalu_gpu_linsys *h, const double *x_dev, double *y_dev)
{
  NG_FINALIZED(h);
  NG_REQUIRE(x_dev && y_dev, NALU_GPU_EINVAL, "null vector");
  NG_TRY(materialize_zero(h));
  return launch_spmv<false, false>(h, x_dev, y_dev, false, nullptr);
}

extern "C" int nalu_gpu_dot(nalu_gpu_linsys *h, const double *a, const double *b, int64_t n, double *out)
{
  NG_FINALIZED(h);
  NG_REQUIRE(a && b && out && n >= 0, NALU_GPU_EINVAL, "bad dot arguments");
  const int vb = vec_grid(n);
  k_dot<<<vb, 256, 0, h->stream>>>(n, a, b, h->dPartials);
  NG_HIP(hipGetLastError());
  NG_TRY(finish_reduction(h, 1, vb));
  NG_HIP(hipMemcpyAsync(h->hPinned, h->dReduce, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  *out = *(double *)h->hPinned;
  return NALU_GPU_OK;
}

// ---- the one entry point that allocates nothing (include/nalu_gpu.h: nalu_gpu_dot_static) ----
// Scratch of the dot product's first stage (one partial sum per workgroup of k_dot) in a
// file-scope device array instead of the handle's hipMalloc'ed workspace.  `dotScratch` is a
// host-side stand-in here, NOT a device address: every GPU holds its own copy at its own
// address, looked up once per device with hipGetSymbolAddress and kept in g_dotScratchPtr;
// host code passes that pointer, never the array's name, to kernels.
static __device__ __align__(256) double dotScratch[MAX_BLOCKS];
namespace {
constexpr int DOT_STATIC_MAX_DEVICES = 64;
double *g_dotScratchPtr[DOT_STATIC_MAX_DEVICES] = {};
std::mutex g_dotScratchMutex;
}  // namespace

extern "C" int nalu_gpu_dot_static(int device, void *hip_stream, const double *a_dev, const double *b_dev,
                                   int64_t n, double *out_dev)
{
  NG_REQUIRE(n >= 0 && n <= NALU_GPU_DOT_STATIC_MAX_N, NALU_GPU_EINVAL,
             "dot_static: n = %lld is outside [0, %lld]", (long long)n, (long long)NALU_GPU_DOT_STATIC_MAX_N);
  NG_REQUIRE(out_dev && (n == 0 || (a_dev && b_dev)), NALU_GPU_EINVAL, "dot_static: null device pointer");
  int ndev = 0;
  NG_HIP(hipGetDeviceCount(&ndev));
  NG_REQUIRE(device >= 0 && device < ndev && device < DOT_STATIC_MAX_DEVICES, NALU_GPU_EINVAL,
             "dot_static: device %d of %d", device, ndev);
  NG_DEVICE(device);
  double *scratch = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_dotScratchMutex);
    if (!g_dotScratchPtr[device]) {  // first call on this device, with this device current
      void *p = nullptr;
      NG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(dotScratch)));
      NG_REQUIRE(p, NALU_GPU_EHIP, "dot_static: no device address for the scratch array");
      g_dotScratchPtr[device] = (double *)p;
    }
    scratch = g_dotScratchPtr[device];
  }
  // the same two kernels, grid and summation order as nalu_gpu_dot: bit-identical result.  Every
  // partial that the second stage reads is written by this call's first stage (vb of them), so
  // nothing of an earlier call is left to clear.
  const hipStream_t st = (hipStream_t)hip_stream;
  const int vb = vec_grid(n);
  k_dot<<<vb, 256, 0, st>>>(n, a_dev, b_dev, scratch);
  k_reduce_partials<EPI_NONE, 1><<<1, RED_T, 0, st>>>(scratch, vb, 1, out_dev, nullptr, nullptr);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_norm2(nalu_gpu_linsys *h, const double *a, int64_t n, double *out)
{
  double d = 0.0;
  NG_TRY(nalu_gpu_dot(h, a, a, n, &d));
  *out = std::sqrt(d);
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_axpy(nalu_gpu_linsys *h, double alpha, const double *x, double *y, int64_t n)
{
  NG_FINALIZED(h);
  NG_REQUIRE(x && y && n >= 0, NALU_GPU_EINVAL, "bad axpy arguments");
  k_axpy<<<vec_grid(n), 256, 0, h->stream>>>(n, alpha, x, y);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_check_system(nalu_gpu_linsys *h, nalu_gpu_check_info *info)
{
  NG_FINALIZED(h);
  NG_REQUIRE(info, NALU_GPU_EINVAL, "info is null");
  NG_TRY(materialize_zero(h));
  int *dOut = (int *)(h->dReduce + REDUCE_CAP);
  NG_HIP(hipMemsetAsync(dOut, 0, 8 * sizeof(int), h->stream));
  if (h->nSlicesOwned) k_check_rows<<<h->nSlicesOwned, SELL_C, 0, h->stream>>>(sell_view(h), h->dRhs, dOut);
  NG_HIP(hipGetLastError());
  int *o = (int *)((char *)h->hPinned + 3072);
  NG_HIP(hipMemcpyAsync(o, dOut, 6 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  info->lhs_nan_rows = o[0];
  info->rhs_nan_rows = o[1];
  info->zero_rows = o[2];
  info->first_lhs_nan_row = o[0] ? h->nOwned - o[3] : -1;
  info->first_rhs_nan_row = o[1] ? h->nOwned - o[4] : -1;
  info->first_zero_row = o[2] ? h->nOwned - o[5] : -1;
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_set_debug(nalu_gpu_linsys *h, int enabled)
{
  NG_REQUIRE(h, NALU_GPU_EINVAL, "null handle");
  h->debug = enabled != 0;
  return NALU_GPU_OK;
}

// one pass of the solve; *flowGaveUp: some rank's dataflow SGS sweep gave up waiting (the result is
// then worthless and nothing has been written to the caller's arrays)
static int solve_once(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, double *x_out_dev,
                      nalu_gpu_solve_info *info, double *history, int history_cap, bool *flowGaveUp)
{
  *flowGaveUp = false;
  NG_TRY(materialize_zero(h));
  if (h->debug) {  // realm_.debug() (src/TpetraLinearSystem.C:1716-1721)
    nalu_gpu_check_info ci;
    NG_TRY(nalu_gpu_check_system(h, &ci));
    // the solve is collective: every rank learns what any rank found, and all leave together
    long long tot[3] = {ci.lhs_nan_rows, ci.rhs_nan_rows, ci.zero_rows};
    NG_TRY(agree_counts(h, 3, tot));
    if (tot[0] || tot[1]) {
      if (ci.lhs_nan_rows || ci.rhs_nan_rows)
        return set_error(NALU_GPU_EINVAL, "bad %s: NaN in local row %lld (checkForNaN, src/TpetraLinearSystem.C:1782-1810)",
                         ci.lhs_nan_rows ? "LHS" : "RHS",
                         (long long)(ci.lhs_nan_rows ? ci.first_lhs_nan_row : ci.first_rhs_nan_row));
      return set_error(NALU_GPU_EINVAL, "bad %s: NaN in %lld row(s) of another rank (checkForNaN, "
                       "src/TpetraLinearSystem.C:1782-1810)", tot[0] ? "LHS" : "RHS", tot[0] ? tot[0] : tot[1]);
    }
    if (tot[2]) {
      if (ci.zero_rows)
        return set_error(NALU_GPU_EINVAL,
                         "ERROR checkForZeroRow in solve(): %lld row(s) with sum |a_ij| < 1e-10, first local row %lld",
                         (long long)ci.zero_rows, (long long)ci.first_zero_row);
      return set_error(NALU_GPU_EINVAL, "ERROR checkForZeroRow in solve(): %lld row(s) with sum |a_ij| < 1e-10 on "
                       "another rank", tot[2]);
    }
  }
  const int checkEvery = cfg->check_every > 0 ? cfg->check_every : 10;
  NG_TRY(ensure_hist(h, cfg->max_iterations + 2));

  // preconditioner_->compute() (src/LinearSolver.C:205), timed like timerPrecond_ (:195-211)
  const auto t0 = std::chrono::steady_clock::now();
  timer_begin(h, NALU_GPU_T_PRECOND);
  int *dBad = (int *)(h->dReduce + REDUCE_CAP);  // 8 ints behind the reduce buffer
  NG_HIP(hipMemsetAsync(dBad, 0, 8 * sizeof(int), h->stream));
  if (h->nOwned)
    k_extract_dinv<<<(h->nOwned + 255) / 256, 256, 0, h->stream>>>(h->nOwned, h->dDiagPos, h->dVals,
                                                                   cfg->precond != NALU_GPU_PRECOND_NONE,
                                                                   h->dDinv, dBad);
  if (cfg->precond == NALU_GPU_PRECOND_SGS) NG_TRY(sgs_prepare(h));
  if (cfg->precond == NALU_GPU_PRECOND_MT_SGS) NG_TRY(ensure_mc_colors(h));
  NG_HIP(hipGetLastError());
  NG_TRY(halo_exchange(h, h->dDinv));
  if (cfg->precond == NALU_GPU_PRECOND_SGS) NG_TRY(sgs_compute(h));
  if (cfg->precond == NALU_GPU_PRECOND_MT_SGS) NG_TRY(mc_sgs_compute(h));
  if (cfg->precond == NALU_GPU_PRECOND_CHEBYSHEV) NG_TRY(cheb_compute(h));
  timer_end(h, NALU_GPU_T_PRECOND);
  {
    int *bad = (int *)((char *)h->hPinned + 3072);
    NG_HIP(hipMemcpyAsync(bad, dBad, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    NG_HIP(hipStreamSynchronize(h->stream));
    // every rank must leave together (the solve is collective): the count is summed over the ranks
    // before anyone decides
    const int nBad = bad[0], firstBad = h->nOwned - bad[1];
    long long tot[1] = {nBad};
    NG_TRY(agree_counts(h, 1, tot));
    if (tot[0] > 0) {
      if (nBad > 0)
        return set_error(NALU_GPU_EINVAL,
                         "%d row(s) have a zero, missing or non-finite diagonal (first: local row %d, global id "
                         "%lld): the %s preconditioner cannot be built (Ifpack2 would produce inf; "
                         "checkForZeroRow, src/TpetraLinearSystem.C:1716-1721, catches it in debug runs)",
                         nBad, firstBad, (long long)h->gids[firstBad], "relaxation");
      return set_error(NALU_GPU_EINVAL,
                       "%lld row(s) on other ranks have a zero, missing or non-finite diagonal: the relaxation "
                       "preconditioner cannot be built (every rank leaves the solve together)", tot[0]);
    }
  }
  const auto t1 = std::chrono::steady_clock::now();

  timer_begin(h, NALU_GPU_T_SOLVE);
  if (cfg->method == NALU_GPU_METHOD_BICGSTAB) NG_TRY(solve_bicgstab(h, cfg, checkEvery));
  else if (cfg->method == NALU_GPU_METHOD_TFQMR) NG_TRY(solve_tfqmr(h, cfg, checkEvery));
  else if (cfg->method == NALU_GPU_METHOD_CG &&
      (cfg->precond == NALU_GPU_PRECOND_SGS || cfg->precond == NALU_GPU_PRECOND_CHEBYSHEV ||
       cfg->precond == NALU_GPU_PRECOND_MT_SGS))
    NG_TRY(solve_cg_general(h, cfg, checkEvery));
  else if (cfg->method == NALU_GPU_METHOD_CG) NG_TRY(solve_cg(h, cfg, checkEvery));
  else NG_TRY(solve_gmres(h, cfg, checkEvery));
  timer_end(h, NALU_GPU_T_SOLVE);

  // residual_norm (src/LinearSolver.C:153-178) and ownedRhs_->norm2()
  // (src/TpetraLinearSystem.C:1756); x gets its shared/ghost entries (sync_field :1753)
  int sb = 0;
  NG_TRY((spmv_with_halo<false, false>(h, h->dX, h->dQ, false, &sb)));
  const int vb = vec_grid(h->nOwned);
  k_resid_norms<<<vb, 256, 0, h->stream>>>(h->nOwned, h->dQ, h->dRhs, h->dPartials);
  NG_HIP(hipGetLastError());
  NG_TRY(finish_reduction(h, 2, vb));
  double *pin = (double *)((char *)h->hPinned + 1024);
  NG_HIP(hipMemcpyAsync(pin, h->dReduce, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  int *flowErr = (int *)((char *)h->hPinned + 3072) + 2;
  NG_HIP(hipMemcpyAsync(flowErr, dBad + 2, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  SolverScalars sc;
  NG_TRY(read_scalars(h, &sc));
  {  // a sweep that gave up on any rank voids the solve on all of them (SGS is rank-local, the solve is not)
    long long gaveUp[1] = {*flowErr != 0};
    NG_TRY(agree_counts(h, 1, gaveUp));
    if (gaveUp[0]) {
      *flowGaveUp = true;
      return NALU_GPU_OK;
    }
  }
  const auto t2 = std::chrono::steady_clock::now();
  info->iters = sc.iter;
  info->final_resid = std::sqrt(pin[0]);
  info->rhs_norm2 = std::sqrt(pin[1]);
  info->precond_seconds = std::chrono::duration<double>(t1 - t0).count();
  info->solve_seconds = std::chrono::duration<double>(t2 - t1).count();
  info->history_len = std::min(sc.iter + 1, h->histCap);
  if (history && history_cap > 0) {
    const int nh = std::min(info->history_len, history_cap);
    NG_HIP(hipMemcpy(history, h->dHist, sizeof(double) * (size_t)nh, hipMemcpyDeviceToHost));
    info->history_len = nh;
  }
  if (x_out_dev)
    NG_HIP(hipMemcpyAsync(x_out_dev, h->dX, sizeof(double) * (size_t)h->nCols, hipMemcpyDeviceToDevice, h->stream));
  NG_HIP(hipStreamSynchronize(h->stream));
  return NALU_GPU_OK;
}

extern "C" int nalu_gpu_solve(nalu_gpu_linsys *h, const nalu_gpu_solver_config *cfg, double *x_out_dev,
                              nalu_gpu_solve_info *info, double *history, int history_cap)
{
  NG_FINALIZED(h);
  NG_REQUIRE(cfg && info, NALU_GPU_EINVAL, "null config/info");
  NG_REQUIRE(cfg->method == NALU_GPU_METHOD_CG || cfg->method == NALU_GPU_METHOD_GMRES ||
                 cfg->method == NALU_GPU_METHOD_BICGSTAB || cfg->method == NALU_GPU_METHOD_TFQMR,
             NALU_GPU_EUNSUPPORTED, "solver method %d (supported: cg, gmres, biCgStab, tfqmr)", cfg->method);
  NG_REQUIRE(cfg->precond == NALU_GPU_PRECOND_NONE || cfg->precond == NALU_GPU_PRECOND_JACOBI ||
                 cfg->precond == NALU_GPU_PRECOND_SGS || cfg->precond == NALU_GPU_PRECOND_CHEBYSHEV ||
                 cfg->precond == NALU_GPU_PRECOND_MT_SGS,
             NALU_GPU_EUNSUPPORTED,
             "preconditioner %d (supported: none, jacobi, sgs, mt_sgs, chebyshev; the reference throws on unknown "
             "names, src/LinearSolverConfig.C:97-99)", cfg->precond);
  NG_REQUIRE(cfg->max_iterations >= 0 && cfg->tolerance >= 0.0, NALU_GPU_EINVAL, "bad solver limits");
  bool gaveUp = false;
  NG_TRY(solve_once(h, cfg, x_out_dev, info, history, history_cap, &gaveUp));
  if (gaveUp) {
    // A dataflow SGS sweep gave up waiting for rows of another chunk (a workgroup that never got its
    // turn).  The schedule with one launch per level computes the same sweeps bit for bit and waits
    // for nothing: this solve is redone on it, from scratch, and so is every later one on this handle
    // (nalu_gpu_get_sgs_schedule reports -2).
    sgs_flow_disable(h);
    NG_TRY(solve_once(h, cfg, x_out_dev, info, history, history_cap, &gaveUp));
    NG_REQUIRE(!gaveUp, NALU_GPU_ESTATE, "sgs: internal error, the level schedule reported a dataflow failure");
  }
  return NALU_GPU_OK;
}
