// precond.hip -- symmetric Gauss-Seidel preconditioner (gfx950).
//
// Ifpack2 RELAXATION "Symmetric Gauss-Seidel", 1 sweep, zero starting solution, damping 1
// (src/LinearSolverConfig.C:71-75; Ifpack2 itself is Trilinos code, not in the tree):
//     z = (D + U)^-1 D (D + L)^-1 r     on the rank's owned block (off-rank couplings are
// ignored, as Ifpack2's processor-local sweeps do).  The sweeps are sequential in the row
// number.  They are reproduced EXACTLY -- same operations in the same order for every row,
// so results are bit-identical to the CPU oracle's sequential loops -- by level scheduling:
// row i goes to level 1 + max(level of its lower-numbered neighbours), rows of one level are
// independent, and one kernel launch per level walks the levels in order (the reverse
// construction for the backward sweep).  A multicolour ordering would be faster but is a
// different preconditioner (different residual history); this one is the reference's.
// Lattice 256^3: 766 levels of ~22k rows each way.
//
// Level-major storage.  The rows of a level are far apart in the SELL layout (255 rows on the
// lattice), so sweeping them in place costs a 128-byte line per 8-byte value.  compute() (once
// per solve) therefore gathers the strictly lower / strictly upper entries into level-major ELL
// blocks -- level l holds cnt_l rows x w_l entries, entry-major, with the column stored as the
// POSITION of that row in the same level order -- and apply() works on vectors permuted to that
// order: r is gathered once, the level kernels stream contiguous values and read t / z of the
// previous levels nearly contiguously, z is scattered back once.  Same operations in the same
// order per row as before: bit-identical results.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "internal.hpp"

using namespace nalu_gpu;

namespace {

// forward: t_i = (r_i - sum_{c<i} a_ic t_c) / d_i        (entries in column order)
__global__ __launch_bounds__(256) void k_sgs_forward(SellView A, const int *__restrict__ rows, int n,
                                                      const SolverScalars *__restrict__ sc,
                                                      const double *__restrict__ dinv,
                                                      const double *__restrict__ r, double *t)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off = A.sliceOff[slice] + lane;
  const int w = A.sliceWidth[slice];
  double s = r[i];
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C;
    const int c = A.cols[pos];
    if (c < i) s -= A.vals[pos] * t[c];
  }
  t[i] = s * dinv[i];
}

// backward: z_i = t_i - (sum_{i<c<n} a_ic z_c) / d_i
__global__ __launch_bounds__(256) void k_sgs_backward(SellView A, const int *__restrict__ rows, int n,
                                                       const SolverScalars *__restrict__ sc,
                                                       const double *__restrict__ dinv,
                                                       const double *__restrict__ t, double *z)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off = A.sliceOff[slice] + lane;
  const int w = A.sliceWidth[slice];
  double s = 0.0;
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C;
    const int c = A.cols[pos];
    if (c > i && c < A.nRows) s += A.vals[pos] * z[c];
  }
  z[i] = t[i] - s * dinv[i];
}

// ---- level-major sweeps
// values of the level-major blocks from the SELL matrix (compute(), once per solve)
__global__ __launch_bounds__(256) void k_lm_gather_vals(long long n, const long long *__restrict__ src,
                                                         const double *__restrict__ vals, double *__restrict__ out)
{
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long p = src[i];
  out[i] = p >= 0 ? vals[p] : 0.0;
}

// out[q] = v[rows[q]] (gather) or out[rows[q]] = v[q] (scatter)
template <bool SCATTER>
__global__ __launch_bounds__(256) void k_lm_permute(int n, const int *__restrict__ rows, const int *__restrict__ done,
                                                     const double *__restrict__ v, double *__restrict__ out)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  if (SCATTER) out[rows[q]] = v[q];
  else out[q] = v[rows[q]];
}

// one level of the forward sweep: t_q = (r_q - sum_k a_k t[col_k]) * dinv_q, k ascending (= column order)
__global__ __launch_bounds__(256) void k_lm_forward(int cnt, int w, const double *__restrict__ av,
                                                     const int *__restrict__ ac, const int *__restrict__ done,
                                                     const double *__restrict__ rq, const double *__restrict__ dinvq,
                                                     double *tq, int o)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  double s = rq[o + q];
  for (int k = 0; k < w; ++k) {
    const int c = ac[(size_t)k * cnt + q];
    if (c >= 0) s -= av[(size_t)k * cnt + q] * tq[c];
  }
  tq[o + q] = s * dinvq[o + q];
}

// one level of the backward sweep: z_q = t - (sum_k a_k z[col_k]) * dinv_q
__global__ __launch_bounds__(256) void k_lm_backward(int cnt, int w, const double *__restrict__ av,
                                                      const int *__restrict__ ac, const int *__restrict__ done,
                                                      const int *__restrict__ tpos, const double *__restrict__ tq,
                                                      const double *__restrict__ dinvq, double *zq, int o)
{
  if (done && *done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= cnt) return;
  double s = 0.0;
  for (int k = 0; k < w; ++k) {
    const int c = ac[(size_t)k * cnt + q];
    if (c >= 0) s += av[(size_t)k * cnt + q] * zq[c];
  }
  zq[o + q] = tq[tpos[o + q]] - s * dinvq[o + q];
}

// Multicolour SGS ("mt_sgs"): rows of one colour are not coupled, so a sweep is one launch per
// colour.  t and z start at zero: only colours already visited contribute, and no colour
// comparison is needed.  Columns of regular rows are computed from the slice offsets.
template <bool FWD>
__global__ __launch_bounds__(256) void k_mc_sweep(SellView A, const int *__restrict__ rows, int n,
                                                   const SolverScalars *__restrict__ sc,
                                                   const double *__restrict__ dinv,
                                                   const double *__restrict__ r, const double *__restrict__ t,
                                                   double *out)
{
  if (sc && sc->done) return;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int i = rows[q];
  const int slice = i / SELL_C, lane = i % SELL_C;
  const long long off0 = A.sliceOff[slice];
  const long long off = off0 + lane;
  const int w = A.sliceWidth[slice];
  const int *kOff = A.kOff + off0 / SELL_C;
  const bool ex = (A.exMask[(size_t)slice * 4 + (lane >> 6)] >> (lane & 63)) & 1ULL;
  double s = FWD ? r[i] : 0.0;
  for (int k = 0; k < w; ++k) {
    const long long pos = off + (long long)k * SELL_C;
    const int c = ex ? A.cols[pos] : i + kOff[k];
    if (c != i && c < A.nRows) {
      if (FWD) s -= A.vals[pos] * out[c];
      else s += A.vals[pos] * out[c];
    }
  }
  out[i] = FWD ? s * dinv[i] : t[i] - s * dinv[i];
}

// rows sorted by level (ascending row inside a level) + level offsets
void bucket_by_level(const std::vector<int> &level, int nLevels, std::vector<int> &rows,
                     std::vector<int> &off)
{
  const int n = (int)level.size();
  off.assign((size_t)nLevels + 1, 0);
  for (int i = 0; i < n; ++i) ++off[level[i] + 1];
  for (int l = 0; l < nLevels; ++l) off[l + 1] += off[l];
  rows.resize((size_t)n);
  std::vector<int> cur(off.begin(), off.end() - 1);
  for (int i = 0; i < n; ++i) rows[cur[level[i]]++] = i;
}

}  // namespace

int nalu_gpu::ensure_sgs_levels(nalu_gpu_linsys *h)
{
  if (h->sgsReady) return NALU_GPU_OK;
  const int n = h->nOwned;
  std::vector<int> lf((size_t)n, 0), lb((size_t)n, 0);
  int nf = 0, nb = 0;
  for (int i = 0; i < n; ++i) {  // sequential by construction: a level depends on lower rows
    int l = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c < i) l = std::max(l, lf[c] + 1);
    }
    lf[i] = l;
    nf = std::max(nf, l + 1);
  }
  for (int i = n - 1; i >= 0; --i) {
    int l = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c > i && c < n) l = std::max(l, lb[c] + 1);
    }
    lb[i] = l;
    nb = std::max(nb, l + 1);
  }
  std::vector<int> rows;
  bucket_by_level(lf, nf, rows, h->sgsFwdOff);
  NG_HIP(hipMalloc(&h->dSgsFwdRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dSgsFwdRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  std::vector<int> rowsF = rows;
  bucket_by_level(lb, nb, rows, h->sgsBwdOff);
  NG_HIP(hipMalloc(&h->dSgsBwdRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dSgsBwdRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  const std::vector<int> &rowsB = rows;

  // level-major blocks (see the header comment)
  std::vector<int> posF((size_t)n), posB((size_t)n);
  for (int q = 0; q < n; ++q) { posF[rowsF[q]] = q; posB[rowsB[q]] = q; }
  auto build = [&](bool fwd, const std::vector<int> &order, const std::vector<int> &off,
                   const std::vector<int> &posOf, std::vector<long long> &entOff, std::vector<int> &width,
                   std::vector<long long> &src, std::vector<int> &col) {
    const int nl = (int)off.size() - 1;
    entOff.assign((size_t)nl + 1, 0);
    width.assign((size_t)std::max(nl, 1), 0);
    for (int l = 0; l < nl; ++l) {
      int w = 0;
      for (int q = off[l]; q < off[l + 1]; ++q) {
        const int i = order[q];
        int c0 = 0;
        for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
          const int c = h->colIdx[k];
          c0 += fwd ? (c < i) : (c > i && c < n);
        }
        w = std::max(w, c0);
      }
      width[l] = w;
      entOff[l + 1] = entOff[l] + (long long)w * (off[l + 1] - off[l]);
    }
    src.assign((size_t)std::max<long long>(entOff[nl], 1), -1);
    col.assign((size_t)std::max<long long>(entOff[nl], 1), -1);
    for (int l = 0; l < nl; ++l) {
      const int cnt = off[l + 1] - off[l];
      for (int q = 0; q < cnt; ++q) {
        const int i = order[off[l] + q];
        const long long so = h->hSliceOff[i / SELL_C] + i % SELL_C;
        int kk = 0;
        for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
          const int c = h->colIdx[k];
          if (fwd ? (c < i) : (c > i && c < n)) {
            const size_t idx = (size_t)entOff[l] + (size_t)kk * cnt + q;
            src[idx] = so + (long long)(k - h->rowPtr[i]) * SELL_C;
            col[idx] = posOf[c];
            ++kk;
          }
        }
      }
    }
  };
  std::vector<long long> srcF, srcB;
  std::vector<int> colF, colB, tposB((size_t)std::max(n, 1));
  build(true, rowsF, h->sgsFwdOff, posF, h->lmFwdEntOff, h->lmFwdW, srcF, colF);
  build(false, rowsB, h->sgsBwdOff, posB, h->lmBwdEntOff, h->lmBwdW, srcB, colB);
  for (int q = 0; q < n; ++q) tposB[q] = posF[rowsB[q]];
  auto up = [&](auto &vec, auto **dst) -> int {
    using T = typename std::remove_reference<decltype(vec[0])>::type;
    NG_HIP(hipMalloc(dst, sizeof(T) * vec.size()));
    NG_HIP(hipMemcpy(*dst, vec.data(), sizeof(T) * vec.size(), hipMemcpyHostToDevice));
    return NALU_GPU_OK;
  };
  NG_TRY(up(srcF, &h->dLmFwdSrc)); NG_TRY(up(colF, &h->dLmFwdCol));
  NG_TRY(up(srcB, &h->dLmBwdSrc)); NG_TRY(up(colB, &h->dLmBwdCol));
  NG_TRY(up(tposB, &h->dLmTpos));
  h->lmFwdN = (long long)srcF.size();
  h->lmBwdN = (long long)srcB.size();
  NG_HIP(hipMalloc(&h->dLmFwdVal, sizeof(double) * srcF.size()));
  NG_HIP(hipMalloc(&h->dLmBwdVal, sizeof(double) * srcB.size()));
  const size_t nv = sizeof(double) * (size_t)std::max(n, 1);
  NG_HIP(hipMalloc(&h->dLmDinvF, nv)); NG_HIP(hipMalloc(&h->dLmDinvB, nv));
  NG_HIP(hipMalloc(&h->dLmR, nv)); NG_HIP(hipMalloc(&h->dLmT, nv)); NG_HIP(hipMalloc(&h->dLmZ, nv));
  h->sgsReady = true;
  return NALU_GPU_OK;
}

// Ifpack2 compute() for SGS: refresh the level-major copies of the matrix values and of dinv
// (the matrix changes between solves, the level structure does not).  dDinv must be ready.
int nalu_gpu::sgs_compute(nalu_gpu_linsys *h)
{
  NG_TRY(ensure_sgs_levels(h));
  const int n = h->nOwned;
  if (n == 0) return NALU_GPU_OK;
  k_lm_gather_vals<<<(unsigned)((h->lmFwdN + 255) / 256), 256, 0, h->stream>>>(h->lmFwdN, h->dLmFwdSrc, h->dVals,
                                                                                h->dLmFwdVal);
  k_lm_gather_vals<<<(unsigned)((h->lmBwdN + 255) / 256), 256, 0, h->stream>>>(h->lmBwdN, h->dLmBwdSrc, h->dVals,
                                                                                h->dLmBwdVal);
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsFwdRows, nullptr, h->dDinv, h->dLmDinvF);
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsBwdRows, nullptr, h->dDinv, h->dLmDinvB);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// z = SGS(r) on the owned rows (level-major sweeps; sgs_compute must have run for the current
// matrix).  r and z may alias each other.  NG_SGS_INPLACE=1 selects the first implementation (the
// sweeps on the SELL matrix itself, 5x slower at 256^3), kept as the cross-check of the tests.
int nalu_gpu::apply_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone)
{
  NG_TRY(ensure_sgs_levels(h));
  const int nf = (int)h->sgsFwdOff.size() - 1, nb = (int)h->sgsBwdOff.size() - 1;
  static const bool inplace = getenv("NG_SGS_INPLACE") && atoi(getenv("NG_SGS_INPLACE")) != 0;
  if (inplace) {
    const SellView A = sell_view(h);
    const SolverScalars *sc = useDone ? h->dScal : nullptr;
    for (int l = 0; l < nf; ++l) {
      const int o = h->sgsFwdOff[l], cnt = h->sgsFwdOff[l + 1] - o;
      k_sgs_forward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dSgsFwdRows + o, cnt, sc, h->dDinv, r,
                                                              h->dT);
    }
    for (int l = 0; l < nb; ++l) {
      const int o = h->sgsBwdOff[l], cnt = h->sgsBwdOff[l + 1] - o;
      k_sgs_backward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dSgsBwdRows + o, cnt, sc, h->dDinv,
                                                               h->dT, z);
    }
    NG_HIP(hipGetLastError());
    return NALU_GPU_OK;
  }
  const int n = h->nOwned;
  if (n == 0) return NALU_GPU_OK;
  const int *done = useDone ? &h->dScal->done : nullptr;
  // ~2 x (number of levels) dependent launches (1532 at 256^3), each a kernel boundary plus two
  // dependent memory round trips (column positions -> t values): ~6 us per level, latency-bound.
  // (Replaying them from a HIP graph was measured and changes nothing: the time is on the device.)
  k_lm_permute<false><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsFwdRows, done, r, h->dLmR);
  for (int l = 0; l < nf; ++l) {
    const int o = h->sgsFwdOff[l], cnt = h->sgsFwdOff[l + 1] - o;
    k_lm_forward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, h->lmFwdW[l], h->dLmFwdVal + h->lmFwdEntOff[l],
                                                           h->dLmFwdCol + h->lmFwdEntOff[l], done, h->dLmR,
                                                           h->dLmDinvF, h->dLmT, o);
  }
  for (int l = 0; l < nb; ++l) {
    const int o = h->sgsBwdOff[l], cnt = h->sgsBwdOff[l + 1] - o;
    k_lm_backward<<<(cnt + 255) / 256, 256, 0, h->stream>>>(cnt, h->lmBwdW[l], h->dLmBwdVal + h->lmBwdEntOff[l],
                                                            h->dLmBwdCol + h->lmBwdEntOff[l], done, h->dLmTpos,
                                                            h->dLmT, h->dLmDinvB, h->dLmZ, o);
  }
  k_lm_permute<true><<<(n + 255) / 256, 256, 0, h->stream>>>(n, h->dSgsBwdRows, done, h->dLmZ, z);
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}

// greedy colouring in row order (the oracle's orc_greedy_colors): colour of row i = smallest
// colour none of its lower-numbered owned neighbours has
int nalu_gpu::ensure_mc_colors(nalu_gpu_linsys *h)
{
  if (h->mcReady) return NALU_GPU_OK;
  const int n = h->nOwned;
  std::vector<int> color((size_t)n, 0);
  int nc = 0;
  for (int i = 0; i < n; ++i) {
    unsigned long long used = 0;
    for (int64_t k = h->rowPtr[i]; k < h->rowPtr[i + 1]; ++k) {
      const int c = h->colIdx[k];
      if (c < i) used |= 1ull << color[c];
    }
    int col = 0;
    while (col < 63 && ((used >> col) & 1ull)) ++col;
    NG_REQUIRE(col < 63, NALU_GPU_EUNSUPPORTED, "mt_sgs: the graph needs more than 63 colours");
    color[i] = col;
    nc = std::max(nc, col + 1);
  }
  std::vector<int> rows;
  bucket_by_level(color, nc, rows, h->mcOff);
  NG_HIP(hipMalloc(&h->dMcRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dMcRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  h->mcReady = true;
  return NALU_GPU_OK;
}

// z = multicolour SGS(r) on the owned rows; uses h->dT as the intermediate vector.
int nalu_gpu::apply_mc_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone)
{
  NG_TRY(ensure_mc_colors(h));
  const SellView A = sell_view(h);
  const SolverScalars *sc = useDone ? h->dScal : nullptr;
  const int nc = (int)h->mcOff.size() - 1;
  const size_t bytes = sizeof(double) * (size_t)h->nOwned;
  NG_HIP(hipMemsetAsync(h->dT, 0, bytes, h->stream));
  NG_HIP(hipMemsetAsync(z, 0, bytes, h->stream));
  for (int c = 0; c < nc; ++c) {
    const int o = h->mcOff[c], cnt = h->mcOff[c + 1] - o;
    if (cnt)
      k_mc_sweep<true><<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dMcRows + o, cnt, sc, h->dDinv, r,
                                                                 nullptr, h->dT);
  }
  for (int c = nc - 1; c >= 0; --c) {
    const int o = h->mcOff[c], cnt = h->mcOff[c + 1] - o;
    if (cnt)
      k_mc_sweep<false><<<(cnt + 255) / 256, 256, 0, h->stream>>>(A, h->dMcRows + o, cnt, sc, h->dDinv,
                                                                  nullptr, h->dT, z);
  }
  NG_HIP(hipGetLastError());
  return NALU_GPU_OK;
}
