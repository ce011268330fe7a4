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
#include <algorithm>
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
  bucket_by_level(lb, nb, rows, h->sgsBwdOff);
  NG_HIP(hipMalloc(&h->dSgsBwdRows, sizeof(int) * (size_t)std::max(n, 1)));
  NG_HIP(hipMemcpy(h->dSgsBwdRows, rows.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  h->sgsReady = true;
  return NALU_GPU_OK;
}

// z = SGS(r) on the owned rows; uses h->dT as the intermediate vector.  r and z may alias
// neither each other nor dT.
int nalu_gpu::apply_sgs(nalu_gpu_linsys *h, const double *r, double *z, bool useDone)
{
  NG_TRY(ensure_sgs_levels(h));
  const SellView A = sell_view(h);
  const SolverScalars *sc = useDone ? h->dScal : nullptr;
  const int nf = (int)h->sgsFwdOff.size() - 1, nb = (int)h->sgsBwdOff.size() - 1;
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
