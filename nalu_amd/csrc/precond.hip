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
