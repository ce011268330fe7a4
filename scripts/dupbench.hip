// dupbench.hip -- does the L2 fetch a line twice when two workgroups ask for it at the same
// time?  (development aid).  65536 workgroups of 256 threads; workgroup s reads its own
// 256-double line of 11 component arrays (the row kernel's own nodal data) and, depending on
// the mode, the line of a neighbouring workgroup.  Run each mode under
// `rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum` and compare with mode 0 (1.47 GB compulsory).
//   0 own only            1 own + line s+1 at once      2 own, then line s+1 one round trip later
//   3 own + line s-1      4 own, then line s+1 two round trips later
//   5 own + line s+256 at once (the z neighbour)        6 the same one round trip later
//   7 own + all four neighbour lines (s-1, s+1, s-256, s+256): the row kernel's gathers without
//     its arithmetic -- 57.7 M lines through the L2 per launch
//   8 own + eight 8-byte stores per row (the matrix values + rhs of the row kernel)
//   9 the same with nontemporal stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NF = 11;
struct Fields { const double *f[NF]; };

__device__ __forceinline__ int xcd_run(int b, int n, int G)
{
  const int full = n / (8 * G) * (8 * G);
  if (b >= full) return b;
  const int xcd = b & 7, idx = b >> 3;
  return ((idx / G) * 8 + xcd) * G + idx % G;
}

__global__ __launch_bounds__(256) void k(Fields F, int nSlices, int mode, double *out, double *big)
{
  const int s = xcd_run(blockIdx.x, nSlices, 8);
  const long long n = (long long)nSlices * 256;
  const long long row = (long long)s * 256 + threadIdx.x;
  double own[NF], acc = 0.0;
#pragma unroll
  for (int i = 0; i < NF; ++i) own[i] = F.f[i][row];
  long long d = mode == 3 ? -256 : (mode >= 5 ? 256 * 256 : 256);
  if (mode == 2 || mode == 4 || mode == 6) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < NF; ++i) t += own[i];
    if (t == 1.2345e300) d += 1;  // never true: makes the neighbour address depend on the own data
    if (mode == 4) {
      long long r2 = row + (t == 2.5e300 ? 1 : 0);
      double u = F.f[0][r2] + F.f[5][r2];  // a second dependent round trip (hits)
      if (u == 1.2345e300) d += 1;
    }
  }
  if (mode == 7) {
    const long long ds[4] = {-256, 256, -256 * 256, 256 * 256};
    for (int j = 0; j < 4; ++j) {
      long long c = row + ds[j];
      if (c < 0 || c >= n) c = row;
#pragma unroll
      for (int i = 0; i < NF; ++i) acc += F.f[i][c];
    }
  } else if (mode != 0) {
    long long c = row + d;
    if (c < 0) c = row;
    if (c >= n) c = row;
#pragma unroll
    for (int i = 0; i < NF; ++i) acc += F.f[i][c];
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) acc += own[i];
  if (mode == 8 || mode == 9) {
    const long long base = (long long)s * 256 * 8 + threadIdx.x;  // SELL-like: 8 entries x 256 rows
    if (mode == 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) big[base + j * 256] = acc + j;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(acc + j, big + base + j * 256);
    }
  }
  if (acc == 1.2345e300) out[0] = acc;
}

int main(int argc, char **argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int nSlices = 65536;
  const long long n = (long long)nSlices * 256;
  Fields F;
  for (int i = 0; i < NF; ++i) {
    double *p;
    CK(hipMalloc(&p, n * 8));
    CK(hipMemset(p, 0, n * 8));
    F.f[i] = p;
  }
  double *out, *big;
  CK(hipMalloc(&out, 8));
  CK(hipMalloc(&big, n * 8 * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    k<<<nSlices, 256>>>(F, nSlices, mode, out, big);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 2) printf("mode %d: %.3f ms, own bytes %.2f GB -> %.2f TB/s of own bytes\n", mode, ms, n * 8.0 * NF / 1e9, n * 8.0 * NF / ms / 1e9);
  }
  return 0;
}
