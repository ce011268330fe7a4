// storebench.hip -- what does a wave's store cost the L2 (TCC_REQ), by store width?  (development aid, round 3)
// 65536 workgroups x 256 rows; every row writes 8 doubles in the SELL-256 layout (entry k of row r at
// slice * 2048 + k * 256 + r), as the assembly kernels do.
//   0: eight 8-byte stores per lane (one 512-byte run per wave instruction)
//   1: lane pairs swap one value by DPP (quad_perm 1,0,3,2), then FOUR 16-byte stores per lane: even lanes
//      write rows (2i, 2i+1) of entry k, odd lanes rows (2i, 2i+1) of entry k + 1 (two 512-byte runs per
//      wave instruction)
//   2: as 0 with nontemporal stores      3: as 1 with nontemporal stores
// Run under `rocprofv3 --pmc WRITE_SIZE TCC_REQ_sum` for the request counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double swap_pair(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void k(const double *in, int mode, double *big)
{
  const int s = blockIdx.x, r = threadIdx.x;
  const double a = in[(long long)s * 256 + r];
  double v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = a * (j + 1);
  double *base = big + (long long)s * 2048;
  if (mode == 0 || mode == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (mode == 0) base[j * 256 + r] = v[j];
      else __builtin_nontemporal_store(v[j], base + j * 256 + r);
    }
  } else {
    const bool odd = r & 1;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      // even lane keeps v[j] and gets the odd lane's v[j]; odd lane keeps v[j+1] and gets the even lane's v[j+1]
      const double give = odd ? v[j] : v[j + 1];
      const double got = swap_pair(give);
      double2 w;
      w.x = odd ? got : v[j];
      w.y = odd ? v[j + 1] : got;
      double2 *dst = (double2 *)(base + (j + (odd ? 1 : 0)) * 256 + (r & ~1));
      if (mode == 1) *dst = w;
      else {
        typedef double v2d __attribute__((ext_vector_type(2)));
        v2d ww = {w.x, w.y};
        __builtin_nontemporal_store(ww, (v2d *)dst);
      }
    }
  }
}

int main(int argc, char **argv)
{
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const int nSlices = 65536;
  const long long n = (long long)nSlices * 256;
  double *in, *big;
  CK(hipMalloc(&in, n * 8));
  CK(hipMalloc(&big, n * 8 * 8));
  CK(hipMemset(in, 0, n * 8));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    k<<<nSlices, 256>>>(in, mode, big);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 2) printf("mode %d: %.3f ms, %.2f GB written -> %.2f TB/s\n", mode, ms, n * 64.0 / 1e9, n * 64.0 / ms / 1e9);
  }
  // check: entry j of row r = in * (j+1) = 0 here; use a pattern instead
  return 0;
}
