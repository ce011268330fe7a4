// mallbench.hip -- read rate as a function of working-set size (L2 / Infinity Cache / HBM), development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int U>
__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, long long n, double *out)
{
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n) ? a[i + u * 256] : double2{};
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
  }
  if (s == 123.456) out[0] = s;
}

int main()
{
  double *a, *out;
  const long long maxB = 2048LL << 20;
  CK(hipMalloc(&a, maxB)); CK(hipMalloc(&out, 64)); CK(hipMemset(a, 0, maxB));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (long long mb : {8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048}) {
    const long long B = mb << 20, n = B / 16;
    const int reps = (int)(16384 / mb) + 2;
    for (int nb : {2048, 8192}) {
      for (int i = 0; i < 3; ++i) k_read<4><<<nb, 256>>>((const double2 *)a, n, out);
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) k_read<4><<<nb, 256>>>((const double2 *)a, n, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("working set %5lld MiB nb=%5d: %8.1f GB/s\n", mb, nb, (double)B * reps / ms / 1e6); fflush(stdout);
    }
  }
  return 0;
}
