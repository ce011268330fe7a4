// membench.hip -- calibration of streaming HBM rates on one MI355X (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <typename T> __device__ double tsum(T v);
template <> __device__ double tsum<double>(double v) { return v; }
template <> __device__ double tsum<double2>(double2 v) { return v.x + v.y; }
template <> __device__ double tsum<double4>(double4 v) { return v.x + v.y + v.z + v.w; }

// grid-stride read: each thread reads U elements of type T per trip, trips strided by the grid
template <typename T, int U>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ a, long long n, double *out)
{
  double s = 0.0;
  const long long stride = (long long)gridDim.x * 256 * U;
  for (long long i = (long long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n) ? a[i + u * 256] : T{};
#pragma unroll
    for (int u = 0; u < U; ++u) s += tsum<T>(v[u]);
  }
  if (s == 123.456) out[0] = s;
}

// block-contiguous read: workgroup b reads its own contiguous chunk
template <typename T, int U>
__global__ __launch_bounds__(256) void k_read_chunk(const T *__restrict__ a, long long n, long long per, double *out)
{
  double s = 0.0;
  const long long b0 = (long long)blockIdx.x * per, b1 = min(n, b0 + per);
  for (long long i = b0 + threadIdx.x; i < b1; i += 256 * U) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < b1) ? a[i + u * 256] : T{};
#pragma unroll
    for (int u = 0; u < U; ++u) s += tsum<T>(v[u]);
  }
  if (s == 123.456) out[0] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void k_copy(const T *__restrict__ a, T *__restrict__ b, long long n)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = a[i];
}

template <typename T>
__global__ __launch_bounds__(256) void k_write(T *__restrict__ b, long long n)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) b[i] = T{};
}

template <typename T>
__global__ __launch_bounds__(256) void k_write_nt(T *__restrict__ b, long long n)
{
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) __builtin_nontemporal_store(T{}, &b[i]);
}

// CG-update shaped: 5 reads, 2 writes
template <bool NT>
__global__ __launch_bounds__(256) void k_upd(long long n, const double *__restrict__ p, const double *__restrict__ q, const double *__restrict__ d, double *__restrict__ x, double *__restrict__ r, double *out)
{
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double xi = x[i] + 0.5 * p[i];
    const double ri = r[i] - 0.5 * q[i];
    if (NT) { __builtin_nontemporal_store(xi, &x[i]); __builtin_nontemporal_store(ri, &r[i]); }
    else { x[i] = xi; r[i] = ri; }
    acc += ri * d[i] * ri;
  }
  if (acc == 123.456) out[0] = acc;
}

template <typename F> float timeit(F f, int reps = 10)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const long long bytes = (argc > 1 ? atoll(argv[1]) : 1024) * 1024LL * 1024LL / 1024 * 1024 / 1024;  // MiB
  const long long B = (argc > 1 ? atoll(argv[1]) : 1024) << 20;
  (void)bytes;
  double *a, *b, *out;
  CK(hipMalloc(&a, B)); CK(hipMalloc(&b, B)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, B)); CK(hipMemset(b, 0, B));
  const long long n = B / 8;
  auto rep = [&](const char *name, float ms, double mult) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, mult * B / ms / 1e6); fflush(stdout); };
  for (int nb : {2048}) {
    char nm[128];
    snprintf(nm, 128, "read f64   grid-stride U1 nb=%d", nb); rep(nm, timeit([&] { k_read<double, 1><<<nb, 256>>>(a, n, out); }), 1);
    snprintf(nm, 128, "read f64   grid-stride U4 nb=%d", nb); rep(nm, timeit([&] { k_read<double, 4><<<nb, 256>>>(a, n, out); }), 1);
    snprintf(nm, 128, "read f64x2 grid-stride U1 nb=%d", nb); rep(nm, timeit([&] { k_read<double2, 1><<<nb, 256>>>((double2 *)a, n / 2, out); }), 1);
    snprintf(nm, 128, "read f64x2 grid-stride U4 nb=%d", nb); rep(nm, timeit([&] { k_read<double2, 4><<<nb, 256>>>((double2 *)a, n / 2, out); }), 1);
    snprintf(nm, 128, "read f64x4 grid-stride U2 nb=%d", nb); rep(nm, timeit([&] { k_read<double4, 2><<<nb, 256>>>((double4 *)a, n / 4, out); }), 1);
    snprintf(nm, 128, "read f64   chunk U7 nb=%d", nb); rep(nm, timeit([&] { k_read_chunk<double, 7><<<nb, 256>>>(a, n, (n + nb - 1) / nb, out); }), 1);
    snprintf(nm, 128, "read f64x2 chunk U4 nb=%d", nb); rep(nm, timeit([&] { k_read_chunk<double2, 4><<<nb, 256>>>((double2 *)a, n / 2, (n / 2 + nb - 1) / nb, out); }), 1);
  }
  rep("copy f64x2 nb=2048 (r+w bytes)", timeit([&] { k_copy<double2><<<2048, 256>>>((double2 *)a, (double2 *)b, n / 2); }), 2);
  rep("copy f32x4 nb=8192 (r+w bytes)", timeit([&] { k_copy<float4><<<8192, 256>>>((float4 *)a, (float4 *)b, n / 2); }), 2);
  rep("write f64x2 nb=2048", timeit([&] { k_write<double2><<<2048, 256>>>((double2 *)b, n / 2); }), 1);
  rep("write f64 nb=2048", timeit([&] { k_write<double><<<2048, 256>>>(b, n); }), 1);
  rep("write f64 nb=16384", timeit([&] { k_write<double><<<16384, 256>>>(b, n); }), 1);
  rep("write f64x2 nb=16384", timeit([&] { k_write<double2><<<16384, 256>>>((double2 *)b, n / 2); }), 1);
  rep("write nt f64 nb=2048", timeit([&] { k_write_nt<double><<<2048, 256>>>(b, n); }), 1);
  rep("write nt f64 nb=16384", timeit([&] { k_write_nt<double><<<16384, 256>>>(b, n); }), 1);
  rep("write nt f32 nb=2048", timeit([&] { k_write_nt<float><<<2048, 256>>>((float *)b, n * 2); }), 1);
  {
    const long long m = n / 5;  // five vectors in a, two of them written
    double *v = a;
    const long long m16 = 16777216;
    for (int nb : {1024, 2048, 4096, 8192, 16384, 32768, 65536}) {
      char nm[128];
      snprintf(nm, 128, "cg-update shape n=256^3 nb=%d", nb);
      float ms = timeit([&] { k_upd<false><<<nb, 256>>>(m16, v, v + m16, v + 2 * m16, v + 3 * m16, v + 4 * m16, out); });
      printf("%-44s %8.3f ms  %7.1f GB/s\n", nm, ms, 7.0 * m16 * 8 / ms / 1e6);
      snprintf(nm, 128, "write f64 nb=%d", nb); rep(nm, timeit([&] { k_write<double><<<nb, 256>>>(b, n); }), 1);
      snprintf(nm, 128, "copy f64 nb=%d (r+w)", nb); rep(nm, timeit([&] { k_copy<double><<<nb, 256>>>(a, b, n); }), 2);
    }
  }
  return 0;
}
