// flowbench.hip -- latency of one producer -> consumer hop between workgroups through memory
// (development aid for the dataflow SGS sweep).
//
// Workgroup k waits until slot[k - stride] holds a value (data-as-flag: the slots start as the
// all-ones NaN), adds one and stores slot[k].  stride 1: every hop crosses XCDs (workgroups are
// dealt round-robin over the 8 XCDs); stride 8 / 64: eight independent chains, each inside one XCD.
//   mode 0: agent-scope relaxed atomic load / store (what the memory model prescribes)
//   mode 1: load that only bypasses the CU's L1 (sc0) + plain store: valid inside one XCD only
// Every spin loop is bounded: a workgroup that gives up sets err and all others leave too.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr unsigned long long SENT = 0xFFFFFFFFFFFFFFFFull;
constexpr int MAX_SPIN = 1 << 18;

template <int MODE>
__device__ __forceinline__ unsigned long long poll_load(const unsigned long long *p)
{
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__device__ __forceinline__ void flow_store(unsigned long long *p, unsigned long long v)
{
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(64) void k_chain(unsigned long long *slot, int stride, int *err, int *xccBad,
                                              int work)
{
  const int k = blockIdx.x;
  if (threadIdx.x == 0) {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if ((x & 0xf) != (k & 7)) atomicAdd(xccBad, 1);
  }
  double v = 0.0;
  if (k >= stride) {
    unsigned long long bits = SENT;
    int spins = 0;
    for (;;) {
      bits = poll_load<MODE>(slot + (k - stride));
      if (bits != SENT) break;
      if (++spins >= MAX_SPIN) { *(volatile int *)err = 1; return; }
      if ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
      __builtin_amdgcn_s_sleep(1);
    }
    v = __longlong_as_double((long long)bits);
  }
  for (int i = 0; i < work; ++i) v = v * 1.0000001 + 1e-9;  // a dependent chain of fp64 work per hop
  v += 1.0;
  if (threadIdx.x == 0) flow_store<MODE>(slot + k, (unsigned long long)__double_as_longlong(v));
}

template <int MODE>
static void run(int G, int stride, int work, unsigned long long *slot, int *err, int *bad)
{
  CK(hipMemset(slot, 0xFF, sizeof(unsigned long long) * (size_t)G));
  CK(hipMemset(err, 0, 4));
  CK(hipMemset(bad, 0, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k_chain<MODE><<<G, 64>>>(slot, stride, err, bad, work);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  int hErr = 0, hBad = 0;
  unsigned long long last = 0;
  CK(hipMemcpy(&hErr, err, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hBad, bad, 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&last, slot + (G - 1), 8, hipMemcpyDeviceToHost));
  double lastV;
  memcpy(&lastV, &last, 8);
  const int hops = (G - 1) / stride;
  printf("mode %d G %6d stride %3d work %4d: %8.3f ms  %7.1f ns per hop  last %.3f  err %d  xcc!=k%%8 %d\n", MODE, G,
         stride, work, ms, hops ? 1e6 * ms / hops : 0.0, lastV, hErr, hBad);
  fflush(stdout);
}

int main(int argc, char **argv)
{
  const int maxG = 1 << 16;
  unsigned long long *slot;
  int *err, *bad;
  CK(hipMalloc(&slot, sizeof(unsigned long long) * (size_t)maxG));
  CK(hipMalloc(&err, 4));
  CK(hipMalloc(&bad, 4));
  const int which = argc > 1 ? atoi(argv[1]) : 0;
  if (which == 0) {  // all workgroups resident
    for (int rep = 0; rep < 2; ++rep) {
      run<0>(2048, 1, 0, slot, err, bad);
      run<0>(2048, 8, 0, slot, err, bad);
      run<0>(2048, 64, 0, slot, err, bad);
      run<1>(2048, 8, 0, slot, err, bad);
      run<1>(2048, 64, 0, slot, err, bad);
    }
    run<0>(2048, 8, 64, slot, err, bad);
    run<1>(2048, 8, 64, slot, err, bad);
  } else {  // more workgroups than the chip holds: relies on dispatch in blockIdx order
    run<0>(maxG, 1, 0, slot, err, bad);
    run<0>(maxG, 8, 0, slot, err, bad);
    run<0>(maxG, 64, 0, slot, err, bad);
    run<1>(maxG, 64, 0, slot, err, bad);
  }
  return 0;
}
