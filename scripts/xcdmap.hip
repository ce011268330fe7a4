// xcdmap.hip -- which XCD and CU does workgroup b run on? (development aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *xcc, int *cu, long long *t, int spin)
{
  if (threadIdx.x == 0) {
    int x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    xcc[blockIdx.x] = x & 0xf;
    cu[blockIdx.x] = h;
    t[blockIdx.x] = wall_clock64();
  }
  // keep the workgroup alive for a while so that occupancy fills up
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}
int main()
{
  const int nb = 4096;
  int *dx, *dc; long long *dt;
  hipMalloc(&dx, nb * 4); hipMalloc(&dc, nb * 4); hipMalloc(&dt, nb * 8);
  for (int spin : {0, 20000}) {
    k<<<nb, 256>>>(dx, dc, dt, spin);
    hipDeviceSynchronize();
    std::vector<int> x(nb), c(nb); std::vector<long long> t(nb);
    hipMemcpy(x.data(), dx, nb * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, nb * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), dt, nb * 8, hipMemcpyDeviceToHost);
    int rr = 0;
    for (int b = 0; b < nb; ++b) rr += (x[b] == b % 8);
    printf("spin %d: blocks with xcc == b %% 8: %d of %d\n", spin, rr, nb);
    printf("first 32 xcc:");
    for (int b = 0; b < 32; ++b) printf(" %d", x[b]);
    printf("\nblock 2040..2080 xcc:");
    for (int b = 2040; b < 2080; ++b) printf(" %d", x[b]);
    // dispatch time order within xcc 0: are idx dispatched in order?
    long long t0 = t[0];
    printf("\nstart ticks of xcc0 blocks idx 0..40 step 1 (x8):");
    for (int b = 0; b < 8 * 40; b += 8) printf(" %lld", t[b] - t0);
    printf("\nstart ticks idx 250..300:");
    for (int b = 8 * 250; b < 8 * 300; b += 8) printf(" %lld", t[b] - t0);
    printf("\n");
  }
  return 0;
}
