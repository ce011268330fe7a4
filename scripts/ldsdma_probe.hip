// probe (development aid): semantics of the direct-to-LDS 16-byte load on gfx950 --
// where lane l's data lands (base + 16 l, also with lanes switched off), 8-byte-aligned sources,
// and that s_waitcnt vmcnt(0) makes the data visible to ds_read.
// build: hipcc --offload-arch=gfx950 -O2 scripts/ldsdma_probe.hip -o /tmp/ldsdma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double *a, double *out, int shift, unsigned long long mask)
{
  __shared__ double lds[256];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) lds[i] = -1.0;
  __syncthreads();
  if ((mask >> lane) & 1ULL)
    __builtin_amdgcn_global_load_lds(a + shift + 2 * lane, (__attribute__((address_space(3))) void *)(lds + 16), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main()
{
  std::vector<double> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  double *a, *o;
  hipMalloc(&a, 8192); hipMalloc(&o, 2048);
  hipMemcpy(a, h.data(), 8192, hipMemcpyHostToDevice);
  for (int c = 0; c < 3; ++c) {
    const int shift = c == 1 ? 1 : 0;                       // case 1: 8-byte aligned only
    const unsigned long long mask = c == 2 ? 0xFFFF0000FFFF00FFULL : ~0ULL;
    k<<<1, 64>>>(a, o, shift, mask);
    std::vector<double> r(256);
    if (hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost) != hipSuccess) { printf("case %d: error %s\n", c, hipGetErrorString(hipGetLastError())); return 1; }
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
      double want = -1.0;
      if (i >= 16 && i < 16 + 128) { const int l = (i - 16) / 2; if ((mask >> l) & 1ULL) want = shift + (i - 16); }
      if (r[i] != want) { if (bad < 5) printf("  case %d: lds[%d] = %g, want %g\n", c, i, r[i], want); ++bad; }
    }
    printf("case %d (shift %d, mask %llx): %s (%d wrong)\n", c, shift, mask, bad ? "DIFFERENT" : "as expected", bad);
  }
  return 0;
}
