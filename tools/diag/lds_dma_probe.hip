// Probe: global_load_lds_dwordx4 on gfx950 - where do the lanes' 16 bytes land in the LDS, and what does an
// asynchronous window copy cost next to a register-staged one?   hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define GAS(p) ((const __attribute__((address_space(1))) void*)(p))
#define LAS(p) ((__attribute__((address_space(3))) void*)(p))
__global__ void k_layout(const float4* src, float4* dst, int n) {
  extern __shared__ float4 sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = wave * 64; i < n; i += blockDim.x)
    __builtin_amdgcn_global_load_lds(GAS(src + i + lane), LAS(sm + i), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = sm[i];
}
int main() {
  const int n = 1024;
  std::vector<float> h(n * 4), o(n * 4);
  for (int i = 0; i < n * 4; ++i) h[i] = (float)i;
  float4 *s, *d;
  hipMalloc(&s, n * 16); hipMalloc(&d, n * 16);
  hipMemcpy(s, h.data(), n * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_layout, dim3(1), dim3(512), n * 16, 0, s, d, n);
  hipMemcpy(o.data(), d, n * 16, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n * 4; ++i) if (o[i] != h[i]) { if (bad < 8) printf("mismatch at %d: %g vs %g\n", i, o[i], h[i]); ++bad; }
  printf("layout: %d mismatches of %d (lane l of a wave lands at base + 16 l when 0)\n", bad, n * 4);
  return bad != 0;
}
