// Throughput / same-address chain cost of fp32 global atomics on gfx950 (diagnostic; not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_atomic(float* buf, int A, int per_thread) {
  for (int k = 0; k < per_thread; ++k) {
    const int idx = (threadIdx.x + k * blockDim.x) % A;
    atomicAdd(buf + idx, 1.0f);
  }
}
__global__ void k_store(float* buf, int A, int per_thread) {
  for (int k = 0; k < per_thread; ++k) {
    const int idx = (threadIdx.x + k * blockDim.x) % A;
    buf[(size_t)blockIdx.x * A + idx] = 1.0f;
  }
}
int main() {
  float* buf; hipMalloc(&buf, 1ull << 30); hipMemset(buf, 0, 1ull << 30);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfg[][3] = {{256, 256, 1}, {1024, 256, 1}, {2048, 256, 1}, {256, 1024, 2}, {256, 4096, 8}, {1024, 4096, 8},
                        {256, 65536, 128}, {64, 65536, 128}, {256, 196608, 384}, {64, 196608, 384}, {2048, 3072, 6}};
  for (auto& c : cfg) {
    for (int mode = 0; mode < 2; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) k_atomic<<<c[0], 512>>>(buf, c[1], c[2]); else k_store<<<c[0], 512>>>(buf, c[1], c[2]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double n = (double)c[0] * 512 * c[2];
      printf("%s WGs %5d addresses %7d per-thread %4d : %8.1f us  %7.2f G/s  chain %d -> %.1f ns each\n", mode ? "store " : "atomic",
             c[0], c[1], c[2], best * 1e3, n / best / 1e6, c[0] * (512 * c[2] / c[1]), best * 1e6 / (c[0] * (512.0 * c[2] / c[1])));
    }
  }
  return 0;
}
