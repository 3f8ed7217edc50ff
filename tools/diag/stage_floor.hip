// What a U-Net stage-sized launch costs at best ("show the floor with a measurement").
// A training step of the conv U-Net is 30 dependent launches that each move one or two 17 MB stage tensors
// (2048 windows x 2 x 512 floats = 8.4 MB in, 8.4 MB out, sometimes a skip).  This probe times chains of 30 dependent
// launches of (a) an empty kernel, (b) a pure copy of one stage tensor with the stage kernels' geometry (512 workgroups x
// 512 threads, a workgroup takes 4 consecutive windows, every load of the pass in flight before the first store),
// (c) the same with a per-channel sum reduced to 64 double atomics per workgroup (what a BatchNorm stage must add).
//   hipcc --offload-arch=gfx950 -O3 -o tools/diag/stage_floor tools/diag/stage_floor.hip && ./tools/diag/stage_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_empty() {}
template <bool SUMS>
__global__ __launch_bounds__(512) void k_stage_copy(const float4* __restrict__ in, float4* __restrict__ out, double* __restrict__ sums,
                                                    int n4_per_wg) {
  const float4* src = in + (size_t)blockIdx.x * n4_per_wg;
  float4* dst = out + (size_t)blockIdx.x * n4_per_wg;
  float4 v[4];
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int i = threadIdx.x + k * 512; v[k] = src[i < n4_per_wg ? i : 0]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = threadIdx.x + k * 512;
    if (i < n4_per_wg) { dst[i] = v[k]; acc += v[k].x + v[k].y + v[k].z + v[k].w; }
  }
  if (SUMS) {
    __shared__ float red[64];
    if (threadIdx.x < 64) red[threadIdx.x] = 0.f;
    __syncthreads();
    atomicAdd(&red[threadIdx.x & 63], acc);
    __syncthreads();
    if (threadIdx.x < 64) atomicAdd(sums + (blockIdx.x & 15) * 64 + threadIdx.x, (double)red[threadIdx.x]);
  }
}
int main() {
  const size_t n4 = (size_t)2048 * 2 * 512 / 4;          // one stage tensor
  float4 *a, *b; double* s;
  hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&s, 16 * 64 * 8);
  hipMemset(a, 0, n4 * 16); hipMemset(s, 0, 16 * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 30; ++i) launch(i);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 20; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 30; ++i) launch(i);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-64s %7.2f us per launch (chain of 30: %.1f us)\n", name, best * 1e3f / 30, best * 1e3f);
  };
  timeit("empty kernel, 512 x 512", [&](int) { k_empty<<<512, 512>>>(); });
  for (int wgs : {512, 1024, 2048}) {
    char nm[96];
    snprintf(nm, 96, "copy 8.4 MB -> 8.4 MB, %d workgroups x 512 threads", wgs);
    timeit(nm, [&](int i) { if (i & 1) k_stage_copy<false><<<wgs, 512>>>(a, b, s, (int)(n4 / wgs)); else k_stage_copy<false><<<wgs, 512>>>(b, a, s, (int)(n4 / wgs)); });
    snprintf(nm, 96, "the same + 64 double atomics per workgroup (16 replicas), %d", wgs);
    timeit(nm, [&](int i) { if (i & 1) k_stage_copy<true><<<wgs, 512>>>(a, b, s, (int)(n4 / wgs)); else k_stage_copy<true><<<wgs, 512>>>(b, a, s, (int)(n4 / wgs)); });
  }
  return 0;
}
