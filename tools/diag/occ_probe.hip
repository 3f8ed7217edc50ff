// Residency probe (gfx950): what hipOccupancyMaxActiveBlocksPerMultiprocessor answers for a 256-thread kernel at several
// dynamic-LDS sizes, and what the hardware does (census: how many workgroups of a spinning kernel are resident at once).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(int* counter, int* maxseen, int hold) {
  extern __shared__ float4 sm[];
  if (threadIdx.x == 0) {
    sm[0].x = 1.f;
    const int now = atomicAdd(counter, 1) + 1;
    atomicMax(maxseen, now);
    for (int i = 0; i < hold; ++i) __builtin_amdgcn_s_sleep(100);
    atomicAdd(counter, -1);
  }
  __syncthreads();
}
int main() {
  int *c, *mx;
  hipMalloc(&c, 4); hipMalloc(&mx, 4);
  for (int kb : {8, 16, 24, 32, 40, 48, 53, 56, 64, 72, 80, 96, 128, 160}) {
    const size_t lds = (size_t)kb * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int per = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k, 256, lds);
    hipMemset(c, 0, 4); hipMemset(mx, 0, 4);
    k<<<256 * 10, 256, lds>>>(c, mx, 200);
    hipDeviceSynchronize();
    int h = 0; hipMemcpy(&h, mx, 4, hipMemcpyDeviceToHost);
    printf("LDS %3d KB: API %d per CU; census max resident %d (= %.2f per CU)\n", kb, per, h, h / 256.0);
  }
  return 0;
}
