#include <hip/hip_runtime.h>
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float swap16_add(float v) {
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
__device__ __forceinline__ float swap32_add(float v) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
template <int W> __device__ __forceinline__ float group_sum(float v) {
  if constexpr (W >= 2) v = dpp_add<0xB1>(v);
  if constexpr (W >= 4) v = dpp_add<0x4E>(v);
  if constexpr (W >= 8) v = dpp_add<0x141>(v);
  if constexpr (W >= 16) v = dpp_add<0x140>(v);
  if constexpr (W >= 32) v = swap16_add(v);
  if constexpr (W >= 64) v = swap32_add(v);
  return v;
}
template <int W> __device__ __forceinline__ float group_sum_ref(float v) {
  for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__global__ void k(const float* in, float* out, float* ref) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = group_sum<2>(v); out[64 + threadIdx.x] = group_sum<4>(v); out[128 + threadIdx.x] = group_sum<8>(v);
  out[192 + threadIdx.x] = group_sum<16>(v); out[256 + threadIdx.x] = group_sum<32>(v); out[320 + threadIdx.x] = group_sum<64>(v);
  ref[threadIdx.x] = group_sum_ref<2>(v); ref[64 + threadIdx.x] = group_sum_ref<4>(v); ref[128 + threadIdx.x] = group_sum_ref<8>(v);
  ref[192 + threadIdx.x] = group_sum_ref<16>(v); ref[256 + threadIdx.x] = group_sum_ref<32>(v); ref[320 + threadIdx.x] = group_sum_ref<64>(v);
}
int main() {
  float h[64], *d, *o, *r, ho[384], hr[384];
  for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i % 20)) + i;   // integers: exact sums
  hipMalloc(&d, 256); hipMalloc(&o, 384 * 4); hipMalloc(&r, 384 * 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, r);
  hipMemcpy(ho, o, 384 * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 384 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 384; ++i) if (ho[i] != hr[i]) { if (bad < 10) printf("mismatch %d: %f vs %f\n", i, ho[i], hr[i]); ++bad; }
  printf("bad = %d\n", bad);
  return bad != 0;
}
