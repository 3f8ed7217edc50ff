// Probe of v_mfma_f32_4x4x1_16B_f32 operand / result lane maps and issue rate on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + l, b = 1000.0f * (1.0f + l);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void rate(float* out, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.0f * iters);
}
__global__ void rate16(float* out, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
  }
  long long t1 = clock64();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (4.0f * iters);
}
template <int NACC>
__global__ void rateN(float* out, int iters) {
  f32x4 c[NACC];
  for (int i = 0; i < NACC; ++i) c[i] = f32x4{0, 0, 0, 0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) c[k] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[k], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0; for (int k = 0; k < NACC; ++k) s += c[k][k & 3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / ((float)NACC * iters);
}
// mixed: 1 x 16x16x4 followed by 4 x 4x4x1 on one accumulator (the attention inner pattern), 2 query tiles
__global__ void mixed(float* out, int iters) {
  f32x4 s0, s1, o0 = {0,0,0,0}, o1 = {0,0,0,0};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile("" : "+v"(a), "+v"(b));
    s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, f32x4{0,0,0,0}, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, f32x4{0,0,0,0}, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(s0[j], b, o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(s1[j], b, o1, 0, 0, 0);
    }
  }
  __syncthreads();
  long long t1 = clock64();
  out[threadIdx.x] = o0[0] + o1[1];
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / iters;
}
// sweep-B pattern of attention backward: per key-tile pair 4 x 16x16x4 (S, dP for 2 tiles) + 16 x 4x4x1 (dV, dK)
template <int N4>
__global__ void mixedB(float* out, int iters) {
  f32x4 o0 = {0,0,0,0}, o1 = o0, o2 = o0, o3 = o0;
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    asm volatile("" : "+v"(a), "+v"(b));
    f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, f32x4{0,0,0,0}, 0, 0, 0);
    f32x4 d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, f32x4{0,0,0,0}, 0, 0, 0);
    f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, f32x4{0,0,0,0}, 0, 0, 0);
    f32x4 d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, f32x4{0,0,0,0}, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0 = __builtin_amdgcn_mfma_f32_4x4x1f32(s0[j], b, o0, 0, 0, 0);
      if (N4 >= 2) o1 = __builtin_amdgcn_mfma_f32_4x4x1f32(d0[j], b, o1, 0, 0, 0);
      if (N4 >= 3) o2 = __builtin_amdgcn_mfma_f32_4x4x1f32(s1[j], b, o2, 0, 0, 0);
      if (N4 >= 4) o3 = __builtin_amdgcn_mfma_f32_4x4x1f32(d1[j], b, o3, 0, 0, 0);
    }
    if (N4 < 4) o3 += d1; if (N4 < 3) o2 += s1; if (N4 < 2) o1 += d0;
  }
  __syncthreads();
  long long t1 = clock64();
  out[threadIdx.x] = o0[0] + o1[1] + o2[2] + o3[3];
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / iters;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  float h[256 + 8];
  probe<<<1, 64>>>(d);
  hipMemcpy(h, d, 256 * 4, hipMemcpyDeviceToHost);
  for (int l = 0; l < 12; ++l) { printf("lane %2d:", l); for (int r = 0; r < 4; ++r) printf(" %9.0f", h[l * 4 + r]); printf("\n"); }
  printf("lane 17: %9.0f %9.0f %9.0f %9.0f\n", h[68], h[69], h[70], h[71]);
  rate<<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
  printf("4x4x1_16B: %.2f clock64 ticks per instruction (one wave)\n", h[64]);
  rate16<<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost);
  printf("16x16x4 : %.2f clock64 ticks per instruction (one wave)\n", h[64]);
  rateN<1><<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("4x4x1 x1 acc (dependent): %.2f\n", h[64]);
  rateN<2><<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("4x4x1 x2 acc: %.2f\n", h[64]);
  rateN<8><<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("4x4x1 x8 acc: %.2f\n", h[64]);
  rateN<16><<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("4x4x1 x16 acc: %.2f\n", h[64]);
  mixed<<<1, 64>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("mixed tile-pair (2x16x16x4 + 8x4x4x1): %.2f ticks\n", h[64]);
  mixed<<<1, 256>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("mixed, 4 waves (1/SIMD): %.2f ticks\n", h[64]);
  mixed<<<1, 512>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("mixed, 8 waves (2/SIMD): %.2f ticks per wave-iteration\n", h[64]);
  for (int nw = 1; nw <= 4; nw *= 2) {
    mixed<<<1, 256 * nw>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("fwd pattern, %d waves/SIMD: %.1f ticks per wave-iter -> %.1f cycles per tile per SIMD\n", nw, h[64], h[64] / nw / 2);
    mixedB<4><<<1, 256 * nw>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("bwd-B pattern (16 4x4x1), %d waves/SIMD: %.1f -> %.1f per tile\n", nw, h[64], h[64] / nw / 2);
    mixedB<2><<<1, 256 * nw>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("bwd pattern (8 4x4x1), %d waves/SIMD: %.1f -> %.1f per tile\n", nw, h[64], h[64] / nw / 2);
    mixedB<1><<<1, 256 * nw>>>(d, 10000); hipMemcpy(h, d, 65 * 4, hipMemcpyDeviceToHost); printf("bwd pattern (4 4x4x1), %d waves/SIMD: %.1f -> %.1f per tile\n", nw, h[64], h[64] / nw / 2);
  }
  return 0;
}
