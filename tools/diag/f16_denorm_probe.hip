// Does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs?  A = a (every element), B = 1: D = 32 a exactly.
// hipcc --offload-arch=gfx950 -O2 -o f16_denorm_probe f16_denorm_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* av, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    const _Float16 a = (_Float16)av[i];
    f16x8 A, B;
    for (int e = 0; e < 8; ++e) { A[e] = a; B[e] = (_Float16)1.0f; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[2 * i] = (float)a; out[2 * i + 1] = c[0]; }
  }
}
int main() {
  const int n = 6;
  float h[n] = {1.0f, 6.2e-5f, 3.0e-5f, 9.5367431640625e-7f /* 2^-20 */, 5.9604644775390625e-8f /* 2^-24 */, 1e-9f};
  float *d, *o, r[2 * n];
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d, o, n);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("a = %.9g (fp16: %.9g)  mfma sum = %.9g  expected %.9g\n", h[i], r[2 * i], r[2 * i + 1], 32.0 * r[2 * i]);
  return 0;
}
