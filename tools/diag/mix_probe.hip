// v_fma_mix / v_cvt_pk_f16_f32 semantics probe (gfx950): the fp16-pair helpers of ral_attnm.hip against a host evaluation.
//   hipcc --offload-arch=gfx950 -O3 -o mix_probe tools/diag/mix_probe.hip && ./mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
__global__ void k(const float* x, const float* y, unsigned* out) {
  const int i = threadIdx.x;
  const float x0 = x[2 * i], x1 = x[2 * i + 1], y0 = y[2 * i], y1 = y[2 * i + 1];
  unsigned h1, h2, a1, a2, b1, b2;
  asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h1) : "v"(x0), "v"(y0));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h1) : "v"(x1), "v"(y1));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x0), "v"(y0), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x1), "v"(y1), "v"(h1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a1) : "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(a2) : "v"(x0), "v"(a1));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(a2) : "v"(x1), "v"(a1));
  const float one = 1.0f;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b1) : "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, %3, -%2 op_sel_hi:[0,0,1]" : "=v"(b2) : "v"(x0), "v"(b1), "v"(one));
  asm("v_fma_mixhi_f16 %0, %1, %3, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(b2) : "v"(x1), "v"(b1), "v"(one));
  out[6 * i] = h1; out[6 * i + 1] = h2; out[6 * i + 2] = a1; out[6 * i + 3] = a2; out[6 * i + 4] = b1; out[6 * i + 5] = b2;
}
static float h2f(unsigned short h) { _Float16 v; memcpy(&v, &h, 2); return (float)v; }
int main() {
  float hx[128], hy[128]; unsigned ho[384];
  for (int i = 0; i < 128; ++i) { hx[i] = ldexpf(1.0f + 0.37f * (i % 7) + 1e-3f * i, (i % 20) - 12); hy[i] = 3.1f - 0.05f * i; }
  float *dx, *dy; unsigned* d_o;
  hipMalloc(&dx, 512); hipMalloc(&dy, 512); hipMalloc(&d_o, 1536);
  hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice); hipMemcpy(dy, hy, 512, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dy, d_o); hipMemcpy(ho, d_o, 1536, hipMemcpyDeviceToHost);
  double eprod = 0, elit = 0, ereg = 0;
  for (int i = 0; i < 64; ++i)
    for (int e = 0; e < 2; ++e) {
      const double x = hx[2 * i + e], y = hy[2 * i + e];
      auto half = [&](unsigned w) { return (double)h2f((unsigned short)(e ? w >> 16 : w & 0xffff)); };
      const double p = half(ho[6 * i]) + half(ho[6 * i + 1]);
      const double a = half(ho[6 * i + 2]) + half(ho[6 * i + 3]);
      const double b = half(ho[6 * i + 4]) + half(ho[6 * i + 5]);
      eprod = fmax(eprod, fabs(p - x * y) / fabs(x * y)); elit = fmax(elit, fabs(a - x) / fabs(x)); ereg = fmax(ereg, fabs(b - x) / fabs(x));
      if (i < 3) printf("x=%g y=%g: prod pair %g (want %g)  pair(x) literal-1.0 %g  register-1.0 %g\n", x, y, p, x * y, a, b);
    }
  printf("max rel err: pair_prod %.3g, pair_of with literal 1.0 %.3g, with a register 1.0 %.3g\n", eprod, elit, ereg);
  return 0;
}
