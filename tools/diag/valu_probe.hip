// VALU issue-cost probe (gfx950): cycles per wave-instruction of v_fma_f32, v_pk_fma_f32, v_exp_f32, v_mul_f32 and of the
// attention inner-loop mixes, at 1 / 2 / 4 waves per SIMD (workgroups of 256 / 512 / 1024 threads, one per CU).
//   hipcc --offload-arch=gfx950 -O3 -o valu_probe tools/diag/valu_probe.hip && ./valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP 64     // instructions of the measured kind per loop iteration
#define ITERS 256

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ void k_probe(float* out, long long* cyc, float seed, int iters) {
  float a[16];
  f32x2 p[8];
  for (int i = 0; i < 16; ++i) a[i] = seed + (float)threadIdx.x * 1e-3f + (float)i;
  for (int i = 0; i < 8; ++i) p[i] = f32x2{a[2 * i], a[2 * i + 1]};
  const float m = 1.0000001f, c = 1e-7f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {          // 64 independent-ish v_fma_f32 (16 chains)
#pragma unroll
      for (int r = 0; r < REP / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
    } else if (KIND == 1) {   // 64 v_pk_fma_f32 (8 chains)
#pragma unroll
      for (int r = 0; r < REP / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
    } else if (KIND == 2) {   // 64 v_exp_f32
#pragma unroll
      for (int r = 0; r < REP / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
    } else if (KIND == 3) {   // 64 v_pk_add_f32
#pragma unroll
      for (int r = 0; r < REP / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
    } else if (KIND == 4) {   // forward-tile mix, packed: 4 exp + 2 pk_add + 8 pk_fma (+1 mfma), x4
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 + (i & 3)]) : "v"(m2), "v"(c2));
      }
    } else if (KIND == 5) {   // forward-tile mix, scalar: 4 exp + 4 add + 16 fma (+1 mfma), x4
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[8 + (i & 1)]) : "v"(c));
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[10 + (i & 3)]) : "v"(m), "v"(c));
      }
    } else if (KIND == 6) {   // 64 v_mul_f32
#pragma unroll
      for (int r = 0; r < REP / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
    } else if (KIND == 7) {   // 64 v_pk_mul_f32
#pragma unroll
      for (int r = 0; r < REP / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
    } else if (KIND == 9 || KIND == 10 || KIND == 11) {   // forward-tile VALU mix beside bf16 MFMAs: 9 = none, 10 = one 16x16x32 bf16, 11 = two
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)a[i];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (KIND >= 10) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
        if (KIND >= 11) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i]));
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 + (i & 3)]) : "v"(m2), "v"(c2));
      }
    } else if (KIND == 12) {   // 16 bare bf16 mfma 16x16x32
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      bf16x8 ab; for (int i = 0; i < 8; ++i) ab[i] = (__bf16)a[i];
      f32x4 ac[4] = {acc, acc, acc, acc};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, ac[i], 0, 0, 0);
      acc = ac[0] + ac[1] + ac[2] + ac[3];
    } else if (KIND == 13) {   // fma with an SGPR operand
      const float sm = __builtin_amdgcn_readfirstlane(m);
#pragma unroll
      for (int r = 0; r < REP / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sm), "v"(c));
    } else if (KIND == 14) {   // pk_fma with an SGPR-pair operand
      f32x2 sm2;
      sm2[0] = __builtin_amdgcn_readfirstlane(m); sm2[1] = __builtin_amdgcn_readfirstlane(m * 1.0000001f);
#pragma unroll
      for (int r = 0; r < REP / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(sm2), "v"(c2));
    } else if (KIND == 17 || KIND == 18) {   // pk_fma with 16 DIFFERENT SGPR pairs (17), or the same operands as VGPR pairs (18)
      f32x2 sp[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { sp[i][0] = __builtin_amdgcn_readfirstlane(m + i * 1e-8f); sp[i][1] = __builtin_amdgcn_readfirstlane(m - i * 1e-8f); }
#pragma unroll
      for (int r = 0; r < REP / 16; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (KIND == 17) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "s"(sp[i]), "v"(c2));
          else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(sp[i]), "v"(c2));
        }
    } else if (KIND == 19) {   // the forward loop body of k_attn_fwd_v: 16 pk_fma (SGPR pairs) + 8 add + 4 exp, registers only
      f32x2 sp[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { sp[i][0] = __builtin_amdgcn_readfirstlane(m + i * 1e-8f); sp[i][1] = __builtin_amdgcn_readfirstlane(m - i * 1e-8f); }
      f32x2 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t[j]) : "v"(p[0]), "s"(sp[2 * j]), "v"(c2));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t[j]) : "v"(p[1]), "s"(sp[2 * j + 1]));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float sv;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(sv) : "v"(t[j][0]), "v"(t[j][1]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(sv));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(sv));
        f32x2 pv = {sv, sv};
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2]) : "v"(pv), "s"(sp[8 + 2 * j]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[3]) : "v"(pv), "s"(sp[9 + 2 * j]));
      }
    } else if (KIND == 15) {   // pk_fma, VGPR pair broadcast through op_sel (low half of src0 for both lanes)
#pragma unroll
      for (int r = 0; r < REP / 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(m2), "v"(c2));
    } else if (KIND == 16) {   // roles: waves 0-3 of the workgroup issue fp32 MFMAs only, waves 4-7 (same SIMDs) the VALU mix only
      if (((threadIdx.x >> 8) & 1) == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc, 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i]));
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 + (i & 3)]) : "v"(m2), "v"(c2));
        }
      }
    } else if (KIND == 20) {   // 16 bf16 MFMAs 16x16x32 forced by inline asm (4 independent accumulators, operands in registers)
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      s16x8 ab;
      for (int i = 0; i < 8; ++i) ab[i] = (short)(__float_as_uint(a[i]) >> 16);
      f32x4 ac[4] = {acc, acc, acc, acc};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, %0" : "+v"(ac[i]) : "v"(ab));
      acc = ac[0] + ac[1] + ac[2] + ac[3];
    } else if (KIND == 21 || KIND == 22) {   // 16 bare f16 MFMAs: 21 = 16x16x16 (K = 16: two operand registers), 22 = 16x16x32
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      h4 a4; h8 a8;
      for (int i = 0; i < 4; ++i) a4[i] = (_Float16)a[i];
      for (int i = 0; i < 8; ++i) a8[i] = (_Float16)a[i];
      f32x4 ac[4] = {acc, acc, acc, acc};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (KIND == 21) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(ac[i]) : "v"(a4));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(ac[i]) : "v"(a8));
        }
      acc = ac[0] + ac[1] + ac[2] + ac[3];
    } else if (KIND >= 23 && KIND <= 26) {   // backward sweep-A tile (2 MFMAs, 4 exp, 4 mul, 8 pk_fma): 23 = fp32 MFMAs, 24 = 16x16x16 f16, 25 = 16x16x32 f16, 26 = none
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      typedef _Float16 h8 __attribute__((ext_vector_type(8)));
      h4 a4; h8 a8;
      for (int i = 0; i < 4; ++i) a4[i] = (_Float16)a[i];
      for (int i = 0; i < 8; ++i) a8[i] = (_Float16)a[i];
      f32x4 ac2 = acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (KIND == 23) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], a[1], acc, 0, 0, 0); ac2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], a[3], ac2, 0, 0, 0); }
        if (KIND == 24) { asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(a4)); asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %1, %0" : "+v"(ac2) : "v"(a4)); }
        if (KIND == 25) { asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(acc) : "v"(a8)); asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(ac2) : "v"(a8)); }
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[4 + i]));
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[8 + i]) : "v"(m));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[2 + (i & 3)]) : "v"(m2), "v"(c2));
      }
      acc += ac2;
    } else if (KIND == 8) {   // 16 bare mfma 16x16x4 f32 (4 accumulators)
      f32x4 ac[4] = {acc, acc, acc, acc};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) ac[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[i + 4], ac[i], 0, 0, 0);
      acc = ac[0] + ac[1] + ac[2] + ac[3];
    }
  }
  __syncthreads();   // every wave done: the oldest wave of a SIMD has issue priority and alone would show its solo speed
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += a[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + acc[0] + acc[1] + acc[2] + acc[3];
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[256 + blockIdx.x] = r1 - r0; }
}

template <int KIND>
static void run(const char* name, int per_iter) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 512 * sizeof(long long));
  printf("%-44s", name);
  for (int threads : {64, 256, 512, 1024}) {
    k_probe<KIND><<<256, threads>>>(out, cyc, 1.0f, ITERS);
    k_probe<KIND><<<256, threads>>>(out, cyc, 1.0f, ITERS);
    hipDeviceSynchronize();
    std::vector<long long> h(512);
    hipMemcpy(h.data(), cyc, 512 * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    const double per_wave_instr = s / 256 / ITERS / per_iter;           // wall cycles per instruction of ONE wave
    const int wps = threads >= 256 ? threads / 256 : 1;                  // waves per SIMD
    printf("  %4d thr: %6.2f cyc/instr/wave -> %5.2f per SIMD slot", threads, per_wave_instr, per_wave_instr / wps);
    if (threads == 1024) {   // sustained clock: the same kernel for ~20 ms (s_memtime = shader cycles, s_memrealtime = 100 MHz)
      k_probe<KIND><<<256, threads>>>(out, cyc, 1.0f, ITERS * 400);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), cyc, 512 * sizeof(long long), hipMemcpyDeviceToHost);
      double c = 0, r = 0; for (int i = 0; i < 256; ++i) { c += (double)h[i]; r += (double)h[256 + i]; }
      printf("  | sustained: %.0f us, clock %.2f GHz, %.2f cyc/slot", r / 256 / 100.0, c / r * 0.1, c / 256 / (ITERS * 400) / per_iter / wps);
    }
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}

int main() {
  // s_memtime ticks at a fixed 100 MHz on some parts: calibrate against a known loop if the numbers look 24x too small
  run<0>("v_fma_f32", REP);
  run<1>("v_pk_fma_f32", REP);
  run<2>("v_exp_f32", REP);
  run<3>("v_pk_add_f32", REP);
  run<6>("v_mul_f32", REP);
  run<7>("v_pk_mul_f32", REP);
  run<8>("v_mfma_f32_16x16x4_f32 (per mfma)", 16);
  run<12>("v_mfma_f32_16x16x32_bf16 (per mfma)", 16);
  run<20>("v_mfma_f32_16x16x32_bf16, asm-forced (per mfma)", 16);
  run<21>("v_mfma_f32_16x16x16_f16 (per mfma)", 16);
  run<22>("v_mfma_f32_16x16x32_f16 (per mfma)", 16);
  run<26>("bwd tile A, no mfma (4exp 4mul 8pkfma)", 4);
  run<23>("bwd tile A + 2 fp32 mfma 16x16x4", 4);
  run<24>("bwd tile A + 2 f16 mfma 16x16x16", 4);
  run<25>("bwd tile A + 2 f16 mfma 16x16x32", 4);
  if (getenv("PROBE_SHORT")) return 0;
  run<13>("v_fma_f32 with an SGPR operand", REP);
  run<14>("v_pk_fma_f32 with an SGPR-pair operand", REP);
  run<15>("v_pk_fma_f32 op_sel broadcast", REP);
  run<17>("v_pk_fma_f32, 16 different SGPR pairs", REP);
  run<18>("v_pk_fma_f32, the same as VGPR pairs", REP);
  run<19>("k_attn_fwd_v loop body (per 4 keys)", 1);
  run<16>("roles: 4 mfma f32 (waves 0-3) | 4 VALU tiles (4-7)", 4);
  run<9>("fwd tile packed, no mfma", 4);
  run<10>("fwd tile packed + 1 bf16 mfma", 4);
  run<11>("fwd tile packed + 2 bf16 mfma", 4);
  run<4>("fwd tile packed (per tile: 4exp 2pkadd 8pkfma 1mfma)", 4);
  run<5>("fwd tile scalar (per tile: 4exp 4add 16fma 1mfma)", 4);
  return 0;
}
