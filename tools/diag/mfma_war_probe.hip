// Does a load that overwrites the A operand registers of a matrix instruction RIGHT AFTER that instruction can change its
// result?  (ral_attnm.hip, round 5: dV of the last key tile was wrong by 2-16 % until the A operand's registers were kept
// alive; the compiler had re-used them as the destination of the next ds_read_b64_tr_b16.)
//   variants: NPRE matrix instructions in front (independent accumulators, to fill the matrix pipe), then the victim
//   (DEP: into the accumulator of the previous one), then KIND of overwrite: 0 none, 1 ds_read_b64_tr_b16, 2 ds_read_b64, 3 v_mov
//   hipcc --offload-arch=gfx950 -O3 -o mfma_war_probe tools/diag/mfma_war_probe.hip && ./mfma_war_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int NPRE, bool DEP, int KIND>
__global__ void k(const unsigned* in, float* out) {
  __shared__ unsigned lds[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0x3c003c00u + 0x00010001u * (i & 7);   // fp16 values near 1
  __syncthreads();
  u32x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = in[lane * 8 + j]; b[j] = in[lane * 8 + 4 + j]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f}, pre[4] = {c, c, c, c};
  u32x2 alo = {a[0], a[1]};
  u32x2 ahi = {a[2], a[3]};
  const unsigned addr = (unsigned)(size_t)(lds) + lane * 8;
  // everything in one asm block: no compiler scheduling, no hazard recogniser
#define PRE(i) "v_mfma_f32_16x16x32_f16 %" #i ", %[b], %[b], %" #i "\n"
  if (NPRE >= 1) asm volatile(PRE(0) : "+v"(pre[0]) : [b] "v"(b));
  if (NPRE >= 2) asm volatile(PRE(0) : "+v"(pre[1]) : [b] "v"(b));
  if (NPRE >= 3) asm volatile(PRE(0) : "+v"(pre[2]) : [b] "v"(b));
  if (NPRE >= 4) asm volatile(PRE(0) : "+v"(pre[3]) : [b] "v"(b));
  u32x2 aa = alo; const u32x2 bb = {b[0], b[1]};   // the victim is the K = 16 form: its A operand is one 64-bit pair, the size of the overwriting read
  if (DEP) {
    // two matrix instructions into the same accumulator, then the overwrite of the A registers (low half)
    if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n ds_read_b64_tr_b16 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(aa) : "v"(bb), "v"(addr) : "memory");
    if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n ds_read_b64 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(aa) : "v"(bb), "v"(addr) : "memory");
    if (KIND == 3) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_pk_mov_b32 %1, 0, 0" : "+v"(c), "+v"(aa) : "v"(bb) : "memory");
    if (KIND == 0) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n" : "+v"(c), "+v"(aa) : "v"(bb) : "memory");
  } else {
    if (KIND == 1) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n ds_read_b64_tr_b16 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(aa) : "v"(bb), "v"(addr) : "memory");
    if (KIND == 2) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n ds_read_b64 %1, %3\n s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(aa) : "v"(bb), "v"(addr) : "memory");
    if (KIND == 3) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n v_pk_mov_b32 %1, 0, 0" : "+v"(c), "+v"(aa) : "v"(bb) : "memory");
    if (KIND == 0) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0\n" : "+v"(c), "+v"(aa) : "v"(bb) : "memory");
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += pre[i][0] * 0.f;
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = c[j] + s;
}

static unsigned hin[512];
static float ref[256], got[256];
template <int NPRE, bool DEP, int KIND>
static void run(const unsigned* din, float* dout, const char* name) {
  hipMemset(dout, 0, 1024);
  k<NPRE, DEP, KIND><<<1, 64>>>(din, dout);
  hipMemcpy(got, dout, 1024, hipMemcpyDeviceToHost);
  if (KIND == 0) { for (int i = 0; i < 256; ++i) ref[i] = got[i]; printf("%-52s reference\n", name); return; }
  int bad = 0; double worst = 0;
  for (int i = 0; i < 256; ++i) { const double d = fabs((double)got[i] - ref[i]); if (d > 1e-6 * fabs(ref[i])) ++bad; if (d > worst) worst = d; }
  printf("%-52s %3d of 256 results differ (largest difference %.3g, |ref| ~ %.3g)\n", name, bad, worst, fabs((double)ref[0]));
}
int main() {
  for (int i = 0; i < 512; ++i) { const unsigned short h = 0x3c00 + (i * 37) % 512; hin[i] = h | ((unsigned)(0x3c00 + (i * 91) % 512) << 16); }
  unsigned* din; float* dout;
  hipMalloc(&din, 2048); hipMalloc(&dout, 1024);
  hipMemcpy(din, hin, 2048, hipMemcpyHostToDevice);
#define ROW(NPRE, DEP) \
  run<NPRE, DEP, 0>(din, dout, "NPRE=" #NPRE " DEP=" #DEP " no overwrite"); \
  run<NPRE, DEP, 1>(din, dout, "NPRE=" #NPRE " DEP=" #DEP " ds_read_b64_tr_b16 into A"); \
  run<NPRE, DEP, 2>(din, dout, "NPRE=" #NPRE " DEP=" #DEP " ds_read_b64 into A"); \
  run<NPRE, DEP, 3>(din, dout, "NPRE=" #NPRE " DEP=" #DEP " v_mov into A");
  ROW(0, false) ROW(0, true) ROW(2, false) ROW(2, true) ROW(4, false) ROW(4, true)
  return 0;
}
