// LDS cycles per wave-instruction for the access patterns of ral_attnm.hip (gfx950), 16 waves per CU issuing the same
// instruction back to back (s_waitcnt every 8): the CU-level throughput cost, in cycles per instruction.
//   0 ds_write_b64   lane (r, g) -> row r (40-byte rows), chunk g          (one dS piece tile)
//   1 ds_write2_b64  the two piece tiles in one instruction
//   2 ds_read_b64_tr_b16 of a piece tile (rows 4g+tq, chunk tp)
//   3 ds_read_b64_tr_b16 of plane images (tokens 4g+tq, 16-byte tokens, chunks from two images)
//   4 ds_read_b64   plane operand of a tile (token r, half g&1)
//   5 ds_read_b128  (l4 / d4: 4 floats at 4g)
//   6 ds_add_f32    64 consecutive floats
//   7 ds_write_b64  conflict-free reference (lane * 8)
//   8 ds_read_b64   conflict-free reference
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(long long* cyc, float* sink, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4, tq = r >> 2, tp = r & 3;
  float* my = lds + wave * 2048;     // 8 KB per wave
  for (int i = lane; i < 2048; i += 64) my[i] = 1.0f;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)my;
  unsigned a;
  if (KIND == 0 || KIND == 1) a = base + r * 40 + g * 8;
  if (KIND == 2) a = base + (4 * g + tq) * 40 + tp * 8;
  if (KIND == 3) a = base + (tp < 2 ? 0 : 4096) + (4 * g + tq) * 16 + (tp & 1) * 8;
  if (KIND == 4) a = base + r * 16 + (g & 1) * 8;
  if (KIND == 5) a = base + g * 16;
  if (KIND == 6 || (KIND >= 20 && KIND <= 25) || KIND == 9 || KIND == 12 || KIND == 13 || KIND == 14 || KIND == 15) a = base + lane * 4;
  if (KIND == 10 || KIND == 11) a = base + lane * 8;
  if (KIND == 7 || KIND == 8) a = base + lane * 8;
  float v0 = 1.f, v1 = 2.f, acc = 0.f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define R8(S) S S S S S S S S
    if (KIND == 0 || KIND == 7) asm volatile(R8("ds_write_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(*(double*)&v0) : "memory");
    if (KIND == 1) asm volatile(R8("ds_write2_b64 %0, %1, %1 offset1:80\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(*(double*)&v0) : "memory");
    if (KIND == 2 || KIND == 3) { double d; asm volatile(R8("ds_read_b64_tr_b16 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory"); acc += (float)d; }
    if (KIND == 4 || KIND == 8) { double d; asm volatile(R8("ds_read_b64 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory"); acc += (float)d; }
    if (KIND == 5) { float4 d; asm volatile(R8("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a) : "memory"); acc += d.x; }
    if (KIND == 6) asm volatile(R8("ds_add_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory");
    if (KIND >= 20 && KIND <= 23) { const int nl = KIND == 20 ? 1 : KIND == 21 ? 4 : KIND == 22 ? 16 : 32;
      if (lane < nl) asm volatile(R8("ds_add_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory"); }
    if (KIND == 24) { if ((lane & 15) == 0) asm volatile(R8("ds_add_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory"); }
    if (KIND == 25) asm volatile(R8("ds_add_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(base + (lane & 3) * 4), "v"(v1) : "memory");
    if (KIND == 9) asm volatile(R8("ds_add_u32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory");
    if (KIND == 10) asm volatile(R8("ds_add_u64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(*(double*)&v0) : "memory");
    if (KIND == 11) asm volatile(R8("ds_add_f64 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(*(double*)&v0) : "memory");
    if (KIND == 12) asm volatile(R8("ds_pk_add_f16 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory");
    if (KIND == 13) asm volatile(R8("ds_max_f32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory");
    if (KIND == 14) asm volatile(R8("ds_write_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v1) : "memory");
    if (KIND == 15) { float d; asm volatile(R8("ds_add_rtn_f32 %0, %1, %2\n") "s_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(a), "v"(v1) : "memory"); acc += d; }
  }
  __syncthreads();
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + my[lane];
}
template <int KIND> static void run(const char* name) {
  long long* cyc; float* sink; hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 256 * 1024 * 4);
  const int iters = 200;
  for (int threads : {256, 1024}) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    k<KIND><<<256, threads, 131072>>>(cyc, sink, iters); k<KIND><<<256, threads, 131072>>>(cyc, sink, iters);
    hipDeviceSynchronize();
    long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-44s %4d threads: %6.2f cycles per wave-instruction (CU level)\n", name, threads, s / 256 / iters / 8 / (threads / 64));
  }
  hipFree(cyc); hipFree(sink);
}
int main() {
  run<7>("ds_write_b64 conflict-free (lane * 8)");
  run<0>("ds_write_b64 piece tile (row r, chunk g)");
  run<1>("ds_write2_b64 two piece tiles");
  run<8>("ds_read_b64 conflict-free");
  run<2>("ds_read_b64_tr_b16 piece tile");
  run<3>("ds_read_b64_tr_b16 plane images");
  run<4>("ds_read_b64 plane operand");
  run<5>("ds_read_b128 row constants");
  run<6>("ds_add_f32 consecutive");
  run<15>("ds_add_rtn_f32");
  run<20>("ds_add_f32, 1 active lane");
  run<21>("ds_add_f32, 4 active lanes");
  run<22>("ds_add_f32, 16 active lanes");
  run<23>("ds_add_f32, 32 active lanes");
  run<24>("ds_add_f32, lanes 0,16,32,48");
  run<25>("ds_add_f32, 64 lanes on 4 addresses");
  run<9>("ds_add_u32 consecutive");
  run<10>("ds_add_u64 consecutive");
  run<11>("ds_add_f64 consecutive");
  run<12>("ds_pk_add_f16");
  run<13>("ds_max_f32");
  run<14>("ds_write_b32 consecutive");
  return 0;
}
