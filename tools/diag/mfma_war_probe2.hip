// Second WAR probe: the K = 32 matrix instruction with the LOW HALF of its A operand (or of its B operand) overwritten by
// the next instruction, as in the failing build of k_attn_bwd_m (v_mfma_f32_16x16x32_f16 v[20:23], v[36:39], v[48:51], v[20:23]
// followed by ds_read_b64_tr_b16 v[36:37]).  Fixed registers inside one asm block; 1 or 16 waves per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// KIND 0: no overwrite; 1: tr read into A lo; 2: tr read into B lo; 3: ds_read_b64 into A lo; 4: v_pk_mov into A lo
// NDEP: matrix instructions chained into the same accumulator in front of the victim
template <int KIND, int NDEP>
__global__ void k(const unsigned* in, float* out) {
  __shared__ unsigned lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0x3c003c00u + 0x00010001u * (i & 7);
  __syncthreads();
  u32x4 a, b;
  for (int j = 0; j < 4; ++j) { a[j] = in[lane * 8 + j]; b[j] = in[lane * 8 + 4 + j]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  const unsigned addr = (unsigned)(size_t)(lds) + wave * 512 + lane * 8;
#define LOADREGS "v_mov_b32 v40, %1\n v_mov_b32 v41, %2\n v_mov_b32 v42, %3\n v_mov_b32 v43, %4\n v_mov_b32 v44, %5\n v_mov_b32 v45, %6\n v_mov_b32 v46, %7\n v_mov_b32 v47, %8\n" \
                 "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n s_nop 7\n"
#define MF "v_mfma_f32_16x16x32_f16 v[48:51], v[40:43], v[44:47], v[48:51]\n"
#define TAIL "s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n v_mov_b32 %0, v48\n"
#define CHAIN (NDEP == 0 ? "" : "")
  float r0;
#define BODY(OVER) \
  if (NDEP == 0) asm volatile(LOADREGS MF OVER TAIL : "=v"(r0) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(addr) \
                              : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory"); \
  if (NDEP == 1) asm volatile(LOADREGS MF MF OVER TAIL : "=v"(r0) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(addr) \
                              : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory"); \
  if (NDEP == 3) asm volatile(LOADREGS MF MF MF MF OVER TAIL : "=v"(r0) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(addr) \
                              : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "memory");
  if (KIND == 0) { BODY("") }
  if (KIND == 1) { BODY("ds_read_b64_tr_b16 v[40:41], %9\n") }
  if (KIND == 2) { BODY("ds_read_b64_tr_b16 v[44:45], %9\n") }
  if (KIND == 3) { BODY("ds_read_b64 v[40:41], %9\n") }
  if (KIND == 4) { BODY("v_pk_mov_b32 v[40:41], 0, 0\n") }
  if (KIND == 5) { BODY("ds_read_b64_tr_b16 v[40:41], %9\n ds_read_b64_tr_b16 v[44:45], %9 offset:64\n ds_read_b64_tr_b16 v[46:47], %9 offset:128\n") }
  out[threadIdx.x] = r0;
}
static float ref[1024], got[1024];
static unsigned hin[512];
template <int KIND, int NDEP>
static void run(const unsigned* din, float* dout, int threads, const char* name) {
  int bad = 0; double worst = 0;
  for (int rep = 0; rep < 20; ++rep) {
    k<0, NDEP><<<64, threads>>>(din, dout); hipMemcpy(ref, dout, 4096, hipMemcpyDeviceToHost);
    k<KIND, NDEP><<<64, threads>>>(din, dout); hipMemcpy(got, dout, 4096, hipMemcpyDeviceToHost);
    for (int i = 0; i < threads; ++i) { const double d = fabs((double)got[i] - ref[i]); if (d > 1e-6 * fabs(ref[i])) ++bad; if (d > worst) worst = d; }
  }
  printf("%-40s %4d threads  chain %d: %5d results differ over 20 runs (largest %.3g, ref %.3g)\n", name, threads, NDEP, bad, worst, fabs((double)ref[0]));
}
int main() {
  for (int i = 0; i < 512; ++i) { const unsigned short h = 0x3c00 + (i * 37) % 512; hin[i] = h | ((unsigned)(0x3c00 + (i * 91) % 512) << 16); }
  unsigned* din; float* dout;
  hipMalloc(&din, 2048); hipMalloc(&dout, 4096);
  hipMemcpy(din, hin, 2048, hipMemcpyHostToDevice);
  for (int threads : {64, 1024}) {
#define ALL(ND) run<1, ND>(din, dout, threads, "tr read into A lo"); run<2, ND>(din, dout, threads, "tr read into B lo"); run<3, ND>(din, dout, threads, "ds_read_b64 into A lo"); \
                run<4, ND>(din, dout, threads, "v_pk_mov into A lo"); run<5, ND>(din, dout, threads, "three tr reads into A lo, B");
    ALL(0) ALL(1) ALL(3)
  }
  return 0;
}
