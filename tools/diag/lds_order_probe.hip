// LDS ordering probe for ds_read_b64_tr_b16 (gfx950): (1) does a transposing read issued right after a ds_write_b64 of the
// same wave see the written data?  (2) do several transposing reads (mixed with plain reads) complete IN ORDER, i.e. is a
// counted s_waitcnt lgkmcnt(N) enough for the oldest of them?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out, int iters) {
  __shared__ unsigned lds[16 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned* my = lds + wave * 1024;                       // 4 KB per wave
  for (int i = lane; i < 1024; i += 64) my[i] = 0xdead0000u + i;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)my;
  const int r = lane & 15, g = lane >> 4, tq = r >> 2, tp = r & 3;
  unsigned bad1 = 0, bad2 = 0;
  for (int it = 0; it < iters; ++it) {
    // (1) lane (r, g) writes 8 bytes at row r, chunk g of a 16 x 16 halves tile (row stride 40 bytes); the transposing read of
    // lane (i, g') then returns halves [row 4g'+0..3][column i]
    const unsigned tag = (unsigned)(it * 64 + lane) & 0x3fffu;
    const unsigned h0 = (tag << 2) | 0, h1 = (tag << 2) | 1, h2 = (tag << 2) | 2, h3 = (tag << 2) | 3;   // 16-bit values: (writer lane, iteration, slot)
    u32x2 w = {(h0 & 0xffff) | (h1 << 16), (h2 & 0xffff) | (h3 << 16)};
    const unsigned waddr = base + r * 40 + g * 8, raddr = base + (4 * g + tq) * 40 + tp * 8;
    u32x2 rd;
    asm volatile("ds_write_b64 %1, %2\n ds_read_b64_tr_b16 %0, %3\n s_waitcnt lgkmcnt(0)" : "=v"(rd) : "v"(waddr), "v"(w), "v"(raddr) : "memory");
    // expected: element q of lane (i, g') = half written by lane (row 4g'+q, chunk i>>2) slot i&3
    for (int q = 0; q < 4; ++q) {
      const unsigned got = (q & 1) ? (rd[q >> 1] >> 16) : (rd[q >> 1] & 0xffff);
      const int wl = (r >> 2) * 16 + (4 * g + q);         // writer lane: (r = 4g'+q, g = i >> 2)
      const unsigned want = ((((unsigned)(it * 64 + wl) & 0x3fffu) << 2) | (r & 3)) & 0xffff;
      if (got != want) ++bad1;
    }
    // (2) four reads in flight (tr, plain, tr, tr) from a second region that holds known constants; use the first after lgkmcnt(3)
    unsigned* c = my + 512;
    if (it == 0) { for (int i = lane; i < 256; i += 64) c[i] = 0x11110000u + i; }
    const unsigned a0 = base + 2048 + lane * 8;
    u32x2 x0, x1, x2, x3, first;
    asm volatile("ds_read_b64 %0, %5\n ds_read_b64_tr_b16 %1, %5 offset:512\n ds_read_b64_tr_b16 %2, %5\n ds_read_b64_tr_b16 %3, %5 offset:512\n"
                 "s_waitcnt lgkmcnt(3)\n v_pk_mov_b32 %4, %0, %0\n s_waitcnt lgkmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(first) : "v"(a0) : "memory");
    if (first[0] != 0x11110000u + lane * 2) ++bad2;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = bad1; out[65536 + blockIdx.x * blockDim.x + threadIdx.x] = bad2;
}
int main() {
  unsigned* d; hipMalloc(&d, 2 * 65536 * 4);
  static unsigned h[2 * 65536];
  for (int threads : {64, 1024}) {
    hipMemset(d, 0, sizeof(h));
    k<<<64, threads>>>(d, 2000); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    long b1 = 0, b2 = 0; for (int i = 0; i < 64 * threads; ++i) { b1 += h[i]; b2 += h[65536 + i]; }
    printf("%4d threads per workgroup: write -> transposing read mismatches %ld, counted-wait mismatches %ld (of %ld checks)\n", threads, b1, b2, 2000L * 64 * threads);
  }
  return 0;
}
