// Attention backward with EVERY contraction on the f16 matrix cores, in ONE sweep (round 5).
//
// Reference op: the backward of MSAttention.forward (model/raletransformer.py:291-322, model/transformer.py:289-323, R-wave
// bias model/transformer.py:508-558).  P is recomputed from q, k and the saved log-sum-exp, as in the two-sweep kernels of rounds 1-4
// (k_attn_bwd_w in ral_attn.hip, k_attn_bwd in ral_bwd.hip: today the strict-fp32 path), but those kernels evaluate every score tile TWICE (sweep A: dQ with the query on the lane, sweep B: dK, dV
// with the key on the lane) and contract dS / P with k, q, dO on the vector ALU: 8 exponentials, 8 multiplies and 24
// packed FMAs per tile pair - 245 issue cycles at best, 341-520 measured - beside a matrix pipe that is 16 % busy.
//
// Here a score tile is evaluated ONCE, with the key on the lane (S[query 4g+j][key r] in the registers of lane (r, g)):
//   * P and dS leave the vector ALU as fp16 PAIRS (h1 = fp16(x), h2 = fp16(x - h1): v_cvt_pk_f16_f32 and the mixed-precision
//     FMA v_fma_mixlo/hi_f16, which forms p * dp - h1 exactly before its one rounding): 14 instructions per tile instead of
//     4 multiplies + 24 packed FMAs, and the second evaluation's 4 exponentials are gone;
//   * the accumulator layout of that tile IS the B operand of a K = 32 matrix instruction whose contraction index runs over
//     [P pieces of queries 4g .. 4g+3 | dS pieces of the same queries]: dV^T and dK^T of the 16 keys come out of ONE
//     v_mfma_f32_16x16x32_f16 per piece (rows 0-7: dO^T planes against P, rows 8-15: q^T planes against dS - the A operand
//     is a transposing LDS read, ds_read_b64_tr_b16, of the SAME token-major planes the S / dP tiles read by rows, masked
//     into its half of K);
//   * only dS crosses lanes, once: its two pieces go through a 16 x 16 LDS tile of the wave ([key][query] rows of 40 bytes,
//     written as they stand, read back with the transposing read) and meet k^T in a third K = 32 instruction (dQ^T);
//   * per tile: 5 matrix instructions (~90 cycles of the matrix pipe) beside 4 exponentials + 14 conversions (~70 issue
//     cycles): the two pipes are balanced, where the two-sweep form left the matrix pipe idle.
// Range.  p is evaluated as 2^8 p (the shift sits in the C operand of the S tile, next to -lse) so that probabilities down to
// 2^-22 keep 22 bits in their pair; dO and v are multiplied by one power of two per task (largest magnitudes into [4, 8) and
// [1, 2)) so that |2^8 p (dP - delta)| < 2^15 for ANY inputs.  q log2 e and k each get THEIR OWN power of two (largest magnitude
// into [8, 16)): their planes are also the operands of the dK / dQ products, where a pair of a small number (second piece
// subnormal in fp16) would carry 14 bits - with q and k merely balanced against each other (as the two-sweep kernels do,
// which contract with fp32 quads) dq / dk of a head with |scores| ~ 1e-6 were off by 2.5e-5.  The score tile therefore comes
// out as (cq ck) (s - lse + 8) - its C operand carries the same factor - and is multiplied by 1 / (cq ck) on the way to the
// exponential (four v_mul_f32 per tile).  All factors are powers of two and leave with the results.  Non-finite inputs give
// non-finite outputs of their head.
#include "ral_device.hpp"
#include "ral_kernels.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#define RAL_LOG2E 1.4426950408889634f
#define RAL_LN2 0.6931471805599453f
#define RAL_PSH 8.0f                 // P tiles are evaluated as 2^RAL_PSH p
#define RAL_PSH_INV 0.00390625f      // 2^-RAL_PSH

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;

// fp16 pairs, two elements per register (element 0 in the low half): h1 = fp16(x), h2 = fp16(x - h1).
// h1 is formed by plain casts (the compiler emits v_cvt_pk_f16_f32 and keeps the wait states a consumer of a transcendental /
// matrix-core result needs - it does NOT see inside inline asm, and an asm statement that read such a result directly
// returned stale registers: dV was off by 2-16 % in the first version of this kernel).  The residual is one mixed-precision
// FMA per element (v_fma_mixlo/hi_f16 takes h1's half through op_sel and rounds x * y - h1 once); these asm statements
// depend on h1, so they issue behind the compiler-visible consumer of x.
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
RAL_DEV unsigned pair_hi2(float x0, float x1) { return __builtin_bit_cast(unsigned, h16x2{(_Float16)x0, (_Float16)x1}); }
RAL_DEV unsigned pair_lo2(float x0, float y0, float x1, float y1, unsigned h1) {   // fp16(x y - h1), x y exact inside the FMA
  unsigned h2;
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x0), "v"(y0), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x1), "v"(y1), "v"(h1));
  return h2;
}
RAL_DEV void pair_prod2(float x0, float y0, float x1, float y1, unsigned& h1, unsigned& h2) {
  h1 = pair_hi2(x0 * y0, x1 * y1);
  h2 = pair_lo2(x0, y0, x1, y1, h1);
}
RAL_DEV void pair_of2(float x0, float x1, unsigned& h1, unsigned& h2) {
  h1 = pair_hi2(x0, x1);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(h2) : "v"(x0), "v"(h1));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(h2) : "v"(x1), "v"(h1));
}
// the registers of fp16 pieces an asm statement produced, before a matrix instruction reads them (a vector write needs
// two wait states in front of a matrix-core read of the same register; the compiler cannot count them across inline asm)
RAL_DEV void pair_settle(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
  asm("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// the 16 bytes of a token in a plane image: [h1 of the quad | h2 of the quad] of c * x
RAL_DEV uint4 pair_quad(float4 x, float c) {
  uint4 u;
  pair_prod2(x.x, c, x.y, c, u.x, u.z);
  pair_prod2(x.z, c, x.w, c, u.y, u.w);
  return u;
}
RAL_DEV f32x4 mm16(u32x2 a, u32x2 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
}
RAL_DEV f32x4 mm32(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// transposing read: lane 4 q + p of a 16-lane group passes the address of row q, 16-bit columns 4 p .. 4 p + 3 of a 4 x 16
// block; lane i of the group receives column i of the four rows.  (EXEC must be all ones.)
RAL_DEV u32x2 tr_read(const float* a) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a)));
}
// power of two that puts a magnitude with the bits `maxbits` into [2^e, 2^(e+1)) (capped at 2^60), and its inverse
RAL_DEV float pow2_to(unsigned maxbits, int e, int cap = 60) {
  const int f = 254 + e - (int)(maxbits >> 23);
  return maxbits == 0u ? 1.0f : __uint_as_float((unsigned)(f < 127 + cap ? (f > 1 ? f : 1) : 127 + cap) << 23);
}
// the powers of two of q log2 e and k (bits of their largest magnitudes): planes in [8, 16), capped at 2^50 each so that the
// C operand of the score tile, (cq ck) (8 - lse log2 e), stays finite
// Two modes, chosen per head (returns true for the second):
//   balanced: q' = 2^-a q log2 e, k' = 2^a k (pair_balance: the product - the score - is unchanged, nothing to undo), taken
//     when both then sit in [2^-3, 2^14): their pairs carry 22 bits, also as operands of the dK / dQ products;
//   independent: each its own power of two (largest magnitude into [8, 16)) and the score tile times 1 / (cq ck) in front of
//     the exponential - four more v_mul per tile (5 % of the kernel), for heads whose scores are tiny (< 2^-6) or huge.
RAL_DEV bool qk_scales(unsigned mqbits, unsigned mkbits, float& cq, float& ck) {
  const int eq = (int)(mqbits >> 23), ek = (int)(mkbits >> 23);   // biased exponents of max |q log2 e|, max |k|
  const int e2 = eq + ek - 254;                                     // exponent of their product
  if (mqbits != 0u && mkbits != 0u && e2 >= -6 && e2 < 26) {
    pair_balance(__uint_as_float(mqbits), __uint_as_float(mkbits), cq, ck);
    return false;
  }
  cq = pow2_to(mqbits, 3, 50); ck = pow2_to(mkbits, 3, 50);
  return true;
}
RAL_DEV float pow2_inv(float p) { return __uint_as_float((254u << 23) - __float_as_uint(p)); }

RAL_DEV float f4amax(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

#ifndef RAL_ATTNM_FENCE
#define RAL_ATTNM_FENCE 1
#endif
#if RAL_ATTNM_FENCE
#define RAL_XB_FENCE() asm volatile("" ::: "memory")
#else
#define RAL_XB_FENCE() do {} while (0)
#endif
#ifndef RAL_ATTNM_WPE
#define RAL_ATTNM_WPE 4
#endif

// One wave per head (two heads at N = 32), N = 32, 64, 128: operands in a private LDS slice, no workgroup barrier in the task loop.  Loop order: query tile outside (dQ^T of the tile in one
// accumulator), key tiles inside (dV^T / dK^T of every key tile of the head in KT accumulators, statically indexed).
template <int NT, bool TAB>
__global__ __launch_bounds__(256, RAL_ATTNM_WPE) void k_attn_bwd_m(const float* qkv, const float* o_hm, const float* do_hm,
                                                                   const float* lse, const float* __restrict__ table,
                                                                   float* __restrict__ tpart, float* dqkv, int H, int Len_rt,
                                                                   int ntask) {
  constexpr int HW = NT >= 64 ? 1 : 64 / NT;   // heads per task
  constexpr int T = HW * NT, TPL = T / 64;     // tokens per task, tokens per lane
  constexpr int KT = NT / 16;                  // key (query) tiles per head
  constexpr int BUF = 160;                     // floats of one dS piece tile: 16 key rows of 40 bytes
  constexpr int PPAD = 32;                     // floats between plane images: the transposing reads take k with v (dO with q) chunks in one
                                               // instruction, and images a multiple of 64 dwords apart would put them on the same banks
  constexpr int WSZ = T * 18 + 4 * PPAD + 4 * BUF;   // floats of LDS per wave
  constexpr int SH = TAB ? 8 : 0, MSK = NT - 1;   // tiles rotated by 8 tokens with a table (see k_attn_bwd_w)
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // plane images, 16 bytes per token: [h1 x 4 | h2 x 4] of q log2 e cq, k ck, v cv, dO cd
  float* Qp = sm + wave * WSZ;
  float* Kp = Qp + 4 * T + PPAD;
  float* Vp = Kp + 4 * T + PPAD;
  float* Dp = Vp + 4 * T + PPAD;
  float* Ls = Dp + 4 * T + PPAD;   // RAL_PSH - lse log2 e
  float* Dl = Ls + T;              // -rowsum(dO O) cd cv
  float* Xb = Dl + T;              // dS piece tiles: [parity][piece][key 16][query 16 (+4 pad)] halves
  const int Len = TAB ? Len_rt : 0;
  const int ntab = TAB ? (2 * Len - 1) * H : 0;
  const int nwv = blockDim.x >> 6;
  float* tab = sm + nwv * WSZ;     // bias * log2(e), (2 Len - 1, H)
  double* dtab = reinterpret_cast<double*>(tab + ((ntab + 1) & ~1));   // table gradient in DOUBLES (ds_add_f64: 8 LDS cycles; ds_add_f32: 192)
  const int off = (NT - Len) >> 1;
  if constexpr (TAB) {
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) { tab[i] = table[i] * RAL_LOG2E; dtab[i] = 0.; }
    __syncthreads();
  }
  auto meets = [&](int x0, int w) -> bool {
    return TAB && (off < SH || (x0 < off + Len - SH && x0 + w > off - SH));
  };
  const int stride = gridDim.x * nwv;
  int task = blockIdx.x * nwv + wave;
  // per-task factors (wave-uniform): results x these; su = 1 / (cq ck) brings the score tile back to log2 units
  float sc_dq = 1.f, sc_dk = 1.f, sc_dv = 1.f, sc_tab = 1.f, su = 1.f;
  bool need_su = false;             // independent q / k scales: the score tile is multiplied by su (see qk_scales)

  auto task_ptrs = [&](int tk, const float4*& gq, const float4*& gk, const float4*& gv, const float4*& gd, const float4*& go,
                       const float*& gl) {
    const int hh = tk * HW, win = hh / H, h0 = hh - win * H;
    gq = reinterpret_cast<const float4*>(qkv) + ((size_t)win * 3 * H + h0) * NT;
    gk = gq + (size_t)H * NT;
    gv = gk + (size_t)H * NT;
    const size_t hq = ((size_t)win * H + h0) * NT;
    gd = reinterpret_cast<const float4*>(do_hm) + hq;
    go = reinterpret_cast<const float4*>(o_hm) + hq;
    gl = lse + hq;
  };
  auto put = [&](int t, float4 q, float4 k, float4 v, float4 d, float4 o, float l, float cq, float ck, float cv, float cd) {
    reinterpret_cast<uint4*>(Qp)[t] = pair_quad(q, RAL_LOG2E * cq);
    reinterpret_cast<uint4*>(Kp)[t] = pair_quad(k, ck);
    reinterpret_cast<uint4*>(Vp)[t] = pair_quad(v, cv);
    reinterpret_cast<uint4*>(Dp)[t] = pair_quad(d, cd);
    Ls[t] = fmaf(-l, RAL_LOG2E, RAL_PSH) * (cq * ck);
    Dl[t] = -f4dot(d, o) * (cd * cv);
  };
  // the task's powers of two; sets the output factors
  auto scales = [&](float mq, float mk, float mv, float md, float& cq, float& ck, float& cv, float& cd) {
    need_su = qk_scales(__float_as_uint(group_max<64>(mq) * RAL_LOG2E), __float_as_uint(group_max<64>(mk)), cq, ck);
    cv = pow2_to(__float_as_uint(group_max<64>(mv)), 0);
    cd = pow2_to(__float_as_uint(group_max<64>(md)), 2);
    const float icd = pow2_inv(cd), icv = pow2_inv(cv);
    su = pow2_inv(cq) * pow2_inv(ck);
    sc_tab = RAL_PSH_INV * icd * icv;        // dS = sc_tab dS'
    sc_dq = 0.5f * sc_tab * pow2_inv(ck);    // q = 0.5 (h Wq^T + b); the k planes carried ck
    sc_dk = RAL_LN2 * sc_tab * pow2_inv(cq); // the q planes carried log2(e) cq
    sc_dv = RAL_PSH_INV * icd;
  };
  auto stage = [&](int tk) {
    const float4 *gq, *gk, *gv, *gd, *go; const float* gl;
    task_ptrs(tk, gq, gk, gv, gd, go, gl);
    float4 q[TPL], k[TPL], v[TPL], d[TPL], o[TPL]; float l[TPL];
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
      const int t = lane + 64 * u;
      q[u] = gq[t]; k[u] = gk[t]; v[u] = gv[t]; d[u] = gd[t]; o[u] = go[t]; l[u] = gl[t];
    }
    float md = 0.f, mv = 0.f, mq = 0.f, mk = 0.f;
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
      md = fmaxf(md, f4amax(d[u])); mv = fmaxf(mv, f4amax(v[u])); mq = fmaxf(mq, f4amax(q[u])); mk = fmaxf(mk, f4amax(k[u]));
    }
    float cq, ck, cv, cd;
    scales(mq, mk, mv, md, cq, ck, cv, cd);
#pragma unroll
    for (int u = 0; u < TPL; ++u) put(lane + 64 * u, q[u], k[u], v[u], d[u], o[u], l[u], cq, ck, cv, cd);
  };
  // lane roles of the transposing reads: row tq of the block, column quad tp
  const int tq = r >> 2, tp = r & 3;
  const bool lowrows = r < 8;
  while (task < ntask) {
    // (the operands of a task are requested and deposited at the top of its trip: requesting the NEXT task's under the
    // current one's tiles - 21 registers held across the sweep - measured 116 against 110 us per launch at N = 64)
    stage(task);
    const int hh = task * HW, win = hh / H, h0 = hh - win * H;
    float* dbase = dqkv + (size_t)win * 3 * H * NT * 4;
    auto sweep = [&](auto su_on) {    // (instantiated for both scaling modes: the choice is per task, wave-uniform)
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      const int head = h0 + hl, tb = hl * NT;
      const float* Qh = Qp + 4 * tb; const float* Kh = Kp + 4 * tb; const float* Vh = Vp + 4 * tb; const float* Dh = Dp + 4 * tb;
      f32x4 acc[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int qt = 0; qt < KT; ++qt) {
        const int q0 = qt * 16;
        const int qr = (q0 + SH + r) & MSK;          // this lane's query as a row of the S tile / a column of dQ^T
        const int q4 = (q0 + SH + 4 * g) & MSK;      // first of the four queries of this lane group
        const u32x2 Aq = *reinterpret_cast<const u32x2*>(Qh + 4 * qr + 2 * (g >> 1));
        const u32x2 Ad = *reinterpret_cast<const u32x2*>(Dh + 4 * qr + 2 * (g >> 1));
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + tb + q4);
        const float4 d4 = *reinterpret_cast<const float4*>(Dl + tb + q4);
        const f32x4 cl = {l4.x, l4.y, l4.z, l4.w}, cdl = {d4.x, d4.y, d4.z, d4.w};
        // A operand of the dV^T / dK^T product: rows 0-7 = dO planes^T in the P half of K, rows 8-15 = q planes^T in the dS half
        const u32x2 xt = tr_read((tp < 2 ? Dh : Qh) + 4 * (q4 + tq) + 2 * (tp & 1));
        const u32x4 a8 = lowrows ? u32x4{xt[0], xt[1], 0u, 0u} : u32x4{0u, 0u, xt[0], xt[1]};
        const bool qin = meets(q0, 16);
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        auto dq_step = [&](int kp) {   // dQ^T += k^T planes x dS^T of key tile kp (its pieces were written a tile ago)
          const float* bp = Xb + (kp & 1) * 2 * BUF + (4 * g + tq) * 10 + 2 * tp;
          const u32x2 b1 = tr_read(bp), b2 = tr_read(bp + BUF);
          const int k4 = (kp * 16 + SH + 4 * g) & MSK;
          const u32x2 kx = tr_read((tp < 2 ? Kh : Vh) + 4 * (k4 + tq) + 2 * (tp & 1));   // (rows 8-15: v planes, results unused)
          dq = mm32(u32x4{kx[0], kx[1], kx[0], kx[1]}, u32x4{b1[0], b1[1], b2[0], b2[1]}, dq);
        };
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k0 = kt * 16;
          const int kr = (k0 + SH + r) & MSK;        // this lane's key: the column of the S tile
          // (keeping these operands of all key tiles of a short head in registers measured the same: 111 against 109 us at N = 64)
          const u32x2 Bk = *reinterpret_cast<const u32x2*>(Kh + 4 * kr + 2 * (g & 1));
          const u32x2 Bv = *reinterpret_cast<const u32x2*>(Vh + 4 * kr + 2 * (g & 1));
          f32x4 s = mm16(Aq, Bk, cl);                // (cq ck) (S - lse + 8), log2 units   [query 4g+j][key r]
          const f32x4 dp = mm16(Ad, Bv, cdl);        // (dP - delta) cd cv
          if constexpr (decltype(su_on)::value) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] *= su;
          }
          float p[4];
          unsigned ph1[2], ph2[2], sh1[2], sh2[2];
          if (TAB && qin && meets(k0, 16)) {
            const int ki = kr - off;
            const bool kok = (unsigned)ki < (unsigned)Len;
            const int rel0 = (q4 - off) - ki + Len - 1;
            int e[4]; bool in[4]; float b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              in[j] = kok && (unsigned)(q4 + j - off) < (unsigned)Len;
              e[j] = min(max(rel0 + j, 0), 2 * Len - 2) * H + head;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tab[e[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_exp2f(s[j] + (in[j] ? b[j] : 0.f));
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (in[j]) atomicAdd(dtab + e[j], (double)(p[j] * dp[j] * sc_tab));
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_exp2f(s[j]);
          }
          pair_of2(p[0], p[1], ph1[0], ph2[0]); pair_of2(p[2], p[3], ph1[1], ph2[1]);
          pair_prod2(p[0], dp[0], p[1], dp[1], sh1[0], sh2[0]); pair_prod2(p[2], dp[2], p[3], dp[3], sh1[1], sh2[1]);
          pair_settle(ph2[0], ph2[1], sh2[0], sh2[1]);
          acc[kt] = mm32(a8, u32x4{ph1[0], ph1[1], sh1[0], sh1[1]}, acc[kt]);
          acc[kt] = mm32(a8, u32x4{ph2[0], ph2[1], sh2[0], sh2[1]}, acc[kt]);
          // dS pieces of the tile, [key r][queries 4g .. 4g+3]
          float* bw = Xb + (kt & 1) * 2 * BUF + r * 10 + 2 * g;
          *reinterpret_cast<u32x2*>(bw) = u32x2{sh1[0], sh1[1]};
          *reinterpret_cast<u32x2*>(bw + BUF) = u32x2{sh2[0], sh2[1]};
          RAL_XB_FENCE();
          if (kt > 0) dq_step(kt - 1);
        }
// (fence + keep-alive below: without them hipcc interleaves the reads of this last step with the last tile and re-uses the
        // registers of a8 for them; that build returned dV of the last key tile off by 2-16 %, deterministically, although its
        // instruction stream reads correct and tools/diag/mfma_war_probe*.hip / lds_order_probe.hip find no hazard in the
        // hardware - unexplained; every shape is checked against fp64 in tests/test_gpu_attention.py)
        asm volatile("" ::: "memory");
        dq_step(KT - 1);
        // dQ[query r][d]: rows d (h1 planes of k) + rows 4 + d (h2 planes) of the accumulator = lane groups 0 and 1
        {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = swap16_add(dq[j]) * sc_dq;
          if (g == 0) *reinterpret_cast<float4*>(dbase + ((size_t)head * NT + qr) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        asm volatile("" :: "v"(a8));
      }
      // dV (rows 0-7: lane groups 0, 1) and dK (rows 8-15: lane groups 2, 3) of every key tile
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int kr = (kt * 16 + SH + r) & MSK;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swap16_add(acc[kt][j]);
        const float sc = g < 2 ? sc_dv : sc_dk;
        float* dst = dbase + (size_t)(g < 2 ? 2 : 1) * H * NT * 4 + ((size_t)head * NT + kr) * 4;
        if ((g & 1) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
      }
    }
    };
    if (need_su) sweep(std::true_type{}); else sweep(std::false_type{});
    task += stride;
  }
  if constexpr (TAB) {
    __syncthreads();
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tpart[(size_t)blockIdx.x * ntab + i] = (float)dtab[i];
  }
}

// =====================================================================================================================
// Long windows (N >= 256; any N = 16 KT W with W waves per head dividing 8): a 512-thread workgroup per (window, head group),
// staged in two passes - all operands of the item as plane images in the LDS, scales per head from LDS maxima - but with
// the one-sweep tiles above.  A wave owns KT key tiles of one head for the whole item: their K / V operands and k^T planes
// stay in registers, dV^T / dK^T in KT accumulators; it walks over ALL query tiles of the head, and the dQ^T tile of a query
// tile (summed over the wave's keys in registers) is added to an fp32 image of dQ in the LDS with ONE ds_add_f32 per lane
// (a 4 x 4 register / row transpose puts (query r, dim g) on lane (r, g)); the image leaves after a barrier, scaled.
// LDS per token: 64 bytes of planes + -lse, -delta + 32 bytes of dQ = 104; N = 512: 53 KB + 8 x 2.5 KB of dS tiles per
// workgroup, two workgroups per CU.
// =====================================================================================================================
template <int KT, bool TAB>
__global__ __launch_bounds__(512, 4) void k_attn_bwd_mh(const float* __restrict__ qkv, const float* __restrict__ o_hm,
                                                        const float* __restrict__ do_hm, const float* __restrict__ lse,
                                                        const float* __restrict__ table, float* __restrict__ tpart,
                                                        float* __restrict__ dqkv, int N, int H, int HG, int Len_rt, int B) {
  constexpr int BUF = 160;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  const int T = HG * N;
  constexpr int PPAD = 32;                                // (see k_attn_bwd_m)
  float* Qp = sm;
  float* Kp = Qp + 4 * T + PPAD;
  float* Vp = Kp + 4 * T + PPAD;
  float* Dp = Vp + 4 * T + PPAD;
  float* Ls = Dp + 4 * T + PPAD;
  float* Dl = Ls + T;
  // dQ image, (T, 4) DOUBLES in units of 1 / sc_dq: on gfx950 ds_add_f64 takes 8 LDS cycles per wave-instruction, ds_add_f32
  // 192 (three per lane; tools/diag/lds_cost_probe.hip) - with fp32 adds this image was 48 of the 74 LDS cycles of a tile
  double* dQb = reinterpret_cast<double*>(Dl + T);
  unsigned* mx = reinterpret_cast<unsigned*>(dQb + 4 * T);   // [HG][4]: bits of max |dO|, |v|, |q log2 e|, |k|
  float* Xw = reinterpret_cast<float*>(mx + 4 * HG);      // dS piece tiles, 4 BUF per wave
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  float* Xb = Xw + wave * 4 * BUF;
  const int Len = TAB ? Len_rt : 0;
  const int ntab = TAB ? (2 * Len - 1) * H : 0;
  float* tab = Xw + nw * 4 * BUF;
  double* dtab = reinterpret_cast<double*>(tab + ((ntab + 1) & ~1));
  const int off = (N - Len) >> 1;
  for (int i = threadIdx.x; i < ntab; i += blockDim.x) { tab[i] = table[i] * RAL_LOG2E; dtab[i] = 0.; }
  const int ngrp = H / HG;
  const int wph = N / (16 * KT);                          // waves per head
  const int hl = wave / wph, ks = (wave - hl * wph) * KT * 16;   // this wave's head of the group and first key
  const int tb = hl * N;
  const int tq = r >> 2, tp = r & 3;
  const bool lowrows = r < 8;
  auto meets = [&](int x0, int w) -> bool { return TAB && x0 < off + Len && x0 + w > off; };
  for (int item = blockIdx.x; item < B * ngrp; item += gridDim.x) {
    const int win = item / ngrp, h0 = (item - win * ngrp) * HG;
    const float* base = qkv + (size_t)win * 3 * H * N * 4;
    float* dbase = dqkv + (size_t)win * 3 * H * N * 4;
    const size_t hq0 = ((size_t)win * H + h0) * N;
    __syncthreads();   // every wave is done with the previous item
    if (threadIdx.x < 4 * HG) mx[threadIdx.x] = 0u;
    __syncthreads();
    // ---- staging pass 1: fp32 quads into the plane slots, -lse, -delta (unscaled), the heads' maxima
    {
      const float4* gq = reinterpret_cast<const float4*>(base + (size_t)h0 * N * 4);
      const float4* gk = reinterpret_cast<const float4*>(base + (size_t)(H + h0) * N * 4);
      const float4* gv = reinterpret_cast<const float4*>(base + (size_t)(2 * H + h0) * N * 4);
      const float4* gd = reinterpret_cast<const float4*>(do_hm) + hq0;
      const float4* go = reinterpret_cast<const float4*>(o_hm) + hq0;
      const int bd = blockDim.x;
      for (int i0 = 0; i0 < T; i0 += 2 * bd) {      // (T is a multiple of 256: whole waves fall on one side of it)
        const int ia = i0 + threadIdx.x, ib = ia + bd;
        const bool hb = ib < T;
        const int ic = hb ? ib : ia;
        const float4 q0 = gq[ia], q1 = gq[ic], k0 = gk[ia], k1 = gk[ic], v0 = gv[ia], v1 = gv[ic];
        const float4 d0 = gd[ia], d1 = gd[ic], o0 = go[ia], o1 = go[ic];
        const float l0 = lse[hq0 + ia], l1 = lse[hq0 + ic];
        auto one = [&](int t, float4 q, float4 k, float4 v, float4 d, float4 o, float l) {
          reinterpret_cast<float4*>(Qp)[t] = q;
          reinterpret_cast<float4*>(Kp)[t] = k;
          reinterpret_cast<float4*>(Vp)[t] = v;
          reinterpret_cast<float4*>(Dp)[t] = d;
          reinterpret_cast<double4*>(dQb)[t] = make_double4(0., 0., 0., 0.);
          Ls[t] = fmaf(-l, RAL_LOG2E, RAL_PSH); Dl[t] = -f4dot(d, o);   // (both times their factors in pass 2)
          const float md = group_max<64>(f4amax(d)), mv = group_max<64>(f4amax(v));
          const float mq = group_max<64>(f4amax(q)), mk = group_max<64>(f4amax(k));
          if (lane == 0) {
            unsigned* m4 = mx + 4 * (t / N);
            atomicMax(m4, __float_as_uint(md)); atomicMax(m4 + 1, __float_as_uint(mv));
            atomicMax(m4 + 2, __float_as_uint(mq * RAL_LOG2E)); atomicMax(m4 + 3, __float_as_uint(mk));
          }
        };
        if (ia < T) one(ia, q0, k0, v0, d0, o0, l0);
        if (hb) one(ib, q1, k1, v1, d1, o1, l1);
      }
    }
    __syncthreads();
    // ---- staging pass 2: every tensor times its head's power of two, split in place
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const unsigned* m4 = mx + 4 * (t / N);
      const float cd = pow2_to(m4[0], 2), cv = pow2_to(m4[1], 0);
      float cq, ck;
      qk_scales(m4[2], m4[3], cq, ck);
      const float4 q = reinterpret_cast<const float4*>(Qp)[t], k = reinterpret_cast<const float4*>(Kp)[t];
      const float4 v = reinterpret_cast<const float4*>(Vp)[t], d = reinterpret_cast<const float4*>(Dp)[t];
      reinterpret_cast<uint4*>(Qp)[t] = pair_quad(q, RAL_LOG2E * cq);
      reinterpret_cast<uint4*>(Kp)[t] = pair_quad(k, ck);
      reinterpret_cast<uint4*>(Vp)[t] = pair_quad(v, cv);
      reinterpret_cast<uint4*>(Dp)[t] = pair_quad(d, cd);
      Dl[t] *= cd * cv;
      Ls[t] *= cq * ck;
    }
    __syncthreads();
    // the head's output factors
    float sc_dk, sc_dv, sc_tab, su;
    bool need_su;
    {
      const unsigned* m4 = mx + 4 * (hl < HG ? hl : 0);
      const float cd = pow2_to(m4[0], 2), cv = pow2_to(m4[1], 0);
      float cq, ck;
      need_su = qk_scales(m4[2], m4[3], cq, ck);
      su = pow2_inv(cq) * pow2_inv(ck);
      sc_tab = RAL_PSH_INV * pow2_inv(cd) * pow2_inv(cv);
      sc_dk = RAL_LN2 * sc_tab * pow2_inv(cq);
      sc_dv = RAL_PSH_INV * pow2_inv(cd);
    }
    auto sweep = [&](auto su_on) {    // (both scaling modes: the choice is per head, wave-uniform)
      const int head = h0 + hl;
      const float* Qh = Qp + 4 * tb; const float* Kh = Kp + 4 * tb; const float* Vh = Vp + 4 * tb; const float* Dh = Dp + 4 * tb;
      // this wave's keys: S / dP column operands and k^T planes, for the whole sweep
      u32x2 Bk[KT], Bv[KT];

      f32x4 acc[KT];
      bool kin[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int k0 = ks + 16 * kt;
        Bk[kt] = *reinterpret_cast<const u32x2*>(Kh + 4 * (k0 + r) + 2 * (g & 1));
        Bv[kt] = *reinterpret_cast<const u32x2*>(Vh + 4 * (k0 + r) + 2 * (g & 1));
        acc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        kin[kt] = meets(k0, 16);
      }
#pragma unroll 1
      for (int q0 = 0; q0 < N; q0 += 16) {
        const int qr = q0 + r, q4 = q0 + 4 * g;
        const u32x2 Aq = *reinterpret_cast<const u32x2*>(Qh + 4 * qr + 2 * (g >> 1));
        const u32x2 Ad = *reinterpret_cast<const u32x2*>(Dh + 4 * qr + 2 * (g >> 1));
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + tb + q4);
        const float4 d4 = *reinterpret_cast<const float4*>(Dl + tb + q4);
        const f32x4 cl = {l4.x, l4.y, l4.z, l4.w}, cdl = {d4.x, d4.y, d4.z, d4.w};
        const u32x2 xt = tr_read((tp < 2 ? Dh : Qh) + 4 * (q4 + tq) + 2 * (tp & 1));
        const u32x4 a8 = lowrows ? u32x4{xt[0], xt[1], 0u, 0u} : u32x4{0u, 0u, xt[0], xt[1]};
        const bool qin = meets(q0, 16);
        f32x4 dq = {0.f, 0.f, 0.f, 0.f};
        auto dq_step = [&](int kp) {
          const float* bp = Xb + (kp & 1) * 2 * BUF + (4 * g + tq) * 10 + 2 * tp;
          const u32x2 b1 = tr_read(bp), b2 = tr_read(bp + BUF);
          // (k planes^T of the wave's four key tiles read again per query tile: kept in registers they are 8 more and the kernel
          // spills at the 128 of four waves per SIMD - 569 against 550 us at N = 512, same box; rows 8-15: v planes, unused)
          const u32x2 kx = tr_read((tp < 2 ? Kh : Vh) + 4 * (ks + 16 * kp + 4 * g + tq) + 2 * (tp & 1));
          dq = mm32(u32x4{kx[0], kx[1], kx[0], kx[1]}, u32x4{b1[0], b1[1], b2[0], b2[1]}, dq);
        };
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          const int k0 = ks + 16 * kt;
          f32x4 s = mm16(Aq, Bk[kt], cl);           // (Bk / Bv read per tile instead of kept: 546 against 531 us at N = 512, same box)
          const f32x4 dp = mm16(Ad, Bv[kt], cdl);
          if constexpr (decltype(su_on)::value) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] *= su;
          }
          float p[4];
          unsigned ph1[2], ph2[2], sh1[2], sh2[2];
          if (TAB && qin && kin[kt]) {
            const int ki = k0 + r - off;
            const bool kok = (unsigned)ki < (unsigned)Len;
            const int rel0 = (q4 - off) - ki + Len - 1;
            int e[4]; bool in[4]; float b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              in[j] = kok && (unsigned)(q4 + j - off) < (unsigned)Len;
              e[j] = min(max(rel0 + j, 0), 2 * Len - 2) * H + head;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = tab[e[j]];
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_exp2f(s[j] + (in[j] ? b[j] : 0.f));
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (in[j]) atomicAdd(dtab + e[j], (double)(p[j] * dp[j] * sc_tab));
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_exp2f(s[j]);
          }
          pair_of2(p[0], p[1], ph1[0], ph2[0]); pair_of2(p[2], p[3], ph1[1], ph2[1]);
          pair_prod2(p[0], dp[0], p[1], dp[1], sh1[0], sh2[0]); pair_prod2(p[2], dp[2], p[3], dp[3], sh1[1], sh2[1]);
          pair_settle(ph2[0], ph2[1], sh2[0], sh2[1]);
          acc[kt] = mm32(a8, u32x4{ph1[0], ph1[1], sh1[0], sh1[1]}, acc[kt]);
          acc[kt] = mm32(a8, u32x4{ph2[0], ph2[1], sh2[0], sh2[1]}, acc[kt]);
          RAL_XB_FENCE();
          float* bw = Xb + (kt & 1) * 2 * BUF + r * 10 + 2 * g;
          *reinterpret_cast<u32x2*>(bw) = u32x2{sh1[0], sh1[1]};
          *reinterpret_cast<u32x2*>(bw + BUF) = u32x2{sh2[0], sh2[1]};
          RAL_XB_FENCE();
          if (kt > 0) dq_step(kt - 1);
        }
        asm volatile("" ::: "memory");   // (see k_attn_bwd_m)
        dq_step(KT - 1);
        // lane (r, g) <- dQ^T rows g (h1 planes of k) and 4 + g (h2 planes) of query r; one add into the image
        {
          float v[4] = {dq[0], dq[1], dq[2], dq[3]};
          rows_transpose4(v);
          atomicAdd(dQb + 4 * (tb + q0 + r) + g, (double)(v[0] + v[1]));
        }
        asm volatile("" :: "v"(a8));
      }
      // dV (lane groups 0, 1) and dK (lane groups 2, 3) of the wave's key tiles
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int kr = ks + 16 * kt + r;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = swap16_add(acc[kt][j]);
        const float sc = g < 2 ? sc_dv : sc_dk;
        float* dst = dbase + (size_t)(g < 2 ? 2 : 1) * H * N * 4 + ((size_t)head * N + kr) * 4;
        if ((g & 1) == 0) *reinterpret_cast<float4*>(dst) = make_float4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc);
      }
    };
    if (hl < HG) { if (need_su) sweep(std::true_type{}); else sweep(std::false_type{}); }
    __syncthreads();
    // dQ leaves, scaled per head
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const unsigned* m4 = mx + 4 * (t / N);
      float cq, ck;
      qk_scales(m4[2], m4[3], cq, ck);
      const float sc = 0.5f * RAL_PSH_INV * pow2_inv(pow2_to(m4[0], 2)) * pow2_inv(pow2_to(m4[1], 0)) * pow2_inv(ck);
      const double4 v = reinterpret_cast<const double4*>(dQb)[t];
      reinterpret_cast<float4*>(dbase + (size_t)h0 * N * 4)[t] = make_float4((float)v.x * sc, (float)v.y * sc, (float)v.z * sc, (float)v.w * sc);
    }
  }
  if constexpr (TAB) {
    __syncthreads();
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tpart[(size_t)blockIdx.x * ntab + i] = (float)dtab[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static int attnm_mode() {   // RAL_ATTN_BWD_M=0: never
  static const int m = (int)ral_knob("ATTN_BWD_M", 1);
  return m;
}
bool attn_bwd_m_takes(int N, int H, int Len, bool table) {
  if (!attnm_mode()) return false;
  if (N != 32 && N != 64 && N != 128) return false;
  if (N == 32 && (H & 1)) return false;
  if (table && (2 * Len - 1) * H > 2048) return false;
  return true;
}
size_t attn_bwd_m_scratch_floats(int N, int H, int Len, bool table, int B) {
  if (!table || !attn_bwd_m_takes(N, H, Len, table)) return 0;
  return (size_t)attnw_grid_max(N, H, B) * (size_t)((2 * Len - 1) * H);
}
// ---- long windows
static int attnmh_kt(int N) { return N >= 1024 ? 8 : 4; }
size_t attn_bwd_mh_lds(int N, int H, int hg, int Len) {
  return ((size_t)26 * hg * N + 4 * 32 + 4 * hg + 8 * 4 * 160 + (Len > 0 ? (size_t)3 * (2 * Len - 1) * H + 2 : 0) + 4) * sizeof(float);
}
static int attnmh_hg(int N, int H) {   // heads per item: eight waves of KT key tiles each
  const int wph = N / (16 * attnmh_kt(N));
  return wph >= 8 ? 1 : 8 / wph;
}
bool attn_bwd_mh_takes(int N, int H, int Len, bool table) {
  static const int on = (int)ral_knob("ATTN_BWD_MH", 1);
  if (!on || N < 256) return false;
  const int kt = attnmh_kt(N), wph = N / (16 * kt);
  if (N % (16 * kt) != 0 || (wph != 1 && wph != 2 && wph != 4 && wph != 8)) return false;
  const int hg = attnmh_hg(N, H);
  if (H % hg != 0) return false;
  if (table && (2 * Len - 1) * H > 2048) return false;
  return attn_bwd_mh_lds(N, H, hg, Len) <= 150 * 1024;
}
static int attnmh_grid(int N, int H, int B) {
  const int items = B * (H / attnmh_hg(N, H));
  return items < 1024 ? items : 1024;
}
size_t attn_bwd_mh_scratch_floats(int N, int H, int Len, bool table, int B) {
  if (!table || !attn_bwd_mh_takes(N, H, Len, table)) return 0;
  return (size_t)attnmh_grid(N, H, B) * (size_t)((2 * Len - 1) * H);
}
void launch_attn_bwd_mh(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                        float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, hipStream_t s) {
  const int hg = attnmh_hg(N, H), grid = attnmh_grid(N, H, B), kt = attnmh_kt(N);
  const size_t lds = attn_bwd_mh_lds(N, H, hg, Len);
  const int ntab = table ? (2 * Len - 1) * H : 0;
#define GO(k, tab) { RAL_SET_LDS((k_attn_bwd_mh<k, tab>), lds); \
    k_attn_bwd_mh<k, tab><<<grid, 512, lds, s>>>(qkv, o_hm, do_hm, lse, table, tpart, dqkv, N, H, hg, table ? Len : 0, B); }
  if (kt == 4) { if (table) GO(4, true) else GO(4, false) }
  else { if (table) GO(8, true) else GO(8, false) }
#undef GO
  if (table) launch_attn_tpart_reduce(tpart, gtable, ntab, grid, s);
}

void launch_attn_bwd_m(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                       float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, hipStream_t s) {
  const int hw = N >= 64 ? 1 : 64 / N, T = hw * N;
  const int ntask = B * H / hw;
  const int ntab = table ? (2 * Len - 1) * H : 0;
  const int nwv = 4;   // (two-wave workgroups: 198 / 125 against 176 / 111 us at N = 128 / 64)
  const size_t lds = ((size_t)nwv * (T * 18 + 4 * 32 + 4 * 160) + 3 * ntab + 2) * sizeof(float);
  int grid = 0;
  auto grid_of = [&](auto kern) {
    const int gmax = attnw_grid_max(N, H, B);          // (the scratch is sized for one workgroup per four tasks)
    const int occ = ral_occupancy(reinterpret_cast<const void*>(kern), 64 * nwv, lds, 3);
    const int slots = ral_num_cus() * (occ > 8 ? 8 : occ);
    const int need = (ntask + nwv - 1) / nwv;
    int g = slots < need ? slots : need;
    return g < gmax ? g : gmax;
  };
#define GO(n, tab) { RAL_SET_LDS((k_attn_bwd_m<n, tab>), lds); grid = grid_of(k_attn_bwd_m<n, tab>); \
    k_attn_bwd_m<n, tab><<<grid, 64 * nwv, lds, s>>>(qkv, o_hm, do_hm, lse, table, tpart, dqkv, H, Len, ntask); }
  if (N == 32) { if (table) GO(32, true) else GO(32, false) }
  else if (N == 64) { if (table) GO(64, true) else GO(64, false) }
  else { if (table) GO(128, true) else GO(128, false) }
#undef GO
  if (table) launch_attn_tpart_reduce(tpart, gtable, ntab, grid, s);
}
