// Device-side building blocks for the RA-LENet kernels (gfx950 / CDNA4 only).
//
// Every contraction on the path goes through v_mfma_f32_16x16x4_f32 (exact fp32, the
// f32 matrix rate of gfx950 = 157 TFLOP/s).  Operand conventions used everywhere:
//
//   D[i][j] = sum_k A[i][k] * B[k][j]      i, j in [0,16), k in [0,4)
//   lane l: r = l & 15, g = l >> 4
//   A operand: lane holds A[i = r][k = g]          B operand: lane holds B[k = g][j = r]
//   C/D:       lane holds D[row = 4g + q][col = r], q = 0..3  (one f32x4)
//
// "weights x activations" products put the OUTPUT CHANNEL on the MFMA row and the
// TOKEN on the MFMA column, so a lane ends up with 4 consecutive channels of one
// token: exactly one float4 of a token-major row, or one head's 4-vector.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RAL_DEV __device__ __forceinline__

RAL_DEV f32x4 mfma4(float a, float b, f32x4 c) {
#ifdef RAL_NOGEMM   // diagnostic builds (make VARIANT=nogemm EXTRA=-DRAL_NOGEMM): what a kernel costs without its
  return c;         // matrix work AND the operand fetches that only feed it (the compiler drops them as dead)
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}

// Activation tile layouts (global memory and LDS use the same two forms):
//   LAY_TOK: token-major rows,   elem(t, c) at  t*ld + c
//   LAY_HM : head-major quads,   elem(t, c) at ((c>>2)*ntok + t)*4 + (c&3)   ("ld" = ntok)
enum { LAY_TOK = 0, LAY_HM = 1 };

template <int LAY>
RAL_DEV int xoff(int ld, int t, int c) {
  if (LAY == LAY_TOK) return t * ld + c;
  return (((c >> 2) * ld + t) << 2) + (c & 3);
}

// ---------------------------------------------------------------------------------
// acc[tt] (+)= W[m0.., :K] x X[t0 + 16 tt .., :K]^T
//   WT == false: W is (M, K) row-major in global memory, A[i][k] = W[(m0+i)*ldw + k]
//   WT == true : W is (K, M) row-major,                  A[i][k] = W[k*ldw + m0 + i]
//   X is an LDS tile in layout LAY.  K is a multiple of 16, or exactly 8.
// The k index handled by (lane group g, step s) is k0 + 4g + s (K%16==0) so that one
// 16-byte read feeds four MFMAs; both operands use the same permutation.
template <int K, int TT, bool WT, int LAY>
RAL_DEV void gemm_wx(const float* __restrict__ W, int ldw, int m0, int M, const float* Xs, int ldx,
                     int t0, f32x4 (&acc)[TT]) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  int mrow = m0 + r;
  if (mrow >= M) mrow = M - 1;  // clamp: rows >= M are computed on garbage and discarded
  if constexpr (K % 16 == 0) {
    // weight fragments of up to KB k-chunks are requested together (one L2 round trip for all of them)
    constexpr int NK = K / 16, KBMAX = WT ? 2 : 8, KB = NK < KBMAX ? NK : KBMAX;
#pragma unroll 1
    for (int kb = 0; kb < NK; kb += KB) {
      float4 wv[KB];
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const int k0 = (kb + c) * 16;
        if constexpr (!WT) {
          wv[c] = *reinterpret_cast<const float4*>(W + (size_t)mrow * ldw + k0 + 4 * g);
        } else {
          wv[c].x = W[(size_t)(k0 + 4 * g + 0) * ldw + mrow];
          wv[c].y = W[(size_t)(k0 + 4 * g + 1) * ldw + mrow];
          wv[c].z = W[(size_t)(k0 + 4 * g + 2) * ldw + mrow];
          wv[c].w = W[(size_t)(k0 + 4 * g + 3) * ldw + mrow];
        }
      }
#pragma unroll
      for (int c = 0; c < KB; ++c) {
        const int k0 = (kb + c) * 16;
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const float4 x4 = *reinterpret_cast<const float4*>(Xs + xoff<LAY>(ldx, t0 + 16 * tt + r, k0 + 4 * g));
          acc[tt] = mfma4(wv[c].x, x4.x, acc[tt]);
          acc[tt] = mfma4(wv[c].y, x4.y, acc[tt]);
          acc[tt] = mfma4(wv[c].z, x4.z, acc[tt]);
          acc[tt] = mfma4(wv[c].w, x4.w, acc[tt]);
        }
      }
    }
  } else {
    static_assert(K == 8, "K must be a multiple of 16 or exactly 8");
    float wv[2];
    if constexpr (!WT) {
      const float2 w2 = *reinterpret_cast<const float2*>(W + (size_t)mrow * ldw + 2 * g);
      wv[0] = w2.x; wv[1] = w2.y;
    } else {
      wv[0] = W[(size_t)(2 * g) * ldw + mrow];
      wv[1] = W[(size_t)(2 * g + 1) * ldw + mrow];
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const float2 x2 = *reinterpret_cast<const float2*>(Xs + xoff<LAY>(ldx, t0 + 16 * tt + r, 2 * g));
      acc[tt] = mfma4(wv[0], x2.x, acc[tt]);
      acc[tt] = mfma4(wv[1], x2.y, acc[tt]);
    }
  }
}

// One GEMM phase of a workgroup: out(M rows x ntiles*16 tokens) = W x X^T, work units
// (16-row m tile, TTB token tiles) dealt round-robin to the waves.  epi(row0, tok, v)
// receives rows row0..row0+3 (row0 % 4 == 0, row0 < M) of token `tok`.
template <int K, int TTB, bool WT, int LAY, class Epi>
RAL_DEV void gemm_phase_t(const float* __restrict__ W, int ldw, int M, const float* Xs, int ldx,
                          int ntiles, Epi& epi) {
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int mt = (M + 15) >> 4, tg = ntiles / TTB;
  for (int u = wave; u < mt * tg; u += nw) {
    const int m = u % mt, tgi = u / mt;
    f32x4 acc[TTB];
#pragma unroll
    for (int tt = 0; tt < TTB; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    gemm_wx<K, TTB, WT, LAY>(W, ldw, m * 16, M, Xs, ldx, tgi * TTB * 16, acc);
    const int row0 = m * 16 + 4 * g;
    if (row0 < M) {
#pragma unroll
      for (int tt = 0; tt < TTB; ++tt) epi(row0, (tgi * TTB + tt) * 16 + r, acc[tt]);
    }
  }
}

// The number of token tiles that share one weight-fragment load is the largest of {4, 2, 1} dividing the
// tile count (a wide level with few tokens would otherwise re-read every weight once per 16 tokens).
// TTBHINT is kept for call-site documentation only.
template <int K, int TTBHINT, bool WT, int LAY, class Epi>
RAL_DEV void gemm_phase(const float* __restrict__ W, int ldw, int M, const float* Xs, int ldx,
                        int ntiles, Epi epi) {
  if ((ntiles & 3) == 0) gemm_phase_t<K, 4, WT, LAY>(W, ldw, M, Xs, ldx, ntiles, epi);
  else if ((ntiles & 1) == 0) gemm_phase_t<K, 2, WT, LAY>(W, ldw, M, Xs, ldx, ntiles, epi);
  else gemm_phase_t<K, 1, WT, LAY>(W, ldw, M, Xs, ldx, ntiles, epi);
}

// ---------------------------------------------------------------------------------
// fp32 products on the bf16 matrix cores (pilot: the QKV projection of the wide levels).
// x = x1 + x2 + x3 with three bf16 pieces carries the 24 significant bits of an fp32 value (x - x1 and x - x1 - x2 are
// exact in fp32); the six piece products whose weight is >= 2^-16 (W1X1, W1X2, W2X1, W1X3, W2X2, W3X1), accumulated in
// fp32 by v_mfma_f32_16x16x32_bf16, reproduce the fp32 product to ~2^-22 relative - 6 x 16 matrix-core cycles per
// 16 x 16 x 32 instead of 8 x 34.5 cycles of the fp32 MFMA, which runs on the vector ALU's multipliers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Bf3 { __bf16 a, b, c; };
RAL_DEV Bf3 bf16_split3(float x) {
  Bf3 r;
  r.a = (__bf16)x;
  const float e1 = x - (float)r.a;
  r.b = (__bf16)e1;
  r.c = (__bf16)(e1 - (float)r.b);
  return r;
}
// row stride (bf16 elements) of a K-contiguous bf16 tile of width K: + 8 keeps the 16-byte fragment reads of 16
// consecutive rows on distinct banks
constexpr int ldb_of(int K) { return K + 8; }

// acc[mi][tt] += W[m0 + 16 mi .., :K] x X[t0 + 16 tt .., :K]^T from three bf16 planes each: Wb[p] (rows of K, row-major,
// global, plane stride wplane elements), Xb[p] (LDS rows of ldx elements, plane stride xplane).  K % 32 == 0.  An
// MT x TT register block: (MT + TT) x 3 fragment loads of 16 bytes feed MT * TT * 6 MFMAs - with 2 x 4 the LDS sees 14
// bytes per cycle per SIMD of its 32 and the vector L1 7 of its 16 (one 16 x 16 x 32 bf16 MFMA issues in 18.7 cycles,
// tools/diag/valu_probe.hip), where a 1 x 2 block would need 27 and 14.
template <int K, int MT, int TT>
RAL_DEV void gemm_wx_b3(const __bf16* __restrict__ Wb, size_t wplane, int m0, const __bf16* Xb, int xplane, int ldx,
                        int t0, f32x4 (&acc)[MT][TT]) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  auto mma6 = [&](const bf16x8 (&a)[3], const bf16x8& b1, const bf16x8& b2, const bf16x8& b3, f32x4& c) {   // smallest terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b3, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b1, c, 0, 0, 0);
  };
  // weight fragments of chunk kc + 1 are requested before the MFMAs of chunk kc (the loop stays rolled: unrolled, hipcc
  // hoists every chunk's loads to the top and spills)
  bf16x8 a[MT][3], an[MT][3];
  const __bf16* wp = Wb + (size_t)(m0 + r) * K + 8 * g;
#pragma unroll
  for (int mi = 0; mi < MT; ++mi)
#pragma unroll
    for (int p = 0; p < 3; ++p) an[mi][p] = *reinterpret_cast<const bf16x8*>(wp + p * wplane + (size_t)mi * 16 * K);
#pragma unroll 1
  for (int kc = 0; kc < K / 32; ++kc) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[mi][p] = an[mi][p];
    const int kn = kc + 1 < K / 32 ? kc + 1 : kc;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int p = 0; p < 3; ++p) an[mi][p] = *reinterpret_cast<const bf16x8*>(wp + p * wplane + (size_t)mi * 16 * K + kn * 32);
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
      const int xo = (t0 + 16 * tt + r) * ldx + kc * 32 + 8 * g;
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Xb + xo), b2 = *reinterpret_cast<const bf16x8*>(Xb + xplane + xo),
                   b3 = *reinterpret_cast<const bf16x8*>(Xb + 2 * xplane + xo);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) mma6(a[mi], b1, b2, b3, acc[mi][tt]);
    }
  }
}

// row stride (floats) of a token-major LDS tile of width C: +4 breaks the power-of-two stride for the
// b128 fragment reads; the 8-wide level keeps its rows dense so that 1024-sample windows still fit
template <int C> struct LDof { static constexpr int v = (C == 8) ? 8 : C + 4; };
static inline int ld_of(int C) { return C == 8 ? 8 : C + 4; }

// token tiles processed per weight-fragment load, by channel width (keeps ntiles % TTB == 0
// for every window length that is a multiple of 256)
template <int C> struct TTBof { static constexpr int v = (C <= 32) ? 4 : (C == 64 ? 2 : 1); };

// ---------------------------------------------------------------------------------
// small math
// Exact-erf GELU (nn.GELU default) and its derivative from ONE evaluation of
//   erfc(z) ~= (a1 t + ... + a5 t^5) e^{-z^2},  t = 1/(1 + p z)      (Abramowitz-Stegun 7.1.26,
// |error| <= 1.5e-7 absolute = fp32 rounding level): ~15 VALU instructions with one v_rcp and one
// v_exp, against ~45 for libm erff + expf with both of its branches taken by a divergent wave.
RAL_DEV void gelu_pair(float x, float& g, float& dg) {
#ifdef RAL_NOGELU   // diagnostic builds: what a kernel costs without its GELU evaluations
  g = x; dg = 1.f;
  return;
#endif
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));   // exp(-x^2/2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float hq = 0.5f * p * t * e;               // 0.5 * erfc(|x|/sqrt2) = Phi(-|x|)
  const float cdf = x >= 0.f ? 1.0f - hq : hq;
  g = x * cdf;
  dg = fmaf(x, 0.39894228040143267794f * e, cdf);
}
RAL_DEV float gelu_f(float x) { float g, d; gelu_pair(x, g, d); return g; }
RAL_DEV float gelu_grad_f(float x) { float g, d; gelu_pair(x, g, d); return d; }

// Cross-lane sums without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32, ~100 cycles of latency per
// step): DPP quad permutes / row mirrors inside a row of 16 lanes, and the gfx950 row-swap instructions across
// rows.  After each step every lane of the group holds the same partial sum, so mirrors can replace xors.
template <int CTRL> RAL_DEV float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
RAL_DEV float swap16_add(float v) {   // v[lane] + v[lane ^ 16]
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
RAL_DEV float swap32_add(float v) {   // v[lane] + v[lane ^ 32]
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}
RAL_DEV float rows_sum(float v) { return swap32_add(swap16_add(v)); }
// 4 x 4 transpose between registers and the four 16-lane rows of a wave: in, register j of lane (r, g) holds E(j, g);
// out, register j of lane (r, g) holds E(g, j).  (v_permlane32_swap a, b: a = [a.lanes 0-31, b.lanes 0-31], b =
// [a.lanes 32-63, b.lanes 32-63]; v_permlane16_swap likewise inside each half.)
RAL_DEV void rows_transpose4(float (&v)[4]) {
  auto sw32 = [](float& a, float& b) {
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(p[0]); b = __uint_as_float(p[1]);
  };
  auto sw16 = [](float& a, float& b) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(p[0]); b = __uint_as_float(p[1]);
  };
  sw32(v[0], v[2]); sw32(v[1], v[3]);
  sw16(v[0], v[1]); sw16(v[2], v[3]);
}   // over the 4 rows of 16 lanes (same column)
template <int W>
RAL_DEV float group_sum(float v) {  // sum over W consecutive lanes (W power of two <= 64), result in every lane
  if constexpr (W >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (W >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (W >= 8) v = dpp_add<0x141>(v);   // row_half_mirror
  if constexpr (W >= 16) v = dpp_add<0x140>(v);  // row_mirror
  if constexpr (W >= 32) v = swap16_add(v);
  if constexpr (W >= 64) v = swap32_add(v);
  return v;
}

RAL_DEV float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RAL_DEV float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
RAL_DEV float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
RAL_DEV float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
RAL_DEV float f4hsum(float4 a) { return (a.x + a.y) + (a.z + a.w); }
RAL_DEV float f4dot(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
RAL_DEV float4 tofloat4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }

// LayerNorm statistics of one row held as float4 per lane across LPR lanes (C = 4*LPR).
// Returns centred values in d, and rstd (biased variance, eps 1e-5 as nn.LayerNorm).
template <int LPR>
RAL_DEV void ln_stats(float4 v, float4& d, float& rstd) {
  constexpr float invC = 1.0f / (4 * LPR);
  const float mean = group_sum<LPR>(f4hsum(v)) * invC;
  d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
  const float var = group_sum<LPR>(f4dot(d, d)) * invC;
  rstd = 1.0f / sqrtf(var + 1e-5f);
}

// ---------------------------------------------------------------------------------
// Phase stamps (diagnostic builds only: `make STAMP=<name> STAMPTU=FWD|BWD|DW STAMPCOND='<expr>'`,
// tools/diag/stamp_kernel.py; a translation unit opts in by defining RAL_STAMP_HERE before this header).  Thread 0 of workgroup 0 adds
// the cycles since the previous stamp to slot i; the product library compiles these macros to nothing.
#if defined(RAL_STAMP) && defined(RAL_STAMP_HERE)
// every translation unit has its own slot array; RAL_STAMPS_DEFINE(name) exports its accessor
static __device__ unsigned long long g_ral_stamps[32];
#define RAL_STAMPS_DEFINE(name)                                                                        \
  extern "C" int name(unsigned long long* out, int reset) {                                            \
    unsigned long long z[32] = {0};                                                                    \
    if (reset) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ral_stamps), z, sizeof(z));                  \
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ral_stamps), sizeof(z));                         \
  }
#ifndef RAL_STAMP_COND
#define RAL_STAMP_COND true   /* e.g. -DRAL_STAMP_COND='(C==128)' to stamp one channel width only */
#endif
#define RAL_STAMP_INIT() long long ral_st_prev_ = clock64()
#define RAL_STAMP_AT(i)                                                                                \
  do {                                                                                                 \
    if (RAL_STAMP_COND && blockIdx.x == 0 && threadIdx.x == 0) {                                       \
      const long long t_ = clock64();                                                                  \
      atomicAdd(&g_ral_stamps[i], (unsigned long long)(t_ - ral_st_prev_));                            \
      ral_st_prev_ = t_;                                                                               \
    }                                                                                                  \
  } while (0)
#else
#define RAL_STAMPS_DEFINE(name)
#define RAL_STAMP_INIT() do {} while (0)
#define RAL_STAMP_AT(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------
// Staging loops.  hipcc waits for each global load right before its use, so a plain
// `for (i = tid; i < n; i += blockDim) dst[i] = src[i]` costs one full HBM latency PER ITERATION.
// These helpers issue U independent 16-byte loads first and consume them afterwards.
template <int U, class F>
RAL_DEV void for_each_f4(const float* __restrict__ src, int n4, F f) {   // f(index, value) over a flat float4 array
  const float4* s = reinterpret_cast<const float4*>(src);
  const int bd = blockDim.x;
  int i = threadIdx.x;
  for (; i + (U - 1) * bd < n4; i += U * bd) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[i + u * bd];
#pragma unroll
    for (int u = 0; u < U; ++u) f(i + u * bd, v[u]);
  }
  for (; i < n4; i += bd) f(i, s[i]);
}

// rows x width (floats, width % 4 == 0) block of a row-major global array with row stride gld: f(row, col, value)
template <int U, class F>
RAL_DEV void for_each_row_f4(const float* __restrict__ src, int gld, int rows, int width, F f) {
  const int q = width >> 2, n4 = rows * q, bd = blockDim.x;
  int i = threadIdx.x;
  for (; i + (U - 1) * bd < n4; i += U * bd) {
    float4 v[U];
    int rr[U], cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = i + u * bd;
      rr[u] = j / q; cc[u] = (j - rr[u] * q) << 2;
      v[u] = *reinterpret_cast<const float4*>(src + (size_t)rr[u] * gld + cc[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) f(rr[u], cc[u], v[u]);
  }
  for (; i < n4; i += bd) {
    const int row = i / q, c = (i - row * q) << 2;
    f(row, c, *reinterpret_cast<const float4*>(src + (size_t)row * gld + c));
  }
}

// coalesced copy of an LDS tile (rows x width, row stride ld) to global rows of `gld` floats
RAL_DEV void copy_out(float* __restrict__ dst, int gld, const float* src, int ld, int rows, int width) {
  const int q = width >> 2;
  for (int i = threadIdx.x; i < rows * q; i += blockDim.x) {
    const int row = i / q, c = (i - row * q) << 2;
    *reinterpret_cast<float4*>(dst + (size_t)row * gld + c) = *reinterpret_cast<const float4*>(src + row * ld + c);
  }
}
RAL_DEV void copy_in(float* dst, int ld, const float* __restrict__ src, int gld, int rows, int width) {
  for_each_row_f4<4>(src, gld, rows, width, [&](int row, int c, float4 v) {
    *reinterpret_cast<float4*>(dst + row * ld + c) = v;
  });
}
// flat float4 copy (n4 float4s)
RAL_DEV void copy_flat(float* dst, const float* __restrict__ src, int n4) {
  for_each_f4<4>(src, n4, [&](int i, float4 v) { reinterpret_cast<float4*>(dst)[i] = v; });
}

// Parameters of one TransformerBlock inside the flat parameter (or gradient) buffer.
struct BlockP {
  float* wqkv;  // (3C, C): to_q.weight then to_kv.weight
  float* bqkv;  // (3C)
  float* wp;    // (C, C)   attn.proj
  float* bp;    // (C)
  float* ln1w; float* ln1b; float* ln2w; float* ln2b;
  float* w1;    // (4C, C)  mlp.fc1
  float* b1;    // (4C)
  float* w2;    // (C, 4C)  mlp.fc2
  float* b2;    // (C)
  float* le;    // (3) leconv.partial_conv3.weight or nullptr
};
