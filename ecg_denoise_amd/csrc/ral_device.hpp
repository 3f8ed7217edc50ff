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
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Host side, every translation unit: switches (ral_api.hip): compile-time defaults, changed only through ral_global_option (a diagnostic build, -DRAL_DIAG,
// also reads RAL_<NAME> from the environment); ral_env_int: the two validated environment variables of the product build
long long ral_knob(const char* name, long long dflt);
int ral_env_int(const char* name, int dflt, int lo, int hi);
int ral_num_cus();                                                          // compute units of the current device (cached per device)
int ral_occupancy(const void* kernel, int threads, size_t lds, int dflt);   // workgroups per CU, queried once per (kernel, threads, LDS bytes)


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
#ifdef RAL_GW_NOW   // diagnostic: the weight fragment of chunk 0 every time (same instructions, L1-hot)
        const int k0 = 0;
#else
        const int k0 = (kb + c) * 16;
#endif
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

// row stride (2-byte elements) of a K-contiguous split-plane tile of width K: + 8 keeps the 16-byte fragment reads of 16
// consecutive rows on distinct banks
constexpr int ldb_of(int K) { return K + 8; }

// ---------------------------------------------------------------------------------
// fp32 products on the f16 matrix cores: x = h1 + 2^-11 h2 with two fp16 pieces (h1 = fp16(x), h2 = fp16(2^11 (x - h1));
// the residual is scaled so that it stays in fp16's normal range whenever x does) carries 22-23 significant bits, and the
// three products h1 h1, h1 h2, h2 h1 reproduce the fp32 product to ~2^-21 relative (the dropped h2 h2 term is 2^-22).  The
// cross terms are summed in their own accumulator and enter the result as acc + 2^-11 accx.  3 x 16 matrix-core cycles
// per 16 x 16 x 32 against 6 x 16 for the bf16 triple split and 8 x 34.5 for the fp32 MFMA; operands are 4 bytes per
// element, as in fp32.  Nothing is clamped: an operand beyond fp16's largest finite value (65504 - LayerNorm / GELU /
// attention outputs and weights are orders of magnitude below it) becomes h1 = Inf, h2 = -Inf and the product NaN, a NaN
// stays a NaN: the result is non-finite and visible (loss = NaN), never a silently saturated number.  `f16_split = 0`
// is the path for a model that needs that range.  Gradient tensors are far BELOW fp16's range: they are multiplied by a
// power of two first (h2_row_scale below, per token; one per launch in the weight-gradient kernels).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct H2 { _Float16 a, b; };
RAL_DEV H2 f16_split2(float x) {
  H2 r;
  r.a = (_Float16)x;
  r.b = (_Float16)((x - (float)r.a) * 2048.f);
  return r;
}
#define RAL_H2_SCALE (1.0f / 2048.f)
// The same split with the residual left as it is (h2 = fp16(x - h1)): for operands that were SCALED into fp16's upper
// range first (|x| up to ~2^14).  The residual of a large element is a normal number, that of a small one a subnormal
// with an absolute error of 2^-25 - which v_mfma_f32_16x16x32_f16 multiplies exactly (tools/diag/f16_denorm_probe.hip) -,
// so x = h1 + h2 to 2^-25 of the operand's scale and all three products go into ONE accumulator.
RAL_DEV H2 f16_split2u(float x) {
  H2 r;
  r.a = (_Float16)x;
  r.b = (_Float16)(x - (float)r.a);
  return r;
}
// The same without the clamp, for operands that were brought into range by a power of two first (attention tiles,
// ral_attn.hip): a non-finite element stays non-finite (h1 = Inf / NaN, h2 = NaN) and reaches the result, as in fp32.
RAL_DEV H2 f16_split2n(float x) {
  H2 r;
  r.a = (_Float16)x;
  r.b = (_Float16)(x - (float)r.a);
  return r;
}
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
struct H2x4 { h16x4 a, b; };   // x = a + b to ~2^-23 relative (2^-25 absolute for small x), element-wise
RAL_DEV H2x4 split4(float4 x) {
  H2x4 r;
  const H2 s0 = f16_split2n(x.x), s1 = f16_split2n(x.y), s2 = f16_split2n(x.z), s3 = f16_split2n(x.w);
  r.a = h16x4{s0.a, s1.a, s2.a, s3.a}; r.b = h16x4{s0.b, s1.b, s2.b, s3.b};
  return r;
}
// q . k on fp16 pairs: the two operands of a score are BALANCED by one power of two, q' = 2^-a q, k' = 2^a k with a = half the
// difference of their largest exponents, so that the product is unchanged (no unscaling in front of the exponential) and
// both sides sit at the geometric mean of their magnitudes: finite while max|q| max|k| < 2^31, and the absolute error of
// the pieces (2^-25 below 2^-2) is as small relative to the score as it can be.
RAL_DEV void pair_balance(float mq, float mk, float& cq, float& ck) {
  const int eq = (int)(__float_as_uint(mq) >> 23), ek = (int)(__float_as_uint(mk) >> 23);
  int a = (mq > 0.f && mk > 0.f) ? (eq - ek) / 2 : 0;
  a = a < -100 ? -100 : (a > 100 ? 100 : a);
  cq = __uint_as_float((unsigned)(127 - a) << 23);
  ck = __uint_as_float((unsigned)(127 + a) << 23);
}
// Gradient rows are far below fp16's normal range (a mean-squared-error gradient is ~1e-6), so a row that feeds such a
// product is multiplied by a power of two first: the scale that puts the row's largest magnitude (bits of |max| as an
// unsigned) into [2^13, 2^14), capped at 2^60; the product is multiplied by the inverse afterwards (exact).  Entries far
// below the row's maximum lose relative precision exactly where they no longer matter for the row's dot products.
RAL_DEV float h2_row_scale(unsigned maxbits) {
  const int f = 267 - (int)(maxbits >> 23);
  return maxbits == 0u ? 1.0f : __uint_as_float((unsigned)(f < 187 ? f : 187) << 23);
}
RAL_DEV float h2_row_unscale(unsigned maxbits) {
  const int f = 267 - (int)(maxbits >> 23);
  return maxbits == 0u ? 1.0f : __uint_as_float((unsigned)(254 - (f < 187 ? f : 187)) << 23);
}

// Weight matrices reach these products as TILED split planes (k_tile_planes): tile (mt, kt) = rows 16 mt .., columns
// 32 kt .. of W, both planes, is 2 KB of contiguous memory - plane p at + 512 p elements, the 16 bytes of lane (r, g) =
// W[16 mt + r][32 kt + 8 g .. + 7] at + 8 (16 g + r) - so that one fragment load of a wave is eight whole 128-byte lines.
// (Row-major planes made it sixteen half-used lines per load; at these widths the products wait for the weight stream
// through the vector L1, not for the matrix cores: with chunk 0's fragments re-used for every chunk fc2 at C = 128 ran
// 4.2 x faster, without the MFMAs it did not change.)  KT = k-tiles per tile row of the matrix.
// the inverse of the power of two the planes of an (M x K) matrix were multiplied by (k_weight_scales, ral_fwd.hip): the
// first float of the slot behind the matrix's planes
RAL_DEV float wplane_unscale(const _Float16* Wt, int M, int K) {
  return __builtin_nontemporal_load(reinterpret_cast<const float*>(Wt + 2 * (size_t)M * K));
}
RAL_DEV const _Float16* wtile(const _Float16* Wt, int KT, int mt, int kt, int p) {
  // the tile is the same for the whole wave: saying so keeps its address in scalar registers (one shared lane offset)
  const int tile = __builtin_amdgcn_readfirstlane((mt * KT + kt) * 2 + p);
  return Wt + (size_t)tile * 512 + (threadIdx.x & 63) * 8;
}
// acc[mi][tt] += W1 X1, accx[mi][tt] += W1 X2 + W2 X1 for an MT x TT register block over k-tiles kt0 .. kt0 + K / 32 of the
// tiled matrix Wt, m-tiles mt0 ..; Xh[p]: LDS rows of ldx elements, plane stride xplane.
// A weight fragment takes ~1 us from the L2 under load and its three MFMAs 48 cycles, so the number of round trips per
// unit is what the product costs: one-row-tile units (MT = 1) request the fragments of up to eight chunks together (64
// registers), two-row-tile units (the fc1 phase, whose GELU epilogue gives the other waves work meanwhile) keep one chunk
// in flight under the MFMAs of the current one.
// MODE 0: residual pieces scaled by 2^11 (f16_split2): result = acc + 2^-11 accx.  MODE 1 / 2: both operands were scaled
// into fp16's upper range and carry UNSCALED residuals (f16_split2u): 1 = the three products go into acc alone (blocks of
// at least four tiles: with fewer, consecutive MFMAs into one tile wait for each other), 2 = cross terms still in accx,
// result = acc + accx
template <int MT, int TT, int ONE = 0>
RAL_DEV void h2_mma(const f16x8 (&a)[MT][2], const _Float16* xr, int xplane, int ldx, f32x4 (&acc)[MT][TT], f32x4 (&accx)[MT][TT]) {
  f16x8 b1[TT], b2[TT];
#pragma unroll
  for (int tt = 0; tt < TT; ++tt) {
    b1[tt] = *reinterpret_cast<const f16x8*>(xr + 16 * tt * ldx); b2[tt] = *reinterpret_cast<const f16x8*>(xr + xplane + 16 * tt * ldx);
  }
#ifdef RAL_H2_ONEACC   // diagnostic (wrong scaling of the cross terms): what ONE accumulator per tile would be worth in registers / time
  constexpr bool one = true;
#else
  constexpr bool one = ONE == 1;
#endif
  if constexpr (one) {
    // product by product over all tiles of the block: consecutive MFMAs never write the same accumulator
    // (three back-to-back into one tile would each wait for the one before)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
          acc[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][p == 0 ? 1 : 0], p == 1 ? b2[tt] : b1[tt], acc[mi][tt], 0, 0, 0);
  } else {
#pragma unroll
    for (int tt = 0; tt < TT; ++tt)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        accx[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][1], b1[tt], accx[mi][tt], 0, 0, 0);
        acc[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][0], b1[tt], acc[mi][tt], 0, 0, 0);
        accx[mi][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mi][0], b2[tt], accx[mi][tt], 0, 0, 0);
      }
  }
}
struct NoHook { RAL_DEV void operator()() const {} };
constexpr int h2_group(int kc, int gmax) {   // largest divisor of kc that is <= gmax
  int g = kc < gmax ? kc : gmax;
  while (kc % g) --g;
  return g;
}
// hook(): called once, after the first weight fragments of the unit are requested and before anything waits for them -
// the place to request global data of a LATER phase (loads return in order: requested earlier, they would be waited for
// together with the fragments)
template <int K, int MT, int TT, class Hook = NoHook, int GMAX_ = 0, int ONE = 0>
RAL_DEV void gemm_wx_h2(const _Float16* __restrict__ Wt, int KT, int mt0, int kt0, const _Float16* Xh, int xplane, int ldx,
                        int t0, f32x4 (&acc)[MT][TT], f32x4 (&accx)[MT][TT], Hook hook = Hook()) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  constexpr int KC = K / 32;
  const _Float16* xr = Xh + (t0 + r) * ldx + 8 * g;
  if constexpr (MT == 1) {
    // (with a hook the caller keeps prefetched data in registers meanwhile: four chunks at a time then)
    constexpr int GMAX = GMAX_ > 0 ? GMAX_ : (std::is_same<Hook, NoHook>::value ? 8 : 4), GS = h2_group(KC, GMAX);
#pragma unroll 1
    for (int k0 = 0; k0 < KC; k0 += GS) {
      f16x8 a[GS][1][2];
#pragma unroll
      for (int j = 0; j < GS; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#ifdef RAL_H2_NOW   // diagnostic: chunk 0's weight fragments every time (same instructions, L1-hot)
          a[j][0][p] = *reinterpret_cast<const f16x8*>(wtile(Wt, KT, mt0, kt0, p));
#else
          a[j][0][p] = *reinterpret_cast<const f16x8*>(wtile(Wt, KT, mt0, kt0 + k0 + j, p));
#endif
        }
      if (k0 == 0) hook();
#pragma unroll
      for (int j = 0; j < GS; ++j) h2_mma<1, TT, ONE>(a[j], xr + (k0 + j) * 32, xplane, ldx, acc, accx);
    }
  } else if constexpr (GMAX_ < 0) {
    // GMAX_ = -1: every K-chunk's fragments requested up front (KC * MT * 8 registers): ONE round trip per unit.  With one chunk in
    // flight (below) a unit of KC chunks costs ~KC - 1 round trips: the three MFMAs of a chunk are ~200 cycles, a fragment from the
    // L2 under load ~2000, and chunk k + 1 is only requested when chunk k is consumed.
    f16x8 a[KC][MT][2];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int p = 0; p < 2; ++p) a[kc][mi][p] = *reinterpret_cast<const f16x8*>(wtile(Wt, KT, mt0 + mi, kt0 + kc, p));
    hook();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) h2_mma<MT, TT, ONE>(a[kc], xr + kc * 32, xplane, ldx, acc, accx);
  } else {
    f16x8 a[MT][2], an[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int p = 0; p < 2; ++p) an[mi][p] = *reinterpret_cast<const f16x8*>(wtile(Wt, KT, mt0 + mi, kt0, p));
    hook();
#pragma unroll 1
    for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int p = 0; p < 2; ++p) a[mi][p] = an[mi][p];
#ifdef RAL_H2_NOW
      const int kn = 0;
#else
      const int kn = kc + 1 < KC ? kc + 1 : kc;
#endif
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int p = 0; p < 2; ++p) an[mi][p] = *reinterpret_cast<const f16x8*>(wtile(Wt, KT, mt0 + mi, kt0 + kn, p));
      h2_mma<MT, TT, ONE>(a, xr + kc * 32, xplane, ldx, acc, accx);
    }
  }
}

// One GEMM phase of a workgroup on split operands: out(M rows x ntiles * 16 tokens) = W x X^T (+ bias; row 0 of the output
// = row 16 mt0 of W, bias[0] its bias), work units of MT x 2 tiles dealt round-robin to the waves; M % (16 MT) == 0, ntiles even.  The bias rows of a unit are requested
// before its products.  epi(row0, tok, v) as in gemm_phase.
// wun: the inverse of the matrix's power-of-two scale (wplane_unscale), applied to the accumulators before the bias
template <int K, int MT, int GMAX, int ONE, class Epi, class Hook>
RAL_DEV void gemm_phase_h2_t(const _Float16* __restrict__ Wt, int KT, int mt0, int kt0, int M, const float* __restrict__ bias, float wun,
                             const _Float16* Xh, int xplane, int ldx, int ntiles, Epi& epi, Hook& hook) {
  constexpr int TT = 2;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int mb = M / (16 * MT), tb = ntiles / TT;
  for (int u = wave; u < mb * tb; u += nw) {
    const int m = u % mb, t = u / mb;
    float4 bv[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
      bv[mi] = bias ? *reinterpret_cast<const float4*>(bias + (m * MT + mi) * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc[MT][TT], accx[MT][TT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) { acc[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if constexpr (std::is_same<Hook, NoHook>::value) gemm_wx_h2<K, MT, TT, NoHook, GMAX, ONE>(Wt, KT, mt0 + m * MT, kt0, Xh, xplane, ldx, t * TT * 16, acc, accx);
    else {
      auto h1 = [&]() { if (u == wave) hook(); };
      gemm_wx_h2<K, MT, TT, decltype(h1), GMAX, ONE>(Wt, KT, mt0 + m * MT, kt0, Xh, xplane, ldx, t * TT * 16, acc, accx, h1);
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
      for (int tt = 0; tt < TT; ++tt) {
        f32x4 v = ONE == 1 ? acc[mi][tt] : (ONE == 2 ? acc[mi][tt] + accx[mi][tt] : acc[mi][tt] + accx[mi][tt] * RAL_H2_SCALE);
        v *= wun;
        v[0] += bv[mi].x; v[1] += bv[mi].y; v[2] += bv[mi].z; v[3] += bv[mi].w;
        epi((m * MT + mi) * 16 + 4 * g, (t * TT + tt) * 16 + r, v);
      }
  }
  if (wave >= mb * tb) hook();   // a wave without a unit still issues its share
}
// the larger block when it still gives every wave a unit
// (Wt, KT): tiled planes of the matrix; the product uses its rows 16 mt0 .. + M and columns 32 kt0 .. + K
// hook: see gemm_wx_h2; runs exactly once in every wave (inside its first unit)
// GMAX: chunks of weight fragments requested together by a one-row-tile unit (0: eight, four with a hook)
template <int K, int GMAX = 0, int ONE = 0, class Epi, class Hook = NoHook>
RAL_DEV void gemm_phase_h2(const _Float16* __restrict__ Wt, int KT, int mt0, int kt0, int M, const float* __restrict__ bias, float wun,
                           const _Float16* Xh, int xplane, int ldx, int ntiles, Epi epi, Hook hook = Hook()) {
  const int nw = blockDim.x >> 6;
  if (M % 32 == 0 && (M / 32) * (ntiles / 2) >= nw) gemm_phase_h2_t<K, 2, GMAX, ONE>(Wt, KT, mt0, kt0, M, bias, wun, Xh, xplane, ldx, ntiles, epi, hook);
  else gemm_phase_h2_t<K, 1, GMAX, ONE>(Wt, KT, mt0, kt0, M, bias, wun, Xh, xplane, ldx, ntiles, epi, hook);
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
// The forward alone needs no derivative, and then no reciprocal either: 0.5 erfc(a) = 2^-(1 + a P(a)) with a = |x| / sqrt2 and
// P a degree-7 polynomial (least-squares fit of -log2 erfc(a) / a on [0, 5.5] weighted by erfc(a), i.e. uniform in the
// error of Phi: 8e-9, below the 1.5e-7 of the formula above; beyond the fitted range a P(a) keeps growing and the term
// underflows to 0), so GELU(x) = max(x, 0) - |x| * 2^-(1 + a P(a)): 8 FMAs that the compiler packs in pairs + ONE
// quarter-rate instruction - 12 issue slots per element against 16 for the pair formula without its derivative.
// (tools/diag/gelu_fit.py re-derives the coefficients and checks the fp32 evaluation against scipy's erfc.)
RAL_DEV float gelu_f(float x) {
#ifdef RAL_NOGELU
  return x;
#endif
  const float ax = fabsf(x), a = ax * 0.70710678118654752440f;
  float p = fmaf(4.527986043e-05f, a, -4.451398044e-04f);
  p = fmaf(p, a, 1.489045845e-03f);
  p = fmaf(p, a, 7.741376838e-04f);
  p = fmaf(p, a, -2.825238065e-02f);
  p = fmaf(p, a, 1.484806716e-01f);
  p = fmaf(p, a, 9.184166615e-01f);
  p = fmaf(p, a, 1.627908569e+00f);
  const float h = __builtin_amdgcn_exp2f(fmaf(-a, p, -1.0f));   // 0.5 erfc(|x| / sqrt2) = Phi(-|x|)
  return fmaf(-ax, h, fmaxf(x, 0.f));
}
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
RAL_DEV float rows_max(float v) {   // max over the four rows of 16 lanes (same column)
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
  p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
}
template <int W> RAL_DEV float group_max(float v) {   // max over W consecutive lanes, result in every lane
#define RAL_DPP_MAX(CTRL) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)))
  if constexpr (W >= 2) RAL_DPP_MAX(0xB1);
  if constexpr (W >= 4) RAL_DPP_MAX(0x4E);
  if constexpr (W >= 8) RAL_DPP_MAX(0x141);
  if constexpr (W >= 16) RAL_DPP_MAX(0x140);
#undef RAL_DPP_MAX
  if constexpr (W >= 32) {
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
  }
  if constexpr (W >= 64) {
    const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(p[0]), __uint_as_float(p[1]));
  }
  return v;
}
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

// Workgroup barrier that orders LDS accesses ONLY: __syncthreads() also waits for every outstanding GLOBAL load of the wave
// (s_waitcnt vmcnt(0) in front of s_barrier), which puts the HBM round trip of a prefetch issued earlier back on the critical
// path; here the prefetched values stay in flight (the compiler waits for them where their registers are first used).
RAL_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// PatchSeparate's token order (reference quirk A5: the output rows are [first channel halves of all input tokens ; second
// halves]) on a window of T output slots of which the first Tv exist (Tv <= T, both even; a window length that is not a multiple
// of 256 runs on padded slots, ral_api.hip): output row `row` reads / back-propagates into input slot l, channel half c1 - float
// offset l * 2D + c1 * D inside the window.  Existing rows map onto the existing input slots [0, Tv / 2) exactly as the reference
// orders them; the padding rows are mapped one to one onto the padding input slots, so that the backward writes every input slot
// exactly once (zeros into the padding).
RAL_DEV size_t sep_src(int row, int T, int Tv, int D) {
  const int hv = Tv >> 1;
  int l, c1;
  if (row < Tv) { c1 = row >= hv ? 1 : 0; l = row - c1 * hv; }
  else { const int hp = (T >> 1) - hv, p = row - Tv; c1 = p >= hp ? 1 : 0; l = hv + p - c1 * hp; }
  return (size_t)l * 2 * D + (size_t)c1 * D;
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
  const float* asc;   // the block's activation scales (ASC_* below; k_act_scales) or nullptr = every scale 1 (gradient / transposed views)
};

// Powers of two for the ACTIVATION operands of the fp16-pair products of one block, from bounds that hold for ANY input:
//   LayerNorm output   |h_i| <= sqrt(C) max|gamma| + max|beta|,   |h|_2 <= sqrt(C) max|gamma| + |beta|_2
//   attention output   |o_i| <= max |v_i| <= max_j (|Wv_j|_2 |h1|_2 + |bv_j|)          (a convex combination of value rows)
//   hidden (fc2 input) |a2_i| <= max(1, sum|le|) max_j (|W1_j|_2 |h2|_2 + |b1_j|)      (|GELU(z)| <= |z|; the 3-tap conv on channel 0)
// each scale puts its bound into [2^13, 2^14): no operand of a forward or weight-gradient product can reach fp16's largest
// finite value whatever the weights are (a model whose hidden values pass 65504 used to end in NaN on this path where the
// reference's fp32 nn.Linear is finite), and activations far BELOW their usual size (weights x 1e-6) keep their 22 bits.
// The inverse of every scale is a power of two as well and is folded into the product's weight-plane unscale: exact.
enum { ASC_LN1 = 0, ASC_LN1_INV = 1, ASC_O = 2, ASC_O_INV = 3, ASC_LN2 = 4, ASC_LN2_INV = 5, ASC_HID = 6, ASC_HID_INV = 7, ASC_N = 8 };
RAL_DEV float asc_get(const float* asc, int k) { return asc ? __builtin_nontemporal_load(asc + k) : 1.0f; }

// ---------------------------------------------------------------------------------
// loss + metrics: per window sse = sum (p-t)^2, sy2 = sum t^2 over leads*L
//   snr = 10 log10(sy2/sse), rmse = sqrt(sse/n), dy = 2 (p-t) / (global_B * n)
//   loss_sum += sse / n   (caller divides by the global batch)
// ---------------------------------------------------------------------------------
// One thread per workgroup adds the workgroup's share.  With `fin` (ral_loss_mean) the LAST workgroup to arrive also
// finishes the job on the device: fin[0] = total * fin_scale (the mean over the global batch), and the accumulator and
// the arrival counter go back to zero for the next call - no fill kernel before the launch, no division kernel after it
// (they were two torch kernels per training step).  scratch: {double sum, unsigned long long arrivals}, zero on entry.
// fin3 (ral_loss_means): the sums of the windows' SNR and RMSE ride along and their means over the global batch leave as
// fin[1], fin[2] - the trainer's per-step metrics then cost no kernel of their own.  The four words then sit in FOUR cache
// lines (scratch of 64 doubles: sum [0], arrivals [16], SNR [32], RMSE [48]): same-line atomics are served one after the other,
// ~12 ns each (256 workgroups: 12.8 us with two words in one line, 18.8 with four), different lines side by side.
RAL_DEV void loss_commit(double* sum, double mine, double* fin, double fin_scale, int fin3 = 0, double msnr = 0.0, double mrmse = 0.0) {
  atomicAdd(sum, mine);
  if (!fin) return;
  if (fin3) { atomicAdd(sum + 32, msnr); atomicAdd(sum + 48, mrmse); }
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(sum + (fin3 ? 16 : 1));
  __threadfence();
  if (atomicAdd(cnt, 1ull) == (unsigned long long)gridDim.x - 1ull) {
    __threadfence();
    const double tot = atomicAdd(sum, 0.0);
    fin[0] = tot * fin_scale;
    atomicExch(reinterpret_cast<unsigned long long*>(sum), 0ull);
    if (fin3) {
      fin[1] = atomicAdd(sum + 32, 0.0) * fin_scale;
      fin[2] = atomicAdd(sum + 48, 0.0) * fin_scale;
      atomicExch(reinterpret_cast<unsigned long long*>(sum + 32), 0ull);
      atomicExch(reinterpret_cast<unsigned long long*>(sum + 48), 0ull);
    }
    atomicExch(cnt, 0ull);
  }
}
