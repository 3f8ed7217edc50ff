// Narrow levels (C = 8, 16, 32): the post-attention half of a TransformerBlock, wave-autonomous.
//
// Reference ops (model/raletransformer.py:392-410, 143-158):  x1 = x + o Wp^T + bp;  g = LN2(x1);  u = g W1^T + b1;
// a = GELU(u)  [local enhancement: a[:, 0] = conv3(a[:, 0]) over tokens, a = GELU(a)];  x2 = x1 + a W2^T + b2.
//
// k_mlp_fwd (ral_fwd.hip) gives a window to a 512-thread workgroup and walks it phase by phase: stage, proj, LayerNorm,
// then per hidden chunk fc1 / conv / fc2, each phase closed by a workgroup barrier and opened by an L2 round trip for its
// weight fragments; the whole window sits in the LDS, one or two workgroups per CU.  At these widths a phase is a few
// MFMAs per wave, so the kernel is its barriers and round trips (wave_parked 0.53, r03 counters).  Here a WAVE carries a
// strip of 64 tokens (four 16-token tiles; two at C = 32) through the whole chain in registers:
//   * an accumulator tile IS the next product's B operand: D[channel 4g+q][token r] is held as register q of lane (r, g),
//     and v_mfma_f32_16x16x4_f32 wants B[k = g][j = r] - MFMA number q of a 16-channel block takes register q, with the
//     weight fragment W[m][16 b + 4 g + q] as its A operand (one 16-byte LDS read feeds the four MFMAs of a block).  So
//     proj -> LayerNorm -> fc1 -> GELU -> fc2 never leaves the registers and there is no barrier in the strip loop;
//   * the block's weights (40 KB at C = 32) are staged in the LDS once per (persistent) workgroup, zero-padded to whole
//     16 x 16 tiles, so the C = 8 level runs the same code (its rows / columns 8 .. 15 are zeros);
//   * LayerNorm statistics: the channels of a token are spread over (channel tile, lane group g, register q) of ONE lane
//     column - two v_permlane swaps per sum;
//   * the local-enhancement conv needs GELU(u[:, 0]) of the neighbouring tokens: inside a tile a DPP row shift, across
//     tiles the other tile's edge lane (pass 1 computes row 0 of fc1 for all four tiles first), across strips the two
//     edge tokens are re-computed on the vector ALU by the two halves of the wave (C^2 + 3 C multiply-adds each);
//   * x1 (training) and x2 leave as 16-byte stores straight from the accumulators; u_pre is not stored (the narrow-level
//     backward re-computes it).
// Arithmetic: fp32 MFMA and fp32 vector ALU, the same products in a different summation order than k_mlp_fwd.
#include "ral_device.hpp"
#include "ral_kernels.hpp"
#include <stdio.h>
#include <stdlib.h>

#ifndef RAL_MLPW_WPE
#define RAL_MLPW_WPE 3
#endif

RAL_DEV float lane_value(float v, int l) {   // v of lane l (a constant), in every lane
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
template <int CTRL> RAL_DEV float dpp_shift(float v) {   // row shift inside a row of 16 lanes; the vacated lane reads 0
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int C>
struct MlpwShape {
  static constexpr int KP = C < 16 ? 16 : C;        // channel count padded to whole tiles
  static constexpr int MT = KP / 16;                // channel tiles
  static constexpr int HID = 4 * C, HT = HID / 16;  // hidden channels / tiles
  static constexpr int LDC = KP + 4, LDH = HID + 4; // LDS row strides of the weight matrices (K = C, K = 4C)
  static constexpr int S = C >= 32 ? 2 : 4;         // 16-token tiles per strip (x1 and LN2(x1) of a strip stay in registers)
  // floats: Wp [KP][LDC] | W1 [HID][LDC] | W2 [KP][LDH] | bp, g2, be2, b2 [KP each] | b1 [HID] | per-wave halo scratch 4 x 64
  static constexpr int W1O = KP * LDC, W2O = W1O + HID * LDC, VO = W2O + KP * LDH, B1O = VO + 4 * KP, SCR = B1O + HID;
  static constexpr int TOTAL = SCR + 4 * 64;
};

template <int C>
__global__ __launch_bounds__(256, RAL_MLPW_WPE) void k_mlp_fwd_w(const float* __restrict__ x, const float* __restrict__ o_hm,
                                                                 BlockP w, float* __restrict__ x1_out, float* __restrict__ x2_out,
                                                                 int N, int B) {
  using SH = MlpwShape<C>;
  constexpr int KP = SH::KP, MT = SH::MT, HID = SH::HID, HT = SH::HT, LDC = SH::LDC, LDH = SH::LDH, S = SH::S;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  float* Wp = sm; float* W1 = sm + SH::W1O; float* W2 = sm + SH::W2O;
  float* bp = sm + SH::VO; float* g2 = bp + KP; float* be2 = g2 + KP; float* b2 = be2 + KP; float* b1 = sm + SH::B1O;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* scr = sm + SH::SCR + wave * 64;
  // ---- the block's weights -> LDS, zero-padded to whole tiles (once per workgroup)
  for (int i = threadIdx.x; i < SH::SCR; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) Wp[(i / C) * LDC + i % C] = w.wp[i];
  for (int i = threadIdx.x; i < HID * C; i += blockDim.x) W1[(i / C) * LDC + i % C] = w.w1[i];
  for (int i = threadIdx.x; i < C * HID; i += blockDim.x) W2[(i / HID) * LDH + i % HID] = w.w2[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) { bp[i] = w.bp[i]; g2[i] = w.ln2w[i]; be2[i] = w.ln2b[i]; b2[i] = w.b2[i]; }
  for (int i = threadIdx.x; i < HID; i += blockDim.x) b1[i] = w.b1[i];
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  __syncthreads();
  const bool cv = (C >= 16) || (4 * g < C);           // this lane's channel quad exists (C = 8: lane groups 0, 1)
  const int spw = N / (16 * S), nstrip = B * spw;      // strips per window
  constexpr float invC = 1.0f / C;
  // out[mo] (+)= W[16 mo + r][K block kb] x the accumulator tile `bt`  (K index of MFMA q: 16 kb + 4 g + q)
  auto mma_block = [&](const float* W, int ld, int mo, int kb, f32x4 bt, f32x4 acc) -> f32x4 {
    const float4 wa = *reinterpret_cast<const float4*>(W + (16 * mo + r) * ld + 16 * kb + 4 * g);
    acc = mfma4(wa.x, bt[0], acc); acc = mfma4(wa.y, bt[1], acc); acc = mfma4(wa.z, bt[2], acc); acc = mfma4(wa.w, bt[3], acc);
    return acc;
  };
  auto vec4 = [&](const float* v, int tile) -> f32x4 {   // entries 16 tile + 4 g .. + 3 of a padded vector
    const float4 t = *reinterpret_cast<const float4*>(v + 16 * tile + 4 * g);
    return f32x4{t.x, t.y, t.z, t.w};
  };
  for (int strip = blockIdx.x * 4 + wave; strip < nstrip; strip += gridDim.x * 4) {
    const int win = strip / spw, t0 = (strip - win * spw) * 16 * S;
    const size_t wo = (size_t)win * N * C;
    const float* xw = x + wo; const float* ow = o_hm + wo;
    // ---- the two edge tokens of the neighbouring strips on the vector ALU (lanes 0 .. C-1: token t0 - 1, lanes 32 .. 32 + C-1:
    //      token t0 + 64): a0 = GELU(u[:, 0]) for the local-enhancement conv; 0 outside the window
    float hl = 0.f, hr = 0.f;
    if (le) {
      const int c = lane & 31, side = lane >> 5;
      const int th = side ? t0 + 16 * S : t0 - 1;
      const bool tin = th >= 0 && th < N, lv = c < C;
      const int thc = tin ? th : 0;
      scr[lane] = lv ? ow[((size_t)(c >> 2) * N + thc) * 4 + (c & 3)] : 0.f;
      float a = lv ? xw[(size_t)thc * C + c] + bp[c] : 0.f;
      const float* wr = Wp + (lv ? c : 0) * LDC;
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const float4 ov = *reinterpret_cast<const float4*>(scr + side * 32 + k);
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
        a = fmaf(wv.x, ov.x, a); a = fmaf(wv.y, ov.y, a); a = fmaf(wv.z, ov.z, a); a = fmaf(wv.w, ov.w, a);
      }
      a = lv ? a : 0.f;
      const float mean = group_sum<32>(a) * invC;
      const float d = lv ? a - mean : 0.f;
      const float rstd = 1.0f / sqrtf(group_sum<32>(d * d) * invC + 1e-5f);
      const float gg = lv ? (d * rstd * g2[c] + be2[c]) * W1[c] : 0.f;          // W1 row 0
      const float u0 = group_sum<32>(gg) + b1[0];
      const float a0 = tin ? gelu_f(u0) : 0.f;
      hl = lane_value(a0, 0); hr = lane_value(a0, 32);
    }
    // ---- pass 1: per tile  x1 = x + o Wp^T + bp,  g = LN2(x1),  a0 = GELU(u[:, 0])
    f32x4 x1t[S][MT], gt[S][MT];
    float a0r[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int tok = t0 + 16 * s + r;
      f32x4 ob[MT], acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 ov = cv ? *reinterpret_cast<const float4*>(ow + ((size_t)(4 * m + g) * N + tok) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 xv = cv ? *reinterpret_cast<const float4*>(xw + (size_t)tok * C + 16 * m + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        ob[m] = f32x4{ov.x, ov.y, ov.z, ov.w};
        acc[m] = f32x4{xv.x, xv.y, xv.z, xv.w} + vec4(bp, m);
      }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo)
#pragma unroll
        for (int kb = 0; kb < MT; ++kb) acc[mo] = mma_block(Wp, LDC, mo, kb, ob[kb], acc[mo]);
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        x1t[s][m] = acc[m];
        if (x1_out && cv) *reinterpret_cast<float4*>(x1_out + wo + (size_t)tok * C + 16 * m + 4 * g) = tofloat4(acc[m]);
        sum += (acc[m][0] + acc[m][1]) + (acc[m][2] + acc[m][3]);
      }
      const float mean = rows_sum(sum) * invC;
      float var = 0.f;
      f32x4 d[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        d[m] = cv ? acc[m] - mean : f32x4{0.f, 0.f, 0.f, 0.f};
        var += (d[m][0] * d[m][0] + d[m][1] * d[m][1]) + (d[m][2] * d[m][2] + d[m][3] * d[m][3]);
      }
      const float rstd = 1.0f / sqrtf(rows_sum(var) * invC + 1e-5f);
      float u0 = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        gt[s][m] = d[m] * rstd * vec4(g2, m) + vec4(be2, m);
        const f32x4 w10 = vec4(W1, m);                   // row 0 of W1 (padded columns are zero)
        u0 += (gt[s][m][0] * w10[0] + gt[s][m][1] * w10[1]) + (gt[s][m][2] * w10[2] + gt[s][m][3] * w10[3]);
      }
      a0r[s] = le ? gelu_f(rows_sum(u0) + b1[0]) : 0.f;
    }
    // ---- pass 2: per tile  hidden = GELU chain(g W1^T + b1),  x2 = x1 + hidden W2^T + b2
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int tok = t0 + 16 * s + r;
      f32x4 out[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) out[m] = x1t[s][m] + vec4(b2, m);
      float c0 = 0.f;
      if (le) {
        const float left = s == 0 ? hl : lane_value(a0r[s > 0 ? s - 1 : 0], 15);
        const float right = s == S - 1 ? hr : lane_value(a0r[s < S - 1 ? s + 1 : S - 1], 0);
        const float sm1 = dpp_shift<0x111>(a0r[s]), sp1 = dpp_shift<0x101>(a0r[s]);   // row_shr:1 / row_shl:1
        const float am = r == 0 ? left : sm1;                // a0 of token r - 1
        const float ap = r == 15 ? right : sp1;              // a0 of token r + 1
        c0 = gelu_f(lw0 * am + lw1 * a0r[s] + lw2 * ap);
      }
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        f32x4 h = vec4(b1, ht);
#pragma unroll
        for (int kb = 0; kb < MT; ++kb) h = mma_block(W1, LDC, ht, kb, gt[s][kb], h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = gelu_f(h[q]);
          if (le) a = (ht == 0 && q == 0 && g == 0) ? c0 : gelu_f(a);
          h[q] = a;
        }
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) out[mo] = mma_block(W2, LDH, mo, ht, h, out[mo]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (cv) *reinterpret_cast<float4*>(x2_out + wo + (size_t)tok * C + 16 * m + 4 * g) = tofloat4(out[m]);
    }
  }
}

// ---------------------------------------------------------------------------------
// The same strip kernel with every product on the f16 matrix cores (fp16-pair operands, ral_device.hpp): the fp32 MFMA is
// what the narrow levels pay for (41.7 cycles per v_mfma_f32_16x16x4_f32 beside vector work, and it blocks the vector ALU;
// v_mfma_f32_16x16x16_f16: 10.6, overlapping).  An accumulator tile D[channel 4g+q][token r] is the B operand of
// v_mfma_f32_16x16x16_f16 as it stands (lane (r, g) holds K = 4g .. 4g+3 of column r): it is split into its two fp16 pieces
// in registers - ONCE per tile, for all the output tiles it feeds - and a 16-channel block of a product is three MFMAs
// (h1 h1 into acc; h2 h1 and h1 h2, residuals scaled by 2^11, into accx; result acc + 2^-11 accx) instead of four fp32 ones.
//   * weights: split when the (persistent) workgroup stages them - every matrix multiplied by its own power of two first
//     (largest magnitude into [2^13, 2^14)), the inverse applied to the accumulators - as planes of K-contiguous rows at a
//     stride of K + 8 halves (the 8-byte fragment reads of lanes 0-31 then cover all 64 banks);
//   * activations (attention output, LayerNorm output, GELU outputs) are not scaled: 22 bits for magnitudes in
//     [6e-5, 65504] and - the residual piece carries 2^11 - the same down to ~3e-8, non-finite beyond 65504 (nothing clamped,
//     as in the wide-level kernels).  tests/test_gpu_configs.py::test_small_hidden_activations_keep_the_fp32_tolerance drives
//     every hidden value of a block to 1e-4 .. 5e-7 (fc1 x 1e-4 / 1e-6, fc2 x 1e+4 / 1e+6) at the ordinary tolerances;
//   * the local-enhancement edge tokens stay on the vector ALU in fp32 (fp32 copies of Wp and of row 0 of W1 for them).
template <int C>
struct MlpwhShape {
  static constexpr int KP = C < 16 ? 16 : C, MT = KP / 16, HID = 4 * C, HT = HID / 16, S = C >= 32 ? 1 : 4;   // (C = 32: two tiles of state spilled)
  static constexpr int LDC = KP + 4;                           // fp32 Wp rows (edge tokens)
  static constexpr int LDA = KP + 8, LDB = HID + 8;            // halves: rows of K = C matrices / of W2 (K = 4C)
  // floats: Wp fp32 [KP][LDC] | w10 [KP] | bp, g2, be2, b2 [KP each] | b1 [HID] | unscale[4] | max bits[4] | scratch 4 x 64
  static constexpr int W10O = KP * LDC, VO = W10O + KP, B1O = VO + 4 * KP, UNO = B1O + HID, SCR = UNO + 8, FTOT = SCR + 4 * 64;
  // halves behind the floats: Wp planes [2][KP][LDA] | W1 planes [2][HID][LDA] | W2 planes [2][KP][LDB]
  static constexpr int HWP = 0, HW1 = HWP + 2 * KP * LDA, HW2 = HW1 + 2 * HID * LDA, HTOT = HW2 + 2 * KP * LDB;
  static constexpr size_t BYTES = (size_t)FTOT * 4 + (size_t)HTOT * 2;
};
RAL_DEV H2x4 split4s(f32x4 x) {   // pieces with the residual scaled by 2^11 (f16_split2)
  H2x4 r;
  const H2 s0 = f16_split2(x[0]), s1 = f16_split2(x[1]), s2 = f16_split2(x[2]), s3 = f16_split2(x[3]);
  r.a = h16x4{s0.a, s1.a, s2.a, s3.a}; r.b = h16x4{s0.b, s1.b, s2.b, s3.b};
  return r;
}
// fp32 matrix (rows x cols, row stride src_ld... dense) -> two planes of rows x ld halves, scaled by `scale`
RAL_DEV void stage_planes(const float* __restrict__ W, int rows, int cols, float scale, _Float16* dst, int ld, int plane) {
  for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
    const int m = i / cols, k = i - m * cols;
    const H2 h = f16_split2(W[i] * scale);
    dst[m * ld + k] = h.a; dst[plane + m * ld + k] = h.b;
  }
}

template <int C>
__global__ __launch_bounds__(256, RAL_MLPW_WPE) void k_mlp_fwd_wh(const float* __restrict__ x, const float* __restrict__ o_hm,
                                                                  BlockP w, float* __restrict__ x1_out, float* __restrict__ x2_out,
                                                                  int N, int B) {
  using SH = MlpwhShape<C>;
  constexpr int KP = SH::KP, MT = SH::MT, HID = SH::HID, HT = SH::HT, LDC = SH::LDC, LDA = SH::LDA, LDB = SH::LDB, S = SH::S;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  float* Wp = sm; float* w10 = sm + SH::W10O;
  float* bp = sm + SH::VO; float* g2 = bp + KP; float* be2 = g2 + KP; float* b2 = be2 + KP; float* b1 = sm + SH::B1O;
  float* uns = sm + SH::UNO; unsigned* mxb = reinterpret_cast<unsigned*>(uns + 4);
  _Float16* hb = reinterpret_cast<_Float16*>(sm + SH::FTOT);
  _Float16* WpH = hb + SH::HWP; _Float16* W1H = hb + SH::HW1; _Float16* W2H = hb + SH::HW2;
  constexpr int PWP = KP * LDA, PW1 = HID * LDA, PW2 = KP * LDB;   // plane strides
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* scr = sm + SH::SCR + wave * 64;
  // ---- the block's weights -> LDS (once per workgroup): zero everything, matrix maxima, planes
  for (int i = threadIdx.x; i < (int)(SH::BYTES / 4); i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) m0 = fmaxf(m0, fabsf(w.wp[i]));
    for (int i = threadIdx.x; i < HID * C; i += blockDim.x) { m1 = fmaxf(m1, fabsf(w.w1[i])); m2 = fmaxf(m2, fabsf(w.w2[i])); }
    m0 = group_max<64>(m0); m1 = group_max<64>(m1); m2 = group_max<64>(m2);
    if (lane == 0) { atomicMax(mxb, __float_as_uint(m0)); atomicMax(mxb + 1, __float_as_uint(m1)); atomicMax(mxb + 2, __float_as_uint(m2)); }
  }
  __syncthreads();
  const float sp = h2_row_scale(mxb[0]), s1 = h2_row_scale(mxb[1]), s2 = h2_row_scale(mxb[2]);
  // the three activation operands (attention output, LayerNorm output, hidden tile) times the block's powers of two (ASC_*,
  // ral_device.hpp): o and the hidden tile when they are split, the LayerNorm output through its affine (g2, be2 carry the scale,
  // row 0 of W1 for the local-enhancement channel - w10 - its inverse: gv * w10 is unchanged), the inverses in the unscales
  const float so = asc_get(w.asc, ASC_O), sg = asc_get(w.asc, ASC_LN2), sgi = asc_get(w.asc, ASC_LN2_INV), sh = asc_get(w.asc, ASC_HID);
  const float unp = h2_row_unscale(mxb[0]) * asc_get(w.asc, ASC_O_INV), un1 = h2_row_unscale(mxb[1]) * sgi,
              un2 = h2_row_unscale(mxb[2]) * asc_get(w.asc, ASC_HID_INV);
  stage_planes(w.wp, C, C, sp, WpH, LDA, PWP);
  stage_planes(w.w1, HID, C, s1, W1H, LDA, PW1);
  stage_planes(w.w2, C, HID, s2, W2H, LDB, PW2);
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) Wp[(i / C) * LDC + i % C] = w.wp[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) { w10[i] = w.w1[i] * sgi; bp[i] = w.bp[i]; g2[i] = w.ln2w[i] * sg; be2[i] = w.ln2b[i] * sg; b2[i] = w.b2[i]; }
  for (int i = threadIdx.x; i < HID; i += blockDim.x) b1[i] = w.b1[i];
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  __syncthreads();
  const bool cv = (C >= 16) || (4 * g < C);
  const int spw = N / (16 * S), nstrip = B * spw;
  constexpr float invC = 1.0f / C;
  // (acc, accx)[mo] += W[16 mo + r][K block kb] x the split tile `bt`
  auto mma_h = [&](const _Float16* Wh, int plane, int ld, int mo, int kb, const H2x4& bt, f32x4& acc, f32x4& accx) {
    const _Float16* p = Wh + (16 * mo + r) * ld + 16 * kb + 4 * g;
    const h16x4 a1 = *reinterpret_cast<const h16x4*>(p), a2 = *reinterpret_cast<const h16x4*>(p + plane);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, bt.a, acc, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, bt.a, accx, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, bt.b, accx, 0, 0, 0);
  };
  // C = 32: the K = 2 x 16 channels of a row in ONE v_mfma_f32_16x16x32_f16 (the issue cycles of the K = 16 instruction): the lane's
  // k slots 8 g .. 8 g + 7 are channels 4 g .. 4 g + 3 and 16 + 4 g .. + 3 on both operands
  auto mma_h32 = [&](const _Float16* Wh, int plane, int ld, int mo, const H2x4& b0, const H2x4& b1_, f32x4& acc, f32x4& accx) {
    const _Float16* p = Wh + (16 * mo + r) * ld + 4 * g;
    const h16x4 a10 = *reinterpret_cast<const h16x4*>(p), a11 = *reinterpret_cast<const h16x4*>(p + 16);
    const h16x4 a20 = *reinterpret_cast<const h16x4*>(p + plane), a21 = *reinterpret_cast<const h16x4*>(p + plane + 16);
    const f16x8 a1 = __builtin_shufflevector(a10, a11, 0, 1, 2, 3, 4, 5, 6, 7), a2 = __builtin_shufflevector(a20, a21, 0, 1, 2, 3, 4, 5, 6, 7);
    const f16x8 p1 = __builtin_shufflevector(b0.a, b1_.a, 0, 1, 2, 3, 4, 5, 6, 7), p2 = __builtin_shufflevector(b0.b, b1_.b, 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, p1, acc, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, p1, accx, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, p2, accx, 0, 0, 0);
  };
  auto vec4 = [&](const float* v, int tile) -> f32x4 {
    const float4 t = *reinterpret_cast<const float4*>(v + 16 * tile + 4 * g);
    return f32x4{t.x, t.y, t.z, t.w};
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  for (int strip = blockIdx.x * 4 + wave; strip < nstrip; strip += gridDim.x * 4) {
    const int win = strip / spw, t0 = (strip - win * spw) * 16 * S;
    const size_t wo = (size_t)win * N * C;
    const float* xw = x + wo; const float* ow = o_hm + wo;
    float hl = 0.f, hr = 0.f;
    if (le) {   // the two edge tokens of the neighbouring strips, fp32 on the vector ALU (see k_mlp_fwd_w)
      const int c = lane & 31, side = lane >> 5;
      const int th = side ? t0 + 16 * S : t0 - 1;
      const bool tin = th >= 0 && th < N, lv = c < C;
      const int thc = tin ? th : 0;
      scr[lane] = lv ? ow[((size_t)(c >> 2) * N + thc) * 4 + (c & 3)] : 0.f;
      float a = lv ? xw[(size_t)thc * C + c] + bp[c] : 0.f;
      const float* wr = Wp + (lv ? c : 0) * LDC;
#pragma unroll
      for (int k = 0; k < KP; k += 4) {
        const float4 ov = *reinterpret_cast<const float4*>(scr + side * 32 + k);
        const float4 wv = *reinterpret_cast<const float4*>(wr + k);
        a = fmaf(wv.x, ov.x, a); a = fmaf(wv.y, ov.y, a); a = fmaf(wv.z, ov.z, a); a = fmaf(wv.w, ov.w, a);
      }
      a = lv ? a : 0.f;
      const float mean = group_sum<32>(a) * invC;
      const float d = lv ? a - mean : 0.f;
      const float rstd = 1.0f / sqrtf(group_sum<32>(d * d) * invC + 1e-5f);
      const float gg = lv ? (d * rstd * g2[c] + be2[c]) * w10[c] : 0.f;
      const float u0 = group_sum<32>(gg) + b1[0];
      const float a0 = tin ? gelu_f(u0) : 0.f;
      hl = lane_value(a0, 0); hr = lane_value(a0, 32);
    }
    // ---- pass 1: per tile  x1 = x + o Wp^T + bp,  g = LN2(x1) (kept as its two pieces),  a0 = GELU(u[:, 0])
    f32x4 x1t[S][MT];
    H2x4 gth[S][MT];
    float a0r[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int tok = t0 + 16 * s + r;
      H2x4 ob[MT];
      f32x4 xin[MT], acc[MT], accx[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float4 ov = cv ? *reinterpret_cast<const float4*>(ow + ((size_t)(4 * m + g) * N + tok) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 xv = cv ? *reinterpret_cast<const float4*>(xw + (size_t)tok * C + 16 * m + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        ob[m] = split4s(f32x4{ov.x, ov.y, ov.z, ov.w} * so);
        xin[m] = f32x4{xv.x, xv.y, xv.z, xv.w} + vec4(bp, m);
        acc[m] = zero4; accx[m] = zero4;
      }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) {
        if constexpr (MT == 2) mma_h32(WpH, PWP, LDA, mo, ob[0], ob[MT - 1], acc[mo], accx[mo]);
        else mma_h(WpH, PWP, LDA, mo, 0, ob[0], acc[mo], accx[mo]);
      }
      float sum = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m] = xin[m] + (acc[m] + accx[m] * RAL_H2_SCALE) * unp;
        x1t[s][m] = acc[m];
        if (x1_out && cv) *reinterpret_cast<float4*>(x1_out + wo + (size_t)tok * C + 16 * m + 4 * g) = tofloat4(acc[m]);
        sum += (acc[m][0] + acc[m][1]) + (acc[m][2] + acc[m][3]);
      }
      const float mean = rows_sum(sum) * invC;
      float var = 0.f;
      f32x4 d[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        d[m] = cv ? acc[m] - mean : zero4;
        var += (d[m][0] * d[m][0] + d[m][1] * d[m][1]) + (d[m][2] * d[m][2] + d[m][3] * d[m][3]);
      }
      const float rstd = 1.0f / sqrtf(rows_sum(var) * invC + 1e-5f);
      float u0 = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 gv = d[m] * rstd * vec4(g2, m) + vec4(be2, m);
        const f32x4 wr0 = vec4(w10, m);                  // row 0 of W1 (padded columns are zero)
        u0 += (gv[0] * wr0[0] + gv[1] * wr0[1]) + (gv[2] * wr0[2] + gv[3] * wr0[3]);
        gth[s][m] = split4s(gv);
      }
      a0r[s] = le ? gelu_f(rows_sum(u0) + b1[0]) : 0.f;
    }
    // ---- pass 2: per tile  hidden = GELU chain(g W1^T + b1),  x2 = x1 + hidden W2^T + b2
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int tok = t0 + 16 * s + r;
      f32x4 out[MT], outx[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) { out[m] = zero4; outx[m] = zero4; }
      float c0 = 0.f;
      if (le) {
        const float left = s == 0 ? hl : lane_value(a0r[s > 0 ? s - 1 : 0], 15);
        const float right = s == S - 1 ? hr : lane_value(a0r[s < S - 1 ? s + 1 : S - 1], 0);
        const float sm1 = dpp_shift<0x111>(a0r[s]), sp1 = dpp_shift<0x101>(a0r[s]);   // row_shr:1 / row_shl:1
        const float am = r == 0 ? left : sm1;
        const float ap = r == 15 ? right : sp1;
        c0 = gelu_f(lw0 * am + lw1 * a0r[s] + lw2 * ap);
      }
#pragma unroll
      for (int ht = 0; ht < HT; ++ht) {
        f32x4 h = zero4, hx = zero4;
        if constexpr (MT == 2) mma_h32(W1H, PW1, LDA, ht, gth[s][0], gth[s][MT - 1], h, hx);
        else mma_h(W1H, PW1, LDA, ht, 0, gth[s][0], h, hx);
        h = (h + hx * RAL_H2_SCALE) * un1 + vec4(b1, ht);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float a = gelu_f(h[q]);
          if (le) a = (ht == 0 && q == 0 && g == 0) ? c0 : gelu_f(a);
          h[q] = a;
        }
        const H2x4 hh = split4s(h * sh);
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) mma_h(W2H, PW2, LDB, mo, ht, hh, out[mo], outx[mo]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (cv) *reinterpret_cast<float4*>(x2_out + wo + (size_t)tok * C + 16 * m + 4 * g) =
            tofloat4(x1t[s][m] + vec4(b2, m) + (out[m] + outx[m] * RAL_H2_SCALE) * un2);
    }
  }
}

// Measured at batch 2048 (rocprofv3, serialised step, us per launch: this kernel / k_mlp_fwd): C = 16 (N = 256): 67.6 / 78.7;
// C = 8 (N = 512): 73.9 / 65.5 - half of every padded MFMA tile is zeros there, k_mlp_fwd has a K = 8 path; C = 32
// (N = 128): 107 / 92 - two tiles of state per strip already spill (284 bytes per lane at 168 registers).  Both forms sit
// at about half their issue bound (C = 16: 36 fp32 MFMAs + 32 GELU evaluations per lane and tile = ~2 400 cycles, measured
// 4 200 / 4 900): what the narrow levels pay for is the fp32 MFMA itself, not the barriers.  Default: C = 16 only
// (RAL_MLP_FWD_W=2: all three widths, 0: never).
// which narrow-level forward: 0 = k_mlp_fwd (ral_fwd.hip), 1 = k_mlp_fwd_w (fp32 MFMA), 2 = k_mlp_fwd_wh (f16 matrix cores;
// only when the model allows fp16-pair products, f16_split > 0).  RAL_MLP_FWD_W: 0 never a strip kernel, 1 fp32 strips
// at C = 16 only, 2 fp32 strips at every width, 3 (default) f16 strips at every width where allowed, else fp32 strips at
// C = 16.
// Measured at batch 2048 (rocprofv3, serialised step, us per launch; k_mlp_fwd / fp32 strips / f16 strips), first with the
// two-transcendental GELU: C = 16: 78.7 / 68.5 / 62.1; C = 8: 66.0 / 73.9 / 71.4; C = 32: 93.4 / 107 / 137 (two tiles of state
// per strip spilled at 168 registers); then with the one-exponential forward GELU and one-tile strips at C = 32 (104
// registers): C = 16: - / - / 51.3; C = 8: 64.1 / - / 62.1; C = 32: 86.5 / - / 71.2.  The f16 form removes the matrix share
// (36 fp32 MFMAs = 1 500 of ~4 200 cycles per tile at C = 16 -> 27 f16 ones that overlap); what is left is the 32 GELU
// evaluations per lane and tile.
int mlp_fwd_w_kind(int C, int N, bool want_upre, bool f16_ok) {
  static const int on = (int)ral_knob("MLP_FWD_W", 3);
  if (!on || want_upre || N % 64 != 0 || !(C == 8 || C == 16 || C == 32)) return 0;   // (64: a whole number of strips at every width)
  if (on >= 3 && f16_ok) return 2;
  if (on == 2) return 1;
  return C == 16 ? 1 : 0;
}
bool mlp_fwd_w_takes(int C, int N, bool want_upre) { return mlp_fwd_w_kind(C, N, want_upre, false) != 0; }

template <int C>
static void go_mlp_fwd_w(const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s) {
  const size_t lds = (size_t)MlpwShape<C>::TOTAL * sizeof(float);
  RAL_SET_LDS((k_mlp_fwd_w<C>), lds);
  static const int genv = (int)ral_knob("GRID_MLPW", 0);
  static int occ = 0;
  if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mlp_fwd_w<C>, 256, lds) != hipSuccess || occ < 1)) occ = 3;
  const int nwg = (B * (N / (16 * MlpwShape<C>::S)) + 3) / 4;
  int grid = genv > 0 ? genv : ral_num_cus() * (occ > 4 ? 4 : occ);
  if (grid > nwg) grid = nwg;
  k_mlp_fwd_w<C><<<grid, 256, lds, s>>>(x, o, w, x1, x2, N, B);
}
template <int C>
static void go_mlp_fwd_wh(const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s) {
  const size_t lds = MlpwhShape<C>::BYTES;
  RAL_SET_LDS((k_mlp_fwd_wh<C>), lds);
  static const int genv = (int)ral_knob("GRID_MLPW", 0);
  static int occ = 0;
  if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mlp_fwd_wh<C>, 256, lds) != hipSuccess || occ < 1)) occ = 3;
  const int nwg = (B * (N / (16 * MlpwhShape<C>::S)) + 3) / 4;
  int grid = genv > 0 ? genv : ral_num_cus() * (occ > 4 ? 4 : occ);
  if (grid > nwg) grid = nwg;
  k_mlp_fwd_wh<C><<<grid, 256, lds, s>>>(x, o, w, x1, x2, N, B);
}
void launch_mlp_fwd_w(int C, int kind, const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s) {
  if (kind == 2) {
    if (C == 8) go_mlp_fwd_wh<8>(x, o, w, x1, x2, N, B, s);
    else if (C == 16) go_mlp_fwd_wh<16>(x, o, w, x1, x2, N, B, s);
    else go_mlp_fwd_wh<32>(x, o, w, x1, x2, N, B, s);
    return;
  }
  if (C == 8) go_mlp_fwd_w<8>(x, o, w, x1, x2, N, B, s);
  else if (C == 16) go_mlp_fwd_w<16>(x, o, w, x1, x2, N, B, s);
  else go_mlp_fwd_w<32>(x, o, w, x1, x2, N, B, s);
}

// =================================================================================
// Narrow levels, BACKWARD of the same half block as a strip kernel (C = 8, 16): replaces k_mlp_bwd_s (ral_bwd.hip), which
// walks a window through ~22 workgroup barriers and as many L2 round trips (wave_parked 0.59).  A WAVE carries 16 tokens:
//   u = W1 LN2(x1) + b1 (re-computed), da2 = W2^T dx2, du = da2 GELU-chain'(u), dg = W1^T du, dx1 = dx2 + LN2bwd(dg),
//   do = Wp^T dx1 - all of them products over CHANNELS, chained through the accumulators as in the forward (an accumulator
//   tile D[channel 4g+q][token r] is the next product's B operand);
//   the weight gradients dW1 = du LN2(x1)^T, dW2 = dx2 a2^T contract over TOKENS, i.e. over the lane column of those tiles:
//   each operand goes through a per-wave LDS tile once (16-byte write of the tile as [token][channel], four 4-byte reads
//   of [channel r][tokens 4g .. 4g+3]) and the products accumulate in registers over all the tiles of the wave
//   (8 C^2 / 64 registers: 32 at C = 16);
//   the local-enhancement conv couples neighbouring tokens through hidden channel 0 only: u[:, 0] of two tokens and
//   da2[:, 0] of one token on each side of the tile are re-computed on the vector ALU (lane group g = halo slot, lane r =
//   channel: a LayerNorm and two dot products of C terms), inside the tile the neighbours come from DPP row shifts;
//   small gradients (b1, b2, LayerNorm, conv taps) are per-lane sums reduced once per kernel; the workgroup's weight-
//   gradient tiles meet in the LDS (the weight copies are no longer needed then) and leave with one atomic per element.
// fp32 MFMA and fp32 vector ALU throughout (this form); weights staged once per persistent workgroup.
// =================================================================================
// The flush at the end of the strip-backward kernels - every workgroup adds its sums to the gradient buffer with atomics, a
// chain of (workgroups) same-address atomics per element - measures 20 - 27 us when it is all a launch does (make VARIANT=..
// EXTRA=-DRAL_DIAG_SKIP=1 against 3) but 7 - 10 us at the end of a real launch, where the workgroups do not arrive together.
// One row of sums per workgroup in scratch memory and a kernel that adds the rows up (built, parity-clean, removed): the strip
// kernels 4 - 10 us shorter, the extra kernel 11 us (~4 with more row groups): nothing gained.
template <int C>
struct MlpbwShape {
  static constexpr int KP = C < 16 ? 16 : C, MT = KP / 16, HID = 4 * C, HT = HID / 16;
  static constexpr int LDC = KP + 4, LDH = HID + 4, LDT = 20;            // LDT: row stride of a 16 x 16 transpose tile
  static constexpr int NTT = 2 * MT + 2;                                   // transpose tiles per wave: LN2(x1), dx2 (MT each), du, a2
  // floats: W1 [HID][LDC] | W2T [HID][LDC] | W1T [KP][LDH] | WpT [KP][LDC] | g2, be2, w10, w2c0 [KP each] | b1 [HID] |
  //         per wave: NTT x 16 x LDT transpose tiles
  static constexpr int W2TO = HID * LDC, W1TO = W2TO + HID * LDC, WPTO = W1TO + KP * LDH, VO = WPTO + KP * LDC, B1O = VO + 4 * KP,
                       SCR = B1O + HID, WSCR = NTT * 16 * LDT, TOTAL = SCR + 4 * WSCR;
  static_assert(TOTAL >= 16 * C * C + 2 * HID + 4 * KP + 8, "the flush staging (weight-gradient tiles in doubles) fits the kernel's LDS");
};

template <int C>
__global__ __launch_bounds__(256, RAL_MLPW_WPE) void k_mlp_bwd_w(const float* __restrict__ dx2, const float* __restrict__ x1,
                                                                 BlockP w, BlockP gr, float* __restrict__ dx1_out,
                                                                 float* __restrict__ do_hm, int N, int B, int want_dw) {
  using SH = MlpbwShape<C>;
  constexpr int KP = SH::KP, MT = SH::MT, HID = SH::HID, HT = SH::HT, LDC = SH::LDC, LDH = SH::LDH, LDT = SH::LDT;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  float* W1 = sm; float* W2T = sm + SH::W2TO; float* W1T = sm + SH::W1TO; float* WpT = sm + SH::WPTO;
  float* g2 = sm + SH::VO; float* be2 = g2 + KP; float* w10 = be2 + KP; float* w2c0 = w10 + KP; float* b1 = sm + SH::B1O;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* TT = sm + SH::SCR + wave * SH::WSCR;          // this wave's transpose tiles
  // ---- weights -> LDS (once per workgroup), zero-padded to whole tiles
  for (int i = threadIdx.x; i < SH::SCR; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < HID * C; i += blockDim.x) {
    const int h = i / C, c = i - h * C;
    const float a = w.w1[i];                         // W1[h][c]
    W1[h * LDC + c] = a; W1T[c * LDH + h] = a;
    W2T[h * LDC + c] = w.w2[c * HID + h];            // W2[c][h]
  }
  for (int i = threadIdx.x; i < C * C; i += blockDim.x) { const int c = i / C, j = i - c * C; WpT[j * LDC + c] = w.wp[i]; }   // Wp[c][j]
  for (int i = threadIdx.x; i < C; i += blockDim.x) { g2[i] = w.ln2w[i]; be2[i] = w.ln2b[i]; w10[i] = w.w1[i]; w2c0[i] = w.w2[i * HID]; }
  for (int i = threadIdx.x; i < HID; i += blockDim.x) b1[i] = w.b1[i];
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  __syncthreads();
  const bool cv = (C >= 16) || (4 * g < C);            // this lane's channel quad exists (C = 8: lane groups 0, 1)
  constexpr float invC = 1.0f / C;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto mma_block = [&](const float* W, int ld, int mo, int kb, f32x4 bt, f32x4 acc) -> f32x4 {
    const float4 wa = *reinterpret_cast<const float4*>(W + (16 * mo + r) * ld + 16 * kb + 4 * g);
    acc = mfma4(wa.x, bt[0], acc); acc = mfma4(wa.y, bt[1], acc); acc = mfma4(wa.z, bt[2], acc); acc = mfma4(wa.w, bt[3], acc);
    return acc;
  };
  auto vec4 = [&](const float* v, int tile) -> f32x4 {
    const float4 t = *reinterpret_cast<const float4*>(v + 16 * tile + 4 * g);
    return f32x4{t.x, t.y, t.z, t.w};
  };
  // transposed fragment of a tile written as [token][channel]: X[channel r][tokens 4g .. 4g+3]
  auto tr_read = [&](const float* T) -> f32x4 {
    return f32x4{T[(4 * g + 0) * LDT + r], T[(4 * g + 1) * LDT + r], T[(4 * g + 2) * LDT + r], T[(4 * g + 3) * LDT + r]};
  };
  // accumulators that live over all the tiles of the wave
  f32x4 dW1[HT][MT], dW2[MT][HT], sb1[HT], sb2[MT], sgam[MT], sbet[MT];
#pragma unroll
  for (int h = 0; h < HT; ++h) { sb1[h] = zero4;
#pragma unroll
    for (int m = 0; m < MT; ++m) { dW1[h][m] = zero4; dW2[m][h] = zero4; } }
#pragma unroll
  for (int m = 0; m < MT; ++m) { sb2[m] = zero4; sgam[m] = zero4; sbet[m] = zero4; }
  float gle0 = 0.f, gle1 = 0.f, gle2 = 0.f;
  const int tpw = N >> 4, ntile = B * tpw;
  // The operands of a tile - 16 tokens of x1 and dx2 and, for the local enhancement, one channel of each of the four halo
  // tokens per lane - are REQUESTED one tile ahead: a wave has nothing else in flight while it waits, and with three waves
  // per SIMD a round trip per 16 tokens was most of what the tile cost.
  float4 pxa[MT], pda[MT];
  constexpr int CPL = C > 16 ? C / 16 : 1;          // halo channels per lane (a halo token is spread over the 16 lanes of a row)
  float pxh[CPL], pdh[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) { pxh[c] = 0.f; pdh[c] = 0.f; }
  auto request = [&](int tl, float4 (&xa)[MT], float4 (&da)[MT], float (&hx)[CPL], float (&hd)[CPL]) {
    const int tlc = tl < ntile ? tl : ntile - 1;                      // (past the end: a valid address, never used)
    const int win_ = tlc / tpw, t0_ = (tlc - win_ * tpw) << 4;
    const float* xw_ = x1 + (size_t)win_ * N * C; const float* dw_ = dx2 + (size_t)win_ * N * C;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int off = (t0_ + r) * C + (cv ? 16 * m + 4 * g : 0);
      xa[m] = *reinterpret_cast<const float4*>(xw_ + off);
      da[m] = *reinterpret_cast<const float4*>(dw_ + off);
    }
    if (le) {
      const int th = g < 2 ? t0_ - 2 + g : t0_ + 14 + g;
      const int thc = (th >= 0 && th < N) ? th : 0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int rc = r + 16 * c < C ? r + 16 * c : 0;
        hx[c] = xw_[(size_t)thc * C + rc]; hd[c] = dw_[(size_t)thc * C + rc];
      }
    }
  };
  request(blockIdx.x * 4 + wave, pxa, pda, pxh, pdh);
#ifndef RAL_DIAG_SKIP
#define RAL_DIAG_SKIP 0
#endif
  for (int tile = blockIdx.x * 4 + wave; tile < ((RAL_DIAG_SKIP & 1) ? 0 : ntile); tile += gridDim.x * 4) {
    const int win = tile / tpw, t0 = (tile - win * tpw) << 4, tok = t0 + r;
    const size_t wo = (size_t)win * N * C;
    f32x4 xv[MT], dv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xv[m] = cv ? f32x4{pxa[m].x, pxa[m].y, pxa[m].z, pxa[m].w} : zero4;
      dv[m] = cv ? f32x4{pda[m].x, pda[m].y, pda[m].z, pda[m].w} : zero4;
    }
    float cxh[CPL], cdh[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { cxh[c] = pxh[c]; cdh[c] = pdh[c]; }
    request(tile + gridDim.x * 4, pxa, pda, pxh, pdh);                 // the next tile's operands
    float hA0[4] = {0.f, 0.f, 0.f, 0.f}, hD[2] = {0.f, 0.f};   // GELU(u[:, 0]) of the halo tokens; da2[:, 0] of t0-1 and t0+16
    if (le) {   // halo tokens t0-2, t0-1, t0+16, t0+17: lane group g = slot, lane r = channel
      const int th = g < 2 ? t0 - 2 + g : t0 + 14 + g;
      const bool tin = th >= 0 && th < N;
      float xs = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) xs += (r + 16 * c < C) ? cxh[c] : 0.f;
      const float mean = group_sum<16>(xs) * invC;
      float dd[CPL], vs = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) { dd[c] = (r + 16 * c < C) ? cxh[c] - mean : 0.f; vs += dd[c] * dd[c]; }
      const float rstd = 1.0f / sqrtf(group_sum<16>(vs) * invC + 1e-5f);
      float gg = 0.f, dh = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const bool lv = r + 16 * c < C;
        const int rc = lv ? r + 16 * c : 0;
        gg += lv ? (dd[c] * rstd * g2[rc] + be2[rc]) * w10[rc] : 0.f;
        dh += lv ? cdh[c] * w2c0[rc] : 0.f;
      }
      const float u0 = group_sum<16>(gg) + b1[0];
      const float a0 = tin ? gelu_f(u0) : 0.f;
      const float d0 = tin ? group_sum<16>(dh) : 0.f;
      hA0[0] = lane_value(a0, 0); hA0[1] = lane_value(a0, 16); hA0[2] = lane_value(a0, 32); hA0[3] = lane_value(a0, 48);
      hD[0] = lane_value(d0, 16); hD[1] = lane_value(d0, 32);
    }
    // ---- LN2 forward of the tile
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) sum += (xv[m][0] + xv[m][1]) + (xv[m][2] + xv[m][3]);
    const float mean = rows_sum(sum) * invC;
    float var = 0.f;
    f32x4 xh[MT], gx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xh[m] = cv ? xv[m] - mean : zero4;
      var += (xh[m][0] * xh[m][0] + xh[m][1] * xh[m][1]) + (xh[m][2] * xh[m][2] + xh[m][3] * xh[m][3]);
    }
    const float rstd = 1.0f / sqrtf(rows_sum(var) * invC + 1e-5f);
    float u0p = 0.f, d0p = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xh[m] = xh[m] * rstd;
      gx[m] = cv ? xh[m] * vec4(g2, m) + vec4(be2, m) : zero4;
      const f32x4 wr0 = vec4(w10, m), wc0 = vec4(w2c0, m);
      u0p += (gx[m][0] * wr0[0] + gx[m][1] * wr0[1]) + (gx[m][2] * wr0[2] + gx[m][3] * wr0[3]);
      d0p += (dv[m][0] * wc0[0] + dv[m][1] * wc0[1]) + (dv[m][2] * wc0[2] + dv[m][3] * wc0[3]);
    }
    // ---- local enhancement through hidden channel 0 (every lane of a token column computes the same numbers)
    float a2_0 = 0.f, du_0 = 0.f;
    if (le) {
      const float u0 = rows_sum(u0p) + b1[0], da0 = rows_sum(d0p);       // u[tok, 0], da2[tok, 0]
      float A0, dA;
      gelu_pair(u0, A0, dA);
      const float sm1 = dpp_shift<0x111>(A0), sp1 = dpp_shift<0x101>(A0);   // row_shr:1 / row_shl:1
      const float Am = r == 0 ? hA0[1] : sm1, Ap = r == 15 ? hA0[2] : sp1;   // A0 of tokens tok - 1, tok + 1
      float c0g, c0d;
      gelu_pair(lw0 * Am + lw1 * A0 + lw2 * Ap, c0g, c0d);
      a2_0 = c0g;
      const float DC = da0 * c0d;                                         // d loss / d conv output at tok
      // the conv outputs of the two tokens next to the tile (their gradient reaches A0 of the edge tokens)
      const float A_first = lane_value(A0, 0), A_last = lane_value(A0, 15);
      const float DCl = (t0 - 1 >= 0) ? hD[0] * gelu_grad_f(lw0 * hA0[0] + lw1 * hA0[1] + lw2 * A_first) : 0.f;
      const float DCr = (t0 + 16 < N) ? hD[1] * gelu_grad_f(lw0 * A_last + lw1 * hA0[2] + lw2 * hA0[3]) : 0.f;
      const float dm1 = dpp_shift<0x111>(DC), dp1 = dpp_shift<0x101>(DC);
      const float DCm = r == 0 ? DCl : dm1, DCp = r == 15 ? DCr : dp1;     // DC of tokens tok - 1, tok + 1
      du_0 = (lw0 * DCp + lw1 * DC + lw2 * DCm) * dA;
      if (g == 0) { gle0 += DC * Am; gle1 += DC * A0; gle2 += DC * Ap; }
    }
    // ---- operands of the weight-gradient products that do not depend on the hidden chunk: LN2(x1) and dx2, transposed
    f32x4 gxT[MT], dvT[MT];
    if (want_dw) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        *reinterpret_cast<float4*>(TT + (m * 16 + r) * LDT + 4 * g) = tofloat4(gx[m]);
        *reinterpret_cast<float4*>(TT + ((MT + m) * 16 + r) * LDT + 4 * g) = tofloat4(dv[m]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) { gxT[m] = tr_read(TT + m * 16 * LDT); dvT[m] = tr_read(TT + (MT + m) * 16 * LDT); sb2[m] += dv[m]; }
    }
    // ---- hidden chunks of 16 channels
    f32x4 dg[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) dg[m] = zero4;
    float* Tdu = TT + 2 * MT * 16 * LDT; float* Ta2 = Tdu + 16 * LDT;
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
      f32x4 u = vec4(b1, ht), da2 = zero4;
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) { u = mma_block(W1, LDC, ht, kb, gx[kb], u); da2 = mma_block(W2T, LDC, ht, kb, dv[kb], da2); }
      // (straight-line over the four elements, the uniform `le` test outside: hipcc then packs the polynomials of two
      // elements into v_pk_* instructions - with the test inside the element loop every evaluation was scalar code
      // behind a branch, 16 branches per tile)
      f32x4 du, a2, a1v, d1v;
#pragma unroll
      for (int q = 0; q < 4; ++q) { float a_, d_; gelu_pair(u[q], a_, d_); a1v[q] = a_; d1v[q] = d_; }
      if (le) {
        f32x4 g2v, d2v;
#pragma unroll
        for (int q = 0; q < 4; ++q) { float a_, d_; gelu_pair(a1v[q], a_, d_); g2v[q] = a_; d2v[q] = d_; }
        du = da2 * d2v * d1v; a2 = g2v;
        if (ht == 0 && g == 0) { du[0] = du_0; a2[0] = a2_0; }              // hidden channel 0: through the conv
      } else { du = da2 * d1v; a2 = a1v; }
#pragma unroll
      for (int mo = 0; mo < MT; ++mo) dg[mo] = mma_block(W1T, LDH, mo, ht, du, dg[mo]);
      if (want_dw) {
        *reinterpret_cast<float4*>(Tdu + r * LDT + 4 * g) = tofloat4(du);
        *reinterpret_cast<float4*>(Ta2 + r * LDT + 4 * g) = tofloat4(a2);
        const f32x4 duT = tr_read(Tdu), a2T = tr_read(Ta2);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dW1[ht][m] = mfma4(duT[q], gxT[m][q], dW1[ht][m]);      // rows: hidden 16 ht + ., columns: channel 16 m + .
            dW2[m][ht] = mfma4(dvT[m][q], a2T[q], dW2[m][ht]);      // rows: channel 16 m + ., columns: hidden 16 ht + .
          }
        }
        sb1[ht] += du;
      }
    }
    // ---- LN2 backward, dx1 = dx2 + dLN, do = Wp^T dx1
    float s1 = 0.f, s2 = 0.f;
    f32x4 dyh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      dyh[m] = cv ? dg[m] * vec4(g2, m) : zero4;
      s1 += (dyh[m][0] + dyh[m][1]) + (dyh[m][2] + dyh[m][3]);
      s2 += (dyh[m][0] * xh[m][0] + dyh[m][1] * xh[m][1]) + (dyh[m][2] * xh[m][2] + dyh[m][3] * xh[m][3]);
      if (cv) { sgam[m] += dg[m] * xh[m]; sbet[m] += dg[m]; }
    }
    const float m1 = rows_sum(s1) * invC, m2 = rows_sum(s2) * invC;
    f32x4 dx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      dx[m] = cv ? dv[m] + (dyh[m] - m1 - xh[m] * m2) * rstd : zero4;
      if (cv) *reinterpret_cast<float4*>(dx1_out + wo + (size_t)tok * C + 16 * m + 4 * g) = tofloat4(dx[m]);
    }
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      f32x4 o = zero4;
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) o = mma_block(WpT, LDC, mo, kb, dx[kb], o);
      if (cv) *reinterpret_cast<float4*>(do_hm + wo + ((size_t)(4 * mo + g) * N + tok) * 4) = tofloat4(o);
    }
  }
  // ---- flush: small gradients (sums over the 16 token lanes of a row), then the weight-gradient tiles through the LDS
  __syncthreads();                                     // every wave is done with the weight copies
  // dW1 [HID][C] | dW2 [C][HID] as DOUBLES (a full-wave ds_add_f32 takes 192 LDS cycles on gfx950, ds_add_f64 8:
  // tools/diag/lds_cost_probe.hip; with fp32 tiles this flush was ~6 000 LDS cycles per wave, all waves of a CU at once) |
  // b1 [HID] | b2, gamma, beta [KP each] | le [4] as floats (sixteen or fewer lanes per add)
  double* stg = reinterpret_cast<double*>(sm);
  float* sB1 = sm + 16 * C * C; float* sB2 = sB1 + HID; float* sG = sB2 + KP; float* sBe = sG + KP; float* sLe = sBe + KP;
  for (int i = threadIdx.x; i < 8 * C * C; i += blockDim.x) stg[i] = 0.;
  for (int i = threadIdx.x; i < HID + 3 * KP + 4; i += blockDim.x) sB1[i] = 0.f;
  __syncthreads();
  if (want_dw) {
#pragma unroll
    for (int h = 0; h < HT; ++h) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = group_sum<16>(sb1[h][q]);
        if (r == 0) atomicAdd(sB1 + 16 * h + 4 * g + q, v);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int hid = 16 * h + 4 * g + q, c = 16 * m + r;                 // dW1 tile: row hidden, column channel
          if (c < C) atomicAdd(stg + hid * C + c, (double)dW1[h][m][q]);
          const int c2 = 16 * m + 4 * g + q, hid2 = 16 * h + r;               // dW2 tile: row channel, column hidden
          if (c2 < C) atomicAdd(stg + 4 * C * C + c2 * HID + hid2, (double)dW2[m][h][q]);
        }
      }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = group_sum<16>(sb2[m][q]);
        if (r == 0 && 16 * m + 4 * g + q < C) atomicAdd(sB2 + 16 * m + 4 * g + q, v);
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float vg = group_sum<16>(sgam[m][q]), vb = group_sum<16>(sbet[m][q]);
      if (r == 0 && 16 * m + 4 * g + q < C) { atomicAdd(sG + 16 * m + 4 * g + q, vg); atomicAdd(sBe + 16 * m + 4 * g + q, vb); }
    }
  if (le) {
    const float v0 = group_sum<64>(gle0), v1 = group_sum<64>(gle1), v2 = group_sum<64>(gle2);
    if (lane == 0) { atomicAdd(sLe, v0); atomicAdd(sLe + 1, v1); atomicAdd(sLe + 2, v2); }
  }
  __syncthreads();
  if (RAL_DIAG_SKIP & 2) return;
  if (want_dw) {
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) { atomicAdd(gr.w1 + i, (float)stg[i]); atomicAdd(gr.w2 + i, (float)stg[4 * C * C + i]); }
    for (int i = threadIdx.x; i < HID; i += blockDim.x) atomicAdd(gr.b1 + i, sB1[i]);
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(gr.b2 + i, sB2[i]);
  }
  for (int i = threadIdx.x; i < C; i += blockDim.x) { atomicAdd(gr.ln2w + i, sG[i]); atomicAdd(gr.ln2b + i, sBe[i]); }
  if (le && threadIdx.x < 3) atomicAdd(gr.le + threadIdx.x, sLe[threadIdx.x]);
}

// ---- the same chain with TWO waves per 16-token tile (C = 32).  The weight-gradient tiles of a level are
// 2 x (4C x C) / 64 accumulator registers per wave that carries all of them: 128 at C = 32, with everything else the chain keeps
// that is 256 registers and ~100 spilled.  Here a PAIR of waves shares a tile and each takes half of the hidden range - half of
// the fc1 products, GELU evaluations and weight-gradient tiles (64 registers); nothing of the expensive part is computed twice.
// What both compute: LayerNorm forward and backward of the tile (a few dozen instructions).  They meet twice per tile: LN2(x1)
// and dx2 tiles for the transposed reads are written once per pair, and the two partial d LN2(x1) tiles - each wave's sum over
// its own hidden channels - are exchanged through the LDS.  After that wave 0 keeps channel tile 0 (dx1 store, proj^T product,
// bias / LayerNorm gradients), wave 1 tile 1.  The local enhancement (hidden channel 0) belongs to wave 0.
#ifndef RAL_MLPW2_NP
#define RAL_MLPW2_NP 2     // pairs per workgroup.  4 (100 KB of LDS, one workgroup per CU): 177.7 us per launch against 182.0 with 2 (78 KB, two
#endif                     // per CU), but the STEP is 12.85 ms with 2 against 12.98 with 4 and 12.94 with k_mlp_bwd_s (tools/diag/ab_step.sh, four
                           // interleaved rounds on one box): the other lane's kernels find room beside the smaller workgroups
template <int C>
struct Mlpbw2Shape {
  static constexpr int KP = C, MT = KP / 16, HID = 4 * C, HT = HID / 16, HW = HT / 2, NP = RAL_MLPW2_NP;   // NP: pairs per workgroup
  static constexpr int LDC = KP + 4, LDH = HID + 4, LDT = 20;
  // floats: W1 [HID][LDC] | W2T [HID][LDC] | W1T [KP][LDH] | WpT [KP][LDC] | g2, be2, w10, w2c0 [KP each] | b1 [HID] |
  //         per pair: LN2(x1) tiles [MT], dx2 tiles [MT] (shared), then per wave: du | a2 tiles (the wave's partial
  //         d LN2(x1), MT x 64 lanes x 4 floats, goes over them at the end of a tile)
  static constexpr int W2TO = HID * LDC, W1TO = W2TO + HID * LDC, WPTO = W1TO + KP * LDH, VO = WPTO + KP * LDC, B1O = VO + 4 * KP,
                       SCR = B1O + HID, WTILES = 2 * 16 * LDT, PSCR = 2 * MT * 16 * LDT + 2 * WTILES, TOTAL = SCR + NP * PSCR;
  static_assert(MT == 2, "a pair splits the channel tiles of the tail one each");
  static_assert(256 * MT <= WTILES, "the partial d LN2(x1) of a wave fits its du / a2 tiles");
  static_assert(TOTAL >= 16 * C * C + 2 * HID + 4 * KP + 8, "the flush staging (weight-gradient tiles in doubles) fits the kernel's LDS");
};

// H16: the four CHANNEL products of a tile (fc1, fc2^T, fc1^T, proj^T: 104 of a wave's 168 fp32 matrix instructions per tile, 34
// issue cycles each beside the vector work) as fp16-pair products - three 18-cycle instructions for four.  The weights sit in the
// LDS as two fp16 planes per matrix (one power of two per matrix - largest |w| into [2^13, 2^14) -, second piece as it stands: the
// three products of a pair go into ONE accumulator; no transposed copy of W1: the A operand
// of fc1^T is a transposing read, ds_read_b64_tr_b16, of the same rows), the tile's LayerNorm rows and its gradient rows (dx2, du, dx1) pass one power of
// two per tile (largest |dx2| of the tile into [2^8, 2^9): seven binades of headroom for du and dx1) that leaves with the
// epilogues.  The weight-gradient products stay on the fp32 instruction (their operands are fp32 tiles transposed through the LDS).
template <int C>
struct Mlpbw2hShape {
  using S2 = Mlpbw2Shape<C>;
  static constexpr int KP = S2::KP, MT = S2::MT, HID = S2::HID, NP = S2::NP, LDA = KP + 8;       // LDA: plane row stride in halves
  static constexpr int PA = HID * LDA, PP = KP * LDA;                                           // plane strides of W1 / W2T, WpT
  static constexpr int HW1 = 0, HW2T = HW1 + 2 * PA, HWPT = HW2T + 2 * PA, HTOT = HWPT + 2 * PP;
  static constexpr int VO = (HTOT / 2 + 3) & ~3, B1O = VO + 4 * KP, UNO = B1O + HID, SCR = UNO + 8, TOTAL = SCR + NP * S2::PSCR;
  static_assert(TOTAL >= 16 * C * C + 2 * HID + 4 * KP + 8, "the flush staging fits the kernel's LDS");
};

typedef short mw_s16x4 __attribute__((ext_vector_type(4)));
typedef mw_s16x4 __attribute__((address_space(3))) * mw_lds_s16x4;

template <int C, bool H16>
__global__ __launch_bounds__(128 * RAL_MLPW2_NP, 2) void k_mlp_bwd_w2(const float* __restrict__ dx2, const float* __restrict__ x1,
                                                       BlockP w, BlockP gr, float* __restrict__ dx1_out,
                                                       float* __restrict__ do_hm, int N, int B, int want_dw) {
  using SH = Mlpbw2Shape<C>; using SP = Mlpbw2hShape<C>;
  constexpr int KP = SH::KP, MT = SH::MT, HID = SH::HID, HW = SH::HW, NP = SH::NP, LDC = SH::LDC, LDH = SH::LDH, LDT = SH::LDT;
  constexpr int VO = H16 ? SP::VO : SH::VO, B1O = H16 ? SP::B1O : SH::B1O, SCR = H16 ? SP::SCR : SH::SCR;
  constexpr int LDA = SP::LDA, PA = SP::PA, PP = SP::PP;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  float* W1 = sm; float* W2T = sm + SH::W2TO; float* W1T = sm + SH::W1TO; float* WpT = sm + SH::WPTO;          // (fp32 form)
  _Float16* hb = reinterpret_cast<_Float16*>(sm);                                                               // (H16: plane images)
  _Float16* W1H = hb + SP::HW1; _Float16* W2TH = hb + SP::HW2T; _Float16* WpTH = hb + SP::HWPT;
  unsigned* mxb = reinterpret_cast<unsigned*>(sm + SP::UNO);
  float* g2 = sm + VO; float* be2 = g2 + KP; float* w10 = be2 + KP; float* w2c0 = w10 + KP; float* b1 = sm + B1O;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave >> 1, half = wave & 1;
  float* PT = sm + SCR + pair * SH::PSCR;                             // the pair's LN2(x1) | dx2 tiles
  float* TW = PT + 2 * MT * 16 * LDT + half * SH::WTILES;             // this wave's du | a2 tiles
  const float* TO = PT + 2 * MT * 16 * LDT + (half ^ 1) * SH::WTILES; // the other wave's
  // ---- weights -> LDS (once per workgroup)
  for (int i = threadIdx.x; i < SCR; i += blockDim.x) sm[i] = 0.f;
  __syncthreads();
  float unp = 1.f, un1 = 1.f, un2 = 1.f;            // H16: inverses of the matrices' powers of two
  if constexpr (H16) {
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) m0 = fmaxf(m0, fabsf(w.wp[i]));
    for (int i = threadIdx.x; i < HID * C; i += blockDim.x) { m1 = fmaxf(m1, fabsf(w.w1[i])); m2 = fmaxf(m2, fabsf(w.w2[i])); }
    m0 = group_max<64>(m0); m1 = group_max<64>(m1); m2 = group_max<64>(m2);
    if (lane == 0) { atomicMax(mxb, __float_as_uint(m0)); atomicMax(mxb + 1, __float_as_uint(m1)); atomicMax(mxb + 2, __float_as_uint(m2)); }
    __syncthreads();
    const float sp = h2_row_scale(mxb[0]), s1 = h2_row_scale(mxb[1]), s2 = h2_row_scale(mxb[2]);
    unp = h2_row_unscale(mxb[0]); un1 = h2_row_unscale(mxb[1]); un2 = h2_row_unscale(mxb[2]);
    for (int i = threadIdx.x; i < HID * C; i += blockDim.x) {
      const int h = i / C, c = i - h * C;
      const H2 a = f16_split2u(w.w1[i] * s1);                     // W1[h][c]
      W1H[h * LDA + c] = a.a; W1H[PA + h * LDA + c] = a.b;
      const H2 b = f16_split2u(w.w2[c * HID + h] * s2);           // W2[c][h]
      W2TH[h * LDA + c] = b.a; W2TH[PA + h * LDA + c] = b.b;
    }
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
      const int c = i / C, j = i - c * C;
      const H2 a = f16_split2u(w.wp[i] * sp);                     // Wp[c][j] -> WpT[j][c]
      WpTH[j * LDA + c] = a.a; WpTH[PP + j * LDA + c] = a.b;
    }
  } else {
    for (int i = threadIdx.x; i < HID * C; i += blockDim.x) {
      const int h = i / C, c = i - h * C;
      const float a = w.w1[i];                         // W1[h][c]
      W1[h * LDC + c] = a; W1T[c * LDH + h] = a;
      W2T[h * LDC + c] = w.w2[c * HID + h];            // W2[c][h]
    }
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) { const int c = i / C, j = i - c * C; WpT[j * LDC + c] = w.wp[i]; }   // Wp[c][j]
  }
  for (int i = threadIdx.x; i < C; i += blockDim.x) { g2[i] = w.ln2w[i]; be2[i] = w.ln2b[i]; w10[i] = w.w1[i]; w2c0[i] = w.w2[i * HID]; }
  for (int i = threadIdx.x; i < HID; i += blockDim.x) b1[i] = w.b1[i];
  const bool le = w.le != nullptr;
  const bool leh = le && half == 0;                    // the wave that carries hidden channel 0 through the conv
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  __syncthreads();
  constexpr float invC = 1.0f / C;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto mma_block = [&](const float* W, int ld, int mo, int kb, f32x4 bt, f32x4 acc) -> f32x4 {
    const float4 wa = *reinterpret_cast<const float4*>(W + (16 * mo + r) * ld + 16 * kb + 4 * g);
    acc = mfma4(wa.x, bt[0], acc); acc = mfma4(wa.y, bt[1], acc); acc = mfma4(wa.z, bt[2], acc); acc = mfma4(wa.w, bt[3], acc);
    return acc;
  };
  // H16: (acc, accx) += W[16 mo + r][K block kb] x the split tile `bt`; with TR the A operand is the TRANSPOSE of the stored rows:
  // A[row = column 16 mo + r of the image][k = image rows 16 kb + 4 g ..] (lane 4 q + p of a 16-lane group passes the address of
  // image row 16 kb + 4 g + q, columns 16 mo + 4 p .. + 3 and receives column r of the four rows)
  auto mma_h = [&](const _Float16* Wh, int plane, int mo, int kb, const H2x4& bt, f32x4& acc, auto tr) {
    h16x4 a1, a2;
    if constexpr (decltype(tr)::value) {
      const _Float16* pw = Wh + (16 * kb + 4 * g + (r >> 2)) * LDA + 16 * mo + 4 * (r & 3);
      a1 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((mw_lds_s16x4)(pw)));
      a2 = __builtin_bit_cast(h16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((mw_lds_s16x4)(pw + plane)));
    } else {
      const _Float16* pw = Wh + (16 * mo + r) * LDA + 16 * kb + 4 * g;
      a1 = *reinterpret_cast<const h16x4*>(pw); a2 = *reinterpret_cast<const h16x4*>(pw + plane);
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, bt.a, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, bt.a, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, bt.b, acc, 0, 0, 0);
  };
  // the K = 2 x 16 = 32 channels of a row in ONE instruction (v_mfma_f32_16x16x32_f16 takes the issue cycles of the K = 16 one): the
  // lane's k slots 8 g .. 8 g + 7 are channels 4 g .. 4 g + 3 and 16 + 4 g .. + 3 on both operands - any order, as long as it is the same
  auto mma_h32 = [&](const _Float16* Wh, int plane, int mo, const H2x4& b0, const H2x4& b1_, f32x4& acc) {
    const _Float16* pw = Wh + (16 * mo + r) * LDA + 4 * g;
    const h16x4 a10 = *reinterpret_cast<const h16x4*>(pw), a11 = *reinterpret_cast<const h16x4*>(pw + 16);
    const h16x4 a20 = *reinterpret_cast<const h16x4*>(pw + plane), a21 = *reinterpret_cast<const h16x4*>(pw + plane + 16);
    const f16x8 a1 = __builtin_shufflevector(a10, a11, 0, 1, 2, 3, 4, 5, 6, 7), a2 = __builtin_shufflevector(a20, a21, 0, 1, 2, 3, 4, 5, 6, 7);
    const f16x8 p1 = __builtin_shufflevector(b0.a, b1_.a, 0, 1, 2, 3, 4, 5, 6, 7), p2 = __builtin_shufflevector(b0.b, b1_.b, 0, 1, 2, 3, 4, 5, 6, 7);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, p1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, p2, acc, 0, 0, 0);
  };
  auto vec4 = [&](const float* v, int tile) -> f32x4 {
    const float4 t = *reinterpret_cast<const float4*>(v + 16 * tile + 4 * g);
    return f32x4{t.x, t.y, t.z, t.w};
  };
  auto tr_read = [&](const float* T) -> f32x4 {
    return f32x4{T[(4 * g + 0) * LDT + r], T[(4 * g + 1) * LDT + r], T[(4 * g + 2) * LDT + r], T[(4 * g + 3) * LDT + r]};
  };
  auto mine = [&](const f32x4 (&a)[MT]) -> f32x4 {   // the channel tile this wave keeps (element-wise selects: an indexed array goes to scratch)
    const f32x4 a0 = a[0], a1 = a[1];
    return f32x4{half ? a1[0] : a0[0], half ? a1[1] : a0[1], half ? a1[2] : a0[2], half ? a1[3] : a0[3]};
  };
  // accumulators that live over all the tiles of the wave: hidden tiles HW half .. HW half + HW - 1, channel tile `half`
  f32x4 dW1[HW][MT], dW2[MT][HW], sb1[HW], sb2 = zero4, sgam = zero4, sbet = zero4;
#pragma unroll
  for (int h = 0; h < HW; ++h) { sb1[h] = zero4;
#pragma unroll
    for (int m = 0; m < MT; ++m) { dW1[h][m] = zero4; dW2[m][h] = zero4; } }
  float gle0 = 0.f, gle1 = 0.f, gle2 = 0.f;
  const int tpw = N >> 4, ntile = B * tpw;
  // operands requested one tile ahead (k_mlp_bwd_w); the halo tokens by the wave of the conv only
  float4 pxa[MT], pda[MT];
  constexpr int CPL = C / 16;
  float pxh[CPL], pdh[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) { pxh[c] = 0.f; pdh[c] = 0.f; }
  auto request = [&](int tl, float4 (&xa)[MT], float4 (&da)[MT], float (&hx)[CPL], float (&hd)[CPL]) {
    const int tlc = tl < ntile ? tl : ntile - 1;                      // (past the end: a valid address, its gradient rows are zeroed)
    const int win_ = tlc / tpw, t0_ = (tlc - win_ * tpw) << 4;
    const float* xw_ = x1 + (size_t)win_ * N * C; const float* dw_ = dx2 + (size_t)win_ * N * C;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int off = (t0_ + r) * C + 16 * m + 4 * g;
      xa[m] = *reinterpret_cast<const float4*>(xw_ + off);
      da[m] = *reinterpret_cast<const float4*>(dw_ + off);
    }
    if (leh) {
      const int th = g < 2 ? t0_ - 2 + g : t0_ + 14 + g;
      const int thc = (th >= 0 && th < N) ? th : 0;
#pragma unroll
      for (int c = 0; c < CPL; ++c) { hx[c] = xw_[(size_t)thc * C + r + 16 * c]; hd[c] = dw_[(size_t)thc * C + r + 16 * c]; }
    }
  };
  // every pair of the workgroup makes the same number of trips (the two meetings per tile are workgroup barriers); a trip past
  // the last tile runs on the last tile's x1 with a zero gradient tile - every sum it adds to is a sum of zeros - and stores nothing
  const int stride = gridDim.x * NP, ntrip = (ntile + stride - 1) / stride;
  int tile = blockIdx.x * NP + pair;
  request(tile, pxa, pda, pxh, pdh);
  for (int trip = 0; trip < ((RAL_DIAG_SKIP & 1) ? 0 : ntrip); ++trip, tile += stride) {
    const bool live = tile < ntile;
    const int tlc = live ? tile : ntile - 1;
    const int win = tlc / tpw, t0 = (tlc - win * tpw) << 4, tok = t0 + r;
    const size_t wo = (size_t)win * N * C;
    f32x4 xv[MT], dv[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xv[m] = f32x4{pxa[m].x, pxa[m].y, pxa[m].z, pxa[m].w};
      dv[m] = live ? f32x4{pda[m].x, pda[m].y, pda[m].z, pda[m].w} : zero4;
    }
    // H16: the power of two that puts the largest magnitude of a tile's rows into [2^8, 2^9) (both waves of the pair form the same one)
    auto tile_scale = [&](const f32x4 (&t)[MT]) -> float {
      float tmax = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) tmax = fmaxf(tmax, fmaxf(fmaxf(fabsf(t[m][0]), fabsf(t[m][1])), fmaxf(fabsf(t[m][2]), fabsf(t[m][3]))));
      const unsigned tbits = __float_as_uint(group_max<64>(tmax));
      const int tf = 262 - (int)(tbits >> 23);
      return tbits == 0u ? 1.0f : __uint_as_float((unsigned)(tf < 187 ? (tf > 1 ? tf : 1) : 187) << 23);
    };
    float sc_t = 1.f, inv_t = 1.f;       // the tile's gradient rows (dx2, then du and dx1 with the same factor: it leaves with the epilogues)
    H2x4 dvh[MT];
    if constexpr (H16) {
      sc_t = tile_scale(dv);
      inv_t = __uint_as_float((254u << 23) - __float_as_uint(sc_t));
#pragma unroll
      for (int m = 0; m < MT; ++m) dvh[m] = split4(tofloat4(dv[m] * sc_t));
    }
    float cxh[CPL], cdh[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) { cxh[c] = pxh[c]; cdh[c] = live ? pdh[c] : 0.f; }
    request(tile + stride, pxa, pda, pxh, pdh);                 // the next tile's operands
    float hA0[4] = {0.f, 0.f, 0.f, 0.f}, hD[2] = {0.f, 0.f};   // GELU(u[:, 0]) of the halo tokens; da2[:, 0] of t0-1 and t0+16
    if (leh) {   // halo tokens t0-2, t0-1, t0+16, t0+17: lane group g = slot, lane r (+ 16 c) = channel
      const int th = g < 2 ? t0 - 2 + g : t0 + 14 + g;
      const bool tin = th >= 0 && th < N;
      float xs = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) xs += cxh[c];
      const float mean = group_sum<16>(xs) * invC;
      float dd[CPL], vs = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) { dd[c] = cxh[c] - mean; vs += dd[c] * dd[c]; }
      const float rstd = 1.0f / sqrtf(group_sum<16>(vs) * invC + 1e-5f);
      float gg = 0.f, dh = 0.f;
#pragma unroll
      for (int c = 0; c < CPL; ++c) {
        const int rc = r + 16 * c;
        gg += (dd[c] * rstd * g2[rc] + be2[rc]) * w10[rc];
        dh += cdh[c] * w2c0[rc];
      }
      const float u0 = group_sum<16>(gg) + b1[0];
      const float a0 = tin ? gelu_f(u0) : 0.f;
      const float d0 = tin ? group_sum<16>(dh) : 0.f;
      hA0[0] = lane_value(a0, 0); hA0[1] = lane_value(a0, 16); hA0[2] = lane_value(a0, 32); hA0[3] = lane_value(a0, 48);
      hD[0] = lane_value(d0, 16); hD[1] = lane_value(d0, 32);
    }
    // ---- LN2 forward of the tile (both waves)
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) sum += (xv[m][0] + xv[m][1]) + (xv[m][2] + xv[m][3]);
    const float mean = rows_sum(sum) * invC;
    float var = 0.f;
    f32x4 xh[MT], gx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xh[m] = xv[m] - mean;
      var += (xh[m][0] * xh[m][0] + xh[m][1] * xh[m][1]) + (xh[m][2] * xh[m][2] + xh[m][3] * xh[m][3]);
    }
    const float rstd = 1.0f / sqrtf(rows_sum(var) * invC + 1e-5f);
#pragma unroll
    for (int m = 0; m < MT; ++m) { xh[m] = xh[m] * rstd; gx[m] = xh[m] * vec4(g2, m) + vec4(be2, m); }
    // ---- local enhancement through hidden channel 0 (wave 0 of the pair; every lane of a token column computes the same numbers)
    float a2_0 = 0.f, du_0 = 0.f;
    if (leh) {
      float u0p = 0.f, d0p = 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 wr0 = vec4(w10, m), wc0 = vec4(w2c0, m);
        u0p += (gx[m][0] * wr0[0] + gx[m][1] * wr0[1]) + (gx[m][2] * wr0[2] + gx[m][3] * wr0[3]);
        d0p += (dv[m][0] * wc0[0] + dv[m][1] * wc0[1]) + (dv[m][2] * wc0[2] + dv[m][3] * wc0[3]);
      }
      const float u0 = rows_sum(u0p) + b1[0], da0 = rows_sum(d0p);       // u[tok, 0], da2[tok, 0]
      float A0, dA;
      gelu_pair(u0, A0, dA);
      const float sm1 = dpp_shift<0x111>(A0), sp1 = dpp_shift<0x101>(A0);   // row_shr:1 / row_shl:1
      const float Am = r == 0 ? hA0[1] : sm1, Ap = r == 15 ? hA0[2] : sp1;   // A0 of tokens tok - 1, tok + 1
      float c0g, c0d;
      gelu_pair(lw0 * Am + lw1 * A0 + lw2 * Ap, c0g, c0d);
      a2_0 = c0g;
      const float DC = da0 * c0d;                                         // d loss / d conv output at tok
      const float A_first = lane_value(A0, 0), A_last = lane_value(A0, 15);
      const float DCl = (t0 - 1 >= 0) ? hD[0] * gelu_grad_f(lw0 * hA0[0] + lw1 * hA0[1] + lw2 * A_first) : 0.f;
      const float DCr = (t0 + 16 < N) ? hD[1] * gelu_grad_f(lw0 * A_last + lw1 * hA0[2] + lw2 * hA0[3]) : 0.f;
      const float dm1 = dpp_shift<0x111>(DC), dp1 = dpp_shift<0x101>(DC);
      const float DCm = r == 0 ? DCl : dm1, DCp = r == 15 ? DCr : dp1;     // DC of tokens tok - 1, tok + 1
      du_0 = (lw0 * DCp + lw1 * DC + lw2 * DCm) * dA;
      if (g == 0) { gle0 += DC * Am; gle1 += DC * A0; gle2 += DC * Ap; }
    }
    // ---- first meeting: LN2(x1) tiles from wave 0, dx2 tiles from wave 1; behind the barrier the other wave has also read this
    // wave's partial of the previous tile, so the du / a2 tiles may be written again
    if (want_dw) {
#pragma unroll
      for (int m = 0; m < MT; ++m) *reinterpret_cast<float4*>(PT + ((half ? MT : 0) + m) * 16 * LDT + r * LDT + 4 * g) = tofloat4(half ? dv[m] : gx[m]);
    }
    __syncthreads();
    f32x4 gxT[MT], dvT[MT];
    if (want_dw) {
#pragma unroll
      for (int m = 0; m < MT; ++m) { gxT[m] = tr_read(PT + m * 16 * LDT); dvT[m] = tr_read(PT + (MT + m) * 16 * LDT); }
      sb2 += mine(dv);
    }
    // ---- this wave's hidden chunks of 16 channels
    f32x4 dg[MT];
    H2x4 gxh[MT];
    float ung = un1;                     // H16: fc1's epilogue factor (the weights' and the LayerNorm tile's powers of two)
    if constexpr (H16) {
      const float sg = tile_scale(gx);
      ung = un1 * __uint_as_float((254u << 23) - __float_as_uint(sg));
#pragma unroll
      for (int m = 0; m < MT; ++m) gxh[m] = split4(tofloat4(gx[m] * sg));
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) dg[m] = zero4;
    float* Tdu = TW; float* Ta2 = TW + 16 * LDT;
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      const int ht = HW * half + hl;
      f32x4 u = vec4(b1, ht), da2 = zero4;
      if constexpr (H16) {
        f32x4 ua = zero4;
        mma_h32(W1H, PA, ht, gxh[0], gxh[1], ua);
        mma_h32(W2TH, PA, ht, dvh[0], dvh[1], da2);
        u = ua * ung + u;
        da2 = da2 * (un2 * inv_t);
      } else {
#pragma unroll
        for (int kb = 0; kb < MT; ++kb) { u = mma_block(W1, LDC, ht, kb, gx[kb], u); da2 = mma_block(W2T, LDC, ht, kb, dv[kb], da2); }
      }
      f32x4 du, a2, a1v, d1v;
#pragma unroll
      for (int q = 0; q < 4; ++q) { float a_, d_; gelu_pair(u[q], a_, d_); a1v[q] = a_; d1v[q] = d_; }
      if (le) {
        f32x4 g2v, d2v;
#pragma unroll
        for (int q = 0; q < 4; ++q) { float a_, d_; gelu_pair(a1v[q], a_, d_); g2v[q] = a_; d2v[q] = d_; }
        du = da2 * d2v * d1v; a2 = g2v;
        if (hl == 0 && half == 0 && g == 0) { du[0] = du_0; a2[0] = a2_0; }    // hidden channel 0: through the conv
      } else { du = da2 * d1v; a2 = a1v; }
      if constexpr (H16) {
        const H2x4 duh = split4(tofloat4(du * sc_t));
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) mma_h(W1H, PA, mo, ht, duh, dg[mo], std::true_type{});
      } else {
#pragma unroll
        for (int mo = 0; mo < MT; ++mo) dg[mo] = mma_block(W1T, LDH, mo, ht, du, dg[mo]);
      }
      if (want_dw) {
        *reinterpret_cast<float4*>(Tdu + r * LDT + 4 * g) = tofloat4(du);
        *reinterpret_cast<float4*>(Ta2 + r * LDT + 4 * g) = tofloat4(a2);
        const f32x4 duT = tr_read(Tdu), a2T = tr_read(Ta2);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dW1[hl][m] = mfma4(duT[q], gxT[m][q], dW1[hl][m]);      // rows: hidden 16 ht + ., columns: channel 16 m + .
            dW2[m][hl] = mfma4(dvT[m][q], a2T[q], dW2[m][hl]);      // rows: channel 16 m + ., columns: hidden 16 ht + .
          }
        }
        sb1[hl] += du;
      }
    }
    if constexpr (H16) {
#pragma unroll
      for (int m = 0; m < MT; ++m) dg[m] = dg[m] * (un1 * inv_t);
    }
    // ---- second meeting: the two partial d LN2(x1) tiles (a + b = b + a: both waves hold the same sum)
#pragma unroll
    for (int m = 0; m < MT; ++m) *reinterpret_cast<float4*>(TW + (m * 64 + lane) * 4) = tofloat4(dg[m]);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 o = *reinterpret_cast<const float4*>(TO + (m * 64 + lane) * 4);
      dg[m] += f32x4{o.x, o.y, o.z, o.w};
    }
    // ---- LN2 backward (both waves), then this wave's channel tile: dx1 = dx2 + dLN, do = Wp^T dx1 rows 16 half ..
    float s1 = 0.f, s2 = 0.f;
    f32x4 dyh[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      dyh[m] = dg[m] * vec4(g2, m);
      s1 += (dyh[m][0] + dyh[m][1]) + (dyh[m][2] + dyh[m][3]);
      s2 += (dyh[m][0] * xh[m][0] + dyh[m][1] * xh[m][1]) + (dyh[m][2] * xh[m][2] + dyh[m][3] * xh[m][3]);
    }
    { const f32x4 dgm = mine(dg); sgam += dgm * mine(xh); sbet += dgm; }
    const float m1 = rows_sum(s1) * invC, m2 = rows_sum(s2) * invC;
    f32x4 dx[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) dx[m] = dv[m] + (dyh[m] - m1 - xh[m] * m2) * rstd;
    f32x4 o = zero4;
    if constexpr (H16) {
      mma_h32(WpTH, PP, half, split4(tofloat4(dx[0] * sc_t)), split4(tofloat4(dx[1] * sc_t)), o);
      o = o * (unp * inv_t);
    } else {
#pragma unroll
      for (int kb = 0; kb < MT; ++kb) o = mma_block(WpT, LDC, half, kb, dx[kb], o);
    }
    if (live) {
      *reinterpret_cast<float4*>(dx1_out + wo + (size_t)tok * C + 16 * half + 4 * g) = tofloat4(mine(dx));
      *reinterpret_cast<float4*>(do_hm + wo + ((size_t)(4 * half + g) * N + tok) * 4) = tofloat4(o);
    }
  }
  // ---- flush (as k_mlp_bwd_w: weight-gradient tiles through doubles in the LDS)
  __syncthreads();
  double* stg = reinterpret_cast<double*>(sm);
  float* sB1 = sm + 16 * C * C; float* sB2 = sB1 + HID; float* sG = sB2 + KP; float* sBe = sG + KP; float* sLe = sBe + KP;
  for (int i = threadIdx.x; i < 8 * C * C; i += blockDim.x) stg[i] = 0.;
  for (int i = threadIdx.x; i < HID + 3 * KP + 4; i += blockDim.x) sB1[i] = 0.f;
  __syncthreads();
  if (want_dw) {
#pragma unroll
    for (int h = 0; h < HW; ++h) {
      const int hb = 16 * (HW * half + h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float v = group_sum<16>(sb1[h][q]);
        if (r == 0) atomicAdd(sB1 + hb + 4 * g + q, v);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          atomicAdd(stg + (hb + 4 * g + q) * C + 16 * m + r, (double)dW1[h][m][q]);                    // dW1 tile: row hidden, column channel
          atomicAdd(stg + 4 * C * C + (16 * m + 4 * g + q) * HID + hb + r, (double)dW2[m][h][q]);      // dW2 tile: row channel, column hidden
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = group_sum<16>(sb2[q]);
      if (r == 0) atomicAdd(sB2 + 16 * half + 4 * g + q, v);
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float vg = group_sum<16>(sgam[q]), vb = group_sum<16>(sbet[q]);
    if (r == 0) { atomicAdd(sG + 16 * half + 4 * g + q, vg); atomicAdd(sBe + 16 * half + 4 * g + q, vb); }
  }
  if (leh) {
    const float v0 = group_sum<64>(gle0), v1 = group_sum<64>(gle1), v2 = group_sum<64>(gle2);
    if (lane == 0) { atomicAdd(sLe, v0); atomicAdd(sLe + 1, v1); atomicAdd(sLe + 2, v2); }
  }
  __syncthreads();
  if (RAL_DIAG_SKIP & 2) return;
  if (want_dw) {
    for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) { atomicAdd(gr.w1 + i, (float)stg[i]); atomicAdd(gr.w2 + i, (float)stg[4 * C * C + i]); }
    for (int i = threadIdx.x; i < HID; i += blockDim.x) atomicAdd(gr.b1 + i, sB1[i]);
    for (int i = threadIdx.x; i < C; i += blockDim.x) atomicAdd(gr.b2 + i, sB2[i]);
  }
  for (int i = threadIdx.x; i < C; i += blockDim.x) { atomicAdd(gr.ln2w + i, sG[i]); atomicAdd(gr.ln2b + i, sBe[i]); }
  if (le && threadIdx.x < 3) atomicAdd(gr.le + threadIdx.x, sLe[threadIdx.x]);
}

// Measured at batch 2048 (rocprofv3, serialised step, us per launch; k_mlp_bwd_s / this kernel): C = 16 (N = 256): 181 / 150.5;
// C = 8 (N = 512): 247 / 153.  An f16 form of it (every product as fp16 pairs, one power of two per tile on the gradient
// side; built, parity-clean, removed again) measured 146.5 / 152.8: the kernel is not bound by its matrix instructions.
// Round 5 (operands requested a tile ahead, flush through doubles): C = 16: 111, C = 8: 139 - and the f16 form again (as
// k_mlp_bwd_w2's, transposing reads for fc1^T): 108 / 137.6, still not worth its code; C = 32 (N = 128): k_mlp_bwd_s 241,
// k_mlp_bwd_w2 178 - 182 on the fp32 instruction, 167 with the channel products as fp16 pairs (default where the model allows).  Its tile loop is 1 571 instructions per wave and tile (~1 000 vector, 168 matrix, 75 transcendental)
// and takes ~27 000 cycles per pair of tiles and SIMD: both pipes are under half busy, the waves wait on each other's chain
// (fc1 MFMAs -> GELU -> fc1^T MFMAs -> LDS transposes -> weight-gradient MFMAs) with two waves per SIMD to hide it.
// MLP_BWD_W: 0 never, 1: C = 16 only, 2: C = 8 and 16, 3 (default): C = 32 too (k_mlp_bwd_w2, two waves per tile).
int mlp_bwd_w_kind(int C, int N, bool f16_ok) {
  static const int on = (int)ral_knob("MLP_BWD_W", 3);
  static const int h16 = (int)ral_knob("MLP_BWD_W_F16", 1);   // C = 32: the channel products as fp16 pairs where the model allows them
  if (!on || N % 16 != 0 || !(C == 16 || (on >= 2 && C == 8) || (on >= 3 && C == 32))) return 0;
  return (C == 32 && f16_ok && h16) ? 2 : 1;
}
template <int C>
static void go_mlp_bwd_w(const float* dx2, const float* x1, const BlockP& w, const BlockP& gr, float* dx1, float* do_hm, int N, int B,
                         bool want_dw, hipStream_t s) {
  static const int genv = (int)ral_knob("GRID_MLPBW", 0);
  const int nwg = (B * (N / 16) + 3) / 4;
  const size_t lds = (size_t)MlpbwShape<C>::TOTAL * sizeof(float);
  RAL_SET_LDS((k_mlp_bwd_w<C>), lds);
  static int occ = 0;
  if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mlp_bwd_w<C>, 256, lds) != hipSuccess || occ < 1)) occ = 3;
  int grid = genv > 0 ? genv : ral_num_cus() * (occ > 3 ? 3 : occ);
  if (grid > nwg) grid = nwg;
  k_mlp_bwd_w<C><<<grid, 256, lds, s>>>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw ? 1 : 0);
}
template <bool H16>
static void go_mlp_bwd_w2(const float* dx2, const float* x1, const BlockP& w, const BlockP& gr, float* dx1, float* do_hm, int N, int B,
                          bool want_dw, hipStream_t s) {
  using SH = Mlpbw2Shape<32>;
  static const int genv = (int)ral_knob("GRID_MLPBW", 0);
  const int nwg = (B * (N / 16) + SH::NP - 1) / SH::NP;
  const size_t lds = (size_t)(H16 ? Mlpbw2hShape<32>::TOTAL : SH::TOTAL) * sizeof(float);
  RAL_SET_LDS((k_mlp_bwd_w2<32, H16>), lds);
  int grid = genv > 0 ? genv : ral_num_cus() * (4 / SH::NP);      // (eight waves per CU: 232 registers)
  if (grid > nwg) grid = nwg;
  k_mlp_bwd_w2<32, H16><<<grid, 128 * SH::NP, lds, s>>>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw ? 1 : 0);
}
void launch_mlp_bwd_w(int C, int kind, const float* dx2, const float* x1, const BlockP& w, const BlockP& gr, float* dx1, float* do_hm, int N, int B,
                      bool want_dw, hipStream_t s) {
  if (C == 32 && kind == 2) go_mlp_bwd_w2<true>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw, s);
  else if (C == 32) go_mlp_bwd_w2<false>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw, s);
  else if (C == 8) go_mlp_bwd_w<8>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw, s);
  else go_mlp_bwd_w<16>(dx2, x1, w, gr, dx1, do_hm, N, B, want_dw, s);
}
