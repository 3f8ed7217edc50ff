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

// Measured at batch 2048 (rocprofv3, serialised step, us per launch: this kernel / k_mlp_fwd): C = 16 (N = 256): 67.6 / 78.7;
// C = 8 (N = 512): 73.9 / 65.5 - half of every padded MFMA tile is zeros there, k_mlp_fwd has a K = 8 path; C = 32
// (N = 128): 107 / 92 - two tiles of state per strip already spill (284 bytes per lane at 168 registers).  Both forms sit
// at about half their issue bound (C = 16: 36 fp32 MFMAs + 32 GELU evaluations per lane and tile = ~2 400 cycles, measured
// 4 200 / 4 900): what the narrow levels pay for is the fp32 MFMA itself, not the barriers.  Default: C = 16 only
// (RAL_MLP_FWD_W=2: all three widths, 0: never).
bool mlp_fwd_w_takes(int C, int N, bool want_upre) {
  static const int on = [] { const char* v = getenv("RAL_MLP_FWD_W"); return v ? atoi(v) : 1; }();
  if (!on || want_upre || N % 64 != 0) return false;   // (64: a whole number of strips at every width)
  return on >= 2 ? (C == 8 || C == 16 || C == 32) : C == 16;
}

template <int C>
static void go_mlp_fwd_w(const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s) {
  const size_t lds = (size_t)MlpwShape<C>::TOTAL * sizeof(float);
  RAL_SET_LDS((k_mlp_fwd_w<C>), lds);
  static const int genv = [] { const char* v = getenv("RAL_GRID_MLPW"); return v ? atoi(v) : 0; }();
  static int occ = 0;
  if (!occ && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_mlp_fwd_w<C>, 256, lds) != hipSuccess || occ < 1)) occ = 3;
  const int nwg = (B * (N / (16 * MlpwShape<C>::S)) + 3) / 4;
  int grid = genv > 0 ? genv : 256 * (occ > 4 ? 4 : occ);
  if (grid > nwg) grid = nwg;
  k_mlp_fwd_w<C><<<grid, 256, lds, s>>>(x, o, w, x1, x2, N, B);
}
void launch_mlp_fwd_w(int C, const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s) {
  if (C == 8) go_mlp_fwd_w<8>(x, o, w, x1, x2, N, B, s);
  else if (C == 16) go_mlp_fwd_w<16>(x, o, w, x1, x2, N, B, s);
  else go_mlp_fwd_w<32>(x, o, w, x1, x2, N, B, s);
}
