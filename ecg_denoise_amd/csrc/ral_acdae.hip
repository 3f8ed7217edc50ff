// ACDAE comparison baseline (reference model/ACDAE.py:10-86) for gfx950: a convolutional
// auto-encoder with MaxPool / linear Upsample / ECA channel attention and NO BatchNorm, so every window is
// independent end to end: one workgroup per window and layer, no grid-wide reduction anywhere.
//
//   EncBlock i (:25-39):  Conv1d(C_i, C_{i+1}, k, pad (k-1)/2) -> MaxPool1d(2) -> LeakyReLU(0.01)      k = 13, 7, 7, 7
//   DecBlock i (:42-59):  ConvTranspose1d(stride 1, pad (k-1)/2) -> Upsample(x2, linear, align_corners=False)
//                         -> LeakyReLU -> ECA (:10-22: channel means -> 3-tap conv over the CHANNEL axis, no bias
//                         -> sigmoid -> scale);  the encoder feature is added after the ECA            k = 7, 7, 7, 13
//   channels 2 -> 16 -> 32 -> 64 -> 128 -> 64 -> 32 -> 16 -> 2
//
// Kernels (256 threads, one window per workgroup iteration, inputs staged in LDS with a zero halo, weights from L2):
//   k_acd_enc_fwd   conv + pool + LeakyReLU; keeps which element of each pair was the maximum (1 byte)
//   k_acd_dec_fwd   transposed conv -> LDS -> upsample + LeakyReLU (stored: the ECA input) -> means -> scale -> + skip
//   k_acd_dec_bwd   ECA backward, LeakyReLU', upsample adjoint -> dT (stored), input gradient (correlation with W)
//   k_acd_enc_bwd   (main + skip) gradient -> LeakyReLU', un-pool -> dC (stored), input gradient
//   k_acd_dw_m      weight / bias gradients of one conv: dW[co][ci][k] = sum over windows and positions of Y X as an
//                   fp32-MFMA GEMM with index-gathered operands (k_acd_dw: the scalar form, kept for 2 output channels)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ral_acdae.hpp"
#include "ral_device.hpp"

namespace {
const int ACH[5] = {2, 16, 32, 64, 128};
const int AKS[4] = {13, 7, 7, 7};

RAL_DEV float lrelu01(float v) { return v > 0.f ? v : 0.01f * v; }

// ---------------------------------------------------------------------------------
// encoder block forward.  in: (B, CIN, Lin)  ->  e: (B, COUT, Lin / 2), am: (B, COUT, Lin / 2) bytes (argmax of the pair)
// ---------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_acd_enc_fwd(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ e,
                                                     unsigned char* __restrict__ am, int CIN, int COUT, int Lin, int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8;          // rows: HALO zeros | Lin values | HALO zeros (HALO >= PAD, 16-byte aligned)
  const int LP = Lin + 2 * HALO, Lo = Lin >> 1;
  float* xs = reinterpret_cast<float*>(smem4);
  for (int i = threadIdx.x; i < CIN * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    xs[c * LP + (h < HALO ? h : Lin + h)] = 0.f;
  }
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    const float* xw = in + (size_t)win * CIN * Lin;
    for (int i = threadIdx.x; i < CIN * (Lin >> 2); i += blockDim.x) {
      const int c = i / (Lin >> 2), p = (i - c * (Lin >> 2)) << 2;
      *reinterpret_cast<float4*>(xs + c * LP + HALO + p) = *reinterpret_cast<const float4*>(xw + c * Lin + p);
    }
    __syncthreads();
    const int q = Lin >> 2;                              // 4 conv positions (2 pooled outputs) per slot
    for (int slot = threadIdx.x; slot < COUT * q; slot += blockDim.x) {
      const int co = slot / q, l0 = (slot - co * q) << 2;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const float* wr = w + (size_t)co * CIN * KS;
      for (int ci = 0; ci < CIN; ++ci) {
        const float* row = xs + ci * LP + HALO + l0 - PAD;
        float xv[KS + 3];
#pragma unroll
        for (int t = 0; t < KS + 3; ++t) xv[t] = row[t];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float wk = wr[ci * KS + k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk, xv[j + k], acc[j]);
        }
      }
      const float b = bias[co];
      const size_t o = ((size_t)win * COUT + co) * Lo + (l0 >> 1);
      // MaxPool1d(2): the FIRST element wins a tie (torch's argmax convention)
      const bool s0 = acc[1] > acc[0], s1 = acc[3] > acc[2];
      *reinterpret_cast<float2*>(e + o) = make_float2(lrelu01((s0 ? acc[1] : acc[0]) + b), lrelu01((s1 ? acc[3] : acc[2]) + b));
      am[o] = s0; am[o + 1] = s1;
    }
  }
}

// ---------------------------------------------------------------------------------
// The same encoder block with the convolution on the fp32 MFMA (input channels >= 16): rows = positions, columns = 32
// output channels of this workgroup (blockIdx.y), K = (input channel, tap) pairs; the A operand is gathered from the
// zero-haloed input tile by index, the B operand from the workgroup's slice of the weights, staged in LDS once.  A lane
// ends with four consecutive positions of one output channel: the two pooling pairs of the epilogue.
// ---------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_acd_enc_fwd_m(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ e,
                                                       unsigned char* __restrict__ am, int CIN, int COUT, int Lin, int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8, CB = 32;
  const int LP = Lin + 2 * HALO, Lo = Lin >> 1, KT = CIN * KS;
  float* xs = reinterpret_cast<float*>(smem4);      // CIN x LP
  float* ws = xs + CIN * LP;                        // CB x KT
  const int cb0 = blockIdx.y * CB;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < CIN * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    xs[c * LP + (h < HALO ? h : Lin + h)] = 0.f;
  }
  for (int i = threadIdx.x; i < CB * KT; i += blockDim.x) ws[i] = w[(size_t)cb0 * KT + i];
  const int ptiles = Lin >> 4;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    const float* xw = in + (size_t)win * CIN * Lin;
    for (int i = threadIdx.x; i < CIN * (Lin >> 2); i += blockDim.x) {
      const int c = i / (Lin >> 2), p = (i - c * (Lin >> 2)) << 2;
      *reinterpret_cast<float4*>(xs + c * LP + HALO + p) = *reinterpret_cast<const float4*>(xw + c * Lin + p);
    }
    __syncthreads();
    for (int pt = wave; pt < ptiles; pt += 4) {           // a wave: 16 positions x both column tiles (one A gather feeds two MFMAs)
      const int p0 = pt << 4;
      const float* arow = xs + HALO - PAD + p0 + r;      // + ci LP + k
      const float* brow = ws + r * KT;                   // + kk (second tile: + 16 KT)
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      int ci = 0, k = g;                                  // this lane's (input channel, tap) of the current step
      if (k >= KS) { k -= KS; ci = 1; }
      // (KT is a multiple of 16: the channel counts are multiples of 16.)  Four steps per trip: their operand reads are
      // issued before the first MFMA needs one.
      for (int kk = g; kk < KT + g; kk += 16) {
        float av[4], b0[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = arow[ci * LP + k]; b0[u] = brow[kk + 4 * u]; b1[u] = brow[16 * KT + kk + 4 * u];
          k += 4;
          if (k >= KS) { k -= KS; ++ci; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc[0] = mfma4(av[u], b0[u], acc[0]); acc[1] = mfma4(av[u], b1[u], acc[1]); }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int co = cb0 + 16 * t + r, l0 = p0 + 4 * g;   // positions l0 .. l0 + 3 of channel co
        const float b = bias[co];
        const size_t o = ((size_t)win * COUT + co) * Lo + (l0 >> 1);
        const bool s0 = acc[t][1] > acc[t][0], s1 = acc[t][3] > acc[t][2];   // MaxPool1d(2): the FIRST element wins a tie
        *reinterpret_cast<float2*>(e + o) = make_float2(lrelu01((s0 ? acc[t][1] : acc[t][0]) + b), lrelu01((s1 ? acc[t][3] : acc[t][2]) + b));
        am[o] = s0; am[o + 1] = s1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// decoder block forward.  in: (B, CIN, Lin) -> a: (B, COUT, 2 Lin) = LeakyReLU(upsample(convT(in))),
//   s: (B, COUT) ECA scale, out: (B, COUT, 2 Lin) = a * s (+ skip)
// ---------------------------------------------------------------------------------
RAL_DEV float up2(const float* t, int n, int j) {   // linear x2 upsample, align_corners = False: src = (j + 0.5) / 2 - 0.5, clamped at 0
  const int i = j >> 1;
  if (j & 1) return 0.75f * t[i] + 0.25f * t[i + 1 < n ? i + 1 : n - 1];
  return i > 0 ? 0.25f * t[i - 1] + 0.75f * t[i] : t[0];
}

template <int KS>
__global__ __launch_bounds__(256) void k_acd_dec_fwd(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const float* __restrict__ ecaw,
                                                     const float* __restrict__ skip, float* __restrict__ a_out,
                                                     float* __restrict__ s_out, float* __restrict__ out,
                                                     int CIN, int COUT, int Lin, int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8;
  const int LP = Lin + 2 * HALO, Lo = Lin * 2;
  float* xs = reinterpret_cast<float*>(smem4);      // CIN x LP
  float* ts = xs + CIN * LP;                        // COUT x Lin   transposed-conv output
  float* as = ts + COUT * Lin;                      // COUT x Lo    LeakyReLU(upsample)
  float* ms = as + COUT * Lo;                       // COUT + 2     channel means with a zero on both sides
  float* ss = ms + COUT + 2;                        // COUT         scale
  for (int i = threadIdx.x; i < CIN * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    xs[c * LP + (h < HALO ? h : Lin + h)] = 0.f;
  }
  if (threadIdx.x == 0) { ms[0] = 0.f; ms[COUT + 1] = 0.f; }
  const float e0 = ecaw[0], e1 = ecaw[1], e2 = ecaw[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    const float* xw = in + (size_t)win * CIN * Lin;
    for (int i = threadIdx.x; i < CIN * (Lin >> 2); i += blockDim.x) {
      const int c = i / (Lin >> 2), p = (i - c * (Lin >> 2)) << 2;
      *reinterpret_cast<float4*>(xs + c * LP + HALO + p) = *reinterpret_cast<const float4*>(xw + c * Lin + p);
    }
    __syncthreads();
    // t[co][j] = b[co] + sum_ci sum_k w[ci][co][k] in[ci][j + PAD - k]
    const int q = Lin >> 2;
    for (int slot = threadIdx.x; slot < COUT * q; slot += blockDim.x) {
      const int co = slot / q, l0 = (slot - co * q) << 2;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int ci = 0; ci < CIN; ++ci) {
        const float* row = xs + ci * LP + HALO + l0 + PAD - (KS - 1);      // in[l0 + PAD - (KS-1) + t]
        const float* wr = w + ((size_t)ci * COUT + co) * KS;
        float xv[KS + 3];
#pragma unroll
        for (int t = 0; t < KS + 3; ++t) xv[t] = row[t];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float wk = wr[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk, xv[j + (KS - 1) - k], acc[j]);
        }
      }
      const float b = bias[co];
      *reinterpret_cast<float4*>(ts + co * Lin + l0) = make_float4(acc[0] + b, acc[1] + b, acc[2] + b, acc[3] + b);
    }
    __syncthreads();
    // upsample + LeakyReLU, channel means (one wave per channel at a time)
    for (int co = wave; co < COUT; co += 4) {
      float sum = 0.f;
      for (int j = lane; j < Lo; j += 64) {
        const float v = lrelu01(up2(ts + co * Lin, Lin, j));
        as[co * Lo + j] = v;
        sum += v;
      }
      sum = group_sum<64>(sum);
      if (lane == 0) ms[co + 1] = sum / (float)Lo;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < COUT; c += blockDim.x) {
      const float z = e0 * ms[c] + e1 * ms[c + 1] + e2 * ms[c + 2];     // conv k3, pad 1 over the channel axis
      const float sg = 1.0f / (1.0f + __expf(-z));
      ss[c] = sg;
      s_out[(size_t)win * COUT + c] = sg;
    }
    __syncthreads();
    const size_t ob = (size_t)win * COUT * Lo;
    for (int i = threadIdx.x; i < (COUT * Lo) >> 2; i += blockDim.x) {
      const int c = (i << 2) / Lo;
      const float4 a = reinterpret_cast<const float4*>(as)[i];
      reinterpret_cast<float4*>(a_out + ob)[i] = a;
      float4 o = f4scale(a, ss[c]);
      if (skip) o = f4add(o, reinterpret_cast<const float4*>(skip + ob)[i]);
      reinterpret_cast<float4*>(out + ob)[i] = o;
    }
  }
}

// ---------------------------------------------------------------------------------
// decoder block backward (data).  g: gradient at the block output (B, COUT, 2 Lin)  ->  dt: (B, COUT, Lin) gradient at the
// transposed-conv output (kept for the weight gradient), gin: (B, CIN, Lin), geca[3] (accumulated)
// ---------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_acd_dec_bwd(const float* __restrict__ g, const float* __restrict__ a,
                                                     const float* __restrict__ s, const float* __restrict__ w,
                                                     const float* __restrict__ ecaw, float* __restrict__ dt_out,
                                                     float* __restrict__ gin, float* __restrict__ geca,
                                                     int CIN, int COUT, int Lin, int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8;
  const int LP = Lin + 2 * HALO, Lo = Lin * 2;
  float* du = reinterpret_cast<float*>(smem4);      // COUT x Lo : g, then dU
  float* dts = du + COUT * Lo;                      // COUT x LP : dT with a zero halo
  float* ms = dts + COUT * LP;                      // COUT + 2 means (zero ends)
  float* dz = ms + COUT + 2;                        // COUT + 2 (zero ends)
  float* dm = dz + COUT + 2;                        // COUT     d mean / Lo
  float* ge = dm + COUT;                            // 4        ECA weight gradient of this workgroup
  for (int i = threadIdx.x; i < COUT * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    dts[c * LP + (h < HALO ? h : Lin + h)] = 0.f;
  }
  if (threadIdx.x == 0) { ms[0] = ms[COUT + 1] = 0.f; dz[0] = dz[COUT + 1] = 0.f; ge[0] = ge[1] = ge[2] = 0.f; }
  const float e0 = ecaw[0], e1 = ecaw[1], e2 = ecaw[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    const size_t ob = (size_t)win * COUT * Lo;
    // per channel: mean(a), ds = sum g a
    for (int co = wave; co < COUT; co += 4) {
      float sa = 0.f, sga = 0.f;
      for (int j = lane; j < Lo; j += 64) {
        const float av = a[ob + (size_t)co * Lo + j], gv = g[ob + (size_t)co * Lo + j];
        du[co * Lo + j] = gv;
        sa += av; sga += gv * av;
      }
      sa = group_sum<64>(sa); sga = group_sum<64>(sga);
      if (lane == 0) {
        const float sg = s[(size_t)win * COUT + co];
        ms[co + 1] = sa / (float)Lo;
        dz[co + 1] = sga * sg * (1.0f - sg);         // d z = d s * sigmoid'
      }
    }
    __syncthreads();
    // z[c] = e0 m[c-1] + e1 m[c] + e2 m[c+1]  =>  d m[c] = e0 dz[c+1] + e1 dz[c] + e2 dz[c-1];  d e_j = sum_c dz[c] m[c+j-1]
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int c = threadIdx.x; c < COUT; c += blockDim.x) {
      dm[c] = (e0 * dz[c + 2] + e1 * dz[c + 1] + e2 * dz[c]) / (float)Lo;
      g0 += dz[c + 1] * ms[c]; g1 += dz[c + 1] * ms[c + 1]; g2 += dz[c + 1] * ms[c + 2];
    }
    if ((int)threadIdx.x < COUT) { atomicAdd(ge + 0, g0); atomicAdd(ge + 1, g1); atomicAdd(ge + 2, g2); }
    __syncthreads();
    // dU = (g s + d mean / Lo) * LeakyReLU'(u)     (sign(a) = sign(u))
    for (int i = threadIdx.x; i < COUT * Lo; i += blockDim.x) {
      const int c = i / Lo;
      const float av = a[ob + i];
      const float d = du[i] * s[(size_t)win * COUT + c] + dm[c];
      du[i] = av > 0.f ? d : 0.01f * d;
    }
    __syncthreads();
    // upsample adjoint: u[2i] = .25 t[i-1] + .75 t[i] (u[0] = t[0]);  u[2i+1] = .75 t[i] + .25 t[i+1] (u[2n-1] = t[n-1])
    for (int i = threadIdx.x; i < COUT * Lin; i += blockDim.x) {
      const int c = i / Lin, p = i - c * Lin;
      const float* d = du + c * Lo;
      float v = (p > 0 ? 0.75f : 1.0f) * d[2 * p] + (p < Lin - 1 ? 0.75f : 1.0f) * d[2 * p + 1];
      if (p > 0) v += 0.25f * d[2 * p - 1];
      if (p < Lin - 1) v += 0.25f * d[2 * p + 2];
      dts[c * LP + HALO + p] = v;
      dt_out[((size_t)win * COUT + c) * Lin + p] = v;
    }
    __syncthreads();
    // gin[ci][i] = sum_co sum_k w[ci][co][k] dT[co][i - PAD + k]     (adjoint of the transposed conv)
    const int q = Lin >> 2;
    for (int slot = threadIdx.x; slot < CIN * q; slot += blockDim.x) {
      const int ci = slot / q, l0 = (slot - ci * q) << 2;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int co = 0; co < COUT; ++co) {
        const float* row = dts + co * LP + HALO + l0 - PAD;
        const float* wr = w + ((size_t)ci * COUT + co) * KS;
        float xv[KS + 3];
#pragma unroll
        for (int t = 0; t < KS + 3; ++t) xv[t] = row[t];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float wk = wr[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk, xv[j + k], acc[j]);
        }
      }
      *reinterpret_cast<float4*>(gin + ((size_t)win * CIN + ci) * Lin + l0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(geca + threadIdx.x, ge[threadIdx.x]);
}

// ---------------------------------------------------------------------------------
// encoder block backward (data).  g1 (+ g2): gradient at e = LeakyReLU(pool(c)), (B, COUT, Lin / 2)
//   -> dc: (B, COUT, Lin) gradient at the conv output (kept for the weight gradient), gin: (B, CIN, Lin) or null
// ---------------------------------------------------------------------------------
template <int KS>
__global__ __launch_bounds__(256) void k_acd_enc_bwd(const float* __restrict__ g1, const float* __restrict__ g2,
                                                     const float* __restrict__ e, const unsigned char* __restrict__ am,
                                                     const float* __restrict__ w, float* __restrict__ dc_out,
                                                     float* __restrict__ gin, int CIN, int COUT, int Lin, int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8;
  const int LP = Lin + 2 * HALO, Lo = Lin >> 1;
  float* dcs = reinterpret_cast<float*>(smem4);     // COUT x LP, zero halo
  for (int i = threadIdx.x; i < COUT * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    dcs[c * LP + (h < HALO ? h : Lin + h)] = 0.f;
  }
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    const size_t eb = (size_t)win * COUT * Lo;
    for (int i = threadIdx.x; i < COUT * Lo; i += blockDim.x) {
      const int c = i / Lo, p = i - c * Lo;
      float gv = g1[eb + i];
      if (g2) gv += g2[eb + i];
      if (!(e[eb + i] > 0.f)) gv *= 0.01f;
      const bool hi = am[eb + i] != 0;
      const float2 d = make_float2(hi ? 0.f : gv, hi ? gv : 0.f);
      *reinterpret_cast<float2*>(dcs + c * LP + HALO + 2 * p) = d;
      *reinterpret_cast<float2*>(dc_out + ((size_t)win * COUT + c) * Lin + 2 * p) = d;
    }
    __syncthreads();
    if (!gin) continue;
    // gin[ci][i] = sum_co sum_k w[co][ci][k] dC[co][i + PAD - k]
    const int q = Lin >> 2;
    for (int slot = threadIdx.x; slot < CIN * q; slot += blockDim.x) {
      const int ci = slot / q, l0 = (slot - ci * q) << 2;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int co = 0; co < COUT; ++co) {
        const float* row = dcs + co * LP + HALO + l0 + PAD - (KS - 1);
        const float* wr = w + ((size_t)co * CIN + ci) * KS;
        float xv[KS + 3];
#pragma unroll
        for (int t = 0; t < KS + 3; ++t) xv[t] = row[t];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          const float wk = wr[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk, xv[j + (KS - 1) - k], acc[j]);
        }
      }
      *reinterpret_cast<float4*>(gin + ((size_t)win * CIN + ci) * Lin + l0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
}

// ---------------------------------------------------------------------------------
// weight and bias gradient of one conv.  Y: (B, CY, L) gradient at the conv output, X: (B, CX, L) its input.
//   CONV  (encoder, w (CY, CX, KS)): dW[cy][cx][k] = sum Y[cy][l] X[cx][l + k - PAD]
//   CONVT (decoder, w (CX, CY, KS)): dW[cx][cy][k] = sum Y[cy][j] X[cx][j + PAD - k]
// grid: (CY/8 * CX/8 channel blocks (ceil), window splits); a workgroup walks its windows with the two 8-row tiles in LDS;
// thread -> (cy, cx, k) entries e = tid, tid + 256 (8 * 8 * KS <= 832 entries: up to 4 per thread)
// ---------------------------------------------------------------------------------
template <int KS, bool CONVT>
__global__ __launch_bounds__(256) void k_acd_dw(const float* __restrict__ Y, const float* __restrict__ X,
                                                float* __restrict__ gw, float* __restrict__ gb, int CY, int CX, int L,
                                                int B) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8, NE = 64 * KS, EPT = (NE + 255) / 256;
  const int LP = L + 2 * HALO;
  float* ys = reinterpret_cast<float*>(smem4);      // 8 x L
  float* xs = ys + 8 * L;                           // 8 x LP, zero halo
  const int nbx = (CX + 7) >> 3;
  const int cy0 = (blockIdx.x / nbx) * 8, cx0 = (blockIdx.x % nbx) * 8;
  const int ny = min(8, CY - cy0), nx = min(8, CX - cx0);
  for (int i = threadIdx.x; i < 8 * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    xs[c * LP + (h < HALO ? h : L + h)] = 0.f;
  }
  float acc[EPT], bacc = 0.f;
  int ey[EPT], ex[EPT], ek[EPT];
#pragma unroll
  for (int t = 0; t < EPT; ++t) {
    const int en = threadIdx.x + 256 * t;
    acc[t] = 0.f;
    ey[t] = (en / (8 * KS)) & 7; ex[t] = (en / KS) % 8; ek[t] = en % KS;
    if (en >= NE) ey[t] = -1;
  }
  const int wpb = (B + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * wpb, w1 = min(B, w0 + wpb);
  for (int win = w0; win < w1; ++win) {
    __syncthreads();
    for (int i = threadIdx.x; i < ny * (L >> 2); i += blockDim.x) {
      const int c = i / (L >> 2), p = (i - c * (L >> 2)) << 2;
      *reinterpret_cast<float4*>(ys + c * L + p) = *reinterpret_cast<const float4*>(Y + ((size_t)win * CY + cy0 + c) * L + p);
    }
    for (int i = threadIdx.x; i < nx * (L >> 2); i += blockDim.x) {
      const int c = i / (L >> 2), p = (i - c * (L >> 2)) << 2;
      *reinterpret_cast<float4*>(xs + c * LP + HALO + p) = *reinterpret_cast<const float4*>(X + ((size_t)win * CX + cx0 + c) * L + p);
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < EPT; ++t) {
      if (ey[t] < 0 || ey[t] >= ny || ex[t] >= nx) continue;
      const float* yr = ys + ey[t] * L;
      const float* xr = xs + ex[t] * LP + HALO + (CONVT ? PAD - ek[t] : ek[t] - PAD);
      float sum = 0.f;
      for (int l = 0; l < L; l += 4) {
        const float4 yv = *reinterpret_cast<const float4*>(yr + l);
        sum = fmaf(yv.x, xr[l], fmaf(yv.y, xr[l + 1], fmaf(yv.z, xr[l + 2], fmaf(yv.w, xr[l + 3], sum))));
      }
      acc[t] += sum;
    }
    if (gb && cx0 == 0 && (int)threadIdx.x < ny * 8) {      // bias gradient: row sums of Y (8 partial sums per row)
      const int c = threadIdx.x >> 3, part = threadIdx.x & 7;
      float sum = 0.f;
      for (int l = part; l < L; l += 8) sum += ys[c * L + l];
      bacc += sum;
    }
  }
#pragma unroll
  for (int t = 0; t < EPT; ++t) {
    if (ey[t] < 0 || ey[t] >= ny || ex[t] >= nx) continue;
    const size_t o = CONVT ? (((size_t)(cx0 + ex[t]) * CY + cy0 + ey[t]) * KS + ek[t])
                           : (((size_t)(cy0 + ey[t]) * CX + cx0 + ex[t]) * KS + ek[t]);
    atomicAdd(gw + o, acc[t]);
  }
  if (gb && cx0 == 0 && (int)threadIdx.x < ny * 8) atomicAdd(gb + cy0 + (threadIdx.x >> 3), bacc);
}

// ---------------------------------------------------------------------------------
// The same weight gradient as an fp32-MFMA GEMM (output channels >= 16): M = output channel, N = (input channel, tap)
// pairs, K = positions, summed over windows.  A workgroup owns 16 output channels x up to 16 column tiles (four per wave,
// in accumulators across its window range); both operands are gathered from the staged rows by index (the tap shift is
// part of the B operand's address; the zero halo covers the borders).  grid: (row tiles x column-tile groups, splits).
// ---------------------------------------------------------------------------------
template <int KS, bool CONVT>
__global__ __launch_bounds__(256) void k_acd_dw_m(const float* __restrict__ Y, const float* __restrict__ X,
                                                  float* __restrict__ gw, float* __restrict__ gb, int CY, int CX, int L,
                                                  int B, int NG) {
  extern __shared__ float4 smem4[];
  constexpr int PAD = (KS - 1) / 2, HALO = 8, TPW = 4;
  const int LP = L + 2 * HALO;
  const int NN = CX * KS, NT = (NN + 15) >> 4;
  const int mt = blockIdx.x / NG, ng = blockIdx.x - mt * NG;
  const int m0 = mt * 16;
  const int t0 = ng * 4 * TPW, t1 = min(NT, t0 + 4 * TPW);          // column tiles of this workgroup
  const int cx0 = (t0 * 16) / KS, cx1 = min(CX, ((t1 * 16 - 1) / KS) + 1), nxr = cx1 - cx0;   // input channels they touch
  float* ys = reinterpret_cast<float*>(smem4);      // 16 x L
  float* xs = ys + 16 * L;                          // nxr x LP, zero halo
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nxr * 2 * HALO; i += blockDim.x) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    xs[c * LP + (h < HALO ? h : L + h)] = 0.f;
  }
  f32x4 acc[TPW];
  int xoff[TPW];                                    // LDS offset of this lane's (input channel, tap) column, -1 if none
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nn = (t0 + wave + 4 * t) * 16 + r;
    const int cx = nn / KS, k = nn - cx * KS;
    xoff[t] = (t0 + wave + 4 * t < t1 && nn < NN) ? (cx - cx0) * LP + HALO + (CONVT ? PAD - k : k - PAD) : -1;
  }
  float bsum = 0.f;
  const int wpb = (B + gridDim.y - 1) / gridDim.y, w0 = blockIdx.y * wpb, w1 = min(B, w0 + wpb);
  const int ny = min(16, CY - m0);
  for (int win = w0; win < w1; ++win) {
    __syncthreads();
    for (int i = threadIdx.x; i < ny * (L >> 2); i += blockDim.x) {
      const int c = i / (L >> 2), p = (i - c * (L >> 2)) << 2;
      *reinterpret_cast<float4*>(ys + c * L + p) = *reinterpret_cast<const float4*>(Y + ((size_t)win * CY + m0 + c) * L + p);
    }
    for (int i = threadIdx.x; i < nxr * (L >> 2); i += blockDim.x) {
      const int c = i / (L >> 2), p = (i - c * (L >> 2)) << 2;
      *reinterpret_cast<float4*>(xs + c * LP + HALO + p) = *reinterpret_cast<const float4*>(X + ((size_t)win * CX + cx0 + c) * L + p);
    }
    __syncthreads();
    const float* yr = ys + (r < ny ? r : 0) * L;
    for (int p0 = 0; p0 < L; p0 += 4) {
      const float av = r < ny ? yr[p0 + g] : 0.f;
      bsum += av;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const float bv = xoff[t] >= 0 ? xs[xoff[t] + p0 + g] : 0.f;
        acc[t] = mfma4(av, bv, acc[t]);
      }
    }
  }
  // lane (r, g) of tile t holds rows m0 + 4 g + v of column nn = (t0 + wave + 4 t) 16 + r
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int nn = (t0 + wave + 4 * t) * 16 + r;
    if (xoff[t] < 0) continue;
    const int cx = nn / KS, k = nn - cx * KS;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int cy = m0 + 4 * g + v;
      if (cy < CY) atomicAdd(gw + (CONVT ? ((size_t)cx * CY + cy) * KS + k : (size_t)cy * NN + nn), acc[t][v]);
    }
  }
  if (gb && ng == 0 && wave == 0) {                 // bias gradient: row sums of Y (lane (r, g) saw positions g, g + 4, ...)
    bsum = rows_sum(bsum);
    if (g == 0 && r < ny) atomicAdd(gb + m0 + r, bsum);
  }
}

}  // namespace

// =================================================================================
// host model
// =================================================================================
struct AEntry { std::string name; int64_t offset; int ndim; int64_t shape[4]; };
struct ALayout {
  std::vector<AEntry> e;
  int64_t nparam = 0;
  int64_t ew[4], eb[4], dw[4], db[4], de[4];
};
static int64_t aalloc(int64_t& cur, int64_t n) { const int64_t o = cur; cur += (n + 3) & ~int64_t(3); return o; }
static void abuild(ALayout& L) {
  int64_t cur = 0;
  auto push = [&](const std::string& n, int64_t off, std::initializer_list<int64_t> shp) {
    AEntry e; e.name = n; e.offset = off; e.ndim = (int)shp.size();
    int i = 0;
    for (auto s : shp) e.shape[i++] = s;
    for (; i < 4; ++i) e.shape[i] = 1;
    L.e.push_back(e);
  };
  for (int i = 0; i < 4; ++i) {
    const std::string p = "EncList." + std::to_string(i) + ".conv.";
    L.ew[i] = aalloc(cur, (int64_t)ACH[i + 1] * ACH[i] * AKS[i]); L.eb[i] = aalloc(cur, ACH[i + 1]);
    push(p + "weight", L.ew[i], {ACH[i + 1], ACH[i], AKS[i]});
    push(p + "bias", L.eb[i], {ACH[i + 1]});
  }
  for (int i = 0; i < 4; ++i) {
    const std::string p = "DecList." + std::to_string(i) + ".";
    const int cin = ACH[4 - i], cout = ACH[3 - i], k = AKS[3 - i];
    L.dw[i] = aalloc(cur, (int64_t)cin * cout * k); L.db[i] = aalloc(cur, cout); L.de[i] = aalloc(cur, 3);
    push(p + "conv.weight", L.dw[i], {cin, cout, k});
    push(p + "conv.bias", L.db[i], {cout});
    push(p + "ECA.conv.weight", L.de[i], {1, 1, 3});
  }
  L.nparam = cur;
}

struct AcdaeModel {
  AcdaePublic pub;
  ALayout lay;
  char* slab = nullptr;
  // per window: encoder outputs e[i] (C_{i+1} x L / 2^{i+1}) + argmax bytes, decoder ECA inputs a[i], scales s[i], block
  // outputs d[i] (d[3] = y is the caller's), gradients: dc[i] (conv-output gradients, C_{i+1} x L / 2^i), dt[i], G tensors
  float *e[4], *a[4], *s[4], *d[3];
  unsigned char* am[4];
  float *dc[4], *dt[4], *gd[3], *ge[4];   // gd[i]: gradient at d[i]; ge[i]: input gradient of encoder i + 1 (at e[i])
  const float* last_x = nullptr;
  int last_B = 0;
};

int acdae_check_cfg(const ral_config* c, char* err, size_t cap) {
  if (c->leads != 2) { snprintf(err, cap, "ACDAE has 2 input channels (got leads=%d)", c->leads); return -1; }
  if (c->L <= 0 || c->L % 64 != 0 || c->L > 1024) { snprintf(err, cap, "ACDAE: L must be a multiple of 64 and <= 1024 (got %d)", c->L); return -1; }
  if (c->max_batch <= 0) { snprintf(err, cap, "max_batch must be positive"); return -1; }
  return 0;
}
int acdae_layout_count(const ral_config*) { ALayout L; abuild(L); return (int)L.e.size(); }
int acdae_layout_entry(const ral_config*, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset, int32_t* ndim,
                       int64_t shape[4]) {
  ALayout L; abuild(L);
  if (idx < 0 || idx >= (int)L.e.size() || (int)L.e[idx].name.size() + 1 > name_cap) return -1;
  strcpy(name, L.e[idx].name.c_str());
  *kind = RAL_PARAM; *offset = L.e[idx].offset; *ndim = L.e[idx].ndim;
  for (int i = 0; i < 4; ++i) shape[i] = L.e[idx].shape[i];
  return 0;
}
int64_t acdae_param_floats(const ral_config*) { ALayout L; abuild(L); return L.nparam; }

static size_t acdae_plan(const ral_config& c, AcdaeModel* m, char* base) {
  size_t cur = 0;
  const size_t B = c.max_batch;
  auto take = [&](size_t bytes) -> char* { char* p = base ? base + cur : nullptr; cur += (bytes + 255) & ~size_t(255); return p; };
  for (int i = 0; i < 4; ++i) {
    const size_t n = B * ACH[i + 1] * (c.L >> (i + 1));
    float* pe = reinterpret_cast<float*>(take(n * 4));
    unsigned char* pa = reinterpret_cast<unsigned char*>(take(n));
    if (m) { m->e[i] = pe; m->am[i] = pa; }
  }
  for (int i = 0; i < 4; ++i) {        // decoder i: COUT = ACH[3 - i], output length L >> (3 - i)
    const size_t n = B * ACH[3 - i] * (c.L >> (3 - i));
    float* pa = reinterpret_cast<float*>(take(n * 4));
    float* ps = reinterpret_cast<float*>(take(B * ACH[3 - i] * 4));
    float* pd = i < 3 ? reinterpret_cast<float*>(take(n * 4)) : nullptr;
    if (m) { m->a[i] = pa; m->s[i] = ps; if (i < 3) m->d[i] = pd; }
  }
  if (c.train) {
    for (int i = 0; i < 4; ++i) {
      float* p1 = reinterpret_cast<float*>(take(B * ACH[i + 1] * (c.L >> i) * 4));            // dc[i]
      float* p2 = reinterpret_cast<float*>(take(B * ACH[3 - i] * (c.L >> (4 - i)) * 4));      // dt[i]: COUT x Lin of decoder i
      float* p3 = i < 3 ? reinterpret_cast<float*>(take(B * ACH[3 - i] * (c.L >> (3 - i)) * 4)) : nullptr;   // gd[i]
      float* p4 = reinterpret_cast<float*>(take(B * ACH[i + 1] * (c.L >> (i + 1)) * 4));      // ge[i]: gradient at e[i]
      if (m) { m->dc[i] = p1; m->dt[i] = p2; if (i < 3) m->gd[i] = p3; m->ge[i] = p4; }
    }
  }
  return cur;
}
int64_t acdae_workspace_bytes(const ral_config* c) { return (int64_t)acdae_plan(*c, nullptr, nullptr); }

AcdaeModel* acdae_create(const ral_config* c, char* err, size_t cap) {
  AcdaeModel* m = new AcdaeModel();
  memset(&m->pub, 0, sizeof(m->pub));
  m->pub.cfg = *c;
  abuild(m->lay);
  m->pub.nparam = m->lay.nparam;
  const size_t bytes = acdae_plan(*c, nullptr, nullptr);
  if (hipMalloc(reinterpret_cast<void**>(&m->slab), bytes) != hipSuccess) {
    snprintf(err, cap, "hipMalloc(%zu) failed", bytes);
    delete m;
    return nullptr;
  }
  acdae_plan(*c, m, m->slab);
  return m;
}
void acdae_destroy(AcdaeModel* m) { if (m) { if (m->slab) (void)hipFree(m->slab); delete m; } }
AcdaePublic* acdae_public(AcdaeModel* m) { return &m->pub; }
int acdae_bind(AcdaeModel* m, float* params, float* grads, float* am, float* av) {
  m->pub.params = params; m->pub.grads = grads; m->pub.am = am; m->pub.av = av;
  return 0;
}

#define ACD_LDS(kernel, bytes) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

int acdae_forward(AcdaeModel* m, const float* x, float* y, int B, hipStream_t st, char* err, size_t cap) {
  AcdaePublic& P = m->pub;
  if (!P.params) { snprintf(err, cap, "ral_bind was not called"); return -1; }
  if (B <= 0 || B > P.cfg.max_batch) { snprintf(err, cap, "batch %d outside (0, %d]", B, P.cfg.max_batch); return -1; }
  const int L = P.cfg.L, grid = B < 1024 ? B : 1024;
  const ALayout& Y = m->lay;
  m->last_x = x; m->last_B = B;
  const float* in = x;
  for (int i = 0; i < 4; ++i) {
    const int cin = ACH[i], cout = ACH[i + 1], lin = L >> i;
    const size_t lds = (size_t)cin * (lin + 16) * sizeof(float);
    static const bool enc_mfma = ral_knob("ACDAE_ENC_MFMA", 1) != 0;
    if (AKS[i] == 13) { ACD_LDS(k_acd_enc_fwd<13>, lds); k_acd_enc_fwd<13><<<grid, 256, lds, st>>>(in, P.params + Y.ew[i], P.params + Y.eb[i], m->e[i], m->am[i], cin, cout, lin, B); }
    else if (enc_mfma && cin >= 16 && cout % 32 == 0 && lin % 16 == 0) {
      const size_t l2 = lds + (size_t)32 * cin * 7 * sizeof(float);
      const int gx = B < 512 ? B : 512;
      ACD_LDS(k_acd_enc_fwd_m<7>, l2);
      k_acd_enc_fwd_m<7><<<dim3(gx, cout / 32), 256, l2, st>>>(in, P.params + Y.ew[i], P.params + Y.eb[i], m->e[i], m->am[i], cin, cout, lin, B);
    }
    else { ACD_LDS(k_acd_enc_fwd<7>, lds); k_acd_enc_fwd<7><<<grid, 256, lds, st>>>(in, P.params + Y.ew[i], P.params + Y.eb[i], m->e[i], m->am[i], cin, cout, lin, B); }
    in = m->e[i];
  }
  for (int i = 0; i < 4; ++i) {
    const int cin = ACH[4 - i], cout = ACH[3 - i], lin = L >> (4 - i), ks = AKS[3 - i];
    const size_t lds = ((size_t)cin * (lin + 16) + (size_t)cout * lin * 3 + 2 * cout + 8) * sizeof(float);
    const float* skip = i < 3 ? m->e[2 - i] : nullptr;
    float* out = i < 3 ? m->d[i] : y;
    if (ks == 13) { ACD_LDS(k_acd_dec_fwd<13>, lds); k_acd_dec_fwd<13><<<grid, 256, lds, st>>>(in, P.params + Y.dw[i], P.params + Y.db[i], P.params + Y.de[i], skip, m->a[i], m->s[i], out, cin, cout, lin, B); }
    else { ACD_LDS(k_acd_dec_fwd<7>, lds); k_acd_dec_fwd<7><<<grid, 256, lds, st>>>(in, P.params + Y.dw[i], P.params + Y.db[i], P.params + Y.de[i], skip, m->a[i], m->s[i], out, cin, cout, lin, B); }
    in = out;
  }
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "ACDAE forward launch failed"); return -1; }
  return 0;
}

template <int KS, bool CONVT>
static void launch_acd_dw(const float* Yg, const float* Xg, float* gw, float* gb, int CY, int CX, int L, int B, hipStream_t st) {
  const size_t lds = ((size_t)8 * L + (size_t)8 * (L + 16)) * sizeof(float);
  const int nblk = ((CY + 7) / 8) * ((CX + 7) / 8);
  int splits = 2048 / nblk;                      // ~2048 workgroups per launch
  if (splits < 1) splits = 1;
  if (splits > B) splits = B;
  static const bool mfma_on = ral_knob("ACDAE_DW_MFMA", 1) != 0;
  if (mfma_on && CY >= 16) {
    const int NT = (CX * KS + 15) / 16, NG = (NT + 15) / 16, MT = (CY + 15) / 16;
    const int cxr = (16 * 16) / KS + 2 < CX ? (16 * 16) / KS + 2 : CX;          // input-channel rows a column group touches
    const size_t l2 = ((size_t)16 * L + (size_t)cxr * (L + 16)) * sizeof(float);
    int sp = 512 / (MT * NG);
    if (sp < 1) sp = 1;
    if (sp > B) sp = B;
    ACD_LDS((k_acd_dw_m<KS, CONVT>), l2);
    k_acd_dw_m<KS, CONVT><<<dim3(MT * NG, sp), 256, l2, st>>>(Yg, Xg, gw, gb, CY, CX, L, B, NG);
    return;
  }
  ACD_LDS((k_acd_dw<KS, CONVT>), lds);
  k_acd_dw<KS, CONVT><<<dim3(nblk, splits), 256, lds, st>>>(Yg, Xg, gw, gb, CY, CX, L, B);
}

int acdae_backward(AcdaeModel* m, const float* dy, float* dx, int B, hipStream_t st, char* err, size_t cap) {
  AcdaePublic& P = m->pub;
  if (!P.cfg.train || !P.grads) { snprintf(err, cap, "ACDAE backward needs train=1 and a bound gradient buffer"); return -1; }
  if (B != m->last_B) { snprintf(err, cap, "backward batch %d != forward batch %d", B, m->last_B); return -1; }
  const int L = P.cfg.L, grid = B < 1024 ? B : 1024;
  const ALayout& Y = m->lay;
  (void)hipMemsetAsync(P.grads, 0, (size_t)Y.nparam * sizeof(float), st);
  // decoder blocks, last to first: g = gradient at the block output
  const float* g = dy;
  for (int i = 3; i >= 0; --i) {
    const int cin = ACH[4 - i], cout = ACH[3 - i], lin = L >> (4 - i), ks = AKS[3 - i];
    const float* in = i == 0 ? m->e[3] : m->d[i - 1];
    float* gin = i == 0 ? m->ge[3] : m->gd[i - 1];
    const size_t lds = ((size_t)cout * lin * 2 + (size_t)cout * (lin + 16) + 3 * cout + 16) * sizeof(float);
    if (ks == 13) {
      ACD_LDS(k_acd_dec_bwd<13>, lds);
      k_acd_dec_bwd<13><<<grid, 256, lds, st>>>(g, m->a[i], m->s[i], P.params + Y.dw[i], P.params + Y.de[i], m->dt[i], gin, P.grads + Y.de[i], cin, cout, lin, B);
      launch_acd_dw<13, true>(m->dt[i], in, P.grads + Y.dw[i], P.grads + Y.db[i], cout, cin, lin, B, st);
    } else {
      ACD_LDS(k_acd_dec_bwd<7>, lds);
      k_acd_dec_bwd<7><<<grid, 256, lds, st>>>(g, m->a[i], m->s[i], P.params + Y.dw[i], P.params + Y.de[i], m->dt[i], gin, P.grads + Y.de[i], cin, cout, lin, B);
      launch_acd_dw<7, true>(m->dt[i], in, P.grads + Y.dw[i], P.grads + Y.db[i], cout, cin, lin, B, st);
    }
    g = gin;
  }
  // encoder blocks: gradient at e[i] = input gradient of block i + 1 (ge[i]) + the decoder's skip gradient (gd[2 - i])
  for (int i = 3; i >= 0; --i) {
    const int cin = ACH[i], cout = ACH[i + 1], lin = L >> i;
    const float* in = i == 0 ? m->last_x : m->e[i - 1];
    const float* g2 = i < 3 ? m->gd[2 - i] : nullptr;
    float* gin = i == 0 ? dx : m->ge[i - 1];
    const size_t lds = (size_t)cout * (lin + 16) * sizeof(float);
    if (AKS[i] == 13) {
      ACD_LDS(k_acd_enc_bwd<13>, lds);
      k_acd_enc_bwd<13><<<grid, 256, lds, st>>>(m->ge[i], g2, m->e[i], m->am[i], P.params + Y.ew[i], m->dc[i], gin, cin, cout, lin, B);
      launch_acd_dw<13, false>(m->dc[i], in, P.grads + Y.ew[i], P.grads + Y.eb[i], cout, cin, lin, B, st);
    } else {
      ACD_LDS(k_acd_enc_bwd<7>, lds);
      k_acd_enc_bwd<7><<<grid, 256, lds, st>>>(m->ge[i], g2, m->e[i], m->am[i], P.params + Y.ew[i], m->dc[i], gin, cin, cout, lin, B);
      launch_acd_dw<7, false>(m->dc[i], in, P.grads + Y.ew[i], P.grads + Y.eb[i], cout, cin, lin, B, st);
    }
  }
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "ACDAE backward launch failed"); return -1; }
  return 0;
}
