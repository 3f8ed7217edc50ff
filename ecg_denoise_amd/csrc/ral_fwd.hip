// Forward kernels of the RA-LENet path (gfx950).  One workgroup = one ECG window
// (or one window x head group for attention); every inter-kernel tensor of a window
// is 8*L floats, so a whole stage tile lives in LDS.
//
// Reference behaviour restated (see oracle/ralenet_oracle.py for the line map):
//   TransformerBlock  model/raletransformer.py:383-410, model/transformer.py:289-323
//   PatchMerging / PatchSeparate  model/raletransformer.py:411-459
#ifdef RAL_STAMP_TU_FWD
#define RAL_STAMP_HERE
#endif
#include "ral_device.hpp"
#include "ral_kernels.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

// =================================================================================
// K1: h = LN1(x*sqrt(C) + PE);  [q|k|v] = h Wqkv^T + b;  q *= 0.5  ->  qkv (HM layout)
// =================================================================================
template <int C>
__global__ __launch_bounds__(256) void k_qkv_fwd(const float* __restrict__ x, const float* __restrict__ pe,
                                                 BlockP w, float* __restrict__ qkv, int N, int B) {
  extern __shared__ float4 smem4[];
  float* Hs = reinterpret_cast<float*>(smem4);
  constexpr int LD = LDof<C>::v, LPR = C / 4, RPP = 256 / LPR;
  const float sqrtC = sqrtf((float)C);
  const int cq = (threadIdx.x % LPR) * 4;
  const float4 gam = *reinterpret_cast<const float4*>(w.ln1w + cq);
  const float4 bet = *reinterpret_cast<const float4*>(w.ln1b + cq);
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* xw = x + (size_t)win * N * C;
    for (int row = threadIdx.x / LPR; row < N; row += RPP) {
      float4 v = *reinterpret_cast<const float4*>(xw + row * C + cq);
      const float4 p = *reinterpret_cast<const float4*>(pe + row * C + cq);
      v = f4add(f4scale(v, sqrtC), p);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      *reinterpret_cast<float4*>(Hs + row * LD + cq) = f4add(f4mul(f4scale(d, rstd), gam), bet);
    }
    __syncthreads();
    float* qw = qkv + (size_t)win * 3 * N * C;
    gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(w.wqkv, C, 3 * C, Hs, LD, N >> 4, [&](int row0, int tok, f32x4 a) {
      float4 v = f4add(tofloat4(a), *reinterpret_cast<const float4*>(w.bqkv + row0));
      if (row0 < C) v = f4scale(v, 0.5f);  // q * head_dim^-0.5, head_dim = 4
      *reinterpret_cast<float4*>(qw + ((size_t)(row0 >> 2) * N + tok) * 4) = v;
    });
    __syncthreads();
  }
}

// =================================================================================
// K1h (wide levels, C >= 64): the same projection as two fp16 pieces per operand, three products per term on the f16
// matrix cores (gemm_wx_h2, ral_device.hpp: fp32-accurate to ~2^-21, 5.7 x fewer matrix cycles than the fp32 MFMA).
// wt: the tiled split planes of Wqkv (k_tile_planes).  A workgroup takes GT consecutive tokens of the batch at a time
// (whole windows or whole parts of one: N | GT or GT | N).  The general form; K1w below is the one the bench shapes take.
// =================================================================================
template <int C, int GT>
__global__ __launch_bounds__(256) void k_qkv_fwd_h(const float* __restrict__ x, const float* __restrict__ pe,
                                                      BlockP w, const _Float16* __restrict__ wt,
                                                      float* __restrict__ qkv, int N, int B) {
  extern __shared__ float4 smem4[];
  _Float16* Hh = reinterpret_cast<_Float16*>(smem4);          // 2 planes x GT x LDB
  constexpr int LDB = ldb_of(C), LPR = C / 4, RPP = 256 / LPR, TT = 4;
  constexpr int MT = (3 * C / 16) % 8 == 0 ? 2 : 1;
  constexpr int xplane = GT * LDB;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const float sqrtC = sqrtf((float)C);
  const int cq = (threadIdx.x % LPR) * 4;
  // (the LayerNorm affine carries the operand's power of two, the weight-plane unscale its inverse: ASC_LN1, ral_device.hpp)
  const float sg = asc_get(w.asc, ASC_LN1);
  const float4 gam = f4scale(*reinterpret_cast<const float4*>(w.ln1w + cq), sg);
  const float4 bet = f4scale(*reinterpret_cast<const float4*>(w.ln1b + cq), sg);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const long total = (long)B * N;
  const int ngroups = (int)((total + GT - 1) / GT);
  const float wun = wplane_unscale(wt, 3 * C, C) * asc_get(w.asc, ASC_LN1_INV);
  RAL_STAMP_INIT();
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    RAL_STAMP_AT(8);
    const long g0 = (long)grp * GT;
    for (int row = threadIdx.x / LPR; row < GT; row += RPP) {
      const long gt = g0 + row < total ? g0 + row : total - 1;
      const int tok = (int)(gt % N);
      float4 v = *reinterpret_cast<const float4*>(x + gt * C + cq);
      const float4 p = *reinterpret_cast<const float4*>(pe + tok * C + cq);
      v = f4add(f4scale(v, sqrtC), p);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      const float4 h = f4add(f4mul(f4scale(d, rstd), gam), bet);
      const H2 s0 = f16_split2(h.x), s1 = f16_split2(h.y), s2 = f16_split2(h.z), s3 = f16_split2(h.w);
      *reinterpret_cast<f16x4*>(Hh + row * LDB + cq) = f16x4{s0.a, s1.a, s2.a, s3.a};
      *reinterpret_cast<f16x4*>(Hh + xplane + row * LDB + cq) = f16x4{s0.b, s1.b, s2.b, s3.b};
    }
    RAL_STAMP_AT(9);
    __syncthreads();
    RAL_STAMP_AT(10);
    constexpr int MU = 3 * C / (16 * MT), TU = GT / (16 * TT);
    for (int u = wave; u < MU * TU; u += 4) {
      const int mu = u % MU, tu = u / MU;
      f32x4 acc[MT][TT], accx[MT][TT];
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { acc[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      gemm_wx_h2<C, MT, TT>(wt, C / 32, mu * MT, 0, Hh, xplane, LDB, tu * TT * 16, acc, accx);
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        const int row0 = (mu * MT + mi) * 16 + 4 * g;
        const float4 bias = *reinterpret_cast<const float4*>(w.bqkv + row0);
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
          const long gt = g0 + (tu * TT + tt) * 16 + r;
          if (gt < total) {
            const long win = gt / N;
            const int tok = (int)(gt - win * N);
            float4 v = f4add(f4scale(f4add(tofloat4(acc[mi][tt]), f4scale(tofloat4(accx[mi][tt]), RAL_H2_SCALE)), wun), bias);
            if (row0 < C) v = f4scale(v, 0.5f);  // q * head_dim^-0.5, head_dim = 4
            *reinterpret_cast<float4*>(qkv + win * 3 * N * C + ((size_t)(row0 >> 2) * N + tok) * 4) = v;
          }
        }
      }
    }
    RAL_STAMP_AT(11);
    __syncthreads();
    RAL_STAMP_AT(12);
  }
}

// =================================================================================
// K1w: the fp16-split projection with the WEIGHTS STATIONARY in registers.  A workgroup has one wave per 32 output rows
// (3C / 32 waves); a wave loads the two fp16 planes of its 32 x C weight block once (C / 2 registers) and keeps them for
// the whole kernel, so the only operand stream of the GEMM is the LDS tile of the current 64 tokens.  The token groups
// are double-buffered: the x rows of group i + 1 are requested before the products of group i are issued, normalised and
// split after them, and ONE barrier closes the iteration.  Positions repeat with period N | GT, so the positional rows a
// thread adds are loaded once.
// =================================================================================
template <int C, int GT>
__global__ __launch_bounds__(6 * C) void k_qkv_fwd_ws(const float* __restrict__ x, const float* __restrict__ pe,
                                                      BlockP w, const _Float16* __restrict__ wt,
                                                      float* __restrict__ qkv, int N, int B) {
  extern __shared__ float4 smem4[];
  constexpr int NT = 6 * C, LDB = ldb_of(C), LPR = C / 4, RPP = NT / LPR, NP = (GT + RPP - 1) / RPP, KC = C / 32;
  constexpr int xplane = GT * LDB, bufsz = 2 * xplane;
  _Float16* Hh = reinterpret_cast<_Float16*>(smem4);          // 2 buffers x 2 planes x GT x LDB
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const float sqrtC = sqrtf((float)C);
  const int cq = (threadIdx.x % LPR) * 4, rowt = threadIdx.x / LPR;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const long total = (long)B * N;
  const int ngroups = (int)((total + GT - 1) / GT);
  // the wave's weight block: rows 32 wave .. + 31, both planes, all of K
  f16x8 wf[2][KC][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int p = 0; p < 2; ++p) wf[mi][kc][p] = *reinterpret_cast<const f16x8*>(wtile(wt, KC, 2 * wave + mi, kc, p));
  const float sg = asc_get(w.asc, ASC_LN1);         // (as in k_qkv_fwd_h)
  const float wun = wplane_unscale(wt, 3 * C, C) * asc_get(w.asc, ASC_LN1_INV);
  const float4 gam = f4scale(*reinterpret_cast<const float4*>(w.ln1w + cq), sg);
  const float4 bet = f4scale(*reinterpret_cast<const float4*>(w.ln1b + cq), sg);
  float4 pev[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int row = rowt + k * RPP;
    pev[k] = *reinterpret_cast<const float4*>(pe + ((row < GT ? row : 0) % N) * C + cq);
  }
  float4 bias[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) bias[mi] = *reinterpret_cast<const float4*>(w.bqkv + 32 * wave + 16 * mi + 4 * g);
  float4 xv[NP];
  auto request = [&](int grp) {
    const long g0 = (long)grp * GT;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int row = rowt + k * RPP;
      long gt = g0 + (row < GT ? row : 0);
      if (gt >= total) gt = total - 1;
      xv[k] = *reinterpret_cast<const float4*>(x + gt * C + cq);
    }
  };
  auto stage = [&](_Float16* buf) {   // LayerNorm + split of the requested rows -> buf
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int row = rowt + k * RPP;
      const float4 v = f4add(f4scale(xv[k], sqrtC), pev[k]);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      const float4 h = f4add(f4mul(f4scale(d, rstd), gam), bet);
      const H2 s0 = f16_split2(h.x), s1 = f16_split2(h.y), s2 = f16_split2(h.z), s3 = f16_split2(h.w);
      if (row < GT) {
        *reinterpret_cast<f16x4*>(buf + row * LDB + cq) = f16x4{s0.a, s1.a, s2.a, s3.a};
        *reinterpret_cast<f16x4*>(buf + xplane + row * LDB + cq) = f16x4{s0.b, s1.b, s2.b, s3.b};
      }
    }
  };
  int grp = blockIdx.x, it = 0;
  if (grp < ngroups) { request(grp); stage(Hh); }
  __syncthreads();
  for (; grp < ngroups; grp += gridDim.x, ++it) {
    const _Float16* cur = Hh + (it & 1) * bufsz;
    const int nxt = grp + gridDim.x;
    if (nxt < ngroups) request(nxt);
    const long g0 = (long)grp * GT;
#pragma unroll 1
    for (int tt = 0; tt < GT / 16; ++tt) {
      f32x4 acc[2], accx[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) { acc[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[mi] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      const _Float16* xr = cur + (16 * tt + r) * LDB + 8 * g;
      f16x8 b1[KC], b2[KC];
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        b1[kc] = *reinterpret_cast<const f16x8*>(xr + kc * 32);
        b2[kc] = *reinterpret_cast<const f16x8*>(xr + xplane + kc * 32);
      }
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][kc][1], b1[kc], accx[mi], 0, 0, 0);
          acc[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][kc][0], b1[kc], acc[mi], 0, 0, 0);
          accx[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mi][kc][0], b2[kc], accx[mi], 0, 0, 0);
        }
      const long gt = g0 + 16 * tt + r;
      if (gt < total) {
        const long win = gt / N;
        const int tok = (int)(gt - win * N);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row0 = 32 * wave + 16 * mi + 4 * g;
          float4 v = f4add(f4scale(f4add(tofloat4(acc[mi]), f4scale(tofloat4(accx[mi]), RAL_H2_SCALE)), wun), bias[mi]);
          if (row0 < C) v = f4scale(v, 0.5f);  // q * head_dim^-0.5, head_dim = 4
          *reinterpret_cast<float4*>(qkv + win * 3 * N * C + ((size_t)(row0 >> 2) * N + tok) * 4) = v;
        }
      }
    }
    if (nxt < ngroups) stage(Hh + ((it + 1) & 1) * bufsz);
    __syncthreads();
  }
}

// The weight matrices of the levels that run on the f16 matrix cores -> tiled split planes (layout: wtile, ral_device.hpp).
// desc[d] = {offset of the matrix in the parameter buffer (floats), rows M, columns K, first work item}; a work item is
// eight consecutive columns of one row (16 bytes of each plane); the tiled planes of a matrix take the bytes of the
// matrix's own place in a buffer shaped like the parameter buffer (2 planes x 2 bytes = 4 bytes per weight).
// Every matrix is multiplied by ITS OWN power of two first (k_weight_scales: largest |w| into [2^13, 2^14)), so that the
// pieces carry 22 bits relative to the matrix's scale whatever that scale is (weights of 1e-7 were 11-bit numbers with a
// fixed factor: fp16 has nothing below 6e-8) and nothing can reach fp16's upper end.  The inverse sits in the first bytes
// of the slot BEHIND the matrix's planes (the place of the bias that follows every weight matrix in the parameter buffer;
// wplane_unscale, ral_device.hpp) and the consumers multiply their accumulators by it.
// unscaled: the backward's planes carry unscaled residuals (products with operands that are scaled themselves go into one
// accumulator, ral_device.hpp), the forward's the 2^11-scaled residual
// (one workgroup per matrix, 1024 threads with four 16-byte loads in flight each: the largest matrix, 64 K floats, is four
// round trips - with 256 threads and one load per iteration it was 64 and the kernel 23 us, twice per step)
__global__ __launch_bounds__(1024) void k_weight_scales(const float* __restrict__ params, _Float16* __restrict__ wt, const int4* __restrict__ desc) {
  const int4 D = desc[blockIdx.x];
  const int n4 = (D.y * D.z) >> 2;
  const float4* w4 = reinterpret_cast<const float4*>(params + D.x);
  float m = 0.f;
  for (int i0 = 0; i0 < n4; i0 += 4 * 1024) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = i0 + k * 1024 + (int)threadIdx.x; v[k] = w4[i < n4 ? i : 0]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k].x), fabsf(v[k].y)), fmaxf(fabsf(v[k].z), fabsf(v[k].w))));
  }
  __shared__ float red[16];
  m = group_max<64>(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = red[0];
#pragma unroll
    for (int k = 1; k < 16; ++k) mm = fmaxf(mm, red[k]);
    const unsigned bits = __float_as_uint(mm);
    float* slot = reinterpret_cast<float*>(wt + 2 * ((size_t)D.x + (size_t)D.y * D.z));
    slot[0] = h2_row_unscale(bits);
    slot[1] = h2_row_scale(bits);
  }
}
__global__ void k_tile_planes(const float* __restrict__ params, _Float16* __restrict__ wt, const int4* __restrict__ desc,
                              int ndesc, int nwork, int unscaled) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nwork; i += gridDim.x * blockDim.x) {
    int d = 0;
    while (d + 1 < ndesc && desc[d + 1].w <= i) ++d;
    const int4 D = desc[d];
    const float scale = reinterpret_cast<const float*>(wt + 2 * ((size_t)D.x + (size_t)D.y * D.z))[1];
    const int j = i - D.w, K8 = D.z >> 3, row = j / K8, k8 = j - row * K8;
    const float4 v0 = *reinterpret_cast<const float4*>(params + D.x + (size_t)row * D.z + 8 * k8);
    const float4 v1 = *reinterpret_cast<const float4*>(params + D.x + (size_t)row * D.z + 8 * k8 + 4);
    const float xs[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    f16x8 h1, h2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const H2 s2 = unscaled ? f16_split2u(xs[e] * scale) : f16_split2(xs[e] * scale); h1[e] = s2.a; h2[e] = s2.b; }
    const int mt = row >> 4, r = row & 15, kt = k8 >> 2, g = k8 & 3;
    _Float16* dst = wt + 2 * (size_t)D.x + ((size_t)(mt * (D.z >> 5) + kt) * 2) * 512 + (g * 16 + r) * 8;
    *reinterpret_cast<f16x8*>(dst) = h1;
    *reinterpret_cast<f16x8*>(dst + 512) = h2;
  }
}
// Activation scales of every transformer block (ASC_*, ral_device.hpp): one workgroup per block.  desc[b] = 12 ints: C, then the
// float offsets of ln1w, ln1b, wqkv, bqkv, ln2w, ln2b, w1, b1, le (-1: no local enhancement), 0, 0.  Row norms with 16 lanes per
// row (four rows per wave and pass).
__global__ __launch_bounds__(1024) void k_act_scales(const float* __restrict__ params, const int* __restrict__ desc, float* __restrict__ asc) {
  const int* D = desc + 12 * blockIdx.x;
  const int C = D[0];
  __shared__ float red[8][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l16 = lane & 15, sub = lane >> 4;
  // 0: max|g1| 1: max|b1ln| 2: |b1ln|^2 3: max|g2| 4: max|b2ln| 5: |b2ln|^2
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < C; i += blockDim.x) {
    const float g1 = params[D[1] + i], b1 = params[D[2] + i], g2 = params[D[5] + i], b2 = params[D[6] + i];
    v[0] = fabsf(g1); v[1] = fabsf(b1); v[2] = b1 * b1; v[3] = fabsf(g2); v[4] = fabsf(b2); v[5] = b2 * b2;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float r_ = (k == 2 || k == 5) ? group_sum<64>(v[k]) : group_max<64>(v[k]);
    if (lane == 0) red[k][wave] = r_;
  }
  __syncthreads();
  float st[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    float a = red[k][0];
    for (int w_ = 1; w_ < 16; ++w_) a = (k == 2 || k == 5) ? a + red[k][w_] : fmaxf(a, red[k][w_]);
    st[k] = a;
  }
  const float sqC = sqrtf((float)C);
  const float e1 = sqC * st[0] + st[1], n1 = sqC * st[0] + sqrtf(st[2]);      // element bound / L2 bound of LN1's output
  const float e2 = sqC * st[3] + st[4], n2 = sqC * st[3] + sqrtf(st[5]);
  // max_j (|W_j|_2 * nrm + |b_j|) over the rows of a (rows x C) matrix
  auto row_bound = [&](const float* W, const float* bias, int rows, float nrm) -> float {
    float best = 0.f;
    for (int r0 = (wave * 4 + sub); r0 < rows; r0 += 64) {
      float ss = 0.f;
      for (int c = l16 * 4; c < C; c += 64) {
        const float4 q = *reinterpret_cast<const float4*>(W + (size_t)r0 * C + c);
        ss += f4dot(q, q);
      }
      ss = group_sum<16>(ss);
      best = fmaxf(best, sqrtf(ss) * nrm + fabsf(bias[r0]));
    }
    return group_max<64>(best);
  };
  const float bv = row_bound(params + D[3] + 2 * C * C, params + D[4] + 2 * C, C, n1);
  const float bu = row_bound(params + D[7], params + D[8], 4 * C, n2);
  __syncthreads();
  if (lane == 0) { red[6][wave] = bv; red[7][wave] = bu; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float mv = red[6][0], mu = red[7][0];
    for (int w_ = 1; w_ < 16; ++w_) { mv = fmaxf(mv, red[6][w_]); mu = fmaxf(mu, red[7][w_]); }
    if (D[9] >= 0) {
      const float sl = fabsf(params[D[9]]) + fabsf(params[D[9] + 1]) + fabsf(params[D[9] + 2]);
      mu *= fmaxf(1.0f, sl);
    }
    const float bnd[4] = {e1, mv, e2, mu};
    float* out = asc + ASC_N * blockIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned bits = __float_as_uint(bnd[k] * 1.001f);    // (the bounds' own rounding)
      out[2 * k] = h2_row_scale(bits);
      out[2 * k + 1] = h2_row_unscale(bits);
    }
  }
}
void launch_act_scales(const float* params, const void* desc, float* asc, int nblk, hipStream_t s) {
  k_act_scales<<<nblk, 1024, 0, s>>>(params, reinterpret_cast<const int*>(desc), asc);
}

void launch_tile_planes(const float* params, void* wt, const void* desc, int ndesc, int nwork, int unscaled, hipStream_t s) {
  if (ndesc <= 0) return;
  k_weight_scales<<<ndesc, 1024, 0, s>>>(params, reinterpret_cast<_Float16*>(wt), reinterpret_cast<const int4*>(desc));
  const int blocks = (nwork + 255) / 256;
  k_tile_planes<<<blocks < 2048 ? blocks : 2048, 256, 0, s>>>(params, reinterpret_cast<_Float16*>(wt), reinterpret_cast<const int4*>(desc), ndesc, nwork, unscaled);
}

// =================================================================================
// K2: softmax(q k^T + bias) v per (window, head); full N x N, head_dim 4.  Nothing
// N x N is ever stored.  S^T tiles come from the fp32 MFMA with the KEY on the row and
// the QUERY on the lane column: a lane owns one query and four keys per tile, so the
// whole online-softmax state (running max, row sum, P.V) is LANE-PRIVATE on the VALU
// (packed fp32 math) and the four lane groups of a query are merged once per block.
// q is pre-multiplied by log2(e) when staged, so p = exp2(s - m) needs no scaling.
// =================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
RAL_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
RAL_DEV f32x2 splat2(float v) { return f32x2{v, v}; }
#define RAL_LOG2E 1.4426950408889634f
#define RAL_LN2 0.6931471805599453f

RAL_STAMPS_DEFINE(ral_debug_stamps_fwd)

// Softmax shift without a running maximum: softmax is invariant to any per-row shift, and
//   s[q][k] = q.k + bias <= |q| max_k |k| + max(bias, 0) =: m[q]   (Cauchy-Schwarz)
// is known before the sweep.  -m[q] is fed to the MFMA as its C operand, so the tile comes out
// as s - m <= 0 ready for exp2 with no VALU subtract, no max and no rescale.  If a row's bound is
// so loose that every term underflows (row sum < 1e-30) the task is redone with the exact
// running-max recurrence (never seen on real data; forced by tests/test_gpu_configs.py::test_attention_forward_exact_fallback).
// NT > 0: window length as a compile-time constant, TAB = false: no R-wave table (short windows: a task is 2-4 tiles, its
// set-up and loop control weigh as much as the tiles - see k_attn_bwd)
// F16: the S tile on the f16 matrix cores - q log2 e and k staged as token-interleaved fp16-pair planes ([h1 x 4 | h2 x 4],
// 16 bytes per token as the fp32 quad they replace), one v_mfma_f32_16x16x16_f16 per tile (lane group g' = piece pair
// (g' >> 1, g' & 1) over the four dims; see k_attn_bwd_w in ral_attn.hip), 10.6 instead of 41.7 cycles beside the tile's
// vector work (tools/diag/valu_probe.hip)
typedef _Float16 fh16x4 __attribute__((ext_vector_type(4)));
RAL_DEV void put_pair_planes(float* X, int t, float4 x) {
  const H2 s0 = f16_split2n(x.x), s1 = f16_split2n(x.y), s2 = f16_split2n(x.z), s3 = f16_split2n(x.w);
  *reinterpret_cast<fh16x4*>(X + 4 * t) = fh16x4{s0.a, s1.a, s2.a, s3.a};
  *reinterpret_cast<fh16x4*>(X + 4 * t + 2) = fh16x4{s0.b, s1.b, s2.b, s3.b};
}
// RAG: only the first NE of the window's N token slots exist (a window length that is not a multiple of 256 runs on padded
// slots, ral_api.hip): keys past NE are masked out of the softmax (s = -inf), the R-wave window is centred in the NE tokens;
// the padding queries are computed like any other (their rows are never used)
template <int QT, int NT = 0, bool TAB = true, bool F16 = false, bool RAG = false>
__global__ __launch_bounds__(512, (QT >= 4 ? 3 : 4)) void k_attn_fwd(const float* __restrict__ qkv, float* __restrict__ o_hm,
                                                  float* __restrict__ lse, const float* __restrict__ table,
                                                  int N_rt, int H, int HG, int Len, int B, int NE_rt = 0) {
  extern __shared__ float4 smem4[];
  const int N = NT ? NT : N_rt;
  const int NE = RAG ? NE_rt : N, NEt = RAG ? ((NE + 15) & ~15) : N;   // existing keys; key tiles that hold one
  if constexpr (!TAB) { table = nullptr; Len = 0; }
  float* Qs = reinterpret_cast<float*>(smem4);
  float* Ks = Qs + HG * N * 4;
  float* Vs = Ks + HG * N * 4;
  float* Mq = Vs + HG * N * 4;                  // HG*N : |q| (log2 units)
  int* Kmax = reinterpret_cast<int*>(Mq + HG * N);  // HG : max |k|^2 as float bits
  float* tab = reinterpret_cast<float*>(Kmax + HG + 4);  // (2Len-1) x HG, times log2(e)
  float* Bmax = tab + (table ? (2 * Len - 1) * HG : 0);   // HG : max(bias, 0)
  int* Qmax = reinterpret_cast<int*>(Bmax + HG);          // HG : max |q|^2 as float bits (F16: balances the pair planes)
  const int ngrp = H / HG;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int off = (NE - Len) >> 1;
  const int kb0 = table ? (off & ~15) : NEt, kb1 = table ? ((off + Len + 15) & ~15) : NEt;  // biased key tiles
  for (int item = blockIdx.x; item < B * ngrp; item += gridDim.x) {
    const int win = item / ngrp, h0 = (item - win * ngrp) * HG;
    const float* base = qkv + (size_t)win * 3 * H * N * 4;
    if ((int)threadIdx.x < HG) { Kmax[threadIdx.x] = 0; Qmax[threadIdx.x] = 0; if (table) Bmax[threadIdx.x] = 0.f; }
    __syncthreads();
    {   // one staging pass with the q, k and v loads of two indices in flight together
      const float4* gq = reinterpret_cast<const float4*>(base + (size_t)h0 * N * 4);
      const float4* gk = reinterpret_cast<const float4*>(base + (size_t)(H + h0) * N * 4);
      const float4* gv = reinterpret_cast<const float4*>(base + (size_t)(2 * H + h0) * N * 4);
      const int n4 = HG * N, bd = blockDim.x;
      int i = threadIdx.x;
      for (; i + bd < n4; i += 2 * bd) {
        float4 q0 = gq[i], q1 = gq[i + bd];
        const float4 k0 = gk[i], k1 = gk[i + bd], v0 = gv[i], v1 = gv[i + bd];
        q0 = f4scale(q0, RAL_LOG2E); q1 = f4scale(q1, RAL_LOG2E);
        reinterpret_cast<float4*>(Qs)[i] = q0; reinterpret_cast<float4*>(Qs)[i + bd] = q1;
        reinterpret_cast<float4*>(Ks)[i] = k0; reinterpret_cast<float4*>(Ks)[i + bd] = k1;
        if constexpr (F16) { atomicMax(Qmax + i / N, __float_as_int(f4dot(q0, q0))); atomicMax(Qmax + (i + bd) / N, __float_as_int(f4dot(q1, q1))); }
        reinterpret_cast<float4*>(Vs)[i] = v0; reinterpret_cast<float4*>(Vs)[i + bd] = v1;
        Mq[i] = sqrtf(f4dot(q0, q0)); Mq[i + bd] = sqrtf(f4dot(q1, q1));
        atomicMax(Kmax + i / N, __float_as_int(f4dot(k0, k0)));   // non-negative floats order like ints
        atomicMax(Kmax + (i + bd) / N, __float_as_int(f4dot(k1, k1)));
      }
      for (; i < n4; i += bd) {
        const float4 q0 = f4scale(gq[i], RAL_LOG2E), k0 = gk[i], v0 = gv[i];
        reinterpret_cast<float4*>(Qs)[i] = q0; reinterpret_cast<float4*>(Ks)[i] = k0; reinterpret_cast<float4*>(Vs)[i] = v0;
        if constexpr (F16) atomicMax(Qmax + i / N, __float_as_int(f4dot(q0, q0)));
        Mq[i] = sqrtf(f4dot(q0, q0));
        atomicMax(Kmax + i / N, __float_as_int(f4dot(k0, k0)));
      }
    }
    if (table)
      for (int i = threadIdx.x; i < (2 * Len - 1) * HG; i += blockDim.x) {
        const float t = table[(i / HG) * H + h0 + (i % HG)] * RAL_LOG2E;
        tab[i] = t;
        if (t > 0.f) atomicMax(reinterpret_cast<int*>(Bmax) + i % HG, __float_as_int(t));
      }
    __syncthreads();
    if constexpr (F16) {   // second pass: q and k of a head balanced by one power of two (pair_balance) and split IN PLACE
      for (int i = threadIdx.x; i < HG * N; i += blockDim.x) {
        float cq, ck;
        pair_balance(sqrtf(__int_as_float(Qmax[i / N])), sqrtf(__int_as_float(Kmax[i / N])), cq, ck);
        const float4 q = reinterpret_cast<const float4*>(Qs)[i], k = reinterpret_cast<const float4*>(Ks)[i];
        put_pair_planes(Qs, i, f4scale(q, cq)); put_pair_planes(Ks, i, f4scale(k, ck));
      }
      __syncthreads();
    }
    const int qblocks = N / (16 * QT);
    for (int task = wave; task < HG * qblocks; task += nw) {
      const int hl = task / qblocks, q0 = (task - hl * qblocks) * 16 * QT;
      const float* Qh = Qs + hl * N * 4;
      const float* Kh = Ks + hl * N * 4;
      const float4* Vh = reinterpret_cast<const float4*>(Vs + hl * N * 4);
      const float kmx = sqrtf(__int_as_float(Kmax[hl])) * 1.0000002f;
      float qf[QT], mq[QT];
      fh16x4 qh[QT];
      f32x2 l2[QT], o01[QT], o23[QT];
      // MFMA operands of a token: fp32 element g of its quad, or (F16) plane g >> 1 (A, row token) / g & 1 (B, column token)
      auto sc_tile = [&](int kt, int qt, f32x4 c) -> f32x4 {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(*reinterpret_cast<const fh16x4*>(Kh + 4 * (kt + r) + 2 * (g >> 1)), qh[qt], c, 0, 0, 0);
        else return mfma4(Kh[(kt + r) * 4 + g], qf[qt], c);
      };
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + 16 * qt + r;
        if constexpr (F16) qh[qt] = *reinterpret_cast<const fh16x4*>(Qh + 4 * q + 2 * (g & 1));
        else qf[qt] = Qh[q * 4 + g];
        mq[qt] = Mq[hl * N + q] * kmx + (table ? Bmax[hl] : 0.f);
        l2[qt] = f32x2{0.f, 0.f}; o01[qt] = f32x2{0.f, 0.f}; o23[qt] = f32x2{0.f, 0.f};
      }
      auto tile = [&](int kt, auto biased) {
        float4 v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v4[j] = Vh[kt + 4 * g + j];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float nm = -mq[qt];
          f32x4 s = sc_tile(kt, qt, f32x4{nm, nm, nm, nm});   // s - m, log2 units
          if constexpr (RAG) {
            if (kt + 16 > NE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) s[j] = (kt + 4 * g + j < NE) ? s[j] : -INFINITY;
            }
          }
          if constexpr (decltype(biased)::value) {
            const int qi = q0 + 16 * qt + r - off;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ki = kt + 4 * g + j - off;
              if (qi >= 0 && qi < Len && ki >= 0 && ki < Len) s[j] += tab[(qi - ki + Len - 1) * HG + hl];
            }
          }
          const f32x2 p01 = f32x2{__builtin_amdgcn_exp2f(s[0]), __builtin_amdgcn_exp2f(s[1])};
          const f32x2 p23 = f32x2{__builtin_amdgcn_exp2f(s[2]), __builtin_amdgcn_exp2f(s[3])};
          l2[qt] += p01;
          l2[qt] += p23;
          o01[qt] = pk_fma(splat2(p01[0]), f32x2{v4[0].x, v4[0].y}, o01[qt]);
          o23[qt] = pk_fma(splat2(p01[0]), f32x2{v4[0].z, v4[0].w}, o23[qt]);
          o01[qt] = pk_fma(splat2(p01[1]), f32x2{v4[1].x, v4[1].y}, o01[qt]);
          o23[qt] = pk_fma(splat2(p01[1]), f32x2{v4[1].z, v4[1].w}, o23[qt]);
          o01[qt] = pk_fma(splat2(p23[0]), f32x2{v4[2].x, v4[2].y}, o01[qt]);
          o23[qt] = pk_fma(splat2(p23[0]), f32x2{v4[2].z, v4[2].w}, o23[qt]);
          o01[qt] = pk_fma(splat2(p23[1]), f32x2{v4[3].x, v4[3].y}, o01[qt]);
          o23[qt] = pk_fma(splat2(p23[1]), f32x2{v4[3].z, v4[3].w}, o23[qt]);
        }
      };
      // only query blocks that touch the centred R-wave window take the biased key tiles
      const bool qbias = table && (q0 < off + Len) && (q0 + 16 * QT > off);
      const int e0 = qbias ? kb0 : NEt, e1 = qbias ? kb1 : NEt;
      for (int kt = 0; kt < e0; kt += 16) tile(kt, std::false_type{});
      for (int kt = e0; kt < e1; kt += 16) tile(kt, std::true_type{});
      for (int kt = e1; kt < NEt; kt += 16) tile(kt, std::false_type{});
      bool redo = false;
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float lv = l2[qt][0] + l2[qt][1];
        float4 ov = make_float4(o01[qt][0], o01[qt][1], o23[qt][0], o23[qt][1]);
        lv = rows_sum(lv);
        ov = make_float4(rows_sum(ov.x), rows_sum(ov.y), rows_sum(ov.z), rows_sum(ov.w));
        redo = redo || !(lv > 1e-30f);
        if (g == 0) {
          const float inv = 1.0f / lv;
          const int q = q0 + 16 * qt + r;
          const size_t hq = ((size_t)win * H + h0 + hl) * N + q;
          *reinterpret_cast<float4*>(o_hm + hq * 4) = f4scale(ov, inv);
          if (lse) lse[hq] = (mq[qt] + __builtin_amdgcn_logf(lv)) * RAL_LN2;   // natural-log units
        }
      }
      if (__any(redo)) {   // exact running-max recurrence (lane-private state, merged at the end)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          float mx = -INFINITY, l = 0.f;
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
          const int qi = q0 + 16 * qt + r - off;
          for (int kt = 0; kt < NEt; kt += 16) {
            f32x4 s = sc_tile(kt, qt, f32x4{0.f, 0.f, 0.f, 0.f});
            if constexpr (RAG) {
#pragma unroll
              for (int j = 0; j < 4; ++j) s[j] = (kt + 4 * g + j < NE) ? s[j] : -INFINITY;
            }
            if (table) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int ki = kt + 4 * g + j - off;
                if (qi >= 0 && qi < Len && ki >= 0 && ki < Len) s[j] += tab[(qi - ki + Len - 1) * HG + hl];
              }
            }
            const float mn = fmaxf(fmaxf(mx, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
            const float corr = (RAG && mx == mn) ? 1.0f : __builtin_amdgcn_exp2f(mx - mn);   // (RAG: a lane that has seen masked keys only holds -inf)
            mx = mn;
            l *= corr; o = f4scale(o, corr);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = (RAG && s[j] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s[j] - mn);
              const float4 v = Vh[kt + 4 * g + j];
              l += p;
              o.x = fmaf(p, v.x, o.x); o.y = fmaf(p, v.y, o.y); o.z = fmaf(p, v.z, o.z); o.w = fmaf(p, v.w, o.w);
            }
          }
          float mg = fmaxf(mx, __shfl_xor(mx, 16));
          mg = fmaxf(mg, __shfl_xor(mg, 32));
          const float sc = (RAG && mx == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mx - mg);
          l *= sc; o = f4scale(o, sc);
#pragma unroll
          for (int sh = 16; sh <= 32; sh <<= 1) {
            l += __shfl_xor(l, sh);
            o.x += __shfl_xor(o.x, sh); o.y += __shfl_xor(o.y, sh);
            o.z += __shfl_xor(o.z, sh); o.w += __shfl_xor(o.w, sh);
          }
          if (g == 0) {
            const int q = q0 + 16 * qt + r;
            const size_t hq = ((size_t)win * H + h0 + hl) * N + q;
            *reinterpret_cast<float4*>(o_hm + hq * 4) = f4scale(o, 1.0f / l);
            if (lse) lse[hq] = (mg + __builtin_amdgcn_logf(l)) * RAL_LN2;
          }
        }
      }
    }
    __syncthreads();
  }
}

// =================================================================================
// K2v: the same attention with one QUERY PER LANE and the keys streamed through the SCALAR path (N >= 64).
// A wave owns 64 queries of one (window, head); key / value rows are wave-uniform, so they come in by s_load
// (scalar cache, no LDS, no barrier, no staging pass) and enter the FMAs as SGPR operands:
//   per key and lane:  s = q.k - m  (2 packed FMA + 1 add, -m is the first addend)   p = exp2(s)   l += p
//                      o += p v  (2 packed FMA)
// Packed fp32 instructions take an SGPR PAIR at full rate (4.3 cycles per 128 FMAs), whereas v_fma_f32 with an SGPR
// operand drops to half rate (4.2 cycles per 64, tools/diag/valu_probe.hip): every product here is a v_pk_* one.
// Everything a query needs is lane-private, so there is no lane-group merge, and the output rows leave as one
// coalesced 16-byte store per lane.  On gfx950 the fp32 MFMA shares the fp32 multipliers with the vector ALU (a tile of
// k_attn_fwd costs its VALU time PLUS its MFMA time, tools/diag/valu_probe.hip), so doing q.k on the vector ALU costs
// no more than the MFMA did, and the tile bookkeeping of the MFMA form disappears.
// The softmax shift is the same Cauchy-Schwarz bound as above (exact running-max redo if a row underflows).
// =================================================================================

template <bool BIAS>
__global__ __launch_bounds__(256) void k_attn_fwd_v(const float* __restrict__ qkv, float* __restrict__ o_hm,
                                                    float* __restrict__ lse, const float* __restrict__ table,
                                                    int N, int H, int Len, int ntask) {
  const int lane = threadIdx.x & 63;
  const int QB = (N + 63) >> 6;
  const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // wave-uniform by construction
  if (task >= ntask) return;
  const int qb = task % QB, wh = task / QB, head = wh % H, win = wh / H;
  const int q = qb * 64 + lane, qc = q < N ? q : N - 1;
  const size_t wbase = (size_t)win * 3 * H * N;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)head * N;
  const float4* __restrict__ K4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(H + head) * N;
  const float4* __restrict__ V4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(2 * H + head) * N;
  const float4 qv = f4scale(Q4[qc], RAL_LOG2E);
  // max_k |k|^2 of the head: every lane takes N / 64 keys, then a wave-wide max
  float km = 0.f;
  for (int i = lane; i < N; i += 64) { const float4 k = K4[i]; km = fmaxf(km, f4dot(k, k)); }
  km = group_max<64>(km);
  float tval = 0.f, bmax = 0.f;      // R-wave table of this head, entry `lane` (2 Len - 1 <= 63 entries), log2 units
  const int off = (N - Len) >> 1, qi = q - off;
  if constexpr (BIAS) {
    if (lane < 2 * Len - 1) tval = table[lane * H + head] * RAL_LOG2E;
    bmax = group_max<64>(fmaxf(tval, 0.f));
  }
  const float mq = sqrtf(f4dot(qv, qv)) * sqrtf(km) * 1.0000002f + bmax;
  const float nm = -mq;
  float l = 0.f;
  f32x2 o01 = {0.f, 0.f}, o23 = {0.f, 0.f};
  const f32x2 q01 = {qv.x, qv.y}, q23 = {qv.z, qv.w}, nm0 = {nm, 0.f};
  auto body = [&](int kt, auto biased) {     // 4 keys; K4 / V4 rows are wave-uniform (scalar loads)
    float4 k[4], v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { k[j] = K4[kt + j]; v[j] = V4[kt + j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 t = pk_fma(q01, f32x2{k[j].x, k[j].y}, nm0);
      t = pk_fma(q23, f32x2{k[j].z, k[j].w}, t);
      float s = t[0] + t[1];
      if constexpr (decltype(biased)::value) {
        const int ki = kt + j - off;                       // uniform; in the window for these chunks except at ragged ends
        const float b = __shfl(tval, qi - ki + Len - 1);   // (lane-indexed gather of the table entry)
        if (ki >= 0 && ki < Len && qi >= 0 && qi < Len) s += b;
      }
      const float p = __builtin_amdgcn_exp2f(s);
      l += p;
      o01 = pk_fma(splat2(p), f32x2{v[j].x, v[j].y}, o01);
      o23 = pk_fma(splat2(p), f32x2{v[j].z, v[j].w}, o23);
    }
  };
  const int b0 = BIAS ? (off & ~3) : N, b1 = BIAS ? ((off + Len + 3) & ~3) : N;
  for (int kt = 0; kt < b0; kt += 4) body(kt, std::false_type{});
  for (int kt = b0; kt < b1; kt += 4) body(kt, std::true_type{});
  for (int kt = b1; kt < N; kt += 4) body(kt, std::false_type{});
  float mfin = mq;
  if (__any(!(l > 1e-30f))) {   // a row underflowed under its bound: exact running-max recurrence for the whole wave
    float mx = -INFINITY, o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
    l = 0.f;
    for (int kt = 0; kt < N; ++kt) {
      const float4 k = K4[kt], v = V4[kt];
      float s = fmaf(qv.x, k.x, fmaf(qv.y, k.y, fmaf(qv.z, k.z, qv.w * k.w)));
      if constexpr (BIAS) {
        const int ki = kt - off;
        const float b = __shfl(tval, qi - ki + Len - 1);
        if (ki >= 0 && ki < Len && qi >= 0 && qi < Len) s += b;
      }
      const float mn = fmaxf(mx, s);
      const float corr = __builtin_amdgcn_exp2f(mx - mn), p = __builtin_amdgcn_exp2f(s - mn);
      mx = mn;
      l = fmaf(l, corr, p);
      o0 = fmaf(o0, corr, p * v.x); o1 = fmaf(o1, corr, p * v.y); o2 = fmaf(o2, corr, p * v.z); o3 = fmaf(o3, corr, p * v.w);
    }
    mfin = mx;
    o01 = f32x2{o0, o1}; o23 = f32x2{o2, o3};
  }
  if (q < N) {
    const float inv = 1.0f / l;
    const size_t hq = ((size_t)win * H + head) * N + q;
    *reinterpret_cast<float4*>(o_hm + hq * 4) = make_float4(o01[0] * inv, o01[1] * inv, o23[0] * inv, o23[1] * inv);
    if (lse) lse[hq] = (mfin + __builtin_amdgcn_logf(l)) * RAL_LN2;   // natural-log units
  }
}

// =================================================================================
// K3: x1 = x + o Wp^T + bp;  g = LN2(x1);  u = g W1^T + b1;  a = GELU(u)
//     [LE: a[:,0] = conv3(a[:,0]) over tokens; a = GELU(a)];  x2 = x1 + a W2^T + b2
// The hidden (N x 4C) tile is processed in NCH column chunks so that long windows fit.
// =================================================================================
template <int C, int NCH>
__global__ __launch_bounds__(512) void k_mlp_fwd(const float* __restrict__ x, const float* __restrict__ o_hm,
                                                 BlockP w, float* __restrict__ x1_out,
                                                 float* __restrict__ upre_out, float* __restrict__ x2_out,
                                                 int N, int B, int NE /* existing tokens of the N slots (padded windows: < N) */,
                                                 const float* __restrict__ addend, float* __restrict__ sum_out /* optional: sum_out = block output + addend (the bottleneck's x_mid = transformer(x4) + x4, raletransformer.py:659) */) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, HC = 4 * C / NCH, LDU = LDof<HC>::v, LPR = C / 4;
  float* Xs = reinterpret_cast<float*>(smem4);  // N x LD   : x -> x1 -> x2
  float* Gs = Xs + N * LD;                      // N x LD   : o (HM, N*C) then LN2(x1)
  float* Us = Gs + N * LD;                      // N x LDU  : hidden chunk
  float* A0 = Us + N * LDU;                     // N + 2    : GELU(u[:,0]) with zero halo
  const int RPP = blockDim.x / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  if (threadIdx.x == 0) { A0[0] = 0.f; A0[N + 1] = 0.f; }   // zero halo of the LE conv (never overwritten)
  RAL_STAMP_INIT();
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    RAL_STAMP_AT(0);
    const size_t wo = (size_t)win * N * C;
    {   // x (token-major -> padded rows) and o (head-major, flat) staged in one pass: 4 loads in flight per thread
      const float4* gx = reinterpret_cast<const float4*>(x + wo);
      const float4* go = reinterpret_cast<const float4*>(o_hm + wo);
      constexpr int q = C / 4;
      const int n4 = N * q, bd = blockDim.x;
      int i = threadIdx.x;
      for (; i + bd < n4; i += 2 * bd) {
        const float4 x0 = gx[i], x1 = gx[i + bd], o0 = go[i], o1 = go[i + bd];
        *reinterpret_cast<float4*>(Xs + (i / q) * LD + (i % q) * 4) = x0;
        *reinterpret_cast<float4*>(Xs + ((i + bd) / q) * LD + ((i + bd) % q) * 4) = x1;
        reinterpret_cast<float4*>(Gs)[i] = o0; reinterpret_cast<float4*>(Gs)[i + bd] = o1;
      }
      for (; i < n4; i += bd) {
        const float4 x0 = gx[i], o0 = go[i];
        *reinterpret_cast<float4*>(Xs + (i / q) * LD + (i % q) * 4) = x0;
        reinterpret_cast<float4*>(Gs)[i] = o0;
      }
    }
    __syncthreads();
    RAL_STAMP_AT(1);
    // ---- attention output projection + residual (x1 goes to HBM straight from the epilogue registers) ----
    float* x1w = x1_out ? x1_out + wo : nullptr;
    gemm_phase<C, TTBof<C>::v, false, LAY_HM>(w.wp, C, C, Gs, N, N >> 4, [&](int row0, int tok, f32x4 a) {
      float4* px = reinterpret_cast<float4*>(Xs + tok * LD + row0);
      const float4 v = f4add(*px, f4add(tofloat4(a), *reinterpret_cast<const float4*>(w.bp + row0)));
      *px = v;
      if (x1w) *reinterpret_cast<float4*>(x1w + (size_t)tok * C + row0) = v;
    });
    __syncthreads();
    RAL_STAMP_AT(2);
    // ---- LN2 ----
    {
      const float4 gam = *reinterpret_cast<const float4*>(w.ln2w + cq);
      const float4 bet = *reinterpret_cast<const float4*>(w.ln2b + cq);
      for (int row = threadIdx.x / LPR; row < N; row += RPP) {
        const float4 v = *reinterpret_cast<const float4*>(Xs + row * LD + cq);
        float4 d; float rstd;
        ln_stats<LPR>(v, d, rstd);
        *reinterpret_cast<float4*>(Gs + row * LD + cq) = f4add(f4mul(f4scale(d, rstd), gam), bet);
      }
    }
    __syncthreads();
    RAL_STAMP_AT(3);
    float* upw = upre_out ? upre_out + (size_t)win * N * 4 * C : nullptr;
    float* x2w = x2_out + wo;
    const float* addw = sum_out ? addend + wo : nullptr;
    float* sumw = sum_out ? sum_out + wo : nullptr;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
      // fc1 + both GELUs in the epilogue (the VALU work of one wave overlaps the MFMAs of the others); only hidden
      // channel 0 of the LE variant waits for its 3-tap conv over tokens
      gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(w.w1 + (size_t)j0 * C, C, HC, Gs, LD, N >> 4,
                                                 [&](int row0, int tok, f32x4 a) {
        const float4 u = f4add(tofloat4(a), *reinterpret_cast<const float4*>(w.b1 + j0 + row0));
        if (upw) *reinterpret_cast<float4*>(upw + (size_t)tok * 4 * C + j0 + row0) = u;
        float4 h = make_float4(gelu_f(u.x), gelu_f(u.y), gelu_f(u.z), gelu_f(u.w));
        if (le) {
          if (ch == 0 && row0 == 0) A0[tok + 1] = tok < NE ? h.x : 0.f;   // conv input (a slot past NE does not exist: zero, like the halo); Us[tok][0] is filled below
          h = make_float4(gelu_f(h.x), gelu_f(h.y), gelu_f(h.z), gelu_f(h.w));
        }
        *reinterpret_cast<float4*>(Us + tok * LDU + row0) = h;
      });
      __syncthreads();
      RAL_STAMP_AT(4);
      if (le && ch == 0) {
        for (int n = threadIdx.x; n < N; n += blockDim.x)
          Us[n * LDU] = gelu_f(lw0 * A0[n] + lw1 * A0[n + 1] + lw2 * A0[n + 2]);
        __syncthreads();
      }
      RAL_STAMP_AT(5);
      gemm_phase<HC, TTBof<C>::v, false, LAY_TOK>(w.w2 + j0, 4 * C, C, Us, LDU, N >> 4,
                                                  [&](int row0, int tok, f32x4 a) {
        float4* px = reinterpret_cast<float4*>(Xs + tok * LD + row0);
        float4 v = f4add(*px, tofloat4(a));
        if (ch == 0) v = f4add(v, *reinterpret_cast<const float4*>(w.b2 + row0));
        if (ch == NCH - 1) {
          *reinterpret_cast<float4*>(x2w + (size_t)tok * C + row0) = v;   // block output
          if (sumw) *reinterpret_cast<float4*>(sumw + (size_t)tok * C + row0) = f4add(v, *reinterpret_cast<const float4*>(addw + (size_t)tok * C + row0));
        }
        else *px = v;
      });
      __syncthreads();
      RAL_STAMP_AT(6);
    }
  }
}

// =================================================================================
// K3h (wide levels, C >= 64): K3 with its three Linear layers on the f16 matrix cores - every GEMM operand as two fp16
// pieces, three products per term (gemm_phase_h2, ral_device.hpp).  The operand tiles of the products (attention output,
// LayerNorm output, hidden chunk) live in LDS as two K-contiguous fp16 planes (the same 4 bytes per element as fp32), the
// residual tile stays fp32.  wt: the tiled split planes of the weight matrices (k_tile_planes, once per forward), a matrix
// at twice its float offset from `pbase`.
// A work item is WPI consecutive windows (T = WPI * N tokens, T % 32 == 0): at these widths a window is 32 - 64 tokens
// against 144 - 576 KB of weights, and what bounds the products is how often the weights are pulled through the L2 - once
// per ITEM.  Only the 3-tap conv of the local enhancement sees window boundaries (per-window zero halos in A0).
// =================================================================================
template <int C, int NCH, int NTH>
__global__ __launch_bounds__(NTH, (NTH == 256 ? 3 : 4)) void k_mlp_fwd_h(const float* __restrict__ x, const float* __restrict__ o_hm,
                                                   BlockP w, const float* __restrict__ pbase, const _Float16* __restrict__ wt,
                                                   float* __restrict__ x1_out,
                                                   float* __restrict__ upre_out, float* __restrict__ x2_out,
                                                   int N, int B, int WPI, const float* __restrict__ addend, float* __restrict__ sum_out /* see k_mlp_fwd */) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, HC = 4 * C / NCH, LDG = ldb_of(C), LDU = ldb_of(HC), LPR = C / 4, RPP = NTH / LPR;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const int T = WPI * N;
  float* Xs = reinterpret_cast<float*>(smem4);                  // T x LD        : x -> x1 -> x2 (fp32)
  _Float16* Gh = reinterpret_cast<_Float16*>(Xs + T * LD);      // 2 x T x LDG   : o, then LN2(x1)
  const int gplane = T * LDG, uplane = T * LDU;
  _Float16* Uh = Gh + 2 * gplane;                               // 2 x T x LDU   : hidden chunk
  float* A0 = reinterpret_cast<float*>(Uh + 2 * uplane);        // WPI x (N + 2) : GELU(u[:,0]) with zero halos
  const int cq = (threadIdx.x % LPR) * 4;
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  if ((int)threadIdx.x < 2 * WPI) A0[(threadIdx.x >> 1) * (N + 2) + (threadIdx.x & 1) * (N + 1)] = 0.f;
  const _Float16* wph = wt + 2 * (w.wp - pbase);   // tiled split planes of the three weight matrices
  const _Float16* w1h = wt + 2 * (w.w1 - pbase);
  const _Float16* w2h = wt + 2 * (w.w2 - pbase);
  // activation operands times the block's powers of two (ASC_*, ral_device.hpp), their inverses in the weight-plane unscales
  const float so = asc_get(w.asc, ASC_O), sg2 = asc_get(w.asc, ASC_LN2), sh = asc_get(w.asc, ASC_HID);
  const float wunp = wplane_unscale(wph, C, C) * asc_get(w.asc, ASC_O_INV), wun1 = wplane_unscale(w1h, 4 * C, C) * asc_get(w.asc, ASC_LN2_INV),
              wun2 = wplane_unscale(w2h, C, 4 * C) * asc_get(w.asc, ASC_HID_INV);
  auto put_split = [&](_Float16* base, int plane, int off, float4 v) {
    const H2 s0 = f16_split2(v.x), s1 = f16_split2(v.y), s2 = f16_split2(v.z), s3 = f16_split2(v.w);
    *reinterpret_cast<f16x4*>(base + off) = f16x4{s0.a, s1.a, s2.a, s3.a};
    *reinterpret_cast<f16x4*>(base + plane + off) = f16x4{s0.b, s1.b, s2.b, s3.b};
  };
  RAL_STAMP_INIT();
  for (int item = blockIdx.x; item * WPI < B; item += gridDim.x) {
    RAL_STAMP_AT(0);
    const size_t wo = (size_t)item * T * C;
    {   // x -> padded fp32 rows; o (head-major quads per window) -> token-major split planes
      const float4* gx = reinterpret_cast<const float4*>(x + wo);
      const float4* go = reinterpret_cast<const float4*>(o_hm + wo);
      constexpr int q = C / 4;
      const int n4 = T * q, nq = N * q;
      auto put = [&](int i, float4 xv, float4 ov) {
        *reinterpret_cast<float4*>(Xs + (i / q) * LD + (i % q) * 4) = xv;
        const int wl = i / nq, i2 = i - wl * nq, qd = i2 / N, t = i2 - qd * N;
        put_split(Gh, gplane, (wl * N + t) * LDG + qd * 4, f4scale(ov, so));
      };
      for (int i0 = 0; i0 < n4; i0 += 2 * NTH) {
        const int i = i0 + threadIdx.x, j = i + NTH;
        const int ic = i < n4 ? i : 0, jc = j < n4 ? j : 0;
        const float4 x0 = gx[ic], x1 = gx[jc], o0 = go[ic], o1 = go[jc];
        if (i < n4) put(i, x0, o0);
        if (j < n4) put(j, x1, o1);
      }
    }
    __syncthreads();
    RAL_STAMP_AT(1);
    // ---- attention output projection + residual ----
    float* x1w = x1_out ? x1_out + wo : nullptr;
    gemm_phase_h2<C, (NTH == 256 ? -1 : 0)>(wph, C / 32, 0, 0, C, w.bp, wunp, Gh, gplane, LDG, T >> 4, [&](int row0, int tok, f32x4 a) {
      float4* px = reinterpret_cast<float4*>(Xs + tok * LD + row0);
      const float4 v = f4add(*px, tofloat4(a));
      *px = v;
      if (x1w) *reinterpret_cast<float4*>(x1w + (size_t)tok * C + row0) = v;
    });
    __syncthreads();
    RAL_STAMP_AT(2);
    // ---- LN2 -> split planes ----
    {
      const float4 gam = f4scale(*reinterpret_cast<const float4*>(w.ln2w + cq), sg2);
      const float4 bet = f4scale(*reinterpret_cast<const float4*>(w.ln2b + cq), sg2);
      for (int row = threadIdx.x / LPR; row < T; row += RPP) {
        const float4 v = *reinterpret_cast<const float4*>(Xs + row * LD + cq);
        float4 d; float rstd;
        ln_stats<LPR>(v, d, rstd);
        put_split(Gh, gplane, row * LDG + cq, f4add(f4mul(f4scale(d, rstd), gam), bet));
      }
    }
    __syncthreads();
    RAL_STAMP_AT(3);
    float* upw = upre_out ? upre_out + wo * 4 : nullptr;
    float* x2w = x2_out + wo;
    const float* addw = sum_out ? addend + wo : nullptr;
    float* sumw = sum_out ? sum_out + wo : nullptr;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
      gemm_phase_h2<C, (NTH == 256 ? -1 : 0)>(w1h, C / 32, j0 / 16, 0, HC, w.b1 + j0, wun1, Gh, gplane, LDG, T >> 4, [&](int row0, int tok, f32x4 a) {
        const float4 u = tofloat4(a);
        if (upw) *reinterpret_cast<float4*>(upw + (size_t)tok * 4 * C + j0 + row0) = u;
        float4 h = make_float4(gelu_f(u.x), gelu_f(u.y), gelu_f(u.z), gelu_f(u.w));
        if (le) {
          if (ch == 0 && row0 == 0) A0[tok + 1 + 2 * (tok / N)] = h.x;   // conv input; hidden channel 0 is filled below
          h = make_float4(gelu_f(h.x), gelu_f(h.y), gelu_f(h.z), gelu_f(h.w));
        }
        put_split(Uh, uplane, tok * LDU + row0, f4scale(h, sh));
      });
      __syncthreads();
      RAL_STAMP_AT(4);
      if (le && ch == 0) {
        for (int n = threadIdx.x; n < T; n += NTH) {
          const int a = n + 1 + 2 * (n / N);
          const H2 s = f16_split2(gelu_f(lw0 * A0[a - 1] + lw1 * A0[a] + lw2 * A0[a + 1]) * sh);
          Uh[n * LDU] = s.a; Uh[uplane + n * LDU] = s.b;
        }
        __syncthreads();
      }
      RAL_STAMP_AT(5);
      gemm_phase_h2<HC>(w2h, 4 * C / 32, 0, j0 / 32, C, ch == 0 ? w.b2 : nullptr, wun2, Uh, uplane, LDU, T >> 4, [&](int row0, int tok, f32x4 a) {
        float4* px = reinterpret_cast<float4*>(Xs + tok * LD + row0);
        const float4 v = f4add(*px, tofloat4(a));
        if (ch == NCH - 1) {
          *reinterpret_cast<float4*>(x2w + (size_t)tok * C + row0) = v;   // block output
          if (sumw) *reinterpret_cast<float4*>(sumw + (size_t)tok * C + row0) = f4add(v, *reinterpret_cast<const float4*>(addw + (size_t)tok * C + row0));
        }
        else *px = v;
      });
      __syncthreads();
      RAL_STAMP_AT(6);
    }
  }
}


// =================================================================================
// PatchMerging  : (N, C) viewed as (N/2, 2C) -> LN(2C) -> Linear(2C, 2C, no bias)
// PatchSeparate : rows [x[:, :C/2] ; x[:, C/2:]] (2N, C/2) -> LN -> Linear (+ skip)
// D = feature width of the LayerNorm / Linear; T = output tokens per window.
// =================================================================================
template <int D, bool SEP>
__global__ __launch_bounds__(256) void k_resample_fwd(const float* __restrict__ x, const float* __restrict__ wred,
                                                      const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                      const float* __restrict__ skip, float* __restrict__ y,
                                                      int T, int Tv, int B) {
  extern __shared__ float4 smem4[];
  float* Hs = reinterpret_cast<float*>(smem4);
  constexpr int LD = LDof<D>::v, LPR = D / 4, RPP = 256 / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const float4 gam = *reinterpret_cast<const float4*>(lnw + cq);
  const float4 bet = *reinterpret_cast<const float4*>(lnb + cq);
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* xw = x + (size_t)win * T * D;
    for (int row = threadIdx.x / LPR; row < T; row += RPP) {
      // SEP: out token t = c1*(Tv/2) + l  reads  x[l][c1*D + :D] of the (T/2, 2D) input (sep_src: Tv of the T slots exist)
      const float* src = SEP ? xw + sep_src(row, T, Tv, D) : xw + (size_t)row * D;
      const float4 v = *reinterpret_cast<const float4*>(src + cq);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      *reinterpret_cast<float4*>(Hs + row * LD + cq) = f4add(f4mul(f4scale(d, rstd), gam), bet);
    }
    __syncthreads();
    float* yw = y + (size_t)win * T * D;
    const float* sw = skip ? skip + (size_t)win * T * D : nullptr;
    gemm_phase<D, TTBof<D>::v, false, LAY_TOK>(wred, D, D, Hs, LD, T >> 4, [&](int row0, int tok, f32x4 a) {
      float4 v = tofloat4(a);
      if (sw) v = f4add(v, *reinterpret_cast<const float4*>(sw + (size_t)tok * D + row0));
      *reinterpret_cast<float4*>(yw + (size_t)tok * D + row0) = v;
    });
    __syncthreads();
  }
}

// y = a + b (flat), used for x_mid = transformer(x4) + x4
__global__ void k_add(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(y)[i] = f4add(reinterpret_cast<const float4*>(a)[i], reinterpret_cast<const float4*>(b)[i]);
}

// =================================================================================
// host launchers
// =================================================================================
static inline int grid_for(int items) {
  static const int gmax = (int)ral_knob("GRID_FWD", 4096);
  return items < gmax ? items : gmax;
}

// widths whose projection has a split-operand kernel (the caller passes the tiled weight planes to choose it)
bool qkv_fwd_uses_f16(int C) { return C == 32 || C == 64 || C == 128; }

void launch_qkv_fwd(int C, const float* x, const float* pe, const BlockP& w, const void* wt, float* qkv, int N, int B, hipStream_t s) {
  if (wt && qkv_fwd_uses_f16(C) && N % 16 == 0) {
    const _Float16* wtp = reinterpret_cast<const _Float16*>(wt);
    auto go = [&](auto kern, int GT) {
      const size_t ldsb = (size_t)2 * GT * ldb_of(C) * 2;
      RAL_SET_LDS(kern, ldsb);
      const long ngroups = ((long)B * N + GT - 1) / GT;
      const int wgs = 512;
      kern<<<(int)(ngroups < wgs ? ngroups : wgs), 256, ldsb, s>>>(x, pe, w, wtp, qkv, N, B);
    };
    // weight-stationary kernel: one workgroup per CU (C = 128) / two (C = 64), token groups of 64 = whole windows
    static const bool ws = (ral_knob("QKV_WS", 1) != 0);
    auto gows = [&](auto kern, int C_, int per_cu) {
      const size_t ldsb = (size_t)2 * 2 * 64 * ldb_of(C_) * 2;
      RAL_SET_LDS(kern, ldsb);
      const long ngroups = ((long)B * N + 63) / 64;
      static const int gq = (int)ral_knob("GRID_QKVW", 0);
      const int wgs = gq > 0 ? gq : 256 * per_cu;
      kern<<<(int)(ngroups < wgs ? ngroups : wgs), 6 * C_, ldsb, s>>>(x, pe, w, wtp, qkv, N, B);
    };
    if (ws && 64 % N == 0) {
      if (C == 64) { gows(k_qkv_fwd_ws<64, 64>, 64, 2); return; }
      if (C == 128) { gows(k_qkv_fwd_ws<128, 64>, 128, 1); return; }
    }
    if (C == 32 && (N % 64 == 0 || 64 % N == 0)) { go(k_qkv_fwd_h<32, 64>, 64); return; }
    if (C == 64 && (N % 64 == 0 || 64 % N == 0)) { go(k_qkv_fwd_h<64, 64>, 64); return; }
    if (C == 128 && (N % 64 == 0 || 64 % N == 0)) { go(k_qkv_fwd_h<128, 64>, 64); return; }
  }
  const size_t lds = (size_t)N * ld_of(C) * sizeof(float);
  switch (C) {
#define CASE(c) case c: k_qkv_fwd<c><<<grid_for(B), 256, lds, s>>>(x, pe, w, qkv, N, B); break;
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
}

size_t attn_fwd_lds(int N, int HG, int Len) {
  return ((size_t)3 * HG * N * 4 + (size_t)HG * N + 3 * HG + 8 + (Len > 0 ? (size_t)(2 * Len - 1) * HG : 0)) * sizeof(float);
}

void launch_attn_fwd(const float* qkv, float* o_hm, float* lse, const float* table, int N, int H, int HG, int Len,
                     int B, int f16, hipStream_t s, int NE) {
  if (NE > 0 && NE < N) {   // padded windows (NE of the N token slots exist): the generic tile kernel with its key mask
    const size_t lds = attn_fwd_lds(N, HG, Len);
    RAL_SET_LDS((k_attn_fwd<1, 0, true, false, true>), lds);
    k_attn_fwd<1, 0, true, false, true><<<grid_for(B * (H / HG)), 512, lds, s>>>(qkv, o_hm, lse, table, N, H, HG, Len, B, NE);
    return;
  }
  if (attn_fwd_w_takes(N, H, Len, table != nullptr)) { launch_attn_fwd_w(qkv, o_hm, lse, table, N, H, Len, B, f16, s); return; }
  // Window lengths [lo, hi] that take the query-per-lane kernel on the scalar path.  Measured at batch 2048
  // (tools/attn_bench.py, us per launch, MFMA-tile kernel vs scalar path): N = 512: 322 / 333, 256: 184 / 172,
  // 128: 122 / 90, 64: 91 / 51.  The switches ATTN_FWD_V_LO / _HI override (0 / 0 = never).
  static int vlo = 64, vhi = 256;
  static const bool vinit = [] { vlo = (int)ral_knob("ATTN_FWD_V_LO", vlo); vhi = (int)ral_knob("ATTN_FWD_V_HI", vhi); return true; }();
  (void)vinit;
  // with the S tile on the f16 matrix cores the tile kernel takes the long windows from the scalar path again
  // (RAL_ATTN_FWD_H = smallest such N, 0 = never)
  static const int hlo = (int)ral_knob("ATTN_FWD_H", 256);
  const bool tile16 = f16 && hlo > 0 && N >= hlo && N % 32 == 0;
  if (!tile16 && N >= vlo && N <= vhi && N >= 64 && N % 4 == 0 && (!table || 2 * Len - 1 <= 64)) {
    const int ntask = B * H * ((N + 63) / 64);
    if (table) k_attn_fwd_v<true><<<(ntask + 3) / 4, 256, 0, s>>>(qkv, o_hm, lse, table, N, H, Len, ntask);
    else k_attn_fwd_v<false><<<(ntask + 3) / 4, 256, 0, s>>>(qkv, o_hm, lse, table, N, H, 0, ntask);
    return;
  }
  // Workgroup split: 1 / SPLIT of the head group per item and 512 / SPLIT threads, so that 2 * SPLIT workgroups share
  // a CU and one's staging latency and barrier waits hide behind the others' tiles (same waves per CU, same LDS).
  // Measured at batch 2048 (fwd + bwd attention, ms per step): split 1: 7.76, split 2: 7.53 (RAL_ATTN_SPLIT).
  static const int split_env = (int)ral_knob("ATTN_SPLIT", 2);
  int split = split_env;
  while (split > 1 && (HG % split != 0 || N % 32 != 0)) split /= 2;
  if (split > 1) {
    const int hg = HG / split;
    const size_t l2 = attn_fwd_lds(N, hg, Len);
    static const bool nt_off = (ral_knob("ATTNF_NT0", 0) != 0);   // experiment knob: run-time window length everywhere
    if (N == 32 && !table && !nt_off) {
      RAL_SET_LDS((k_attn_fwd<2, 32, false>), l2);
      k_attn_fwd<2, 32, false><<<grid_for(B * (H / hg)), 512 / split, l2, s>>>(qkv, o_hm, lse, table, N, H, hg, Len, B);
      return;
    }
    if (tile16) {
      RAL_SET_LDS((k_attn_fwd<2, 0, true, true>), l2);
      k_attn_fwd<2, 0, true, true><<<grid_for(B * (H / hg)), 512 / split, l2, s>>>(qkv, o_hm, lse, table, N, H, hg, Len, B);
      return;
    }
    RAL_SET_LDS((k_attn_fwd<2>), l2);
    k_attn_fwd<2><<<grid_for(B * (H / hg)), 512 / split, l2, s>>>(qkv, o_hm, lse, table, N, H, hg, Len, B);
    return;
  }
  const size_t lds = attn_fwd_lds(N, HG, Len);
  const int items = B * (H / HG);
  static const bool force1 = (ral_knob("ATTN_QT1", 0) != 0);   // experiment knobs
  static const bool force4 = (ral_knob("ATTN_QT4", 0) != 0);
  if (N % 64 == 0 && force4) {
    RAL_SET_LDS((k_attn_fwd<4>), lds);
    k_attn_fwd<4><<<grid_for(items), 512, lds, s>>>(qkv, o_hm, lse, table, N, H, HG, Len, B);
  } else if (N % 32 == 0 && !force1) {
    RAL_SET_LDS((k_attn_fwd<2>), lds);
    k_attn_fwd<2><<<grid_for(items), 512, lds, s>>>(qkv, o_hm, lse, table, N, H, HG, Len, B);
  } else {
    RAL_SET_LDS((k_attn_fwd<1>), lds);
    k_attn_fwd<1><<<grid_for(items), 512, lds, s>>>(qkv, o_hm, lse, table, N, H, HG, Len, B);
  }
}

size_t mlp_fwd_lds(int C, int N, int nch) {
  return ((size_t)2 * N * ld_of(C) + (size_t)N * ld_of(4 * C / nch) + N + 2 + 2) * sizeof(float);
}

template <int C>
static void launch_mlp_fwd_c(int nch, const float* x, const float* o, const BlockP& w, float* x1, float* upre,
                             float* x2, int N, int B, hipStream_t s, int NE, const float* addend, float* sum_out) {
  const size_t lds = mlp_fwd_lds(C, N, nch);
  if (nch == 1) { RAL_SET_LDS((k_mlp_fwd<C, 1>), lds); k_mlp_fwd<C, 1><<<grid_for(B), 512, lds, s>>>(x, o, w, x1, upre, x2, N, B, NE, addend, sum_out); }
  else if (nch == 2) { RAL_SET_LDS((k_mlp_fwd<C, 2>), lds); k_mlp_fwd<C, 2><<<grid_for(B), 512, lds, s>>>(x, o, w, x1, upre, x2, N, B, NE, addend, sum_out); }
  else { RAL_SET_LDS((k_mlp_fwd<C, 4>), lds); k_mlp_fwd<C, 4><<<grid_for(B), 512, lds, s>>>(x, o, w, x1, upre, x2, N, B, NE, addend, sum_out); }
}

// wide levels on split fp16 operands (RAL_MLP_F16=0: the fp32-MFMA kernel everywhere)
bool mlp_fwd_uses_f16(int C, int N) {
  static const bool on = (ral_knob("MLP_F16", 1) != 0);
  return on && (C == 32 || C == 64 || C == 128) && N % 32 == 0;
}
size_t mlp_fwd_h_lds(int C, int T, int nch) {   // T = tokens of a work item
  return (size_t)T * ld_of(C) * 4 + (size_t)2 * T * ldb_of(C) * 2 + (size_t)2 * T * ldb_of(4 * C / nch) * 2 + (T / 16 * 2 + T + 4) * 4;
}
// windows per work item and hidden chunks of the split-operand kernel: the most tokens (up to RAL_MLP_TOK, a power-of-two
// number of windows dividing the batch) whose tiles fit RAL_MLP_HLDS bytes with at most four hidden chunks
static void mlp_fwd_h_plan(int C, int N, int B, int nth_ /* threads of the workgroup that will run it */, int* wpi_out, int* nch_out) {
  static const int tokmax = (int)ral_knob("MLP_TOK", 0);   // default: one window per item (64 / 128 tokens measured slower: mlp_fwd 1.87 / 1.93 against 1.76 ms per step - one workgroup per CU)
  static const size_t budget = (size_t)ral_knob("MLP_HLDS", 150 * 1024);
  int wpi = 1;
  while (wpi * 2 * N <= tokmax && B % (wpi * 2) == 0 && mlp_fwd_h_lds(C, wpi * 2 * N, 4) <= budget) wpi *= 2;
  int nch = 1;
  // (four-wave workgroups: three per CU.  The hardware admits one workgroup fewer than 160 KB / LDS suggests once the last one
  // would end within a granule of the top: 53 000 bytes run three per CU, 54 576 two - tools/diag/occ_probe.hip, census - while
  // hipOccupancyMaxActiveBlocksPerMultiprocessor still answers three.)
  const size_t b1 = wpi == 1 ? (nth_ == 256 ? 53000 : 78000) : budget;
  while (nch < (nth_ == 256 ? 8 : 4) && mlp_fwd_h_lds(C, wpi * N, nch) > b1) nch *= 2;
  *wpi_out = wpi; *nch_out = nch;
}
template <int C, int NTH>
static void launch_mlp_fwd_hc(const float* x, const float* o, const BlockP& w, const float* pbase, const void* wh,
                              float* x1, float* upre, float* x2, int N, int B, hipStream_t s, const float* addend, float* sum_out) {
  int wpi, nch;
  mlp_fwd_h_plan(C, N, B, NTH, &wpi, &nch);
  const size_t lds = mlp_fwd_h_lds(C, wpi * N, nch);
  const _Float16* whp = reinterpret_cast<const _Float16*>(wh);
  const int grid = grid_for(B / wpi);
  if (nch == 1) { RAL_SET_LDS((k_mlp_fwd_h<C, 1, NTH>), lds); k_mlp_fwd_h<C, 1, NTH><<<grid, NTH, lds, s>>>(x, o, w, pbase, whp, x1, upre, x2, N, B, wpi, addend, sum_out); }
  else if (nch == 2) { RAL_SET_LDS((k_mlp_fwd_h<C, 2, NTH>), lds); k_mlp_fwd_h<C, 2, NTH><<<grid, NTH, lds, s>>>(x, o, w, pbase, whp, x1, upre, x2, N, B, wpi, addend, sum_out); }
  else if (nch == 4 || NTH != 256) { RAL_SET_LDS((k_mlp_fwd_h<C, 4, NTH>), lds); k_mlp_fwd_h<C, 4, NTH><<<grid, NTH, lds, s>>>(x, o, w, pbase, whp, x1, upre, x2, N, B, wpi, addend, sum_out); }
  else if constexpr (NTH == 256) { RAL_SET_LDS((k_mlp_fwd_h<C, 8, NTH>), lds); k_mlp_fwd_h<C, 8, NTH><<<grid, NTH, lds, s>>>(x, o, w, pbase, whp, x1, upre, x2, N, B, wpi, addend, sum_out); }
}

void launch_mlp_fwd(int C, int nch, const float* x, const float* o, const BlockP& w, const float* pbase, const void* wh,
                    float* x1, float* upre, float* x2, int N, int B, int f16_narrow, hipStream_t s, int NE, const float* addend, float* sum_out) {
  if (NE <= 0 || NE > N) NE = N;
  const bool padded = NE < N;   // padded windows: the generic kernel (its local-enhancement conv knows where the window ends)
  if (!padded && wh && mlp_fwd_uses_f16(C, N)) {
    // threads of a k_mlp_fwd_h workgroup: 256 (default; four waves, four or eight hidden chunks, <= 53 000 bytes of LDS, three workgroups per CU, every
    // K-chunk's weight fragments of a proj / fc1 unit requested together: 135-163 registers), 512 (two per CU at 128 registers) or
    // 1024.  Measured: `mlp_fwd` 1.620 / 1.625 ms per step serialised for 512 / 256, the step 12.69 -> 12.67 ms (three interleaved
    // rounds, each in favour), inference 557.6 k -> 562.3 k windows/s
    static const int nth = (int)ral_knob("MLP_HTHREADS", 256);
    if (C == 32) launch_mlp_fwd_hc<32, 512>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out);
    else if (nth == 256 && C == 64) launch_mlp_fwd_hc<64, 256>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out);
    else if (nth == 256) launch_mlp_fwd_hc<128, 256>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out);
    else if (C == 64) { if (nth == 1024) launch_mlp_fwd_hc<64, 1024>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out); else launch_mlp_fwd_hc<64, 512>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out); }
    else { if (nth == 1024) launch_mlp_fwd_hc<128, 1024>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out); else launch_mlp_fwd_hc<128, 512>(x, o, w, pbase, wh, x1, upre, x2, N, B, s, addend, sum_out); }
    return;
  }
  if (!padded && !sum_out)
    if (const int kind = mlp_fwd_w_kind(C, N, upre != nullptr, f16_narrow != 0)) { launch_mlp_fwd_w(C, kind, x, o, w, x1, x2, N, B, s); return; }   // narrow levels: ral_mlpw.hip
  switch (C) {
#define CASE(c) case c: launch_mlp_fwd_c<c>(nch, x, o, w, x1, upre, x2, N, B, s, NE, addend, sum_out); break;
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
}

void launch_resample_fwd(int D, bool sep, const float* x, const float* wred, const float* lnw, const float* lnb,
                         const float* skip, float* y, int T, int Tv, int B, hipStream_t s) {
  const size_t lds = (size_t)T * ld_of(D) * sizeof(float);
#define CASE(d) case d: if (sep) k_resample_fwd<d, true><<<grid_for(B), 256, lds, s>>>(x, wred, lnw, lnb, skip, y, T, Tv, B); \
                        else k_resample_fwd<d, false><<<grid_for(B), 256, lds, s>>>(x, wred, lnw, lnb, skip, y, T, Tv, B); break;
  switch (D) { CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) }
#undef CASE
}

void launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t s) {
  const size_t n4 = n / 4;
  k_add<<<(int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192), 256, 0, s>>>(a, b, y, n4);
}
