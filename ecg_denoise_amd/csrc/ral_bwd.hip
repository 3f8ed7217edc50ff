// Backward (data-gradient) kernels of the RA-LENet path (gfx950).  Weight gradients
// of the Linear layers are separate token-contraction GEMMs (ral_dw.hip); the small
// vector gradients (biases, LayerNorm affines, LE taps, R-wave tables) are reduced
// here in registers / LDS across the windows a workgroup owns and flushed with one
// atomic per element per workgroup.
#ifdef RAL_STAMP_TU_BWD
#define RAL_STAMP_HERE
#endif
#include "ral_device.hpp"
#include "ral_kernels.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

RAL_STAMPS_DEFINE(ral_debug_stamps)

// LDS atomic accumulate of per-channel vectors: red[c..c+3] += v.  The block reduction of the small gradients runs in DOUBLES:
// a full-wave ds_add_f32 takes 192 LDS cycles on gfx950 (three per active lane), ds_add_f64 eight (tools/diag/lds_cost_probe.hip;
// with fp32 this flush was 8 waves x 8 adds x 192 = 12 000 LDS cycles at the end of every workgroup, all workgroups of a CU
// at once).  `redd` sits at the base of the dynamic LDS, whose tiles are dead by then.
RAL_DEV void lds_add4(double* red, int c, float4 v) {
  atomicAdd(red + c, (double)v.x); atomicAdd(red + c + 1, (double)v.y); atomicAdd(red + c + 2, (double)v.z); atomicAdd(red + c + 3, (double)v.w);
}

// =================================================================================
// B3: MLP + attention-projection backward for one block.
//   in : dx2 (grad of block output), x1, u_pre           out: du_pre, dx1, do (HM)
//   grads: b2, b1, le taps, ln2 w/b, bp
// =================================================================================
template <int C, int NCH>
__global__ __launch_bounds__(512) void k_mlp_bwd(const float* __restrict__ dx2, const float* __restrict__ x1,
                                                 const float* __restrict__ upre, BlockP w, BlockP wt, BlockP gr,
                                                 float* __restrict__ dupre, float* __restrict__ dx1,
                                                 float* __restrict__ do_hm, float* __restrict__ a2c0, int N, int B,
                                                 int NE /* existing tokens of the N slots (padded windows: < N; the others carry dx2 = 0) */) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, HC = 4 * C / NCH, LDU = LDof<HC>::v, LPR = C / 4;
  float* Ds = reinterpret_cast<float*>(smem4);  // N x LD : dx2 -> dx1
  float* Gs = Ds + N * LD;                      // N x LD : dg accumulator
  float* Us = Gs + N * LD;                      // N x LDU: u_pre chunk -> du chunk
  float* A0 = Us + N * LDU;                     // N + 2  : gelu(u[:,0]), zero halo
  float* DC0 = A0 + N + 2;                      // N + 2  : d c0, zero halo
  float* U0 = DC0 + N + 2;                      // N      : u_pre[:,0]
  float* C0 = U0 + N;                           // N      : conv output c0
  float* red = C0 + N;                          // 2C + 4 : ln2 grads + le taps (block reduction)
  const int RPP = blockDim.x / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  const float4 gam2 = *reinterpret_cast<const float4*>(w.ln2w + cq);
  // per-thread gradient accumulators (live across the window loop)
  float gle0 = 0.f, gle1 = 0.f, gle2 = 0.f;
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);

  RAL_STAMP_INIT();
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const size_t wo = (size_t)win * N * C;
    RAL_STAMP_AT(15);
    copy_in(Ds, LD, dx2 + wo, C, N, C);
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
      copy_in(Us, LDU, upre + (size_t)win * N * 4 * C + j0, 4 * C, N, HC);
      __syncthreads();
      RAL_STAMP_AT(0);
      if (le && ch == 0) {
        for (int i = threadIdx.x; i < N + 2; i += blockDim.x) {
          const bool halo = (i == 0 || i == N + 1);
          const float u = halo ? 0.f : Us[(i - 1) * LDU];
          A0[i] = (halo || i > NE) ? 0.f : gelu_f(u);            // (a slot past NE does not exist for the conv: zero, like the halo)
          DC0[i] = 0.f;
          if (!halo) U0[i - 1] = u;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
          const float c0 = lw0 * A0[i] + lw1 * A0[i + 1] + lw2 * A0[i + 2];
          C0[i] = c0;
          if (a2c0) a2c0[(size_t)win * N + i] = gelu_f(c0);   // fc2 input of the LE channel, for the fc2 weight-gradient kernel
        }
        __syncthreads();
      }
      RAL_STAMP_AT(1);
      // d a2 = dx2 W2[:, chunk]  -> du (in place over u_pre)
      gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(wt.w2 + (size_t)j0 * C, C, HC, Ds, LD, N >> 4,
                                                [&](int row0, int tok, f32x4 a) {
        float4* pu = reinterpret_cast<float4*>(Us + tok * LDU + row0);
        const float4 u = *pu;
        float uu[4] = {u.x, u.y, u.z, u.w}, out[4];
        // (the uniform `le` test OUTSIDE the element loop and the channel-0 exception patched afterwards: straight-line code
        // over the four elements is what hipcc packs two at a time; behind per-element branches every GELU was scalar)
        if (!le) {
#pragma unroll
          for (int e = 0; e < 4; ++e) out[e] = a[e] * gelu_grad_f(uu[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float a1, d1;
            gelu_pair(uu[e], a1, d1);
            out[e] = a[e] * gelu_grad_f(a1) * d1;
          }
          if (ch == 0 && row0 == 0) {
            DC0[tok + 1] = a[0] * gelu_grad_f(C0[tok]);
            out[0] = 0.f;  // filled by the channel-0 pass below
          }
        }
        *pu = make_float4(out[0], out[1], out[2], out[3]);
      });
      __syncthreads();
      RAL_STAMP_AT(2);
      if (le && ch == 0) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const float da1 = lw0 * DC0[n + 2] + lw1 * DC0[n + 1] + lw2 * DC0[n];
          Us[n * LDU] = n < NE ? da1 * gelu_grad_f(U0[n]) : 0.f;   // (... and takes no gradient)
          const float dc = DC0[n + 1];
          gle0 += dc * A0[n]; gle1 += dc * A0[n + 1]; gle2 += dc * A0[n + 2];
        }
        __syncthreads();
      }
      RAL_STAMP_AT(3);
      if (dupre) copy_out(dupre + (size_t)win * N * 4 * C + j0, 4 * C, Us, LDU, N, HC);   // (only the fc1 weight-gradient kernel reads it)
      RAL_STAMP_AT(4);
      RAL_STAMP_AT(5);
      // dg (+)= du W1[chunk, :]
      gemm_phase<HC, TTBof<C>::v, false, LAY_TOK>(wt.w1 + j0, 4 * C, C, Us, LDU, N >> 4,
                                                 [&](int row0, int tok, f32x4 a) {
        float4* pg = reinterpret_cast<float4*>(Gs + tok * LD + row0);
        *pg = (ch == 0) ? tofloat4(a) : f4add(*pg, tofloat4(a));
      });
      __syncthreads();
      RAL_STAMP_AT(6);
    }
    RAL_STAMP_AT(7);
    // LN2 backward, dx1 = dx2 + dLN
    for (int row = threadIdx.x / LPR; row < N; row += RPP) {
      const float4 v = *reinterpret_cast<const float4*>(x1 + wo + (size_t)row * C + cq);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      const float4 xh = f4scale(d, rstd);
      const float4 dg = *reinterpret_cast<const float4*>(Gs + row * LD + cq);
      const float4 dyh = f4mul(dg, gam2);
      constexpr float invC = 1.0f / C;
      const float m1 = group_sum<LPR>(f4hsum(dyh)) * invC;
      const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invC;
      float4* pd = reinterpret_cast<float4*>(Ds + row * LD + cq);
      const float4 dx = make_float4(rstd * (dyh.x - m1 - xh.x * m2), rstd * (dyh.y - m1 - xh.y * m2),
                                    rstd * (dyh.z - m1 - xh.z * m2), rstd * (dyh.w - m1 - xh.w * m2));
      *pd = f4add(*pd, dx);
      dgam = f4add(dgam, f4mul(dg, xh));
      dbet = f4add(dbet, dg);
    }
    __syncthreads();
    RAL_STAMP_AT(8);
    copy_out(dx1 + wo, C, Ds, LD, N, C);
    RAL_STAMP_AT(9);
    RAL_STAMP_AT(10);
    // do = dx1 Wp  (head-major)
    float* dow = do_hm + wo;
    gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(wt.wp, C, C, Ds, LD, N >> 4, [&](int row0, int tok, f32x4 a) {
      *reinterpret_cast<float4*>(dow + ((size_t)(row0 >> 2) * N + tok) * 4) = tofloat4(a);
    });
    __syncthreads();
    RAL_STAMP_AT(11);
  }
  // ---- flush the small gradients ----
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * C + 4; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, C + cq, dbet);
  if (le) {
    const float s0 = group_sum<64>(gle0), s1 = group_sum<64>(gle1), s2 = group_sum<64>(gle2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(redd + 2 * C, (double)s0); atomicAdd(redd + 2 * C + 1, (double)s1); atomicAdd(redd + 2 * C + 2, (double)s2); }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
#ifndef RAL_NOVECATOMICS   // (diagnostic build: what the same-address chains of the small-vector gradients cost)
    atomicAdd(gr.ln2w + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(gr.ln2b + threadIdx.x, (float)redd[C + threadIdx.x]);
#endif
  }
  if (le && threadIdx.x < 3) atomicAdd(gr.le + threadIdx.x, (float)redd[2 * C + threadIdx.x]);
}

// =================================================================================
// B3h (wide levels, C >= 64): B3 with its three data-gradient products on the f16 matrix cores, every operand as two
// fp16 pieces (gemm_wx_h2, ral_device.hpp).  wtt: tiled split planes of the TRANSPOSED weight matrices (a matrix at twice
// its float offset from `ptbase`).  The gradient rows that feed a product (dx2, du, dx1) are scaled by a power of two
// per TOKEN first (h2_row_scale: they are ~1e-6, fp16 starts at 6e-5) and the product is unscaled in its epilogue - the
// contraction runs over a token's channels, so the scale factors out exactly.  For du the row maximum is only known
// once every wave has its part of the row: the fc2^T phase keeps its (one) unit's du values in registers, raises the
// token's maximum with an LDS atomic, and splits after a barrier - into the LDS bytes of the u_pre chunk, which nobody
// reads any more.  That is why the kernel takes (N, HC) with exactly one 32 x 32 unit per wave: (HC / 32) (N / 32) == 8.
// dx2 is not kept in fp32: the LayerNorm backward reads it again from global memory (L2).
// =================================================================================
// NTH = 512: two workgroups per CU at 128 registers (one 32 x 32 unit of the fc2^T phase per wave: (HC / 32)(N / 32) = 8);
// NTH = 256: three per CU at 168 registers ((HC / 32)(N / 32) = 4: twice the hidden chunks)
#ifndef RAL_MLPBH_ALLK
#define RAL_MLPBH_ALLK 1
#endif
template <int C, int NCH, int NTH = 512>
__global__ __launch_bounds__(NTH, (NTH == 512 ? 4 : 3)) void k_mlp_bwd_h(const float* __restrict__ dx2, const float* __restrict__ x1,
                                                      const float* __restrict__ upre, BlockP w, BlockP wt, const float* __restrict__ ptbase,
                                                      const _Float16* __restrict__ wtt, BlockP gr,
                                                      float* __restrict__ dupre, float* __restrict__ dx1,
                                                      float* __restrict__ do_hm, float* __restrict__ a2c0,
                                                      unsigned* __restrict__ gmax, int N, int B) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, HC = 4 * C / NCH, LDUF = LDof<HC>::v, LDG = ldb_of(C), LDU = ldb_of(HC), LPR = C / 4, RPP = NTH / LPR;
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  _Float16* Dh = reinterpret_cast<_Float16*>(smem4);            // 2 x N x LDG : dx2, later dx1 (scaled, split)
  const int dplane = N * LDG, uplane = N * LDU;
  // NTH = 256: dg is accumulated in registers over the hidden chunks (every wave owns ONE 32 x 32 unit of it: (C / 32)(N / 32) = 4)
  // and meets the LayerNorm backward through the bytes of the u_pre / du chunk, which is free by then
  constexpr bool DGR = NTH == 256;
  float* Gs = reinterpret_cast<float*>(Dh + 2 * dplane);        // N x LD      : dg accumulator (fp32)
  float* Us = DGR ? Gs : Gs + N * LD;                           // N x LDUF    : u_pre chunk (fp32) ...
  _Float16* Uh = reinterpret_cast<_Float16*>(Us);               // 2 x N x LDU : ... then du (scaled, split); N (HC + 8) floats
  float* A0 = Us + N * (HC + 8);                                // N + 2  : gelu(u[:,0]), zero halo
  float* DC0 = A0 + N + 2;                                      // N + 2  : d c0, zero halo
  float* U0 = DC0 + N + 2;                                      // N      : u_pre[:,0]
  float* C0 = U0 + N;                                           // N      : conv output c0
  float* DU0 = C0 + N;                                          // N      : du of hidden channel 0 (local enhancement)
  unsigned* smD = reinterpret_cast<unsigned*>(DU0 + N);         // N      : bits of max |dx2 row| / |dx1 row|
  unsigned* smU = smD + N;                                      // N      : bits of max |du row| of the chunk
  float* red = reinterpret_cast<float*>(smU + N);               // 2C + 4 : ln2 grads + le taps (block reduction)
  const int cq = (threadIdx.x % LPR) * 4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  const float4 gam2 = *reinterpret_cast<const float4*>(w.ln2w + cq);
  const _Float16* w2t = wtt + 2 * (wt.w2 - ptbase);   // (4C x C)
  const _Float16* w1t = wtt + 2 * (wt.w1 - ptbase);   // (C x 4C)
  const _Float16* wpt = wtt + 2 * (wt.wp - ptbase);   // (C x C)
  const float wun2 = wplane_unscale(w2t, 4 * C, C), wun1 = wplane_unscale(w1t, C, 4 * C), wunp = wplane_unscale(wpt, C, C);   // (the planes' powers of two)
  float gle0 = 0.f, gle1 = 0.f, gle2 = 0.f;
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  auto put_split = [&](_Float16* base, int plane, int off, float4 v) {
    const H2 s0 = f16_split2u(v.x), s1 = f16_split2u(v.y), s2 = f16_split2u(v.z), s3 = f16_split2u(v.w);   // (rows arrive scaled)
    *reinterpret_cast<f16x4*>(base + off) = f16x4{s0.a, s1.a, s2.a, s3.a};
    *reinterpret_cast<f16x4*>(base + plane + off) = f16x4{s0.b, s1.b, s2.b, s3.b};
  };
  // largest magnitudes this thread has seen of dx2, du and dx1: the weight-gradient kernels scale those operands by ONE
  // power of two per launch (their contraction runs over tokens) and read it from gmax[0..2]
  float tmx2 = 0.f, tmxu = 0.f, tmx1 = 0.f;
  auto put_row = [&](int row, float4 v, float& tmx) {   // a gradient row held by LPR lanes -> scaled split planes of Dh + its maximum
    const float mx = group_max<LPR>(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    const unsigned mb = __float_as_uint(mx);
    put_split(Dh, dplane, row * LDG + cq, f4scale(v, h2_row_scale(mb)));
    if (cq == 0) smD[row] = mb;
    tmx = fmaxf(tmx, mx);
  };
  const int mb_ = HC / 32, um = wave % mb_, ut = wave / mb_;   // the wave's unit of the fc2^T phase
  RAL_STAMP_INIT();
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const size_t wo = (size_t)win * N * C;
    RAL_STAMP_AT(12);                                            // (loop control; the previous window's last barrier)
    for (int row0 = threadIdx.x / LPR; row0 < N; row0 += 2 * RPP) {
      const int rowb = row0 + RPP < N ? row0 + RPP : row0;
      const float4 va = *reinterpret_cast<const float4*>(dx2 + wo + (size_t)row0 * C + cq);
      const float4 vb = *reinterpret_cast<const float4*>(dx2 + wo + (size_t)rowb * C + cq);
      put_row(row0, va, tmx2);
      if (row0 + RPP < N) put_row(rowb, vb, tmx2);
    }
    if ((int)threadIdx.x < N) smU[threadIdx.x] = 0u;
    f32x4 dgr[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) dgr[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gm_ = wave % (C / 32), gt_ = wave / (C / 32);      // DGR: the wave's unit of dg (rows 32 gm_ .., tokens 32 gt_ ..)
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
      RAL_STAMP_AT(13);                                          // dx2 rows staged (ch = 0) / dg product (ch > 0)
      copy_in(Us, LDUF, upre + (size_t)win * N * 4 * C + j0, 4 * C, N, HC);
      __syncthreads();
      RAL_STAMP_AT(14);                                          // u_pre chunk: global loads waited for in place
      if (le && ch == 0) {
        for (int i = threadIdx.x; i < N + 2; i += blockDim.x) {
          const bool halo = (i == 0 || i == N + 1);
          const float u = halo ? 0.f : Us[(i - 1) * LDUF];
          A0[i] = halo ? 0.f : gelu_f(u);
          DC0[i] = 0.f;
          if (!halo) U0[i - 1] = u;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) {
          const float c0 = lw0 * A0[i] + lw1 * A0[i + 1] + lw2 * A0[i + 2];
          C0[i] = c0;
          if (a2c0) a2c0[(size_t)win * N + i] = gelu_f(c0);   // fc2 input of the LE channel, for the fc2 weight-gradient kernel
        }
        __syncthreads();
      }
      RAL_STAMP_AT(16);                                          // local-enhancement preparation (ch = 0)
      // ---- d a2 = dx2 W2[:, chunk] -> du: the wave's 32 hidden x 32 token unit, values kept in registers ----
      float outv[2][2][4];
      {
        f32x4 acc[2][2], accx[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) { acc[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        gemm_wx_h2<C, 2, 2, NoHook, (DGR && RAL_MLPBH_ALLK ? -1 : 0), 1>(w2t, C / 32, j0 / 16 + um * 2, 0, Dh, dplane, LDG, ut * 32, acc, accx);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const int tok = ut * 32 + tt * 16 + r;
          const float sd = h2_row_unscale(smD[tok]) * wun2;
          float mx = 0.f;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
            const int row0 = (um * 2 + mi) * 16 + 4 * g;
            const f32x4 a = acc[mi][tt] * sd;
            const float4 u = *reinterpret_cast<const float4*>(Us + tok * LDUF + row0);
            const float uu[4] = {u.x, u.y, u.z, u.w};
            float o4[4];
            if (!le) {     // (the `le` test outside the element loop: see k_mlp_bwd)
#pragma unroll
              for (int e = 0; e < 4; ++e) o4[e] = a[e] * gelu_grad_f(uu[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float a1, d1;
                gelu_pair(uu[e], a1, d1);
                o4[e] = a[e] * gelu_grad_f(a1) * d1;
              }
              if (ch == 0 && row0 == 0) {
                DC0[tok + 1] = a[0] * gelu_grad_f(C0[tok]);
                o4[0] = 0.f;  // filled by the channel-0 pass below
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { outv[mi][tt][e] = o4[e]; mx = fmaxf(mx, fabsf(o4[e])); }
          }
          mx = rows_max(mx);
          tmxu = fmaxf(tmxu, mx);
          if (g == 0) atomicMax(smU + tok, __float_as_uint(mx));
        }
      }
      __syncthreads();
      RAL_STAMP_AT(17);                                          // fc2^T product + GELU-derivative epilogue
      if (le && ch == 0) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const float da1 = lw0 * DC0[n + 2] + lw1 * DC0[n + 1] + lw2 * DC0[n];
          const float v = da1 * gelu_grad_f(U0[n]);
          DU0[n] = v;
          tmxu = fmaxf(tmxu, fabsf(v));
          atomicMax(smU + n, __float_as_uint(fabsf(v)));
          const float dc = DC0[n + 1];
          gle0 += dc * A0[n]; gle1 += dc * A0[n + 1]; gle2 += dc * A0[n + 2];
        }
        __syncthreads();
      }
      RAL_STAMP_AT(18);                                          // channel-0 pass of the local enhancement
      // every reader of the u_pre chunk is past a barrier: du goes to global memory (fp32) and, scaled and split, over it
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int tok = ut * 32 + tt * 16 + r;
        const float su = h2_row_scale(smU[tok]);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int row0 = (um * 2 + mi) * 16 + 4 * g;
          float4 v = make_float4(outv[mi][tt][0], outv[mi][tt][1], outv[mi][tt][2], outv[mi][tt][3]);
          if (le && ch == 0 && row0 == 0) v.x = DU0[tok];
          if (dupre) *reinterpret_cast<float4*>(dupre + ((size_t)win * N + tok) * 4 * C + j0 + row0) = v;   // (only the fc1 weight-gradient kernel reads it)
          put_split(Uh, uplane, tok * LDU + row0, f4scale(v, su));
        }
      }
      __syncthreads();
      RAL_STAMP_AT(19);                                          // du written and split
      // ---- dg (+)= du W1[chunk, :] ----
      if constexpr (DGR) {
        f32x4 acc[2][2], accx[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) { acc[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[mi][tt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        gemm_wx_h2<HC, 2, 2, NoHook, (RAL_MLPBH_ALLK ? -1 : 0), 1>(w1t, 4 * C / 32, gm_ * 2, j0 / 32, Uh, uplane, LDU, gt_ * 32, acc, accx);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          const float si = h2_row_unscale(smU[gt_ * 32 + tt * 16 + r]) * wun1;   // (the chunk's power of two of this token's du row)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) dgr[mi][tt] += acc[mi][tt] * si;
        }
      } else
      gemm_phase_h2<HC, 0, 1>(w1t, 4 * C / 32, 0, j0 / 32, C, nullptr, 1.0f, Uh, uplane, LDU, N >> 4, [&](int row0, int tok, f32x4 a) {
        const float si = h2_row_unscale(smU[tok]) * wun1;
        float4* pg = reinterpret_cast<float4*>(Gs + tok * LD + row0);
        const float4 v = f4scale(tofloat4(a), si);
        *pg = (ch == 0) ? v : f4add(*pg, v);
      });
      __syncthreads();
      if ((int)threadIdx.x < N) smU[threadIdx.x] = 0u;
    }
    RAL_STAMP_AT(25);                                            // last chunk's dg product
    if constexpr (DGR) {   // (every reader of the last du chunk is past the barrier above)
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          *reinterpret_cast<float4*>(Gs + (gt_ * 32 + tt * 16 + r) * LD + (gm_ * 2 + mi) * 16 + 4 * g) = tofloat4(dgr[mi][tt]);
      __syncthreads();
    }
    // ---- LN2 backward, dx1 = dx2 + dLN: to global memory and, scaled and split, into Dh ----
    for (int row = threadIdx.x / LPR; row < N; row += RPP) {
      const float4 v = *reinterpret_cast<const float4*>(x1 + wo + (size_t)row * C + cq);
      const float4 d2 = *reinterpret_cast<const float4*>(dx2 + wo + (size_t)row * C + cq);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      const float4 xh = f4scale(d, rstd);
      const float4 dg = *reinterpret_cast<const float4*>(Gs + row * LD + cq);
      const float4 dyh = f4mul(dg, gam2);
      constexpr float invC = 1.0f / C;
      const float m1 = group_sum<LPR>(f4hsum(dyh)) * invC;
      const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invC;
      const float4 dx = f4add(d2, make_float4(rstd * (dyh.x - m1 - xh.x * m2), rstd * (dyh.y - m1 - xh.y * m2),
                                              rstd * (dyh.z - m1 - xh.z * m2), rstd * (dyh.w - m1 - xh.w * m2)));
      *reinterpret_cast<float4*>(dx1 + wo + (size_t)row * C + cq) = dx;
      put_row(row, dx, tmx1);
      dgam = f4add(dgam, f4mul(dg, xh));
      dbet = f4add(dbet, dg);
    }
    __syncthreads();
    RAL_STAMP_AT(26);                                            // LayerNorm backward: x1 / dx2 rows waited for in place
    // ---- do = dx1 Wp (head-major) ----
    float* dow = do_hm + wo;
    gemm_phase_h2<C, 0, 1>(wpt, C / 32, 0, 0, C, nullptr, 1.0f, Dh, dplane, LDG, N >> 4, [&](int row0, int tok, f32x4 a) {
      *reinterpret_cast<float4*>(dow + ((size_t)(row0 >> 2) * N + tok) * 4) = f4scale(tofloat4(a), h2_row_unscale(smD[tok]) * wunp);
    });
    __syncthreads();
    RAL_STAMP_AT(27);                                            // do product
  }
  // ---- flush the small gradients ----
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * C + 4; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, C + cq, dbet);
  if (le) {
    const float s0 = group_sum<64>(gle0), s1 = group_sum<64>(gle1), s2 = group_sum<64>(gle2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(redd + 2 * C, (double)s0); atomicAdd(redd + 2 * C + 1, (double)s1); atomicAdd(redd + 2 * C + 2, (double)s2); }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    atomicAdd(gr.ln2w + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(gr.ln2b + threadIdx.x, (float)redd[C + threadIdx.x]);
  }
  if (le && threadIdx.x < 3) atomicAdd(gr.le + threadIdx.x, (float)redd[2 * C + threadIdx.x]);
  if (gmax) {   // one atomic per workgroup and tensor
    __syncthreads();
    if (threadIdx.x < 3) smD[threadIdx.x] = 0u;
    __syncthreads();
    const float m2 = group_max<64>(tmx2), mu = group_max<64>(tmxu), m1 = group_max<64>(tmx1);
    if ((threadIdx.x & 63) == 0) { atomicMax(smD, __float_as_uint(m2)); atomicMax(smD + 1, __float_as_uint(mu)); atomicMax(smD + 2, __float_as_uint(m1)); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMax(gmax + threadIdx.x, smD[threadIdx.x]);
  }
}

// =================================================================================
// B3s: the same backward for the narrow levels (C <= 32) with the fc1 / fc2 WEIGHT gradients fused in.
// At these widths dW1 (4C x C) and dW2 (C x 4C) fit in a few accumulator registers per wave, and both of
// their operands (du | LN2(x1) and dx2 | a2) are already in LDS here, so the two token-contraction kernels
// of ral_dw.hip -- which re-read u_pre and du_pre from HBM and re-compute the GELU chain and the LayerNorm --
// disappear, and du_pre is never written.  Workgroups are persistent (a few windows each) so that the
// accumulators are flushed with one atomic per element per workgroup.
//   u_pre is not read either: it is re-computed chunk by chunk from LN2(x1), which is staged for dW1 anyway.
//   hidden chunks of HC = C channels (4 chunks); LDS: dx2->dx1 | LN2(x1) | u_pre->du (->dg) | a2, each N x LD
//   per chunk: da2 GEMM (epilogue: du, a2) -> barrier -> [dg tiles in registers, dW tile jobs] -> barrier
//   TW = token tiles of dg per wave (N * max(C,16) / 2048)
// =================================================================================
template <int C, int TW>
__global__ __launch_bounds__(512, 4) void k_mlp_bwd_s(const float* __restrict__ dx2, const float* __restrict__ x1,
                                                      BlockP w, BlockP wt, BlockP gr,
                                                      float* __restrict__ dx1, float* __restrict__ do_hm, int N, int B,
                                                      int want_dw, int NE /* existing tokens of the N slots (padded windows: < N) */) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, HC = C, NCH = 4, LPR = C / 4;
  // Xl and As are only read as B operands of the dW jobs (4-byte reads): at C = 16 they go unpadded so that two
  // workgroups still fit a CU (the 4-way bank conflict of those few reads costs less than the occupancy)
  constexpr int LDB = (C == 16) ? C : LD;
  constexpr int MT = C >= 16 ? C / 16 : 1;          // 16-row tiles along C (half a tile at C = 8)
  constexpr int T = MT * MT, JOBS = 2 * T;          // dW tiles per product and chunk; jobs = both products
  constexpr int KS = JOBS >= 8 ? 1 : 8 / JOBS;      // spare waves split the tokens of a job
  static_assert(JOBS <= 8 && 8 % JOBS == 0, "one dW job per wave");
  float* Ds = reinterpret_cast<float*>(smem4);  // N x LD : dx2 -> dx1
  float* Xl = Ds + N * LD;                      // N x LD : LN2(x1)  (fc1 input)
  float* Us = Xl + N * LDB;                      // N x LD : u_pre chunk -> du chunk; dg after the chunks
  float* As = Us + N * LD;                      // N x LD : a2 chunk (fc2 input)
  float* A0 = As + N * LDB;                      // N + 2  : gelu(u[:,0]), zero halo
  float* DC0 = A0 + N + 2;                      // N + 2  : d c0, zero halo
  float* U0 = DC0 + N + 2;                      // N      : u_pre[:,0]
  float* C0 = U0 + N;                           // N      : conv output c0
  float* red = C0 + N;                          // 2C + 4 : ln2 grads + le taps (block reduction)
  const int RPP = blockDim.x / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const bool le = w.le != nullptr;
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (le) { lw0 = w.le[0]; lw1 = w.le[1]; lw2 = w.le[2]; }
  const float4 gam2 = *reinterpret_cast<const float4*>(w.ln2w + cq);
  const float4 bet2 = *reinterpret_cast<const float4*>(w.ln2b + cq);
  float gle0 = 0.f, gle1 = 0.f, gle2 = 0.f;
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  // this wave's dW job: product 0 = dW2 (rows c, columns hidden j), product 1 = dW1 (rows hidden j, columns c)
  const int job = wave % JOBS, kpart = wave / JOBS;
  const int prod = job / T, mi = (job % T) / MT, nj = (job % T) % MT;
  int arow = mi * 16 + r, bcol = nj * 16 + r;       // operand columns of this lane, clamped into the half tile
  if (arow >= C) arow = C - 1;
  if (bcol >= C) bcol = C - 1;
  f32x4 accw[NCH];
  float bs1[NCH], bs2 = 0.f;                         // column sums of du (db1) and of dx2 (db2)
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) { accw[ch] = f32x4{0.f, 0.f, 0.f, 0.f}; bs1[ch] = 0.f; }
  // this wave's dg tiles: channel tile gm, token tiles [gt0, gt0 + TW)
  const int gm = wave % MT, gt0 = (wave / MT) * TW;

  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const size_t wo = (size_t)win * N * C;
    copy_in(Ds, LD, dx2 + wo, C, N, C);
    for_each_row_f4<4>(x1 + wo, C, N, C, [&](int row, int c, float4 v) {   // (row = LPR consecutive lanes)
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      *reinterpret_cast<float4*>(Xl + row * LDB + c) = f4add(f4mul(f4scale(d, rstd), gam2), bet2);
    });
    f32x4 accg[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) accg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
      // u_pre chunk = LN2(x1) W1[chunk]^T + b1, RE-COMPUTED (the forward does not store u_pre at these levels: the
      // matrix pipe is idle here and 4E of HBM writes plus 4E of reads per block are not)
      if (ch == 0) __syncthreads();   // Xl (and dx2) staged
      gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(w.w1 + (size_t)j0 * C, C, HC, Xl, LDB, N >> 4,
                                                [&](int row0, int tok, f32x4 a) {
        *reinterpret_cast<float4*>(Us + tok * LD + row0) =
            f4add(tofloat4(a), *reinterpret_cast<const float4*>(w.b1 + j0 + row0));
      });
      __syncthreads();
      if (le && ch == 0) {
        for (int i = threadIdx.x; i < N + 2; i += blockDim.x) {
          const bool halo = (i == 0 || i == N + 1);
          const float u = halo ? 0.f : Us[(i - 1) * LD];
          A0[i] = (halo || i > NE) ? 0.f : gelu_f(u);
          DC0[i] = 0.f;
          if (!halo) U0[i - 1] = u;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += blockDim.x) C0[i] = lw0 * A0[i] + lw1 * A0[i + 1] + lw2 * A0[i + 2];
        __syncthreads();
      }
      // d a2 = dx2 W2[:, chunk]  -> du (in place over u_pre), a2 -> As
      gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(wt.w2 + (size_t)j0 * C, C, HC, Ds, LD, N >> 4,
                                                [&](int row0, int tok, f32x4 a) {
        float4* pu = reinterpret_cast<float4*>(Us + tok * LD + row0);
        const float4 u = *pu;
        float uu[4] = {u.x, u.y, u.z, u.w}, out[4], a2[4], a1v[4], d1v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gelu_pair(uu[e], a1v[e], d1v[e]);
        if (!le) {     // (the `le` test outside the element loop: see k_mlp_bwd)
#pragma unroll
          for (int e = 0; e < 4; ++e) { out[e] = a[e] * d1v[e]; a2[e] = a1v[e]; }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float g2, d2;
            gelu_pair(a1v[e], g2, d2);
            out[e] = a[e] * d2 * d1v[e]; a2[e] = g2;
          }
          if (ch == 0 && row0 == 0) {
            DC0[tok + 1] = a[0] * gelu_grad_f(C0[tok]);
            out[0] = 0.f; a2[0] = 0.f;  // both filled by the channel-0 pass below
          }
        }
        *pu = make_float4(out[0], out[1], out[2], out[3]);
        *reinterpret_cast<float4*>(As + tok * LDB + row0) = make_float4(a2[0], a2[1], a2[2], a2[3]);
      });
      __syncthreads();
      if (le && ch == 0) {
        for (int n = threadIdx.x; n < N; n += blockDim.x) {
          const float da1 = lw0 * DC0[n + 2] + lw1 * DC0[n + 1] + lw2 * DC0[n];
          Us[n * LD] = n < NE ? da1 * gelu_grad_f(U0[n]) : 0.f;
          As[n * LDB] = gelu_f(C0[n]);
          const float dc = DC0[n + 1];
          gle0 += dc * A0[n]; gle1 += dc * A0[n + 1]; gle2 += dc * A0[n + 2];
        }
        __syncthreads();
      }
      // dg += du W1[chunk, :]   (this wave's tiles, accumulated in registers over the chunks)
      gemm_wx<HC, TW, false, LAY_TOK>(wt.w1 + j0, 4 * C, gm * 16, C, Us, LD, gt0 * 16, accg);
      // dW job of this wave: contraction over its share of the window's tokens, both operands in LDS
      if (want_dw) {   // (frozen weights, ral_backward_input: no weight gradients)
        const float* Ap = prod ? Us : Ds;
        const float* Bp = prod ? Xl : As;
        const int tlen = N / KS;
        for (int t0 = kpart * tlen; t0 < (kpart + 1) * tlen; t0 += 16) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int t = t0 + 4 * g + s;
            const float av = Ap[t * LD + arow], bv = Bp[t * LDB + bcol];
            accw[ch] = mfma4(av, bv, accw[ch]);
            bs1[ch] += av;
            if (ch == 0) bs2 += av;
          }
        }
      }
      __syncthreads();
    }
    // dg -> LDS (over the du buffer) for the row-wise LayerNorm backward
    {
      const int row0 = gm * 16 + 4 * g;
      if (row0 < C) {
#pragma unroll
        for (int i = 0; i < TW; ++i)
          *reinterpret_cast<float4*>(Us + ((gt0 + i) * 16 + r) * LD + row0) = tofloat4(accg[i]);
      }
    }
    __syncthreads();
    // LN2 backward, dx1 = dx2 + dLN
    for (int row = threadIdx.x / LPR; row < N; row += RPP) {
      const float4 v = *reinterpret_cast<const float4*>(x1 + wo + (size_t)row * C + cq);
      float4 d; float rstd;
      ln_stats<LPR>(v, d, rstd);
      const float4 xh = f4scale(d, rstd);
      const float4 dg = *reinterpret_cast<const float4*>(Us + row * LD + cq);
      const float4 dyh = f4mul(dg, gam2);
      constexpr float invC = 1.0f / C;
      const float m1 = group_sum<LPR>(f4hsum(dyh)) * invC;
      const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invC;
      float4* pd = reinterpret_cast<float4*>(Ds + row * LD + cq);
      const float4 dx = make_float4(rstd * (dyh.x - m1 - xh.x * m2), rstd * (dyh.y - m1 - xh.y * m2),
                                    rstd * (dyh.z - m1 - xh.z * m2), rstd * (dyh.w - m1 - xh.w * m2));
      *pd = f4add(*pd, dx);
      dgam = f4add(dgam, f4mul(dg, xh));
      dbet = f4add(dbet, dg);
    }
    __syncthreads();
    copy_out(dx1 + wo, C, Ds, LD, N, C);
    // do = dx1 Wp  (head-major)
    float* dow = do_hm + wo;
    gemm_phase<C, TTBof<C>::v, false, LAY_TOK>(wt.wp, C, C, Ds, LD, N >> 4, [&](int row0, int tok, f32x4 a) {
      *reinterpret_cast<float4*>(dow + ((size_t)(row0 >> 2) * N + tok) * 4) = tofloat4(a);
    });
    __syncthreads();
  }
  // ---- flush the small gradients ----
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * C + 4; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, C + cq, dbet);
  if (le) {
    const float s0 = group_sum<64>(gle0), s1 = group_sum<64>(gle1), s2 = group_sum<64>(gle2);
    if ((threadIdx.x & 63) == 0) { atomicAdd(redd + 2 * C, (double)s0); atomicAdd(redd + 2 * C + 1, (double)s1); atomicAdd(redd + 2 * C + 2, (double)s2); }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
#ifndef RAL_NOVECATOMICS   // (diagnostic build: what the same-address chains of the small-vector gradients cost)
    atomicAdd(gr.ln2w + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(gr.ln2b + threadIdx.x, (float)redd[C + threadIdx.x]);
#endif
  }
  if (le && threadIdx.x < 3) atomicAdd(gr.le + threadIdx.x, (float)redd[2 * C + threadIdx.x]);
  // ---- flush the weight gradients: fold the token-split partials through LDS, then one atomic per element ----
  if (!want_dw) return;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) bs1[ch] = rows_sum(bs1[ch]);
  bs2 = rows_sum(bs2);
  if constexpr (KS > 1) {
    float* fold = Ds;   // (KS - 1) x JOBS x (NCH x 256 + NCH x 16 + 16) floats <= 3 x 2 x 1104: inside the 4 N x LD tiles
    constexpr int PER = NCH * 256 + NCH * 16 + 16;
    __syncthreads();    // `redd` sits at the base of the LDS, where the fold region starts: its readers (the atomics above) must be done
    float* mine = fold + ((kpart - 1) * JOBS + job) * PER;
    if (kpart > 0) {
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        *reinterpret_cast<f32x4*>(mine + ch * 256 + lane * 4) = accw[ch];
        if (g == 0) mine[NCH * 256 + ch * 16 + r] = bs1[ch];
      }
      if (g == 0) mine[NCH * 256 + NCH * 16 + r] = bs2;
    }
    __syncthreads();
    if (kpart == 0) {
      for (int k = 1; k < KS; ++k) {
        const float* o = fold + ((k - 1) * JOBS + job) * PER;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
          accw[ch] += *reinterpret_cast<const f32x4*>(o + ch * 256 + lane * 4);
          bs1[ch] += o[NCH * 256 + ch * 16 + r];
        }
        bs2 += o[NCH * 256 + NCH * 16 + r];
      }
    }
  }
  // The tiles go through the LDS as the two whole matrices (dW1 4C x C, dW2 C x 4C: 8 C^2 floats behind the fold region)
  // and leave it with consecutive threads adding consecutive floats: global float atomics run at full rate only for
  // 256 contiguous bytes per wave-instruction, and a 16 x 16 accumulator register is four 64-byte pieces of four rows.
  float* stg = Ds + 6656;   // (the fold region is at most 3 x 2 x 1104 floats; the launcher sizes the LDS for 6656 + 8 C^2)
  __syncthreads();   // (the fold region and, at short windows, the small-gradient array `red` overlap the staging area and are read above)
  if (kpart == 0) {
    const int col = nj * 16 + r;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int j0 = ch * HC;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = mi * 16 + 4 * g + q;
        if (row < C && col < C) {
          if (prod) stg[(j0 + row) * C + col] = accw[ch][q];                 // dW1[hidden][c]
          else stg[4 * C * C + row * 4 * C + j0 + col] = accw[ch][q];         // dW2[c][hidden]
        }
      }
      if (prod && nj == 0 && g == 0 && mi * 16 + r < C) atomicAdd(gr.b1 + j0 + mi * 16 + r, bs1[ch]);
    }
    if (!prod && nj == 0 && g == 0 && mi * 16 + r < C) atomicAdd(gr.b2 + mi * 16 + r, bs2);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * C * C; i += blockDim.x) {
    atomicAdd(gr.w1 + i, stg[i]);
    atomicAdd(gr.w2 + i, stg[4 * C * C + i]);
  }
}

// =================================================================================
// B2: attention backward per (window, head group).  P is recomputed from q, k and the
// saved log-sum-exp.  Sweep A keeps a query block on the lanes (dQ accumulates lane-
// privately over keys), sweep B keeps a key block on the lanes (dK, dV accumulate over
// queries): no cross-workgroup sums, no atomics except the tiny R-wave table gradient.
// Output dqkv has the qkv layout; dq already carries the 0.5 of q = 0.5 (h Wq^T + b).
// =================================================================================
typedef float f32x2 __attribute__((ext_vector_type(2)));
RAL_DEV f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
RAL_DEV f32x2 splat2(float v) { return f32x2{v, v}; }
#define RAL_LOG2E 1.4426950408889634f
#define RAL_LN2 0.6931471805599453f

// NT > 0: the window length as a compile-time constant (short windows: a task's sweep is 2-4 iterations, so its set-up,
// loop control and epilogue weigh as much as its tiles; with N known they unroll and their index arithmetic folds);
// TAB = false: no R-wave table (its bounds and branches fold away).
#ifndef RAL_ATTNB_WPE
#define RAL_ATTNB_WPE 4   // waves per SIMD the register budget is sized for (3 = 168 registers, no spills: N = 128 257 vs 250 us, equal elsewhere)
#endif
// RAG: only the first NE of the N token slots exist (see k_attn_fwd): keys past NE are masked (p = 0: no dS, dK, dV), the
// padding queries carry dO = 0 and add nothing; both sweeps stop at the last tile that holds an existing token
template <int QT, int NT = 0, bool TAB = true, bool RAG = false>
__global__ __launch_bounds__(512, RAL_ATTNB_WPE) void k_attn_bwd(const float* __restrict__ qkv, const float* __restrict__ o_hm,
                                                  const float* __restrict__ do_hm, const float* __restrict__ lse,
                                                  const float* __restrict__ table, float* __restrict__ gtable,
                                                  float* __restrict__ dqkv, int N_rt, int H, int HG, int Len, int B, int NE_rt = 0) {
  extern __shared__ float4 smem4[];
  const int N = NT ? NT : N_rt;
  const int NE = RAG ? NE_rt : N, NEt = RAG ? ((NE + 15) & ~15) : N;
  if constexpr (!TAB) { table = nullptr; Len = 0; }
  float* Qs = reinterpret_cast<float*>(smem4);  // q * log2(e)
  float* Ks = Qs + HG * N * 4;
  float* Vs = Ks + HG * N * 4;
  float* dOs = Vs + HG * N * 4;
  float* Ls = dOs + HG * N * 4;  // -lse * log2e      (negated: they enter the MFMAs as C operands,
  float* Dl = Ls + HG * N;       // -rowsum(dO * O)    so tiles come out as s - lse and dP - delta)
  const int ntab = table ? (2 * Len - 1) * HG : 0;
  float* tab = Dl + HG * N;      // bias * log2e
  float* dtab = tab + ntab;
  const int ngrp = H / HG;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int off = (NE - Len) >> 1;
  const int kb0 = table ? (off & ~15) : NEt, kb1 = table ? ((off + Len + 15) & ~15) : NEt;
  for (int i = threadIdx.x; i < ntab; i += blockDim.x) dtab[i] = 0.f;
#ifdef RAL_STAMP
  constexpr int C = -1;   // (stamp conditions name the channel width)
#endif
  RAL_STAMP_INIT();
  for (int item = blockIdx.x; item < B * ngrp; item += gridDim.x) {
    RAL_STAMP_AT(20);
    const int win = item / ngrp, h0 = (item - win * ngrp) * HG;
    const float* base = qkv + (size_t)win * 3 * H * N * 4;
    float* dbase = dqkv + (size_t)win * 3 * H * N * 4;
    const size_t hq0 = ((size_t)win * H + h0) * N;
    // one staging pass: the six loads of an index are issued together (a separate loop per tensor would pay
    // one HBM round trip each)
    {
      const float4* gq = reinterpret_cast<const float4*>(base + (size_t)h0 * N * 4);
      const float4* gk = reinterpret_cast<const float4*>(base + (size_t)(H + h0) * N * 4);
      const float4* gv = reinterpret_cast<const float4*>(base + (size_t)(2 * H + h0) * N * 4);
      const float4* gd = reinterpret_cast<const float4*>(do_hm) + hq0;
      const float4* go = reinterpret_cast<const float4*>(o_hm) + hq0;
      const int n4 = HG * N, bd = blockDim.x;
      int i = threadIdx.x;
#ifdef RAL_ATTNB_NOSTAGE   // diagnostic: no staging loads (the tiles work on whatever the LDS holds)
      i = n4;
#endif
      for (; i + bd < n4; i += 2 * bd) {
        const float4 q0 = gq[i], q1 = gq[i + bd], k0 = gk[i], k1 = gk[i + bd], v0 = gv[i], v1 = gv[i + bd];
        const float4 d0 = gd[i], d1 = gd[i + bd], o0 = go[i], o1 = go[i + bd];
        const float l0 = lse[hq0 + i], l1 = lse[hq0 + i + bd];
        reinterpret_cast<float4*>(Qs)[i] = f4scale(q0, RAL_LOG2E); reinterpret_cast<float4*>(Qs)[i + bd] = f4scale(q1, RAL_LOG2E);
        reinterpret_cast<float4*>(Ks)[i] = k0; reinterpret_cast<float4*>(Ks)[i + bd] = k1;
        reinterpret_cast<float4*>(Vs)[i] = v0; reinterpret_cast<float4*>(Vs)[i + bd] = v1;
        reinterpret_cast<float4*>(dOs)[i] = d0; reinterpret_cast<float4*>(dOs)[i + bd] = d1;
        Dl[i] = -f4dot(d0, o0); Dl[i + bd] = -f4dot(d1, o1);
        Ls[i] = -l0 * RAL_LOG2E; Ls[i + bd] = -l1 * RAL_LOG2E;
      }
      for (; i < n4; i += bd) {
        const float4 q0 = gq[i], k0 = gk[i], v0 = gv[i], d0 = gd[i], o0 = go[i];
        const float l0 = lse[hq0 + i];
        reinterpret_cast<float4*>(Qs)[i] = f4scale(q0, RAL_LOG2E);
        reinterpret_cast<float4*>(Ks)[i] = k0; reinterpret_cast<float4*>(Vs)[i] = v0; reinterpret_cast<float4*>(dOs)[i] = d0;
        Dl[i] = -f4dot(d0, o0); Ls[i] = -l0 * RAL_LOG2E;
      }
    }
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tab[i] = table[(i / HG) * H + h0 + (i % HG)] * RAL_LOG2E;
    __syncthreads();
    RAL_STAMP_AT(21);
#ifdef RAL_ATTNB_NOSWEEP   // diagnostic: staging only
    const int nblk = 0;
#else
    const int nblk = N / (16 * QT);
#endif
    // ---------------- sweep A: dQ (query block on the lanes, loop over key tiles) ----------------
    for (int task = wave; task < HG * nblk; task += nw) {
      const int hl = task / nblk, q0 = (task - hl * nblk) * 16 * QT;
      const float* Qh = Qs + hl * N * 4; const float* Kh = Ks + hl * N * 4;
      const float* Vh = Vs + hl * N * 4; const float* Dh = dOs + hl * N * 4;
      const float4* K4 = reinterpret_cast<const float4*>(Kh);
      float qf[QT], df[QT], lq[QT], dl[QT];
      f32x2 dq01[QT], dq23[QT];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const int q = q0 + 16 * qt + r;
        qf[qt] = Qh[q * 4 + g]; df[qt] = Dh[q * 4 + g];
        lq[qt] = Ls[hl * N + q]; dl[qt] = Dl[hl * N + q];
        dq01[qt] = f32x2{0.f, 0.f}; dq23[qt] = f32x2{0.f, 0.f};
      }
      auto tileA = [&](int kt, auto biased) {
        const float kf = Kh[(kt + r) * 4 + g], vf = Vh[(kt + r) * 4 + g];
        float4 k4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) k4[j] = K4[kt + 4 * g + j];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          f32x4 s = mfma4(kf, qf[qt], f32x4{lq[qt], lq[qt], lq[qt], lq[qt]});          // s - lse
          const f32x4 dp = mfma4(vf, df[qt], f32x4{dl[qt], dl[qt], dl[qt], dl[qt]});   // dP - delta
          if constexpr (RAG) {
            if (kt + 16 > NE) {
#pragma unroll
              for (int j = 0; j < 4; ++j) s[j] = (kt + 4 * g + j < NE) ? s[j] : -INFINITY;
            }
          }
          const int qi = q0 + 16 * qt + r - off;
          if constexpr (decltype(biased)::value) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ki = kt + 4 * g + j - off;
              if (qi >= 0 && qi < Len && ki >= 0 && ki < Len) s[j] += tab[(qi - ki + Len - 1) * HG + hl];
            }
          }
          float ds[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) ds[j] = __builtin_amdgcn_exp2f(s[j]) * dp[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dq01[qt] = pk_fma(splat2(ds[j]), f32x2{k4[j].x, k4[j].y}, dq01[qt]);
            dq23[qt] = pk_fma(splat2(ds[j]), f32x2{k4[j].z, k4[j].w}, dq23[qt]);
          }
          if constexpr (decltype(biased)::value) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ki = kt + 4 * g + j - off;
              if (qi >= 0 && qi < Len && ki >= 0 && ki < Len) atomicAdd(dtab + (qi - ki + Len - 1) * HG + hl, ds[j]);
            }
          }
        }
      };
      const bool qbias = table && (q0 < off + Len) && (q0 + 16 * QT > off);
      const int e0 = qbias ? kb0 : NEt, e1 = qbias ? kb1 : NEt;
      if constexpr (NT > 0 && NT <= 64) {   // compile-time trip count: unrolled, the table variant chosen per tile (wave-uniform)
#pragma unroll
        for (int kt = 0; kt < NT; kt += 16) {
          if (TAB && kt >= e0 && kt < e1) tileA(kt, std::true_type{});
          else tileA(kt, std::false_type{});
        }
      } else {
        for (int kt = 0; kt < e0; kt += 16) tileA(kt, std::false_type{});
        for (int kt = e0; kt < e1; kt += 16) tileA(kt, std::true_type{});
        for (int kt = e1; kt < NEt; kt += 16) tileA(kt, std::false_type{});
      }
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        float4 v = make_float4(dq01[qt][0], dq01[qt][1], dq23[qt][0], dq23[qt][1]);
        v = make_float4(rows_sum(v.x), rows_sum(v.y), rows_sum(v.z), rows_sum(v.w));
        if (g == 0)
          *reinterpret_cast<float4*>(dbase + ((size_t)(h0 + hl) * N + q0 + 16 * qt + r) * 4) = f4scale(v, 0.5f);
      }
    }
    RAL_STAMP_AT(22);
    // ---------------- sweep B: dK, dV (key block on the lanes, loop over query tiles) ----------------
    for (int task = wave; task < HG * nblk; task += nw) {
      const int hl = task / nblk, k0 = (task - hl * nblk) * 16 * QT;
      const float* Qh = Qs + hl * N * 4; const float* Kh = Ks + hl * N * 4;
      const float* Vh = Vs + hl * N * 4; const float* Dh = dOs + hl * N * 4;
      const float4* Q4 = reinterpret_cast<const float4*>(Qh);
      const float4* D4 = reinterpret_cast<const float4*>(Dh);
      float kf[QT], vf[QT];
      f32x2 dk01[QT], dk23[QT], dv01[QT], dv23[QT];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const int k = k0 + 16 * t + r;
        kf[t] = Kh[k * 4 + g]; vf[t] = Vh[k * 4 + g];
        dk01[t] = f32x2{0.f, 0.f}; dk23[t] = f32x2{0.f, 0.f}; dv01[t] = f32x2{0.f, 0.f}; dv23[t] = f32x2{0.f, 0.f};
      }
      auto tileB = [&](int qt, auto biased) {
        const float qa = Qh[(qt + r) * 4 + g], da = Dh[(qt + r) * 4 + g];
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + hl * N + qt + 4 * g);
        const float4 d4 = *reinterpret_cast<const float4*>(Dl + hl * N + qt + 4 * g);
        float4 q4[4], o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { q4[j] = Q4[qt + 4 * g + j]; o4[j] = D4[qt + 4 * g + j]; }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          f32x4 s = mfma4(qa, kf[t], f32x4{l4.x, l4.y, l4.z, l4.w});        // S[query 4g+j][key r] - lse
          const f32x4 dp = mfma4(da, vf[t], f32x4{d4.x, d4.y, d4.z, d4.w});  // dP[query][key] - delta
          if constexpr (RAG) {
            if (k0 + 16 * t + r >= NE) s = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};   // this lane's key does not exist
          }
          if constexpr (decltype(biased)::value) {
            const int ki = k0 + 16 * t + r - off;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int qi = qt + 4 * g + j - off;
              if (qi >= 0 && qi < Len && ki >= 0 && ki < Len) s[j] += tab[(qi - ki + Len - 1) * HG + hl];
            }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float p = __builtin_amdgcn_exp2f(s[j]);
            const float ds = p * dp[j];
            dv01[t] = pk_fma(splat2(p), f32x2{o4[j].x, o4[j].y}, dv01[t]);
            dv23[t] = pk_fma(splat2(p), f32x2{o4[j].z, o4[j].w}, dv23[t]);
            dk01[t] = pk_fma(splat2(ds), f32x2{q4[j].x, q4[j].y}, dk01[t]);
            dk23[t] = pk_fma(splat2(ds), f32x2{q4[j].z, q4[j].w}, dk23[t]);
          }
        }
      };
      const bool kbias = table && (k0 < off + Len) && (k0 + 16 * QT > off);
      const int e0 = kbias ? kb0 : NEt, e1 = kbias ? kb1 : NEt;
      if constexpr (NT > 0 && NT <= 64) {
#pragma unroll
        for (int qt = 0; qt < NT; qt += 16) {
          if (TAB && qt >= e0 && qt < e1) tileB(qt, std::true_type{});
          else tileB(qt, std::false_type{});
        }
      } else {
        for (int qt = 0; qt < e0; qt += 16) tileB(qt, std::false_type{});
        for (int qt = e0; qt < e1; qt += 16) tileB(qt, std::true_type{});
        for (int qt = e1; qt < NEt; qt += 16) tileB(qt, std::false_type{});   // (padding queries: dO = 0, nothing to add)
      }
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float4 vk = make_float4(dk01[t][0], dk01[t][1], dk23[t][0], dk23[t][1]);
        float4 vv = make_float4(dv01[t][0], dv01[t][1], dv23[t][0], dv23[t][1]);
        vk = make_float4(rows_sum(vk.x), rows_sum(vk.y), rows_sum(vk.z), rows_sum(vk.w));
        vv = make_float4(rows_sum(vv.x), rows_sum(vv.y), rows_sum(vv.z), rows_sum(vv.w));
        if (g == 0) {
          const size_t kk = (size_t)(h0 + hl) * N + k0 + 16 * t + r;
          *reinterpret_cast<float4*>(dbase + ((size_t)H * N + kk) * 4) = f4scale(vk, RAL_LN2);  // Qs carried log2(e)
          *reinterpret_cast<float4*>(dbase + ((size_t)2 * H * N + kk) * 4) = vv;
        }
      }
    }
    RAL_STAMP_AT(23);
    __syncthreads();
    RAL_STAMP_AT(24);
    if (ntab) {  // flush this item's table gradient (heads differ between items)
      for (int i = threadIdx.x; i < ntab; i += blockDim.x) {
        const float v = dtab[i];
        if (v != 0.f) atomicAdd(gtable + (i / HG) * H + h0 + (i % HG), v);
        dtab[i] = 0.f;
      }
      __syncthreads();
    }
  }
}

// =================================================================================
// B2v: attention backward on the scalar path (N >= 64), the counterpart of k_attn_fwd_v (ral_fwd.hip): two sweeps with
// one row per lane and the other operand wave-uniform (s_load, SGPR pairs into packed FMAs, no LDS, no barrier, no
// lane-group merge).  P is recomputed from q, k and the saved log-sum-exp.
//   sweep Q  (k_attn_bwd_vq):  lane = query i, uniform key j:   s, dP, dS = P (dP - delta_i),  dq_i += dS k_j
//                              also writes (-lse_i log2 e, -delta_i) for sweep KV and the R-wave table gradient
//   sweep KV (k_attn_bwd_vkv): lane = key j, uniform query i:   s, dP, dS,  dk_j += dS q_i,  dv_j += P do_i
// On gfx950 the fp32 MFMA and the vector ALU share the fp32 multipliers (no overlap on a SIMD: valu_probe.hip), so the
// S / dP tiles of k_attn_bwd cost their full 34 cycles each on top of the vector work; here they are 2 packed FMAs + 1 add
// per 64 scores (11 cycles) and the per-tile bookkeeping, the staging pass and the merges are gone.
// =================================================================================
RAL_DEV float wave_shfl_or0(float v, int src) {   // v of lane `src`, 0 when src is outside the wave
  const float r = __shfl(v, src & 63);
  return ((unsigned)src < 64u) ? r : 0.f;
}

template <bool BIAS>
__global__ __launch_bounds__(256) void k_attn_bwd_vq(const float* __restrict__ qkv, const float* __restrict__ o_hm,
                                                     const float* __restrict__ do_hm, const float* __restrict__ lse,
                                                     const float* __restrict__ table, float* __restrict__ tpart,
                                                     float* __restrict__ dqkv, float* __restrict__ stat2,
                                                     int N, int H, int Len, int ntask) {
  const int lane = threadIdx.x & 63;
  const int QB = (N + 63) >> 6;
  const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (task >= ntask) return;
  const int qb = task % QB, wh = task / QB, head = wh % H, win = wh / H;
  const int q = qb * 64 + lane, qc = q < N ? q : N - 1;
  const size_t wbase = (size_t)win * 3 * H * N;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)head * N;
  const float4* __restrict__ K4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(H + head) * N;
  const float4* __restrict__ V4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(2 * H + head) * N;
  const size_t hq = ((size_t)win * H + head) * N + qc;
  const float4 qv = f4scale(Q4[qc], RAL_LOG2E);
  const float4 dov = reinterpret_cast<const float4*>(do_hm)[hq], ov = reinterpret_cast<const float4*>(o_hm)[hq];
  const float nl = -lse[hq] * RAL_LOG2E, nd = -f4dot(dov, ov);
  if (q < N) *reinterpret_cast<float2*>(stat2 + hq * 2) = make_float2(nl, nd);
  float tval = 0.f, tacc = 0.f;      // R-wave table entry `lane` of this head (log2 units) and its gradient
  const int off = (N - Len) >> 1, qi = q - off;
  if constexpr (BIAS) {
    if (lane < 2 * Len - 1) tval = table[lane * H + head] * RAL_LOG2E;
  }
  const f32x2 q01 = {qv.x, qv.y}, q23 = {qv.z, qv.w}, d01 = {dov.x, dov.y}, d23 = {dov.z, dov.w};
  const f32x2 nl0 = {nl, 0.f}, nd0 = {nd, 0.f};
  f32x2 dq01 = {0.f, 0.f}, dq23 = {0.f, 0.f};
  auto body = [&](int kt, auto biased) {
    float4 k[4], v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { k[j] = K4[kt + j]; v[j] = V4[kt + j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 k01 = {k[j].x, k[j].y}, k23 = {k[j].z, k[j].w};
      f32x2 t = pk_fma(q01, k01, nl0);
      t = pk_fma(q23, k23, t);
      f32x2 u = pk_fma(d01, f32x2{v[j].x, v[j].y}, nd0);
      u = pk_fma(d23, f32x2{v[j].z, v[j].w}, u);
      float s = t[0] + t[1];
      const float dp = u[0] + u[1];
      bool inwin = false;
      if constexpr (decltype(biased)::value) {
        const int ki = kt + j - off;
        const float b = __shfl(tval, qi - ki + Len - 1);
        inwin = ki >= 0 && ki < Len && qi >= 0 && qi < Len;
        if (inwin) s += b;
      }
      const float ds = __builtin_amdgcn_exp2f(s) * dp;
      dq01 = pk_fma(splat2(ds), k01, dq01);
      dq23 = pk_fma(splat2(ds), k23, dq23);
      if constexpr (decltype(biased)::value) {
        // table entry e = qi - ki + Len - 1 lives on lane e: it takes dS from the lane whose query is e + ki - Len + 1
        const int ki = kt + j - off;
        tacc += wave_shfl_or0(inwin ? ds : 0.f, lane + ki - Len + 1 + off - qb * 64);
      }
    }
  };
  const int b0 = BIAS ? (off & ~3) : N, b1 = BIAS ? ((off + Len + 3) & ~3) : N;
  const bool qwin = BIAS && (qb * 64 < off + Len) && (qb * 64 + 64 > off);     // does this query block touch the window?
  const int e0 = qwin ? b0 : N, e1 = qwin ? b1 : N;
  for (int kt = 0; kt < e0; kt += 4) body(kt, std::false_type{});
  for (int kt = e0; kt < e1; kt += 4) body(kt, std::true_type{});
  for (int kt = e1; kt < N; kt += 4) body(kt, std::false_type{});
  if (q < N)
    *reinterpret_cast<float4*>(dqkv + (wbase + (size_t)head * N + q) * 4) =
        make_float4(0.5f * dq01[0], 0.5f * dq01[1], 0.5f * dq23[0], 0.5f * dq23[1]);   // q = 0.5 (h Wq^T + b)
  if constexpr (BIAS) {
    // The table gradient of this (window, head, query block) goes to a partial buffer, (B, 2, H, 64) floats: the
    // window is at most 32 wide, so it touches one or two query blocks (slot 0 / 1).  k_attn_table_reduce sums the
    // partials over the windows: a handful of atomics per table entry instead of one per window (all of a launch's
    // adds land in the table's 7 to 63 cache lines and serialise there: measured 350 ns per add at N = 64).
    if (qwin) tpart[(((size_t)win * 2 + (qb - off / 64)) * H + head) * 64 + lane] = tacc;
  }
}

// gtable[e][h] += sum over windows and slots of tpart[win][slot][h][e]; grid = chunks of windows, 256 threads
__global__ __launch_bounds__(256) void k_attn_table_reduce(const float* __restrict__ tpart, float* __restrict__ gtable,
                                                           int H, int Len, int nslot, int B, int wpb) {
  const int ne = 2 * Len - 1, w0 = blockIdx.x * wpb, w1 = min(B, w0 + wpb);
  for (int i = threadIdx.x; i < H * 64; i += blockDim.x) {
    const int h = i >> 6, e = i & 63;
    if (e >= ne) continue;
    float acc = 0.f;
    for (int w = w0; w < w1; ++w)
      for (int sl = 0; sl < nslot; ++sl) acc += tpart[(((size_t)w * 2 + sl) * H + h) * 64 + e];
    atomicAdd(gtable + e * H + h, acc);
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256) void k_attn_bwd_vkv(const float* __restrict__ qkv, const float* __restrict__ do_hm,
                                                      const float* __restrict__ stat2, const float* __restrict__ table,
                                                      float* __restrict__ dqkv, int N, int H, int Len, int ntask) {
  const int lane = threadIdx.x & 63;
  const int KB = (N + 63) >> 6;
  const int task = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (task >= ntask) return;
  const int kb = task % KB, wh = task / KB, head = wh % H, win = wh / H;
  const int kidx = kb * 64 + lane, kc = kidx < N ? kidx : N - 1;
  const size_t wbase = (size_t)win * 3 * H * N;
  const float4* __restrict__ Q4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)head * N;
  const float4* __restrict__ K4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(H + head) * N;
  const float4* __restrict__ V4 = reinterpret_cast<const float4*>(qkv) + wbase + (size_t)(2 * H + head) * N;
  const size_t hq0 = ((size_t)win * H + head) * N;
  const float4* __restrict__ D4 = reinterpret_cast<const float4*>(do_hm) + hq0;
  const float2* __restrict__ S2 = reinterpret_cast<const float2*>(stat2) + hq0;
  const float4 kv = f4scale(K4[kc], RAL_LOG2E), vv = V4[kc];
  const f32x2 k01 = {kv.x, kv.y}, k23 = {kv.z, kv.w}, v01 = {vv.x, vv.y}, v23 = {vv.z, vv.w};
  float tval = 0.f;
  const int off = (N - Len) >> 1, ki = kidx - off;
  if constexpr (BIAS) {
    if (lane < 2 * Len - 1) tval = table[lane * H + head] * RAL_LOG2E;
  }
  f32x2 dk01 = {0.f, 0.f}, dk23 = {0.f, 0.f}, dv01 = {0.f, 0.f}, dv23 = {0.f, 0.f};
  auto body = [&](int qt, auto biased) {
    float4 qr[4], dr[4];
    float2 st[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { qr[j] = Q4[qt + j]; dr[j] = D4[qt + j]; st[j] = S2[qt + j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 qa = {qr[j].x, qr[j].y}, qb2 = {qr[j].z, qr[j].w}, da = {dr[j].x, dr[j].y}, db = {dr[j].z, dr[j].w};
      f32x2 t = k01 * qa;
      t = pk_fma(k23, qb2, t);
      f32x2 u = v01 * da;
      u = pk_fma(v23, db, u);
      float hs = t[0] + t[1], hd = u[0] + u[1];
      asm volatile("" : "+v"(hs), "+v"(hd));      // (keeps the two horizontal adds scalar: packed, they need three moves)
      f32x2 sd = {hs, hd};
      sd += f32x2{st[j].x, st[j].y};              // (s - lse, dP - delta) in one packed add with the SGPR pair
      if constexpr (decltype(biased)::value) {
        const int qi = qt + j - off;              // uniform
        const float b = __shfl(tval, qi - ki + Len - 1);
        if (ki >= 0 && ki < Len && qi >= 0 && qi < Len) sd[0] += b;
      }
      const float p = __builtin_amdgcn_exp2f(sd[0]);
      const float ds = p * sd[1];
      dk01 = pk_fma(splat2(ds), qa, dk01);
      dk23 = pk_fma(splat2(ds), qb2, dk23);
      dv01 = pk_fma(splat2(p), da, dv01);
      dv23 = pk_fma(splat2(p), db, dv23);
    }
  };
  const int b0 = BIAS ? (off & ~3) : N, b1 = BIAS ? ((off + Len + 3) & ~3) : N;
  const bool kwin = BIAS && (kb * 64 < off + Len) && (kb * 64 + 64 > off);
  const int e0 = kwin ? b0 : N, e1 = kwin ? b1 : N;
  for (int qt = 0; qt < e0; qt += 4) body(qt, std::false_type{});
  for (int qt = e0; qt < e1; qt += 4) body(qt, std::true_type{});
  for (int qt = e1; qt < N; qt += 4) body(qt, std::false_type{});
  if (kidx < N) {
    float4* dk = reinterpret_cast<float4*>(dqkv) + wbase + (size_t)(H + head) * N + kidx;
    float4* dv = reinterpret_cast<float4*>(dqkv) + wbase + (size_t)(2 * H + head) * N + kidx;
    *dk = make_float4(dk01[0], dk01[1], dk23[0], dk23[1]);
    *dv = make_float4(dv01[0], dv01[1], dv23[0], dv23[1]);
  }
}

// =================================================================================
// B1: QKV projection + LN1 backward:  dh = dqkv Wqkv;  dx = dx1 + sqrt(C) * LN1bwd(dh)
//   grads: bqkv, ln1 w/b.   `extra` (optional) is added to dx (skip-connection gradient).
// =================================================================================
// FDW (narrow levels, C <= 32): the weight gradient of the projection, dWqkv[m][c] = sum_t dqkv[t][m] LN1(x)[t][c], and its bias
// gradient are formed HERE - both operands pass through this kernel anyway (dqkv sits in the LDS, the LayerNorm output is one
// FMA away from the row the LayerNorm backward holds), so the separate token-contraction launch of ral_dw.hip, which read dqkv
// (3E) and x (E) again from HBM and re-computed the LayerNorm, disappears for these blocks.  Every wave owns NJ (tile, token
// range) jobs for the whole kernel - fp32 MFMA tiles with the token as the contraction index, accumulators in registers over
// all the windows of the (persistent) workgroup -; the partials of the token ranges meet in the LDS at the end and leave as
// one pass of coalesced atomics.
template <int C> struct QkvDwShape {
  static constexpr int TM = (3 * C + 15) / 16, TN = (C + 15) / 16, TILES = TM * TN;
  static constexpr int KS = C == 32 ? 2 : 8;                        // token ranges per tile
  static constexpr int NJ = TILES * KS / 8;                          // jobs per wave (8 waves)
  static_assert(TILES * KS % 8 == 0, "whole jobs per wave");
};
template <int C, bool FDW = false>
// (dqkv, x, dx1 and extra are deliberately NOT __restrict__: see k_dw - loads the compiler can prove invariant are sunk
// across the compiler barrier of the prefetch, next to their uses, which puts the HBM round trip back in front of them)
#ifndef RAL_QKVB_MINB
#define RAL_QKVB_MINB 1
#endif
__global__ __launch_bounds__(512, RAL_QKVB_MINB) void k_qkv_bwd(const float* dqkv, const float* x,
                                                 const float* __restrict__ pe, const float* dx1,
                                                 const float* extra, BlockP w, BlockP wt, BlockP gr,
                                                 float* __restrict__ dx, int N, int B) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, LPR = C / 4;
  float* DQ = reinterpret_cast<float*>(smem4);  // HM, N x 3C
  float* Dh = DQ + N * 3 * C;                   // N x LD
  float* Hl = Dh + N * LD;                      // FDW: N x LD, LN1 output (the B operand of the weight-gradient tiles)
  float* red = Hl + (FDW ? N * LD : 0);         // 2C
  const int RPP = blockDim.x / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const float sqrtC = sqrtf((float)C);
  const float4 gam1 = *reinterpret_cast<const float4*>(w.ln1w + cq);
  float4 bet1 = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (FDW) bet1 = *reinterpret_cast<const float4*>(w.ln1b + cq);
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  using DWS = QkvDwShape<C>;
  f32x4 accw[FDW ? DWS::NJ : 1];
  float bsw[FDW ? DWS::NJ : 1];
#pragma unroll
  for (int j = 0; j < (FDW ? DWS::NJ : 1); ++j) { accw[j] = f32x4{0.f, 0.f, 0.f, 0.f}; bsw[j] = 0.f; }
  // One workgroup per CU is resident next to the weight-gradient kernels, so nothing else hides this kernel's HBM round
  // trips: the next window's operands are requested under the current window's LayerNorm phase.  NQ float4 of dqkv and
  // NR rows of x / dx1 / extra per thread are kept in flight (the BASELINE shapes have N C = 4096: NQ = 6, NR = 2 cover
  // a whole window); longer windows stage their remainder the plain way.
  constexpr int NQ = 6, NR = 2;
  const int n4 = N * 3 * C / 4;
  const bool pfq = n4 <= NQ * (int)blockDim.x;
  const int rbase = threadIdx.x / LPR;
  const float* const ex = extra ? extra : dx1;   // any readable address: no branch around a load
  // (plain arrays filled through by-value lambdas: a struct handed to a lambda by reference ends up in scratch memory)
  auto row_off = [&](int win, int u) -> size_t {
    return (size_t)win * N * C + (size_t)min(rbase + u * RPP, N - 1) * C + cq;
  };
  auto ld4 = [&](const float* base, size_t o) -> float4 { return *reinterpret_cast<const float4*>(base + o); };
  auto ld_dq = [&](int win, int u) -> float4 {
    return reinterpret_cast<const float4*>(dqkv + (size_t)win * N * 3 * C)[min((int)threadIdx.x + u * (int)blockDim.x, n4 - 1)];
  };
  auto st_dq = [&](int u, float4 v) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) reinterpret_cast<float4*>(DQ)[i] = v;
  };
  float4 pe4[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) pe4[u] = *reinterpret_cast<const float4*>(pe + min(rbase + u * RPP, N - 1) * C + cq);
  auto ln_row = [&](size_t wo, int row, float4 v, float4 p, float4 d1, float4 e) {
    v = f4add(f4scale(v, sqrtC), p);
    float4 d; float rstd;
    ln_stats<LPR>(v, d, rstd);
    const float4 xh = f4scale(d, rstd);
    const float4 dh = *reinterpret_cast<const float4*>(Dh + row * LD + cq);
    const float4 dyh = f4mul(dh, gam1);
    constexpr float invC = 1.0f / C;
    const float m1 = group_sum<LPR>(f4hsum(dyh)) * invC;
    const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invC;
    const float k = rstd * sqrtC;
    float4 out = make_float4(k * (dyh.x - m1 - xh.x * m2), k * (dyh.y - m1 - xh.y * m2),
                             k * (dyh.z - m1 - xh.z * m2), k * (dyh.w - m1 - xh.w * m2));
    out = f4add(out, d1);
    if (extra) out = f4add(out, e);
    *reinterpret_cast<float4*>(dx + wo + (size_t)row * C + cq) = out;
    if constexpr (FDW) *reinterpret_cast<float4*>(Hl + row * LD + cq) = f4add(f4mul(xh, gam1), bet1);
    dgam = f4add(dgam, f4mul(dh, xh));
    dbet = f4add(dbet, dh);
  };

  float4 rx[NR], rd[NR], re[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) rx[u] = rd[u] = re[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const int win0 = blockIdx.x;
    if (win0 < B) {
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const size_t o = row_off(win0, u);
        rx[u] = ld4(x, o); rd[u] = ld4(dx1, o); re[u] = ld4(ex, o);
      }
      if (pfq) {
        float4 q[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) q[u] = ld_dq(win0, u);
#pragma unroll
        for (int u = 0; u < NQ; ++u) st_dq(u, q[u]);
      } else {
        copy_flat(DQ, dqkv + (size_t)win0 * N * 3 * C, n4);
      }
    }
    __syncthreads();
  }
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const size_t wo = (size_t)win * N * C;
    // dh[t][c] = sum_m dqkv[t][m] Wqkv[m][c]   (K = 3C split as C (q rows) + 2C (kv rows))
    {
      auto run = [&](auto ttb_tag) {
        constexpr int TT = decltype(ttb_tag)::value;
        constexpr int mt = (C + 15) >> 4;
        const int tg = (N >> 4) / TT;
        for (int u = wave; u < mt * tg; u += nw) {
          const int m = u % mt, tgi = u / mt;
          f32x4 acc[TT];
#pragma unroll
          for (int tt = 0; tt < TT; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
          gemm_wx<C, TT, false, LAY_HM>(wt.wqkv, 3 * C, m * 16, C, DQ, N, tgi * TT * 16, acc);
          gemm_wx<2 * C, TT, false, LAY_HM>(wt.wqkv + C, 3 * C, m * 16, C, DQ + (C / 4) * N * 4, N, tgi * TT * 16, acc);
          const int row0 = m * 16 + 4 * g;
          if (row0 < C) {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
              *reinterpret_cast<float4*>(Dh + ((tgi * TT + tt) * 16 + r) * LD + row0) = tofloat4(acc[tt]);
          }
        }
      };
      const int nt = N >> 4;
      if ((nt & 3) == 0) run(std::integral_constant<int, 4>{});
      else if ((nt & 1) == 0) run(std::integral_constant<int, 2>{});
      else run(std::integral_constant<int, 1>{});
    }
    __syncthreads();   // Dh complete, DQ free
    const int nxt = win + gridDim.x;
    const bool more = nxt < B;
    const int pw = more ? nxt : win;   // (past the end: a harmless re-read, no branch around the loads)
    float4 q[NQ], nx[NR], nd[NR], ne[NR];
#pragma unroll
    for (int u = 0; u < NQ; ++u) q[u] = ld_dq(pw, u);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const size_t o = row_off(pw, u);
      nx[u] = ld4(x, o); nd[u] = ld4(dx1, o); ne[u] = ld4(ex, o);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int row = rbase + u * RPP;
      if (row < N) ln_row(wo, row, rx[u], pe4[u], rd[u], re[u]);
    }
    for (int row = rbase + NR * RPP; row < N; row += RPP) {
      const size_t o = wo + (size_t)row * C + cq;
      ln_row(wo, row, *reinterpret_cast<const float4*>(x + o), *reinterpret_cast<const float4*>(pe + row * C + cq),
             *reinterpret_cast<const float4*>(dx1 + o), *reinterpret_cast<const float4*>(ex + o));
    }
    if constexpr (FDW) {
      if (gr.wqkv) {   // (frozen weights, ral_backward_input: no weight gradients)
        lds_barrier();   // the window's LayerNorm output is in Hl; dqkv still in DQ (the next window's prefetch stays in flight)
        const int tlen = N / DWS::KS;
        // the jobs of a wave advance together, 16 tokens at a time: the operand reads of all of them are issued before the
        // first product, and consecutive MFMAs go to different accumulators
        const float* Ap[DWS::NJ]; const float* Bp[DWS::NJ];
#pragma unroll
        for (int j = 0; j < DWS::NJ; ++j) {
          const int job = wave + 8 * j, tile = job % DWS::TILES, kpart = job / DWS::TILES;
          const int mi = tile / DWS::TN, nj = tile % DWS::TN;
          int am = mi * 16 + r, bc = nj * 16 + r;                 // operand columns of this lane, clamped into the matrix
          am = am < 3 * C ? am : 3 * C - 1;
          bc = bc < C ? bc : C - 1;
          Ap[j] = DQ + ((am >> 2) * N + kpart * tlen + 4 * g) * 4 + (am & 3);   // dqkv[t][am] at + 4 t (head-major quads)
          Bp[j] = Hl + (kpart * tlen + 4 * g) * LD + bc;                        // LN1(x)[t][bc] at + t LD
        }
        for (int t0 = 0; t0 < tlen; t0 += 16) {
          float av[DWS::NJ][4], bv[DWS::NJ][4];
#pragma unroll
          for (int j = 0; j < DWS::NJ; ++j)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) { av[j][s_] = Ap[j][4 * (t0 + s_)]; bv[j][s_] = Bp[j][(t0 + s_) * LD]; }
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int j = 0; j < DWS::NJ; ++j) {
              accw[j] = mfma4(av[j][s_], bv[j][s_], accw[j]);
              bsw[j] += av[j][s_];
            }
        }
        lds_barrier();   // DQ may take the next window now
      }
    }
    if (more) {
      if (pfq) {
#pragma unroll
        for (int u = 0; u < NQ; ++u) st_dq(u, q[u]);
      } else {
        copy_flat(DQ, dqkv + (size_t)nxt * N * 3 * C, n4);
      }
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) { rx[u] = nx[u]; rd[u] = nd[u]; re[u] = ne[u]; }
    __syncthreads();
  }
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, C + cq, dbet);
  __syncthreads();
  if ((int)threadIdx.x < C) {
#ifndef RAL_NOVECATOMICS
    atomicAdd(gr.ln1w + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(gr.ln1b + threadIdx.x, (float)redd[C + threadIdx.x]);
#endif
  }
  if constexpr (FDW) {
    if (!gr.wqkv) return;
    // ---- flush dWqkv / dbqkv: the token-range partials meet in an LDS image of the matrix (3C x C floats + 3C, behind the small
    // sums above), then consecutive threads add consecutive floats (global float atomics run at full rate for 256 contiguous bytes)
    float* img = reinterpret_cast<float*>(smem4) + 4 * C + 8;   // (past the 2C doubles of redd)
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C * C + 3 * C; i += blockDim.x) img[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < DWS::NJ; ++j) {
      const int job = wave + 8 * j, tile = job % DWS::TILES;
      const int mi = tile / DWS::TN, nj = tile % DWS::TN;
      const int col = nj * 16 + r;
#pragma unroll
      for (int q_ = 0; q_ < 4; ++q_) {
        const int row = mi * 16 + 4 * g + q_;
        if (row < 3 * C && col < C) atomicAdd(img + row * C + col, accw[j][q_]);
      }
      const float bs = rows_sum(bsw[j]);         // column sums of dqkv over this job's tokens: lane (r, *) holds row mi * 16 + r
      if (nj == 0 && g == 0 && mi * 16 + r < 3 * C) atomicAdd(img + 3 * C * C + mi * 16 + r, bs);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C * C; i += blockDim.x) atomicAdd(gr.wqkv + i, img[i]);
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) atomicAdd(gr.bqkv + i, img[3 * C * C + i]);
  }
}

// =================================================================================
// B1h (wide levels, C >= 64, N C <= 4096): B1 with dh = dqkv Wqkv on the f16 matrix cores (two fp16 pieces per operand,
// gemm_phase_h2).  The dqkv window sits in LDS as token-major split planes, every token row scaled by a power of two
// (h2_row_scale; a row's 3C values arrive spread over the workgroup - head-major quads -, so its maximum is raised with LDS
// atomics from the prefetch registers and the split happens one barrier later); the product is unscaled into Dh.
// wtt: tiled split planes of the transposed weights (Wqkv^T: C x 3C).  Everything else as B1.
// =================================================================================
template <int C>
__global__ __launch_bounds__(512, RAL_QKVB_MINB) void k_qkv_bwd_h(const float* dqkv, const float* x,
                                                 const float* __restrict__ pe, const float* dx1,
                                                 const float* extra, BlockP w, BlockP wt, const float* __restrict__ ptbase,
                                                 const _Float16* __restrict__ wtt, BlockP gr,
                                                 float* __restrict__ dx, unsigned* __restrict__ gmax, int N, int B,
                                                 int NS /* token segments per window (1, 2): a work item is one segment; gridDim.x % NS == 0 */) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<C>::v, LPR = C / 4, LDQ = ldb_of(3 * C);
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  // Everything in this kernel is per token (row maxima, the product, the LayerNorm backward), so a work item may be a SEGMENT of
  // a window: NT = N / NS tokens from token t0.  A workgroup always gets the same segment index (gridDim.x % NS == 0), so its
  // positional-encoding rows stay in registers.
  const int NT = N / NS, t0 = ((int)blockIdx.x % NS) * NT, items = B * NS;
  _Float16* Qh = reinterpret_cast<_Float16*>(smem4);   // 2 x NT x LDQ : dqkv, token-major, scaled, split
  const int qplane = NT * LDQ;
  float* Dh = reinterpret_cast<float*>(Qh + 2 * qplane);   // NT x LD
  float* red = Dh + NT * LD;                    // 2C
  unsigned* smQ = reinterpret_cast<unsigned*>(red + 2 * C);   // 2 x NT : bits of max |dqkv row|, by item parity
  const int RPP = blockDim.x / LPR;
  const int cq = (threadIdx.x % LPR) * 4;
  const float sqrtC = sqrtf((float)C);
  const float4 gam1 = *reinterpret_cast<const float4*>(w.ln1w + cq);
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  const _Float16* wq = wtt + 2 * (wt.wqkv - ptbase);
  const float wunq = wplane_unscale(wq, C, 3 * C);
  constexpr int NQ = 6, NR = 2;
  const int n4 = NT * 3 * C / 4;   // <= NQ * blockDim.x (launcher)
  const int rbase = threadIdx.x / LPR;
  const float* const ex = extra ? extra : dx1;
  auto row_off = [&](int item, int u) -> size_t {
    return (size_t)(item / NS) * N * C + (size_t)(t0 + min(rbase + u * RPP, NT - 1)) * C + cq;
  };
  auto ld4 = [&](const float* base, size_t o) -> float4 { return *reinterpret_cast<const float4*>(base + o); };
  auto ld_dq = [&](int item, int u) -> float4 {   // flat float4 i of the segment = (channel quad i / NT, token t0 + i % NT) of the window
    const int i = min((int)threadIdx.x + u * (int)blockDim.x, n4 - 1), qd = i / NT, t = i - qd * NT;
    return reinterpret_cast<const float4*>(dqkv + (size_t)(item / NS) * N * 3 * C)[qd * N + t0 + t];
  };
  float tmxq = 0.f;   // largest |dqkv| this thread has seen: gmax[3], for the weight-gradient kernel (see k_mlp_bwd_h)
  auto max_dq = [&](unsigned* sm, int u, float4 v) {   // flat float4 i = (channel quad, token)
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) {
      const float mx = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
      tmxq = fmaxf(tmxq, mx);
      atomicMax(sm + i % NT, __float_as_uint(mx));
    }
  };
  auto st_dq = [&](const unsigned* sm, int u, float4 v) {
    const int i = threadIdx.x + u * blockDim.x;
    if (i < n4) {
      const int qd = i / NT, t = i - qd * NT;
      v = f4scale(v, h2_row_scale(sm[t]));
      const H2 s0 = f16_split2u(v.x), s1 = f16_split2u(v.y), s2 = f16_split2u(v.z), s3 = f16_split2u(v.w);
      *reinterpret_cast<f16x4*>(Qh + t * LDQ + qd * 4) = f16x4{s0.a, s1.a, s2.a, s3.a};
      *reinterpret_cast<f16x4*>(Qh + qplane + t * LDQ + qd * 4) = f16x4{s0.b, s1.b, s2.b, s3.b};
    }
  };
  float4 pe4[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) pe4[u] = *reinterpret_cast<const float4*>(pe + (t0 + min(rbase + u * RPP, NT - 1)) * C + cq);
  auto ln_row = [&](size_t wo, int row, float4 v, float4 p, float4 d1, float4 e) {
    v = f4add(f4scale(v, sqrtC), p);
    float4 d; float rstd;
    ln_stats<LPR>(v, d, rstd);
    const float4 xh = f4scale(d, rstd);
    const float4 dh = *reinterpret_cast<const float4*>(Dh + row * LD + cq);
    const float4 dyh = f4mul(dh, gam1);
    constexpr float invC = 1.0f / C;
    const float m1 = group_sum<LPR>(f4hsum(dyh)) * invC;
    const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invC;
    const float k = rstd * sqrtC;
    float4 out = make_float4(k * (dyh.x - m1 - xh.x * m2), k * (dyh.y - m1 - xh.y * m2),
                             k * (dyh.z - m1 - xh.z * m2), k * (dyh.w - m1 - xh.w * m2));
    out = f4add(out, d1);
    if (extra) out = f4add(out, e);
    *reinterpret_cast<float4*>(dx + wo + (size_t)row * C + cq) = out;
    dgam = f4add(dgam, f4mul(dh, xh));
    dbet = f4add(dbet, dh);
  };

  float4 rx[NR], rd[NR], re[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) rx[u] = rd[u] = re[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < 2 * NT; i += blockDim.x) smQ[i] = 0u;
  __syncthreads();
  {
    const int win0 = blockIdx.x;
    if (win0 < items) {
#pragma unroll
      for (int u = 0; u < NR; ++u) {
        const size_t o = row_off(win0, u);
        rx[u] = ld4(x, o); rd[u] = ld4(dx1, o); re[u] = ld4(ex, o);
      }
      float4 q[NQ];
#pragma unroll
      for (int u = 0; u < NQ; ++u) q[u] = ld_dq(win0, u);
#pragma unroll
      for (int u = 0; u < NQ; ++u) max_dq(smQ, u, q[u]);
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NQ; ++u) st_dq(smQ, u, q[u]);
    }
    __syncthreads();
  }
  int par = 0;
  for (int win = blockIdx.x; win < items; win += gridDim.x, par ^= 1) {   // (win: the item)
    const size_t wo = (size_t)(win / NS) * N * C + (size_t)t0 * C;
    const unsigned* smc = smQ + par * NT;
    unsigned* smn = smQ + (par ^ 1) * NT;
    if ((int)threadIdx.x < NT) smn[threadIdx.x] = 0u;   // (last read by the previous item's product)
    // dh[t][c] = sum_m dqkv[t][m] Wqkv[m][c]
    gemm_phase_h2<3 * C, 0, 2>(wq, 3 * C / 32, 0, 0, C, nullptr, 1.0f, Qh, qplane, LDQ, NT >> 4, [&](int row0, int tok, f32x4 a) {
      *reinterpret_cast<float4*>(Dh + tok * LD + row0) = f4scale(tofloat4(a), h2_row_unscale(smc[tok]) * wunq);
    });
    __syncthreads();   // Dh complete, Qh free
    const int nxt = win + gridDim.x;
    const bool more = nxt < items;
    const int pw = more ? nxt : win;   // (past the end: a harmless re-read, no branch around the loads)
    float4 q[NQ], nx[NR], nd[NR], ne[NR];
#pragma unroll
    for (int u = 0; u < NQ; ++u) q[u] = ld_dq(pw, u);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const size_t o = row_off(pw, u);
      nx[u] = ld4(x, o); nd[u] = ld4(dx1, o); ne[u] = ld4(ex, o);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int row = rbase + u * RPP;
      if (row < NT) ln_row(wo, row, rx[u], pe4[u], rd[u], re[u]);
    }
    for (int row = rbase + NR * RPP; row < NT; row += RPP) {
      const size_t o = wo + (size_t)row * C + cq;
      ln_row(wo, row, *reinterpret_cast<const float4*>(x + o), *reinterpret_cast<const float4*>(pe + (t0 + row) * C + cq),
             *reinterpret_cast<const float4*>(dx1 + o), *reinterpret_cast<const float4*>(ex + o));
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < NQ; ++u) max_dq(smn, u, q[u]);
    }
    __syncthreads();
    if (more) {
#pragma unroll
      for (int u = 0; u < NQ; ++u) st_dq(smn, u, q[u]);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) { rx[u] = nx[u]; rd[u] = nd[u]; re[u] = ne[u]; }
    __syncthreads();
  }
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, C + cq, dbet);
  __syncthreads();
  if ((int)threadIdx.x < C) {
    atomicAdd(gr.ln1w + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(gr.ln1b + threadIdx.x, (float)redd[C + threadIdx.x]);
  }
  if (gmax) {
    __syncthreads();
    if (threadIdx.x == 0) smQ[0] = 0u;
    __syncthreads();
    const float mq = group_max<64>(tmxq);
    if ((threadIdx.x & 63) == 0) atomicMax(smQ, __float_as_uint(mq));
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(gmax + 3, smQ[0]);
  }
}

// =================================================================================
// PatchMerging / PatchSeparate backward: dh = dy W;  dx = LNbwd(dh) scattered back to
// the input layout;  grads: LN w/b.  (dW of the reduction is a ral_dw.hip product.)
// =================================================================================
template <int D, bool SEP>
// (dy and x are deliberately NOT __restrict__: see k_dw - loads the compiler can prove invariant are sunk across the compiler
// barrier of the prefetch, next to their uses)
__global__ __launch_bounds__(256) void k_resample_bwd(const float* dy, const float* x,
                                                      const float* __restrict__ wred, const float* __restrict__ lnw,
                                                      float* __restrict__ g_lnw, float* __restrict__ g_lnb,
                                                      float* __restrict__ dx, int T, int Tv, int B) {
  extern __shared__ float4 smem4[];
  constexpr int LD = LDof<D>::v, LPR = D / 4, RPP = 256 / LPR;
  float* Ys = reinterpret_cast<float*>(smem4);  // T x LD
  float* Dh = Ys + T * LD;                      // T x LD
  float* red = Dh + T * LD;                     // 2D
  const int cq = (threadIdx.x % LPR) * 4;
  const float4 gam = *reinterpret_cast<const float4*>(lnw + cq);
  float4 dgam = make_float4(0.f, 0.f, 0.f, 0.f), dbet = make_float4(0.f, 0.f, 0.f, 0.f);
  auto src_of = [&](size_t wo, int row) -> size_t {
    return SEP ? wo + sep_src(row, T, Tv, D) : wo + (size_t)row * D;
  };
  auto ln_row = [&](int row, size_t src, float4 v) {   // LayerNorm backward of one row (all LPR lanes of the row take part)
    float4 d; float rstd;
    ln_stats<LPR>(v, d, rstd);
    if (row < T) {
      const float4 xh = f4scale(d, rstd);
      const float4 dh = *reinterpret_cast<const float4*>(Dh + row * LD + cq);
      const float4 dyh = f4mul(dh, gam);
      constexpr float invD = 1.0f / D;
      const float m1 = group_sum<LPR>(f4hsum(dyh)) * invD;
      const float m2 = group_sum<LPR>(f4dot(dyh, xh)) * invD;
      *reinterpret_cast<float4*>(dx + src + cq) =
          make_float4(rstd * (dyh.x - m1 - xh.x * m2), rstd * (dyh.y - m1 - xh.y * m2),
                      rstd * (dyh.z - m1 - xh.z * m2), rstd * (dyh.w - m1 - xh.w * m2));
      dgam = f4add(dgam, f4mul(dh, xh));
      dbet = f4add(dbet, dh);
    }
  };
  if (T == 4 * RPP) {
    // 512-sample windows (T D = 4096: four float4 of dy and four rows of x per thread).  One workgroup per CU runs next to
    // the weight-gradient kernels, so nothing else hides this kernel's HBM round trips: the x rows of the LayerNorm
    // phase are requested before the product, the next window's dy under the LayerNorm phase - one barrier-free round
    // trip per window instead of two exposed ones (the eight launches of a step: 0.38 -> 0.32 ms).
    constexpr int q = D / 4;
    float4 dv[4], xv[4];
    // (filled in place, not through a lambda: an array handed to a lambda by reference ends up in scratch memory)
#define RAL_REQ_DY(win_) do { const float4* s4_ = reinterpret_cast<const float4*>(dy + (size_t)(win_) * T * D); \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) dv[k] = s4_[threadIdx.x + k * 256]; } while (0)
    if ((int)blockIdx.x < B) RAL_REQ_DY(blockIdx.x);
    else { dv[0] = dv[1] = dv[2] = dv[3] = make_float4(0.f, 0.f, 0.f, 0.f); }
    for (int win = blockIdx.x; win < B; win += gridDim.x) {
      const size_t wo = (size_t)win * T * D;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + k * 256;
        *reinterpret_cast<float4*>(Ys + (i / q) * LD + (i % q) * 4) = dv[k];
      }
      __syncthreads();
      size_t src4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        src4[u] = src_of(wo, threadIdx.x / LPR + u * RPP);
        xv[u] = *reinterpret_cast<const float4*>(x + src4[u] + cq);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      gemm_phase<D, TTBof<D>::v, false, LAY_TOK>(wred /* W^T */, D, D, Ys, LD, T >> 4, [&](int row0, int tok, f32x4 a) {
        *reinterpret_cast<float4*>(Dh + tok * LD + row0) = tofloat4(a);
      });
      __syncthreads();
      const int nxt = win + gridDim.x;
      RAL_REQ_DY(nxt < B ? nxt : win);   // (past the end: a harmless re-read, no branch around the loads)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) ln_row(threadIdx.x / LPR + u * RPP, src4[u], xv[u]);
    }
#undef RAL_REQ_DY
    __syncthreads();
  } else {
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const size_t wo = (size_t)win * T * D;
    copy_in(Ys, LD, dy + wo, D, T, D);
    __syncthreads();
    gemm_phase<D, TTBof<D>::v, false, LAY_TOK>(wred /* W^T */, D, D, Ys, LD, T >> 4, [&](int row0, int tok, f32x4 a) {
      *reinterpret_cast<float4*>(Dh + tok * LD + row0) = tofloat4(a);
    });
    __syncthreads();
    // LayerNorm backward, four rows per thread with their x loads in flight together (one HBM round trip, not four)
    for (int row0 = threadIdx.x / LPR; row0 < T; row0 += 4 * RPP) {
      float4 v4[4]; size_t src4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        src4[u] = src_of(wo, min(row0 + u * RPP, T - 1));
        v4[u] = *reinterpret_cast<const float4*>(x + src4[u] + cq);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) ln_row(row0 + u * RPP, src4[u], v4[u]);
    }
    __syncthreads();
  }
  }
  double* redd = reinterpret_cast<double*>(smem4);
  for (int i = threadIdx.x; i < 2 * D; i += blockDim.x) redd[i] = 0.;
  __syncthreads();
  lds_add4(redd, cq, dgam);
  lds_add4(redd, D + cq, dbet);
  __syncthreads();
  if ((int)threadIdx.x < D) {
#ifndef RAL_NOVECATOMICS
    atomicAdd(g_lnw + threadIdx.x, (float)redd[threadIdx.x]);
    atomicAdd(g_lnb + threadIdx.x, (float)redd[D + threadIdx.x]);
#endif
  }
}

// =================================================================================
// output conv backward: dz[b][l][c] = sum_o sum_k w[o][c][k] dy[b][o][l-k+1]  (token-major, 8 ch)
//   grads: transconv w (leads x 8 x 3), b (leads);  z = u0 + x0 is re-formed on the fly
// =================================================================================
template <int LEADS>
__global__ __launch_bounds__(256) void k_final_bwd(const float* __restrict__ dy, const float* __restrict__ u0,
                                                   const float* __restrict__ x0, const float* __restrict__ w,
                                                   float* __restrict__ gw, float* __restrict__ gb,
                                                   float* __restrict__ dz, int L, int Lp, int B) {
  // L: samples per window of dy; Lp >= L: token slots per window of u0 / x0 / dz (slots past L: no gradient, they do not exist)
  constexpr int NG = LEADS * 24 + LEADS;
  __shared__ float red[4][NG];
  float wr[LEADS][8][3];
#pragma unroll
  for (int o = 0; o < LEADS; ++o)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) wr[o][c][k] = w[(o * 8 + c) * 3 + k];
  float acc[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) acc[i] = 0.f;
  const size_t total = (size_t)B * Lp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Lp), l = (int)(i - (size_t)b * Lp);
    if (l >= L) {
      float4* pz0 = reinterpret_cast<float4*>(dz + i * 8);
      pz0[0] = make_float4(0.f, 0.f, 0.f, 0.f); pz0[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    float dyv[LEADS][3];  // dy[o][l-1], dy[o][l], dy[o][l+1]
#pragma unroll
    for (int o = 0; o < LEADS; ++o) {
      const float* dr = dy + ((size_t)b * LEADS + o) * L;
      dyv[o][0] = l > 0 ? dr[l - 1] : 0.f;
      dyv[o][1] = dr[l];
      dyv[o][2] = l < L - 1 ? dr[l + 1] : 0.f;
    }
    // dz[l][c] = sum_o ( w[o][c][0] dy[o][l+1] + w[o][c][1] dy[o][l] + w[o][c][2] dy[o][l-1] )
    float z8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float a = 0.f;
#pragma unroll
      for (int o = 0; o < LEADS; ++o)
        a += wr[o][c][0] * dyv[o][2] + wr[o][c][1] * dyv[o][1] + wr[o][c][2] * dyv[o][0];
      z8[c] = a;
    }
    float4* pz = reinterpret_cast<float4*>(dz + i * 8);
    pz[0] = make_float4(z8[0], z8[1], z8[2], z8[3]);
    pz[1] = make_float4(z8[4], z8[5], z8[6], z8[7]);
    // weight grads: gw[o][c][k] += dy[o][l'] z[l'+k-1][c]  summed over l'  ==  z[l][c] * dy[o][l-k+1]
    const float4 a0 = reinterpret_cast<const float4*>(u0)[i * 2], a1 = reinterpret_cast<const float4*>(u0)[i * 2 + 1];
    const float4 b0 = reinterpret_cast<const float4*>(x0)[i * 2], b1 = reinterpret_cast<const float4*>(x0)[i * 2 + 1];
    const float z[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w,
                        a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
#pragma unroll
    for (int o = 0; o < LEADS; ++o) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[(o * 8 + c) * 3 + k] += z[c] * dyv[o][2 - k];
      acc[LEADS * 24 + o] += dyv[o][1];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const float s = group_sum<64>(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < NG) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if ((int)threadIdx.x < LEADS * 24) atomicAdd(gw + threadIdx.x, s);
    else atomicAdd(gb + threadIdx.x - LEADS * 24, s);
  }
}

// =================================================================================
// conv stem backward.  Pass 1: per-channel sums of dy and dy*xhat (BatchNorm backward).
// Pass 2: d a0 = gamma rstd (dy - mean(dy) - xhat mean(dy xhat)); LeakyReLU'; conv w/b grads.
// ss = [scale(8), shift(8), mean(8), rstd(8)] from k_bn_train8.
// =================================================================================
__global__ __launch_bounds__(256) void k_bn8_bwd_stats(const float* __restrict__ dy, const float* __restrict__ a0,
                                                       const float* __restrict__ ss, double* __restrict__ out,
                                                       size_t ntok) {
  __shared__ double red[16 * 4];
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < ntok; t += (size_t)gridDim.x * blockDim.x) {
    const float4 d0 = reinterpret_cast<const float4*>(dy)[t * 2], d1 = reinterpret_cast<const float4*>(dy)[t * 2 + 1];
    const float4 v0 = reinterpret_cast<const float4*>(a0)[t * 2], v1 = reinterpret_cast<const float4*>(a0)[t * 2 + 1];
    const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float xh = (v[c] - ss[16 + c]) * ss[24 + c];
      acc[c] += d[c];
      acc[8 + c] += d[c] * xh;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float s = group_sum<64>(acc[i]);
    if (lane == 0) red[wave * 16 + i] = (double)s;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w * 16 + threadIdx.x];
    atomicAdd(out + threadIdx.x, t);
  }
}

template <int LEADS>
__global__ __launch_bounds__(256) void k_conv1_bwd(const float* __restrict__ dy, const float* __restrict__ a0,
                                                   const float* __restrict__ x, const float* __restrict__ ss,
                                                   const float* __restrict__ bnw, const double* __restrict__ bst,
                                                   double count, float* __restrict__ gw, float* __restrict__ gb,
                                                   float* __restrict__ dzout, int L, int Lp, int B,
                                                   float* __restrict__ gbnw, float* __restrict__ gbnb, double share) {
  // L: samples per window of x; Lp >= L: token slots per window of dy / a0 / dzout (slots past L do not exist: no gradient)
  constexpr int NG = 8 * LEADS * 3 + 8;
  __shared__ float red[4][NG];
  float k1[8], m1[8], m2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    k1[c] = bnw[c] * ss[24 + c];
    m1[c] = (float)(bst[c] / count);
    m2[c] = (float)(bst[8 + c] / count);
  }
  float acc[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) acc[i] = 0.f;
  const size_t total = (size_t)B * Lp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Lp), l = (int)(i - (size_t)b * Lp);
    if (l >= L) {
      if (dzout) {
        float4* pz0 = reinterpret_cast<float4*>(dzout + i * 8);
        pz0[0] = make_float4(0.f, 0.f, 0.f, 0.f); pz0[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    const float4 d0 = reinterpret_cast<const float4*>(dy)[i * 2], d1 = reinterpret_cast<const float4*>(dy)[i * 2 + 1];
    const float4 v0 = reinterpret_cast<const float4*>(a0)[i * 2], v1 = reinterpret_cast<const float4*>(a0)[i * 2 + 1];
    const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    float xv[LEADS][3];
#pragma unroll
    for (int c = 0; c < LEADS; ++c) {
      const float* xr = x + ((size_t)b * LEADS + c) * L;
      xv[c][0] = l > 0 ? xr[l - 1] : 0.f;
      xv[c][1] = xr[l];
      xv[c][2] = l < L - 1 ? xr[l + 1] : 0.f;
    }
    float dz[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const float xh = (v[o] - ss[16 + o]) * ss[24 + o];
      float da = k1[o] * (d[o] - m1[o] - xh * m2[o]);
      da = v[o] > 0.f ? da : 0.2f * da;
      dz[o] = da;
#pragma unroll
      for (int c = 0; c < LEADS; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[(o * LEADS + c) * 3 + k] += da * xv[c][k];
      acc[8 * LEADS * 3 + o] += da;
    }
    if (dzout) {
      float4* pz = reinterpret_cast<float4*>(dzout + i * 8);
      pz[0] = make_float4(dz[0], dz[1], dz[2], dz[3]);
      pz[1] = make_float4(dz[4], dz[5], dz[6], dz[7]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const float s = group_sum<64>(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < NG) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if ((int)threadIdx.x < 8 * LEADS * 3) atomicAdd(gw + threadIdx.x, s);
    else atomicAdd(gb + threadIdx.x - 8 * LEADS * 3, s);
  }
  // BatchNorm affine grads from the (all-reduced) backward sums: g_w += sum dy*xhat, g_b += sum dy (one workgroup's eight
  // threads; it was a launch of its own).  `share` = this rank's fraction of the global batch: the sums are already all-reduced,
  // and the gradient all-reduce (sum over ranks) that follows must reproduce them exactly once.
  if (blockIdx.x == 0 && threadIdx.x < 8) {
    const int c = threadIdx.x;
    gbnb[c] += (float)(bst[c] * share); gbnw[c] += (float)(bst[8 + c] * share);
  }
}

// dx[b][c][l] = sum_o sum_k w[o][c][k] dz[b][l-k+1][o]   (input gradient of the stem, 12-lead adapter only)
template <int LEADS>
__global__ void k_conv1_bwd_dx(const float* __restrict__ dz, const float* __restrict__ w, float* __restrict__ dx,
                               int L, int Lp, int B) {
  const size_t total = (size_t)B * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (size_t)b * L);
    float acc[LEADS];
#pragma unroll
    for (int c = 0; c < LEADS; ++c) acc[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ll = l - k + 1;
      if (ll < 0 || ll >= L) continue;
      const float* dr = dz + ((size_t)b * Lp + ll) * 8;
#pragma unroll
      for (int o = 0; o < 8; ++o)
#pragma unroll
        for (int c = 0; c < LEADS; ++c) acc[c] += w[(o * LEADS + c) * 3 + k] * dr[o];
    }
#pragma unroll
    for (int c = 0; c < LEADS; ++c) dx[((size_t)b * LEADS + c) * L + l] = acc[c];
  }
}

// =================================================================================
// host launchers
// =================================================================================
// Workgroups of the persistent backward kernels.  Each workgroup ends by adding its share of the parameter gradients
// (LayerNorm affines, biases, local-enhancement taps: 5 C .. 8 C floats) to global memory with atomics and pays its
// set-up once, so fewer, longer-lived workgroups win well before the CUs run out of work.  Measured at batch 2048 (two
// lanes of 1024 windows), training step with 1024 workgroups everywhere: 18.12 ms; k_qkv_bwd at 256: 17.81; k_resample_bwd
// at 256: 17.95; k_mlp_bwd at 512: 18.06; all three: 17.45 ms; k_mlp_bwd_s: flat between 384 and 512, slower above.
// Re-measured once k_qkv_bwd prefetched its operands and the weight-gradient launches had grown to 192 workgroups (same
// box, ms per step): k_qkv_bwd at 128: 17.22, 160: 17.13, 192: 17.12, 224: 17.84 (1024 windows over 224 workgroups leave
// a ragged last round), 256: 17.29 - the CUs it leaves free go to the weight-gradient kernels running beside it; then
// k_mlp_bwd_s at 256: 17.11, 320: 17.08, 384: 16.99, 448: 17.02, 512: 17.04.
// (RAL_GRID_QKVB / RESB / MLPB / MLPS / ATTNB override.)
static inline int env_grid(const char* name, int dflt) { return (int)ral_knob(name + 4, dflt); }   // (name = "RAL_<KNOB>")
static inline int cap(int items, int gmax) { return items < gmax ? items : gmax; }
static inline int ew_grid(size_t n, int per = 256) {
  size_t g = (n + per - 1) / per;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

size_t mlp_bwd_lds(int C, int N, int nch) {
  return ((size_t)2 * N * ld_of(C) + (size_t)N * ld_of(4 * C / nch) + 2 * (N + 2) + 2 * N + 8 * C + 16) * sizeof(float);
}

// narrow levels: fused weight-gradient variant (k_mlp_bwd_s) when the window length gives each wave a whole
// number (1, 2, 4 or 8) of dg token tiles; RAL_FUSE_DW=0 keeps the separate dW kernels
// does the fused narrow-level kernel take (C, N)?  The forward asks too: it does not store u_pre for such blocks
template <int C>
static bool mlp_bwd_s_applies(int N, int* tw_out, size_t* lds_out) {
  static const int maxc = (int)ral_knob("FUSE_DW", 32);
  constexpr int MT = C >= 16 ? C / 16 : 1;
  if (C > maxc || (N * MT) % 128 != 0) return false;
  const int tw = N * MT / 128;
  if (tw != 1 && tw != 2 && tw != 4 && tw != 8) return false;
  const int ldb = C == 16 ? C : ld_of(C);
  size_t lds = ((size_t)2 * N * ld_of(C) + (size_t)2 * N * ldb + 2 * (N + 2) + 2 * N + 2 * C + 8) * sizeof(float);
  if (lds > 80 * 1024) return false;
  const size_t flush = ((size_t)6656 + 8 * C * C) * sizeof(float);   // fold region + the two staged weight-gradient matrices
  if (lds < flush) lds = flush;
  if (tw_out) *tw_out = tw;
  if (lds_out) *lds_out = lds;
  return true;
}
bool mlp_bwd_is_fused(int C, int N) {
  switch (C) {
    case 8: return mlp_bwd_s_applies<8>(N, nullptr, nullptr);
    case 16: return mlp_bwd_s_applies<16>(N, nullptr, nullptr);
    case 32: return mlp_bwd_s_applies<32>(N, nullptr, nullptr);
  }
  return false;
}

template <int C>
static bool launch_mlp_bwd_s(const float* dx2, const float* x1, const BlockP& w, const BlockP& wt,
                             const BlockP& gr, float* dx1, float* do_hm, int N, int B, bool want_dw, hipStream_t s, int NE) {
  // widest level that takes the fused kernel: RAL_FUSE_DW (0 disables it, 8 / 16 narrow it).  Measured at batch 2048:
  // none 102.8k, C <= 8 103.8k, C <= 16 105.5k, C <= 32 105.8k windows/s (at C = 32 the weight-gradient MFMAs are
  // no longer negligible on the critical stream, so the gain flattens)
  int tw; size_t lds;
  if (!mlp_bwd_s_applies<C>(N, &tw, &lds)) return false;
  static const int gs = env_grid("RAL_GRID_MLPS", 384);
  const int grid = cap(B, gs);
#define GO(t) { RAL_SET_LDS((k_mlp_bwd_s<C, t>), lds); k_mlp_bwd_s<C, t><<<grid, 512, lds, s>>>(dx2, x1, w, wt, gr, dx1, do_hm, N, B, want_dw ? 1 : 0, NE); return true; }
  switch (tw) { case 1: GO(1) case 2: GO(2) case 4: GO(4) case 8: GO(8) default: return false; }
#undef GO
}

// wide levels on split fp16 operands (RAL_MLP_F16=0: the fp32-MFMA kernel everywhere): the hidden-chunk count that gives
// the fc2^T phase exactly one 32 x 32 unit per wave and fits the LDS budget, or 0
size_t mlp_bwd_h_lds(int C, int N, int nch, bool small = false) {   // small: the four-wave form (no separate dg tile)
  const int HC = 4 * C / nch;
  return (size_t)2 * N * ldb_of(C) * 2 + (small ? 0 : (size_t)N * ld_of(C) * 4) + (size_t)N * (HC + 8) * 4 + ((size_t)2 * (N + 2) + 5 * N + 2 * C + 4) * 4;
}
// Threads of a k_mlp_bwd_h workgroup (MLPB_HTHREADS).  256 (default): four waves, 37-39 KB of LDS, 150 registers per lane, three
// workgroups per CU; 512: eight waves, 75-78 KB, 128 registers (28-68 bytes of scratch), two per CU.  Measured at batch 2048: the
// four-wave form is the SLOWER kernel on an empty GPU (`mlp_bwd` 2.66 -> 2.76 ms per step serialised) and the faster training step
// (12.98 -> 12.88 ms, 13.57 -> 13.42 on a slower box; interleaved A/B, three rounds each).  What it changes beside the other lane's
// kernels and the 76 KB weight-gradient workgroups: half the LDS and half the waves per workgroup, no spills, and no dg
// read-modify-write through LDS per hidden chunk (k_qkv_bwd_h with half the LDS and waves alone - QKVB_SEG - did not move the step).
static int mlp_bwd_h_threads() {
  static const int t = (int)ral_knob("MLPB_HTHREADS", 256);
  return t == 512 ? 512 : 256;
}
int mlp_bwd_h_nch(int C, int N) {
  static const bool on = (ral_knob("MLP_F16", 1) != 0);
  if (!on || (C != 32 && C != 64 && C != 128) || N % 32 != 0) return 0;
  const int nwaves = mlp_bwd_h_threads() / 64;
  if (nwaves == 4) {   // N C = 4096: one 32 x 32 unit of dg per wave; HC + 8 >= C + 4: the dg tile fits the chunk's bytes
    for (int nch = 1; nch <= 4; nch *= 2)
      if (N * C == 4096 && 4 * C / nch >= C && (4 * C / nch / 32) * (N / 32) == 4 && mlp_bwd_h_lds(C, N, nch, true) <= 54400) return nch;
  } else
  for (int nch = 1; nch <= 4; nch *= 2)
    if ((4 * C / nch / 32) * (N / 32) == 8 && mlp_bwd_h_lds(C, N, nch) <= 79872) return nch;
  if (nwaves != 8)   // (a shape the small workgroups do not cover keeps the eight-wave form)
    for (int nch = 1; nch <= 4; nch *= 2)
      if ((4 * C / nch / 32) * (N / 32) == 8 && mlp_bwd_h_lds(C, N, nch) <= 79872) return -nch;
  return 0;
}
template <int C>
static void launch_mlp_bwd_hc(int nch, const float* dx2, const float* x1, const float* upre, const BlockP& w, const BlockP& wt,
                              const float* ptbase, const void* wtt, const BlockP& gr, float* dupre, float* dx1, float* do_hm,
                              float* a2c0, unsigned* gmax, int N, int B, hipStream_t s) {
  const bool small = nch > 0 && mlp_bwd_h_threads() == 256;
  if (nch < 0) nch = -nch;
  const size_t lds = mlp_bwd_h_lds(C, N, nch, small);
  static const int gm = env_grid("RAL_GRID_MLPB", 512);
  const int grid = cap(B, small ? gm * 3 / 2 : gm);
  const _Float16* wp = reinterpret_cast<const _Float16*>(wtt);
  if (small) {
    if (nch == 2) { RAL_SET_LDS((k_mlp_bwd_h<C, 2, 256>), lds); k_mlp_bwd_h<C, 2, 256><<<grid, 256, lds, s>>>(dx2, x1, upre, w, wt, ptbase, wp, gr, dupre, dx1, do_hm, a2c0, gmax, N, B); }
    else { RAL_SET_LDS((k_mlp_bwd_h<C, 4, 256>), lds); k_mlp_bwd_h<C, 4, 256><<<grid, 256, lds, s>>>(dx2, x1, upre, w, wt, ptbase, wp, gr, dupre, dx1, do_hm, a2c0, gmax, N, B); }
    return;
  }
  if (nch == 1) { RAL_SET_LDS((k_mlp_bwd_h<C, 1>), lds); k_mlp_bwd_h<C, 1><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, ptbase, wp, gr, dupre, dx1, do_hm, a2c0, gmax, N, B); }
  else if (nch == 2) { RAL_SET_LDS((k_mlp_bwd_h<C, 2>), lds); k_mlp_bwd_h<C, 2><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, ptbase, wp, gr, dupre, dx1, do_hm, a2c0, gmax, N, B); }
  else { RAL_SET_LDS((k_mlp_bwd_h<C, 4>), lds); k_mlp_bwd_h<C, 4><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, ptbase, wp, gr, dupre, dx1, do_hm, a2c0, gmax, N, B); }
}

template <int C>
static bool launch_mlp_bwd_c(int nch, const float* dx2, const float* x1, const float* upre, const BlockP& w,
                             const BlockP& wt, const BlockP& gr, float* dupre, float* dx1, float* do_hm, float* a2c0, int N, int B,
                             bool want_dw, hipStream_t s, int NE) {
  if constexpr (C <= 32) {
    if (launch_mlp_bwd_s<C>(dx2, x1, w, wt, gr, dx1, do_hm, N, B, want_dw, s, NE)) return true;
  }
  const size_t lds = mlp_bwd_lds(C, N, nch);
  static const int gm = env_grid("RAL_GRID_MLPB", 512);
  const int grid = cap(B, gm);
  if (nch == 1) { RAL_SET_LDS((k_mlp_bwd<C, 1>), lds); k_mlp_bwd<C, 1><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, gr, dupre, dx1, do_hm, a2c0, N, B, NE); }
  else if (nch == 2) { RAL_SET_LDS((k_mlp_bwd<C, 2>), lds); k_mlp_bwd<C, 2><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, gr, dupre, dx1, do_hm, a2c0, N, B, NE); }
  else { RAL_SET_LDS((k_mlp_bwd<C, 4>), lds); k_mlp_bwd<C, 4><<<grid, 512, lds, s>>>(dx2, x1, upre, w, wt, gr, dupre, dx1, do_hm, a2c0, N, B, NE); }
  return false;
}

// returns true when the fc1 / fc2 weight (and bias) gradients were produced here (narrow levels): the caller then
// skips those two products in launch_block_dw
bool launch_mlp_bwd(int C, int nch, const float* dx2, const float* x1, const float* upre, const BlockP& w,
                    const BlockP& wt, const float* ptbase, const void* wtt, unsigned* gmax, const BlockP& gr, float* dupre, float* dx1, float* do_hm,
                    float* a2c0, int N, int B, bool want_dw, hipStream_t s, int f16_narrow, int NE) {
  if (!want_dw) { dupre = nullptr; a2c0 = nullptr; }   // consumed by the weight-gradient kernels only
  if (NE <= 0 || NE > N) NE = N;
  const bool padded = NE < N;   // padded windows: the generic kernels (their local-enhancement conv knows where the window ends)
  if (!padded && wtt && upre) {   // (upre == nullptr: a level whose forward does not store u_pre - the fused narrow-level kernel re-computes it)
    const int nh = mlp_bwd_h_nch(C, N);
    if (nh && C == 32) { launch_mlp_bwd_hc<32>(nh, dx2, x1, upre, w, wt, ptbase, wtt, gr, dupre, dx1, do_hm, a2c0, want_dw ? gmax : nullptr, N, B, s); return false; }
    if (nh && C == 64) { launch_mlp_bwd_hc<64>(nh, dx2, x1, upre, w, wt, ptbase, wtt, gr, dupre, dx1, do_hm, a2c0, want_dw ? gmax : nullptr, N, B, s); return false; }
    if (nh && C == 128) { launch_mlp_bwd_hc<128>(nh, dx2, x1, upre, w, wt, ptbase, wtt, gr, dupre, dx1, do_hm, a2c0, want_dw ? gmax : nullptr, N, B, s); return false; }
  }
  if (!padded && !upre && mlp_bwd_is_fused(C, N)) {   // strip kernels (ral_mlpw.hip), fc1 / fc2 weight gradients fused
    if (const int kind = mlp_bwd_w_kind(C, N, f16_narrow != 0)) {
      launch_mlp_bwd_w(C, kind, dx2, x1, w, gr, dx1, do_hm, N, B, want_dw, s);
      return true;
    }
  }
  switch (C) {
#define CASE(c) case c: return launch_mlp_bwd_c<c>(nch, dx2, x1, upre, w, wt, gr, dupre, dx1, do_hm, a2c0, N, B, want_dw, s, NE);
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
  return false;
}

size_t attn_bwd_lds(int N, int HG, int Len) {
  return ((size_t)4 * HG * N * 4 + (size_t)2 * HG * N + (Len > 0 ? (size_t)2 * (2 * Len - 1) * HG : 0) + 4) * sizeof(float);
}

// Window lengths that take the scalar-path sweeps.  Measured at batch 2048 (tools/attn_bench.py, us per launch, MFMA-tile
// kernel vs scalar path).  Without an R-wave table: N = 512: 790 / 853, 256: 446 / 464, 128: 285 / 265, 64: 208 / 155;
// with one (the in-window keys cost two lane gathers each, plus the partial-sum pass): 128: 285 / 293, 64: 208 / 192, and
// inside the training step (bench.py --kinds) the N = 64 case with a table came out 2 % slower than the MFMA-tile kernel.
// So: N <= 128 without a table, never with one.  The switches ATTN_BWD_V_LO / _HI force a range for both cases (0 / 0 = never).
bool attn_bwd_uses_stat2(int N, int Len, bool table) {
  static int lo = -1, hi = -1;
  static const bool init = [] { lo = (int)ral_knob("ATTN_BWD_V_LO", -1); hi = (int)ral_knob("ATTN_BWD_V_HI", -1); return true; }();
  (void)init;
  if (N < 64 || N % 4 != 0 || (table && 2 * Len - 1 > 64)) return false;
  if (lo >= 0) return N >= lo && N <= hi;
  return !table && N <= 128;
}

size_t attn_bwd_scratch_floats(int N, int H, int Len, bool table, int B) {
  if (attn_bwd_m_takes(N, H, Len, table)) return attn_bwd_m_scratch_floats(N, H, Len, table, B);
  if (attn_bwd_mh_takes(N, H, Len, table)) return attn_bwd_mh_scratch_floats(N, H, Len, table, B);
  if (attn_bwd_w_takes(N, H, Len, table)) return attn_bwd_w_scratch_floats(N, H, Len, table, B);
  if (!attn_bwd_uses_stat2(N, Len, table)) return 0;
  return (size_t)B * H * N * 2 + (size_t)B * 2 * H * 64;
}

void launch_attn_bwd(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                     float* gtable, float* dqkv, float* stat2, size_t scratch_floats, int N, int H, int HG, int Len, int B,
                     int f16, hipStream_t s, int NE) {
  if (NE > 0 && NE < N) {   // padded windows (NE of the N token slots exist): the generic tile kernel with its key mask
    const size_t lds = attn_bwd_lds(N, HG, Len);
    const int items = B * (H / HG), grid = items < 4096 ? items : 4096;
    RAL_SET_LDS((k_attn_bwd<1, 0, true, true>), lds);
    k_attn_bwd<1, 0, true, true><<<grid, 512, lds, s>>>(qkv, o_hm, do_hm, lse, table, gtable, dqkv, N, H, HG, Len, B, NE);
    return;
  }
  // one sweep with every contraction on the f16 matrix cores (ral_attnm.hip)
  if (f16 && attn_bwd_m_takes(N, H, Len, table != nullptr) &&
      (stat2 ? scratch_floats : 0) >= attn_bwd_m_scratch_floats(N, H, Len, table != nullptr, B)) {
    launch_attn_bwd_m(qkv, o_hm, do_hm, lse, table, gtable, dqkv, stat2, N, H, Len, B, s);
    return;
  }
  if (f16 && attn_bwd_mh_takes(N, H, Len, table != nullptr) &&
      (stat2 ? scratch_floats : 0) >= attn_bwd_mh_scratch_floats(N, H, Len, table != nullptr, B)) {
    launch_attn_bwd_mh(qkv, o_hm, do_hm, lse, table, gtable, dqkv, stat2, N, H, Len, B, s);
    return;
  }
  // short windows: one wave per head, no workgroup barriers (ral_attn.hip)
  if (attn_bwd_w_takes(N, H, Len, table != nullptr) &&
      (stat2 ? scratch_floats : 0) >= attn_bwd_w_scratch_floats(N, H, Len, table != nullptr, B)) {
    launch_attn_bwd_w(qkv, o_hm, do_hm, lse, table, gtable, dqkv, stat2, N, H, Len, B, 0, s);   // (fp32 tiles: strict mode, or the one-sweep kernels switched off)
    return;
  }
  if (stat2 && scratch_floats < (size_t)B * H * N * 2 + (size_t)B * 2 * H * 64) stat2 = nullptr;
  // the two scalar-path sweeps; stat2 is scratch: (B, H, N, 2) floats handed from the query sweep to the key / value
  // sweep, followed by (B, 2, H, 64) floats of R-wave table-gradient partials
  if (stat2 && attn_bwd_uses_stat2(N, Len, table != nullptr)) {
    const int ntask = B * H * ((N + 63) / 64), grid = (ntask + 3) / 4;
    if (table) {
      float* tpart = stat2 + (size_t)B * H * N * 2;
      const int off = (N - Len) / 2, nslot = (off + Len - 1) / 64 - off / 64 + 1;
      k_attn_bwd_vq<true><<<grid, 256, 0, s>>>(qkv, o_hm, do_hm, lse, table, tpart, dqkv, stat2, N, H, Len, ntask);
      k_attn_bwd_vkv<true><<<grid, 256, 0, s>>>(qkv, do_hm, stat2, table, dqkv, N, H, Len, ntask);
      const int wpb = 32;
      k_attn_table_reduce<<<(B + wpb - 1) / wpb, 256, 0, s>>>(tpart, gtable, H, Len, nslot, B, wpb);
    } else {
      k_attn_bwd_vq<false><<<grid, 256, 0, s>>>(qkv, o_hm, do_hm, lse, nullptr, nullptr, dqkv, stat2, N, H, 0, ntask);
      k_attn_bwd_vkv<false><<<grid, 256, 0, s>>>(qkv, do_hm, stat2, nullptr, dqkv, N, H, 0, ntask);
    }
    return;
  }
  static const int split_env = (int)ral_knob("ATTN_SPLIT", 2);   // see launch_attn_fwd
  int split = split_env;
  while (split > 1 && (HG % split != 0 || N % 32 != 0)) split /= 2;
  if (split > 1) {
    const int hg = HG / split;
    const size_t l2 = attn_bwd_lds(N, hg, Len);
    const int it2 = B * (H / hg);
    static const bool nt_off = (ral_knob("ATTNB_NT0", 0) != 0);   // experiment knob: run-time window length everywhere
    const int grid2 = cap(it2, env_grid("RAL_GRID_ATTNB", 8192));
#define NTCASE(n, tab) { RAL_SET_LDS((k_attn_bwd<2, n, tab>), l2); \
      k_attn_bwd<2, n, tab><<<grid2, 512 / split, l2, s>>>(qkv, o_hm, do_hm, lse, table, gtable, dqkv, N, H, hg, Len, B); return; }
    if (!nt_off) {
      if (N == 32 && !table) NTCASE(32, false)
      if (N == 64 && table) NTCASE(64, true)
    }
#undef NTCASE
    RAL_SET_LDS((k_attn_bwd<2>), l2);
    k_attn_bwd<2><<<grid2, 512 / split, l2, s>>>(qkv, o_hm, do_hm, lse, table, gtable, dqkv, N, H, hg, Len, B);
    return;
  }
  const size_t lds = attn_bwd_lds(N, HG, Len);
  const int items = B * (H / HG);
  const int grid = items < 4096 ? items : 4096;
  static const bool force1 = (ral_knob("ATTN_QT1", 0) != 0);   // experiment knob
  if (N % 32 == 0 && !force1) {
    RAL_SET_LDS((k_attn_bwd<2>), lds);
    k_attn_bwd<2><<<grid, 512, lds, s>>>(qkv, o_hm, do_hm, lse, table, gtable, dqkv, N, H, HG, Len, B);
  } else {
    RAL_SET_LDS((k_attn_bwd<1>), lds);
    k_attn_bwd<1><<<grid, 512, lds, s>>>(qkv, o_hm, do_hm, lse, table, gtable, dqkv, N, H, HG, Len, B);
  }
}

size_t qkv_bwd_lds(int C, int N) { return ((size_t)N * 3 * C + (size_t)N * ld_of(C) + 5 * C + 8) * sizeof(float); }
// QKVB_FDW = 1: the narrow levels form the projection's weight gradient inside k_qkv_bwd instead of the separate launch of
// ral_dw.hip.  Built in round 6 for its HBM bytes (-4E per narrow block: 0.5 GB of the step's 33 GB) and measured: the
// weight-gradient kind 2.91 -> 2.51 ms per step serialised, k_qkv_bwd 1.10 -> 1.35 ms - and the STEP 0.06 ms slower (12.90 ->
// 12.96 ms, same box, three interleaved rounds): the separate launch runs on the side stream and is bound by HBM, which the
// chain kernels beside it leave idle; inside k_qkv_bwd the same product is fp32-MFMA tiles fed by 4-byte LDS reads on the
// critical path.  Default OFF; the switch and tests/test_gpu_configs.py keep it alive.
bool qkv_bwd_fuses_dw(int C, int N) {
  static const bool on = (ral_knob("QKVB_FDW", 0) != 0);
  if (!on || C > 32) return false;
  const int ks = C == 32 ? 2 : 8;
  if (qkv_bwd_lds(C, N) + (size_t)N * ld_of(C) * sizeof(float) > 156 * 1024) return false;   // (C = 8 at 1024 tokens: the LayerNorm-output tile does not fit)
  return N % (16 * ks) == 0;
}

bool qkv_bwd_uses_f16(int C, int N) {
  static const bool on = (ral_knob("QKVB_F16", 1) != 0);
  return on && (C == 32 || C == 64 || C == 128) && N % 32 == 0 && N * 3 * C / 4 <= 6 * 512;
}
bool launch_qkv_bwd(int C, const float* dqkv, const float* x, const float* pe, const float* dx1, const float* extra,
                    const BlockP& w, const BlockP& wt, const float* ptbase, const void* wtt, unsigned* gmax, const BlockP& gr, float* dx, int N, int B,
                    bool want_dw, hipStream_t s) {
  static const int gq = env_grid("RAL_GRID_QKVB", 192);
  const int grid = cap(B, gq);
  if (wtt && qkv_bwd_uses_f16(C, N)) {
    // QKVB_SEG = 2: a work item is HALF a window (four-wave workgroups, half the LDS) where half a window still is a whole number of
    // 32-token product units and one prefetch round of 256 threads (N C = 4096, N >= 64).  Measured: `qkv_bwd` 1.105 -> 1.075 ms per
    // step serialised, the step unchanged (13.03 / 13.02 ms, and flat over grids of 128 .. 256 workgroups): default off.
    static const int seg = (int)ral_knob("QKVB_SEG", 1);
    const int NS = (seg == 2 && N * C == 4096 && N % 64 == 0) ? 2 : 1, NT = N / NS, nth = 512 / NS;
    const size_t ldsh = (size_t)2 * NT * ldb_of(3 * C) * 2 + ((size_t)NT * ld_of(C) + 2 * C + 2 * NT) * 4;
    const _Float16* wp = reinterpret_cast<const _Float16*>(wtt);
    const int gridh = NS == 1 ? grid : (cap(B * NS, gq * NS) / NS) * NS;
    if (C == 32) { RAL_SET_LDS((k_qkv_bwd_h<32>), ldsh); k_qkv_bwd_h<32><<<gridh, nth, ldsh, s>>>(dqkv, x, pe, dx1, extra, w, wt, ptbase, wp, gr, dx, gmax, N, B, NS); }
    else if (C == 64) { RAL_SET_LDS((k_qkv_bwd_h<64>), ldsh); k_qkv_bwd_h<64><<<gridh, nth, ldsh, s>>>(dqkv, x, pe, dx1, extra, w, wt, ptbase, wp, gr, dx, gmax, N, B, NS); }
    else { RAL_SET_LDS((k_qkv_bwd_h<128>), ldsh); k_qkv_bwd_h<128><<<gridh, nth, ldsh, s>>>(dqkv, x, pe, dx1, extra, w, wt, ptbase, wp, gr, dx, gmax, N, B, NS); }
    return false;
  }
  const size_t lds = qkv_bwd_lds(C, N);
  if (qkv_bwd_fuses_dw(C, N)) {   // narrow levels: weight gradient of the projection inside the kernel
    size_t ldsf = lds + (size_t)N * ld_of(C) * sizeof(float);
    const size_t img = ((size_t)4 * C + 8 + 3 * C * C + 3 * C) * sizeof(float);
    if (ldsf < img) ldsf = img;
    BlockP g2 = gr;
    if (!want_dw) g2.wqkv = nullptr;   // (frozen weights: the kernel skips the tiles)
    switch (C) {
#define CASE(c) case c: RAL_SET_LDS((k_qkv_bwd<c, true>), ldsf); \
      k_qkv_bwd<c, true><<<grid, 512, ldsf, s>>>(dqkv, x, pe, dx1, extra, w, wt, g2, dx, N, B); break;
      CASE(8) CASE(16) CASE(32)
#undef CASE
    }
    return true;
  }
  switch (C) {
#define CASE(c) case c: RAL_SET_LDS((k_qkv_bwd<c>), lds); \
    k_qkv_bwd<c><<<grid, 512, lds, s>>>(dqkv, x, pe, dx1, extra, w, wt, gr, dx, N, B); break;
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
  return false;
}

void launch_resample_bwd(int D, bool sep, const float* dy, const float* x, const float* wred, const float* lnw,
                         float* g_lnw, float* g_lnb, float* dx, int T, int Tv, int B, hipStream_t s) {
  const size_t lds = ((size_t)2 * T * ld_of(D) + 2 * D + 4) * sizeof(float);
  static const int gr = env_grid("RAL_GRID_RESB", 256);
  const int grid = cap(B, gr);
#define CASE(d) case d: if (sep) { RAL_SET_LDS((k_resample_bwd<d, true>), lds); k_resample_bwd<d, true><<<grid, 256, lds, s>>>(dy, x, wred, lnw, g_lnw, g_lnb, dx, T, Tv, B); } \
                        else { RAL_SET_LDS((k_resample_bwd<d, false>), lds); k_resample_bwd<d, false><<<grid, 256, lds, s>>>(dy, x, wred, lnw, g_lnw, g_lnb, dx, T, Tv, B); } break;
  switch (D) { CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) }
#undef CASE
}

void launch_final_bwd(int leads, const float* dy, const float* u0, const float* x0, const float* w, float* gw,
                      float* gb, float* dz, int L, int Lp, int B, hipStream_t s) {
  const int grid = ew_grid((size_t)B * Lp, 1024);
  if (leads == 1) k_final_bwd<1><<<grid, 256, 0, s>>>(dy, u0, x0, w, gw, gb, dz, L, Lp, B);
  else k_final_bwd<2><<<grid, 256, 0, s>>>(dy, u0, x0, w, gw, gb, dz, L, Lp, B);
}

void launch_bn8_bwd_stats(const float* dy, const float* a0, const float* ss, double* out, size_t ntok, hipStream_t s) {
  k_bn8_bwd_stats<<<ew_grid(ntok, 1024), 256, 0, s>>>(dy, a0, ss, out, ntok);
}

void launch_conv1_bwd(int leads, const float* dy, const float* a0, const float* x, const float* ss, const float* bnw,
                      const double* bst, double count, float* gw, float* gb, float* dz, int L, int Lp, int B, hipStream_t s,
                      float* gbnw, float* gbnb, double share) {
  const int grid = ew_grid((size_t)B * Lp, 1024);
  if (leads == 1) k_conv1_bwd<1><<<grid, 256, 0, s>>>(dy, a0, x, ss, bnw, bst, count, gw, gb, dz, L, Lp, B, gbnw, gbnb, share);
  else k_conv1_bwd<2><<<grid, 256, 0, s>>>(dy, a0, x, ss, bnw, bst, count, gw, gb, dz, L, Lp, B, gbnw, gbnb, share);
}

void launch_conv1_bwd_dx(int leads, const float* dz, const float* w, float* dx, int L, int Lp, int B, hipStream_t s) {
  const int grid = ew_grid((size_t)B * L);
  if (leads == 1) k_conv1_bwd_dx<1><<<grid, 256, 0, s>>>(dz, w, dx, L, Lp, B);
  else k_conv1_bwd_dx<2><<<grid, 256, 0, s>>>(dz, w, dx, L, Lp, B);
}
