// Weight-gradient products of the Linear layers:  dW[m][n] += sum_tokens Y[t][m] * X[t][n].
// Both operands are staged per window (or per token chunk of a window) in LDS; the X
// operand is RE-COMPUTED from the tensors the forward pass saved (LayerNorm, the
// GELU / local-enhancement chain) rather than stored.  Each wave keeps its share of the
// 16x16 output tiles in MFMA accumulators across all the windows its workgroup owns
// and adds them to the gradient buffer once at the end (fp32 global atomics).
#include "ral_device.hpp"
#include "ral_kernels.hpp"

enum { XF_HM = 0, XF_LN = 1, XF_LNPE = 2, XF_A2 = 3, XF_LN_SEP = 4 };

template <int M, int NC, int LAYY, int XF>
__global__ __launch_bounds__(512) void k_dw(const float* __restrict__ Y, const float* __restrict__ X,
                                            const float* __restrict__ pe, const float* __restrict__ lnw,
                                            const float* __restrict__ lnb, const float* __restrict__ le,
                                            float* __restrict__ dW, int N, int TC, int B) {
  extern __shared__ float4 smem4[];
  constexpr int MT = (M + 15) / 16, NT = (NC + 15) / 16, NWV = 8, TPW = (MT * NT + NWV - 1) / NWV;
  constexpr int LAYX = (XF == XF_HM) ? LAY_HM : LAY_TOK;
  constexpr int LDY = LDof<M>::v, LDX = LDof<NC>::v;
  float* Ys = reinterpret_cast<float*>(smem4);
  float* Xs = Ys + (LAYY == LAY_HM ? TC * M : TC * LDY);
  float* A0 = Xs + (LAYX == LAY_HM ? TC * NC : TC * LDX);  // N + 2 (XF_A2 with LE only)
  const int ldy = (LAYY == LAY_HM) ? TC : LDY, ldx = (LAYX == LAY_HM) ? TC : LDX;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  f32x4 acc[TPW];
  int mo[TPW], no[TPW];
  bool valid[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int id = wave + NWV * i;
    valid[i] = id < MT * NT;
    mo[i] = valid[i] ? (id / NT) * 16 : 0;
    no[i] = valid[i] ? (id % NT) * 16 : 0;
  }
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  if (XF == XF_A2 && le) { lw0 = le[0]; lw1 = le[1]; lw2 = le[2]; }

  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* Yw = Y + (size_t)win * N * M;
    const float* Xw = X + (size_t)win * N * NC;
    if (XF == XF_A2 && le) {  // gelu(u[:,0]) of the whole window, zero halo
      for (int i = threadIdx.x; i < N + 2; i += blockDim.x)
        A0[i] = (i == 0 || i == N + 1) ? 0.f : gelu_f(Xw[(size_t)(i - 1) * NC]);
    }
    for (int t0 = 0; t0 < N; t0 += TC) {
      // ---- stage Y ----
      if (LAYY == LAY_HM) {
        for (int i = threadIdx.x; i < (M / 4) * TC; i += blockDim.x) {
          const int q = i / TC, tt = i - q * TC;
          reinterpret_cast<float4*>(Ys)[i] = reinterpret_cast<const float4*>(Yw)[(size_t)q * N + t0 + tt];
        }
      } else {
        copy_in(Ys, LDY, Yw + (size_t)t0 * M, M, TC, M);
      }
      // ---- stage / re-compute X ----
      if constexpr (XF == XF_HM) {
        for (int i = threadIdx.x; i < (NC / 4) * TC; i += blockDim.x) {
          const int q = i / TC, tt = i - q * TC;
          reinterpret_cast<float4*>(Xs)[i] = reinterpret_cast<const float4*>(Xw)[(size_t)q * N + t0 + tt];
        }
      } else if constexpr (XF == XF_A2) {
        if (le) __syncthreads();  // A0 ready
        for (int i = threadIdx.x; i < TC * (NC / 4); i += blockDim.x) {
          const int row = i / (NC / 4), q = i - row * (NC / 4);
          float4 a = *reinterpret_cast<const float4*>(Xw + (size_t)(t0 + row) * NC + 4 * q);
          a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w);
          if (le) {
            if (q == 0) a.x = lw0 * A0[t0 + row] + lw1 * A0[t0 + row + 1] + lw2 * A0[t0 + row + 2];
            a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w);
          }
          *reinterpret_cast<float4*>(Xs + row * LDX + 4 * q) = a;
        }
      } else {  // LayerNorm family
        constexpr int LPR = NC / 4;
        const int RPP = blockDim.x / LPR;
        const int cq = (threadIdx.x % LPR) * 4;
        const float4 gam = *reinterpret_cast<const float4*>(lnw + cq);
        const float4 bet = *reinterpret_cast<const float4*>(lnb + cq);
        const float sq = sqrtf((float)NC);
        for (int row = threadIdx.x / LPR; row < TC; row += RPP) {
          const int t = t0 + row;
          const float* src = (XF == XF_LN_SEP) ? Xw + (size_t)(t % (N / 2)) * 2 * NC + (t / (N / 2)) * NC
                                               : Xw + (size_t)t * NC;
          float4 v = *reinterpret_cast<const float4*>(src + cq);
          if (XF == XF_LNPE) v = f4add(f4scale(v, sq), *reinterpret_cast<const float4*>(pe + t * NC + cq));
          float4 d; float rstd;
          ln_stats<LPR>(v, d, rstd);
          *reinterpret_cast<float4*>(Xs + row * LDX + cq) = f4add(f4mul(f4scale(d, rstd), gam), bet);
        }
      }
      __syncthreads();
      // ---- accumulate over the chunk's tokens ----
      for (int tb = 0; tb < TC; tb += 16) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int t = tb + 4 * g + s;
#pragma unroll
          for (int i = 0; i < TPW; ++i) {
            if (valid[i]) {
              int cm = mo[i] + r, cn = no[i] + r;
              if constexpr (M % 16 != 0) cm = cm < M ? cm : M - 1;   // half tiles (C = 8): stay inside the tile
              if constexpr (NC % 16 != 0) cn = cn < NC ? cn : NC - 1;
              const float a = Ys[xoff<LAYY>(ldy, t, cm)];
              const float b = Xs[xoff<LAYX>(ldx, t, cn)];
              acc[i] = mfma4(a, b, acc[i]);
            }
          }
        }
      }
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    if (!valid[i]) continue;
    const int n = no[i] + r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = mo[i] + 4 * g + j;
      if (m < M && n < NC) atomicAdd(dW + (size_t)m * NC + n, acc[i][j]);
    }
  }
}

// ---------------------------------------------------------------------------------
size_t dw_lds(int M, int NC, bool yhm, bool xhm, int TC, int N) {
  return ((size_t)TC * (yhm ? M : ld_of(M)) + (size_t)TC * (xhm ? NC : ld_of(NC)) + N + 2 + 4) * sizeof(float);
}

// largest token chunk (multiple of 16 dividing N) whose staging fits the LDS budget
int dw_chunk(int M, int NC, bool yhm, bool xhm, int N, size_t budget) {
  int tc = N;
  while (tc > 16 && (dw_lds(M, NC, yhm, xhm, tc, N) > budget || N % tc != 0)) tc -= 16;
  return tc;
}

template <int M, int NC, int LAYY, int XF>
static void launch_dw_t(const float* Y, const float* X, const float* pe, const float* lnw, const float* lnb,
                        const float* le, float* dW, int N, int B, int ksplit, hipStream_t s) {
  const bool yhm = LAYY == LAY_HM, xhm = XF == XF_HM;
  const int TC = dw_chunk(M, NC, yhm, xhm, N, 96 * 1024);
  const size_t lds = dw_lds(M, NC, yhm, xhm, TC, N);
  RAL_SET_LDS((k_dw<M, NC, LAYY, XF>), lds);
  const int grid = B < ksplit ? B : ksplit;
  k_dw<M, NC, LAYY, XF><<<grid, 512, lds, s>>>(Y, X, pe, lnw, lnb, le, dW, N, TC, B);
}

template <int C>
static void launch_block_dw_c(const float* dx2, const float* upre, const float* dupre, const float* x1,
                              const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                              const BlockP& w, const BlockP& gr, int N, int B, int ks, hipStream_t s) {
  launch_dw_t<C, 4 * C, LAY_TOK, XF_A2>(dx2, upre, nullptr, nullptr, nullptr, w.le, gr.w2, N, B, ks, s);
  launch_dw_t<4 * C, C, LAY_TOK, XF_LN>(dupre, x1, nullptr, w.ln2w, w.ln2b, nullptr, gr.w1, N, B, ks, s);
  launch_dw_t<C, C, LAY_TOK, XF_HM>(dx1, o_hm, nullptr, nullptr, nullptr, nullptr, gr.wp, N, B, ks, s);
  launch_dw_t<3 * C, C, LAY_HM, XF_LNPE>(dqkv, x, pe, w.ln1w, w.ln1b, nullptr, gr.wqkv, N, B, ks, s);
}

void launch_block_dw(int C, const float* dx2, const float* upre, const float* dupre, const float* x1,
                     const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                     const BlockP& w, const BlockP& gr, int N, int B, int ksplit, hipStream_t s) {
  switch (C) {
#define CASE(c) case c: launch_block_dw_c<c>(dx2, upre, dupre, x1, dx1, o_hm, dqkv, x, pe, w, gr, N, B, ksplit, s); break;
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
}

void launch_resample_dw(int D, bool sep, const float* dy, const float* x, const float* lnw, const float* lnb,
                        float* dW, int T, int B, int ksplit, hipStream_t s) {
#define CASE(d) case d: if (sep) launch_dw_t<d, d, LAY_TOK, XF_LN_SEP>(dy, x, nullptr, lnw, lnb, nullptr, dW, T, B, ksplit, s); \
                        else launch_dw_t<d, d, LAY_TOK, XF_LN>(dy, x, nullptr, lnw, lnb, nullptr, dW, T, B, ksplit, s); break;
  switch (D) { CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) }
#undef CASE
}
