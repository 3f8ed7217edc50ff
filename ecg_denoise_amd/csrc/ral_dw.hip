// Weight-gradient products of the Linear layers:  dW[m][n] += sum_tokens Y[t][m] * X[t][n].
//
// Both operands are staged per token chunk in LDS; the X operand is RE-COMPUTED from the tensors the
// forward pass saved (LayerNorm, the GELU / local-enhancement chain) rather than stored.  A workgroup
// owns one (M-slice x N-slice) block of dW for the windows blockIdx.x, blockIdx.x + gridDim.x, ...;
// its waves form a WM x WN grid and each wave keeps an MI x NI block of 16x16 tiles in MFMA
// accumulators across ALL those windows (MI + NI LDS fragment reads feed MI*NI MFMAs), adding them to
// the gradient buffer once at the end (fp32 global atomics).  Slices keep the LDS footprint small
// enough for several workgroups per CU, so one workgroup's staging overlaps another's MFMAs.
#include "ral_device.hpp"
#include "ral_kernels.hpp"

enum { XF_HM = 0, XF_LN = 1, XF_LNPE = 2, XF_A2 = 3, XF_LN_SEP = 4 };

template <int T> struct WaveGrid {   // T = 16x16 tiles along a dimension
  static constexpr int m = (T % 4 == 0) ? 4 : (T % 3 == 0 ? 3 : (T % 2 == 0 ? 2 : 1));
};
template <int TM, int TN> struct WaveGrid2 {
  static constexpr int WM = WaveGrid<TM>::m;
  static constexpr int cap = 8 / WM;
  static constexpr int WN = (TN % 4 == 0 && cap >= 4) ? 4 : ((TN % 2 == 0 && cap >= 2) ? 2 : 1);
};

template <int M, int NC, int MS, int NS, int LAYY, int XF>
__global__ __launch_bounds__(512) void k_dw(const float* __restrict__ Y, const float* __restrict__ X,
                                            const float* __restrict__ pe, const float* __restrict__ lnw,
                                            const float* __restrict__ lnb, const float* __restrict__ le,
                                            float* __restrict__ dW, int N, int TC, int B) {
  extern __shared__ float4 smem4[];
  constexpr int TM = (MS + 15) / 16, TN = (NS + 15) / 16;
  constexpr int WM = WaveGrid2<TM, TN>::WM, WN = WaveGrid2<TM, TN>::WN, MI = TM / WM, NI = TN / WN;
  constexpr int LAYX = (XF == XF_HM) ? LAY_HM : LAY_TOK;
  constexpr int LDY = LDof<MS>::v, LDX = LDof<NS>::v;
  constexpr int NSL_N = NC / NS;
  float* Ys = reinterpret_cast<float*>(smem4);
  float* Xs = Ys + (LAYY == LAY_HM ? TC * MS : TC * LDY);
  float* A0 = Xs + (LAYX == LAY_HM ? TC * NS : TC * LDX);  // N + 2 (XF_A2 with LE only)
  const int ldy = (LAYY == LAY_HM) ? TC : LDY, ldx = (LAYX == LAY_HM) ? TC : LDX;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const int mb = (blockIdx.y / NSL_N) * MS, nb = (blockIdx.y % NSL_N) * NS;   // this workgroup's slice of dW
  // narrow slices own fewer than 8 wave blocks: the spare waves take every KW-th 16-token step of the chunk and
  // their partial sums are folded through LDS at the end (the global atomics stay one per element per workgroup)
  constexpr int NWB = WM * WN, KW = 8 / NWB;
  const int wb = wave % NWB, kw = wave / NWB;
  const int m0 = (wb / WN) * MI * 16, n0 = (wb % WN) * NI * 16;               // this wave's block inside the slice
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float lw0 = 0.f, lw1 = 0.f, lw2 = 0.f;
  const bool use_le = (XF == XF_A2) && le != nullptr;
  if (use_le) { lw0 = le[0]; lw1 = le[1]; lw2 = le[2]; }

  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* Yw = Y + (size_t)win * N * M;
    const float* Xw = X + (size_t)win * N * NC;
    if (use_le && nb == 0) {  // gelu(u[:,0]) of the whole window, zero halo (only the slice holding channel 0)
      for (int i = threadIdx.x; i < N + 2; i += blockDim.x)
        A0[i] = (i == 0 || i == N + 1) ? 0.f : gelu_f(Xw[(size_t)(i - 1) * NC]);
    }
    for (int t0 = 0; t0 < N; t0 += TC) {
      // ---- stage the Y slice ----
      if constexpr (LAYY == LAY_HM) {
        for_each_row_f4<4>(Yw + ((size_t)(mb / 4) * N + t0) * 4, N * 4, MS / 4, TC * 4, [&](int q, int c, float4 v) {
          reinterpret_cast<float4*>(Ys)[q * TC + (c >> 2)] = v;
        });
      } else {
        copy_in(Ys, LDY, Yw + (size_t)t0 * M + mb, M, TC, MS);
      }
      // ---- stage / re-compute the X slice ----
      if constexpr (XF == XF_HM) {
        for_each_row_f4<4>(Xw + ((size_t)(nb / 4) * N + t0) * 4, N * 4, NS / 4, TC * 4, [&](int q, int c, float4 v) {
          reinterpret_cast<float4*>(Xs)[q * TC + (c >> 2)] = v;
        });
      } else if constexpr (XF == XF_A2) {
        if (use_le && nb == 0) __syncthreads();  // A0 ready
        for_each_row_f4<4>(Xw + (size_t)t0 * NC + nb, NC, TC, NS, [&](int row, int c4, float4 a) {
          const int q = c4 >> 2;
          a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w);
          if (use_le) {
            if (nb == 0 && q == 0) a.x = lw0 * A0[t0 + row] + lw1 * A0[t0 + row + 1] + lw2 * A0[t0 + row + 2];
            a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w);
          }
          *reinterpret_cast<float4*>(Xs + row * LDX + 4 * q) = a;
        });
      } else {  // LayerNorm family (X is never sliced: NS == NC)
        static_assert(XF == XF_HM || XF == XF_A2 || NS == NC, "LayerNorm operand must keep whole rows");
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
      if (kw < KW) {
        for (int tb = kw * 16; tb < TC; tb += 16 * KW) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const int t = tb + 4 * g + s;
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              int cm = m0 + 16 * i + r;
              if constexpr (MS % 16 != 0) cm = cm < MS ? cm : MS - 1;   // half tiles (C = 8): stay inside the tile
              a[i] = Ys[xoff<LAYY>(ldy, t, cm)];
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              int cn = n0 + 16 * j + r;
              if constexpr (NS % 16 != 0) cn = cn < NS ? cn : NS - 1;
              b[j] = Xs[xoff<LAYX>(ldx, t, cn)];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
          }
        }
      }
      __syncthreads();
    }
  }
  if constexpr (KW > 1) {   // fold the K-split partials: wave (wb, kw > 0) -> LDS -> wave (wb, 0)
    float* red = reinterpret_cast<float*>(smem4);   // staging buffers are free now (last barrier of the loop)
    if (kw > 0 && kw < KW) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          *reinterpret_cast<f32x4*>(red + ((((kw - 1) * NWB + wb) * MI + i) * NI + j) * 256 + lane * 4) = acc[i][j];
    }
    __syncthreads();
    if (kw == 0) {
      for (int k = 1; k < KW; ++k)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((((k - 1) * NWB + wb) * MI + i) * NI + j) * 256 + lane * 4);
            acc[i][j] += v;
          }
    }
  }
  if (kw == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + 16 * j + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + 16 * i + 4 * g + q;
          if (m < MS && n < NS) atomicAdd(dW + (size_t)(mb + m) * NC + nb + n, acc[i][j][q]);
        }
      }
  }
}

// ---------------------------------------------------------------------------------
static size_t dw_lds(int MS, int NS, bool yhm, bool xhm, int TC, int N) {
  return ((size_t)TC * (yhm ? MS : ld_of(MS)) + (size_t)TC * (xhm ? NS : ld_of(NS)) + N + 2 + 4) * sizeof(float);
}

// largest token chunk (multiple of 16 dividing N) whose staging fits the LDS budget
static int dw_chunk(int MS, int NS, bool yhm, bool xhm, int N, size_t budget) {
  int tc = N;
  while (tc > 16 && (dw_lds(MS, NS, yhm, xhm, tc, N) > budget || N % tc != 0)) tc -= 16;
  return tc;
}

static size_t g_dw_budget = 50 * 1024;
void set_dw_lds_budget(size_t bytes) { g_dw_budget = bytes; }

template <int M, int NC, int MS, int NS, int LAYY, int XF>
static void launch_dw_t(const float* Y, const float* X, const float* pe, const float* lnw, const float* lnb,
                        const float* le, float* dW, int N, int B, int ksplit, hipStream_t s) {
  constexpr int TM = (MS + 15) / 16, TN = (NS + 15) / 16;
  constexpr int WM = WaveGrid2<TM, TN>::WM, WN = WaveGrid2<TM, TN>::WN;
  static_assert(TM % WM == 0 && TN % WN == 0 && WM * WN <= 8, "wave grid must tile the slice");
  static_assert(M % MS == 0 && NC % NS == 0, "slices must tile dW");
  const bool yhm = LAYY == LAY_HM, xhm = XF == XF_HM;
  const int TC = dw_chunk(MS, NS, yhm, xhm, N, g_dw_budget);
  constexpr int KW = 8 / (WM * WN);
  const size_t fold = (size_t)(KW - 1) * TM * TN * 256 * sizeof(float);   // K-split partials of the spare waves
  const size_t stage = dw_lds(MS, NS, yhm, xhm, TC, N);
  const size_t lds = stage > fold ? stage : fold;
  RAL_SET_LDS((k_dw<M, NC, MS, NS, LAYY, XF>), lds);
  // ksplit is the split-K count of a fully sliced product; products with fewer slices get proportionally more
  // split-K workgroups so that every launch still fills the chip (at least ~256 workgroups)
  constexpr int nsl = (M / MS) * (NC / NS);
  int ks = ksplit;
  if (ks * nsl < 256) ks = (256 + nsl - 1) / nsl;
  dim3 grid(B < ks ? B : ks, nsl);
  // staging (LayerNorm / GELU re-computation) uses all 8 waves even when only WM*WN of them own MFMA tiles
  constexpr int threads = 512;
  k_dw<M, NC, MS, NS, LAYY, XF><<<grid, threads, lds, s>>>(Y, X, pe, lnw, lnb, le, dW, N, TC, B);
}

// slice widths: at most 256 rows/columns of the wide operand per workgroup
template <int W> struct SliceOf { static constexpr int v = W > 128 ? ((W % 128 == 0) ? 128 : W / 2) : W; };

template <int C>
static void launch_block_dw_c(const float* dx2, const float* upre, const float* dupre, const float* x1,
                              const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                              const BlockP& w, const BlockP& gr, int N, int B, int ks, hipStream_t s) {
  launch_dw_t<C, 4 * C, C, SliceOf<4 * C>::v, LAY_TOK, XF_A2>(dx2, upre, nullptr, nullptr, nullptr, w.le, gr.w2, N, B, ks, s);
  launch_dw_t<4 * C, C, SliceOf<4 * C>::v, C, LAY_TOK, XF_LN>(dupre, x1, nullptr, w.ln2w, w.ln2b, nullptr, gr.w1, N, B, ks, s);
  launch_dw_t<C, C, C, C, LAY_TOK, XF_HM>(dx1, o_hm, nullptr, nullptr, nullptr, nullptr, gr.wp, N, B, ks, s);
  launch_dw_t<3 * C, C, SliceOf<3 * C>::v, C, LAY_HM, XF_LNPE>(dqkv, x, pe, w.ln1w, w.ln1b, nullptr, gr.wqkv, N, B, ks, s);
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
#define CASE(d) case d: if (sep) launch_dw_t<d, d, d, d, LAY_TOK, XF_LN_SEP>(dy, x, nullptr, lnw, lnb, nullptr, dW, T, B, ksplit, s); \
                        else launch_dw_t<d, d, d, d, LAY_TOK, XF_LN>(dy, x, nullptr, lnw, lnb, nullptr, dW, T, B, ksplit, s); break;
  switch (D) { CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) }
#undef CASE
}
