// Weight-gradient products of the Linear layers:  dW[m][n] += sum_tokens Y[t][m] * X[t][n].
//
// A workgroup owns one (M-slice x N-slice) block of dW for the windows blockIdx.x, blockIdx.x + gridDim.x, ...
// and walks their tokens in chunks of TC.  Its 8 waves are split by role:
//   waves 4..7 (producers) stage the X operand of chunk i+1 into one LDS buffer -- every global load of the chunk is issued
//     before the first one is consumed, and X is RE-COMPUTED from what the forward pass saved (LayerNorm, the GELU /
//     local-enhancement chain) rather than stored;
//   waves 0..3 (consumers) stage the Y operand of chunk i+1 (a gradient: a copy, or scale-and-split) and then run the MFMAs
//     of chunk i out of the other buffer: they form a WM x WN grid, each wave keeping an MI x NI block of 16x16 tiles in
//     accumulators across ALL the windows (MI + NI LDS fragment reads feed MI*NI MFMAs).
// One barrier per chunk swaps the buffers, so HBM latency, the re-computation and the MFMAs overlap inside a
// workgroup; one to three workgroups fit a CU.  The accumulators are added to the gradient buffer once at the
// end (fp32 global atomics; a same-address atomic chain costs ~30 ns per link on MI355X, which is what bounds
// the split-K count of the narrow levels).
#ifdef RAL_STAMP_TU_DW
#define RAL_STAMP_HERE
#endif
#include "ral_device.hpp"
#include "ral_kernels.hpp"

enum { XF_HM = 0, XF_LN = 1, XF_LNPE = 2, XF_A2 = 3, XF_LN_SEP = 4 };

// consumer wave grid over TM x TN tiles (at most 4 waves)
template <int TM, int TN> struct WaveGrid4 {
  static constexpr bool sq = (TM % 2 == 0 && TN % 2 == 0);
  static constexpr int WM = sq ? 2 : (TM % 4 == 0 ? 4 : (TM % 3 == 0 ? 3 : (TM % 2 == 0 ? 2 : 1)));
  static constexpr int WN = sq ? 2 : (WM > 1 ? 1 : (TN % 4 == 0 ? 4 : (TN % 3 == 0 ? 3 : (TN % 2 == 0 ? 2 : 1))));
};
// LDS row stride of a token-major tile: = 16 (mod 32) floats, so that the four k-groups of an MFMA fragment
// read (row t0 + g, column c0 + r) hit disjoint banks
constexpr int dw_ld(int w) { int l = 16; while (l < w) l += 32; return l; }
constexpr int dw_tile_floats(int w, bool hm, int tc) { return hm ? (w / 4) * (tc + 1) * 4 : tc * dw_ld(w); }
// widest slice of the sliced operand, and the workgroup count below which a launch gets more split-K workgroups
#ifndef RAL_DW_SLICE
#define RAL_DW_SLICE 128
#endif
// Workgroups of a weight-gradient launch.  These kernels run on side streams UNDER the data-gradient chain, and every
// workgroup ends with its slice's worth of global atomics: measured at batch 2048, launches of >= 256 workgroups give the
// fastest weight-gradient kernels on an empty GPU (3.65 ms per step serialised, 3.74 with 192, 5.0 with 128) but the slower
// training step - they take the CUs and the atomic throughput the chain needs.  Training step by this constant (same
// box, ms): 96: 17.48, 128: 17.37, 160: 17.23, 192: 17.21, 224: 17.31, 256: 17.63.  (Before the chain kernels of
// the narrow levels got faster the optimum was 128.)
#ifndef RAL_DW_MINWG
#define RAL_DW_MINWG 192
#endif
// (LDS of a workgroup's two staging buffers; smaller chunks co-reside more easily with the chain kernels' workgroups but
// were measured slower on the step: RAL_DW_LDS = 40 000: 17.74 ms, 24 000: 18.01 ms against 17.04 at the full 76 KB)
#ifndef RAL_DW_LDS_BYTES
#define RAL_DW_LDS_BYTES (76 * 1024)
#endif
// largest power-of-two chunk whose two buffers fit the LDS budget and whose staging registers (all in flight at
// once) fit next to the accumulators: at most 12 float4 per producer thread, 6 when a wave owns 16 tiles
constexpr int dw_tcmax(int MS, int NS, bool yhm, bool xhm, bool lnpe) {
  const int TM = (MS + 15) / 16, TN = (NS + 15) / 16;
  const int tiles = TM * TN >= 4 ? TM * TN / 4 : 1;
  const int cap = 12;   // (the accumulators are not live in the producer path)
  int tc = 128;
  while (tc > 16 && (2 * (dw_tile_floats(MS, yhm, tc) + dw_tile_floats(NS, xhm, tc)) * 4 > RAL_DW_LDS_BYTES ||
                     tc * (MS + NS * (lnpe ? 2 : 1)) / 4 > cap * 256)) tc /= 2;
  return tc;
}

struct XLoad { float4 v, p; float c; };   // one staged X element: value, positional encoding (LNPE), LE channel (A2)
template <int U, int UMAX, class LD, class ST, class IN>
RAL_DEV void issue_then_store(LD ld, ST st, IN inner) {
  if constexpr (U == UMAX) {
    inner();
  } else {
    const auto v = ld(U);
    issue_then_store<U + 1, UMAX>(ld, st, inner);
    st(U, v);
  }
}

// H (wide levels, behind the split data-gradient kernels): both operands staged as fp16-pair planes - Y (a gradient) times
// the power of two that puts the launch's largest |Y| (ymax, published by k_mlp_bwd_h / k_qkv_bwd_h) into [2^13, 2^14), X
// (LayerNorm / GELU / attention outputs) times the block's power of two for that activation (`xscale`: its bound into [2^13, 2^14), ASC_*), residuals unscaled (f16_split2u) -, token-major rows; the consumers
// fetch their fragments with the transposing LDS read (ds_read_b64_tr_b16: per 16 lanes a block of 4 tokens x 16 channels,
// delivered channel-major - the contraction index of these products is the token) and issue three
// v_mfma_f32_16x16x32_f16 per 32 tokens and tile into the one accumulator, which is unscaled once, at the flush.
template <int M, int NC, int MS, int NS, int LAYY, int XF, bool H = false>
// (Y, X, pe and a2c0 are deliberately NOT __restrict__: loads the compiler can prove invariant are sunk across the
// compiler barrier of the staging code, next to their stores, which costs one HBM round trip per load)
#ifndef RAL_DW_WPE
#define RAL_DW_WPE 2      // workgroups per CU the register budget is sized for (1: 256 registers, for RAL_DW_SLICE = 256 experiments)
#endif
__global__ __launch_bounds__(512, RAL_DW_WPE) void k_dw(const float* Y, const float* X, const float* pe,
                                               const float* __restrict__ lnw, const float* __restrict__ lnb,
                                               const float* a2c0, float* dW, float* dB, const unsigned* __restrict__ ymax,
                                               const float* __restrict__ xscale /* {scale, inverse} of the X operand (ASC_*), or nullptr: 2^8 */,
                                               int N, int TC, int B, int NV = 0 /* XF_LN_SEP: existing tokens of the N slots (0: all) */) {
  extern __shared__ float4 smem4[];
  constexpr bool YHM = LAYY == LAY_HM, XHM = XF == XF_HM;
  constexpr int LDYH = MS + 8, LDXH = NS + 8;   // H: row strides of the token-major planes (2-byte elements)
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const float ysc = H ? h2_row_scale(ymax[0]) : 1.0f;
  // the activation operand times the block's power of two for that activation (its bound into [2^13, 2^14): ral_device.hpp, ASC_*)
  const float XSC = (H && xscale) ? __builtin_nontemporal_load(xscale) : 256.0f;
  const float XSCI = (H && xscale) ? __builtin_nontemporal_load(xscale + 1) : 1.0f / 256.0f;
  constexpr int TM = (MS + 15) / 16, TN = (NS + 15) / 16;
  constexpr int WM = WaveGrid4<TM, TN>::WM, WN = WaveGrid4<TM, TN>::WN, MI = TM / WM, NI = TN / WN;
  constexpr int NWB = WM * WN, KW = 4 / NWB;      // spare consumer waves split the chunk's tokens (folded at the end)
  constexpr int LDY = dw_ld(MS), LDX = dw_ld(NS);
  constexpr int TCMAX = dw_tcmax(MS, NS, YHM, XHM, XF == XF_LNPE);
  constexpr int NPROD = 256;
  constexpr int UY = (TCMAX * MS / 4 + NPROD - 1) / NPROD, UX = (TCMAX * NS / 4 + NPROD - 1) / NPROD;
  static_assert(TM % WM == 0 && TN % WN == 0 && NWB <= 4, "consumer wave grid must tile the slice");
  static_assert(XHM || XF == XF_A2 || NS == NC, "LayerNorm operand must keep whole rows");
  static_assert(MS % 4 == 0 && NS % 4 == 0, "float4 staging");
  // (H: a tile is two planes of TC x LD 2-byte elements = TC x LD floats)
  const int ysz = H ? TC * LDYH : (YHM ? (MS / 4) * (TC + 1) * 4 : TC * LDY), xsz = H ? TC * LDXH : (XHM ? (NS / 4) * (TC + 1) * 4 : TC * LDX);
  float* const buf0 = reinterpret_cast<float*>(smem4);
  float* const buf1 = buf0 + ysz + xsz;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const bool producer = wave >= 4;
  const int mb = (blockIdx.y / (NC / NS)) * MS, nb = (blockIdx.y % (NC / NS)) * NS;   // this workgroup's slice of dW
  const int wb = (wave & 3) % NWB, kw = (wave & 3) / NWB;
  const int m0 = (wb / WN) * MI * 16, n0 = (wb % WN) * NI * 16;                       // this wave's block in the slice
  const int cpw = N / TC;
  const int nwin = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int nci = nwin * cpw;

  float4 gam = make_float4(0.f, 0.f, 0.f, 0.f), bet = gam;   // LayerNorm affine of this thread's column group
  if constexpr (XF == XF_LN || XF == XF_LNPE || XF == XF_LN_SEP) {
    const int c = (threadIdx.x % (NS / 4)) * 4;
    gam = *reinterpret_cast<const float4*>(lnw + c);
    bet = *reinterpret_cast<const float4*>(lnb + c);
  }
  // ---- producer: stage chunk ci into buf (256 threads, no barriers inside).  Straight-line code: indices past
  // the chunk are clamped rather than branched around (the duplicates rewrite identical values), because hipcc
  // waits for a conditional load on the spot and the whole point is to have every load of the chunk in flight.
  // the in-flight loads of one chunk (static indices only): the X operand - the one with the re-computation - is staged by the four
  // producer waves, the Y operand (a gradient: scale and split, or a plain copy) by the four consumer waves in front of their MFMAs.
  // (With both operands on the producers a chunk cost them ~1.3 us at C = 128 while the consumers' tiles take ~0.3 us: the
  // workgroup ran at the producers' pace with half of its waves idle three quarters of the time.)
  struct PackY { float4 y[UY]; };
  struct PackX { XLoad x[UX]; };
  auto with_chunk = [&](int tid, int ci, float* buf, auto&& fn) {
    const int win = blockIdx.x + (ci / cpw) * gridDim.x, t0 = (ci % cpw) * TC;
    const float* Yw = Y + (size_t)win * N * M;
    const float* Xw = X + (size_t)win * N * NC;
    const float* c0w = (XF == XF_A2 && a2c0 != nullptr) ? a2c0 + (size_t)win * N : Xw;   // any readable address if unused
    const bool le0 = XF == XF_A2 && a2c0 != nullptr && nb == 0;   // this slice holds the local-enhancement channel
    float* Ys = buf;
    float* Xs = buf + ysz;
    const int n4y = TC * (MS / 4), n4x = TC * (NS / 4);
    auto load_y = [&](int u) -> float4 {
      const int j = min(tid + u * NPROD, n4y - 1);
#ifdef RAL_DW_NOLOAD   // diagnostic: no HBM traffic
      return make_float4(1e-3f * j, 0.f, 1.f, 2.f);
#endif
      if constexpr (YHM) {
        const int q = j / TC, t = j - q * TC;
        return *reinterpret_cast<const float4*>(Yw + ((size_t)(mb / 4 + q) * N + t0 + t) * 4);
      } else {
        const int row = j / (MS / 4), c = (j - row * (MS / 4)) * 4;
        return *reinterpret_cast<const float4*>(Yw + (size_t)(t0 + row) * M + mb + c);
      }
    };
    auto store_y = [&](int u, float4 v) {
      const int j = min(tid + u * NPROD, n4y - 1);
      if constexpr (H) {
        int row, c;
        if constexpr (YHM) { const int q = j / TC; row = j - q * TC; c = q * 4; }
        else { row = j / (MS / 4); c = (j - row * (MS / 4)) * 4; }
        const H2 s0 = f16_split2u(v.x * ysc), s1 = f16_split2u(v.y * ysc), s2 = f16_split2u(v.z * ysc), s3 = f16_split2u(v.w * ysc);
        _Float16* Yh = reinterpret_cast<_Float16*>(Ys);
        *reinterpret_cast<f16x4*>(Yh + row * LDYH + c) = f16x4{s0.a, s1.a, s2.a, s3.a};
        *reinterpret_cast<f16x4*>(Yh + TC * LDYH + row * LDYH + c) = f16x4{s0.b, s1.b, s2.b, s3.b};
      } else if constexpr (YHM) {
        const int q = j / TC, t = j - q * TC;
        *reinterpret_cast<float4*>(Ys + (q * (TC + 1) + t) * 4) = v;
      } else {
        const int row = j / (MS / 4), c = (j - row * (MS / 4)) * 4;
        *reinterpret_cast<float4*>(Ys + row * LDY + c) = v;
      }
    };
    // X indices past the chunk wrap (n4x is a power of two) so that LayerNorm lane groups still hold whole rows
    auto load_x = [&](int u) -> XLoad {
      const int j = (tid + u * NPROD) & (n4x - 1);
      XLoad o;
      o.p = make_float4(0.f, 0.f, 0.f, 0.f); o.c = 0.f;
#ifdef RAL_DW_NOLOAD
      o.v = make_float4(1e-3f * j, 0.5f, 1.f, 2.f);
      return o;
#endif
      if constexpr (XHM) {
        const int q = j / TC, t = j - q * TC;
        o.v = *reinterpret_cast<const float4*>(Xw + ((size_t)(nb / 4 + q) * N + t0 + t) * 4);
      } else {
        const int row = j / (NS / 4), c = (j - row * (NS / 4)) * 4, t = t0 + row;
        const float* src = (XF == XF_LN_SEP) ? Xw + sep_src(t, N, NV ? NV : N, NC)
                                             : Xw + (size_t)t * NC + nb;
        o.v = *reinterpret_cast<const float4*>(src + c);
        if constexpr (XF == XF_LNPE) o.p = *reinterpret_cast<const float4*>(pe + (size_t)t * NC + c);
        if constexpr (XF == XF_A2) o.c = c0w[t];   // fc2 input of hidden channel 0 (LE variant), saved by k_mlp_bwd
      }
      return o;
    };
    auto put_xh = [&](int row, int c, float4 a) {
      const H2 s0 = f16_split2u(a.x * XSC), s1 = f16_split2u(a.y * XSC), s2 = f16_split2u(a.z * XSC), s3 = f16_split2u(a.w * XSC);
      _Float16* Xh = reinterpret_cast<_Float16*>(Xs);
      *reinterpret_cast<f16x4*>(Xh + row * LDXH + c) = f16x4{s0.a, s1.a, s2.a, s3.a};
      *reinterpret_cast<f16x4*>(Xh + TC * LDXH + row * LDXH + c) = f16x4{s0.b, s1.b, s2.b, s3.b};
    };
    auto store_x = [&](int u, XLoad o) {
      const int j = (tid + u * NPROD) & (n4x - 1);
      if constexpr (XHM) {
        const int q = j / TC, t = j - q * TC;
        if constexpr (H) put_xh(t, q * 4, o.v);
        else *reinterpret_cast<float4*>(Xs + (q * (TC + 1) + t) * 4) = o.v;
      } else {
        const int row = j / (NS / 4), c = (j - row * (NS / 4)) * 4;
        float4 a = o.v;
#ifdef RAL_DW_NOXF     // diagnostic: no LayerNorm / GELU re-computation
        if constexpr (false) {
#else
        if constexpr (XF == XF_A2) {
#endif
          a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w);
          if (a2c0 != nullptr) { a.x = gelu_f(a.x); a.y = gelu_f(a.y); a.z = gelu_f(a.z); a.w = gelu_f(a.w); }
          a.x = (le0 && c == 0) ? o.c : a.x;
        }
#ifdef RAL_DW_NOXF
        else if constexpr (false) {
#else
        else {  // LayerNorm family: a row is NC/4 consecutive lanes
#endif
          constexpr int LPR = NC / 4;
          if constexpr (XF == XF_LNPE) a = f4add(f4scale(a, sqrtf((float)NC)), o.p);
          float4 d; float rstd;
          ln_stats<LPR>(a, d, rstd);
          a = f4add(f4mul(f4scale(d, rstd), gam), bet);   // (column c is the same for every u: 256 % (NS/4) == 0)
        }
        if constexpr (H) put_xh(row, c, a);
        else *reinterpret_cast<float4*>(Xs + row * LDX + c) = a;
      }
    };
    fn(load_y, store_y, load_x, store_x);
  };
  // issue_*(): request every load of chunk ci (nothing waits on them here).  The compiler barrier pins the loads
  // above it -- otherwise they are sunk next to their stores, one HBM round trip each -- and the scheduling barrier
  // keeps the machine scheduler from undoing that.
  auto issue_x = [&](int ci, PackX& p) {
    with_chunk((int)threadIdx.x - 256, ci, buf0, [&](auto&, auto&, auto& load_x, auto&) {
#pragma unroll
      for (int u = 0; u < UX; ++u) p.x[u] = load_x(u);
    });
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto issue_y = [&](int ci, PackY& p) {
    with_chunk((int)threadIdx.x, ci, buf0, [&](auto& load_y, auto&, auto&, auto&) {
#pragma unroll
      for (int u = 0; u < UY; ++u) p.y[u] = load_y(u);
    });
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // commit_*(): wait for the chunk's loads, transform, store into the LDS buffer
  auto commit_x = [&](int ci, float* buf, const PackX& p) {
    with_chunk((int)threadIdx.x - 256, ci, buf, [&](auto&, auto&, auto&, auto& store_x) {
#pragma unroll
      for (int u = 0; u < UX; ++u) store_x(u, p.x[u]);
    });
  };
  auto commit_y = [&](int ci, float* buf, const PackY& p) {
    with_chunk((int)threadIdx.x, ci, buf, [&](auto&, auto& store_y, auto&, auto&) {
#pragma unroll
      for (int u = 0; u < UY; ++u) store_y(u, p.y[u]);
    });
  };

  // The two roles run separate loops with matching barrier counts (the role is wave-uniform), so that the
  // accumulators are not live in the producer path and its loads are not squeezed by register pressure.
  // The producer keeps the loads of the chunk after next in flight across the barrier: while the consumers work on
  // chunk ci it commits chunk ci+1 (requested one iteration ago, normally already there) and requests chunk ci+2,
  // so a chunk's HBM round trip overlaps a whole consumer phase instead of sitting in front of every barrier.
  if (producer) {
    if (nci > 0) {
      if constexpr (XF == XF_A2) {
        // GELU-heavy operand (fc2's input is re-computed from u_pre): two chunks in flight.  While the consumers work
        // on chunk ci the producer commits chunk ci + 1 (requested TWO iterations ago) and requests chunk ci + 3, so
        // a chunk's HBM round trip has a whole iteration to complete.  With one chunk in flight the load latency, the
        // transform and the consumers' MFMAs (which share the producer's SIMD) ADD UP here - measured with the
        // `RAL_DW_NOLOAD / NOXF / NOMFMA` variants of this kernel: 154 us = 55 + 53 + 45 at C = 128; this ordering
        // brings it to 132 us.  The LayerNorm operands did not gain (their packs are larger) and keep one chunk.
        PackX pa, pb;
        const int last = nci - 1;
        issue_x(0, pa);
        commit_x(0, buf0, pa);
        issue_x(last < 1 ? last : 1, pa);
        issue_x(last < 2 ? last : 2, pb);
        __syncthreads();
        // The loop body is a PAIR of chunks with no path around its second half: with `if (ci + 1 < nci) { second half }` inside
        // the loop, the wait-count pass merges, at the loop header, the state "pb requested after pa" with the state of the path
        // that skipped pb's request - and then makes the commit of pa wait for pb's loads as well (s_waitcnt vmcnt(6 .. 0) where
        // 18 .. 12 is right: one chunk in flight again, every other chunk).  An odd last chunk is peeled off below.
        int ci = 0;
        for (; ci + 1 < nci; ci += 2) {
          commit_x(ci + 1, buf1, pa);
          issue_x(ci + 3 < nci ? ci + 3 : last, pa);     // (past the end: a harmless re-read, no branch around the loads)
          __syncthreads();
          if (ci + 2 < nci) commit_x(ci + 2, buf0, pb);
          issue_x(ci + 4 < nci ? ci + 4 : last, pb);
          __syncthreads();
        }
        if (ci < nci) __syncthreads();                 // (odd count: the consumers' last chunk is staged already)
      } else {
        PackX p;
        issue_x(0, p);
        commit_x(0, buf0, p);
        issue_x(nci > 1 ? 1 : 0, p);
        __syncthreads();
        for (int ci = 0; ci < nci; ++ci) {
          if (ci + 1 < nci) commit_x(ci + 1, (ci & 1) ? buf0 : buf1, p);
          issue_x(ci + 2 < nci ? ci + 2 : nci - 1, p);   // (past the end: a harmless re-read, no branch around the loads)
          __syncthreads();
        }
      }
    } else {
      __syncthreads();
    }
    if constexpr (KW > 1) __syncthreads();
    return;
  }

  // ---- consumer: MFMAs over the staged chunks ----
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Bias gradient of the layer = column sums of Y over the tokens: the A fragments pass through registers anyway,
  // so the waves of the first column block (of the first N-slice) add them up on the side.
  const bool bias = dB != nullptr && nb == 0 && n0 == 0;
  float bsum[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) bsum[i] = 0.f;
  // Y operand: chunk ci + 1 is committed (requested one iteration ago) and chunk ci + 2 requested in front of the tiles of chunk ci
  PackY py;
  if (nci > 0) {
    issue_y(0, py);
    commit_y(0, buf0, py);
    issue_y(nci > 1 ? 1 : 0, py);
  }
  __syncthreads();
  for (int ci = 0; ci < nci; ++ci) {
    const float* Ys = (ci & 1) ? buf1 : buf0;
    const float* Xs = Ys + ysz;
    if (ci + 1 < nci) commit_y(ci + 1, (ci & 1) ? buf0 : buf1, py);
    issue_y(ci + 2 < nci ? ci + 2 : nci - 1, py);      // (past the end: a harmless re-read, no branch around the loads)
    if constexpr (H) {
      if (kw < KW) {
        const _Float16* Yh = reinterpret_cast<const _Float16*>(Ys);
        const _Float16* Xh = reinterpret_cast<const _Float16*>(Xs);
        const int l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;   // transposing read: lane 4 q + p of a 16-lane group addresses row q, columns 4 p ..
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        auto frag = [&](const _Float16* base, int ld, int t0, int c0) -> f16x8 {   // tokens t0 + 8 g .. + 7 of channel c0 + (lane & 15)
          const _Float16* a = base + (t0 + 8 * g + tq) * ld + c0 + 4 * tp;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 4 * ld));
          union { s16x4 s[2]; f16x8 h; } u;
          u.s[0] = lo; u.s[1] = hi;
          return u.h;
        };
        for (int tb = kw * 32; tb < TC; tb += 32 * KW) {
          f16x8 a1[MI], a2[MI];
#pragma unroll
          for (int i = 0; i < MI; ++i) { a1[i] = frag(Yh, LDYH, tb, m0 + 16 * i); a2[i] = frag(Yh + TC * LDYH, LDYH, tb, m0 + 16 * i); }
          if (bias) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
              float sm = 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) sm += (float)a1[i][e] + (float)a2[i][e];
              bsum[i] += sm;
            }
          }
          // (the X fragments one column tile at a time: all of them at once would cost the second workgroup per CU)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const f16x8 b1 = frag(Xh, LDXH, tb, n0 + 16 * j), b2 = frag(Xh + TC * LDXH, LDXH, tb, n0 + 16 * j);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#if defined(RAL_DW_NOMFMA)   // diagnostic: without the matrix work (fragments still read)
              acc[i][j][0] += (float)a1[i][0] + (float)a2[i][1] + (float)b1[2] + (float)b2[3];
              continue;
#endif
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[i], b1, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b2, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[i], b1, acc[i][j], 0, 0, 0);
            }
          }
        }
      }
    } else
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
            a[i] = YHM ? Ys[((cm >> 2) * (TC + 1) + t) * 4 + (cm & 3)] : Ys[t * LDY + cm];
            bsum[i] += a[i];
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            int cn = n0 + 16 * j + r;
            if constexpr (NS % 16 != 0) cn = cn < NS ? cn : NS - 1;
            b[j] = XHM ? Xs[((cn >> 2) * (TC + 1) + t) * 4 + (cn & 3)] : Xs[t * LDX + cn];
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
#if defined(RAL_DW_NOMFMA) && !defined(RAL_NOGEMM)   // diagnostic: without its matrix work (operands still read)
            for (int j = 0; j < NI; ++j) acc[i][j][0] += a[i] + b[j];
#else
            for (int j = 0; j < NI; ++j) acc[i][j] = mfma4(a[i], b[j], acc[i][j]);
#endif
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < MI; ++i) bsum[i] = rows_sum(bsum[i]);   // over the 4 k-groups: every lane (r, *) holds column r
  if constexpr (H) {   // out of the operand scales
    const float uy = h2_row_unscale(ymax[0]), ua = uy * XSCI;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      bsum[i] *= uy;
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] *= ua;
    }
  }
  if constexpr (KW > 1) {   // fold the K-split partials: wave (wb, kw > 0) -> LDS -> wave (wb, 0)
    float* red = buf0;      // staging buffers are free now (last barrier of the loop)
    float* redb = red + (KW - 1) * NWB * MI * NI * 256;
    if (kw > 0 && kw < KW) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        if (g == 0) redb[(((kw - 1) * NWB + wb) * MI + i) * 16 + r] = bsum[i];
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
          for (int j = 0; j < NI; ++j)
            acc[i][j] += *reinterpret_cast<const f32x4*>(red + ((((k - 1) * NWB + wb) * MI + i) * NI + j) * 256 + lane * 4);
      for (int k = 1; k < KW; ++k)
#pragma unroll
        for (int i = 0; i < MI; ++i) bsum[i] += redb[(((k - 1) * NWB + wb) * MI + i) * 16 + r];
    }
  }
  if (kw == 0 && bias && g == 0) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + 16 * i + r;
      if (m < MS) atomicAdd(dB + mb + m, bsum[i]);
    }
  }
  if (kw == 0) {
    if constexpr (NI == 4 && MS % 16 == 0 && NS % 64 == 0) {
      // Global float atomics run at the memory side at a fixed byte rate, and at full rate only for wave-instructions
      // that cover 256 contiguous bytes; a 16 x 16 accumulator register as it stands covers four 64-byte pieces of
      // four rows (4 x slower).  The four column tiles of a wave are therefore transposed against the four lane rows
      // first (two lane-swap steps): then one instruction adds 64 consecutive floats of one row of dW.
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4] = {acc[i][0][q], acc[i][1][q], acc[i][2][q], acc[i][3][q]};
          rows_transpose4(v);            // v[row]: row m0 + 16 i + 4 row + q, column n0 + 16 g + r
#pragma unroll
          for (int row = 0; row < 4; ++row)
            atomicAdd(dW + (size_t)(mb + m0 + 16 * i + 4 * row + q) * NC + nb + n0 + 16 * g + r, v[row]);
        }
    } else {
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
}

// ---------------------------------------------------------------------------------
static size_t g_dw_budget = RAL_DW_LDS_BYTES;
void set_dw_lds_budget(size_t bytes) { g_dw_budget = bytes < (size_t)RAL_DW_LDS_BYTES ? bytes : (size_t)RAL_DW_LDS_BYTES; }

template <int M, int NC, int MS, int NS, int LAYY, int XF>
static void launch_dw_t(const float* Y, const float* X, const float* pe, const float* lnw, const float* lnb,
                        const float* a2c0, float* dW, float* dB, int N, int B, int ksplit, hipStream_t s,
                        const unsigned* ymax = nullptr, const float* xscale = nullptr, int NV = 0) {
  static_assert(M % MS == 0 && NC % NS == 0, "slices must tile dW");
  constexpr bool yhm = LAYY == LAY_HM, xhm = XF == XF_HM;
  constexpr int TM = (MS + 15) / 16, TN = (NS + 15) / 16;
  constexpr int KW = 4 / (WaveGrid4<TM, TN>::WM * WaveGrid4<TM, TN>::WN);
  // ksplit is the split-K count of a fully sliced product; products with fewer slices get proportionally more
  // split-K workgroups so that every launch still fills the chip (at least ~256 workgroups)
  constexpr int nsl = (M / MS) * (NC / NS);
  int ks = ksplit;
  if (ks * nsl < RAL_DW_MINWG) ks = (RAL_DW_MINWG + nsl - 1) / nsl;
  dim3 grid(B < ks ? B : ks, nsl);
  const size_t fold = (size_t)(KW - 1) * (TM * TN * 256 + TM * 16) * sizeof(float);   // K-split partials of the spare waves
  if constexpr (MS % 16 == 0 && NS % 16 == 0 && MS >= 32 && NS >= 32) {
    if (ymax) {   // split operands (token chunks of at least one 32-token MFMA step)
      int TC = dw_tcmax(MS, NS, yhm, xhm, XF == XF_LNPE);
      auto bytesh = [&](int tc) { return (size_t)2 * tc * (MS + 8 + NS + 8) * sizeof(float); };
      while (TC > 32 && (N % TC != 0 || bytesh(TC) > g_dw_budget)) TC /= 2;
      if (TC >= 32 && N % TC == 0 && bytesh(TC) <= g_dw_budget + 4096) {
        const size_t lds = bytesh(TC) > fold ? bytesh(TC) : fold;
        RAL_SET_LDS((k_dw<M, NC, MS, NS, LAYY, XF, true>), lds);
        k_dw<M, NC, MS, NS, LAYY, XF, true><<<grid, 512, lds, s>>>(Y, X, pe, lnw, lnb, a2c0, dW, dB, ymax, xscale, N, TC, B, NV);
        return;
      }
    }
  }
  // largest power-of-two token chunk dividing N within the compile-time staging bound and the LDS budget
  int TC = dw_tcmax(MS, NS, yhm, xhm, XF == XF_LNPE);
  auto bytes = [&](int tc) { return (size_t)2 * (dw_tile_floats(MS, yhm, tc) + dw_tile_floats(NS, xhm, tc)) * sizeof(float); };
  while (TC > 16 && (N % TC != 0 || bytes(TC) > g_dw_budget)) TC /= 2;
  const size_t lds = bytes(TC) > fold ? bytes(TC) : fold;
  RAL_SET_LDS((k_dw<M, NC, MS, NS, LAYY, XF>), lds);
  k_dw<M, NC, MS, NS, LAYY, XF><<<grid, 512, lds, s>>>(Y, X, pe, lnw, lnb, a2c0, dW, dB, nullptr, nullptr, N, TC, B, NV);
}

// slice widths: at most RAL_DW_SLICE rows/columns of the wide operand per workgroup
template <int W> struct SliceOf { static constexpr int v = W > RAL_DW_SLICE ? ((W % RAL_DW_SLICE == 0) ? RAL_DW_SLICE : W / 2) : W; };

// gmax (or nullptr): bits of the largest |dx2|, |du|, |dx1|, |dqkv| of this launch's windows (published by the split
// data-gradient kernels): the four products then run on split operands (k_dw<..., true>)
template <int C>
static void launch_block_dw_c(const float* dx2, const float* upre, const float* a2c0, const float* dupre, const float* x1,
                              const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                              const BlockP& w, const BlockP& gr, int N, int B, int ks, bool skip_mlp, const unsigned* gmax, hipStream_t s, bool skip_qkv) {
  static const bool h_on = (ral_knob("DW_F16", 1) != 0);
  const unsigned* gm = (h_on && C >= 32) ? gmax : nullptr;
  if (!skip_mlp) {
  launch_dw_t<C, 4 * C, C, SliceOf<4 * C>::v, LAY_TOK, XF_A2>(dx2, upre, nullptr, nullptr, nullptr, a2c0, gr.w2, gr.b2, N, B, ks, s, gm ? gm + 0 : nullptr, w.asc ? w.asc + ASC_HID : nullptr);
  launch_dw_t<4 * C, C, SliceOf<4 * C>::v, C, LAY_TOK, XF_LN>(dupre, x1, nullptr, w.ln2w, w.ln2b, nullptr, gr.w1, gr.b1, N, B, ks, s, gm ? gm + 1 : nullptr, w.asc ? w.asc + ASC_LN2 : nullptr);
  }
  launch_dw_t<C, C, SliceOf<C>::v, C, LAY_TOK, XF_HM>(dx1, o_hm, nullptr, nullptr, nullptr, nullptr, gr.wp, gr.bp, N, B, ks, s, gm ? gm + 2 : nullptr, w.asc ? w.asc + ASC_O : nullptr);
  if (!skip_qkv)
  launch_dw_t<3 * C, C, SliceOf<3 * C>::v, C, LAY_HM, XF_LNPE>(dqkv, x, pe, w.ln1w, w.ln1b, nullptr, gr.wqkv, gr.bqkv, N, B, ks, s, gm ? gm + 3 : nullptr, w.asc ? w.asc + ASC_LN1 : nullptr);
}

void launch_block_dw(int C, const float* dx2, const float* upre, const float* a2c0, const float* dupre, const float* x1,
                     const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                     const BlockP& w, const BlockP& gr, int N, int B, int ksplit, bool skip_mlp, const unsigned* gmax, hipStream_t s, bool skip_qkv) {
  switch (C) {
#define CASE(c) case c: launch_block_dw_c<c>(dx2, upre, a2c0, dupre, x1, dx1, o_hm, dqkv, x, pe, w, gr, N, B, ksplit, skip_mlp, gmax, s, skip_qkv); break;
    CASE(8) CASE(16) CASE(32) CASE(64) CASE(128)
#undef CASE
  }
}

void launch_resample_dw(int D, bool sep, const float* dy, const float* x, const float* lnw, const float* lnb,
                        float* dW, int T, int Tv, int B, int ksplit, hipStream_t s) {
#define CASE(d) case d: if (sep) launch_dw_t<d, d, d, d, LAY_TOK, XF_LN_SEP>(dy, x, nullptr, lnw, lnb, nullptr, dW, nullptr, T, B, ksplit, s, nullptr, nullptr, Tv); \
                        else launch_dw_t<d, d, d, d, LAY_TOK, XF_LN>(dy, x, nullptr, lnw, lnb, nullptr, dW, nullptr, T, B, ksplit, s); break;
  switch (D) { CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) }
#undef CASE
}
