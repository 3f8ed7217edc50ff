// Attention backward of the SHORT windows (N = 32, 64, 128): wave-autonomous form.
//
// Reference op: MSAttention.forward's backward (model/raletransformer.py:291-322, model/transformer.py:289-323, R-wave bias
// model/transformer.py:508-558) - the same arithmetic as k_attn_bwd (ral_bwd.hip): P is recomputed from q, k and the saved
// log-sum-exp, S and dP are exact fp32 MFMA tiles with -lse / -delta as their C operands, sweep A accumulates dQ with a
// query block on the lanes, sweep B dK / dV with a key block on the lanes.
//
// What is different is who owns a head.  k_attn_bwd gives a (window, head group) item to a 256 / 512-thread workgroup:
// stage -> barrier -> sweeps -> barrier.  At N <= 128 an item's sweeps are as short as its staging pass (N = 64: 4.7 us of
// tiles behind a 43 KB load), the waves of a workgroup sit in the same phase, and with two workgroups per CU half the
// waves of a SIMD wait while the other half computes: 380 cycles per tile visit at N = 64 against 229 at N = 512 (r03).
// Here ONE WAVE owns a head (two heads at N = 32) from its global loads to its stores:
//   * its operands live in a private slice of the LDS (72 B per token: q log2 e, k, v, dO as 16-byte quads, -lse log2 e,
//     -delta), written and read by that wave only - no workgroup barrier anywhere in the task loop;
//   * the NEXT head's operands (q, k, v, dO, O, lse: 21 registers per token and lane) are requested before the current
//     head's sweeps and land under them; delta = rowsum(dO O) and the scalings are applied on the way into the LDS;
//   * the waves of a SIMD are at unrelated points of their tasks, so one wave's loads, LDS traffic and epilogue hide under
//     the others' tiles without any scheduling on our side;
//   * the R-wave table gradient is accumulated in an LDS copy for the whole life of the (persistent) workgroup and leaves
//     as ONE row of partials per workgroup (plain stores; k_attn_tpart_reduce adds the rows): the per-item flush of
//     k_attn_bwd was a 2048-link chain of same-address global atomics per table entry;
//   * a tile takes the biased path only when ITS 16 queries and ITS 16 keys meet the window (k_attn_bwd decides per
//     32-query block: at N = 64 half of all tile visits ran the table code for 16 of 4096 scores);
//   * the sums over the four lane groups at the end of a block are a 4 x 4 register / row transpose plus three adds (7
//     instructions per quad instead of 16) and every lane stores one float (256 contiguous bytes per wave-instruction).
#include "ral_device.hpp"
#include "ral_kernels.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x2 __attribute__((ext_vector_type(2)));
RAL_DEV f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#define RAL_LOG2E 1.4426950408889634f
#define RAL_LN2 0.6931471805599453f

#ifndef RAL_ATTNW_UNROLL
#define RAL_ATTNW_UNROLL 1   // key / query tiles per trip of the sweep loops
#endif
#ifndef RAL_ATTNW_WPE
#define RAL_ATTNW_WPE 3   // waves per SIMD the register budget is sized for (168 registers)
#endif

// sum of a per-lane quad over the four 16-lane rows of the wave: lane (r, g) ends with component g of the total
RAL_DEV float quad_rows_sum(float x, float y, float z, float w) {
  float v[4] = {x, y, z, w};
  rows_transpose4(v);
  return (v[0] + v[1]) + (v[2] + v[3]);
}

RAL_DEV float f4absmax(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// F16: the S and dP tiles on the f16 matrix cores.  A 16 x 16 tile of q k^T at head_dim 4 is a K = 4 product; with both
// operands as fp16 PAIRS (x = h1 + h2, unscaled residual) the four piece products of the four dims are exactly one
// v_mfma_f32_16x16x16_f16: lane group g' carries piece pair (g' >> 1, g' & 1) over the four dims, i.e. the A operand of a
// lane is plane (g >> 1) of its row's token and the B operand plane (g & 1) of its column's token - one 8-byte LDS read
// each, no packing instruction.  Beside the tile's vector work such an MFMA costs 10.6 cycles per SIMD slot against 41.7
// for v_mfma_f32_16x16x4_f32 (tools/diag/valu_probe.hip, kinds 23-26: 164.5 -> 102.2 cycles per sweep-A tile).
// Range: dO and v are multiplied by one power of two per task first (largest magnitude into [2^13, 2^14): dP is linear in
// both, the factor leaves with the results), q and k are not (S goes through the exponential); their pieces carry an
// ABSOLUTE error of 2^-25 below |x| = 2^-2, which is what matters for a score, and 2^-23 relative above.
template <int NT, bool TAB, bool F16, int QT>
__global__ __launch_bounds__(256, RAL_ATTNW_WPE) void k_attn_bwd_w(const float* qkv, const float* o_hm, const float* do_hm,
                                                                   const float* lse, const float* __restrict__ table,
                                                                   float* __restrict__ tpart, float* dqkv, int H, int Len_rt,
                                                                   int ntask) {
  constexpr int HW = NT >= 64 ? 1 : 64 / NT;   // heads per task
  constexpr int T = HW * NT, TPL = T / 64;     // tokens per task, tokens per lane
  constexpr int WSZ = T * (F16 ? 30 : 18);     // floats of LDS per wave
  constexpr int NB = NT / (16 * QT);           // query (key) blocks per head
  static_assert(NT % (16 * QT) == 0 && T % 64 == 0, "window length");
  constexpr int NBU = NB <= 2 ? NB : 1;        // unrolled blocks (a static store count for the wait, see the task loop)
  // Tiles are ROTATED by SH tokens when there is an R-wave table: tile x covers tokens (x + SH .. x + SH + 15) mod NT.  The
  // window is centred (off + Len / 2 = NT / 2, a multiple of 16), so unrotated it always straddles a tile boundary and
  // meets 2 x 2 tiles per head; rotated by 8 a window of up to 16 tokens sits inside ONE tile.  Attention sums over all
  // keys in any order, so the rotation only changes index arithmetic.
  constexpr int SH = TAB ? 8 : 0, MSK = NT - 1;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // fp32 tiles (quads per token): Q32 = q log2 e, K32, D32 = dO (and V32 without F16); pair planes (F16): Qp, Kp, Vp, Dp,
  // plane p of a tile at + p * 2 T floats, the four halves of token t at float index 2 t
  float* Q32 = sm + wave * WSZ;
  float* K32 = Q32 + 4 * T;
  float* D32 = K32 + 4 * T;
  float* V32 = D32 + 4 * T;              // !F16 only
  float* Qp = D32 + 4 * T;               // F16 only (same place as V32)
  float* Kp = Qp + 4 * T;
  float* Vp = Kp + 4 * T;
  float* Dp = Vp + 4 * T;
  float* Ls = F16 ? Dp + 4 * T : V32 + 4 * T;   // -lse * log2(e)   (negated: C operands of the S / dP tiles)
  float* Dl = Ls + T;                           // -rowsum(dO * O) (F16: times the task's scale)
  const int Len = TAB ? Len_rt : 0;
  const int ntab = TAB ? (2 * Len - 1) * H : 0;
  const int nwv = blockDim.x >> 6;   // waves per workgroup (the launcher sizes it so that the wave slices fill the LDS)
  float* tab = sm + nwv * WSZ;   // bias * log2(e), (2 Len - 1, H)
  float* dtab = tab + ntab;
  const int off = (NT - Len) >> 1;
  if constexpr (TAB) {
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) { tab[i] = table[i] * RAL_LOG2E; dtab[i] = 0.f; }
    __syncthreads();
  }
  // does the rotated range [x0, x0 + w) meet the window?  (wave-uniform; a window that reaches into the first SH tokens
  // wraps in rotated coordinates: every tile takes the table path then, the per-element test stays exact)
  auto meets = [&](int x0, int w) -> bool {
    return TAB && (off < SH || (x0 < off + Len - SH && x0 + w > off - SH));
  };
  const int xe0 = !TAB ? NT : (off < SH ? 0 : ((off - SH) & ~15));
  const int xe1 = !TAB ? NT : (off < SH ? NT : ((off + Len - SH + 15) & ~15));
  const int stride = gridDim.x * nwv;
  int task = blockIdx.x * nwv + wave;
  float inv = 1.f;               // F16: 1 / (scale of dO x scale of v) of the task in the LDS

  // operands of one task in flight (PREF: one token per lane): q, k, v, dO, O quads and lse
  float4 pq, pk, pv, pd, po;
  float pl;
  auto task_ptrs = [&](int tk, const float4*& gq, const float4*& gk, const float4*& gv, const float4*& gd, const float4*& go,
                       const float*& gl) {
    const int hh = tk * HW, win = hh / H, h0 = hh - win * H;
    gq = reinterpret_cast<const float4*>(qkv) + ((size_t)win * 3 * H + h0) * NT;
    gk = gq + (size_t)H * NT;
    gv = gk + (size_t)H * NT;
    const size_t hq = ((size_t)win * H + h0) * NT;
    gd = reinterpret_cast<const float4*>(do_hm) + hq;
    go = reinterpret_cast<const float4*>(o_hm) + hq;
    gl = lse + hq;
  };
  auto put = [&](int t, float4 q, float4 k, float4 v, float4 d, float4 o, float l, float cd, float cv, float cq, float ck) {
    const float4 ql = f4scale(q, RAL_LOG2E);
    reinterpret_cast<float4*>(Q32)[t] = ql;
    reinterpret_cast<float4*>(K32)[t] = k;
    reinterpret_cast<float4*>(D32)[t] = d;
    Ls[t] = -l * RAL_LOG2E;
    if constexpr (F16) {
      auto planes = [&](float* X, float4 x) {
        const H2x4 s2 = split4(x);
        *reinterpret_cast<h16x4*>(X + 2 * t) = s2.a;
        *reinterpret_cast<h16x4*>(X + 2 * T + 2 * t) = s2.b;
      };
      planes(Qp, f4scale(ql, cq)); planes(Kp, f4scale(k, ck)); planes(Vp, f4scale(v, cv)); planes(Dp, f4scale(d, cd));
      Dl[t] = -f4dot(d, o) * (cd * cv);
    } else {
      reinterpret_cast<float4*>(V32)[t] = v;
      Dl[t] = -f4dot(d, o);
    }
  };
  // the task's powers of two for dO and v (largest magnitude into [2^13, 2^14)); returns 1 / their product
  auto scales = [&](float md, float mv, float& cd, float& cv) -> float {
    const unsigned bd = __float_as_uint(group_max<64>(md)), bv = __float_as_uint(group_max<64>(mv));
    cd = h2_row_scale(bd); cv = h2_row_scale(bv);
    return h2_row_unscale(bd) * h2_row_unscale(bv);
  };
  auto request = [&](int tk) {
    const float4 *gq, *gk, *gv, *gd, *go; const float* gl;
    task_ptrs(tk, gq, gk, gv, gd, go, gl);
    pq = gq[lane]; pk = gk[lane]; pv = gv[lane]; pd = gd[lane]; po = go[lane]; pl = gl[lane];
  };
  auto deposit = [&]() {
    float cd = 1.f, cv = 1.f, cq = 1.f, ck = 1.f;
    if constexpr (F16) {
      inv = scales(f4absmax(pd), f4absmax(pv), cd, cv);
      pair_balance(group_max<64>(f4absmax(pq)) * RAL_LOG2E, group_max<64>(f4absmax(pk)), cq, ck);
    }
    put(lane, pq, pk, pv, pd, po, pl, cd, cv, cq, ck);
  };
  auto stage = [&](int tk) {   // request + deposit in one go (all loads of the task in flight together)
    const float4 *gq, *gk, *gv, *gd, *go; const float* gl;
    task_ptrs(tk, gq, gk, gv, gd, go, gl);
    float4 q[TPL], k[TPL], v[TPL], d[TPL], o[TPL]; float l[TPL];
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
      const int t = lane + 64 * u;
      q[u] = gq[t]; k[u] = gk[t]; v[u] = gv[t]; d[u] = gd[t]; o[u] = go[t]; l[u] = gl[t];
    }
    float cd = 1.f, cv = 1.f, cq = 1.f, ck = 1.f;
    if constexpr (F16) {
      float md = 0.f, mv = 0.f, mq = 0.f, mk = 0.f;
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        md = fmaxf(md, f4absmax(d[u])); mv = fmaxf(mv, f4absmax(v[u])); mq = fmaxf(mq, f4absmax(q[u])); mk = fmaxf(mk, f4absmax(k[u]));
      }
      inv = scales(md, mv, cd, cv);
      pair_balance(group_max<64>(mq) * RAL_LOG2E, group_max<64>(mk), cq, ck);
    }
#pragma unroll
    for (int u = 0; u < TPL; ++u) put(lane + 64 * u, q[u], k[u], v[u], d[u], o[u], l[u], cd, cv, cq, ck);
  };
  // MFMA operands of the token on this lane's row (A) / column (B) of a tile
  auto opA = [&](const float* X, int tok) -> h16x4 { return *reinterpret_cast<const h16x4*>(X + (g >> 1) * 2 * T + 2 * tok); };
  auto opB = [&](const float* X, int tok) -> h16x4 { return *reinterpret_cast<const h16x4*>(X + (g & 1) * 2 * T + 2 * tok); };
  auto mm = [&](h16x4 a, h16x4 b, f32x4 c) -> f32x4 { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); };
  // Order inside a trip: request the next task's operands, sweep (and store) the current one, THEN wait for the request
  // and move it into the LDS.  With the wait at the top of the next trip (across the back edge) the compiler could not
  // count the stores issued since and waited for them too - a store round trip exposed per task; here the loads, the
  // (statically counted) stores and the wait are straight-line code and the wait is s_waitcnt vmcnt(<stores after>).
  // Two tokens per lane (N = 128) would be 42 registers in flight, which the register allocator answers by waiting for
  // the loads at once and parking them in scratch; there a task is ~13 us of tiles behind a ~2 us load and the other
  // waves of the SIMD cover it: no prefetch, the operands are requested and deposited at the top of the trip.
  constexpr bool PREF = TPL == 1;
  if constexpr (PREF) { if (task < ntask) { request(task); deposit(); } }
  while (task < ntask) {
    const int next = task + stride;
    if constexpr (PREF) { request(next < ntask ? next : task); asm volatile("" ::: "memory"); }   // (the last task re-requests its own operands: no branch around the loads)
    else stage(task);
    const int hh = task * HW, win = hh / H, h0 = hh - win * H;
    float* dbase = dqkv + (size_t)win * 3 * H * NT * 4;
    const float oscale = F16 ? inv : 1.f;
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      const int head = h0 + hl, tb = hl * NT;   // tb: first token of the head in the task's tiles
      const float* Lh = Ls + tb; const float* Eh = Dl + tb;
      // ---------------- sweep A: dQ (query block on the lanes, loop over key tiles) ----------------
#pragma unroll NBU
      for (int qb = 0; qb < NB; ++qb) {
        const int q0 = qb * 16 * QT;
        const float4* K4 = reinterpret_cast<const float4*>(K32) + tb;
        float qf[QT], df[QT];
        h16x4 qh[QT], dh[QT];
        f32x4 lq[QT], dl[QT];
        f32x2 dq01[QT], dq23[QT];
        bool qin[QT];
        int qtok[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int q = (q0 + 16 * qt + SH + r) & MSK;
          qtok[qt] = q;
          if constexpr (F16) { qh[qt] = opB(Qp, tb + q); dh[qt] = opB(Dp, tb + q); }
          else { qf[qt] = Q32[(tb + q) * 4 + g]; df[qt] = D32[(tb + q) * 4 + g]; }
          const float l = Lh[q], d = Eh[q];
          lq[qt] = f32x4{l, l, l, l}; dl[qt] = f32x4{d, d, d, d};
          dq01[qt] = f32x2{0.f, 0.f}; dq23[qt] = f32x2{0.f, 0.f};
          qin[qt] = meets(q0 + 16 * qt, 16);
        }
        auto tileA = [&](int kt, auto biased) {
          const int kr = tb + ((kt + SH + r) & MSK), k4i = (kt + SH + 4 * g) & MSK;
          float kf, vf; h16x4 kh, vh;
          if constexpr (F16) { kh = opA(Kp, kr); vh = opA(Vp, kr); }
          else { kf = K32[kr * 4 + g]; vf = V32[kr * 4 + g]; }
          float4 k4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) k4[j] = K4[k4i + j];
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            f32x4 s, dp;
            if constexpr (F16) { s = mm(kh, qh[qt], lq[qt]); dp = mm(vh, dh[qt], dl[qt]); }
            else { s = mfma4(kf, qf[qt], lq[qt]); dp = mfma4(vf, df[qt], dl[qt]); }   // s - lse, dP - delta   [key 4g+j][query r]
            float ds[4];
            if (decltype(biased)::value && qin[qt]) {
              // (the four table reads are unconditional - clamped index - and issued together: one LDS round trip per tile)
              const int qi = qtok[qt] - off;
              const bool qok = (unsigned)qi < (unsigned)Len;
              const int rel0 = qi - (k4i - off) + Len - 1;
              int e[4]; bool in[4]; float b[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                in[j] = qok && (unsigned)(k4i + j - off) < (unsigned)Len;
                e[j] = min(max(rel0 - j, 0), 2 * Len - 2) * H + head;
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) b[j] = tab[e[j]];
#pragma unroll
              for (int j = 0; j < 4; ++j) ds[j] = __builtin_amdgcn_exp2f(s[j] + (in[j] ? b[j] : 0.f)) * dp[j];
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (in[j]) atomicAdd(dtab + e[j], ds[j] * oscale);
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) ds[j] = __builtin_amdgcn_exp2f(s[j]) * dp[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              dq01[qt] = pk_fma2(f32x2{ds[j], ds[j]}, f32x2{k4[j].x, k4[j].y}, dq01[qt]);
              dq23[qt] = pk_fma2(f32x2{ds[j], ds[j]}, f32x2{k4[j].z, k4[j].w}, dq23[qt]);
            }
          }
        };
        // key tiles that meet the R-wave window run the table code (only for the query tiles that meet it too); the
        // others stay in branch-free loops
        const bool anyq = meets(q0, 16 * QT);
        const int e0 = anyq ? xe0 : NT, e1 = anyq ? xe1 : NT;
#pragma unroll RAL_ATTNW_UNROLL
        for (int kt = 0; kt < e0; kt += 16) tileA(kt, std::false_type{});
        if constexpr (TAB) {
#pragma unroll 1
          for (int kt = e0; kt < e1; kt += 16) tileA(kt, std::true_type{});
#pragma unroll RAL_ATTNW_UNROLL
          for (int kt = e1; kt < NT; kt += 16) tileA(kt, std::false_type{});
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float v = quad_rows_sum(dq01[qt][0], dq01[qt][1], dq23[qt][0], dq23[qt][1]);
          dbase[((size_t)head * NT + qtok[qt]) * 4 + g] = (0.5f * oscale) * v;   // q = 0.5 (h Wq^T + b)
        }
      }
      // ---------------- sweep B: dK, dV (key block on the lanes, loop over query tiles) ----------------
#pragma unroll NBU
      for (int kb = 0; kb < NB; ++kb) {
        const int k0 = kb * 16 * QT;
        const float4* Q4 = reinterpret_cast<const float4*>(Q32) + tb;
        const float4* D4 = reinterpret_cast<const float4*>(D32) + tb;
        float kf[QT], vf[QT];
        h16x4 kh[QT], vh[QT];
        f32x2 dk01[QT], dk23[QT], dv01[QT], dv23[QT];
        bool kin[QT];
        int ktok[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const int k = (k0 + 16 * t + SH + r) & MSK;
          ktok[t] = k;
          if constexpr (F16) { kh[t] = opB(Kp, tb + k); vh[t] = opB(Vp, tb + k); }
          else { kf[t] = K32[(tb + k) * 4 + g]; vf[t] = V32[(tb + k) * 4 + g]; }
          dk01[t] = f32x2{0.f, 0.f}; dk23[t] = f32x2{0.f, 0.f}; dv01[t] = f32x2{0.f, 0.f}; dv23[t] = f32x2{0.f, 0.f};
          kin[t] = meets(k0 + 16 * t, 16);
        }
        auto tileB = [&](int qt, auto biased) {
          const int qr = tb + ((qt + SH + r) & MSK), q4i = (qt + SH + 4 * g) & MSK;
          float qa, da; h16x4 qah, dah;
          if constexpr (F16) { qah = opA(Qp, qr); dah = opA(Dp, qr); }
          else { qa = Q32[qr * 4 + g]; da = D32[qr * 4 + g]; }
          const float4 l4 = *reinterpret_cast<const float4*>(Lh + q4i);
          const float4 d4 = *reinterpret_cast<const float4*>(Eh + q4i);
          float4 q4[4], o4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { q4[j] = Q4[q4i + j]; o4[j] = D4[q4i + j]; }
#pragma unroll
          for (int t = 0; t < QT; ++t) {
            f32x4 s, dp;
            if constexpr (F16) { s = mm(qah, kh[t], f32x4{l4.x, l4.y, l4.z, l4.w}); dp = mm(dah, vh[t], f32x4{d4.x, d4.y, d4.z, d4.w}); }
            else { s = mfma4(qa, kf[t], f32x4{l4.x, l4.y, l4.z, l4.w}); dp = mfma4(da, vf[t], f32x4{d4.x, d4.y, d4.z, d4.w}); }   // [query 4g+j][key r]
            if (decltype(biased)::value && kin[t]) {
              const int ki = ktok[t] - off;
              const bool kok = (unsigned)ki < (unsigned)Len;
              const int rel0 = (q4i - off) - ki + Len - 1;
              float b[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) b[j] = tab[min(max(rel0 + j, 0), 2 * Len - 2) * H + head];
#pragma unroll
              for (int j = 0; j < 4; ++j) s[j] += (kok && (unsigned)(q4i + j - off) < (unsigned)Len) ? b[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float p = __builtin_amdgcn_exp2f(s[j]);
              const float ds = p * dp[j];
              dv01[t] = pk_fma2(f32x2{p, p}, f32x2{o4[j].x, o4[j].y}, dv01[t]);
              dv23[t] = pk_fma2(f32x2{p, p}, f32x2{o4[j].z, o4[j].w}, dv23[t]);
              dk01[t] = pk_fma2(f32x2{ds, ds}, f32x2{q4[j].x, q4[j].y}, dk01[t]);
              dk23[t] = pk_fma2(f32x2{ds, ds}, f32x2{q4[j].z, q4[j].w}, dk23[t]);
            }
          }
        };
        const bool anyk = meets(k0, 16 * QT);
        const int e0 = anyk ? xe0 : NT, e1 = anyk ? xe1 : NT;
#pragma unroll RAL_ATTNW_UNROLL
        for (int qt = 0; qt < e0; qt += 16) tileB(qt, std::false_type{});
        if constexpr (TAB) {
#pragma unroll 1
          for (int qt = e0; qt < e1; qt += 16) tileB(qt, std::true_type{});
#pragma unroll RAL_ATTNW_UNROLL
          for (int qt = e1; qt < NT; qt += 16) tileB(qt, std::false_type{});
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          const float vk = quad_rows_sum(dk01[t][0], dk01[t][1], dk23[t][0], dk23[t][1]);
          const float vv = quad_rows_sum(dv01[t][0], dv01[t][1], dv23[t][0], dv23[t][1]);
          const size_t kk = ((size_t)head * NT + ktok[t]) * 4 + g;
          dbase[(size_t)H * NT * 4 + kk] = vk * (RAL_LN2 * oscale);      // Q32 carried log2(e)
          dbase[(size_t)2 * H * NT * 4 + kk] = vv;
        }
      }
    }
    if constexpr (PREF) deposit();
    task = next;
  }
  if constexpr (TAB) {
    __syncthreads();
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tpart[(size_t)blockIdx.x * ntab + i] = dtab[i];
  }
}

// gtable[i] += sum over the workgroups' rows of partials: one WAVE per table entry (lane l adds rows l, l + 64, ...: every
// load of the launch is independent; a thread per entry walking the rows was a 200-link chain of memory round trips,
// + 50 us behind a 120 us kernel)
__global__ __launch_bounds__(256) void k_attn_tpart_reduce(const float* __restrict__ tpart, float* __restrict__ gtable, int ntab, int nrow) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= ntab) return;
  float a0 = 0.f, a1 = 0.f;
  int b = lane;
  for (; b + 64 < nrow; b += 128) { a0 += tpart[(size_t)b * ntab + i]; a1 += tpart[(size_t)(b + 64) * ntab + i]; }
  if (b < nrow) a0 += tpart[(size_t)b * ntab + i];
  const float t = group_sum<64>(a0 + a1);
  if (lane == 0) atomicAdd(gtable + i, t);
}

// A caller that wants the reduction somewhere else than behind the attention kernel on the chain stream (ral_api.hip runs it on
// the block's weight-gradient stream: nothing on the chain reads the table gradient) sets this slot around launch_attn_bwd; the
// reduction is then recorded instead of launched.
thread_local AttnTabReduce* g_attn_tab_defer = nullptr;
void attn_tab_defer_to(AttnTabReduce* slot) { g_attn_tab_defer = slot; if (slot) slot->ntab = 0; }
void launch_attn_tpart_reduce(const float* tpart, float* gtable, int ntab, int nrow, hipStream_t s) {
  if (g_attn_tab_defer) { *g_attn_tab_defer = AttnTabReduce{tpart, gtable, ntab, nrow}; return; }
  k_attn_tpart_reduce<<<(ntab + 3) / 4, 256, 0, s>>>(tpart, gtable, ntab, nrow);
}

// ---------------------------------------------------------------------------------------------------------------------
int attn_f16_default() {
  static const int m = (int)ral_knob("ATTN_F16", 1);
  return m;
}
static int attnw_mode() {   // RAL_ATTN_BWD_W=0: never (the workgroup kernels of ral_bwd.hip take every shape)
  static const int m = (int)ral_knob("ATTN_BWD_W", 1);
  return m;
}
// upper bound of the grid (what the scratch is sized for): one workgroup per four tasks, at most 1024
int attnw_grid_max(int N, int H, int B) {
  const int hw = N >= 64 ? 1 : 64 / N;
  const int ntask = B * H / hw, g = (ntask + 3) / 4;
  return g < 1024 ? g : 1024;
}
// the grid of a launch: a whole number of resident rounds (workgroups per CU x 256 CUs) - 1024 workgroups on 768 slots
// would run a second round one third full (RAL_GRID_ATTNW overrides)
template <class K>
static int attnw_grid(K kernel, size_t lds, int N, int H, int B) {
  static const int genv = (int)ral_knob("GRID_ATTNW", 0);
  const int gmax = attnw_grid_max(N, H, B);
  if (genv > 0) return genv < gmax ? genv : gmax;
  const int occ = ral_occupancy(reinterpret_cast<const void*>(kernel), 256, lds, 3);
  const int slots = ral_num_cus() * (occ > 4 ? 4 : occ);
  return slots < gmax ? slots : gmax;
}
bool attn_bwd_w_takes(int N, int H, int Len, bool table) {
  if (!attnw_mode()) return false;
  if (N != 32 && N != 64 && N != 128) return false;
  if (N == 32 && (H & 1)) return false;
  if (table && (2 * Len - 1) * H > 2048) return false;
  return true;
}
size_t attn_bwd_w_scratch_floats(int N, int H, int Len, bool table, int B) {
  if (!table || !attn_bwd_w_takes(N, H, Len, table)) return 0;
  return (size_t)attnw_grid_max(N, H, B) * (size_t)((2 * Len - 1) * H);
}
// =====================================================================================================================
// Attention FORWARD (softmax(q k^T + bias) v, raletransformer.py:299-316 / transformer.py:302-310) in the same two forms:
// k_attn_fwd_w (N = 32, 64, 128: one wave per head, operands requested one task ahead) and k_attn_fwd_h (N >= 256: a
// workgroup per (window, head group)), with the S tile on the f16 matrix cores when F16.  The softmax shift is the
// Cauchy-Schwarz bound of k_attn_fwd (ral_fwd.hip): m[q] = |q| max|k| + max(bias, 0) enters the MFMA as its C operand;
// a row whose sum underflows is redone with the exact running maximum.
// =====================================================================================================================
// exact recurrence for one query tile (lane = query r, keys 4g .. 4g + 3 of every tile; merged over the lane groups)
template <bool F16, class SF, class BF, class VF>
RAL_DEV void attn_fwd_exact(int NTK, SF score, BF bias, VF vrow, float4& o, float& mg, float& l) {
  float mx = -INFINITY; l = 0.f; o = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int kt = 0; kt < NTK; kt += 16) {
    f32x4 s = score(kt);
    bias(kt, s);
    const float mn = fmaxf(fmaxf(mx, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
    const float corr = __builtin_amdgcn_exp2f(mx - mn);
    mx = mn; l *= corr; o = f4scale(o, corr);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = __builtin_amdgcn_exp2f(s[j] - mn);
      const float4 v = vrow(kt, j);
      l += p;
      o.x = fmaf(p, v.x, o.x); o.y = fmaf(p, v.y, o.y); o.z = fmaf(p, v.z, o.z); o.w = fmaf(p, v.w, o.w);
    }
  }
  mg = rows_max(mx);
  const float sc = __builtin_amdgcn_exp2f(mx - mg);
  l = rows_sum(l * sc);
  o = make_float4(rows_sum(o.x * sc), rows_sum(o.y * sc), rows_sum(o.z * sc), rows_sum(o.w * sc));
}

template <int NT, bool TAB, bool F16>
__global__ __launch_bounds__(256, (NT == 64 ? 3 : 4)) void k_attn_fwd_w(const float* qkv, float* o_hm, float* lse, const float* __restrict__ table,
                                                       int H, int Len_rt, int ntask) {
  constexpr int QT = 2;
  constexpr int HW = NT >= 64 ? 1 : 64 / NT;
  constexpr int T = HW * NT, TPL = T / 64;
  constexpr int WSZ = T * 13;                  // floats of LDS per wave: q, k operand tiles + v quads + |q|
  constexpr int NB = NT / (16 * QT);
  constexpr int SH = TAB ? 8 : 0, MSK = NT - 1;   // rotated tiles (see k_attn_bwd_w)
  constexpr int NBU = NB <= 2 ? NB : 1;
  extern __shared__ float4 smem4[];
  float* sm = reinterpret_cast<float*>(smem4);
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* Qx = sm + wave * WSZ;   // q log2 e: fp32 quads, or (F16) token-interleaved pair planes - 16 bytes per token either way
  float* Kx = Qx + 4 * T;
  float* V32 = Kx + 4 * T;
  float* Mq = V32 + 4 * T;       // |q| (log2 units)
  const int Len = TAB ? Len_rt : 0;
  const int ntab = TAB ? (2 * Len - 1) * H : 0;
  float* tab = sm + 4 * WSZ;     // bias * log2(e), (2 Len - 1, H)
  float* bmax = tab + ntab;      // H: max(bias, 0) per head
  const int off = (NT - Len) >> 1;
  if constexpr (TAB) {
    for (int i = threadIdx.x; i < ntab; i += blockDim.x) tab[i] = table[i] * RAL_LOG2E;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
      float m = 0.f;
      for (int e = 0; e < 2 * Len - 1; ++e) m = fmaxf(m, table[e * H + h] * RAL_LOG2E);
      bmax[h] = m;
    }
    __syncthreads();
  }
  auto meets = [&](int x0, int w) -> bool { return TAB && (off < SH || (x0 < off + Len - SH && x0 + w > off - SH)); };
  const int xe0 = !TAB ? NT : (off < SH ? 0 : ((off - SH) & ~15));
  const int xe1 = !TAB ? NT : (off < SH ? NT : ((off + Len - SH + 15) & ~15));
  const int stride = gridDim.x * 4;
  int task = blockIdx.x * 4 + wave;
  float kmx = 0.f;               // max |k| of the task in the LDS (a bound for every head of the task)
  float4 pq, pk, pv;
  auto task_ptrs = [&](int tk, const float4*& gq, const float4*& gk, const float4*& gv) {
    const int hh = tk * HW, win = hh / H, h0 = hh - win * H;
    gq = reinterpret_cast<const float4*>(qkv) + ((size_t)win * 3 * H + h0) * NT;
    gk = gq + (size_t)H * NT;
    gv = gk + (size_t)H * NT;
  };
  auto put = [&](int t, float4 q, float4 k, float4 v, float cq, float ck) {
    const float4 ql = f4scale(q, RAL_LOG2E);
    if constexpr (F16) {
      const H2x4 a = split4(f4scale(ql, cq)), b = split4(f4scale(k, ck));
      *reinterpret_cast<h16x4*>(Qx + 4 * t) = a.a; *reinterpret_cast<h16x4*>(Qx + 4 * t + 2) = a.b;
      *reinterpret_cast<h16x4*>(Kx + 4 * t) = b.a; *reinterpret_cast<h16x4*>(Kx + 4 * t + 2) = b.b;
    } else {
      reinterpret_cast<float4*>(Qx)[t] = ql; reinterpret_cast<float4*>(Kx)[t] = k;
    }
    reinterpret_cast<float4*>(V32)[t] = v;
    Mq[t] = sqrtf(f4dot(ql, ql));
  };
  auto request = [&](int tk) {
    const float4 *gq, *gk, *gv; task_ptrs(tk, gq, gk, gv);
    pq = gq[lane]; pk = gk[lane]; pv = gv[lane];
  };
  auto deposit = [&]() {
    kmx = sqrtf(group_max<64>(f4dot(pk, pk))) * 1.0000002f;
    float cq = 1.f, ck = 1.f;
    if constexpr (F16) pair_balance(group_max<64>(f4absmax(pq)) * RAL_LOG2E, group_max<64>(f4absmax(pk)), cq, ck);
    put(lane, pq, pk, pv, cq, ck);
  };
  auto stage = [&](int tk) {
    const float4 *gq, *gk, *gv; task_ptrs(tk, gq, gk, gv);
    float4 q[TPL], k[TPL], v[TPL];
#pragma unroll
    for (int u = 0; u < TPL; ++u) { const int t = lane + 64 * u; q[u] = gq[t]; k[u] = gk[t]; v[u] = gv[t]; }
    float m = 0.f, mq = 0.f, mk = 0.f;
#pragma unroll
    for (int u = 0; u < TPL; ++u) { m = fmaxf(m, f4dot(k[u], k[u])); mq = fmaxf(mq, f4absmax(q[u])); mk = fmaxf(mk, f4absmax(k[u])); }
    kmx = sqrtf(group_max<64>(m)) * 1.0000002f;
    float cq = 1.f, ck = 1.f;
    if constexpr (F16) pair_balance(group_max<64>(mq) * RAL_LOG2E, group_max<64>(mk), cq, ck);
#pragma unroll
    for (int u = 0; u < TPL; ++u) put(lane + 64 * u, q[u], k[u], v[u], cq, ck);
  };
  auto opA = [&](const float* X, int tok) -> h16x4 { return *reinterpret_cast<const h16x4*>(X + 4 * tok + 2 * (g >> 1)); };
  auto opB = [&](const float* X, int tok) -> h16x4 { return *reinterpret_cast<const h16x4*>(X + 4 * tok + 2 * (g & 1)); };
  constexpr bool PREF = TPL == 1;
  if constexpr (PREF) { if (task < ntask) { request(task); deposit(); } }
  while (task < ntask) {
    const int next = task + stride;
    if constexpr (PREF) { request(next < ntask ? next : task); asm volatile("" ::: "memory"); }
    else stage(task);
    const int hh = task * HW, win = hh / H, h0 = hh - win * H;
#pragma unroll
    for (int hl = 0; hl < HW; ++hl) {
      const int head = h0 + hl, tb = hl * NT;
      const size_t hq0 = ((size_t)win * H + head) * NT;
      const float4* V4 = reinterpret_cast<const float4*>(V32) + tb;
      const float badd = TAB ? bmax[head] : 0.f;
#pragma unroll NBU
      for (int qb = 0; qb < NB; ++qb) {
        const int q0 = qb * 16 * QT;
        float qf[QT]; h16x4 qh[QT];
        float mq[QT];
        f32x4 nm[QT];
        f32x2 l2[QT], o01[QT], o23[QT];
        bool qin[QT];
        int qtok[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const int q = (q0 + 16 * qt + SH + r) & MSK;
          qtok[qt] = q;
          if constexpr (F16) qh[qt] = opB(Qx, tb + q); else qf[qt] = Qx[(tb + q) * 4 + g];
          mq[qt] = Mq[tb + q] * kmx + badd;
          nm[qt] = f32x4{-mq[qt], -mq[qt], -mq[qt], -mq[qt]};
          l2[qt] = f32x2{0.f, 0.f}; o01[qt] = f32x2{0.f, 0.f}; o23[qt] = f32x2{0.f, 0.f};
          qin[qt] = meets(q0 + 16 * qt, 16);
        }
        auto addbias = [&](int kt, int qtk, f32x4& s) {
          const int k4i = (kt + SH + 4 * g) & MSK;
          const int qi = qtk - off;
          const bool qok = (unsigned)qi < (unsigned)Len;
          const int rel0 = qi - (k4i - off) + Len - 1;
          float b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) b[j] = tab[min(max(rel0 - j, 0), 2 * Len - 2) * H + head];
#pragma unroll
          for (int j = 0; j < 4; ++j) s[j] += (qok && (unsigned)(k4i + j - off) < (unsigned)Len) ? b[j] : 0.f;
        };
        auto tile = [&](int kt, auto biased) {
          const int kr = tb + ((kt + SH + r) & MSK), k4i = (kt + SH + 4 * g) & MSK;
          float kf; h16x4 kh;
          if constexpr (F16) kh = opA(Kx, kr); else kf = Kx[kr * 4 + g];
          float4 v4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v4[j] = V4[k4i + j];
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) {
            f32x4 s;
            if constexpr (F16) s = __builtin_amdgcn_mfma_f32_16x16x16f16(kh, qh[qt], nm[qt], 0, 0, 0);
            else s = mfma4(kf, qf[qt], nm[qt]);                      // s - m, log2 units   [key 4g+j][query r]
            if (decltype(biased)::value && qin[qt]) addbias(kt, qtok[qt], s);
            const f32x2 p01 = f32x2{__builtin_amdgcn_exp2f(s[0]), __builtin_amdgcn_exp2f(s[1])};
            const f32x2 p23 = f32x2{__builtin_amdgcn_exp2f(s[2]), __builtin_amdgcn_exp2f(s[3])};
            l2[qt] += p01;
            l2[qt] += p23;
            o01[qt] = pk_fma2(f32x2{p01[0], p01[0]}, f32x2{v4[0].x, v4[0].y}, o01[qt]);
            o23[qt] = pk_fma2(f32x2{p01[0], p01[0]}, f32x2{v4[0].z, v4[0].w}, o23[qt]);
            o01[qt] = pk_fma2(f32x2{p01[1], p01[1]}, f32x2{v4[1].x, v4[1].y}, o01[qt]);
            o23[qt] = pk_fma2(f32x2{p01[1], p01[1]}, f32x2{v4[1].z, v4[1].w}, o23[qt]);
            o01[qt] = pk_fma2(f32x2{p23[0], p23[0]}, f32x2{v4[2].x, v4[2].y}, o01[qt]);
            o23[qt] = pk_fma2(f32x2{p23[0], p23[0]}, f32x2{v4[2].z, v4[2].w}, o23[qt]);
            o01[qt] = pk_fma2(f32x2{p23[1], p23[1]}, f32x2{v4[3].x, v4[3].y}, o01[qt]);
            o23[qt] = pk_fma2(f32x2{p23[1], p23[1]}, f32x2{v4[3].z, v4[3].w}, o23[qt]);
          }
        };
        const bool anyq = meets(q0, 16 * QT);
        const int e0 = anyq ? xe0 : NT, e1 = anyq ? xe1 : NT;
#pragma unroll 1
        for (int kt = 0; kt < e0; kt += 16) tile(kt, std::false_type{});
        if constexpr (TAB) {
#pragma unroll 1
          for (int kt = e0; kt < e1; kt += 16) tile(kt, std::true_type{});
#pragma unroll 1
          for (int kt = e1; kt < NT; kt += 16) tile(kt, std::false_type{});
        }
        bool redo = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
          const float lv = rows_sum(l2[qt][0] + l2[qt][1]);
          const float ov = quad_rows_sum(o01[qt][0], o01[qt][1], o23[qt][0], o23[qt][1]);
          redo = redo || !(lv > 1e-30f);
          o_hm[(hq0 + qtok[qt]) * 4 + g] = ov * (1.0f / lv);
          if (lse && g == 0) lse[hq0 + qtok[qt]] = (mq[qt] + __builtin_amdgcn_logf(lv)) * RAL_LN2;   // natural-log units
        }
        if (__any(redo)) {   // (rolled, with the tile's values picked by static index: a run-time index would park the arrays in scratch)
          static_assert(QT == 2, "two query tiles");
#pragma unroll 1
          for (int qt = 0; qt < QT; ++qt) {
            const h16x4 qhv = qt ? qh[1] : qh[0];
            const float qfv = qt ? qf[1] : qf[0];
            const int qtk = qt ? qtok[1] : qtok[0];
            const bool qinv = qt ? qin[1] : qin[0];
            float4 o; float mg, l;
            attn_fwd_exact<F16>(NT,
                                [&](int kt) {
                                  const int kr = tb + ((kt + SH + r) & MSK);
                                  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x16f16(opA(Kx, kr), qhv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                                  else return mfma4(Kx[kr * 4 + g], qfv, f32x4{0.f, 0.f, 0.f, 0.f});
                                },
                                [&](int kt, f32x4& sv) { if (TAB && qinv && meets(kt, 16)) addbias(kt, qtk, sv); },
                                [&](int kt, int j) { return V4[((kt + SH + 4 * g) & MSK) + j]; }, o, mg, l);
            const float inv = 1.0f / l;
            const float og = g == 0 ? o.x : (g == 1 ? o.y : (g == 2 ? o.z : o.w));
            o_hm[(hq0 + qtk) * 4 + g] = og * inv;
            if (lse && g == 0) lse[hq0 + qtk] = (mg + __builtin_amdgcn_logf(l)) * RAL_LN2;
          }
        }
      }
    }
    if constexpr (PREF) deposit();
    task = next;
  }
}

bool attn_fwd_w_takes(int N, int H, int Len, bool table) {
  static const int on = (int)ral_knob("ATTN_FWD_W", 1);
  // measured at batch 2048 (us per launch, this kernel with f16 tiles / the kernels of ral_fwd.hip): N = 32: 39 / 45,
  // 64 (table): 64 / 54, 128 (table): 90 / 91 - the scalar-path forward keeps N = 64 and 128 (RAL_ATTN_FWD_W=2: all three)
  if (!on || (N != 32 && N != 64 && N != 128) || (on < 2 && N != 32)) return false;
  if (N == 32 && (H & 1)) return false;
  if (table && (2 * Len - 1) * H > 2048) return false;
  return true;
}
void launch_attn_fwd_w(const float* qkv, float* o_hm, float* lse, const float* table, int N, int H, int Len, int B, int f16,
                       hipStream_t s) {
  const int hw = N >= 64 ? 1 : 64 / N, T = hw * N;
  const int ntask = B * H / hw;
  const int ntab = table ? (2 * Len - 1) * H : 0;
  const size_t lds = ((size_t)4 * T * 13 + ntab + (table ? H : 0)) * sizeof(float);
  int grid = 0;
#define GO(n, tab, h) { RAL_SET_LDS((k_attn_fwd_w<n, tab, h>), lds); grid = attnw_grid(k_attn_fwd_w<n, tab, h>, lds, N, H, B); \
    k_attn_fwd_w<n, tab, h><<<grid, 256, lds, s>>>(qkv, o_hm, lse, table, H, Len, ntask); }
#define GOH(n, tab) { if (f16) GO(n, tab, true) else GO(n, tab, false) }
  if (N == 32) { if (table) GOH(32, true) else GOH(32, false) }
  else if (N == 64) { if (table) GOH(64, true) else GOH(64, false) }
  else { if (table) GOH(128, true) else GOH(128, false) }
#undef GOH
#undef GO
}

void launch_attn_bwd_w(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                       float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, int f16, hipStream_t s) {
  const int hw = N >= 64 ? 1 : 64 / N, T = hw * N;
  const int ntask = B * H / hw;
  const int ntab = table ? (2 * Len - 1) * H : 0;
  // waves per workgroup (RAL_ATTNW_WAVES).  Measured at batch 2048, us per launch with 4 / 3 / 2 / 1 waves: N = 128: 236 / 274 /
  // 245 / 283 (four-wave workgroups of 61 KB leave a CU two of them, five two-wave ones fit - and are no faster), N = 64:
  // 131 / 135 / 142 / 173, N = 32: 82 / 80 / 83 / 99
  static const int wv_env = (int)ral_knob("ATTNW_WAVES", 0);
  const int nwv = wv_env ? wv_env : 4;
  (void)f16;
  const size_t lds = ((size_t)nwv * T * 18 + 2 * ntab) * sizeof(float);
  int grid = 0;
  auto grid_of = [&](auto kern) {
    static const int genv = (int)ral_knob("GRID_ATTNW", 0);
    const int gmax = attnw_grid_max(N, H, B);          // (the scratch is sized for one workgroup per four tasks)
    if (genv > 0) return genv < gmax ? genv : gmax;
    const int occ = ral_occupancy(reinterpret_cast<const void*>(kern), 64 * nwv, lds, 3);
    const int slots = ral_num_cus() * (occ > 8 ? 8 : occ);
    const int need = (ntask + nwv - 1) / nwv;
    int g = slots < need ? slots : need;
    return g < gmax ? g : gmax;
  };
#define GO(n, tab, h, qt) { RAL_SET_LDS((k_attn_bwd_w<n, tab, h, qt>), lds); grid = grid_of(k_attn_bwd_w<n, tab, h, qt>); \
    k_attn_bwd_w<n, tab, h, qt><<<grid, 64 * nwv, lds, s>>>(qkv, o_hm, do_hm, lse, table, tpart, dqkv, H, Len, ntask); }
#define GOH(n, tab) { GO(n, tab, false, 2) }   // (the f16 tiles of this form went with round 5: k_attn_bwd_m, ral_attnm.hip)
  if (N == 32) { if (table) GOH(32, true) else GOH(32, false) }
  else if (N == 64) {
    if (table) GOH(64, true) else GOH(64, false)
  }
  else { if (table) GOH(128, true) else GOH(128, false) }
#undef GOH
#undef GO
  if (table) launch_attn_tpart_reduce(tpart, gtable, ntab, grid, s);
}
