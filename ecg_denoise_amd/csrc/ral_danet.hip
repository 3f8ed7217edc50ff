// DANet comparison baseline (reference model/DAM.py::Seq2Seq2, :341-349) for gfx950.
//
//   encoder cell i (:51-77)   Conv1d(stride 2; k 17,17,3,3; pad 8,8,1,1; leads->4->8->16->32) -> APReLU -> BatchNorm1d
//   decoder cell i (:158-190) ConvTranspose1d(stride 2; k 4,4,18,18; pad 1,1,8,8; 32->16->8->4->2) -> APReLU -> BatchNorm1d
//                             -> DAM (cells 0..2); cells 1..3 take previous output + encoder feature (:329-336)
//   APReLU (:12-48)           p + alpha * n,  alpha = sigmoid(BN(W3 relu(BN(W0 [mean p; mean n] + b0)) + b3)), BN over the batch
//   DAM (:101-155)            Sattn * (Cattn * x): Cattn = sigmoid(fcn(mean_L x) + fcn(max_L x)) with ONE shared fcn (same
//                             shape as APReLU's, C -> C -> C), Sattn = sigmoid(conv1x1([mean_C x; max_C x]))
//
// Every tensor of a window is 2 L floats (4 KB at L = 512) and every window is independent EXCEPT through the batch
// statistics of the 36 BatchNorms: 4 per cell (two inside APReLU on (B, 2C) / (B, C) descriptors, one on the cell output)
// plus 4 per DAM (the shared fcn sees the average-pooled and the max-pooled batch).  The forward pass is therefore a chain
// of per-window kernels cut at those reductions - each kernel keeps its window in LDS, adds its partial column sums
// (double) to a small buffer, and the NEXT kernel turns the finished sums into mean / rstd on the fly:
//
//   k_dn_conv      input (+ skip) -> conv -> z, [mean p; mean n] -> first Linear of the APReLU fcn -> h1   | sums(h1)
//   k_dn_fcn_mid   relu(BN(h1)) -> second Linear -> h2 (one or two paths)                                     | sums(h2)
//   k_dn_act       alpha = sigmoid(BN(h2)); a = p + alpha n                                                   | sums(a)
//   k_dn_dam1      x = BN(a); mean_L x, max_L x -> first Linear of the DAM fcn, both paths                    | sums(h1 x 2)
//   k_dn_fcn_mid   (two paths)                                                                                | sums(h2 x 2)
//   k_dn_out       x = BN(a) [-> Cattn, Sattn -> Sattn Cattn x]  -> cell output
//
// (4 launches per plain cell, 6 per DAM cell, 38 per forward; in eval mode the same kernels read the running statistics
// and skip the sums.)  The backward pass mirrors it: every BatchNorm backward needs sum(dy) and sum(dy * xhat) over the
// batch, so it is cut at the same places - k_dn_dam_b, k_dn_fcn_bmid, k_dn_bn_b, k_dn_act_b, k_dn_fcn_bmid, k_dn_conv_b -
// with weight gradients accumulated per workgroup in LDS and added to the gradient buffer once at the end.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ral_danet.hpp"
#include "ral_device.hpp"

namespace {
const int ENC_CH[4] = {4, 8, 16, 32}, ENC_K[4] = {17, 17, 3, 3}, ENC_P[4] = {8, 8, 1, 1};
const int DEC_CH[4] = {16, 8, 4, 2}, DEC_K[4] = {4, 4, 18, 18}, DEC_P[4] = {1, 1, 8, 8};
constexpr float BN_EPS = 1e-5f, BN_MOM = 0.1f;
constexpr int NT = 256;           // threads per workgroup
constexpr int MAXC = 64;          // widest descriptor (2 x 32)
constexpr int CH = 12;            // zero halo of the conv tiles in LDS (>= 9 = the widest tap reach, multiple of 4)
// sum-buffer slots of a cell (doubles): forward column sums, then the sums of the BatchNorm backwards
enum { S_AH1 = 0, S_AH2 = 128, S_BN = 192, S_DH1 = 256, S_DH2 = 384, T_D2 = 512, T_D1 = 640, T_BN = 768, T_A2 = 832, T_A1 = 896, S_CELL = 1024 };

struct Bn { const float *g, *b; float *rm, *rv; float *dg, *db; };
struct Fcn {              // Linear(din, dh) -> BN -> ReLU -> Linear(dh, dout) -> BN -> Sigmoid
  const float *w0, *b0, *w3, *b3; float *dw0, *db0, *dw3, *db3;
  Bn bn1, bn2; int din, dh, dout;
};
struct Geo { int cin, c, k, p, tr, lin, lout, dam; };

RAL_DEV float sigm(float v) { return 1.f / (1.f + __expf(-v)); }

// mean / rstd of column j: from the finished batch sums (training) or the running statistics (eval)
RAL_DEV void bn_stat(const double* sums, int ncol, int j, double cnt, bool training, const float* rm, const float* rv,
                     float& mean, float& rstd) {
  if (training) {
    const double m = sums[j] / cnt, v = sums[ncol + j] / cnt - m * m;
    mean = (float)m; rstd = (float)(1.0 / sqrt((v > 0 ? v : 0) + (double)BN_EPS));
  } else {
    mean = rm[j]; rstd = 1.f / sqrtf(rv[j] + BN_EPS);
  }
}
// torch's running-statistic update from finished sums (momentum 0.1, unbiased variance)
RAL_DEV void bn_running(const double* sums, int ncol, int j, double cnt, float* rm, float* rv) {
  const double m = sums[j] / cnt, v = sums[ncol + j] / cnt - m * m;
  rm[j] = (1.f - BN_MOM) * rm[j] + BN_MOM * (float)m;
  rv[j] = (1.f - BN_MOM) * rv[j] + BN_MOM * (float)((v > 0 ? v : 0) * cnt / (cnt > 1 ? cnt - 1 : 1));
}
RAL_DEV float wave_sum(float v) {
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
  return v;
}
// sum over groups of `w` consecutive lanes (w a power of two in [1, 64], every lane of the wave active), result in all lanes
RAL_DEV float seg_sum(float v, int w) {
  switch (w) {
    case 64: return group_sum<64>(v);
    case 32: return group_sum<32>(v);
    case 16: return group_sum<16>(v);
    case 8: return group_sum<8>(v);
    case 4: return group_sum<4>(v);
    case 2: return group_sum<2>(v);
    default: return v;
  }
}
RAL_DEV float hsum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }
RAL_DEV int ec_of(int tid, int C) { return tid % C; }
// items of V floats (V = 4: 16-byte accesses; V = 1 for level lengths that are not a multiple of 4: value in .x, rest 0)
template <int V> RAL_DEV float4 ldv(const float* p, int i) {
  if constexpr (V == 4) return reinterpret_cast<const float4*>(p)[i];
  else return make_float4(p[i], 0.f, 0.f, 0.f);
}
template <int V> RAL_DEV void stv(float* p, int i, float4 v) {
  if constexpr (V == 4) reinterpret_cast<float4*>(p)[i] = v;
  else p[i] = v.x;
}
RAL_DEV float wave_max(float v) {
  for (int s = 32; s > 0; s >>= 1) v = fmaxf(v, __shfl_xor(v, s));
  return v;
}

// ---------------------------------------------------------------------------------
// k_dn_conv: in0 (+ in1) (B, cin, lin) -> z (B, c, lout), desc (B, 2c) = [mean_L max(z,0); mean_L min(z,0)],
//            h1 (B, dh) = W0 desc + b0, column sums of h1
// ---------------------------------------------------------------------------------
template <int K, bool TR>
__global__ __launch_bounds__(NT) void k_dn_conv(const float* __restrict__ in0, const float* __restrict__ in1,
                                                const float* __restrict__ w, const float* __restrict__ bias, Geo G, Fcn F,
                                                float* __restrict__ z, float* __restrict__ desc, float* __restrict__ h1,
                                                double* sums, int B) {
  extern __shared__ float sm[];
  const int LP = G.lin + 2 * CH;         // input rows carry a zero halo of CH on both sides: no tap needs a bounds check
  float* xs = sm;                        // cin x LP
  float* zs = xs + G.cin * LP;           // c x lout
  float* ds = zs + G.c * G.lout;         // 2c
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nin = G.cin * G.lin, nz = G.c * G.lout;
  for (int i = tid; i < G.cin * 2 * CH; i += NT) { const int c = i / (2 * CH), h = i - c * 2 * CH; xs[c * LP + (h < CH ? h : G.lin + h)] = 0.f; }
  double s1 = 0, s2 = 0;                 // column sums of h1 (thread j < dh owns column j)
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < nin; i += NT) {
      const int c = i / G.lin, l = i - c * G.lin;
      xs[c * LP + CH + l] = in0[(size_t)win * nin + i] + (in1 ? in1[(size_t)win * nin + i] : 0.f);
    }
    __syncthreads();
    for (int o = tid; o < nz; o += NT) {
      const int co = o / G.lout, l = o - co * G.lout;
      float acc = bias[co];
      if constexpr (!TR) {   // out[co][l] = sum_ci sum_k w[co][ci][k] in[ci][2 l + k - p]
        for (int ci = 0; ci < G.cin; ++ci) {
          const float* wr = w + ((size_t)co * G.cin + ci) * K;
          const float* row = xs + ci * LP + CH + 2 * l - G.p;
#pragma unroll
          for (int k = 0; k < K; ++k) acc = fmaf(wr[k], row[k], acc);
        }
      } else {               // out[co][t] = sum_ci sum_{k = t + p - 2 s} w[ci][co][k] in[ci][s]: the taps of t's parity
        const int k0 = (l + G.p) & 1, s0 = (l + G.p - k0) >> 1;
        for (int ci = 0; ci < G.cin; ++ci) {
          const float* wr = w + ((size_t)ci * G.c + co) * K + k0;
          const float* row = xs + ci * LP + CH + s0;
#pragma unroll
          for (int kk = 0; kk < K / 2; ++kk) acc = fmaf(wr[2 * kk], row[-kk], acc);
        }
      }
      zs[o] = acc;
      z[(size_t)win * nz + o] = acc;
    }
    __syncthreads();
    for (int c = wave; c < G.c; c += NT / 64) {
      float p = 0.f, n = 0.f;
      for (int l = lane; l < G.lout; l += 64) { const float v = zs[c * G.lout + l]; p += fmaxf(v, 0.f); n += fminf(v, 0.f); }
      p = wave_sum(p); n = wave_sum(n);
      if (lane == 0) { ds[c] = p / G.lout; ds[G.c + c] = n / G.lout; }
    }
    __syncthreads();
    if (tid < 2 * G.c) desc[(size_t)win * 2 * G.c + tid] = ds[tid];
    if (tid < F.dh) {
      float acc = F.b0[tid];
      for (int i = 0; i < F.din; ++i) acc = fmaf(F.w0[tid * F.din + i], ds[i], acc);
      h1[(size_t)win * F.dh + tid] = acc;
      s1 += acc; s2 += (double)acc * acc;
    }
  }
  if (sums && tid < F.dh) { atomicAdd(sums + tid, s1); atomicAdd(sums + F.dh + tid, s2); }
}

// ---------------------------------------------------------------------------------
// k_dn_fcn_mid: h2 = W3 relu(BN1(h1)) + b3 for `np` paths (path p at h1 + p B dh, sums + p 2 dh, ...); column sums of h2;
// workgroup 0 applies the running-statistic update of BN1 (path by path: the shared DAM fcn sees two batches).
// ---------------------------------------------------------------------------------
// One WAVE per window, four windows of a wave in flight, no block barrier inside the window loop: these kernels move
// 64 floats per window, so their time was one memory round trip and two barriers per window and, at 1024 workgroups,
// a 1024-link same-address chain of double atomics per column sum (25 us for half a megabyte).  Lane l < np dh stages
// relu(BN1(h1)) of (path, j) in the wave's LDS row; lane l < np dout forms column l of the second Linear from it.
constexpr int DWIN = 4;                 // windows per wave and pass
constexpr int DLD = 65;                 // LDS row stride of W3 (dout x dh, dh <= 64): odd, so that 32 rows hit 32 banks
__global__ __launch_bounds__(NT) void k_dn_fcn_mid(const float* __restrict__ h1, const double* sums1, Fcn F,
                                                   float* __restrict__ h2, double* sums2, int np, int B, int training) {
  __shared__ float w3s[32 * DLD], mean1[2][MAXC], rstd1[2][MAXC], g1s[MAXC], b1s[MAXC];
  __shared__ float rs[NT / 64][DWIN][2 * MAXC];      // per wave and window: relu(BN1(h1)), both paths
  __shared__ double red[NT / 64][2][MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < np * F.dh; i += NT) {
    const int p = i / F.dh, j = i - p * F.dh;
    bn_stat(sums1 + p * 2 * F.dh, F.dh, j, B, training, F.bn1.rm, F.bn1.rv, mean1[p][j], rstd1[p][j]);
  }
  if (tid < F.dh) { g1s[tid] = F.bn1.g[tid]; b1s[tid] = F.bn1.b[tid]; }
  for (int i = tid; i < F.dout * F.dh; i += NT) w3s[(i / F.dh) * DLD + i % F.dh] = F.w3[i];
  if (training && blockIdx.x == 0 && tid < F.dh)
    for (int p = 0; p < np; ++p) bn_running(sums1 + p * 2 * F.dh, F.dh, tid, B, F.bn1.rm, F.bn1.rv);
  __syncthreads();
  const int nin = np * F.dh, nout = np * F.dout;
  const int pi = lane / F.dh, ji = lane - pi * F.dh;          // this lane's input element (lane < nin)
  const int pc = lane / F.dout, cc = lane - pc * F.dout;      // this lane's output column (lane < nout)
  const float b3 = lane < nout ? F.b3[cc] : 0.f;
  double s1 = 0, s2 = 0;
  const int nwv = gridDim.x * (NT / 64);
  for (int w0 = (blockIdx.x * (NT / 64) + wave) * DWIN; w0 < B; w0 += nwv * DWIN) {
    float v[DWIN];
#pragma unroll
    for (int q = 0; q < DWIN; ++q) {               // the four loads first (clamped: no lane-predicated loads)
      const int win = w0 + q < B ? w0 + q : B - 1;
      v[q] = h1[((size_t)(lane < nin ? pi : 0) * B + win) * F.dh + (lane < nin ? ji : 0)];
    }
#pragma unroll
    for (int q = 0; q < DWIN; ++q)
      if (lane < nin) rs[wave][q][pi * MAXC + ji] = fmaxf(g1s[ji] * (v[q] - mean1[pi][ji]) * rstd1[pi][ji] + b1s[ji], 0.f);
    __builtin_amdgcn_wave_barrier();               // (the wave's own LDS writes are ordered before its reads)
    if (lane < nout) {
#pragma unroll
      for (int q = 0; q < DWIN; ++q) {
        if (w0 + q >= B) break;
        float acc = b3;
        const float* r = rs[wave][q] + pc * MAXC;
        const float* wr = w3s + cc * DLD;
        for (int j = 0; j < F.dh; ++j) acc = fmaf(wr[j], r[j], acc);
        h2[((size_t)pc * B + w0 + q) * F.dout + cc] = acc;
        s1 += acc; s2 += (double)acc * acc;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (sums2) {                                     // the four waves' column sums meet in LDS: one atomic pair per column and workgroup
    if (lane < nout) { red[wave][0][lane] = s1; red[wave][1][lane] = s2; }
    __syncthreads();
    if (tid < nout) {
      double a1 = 0, a2 = 0;
      for (int w = 0; w < NT / 64; ++w) { a1 += red[w][0][tid]; a2 += red[w][1][tid]; }
      const int p = tid / F.dout, c = tid - p * F.dout;
      atomicAdd(sums2 + p * 2 * F.dout + c, a1); atomicAdd(sums2 + p * 2 * F.dout + F.dout + c, a2);
    }
  }
}

// ---------------------------------------------------------------------------------
// k_dn_act: alpha = sigmoid(BN2(h2)); a = max(z, 0) + alpha min(z, 0); per-channel sums of a; BN2 running update
// ---------------------------------------------------------------------------------
// Elementwise kernels of the cell: a workgroup takes EW consecutive windows per pass - one contiguous block of every tensor,
// requested as 16-byte loads with all loads of the pass in flight - and a lane's four values belong to ONE channel row
// (lout is a multiple of 4), so the per-channel sums are wave-level segment sums in front of one LDS atomic per row piece.
// (A wave per channel row with 4-byte loads paid one memory round trip per 64 floats.)
constexpr int EW = 4;                              // windows per workgroup and pass
template <int V>
__global__ __launch_bounds__(NT) void k_dn_act(const float* __restrict__ z, const float* __restrict__ h2, const double* sums2,
                                               Fcn F, Geo G, float* __restrict__ a, double* sumsa, int B, int training) {
  __shared__ float mean2[MAXC], rstd2[MAXC], g2s[MAXC], b2s[MAXC], al[EW][MAXC];
  __shared__ float acc1[MAXC], acc2[MAXC];
  const int tid = threadIdx.x;
  if (tid < G.c) {
    bn_stat(sums2, G.c, tid, B, training, F.bn2.rm, F.bn2.rv, mean2[tid], rstd2[tid]);
    g2s[tid] = F.bn2.g[tid]; b2s[tid] = F.bn2.b[tid];
    acc1[tid] = 0.f; acc2[tid] = 0.f;
    if (training && blockIdx.x == 0) bn_running(sums2, G.c, tid, B, F.bn2.rm, F.bn2.rv);
  }
  const int nz = G.c * G.lout, n4w = nz / V, q4 = G.lout / V;          // items per window / per channel row
  const int segw = ((q4 & (q4 - 1)) == 0 && (EW * n4w) % 64 == 0) ? (q4 < 64 ? q4 : 64) : 1;
  double t1 = 0, t2 = 0;                                                // (thread c < C: running totals of channel c)
  for (int w0 = blockIdx.x * EW; w0 < B; w0 += gridDim.x * EW) {
    const int nwin = (B - w0) < EW ? (B - w0) : EW, n4 = nwin * n4w;
    const float* z4 = z + (size_t)w0 * nz;
    float* a4 = a + (size_t)w0 * nz;
    __syncthreads();
    if (tid < EW * G.c) {
      const int q = tid / G.c, c = tid - q * G.c, win = w0 + q < B ? w0 + q : B - 1;
      al[q][c] = sigm(g2s[c] * (h2[(size_t)win * G.c + c] - mean2[c]) * rstd2[c] + b2s[c]);
    }
    for (int i0 = 0; i0 < n4; i0 += 4 * NT) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const int i = i0 + k * NT + tid; v[k] = ldv<V>(z4, i < n4 ? i : 0); }
      __syncthreads();                                                   // (al of this pass; uniform: n4 is)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * NT + tid;
        float s1 = 0.f, s2 = 0.f;
        int c = 0;
        if (i < n4) {
          const int q = i / n4w, e = i - q * n4w;
          c = e / q4;
          const float al_ = al[q][c];
          float4 r = v[k];
          r.x = r.x > 0.f ? r.x : al_ * r.x; r.y = r.y > 0.f ? r.y : al_ * r.y;
          r.z = r.z > 0.f ? r.z : al_ * r.z; r.w = r.w > 0.f ? r.w : al_ * r.w;
          stv<V>(a4, i, r);
          s1 = hsum4(r); s2 = (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
        }
        if (sumsa) {
          if (segw > 1) { s1 = seg_sum(s1, segw); s2 = seg_sum(s2, segw); if ((tid & (segw - 1)) == 0 && i < n4) { atomicAdd(acc1 + c, s1); atomicAdd(acc2 + c, s2); } }
          else if (i < n4) { atomicAdd(acc1 + c, s1); atomicAdd(acc2 + c, s2); }
        }
      }
    }
    __syncthreads();
    if (tid < G.c) { t1 += acc1[tid]; t2 += acc2[tid]; acc1[tid] = 0.f; acc2[tid] = 0.f; }
  }
  if (sumsa && tid < G.c) { atomicAdd(sumsa + tid, t1); atomicAdd(sumsa + G.c + tid, t2); }
}

// x = BN(a) of one window into LDS (scale / shift per channel precomputed)
RAL_DEV void load_bn(const float* __restrict__ a, float* xs, const float* sc, const float* sh, int c, int lout, int tid) {
  if ((lout & 3) == 0) {     // 16-byte accesses: a lane's four values share a channel
    const float4* a4 = reinterpret_cast<const float4*>(a);
    for (int i = tid; i < (c * lout) >> 2; i += NT) {
      const int ch = (i << 2) / lout;
      const float4 v = a4[i];
      const float s = sc[ch], h = sh[ch];
      reinterpret_cast<float4*>(xs)[i] = make_float4(fmaf(v.x, s, h), fmaf(v.y, s, h), fmaf(v.z, s, h), fmaf(v.w, s, h));
    }
    return;
  }
  for (int o = tid; o < c * lout; o += NT) { const int ch = o / lout; xs[o] = fmaf(a[o], sc[ch], sh[ch]); }
}

// ---------------------------------------------------------------------------------
// k_dn_dam1: x = BN(a); gap = mean_L x, gmp = max_L x (stored, (2, B, c)); h1 = W0 [gap | gmp] + b0 (2, B, c); sums
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_dn_dam1(const float* __restrict__ a, const double* sumsa, Bn bn, Fcn F, Geo G,
                                                float* __restrict__ pool, float* __restrict__ h1, double* sums1, int B,
                                                int training) {
  extern __shared__ float sm[];
  float* xs = sm;                                 // c x lout
  __shared__ float sc[MAXC], sh[MAXC], pl[2][MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = G.c;
  if (tid < C) {
    float m, r;
    bn_stat(sumsa, C, tid, (double)B * G.lout, training, bn.rm, bn.rv, m, r);
    sc[tid] = bn.g[tid] * r; sh[tid] = bn.b[tid] - m * bn.g[tid] * r;
    if (training && blockIdx.x == 0) bn_running(sumsa, C, tid, (double)B * G.lout, bn.rm, bn.rv);
  }
  double s1 = 0, s2 = 0;
  const int pc = tid / C, cc = tid - pc * C;      // thread (path, column)
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    load_bn(a + (size_t)win * C * G.lout, xs, sc, sh, C, G.lout, tid);
    __syncthreads();
    for (int c = wave; c < C; c += NT / 64) {
      float s = 0.f, mx = -INFINITY;
      for (int l = lane; l < G.lout; l += 64) { const float v = xs[c * G.lout + l]; s += v; mx = fmaxf(mx, v); }
      s = wave_sum(s); mx = wave_max(mx);
      if (lane == 0) { pl[0][c] = s / G.lout; pl[1][c] = mx; }
    }
    __syncthreads();
    if (pc < 2) {
      pool[((size_t)pc * B + win) * C + cc] = pl[pc][cc];
      float acc = F.b0[cc];
      for (int i = 0; i < C; ++i) acc = fmaf(F.w0[cc * C + i], pl[pc][i], acc);
      h1[((size_t)pc * B + win) * C + cc] = acc;
      s1 += acc; s2 += (double)acc * acc;
    }
  }
  if (sums1 && pc < 2) { atomicAdd(sums1 + pc * 2 * C + cc, s1); atomicAdd(sums1 + pc * 2 * C + C + cc, s2); }
}

// Cattn of one window: sigmoid(sigmoid(BN2(h2_avg)) + sigmoid(BN2(h2_max))); optionally the two inner sigmoids
RAL_DEV float dam_cattn(const float* h2, int B, int win, int C, int c, const Fcn& F, const float (*mean2)[MAXC],
                        const float (*rstd2)[MAXC], float* sa, float* smx) {
  const float ua = sigm(F.bn2.g[c] * (h2[(size_t)win * C + c] - mean2[0][c]) * rstd2[0][c] + F.bn2.b[c]);
  const float um = sigm(F.bn2.g[c] * (h2[((size_t)B + win) * C + c] - mean2[1][c]) * rstd2[1][c] + F.bn2.b[c]);
  if (sa) { *sa = ua; *smx = um; }
  return sigm(ua + um);
}

// ---------------------------------------------------------------------------------
// k_dn_out: cell output.  dam = 0: out = BN(a).  dam = 1: x = BN(a); out = Sattn * Cattn * x.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_dn_out(const float* __restrict__ a, const double* sumsa, Bn bn, Geo G,
                                               const float* __restrict__ h2, const double* sums2, Fcn F,
                                               const float* __restrict__ saw, const float* __restrict__ sab,
                                               float* __restrict__ out, int B, int training, int upd_bn) {
  extern __shared__ float sm[];
  float* xs = sm;                   // c x lout
  float* ss = xs + G.c * G.lout;    // lout
  __shared__ float sc[MAXC], sh[MAXC], ca[MAXC], mean2[2][MAXC], rstd2[2][MAXC];
  const int tid = threadIdx.x, C = G.c;
  if (tid < C) {
    float m, r;
    bn_stat(sumsa, C, tid, (double)B * G.lout, training, bn.rm, bn.rv, m, r);
    sc[tid] = bn.g[tid] * r; sh[tid] = bn.b[tid] - m * bn.g[tid] * r;
    if (training && upd_bn && blockIdx.x == 0) bn_running(sumsa, C, tid, (double)B * G.lout, bn.rm, bn.rv);
    if (G.dam) {
      for (int p = 0; p < 2; ++p) bn_stat(sums2 + p * 2 * C, C, tid, B, training, F.bn2.rm, F.bn2.rv, mean2[p][tid], rstd2[p][tid]);
      if (training && blockIdx.x == 0)
        for (int p = 0; p < 2; ++p) bn_running(sums2 + p * 2 * C, C, tid, B, F.bn2.rm, F.bn2.rv);
    }
  }
  const int nz = C * G.lout;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    load_bn(a + (size_t)win * nz, xs, sc, sh, C, G.lout, tid);
    if (G.dam && tid < C) ca[tid] = dam_cattn(h2, B, win, C, tid, F, mean2, rstd2, nullptr, nullptr);
    __syncthreads();
    if (!G.dam) {
      for (int o = tid; o < nz; o += NT) out[(size_t)win * nz + o] = xs[o];
      continue;
    }
    for (int l = tid; l < G.lout; l += NT) {
      float s = 0.f, mx = -INFINITY;
      for (int c = 0; c < C; ++c) { const float v = xs[c * G.lout + l]; s += v; mx = fmaxf(mx, v); }
      ss[l] = sigm(fmaf(saw[0], s / C, fmaf(saw[1], mx, sab[0])));
    }
    __syncthreads();
    for (int o = tid; o < nz; o += NT) { const int c = o / G.lout, l = o - c * G.lout; out[(size_t)win * nz + o] = ss[l] * (ca[c] * xs[o]); }
  }
}

// =================================================================================
// backward
// =================================================================================
// per-workgroup gradient accumulators in LDS, flushed with one atomic per entry
RAL_DEV void flush(float* g, const float* acc, int n, int tid) {
  for (int i = tid; i < n; i += NT) if (acc[i] != 0.f) atomicAdd(g + i, acc[i]);
}

// ---------------------------------------------------------------------------------
// k_dn_dam_b: gradient through out = S * Ca * x of a DAM cell.  In: dout, a (-> x), h2 (-> Ca and the inner sigmoids).
// Out: dxp = dout S Ca + ds w0 / C + [c = argmax_c x] ds w1   (the two pooled paths are added by k_dn_bn_b),
//      dy2 (2, B, C) = gradient at the BN2 outputs of the two fcn paths, their BatchNorm-backward sums, dL/d convsa.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_dn_dam_b(const float* dout /* may alias dxp */, const float* __restrict__ a,
                                                 const double* sumsa, Bn bn, Geo G, const float* __restrict__ h2,
                                                 const double* sums2, Fcn F, const float* __restrict__ saw,
                                                 const float* __restrict__ sab, float* dsaw, float* dsab,
                                                 float* dxp, float* __restrict__ dy2, double* t2, int B) {
  extern __shared__ float sm[];
  float* xs = sm;                    // c x lout : x
  float* ts = xs + G.c * G.lout;     // c x lout : dout
  float* ss = ts + G.c * G.lout;     // lout : S
  float* dsl = ss + G.lout;          // lout : ds
  int* am = reinterpret_cast<int*>(dsl + G.lout);   // lout : argmax over channels
  __shared__ float sc[MAXC], sh[MAXC], ca[MAXC], ua[MAXC], um[MAXC], du[MAXC], mean2[2][MAXC], rstd2[2][MAXC], gacc[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = G.c, nz = C * G.lout;
  if (tid < C) {
    float m, r;
    bn_stat(sumsa, C, tid, (double)B * G.lout, 1, nullptr, nullptr, m, r);
    sc[tid] = bn.g[tid] * r; sh[tid] = bn.b[tid] - m * bn.g[tid] * r;
    for (int p = 0; p < 2; ++p) bn_stat(sums2 + p * 2 * C, C, tid, B, 1, nullptr, nullptr, mean2[p][tid], rstd2[p][tid]);
  }
  if (tid < 4) gacc[tid] = 0.f;
  double q1 = 0, q2 = 0;             // thread (path, c): sums of dy2 and dy2 * h2hat
  const int pc = tid / C, cc = tid - pc * C;
  const float w0 = saw[0], w1 = saw[1], b = sab[0];
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    load_bn(a + (size_t)win * nz, xs, sc, sh, C, G.lout, tid);
    for (int o = tid; o < nz; o += NT) ts[o] = dout[(size_t)win * nz + o];
    if (tid < C) ca[tid] = dam_cattn(h2, B, win, C, tid, F, mean2, rstd2, &ua[tid], &um[tid]);
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int l = tid; l < G.lout; l += NT) {
      float s = 0.f, mx = -INFINITY, dS = 0.f; int im = 0;
      for (int c = 0; c < C; ++c) {
        const float v = xs[c * G.lout + l];
        s += v; if (v > mx) { mx = v; im = c; }
        dS = fmaf(ts[c * G.lout + l] * v, ca[c], dS);
      }
      const float S = sigm(fmaf(w0, s / C, fmaf(w1, mx, b)));
      const float d = dS * S * (1.f - S);
      ss[l] = S; dsl[l] = d; am[l] = im;
      g0 = fmaf(d, s / C, g0); g1 = fmaf(d, mx, g1); g2 += d;
    }
    g0 = wave_sum(g0); g1 = wave_sum(g1); g2 = wave_sum(g2);
    if (lane == 0) { atomicAdd(&gacc[0], g0); atomicAdd(&gacc[1], g1); atomicAdd(&gacc[2], g2); }
    __syncthreads();
    for (int c = wave; c < C; c += NT / 64) {       // dCa[c] = sum_l dout x S
      float s = 0.f;
      for (int l = lane; l < G.lout; l += 64) s = fmaf(ts[c * G.lout + l] * xs[c * G.lout + l], ss[l], s);
      s = wave_sum(s);
      if (lane == 0) du[c] = s * ca[c] * (1.f - ca[c]);
    }
    for (int o = tid; o < nz; o += NT) {
      const int c = o / G.lout, l = o - c * G.lout;
      dxp[(size_t)win * nz + o] = ts[o] * ss[l] * ca[c] + dsl[l] * (w0 / C + (am[l] == c ? w1 : 0.f));
    }
    __syncthreads();
    if (pc < 2) {
      const float u = pc ? um[cc] : ua[cc];
      const float d = du[cc] * u * (1.f - u);
      dy2[((size_t)pc * B + win) * C + cc] = d;
      const float hh = (h2[((size_t)pc * B + win) * C + cc] - mean2[pc][cc]) * rstd2[pc][cc];
      q1 += d; q2 += (double)d * hh;
    }
  }
  __syncthreads();
  if (tid == 0) { atomicAdd(dsaw, gacc[0]); atomicAdd(dsaw + 1, gacc[1]); atomicAdd(dsab, gacc[2]); }
  if (pc < 2) { atomicAdd(t2 + pc * 2 * C + cc, q1); atomicAdd(t2 + pc * 2 * C + C + cc, q2); }
}

// ---------------------------------------------------------------------------------
// k_dn_fcn_bmid: BN2 backward + second Linear backward + ReLU / BN1 mask, for `np` paths.
//   dh2 = g2 rstd2 (dy2 - mean(dy2) - h2hat mean(dy2 h2hat));  dW3 += dh2 (x) r;  db3 += dh2;  dr = W3^T dh2;
//   dy1 = dr [BN1(h1) > 0]  (stored, (np, B, dh)) and its BatchNorm-backward sums;  workgroup 0: dg2, dbeta2 from the sums.
// ---------------------------------------------------------------------------------
// (one wave per window, four windows in flight, as k_dn_fcn_mid: the window loop has no block barrier; a wave keeps its share
// of dW3 in registers - entry e = lane + 64 k of the (dout x dh) matrix - and the four waves meet in the LDS accumulator)
__global__ __launch_bounds__(NT) void k_dn_fcn_bmid(const float* __restrict__ dy2, const double* t2, const float* __restrict__ h2,
                                                    const double* sums2, const float* __restrict__ h1, const double* sums1,
                                                    Fcn F, float* __restrict__ dy1, double* t1, int np, int B) {
  extern __shared__ float sm[];
  float* aw3 = sm;                          // dout x dh accumulator
  float* ab3 = aw3 + F.dout * F.dh;         // dout
  __shared__ float w3s[32 * DLD], mean1[2][MAXC], rstd1[2][MAXC], mean2[2][MAXC], rstd2[2][MAXC], m1[2][MAXC], m2[2][MAXC];
  __shared__ float g1s[MAXC], b1s[MAXC], g2s[MAXC];
  __shared__ float rs[NT / 64][DWIN][2 * MAXC], dh2w[NT / 64][DWIN][MAXC];     // per wave and window: relu(BN1(h1)); dh2 (path p at 32 p)
  __shared__ double red[NT / 64][2][MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < F.dout * F.dh + F.dout; i += NT) aw3[i] = 0.f;
  for (int i = tid; i < np * F.dh; i += NT) {
    const int p = i / F.dh, j = i - p * F.dh;
    bn_stat(sums1 + p * 2 * F.dh, F.dh, j, B, 1, nullptr, nullptr, mean1[p][j], rstd1[p][j]);
  }
  for (int i = tid; i < np * F.dout; i += NT) {
    const int p = i / F.dout, c = i - p * F.dout;
    bn_stat(sums2 + p * 2 * F.dout, F.dout, c, B, 1, nullptr, nullptr, mean2[p][c], rstd2[p][c]);
    m1[p][c] = (float)(t2[p * 2 * F.dout + c] / B); m2[p][c] = (float)(t2[p * 2 * F.dout + F.dout + c] / B);
    if (blockIdx.x == 0) { atomicAdd(F.bn2.db + c, (float)t2[p * 2 * F.dout + c]); atomicAdd(F.bn2.dg + c, (float)t2[p * 2 * F.dout + F.dout + c]); }
  }
  if (tid < F.dh) { g1s[tid] = F.bn1.g[tid]; b1s[tid] = F.bn1.b[tid]; }
  if (tid < F.dout) g2s[tid] = F.bn2.g[tid];
  for (int i = tid; i < F.dout * F.dh; i += NT) w3s[(i / F.dh) * DLD + i % F.dh] = F.w3[i];
  __syncthreads();
  const int nin = np * F.dh, nout = np * F.dout, nent = (F.dout * F.dh + 63) / 64;
  const int pi = lane / F.dh, ji = lane - pi * F.dh;          // this lane's hidden element (lane < nin)
  const int pc = lane / F.dout, cc = lane - pc * F.dout;      // this lane's output column (lane < nout)
  float accw[32];                                             // dW3 entries lane + 64 k (dout dh <= 2048)
#pragma unroll
  for (int k = 0; k < 32; ++k) accw[k] = 0.f;
  float accb = 0.f;
  double q1 = 0, q2 = 0;       // lane (path, j): sums of dy1, dy1 * h1hat
  const int nwv = gridDim.x * (NT / 64);
  for (int w0 = (blockIdx.x * (NT / 64) + wave) * DWIN; w0 < B; w0 += nwv * DWIN) {
    float v1[DWIN], v2[DWIN], vd[DWIN];
#pragma unroll
    for (int q = 0; q < DWIN; ++q) {               // all twelve loads first (clamped: no lane-predicated loads)
      const int win = w0 + q < B ? w0 + q : B - 1;
      v1[q] = h1[((size_t)(lane < nin ? pi : 0) * B + win) * F.dh + (lane < nin ? ji : 0)];
      const size_t o = ((size_t)(lane < nout ? pc : 0) * B + win) * F.dout + (lane < nout ? cc : 0);
      v2[q] = h2[o]; vd[q] = dy2[o];
    }
    float hh1[DWIN]; bool pos[DWIN];
#pragma unroll
    for (int q = 0; q < DWIN; ++q) {
      if (lane < nin) {
        hh1[q] = (v1[q] - mean1[pi][ji]) * rstd1[pi][ji];
        const float bnv = g1s[ji] * (v1[q] - mean1[pi][ji]) * rstd1[pi][ji] + b1s[ji];
        pos[q] = bnv > 0.f;
        rs[wave][q][pi * MAXC + ji] = fmaxf(bnv, 0.f);
      }
      if (lane < nout) {
        const float hh = (v2[q] - mean2[pc][cc]) * rstd2[pc][cc];
        const float d2 = (w0 + q < B) ? g2s[cc] * rstd2[pc][cc] * (vd[q] - m1[pc][cc] - hh * m2[pc][cc]) : 0.f;   // (past the end: no contribution)
        dh2w[wave][q][pc * 32 + cc] = d2;
        accb += d2;
      }
    }
    __builtin_amdgcn_wave_barrier();               // (the wave's own LDS writes are ordered before its reads)
#pragma unroll
    for (int q = 0; q < DWIN; ++q) {
      if (w0 + q >= B) break;
      if (lane < nin) {
        float dr = 0.f;
        const float* d2 = dh2w[wave][q] + pi * 32;
        for (int c = 0; c < F.dout; ++c) dr = fmaf(w3s[c * DLD + ji], d2[c], dr);
        const float d = pos[q] ? dr : 0.f;
        dy1[((size_t)pi * B + w0 + q) * F.dh + ji] = d;
        q1 += d; q2 += (double)d * hh1[q];
      }
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const int e = lane + 64 * k;
        if (k < nent && e < F.dout * F.dh) {
          const int c = e / F.dh, j = e - c * F.dh;
          float t = dh2w[wave][q][c] * rs[wave][q][j];
          if (np > 1) t = fmaf(dh2w[wave][q][32 + c], rs[wave][q][MAXC + j], t);
          accw[k] += t;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the four waves meet in the LDS accumulators, then one atomic per entry and workgroup
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const int e = lane + 64 * k;
    if (k < nent && e < F.dout * F.dh) atomicAdd(aw3 + e, accw[k]);
  }
  if (lane < nout) atomicAdd(ab3 + cc, accb);
  if (lane < nin) { red[wave][0][lane] = q1; red[wave][1][lane] = q2; }
  __syncthreads();
  flush(F.dw3, aw3, F.dout * F.dh, tid);
  flush(F.db3, ab3, F.dout, tid);
  if (tid < nin) {
    double a1 = 0, a2 = 0;
    for (int w = 0; w < NT / 64; ++w) { a1 += red[w][0][tid]; a2 += red[w][1][tid]; }
    const int p = tid / F.dh, j = tid - p * F.dh;
    atomicAdd(t1 + p * 2 * F.dh + j, a1); atomicAdd(t1 + p * 2 * F.dh + F.dh + j, a2);
  }
}

// BN1 backward + first Linear backward of one window and path: dh1 -> LDS accumulators, returns d input[i] for thread i
RAL_DEV void fcn_first_bwd(const Fcn& F, const float* dh1s /*LDS dh*/, const float* ins /*LDS din*/, float* aw0, float* ab0, int tid) {
  for (int i = tid; i < F.dh * F.din; i += NT) { const int j = i / F.din, k = i - j * F.din; aw0[i] = fmaf(dh1s[j], ins[k], aw0[i]); }
  if (tid < F.dh) ab0[tid] += dh1s[tid];
}

// ---------------------------------------------------------------------------------
// k_dn_bn_b: gradient at the cell's BatchNorm output, dxo, and the sums its backward needs.
//   dam = 0: dxo = g0 (+ g1)           (the gradients from the two consumers of the cell output)
//   dam = 1: dxo = dxp + dgap / L + [l = argmax_l x] dgmp with (dgap, dgmp) = first-Linear backward of the two fcn paths
//            (BN1 backward from dy1 and its sums; dW0, db0, dg1, dbeta1 accumulated here)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_dn_bn_b(const float* g0 /* may alias dxo */, const float* __restrict__ g1,
                                                const float* __restrict__ a, const double* sumsa, Bn bn, Geo G,
                                                const float* __restrict__ dy1, const double* t1, const float* __restrict__ h1,
                                                const double* sums1, const float* __restrict__ pool, Fcn F,
                                                float* dxo, double* tbn, int B) {
  extern __shared__ float sm[];
  float* xs = sm;                        // c x lout : a, then ahat
  float* aw0 = xs + G.c * G.lout;        // dh x din
  float* ab0 = aw0 + (G.dam ? F.dh * F.din : 0);
  __shared__ float mean[MAXC], rstd[MAXC], mean1[2][MAXC], rstd1[2][MAXC], m1[2][MAXC], m2[2][MAXC];
  __shared__ float dh1s[2][MAXC], ins[2][MAXC], dpool[2][MAXC];
  __shared__ double acc1[MAXC], acc2[MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = G.c, nz = C * G.lout;
  if (tid < C) {
    bn_stat(sumsa, C, tid, (double)B * G.lout, 1, nullptr, nullptr, mean[tid], rstd[tid]);
    acc1[tid] = 0; acc2[tid] = 0;
  }
  if (G.dam) {
    for (int i = tid; i < F.dh * F.din + F.dh; i += NT) aw0[i] = 0.f;
    for (int i = tid; i < 2 * C; i += NT) {
      const int p = i / C, j = i - p * C;
      bn_stat(sums1 + p * 2 * C, C, j, B, 1, nullptr, nullptr, mean1[p][j], rstd1[p][j]);
      m1[p][j] = (float)(t1[p * 2 * C + j] / B); m2[p][j] = (float)(t1[p * 2 * C + C + j] / B);
      if (blockIdx.x == 0) { atomicAdd(F.bn1.db + j, (float)t1[p * 2 * C + j]); atomicAdd(F.bn1.dg + j, (float)t1[p * 2 * C + C + j]); }
    }
  }
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    if ((G.lout & 3) == 0) { for (int i = tid; i < (nz >> 2); i += NT) reinterpret_cast<float4*>(xs)[i] = reinterpret_cast<const float4*>(a + (size_t)win * nz)[i]; }
    else for (int o = tid; o < nz; o += NT) xs[o] = a[(size_t)win * nz + o];
    if (G.dam)
      for (int i = tid; i < 2 * C; i += NT) {
        const int p = i / C, j = i - p * C;
        const size_t o = ((size_t)p * B + win) * C + j;
        const float hh = (h1[o] - mean1[p][j]) * rstd1[p][j];
        dh1s[p][j] = F.bn1.g[j] * rstd1[p][j] * (dy1[o] - m1[p][j] - hh * m2[p][j]);
        ins[p][j] = pool[o];
      }
    __syncthreads();
    if (G.dam) {
      for (int i = tid; i < C * C; i += NT) {
        const int j = i / C, k = i - j * C;
        aw0[i] += dh1s[0][j] * ins[0][k] + dh1s[1][j] * ins[1][k];
      }
      if (tid < C) ab0[tid] += dh1s[0][tid] + dh1s[1][tid];
      for (int i = tid; i < 2 * C; i += NT) {
        const int p = i / C, k = i - p * C;
        float d = 0.f;
        for (int j = 0; j < C; ++j) d = fmaf(F.w0[j * C + k], dh1s[p][j], d);
        dpool[p][k] = d;
      }
      __syncthreads();
    }
    // fast path: a thread per 16-byte item of the window, a channel row = q4 consecutive lanes of one wave (segmented
    // arg-max and sums by lane exchanges); otherwise a wave per channel row
    const int q4 = G.lout >> 2, n4w = nz >> 2;
    const bool rowfast = (G.lout & 3) == 0 && q4 <= 64 && (q4 & (q4 - 1)) == 0 && n4w % 64 == 0;
    if (rowfast) {
      const float4* g04 = reinterpret_cast<const float4*>(g0 + (size_t)win * nz);
      const float4* g14 = g1 ? reinterpret_cast<const float4*>(g1 + (size_t)win * nz) : nullptr;
      float4* d4 = reinterpret_cast<float4*>(dxo + (size_t)win * nz);
      for (int i0 = 0; i0 < n4w; i0 += NT) {
        const int i = i0 + tid;                    // (n4w is a multiple of 64: whole waves are in or out)
        if (i < n4w) {
          const int c = i / q4, l0 = (i - c * q4) << 2;
          float4 d = g04[i];
          if (g14) d = f4add(d, g14[i]);
          const float4 xv = reinterpret_cast<const float4*>(xs)[i];
          if (G.dam) {
            const bool up = bn.g[c] * rstd[c] >= 0.f;
            const float x4[4] = {up ? xv.x : -xv.x, up ? xv.y : -xv.y, up ? xv.z : -xv.z, up ? xv.w : -xv.w};
            float best = x4[0]; int bi = l0;
#pragma unroll
            for (int e = 1; e < 4; ++e) if (x4[e] > best) { best = x4[e]; bi = l0 + e; }
            for (int sft = q4 >> 1; sft > 0; sft >>= 1) {
              const float ob = __shfl_xor(best, sft); const int oi = __shfl_xor(bi, sft);
              if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const float dm = dpool[0][c] / G.lout, dx_ = dpool[1][c];
            d.x += dm + (l0 == bi ? dx_ : 0.f); d.y += dm + (l0 + 1 == bi ? dx_ : 0.f);
            d.z += dm + (l0 + 2 == bi ? dx_ : 0.f); d.w += dm + (l0 + 3 == bi ? dx_ : 0.f);
          }
          d4[i] = d;
          const float mu = mean[c], rs_ = rstd[c];
          float s1 = hsum4(d);
          float s2 = fmaf(d.x, (xv.x - mu) * rs_, fmaf(d.y, (xv.y - mu) * rs_, fmaf(d.z, (xv.z - mu) * rs_, d.w * ((xv.w - mu) * rs_))));
          s1 = seg_sum(s1, q4); s2 = seg_sum(s2, q4);
          if ((tid & (q4 - 1)) == 0) { acc1[c] += s1; acc2[c] += s2; }      // (channel c's row has one leader lane per window)
        }
      }
      continue;
    }
    for (int c = wave; c < C; c += NT / 64) {
      // x = sc a + sh is increasing in a when g * rstd > 0: the window maximum of x sits at the max (min) of a
      int arg = -1;
      if (G.dam) {
        const bool up = bn.g[c] * rstd[c] >= 0.f;
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int l = lane; l < G.lout; l += 64) { const float v = up ? xs[c * G.lout + l] : -xs[c * G.lout + l]; if (v > best) { best = v; bi = l; } }
        for (int s = 32; s > 0; s >>= 1) {
          const float ob = __shfl_xor(best, s); const int oi = __shfl_xor(bi, s);
          if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        arg = bi;
      }
      float s1 = 0.f, s2 = 0.f;
      for (int l = lane; l < G.lout; l += 64) {
        const size_t o = (size_t)win * nz + c * G.lout + l;
        float d = g0[o];
        if (g1) d += g1[o];
        if (G.dam) d += dpool[0][c] / G.lout + (l == arg ? dpool[1][c] : 0.f);
        dxo[o] = d;
        const float ah = (xs[c * G.lout + l] - mean[c]) * rstd[c];
        s1 += d; s2 = fmaf(d, ah, s2);
      }
      s1 = wave_sum(s1); s2 = wave_sum(s2);
      if (lane == 0) { acc1[c] += s1; acc2[c] += s2; }
    }
  }
  __syncthreads();
  if (tid < C) { atomicAdd(tbn + tid, acc1[tid]); atomicAdd(tbn + C + tid, acc2[tid]); }
  if (G.dam) { flush(F.dw0, aw0, C * C, tid); flush(F.db0, ab0, C, tid); }
}

// ---------------------------------------------------------------------------------
// k_dn_act_b: BatchNorm backward of the cell output and the APReLU mix.
//   da = g rstd (dxo - mean(dxo) - ahat mean(dxo ahat)) (stored in place of dxo);  dalpha[c] = sum_l da min(z, 0);
//   dy2 = dalpha alpha (1 - alpha) (B, C) and its BatchNorm-backward sums;  workgroup 0: dg, dbeta of the cell's BatchNorm.
// ---------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(NT) void k_dn_act_b(float* __restrict__ dxo, const float* __restrict__ a, const float* __restrict__ z,
                                                 const double* sumsa, const double* tbn, Bn bn, Geo G,
                                                 const float* __restrict__ h2, const double* sums2, Fcn F,
                                                 float* __restrict__ dy2, double* t2, int B) {
  __shared__ float mean[MAXC], rstd[MAXC], k1[MAXC], k2[MAXC], grs[MAXC], mean2[MAXC], rstd2[MAXC], g2s[MAXC], b2s[MAXC], dal[EW][MAXC];
  const int tid = threadIdx.x, C = G.c, nz = C * G.lout, n4w = nz / V, q4 = G.lout / V;
  const double cnt = (double)B * G.lout;
  if (tid < C) {
    bn_stat(sumsa, C, tid, cnt, 1, nullptr, nullptr, mean[tid], rstd[tid]);
    k1[tid] = (float)(tbn[tid] / cnt); k2[tid] = (float)(tbn[C + tid] / cnt);
    grs[tid] = bn.g[tid] * rstd[tid];
    bn_stat(sums2, C, tid, B, 1, nullptr, nullptr, mean2[tid], rstd2[tid]);
    g2s[tid] = F.bn2.g[tid]; b2s[tid] = F.bn2.b[tid];
    if (blockIdx.x == 0) { atomicAdd(bn.db + tid, (float)tbn[tid]); atomicAdd(bn.dg + tid, (float)tbn[C + tid]); }
  }
  const int segw = ((q4 & (q4 - 1)) == 0 && (EW * n4w) % 64 == 0) ? (q4 < 64 ? q4 : 64) : 1;
  double q1 = 0, q2 = 0;                       // thread (window slot, c): sums of dy2, dy2 * h2hat of channel c
  for (int w0 = blockIdx.x * EW; w0 < B; w0 += gridDim.x * EW) {
    const int nwin = (B - w0) < EW ? (B - w0) : EW, n4 = nwin * n4w;
    float* d4 = dxo + (size_t)w0 * nz;
    const float* a4 = a + (size_t)w0 * nz;
    const float* z4 = z + (size_t)w0 * nz;
    __syncthreads();
    if (tid < EW * C) dal[tid / C][tid % C] = 0.f;
    // this thread's (window, channel) of the epilogue: its h2 value is requested with the pass's other loads
    const int eq = tid / C, ec = tid - eq * C;
    const float h2v = h2[(size_t)((tid < EW * C && w0 + eq < B) ? w0 + eq : w0) * C + (tid < EW * C ? ec : 0)];
    __syncthreads();
    for (int i0 = 0; i0 < n4; i0 += 2 * NT) {
      float4 dv[2], av[2], zv[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) { const int i = i0 + k * NT + tid, j = i < n4 ? i : 0; dv[k] = ldv<V>(d4, j); av[k] = ldv<V>(a4, j); zv[k] = ldv<V>(z4, j); }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = i0 + k * NT + tid;
        float sv = 0.f;
        int q = 0, c = 0;
        if (i < n4) {
          q = i / n4w; const int e = i - q * n4w;
          c = e / q4;
          const float mu = mean[c], rs = rstd[c], gr = grs[c], kk1 = k1[c], kk2 = k2[c];
          float4 d;
          d.x = gr * (dv[k].x - kk1 - (av[k].x - mu) * rs * kk2); d.y = gr * (dv[k].y - kk1 - (av[k].y - mu) * rs * kk2);
          d.z = gr * (dv[k].z - kk1 - (av[k].z - mu) * rs * kk2); d.w = gr * (dv[k].w - kk1 - (av[k].w - mu) * rs * kk2);
          stv<V>(d4, i, d);
          sv = fmaf(d.x, fminf(zv[k].x, 0.f), fmaf(d.y, fminf(zv[k].y, 0.f), fmaf(d.z, fminf(zv[k].z, 0.f), d.w * fminf(zv[k].w, 0.f))));
        }
        if (segw > 1) { sv = seg_sum(sv, segw); if ((tid & (segw - 1)) == 0 && i < n4) atomicAdd(&dal[q][c], sv); }
        else if (i < n4) atomicAdd(&dal[q][c], sv);
      }
    }
    __syncthreads();
    if (tid < EW * C && w0 + eq < B) {
      const float hh = (h2v - mean2[ec]) * rstd2[ec];
      const float al = sigm(fmaf(g2s[ec], hh, b2s[ec]));
      const float d = dal[eq][ec] * al * (1.f - al);
      dy2[(size_t)(w0 + eq) * C + ec] = d;
      q1 += d; q2 += (double)d * hh;
    }
  }
  if (tid < EW * C) { atomicAdd(t2 + ec_of(tid, C), q1); atomicAdd(t2 + C + ec_of(tid, C), q2); }
}

// ---------------------------------------------------------------------------------
// k_dn_conv_b: first Linear + BN1 backward of the APReLU fcn, dz, and the convolution backward.
//   dh1 = BN1 backward(dy1); dW0 += dh1 (x) desc; ddesc = W0^T dh1 = [dP; dN];
//   dz = da (z > 0 ? 1 : alpha) + (z > 0 ? dP : dN) / lout;  dW, db of the conv;  din -> gi0 (and gi1) unless null
// ---------------------------------------------------------------------------------
template <int K, bool TR>
__global__ __launch_bounds__(NT) void k_dn_conv_b(const float* __restrict__ da, const float* __restrict__ z,
                                                  const float* __restrict__ in0, const float* __restrict__ in1,
                                                  const float* __restrict__ w, float* dw, float* dbias, Geo G,
                                                  const float* __restrict__ dy1, const double* t1, const float* __restrict__ h1,
                                                  const double* sums1, const float* __restrict__ h2, const double* sums2,
                                                  const float* __restrict__ desc, Fcn F, float* __restrict__ gi0,
                                                  float* __restrict__ gi1, int B) {
  extern __shared__ float sm[];
  const int LP = G.lin + 2 * CH, LPO = G.lout + 2 * CH;   // both tiles carry a zero halo: no tap needs a bounds check
  float* xs = sm;                          // cin x LP
  float* dzs = xs + G.cin * LP;            // c x LPO
  float* aw = dzs + G.c * LPO;             // conv weight accumulator (c cin k)
  float* ab = aw + G.c * G.cin * G.k;      // c
  float* aw0 = ab + G.c;                   // dh x din
  float* ab0 = aw0 + F.dh * F.din;         // dh
  __shared__ float mean1[MAXC], rstd1[MAXC], m1[MAXC], m2[MAXC], mean2[MAXC], rstd2[MAXC], dh1s[MAXC], dsc[MAXC], al[MAXC], dd[MAXC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, C = G.c;
  const int nin = G.cin * G.lin, nz = C * G.lout, nw = C * G.cin * K;
  for (int i = tid; i < nw + C + F.dh * F.din + F.dh; i += NT) aw[i] = 0.f;
  for (int i = tid; i < G.cin * 2 * CH; i += NT) { const int c = i / (2 * CH), h = i - c * 2 * CH; xs[c * LP + (h < CH ? h : G.lin + h)] = 0.f; }
  for (int i = tid; i < C * 2 * CH; i += NT) { const int c = i / (2 * CH), h = i - c * 2 * CH; dzs[c * LPO + (h < CH ? h : G.lout + h)] = 0.f; }
  if (tid < F.dh) {
    bn_stat(sums1, F.dh, tid, B, 1, nullptr, nullptr, mean1[tid], rstd1[tid]);
    m1[tid] = (float)(t1[tid] / B); m2[tid] = (float)(t1[F.dh + tid] / B);
    if (blockIdx.x == 0) { atomicAdd(F.bn1.db + tid, (float)t1[tid]); atomicAdd(F.bn1.dg + tid, (float)t1[F.dh + tid]); }
  }
  if (tid < C) bn_stat(sums2, C, tid, B, 1, nullptr, nullptr, mean2[tid], rstd2[tid]);
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < nin; i += NT) {
      const int c = i / G.lin, l = i - c * G.lin;
      xs[c * LP + CH + l] = in0[(size_t)win * nin + i] + (in1 ? in1[(size_t)win * nin + i] : 0.f);
    }
    if (tid < F.dh) {
      const size_t o = (size_t)win * F.dh + tid;
      const float hh = (h1[o] - mean1[tid]) * rstd1[tid];
      dh1s[tid] = F.bn1.g[tid] * rstd1[tid] * (dy1[o] - m1[tid] - hh * m2[tid]);
    }
    if (tid < F.din) dsc[tid] = desc[(size_t)win * F.din + tid];
    if (tid < C) al[tid] = sigm(F.bn2.g[tid] * (h2[(size_t)win * C + tid] - mean2[tid]) * rstd2[tid] + F.bn2.b[tid]);
    __syncthreads();
    fcn_first_bwd(F, dh1s, dsc, aw0, ab0, tid);
    if (tid < F.din) {
      float d = 0.f;
      for (int j = 0; j < F.dh; ++j) d = fmaf(F.w0[j * F.din + tid], dh1s[j], d);
      dd[tid] = d / G.lout;
    }
    __syncthreads();
    for (int o = tid; o < nz; o += NT) {
      const int c = o / G.lout, l = o - c * G.lout;
      const float v = z[(size_t)win * nz + o], d = da[(size_t)win * nz + o];
      dzs[c * LPO + CH + l] = v > 0.f ? d + dd[c] : fmaf(d, al[c], dd[C + c]);
    }
    __syncthreads();
    // bias and weight gradients: thread-owned accumulator entries
    for (int c = wave; c < C; c += NT / 64) {
      float s = 0.f;
      for (int l = lane; l < G.lout; l += 64) s += dzs[c * LPO + CH + l];
      s = wave_sum(s);
      if (lane == 0) ab[c] += s;
    }
    for (int i = tid; i < nw; i += NT) {
      float s = 0.f;
      if constexpr (!TR) {   // w[co][ci][k]: sum_l dz[co][l] in[ci][2 l + k - p]
        const int co = i / (G.cin * K), r = i - co * G.cin * K, ci = r / K, k = r - ci * K;
        const float* dr = dzs + co * LPO + CH; const float* xr = xs + ci * LP + CH + k - G.p;
        for (int l = 0; l < G.lout; ++l) s = fmaf(dr[l], xr[2 * l], s);
      } else {               // w[ci][co][k]: sum_s in[ci][s] dz[co][2 s + k - p]
        const int ci = i / (C * K), r = i - ci * C * K, co = r / K, k = r - co * K;
        const float* dr = dzs + co * LPO + CH + k - G.p; const float* xr = xs + ci * LP + CH;
        for (int sx = 0; sx < G.lin; ++sx) s = fmaf(xr[sx], dr[2 * sx], s);
      }
      aw[i] += s;
    }
    // input gradient
    if (gi0) {
      for (int o = tid; o < nin; o += NT) {
        const int ci = o / G.lin, sx = o - ci * G.lin;
        float s = 0.f;
        if constexpr (!TR) {   // din[ci][s] = sum_co sum_{k = s + p - 2 l} w[co][ci][k] dz[co][l]: the taps of s's parity
          const int k0 = (sx + G.p) & 1, l0 = (sx + G.p - k0) >> 1;
          for (int co = 0; co < C; ++co) {
            const float* wr = w + ((size_t)co * G.cin + ci) * K + k0; const float* dr = dzs + co * LPO + CH + l0;
#pragma unroll
            for (int kk = 0; kk < (K + 1) / 2; ++kk) if (k0 + 2 * kk < K) s = fmaf(wr[2 * kk], dr[-kk], s);
          }
        } else {               // din[ci][s] = sum_co sum_k w[ci][co][k] dz[co][2 s + k - p]
          for (int co = 0; co < C; ++co) {
            const float* wr = w + ((size_t)ci * C + co) * K; const float* dr = dzs + co * LPO + CH + 2 * sx - G.p;
#pragma unroll
            for (int k = 0; k < K; ++k) s = fmaf(wr[k], dr[k], s);
          }
        }
        gi0[(size_t)win * nin + o] = s;
        if (gi1) gi1[(size_t)win * nin + o] = s;
      }
    }
  }
  __syncthreads();
  flush(dw, aw, nw, tid);
  flush(dbias, ab, C, tid);
  flush(F.dw0, aw0, F.dh * F.din, tid);
  flush(F.db0, ab0, F.dh, tid);
}

// =================================================================================
// layout (state_dict contract) and host side
// =================================================================================
struct DEntry { std::string name; int kind; int64_t offset; int ndim; int64_t shape[4]; };
struct BnOff { int64_t w, b, rm, rv; };
struct FcnOff { int64_t w0, b0, w3, b3; BnOff bn1, bn2; int din, dh, dout; };
struct CellOff { Geo g; int64_t cw, cb; FcnOff act; BnOff bn; FcnOff dam; int64_t saw, sab; };
struct DLayout { std::vector<DEntry> e; int64_t nparam = 0, nstate = 0; CellOff cell[8]; };

void dbuild(const ral_config& c, DLayout& Y) {
  auto add = [&](const std::string& name, int kind, int64_t off, std::vector<int64_t> shp) {
    DEntry en; en.name = name; en.kind = kind; en.offset = off; en.ndim = (int)shp.size();
    for (int i = 0; i < 4; ++i) en.shape[i] = i < (int)shp.size() ? shp[i] : 1;
    Y.e.push_back(en);
  };
  auto numel = [](const std::vector<int64_t>& s) { int64_t n = 1; for (auto v : s) n *= v; return n; };
  auto param = [&](const std::string& name, std::vector<int64_t> shp) { const int64_t o = Y.nparam; add(name, RAL_PARAM, o, shp); Y.nparam += (numel(shp) + 3) / 4 * 4; return o; };
  auto state = [&](const std::string& name, int64_t n) { const int64_t o = Y.nstate; add(name, RAL_STATE_F32, o, {n}); Y.nstate += (n + 3) / 4 * 4; return o; };
  auto bn = [&](const std::string& pre, int64_t n) {
    BnOff b; b.w = param(pre + ".weight", {n}); b.b = param(pre + ".bias", {n});
    b.rm = state(pre + ".running_mean", n); b.rv = state(pre + ".running_var", n);
    add(pre + ".num_batches_tracked", RAL_COUNTER_I64, 0, {});
    return b;
  };
  auto fcn = [&](const std::string& pre, int din, int dh, int dout) {
    FcnOff f; f.din = din; f.dh = dh; f.dout = dout;
    f.w0 = param(pre + ".0.weight", {dh, din}); f.b0 = param(pre + ".0.bias", {dh});
    f.bn1 = bn(pre + ".1", dh);
    f.w3 = param(pre + ".3.weight", {dout, dh}); f.b3 = param(pre + ".3.bias", {dout});
    f.bn2 = bn(pre + ".4", dout);
    return f;
  };
  int cin = c.leads;
  for (int i = 0; i < 8; ++i) {
    CellOff& K = Y.cell[i];
    const bool enc = i < 4;
    const int j = enc ? i : i - 4;
    const int C = enc ? ENC_CH[j] : DEC_CH[j];
    K.g.cin = cin; K.g.c = C; K.g.k = enc ? ENC_K[j] : DEC_K[j]; K.g.p = enc ? ENC_P[j] : DEC_P[j]; K.g.tr = enc ? 0 : 1;
    K.g.lin = enc ? c.L >> j : c.L >> (4 - j); K.g.lout = enc ? c.L >> (j + 1) : c.L >> (3 - j); K.g.dam = (!enc && j < 3) ? 1 : 0;
    const std::string pre = enc ? "enc.EncoderList.cell" + std::to_string(j) : "dec.DecoderList." + std::to_string(j);
    const std::string cv = enc ? ".conv" : ".deconv";
    K.cw = enc ? param(pre + cv + ".weight", {C, cin, K.g.k}) : param(pre + cv + ".weight", {cin, C, K.g.k});
    K.cb = param(pre + cv + ".bias", {C});
    K.act = fcn(pre + ".activate.fcn", 2 * C, 2 * C, C);
    K.bn = bn(pre + ".bn", C);
    if (K.g.dam) {
      K.dam = fcn(pre + ".dam.fcn1", C, C, C);
      // fcn2 is the same module list as fcn1 (DAM.py:122-131): alias entries at the same offsets
      const size_t first = Y.e.size() - 14;
      for (size_t q = first; q < first + 14; ++q) {
        DEntry en = Y.e[q];
        en.name.replace(en.name.find(".dam.fcn1."), 10, ".dam.fcn2.");
        Y.e.push_back(en);
      }
      K.saw = param(pre + ".dam.convsa.weight", {1, 2, 1}); K.sab = param(pre + ".dam.convsa.bias", {1});
    }
    cin = C;
  }
}
}  // namespace

struct DanetModel {
  DanetPublic pub;
  DLayout lay;
  char* slab = nullptr;
  float *z[8], *a[8], *out[8], *desc[8], *h1[8], *h2[8], *pool[8], *dh1[8], *dh2[8];   // forward tensors per cell
  float *g[8], *gs[4], *dy2[8], *dy1[8], *ddy2[8], *ddy1[8];                           // backward tensors
  double* sums = nullptr;            // 8 x S_CELL
  const float* last_x = nullptr; int last_B = 0; bool last_training = false;
};

int danet_check_cfg(const ral_config* c, char* err, size_t cap) {
  if (c->leads != 2) { snprintf(err, cap, "DANet: the last decoder cell has 2 output channels, so leads must be 2 (got %d)", c->leads); return -1; }
  if (c->L < 32 || c->L % 16 != 0 || c->L > 2048) { snprintf(err, cap, "DANet: L must be a multiple of 16 in [32, 2048], got %d", c->L); return -1; }
  if (c->max_batch < 1) { snprintf(err, cap, "DANet: max_batch must be positive"); return -1; }
  return 0;
}
int danet_layout_count(const ral_config* c) { DLayout Y; dbuild(*c, Y); return (int)Y.e.size(); }
int danet_layout_entry(const ral_config* c, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset, int32_t* ndim,
                       int64_t shape[4]) {
  DLayout Y; dbuild(*c, Y);
  if (idx < 0 || idx >= (int)Y.e.size() || (int)Y.e[idx].name.size() + 1 > name_cap) return -1;
  strcpy(name, Y.e[idx].name.c_str());
  *kind = Y.e[idx].kind; *offset = Y.e[idx].offset; *ndim = Y.e[idx].ndim;
  for (int i = 0; i < 4; ++i) shape[i] = Y.e[idx].shape[i];
  return 0;
}
int64_t danet_param_floats(const ral_config* c) { DLayout Y; dbuild(*c, Y); return Y.nparam; }
int64_t danet_state_floats(const ral_config* c) { DLayout Y; dbuild(*c, Y); return Y.nstate; }

static size_t danet_plan(const ral_config& c, DanetModel* m, char* base) {
  DLayout Yl; const DLayout* Y = m ? &m->lay : &Yl;
  if (!m) dbuild(c, Yl);
  size_t cur = 0;
  const size_t B = c.max_batch;
  auto take = [&](size_t floats) -> float* { float* p = base ? reinterpret_cast<float*>(base + cur) : nullptr; cur += (floats * 4 + 255) & ~size_t(255); return p; };
  for (int i = 0; i < 8; ++i) {
    const Geo& G = Y->cell[i].g;
    const size_t n = B * G.c * G.lout;
    float *pz = take(n), *pa = take(n), *po = take(n), *pd = take(B * 2 * G.c), *p1 = take(B * 2 * G.c), *p2 = take(B * G.c);
    float *pp = G.dam ? take(2 * B * G.c) : nullptr, *q1 = G.dam ? take(2 * B * G.c) : nullptr, *q2 = G.dam ? take(2 * B * G.c) : nullptr;
    if (m) { m->z[i] = pz; m->a[i] = pa; m->out[i] = po; m->desc[i] = pd; m->h1[i] = p1; m->h2[i] = p2; m->pool[i] = pp; m->dh1[i] = q1; m->dh2[i] = q2; }
    if (c.train) {
      float *pg = take(n), *pgs = i < 3 ? take(n) : nullptr, *r2 = take(B * G.c), *r1 = take(B * 2 * G.c);
      float *s2 = G.dam ? take(2 * B * G.c) : nullptr, *s1 = G.dam ? take(2 * B * G.c) : nullptr;
      if (m) { m->g[i] = pg; if (i < 4) m->gs[i] = pgs; m->dy2[i] = r2; m->dy1[i] = r1; m->ddy2[i] = s2; m->ddy1[i] = s1; }
    }
  }
  double* ps = reinterpret_cast<double*>(take(8 * S_CELL * 2));
  if (m) m->sums = ps;
  return cur;
}
int64_t danet_workspace_bytes(const ral_config* c) { return (int64_t)danet_plan(*c, nullptr, nullptr); }

DanetModel* danet_create(const ral_config* c, char* err, size_t cap) {
  DanetModel* m = new DanetModel();
  memset(&m->pub, 0, sizeof(m->pub));
  m->pub.cfg = *c;
  dbuild(*c, m->lay);
  m->pub.nparam = m->lay.nparam; m->pub.nstate = m->lay.nstate;
  const size_t bytes = danet_plan(*c, m, nullptr);
  if (hipMalloc(reinterpret_cast<void**>(&m->slab), bytes) != hipSuccess) {
    snprintf(err, cap, "hipMalloc(%zu) failed", bytes);
    delete m;
    return nullptr;
  }
  danet_plan(*c, m, m->slab);
  return m;
}
void danet_destroy(DanetModel* m) { if (m) { if (m->slab) (void)hipFree(m->slab); delete m; } }
DanetPublic* danet_public(DanetModel* m) { return &m->pub; }
int danet_bind(DanetModel* m, float* params, float* grads, float* am, float* av, float* state) {
  m->pub.params = params; m->pub.grads = grads; m->pub.am = am; m->pub.av = av; m->pub.state = state;
  return 0;
}

namespace {
Bn mk_bn(const DanetPublic& P, const BnOff& o) {
  Bn b; b.g = P.params + o.w; b.b = P.params + o.b; b.rm = P.state + o.rm; b.rv = P.state + o.rv;
  b.dg = P.grads ? P.grads + o.w : nullptr; b.db = P.grads ? P.grads + o.b : nullptr;
  return b;
}
Fcn mk_fcn(const DanetPublic& P, const FcnOff& o) {
  Fcn f; f.w0 = P.params + o.w0; f.b0 = P.params + o.b0; f.w3 = P.params + o.w3; f.b3 = P.params + o.b3;
  f.dw0 = P.grads ? P.grads + o.w0 : nullptr; f.db0 = P.grads ? P.grads + o.b0 : nullptr;
  f.dw3 = P.grads ? P.grads + o.w3 : nullptr; f.db3 = P.grads ? P.grads + o.b3 : nullptr;
  f.bn1 = mk_bn(P, o.bn1); f.bn2 = mk_bn(P, o.bn2); f.din = o.din; f.dh = o.dh; f.dout = o.dout;
  return f;
}
template <class K> void set_lds(K kernel, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
}  // namespace

int danet_forward(DanetModel* m, const float* x, float* y, int B, int training, hipStream_t st, char* err, size_t cap) {
  DanetPublic& P = m->pub;
  if (!P.params || !P.state) { snprintf(err, cap, "ral_bind was not called"); return -1; }
  if (B <= 0 || B > P.cfg.max_batch) { snprintf(err, cap, "batch %d outside (0, %d]", B, P.cfg.max_batch); return -1; }
  if (training && B < 2) { snprintf(err, cap, "DANet: a training forward needs at least 2 windows (BatchNorm over the batch of descriptors)"); return -1; }
  static const int gf = (int)ral_knob("DANET_GRID_F", 1024);
  const int grid = B < gf ? B : gf;
  static const int gdesc = (int)ral_knob("DANET_GRID_D", 256);   // descriptor-level kernels: a wave per window,
  const int gridd = (B + 15) / 16 < gdesc ? (B + 15) / 16 : gdesc;                                   // 16 windows per workgroup and pass
  static const int gact = (int)ral_knob("DANET_GRID_A", 256);    // elementwise kernels: EW = 4 windows per pass
  const int grida = (B + 3) / 4 < gact ? (B + 3) / 4 : gact;
  if (training) (void)hipMemsetAsync(m->sums, 0, sizeof(double) * 8 * S_CELL, st);
  m->last_x = x; m->last_B = B; m->last_training = training != 0;
  for (int i = 0; i < 8; ++i) {
    const CellOff& K = m->lay.cell[i];
    const Geo& G = K.g;
    double* S = m->sums + (size_t)i * S_CELL;
    const Fcn Fa = mk_fcn(P, K.act);
    const Bn bn = mk_bn(P, K.bn);
    const float* in0 = i == 0 ? x : m->out[i - 1];
    const float* in1 = i >= 5 ? m->out[7 - i] : nullptr;     // decoder cells 1..3 add encoder features 2, 1, 0
    float* out = i == 7 ? y : m->out[i];
    const size_t nz = (size_t)G.c * G.lout;
    const size_t l1 = ((size_t)G.cin * (G.lin + 2 * CH) + nz + 2 * G.c) * sizeof(float);
    auto conv = [&](auto kern) {
      set_lds(kern, l1);
      kern<<<grid, NT, l1, st>>>(in0, in1, P.params + K.cw, P.params + K.cb, G, Fa, m->z[i], m->desc[i], m->h1[i],
                                 training ? S + S_AH1 : nullptr, B);
    };
    if (G.k == 17) conv(k_dn_conv<17, false>); else if (G.k == 3) conv(k_dn_conv<3, false>);
    else if (G.k == 4) conv(k_dn_conv<4, true>); else conv(k_dn_conv<18, true>);
    k_dn_fcn_mid<<<gridd, NT, 0, st>>>(m->h1[i], S + S_AH1, Fa, m->h2[i], training ? S + S_AH2 : nullptr, 1, B, training);
    if (G.lout % 4 == 0) k_dn_act<4><<<grida, NT, 0, st>>>(m->z[i], m->h2[i], S + S_AH2, Fa, G, m->a[i], training ? S + S_BN : nullptr, B, training);
    else k_dn_act<1><<<grida, NT, 0, st>>>(m->z[i], m->h2[i], S + S_AH2, Fa, G, m->a[i], training ? S + S_BN : nullptr, B, training);
    Fcn Fd = Fa;
    if (G.dam) {
      Fd = mk_fcn(P, K.dam);
      const size_t l2 = nz * sizeof(float);
      set_lds(k_dn_dam1, l2);
      k_dn_dam1<<<grid, NT, l2, st>>>(m->a[i], S + S_BN, bn, Fd, G, m->pool[i], m->dh1[i], training ? S + S_DH1 : nullptr, B, training);
      k_dn_fcn_mid<<<gridd, NT, 0, st>>>(m->dh1[i], S + S_DH1, Fd, m->dh2[i], training ? S + S_DH2 : nullptr, 2, B, training);
    }
    const size_t l3 = (nz + G.lout) * sizeof(float);
    set_lds(k_dn_out, l3);
    k_dn_out<<<grid, NT, l3, st>>>(m->a[i], S + S_BN, bn, G, m->dh2[i], S + S_DH2, Fd, G.dam ? P.params + K.saw : nullptr,
                                    G.dam ? P.params + K.sab : nullptr, out, B, training, G.dam ? 0 : 1);
  }
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "DANet forward launch failed"); return -1; }
  return 0;
}

int danet_backward(DanetModel* m, const float* dy, float* dx, int B, hipStream_t st, char* err, size_t cap) {
  DanetPublic& P = m->pub;
  if (!P.cfg.train || !P.grads) { snprintf(err, cap, "DANet backward needs train=1 and a bound gradient buffer"); return -1; }
  if (B != m->last_B || !m->last_training) { snprintf(err, cap, "DANet backward needs a training forward of the same batch first"); return -1; }
  static const int gb = (int)ral_knob("DANET_GRID_B", 1024);   // (train step at batch 2048: 3.84 / 3.58 / 4.43 ms with 512 / 1024 / 2048)
  const int grid = B < gb ? B : gb;
  static const int gdesc = (int)ral_knob("DANET_GRID_D", 256);   // descriptor-level kernels (see danet_forward)
  const int gridd = (B + 15) / 16 < gdesc ? (B + 15) / 16 : gdesc;
  static const int gact = (int)ral_knob("DANET_GRID_A", 256);    // elementwise kernels (see danet_forward)
  const int grida = (B + 3) / 4 < gact ? (B + 3) / 4 : gact;
  static const int gwin = (int)ral_knob("DANET_GRID_W", 512);   // window-at-a-time kernels with column-sum flushes (train step at batch 2048: 2.33 / 2.23 / 2.26 / 2.44 ms with 1024 / 512 / 256 / 128)
  const int gridw = B < gwin ? B : gwin;
  (void)hipMemsetAsync(P.grads, 0, (size_t)m->lay.nparam * sizeof(float), st);
  // the backward halves [T_D2, S_CELL) of the eight cells' sum records: one strided fill
  (void)hipMemset2DAsync(m->sums + T_D2, sizeof(double) * S_CELL, 0, sizeof(double) * (S_CELL - T_D2), 8, st);
  for (int i = 7; i >= 0; --i) {
    const CellOff& K = m->lay.cell[i];
    const Geo& G = K.g;
    double* S = m->sums + (size_t)i * S_CELL;
    const Fcn Fa = mk_fcn(P, K.act);
    const Bn bn = mk_bn(P, K.bn);
    const size_t nz = (size_t)G.c * G.lout;
    // gradient(s) at the cell output: the last cell gets dy, decoder cells the next cell's input gradient, encoder cell
    // 3 the first decoder cell's, encoder cells 0..2 the next encoder cell's plus the decoder's skip
    const float* g0 = i == 7 ? dy : m->g[i];
    const float* g1 = i < 3 ? m->gs[i] : nullptr;
    float* work = m->g[i];                         // dxo / da live here (for i == 7: copy of dy's role)
    Fcn Fd = Fa;
    if (G.dam) {
      Fd = mk_fcn(P, K.dam);
      const size_t l1 = (2 * nz + 3 * G.lout) * sizeof(float);
      set_lds(k_dn_dam_b, l1);
      k_dn_dam_b<<<gridw, NT, l1, st>>>(g0, m->a[i], S + S_BN, bn, G, m->dh2[i], S + S_DH2, Fd, P.params + K.saw, P.params + K.sab,
                                        P.grads + K.saw, P.grads + K.sab, work, m->ddy2[i], S + T_D2, B);
      const size_t l2 = ((size_t)Fd.dout * Fd.dh + Fd.dout) * sizeof(float);
      k_dn_fcn_bmid<<<gridd, NT, l2, st>>>(m->ddy2[i], S + T_D2, m->dh2[i], S + S_DH2, m->dh1[i], S + S_DH1, Fd, m->ddy1[i], S + T_D1, 2, B);
      g0 = work; g1 = nullptr;
    }
    const size_t l3 = (nz + (G.dam ? (size_t)Fd.dh * Fd.din + Fd.dh : 0)) * sizeof(float);
    set_lds(k_dn_bn_b, l3);
    k_dn_bn_b<<<gridw, NT, l3, st>>>(g0, g1, m->a[i], S + S_BN, bn, G, m->ddy1[i], S + T_D1, m->dh1[i], S + S_DH1, m->pool[i], Fd,
                                     work, S + T_BN, B);
    if (G.lout % 4 == 0) k_dn_act_b<4><<<grida, NT, 0, st>>>(work, m->a[i], m->z[i], S + S_BN, S + T_BN, bn, G, m->h2[i], S + S_AH2, Fa, m->dy2[i], S + T_A2, B);
    else k_dn_act_b<1><<<grida, NT, 0, st>>>(work, m->a[i], m->z[i], S + S_BN, S + T_BN, bn, G, m->h2[i], S + S_AH2, Fa, m->dy2[i], S + T_A2, B);
    const size_t l4 = ((size_t)Fa.dout * Fa.dh + Fa.dout) * sizeof(float);
    set_lds(k_dn_fcn_bmid, l4);
    k_dn_fcn_bmid<<<gridd, NT, l4, st>>>(m->dy2[i], S + T_A2, m->h2[i], S + S_AH2, m->h1[i], S + S_AH1, Fa, m->dy1[i], S + T_A1, 1, B);
    const float* in0 = i == 0 ? m->last_x : m->out[i - 1];
    const float* in1 = i >= 5 ? m->out[7 - i] : nullptr;
    float* gi0 = i == 0 ? dx : m->g[i - 1];
    float* gi1 = i >= 5 ? m->gs[7 - i] : nullptr;
    const size_t l5 = ((size_t)G.cin * (G.lin + 2 * CH) + (size_t)G.c * (G.lout + 2 * CH) + (size_t)G.c * G.cin * G.k + G.c +
                       (size_t)Fa.dh * Fa.din + Fa.dh) * sizeof(float);
    auto convb = [&](auto kern) {
      set_lds(kern, l5);
      kern<<<grid, NT, l5, st>>>(work, m->z[i], in0, in1, P.params + K.cw, P.grads + K.cw, P.grads + K.cb, G, m->dy1[i], S + T_A1,
                                 m->h1[i], S + S_AH1, m->h2[i], S + S_AH2, m->desc[i], Fa, gi0, gi1, B);
    };
    if (G.k == 17) convb(k_dn_conv_b<17, false>); else if (G.k == 3) convb(k_dn_conv_b<3, false>);
    else if (G.k == 4) convb(k_dn_conv_b<4, true>); else convb(k_dn_conv_b<18, true>);
  }
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "DANet backward launch failed"); return -1; }
  return 0;
}
