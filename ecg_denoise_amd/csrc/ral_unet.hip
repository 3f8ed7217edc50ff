// U-Net baseline (reference model/UNet.py:46-141) for gfx950: 11 conv stages, every tensor of a
// window is leads*L floats, BatchNorm (batch statistics) after every conv.
//
// HBM-bound design: a stage NEVER writes a normalised/activated tensor.  It writes the conv output
// (pre-BatchNorm) once plus per-channel double sums; the consumer applies BatchNorm + LeakyReLU (+ the
// additive skip) while staging its input tile in LDS.  Forward traffic per window = one read and
// one write of leads*L floats per stage (+ one re-read per skip).  Backward mirrors it: a stage reads
// the gradient at its BatchNorm output, applies the BatchNorm-backward correction on load (needs the
// two global sums the PRODUCING kernel accumulated), and emits the gradient at its producers'
// BatchNorm outputs together with their sums.
#ifdef RAL_STAMP_TU_UNET
#define RAL_STAMP_HERE
#endif
#include "ral_unet.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "ral_device.hpp"

#define MAXC 32
enum { ACT_NONE = 0, ACT_LRELU = 1 };
enum { NORM_NONE = 0, NORM_BATCH = 1, NORM_RUNNING = 2 };
enum { CONV = 0, CONVT = 1 };
enum { TY_ZBN = 0 /* z -> BN -> [lrelu] */, TY_ABN = 1 /* lrelu(z) -> BN */, TY_PLAIN = 2 /* no BN */ };

struct Src {            // one input operand of a stage: act(norm(z))
  const float* z;       // (B, C, L) pre-BN tensor
  const double* sums;   // fwd sums of its BatchNorm (sum[C], sumsq[C]) or null
  const float* gamma; const float* beta; const float* running;  // running: mean[C], var[C]
  int norm, act;
  int nrep;             // `sums` is spread over nrep replicas of 64 doubles (1: the record itself)
  // backward outputs for this operand: gradient at its BatchNorm output (+ its two sums)
  float* G; double* bsums; int accumulate;
};

struct Stage {
  Src a, b, r;          // main input, additive skip (b.z == null: none), residual added to the OUTPUT (r)
  const float* w; const float* bias;
  float* out; double* sums_out;
  int cin, cout, ks, mode, stride, pad, lin, lout, post_lrelu;
  double count;         // global elements per channel (windows * lout) for this stage's output BN
  double count_a, count_b, count_r;
  // backward
  const float* Gout; const double* bsums_out; const float* gamma_out; int type;
  float* gw; float* gb;
  float* part; int64_t part_stride; int part_w, part_b;   // per-workgroup gradient partial rows (null: atomics into gw / gb)
  int nrep;             // replicas of every record this stage ADDS to (sums_out forward; a / b / r .bsums backward) and of
                        // bsums_out: workgroup w adds to replica w % nrep.  A same-address atomic chain costs ~30 ns per link
                        // on MI355X: 512 workgroups adding to ONE record were 15 us of a 20 us stage
};

RAL_DEV float lrelu01(float v) { return v > 0.f ? v : 0.01f * v; }
// x / d (x >= 0) for a divisor that is nearly always a power of two here (window lengths, tensor sizes, tile counts):
// sh = log2(d) or -1.  A run-time integer division is ~25 vector instructions; the stage kernels are bound by those.
// Row stride (floats) of a weight matrix kept in LDS as a B operand of the MFMA tiles: lane (r, g) of a k-step reads row r,
// column 4 ks + g.  A 4-byte LDS read serves lanes 0-31 together (16 rows x 2 columns) from 32 banks: conflict-free when the
// stride is 2 x an odd number (the row lengths 32, 64, 96 put the 16 rows in ONE bank pair; rows are staged as 8-byte pieces).
constexpr int wrow_ld(int row) { return (row % 4 == 2) ? row : row + 2; }
RAL_DEV void st_f4_as_f2(float* dst, float4 v) {   // 16 bytes to an 8-byte aligned LDS address
  *reinterpret_cast<float2*>(dst) = make_float2(v.x, v.y);
  *reinterpret_cast<float2*>(dst + 2) = make_float2(v.z, v.w);
}
RAL_DEV int pow2_shift(int d) { return (d > 0 && (d & (d - 1)) == 0) ? __builtin_ctz((unsigned)d) : -1; }
RAL_DEV int qdiv(int x, int d, int sh) { return sh >= 0 ? (x >> sh) : x / d; }

// One BatchNorm record (S1[MAXC], S2[MAXC] doubles) that may be spread over `nrep` replicas of 64 doubles -> out64 (LDS).
// All 256 threads of the workgroup take part (thread t: entry t & 63 of the replicas (t >> 6) + 4 j), the four partial rows
// meet in LDS.  Split in two so that a kernel can REQUEST every record it needs (and its weights) before it waits for the
// first: hipcc waits for a load right before its first use, so loads issued back to back share one memory round trip.
#define UNET_MAXREP 16
// A record is folded by ONE wave (lane = entry, all replicas requested at once, the sum kept in registers until
// fold_store writes it to the record's LDS slot): wave w of the workgroup takes record w, so the records of a kernel's
// prologue are folded side by side and ONE barrier makes all of them visible (fold_rec(slot)[0..63]).
#define UNET_FOLD_SLOTS 5
struct FoldLd { double v[UNET_MAXREP]; };
RAL_DEV FoldLd fold_load(const double* rec, int nrep) {
  const int idx = threadIdx.x & 63;
  FoldLd f;
#pragma unroll
  for (int k = 0; k < UNET_MAXREP; ++k) f.v[k] = rec[(size_t)(k < nrep ? k : 0) * 64 + idx];   // (branch-free: clamped)
  return f;
}
RAL_DEV double* fold_rec(int slot) { __shared__ double rec[UNET_FOLD_SLOTS][64]; return rec[slot]; }
RAL_DEV void fold_store(const FoldLd& f, int nrep, int slot) {
  double sum = 0.0;
#pragma unroll
  for (int k = 0; k < UNET_MAXREP; ++k) sum += (k < nrep) ? f.v[k] : 0.0;
  fold_rec(slot)[threadIdx.x & 63] = sum;
}
// the wave's record among up to five (null = none): pointer and replica count by wave index
struct FoldPick { const double* rec; int nrep; };
RAL_DEV FoldPick fold_pick(const double* r0, int n0, const double* r1, int n1, const double* r2, int n2, const double* r3, int n3,
                           const double* r4, int n4, const void* any) {
  const int w = threadIdx.x >> 6;
  FoldPick p;
  p.rec = w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : (w == 3 ? r3 : (w == 4 ? r4 : nullptr))));
  p.nrep = w == 0 ? n0 : (w == 1 ? n1 : (w == 2 ? n2 : (w == 3 ? n3 : (w == 4 ? n4 : 1))));
  if (!p.rec) { p.rec = reinterpret_cast<const double*>(any); p.nrep = 0; }      // (nrep 0: loads a valid address, stores 0)
  return p;
}
RAL_DEV void fold_record(const double* rec, int nrep, double* out64) {   // one record on its own (barriers inside)
  if (threadIdx.x < 64) fold_store(fold_load(rec, nrep), nrep, 0);
  __syncthreads();
  if (threadIdx.x < 64) out64[threadIdx.x] = fold_rec(0)[threadIdx.x];
  __syncthreads();
}

// scale/shift (and mean/rstd) of one operand's BatchNorm into LDS: ss[0:C] scale, [C:2C] shift, [2C:3C] mean, [3C:4C] rstd.
// src_request: the affine parameters of channel (thread & 63), requested early (any valid address serves an operand without
// them: the values are not used).  src_coeffs_slot: the coefficients from the folded record in `slot`, computed by the
// 64-thread group `grp` of the workgroup (no barrier: the caller places one).
struct SrcReq { float gamma, beta, rmean, rvar; };
RAL_DEV const double* src_rec(const Src& s) { return s.norm == NORM_BATCH ? s.sums : nullptr; }
RAL_DEV SrcReq src_request(const Src& s, int C, const void* any) {
  SrcReq q;
  const int c = (int)(threadIdx.x & 63) < C ? (int)(threadIdx.x & 63) : 0;
  const float* fa = reinterpret_cast<const float*>(any);
  q.gamma = (s.norm != NORM_NONE ? s.gamma : fa)[c];
  q.beta = (s.norm != NORM_NONE ? s.beta : fa)[c];
  q.rmean = (s.norm == NORM_RUNNING ? s.running : fa)[c];
  q.rvar = (s.norm == NORM_RUNNING ? s.running + C : fa)[c];
  return q;
}
RAL_DEV void src_coeffs_slot(const Src& s, int C, double count, float* ss, const SrcReq& pre, int slot, int grp) {
  const int c = (int)threadIdx.x - 64 * grp;
  if (c >= 0 && c < C) {
    const double* rec = fold_rec(slot);
    float mean = 0.f, rstd = 1.f, sc = 1.f, sh = 0.f;
    if (s.norm == NORM_BATCH) {
      const double m = rec[c] / count;
      double var = rec[MAXC + c] / count - m * m;
      if (var < 0.0) var = 0.0;
      mean = (float)m; rstd = (float)(1.0 / sqrt(var + 1e-5));
    } else if (s.norm == NORM_RUNNING) {
      mean = pre.rmean; rstd = 1.0f / sqrtf(pre.rvar + 1e-5f);
    }
    if (s.norm != NORM_NONE) { sc = pre.gamma * rstd; sh = pre.beta - mean * sc; }
    ss[c] = sc; ss[C + c] = sh; ss[2 * C + c] = mean; ss[3 * C + c] = rstd;
  }
}
RAL_DEV void src_coeffs(const Src& s, int C, double count, float* ss) {   // one operand on its own (barriers inside)
  const SrcReq pre = src_request(s, C, s.z);
  if (s.norm == NORM_BATCH && threadIdx.x < 64) fold_store(fold_load(s.sums, s.nrep), s.nrep, 0);
  __syncthreads();
  src_coeffs_slot(s, C, count, ss, pre, 0, 0);
  __syncthreads();
}

RAL_DEV float src_value(const Src& s, const float* ss, int C, int c, float z) {
  float v = z * ss[c] + ss[C + c];
  return s.act == ACT_LRELU ? lrelu01(v) : v;
}

// ---------------------------------------------------------------------------------
// forward stage
// LDS: in tile cin x lin | weights | bias | coeffs a,b,r (4*MAXC each) | stats 2*MAXC
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unet_fwd(Stage st, int B) {
  extern __shared__ float4 smem4[];
  float* in = reinterpret_cast<float*>(smem4);
  float* ws = in + st.cin * st.lin;
  const int nw = st.cin * st.cout * st.ks;
  float* bs = ws + nw;
  float* ca = bs + MAXC; float* cb = ca + 4 * MAXC; float* cr = cb + 4 * MAXC;
  float* red = cr + 4 * MAXC;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) ws[i] = st.w[i];
  for (int i = threadIdx.x; i < st.cout; i += blockDim.x) bs[i] = st.bias[i];
  for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) red[i] = 0.f;
  src_coeffs(st.a, st.cin, st.count_a, ca);
  if (st.b.z) src_coeffs(st.b, st.cin, st.count_b, cb);
  if (st.r.z) src_coeffs(st.r, st.cout, st.count_r, cr);
  __syncthreads();
  const int nin = st.cin * st.lin, nout = st.cout * st.lout;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* za = st.a.z + (size_t)win * nin;
    const float* zb = st.b.z ? st.b.z + (size_t)win * nin : nullptr;
    for (int i = threadIdx.x; i < nin; i += blockDim.x) {
      const int c = i / st.lin;
      float v = src_value(st.a, ca, st.cin, c, za[i]);
      if (zb) v += src_value(st.b, cb, st.cin, c, zb[i]);
      in[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nout; i += blockDim.x) {
      const int co = i / st.lout, l = i - co * st.lout;
      float acc = bs[co];
      if (st.mode == CONV) {
        for (int ci = 0; ci < st.cin; ++ci) {
          const float* wr = ws + (co * st.cin + ci) * st.ks;
          const float* ir = in + ci * st.lin;
          for (int k = 0; k < st.ks; ++k) {
            const int p = l * st.stride - st.pad + k;
            if (p >= 0 && p < st.lin) acc = fmaf(wr[k], ir[p], acc);
          }
        }
      } else {  // ConvTranspose1d(k4, s2, p1): out[j] += w[ci][co][k] in[ci][i], j = 2i - 1 + k
        const int k0 = (l + 1) & 1;
        for (int ci = 0; ci < st.cin; ++ci) {
          const float* wr = ws + (ci * st.cout + co) * st.ks;
          const float* ir = in + ci * st.lin;
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int k = k0 + 2 * kk, p = (l + 1 - k) >> 1;
            if (p >= 0 && p < st.lin) acc = fmaf(wr[k], ir[p], acc);
          }
        }
      }
      if (st.post_lrelu) acc = lrelu01(acc);
      if (st.r.z) acc += src_value(st.r, cr, st.cout, co, st.r.z[(size_t)win * nout + i]);
      st.out[(size_t)win * nout + i] = acc;
      if (st.sums_out) { atomicAdd(red + co, acc); atomicAdd(red + MAXC + co, acc * acc); }
    }
    __syncthreads();
  }
  if (st.sums_out && (int)threadIdx.x < st.cout) {
    double* rec = st.sums_out + (size_t)(blockIdx.x % st.nrep) * 64;
    atomicAdd(rec + threadIdx.x, (double)red[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, (double)red[MAXC + threadIdx.x]);
  }
}

// weight-gradient tile of slot s (of `slots` per workgroup), rotated by the workgroup index: the workgroups of a launch
// finish together, and without the rotation every one of them would start its flush on the same addresses
RAL_DEV int dw_tile(int s, int slots) { return (s + (int)blockIdx.x) % slots; }

// LDS atomic add of the sum over groups of `w` consecutive lanes (w = 1: every lane adds its own value; w in {8, 16,
// 32, 64}: the group's first lane adds the group sum - same-address LDS atomics of a wave execute one after the other)
// (the fast stage kernels keep these accumulators in DOUBLES: ds_add_f64 takes 8 LDS cycles per wave-instruction on gfx950,
// ds_add_f32 three per active lane - 49 for the sixteen lanes of an MFMA tile's column sums; tools/diag/lds_cost_probe.hip)
template <class T>
RAL_DEV void seg_atomic(T* addr, float v, int w) {
  if (w == 1) { atomicAdd(addr, (T)v); return; }
  const int lane = threadIdx.x & 63;
  if (w == 64) v = group_sum<64>(v);
  else if (w == 32) v = group_sum<32>(v);
  else if (w == 16) v = group_sum<16>(v);
  else v = group_sum<8>(v);
  if ((lane & (w - 1)) == 0) atomicAdd(addr, (T)v);
}

// ---------------------------------------------------------------------------------
// forward stage, specialised: channel counts, kernel size and conv type are template parameters, every
// thread produces 4 consecutive positions of one output channel from registers (inputs of a channel are
// read once from the zero-haloed LDS row and reused by the 4 outputs), per-channel statistics go to LDS
// with two atomics per thread.  MODE 0: Conv1d(k3, s2, p1); 1: Conv1d(k1 | k3, s1, same); 2: ConvTranspose1d(k4, s2, p1)
// phase stamps of ONE instantiation (diagnostic builds: make STAMP=1 STAMPTU=UNET STAMPSEL='CIN==4&&COUT==2&&MODE==2';
// tools/diag/stamp_unet_bwd.py): slots 16.. = prologue, load phase, input gradient, weight gradient, flush of the backward stage;
// slots 8.. = prologue, load phase, compute, flush of the forward stage
#ifndef UNET_STAMP_WG
#define UNET_STAMP_WG 0
#endif
#if defined(RAL_STAMP) && defined(RAL_STAMP_HERE) && defined(UNET_STAMP_SEL)
#define UB_STAMP(i) do { if ((UNET_STAMP_SEL) && blockIdx.x == UNET_STAMP_WG && threadIdx.x == 0) { const long long t_ = clock64(); \
    atomicAdd(&g_ral_stamps[i], (unsigned long long)(t_ - ub_prev_)); ub_prev_ = t_; } } while (0)
#define UB_STAMP_INIT() long long ub_prev_ = clock64()
#else
#define UB_STAMP(i) do {} while (0)
#define UB_STAMP_INIT() do {} while (0)
#endif
// ---------------------------------------------------------------------------------
#ifndef UNET_FWD_THREADS
#define UNET_FWD_THREADS 512
#endif
template <int CIN, int COUT, int KS, int MODE>
__global__ __launch_bounds__(UNET_FWD_THREADS) void k_unet_fwd_t(Stage st, int B, int WP) {
  // A workgroup takes WP CONSECUTIVE windows per pass (one pass per workgroup at the launch sizes used): their inputs are
  // one contiguous block of global memory, requested with every load of the pass in flight at once - a window at a time
  // costs one memory round trip per window, which at 4 windows per workgroup was most of a stage's time.
  extern __shared__ float4 smem4[];
  constexpr int HALO = 4;                       // 16-byte aligned zero halo on both sides of every input row
  constexpr int nw = CIN * COUT * KS;
  const int lin = st.lin, lout = st.lout, LP = lin + 2 * HALO;
  float* in = reinterpret_cast<float*>(smem4);  // WP x CIN x LP
  float* rt = in + WP * CIN * LP;               // residual operand (already lrelu(BN(z))): WP x COUT x lout, when st.r.z
  // weights: [co][ci][k] (Conv1d) / [ci][co][k] (ConvTranspose1d) as in global memory; the MFMA path of the Conv1d layers reads
  // rows of one output channel per lane and keeps them at a padded stride (wrow_ld)
  constexpr bool FMFMA = CIN >= 4 && (CIN * KS) % 4 == 0 && (MODE != 2 || COUT >= 16);   // (narrow ConvTranspose1d layers: 16-channel tiles
                                                                                          //  would be mostly padding; measured slower)
  constexpr int FKTOT = CIN * KS, WLD = (FMFMA && MODE != 2) ? wrow_ld(FKTOT) : FKTOT, WSZ = (MODE == 2) ? nw : ((COUT * WLD + 3) & ~3);
  float* ws = rt + (st.r.z ? WP * COUT * lout : 0);
  float* bs = ws + WSZ;
  float* ca = bs + MAXC; float* cb = ca + 4 * MAXC; float* cr = cb + 4 * MAXC;
  double* red = reinterpret_cast<double*>(cr + 4 * MAXC);   // column sums of the output (S1 | S2), doubles
  UB_STAMP_INIT();
  // every global load of the prologue is requested before the first is waited for: the BatchNorm records of the
  // operands, the weights (at most 3 float4 per thread) and the bias
  const SrcReq la = src_request(st.a, CIN, st.w), lb = src_request(st.b.z ? st.b : st.a, CIN, st.w),
               lr = src_request(st.r.z ? st.r : st.a, st.r.z ? COUT : CIN, st.w);
  const FoldPick fp = fold_pick(src_rec(st.a), st.a.nrep, st.b.z ? src_rec(st.b) : nullptr, st.b.nrep,
                                st.r.z ? src_rec(st.r) : nullptr, st.r.nrep, nullptr, 1, nullptr, 1, st.w);
  const FoldLd fl = fold_load(fp.rec, fp.nrep);        // (wave w: record w)
  constexpr int NTF = UNET_FWD_THREADS, NWVF = (nw / 4 + NTF - 1) / NTF;
  float4 wv[NWVF];
#pragma unroll
  for (int k = 0; k < NWVF; ++k) {
    const int i = threadIdx.x + k * NTF;
    wv[k] = reinterpret_cast<const float4*>(st.w)[i < nw / 4 ? i : 0];
  }
  const float bv = st.bias[threadIdx.x < COUT ? threadIdx.x : 0];
#pragma unroll
  for (int k = 0; k < NWVF; ++k) {
    const int i = threadIdx.x + k * NTF;
    if (i < nw / 4) st_f4_as_f2(ws + (WLD == FKTOT ? 4 * i : (4 * i / FKTOT) * WLD + (4 * i) % FKTOT), wv[k]);
  }
  if ((int)threadIdx.x < COUT) bs[threadIdx.x] = bv;
  for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) red[i] = 0.;
  for (int i = threadIdx.x; i < WP * CIN * 2 * HALO; i += blockDim.x) {   // halos stay zero for every pass
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    in[c * LP + (h < HALO ? h : lin + h)] = 0.f;
  }
  if ((threadIdx.x >> 6) < 3) fold_store(fl, fp.nrep, threadIdx.x >> 6);
  __syncthreads();
  src_coeffs_slot(st.a, CIN, st.count_a, ca, la, 0, 0);          // (each operand by its own 64-thread group)
  if (st.b.z) src_coeffs_slot(st.b, CIN, st.count_b, cb, lb, 1, 1);
  if (st.r.z) src_coeffs_slot(st.r, COUT, st.count_r, cr, lr, 2, 2);
  __syncthreads();
  const int nin = CIN * lin, nout = COUT * lout, q = lout >> 2, nslots = COUT * q, nin4 = nin >> 2, nout4 = nout >> 2;
  const int sh_lin = pow2_shift(lin), sh_lout = pow2_shift(lout), sh_nin4 = pow2_shift(nin4), sh_nout4 = pow2_shift(nout4),
            sh_q = pow2_shift(q), sh_nslots = pow2_shift(nslots);
  const bool a_lrelu = st.a.act == ACT_LRELU;
  // Layers with at least four input channels run on the matrix pipe (the vector loop below it is bound by its LDS reads: 5 of
  // the 11.5 us of the 32 -> 32 k3 stage): tiles of 16 output positions x 16 output channels, K = (ci, k) pairs in the order
  // kk = ci KS + k = 4 ks + (lane >> 4), both operands gathered from the LDS tiles by index - per-lane bases that repeat with
  // period FW_P in the k-step plus compile-time multiples of a uniform step (as in the backward kernel's input gradient).
  // A wave keeps one channel tile; a lane ends with 4 consecutive positions of one output channel.
  constexpr int FCT = (COUT + 15) / 16, FKST = FKTOT / 4, FNWAVE = UNET_FWD_THREADS / 64;
  constexpr int FW_P = (KS == 3) ? 3 : 1, FW_CSTEP = 4 * FW_P / KS;
  constexpr int FW_KCH = FKST < 12 ? (FKST < 1 ? 1 : FKST) : 12;
  static_assert(FNWAVE % FCT == 0, "a wave keeps its channel tile");
  const int lane = threadIdx.x & 63, r = lane & 15, g4 = lane >> 4, wave = threadIdx.x >> 6;
  const int cow = ((wave % FCT) << 4) + r, coc = cow < COUT ? cow : COUT - 1;
  int fabase[FW_P], fbbase[FW_P];
  bool fazero[FW_P];
#pragma unroll
  for (int j = 0; j < FW_P; ++j) {
    const int kk0 = 4 * j + g4, ci0 = kk0 / KS, k = kk0 - ci0 * KS, t0 = r + 1 - k;   // (MODE 2: output n0 + r meets input (pos + 1 - k) / 2)
    fabase[j] = ci0 * LP + HALO + (MODE == 1 ? k - (KS - 1) / 2 : (MODE == 0 ? k - 1 : (t0 >> 1)));
    fazero[j] = MODE == 2 && (t0 & 1);
    fbbase[j] = (MODE == 2) ? (ci0 * COUT + coc) * KS + k : coc * WLD + kk0;
  }
  constexpr int FW_BSTEP = (MODE == 2) ? FW_CSTEP * COUT * KS : 4 * FW_P;   // floats of the weight tensor per period of k-steps
  const float fbias = bs[coc];
  // lanes of a wave that hold the same output channel in the compute loop (q consecutive slots, when q is a power of two
  // and every wave of the loop is full): their sums are combined before the LDS atomic - same-address LDS atomics of a
  // wave execute one after the other
  const int segw = ((q & (q - 1)) == 0 && q >= 8 && nslots % 64 == 0) ? (q < 64 ? q : 64) : 1;
  UB_STAMP(8);
  for (int w0 = blockIdx.x * WP; w0 < B; w0 += gridDim.x * WP) {
    const int nwin = (B - w0) < WP ? (B - w0) : WP;
    const float4* za = reinterpret_cast<const float4*>(st.a.z + (size_t)w0 * nin);
    const float4* zb = st.b.z ? reinterpret_cast<const float4*>(st.b.z + (size_t)w0 * nin) : nullptr;
    auto put = [&](int i, float4 v, float4 u) {
      const int wi = qdiv(i, nin4, sh_nin4), e = (i - wi * nin4) << 2, c = qdiv(e, lin, sh_lin), p = e - c * lin;
      const float sa = ca[c], ha = ca[CIN + c];
      v = make_float4(v.x * sa + ha, v.y * sa + ha, v.z * sa + ha, v.w * sa + ha);
      if (a_lrelu) v = make_float4(lrelu01(v.x), lrelu01(v.y), lrelu01(v.z), lrelu01(v.w));
      if (zb) {
        const float sb = cb[c], hb = cb[CIN + c];
        v.x += lrelu01(u.x * sb + hb); v.y += lrelu01(u.y * sb + hb); v.z += lrelu01(u.z * sb + hb); v.w += lrelu01(u.w * sb + hb);
      }
      *reinterpret_cast<float4*>(in + (wi * CIN + c) * LP + HALO + p) = v;
    };
    {   // all streams of a chunk (2 float4 per thread and stream) requested before any is used; indices past the end are
        // clamped (no lane-predicated loads) and only their consumption is skipped
      constexpr int U = 2;
      const int na = nwin * nin4, nr = st.r.z ? nwin * nout4 : 0, nmax = na > nr ? na : nr, bd = blockDim.x;
      const float4* zr = st.r.z ? reinterpret_cast<const float4*>(st.r.z + (size_t)w0 * nout) : za;
      for (int i0 = 0; i0 < nmax; i0 += U * bd) {
        float4 v[U], u[U], rr[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int i = i0 + k * bd + (int)threadIdx.x, ja = i < na ? i : 0, jr = i < nr ? i : 0;
          v[k] = za[ja];
          u[k] = (zb ? zb : za)[ja];
          rr[k] = zr[jr];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int i = i0 + k * bd + (int)threadIdx.x;
          if (i < na) put(i, v[k], u[k]);
          if (i < nr) {
            const int e = (i - qdiv(i, nout4, sh_nout4) * nout4) << 2, c = qdiv(e, lout, sh_lout);
            const float sr = cr[c], hr = cr[COUT + c];
            const float4 z4 = rr[k];
            reinterpret_cast<float4*>(rt)[i] = make_float4(lrelu01(z4.x * sr + hr), lrelu01(z4.y * sr + hr), lrelu01(z4.z * sr + hr), lrelu01(z4.w * sr + hr));
          }
        }
      }
    }
    __syncthreads();
    UB_STAMP(9);
    if constexpr (FMFMA) {
      const int ptiles = (lout + 15) >> 4, sh_pt = pow2_shift(ptiles);
      for (int tile = wave; tile < nwin * ptiles * FCT; tile += FNWAVE) {
        const int tp = tile / FCT, wi = qdiv(tp, ptiles, sh_pt), n0 = (tp - wi * ptiles) << 4;
        const int pos = n0 + r, posc = pos < lout ? pos : lout - 1;   // (rows past the end: clamped address, dropped below)
        const float* xa = in + wi * CIN * LP + (MODE == 1 ? posc : (MODE == 0 ? 2 * posc : (n0 >> 1)));
        f32x4 acc0 = {fbias, fbias, fbias, fbias}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < FKST; kc += FW_KCH) {
          float av[FW_KCH], bv[FW_KCH];
#pragma unroll
          for (int u = 0; u < FW_KCH; ++u) {
            const int ks = kc + u, j = ks % FW_P, i = ks / FW_P;
            if (ks < FKST) {
              av[u] = xa[fabase[j] + i * FW_CSTEP * LP];
              if (MODE == 2 && fazero[j]) av[u] = 0.f;
              bv[u] = ws[fbbase[j] + i * FW_BSTEP];
            }
          }
#pragma unroll
          for (int u = 0; u < FW_KCH; ++u) {
            if (kc + u < FKST) {
              if (u & 1) acc1 = mfma4(av[u], bv[u], acc1); else acc0 = mfma4(av[u], bv[u], acc0);
            }
          }
        }
        f32x4 accv = acc0 + acc1;
        const int p0 = n0 + 4 * g4;
        const bool valid = cow < COUT && p0 < lout;
        float s1 = 0.f, s2 = 0.f;
        if (valid) {
          float acc[4] = {accv[0], accv[1], accv[2], accv[3]};
          if (st.post_lrelu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = lrelu01(acc[j]);
          }
          if (st.r.z) {
            const float4 rr = *reinterpret_cast<const float4*>(rt + (size_t)wi * nout + (size_t)cow * lout + p0);
            acc[0] += rr.x; acc[1] += rr.y; acc[2] += rr.z; acc[3] += rr.w;
          }
          *reinterpret_cast<float4*>(st.out + (size_t)(w0 + wi) * nout + (size_t)cow * lout + p0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
          s1 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
          s2 = (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]);
        }
        if (st.sums_out) {   // the four position groups of a channel meet in one lane: one LDS atomic per channel, tile and sum
          s1 = rows_sum(s1); s2 = rows_sum(s2);
          if (g4 == 0 && cow < COUT) { atomicAdd(red + cow, (double)s1); atomicAdd(red + MAXC + cow, (double)s2); }
        }
      }
    } else
    for (int slot = threadIdx.x; slot < nwin * nslots; slot += blockDim.x) {
      const int wi = qdiv(slot, nslots, sh_nslots), sl = slot - wi * nslots, co = qdiv(sl, q, sh_q), l0 = (sl - co * q) << 2;
      float acc[4] = {bs[co], bs[co], bs[co], bs[co]};
#pragma unroll 4
      for (int ci = 0; ci < CIN; ++ci) {
        const float* row = in + (wi * CIN + ci) * LP + HALO;
        if constexpr (MODE == 0) {                      // out[l] = sum_k w[k] in[2l - 1 + k]
          const float* wr = ws + (co * CIN + ci) * 3;
          const float w0_ = wr[0], w1 = wr[1], w2 = wr[2];
          // in[2 l0 - 1 .. 2 l0 + 7] as three aligned 16-byte reads (nine 4-byte reads at a lane stride of 8 floats are bank conflicts)
          const float4 qa = *reinterpret_cast<const float4*>(row + 2 * l0 - 4), qb = *reinterpret_cast<const float4*>(row + 2 * l0),
                       qc = *reinterpret_cast<const float4*>(row + 2 * l0 + 4);
          const float x[9] = {qa.w, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(w0_, x[2 * j], fmaf(w1, x[2 * j + 1], fmaf(w2, x[2 * j + 2], acc[j])));
        } else if constexpr (MODE == 1 && KS == 3) {    // out[l] = sum_k w[k] in[l - 1 + k]
          const float* wr = ws + (co * CIN + ci) * 3;
          const float w0_ = wr[0], w1 = wr[1], w2 = wr[2];
          const float4 qb = *reinterpret_cast<const float4*>(row + l0);
          const float x[6] = {row[l0 - 1], qb.x, qb.y, qb.z, qb.w, row[l0 + 4]};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(w0_, x[j], fmaf(w1, x[j + 1], fmaf(w2, x[j + 2], acc[j])));
        } else if constexpr (MODE == 1) {               // 1x1
          const float w0_ = ws[co * CIN + ci];
          const float4 x = *reinterpret_cast<const float4*>(row + l0);
          acc[0] = fmaf(w0_, x.x, acc[0]); acc[1] = fmaf(w0_, x.y, acc[1]); acc[2] = fmaf(w0_, x.z, acc[2]); acc[3] = fmaf(w0_, x.w, acc[3]);
        } else {                                        // ConvTranspose1d: out[j] += w[ci][co][k] in[i], j = 2i - 1 + k
          const float* wr = ws + (ci * COUT + co) * 4;
          const float w0_ = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
          const int h = l0 >> 1;
          const float x0 = row[h - 1], x1 = row[h], x2 = row[h + 1], x3 = row[h + 2];
          acc[0] = fmaf(w1, x1, fmaf(w3, x0, acc[0]));
          acc[1] = fmaf(w0_, x2, fmaf(w2, x1, acc[1]));
          acc[2] = fmaf(w1, x2, fmaf(w3, x1, acc[2]));
          acc[3] = fmaf(w0_, x3, fmaf(w2, x2, acc[3]));
        }
      }
      if (st.post_lrelu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = lrelu01(acc[j]);
      }
      if (st.r.z) {
        const float4 rr = *reinterpret_cast<const float4*>(rt + (size_t)wi * nout + (size_t)co * lout + l0);
        acc[0] += rr.x; acc[1] += rr.y; acc[2] += rr.z; acc[3] += rr.w;
      }
      *reinterpret_cast<float4*>(st.out + (size_t)(w0 + wi) * nout + (size_t)co * lout + l0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      if (st.sums_out) {
        seg_atomic(red + co, (acc[0] + acc[1]) + (acc[2] + acc[3]), segw);
        seg_atomic(red + MAXC + co, (acc[0] * acc[0] + acc[1] * acc[1]) + (acc[2] * acc[2] + acc[3] * acc[3]), segw);
      }
    }
    UB_STAMP(10);
    __syncthreads();
    UB_STAMP(11);
  }
  if (st.sums_out && (int)threadIdx.x < COUT) {
    double* rec = st.sums_out + (size_t)(blockIdx.x % st.nrep) * 64;
    atomicAdd(rec + threadIdx.x, red[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, red[MAXC + threadIdx.x]);
  }
  UB_STAMP(12);
}

// `sums` is the record the statistics come from, in nrep replicas; `final` (the caller-visible bn_sums record) receives the
// folded sums when they were kept in replicas
struct BnUpd { const double* sums; double* final_; float* running; int C; double count; int nrep; };
struct BnUpdAll { BnUpd l[10]; };
RAL_DEV void unet_running_update(const BnUpd& b, int c) {
  if (c >= b.C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < b.nrep; ++k) { s1 += b.sums[(size_t)k * 64 + c]; s2 += b.sums[(size_t)k * 64 + MAXC + c]; }
  if (b.final_ != b.sums) { b.final_[c] = s1; b.final_[MAXC + c] = s2; }
  const double m = s1 / b.count;
  double var = s2 / b.count - m * m;
  if (var < 0.0) var = 0.0;
  b.running[c] = 0.9f * b.running[c] + 0.1f * (float)m;
  b.running[b.C + c] = 0.9f * b.running[b.C + c] + 0.1f * (float)(b.count > 1.0 ? var * b.count / (b.count - 1.0) : var);
}

// final: y = BN9(z9) elementwise; in training its first workgroups also update the running statistics of the ten layers
// (and leaves the folded sums in the caller-visible records)
__global__ void k_unet_out(Src s, float* __restrict__ y, int C, int L, double count, size_t total, BnUpdAll u, int do_running) {
  __shared__ float ss[4 * MAXC];
  src_coeffs(s, C, count, ss);
  __syncthreads();
  if (do_running) {      // layer b by workgroup b (the grid has at least ten: the launcher sees to it), beside its share of y
    for (int b = blockIdx.x; b < 10; b += gridDim.x)
      if ((int)threadIdx.x < MAXC) unet_running_update(u.l[b], threadIdx.x);
  }
  const float4* z4 = reinterpret_cast<const float4*>(s.z);
  float4* y4 = reinterpret_cast<float4*>(y);
  const size_t n4 = total >> 2;                     // (L is a multiple of 16: rows are whole float4s)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(((i << 2) / L) % C);
    const float sc = ss[c], sh = ss[C + c];
    const float4 v = z4[i];
    y4[i] = make_float4(v.x * sc + sh, v.y * sc + sh, v.z * sc + sh, v.w * sc + sh);
  }
}

// sums of the gradient at a BatchNorm output: S1 = sum G, S2 = sum G * zhat   (used for the last layer).  One wave per
// (window, channel) row of L floats (L a multiple of 4): 16-byte loads, four of each operand in flight per lane, a wave-level
// sum, then one LDS atomic per row and sum.
__global__ __launch_bounds__(256) void k_unet_gsums(const float* __restrict__ G, Src s, int C, int L, double count,
                                                    double* __restrict__ bsums, int nrep, size_t total) {
  __shared__ float ss[4 * MAXC];
  __shared__ float red[2 * MAXC];
  src_coeffs(s, C, count, ss);
  for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) red[i] = 0.;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, L4 = L >> 2;
  const size_t rows = total / L;
  for (size_t row = (size_t)blockIdx.x * 4 + wave; row < rows; row += (size_t)gridDim.x * 4) {
    const int c = (int)(row % C);
    const float4* g4 = reinterpret_cast<const float4*>(G + row * L);
    const float4* z4 = reinterpret_cast<const float4*>(s.z + row * L);
    const float mu = ss[2 * C + c], rs = ss[3 * C + c];
    float s1 = 0.f, s2 = 0.f;
    auto take = [&](float4 g, float4 z) {
      s1 += f4hsum(g);
      s2 += g.x * ((z.x - mu) * rs) + g.y * ((z.y - mu) * rs) + g.z * ((z.z - mu) * rs) + g.w * ((z.w - mu) * rs);
    };
    int i = lane;
    for (; i + 3 * 64 < L4; i += 4 * 64) {
      float4 gv[4], zv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) gv[k] = g4[i + k * 64];
#pragma unroll
      for (int k = 0; k < 4; ++k) zv[k] = z4[i + k * 64];
#pragma unroll
      for (int k = 0; k < 4; ++k) take(gv[k], zv[k]);
    }
    for (; i < L4; i += 64) take(g4[i], z4[i]);
    s1 = group_sum<64>(s1); s2 = group_sum<64>(s2);
    if (lane == 0) { atomicAdd(red + c, s1); atomicAdd(red + MAXC + c, s2); }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double* rec = bsums + (size_t)(blockIdx.x % nrep) * 64;
    atomicAdd(rec + threadIdx.x, red[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, red[MAXC + threadIdx.x]);
  }
}

// The tail of a TRAINING step's forward in one pass over the last layer's tensor (ral_forward_loss_means): y = BN9(z9) as
// k_unet_out writes it (and the running statistics), the loss / SNR / RMSE sums and dy = 2 (y - t) / (global windows x n) as
// k_loss_w forms them (same summation order), and the two sums of dy at the output BatchNorm that the backward pass starts
// from (k_unet_gsums) - three kernels, each one a memory round trip over B x leads x L floats plus its tail of atomics,
// measured 10.1 + 12.9 + 6.3 us of the 424 us step at 2048 x 2 x 512.  One wave per window, two windows in flight.
#define UOL_WAVES 8
__global__ __launch_bounds__(64 * UOL_WAVES) void k_unet_out_loss(Src s, float* __restrict__ y, const float* __restrict__ target,
                                                                  float* __restrict__ dy, float* __restrict__ snr,
                                                                  float* __restrict__ rmse, double* __restrict__ loss_sum, int C, int L,
                                                                  int B, double count, float gscale, double* __restrict__ fin,
                                                                  double fin_scale, int fin3, BnUpdAll u, double* __restrict__ bsums,
                                                                  int nrep) {
  __shared__ float ss[4 * MAXC];
  __shared__ float redg[2 * MAXC];
  __shared__ double red[3][UOL_WAVES];
  src_coeffs(s, C, count, ss);
  for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) redg[i] = 0.f;
  __syncthreads();
  for (int b = blockIdx.x; b < 10; b += gridDim.x)
    if ((int)threadIdx.x < MAXC) unet_running_update(u.l[b], threadIdx.x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = C * L, n4 = n >> 2, L4 = L >> 2;
  const bool rowwise = (L4 & 63) == 0;          // the 64 lanes of a load slice share a channel: one LDS atomic per slice and sum
  double mine = 0.0, msnr = 0.0, mrmse = 0.0;
  const int stride = gridDim.x * UOL_WAVES;
  for (int w = blockIdx.x * UOL_WAVES + wave; w < B; w += 2 * stride) {
    const bool two = w + stride < B;
    const int wb = two ? w + stride : w;
    const float4* za = reinterpret_cast<const float4*>(s.z + (size_t)w * n);
    const float4* ta = reinterpret_cast<const float4*>(target + (size_t)w * n);
    const float4* zb = reinterpret_cast<const float4*>(s.z + (size_t)wb * n);
    const float4* tb = reinterpret_cast<const float4*>(target + (size_t)wb * n);
    float va0 = 0.f, va1 = 0.f, vb0 = 0.f, vb1 = 0.f;
    for (int i0 = 0; i0 < n4; i0 += 4 * 64) {
      float4 zav[4], tav[4], zbv[4], tbv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane, j = i < n4 ? i : 0;
        zav[k] = za[j]; tav[k] = ta[j]; zbv[k] = zb[j]; tbv[k] = tb[j];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane;
        const bool in = i < n4;
        const int c = in ? i / L4 : 0;
        const float sc = ss[c], sh = ss[C + c], mu = ss[2 * C + c], rs = ss[3 * C + c];
        float g1 = 0.f, g2 = 0.f;
        auto one = [&](const float4 z, const float4 t, size_t wo, float& v0, float& v1) {
          const float4 yv = make_float4(z.x * sc + sh, z.y * sc + sh, z.z * sc + sh, z.w * sc + sh);
          reinterpret_cast<float4*>(y + wo)[i] = yv;
          const float4 d = f4sub(yv, t);
          v0 += f4dot(d, d); v1 += f4dot(t, t);
          const float4 gq = f4scale(d, gscale);
          reinterpret_cast<float4*>(dy + wo)[i] = gq;
          g1 += f4hsum(gq);
          g2 += gq.x * ((z.x - mu) * rs) + gq.y * ((z.y - mu) * rs) + gq.z * ((z.z - mu) * rs) + gq.w * ((z.w - mu) * rs);
        };
        if (in) {
          one(zav[k], tav[k], (size_t)w * n, va0, va1);
          if (two) one(zbv[k], tbv[k], (size_t)wb * n, vb0, vb1);
        }
        if (rowwise) {                             // (n4 is a multiple of 64 then: the slice is whole)
          g1 = group_sum<64>(g1); g2 = group_sum<64>(g2);
          if (lane == 0 && i0 + k * 64 < n4) { atomicAdd(redg + c, g1); atomicAdd(redg + MAXC + c, g2); }
        } else if (in) { atomicAdd(redg + c, g1); atomicAdd(redg + MAXC + c, g2); }
      }
    }
    const float sa0 = group_sum<64>(va0), sa1 = group_sum<64>(va1), sb0 = group_sum<64>(vb0), sb1 = group_sum<64>(vb1);
    if (lane == 0) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 1 && !two) break;
        const double sse = (double)(h ? sb0 : sa0);
        const float mse = (float)(sse / n), my2 = (float)((double)(h ? sb1 : sa1) / n);
        const float sn = 10.0f * log10f(my2 / mse), rm = sqrtf(mse);
        const int ww = h ? wb : w;
        if (snr) snr[ww] = sn;
        if (rmse) rmse[ww] = rm;
        mine += sse / n; msnr += (double)sn; mrmse += (double)rm;
      }
    }
  }
  if (lane == 0) { red[0][wave] = mine; red[1][wave] = msnr; red[2][wave] = mrmse; }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    double* rec = bsums + (size_t)(blockIdx.x % nrep) * 64;
    atomicAdd(rec + threadIdx.x, (double)redg[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, (double)redg[MAXC + threadIdx.x]);
  }
  if (threadIdx.x == 0 && loss_sum) {
    double t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = ((red[k][0] + red[k][1]) + (red[k][2] + red[k][3])) + ((red[k][4] + red[k][5]) + (red[k][6] + red[k][7]));
    loss_commit(loss_sum, t[0], fin, fin_scale, fin3, t[1], t[2]);
  }
}

// ---------------------------------------------------------------------------------
// backward stage
// LDS: in tile | dconv tile (cout x lout) | d_in tile (cin x lin) | weights | coeffs a,b,r,out | sums a,b,r (2*MAXC each) | gb
// ---------------------------------------------------------------------------------
#define UNET_MAXW 12   // weight-gradient entries per thread (3072 / 256)
__global__ __launch_bounds__(256) void k_unet_bwd(Stage st, int B) {
  extern __shared__ float4 smem4[];
  const int nin = st.cin * st.lin, nout = st.cout * st.lout, nw = st.cin * st.cout * st.ks;
  float* in = reinterpret_cast<float*>(smem4);
  float* dc = in + nin;
  float* din = dc + nout;
  float* ws = din + nin;
  float* ca = ws + nw; float* cb = ca + 4 * MAXC; float* cr = cb + 4 * MAXC; float* co_ = cr + 4 * MAXC;
  float* sa = co_ + 4 * MAXC; float* sb = sa + 2 * MAXC; float* sr = sb + 2 * MAXC;
  float* gbs = sr + 2 * MAXC;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) ws[i] = st.w[i];
  for (int i = threadIdx.x; i < 7 * MAXC; i += blockDim.x) sa[i] = 0.f;   // sa, sb, sr, gbs
  src_coeffs(st.a, st.cin, st.count_a, ca);
  if (st.b.z) src_coeffs(st.b, st.cin, st.count_b, cb);
  if (st.r.z) src_coeffs(st.r, st.cout, st.count_r, cr);
  // BN-backward coefficients of THIS stage's output: co_[c] = gamma*rstd, [C+c] = S1/n, [2C+c] = S2/n, mean/rstd after
  if (st.type != TY_PLAIN) {
    Src o; o.norm = NORM_BATCH; o.sums = st.sums_out; o.nrep = 1; o.gamma = st.gamma_out; o.beta = st.gamma_out; o.act = ACT_NONE;
    float* tmp = gbs + MAXC;  // 4*MAXC scratch
    src_coeffs(o, st.cout, st.count, tmp);
    __shared__ double brec[64];
    fold_record(st.bsums_out, st.nrep, brec);
    for (int c = threadIdx.x; c < st.cout; c += blockDim.x) {
      co_[c] = st.gamma_out[c] * tmp[3 * st.cout + c];
      co_[MAXC + c] = (float)(brec[c] / st.count);
      co_[2 * MAXC + c] = (float)(brec[MAXC + c] / st.count);
      co_[3 * MAXC + c] = tmp[2 * st.cout + c];                         // mean
      gbs[5 * MAXC + c] = tmp[3 * st.cout + c];                          // rstd
    }
  }
  float gwacc[UNET_MAXW];
#pragma unroll
  for (int i = 0; i < UNET_MAXW; ++i) gwacc[i] = 0.f;
  __syncthreads();
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    const float* za = st.a.z + (size_t)win * nin;
    const float* zb = st.b.z ? st.b.z + (size_t)win * nin : nullptr;
    for (int i = threadIdx.x; i < nin; i += blockDim.x) {
      const int c = i / st.lin;
      float v = src_value(st.a, ca, st.cin, c, za[i]);
      if (zb) v += src_value(st.b, cb, st.cin, c, zb[i]);
      in[i] = v;
    }
    // gradient at the conv output
    for (int i = threadIdx.x; i < nout; i += blockDim.x) {
      const int c = i / st.lout;
      float g = st.Gout[(size_t)win * nout + i];
      if (st.type != TY_PLAIN) {
        const float zo = st.out[(size_t)win * nout + i];
        const float zh = (zo - co_[3 * MAXC + c]) * gbs[5 * MAXC + c];
        g = co_[c] * (g - co_[MAXC + c] - zh * co_[2 * MAXC + c]);
        if (st.type == TY_ABN) g = zo > 0.f ? g : 0.01f * g;   // stored tensor is lrelu(conv)
      }
      dc[i] = g;
      atomicAdd(gbs + c, g);
    }
    __syncthreads();
    // weight gradients: thread owns entries e = tid + 256 j of the weight tensor
#pragma unroll
    for (int j = 0; j < UNET_MAXW; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e < nw) {
        int ci, co, k;
        if (st.mode == CONV) { co = e / (st.cin * st.ks); ci = (e / st.ks) % st.cin; k = e % st.ks; }
        else { ci = e / (st.cout * st.ks); co = (e / st.ks) % st.cout; k = e % st.ks; }
        const float* ir = in + ci * st.lin;
        const float* dr = dc + co * st.lout;
        float s = 0.f;
        if (st.mode == CONV) {
          for (int l = 0; l < st.lout; ++l) {
            const int p = l * st.stride - st.pad + k;
            if (p >= 0 && p < st.lin) s = fmaf(dr[l], ir[p], s);
          }
        } else {
          for (int i2 = 0; i2 < st.lin; ++i2) {
            const int jj = 2 * i2 - 1 + k;
            if (jj >= 0 && jj < st.lout) s = fmaf(ir[i2], dr[jj], s);
          }
        }
        gwacc[j] += s;
      }
    }
    // input gradient
    for (int i = threadIdx.x; i < nin; i += blockDim.x) {
      const int ci = i / st.lin, p = i - ci * st.lin;
      float acc = 0.f;
      if (st.mode == CONV) {
        for (int co = 0; co < st.cout; ++co) {
          const float* wr = ws + (co * st.cin + ci) * st.ks;
          const float* dr = dc + co * st.lout;
          for (int k = 0; k < st.ks; ++k) {
            const int t = p + st.pad - k;
            if (t >= 0 && t % st.stride == 0 && t / st.stride < st.lout) acc = fmaf(wr[k], dr[t / st.stride], acc);
          }
        }
      } else {
        for (int co = 0; co < st.cout; ++co) {
          const float* wr = ws + (ci * st.cout + co) * st.ks;
          const float* dr = dc + co * st.lout;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int jj = 2 * p - 1 + k;
            if (jj >= 0 && jj < st.lout) acc = fmaf(wr[k], dr[jj], acc);
          }
        }
      }
      // distribute to the producers of the input: G = d_in * act'(bn value) (+ accumulate), and their sums
      if (st.a.G) {
        const float bn = za[i] * ca[ci] + ca[st.cin + ci];
        float ga = (st.a.act == ACT_LRELU && bn <= 0.f) ? 0.01f * acc : acc;
        if (st.a.bsums) {
          const float zh = (za[i] - ca[2 * st.cin + ci]) * ca[3 * st.cin + ci];
          atomicAdd(sa + ci, ga); atomicAdd(sa + MAXC + ci, ga * zh);
        }
        float* dst = st.a.G + (size_t)win * nin + i;
        *dst = st.a.accumulate ? *dst + ga : ga;
      }
      if (zb && st.b.G) {
        const float bn = zb[i] * cb[ci] + cb[st.cin + ci];
        const float gbv = bn <= 0.f ? 0.01f * acc : acc;
        const float zh = (zb[i] - cb[2 * st.cin + ci]) * cb[3 * st.cin + ci];
        atomicAdd(sb + ci, gbv); atomicAdd(sb + MAXC + ci, gbv * zh);
        float* dst = st.b.G + (size_t)win * nin + i;
        *dst = st.b.accumulate ? *dst + gbv : gbv;
      }
    }
    // residual operand (added to the OUTPUT): its gradient is the plain output gradient
    if (st.r.z && st.r.G) {
      for (int i = threadIdx.x; i < nout; i += blockDim.x) {
        const int c = i / st.lout;
        const float zr = st.r.z[(size_t)win * nout + i];
        const float bn = zr * cr[c] + cr[st.cout + c];
        const float g0 = st.Gout[(size_t)win * nout + i];
        const float gr = bn <= 0.f ? 0.01f * g0 : g0;
        const float zh = (zr - cr[2 * st.cout + c]) * cr[3 * st.cout + c];
        atomicAdd(sr + c, gr); atomicAdd(sr + MAXC + c, gr * zh);
        float* dst = st.r.G + (size_t)win * nout + i;
        *dst = st.r.accumulate ? *dst + gr : gr;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < UNET_MAXW; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < nw) atomicAdd(st.gw + e, gwacc[j]);
  }
  if ((int)threadIdx.x < st.cout) atomicAdd(st.gb + threadIdx.x, gbs[threadIdx.x]);
  if ((int)threadIdx.x < st.cin) {
    if (st.a.G && st.a.bsums) {
      double* rec = st.a.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
      atomicAdd(rec + threadIdx.x, (double)sa[threadIdx.x]);
      atomicAdd(rec + MAXC + threadIdx.x, (double)sa[MAXC + threadIdx.x]);
    }
    if (st.b.z && st.b.G) {
      double* rec = st.b.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
      atomicAdd(rec + threadIdx.x, (double)sb[threadIdx.x]);
      atomicAdd(rec + MAXC + threadIdx.x, (double)sb[MAXC + threadIdx.x]);
    }
  }
  if (st.r.z && st.r.G && (int)threadIdx.x < st.cout) {
    double* rec = st.r.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
    atomicAdd(rec + threadIdx.x, (double)sr[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, (double)sr[MAXC + threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------------
// backward stage, specialised (same template parameters as k_unet_fwd_t).  A workgroup takes WP consecutive windows
// per pass (one pass at the launch sizes used) and touches global memory in ONE batched load phase:
//   1. load phase, every load of the pass in flight at once: the input operand(s) - kept twice in zero-haloed LDS rows,
//      transformed (BatchNorm + LeakyReLU + skip: the conv operand) and raw (the epilogue's activation derivative and
//      z-hat) -, the old value of a gradient tensor that is accumulated into, and the gradient at the conv output with
//      its BatchNorm-backward correction applied (+ bias gradient, + the residual operand's gradient);
//   2. input gradient d_in[ci][p] = sum over (co, k) W(co, ci, k) D(co, k, p) as fp32-MFMA tiles of 16 positions x 16
//      input channels (padded), both operands gathered from the LDS tiles by index (zero halos cover the taps outside;
//      no im2col copy); a lane ends with 4 consecutive positions of one input channel, the unit of the epilogue
//      (activation derivative, BatchNorm-backward sums of the producers, store) - which reads LDS only;
//   3. weight gradient gw(co, ci, k) += sum_p D[co][p] I(ci, k, p) as MFMA tiles of 16 output channels x 16 (ci, k)
//      pairs (padded), K = output positions; layers with fewer than four tiles split the positions over the waves.  The
//      accumulators live in registers across all windows of the workgroup.
// Every layer runs on the matrix pipe, the narrow ones with mostly-padding tiles: the FLOPs are free here (0.76 MFLOP
// per window), what the scalar loops they replace cost was LDS bank conflicts and same-address LDS atomics.
// The workgroup's weight / bias gradient partial leaves as plain stores into ITS row of a scratch matrix
// (st.part: workgroups x parameters), folded by k_unet_fold after the last stage - no global atomics except the 2 x C
// BatchNorm-backward sums the NEXT stage needs.  (st.part == nullptr: atomics straight into the gradient buffer.)
// ---------------------------------------------------------------------------------
#define UNET_BWD_THREADS 512
template <int CIN, int COUT, int KS, int MODE>
__global__ __launch_bounds__(UNET_BWD_THREADS, 2) void k_unet_bwd_t(Stage st, int B, int WP) {
  extern __shared__ float4 smem4[];
  UB_STAMP_INIT();
  constexpr int HALO = 4, nw = CIN * COUT * KS, NT = UNET_BWD_THREADS, NWAVE = NT / 64;
  // rows of the gradient tile: halo 4 | lout | halo 6.  The weight-gradient tiles read 16 ROWS per 4-byte access (lanes 0-31:
  // 16 output channels x 2 positions, 32 banks): a stride of 2 x an odd number keeps them in different banks (lout + 8 put
  // four rows in each bank); rows are therefore 8-byte aligned and written as 8-byte pieces
  const int lin = st.lin, lout = st.lout, LP = lin + 2 * HALO, LPO = lout + 2 * HALO + 2;
  const bool has_b = st.b.z != nullptr, want_din = st.a.G != nullptr;
  const bool acc_a = want_din && st.a.accumulate != 0;
  float* in = reinterpret_cast<float*>(smem4);                        // WP x CIN x LP   conv operand
  float* zra = in + WP * CIN * LP;                                    // WP x CIN x LP   raw z of operand a (want_din)
  float* aux = zra + (want_din ? WP * CIN * LP : 0);                  // WP x CIN x LP   raw z of operand b | old G_a
  float* dc = aux + ((has_b || acc_a) ? WP * CIN * LP : 0);           // WP x COUT x LPO gradient at the conv output
  // weights as in global memory ([co][ci][k] / ConvTranspose1d: [ci][co][k]); the ConvTranspose1d layers read one row per
  // lane in the input gradient and keep the rows at a padded stride (wrow_ld)
  constexpr int BWROW = COUT * KS, BWLD = (MODE == 2) ? wrow_ld(BWROW) : BWROW, BWSZ = (MODE == 2) ? ((CIN * BWLD + 3) & ~3) : nw;
  float* ws = dc + ((WP * COUT * LPO + 3) & ~3);  // weights (16-byte aligned)
  float* gws = ws + BWSZ;                        // weight-gradient partial of the workgroup
  float* ca = gws + nw; float* cb = ca + 4 * MAXC; float* cr = cb + 4 * MAXC; float* co_ = cr + 4 * MAXC;
  // backward BatchNorm sums of the operands / the residual and the bias gradient: doubles (see seg_atomic)
  double* sa = reinterpret_cast<double*>(co_ + 5 * MAXC); double* sb = sa + 2 * MAXC; double* sr = sb + 2 * MAXC;
  double* gbs = sr + 2 * MAXC;                   // MAXC bias grads
  // every global load of the prologue is requested before the first is waited for: the BatchNorm records and affine
  // parameters of the operands and of this stage's output (forward statistics and backward sums) and the weights
  Src o; o.norm = st.type != TY_PLAIN ? NORM_BATCH : NORM_NONE; o.sums = st.sums_out; o.nrep = 1; o.gamma = st.gamma_out; o.beta = st.gamma_out; o.act = ACT_NONE;
  const SrcReq la = src_request(st.a, CIN, st.w), lb = src_request(has_b ? st.b : st.a, CIN, st.w),
               lr = src_request(st.r.z ? st.r : st.a, st.r.z ? COUT : CIN, st.w), lo = src_request(o, COUT, st.w);
  const FoldPick fp = fold_pick(src_rec(st.a), st.a.nrep, has_b ? src_rec(st.b) : nullptr, st.b.nrep, st.r.z ? src_rec(st.r) : nullptr, st.r.nrep,
                                st.type != TY_PLAIN ? st.sums_out : nullptr, 1, st.type != TY_PLAIN ? st.bsums_out : nullptr, st.nrep, st.w);
  const FoldLd fl = fold_load(fp.rec, fp.nrep);        // (wave w: record w)
  constexpr int NWV = (nw / 4 + NT - 1) / NT;
  float4 wv[NWV];
#pragma unroll
  for (int k = 0; k < NWV; ++k) {
    const int i = threadIdx.x + k * NT;
    wv[k] = reinterpret_cast<const float4*>(st.w)[i < nw / 4 ? i : 0];
  }
#pragma unroll
  for (int k = 0; k < NWV; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < nw / 4) st_f4_as_f2(ws + (BWLD == BWROW ? 4 * i : (4 * i / BWROW) * BWLD + (4 * i) % BWROW), wv[k]);
  }
  for (int i = threadIdx.x; i < nw; i += NT) gws[i] = 0.f;
  for (int i = threadIdx.x; i < 7 * MAXC; i += NT) sa[i] = 0.;   // sa, sb, sr, gbs[0:MAXC]
  for (int i = threadIdx.x; i < WP * CIN * 2 * HALO; i += NT) {
    const int c = i / (2 * HALO), h = i % (2 * HALO);
    in[c * LP + (h < HALO ? h : lin + h)] = 0.f;
  }
  for (int i = threadIdx.x; i < WP * COUT * 16; i += NT) {
    const int c = i >> 4, h = i & 15;
    if (h < 2 * HALO + 2) dc[c * LPO + (h < HALO ? h : lout + h)] = 0.f;
  }
  if ((threadIdx.x >> 6) < UNET_FOLD_SLOTS) fold_store(fl, fp.nrep, threadIdx.x >> 6);
  __syncthreads();
  src_coeffs_slot(st.a, CIN, st.count_a, ca, la, 0, 0);          // (each record by its own 64-thread group)
  if (has_b) src_coeffs_slot(st.b, CIN, st.count_b, cb, lb, 1, 1);
  if (st.r.z) src_coeffs_slot(st.r, COUT, st.count_r, cr, lr, 2, 2);
  if (st.type != TY_PLAIN) {   // BN-backward coefficients of this stage's output, by thread group 3
    const int c = (int)threadIdx.x - 192;
    if (c >= 0 && c < COUT) {
      const double* rec = fold_rec(3);
      const double* brec = fold_rec(4);
      const double m = rec[c] / st.count;
      double var = rec[MAXC + c] / st.count - m * m;
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + 1e-5));
      co_[c] = lo.gamma * rstd;                                          // gamma * rstd
      co_[MAXC + c] = (float)(brec[c] / st.count);                       // mean(G)
      co_[2 * MAXC + c] = (float)(brec[MAXC + c] / st.count);            // mean(G * zhat)
      co_[3 * MAXC + c] = (float)m;                                      // mean
      co_[4 * MAXC + c] = rstd;                                          // rstd
    }
  }
  __syncthreads();
  const int nin = CIN * lin, nout = COUT * lout, nin4 = nin >> 2, nout4 = nout >> 2;
  const int sh_lin = pow2_shift(lin), sh_lout = pow2_shift(lout), sh_nin4 = pow2_shift(nin4), sh_nout4 = pow2_shift(nout4);
  const bool a_lrelu = st.a.act == ACT_LRELU;
  // lanes of a wave that hold the same output channel in the staging loop below (lout / 4 consecutive threads, when
  // that is a power of two and every wave of the loop is full): their sums are combined before the LDS atomic
  const int gq = lout >> 2;
  const int segw = ((gq & (gq - 1)) == 0 && gq >= 8 && nout4 % 64 == 0) ? (gq < 64 ? gq : 64) : 1;
  const int lane = threadIdx.x & 63, r = lane & 15, g4 = lane >> 4, wave = threadIdx.x >> 6;
  // weight-gradient work units of the workgroup: (tile, position range); layers with fewer tiles than waves split the
  // output positions KSPLIT ways so that every wave has one
  // Conv1d layers: rows = output channels, columns = (ci, k) pairs, contraction over OUTPUT positions.  ConvTranspose1d layers
  // (TDW): rows = (co, k) pairs, columns = input channels, contraction over INPUT positions i - gw[ci][co][k] = sum_i
  // in[ci][i] D[co][2 i - 1 + k] - half as many terms as over output positions, where every second one is a zero (the
  // decoder's weight gradients were 128 MFMAs per window and layer: 4-5 us of a stage at the fp32 MFMA's rate)
  constexpr bool TDW = MODE == 2;
  constexpr int DW_M = TDW ? COUT * KS : COUT, DW_N = TDW ? CIN : CIN * KS;
  constexpr int DW_MT = (DW_M + 15) / 16, DW_NT = (DW_N + 15) / 16, DW_NU = DW_MT * DW_NT;
  constexpr int KSPLIT = DW_NU >= NWAVE ? 1 : (NWAVE + DW_NU - 1) / DW_NU, DW_UNITS = DW_NU * KSPLIT, DW_TPW = (DW_UNITS + NWAVE - 1) / NWAVE;
  const int dw_klen = TDW ? lin : lout;
  const int kchunk = (((dw_klen + 15) >> 4) + KSPLIT - 1) / KSPLIT * 16;   // contraction positions per split (multiple of 16)
  f32x4 dwacc[DW_TPW];
#pragma unroll
  for (int ti = 0; ti < DW_TPW; ++ti) dwacc[ti] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool has_r = st.r.z && st.r.G;
  const bool has_z = st.type != TY_PLAIN;
  // input-gradient tiles: 16 positions x 16 input channels, K = (co, k) pairs in the order kk = 4 ks + (lane >> 4).  A wave
  // keeps ONE channel tile (tile index = wave + i NWAVE, channel tile = tile % CT), so everything that depends on the lane's
  // channel and k-pairs is a per-lane constant: the LDS offsets of both operands repeat with period DIN_P in the k-step
  // (k3: kk + 12 is the same tap four channels on) and the epilogue's BatchNorm coefficients are read once.
  constexpr int KTOT = COUT * KS, CT = (CIN + 15) / 16, KST = KTOT / 4;
  constexpr int DIN_P = (KS == 3) ? 3 : 1, DIN_CSTEP = 4 * DIN_P / KS, DIN_BSTEP = (MODE == 2) ? DIN_CSTEP * KS : DIN_CSTEP * CIN * KS;
  constexpr int DIN_KCH = KST < 12 ? (KST < 1 ? 1 : KST) : 12;
  static_assert(KTOT % 4 == 0 || CIN <= 2, "k-steps of the input-gradient tiles are whole");
  static_assert(NWAVE % CT == 0, "a wave keeps its channel tile");
  const int cibw = ((wave % CT) << 4) + r, cibc = cibw < CIN ? cibw : CIN - 1;
  const bool cokw = cibw < CIN;
  int abase[DIN_P], bbase[DIN_P];
  bool azero[DIN_P];
#pragma unroll
  for (int j = 0; j < DIN_P; ++j) {
    const int kk0 = 4 * j + g4, co0 = kk0 / KS, k = kk0 - co0 * KS, t0 = r + 1 - k;   // (MODE 0: position n0 + r meets output (pos + 1 - k) / 2)
    abase[j] = co0 * LPO + HALO + (MODE == 1 ? (KS - 1) / 2 - k : (MODE == 2 ? k - 1 : (t0 >> 1)));
    azero[j] = MODE == 0 && (t0 & 1);
    bbase[j] = (MODE == 2) ? cibc * BWLD + kk0 : (co0 * CIN + cibc) * KS + k;
  }
  const float4 eca = make_float4(ca[cibc], ca[CIN + cibc], ca[2 * CIN + cibc], ca[3 * CIN + cibc]);
  const float4 ecb = has_b ? make_float4(cb[cibc], cb[CIN + cibc], cb[2 * CIN + cibc], cb[3 * CIN + cibc]) : eca;
  UB_STAMP(16);
  for (int w0 = blockIdx.x * WP; w0 < B; w0 += gridDim.x * WP) {
    const int nwin = (B - w0) < WP ? (B - w0) : WP;
    // ---- load phase: all five streams of a chunk (NT x 2 float4 of each) are requested before any is used ----
    const float4* za = reinterpret_cast<const float4*>(st.a.z + (size_t)w0 * nin);
    const float4* zx = has_b ? reinterpret_cast<const float4*>(st.b.z + (size_t)w0 * nin)
                             : (acc_a ? reinterpret_cast<const float4*>(st.a.G + (size_t)w0 * nin) : nullptr);
    const float4* G4 = reinterpret_cast<const float4*>(st.Gout + (size_t)w0 * nout);
    const float4* Z4 = has_z ? reinterpret_cast<const float4*>(st.out + (size_t)w0 * nout) : nullptr;
    const float4* R4 = has_r ? reinterpret_cast<const float4*>(st.r.z + (size_t)w0 * nout) : nullptr;
    float4* RG4 = has_r ? reinterpret_cast<float4*>(st.r.G + (size_t)w0 * nout) : nullptr;
    auto put = [&](int i, float4 v, float4 u) {
      const int wi = qdiv(i, nin4, sh_nin4), e = (i - wi * nin4) << 2, c = qdiv(e, lin, sh_lin), p = e - c * lin;
      const int oo = (wi * CIN + c) * LP + HALO + p;
      if (want_din) *reinterpret_cast<float4*>(zra + oo) = v;
      if (zx) *reinterpret_cast<float4*>(aux + oo) = u;
      const float s_ = ca[c], h = ca[CIN + c];
      v = make_float4(v.x * s_ + h, v.y * s_ + h, v.z * s_ + h, v.w * s_ + h);
      if (a_lrelu) v = make_float4(lrelu01(v.x), lrelu01(v.y), lrelu01(v.z), lrelu01(v.w));
      if (has_b) {
        const float s2 = cb[c], h2 = cb[CIN + c];
        v.x += lrelu01(u.x * s2 + h2); v.y += lrelu01(u.y * s2 + h2); v.z += lrelu01(u.z * s2 + h2); v.w += lrelu01(u.w * s2 + h2);
      }
      *reinterpret_cast<float4*>(in + oo) = v;
    };
    auto putg = [&](int i, float4 g4v, float4 z4, float4 zr) {
      const int wi = qdiv(i, nout4, sh_nout4), e = (i - wi * nout4) << 2, c = qdiv(e, lout, sh_lout), p = e - c * lout;
      float g[4] = {g4v.x, g4v.y, g4v.z, g4v.w};
      if (has_z) {
        const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
        const float k = co_[c], m1 = co_[MAXC + c], m2 = co_[2 * MAXC + c], mu = co_[3 * MAXC + c], rs = co_[4 * MAXC + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float zh = (zz[j] - mu) * rs;
          float v = k * (g[j] - m1 - zh * m2);
          if (st.type == TY_ABN && zz[j] <= 0.f) v *= 0.01f;   // stored tensor is lrelu(conv)
          g[j] = v;
        }
      }
      st_f4_as_f2(dc + (wi * COUT + c) * LPO + HALO + p, make_float4(g[0], g[1], g[2], g[3]));
      seg_atomic(gbs + c, (g[0] + g[1]) + (g[2] + g[3]), segw);
      if (has_r) {   // residual operand lrelu(BN(z_r)) was added to the output: its gradient is Gout * lrelu'
        const float zrr[4] = {zr.x, zr.y, zr.z, zr.w}, gg[4] = {g4v.x, g4v.y, g4v.z, g4v.w};
        const float s_ = cr[c], h = cr[COUT + c], mu = cr[2 * COUT + c], rs = cr[3 * COUT + c];
        float gr[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gr[j] = (zrr[j] * s_ + h <= 0.f) ? 0.01f * gg[j] : gg[j];
          s1 += gr[j]; s2 += gr[j] * (zrr[j] - mu) * rs;
        }
        float4 outv = make_float4(gr[0], gr[1], gr[2], gr[3]);
        if (st.r.accumulate) outv = f4add(outv, RG4[i]);
        RG4[i] = outv;
        seg_atomic(sr + c, s1, segw); seg_atomic(sr + MAXC + c, s2, segw);
      }
    };
    {
      constexpr int U = 2;
      const int na = nwin * nin4, ng = nwin * nout4;                   // float4 per stream (equal here: every tensor of a window
      const int nmax = na > ng ? na : ng;                              //  has leads * L floats; kept general)
      for (int i0 = 0; i0 < nmax; i0 += U * NT) {
        float4 va[U], vx[U], vg[U], vz[U], vr[U];
        int ia[U], ig[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int i = i0 + k * NT + (int)threadIdx.x;
          ia[k] = i < na ? i : -1; ig[k] = i < ng ? i : -1;
          const int ja = i < na ? i : 0, jg = i < ng ? i : 0;            // (clamped: no lane-predicated loads)
          va[k] = za[ja];
          vx[k] = (zx ? zx : za)[ja];
          vg[k] = G4[jg];
          vz[k] = (has_z ? Z4 : G4)[jg];
          vr[k] = (has_r ? R4 : G4)[jg];
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (ia[k] >= 0) put(ia[k], va[k], vx[k]);
          if (ig[k] >= 0) putg(ig[k], vg[k], vz[k], vr[k]);
        }
      }
    }
    __syncthreads();
    UB_STAMP(17);
    // ---- input gradient ----
    if (want_din) {
      const int ptiles = (lin + 15) >> 4, sh_pt = pow2_shift(ptiles);
      for (int tile = wave; tile < nwin * ptiles * CT; tile += NWAVE) {
        const int tp = tile / CT, wi = qdiv(tp, ptiles, sh_pt), n0 = (tp - wi * ptiles) << 4;   // (channel tile: tile % CT == wave % CT)
        const int pos = n0 + r, posc = pos < lin ? pos : lin - 1;   // (rows past the end are computed on a clamped address and dropped)
        const float* da = dc + wi * COUT * LPO + (MODE == 1 ? posc : (MODE == 2 ? 2 * posc : (n0 >> 1)));
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // operands of up to twelve k-steps are fetched together (addresses: a per-lane base + a compile-time multiple of a
        // uniform step), then their MFMAs issue back to back on two accumulators
#pragma unroll
        for (int kc = 0; kc < KST; kc += DIN_KCH) {
          float av[DIN_KCH], bv[DIN_KCH];
#pragma unroll
          for (int u = 0; u < DIN_KCH; ++u) {
            const int ks = kc + u, j = ks % DIN_P, i = ks / DIN_P;
            if (ks < KST) {
              av[u] = da[abase[j] + i * DIN_CSTEP * LPO];
              if (MODE == 0 && azero[j]) av[u] = 0.f;
              bv[u] = ws[bbase[j] + i * DIN_BSTEP];
            }
          }
#pragma unroll
          for (int u = 0; u < DIN_KCH; ++u) {
            if (kc + u < KST) {
              if (u & 1) acc1 = mfma4(av[u], bv[u], acc1); else acc0 = mfma4(av[u], bv[u], acc0);
            }
          }
        }
        const f32x4 accv = acc0 + acc1;
        // epilogue of the lane's (input channel cibw, positions p0 .. p0 + 3): activation derivative, store (to one or two
        // producers), contributions to their BatchNorm-backward sums - all operands from the LDS
        const int p0 = n0 + 4 * g4;
        const bool valid = cokw && p0 < lin;
        float s1a = 0.f, s2a = 0.f, s1b = 0.f, s2b = 0.f;
        if (valid) {
          const int oo = (wi * CIN + cibw) * LP + HALO + p0;
          const size_t og = (size_t)(w0 + wi) * nin + (size_t)cibw * lin + p0;
          const float acc[4] = {accv[0], accv[1], accv[2], accv[3]};
          const float4 z4 = *reinterpret_cast<const float4*>(zra + oo);
          const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
          float ga[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            ga[j] = (a_lrelu && zz[j] * eca.x + eca.y <= 0.f) ? 0.01f * acc[j] : acc[j];
            s1a += ga[j]; s2a += ga[j] * (zz[j] - eca.z) * eca.w;
          }
          float4 outv = make_float4(ga[0], ga[1], ga[2], ga[3]);
          if (acc_a) outv = f4add(outv, *reinterpret_cast<const float4*>(aux + oo));
          *reinterpret_cast<float4*>(st.a.G + og) = outv;
          if (has_b && st.b.G) {
            const float4 y4 = *reinterpret_cast<const float4*>(aux + oo);
            const float yy[4] = {y4.x, y4.y, y4.z, y4.w};
            float gb4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              gb4[j] = (yy[j] * ecb.x + ecb.y <= 0.f) ? 0.01f * acc[j] : acc[j];
              s1b += gb4[j]; s2b += gb4[j] * (yy[j] - ecb.z) * ecb.w;
            }
            float4 ob = make_float4(gb4[0], gb4[1], gb4[2], gb4[3]);
            if (st.b.accumulate) ob = f4add(ob, *reinterpret_cast<const float4*>(st.b.G + og));
            *reinterpret_cast<float4*>(st.b.G + og) = ob;
          }
        }
        // the sums of a channel are added up across the four position groups of the tile (lanes r, r + 16, r + 32,
        // r + 48) before ONE LDS atomic per channel and tile: same-address LDS atomics of a wave run one after the other
        s1a = rows_sum(s1a); s2a = rows_sum(s2a);
        if (g4 == 0 && cokw && st.a.bsums) { atomicAdd(sa + cibw, (double)s1a); atomicAdd(sa + MAXC + cibw, (double)s2a); }
        if (has_b && st.b.G) {
          s1b = rows_sum(s1b); s2b = rows_sum(s2b);
          if (g4 == 0 && cokw) { atomicAdd(sb + cibw, (double)s1b); atomicAdd(sb + MAXC + cibw, (double)s2b); }
        }
      }
    }
    UB_STAMP(18);
    // ---- weight gradient ----
    // D rows / columns of padding lanes (co >= COUT, (ci, k) pair >= CIN KS) are dropped at the flush, so those lanes read a
    // clamped (valid, finite) address instead of being masked; the address of every read is a per-lane base + a
    // compile-time offset, advanced by a uniform step per 16 positions
#pragma unroll
    for (int ti = 0; ti < DW_TPW; ++ti) {
      const int unit = wave + ti * NWAVE;
      if (unit >= DW_UNITS) continue;
      const int tile = unit / KSPLIT, ks = unit - tile * KSPLIT;
      const int m0 = (tile / DW_NT) << 4, n0 = (tile % DW_NT) << 4;
      const int mm = m0 + r, nn = n0 + r;
      const int mc = mm < DW_M ? mm : 0, nc = nn < DW_N ? nn : 0;          // (padding lanes: clamped, dropped at the flush)
      const int q_lo = ks * kchunk, q_hi = (q_lo + kchunk) < dw_klen ? (q_lo + kchunk) : dw_klen;
      // A row: TDW: (co, k) = (mc / KS, mc % KS), element i = D[co][2 i - 1 + k]; else co = mc, element pp = D[co][pp]
      // B row: TDW: ci = nc, element i = in[ci][i]; else (ci, k) = (nc / KS, nc % KS), element pp = in[ci][f(pp, k)]
      const int ci = TDW ? nc : nc / KS, k = TDW ? mc % KS : nc - (nc / KS) * KS, co = TDW ? mc / KS : mc;
      const float* dr = dc + co * LPO + HALO + (TDW ? 2 * g4 - 1 + k : g4);
      const float* ir = in + ci * LP + HALO + (TDW ? g4 : (MODE == 1 ? g4 - (KS - 1) / 2 + k : 2 * g4 - 1 + k));
      constexpr int AQ = TDW ? 8 : 4;                                // floats of the A row per 4 contraction positions
      constexpr int BQ = TDW ? 4 : (MODE == 1 ? 4 : 8);              // floats of the B row per 4 contraction positions
      f32x4 accw = dwacc[ti], accx = {0.f, 0.f, 0.f, 0.f};
      for (int wi = 0; wi < nwin; ++wi) {
        const float* d2 = dr + wi * COUT * LPO;
        const float* i2 = ir + wi * CIN * LP;
        int q0 = q_lo;
        for (; q0 + 16 <= q_hi; q0 += 16) {
          float av[4], bv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            av[u] = d2[(q0 >> 2) * AQ + u * AQ];
            bv[u] = i2[(q0 >> 2) * BQ + u * BQ];
          }
          accw = mfma4(av[0], bv[0], accw); accx = mfma4(av[1], bv[1], accx);
          accw = mfma4(av[2], bv[2], accw); accx = mfma4(av[3], bv[3], accx);
        }
        if (q0 < q_hi) {   // ragged end (length not a multiple of 16)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool pk = q0 + 4 * u + g4 < q_hi;
            const float a_ = pk ? d2[(q0 >> 2) * AQ + u * AQ] : 0.f;
            const float b_ = pk ? i2[(q0 >> 2) * BQ + u * BQ] : 0.f;
            accw = mfma4(a_, b_, accw);
          }
        }
      }
      dwacc[ti] = accw + accx;
    }
    UB_STAMP(19);
    __syncthreads();
    UB_STAMP(20);
  }
  // ---- the workgroup's partial sums leave ----
  // weight-gradient tiles: lane (r, g) holds rows m0 + 4 g + v of column n0 + r; position splits of one tile meet in the LDS
  // copy of the weight tensor
#pragma unroll
  for (int ti = 0; ti < DW_TPW; ++ti) {
    const int unit = wave + ti * NWAVE;
    if (unit >= DW_UNITS) continue;
    const int tile = unit / KSPLIT;
    const int m0 = (tile / DW_NT) << 4, n0 = (tile % DW_NT) << 4, nn = n0 + r;
    if (nn >= DW_N) continue;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int mm = m0 + 4 * g4 + v;
      if (mm < DW_M) {
        // [co][ci][k] (Conv1d): row co, column (ci, k); [ci][co][k] (ConvTranspose1d): row (co, k), column ci
        float* dst = gws + (TDW ? nn * (COUT * KS) + mm : mm * CIN * KS + nn);
        if (KSPLIT == 1) *dst = dwacc[ti][v]; else atomicAdd(dst, dwacc[ti][v]);
      }
    }
  }
  __syncthreads();
  if (st.part) {
    float* row = st.part + (size_t)blockIdx.x * st.part_stride;
    for (int i = threadIdx.x; i < (nw >> 2); i += NT)
      reinterpret_cast<float4*>(row + st.part_w)[i] = reinterpret_cast<const float4*>(gws)[i];
    if ((int)threadIdx.x < COUT) row[st.part_b + threadIdx.x] = (float)gbs[threadIdx.x];
  } else {
    for (int i = threadIdx.x; i < nw; i += NT) atomicAdd(st.gw + i, gws[i]);
    if ((int)threadIdx.x < COUT) atomicAdd(st.gb + threadIdx.x, (float)gbs[threadIdx.x]);
  }
  if ((int)threadIdx.x < CIN) {
    if (st.a.G && st.a.bsums) {
      double* rec = st.a.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
      atomicAdd(rec + threadIdx.x, (double)sa[threadIdx.x]);
      atomicAdd(rec + MAXC + threadIdx.x, (double)sa[MAXC + threadIdx.x]);
    }
    if (st.b.z && st.b.G) {
      double* rec = st.b.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
      atomicAdd(rec + threadIdx.x, (double)sb[threadIdx.x]);
      atomicAdd(rec + MAXC + threadIdx.x, (double)sb[MAXC + threadIdx.x]);
    }
  }
  if (st.r.z && st.r.G && (int)threadIdx.x < COUT) {
    double* rec = st.r.bsums + (size_t)(blockIdx.x % st.nrep) * 64;
    atomicAdd(rec + threadIdx.x, (double)sr[threadIdx.x]);
    atomicAdd(rec + MAXC + threadIdx.x, (double)sr[MAXC + threadIdx.x]);
  }
  UB_STAMP(21);
}

// Fold of the per-workgroup weight / bias gradient partials (rows x stride scratch matrix written by k_unet_bwd_t) into the
// gradient buffer: a workgroup owns 64 consecutive columns of the conv-parameter column list `cols`, its four waves take a
// quarter of the rows each (256-byte row pieces), the quarters meet in LDS (a fixed summation order; the partials themselves
// still carry the order of their workgroup's LDS atomics).  The last workgroup writes the BatchNorm affine gradients from the backward
// sums (g_gamma = S2, g_beta = S1).
// `bsums`: the backward record in nrep replicas; `final_`: the caller-visible bn_sums record, written when they differ
struct BnGrad { const double* bsums; double* final_; float* gw; float* gb; int C; int nrep; };
struct BnGradAll { BnGrad l[10]; };
RAL_DEV void bn_affine_grads(const BnGrad& b, int c, double share) {
  if (c >= b.C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < b.nrep; ++k) { s1 += b.bsums[(size_t)k * 64 + c]; s2 += b.bsums[(size_t)k * 64 + MAXC + c]; }
  if (b.final_ != b.bsums) { b.final_[c] = s1; b.final_[MAXC + c] = s2; }
  b.gb[c] = (float)(s1 * share); b.gw[c] = (float)(s2 * share);
}
#define UNET_FOLD_WAVES 16
__global__ __launch_bounds__(64 * UNET_FOLD_WAVES) void k_unet_fold(const float* __restrict__ part, int64_t stride, int rows, const int* __restrict__ cols,
                                                   int ncols, float* __restrict__ grads, BnGradAll u, double share,
                                                   double* __restrict__ zero, int nzero) {
  if (blockIdx.x == gridDim.x - 1) {
    for (int t = threadIdx.x; t < 10 * MAXC; t += blockDim.x) bn_affine_grads(u.l[t / MAXC], t % MAXC, share);
    // the replica records of both directions are no longer needed (the forward ones were folded by k_unet_out, the backward
    // ones right here): zeroed for the next step, which then needs no fill kernel
    __syncthreads();
    for (int i = threadIdx.x; i < nzero; i += blockDim.x) zero[i] = 0.0;
    return;
  }
  // sixteen waves share the rows of 64 columns (eight 256-byte row pieces in flight per wave: four round trips at 512 rows;
  // with four waves it was sixteen and the kernel 13 us for 29 MB)
  __shared__ float red[UNET_FOLD_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int col = j < ncols ? cols[j] : -1;
  float acc = 0.f;
  if (col >= 0) {
    const int r0 = (int)((int64_t)rows * wave / UNET_FOLD_WAVES), r1 = (int)((int64_t)rows * (wave + 1) / UNET_FOLD_WAVES);
    int rr = r0;
    for (; rr + 8 <= r1; rr += 8) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = part[(size_t)(rr + k) * stride + col];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    }
    for (; rr < r1; ++rr) acc += part[(size_t)rr * stride + col];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && col >= 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < UNET_FOLD_WAVES; ++k) t += red[k][lane];
    grads[col] = t;
  }
}

// BatchNorm affine gradients of all layers from the backward sums: g_gamma = S2, g_beta = S1 (the path without the fold)
// `share` = local / global windows: under data parallelism the sums are global and the later gradient all-reduce adds
// the ranks' copies, so each rank contributes its share
__global__ void k_unet_bn_grads(BnGradAll u, double share) {
  bn_affine_grads(u.l[blockIdx.x], threadIdx.x, share);
}

// =================================================================================
// Fused inference forward (eval-mode BatchNorm): ONE kernel for the whole network.  A window's tensors are
// leads*L floats each, so all eleven stages of a window live in LDS (53 KB at L = 512: three workgroups per CU);
// HBM sees the input window once and the output window once (8 KB per window at 2 x 512 instead of ~100 KB of
// stage-granular traffic).  The seven wide layers (8->16 ... 16->8 channels) are GEMMs on the fp32 MFMA:
//   out[co][n] = sum_{ci,k} W[co][(ci,k)] X[n][(ci,k)]     X = im2col rows, written by the PRODUCING layer's epilogue
// (the epilogue applies bias / LeakyReLU / BatchNorm scale-shift / skip and scatters each value to the 1-3 im2col
// slots that read it); a transposed conv is two such GEMMs, one per output parity, over one shared row layout
// (row m = [in[m-1], in[m]] per channel: parity 0 reads row n, parity 1 row n + 1).  The four narrow layers
// (2->4, 4->8, 8->4, 4->2 channels) stay on the vector ALU.  Weights are re-packed into the GEMM layouts, and the
// BatchNorm running statistics folded into (scale, shift), by k_unet_pack right before (10 528 floats).
// Reference: model/UNet.py:46-141.
// =================================================================================
RAL_STAMPS_DEFINE(ral_debug_stamps_unet)

namespace uinf {
constexpr int PW2 = 0, PW3 = 512, PW4 = 2048, PW5 = 3072, PW6 = 6144, PW7 = 7168, PW8 = 9216;   // GEMM weights (M x K)
constexpr int PW0 = 9728, PW1 = 9760, PW9 = 9856, PW10 = 9984;                                  // vector-ALU layers
constexpr int PB = 10016;     // biases: b0@0 b1@4 b2@12 b3@28 b4@60 b5@92 b6@124 b7@156 b8@172 b9@180 b10@184
constexpr int PS = 10208;     // BatchNorm (scale[C], shift[C]) per layer at SO[i]
constexpr int PTOT = 10528;
constexpr int BO[11] = {0, 4, 12, 28, 60, 92, 124, 156, 172, 180, 184};
constexpr int SO[10] = {0, 8, 24, 56, 120, 184, 248, 280, 296, 304};
constexpr int BNC[10] = {4, 8, 16, 32, 32, 32, 16, 8, 4, 2};
}  // namespace uinf

struct UPackArgs {
  const float* params; const float* state; float* pk;
  int64_t w[11], b[11], bnw[10], bnb[10], run[10];
  int leads;
};

__global__ __launch_bounds__(256) void k_unet_pack(UPackArgs a) {   // one packed float per thread
  using namespace uinf;
  const float* P = a.params;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PTOT) return;
  const int lead16 = 16 * a.leads;
  float v = 0.f;
  // conv k3: PyTorch (cout, cin, 3) is already the GEMM row layout [(ci, k)]; layer 2 is padded from K = 24 to 32
  if (i < PW3) { const int r = i >> 5, c = i & 31; v = c < 24 ? P[a.w[2] + r * 24 + c] : 0.f; }
  else if (i < PW4) v = P[a.w[3] + (i - PW3)];
  else if (i < PW5) v = P[a.w[4] + (i - PW4)];
  else if (i < PW6) v = P[a.w[5] + (i - PW5)];
  else if (i < PW7) v = P[a.w[6] + (i - PW6)];
  // transposed conv (cin, cout, 4): out[2n] = w[..][1] in[n] + w[..][3] in[n-1];  out[2n+1] = w[..][2] in[n] + w[..][0] in[n+1]
  // column (ci, t) of im2col row m holds in[m - 1 + t]; parity 0 reads row n, parity 1 row n + 1
  else if (i < PW8) {
    const int j = i - PW7, par = j / 1024, co = (j / 64) % 16, ci = (j % 64) / 2, t = j & 1;
    v = P[a.w[7] + (ci * 16 + co) * 4 + (par == 0 ? (t ? 1 : 3) : (t ? 0 : 2))];
  } else if (i < PW0) {
    const int j = i - PW8, par = j / 256, co = (j / 32) % 8, ci = (j % 32) / 2, t = j & 1;
    v = P[a.w[8] + (ci * 8 + co) * 4 + (par == 0 ? (t ? 1 : 3) : (t ? 0 : 2))];
  }
  else if (i < PW1) { const int j = i - PW0; v = j < 4 * a.leads * 3 ? P[a.w[0] + j] : 0.f; }
  else if (i < PW9) v = P[a.w[1] + (i - PW1)];
  else if (i < PW10) v = P[a.w[9] + (i - PW9)];
  else if (i < PB) { const int j = i - PW10; v = j < lead16 ? P[a.w[10] + j] : 0.f; }
  else if (i < PS) {
    const int j = i - PB;
    const int bc[11] = {4, 8, 16, 32, 32, 32, 32, 16, 8, 4, a.leads};
    for (int l = 0; l < 11; ++l)
      if (j >= BO[l] && j < BO[l] + bc[l]) v = P[a.b[l] + (j - BO[l])];
  } else {
    const int j = i - PS;
    for (int l = 0; l < 10; ++l) {
      const int C = l == 9 ? a.leads : BNC[l];
      if (j >= SO[l] && j < SO[l] + 2 * C) {
        const int c = (j - SO[l]) % C;
        const float sc = P[a.bnw[l] + c] / sqrtf(a.state[a.run[l] + C + c] + 1e-5f);   // (as src_coeffs, NORM_RUNNING)
        v = (j - SO[l]) < C ? sc : P[a.bnb[l] + c] - a.state[a.run[l] + c] * sc;
      }
    }
  }
  a.pk[i] = v;
}

// One 16 x 16 output tile of a GEMM layer, in two halves so that the global weight-fragment loads of a layer can be
// issued BEFORE the barrier that ends the previous layer (their L2 latency then hides behind the other waves' epilogues):
//   uinf_frag  - the A fragments of rows m0.. of W (M x K row-major, global): K / 16 float4 per lane (k = 16 c + 4 g + s)
//   uinf_mma   - the MFMAs against im2col rows t0.. of Xs (LDS, stride ldx); epi(row0, row index, 4 channels of that row)
template <int K> struct UFrag { float4 w[K / 16]; };
template <int K>
RAL_DEV UFrag<K> uinf_frag(const float* __restrict__ W, int M, int m0) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  int mrow = m0 + r;
  if (mrow >= M) mrow = M - 1;          // rows >= M are computed on a duplicate and discarded
  UFrag<K> f;
#pragma unroll
  for (int c = 0; c < K / 16; ++c) f.w[c] = *reinterpret_cast<const float4*>(W + (size_t)mrow * K + c * 16 + 4 * g);
  return f;
}
template <int K, class Epi>
RAL_DEV void uinf_mma(const UFrag<K>& f, int M, const float* Xs, int ldx, int m0, int t0, Epi epi) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < K / 16; ++c) {
    const float4 x4 = *reinterpret_cast<const float4*>(Xs + (t0 + r) * ldx + c * 16 + 4 * g);
    acc = mfma4(f.w[c].x, x4.x, acc);
    acc = mfma4(f.w[c].y, x4.y, acc);
    acc = mfma4(f.w[c].z, x4.z, acc);
    acc = mfma4(f.w[c].w, x4.w, acc);
  }
  const int row0 = m0 + 4 * g;
  if (row0 < M) epi(row0, t0 + r, tofloat4(acc));
}
// a layer's units dealt to the 4 waves: the first unit's fragments were prefetched by the caller (pf), later ones load here
template <int K, class WOf, class Epi>
RAL_DEV void uinf_layer(const UFrag<K>& pf, int nunits, int M, WOf wof, Epi epi) {
  const int wave = threadIdx.x >> 6;
  for (int u = wave; u < nunits; u += 4) {
    const float* W; const float* Xs; int ldx, m0, t0, tag;
    wof(u, W, Xs, ldx, m0, t0, tag);
    if (u == wave) uinf_mma<K>(pf, M, Xs, ldx, m0, t0, [&](int row0, int n, float4 a) { epi(tag, row0, n, a); });
    else { const UFrag<K> f = uinf_frag<K>(W, M, m0); uinf_mma<K>(f, M, Xs, ldx, m0, t0, [&](int row0, int n, float4 a) { epi(tag, row0, n, a); }); }
  }
}

RAL_DEV float4 f4lrelu(float4 v) { return make_float4(lrelu01(v.x), lrelu01(v.y), lrelu01(v.z), lrelu01(v.w)); }
RAL_DEV float4 f4fma(float4 a, float4 b, float4 c) { return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)); }

template <int LEADS>
__global__ __launch_bounds__(256, 2) void k_unet_infer(const float* __restrict__ pk, const float* __restrict__ x,
                                                    float* __restrict__ y, int L, int B) {
  using namespace uinf;
  extern __shared__ float4 smem4[];
  const int N1 = L >> 1, N2 = L >> 2, N3 = L >> 3, N4 = L >> 4;
  constexpr int LD2 = 36, LD3 = 52, LD4 = 36, LD5 = 100, LD7 = 68, LD8 = 36;   // im2col row strides (K + 4)
  const int LP0 = N1 + 8;                      // E0c rows carry a zero halo of 4 on both sides
  float* E0c = reinterpret_cast<float*>(smem4);          // e0, channel-major 4 x LP0 (skip of the last-but-one layer)
  float* RA = E0c + 4 * LP0;                             // X2 (N3 x 36) | later X5 (N4 x 100)
  float* S1 = RA + N4 * LD5;                             // e1, token-major N2 x 8
  float* S2 = S1 + N2 * 8;                               // e2, token-major N3 x 16
  float* RB = S2 + N3 * 16;                              // x staging (LEADS x (L + 8)) | X3 (N4 x 52) | later X7 ((N4 + 1) x 68)
  float* S3 = RB + (N4 + 1) * LD7;                       // e3 = X4, token-major N4 x 36 | later d2, channel-major 4 x N1
  float* RC = S3 + N4 * LD4;                             // X6 (N4 x 36) | later d1, channel-major 8 x N2
  float* X8 = RC + N4 * LD4;                             // (N3 + 1) x 36
  float* WV = X8 + (N3 + 1) * LD8;                       // weights of the four vector-ALU layers (PW0 .. PW10 + 32)
  // biases and BatchNorm (scale, shift) pairs of all layers (PB .. PTOT, 512 floats).  They are read in the epilogues
  // through the LDS, not from `pk`: hipcc hoists every loop-invariant global load out of the window loop, and with the
  // 21 weight-fragment quads of the seven GEMM layers already living in registers across windows the ~20 epilogue quads
  // went to scratch memory (156 bytes per lane: 20 MB of spill stores per launch, more than the kernel's output)
  float* CB = WV + 288;
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 288; i += 256) WV[i] = pk[PW0 + i];
  for (int i = tid; i < PTOT - PB; i += 256) CB[i] = pk[PB + i];
  for (int i = tid; i < 32; i += 256) { const int c = i >> 3, h = i & 7; E0c[c * LP0 + (h < 4 ? h : N1 + h)] = 0.f; }
  for (int i = tid; i < 16; i += 256) { X8[i * 2] = 0.f; X8[N3 * LD8 + i * 2 + 1] = 0.f; }   // in[-1] and in[N3] of the shared rows
  const float* wv0 = WV, *wv1 = WV + (PW1 - PW0), *wv9 = WV + (PW9 - PW0), *wv10 = WV + (PW10 - PW0);
  const float* bias = CB;
  const float* bns = CB + (PS - PB);
  __syncthreads();
  RAL_STAMP_INIT();
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    RAL_STAMP_AT(0);
    // ---- stage x (zero halo), clear the im2col rows of layer 2 (their K padding and the in[-1] slot must be zeros) ----
    {
      const int LPX = L + 8;
      const float* xw = x + (size_t)win * LEADS * L;
      for (int i = tid; i < LEADS * (L >> 2); i += 256) {
        const int c = i / (L >> 2), p = (i - c * (L >> 2)) << 2;
        *reinterpret_cast<float4*>(RB + c * LPX + 4 + p) = *reinterpret_cast<const float4*>(xw + c * L + p);
      }
      for (int i = tid; i < LEADS * 8; i += 256) { const int c = i >> 3, h = i & 7; RB[c * LPX + (h < 4 ? h : L + h)] = 0.f; }
      for (int i = tid; i < (N3 * LD2) >> 2; i += 256) reinterpret_cast<float4*>(RA)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    RAL_STAMP_AT(1);
    // ---- layer 0: Conv1d(LEADS, 4, k3, s2, p1) -> BN -> LeakyReLU -> E0c ----
    for (int slot = tid; slot < N1; slot += 256) {      // 4 channels x N1 / 4 position quads
      const int q = N1 >> 2, co = slot / q, l0 = (slot - co * q) << 2;
      float acc[4] = {bias[BO[0] + co], bias[BO[0] + co], bias[BO[0] + co], bias[BO[0] + co]};
#pragma unroll
      for (int ci = 0; ci < LEADS; ++ci) {
        const float* row = RB + ci * (L + 8) + 4;
        const float* wr = wv0 + (co * LEADS + ci) * 3;
        // in[2 l0 - 1 .. 2 l0 + 7] as three aligned 16-byte reads (nine 4-byte reads at a lane stride of 8 floats are 8-way bank conflicts)
        const float4 qa = *reinterpret_cast<const float4*>(row + 2 * l0 - 4), qb = *reinterpret_cast<const float4*>(row + 2 * l0),
                     qc = *reinterpret_cast<const float4*>(row + 2 * l0 + 4);
        const float xv[9] = {qa.w, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wr[0], xv[2 * j], fmaf(wr[1], xv[2 * j + 1], fmaf(wr[2], xv[2 * j + 2], acc[j])));
      }
      const float sc = bns[SO[0] + co], sh = bns[SO[0] + 4 + co];
      *reinterpret_cast<float4*>(E0c + co * LP0 + 4 + l0) =
          make_float4(lrelu01(fmaf(acc[0], sc, sh)), lrelu01(fmaf(acc[1], sc, sh)), lrelu01(fmaf(acc[2], sc, sh)), lrelu01(fmaf(acc[3], sc, sh)));
    }
    __syncthreads();
    RAL_STAMP_AT(2);
    // ---- layer 1: Conv1d(4, 8, k3, s2, p1) -> BN -> LeakyReLU -> S1 (skip) and the im2col rows of layer 2 ----
    for (int slot = tid; slot < 2 * N2; slot += 256) {   // 8 channels x N2 / 4
      const int q = N2 >> 2, co = slot / q, l0 = (slot - co * q) << 2;
      float acc[4] = {bias[BO[1] + co], bias[BO[1] + co], bias[BO[1] + co], bias[BO[1] + co]};
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const float* row = E0c + ci * LP0 + 4;
        const float* wr = wv1 + (co * 4 + ci) * 3;
        // in[2 l0 - 1 .. 2 l0 + 7] as three aligned 16-byte reads (nine 4-byte reads at a lane stride of 8 floats are 8-way bank conflicts)
        const float4 qa = *reinterpret_cast<const float4*>(row + 2 * l0 - 4), qb = *reinterpret_cast<const float4*>(row + 2 * l0),
                     qc = *reinterpret_cast<const float4*>(row + 2 * l0 + 4);
        const float xv[9] = {qa.w, qb.x, qb.y, qb.z, qb.w, qc.x, qc.y, qc.z, qc.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = fmaf(wr[0], xv[2 * j], fmaf(wr[1], xv[2 * j + 1], fmaf(wr[2], xv[2 * j + 2], acc[j])));
      }
      const float sc = bns[SO[1] + co], sh = bns[SO[1] + 8 + co];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int p = l0 + j;
        const float v = lrelu01(fmaf(acc[j], sc, sh));
        S1[p * 8 + co] = v;
        // consumer (k3, s2, p1) reads in[2n - 1 + k]: odd p feeds (n = (p+1)/2, k = 0) and (n = (p-1)/2, k = 2); even p (n = p/2, k = 1)
        if (p & 1) { if (((p + 1) >> 1) < N3) RA[((p + 1) >> 1) * LD2 + co * 3] = v; RA[((p - 1) >> 1) * LD2 + co * 3 + 2] = v; }
        else RA[(p >> 1) * LD2 + co * 3 + 1] = v;
      }
    }
    for (int i = tid; i < 16; i += 256) RB[i * 3] = 0.f;          // X3 row 0, tap 0 = e2[-1] (x staging is dead)
    // unit -> (weights, im2col rows, row stride, first output row, first im2col row, tag) of every GEMM layer
    auto w2 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { W = pk + PW2; Xs = RA; ldx = LD2; m0 = 0; t0 = u * 16; tag = 0; };
    auto w3 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { W = pk + PW3; Xs = RB; ldx = LD3; m0 = (u & 1) * 16; t0 = (u >> 1) * 16; tag = 0; };
    auto w4 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { W = pk + PW4; Xs = S3; ldx = LD4; m0 = (u & 1) * 16; t0 = (u >> 1) * 16; tag = 0; };
    auto w5 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { W = pk + PW5; Xs = RA; ldx = LD5; m0 = (u & 1) * 16; t0 = (u >> 1) * 16; tag = 0; };
    auto w6 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { W = pk + PW6; Xs = RC; ldx = LD4; m0 = (u & 1) * 16; t0 = (u >> 1) * 16; tag = 0; };
    auto w7 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { tag = u & 1; W = pk + PW7 + tag * 1024; Xs = RB + tag * LD7; ldx = LD7; m0 = 0; t0 = (u >> 1) * 16; };
    auto w8 = [&](int u, const float*& W, const float*& Xs, int& ldx, int& m0, int& t0, int& tag) { tag = u & 1; W = pk + PW8 + tag * 256; Xs = X8 + tag * LD8; ldx = LD8; m0 = 0; t0 = (u >> 1) * 16; };
    const UFrag<32> f2 = uinf_frag<32>(pk + PW2, 16, 0);
    __syncthreads();
    RAL_STAMP_AT(3);
    // ---- layer 2: Conv1d(8, 16) as GEMM 16 x 32(24) over N3 rows -> BN -> LeakyReLU -> S2 (skip), X3 ----
    uinf_layer<32>(f2, N3 >> 4, 16, w2, [&](int, int row0, int p, float4 a) {
      const float4 v = f4lrelu(f4fma(f4add(a, *reinterpret_cast<const float4*>(bias + BO[2] + row0)),
                                     *reinterpret_cast<const float4*>(bns + SO[2] + row0), *reinterpret_cast<const float4*>(bns + SO[2] + 16 + row0)));
      *reinterpret_cast<float4*>(S2 + p * 16 + row0) = v;
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = row0 + e;
        if (p & 1) { if (((p + 1) >> 1) < N4) RB[((p + 1) >> 1) * LD3 + c * 3] = vv[e]; RB[((p - 1) >> 1) * LD3 + c * 3 + 2] = vv[e]; }
        else RB[(p >> 1) * LD3 + c * 3 + 1] = vv[e];
      }
    });
    const UFrag<48> f3 = uinf_frag<48>(pk + PW3, 32, (wave & 1) * 16);
    __syncthreads();
    RAL_STAMP_AT(4);
    // ---- layer 3: Conv1d(16, 32) as GEMM 32 x 48 -> BN -> LeakyReLU -> S3 (= im2col of the 1x1 conv, and the residual) ----
    uinf_layer<48>(f3, 2 * (N4 >> 4), 32, w3, [&](int, int row0, int p, float4 a) {
      *reinterpret_cast<float4*>(S3 + p * LD4 + row0) =
          f4lrelu(f4fma(f4add(a, *reinterpret_cast<const float4*>(bias + BO[3] + row0)),
                        *reinterpret_cast<const float4*>(bns + SO[3] + row0), *reinterpret_cast<const float4*>(bns + SO[3] + 32 + row0)));
    });
    for (int i = tid; i < 32; i += 256) { RA[i * 3] = 0.f; RA[(N4 - 1) * LD5 + i * 3 + 2] = 0.f; }   // X5: in[-1], in[N4] (X2 is dead)
    const UFrag<32> f4 = uinf_frag<32>(pk + PW4, 32, (wave & 1) * 16);
    __syncthreads();
    RAL_STAMP_AT(5);
    // ---- layer 4: bottleneck.0, 1x1 conv -> LeakyReLU -> BN -> X5 (k3, s1, p1 im2col) ----
    uinf_layer<32>(f4, 2 * (N4 >> 4), 32, w4, [&](int, int row0, int p, float4 a) {
      const float4 v = f4fma(f4lrelu(f4add(a, *reinterpret_cast<const float4*>(bias + BO[4] + row0))),
                             *reinterpret_cast<const float4*>(bns + SO[4] + row0), *reinterpret_cast<const float4*>(bns + SO[4] + 32 + row0));
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = row0 + e;
        RA[p * LD5 + c * 3 + 1] = vv[e];
        if (p + 1 < N4) RA[(p + 1) * LD5 + c * 3] = vv[e];
        if (p > 0) RA[(p - 1) * LD5 + c * 3 + 2] = vv[e];
      }
    });
    // (the largest fragment, 24 registers: re-loaded per window - an opaque pointer keeps hipcc from hoisting it out of
    // the window loop next to the other six layers' fragments, which is what pushed the kernel over 256 registers)
    const float* pk5 = pk + PW5;
    asm volatile("" : "+s"(pk5));
    const UFrag<96> f5 = uinf_frag<96>(pk5, 32, (wave & 1) * 16);
    __syncthreads();
    RAL_STAMP_AT(6);
    // ---- layer 5: bottleneck.3, k3 conv as GEMM 32 x 96 -> LeakyReLU -> BN -> X6 ----
    uinf_layer<96>(f5, 2 * (N4 >> 4), 32, w5, [&](int, int row0, int p, float4 a) {
      *reinterpret_cast<float4*>(RC + p * LD4 + row0) =
          f4fma(f4lrelu(f4add(a, *reinterpret_cast<const float4*>(bias + BO[5] + row0))),
                *reinterpret_cast<const float4*>(bns + SO[5] + row0), *reinterpret_cast<const float4*>(bns + SO[5] + 32 + row0));
    });
    for (int i = tid; i < 32; i += 256) { RB[i * 2] = 0.f; RB[N4 * LD7 + i * 2 + 1] = 0.f; }         // X7: in[-1], in[N4] (X3 is dead)
    const UFrag<32> f6 = uinf_frag<32>(pk + PW6, 32, (wave & 1) * 16);
    __syncthreads();
    RAL_STAMP_AT(7);
    // ---- layer 6: bottleneck.6, 1x1 conv, + e3 -> shared rows of the first transposed conv ----
    uinf_layer<32>(f6, 2 * (N4 >> 4), 32, w6, [&](int, int row0, int p, float4 a) {
      const float4 v = f4add(f4add(a, *reinterpret_cast<const float4*>(bias + BO[6] + row0)), *reinterpret_cast<const float4*>(S3 + p * LD4 + row0));
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { RB[p * LD7 + (row0 + e) * 2 + 1] = vv[e]; RB[(p + 1) * LD7 + (row0 + e) * 2] = vv[e]; }
    });
    const UFrag<64> f7 = uinf_frag<64>(pk + PW7 + (wave & 1) * 1024, 16, 0);
    __syncthreads();
    RAL_STAMP_AT(8);
    // ---- layer 7: ConvTranspose1d(32, 16) = two GEMMs 16 x 64 -> BN -> LeakyReLU, + e2 -> shared rows of layer 8 ----
    uinf_layer<64>(f7, 2 * (N4 >> 4), 16, w7, [&](int par, int row0, int n, float4 a) {
      const int j = 2 * n + par;
      const float4 v = f4add(f4lrelu(f4fma(f4add(a, *reinterpret_cast<const float4*>(bias + BO[7] + row0)),
                                           *reinterpret_cast<const float4*>(bns + SO[6] + row0), *reinterpret_cast<const float4*>(bns + SO[6] + 16 + row0))),
                             *reinterpret_cast<const float4*>(S2 + j * 16 + row0));
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { X8[j * LD8 + (row0 + e) * 2 + 1] = vv[e]; X8[(j + 1) * LD8 + (row0 + e) * 2] = vv[e]; }
    });
    const UFrag<32> f8 = uinf_frag<32>(pk + PW8 + (wave & 1) * 256, 8, 0);
    __syncthreads();
    RAL_STAMP_AT(9);
    // ---- layer 8: ConvTranspose1d(16, 8) = two GEMMs 8 x 32 -> BN -> LeakyReLU, + e1 -> d1 (channel-major, over X6) ----
    uinf_layer<32>(f8, 2 * (N3 >> 4), 8, w8, [&](int par, int row0, int n, float4 a) {
      const int j = 2 * n + par;
      const float4 v = f4add(f4lrelu(f4fma(f4add(a, *reinterpret_cast<const float4*>(bias + BO[8] + row0)),
                                           *reinterpret_cast<const float4*>(bns + SO[7] + row0), *reinterpret_cast<const float4*>(bns + SO[7] + 8 + row0))),
                             *reinterpret_cast<const float4*>(S1 + j * 8 + row0));
      RC[(row0 + 0) * N2 + j] = v.x; RC[(row0 + 1) * N2 + j] = v.y; RC[(row0 + 2) * N2 + j] = v.z; RC[(row0 + 3) * N2 + j] = v.w;
    });
    __syncthreads();
    RAL_STAMP_AT(10);
    // ---- layer 9: ConvTranspose1d(8, 4) -> BN -> LeakyReLU, + e0 -> d2 (channel-major, over e3) ----
    for (int slot = tid; slot < N1; slot += 256) {
      const int q = N1 >> 2, co = slot / q, l0 = (slot - co * q) << 2, h = l0 >> 1;
      float acc[4] = {bias[BO[9] + co], bias[BO[9] + co], bias[BO[9] + co], bias[BO[9] + co]};
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) {
        const float* row = RC + ci * N2;
        const float* wr = wv9 + (ci * 4 + co) * 4;
        const float x0 = h > 0 ? row[h - 1] : 0.f, x1 = row[h], x2 = row[h + 1], x3 = h + 2 < N2 ? row[h + 2] : 0.f;
        acc[0] = fmaf(wr[1], x1, fmaf(wr[3], x0, acc[0]));
        acc[1] = fmaf(wr[0], x2, fmaf(wr[2], x1, acc[1]));
        acc[2] = fmaf(wr[1], x2, fmaf(wr[3], x1, acc[2]));
        acc[3] = fmaf(wr[0], x3, fmaf(wr[2], x2, acc[3]));
      }
      const float sc = bns[SO[8] + co], sh = bns[SO[8] + 4 + co];
      const float4 sk = *reinterpret_cast<const float4*>(E0c + co * LP0 + 4 + l0);
      *reinterpret_cast<float4*>(S3 + co * N1 + l0) =
          make_float4(lrelu01(fmaf(acc[0], sc, sh)) + sk.x, lrelu01(fmaf(acc[1], sc, sh)) + sk.y,
                      lrelu01(fmaf(acc[2], sc, sh)) + sk.z, lrelu01(fmaf(acc[3], sc, sh)) + sk.w);
    }
    __syncthreads();
    RAL_STAMP_AT(11);
    // ---- layer 10: ConvTranspose1d(4, LEADS) -> BN -> y ----
    float* yw = y + (size_t)win * LEADS * L;
    for (int slot = tid; slot < LEADS * (L >> 2); slot += 256) {
      const int q = L >> 2, co = slot / q, l0 = (slot - co * q) << 2, h = l0 >> 1;
      float acc[4] = {bias[BO[10] + co], bias[BO[10] + co], bias[BO[10] + co], bias[BO[10] + co]};
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const float* row = S3 + ci * N1;
        const float* wr = wv10 + (ci * LEADS + co) * 4;
        const float x0 = h > 0 ? row[h - 1] : 0.f, x1 = row[h], x2 = row[h + 1], x3 = h + 2 < N1 ? row[h + 2] : 0.f;
        acc[0] = fmaf(wr[1], x1, fmaf(wr[3], x0, acc[0]));
        acc[1] = fmaf(wr[0], x2, fmaf(wr[2], x1, acc[1]));
        acc[2] = fmaf(wr[1], x2, fmaf(wr[3], x1, acc[2]));
        acc[3] = fmaf(wr[0], x3, fmaf(wr[2], x2, acc[3]));
      }
      const float sc = bns[SO[9] + co], sh = bns[SO[9] + LEADS + co];
      *reinterpret_cast<float4*>(yw + co * L + l0) = make_float4(fmaf(acc[0], sc, sh), fmaf(acc[1], sc, sh), fmaf(acc[2], sc, sh), fmaf(acc[3], sc, sh));
    }
    __syncthreads();
    RAL_STAMP_AT(12);
  }
}

static size_t uinf_lds_floats(int L) {
  const int N1 = L / 2, N2 = L / 4, N3 = L / 8, N4 = L / 16;
  return (size_t)4 * (N1 + 8) + (size_t)N4 * 100 + (size_t)N2 * 8 + (size_t)N3 * 16 + (size_t)(N4 + 1) * 68 + (size_t)N4 * 36 * 2 +
         (size_t)(N3 + 1) * 36 + 288 + (uinf::PTOT - uinf::PB);
}

// =================================================================================
// host model
// =================================================================================
struct UEntry { std::string name; int kind; int64_t offset; int ndim; int64_t shape[4]; };

struct ULayout {
  std::vector<UEntry> e;
  int64_t nparam = 0, nstate = 0;
  int64_t w[11], b[11];         // conv weights/biases: 0-3 enc, 4-6 bottleneck convs, 7-10 dec
  int64_t bnw[10], bnb[10], run[10];  // BN: 0-3 enc, 4-5 bottleneck, 6-9 dec
  int bnC[10];
  int ch[5];
};

static int64_t ualloc(int64_t& cur, int64_t n) { const int64_t o = cur; cur += (n + 3) & ~int64_t(3); return o; }
static void upush(ULayout& L, const std::string& n, int kind, int64_t off, std::initializer_list<int64_t> shp) {
  UEntry e; e.name = n; e.kind = kind; e.offset = off; e.ndim = (int)shp.size();
  int i = 0;
  for (auto s : shp) e.shape[i++] = s;
  for (; i < 4; ++i) e.shape[i] = 1;
  L.e.push_back(e);
}

static void ubuild(const ral_config& c, ULayout& L) {
  int64_t cur = 0, st = 0;
  const int ch[5] = {c.leads, 4, 8, 16, 32};
  memcpy(L.ch, ch, sizeof(ch));
  auto bn = [&](int idx, const std::string& pre, int C) {
    L.bnC[idx] = C;
    L.bnw[idx] = ualloc(cur, C); L.bnb[idx] = ualloc(cur, C);
    L.run[idx] = st; st += 2 * C;
    upush(L, pre + ".weight", RAL_PARAM, L.bnw[idx], {C});
    upush(L, pre + ".bias", RAL_PARAM, L.bnb[idx], {C});
    upush(L, pre + ".running_mean", RAL_STATE_F32, L.run[idx], {C});
    upush(L, pre + ".running_var", RAL_STATE_F32, L.run[idx] + C, {C});
    upush(L, pre + ".num_batches_tracked", RAL_COUNTER_I64, 0, {});
  };
  for (int i = 0; i < 4; ++i) {
    const std::string p = "EncList." + std::to_string(i);
    L.w[i] = ualloc(cur, ch[i + 1] * ch[i] * 3); L.b[i] = ualloc(cur, ch[i + 1]);
    upush(L, p + ".conv.weight", RAL_PARAM, L.w[i], {ch[i + 1], ch[i], 3});
    upush(L, p + ".conv.bias", RAL_PARAM, L.b[i], {ch[i + 1]});
    bn(i, p + ".bn", ch[i + 1]);
  }
  for (int i = 0; i < 4; ++i) {
    const std::string p = "DecList." + std::to_string(i);
    const int cin = ch[4 - i], cout = ch[3 - i];
    L.w[7 + i] = ualloc(cur, cin * cout * 4); L.b[7 + i] = ualloc(cur, cout);
    upush(L, p + ".conv.weight", RAL_PARAM, L.w[7 + i], {cin, cout, 4});
    upush(L, p + ".conv.bias", RAL_PARAM, L.b[7 + i], {cout});
    bn(6 + i, p + ".bn", cout);
  }
  const int ks[3] = {1, 3, 1};
  const char* cn[3] = {"bottleneck.0", "bottleneck.3", "bottleneck.6"};
  const char* bnn[2] = {"bottleneck.2", "bottleneck.5"};
  for (int i = 0; i < 3; ++i) {
    L.w[4 + i] = ualloc(cur, 32 * 32 * ks[i]); L.b[4 + i] = ualloc(cur, 32);
    upush(L, std::string(cn[i]) + ".weight", RAL_PARAM, L.w[4 + i], {32, 32, ks[i]});
    upush(L, std::string(cn[i]) + ".bias", RAL_PARAM, L.b[4 + i], {32});
    if (i < 2) bn(4 + i, bnn[i], 32);
  }
  L.nparam = cur; L.nstate = st;
}

struct UNetModel {
  UNetPublic pub;
  ULayout lay;
  char* slab = nullptr;
  float* z[11];       // conv outputs: 0-3 enc, 4 a4, 5 a5, 6 r, 7-10 dec (z6..z9 in the text above)
  float* pack = nullptr;   // fused inference: re-packed weights + folded BatchNorm coefficients (uinf::PTOT floats)
  int infer_grid = 0;      // resident workgroups of the fused inference kernel (occupancy query, first call)
  bool fused = true;       // eval forward as one kernel where it applies (ral_set_option "unet_fused"; RAL_UNET_FUSED=0)
  float* G[11];       // gradients at the BatchNorm outputs (same indexing; G[6] = d r)
  int C[11], Ln[11];  // channels / length of z[i]
  const float* last_x = nullptr;
  const float* last_dy = nullptr;   // gradient at the output BatchNorm = the caller's dy (read by the last stage's backward)
  const float* gsums_dy = nullptr;  // unet_forward_loss left the sums of THIS dy in the output layer's backward record ...
  int gsums_nrep = 0;               // ... spread over this many replicas (unet_backward_start then skips k_unet_gsums)
  int last_B = 0;
  // weight / bias gradient partials of the backward stage kernels: one row of `part_stride` floats (the flat parameter
  // layout) per workgroup, folded by k_unet_fold; `cols` = the flat offsets of all conv weights and biases
  float* part = nullptr; int64_t part_stride = 0; int part_rows_max = 0;
  int* cols = nullptr; int ncols = 0;
  bool fold = false; int bwd_rows = 0;
  // replicas of the BatchNorm records for the one-call forward / backward (see Stage::nrep): [fwd | bwd][10][UNET_MAXREP][64]
  double* rep = nullptr;
  int nrep_f = 1, nrep_b = 1;     // replicas the current forward / backward pass adds to (1: straight into bn_sums)
  bool rep_bwd_clean = false;     // the backward half of `rep` was zeroed by the forward pass's fill and not used since
  bool rep_all_clean = false;     // both halves were zeroed by the last backward pass's fold kernel and not used since
};

int unet_check_cfg(const ral_config* c, char* err, size_t cap) {
  if (c->leads != 1 && c->leads != 2) { snprintf(err, cap, "leads must be 1 or 2 (got %d)", c->leads); return -1; }
  if (c->L <= 0 || c->L % 16 != 0 || c->L > 2048) { snprintf(err, cap, "U-Net: L must be a multiple of 16 and <= 2048 (got %d)", c->L); return -1; }
  if (c->max_batch <= 0) { snprintf(err, cap, "max_batch must be positive"); return -1; }
  return 0;
}
int unet_layout_count(const ral_config* c) { ULayout L; ubuild(*c, L); return (int)L.e.size(); }
int unet_layout_entry(const ral_config* c, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset,
                      int32_t* ndim, int64_t shape[4]) {
  ULayout L; ubuild(*c, L);
  if (idx < 0 || idx >= (int)L.e.size() || (int)L.e[idx].name.size() + 1 > name_cap) return -1;
  strcpy(name, L.e[idx].name.c_str());
  *kind = L.e[idx].kind; *offset = L.e[idx].offset; *ndim = L.e[idx].ndim;
  for (int i = 0; i < 4; ++i) shape[i] = L.e[idx].shape[i];
  return 0;
}
int64_t unet_param_floats(const ral_config* c) { ULayout L; ubuild(*c, L); return L.nparam; }
int64_t unet_state_floats(const ral_config* c) { ULayout L; ubuild(*c, L); return L.nstate; }
#define UNET_PART_ROWS 1024   // most workgroups a backward stage is ever launched with
static size_t unet_plan(const ral_config& c, UNetModel* m, char* base) {
  const int ch[5] = {c.leads, 4, 8, 16, 32};
  const int Cs[11] = {4, 8, 16, 32, 32, 32, 32, 16, 8, 4, ch[0]};
  const int Ls[11] = {c.L / 2, c.L / 4, c.L / 8, c.L / 16, c.L / 16, c.L / 16, c.L / 16, c.L / 8, c.L / 4, c.L / 2, c.L};
  size_t cur = 0;
  for (int pass = 0; pass < (c.train ? 2 : 1); ++pass)
    for (int i = 0; i < 11; ++i) {
      const size_t bytes = ((size_t)c.max_batch * Cs[i] * Ls[i] * sizeof(float) + 255) & ~size_t(255);
      if (m) { (pass ? m->G : m->z)[i] = base ? reinterpret_cast<float*>(base + cur) : nullptr; m->C[i] = Cs[i]; m->Ln[i] = Ls[i]; }
      cur += bytes;
    }
  if (m) m->pack = base ? reinterpret_cast<float*>(base + cur) : nullptr;
  cur += ((size_t)uinf::PTOT * sizeof(float) + 255) & ~size_t(255);
  if (c.train) {
    ULayout Y; ubuild(c, Y);
    const int64_t stride = (Y.nparam + 63) & ~int64_t(63);
    const int rows = c.max_batch < UNET_PART_ROWS ? c.max_batch : UNET_PART_ROWS;
    if (m) { m->part = base ? reinterpret_cast<float*>(base + cur) : nullptr; m->part_stride = stride; m->part_rows_max = rows; }
    cur += ((size_t)rows * stride * sizeof(float) + 255) & ~size_t(255);
    if (m) m->cols = base ? reinterpret_cast<int*>(base + cur) : nullptr;
    cur += ((size_t)Y.nparam * sizeof(int) + 255) & ~size_t(255);
    if (m) m->rep = base ? reinterpret_cast<double*>(base + cur) : nullptr;
    cur += (size_t)2 * 10 * UNET_MAXREP * 64 * sizeof(double);
  }
  return cur;
}
int64_t unet_workspace_bytes(const ral_config* c) { return (int64_t)unet_plan(*c, nullptr, nullptr); }

UNetModel* unet_create(const ral_config* c, char* err, size_t cap) {
  UNetModel* m = new UNetModel();
  memset(&m->pub, 0, sizeof(m->pub));
  m->pub.cfg = *c;
  ubuild(*c, m->lay);
  m->pub.nparam = m->lay.nparam;
  const size_t bytes = unet_plan(*c, nullptr, nullptr);
  if (hipMalloc(reinterpret_cast<void**>(&m->slab), bytes) != hipSuccess) {
    snprintf(err, cap, "hipMalloc(%zu) failed", bytes);
    delete m;
    return nullptr;
  }
  unet_plan(*c, m, m->slab);
  m->fused = (ral_knob("UNET_FUSED", 1) != 0);
  if (c->train) {
    // the specialised backward kernels (all of them apply when every level's length is a multiple of 4) leave their
    // weight-gradient partials in scratch rows; otherwise some stage runs the generic kernel and everything stays atomic
    m->fold = c->L % 64 == 0 && (ral_knob("UNET_FOLD", 1) != 0);
    std::vector<int> cols;
    const int ch[5] = {c->leads, 4, 8, 16, 32};
    const int Cs[11] = {4, 8, 16, 32, 32, 32, 32, 16, 8, 4, ch[0]};
    const int Ci[11] = {ch[0], 4, 8, 16, 32, 32, 32, 32, 16, 8, 4};
    const int Ks[11] = {3, 3, 3, 3, 1, 3, 1, 4, 4, 4, 4};
    for (int si = 0; si < 11; ++si) {
      for (int i = 0; i < Cs[si] * Ci[si] * Ks[si]; ++i) cols.push_back((int)m->lay.w[si] + i);
      for (int i = 0; i < Cs[si]; ++i) cols.push_back((int)m->lay.b[si] + i);
    }
    m->ncols = (int)cols.size();
    if (hipMemcpy(m->cols, cols.data(), cols.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      snprintf(err, cap, "hipMemcpy(cols) failed");
      (void)hipFree(m->slab); delete m;
      return nullptr;
    }
  }
  return m;
}
void unet_destroy(UNetModel* u) { if (u) { if (u->slab) (void)hipFree(u->slab); delete u; } }
UNetPublic* unet_public(UNetModel* u) { return &u->pub; }
int unet_bind(UNetModel* u, float* params, float* grads, float* am, float* av, float* state, double* bn_sums) {
  u->pub.params = params; u->pub.grads = grads; u->pub.am = am; u->pub.av = av; u->pub.state = state; u->pub.bn_sums = bn_sums;
  return 0;
}

// BatchNorm index feeding each z tensor (z index -> bn index), -1: none (z[6] = r)
static const int BN_OF_Z[11] = {0, 1, 2, 3, 4, 5, -1, 6, 7, 8, 9};

static double* unet_rep(UNetModel* m, int dir, int bi) { return m->rep + ((size_t)(dir * 10 + bi) * UNET_MAXREP) * 64; }

static Src make_src(UNetModel* m, int zi, int act, bool training, bool with_grad, int accumulate) {
  Src s; memset(&s, 0, sizeof(s));
  const UNetPublic& P = m->pub;
  s.z = m->z[zi];
  const int bi = BN_OF_Z[zi];
  if (bi >= 0) {
    s.norm = training ? NORM_BATCH : NORM_RUNNING;
    // forward statistics: the stages of a running one-call forward read the replicas their producers add to; everything
    // after it (and every stage-by-stage caller) reads the folded record in bn_sums
    const bool frep = !with_grad && m->nrep_f > 1;
    s.sums = frep ? unet_rep(m, 0, bi) : (P.bn_sums ? P.bn_sums + 128 * bi : nullptr);
    s.nrep = frep ? m->nrep_f : 1;
    s.gamma = P.params + m->lay.bnw[bi]; s.beta = P.params + m->lay.bnb[bi];
    s.running = P.state + m->lay.run[bi];
    if (with_grad) s.bsums = m->nrep_b > 1 ? unet_rep(m, 1, bi) : P.bn_sums + 128 * bi + 64;
  } else {
    s.norm = NORM_NONE;
  }
  s.act = act;
  if (with_grad) { s.G = m->G[zi]; s.accumulate = accumulate; }
  return s;
}

static size_t fwd_lds(const Stage& s) {
  return ((size_t)s.cin * s.lin + (size_t)s.cin * s.cout * s.ks + MAXC + 12 * MAXC + 2 * MAXC + 8) * sizeof(float);
}
static size_t bwd_lds(const Stage& s) {
  return ((size_t)2 * s.cin * s.lin + (size_t)s.cout * s.lout + (size_t)s.cin * s.cout * s.ks + 16 * MAXC + 6 * MAXC + 7 * MAXC + 8) * sizeof(float);
}

// stage table: index 0-3 enc, 4-6 bottleneck, 7-10 dec
static Stage make_stage(UNetModel* m, int si, const float* x, bool training, bool bwd, int B, double gwin) {
  const ral_config& c = m->pub.cfg;
  const UNetPublic& P = m->pub;
  const ULayout& Y = m->lay;
  Stage s; memset(&s, 0, sizeof(s));
  const double cnt = gwin;   // windows behind every BatchNorm statistic: B, or the global batch under data parallelism
  auto cnt_of = [&](int zi) { return cnt * m->Ln[zi]; };
  s.w = P.params + Y.w[si]; s.bias = P.params + Y.b[si];
  s.out = m->z[si]; s.cout = m->C[si]; s.lout = m->Ln[si];
  const int bo = BN_OF_Z[si];
  // (forward: the record this stage adds to, in nrep replicas; backward: the folded forward record of its output)
  s.nrep = bwd ? m->nrep_b : m->nrep_f;
  s.sums_out = (bo >= 0 && P.bn_sums) ? ((!bwd && m->nrep_f > 1) ? unet_rep(m, 0, bo) : P.bn_sums + 128 * bo) : nullptr;
  s.count = cnt_of(si);
  s.stride = 1; s.pad = 0; s.mode = CONV;
  if (si <= 3) {                      // encoder: Conv1d(k3, s2, p1) -> BN -> LeakyReLU
    s.ks = 3; s.stride = 2; s.pad = 1; s.type = TY_ZBN;
    if (si == 0) { memset(&s.a, 0, sizeof(Src)); s.a.z = x; s.a.norm = NORM_NONE; s.a.act = ACT_NONE; s.cin = c.leads; s.lin = c.L; }
    else { s.a = make_src(m, si - 1, ACT_LRELU, training, bwd, 1); s.cin = m->C[si - 1]; s.lin = m->Ln[si - 1]; s.count_a = cnt_of(si - 1); }
  } else if (si == 4) {               // bottleneck.0: 1x1 conv -> LeakyReLU -> BN(2)
    s.ks = 1; s.post_lrelu = 1; s.type = TY_ABN;
    s.a = make_src(m, 3, ACT_LRELU, training, bwd, 1); s.cin = 32; s.lin = m->Ln[3]; s.count_a = cnt_of(3);
  } else if (si == 5) {               // bottleneck.3: k3 conv -> LeakyReLU -> BN(5)
    s.ks = 3; s.pad = 1; s.post_lrelu = 1; s.type = TY_ABN;
    s.a = make_src(m, 4, ACT_NONE, training, bwd, 0); s.cin = 32; s.lin = m->Ln[4]; s.count_a = cnt_of(4);
  } else if (si == 6) {               // bottleneck.6: 1x1 conv, + x (x = lrelu(BN3(z3)))
    s.ks = 1; s.type = TY_PLAIN;
    s.a = make_src(m, 5, ACT_NONE, training, bwd, 0); s.cin = 32; s.lin = m->Ln[5]; s.count_a = cnt_of(5);
    s.r = make_src(m, 3, ACT_LRELU, training, bwd, 0); s.count_r = cnt_of(3);
  } else {                            // decoder: ConvTranspose1d(k4, s2, p1) -> BN -> [LeakyReLU] (+ skip in the consumer)
    s.ks = 4; s.mode = CONVT; s.type = TY_ZBN;
    const int i = si - 7;
    if (i == 0) { s.a = make_src(m, 6, ACT_NONE, training, bwd, 0); }
    else {
      s.a = make_src(m, si - 1, ACT_LRELU, training, bwd, 0);
      s.b = make_src(m, 3 - i, ACT_LRELU, training, bwd, 0);   // feats[2 - (i-1)] = enc stage (3 - i)
      s.count_b = cnt_of(3 - i);
    }
    s.cin = m->C[si - 1]; s.lin = m->Ln[si - 1]; s.count_a = cnt_of(si - 1);
  }
  if (bwd) {
    s.Gout = m->G[si];
    s.bsums_out = bo >= 0 ? (m->nrep_b > 1 ? unet_rep(m, 1, bo) : P.bn_sums + 128 * bo + 64) : nullptr;
    s.gamma_out = bo >= 0 ? P.params + Y.bnw[bo] : nullptr;
    s.gw = P.grads + Y.w[si]; s.gb = P.grads + Y.b[si];
    if (si == 10) s.Gout = m->last_dy;        // the gradient at the output BatchNorm is the caller's dy itself
    if (m->fold) { s.part = m->part; s.part_stride = m->part_stride; s.part_w = (int)Y.w[si]; s.part_b = (int)Y.b[si]; }
  }
  return s;
}


// residual operand r of the specialised kernel is always lrelu(BN(z)) (bottleneck.6 + x); main operand act is a runtime flag
template <int CIN, int COUT, int KS, int MODE>
static void launch_fwd_t(const Stage& st, int B, int grid, hipStream_t s) {
  // windows per workgroup pass: all of a workgroup's windows at once (up to 4), so the grid is B / WP workgroups - fewer
  // while the tiles of a pass would not leave room for two workgroups per CU (long windows: at L = 2048 the 32-channel
  // stages take 64 KB at WP = 2 already); dynamic LDS above 64 KB is opted into per instantiation
  auto lds_of = [&](int WP) {
    return ((size_t)WP * CIN * (st.lin + 8) + (st.r.z ? (size_t)WP * COUT * st.lout : 0) + (size_t)COUT * (CIN * KS + 4) + MAXC +
            12 * MAXC + 4 * MAXC + 8) * sizeof(float);   // (weights: rows padded by up to 4 floats; the column sums are doubles)
  };
  int WP = (B + grid - 1) / grid;
  if (WP > 4) WP = 4;
  if (WP < 1) WP = 1;
  while (WP > 1 && lds_of(WP) > 80 * 1024) --WP;
  const int nwg = (B + WP - 1) / WP;
  const int g2 = nwg < grid ? nwg : grid;         // (the kernel loops over passes)
  const size_t lds = lds_of(WP);
  static size_t cur = 0;                          // (one per instantiation)
  if (lds > cur) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_unet_fwd_t<CIN, COUT, KS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); cur = lds; }
  k_unet_fwd_t<CIN, COUT, KS, MODE><<<g2, UNET_FWD_THREADS, lds, s>>>(st, B, WP);
}

static bool launch_unet_fwd_fast(const Stage& st, int si, int leads, int B, int grid, hipStream_t s) {
  if (st.lout % 4 || st.lin % 4) return false;
  switch (si) {
    case 0: if (leads == 1) launch_fwd_t<1, 4, 3, 0>(st, B, grid, s); else launch_fwd_t<2, 4, 3, 0>(st, B, grid, s); return true;
    case 1: launch_fwd_t<4, 8, 3, 0>(st, B, grid, s); return true;
    case 2: launch_fwd_t<8, 16, 3, 0>(st, B, grid, s); return true;
    case 3: launch_fwd_t<16, 32, 3, 0>(st, B, grid, s); return true;
    case 4: launch_fwd_t<32, 32, 1, 1>(st, B, grid, s); return true;
    case 5: launch_fwd_t<32, 32, 3, 1>(st, B, grid, s); return true;
    case 6: launch_fwd_t<32, 32, 1, 1>(st, B, grid, s); return true;
    case 7: launch_fwd_t<32, 16, 4, 2>(st, B, grid, s); return true;
    case 8: launch_fwd_t<16, 8, 4, 2>(st, B, grid, s); return true;
    case 9: launch_fwd_t<8, 4, 4, 2>(st, B, grid, s); return true;
    case 10: if (leads == 1) launch_fwd_t<4, 1, 4, 2>(st, B, grid, s); else launch_fwd_t<4, 2, 4, 2>(st, B, grid, s); return true;
  }
  return false;
}

// ---- forward, one stage at a time (a data-parallel caller all-reduces the stage's BatchNorm sums in between) ----
int unet_forward_stage(UNetModel* m, const float* x, int B, int training, int si, int64_t gwin, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.params || !P.state) { snprintf(err, cap, "ral_bind was not called"); return -1; }
  if (B <= 0 || B > P.cfg.max_batch) { snprintf(err, cap, "batch %d outside (0, %d]", B, P.cfg.max_batch); return -1; }
  if (si < 0 || si > 10) { snprintf(err, cap, "U-Net stage %d outside [0, 10]", si); return -1; }
  if (training && (!P.cfg.train || !P.bn_sums)) { snprintf(err, cap, "training forward needs train=1 and bn_sums"); return -1; }
  if (si == 0) {
    m->last_x = x; m->last_B = B; m->gsums_dy = nullptr;
    if (training) {
      // (one fill for both halves of the replica records: the backward pass of this step finds its half zeroed)
      if (m->nrep_f > 1) {
        if (!m->rep_all_clean) (void)hipMemsetAsync(unet_rep(m, 0, 0), 0, (size_t)2 * 10 * UNET_MAXREP * 64 * sizeof(double), s);
        m->rep_all_clean = false; m->rep_bwd_clean = true;
      }
      else (void)hipMemsetAsync(P.bn_sums, 0, 1280 * sizeof(double), s);
    }
  } else if (x != m->last_x || B != m->last_B) { snprintf(err, cap, "U-Net stages must follow stage 0 of the same batch"); return -1; }
  static const int fgmax = (int)ral_knob("UNET_FWD_GRID", 512);   // (train forward at batch 2048: 0.31 / 0.27 / 0.29 / 0.41 ms with 256 / 512 / 1024 / 2048 workgroups)
  static const int egmax = (int)ral_knob("UNET_EVAL_GRID", 512);   // (stage-by-stage eval forward at batch 2048: 173 / 158 / 177 us with 256 / 512 / 1024 workgroups)
  const int gcap = training ? fgmax : egmax;
  const int grid = B < gcap ? B : gcap;
  Stage st = make_stage(m, si, x, training != 0, false, B, (double)gwin);
  if (!training) st.sums_out = nullptr;
  if (!launch_unet_fwd_fast(st, si, P.cfg.leads, B, grid, s)) k_unet_fwd<<<grid, 256, fwd_lds(st), s>>>(st, B);
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net forward launch failed"); return -1; }
  return 0;
}

int unet_forward_finish(UNetModel* m, float* y, int B, int training, int64_t gwin, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (B != m->last_B) { snprintf(err, cap, "finish batch %d != stage batch %d", B, m->last_B); return -1; }
  Src o = make_src(m, 10, ACT_NONE, training != 0, false, 0);
  const size_t total = (size_t)B * m->C[10] * m->Ln[10];
  BnUpdAll u;
  memset(&u, 0, sizeof(u));
  if (training) {
    for (int zi = 0, k = 0; zi < 11; ++zi) {
      const int bi = BN_OF_Z[zi];
      if (bi < 0) continue;
      u.l[k++] = BnUpd{m->nrep_f > 1 ? unet_rep(m, 0, bi) : P.bn_sums + 128 * bi, P.bn_sums + 128 * bi, P.state + m->lay.run[bi],
                       m->C[zi], (double)gwin * m->Ln[zi], m->nrep_f};
    }
  }
  // (the running update reads the layers' complete records: every stage kernel has finished; the output layer's record is
  // read by this kernel's own coefficient fold, not written)
  k_unet_out<<<(int)((total / 4 + 255) / 256 < 1024 ? (total / 4 + 255) / 256 : 1024), 256, 0, s>>>(o, y, m->C[10], m->Ln[10],
                                                                                               (double)gwin * m->Ln[10], total, u, training ? 1 : 0);
  m->nrep_f = 1;
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net forward launch failed"); return -1; }
  return 0;
}

// The eval-mode forward as one kernel (+ the weight re-pack): windows of 256 / 512 / 768 / 1024 samples (the GEMM layers
// need whole 16-row tiles at the deepest level and the window's tensors must fit LDS).  RAL_UNET_FUSED=0 keeps the
// stage-by-stage path (which serves every other length, and training).
static bool unet_infer_fused_applies(const UNetModel* m) {
  const ral_config& c = m->pub.cfg;
  return m->fused && c.L % 256 == 0 && c.L <= 1024;
}
int unet_set_option(UNetModel* m, const char* key, int value) {
  if (!strcmp(key, "unet_fused")) { m->fused = value != 0; return 0; }
  return -1;
}

static int unet_forward_fused(UNetModel* m, const float* x, float* y, int B, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.params || !P.state) { snprintf(err, cap, "ral_bind was not called"); return -1; }
  if (B <= 0 || B > P.cfg.max_batch) { snprintf(err, cap, "batch %d outside (0, %d]", B, P.cfg.max_batch); return -1; }
  UPackArgs a;
  a.params = P.params; a.state = P.state; a.pk = m->pack; a.leads = P.cfg.leads;
  for (int i = 0; i < 11; ++i) { a.w[i] = m->lay.w[i]; a.b[i] = m->lay.b[i]; }
  for (int i = 0; i < 10; ++i) { a.bnw[i] = m->lay.bnw[i]; a.bnb[i] = m->lay.bnb[i]; a.run[i] = m->lay.run[i]; }
  k_unet_pack<<<(uinf::PTOT + 255) / 256, 256, 0, s>>>(a);
  const size_t lds = uinf_lds_floats(P.cfg.L) * sizeof(float);
  // persistent grid = exactly the workgroups that are resident at once (a grid one LDS granule too optimistic runs its
  // surplus workgroups as a second round: measured 76 us instead of 45 us at batch 2048)
  const void* kfn = P.cfg.leads == 1 ? reinterpret_cast<const void*>(k_unet_infer<1>) : reinterpret_cast<const void*>(k_unet_infer<2>);
  (void)hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (m->infer_grid == 0) {
    int per_cu = 0, dev = 0, ncu = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    const hipError_t e = P.cfg.leads == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_unet_infer<1>, 256, lds)
                                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_unet_infer<2>, 256, lds);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    if (ral_knob("UNET_DEBUG", 0)) fprintf(stderr, "k_unet_infer: lds %zu B, occupancy query -> %d workgroups per CU (rc %d), %d CUs\n", lds, per_cu, (int)e, ncu);
    { const int v = (int)ral_knob("UNET_WG_PER_CU", 0); if (v > 0) per_cu = v; }
    m->infer_grid = per_cu * ncu;
  }
  const int grid = B < m->infer_grid ? B : m->infer_grid;
  if (P.cfg.leads == 1) k_unet_infer<1><<<grid, 256, lds, s>>>(m->pack, x, y, P.cfg.L, B);
  else k_unet_infer<2><<<grid, 256, lds, s>>>(m->pack, x, y, P.cfg.L, B);
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net fused forward launch failed"); return -1; }
  return 0;
}

static int unet_nrep() {
  static const int n = [] { int k = (int)ral_knob("UNET_NREP", UNET_MAXREP); return k < 1 ? 1 : (k > UNET_MAXREP ? UNET_MAXREP : k); }();
  return n;
}

int unet_forward(UNetModel* m, const float* x, float* y, int B, int training, hipStream_t s, char* err, size_t cap) {
  if (!training && unet_infer_fused_applies(m)) return unet_forward_fused(m, x, y, B, s, err, cap);
  m->nrep_f = (training && m->pub.cfg.train) ? unet_nrep() : 1;
  for (int si = 0; si < 11; ++si)
    if (unet_forward_stage(m, x, B, training, si, B, s, err, cap)) { m->nrep_f = 1; return -1; }
  const int rc = unet_forward_finish(m, y, B, training, B, s, err, cap);
  m->nrep_f = 1;
  return rc;
}

int unet_stage_bn(int si) { return (si >= 0 && si <= 10) ? BN_OF_Z[si] : -1; }


template <int CIN, int COUT, int KS, int MODE>
static void launch_bwd_t(const Stage& st, int B, int grid, int wp_req, hipStream_t s) {
  const bool want_din = st.a.G != nullptr, third = st.b.z != nullptr || (want_din && st.a.accumulate);
  auto lds_of = [&](int WP) {
    return ((size_t)WP * CIN * (st.lin + 8) * (1 + (want_din ? 1 : 0) + (third ? 1 : 0)) + (size_t)WP * COUT * (st.lout + 10) +
            (size_t)2 * CIN * COUT * KS + 4 * CIN + 31 * MAXC + 12) * sizeof(float);   // (weights: rows padded by up to 4 floats; 7 MAXC doubles of sums)
  };
  int WP = (B + grid - 1) / grid;                 // windows per workgroup
  if (WP > wp_req) WP = wp_req;
  while (WP > 1 && lds_of(WP) > 80 * 1024) WP = WP > 2 ? 2 : 1;  // two workgroups per CU; passes of equal size (4, 2 or 1 windows)
  const size_t lds = lds_of(WP);
  static size_t cur = 0;                          // (one per instantiation)
  if (lds > cur) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_unet_bwd_t<CIN, COUT, KS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); cur = lds; }
  k_unet_bwd_t<CIN, COUT, KS, MODE><<<grid, UNET_BWD_THREADS, lds, s>>>(st, B, WP);
}

static bool launch_unet_bwd_fast(const Stage& st, int si, int leads, int B, int grid, int wp, hipStream_t s) {
  if (st.lout % 4 || st.lin % 4) return false;
  switch (si) {
    case 0: if (leads == 1) launch_bwd_t<1, 4, 3, 0>(st, B, grid, wp, s); else launch_bwd_t<2, 4, 3, 0>(st, B, grid, wp, s); return true;
    case 1: launch_bwd_t<4, 8, 3, 0>(st, B, grid, wp, s); return true;
    case 2: launch_bwd_t<8, 16, 3, 0>(st, B, grid, wp, s); return true;
    case 3: launch_bwd_t<16, 32, 3, 0>(st, B, grid, wp, s); return true;
    case 4: launch_bwd_t<32, 32, 1, 1>(st, B, grid, wp, s); return true;
    case 5: launch_bwd_t<32, 32, 3, 1>(st, B, grid, wp, s); return true;
    case 6: launch_bwd_t<32, 32, 1, 1>(st, B, grid, wp, s); return true;
    case 7: launch_bwd_t<32, 16, 4, 2>(st, B, grid, wp, s); return true;
    case 8: launch_bwd_t<16, 8, 4, 2>(st, B, grid, wp, s); return true;
    case 9: launch_bwd_t<8, 4, 4, 2>(st, B, grid, wp, s); return true;
    case 10: if (leads == 1) launch_bwd_t<4, 1, 4, 2>(st, B, grid, wp, s); else launch_bwd_t<4, 2, 4, 2>(st, B, grid, wp, s); return true;
  }
  return false;
}

// ---- backward, one stage at a time: stage si needs the BatchNorm-backward sums of ITS OUTPUT's BatchNorm, which
// the backward of its consumers (stages > si, and unet_backward_start for the last layer) has accumulated ----
// the ten backward halves [64, 128) of the 128-double BatchNorm records, one strided fill
static void unet_zero_bwd_records(UNetModel* m, hipStream_t s) {
  if (m->nrep_b > 1) {
    if (!m->rep_bwd_clean) (void)hipMemsetAsync(unet_rep(m, 1, 0), 0, (size_t)10 * UNET_MAXREP * 64 * sizeof(double), s);
    m->rep_bwd_clean = false; m->rep_all_clean = false;
  } else (void)hipMemset2DAsync(m->pub.bn_sums + 64, 128 * sizeof(double), 0, 64 * sizeof(double), 10, s);
}

// Training forward + loss + metrics in one call (ral_forward_loss_means): the stage kernels, then ONE kernel for the output
// BatchNorm, the loss sums, dy and the backward pass's first two sums (k_unet_out_loss).  Whole-batch statistics only (the
// data-parallel path drives the stages itself); the sums are left where unet_backward(dy) looks for them.
int unet_forward_loss(UNetModel* m, const float* x, const float* target, float* y, int B, float* dy, float* snr, float* rmse,
                      double* loss_sum, double* fin, double fin_scale, int fin3, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.cfg.train || !P.grads || !P.bn_sums) { snprintf(err, cap, "forward + loss needs train=1, grads and bn_sums bound"); return -1; }
  if (!target || !y || !dy) { snprintf(err, cap, "forward + loss: null pointer"); return -1; }
  m->nrep_f = unet_nrep();
  for (int si = 0; si < 11; ++si)
    if (unet_forward_stage(m, x, B, 1, si, B, s, err, cap)) { m->nrep_f = 1; return -1; }
  Src o = make_src(m, 10, ACT_NONE, true, false, 0);
  BnUpdAll u;
  memset(&u, 0, sizeof(u));
  for (int zi = 0, k = 0; zi < 11; ++zi) {
    const int bi = BN_OF_Z[zi];
    if (bi < 0) continue;
    u.l[k++] = BnUpd{m->nrep_f > 1 ? unet_rep(m, 0, bi) : P.bn_sums + 128 * bi, P.bn_sums + 128 * bi, P.state + m->lay.run[bi],
                     m->C[zi], (double)B * m->Ln[zi], m->nrep_f};
  }
  m->nrep_b = unet_nrep();
  unet_zero_bwd_records(m, s);
  const int C = m->C[10], L = m->Ln[10], n = C * L;
  static const int gmax = [] { const int v = (int)ral_knob("LOSS_GRID", 256); return v < 1 ? 1 : v; }();
  const int g = (B + UOL_WAVES - 1) / UOL_WAVES;
  k_unet_out_loss<<<g < gmax ? g : gmax, 64 * UOL_WAVES, 0, s>>>(o, y, target, dy, snr, rmse, loss_sum, C, L, B, (double)B * L,
                                                               (float)(2.0 / ((double)B * n)), fin, fin_scale, fin3, u,
                                                               m->nrep_b > 1 ? unet_rep(m, 1, 9) : P.bn_sums + 128 * 9 + 64, m->nrep_b);
  m->gsums_dy = dy; m->gsums_nrep = m->nrep_b;
  m->nrep_f = 1; m->nrep_b = 1;
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net forward launch failed"); return -1; }
  return 0;
}

int unet_backward_start(UNetModel* m, const float* dy, int B, int64_t gwin, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.cfg.train || !P.grads || !P.bn_sums) { snprintf(err, cap, "backward needs train=1, grads and bn_sums bound"); return -1; }
  if (B != m->last_B) { snprintf(err, cap, "backward batch %d != forward batch %d", B, m->last_B); return -1; }
  // with the fold every gradient entry is WRITTEN by k_unet_fold; the atomic path accumulates into a zeroed buffer
  if (!m->fold) (void)hipMemsetAsync(P.grads, 0, (size_t)m->lay.nparam * sizeof(float), s);
  // dy == NULL: "the gradient unet_forward_loss wrote, untouched" - its sums are already in the output layer's backward record.
  // A dy the caller passes explicitly is summed again, whatever its address: its CONTENTS may have changed since (loss
  // scaling, gradient accumulation), and the stage kernels read the new values.
  bool have_sums = false;
  if (!dy) {
    if (!m->gsums_dy) { snprintf(err, cap, "U-Net backward with dy = NULL must follow ral_forward_loss_means"); return -1; }
    dy = m->gsums_dy;
    have_sums = m->gsums_nrep == m->nrep_b;
  }
  m->last_dy = dy;
  m->bwd_rows = 0;
  m->gsums_dy = nullptr;
  if (have_sums) return 0;
  unet_zero_bwd_records(m, s);
  const size_t total = (size_t)B * m->C[10] * m->Ln[10];
  Src o = make_src(m, 10, ACT_NONE, true, false, 0);
  const size_t grows = (total / m->Ln[10] + 3) / 4;      // one wave per (window, channel) row
  k_unet_gsums<<<(int)(grows < 512 ? grows : 512), 256, 0, s>>>(
      dy, o, m->C[10], m->Ln[10], (double)gwin * m->Ln[10], m->nrep_b > 1 ? unet_rep(m, 1, 9) : P.bn_sums + 128 * 9 + 64, m->nrep_b, total);
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net backward launch failed"); return -1; }
  return 0;
}

int unet_backward_stage(UNetModel* m, int B, int si, int64_t gwin, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (B != m->last_B) { snprintf(err, cap, "backward batch %d != forward batch %d", B, m->last_B); return -1; }
  if (si < 0 || si > 10) { snprintf(err, cap, "U-Net stage %d outside [0, 10]", si); return -1; }
  // every workgroup ends with ~2 000 global atomics (its share of dW, db and the BatchNorm-backward sums): the chip retires
  // ~75 of them per ns, so 1024 workgroups spend 30 us per launch on them alone.  Measured train step at batch 2048 with
  // 1024 / 512 / 384 workgroups: 1.23 / 1.08 / 1.12 ms (RAL_UNET_BWD_GRID)
  static const int gmax0 = (int)ral_knob("UNET_BWD_GRID", 512);
  static const int wp = (int)ral_knob("UNET_BWD_WP", 2);
  const int gmax = (m->fold && gmax0 > m->part_rows_max) ? m->part_rows_max : gmax0;
  const int grid = B < gmax ? B : gmax;
  if (m->last_dy == nullptr) { snprintf(err, cap, "U-Net backward stages must follow ral_unet_backward_start"); return -1; }
  m->bwd_rows = grid;                             // (the same for every stage of a backward pass: it depends on B only)
  // consumers run in reverse order; an encoder tensor's gradient is first WRITTEN by its decoder-side consumer
  // (skip / residual, accumulate = 0) and later ACCUMULATED by the next encoder / bottleneck stage (accumulate = 1)
  Stage st = make_stage(m, si, m->last_x, true, true, B, (double)gwin);
  if (si == 0) { st.a.G = nullptr; }
  const size_t lds = bwd_lds(st);
  static size_t cur = 0;
  if (lds > cur) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_unet_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); cur = lds; }
  if (!launch_unet_bwd_fast(st, si, P.cfg.leads, B, grid, wp < 1 ? 1 : wp, s)) k_unet_bwd<<<grid, 256, lds, s>>>(st, B);
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net backward launch failed"); return -1; }
  return 0;
}

int unet_backward_finish(UNetModel* m, int B, int64_t gwin, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  BnGradAll u;
  for (int bi = 0; bi < 10; ++bi)
    u.l[bi] = BnGrad{m->nrep_b > 1 ? unet_rep(m, 1, bi) : P.bn_sums + 128 * bi + 64, P.bn_sums + 128 * bi + 64,
                     P.grads + m->lay.bnw[bi], P.grads + m->lay.bnb[bi], m->lay.bnC[bi], m->nrep_b};
  if (m->fold) {
    if (m->bwd_rows <= 0) { snprintf(err, cap, "U-Net backward finish without its stages"); return -1; }
    const bool reps = m->nrep_b > 1;
    k_unet_fold<<<(m->ncols + 63) / 64 + 1, 64 * UNET_FOLD_WAVES, 0, s>>>(m->part, m->part_stride, m->bwd_rows, m->cols, m->ncols, P.grads, u,
                                                        (double)B / (double)gwin, reps ? unet_rep(m, 0, 0) : nullptr,
                                                        reps ? 2 * 10 * UNET_MAXREP * 64 : 0);
    if (reps) m->rep_all_clean = true;
  } else {
    k_unet_bn_grads<<<10, MAXC, 0, s>>>(u, (double)B / (double)gwin);
  }
  m->last_dy = nullptr;
  m->nrep_b = 1;
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net backward launch failed"); return -1; }
  return 0;
}

int unet_backward(UNetModel* m, const float* dy, float* dx, int B, hipStream_t s, char* err, size_t cap) {
  if (dx) { snprintf(err, cap, "U-Net input gradient is not provided"); return -1; }
  m->nrep_b = unet_nrep();
  if (unet_backward_start(m, dy, B, B, s, err, cap)) { m->nrep_b = 1; return -1; }
  for (int si = 10; si >= 0; --si)
    if (unet_backward_stage(m, B, si, B, s, err, cap)) { m->nrep_b = 1; return -1; }
  const int rc = unet_backward_finish(m, B, B, s, err, cap);
  m->nrep_b = 1;
  return rc;
}
