// HBM-bound kernels around the transformer stack: conv stem (+LeakyReLU +BatchNorm),
// output conv, loss/SNR/RMSE reduction, fused flat Adam.  (gfx950)
//
// Reference behaviour: model/raletransformer.py:568-572, 636, 676-678;
// local_utils/evaluate.py:10-51; denoise_train.py:24,53 (Adam lr 1e-3, mse mean).
#include "ral_device.hpp"
#include "ral_kernels.hpp"

// block-wide sum of NV per-thread values -> double atomics into out[0..NV)
template <int NV>
RAL_DEV void block_atomic_sums(const float (&v)[NV], double* __restrict__ out, double* red /* LDS NV*nwaves */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float s = group_sum<64>(v[i]);
    if (lane == 0) red[wave * NV + i] = (double)s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w * NV + threadIdx.x];
    atomicAdd(out + threadIdx.x, t);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------
// conv1: Conv1d(leads, 8, k3, p1) + LeakyReLU(0.2) -> a0 (B, L, 8) token-major.
// MODE 0: training, also accumulates per-channel sum / sum of squares (double).
// MODE 1: eval, applies BatchNorm with the running statistics and writes x0 directly.
// ---------------------------------------------------------------------------------
template <int LEADS, int MODE>
__global__ __launch_bounds__(256) void k_conv1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ out,
                                                   double* __restrict__ stats, const float* __restrict__ bnw,
                                                   const float* __restrict__ bnb, const float* __restrict__ rmean,
                                                   const float* __restrict__ rvar, int L, int Lp, int B) {
  // L: samples of a window (row stride of x); Lp >= L: token slots per window in `out` (a window length that is not a multiple
  // of 256 runs on the next multiple, ral_api.hip: the slots past L are written as zeros and are not part of the statistics)
  __shared__ double red[16 * 4];
  float wr[8][LEADS][3], br[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    br[o] = bias[o];
#pragma unroll
    for (int c = 0; c < LEADS; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) wr[o][c][k] = w[(o * LEADS + c) * 3 + k];
  }
  float sc[8], sh[8];
  if (MODE == 1) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      sc[o] = bnw[o] / sqrtf(rvar[o] + 1e-5f);
      sh[o] = bnb[o] - rmean[o] * sc[o];
    }
  }
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const size_t total = (size_t)B * Lp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Lp), l = (int)(i - (size_t)b * Lp);
    if (l >= L) {
      float4* pz = reinterpret_cast<float4*>(out + i * 8);
      pz[0] = make_float4(0.f, 0.f, 0.f, 0.f); pz[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      continue;
    }
    float xv[LEADS][3];
#pragma unroll
    for (int c = 0; c < LEADS; ++c) {
      const float* xr = x + ((size_t)b * LEADS + c) * L;
      xv[c][0] = l > 0 ? xr[l - 1] : 0.f;
      xv[c][1] = xr[l];
      xv[c][2] = l < L - 1 ? xr[l + 1] : 0.f;
    }
    float y[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float a = br[o];
#pragma unroll
      for (int c = 0; c < LEADS; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) a = fmaf(wr[o][c][k], xv[c][k], a);
      a = a > 0.f ? a : 0.2f * a;
      if (MODE == 0) { acc[o] += a; acc[8 + o] += a * a; }
      else a = a * sc[o] + sh[o];
      y[o] = a;
    }
    float4* po = reinterpret_cast<float4*>(out + i * 8);
    po[0] = make_float4(y[0], y[1], y[2], y[3]);
    po[1] = make_float4(y[4], y[5], y[6], y[7]);
  }
  if (MODE == 0) block_atomic_sums<16>(acc, stats, red);
}

// stats: [sum(8), sumsq(8)] double; ss: [scale(8), shift(8), mean(8), rstd(8)] float
// The stem's BatchNorm in ONE launch (training): every workgroup forms the eight channels' scale / shift from the (all-reduced)
// sums itself - eight lanes of double arithmetic - and applies them to its share of the tokens; workgroup 0 also leaves ss
// (scale, shift, mean, rstd: the backward reads them) and the running statistics, (a finalise kernel and an apply kernel were two
// launches on the one stream that runs at the start of a step.)
__global__ __launch_bounds__(256) void k_bn_train8(const double* __restrict__ stats, double count, const float* __restrict__ bnw,
                                                   const float* __restrict__ bnb, float* __restrict__ ss, float* __restrict__ rmean,
                                                   float* __restrict__ rvar, const float* __restrict__ a0, float* __restrict__ x0,
                                                   size_t ntok) {
  __shared__ float sc_[8], sh_[8];
  if (threadIdx.x < 8) {
    const int c = threadIdx.x;
    const double mean = stats[c] / count;
    double var = stats[8 + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    const float sc = bnw[c] * rstd, sh = bnb[c] - (float)mean * sc;
    sc_[c] = sc; sh_[c] = sh;
    if (blockIdx.x == 0) {
      ss[c] = sc; ss[8 + c] = sh; ss[16 + c] = (float)mean; ss[24 + c] = rstd;
      rmean[c] = 0.9f * rmean[c] + 0.1f * (float)mean;
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      rvar[c] = 0.9f * rvar[c] + 0.1f * (float)unb;
    }
  }
  __syncthreads();
  const float4 s0 = make_float4(sc_[0], sc_[1], sc_[2], sc_[3]), s1 = make_float4(sc_[4], sc_[5], sc_[6], sc_[7]);
  const float4 h0 = make_float4(sh_[0], sh_[1], sh_[2], sh_[3]), h1 = make_float4(sh_[4], sh_[5], sh_[6], sh_[7]);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntok * 2; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(a0)[i];
    const bool hi = i & 1;
    reinterpret_cast<float4*>(x0)[i] = f4add(f4mul(v, hi ? s1 : s0), hi ? h1 : h0);
  }
}

// ---------------------------------------------------------------------------------
// output stage: z = u0 + x0 (token-major, 8 ch); y = Conv1d(8, leads, k3, p1)(z^T)
// ---------------------------------------------------------------------------------
template <int LEADS>
__global__ __launch_bounds__(256) void k_final_fwd(const float* __restrict__ u0, const float* __restrict__ x0,
                                                   const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, int L, int Lp, int B) {
  // L: samples per window of y; Lp >= L: token slots per window of u0 / x0 (the slots past L do not exist for the conv: zero halo)
  float wr[LEADS][8][3];
#pragma unroll
  for (int o = 0; o < LEADS; ++o)
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) wr[o][c][k] = w[(o * 8 + c) * 3 + k];
  const size_t total = (size_t)B * L;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / L), l = (int)(i - (size_t)b * L);
    float acc[LEADS];
#pragma unroll
    for (int o = 0; o < LEADS; ++o) acc[o] = bias[o];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ll = l + k - 1;
      if (ll < 0 || ll >= L) continue;
      const size_t t = (size_t)b * Lp + ll;
      const float4 a0 = reinterpret_cast<const float4*>(u0)[t * 2], a1 = reinterpret_cast<const float4*>(u0)[t * 2 + 1];
      const float4 b0 = reinterpret_cast<const float4*>(x0)[t * 2], b1 = reinterpret_cast<const float4*>(x0)[t * 2 + 1];
      const float z[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w,
                          a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
#pragma unroll
      for (int o = 0; o < LEADS; ++o)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[o] = fmaf(wr[o][c][k], z[c], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < LEADS; ++o) y[((size_t)b * LEADS + o) * L + l] = acc[o];
  }
}

// (loss_commit: ral_device.hpp)
__global__ __launch_bounds__(256) void k_loss(const float* __restrict__ pred, const float* __restrict__ target,
                                              float* __restrict__ dy, float* __restrict__ snr,
                                              float* __restrict__ rmse, double* __restrict__ loss_sum, int n,
                                              float gscale, double* __restrict__ fin, double fin_scale, int fin3) {
  __shared__ double red[2 * 4];
  const size_t base = (size_t)blockIdx.x * n;
  float v[2] = {0.f, 0.f};
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float p = pred[base + i], t = target[base + i], d = p - t;
    v[0] += d * d;
    v[1] += t * t;
    if (dy) dy[base + i] = d * gscale;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float s0 = group_sum<64>(v[0]), s1 = group_sum<64>(v[1]);
  if (lane == 0) { red[wave * 2] = s0; red[wave * 2 + 1] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double sse = 0, sy2 = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { sse += red[w * 2]; sy2 += red[w * 2 + 1]; }
    const float mse = (float)(sse / n), my2 = (float)(sy2 / n);
    const float sn = 10.0f * log10f(my2 / mse), rm = sqrtf(mse);
    if (snr) snr[blockIdx.x] = sn;
    if (rmse) rmse[blockIdx.x] = rm;
    if (loss_sum) loss_commit(loss_sum, sse / n, fin, fin_scale, fin3, (double)sn, (double)rm);
  }
}

// The same reduction with one WAVE per window (n a multiple of 4): 16-byte loads, four of each operand in flight per lane,
// persistent workgroups (at most 512) that add their share of the loss with ONE atomic each - a workgroup per window
// was 2048 same-address double atomics in a row (~12 ns per link: most of the kernel's 30 us at batch 2048) on top of one
// memory round trip per 4-byte element.
#define LOSS_W_WAVES 8
__global__ __launch_bounds__(64 * LOSS_W_WAVES) void k_loss_w(const float* __restrict__ pred, const float* __restrict__ target,
                                                float* __restrict__ dy, float* __restrict__ snr,
                                                float* __restrict__ rmse, double* __restrict__ loss_sum, int n, int B,
                                                float gscale, double* __restrict__ fin, double fin_scale, int fin3) {
  __shared__ double red[3][LOSS_W_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n4 = n >> 2;
  double mine = 0.0, msnr = 0.0, mrmse = 0.0;   // (lane 0: sums over this wave's windows of sse / n, SNR, RMSE)
  // a wave takes its windows two at a time: both windows' loads (up to 4 x 16 bytes per lane and operand, indices past the
  // end clamped) are requested before either is reduced - one memory round trip per pair
  const int stride = gridDim.x * LOSS_W_WAVES;
  for (int w = blockIdx.x * LOSS_W_WAVES + wave; w < B; w += 2 * stride) {
    const bool two = w + stride < B;
    const int wb = two ? w + stride : w;
    const float4* pa = reinterpret_cast<const float4*>(pred + (size_t)w * n);
    const float4* ta = reinterpret_cast<const float4*>(target + (size_t)w * n);
    const float4* pb = reinterpret_cast<const float4*>(pred + (size_t)wb * n);
    const float4* tb = reinterpret_cast<const float4*>(target + (size_t)wb * n);
    float4* da = dy ? reinterpret_cast<float4*>(dy + (size_t)w * n) : nullptr;
    float4* db = dy ? reinterpret_cast<float4*>(dy + (size_t)wb * n) : nullptr;
    float va0 = 0.f, va1 = 0.f, vb0 = 0.f, vb1 = 0.f;
    for (int i0 = 0; i0 < n4; i0 += 4 * 64) {
      float4 pav[4], tav[4], pbv[4], tbv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane, j = i < n4 ? i : 0;
        pav[k] = pa[j]; tav[k] = ta[j]; pbv[k] = pb[j]; tbv[k] = tb[j];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * 64 + lane;
        if (i < n4) {
          const float4 d = f4sub(pav[k], tav[k]);
          va0 += f4dot(d, d); va1 += f4dot(tav[k], tav[k]);
          if (da) da[i] = f4scale(d, gscale);
          if (two) {
            const float4 e = f4sub(pbv[k], tbv[k]);
            vb0 += f4dot(e, e); vb1 += f4dot(tbv[k], tbv[k]);
            if (db) db[i] = f4scale(e, gscale);
          }
        }
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
  if (threadIdx.x == 0 && loss_sum) {
    double t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = ((red[k][0] + red[k][1]) + (red[k][2] + red[k][3])) + ((red[k][4] + red[k][5]) + (red[k][6] + red[k][7]));
    loss_commit(loss_sum, t[0], fin, fin_scale, fin3, t[1], t[2]);
  }
}

// ---------------------------------------------------------------------------------
// fused flat Adam (torch.optim.Adam defaults, no weight decay / amsgrad)
// ---------------------------------------------------------------------------------
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, size_t n4, float step, float b1, float b2, float omb1, float omb2,
                       float eps, float sqrt_bc2, float gscale, double* __restrict__ zero64) {
  // (zero64: the 64 BatchNorm sums of the model, cleared here for the next step's stem: the last kernel of a step instead of a
  // fill kernel in front of the first one)
  if (zero64 && blockIdx.x == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = f4scale(reinterpret_cast<const float4*>(g)[i], gscale);
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
#define UPD(f)                                              \
    mm.f = b1 * mm.f + omb1 * gg.f;                         \
    vv.f = b2 * vv.f + omb2 * gg.f * gg.f;                  \
    pp.f -= step * mm.f / (sqrtf(vv.f) / sqrt_bc2 + eps);
    UPD(x) UPD(y) UPD(z) UPD(w)
#undef UPD
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
}

// ---------------------------------------------------------------------------------
static inline int ew_grid(size_t n, int per = 256) {
  size_t g = (n + per - 1) / per;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

void launch_conv1_fwd(int leads, int mode, const float* x, const float* w, const float* b, float* out, double* stats,
                      const float* bnw, const float* bnb, const float* rmean, const float* rvar, int L, int Lp, int B,
                      hipStream_t s) {
  // training: every workgroup ends with 16 double atomics on the same 16 addresses (~15 ns per link of a same-address
  // chain): 4096 workgroups were a 56 us kernel for 38 MB of traffic, 512 are an 18 us one
  const int grid = ew_grid((size_t)B * Lp, mode == 0 ? 2048 : 256);
#define CASE(ld)                                                                                             \
  case ld:                                                                                                   \
    if (mode == 0) k_conv1_fwd<ld, 0><<<grid, 256, 0, s>>>(x, w, b, out, stats, bnw, bnb, rmean, rvar, L, Lp, B); \
    else k_conv1_fwd<ld, 1><<<grid, 256, 0, s>>>(x, w, b, out, stats, bnw, bnb, rmean, rvar, L, Lp, B);           \
    break;
  switch (leads) { CASE(1) CASE(2) }
#undef CASE
}

void launch_bn_train8(const double* stats, double count, const float* bnw, const float* bnb, float* ss, float* rmean, float* rvar,
                      const float* a0, float* x0, size_t ntok, hipStream_t s) {
  const int g = ew_grid(ntok * 2);
  k_bn_train8<<<g < 2048 ? g : 2048, 256, 0, s>>>(stats, count, bnw, bnb, ss, rmean, rvar, a0, x0, ntok);
}

void launch_final_fwd(int leads, const float* u0, const float* x0, const float* w, const float* b, float* y, int L, int Lp,
                      int B, hipStream_t s) {
  const int grid = ew_grid((size_t)B * L);
  if (leads == 1) k_final_fwd<1><<<grid, 256, 0, s>>>(u0, x0, w, b, y, L, Lp, B);
  else k_final_fwd<2><<<grid, 256, 0, s>>>(u0, x0, w, b, y, L, Lp, B);
}

void launch_loss(const float* pred, const float* target, float* dy, float* snr, float* rmse, double* loss_sum,
                 int n, int B, float gscale, hipStream_t s, double* fin, double fin_scale, int fin3) {
  if (n % 4 == 0) {
    // (a workgroup ends with atomics on the same words - its share of the sum(s) and its arrival: at batch 2048 x 1024 floats
    // 512 four-wave workgroups were 19 us, 256 eight-wave ones 12.8 with two words in one cache line and 18.8 with four;
    // 128 / 64 workgroups: 13.8 / 13.1 - fewer links but half the CUs pulling the 24 MB.  256 and one line per word)
    static const int gmax = [] { const int v = (int)ral_knob("LOSS_GRID", 256); return v < 1 ? 1 : v; }();
    const int g = (B + LOSS_W_WAVES - 1) / LOSS_W_WAVES;
    k_loss_w<<<g < gmax ? g : gmax, 64 * LOSS_W_WAVES, 0, s>>>(pred, target, dy, snr, rmse, loss_sum, n, B, gscale, fin, fin_scale, fin3);
  } else {
    k_loss<<<B, 256, 0, s>>>(pred, target, dy, snr, rmse, loss_sum, n, gscale, fin, fin_scale, fin3);
  }
}

// the three buffers a backward pass starts from, zeroed by ONE launch (they were three fill kernels of ~5 us, each with its launch
// gap, between the loss and the first backward kernel: nothing else runs there)
__global__ void k_zero_bwd(float4* __restrict__ grads, size_t n4, double* __restrict__ sums, int nsums, unsigned* __restrict__ gmax, int ngmax) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n4; i += (size_t)gridDim.x * blockDim.x) grads[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i0 < (size_t)nsums) sums[i0] = 0.0;
  if (i0 < (size_t)ngmax) gmax[i0] = 0u;
}
void launch_zero_bwd(float* grads, size_t nfloat, double* sums, int nsums, unsigned* gmax, int ngmax, hipStream_t s) {
  k_zero_bwd<<<512, 256, 0, s>>>(reinterpret_cast<float4*>(grads), nfloat / 4, sums, nsums, gmax, ngmax);
}

void launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                 int step, float gscale, hipStream_t s, double* zero64) {
  // scalars are formed in double on the host and rounded once, as torch.optim.Adam does
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2 = 1.0 - pow(b2, (double)step);
  k_adam<<<ew_grid(n / 4), 256, 0, s>>>(p, g, m, v, n / 4, (float)(lr / bc1), (float)b1, (float)b2, (float)(1.0 - b1),
                                        (float)(1.0 - b2), (float)eps, (float)sqrt(bc2), gscale, zero64);
}

// ---------------------------------------------------------------------------------
// 12-lead adapter convolutions (reference model/ralenet_12leads.py:683-709):
// Conv1d(cin, cout, k13, p6) [+ LeakyReLU(0.01)], channel-major (B, C, L).  One workgroup per window:
// input rows staged in LDS with a zero halo, weights in LDS.
// ---------------------------------------------------------------------------------
#define K13 13
__global__ __launch_bounds__(256) void k_conv13_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int B,
                                                    int cin, int cout, int L, int lrelu) {
  extern __shared__ float4 smem4[];
  float* xs = reinterpret_cast<float*>(smem4);  // cin x (L + 12)
  float* ws = xs + cin * (L + 12);              // cout x cin x 13
  const int LP = L + 12;
  for (int i = threadIdx.x; i < cout * cin * K13; i += blockDim.x) ws[i] = w[i];
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < cin * LP; i += blockDim.x) {
      const int c = i / LP, p = i - c * LP - 6;
      xs[i] = (p >= 0 && p < L) ? x[((size_t)win * cin + c) * L + p] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cout * L; i += blockDim.x) {
      const int co = i / L, l = i - co * L;
      float acc = bias[co];
      for (int ci = 0; ci < cin; ++ci) {
        const float* wr = ws + (co * cin + ci) * K13;
        const float* xr = xs + ci * LP + l;
#pragma unroll
        for (int k = 0; k < K13; ++k) acc = fmaf(wr[k], xr[k], acc);
      }
      if (lrelu) acc = acc > 0.f ? acc : 0.01f * acc;
      y[((size_t)win * cout + co) * L + l] = acc;
    }
  }
}

// gradients: dz = dy * lrelu'(y);  dx = conv^T(dz);  gw += dz (*) x;  gb += sum dz
__global__ __launch_bounds__(256) void k_conv13_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ dy, const float* __restrict__ w,
                                                    float* __restrict__ gw, float* __restrict__ gb,
                                                    float* __restrict__ dx, int B, int cin, int cout, int L, int lrelu) {
  extern __shared__ float4 smem4[];
  const int LP = L + 12, nw = cout * cin * K13;
  float* xs = reinterpret_cast<float*>(smem4);  // cin x LP
  float* ds = xs + cin * LP;                    // cout x LP
  float* ws = ds + cout * LP;
  for (int i = threadIdx.x; i < nw; i += blockDim.x) ws[i] = w[i];
  float gacc[4] = {0.f, 0.f, 0.f, 0.f};          // 936 weight entries max / 256 threads
  float gbacc = 0.f;
  for (int win = blockIdx.x; win < B; win += gridDim.x) {
    __syncthreads();
    for (int i = threadIdx.x; i < cin * LP; i += blockDim.x) {
      const int c = i / LP, p = i - c * LP - 6;
      xs[i] = (p >= 0 && p < L) ? x[((size_t)win * cin + c) * L + p] : 0.f;
    }
    for (int i = threadIdx.x; i < cout * LP; i += blockDim.x) {
      const int c = i / LP, p = i - c * LP - 6;
      float g = 0.f;
      if (p >= 0 && p < L) {
        const size_t o = ((size_t)win * cout + c) * L + p;
        g = dy[o];
        if (lrelu && y[o] <= 0.f) g *= 0.01f;
      }
      ds[i] = g;
    }
    __syncthreads();
    if (dx) {
      for (int i = threadIdx.x; i < cin * L; i += blockDim.x) {
        const int ci = i / L, l = i - ci * L;
        float acc = 0.f;
        for (int co = 0; co < cout; ++co) {
          const float* wr = ws + (co * cin + ci) * K13;
          const float* dr = ds + co * LP + l + 12;   // dz[co][l + 6 - k] at padded index l + 12 - k
#pragma unroll
          for (int k = 0; k < K13; ++k) acc = fmaf(wr[k], dr[-k], acc);
        }
        dx[((size_t)win * cin + ci) * L + l] = acc;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = threadIdx.x + 256 * j;
      if (e < nw) {
        const int co = e / (cin * K13), ci = (e / K13) % cin, k = e % K13;
        const float* dr = ds + co * LP + 6;       // dz[co][l]
        const float* xr = xs + ci * LP + k;       // x[ci][l - 6 + k] at padded index l + k
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(dr[l], xr[l], s);
        gacc[j] += s;
      }
    }
    if ((int)threadIdx.x < cout) {
      const float* dr = ds + threadIdx.x * LP + 6;
      float s = 0.f;
      for (int l = 0; l < L; ++l) s += dr[l];
      gbacc += s;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e = threadIdx.x + 256 * j;
    if (e < nw) atomicAdd(gw + e, gacc[j]);
  }
  if ((int)threadIdx.x < cout) atomicAdd(gb + threadIdx.x, gbacc);
}

int launch_conv13_fwd(const float* x, const float* w, const float* b, float* y, int B, int cin, int cout, int L,
                      int lrelu, hipStream_t s) {
  if (cin * cout * K13 > 1024 || cin > 12 || cout > 12) return -1;
  const size_t lds = ((size_t)cin * (L + 12) + (size_t)cout * cin * K13 + 4) * sizeof(float);
  RAL_SET_LDS(k_conv13_fwd, lds);
  k_conv13_fwd<<<B < 2048 ? B : 2048, 256, lds, s>>>(x, w, b, y, B, cin, cout, L, lrelu);
  return 0;
}

int launch_conv13_bwd(const float* x, const float* y, const float* dy, const float* w, float* gw, float* gb,
                      float* dx, int B, int cin, int cout, int L, int lrelu, hipStream_t s) {
  if (cin * cout * K13 > 1024 || cin > 12 || cout > 12) return -1;
  const size_t lds = ((size_t)(cin + cout) * (L + 12) + (size_t)cout * cin * K13 + 4) * sizeof(float);
  RAL_SET_LDS(k_conv13_bwd, lds);
  k_conv13_bwd<<<B < 512 ? B : 512, 256, lds, s>>>(x, y, dy, w, gw, gb, dx, B, cin, cout, L, lrelu);
  return 0;
}

// ---------------------------------------------------------------------------------
// transposed copies of the weight matrices for the backward GEMMs (dX = dY W): with W^T row-major the
// A-operand fragment of those products is one 16-byte load, like in the forward pass
// ---------------------------------------------------------------------------------
__global__ void k_transpose_mats(const float* __restrict__ src, float* __restrict__ dst, const int4* __restrict__ desc,
                                 int nmat) {
  // desc[i] = (offset, rows, cols, first flat element index of this matrix in the launch)
  const int total = desc[nmat - 1].w + desc[nmat - 1].y * desc[nmat - 1].z;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    int lo = 0, hi = nmat - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].w <= e) lo = mid; else hi = mid - 1; }
    const int4 d = desc[lo];
    const int i = e - d.w, r = i / d.z, c = i - r * d.z;
    dst[d.x + c * d.y + r] = src[d.x + i];
  }
}

void launch_transpose_mats(const float* src, float* dst, const void* desc, int nmat, int total, hipStream_t s) {
  int grid = (total + 255) / 256;
  if (grid > 2048) grid = 2048;
  k_transpose_mats<<<grid, 256, 0, s>>>(src, dst, reinterpret_cast<const int4*>(desc), nmat);
}

// =================================================================================
// Record windowing + noise mixing in front of the model: one iteration of the reference's
// `batch_norm_snr_iter` (local_utils/local_utils.py:116-130) on the GPU --
//   clean = np_norm(segment, dim=0)                (per-lead z-score over the whole segment, population std, :261-266)
//   noisy = clean + sqrt(P_clean / 10^(snr/10) / P_noise) * noise      (`Gnoisegen`, :86-114)
//   '(b l) c -> b c l' windows of L samples, fp32
// Statistics are double sums (like the reference's float64 numpy arithmetic); the element-wise pass works in double and
// rounds once, so outputs equal the reference's `torch.FloatTensor(...)` casts up to the summation order of the sums.
// Both kernels are one HBM pass: (T, leads) row-major in, channel-major windows out.
// =================================================================================
#define RAL_PREP_MAXL 16
__global__ __launch_bounds__(256) void k_prep_stats(const float* __restrict__ sig, const float* __restrict__ noise,
                                                    long long T, int leads, double* __restrict__ sums) {
  __shared__ double red[2 * RAL_PREP_MAXL + 1];
  for (int i = threadIdx.x; i < 2 * leads + 1; i += blockDim.x) red[i] = 0.0;
  __syncthreads();
  double sx[RAL_PREP_MAXL], sxx[RAL_PREP_MAXL], snn = 0.0;
#pragma unroll
  for (int c = 0; c < RAL_PREP_MAXL; ++c) { sx[c] = 0.0; sxx[c] = 0.0; }
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < T; t += (long long)gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < RAL_PREP_MAXL; ++c) {
      if (c < leads) {
        const double x = sig[t * leads + c], n = noise[t * leads + c];
        sx[c] += x; sxx[c] += x * x; snn += n * n;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < RAL_PREP_MAXL; ++c) {
    if (c < leads) { atomicAdd(&red[c], sx[c]); atomicAdd(&red[leads + c], sxx[c]); }
  }
  atomicAdd(&red[2 * leads], snn);
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * leads + 1; i += blockDim.x) atomicAdd(sums + i, red[i]);
}

__global__ __launch_bounds__(256) void k_prep_mix(const float* __restrict__ sig, const float* __restrict__ noise,
                                                  const double* __restrict__ sums, long long T, int leads, int L,
                                                  double snr_db, float* __restrict__ noisy, float* __restrict__ clean) {
  // P_clean = sum(clean^2) / T = leads (every lead has unit population variance); P_noise = sum(noise^2) / T
  const double scale = sqrt((double)leads / pow(10.0, snr_db / 10.0) / (sums[2 * leads] / (double)T));
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < T; t += (long long)gridDim.x * blockDim.x) {
    const long long b = t / L;
    const int l = (int)(t - b * L);
    for (int c = 0; c < leads; ++c) {
      const double mean = sums[c] / (double)T;
      const double var = sums[leads + c] / (double)T - mean * mean;
      const double xn = ((double)sig[t * leads + c] - mean) / sqrt(var);
      const size_t o = ((size_t)b * leads + c) * L + l;
      clean[o] = (float)xn;
      noisy[o] = (float)(xn + scale * (double)noise[t * leads + c]);
    }
  }
}

int launch_prep_windows(const float* sig, const float* noise, long long T, int leads, int L, double snr_db, double* sums,
                        float* noisy, float* clean, hipStream_t s) {
  if (leads < 1 || leads > RAL_PREP_MAXL || L < 1 || T < L || T % L != 0) return -1;
  (void)hipMemsetAsync(sums, 0, (2 * leads + 1) * sizeof(double), s);
  const int grid = (int)((T + 255) / 256 < 2048 ? (T + 255) / 256 : 2048);
  k_prep_stats<<<grid, 256, 0, s>>>(sig, noise, T, leads, sums);
  k_prep_mix<<<grid, 256, 0, s>>>(sig, noise, sums, T, leads, L, snr_db, noisy, clean);
  return 0;
}

// =================================================================================
// Streaming of long records around the inference forward (BASELINE config 5).  The reference cuts the
// 650 000-sample MIT-BIH records into fixed chunks and z-scores them (local_utils/local_utils.py:116-130, np_norm
// :261-266); here a group of R records (R, leads, T) is cut into windows of L samples every `hop` samples (the last
// window of a record is right-aligned), every window is z-scored per lead, and after the model the windows are
// de-normalised and stitched back: overlapping regions keep the centre of each window, the record edges keep the
// whole window.  One wave per (window, lead); mean and standard deviation go to a side buffer for the way back.
// =================================================================================
RAL_DEV void stream_geom(long long T, int L, int hop, int& n_reg, int& n) {
  n_reg = (int)((T - L) / hop) + 1;
  n = n_reg + (((T - L) % hop) != 0 ? 1 : 0);
}

__global__ __launch_bounds__(64) void k_stream_windows(const float* __restrict__ rec, long long T, int leads, int L, int hop,
                                                       long long w0, float* __restrict__ win, float* __restrict__ stats) {
  int n_reg, n;
  stream_geom(T, L, hop, n_reg, n);
  const int i = blockIdx.x / leads, c = blockIdx.x - i * leads;   // window of this launch, lead
  const long long gw = w0 + i;
  const long long r = gw / n;
  const int k = (int)(gw - r * n);
  const long long start = k < n_reg ? (long long)k * hop : T - L;
  const float* src = rec + (r * leads + c) * T + start;
  const int lane = threadIdx.x;
  float sum = 0.f;
  for (int l = lane; l < L; l += 64) sum += src[l];
  const float mean = group_sum<64>(sum) / (float)L;
  float ss = 0.f;
  for (int l = lane; l < L; l += 64) { const float d = src[l] - mean; ss = fmaf(d, d, ss); }   // (second pass: L1 hits)
  const float sd = fmaxf(sqrtf(group_sum<64>(ss) / (float)L), 1e-6f);   // population std, as np_norm; constant leads stay finite
  const float inv = 1.0f / sd;
  float* dst = win + ((size_t)i * leads + c) * L;
  for (int l = lane; l < L; l += 64) dst[l] = (src[l] - mean) * inv;
  if (lane == 0) { stats[(gw * leads + c) * 2] = mean; stats[(gw * leads + c) * 2 + 1] = sd; }
}

__global__ __launch_bounds__(256) void k_stream_stitch(const float* __restrict__ y, const float* __restrict__ stats, long long R,
                                                       long long T, int leads, int L, int hop, float* __restrict__ out) {
  int n_reg, n;
  stream_geom(T, L, hop, n_reg, n);
  const int h = (L - hop) >> 1;
  const long long last_begin = n > 1 ? (long long)(n - 2) * hop + L - h : 0;   // first sample the last window keeps
  const long long total = R * leads * T;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long rc = e / T, t = e - rc * T;
    const long long r = rc / leads;
    const int c = (int)(rc - r * leads);
    int k; long long off;
    if (n == 1 || t >= last_begin) { k = n - 1; off = t - (k < n_reg ? (long long)k * hop : T - L); }
    else { k = t < h ? 0 : (int)((t - h) / hop); off = t - (long long)k * hop; }
    const long long gw = r * n + k;
    const float mean = stats[(gw * leads + c) * 2], sd = stats[(gw * leads + c) * 2 + 1];
    out[e] = fmaf(y[(gw * leads + c) * L + off], sd, mean);
  }
}

int launch_stream_windows(const float* rec, long long R, long long T, int leads, int L, int hop, long long w0, int nw,
                          float* win, float* stats, hipStream_t s) {
  if (R < 1 || leads < 1 || L < 64 || L % 64 != 0 || L > 2048 || T < L || hop < 1 || hop > L || nw < 1 || w0 < 0) return -1;
  const long long n_reg = (T - L) / hop + 1, n = n_reg + (((T - L) % hop) != 0 ? 1 : 0);
  if (w0 + nw > R * n) return -1;
  k_stream_windows<<<nw * leads, 64, 0, s>>>(rec, T, leads, L, hop, w0, win, stats);
  return 0;
}

int launch_stream_stitch(const float* y, const float* stats, long long R, long long T, int leads, int L, int hop, float* out,
                         hipStream_t s) {
  if (R < 1 || leads < 1 || T < L || hop < 1 || hop > L || ((L - hop) & 1)) return -1;
  const long long total = R * leads * T;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  k_stream_stitch<<<grid, 256, 0, s>>>(y, stats, R, T, leads, L, hop, out);
  return 0;
}
