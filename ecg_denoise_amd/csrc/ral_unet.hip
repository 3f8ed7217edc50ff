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
#include "ral_unet.hpp"

#include <stdio.h>
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
};

RAL_DEV float lrelu01(float v) { return v > 0.f ? v : 0.01f * v; }

// scale/shift (and mean/rstd) of one operand's BatchNorm into LDS: ss[0:C] scale, [C:2C] shift, [2C:3C] mean, [3C:4C] rstd
RAL_DEV void src_coeffs(const Src& s, int C, double count, float* ss) {
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mean = 0.f, rstd = 1.f, sc = 1.f, sh = 0.f;
    if (s.norm == NORM_BATCH) {
      const double m = s.sums[c] / count;
      double var = s.sums[MAXC + c] / count - m * m;
      if (var < 0.0) var = 0.0;
      mean = (float)m; rstd = (float)(1.0 / sqrt(var + 1e-5));
    } else if (s.norm == NORM_RUNNING) {
      mean = s.running[c]; rstd = 1.0f / sqrtf(s.running[C + c] + 1e-5f);
    }
    if (s.norm != NORM_NONE) { sc = s.gamma[c] * rstd; sh = s.beta[c] - mean * sc; }
    ss[c] = sc; ss[C + c] = sh; ss[2 * C + c] = mean; ss[3 * C + c] = rstd;
  }
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
    atomicAdd(st.sums_out + threadIdx.x, (double)red[threadIdx.x]);
    atomicAdd(st.sums_out + MAXC + threadIdx.x, (double)red[MAXC + threadIdx.x]);
  }
}

// final: y = BN9(z9) elementwise;  also running-stat updates of all layers
__global__ void k_unet_out(Src s, float* __restrict__ y, int C, int L, double count, size_t total) {
  __shared__ float ss[4 * MAXC];
  src_coeffs(s, C, count, ss);
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    y[i] = s.z[i] * ss[c] + ss[C + c];
  }
}

struct BnUpd { const double* sums; float* running; int C; double count; };
struct BnUpdAll { BnUpd l[10]; };
__global__ void k_unet_running(BnUpdAll u) {
  const BnUpd& b = u.l[blockIdx.x];
  const int c = threadIdx.x;
  if (c >= b.C) return;
  const double m = b.sums[c] / b.count;
  double var = b.sums[MAXC + c] / b.count - m * m;
  if (var < 0.0) var = 0.0;
  b.running[c] = 0.9f * b.running[c] + 0.1f * (float)m;
  b.running[b.C + c] = 0.9f * b.running[b.C + c] + 0.1f * (float)(b.count > 1.0 ? var * b.count / (b.count - 1.0) : var);
}

// sums of the gradient at a BatchNorm output: S1 = sum G, S2 = sum G * zhat   (used for the last layer)
__global__ __launch_bounds__(256) void k_unet_gsums(const float* __restrict__ G, Src s, int C, int L, double count,
                                                    double* __restrict__ bsums, size_t total) {
  __shared__ float ss[4 * MAXC];
  __shared__ float red[2 * MAXC];
  src_coeffs(s, C, count, ss);
  for (int i = threadIdx.x; i < 2 * MAXC; i += blockDim.x) red[i] = 0.f;
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    const float g = G[i], zh = (s.z[i] - ss[2 * C + c]) * ss[3 * C + c];
    atomicAdd(red + c, g); atomicAdd(red + MAXC + c, g * zh);
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    atomicAdd(bsums + threadIdx.x, (double)red[threadIdx.x]);
    atomicAdd(bsums + MAXC + threadIdx.x, (double)red[MAXC + threadIdx.x]);
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
    Src o; o.norm = NORM_BATCH; o.sums = st.sums_out; o.gamma = st.gamma_out; o.beta = st.gamma_out; o.act = ACT_NONE;
    float* tmp = gbs + MAXC;  // 4*MAXC scratch
    src_coeffs(o, st.cout, st.count, tmp);
    __syncthreads();
    for (int c = threadIdx.x; c < st.cout; c += blockDim.x) {
      co_[c] = st.gamma_out[c] * tmp[3 * st.cout + c];
      co_[MAXC + c] = (float)(st.bsums_out[c] / st.count);
      co_[2 * MAXC + c] = (float)(st.bsums_out[MAXC + c] / st.count);
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
      atomicAdd(st.a.bsums + threadIdx.x, (double)sa[threadIdx.x]);
      atomicAdd(st.a.bsums + MAXC + threadIdx.x, (double)sa[MAXC + threadIdx.x]);
    }
    if (st.b.z && st.b.G) {
      atomicAdd(st.b.bsums + threadIdx.x, (double)sb[threadIdx.x]);
      atomicAdd(st.b.bsums + MAXC + threadIdx.x, (double)sb[MAXC + threadIdx.x]);
    }
  }
  if (st.r.z && st.r.G && (int)threadIdx.x < st.cout) {
    atomicAdd(st.r.bsums + threadIdx.x, (double)sr[threadIdx.x]);
    atomicAdd(st.r.bsums + MAXC + threadIdx.x, (double)sr[MAXC + threadIdx.x]);
  }
}

// BatchNorm affine gradients of all layers from the backward sums: g_gamma = S2, g_beta = S1
struct BnGrad { const double* bsums; float* gw; float* gb; int C; };
struct BnGradAll { BnGrad l[10]; };
__global__ void k_unet_bn_grads(BnGradAll u) {
  const BnGrad& b = u.l[blockIdx.x];
  const int c = threadIdx.x;
  if (c < b.C) { b.gb[c] = (float)b.bsums[c]; b.gw[c] = (float)b.bsums[MAXC + c]; }
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
  float* G[11];       // gradients at the BatchNorm outputs (same indexing; G[6] = d r)
  int C[11], Ln[11];  // channels / length of z[i]
  const float* last_x = nullptr;
  int last_B = 0;
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

static Src make_src(UNetModel* m, int zi, int act, bool training, bool with_grad, int accumulate) {
  Src s; memset(&s, 0, sizeof(s));
  const UNetPublic& P = m->pub;
  s.z = m->z[zi];
  const int bi = BN_OF_Z[zi];
  if (bi >= 0) {
    s.norm = training ? NORM_BATCH : NORM_RUNNING;
    s.sums = P.bn_sums ? P.bn_sums + 128 * bi : nullptr;
    s.gamma = P.params + m->lay.bnw[bi]; s.beta = P.params + m->lay.bnb[bi];
    s.running = P.state + m->lay.run[bi];
    if (with_grad) s.bsums = P.bn_sums + 128 * bi + 64;
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
static Stage make_stage(UNetModel* m, int si, const float* x, bool training, bool bwd, int B) {
  const ral_config& c = m->pub.cfg;
  const UNetPublic& P = m->pub;
  const ULayout& Y = m->lay;
  Stage s; memset(&s, 0, sizeof(s));
  const double cnt = (double)B;
  auto cnt_of = [&](int zi) { return cnt * m->Ln[zi]; };
  s.w = P.params + Y.w[si]; s.bias = P.params + Y.b[si];
  s.out = m->z[si]; s.cout = m->C[si]; s.lout = m->Ln[si];
  const int bo = BN_OF_Z[si];
  s.sums_out = (bo >= 0 && P.bn_sums) ? P.bn_sums + 128 * bo : nullptr;
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
    s.bsums_out = bo >= 0 ? P.bn_sums + 128 * bo + 64 : nullptr;
    s.gamma_out = bo >= 0 ? P.params + Y.bnw[bo] : nullptr;
    s.gw = P.grads + Y.w[si]; s.gb = P.grads + Y.b[si];
  }
  return s;
}

int unet_forward(UNetModel* m, const float* x, float* y, int B, int training, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.params || !P.state) { snprintf(err, cap, "ral_bind was not called"); return -1; }
  if (B <= 0 || B > P.cfg.max_batch) { snprintf(err, cap, "batch %d outside (0, %d]", B, P.cfg.max_batch); return -1; }
  if (training && (!P.cfg.train || !P.bn_sums)) { snprintf(err, cap, "training forward needs train=1 and bn_sums"); return -1; }
  m->last_x = x; m->last_B = B;
  if (training) (void)hipMemsetAsync(P.bn_sums, 0, 1280 * sizeof(double), s);
  const int grid = B < 1024 ? B : 1024;
  for (int si = 0; si < 11; ++si) {
    Stage st = make_stage(m, si, x, training != 0, false, B);
    if (!training) st.sums_out = nullptr;
    k_unet_fwd<<<grid, 256, fwd_lds(st), s>>>(st, B);
  }
  Src o = make_src(m, 10, ACT_NONE, training != 0, false, 0);
  const size_t total = (size_t)B * m->C[10] * m->Ln[10];
  k_unet_out<<<(int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048), 256, 0, s>>>(o, y, m->C[10], m->Ln[10],
                                                                                          (double)B * m->Ln[10], total);
  if (training) {
    BnUpdAll u;
    for (int zi = 0, k = 0; zi < 11; ++zi) {
      const int bi = BN_OF_Z[zi];
      if (bi < 0) continue;
      u.l[k++] = BnUpd{P.bn_sums + 128 * bi, P.state + m->lay.run[bi], m->C[zi], (double)B * m->Ln[zi]};
    }
    k_unet_running<<<10, MAXC, 0, s>>>(u);
  }
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net forward launch failed"); return -1; }
  return 0;
}

int unet_backward(UNetModel* m, const float* dy, float* dx, int B, hipStream_t s, char* err, size_t cap) {
  UNetPublic& P = m->pub;
  if (!P.cfg.train || !P.grads || !P.bn_sums) { snprintf(err, cap, "backward needs train=1, grads and bn_sums bound"); return -1; }
  if (B != m->last_B) { snprintf(err, cap, "backward batch %d != forward batch %d", B, m->last_B); return -1; }
  if (dx) { snprintf(err, cap, "U-Net input gradient is not provided"); return -1; }
  (void)hipMemsetAsync(P.grads, 0, (size_t)m->lay.nparam * sizeof(float), s);
  for (int bi = 0; bi < 10; ++bi) (void)hipMemsetAsync(P.bn_sums + 128 * bi + 64, 0, 64 * sizeof(double), s);
  // last layer: gradient at BN9's output is dy itself
  const size_t total = (size_t)B * m->C[10] * m->Ln[10];
  (void)hipMemcpyAsync(m->G[10], dy, total * sizeof(float), hipMemcpyDeviceToDevice, s);
  Src o = make_src(m, 10, ACT_NONE, true, false, 0);
  k_unet_gsums<<<(int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024), 256, 0, s>>>(
      dy, o, m->C[10], m->Ln[10], (double)B * m->Ln[10], P.bn_sums + 128 * 9 + 64, total);
  const int grid = B < 1024 ? B : 1024;
  // consumers run in reverse order; an encoder tensor's gradient is first WRITTEN by its decoder-side consumer
  // (skip / residual, accumulate = 0) and later ACCUMULATED by the next encoder / bottleneck stage (accumulate = 1)
  for (int si = 10; si >= 0; --si) {
    Stage st = make_stage(m, si, m->last_x, true, true, B);
    if (si == 0) { st.a.G = nullptr; }
    const size_t lds = bwd_lds(st);
    static size_t cur = 0;
    if (lds > cur) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_unet_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); cur = lds; }
    k_unet_bwd<<<grid, 256, lds, s>>>(st, B);
  }
  BnGradAll u;
  for (int bi = 0; bi < 10; ++bi)
    u.l[bi] = BnGrad{P.bn_sums + 128 * bi + 64, P.grads + m->lay.bnw[bi], P.grads + m->lay.bnb[bi], m->lay.bnC[bi]};
  k_unet_bn_grads<<<10, MAXC, 0, s>>>(u);
  if (hipGetLastError() != hipSuccess) { snprintf(err, cap, "U-Net backward launch failed"); return -1; }
  return 0;
}
