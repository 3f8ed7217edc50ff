// db8 wavelet-threshold baseline (local_utils/denoisefunc.py:7-33: pywt.wavedec 'db8' at the maximum level, soft
// threshold of every detail band at 0.04 * max(band), pywt.waverec), one workgroup per 1-D record.
//
// A record of L samples (2 KB at L = 512) and its whole coefficient pyramid (L + 15 * levels floats) stay in LDS: HBM
// sees one read and one write of the record.  Every level is a 16-tap stride-2 filter pair over the half-sample
// symmetric extension of the previous approximation band (PyWavelets MODE_SYMMETRIC), band length floor((n + 15) / 2);
// reconstruction is the valid part of the up-sampled synthesis pair (2 m - 14 samples from bands of m), with the surplus
// sample of an odd-length level dropped as pywt.waverec does.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ralenet.h"
#include "ral_device.hpp"

namespace {
constexpr int WF = 16;   // db8 filter length
// pywt.Wavelet('db8').dec_lo (the synthesis / high-pass filters are its mirror images)
__constant__ float c_dec_lo[WF] = {
    -0.00011747678400228192f, 0.0006754494059985568f, -0.0003917403729959771f, -0.00487035299301066f,
    0.008746094047015655f,    0.013981027917015516f,  -0.04408825393106472f,   -0.01736930100202211f,
    0.128747426620186f,       0.00047248457399797254f, -0.2840155429624281f,   -0.015829105256023893f,
    0.5853546836548691f,      0.6756307362980128f,    0.3128715909144659f,     0.05441584224308161f};

constexpr int WMAXLEV = 12;
struct WavePlan { int nlev; int n[WMAXLEV + 1]; int off[WMAXLEV + 1]; int total; };   // n[k]: band length after k levels

RAL_DEV int sym_index(int t, int n) {   // half-sample symmetric extension (one reflection suffices: n >= 15)
  t = t < 0 ? -t - 1 : t;
  return t >= n ? 2 * n - 1 - t : t;
}

__global__ __launch_bounds__(128) void k_wavelet_denoise(const float* __restrict__ x, float* __restrict__ y, int L,
                                                         WavePlan P, float thr) {
  extern __shared__ float sm[];
  float* A0 = sm;                 // approximation bands, ping-pong
  float* A1 = sm + L;
  float* D = sm + 2 * L;          // detail pyramid: level k at D + off[k], n[k] coefficients
  __shared__ float red[2];
  const size_t row = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  float flo[WF], fhi[WF];         // dec_lo[j], dec_hi[j] = (-1)^(j+1) dec_lo[15 - j]
#pragma unroll
  for (int j = 0; j < WF; ++j) { flo[j] = c_dec_lo[j]; fhi[j] = (j & 1) ? c_dec_lo[WF - 1 - j] : -c_dec_lo[WF - 1 - j]; }
  for (int i = tid; i < L; i += nt) A0[i] = x[row * L + i];
  __syncthreads();
  float* a = A0; float* b = A1;
  // ---- analysis ----
  for (int k = 1; k <= P.nlev; ++k) {
    const int n = P.n[k - 1], m = P.n[k];
    float* d = D + P.off[k];
    for (int o = tid; o < m; o += nt) {
      float sa = 0.f, sd = 0.f;
#pragma unroll
      for (int j = 0; j < WF; ++j) {
        const float v = a[sym_index(2 * o + 1 - j, n)];
        sa = fmaf(flo[j], v, sa); sd = fmaf(fhi[j], v, sd);
      }
      b[o] = sa; d[o] = sd;
    }
    __syncthreads();
    float* t = a; a = b; b = t;
  }
  // ---- soft threshold of every detail band at thr * max(band) ----
  for (int k = 1; k <= P.nlev; ++k) {
    float* d = D + P.off[k];
    const int m = P.n[k];
    float mx = -INFINITY;
    for (int o = tid; o < m; o += nt) mx = fmaxf(mx, d[o]);
    for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    const float value = thr * fmaxf(red[0], nt > 64 ? red[1] : red[0]);
    for (int o = tid; o < m; o += nt) {
      const float c = d[o], mag = fabsf(c);
      d[o] = mag > 0.f ? c * fmaxf(1.f - value / mag, 0.f) : 0.f;
    }
    __syncthreads();
  }
  // ---- synthesis: rec_lo[j] = dec_lo[15 - j], rec_hi[j] = dec_hi[15 - j] ----
  for (int k = P.nlev; k >= 1; --k) {
    const int m = P.n[k];                 // (a may hold one sample more than d: it is simply not read)
    const float* d = D + P.off[k];
    const int half = m - WF / 2 + 1;      // output pairs
    for (int q = tid; q < half; q += nt) {
      float e = 0.f, od = 0.f;
#pragma unroll
      for (int j = 0; j < WF / 2; ++j) {
        const float av = a[q + WF / 2 - 1 - j], dv = d[q + WF / 2 - 1 - j];
        e = fmaf(flo[WF - 1 - 2 * j], av, fmaf(fhi[WF - 1 - 2 * j], dv, e));
        od = fmaf(flo[WF - 2 - 2 * j], av, fmaf(fhi[WF - 2 - 2 * j], dv, od));
      }
      b[2 * q] = e; b[2 * q + 1] = od;
    }
    __syncthreads();
    float* t = a; a = b; b = t;
  }
  for (int i = tid; i < L; i += nt) y[row * L + i] = a[i];
}
}  // namespace

// (the C entry point, ral_wavelet_denoise, is in ral_api.hip with the other argument checks)
int launch_wavelet_denoise(const float* x, float* y, long long rows, int L, float threshold, hipStream_t stream) {
  if (rows < 0 || L < 2 || (L & 1) || L > 8192) return -1;
  if (rows == 0) return 0;
  WavePlan P;
  P.nlev = 0; P.n[0] = L; P.off[0] = 0; P.total = 0;
  if (L >= WF - 1) {
    int lev = 0;
    while (L >= (WF - 1) * (2 << lev)) ++lev;   // pywt.dwt_max_level: floor(log2(L / (F - 1)))
    P.nlev = lev < WMAXLEV ? lev : WMAXLEV;
  }
  for (int k = 1; k <= P.nlev; ++k) {
    P.n[k] = (P.n[k - 1] + WF - 1) / 2;
    P.off[k] = P.total;
    P.total += P.n[k];
  }
  const size_t lds = (size_t)(2 * L + P.total + 16) * sizeof(float);
  if (lds > 150 * 1024) return -1;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wavelet_denoise), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k_wavelet_denoise<<<(unsigned)rows, 128, lds, stream>>>(x, y, L, P, threshold);
  return 0;
}
