// Host-side launch interface of the RA-LENet / U-Net kernels (internal to libralenet).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "ral_device.hpp"

// gfx950 has 160 KB of LDS per CU; dynamic LDS above 64 KB must be opted into per kernel.
#define RAL_SET_LDS(kernel, bytes)                                                                     \
  do {                                                                                                 \
    static size_t ral_cur_ = 0;                                                                        \
    if ((size_t)(bytes) > ral_cur_) {                                                                  \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));             \
      ral_cur_ = (size_t)(bytes);                                                                      \
    }                                                                                                  \
  } while (0)

// ---- forward (ral_fwd.hip)
// Linear layers of the wide levels on the f16 matrix cores (two fp16 pieces per operand, three products per term): the
// weight matrices are re-written once per forward as tiled split planes (launch_tile_planes: desc = int4 {float offset, rows,
// columns, first work item} per matrix, nwork = sum of rows * columns / 8; a matrix's planes sit at twice its float offset
// in `wt`).  wt == nullptr selects the fp32-MFMA kernels.
bool qkv_fwd_uses_f16(int C);
void launch_tile_planes(const float* params, void* wt, const void* desc, int ndesc, int nwork, int unscaled_residual, hipStream_t s);
// the activation scales of the nblk transformer blocks (ASC_N floats each; desc: 12 ints per block - see k_act_scales)
void launch_act_scales(const float* params, const void* desc, float* asc, int nblk, hipStream_t s);
void launch_qkv_fwd(int C, const float* x, const float* pe, const BlockP& w, const void* wt /* of Wqkv */, float* qkv, int N, int B, hipStream_t s);
size_t attn_fwd_lds(int N, int HG, int Len);
// NE: existing tokens of the N slots (0 or N: all; fewer: padded windows, the generic kernel masks the keys past NE)
void launch_attn_fwd(const float* qkv, float* o_hm, float* lse, const float* table, int N, int H, int HG, int Len,
                     int B, int f16, hipStream_t s, int NE = 0);
size_t mlp_fwd_lds(int C, int N, int nch);
// pbase / wt: the parameter buffer and the tiled-plane buffer; the levels where mlp_fwd_uses_f16(C, N) run their Linear
// layers on the planes (wt == nullptr: fp32 MFMA everywhere)
bool mlp_fwd_uses_f16(int C, int N);
// f16_narrow: the model allows fp16-pair products (f16_split > 0): the narrow levels may take their f16 strip kernel
void launch_mlp_fwd(int C, int nch, const float* x, const float* o, const BlockP& w, const float* pbase, const void* wt,
                    float* x1, float* upre, float* x2, int N, int B, int f16_narrow, hipStream_t s, int NE = 0 /* existing tokens; see launch_attn_fwd */,
                    const float* addend = nullptr, float* sum_out = nullptr /* optional second output: block output + addend */);
// narrow levels (C <= 32), wave-autonomous (ral_mlpw.hip): kind 0 = not taken, 1 = fp32-MFMA strips, 2 = f16 strips
int mlp_fwd_w_kind(int C, int N, bool want_upre, bool f16_ok);
bool mlp_fwd_w_takes(int C, int N, bool want_upre);
void launch_mlp_fwd_w(int C, int kind, const float* x, const float* o, const BlockP& w, float* x1, float* x2, int N, int B, hipStream_t s);
// narrow levels, backward as a strip kernel (ral_mlpw.hip; fc1 / fc2 weight gradients fused like k_mlp_bwd_s)
// kind: 0 = not taken, 1 = fp32-MFMA strips, 2 = f16 strips (f16_ok: the model allows fp16-pair products)
int mlp_bwd_w_kind(int C, int N, bool f16_ok);
void launch_mlp_bwd_w(int C, int kind, const float* dx2, const float* x1, const BlockP& w, const BlockP& gr, float* dx1, float* do_hm, int N, int B,
                      bool want_dw, hipStream_t s);
void launch_resample_fwd(int D, bool sep, const float* x, const float* wred, const float* lnw, const float* lnb,
                         const float* skip, float* y, int T, int Tv /* existing output tokens (<= T slots) */, int B, hipStream_t s);
void launch_add(const float* a, const float* b, float* y, size_t n, hipStream_t s);

// ---- stem / head / loss / optimiser (ral_misc.hip)
void launch_conv1_fwd(int leads, int mode, const float* x, const float* w, const float* b, float* out, double* stats,
                      const float* bnw, const float* bnb, const float* rmean, const float* rvar, int L, int Lp /* token slots per window (>= L) */, int B,
                      hipStream_t s);
// the stem BatchNorm of a training forward: scale / shift / mean / rstd into ss, running statistics updated, x0 = a0 * scale + shift
void launch_bn_train8(const double* stats, double count, const float* bnw, const float* bnb, float* ss, float* rmean, float* rvar,
                      const float* a0, float* x0, size_t ntok, hipStream_t s);
void launch_final_fwd(int leads, const float* u0, const float* x0, const float* w, const float* b, float* y, int L, int Lp,
                      int B, hipStream_t s);
// fin != nullptr: loss_sum is a {double, counter} scratch that is zero on entry and left zero; fin[0] = sum * fin_scale
// fin3: the scratch has 64 doubles (sum [0], counter [16], SNR sum [32], RMSE sum [48]: a cache line each); fin[1], fin[2] = the sums of the windows' SNR / RMSE * fin_scale
void launch_loss(const float* pred, const float* target, float* dy, float* snr, float* rmse, double* loss_sum,
                 int n, int B, float gscale, hipStream_t s, double* fin = nullptr, double fin_scale = 1.0, int fin3 = 0);
// grads[0, nfloat) (nfloat % 4 == 0), sums[0, nsums) and gmax[0, ngmax) = 0 in one launch (nsums, ngmax <= 131072)
void launch_zero_bwd(float* grads, size_t nfloat, double* sums, int nsums, unsigned* gmax, int ngmax, hipStream_t s);
void launch_adam(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                 int step, float gscale, hipStream_t s, double* zero64 = nullptr /* optional: 64 doubles cleared by the same launch */);

int launch_prep_windows(const float* sig, const float* noise, long long T, int leads, int L, double snr_db, double* sums,
                        float* noisy, float* clean, hipStream_t s);
int launch_stream_windows(const float* rec, long long R, long long T, int leads, int L, int hop, long long w0, int nw,
                          float* win, float* stats, hipStream_t s);
int launch_stream_stitch(const float* y, const float* stats, long long R, long long T, int leads, int L, int hop, float* out,
                         hipStream_t s);
int launch_conv13_fwd(const float* x, const float* w, const float* b, float* y, int B, int cin, int cout, int L,
                      int lrelu, hipStream_t s);
int launch_conv13_bwd(const float* x, const float* y, const float* dy, const float* w, float* gw, float* gb,
                      float* dx, int B, int cin, int cout, int L, int lrelu, hipStream_t s);

void launch_transpose_mats(const float* src, float* dst, const void* desc, int nmat, int total, hipStream_t s);

// ---- backward (ral_bwd.hip)
size_t mlp_bwd_lds(int C, int N, int nch);
bool mlp_bwd_is_fused(int C, int N);   // narrow levels: fused weight gradients, u_pre re-computed (not stored by the forward)
// ptbase / wtt: the transposed-parameter buffer and the tiled split planes of its weight matrices (launch_tile_planes over
// it); the (C, N) for which mlp_bwd_h_nch is non-zero run their data-gradient products on them (wtt == nullptr: fp32 MFMA)
int mlp_bwd_h_nch(int C, int N);
bool qkv_bwd_uses_f16(int C, int N);
// gmax (4 unsigned, zeroed by the caller): the split kernels raise it to the bits of the largest |dx2|, |du|, |dx1| (mlp) and
// |dqkv| (qkv) of the launch - the scales of the split weight-gradient products (launch_block_dw)
bool launch_mlp_bwd(int C, int nch, const float* dx2, const float* x1, const float* upre, const BlockP& w,
                    const BlockP& wt, const float* ptbase, const void* wtt, unsigned* gmax, const BlockP& gr, float* dupre, float* dx1, float* do_hm,
                    float* a2c0, int N, int B, bool want_dw, hipStream_t s, int f16_narrow = 0, int NE = 0 /* existing tokens; see launch_attn_fwd */);
size_t attn_bwd_lds(int N, int HG, int Len);
bool attn_bwd_uses_stat2(int N, int Len, bool table);   // does launch_attn_bwd need its (B, H, N, 2) scratch for this shape?
// scratch / scratch_floats: caller-owned; attn_bwd_scratch_floats() says how much the kernels chosen for a shape need
size_t attn_bwd_scratch_floats(int N, int H, int Len, bool table, int B);
void launch_attn_bwd(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                     float* gtable, float* dqkv, float* scratch, size_t scratch_floats, int N, int H, int HG, int Len, int B,
                     int f16, hipStream_t s, int NE = 0);
// f16: S and dP tiles as fp16-pair products on the f16 matrix cores (0: exact fp32 MFMA).  attn_f16_default(): 1 unless
// RAL_ATTN_F16=0 (what the handle-free operator entry points use; a model handle follows its f16_split option)
int attn_f16_default();
// wave-autonomous kernels of the short windows (ral_attn.hip)
bool attn_fwd_w_takes(int N, int H, int Len, bool table);
void launch_attn_fwd_w(const float* qkv, float* o_hm, float* lse, const float* table, int N, int H, int Len, int B, int f16,
                       hipStream_t s);
bool attn_bwd_w_takes(int N, int H, int Len, bool table);
size_t attn_bwd_w_scratch_floats(int N, int H, int Len, bool table, int B);
void launch_attn_bwd_w(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                       float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, int f16, hipStream_t s);
int attnw_grid_max(int N, int H, int B);   // upper bound of the wave-autonomous kernels' grids (what their scratch is sized for)
void launch_attn_tpart_reduce(const float* tpart, float* gtable, int ntab, int nrow, hipStream_t s);
// deferred form: while a slot is set (attn_tab_defer_to(&slot) ... attn_tab_defer_to(nullptr)), launch_attn_tpart_reduce records
// its arguments there (ntab > 0) instead of launching; the caller launches it later with launch_attn_tpart_reduce on the stream of
// its choice (after un-setting the slot)
struct AttnTabReduce { const float* tpart; float* gtable; int ntab, nrow; };
void attn_tab_defer_to(AttnTabReduce* slot);
// one sweep, every contraction on the f16 matrix cores (ral_attnm.hip; only with f16 != 0)
bool attn_bwd_m_takes(int N, int H, int Len, bool table);
size_t attn_bwd_m_scratch_floats(int N, int H, int Len, bool table, int B);
void launch_attn_bwd_m(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                       float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, hipStream_t s);
bool attn_bwd_mh_takes(int N, int H, int Len, bool table);   // its workgroup form for N >= 256
size_t attn_bwd_mh_scratch_floats(int N, int H, int Len, bool table, int B);
void launch_attn_bwd_mh(const float* qkv, const float* o_hm, const float* do_hm, const float* lse, const float* table,
                        float* gtable, float* dqkv, float* tpart, int N, int H, int Len, int B, hipStream_t s);
size_t qkv_bwd_lds(int C, int N);
// returns true when the projection's weight / bias gradients were produced here (narrow levels: qkv_bwd_fuses_dw): the caller
// then skips that product in launch_block_dw
bool qkv_bwd_fuses_dw(int C, int N);
bool launch_qkv_bwd(int C, const float* dqkv, const float* x, const float* pe, const float* dx1, const float* extra,
                    const BlockP& w, const BlockP& wt, const float* ptbase, const void* wtt /* as launch_mlp_bwd */, unsigned* gmax,
                    const BlockP& gr, float* dx, int N, int B, bool want_dw, hipStream_t s);
void launch_resample_bwd(int D, bool sep, const float* dy, const float* x, const float* wred, const float* lnw,
                         float* g_lnw, float* g_lnb, float* dx, int T, int Tv, int B, hipStream_t s);
void launch_final_bwd(int leads, const float* dy, const float* u0, const float* x0, const float* w, float* gw,
                      float* gb, float* dz, int L, int Lp, int B, hipStream_t s);
void launch_bn8_bwd_stats(const float* dy, const float* a0, const float* ss, double* out, size_t ntok, hipStream_t s);
void launch_conv1_bwd(int leads, const float* dy, const float* a0, const float* x, const float* ss, const float* bnw,
                      const double* bst, double count, float* gw, float* gb, float* dz, int L, int Lp, int B, hipStream_t s,
                      float* gbnw, float* gbnb /* BatchNorm affine gradients += the backward sums x share */, double share);
void launch_conv1_bwd_dx(int leads, const float* dz, const float* w, float* dx, int L, int Lp, int B, hipStream_t s);

// ---- weight gradients (ral_dw.hip)
void launch_block_dw(int C, const float* dx2, const float* upre, const float* a2c0, const float* dupre, const float* x1,
                     const float* dx1, const float* o_hm, const float* dqkv, const float* x, const float* pe,
                     const BlockP& w, const BlockP& gr, int N, int B, int ksplit, bool skip_mlp, const unsigned* gmax /* 4 maxima of the split data-gradient kernels, or nullptr */, hipStream_t s,
                     bool skip_qkv = false /* the projection's weight gradient was formed inside k_qkv_bwd */);
void set_dw_lds_budget(size_t bytes);
void launch_resample_dw(int D, bool sep, const float* dy, const float* x, const float* lnw, const float* lnb,
                        float* dW, int T, int Tv, int B, int ksplit, hipStream_t s);

// db8 wavelet-threshold baseline (ral_wavelet.hip); non-zero = rejected arguments
int launch_wavelet_denoise(const float* x, float* y, long long rows, int L, float threshold, hipStream_t s);
