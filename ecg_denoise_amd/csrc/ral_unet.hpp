// U-Net baseline (model/UNet.py) — internal interface used by ral_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ralenet.h"

struct UNetModel;
struct UNetPublic {  // leading members of UNetModel that ral_api.hip reads
  ral_config cfg;
  float *params, *grads, *am, *av, *state;
  double* bn_sums;
  int64_t nparam;
};

int unet_check_cfg(const ral_config* c, char* err, size_t cap);
int unet_layout_count(const ral_config* c);
int unet_layout_entry(const ral_config* c, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset,
                      int32_t* ndim, int64_t shape[4]);
int64_t unet_param_floats(const ral_config* c);
int64_t unet_state_floats(const ral_config* c);
int64_t unet_workspace_bytes(const ral_config* c);
UNetModel* unet_create(const ral_config* c, char* err, size_t cap);
void unet_destroy(UNetModel* u);
int unet_bind(UNetModel* u, float* params, float* grads, float* am, float* av, float* state, double* bn_sums);
int unet_forward(UNetModel* u, const float* x, float* y, int B, int training, hipStream_t s, char* err, size_t cap);
// training forward + MSE loss / SNR / RMSE sums + dy in one pass over the output (see unet_forward_loss, ral_unet.hip)
int unet_forward_loss(UNetModel* u, const float* x, const float* target, float* y, int B, float* dy, float* snr, float* rmse,
                      double* loss_sum, double* fin, double fin_scale, int fin3, hipStream_t s, char* err, size_t cap);
int unet_backward(UNetModel* u, const float* dy, float* dx, int B, hipStream_t s, char* err, size_t cap);
// staged execution (data-parallel sync-BatchNorm): gwin = windows of the global batch
int unet_forward_stage(UNetModel* u, const float* x, int B, int training, int si, int64_t gwin, hipStream_t s, char* err, size_t cap);
int unet_forward_finish(UNetModel* u, float* y, int B, int training, int64_t gwin, hipStream_t s, char* err, size_t cap);
int unet_backward_start(UNetModel* u, const float* dy, int B, int64_t gwin, hipStream_t s, char* err, size_t cap);
int unet_backward_stage(UNetModel* u, int B, int si, int64_t gwin, hipStream_t s, char* err, size_t cap);
int unet_backward_finish(UNetModel* u, int B, int64_t gwin, hipStream_t s, char* err, size_t cap);
int unet_stage_bn(int si);
int unet_set_option(UNetModel* u, const char* key, int value);   // "unet_fused": eval forward as one kernel (default 1)
UNetPublic* unet_public(UNetModel* u);
