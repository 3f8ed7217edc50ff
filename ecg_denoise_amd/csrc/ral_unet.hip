// U-Net baseline (model/UNet.py:46-141) — placeholder until the conv kernels land.
#include "ral_unet.hpp"
#include <stdio.h>
struct UNetModel { UNetPublic pub; };
static int nyi(char* err, size_t cap) { snprintf(err, cap, "U-Net variant is not built yet"); return -1; }
int unet_check_cfg(const ral_config*, char* err, size_t cap) { return nyi(err, cap); }
int unet_layout_count(const ral_config*) { return -1; }
int unet_layout_entry(const ral_config*, int, char*, int, int32_t*, int64_t*, int32_t*, int64_t*) { return -1; }
int64_t unet_param_floats(const ral_config*) { return -1; }
int64_t unet_state_floats(const ral_config*) { return -1; }
int64_t unet_workspace_bytes(const ral_config*) { return -1; }
UNetModel* unet_create(const ral_config*, char* err, size_t cap) { nyi(err, cap); return nullptr; }
void unet_destroy(UNetModel* u) { delete u; }
int unet_bind(UNetModel*, float*, float*, float*, float*, float*, double*) { return -1; }
int unet_forward(UNetModel*, const float*, float*, int, int, hipStream_t, char* err, size_t cap) { return nyi(err, cap); }
int unet_backward(UNetModel*, const float*, float*, int, hipStream_t, char* err, size_t cap) { return nyi(err, cap); }
UNetPublic* unet_public(UNetModel* u) { return &u->pub; }
