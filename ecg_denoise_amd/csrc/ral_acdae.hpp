// ACDAE comparison baseline (model/ACDAE.py) - internal interface used by ral_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ralenet.h"

struct AcdaeModel;
struct AcdaePublic {  // leading members of AcdaeModel that ral_api.hip reads
  ral_config cfg;
  float *params, *grads, *am, *av;
  int64_t nparam;
};

int acdae_check_cfg(const ral_config* c, char* err, size_t cap);
int acdae_layout_count(const ral_config* c);
int acdae_layout_entry(const ral_config* c, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset,
                       int32_t* ndim, int64_t shape[4]);
int64_t acdae_param_floats(const ral_config* c);
int64_t acdae_workspace_bytes(const ral_config* c);
AcdaeModel* acdae_create(const ral_config* c, char* err, size_t cap);
void acdae_destroy(AcdaeModel* m);
AcdaePublic* acdae_public(AcdaeModel* m);
int acdae_bind(AcdaeModel* m, float* params, float* grads, float* am, float* av);
int acdae_forward(AcdaeModel* m, const float* x, float* y, int B, hipStream_t s, char* err, size_t cap);
int acdae_backward(AcdaeModel* m, const float* dy, float* dx, int B, hipStream_t s, char* err, size_t cap);
