// DANet comparison baseline (model/DAM.py::Seq2Seq2) - internal interface used by ral_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ralenet.h"

struct DanetModel;
struct DanetPublic {  // leading members of DanetModel that ral_api.hip reads
  ral_config cfg;
  float *params, *grads, *am, *av, *state;
  int64_t nparam, nstate;
};

int danet_check_cfg(const ral_config* c, char* err, size_t cap);
int danet_layout_count(const ral_config* c);
int danet_layout_entry(const ral_config* c, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset,
                       int32_t* ndim, int64_t shape[4]);
int64_t danet_param_floats(const ral_config* c);
int64_t danet_state_floats(const ral_config* c);
int64_t danet_workspace_bytes(const ral_config* c);
DanetModel* danet_create(const ral_config* c, char* err, size_t cap);
void danet_destroy(DanetModel* m);
DanetPublic* danet_public(DanetModel* m);
int danet_bind(DanetModel* m, float* params, float* grads, float* am, float* av, float* state);
int danet_forward(DanetModel* m, const float* x, float* y, int B, int training, hipStream_t s, char* err, size_t cap);
int danet_backward(DanetModel* m, const float* dy, float* dx, int B, hipStream_t s, char* err, size_t cap);
