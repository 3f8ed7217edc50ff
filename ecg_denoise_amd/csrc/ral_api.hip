// C ABI (include/ralenet.h) + host orchestration of the RA-LENet kernels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <ctype.h>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/ralenet.h"
#include "ral_kernels.hpp"
#include "ral_unet.hpp"
#include "ral_acdae.hpp"
#include "ral_danet.hpp"

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return -1;
}
extern "C" const char* ral_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------------------------------
// Switches.  Every tuning / experiment switch of the library has a compile-time default and a name in the list below.  The
// product build reads NO environment variable for them: the only way to change one is ral_global_option() (C ABI; process-
// wide, to be called before the first use - most are read once), which the tests use to pick kernels.  The library reads
// exactly two environment variables, both validated: RAL_LANES (1 .. 4 micro-batch chains) and RAL_NO_SIDE_STREAM (0 / 1).
// A diagnostic build (-DRAL_DIAG) additionally takes RAL_<NAME> from the environment for every switch (tools/diag/*.sh).
static const char* const KNOB_NAMES[] = {
  "ACDAE_DW_MFMA", "ACDAE_ENC_MFMA", "ATTNB_NT0", "ATTNF_NT0", "ATTNW_WAVES",
  "ATTN_BWD_LDS", "ATTN_BWD_M", "ATTN_BWD_MH", "ATTN_BWD_V_HI", "ATTN_BWD_V_LO", "ATTN_BWD_W", "ATTN_F16", "ATTN_FWD_H",
  "ATTN_FWD_LDS", "ATTN_FWD_V_HI", "ATTN_FWD_V_LO", "ATTN_FWD_W", "ATTN_QT1", "ATTN_QT4", "ATTN_SPLIT", "BUCKET_WAIT", "DANET_GRID_A",
  "DANET_GRID_B", "DANET_GRID_D", "DANET_GRID_F", "DANET_GRID_W", "DIAG_SKIP_WIDE_DW", "DW_F16", "DW_KSPLIT_128", "DW_KSPLIT_16",
  "DW_KSPLIT_32", "DW_KSPLIT_64", "DW_KSPLIT_8", "DW_LDS", "DW_PRIO", "DW_SETS", "F16_SPLIT", "FUSE_DW", "GRID_ATTNB", 
  "GRID_ATTNW", "GRID_FWD", "GRID_MLPB", "GRID_MLPBW", "GRID_MLPS", "GRID_MLPW", "GRID_QKVB", "GRID_QKVW", "GRID_RESB", "LOSS_GRID",
  "MLP_BWD_W", "MLP_BWD_W_F16", "MLP_F16", "MLP_FWD_W", "MLP_HLDS", "MLP_HTHREADS", "MLPB_HTHREADS", "MLP_LDS", "MLP_TOK", "PREP_OVERLAP", "QKVB_F16", "QKVB_FDW", "QKVB_SEG", "TABRED_SIDE", "QKV_WS", "UNET_BWD_GRID",
  "UNET_BWD_WP", "UNET_DEBUG", "UNET_EVAL_GRID", "UNET_FOLD", "UNET_FUSED", "UNET_FWD_GRID", "UNET_NREP", "UNET_WG_PER_CU"};
static std::mutex g_knob_mu;
static std::map<std::string, long long>& knob_table() { static std::map<std::string, long long> t; return t; }
static std::map<std::string, int>& knob_read() { static std::map<std::string, int> t; return t; }   // switches a consumer has read already
// grid caps, thread counts and split counts: 0 would be a launch with no workgroup (a launch error), not "automatic"
static const char* const KNOB_MIN1[] = {
  "DANET_GRID_A", "DANET_GRID_B", "DANET_GRID_D", "DANET_GRID_F", "DANET_GRID_W", "GRID_FWD", "GRID_ATTNB", "GRID_MLPB", "GRID_MLPS",
  "GRID_QKVB", "GRID_RESB", "LOSS_GRID", "UNET_FWD_GRID", "UNET_EVAL_GRID", "UNET_BWD_GRID", "UNET_BWD_WP", "UNET_NREP", "DW_SETS",
  "DW_KSPLIT_8", "DW_KSPLIT_16", "DW_KSPLIT_32", "DW_KSPLIT_64", "DW_KSPLIT_128"};
long long ral_knob(const char* name, long long dflt) {
  {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    knob_read()[name] = 1;
    auto it = knob_table().find(name);
    if (it != knob_table().end()) return it->second;
  }
#ifdef RAL_DIAG
  const std::string e = std::string("RAL_") + name;
  if (const char* v = getenv(e.c_str())) return atoll(v);
#endif
  return dflt;
}
// a validated integer environment variable (the product build has two): out-of-range or non-numeric text keeps the default
int ral_env_int(const char* name, int dflt, int lo, int hi) {
  const char* v = getenv(name);
  if (!v || !*v) return dflt;
  char* end = nullptr;
  const long x = strtol(v, &end, 10);
  if (end == v || *end != 0 || x < lo || x > hi) {
    fprintf(stderr, "libralenet: %s=\"%s\" ignored (an integer in [%d, %d] is expected)\n", name, v, lo, hi);
    return dflt;
  }
  return (int)x;
}
int ral_num_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  hipDeviceProp_t prop;
  const int n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  cus[dev] = n;
  return n;
}
int ral_occupancy(const void* kernel, int threads, size_t lds, int dflt) {
  static std::mutex mu;
  static std::map<std::tuple<const void*, int, size_t>, int> cache;
  const auto key = std::make_tuple(kernel, threads, lds);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess || occ < 1) occ = dflt;
  cache[key] = occ;
  return occ;
}
extern "C" int ral_global_option(const char* key, long long value) {
  if (!key) return fail("null key");
  std::string k(key);
  for (auto& c : k) c = (char)toupper((unsigned char)c);
  if (k.rfind("RAL_", 0) == 0) k = k.substr(4);
  bool known = false;
  for (const char* n : KNOB_NAMES) known = known || k == n;
  if (!known) return fail("unknown switch %s", key);
  if (value < 0) return fail("switch %s: negative value %lld", key, value);
  for (const char* n : KNOB_MIN1)
    if (k == n && value < 1) return fail("switch %s: value %lld out of range (a grid / thread / split count: >= 1)", key, value);
  if (k == "MLP_HTHREADS" && value != 256 && value != 512 && value != 1024) return fail("switch %s: 256, 512 or 1024 threads (got %lld)", key, value);
  if (k == "MLPB_HTHREADS" && value != 256 && value != 512) return fail("switch %s: 256 or 512 threads (got %lld)", key, value);
  std::lock_guard<std::mutex> lk(g_knob_mu);
  // most consumers read a switch once and keep it: a change after that first read would be ignored silently, so it is refused
  // (the same value again is fine: tests and tools set their switches at start-up, possibly more than once)
  auto cur = knob_table().find(k);
  if (knob_read().count(k) && !(cur != knob_table().end() && cur->second == value))
    return fail("switch %s was already read by the library (switches are latched at first use): set it before the first model is created", key);
  knob_table()[k] = value;
  return 0;
}

#define HIP_OK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

// Event records / stream waits of the fork-join scheduler: a failed one would silently drop an ordering edge (a race,
// not an error), so the first failure is remembered and the entry point that issued it returns it (sched_check).
static thread_local hipError_t g_sched_err = hipSuccess;
static thread_local const char* g_sched_what = "";
#define EV(expr)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess && g_sched_err == hipSuccess) { g_sched_err = e_; g_sched_what = #expr; } \
  } while (0)
// every entry point that records events / stream waits starts from a clean slate (an error remembered by a call that left
// through another failure must not surface in a later, unrelated one) and ends with sched_check()
static void sched_reset() { g_sched_err = hipSuccess; g_sched_what = ""; }
static int sched_check() {
  if (g_sched_err == hipSuccess) return 0;
  const hipError_t e = g_sched_err;
  g_sched_err = hipSuccess;
  return fail("stream scheduling: %s failed: %s (an ordering edge was lost: the results of this call are undefined)",
              g_sched_what, hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------
// layout
// ---------------------------------------------------------------------------------
static const int CH[5] = {8, 16, 32, 64, 128};
static const int RWLEN[4] = {32, 16, 8, 4};
struct StageDef { const char* name; int level; int rw; };
static const StageDef STAGES[9] = {
    {"dtransformer1", 0, 1}, {"dtransformer2", 1, 2}, {"dtransformer3", 2, 3}, {"dtransformer34", 3, 4},
    {"transformer", 4, 0},   {"utransformer4", 4, 0}, {"utranformer3", 3, 4},  {"utransformer2", 2, 3},
    {"utransformer1", 1, 2}};

struct Entry {
  std::string name;
  int kind;
  int64_t offset;
  int ndim;
  int64_t shape[4];
};

struct BlockOff { int64_t wqkv, bqkv, wp, bp, ln1w, ln1b, ln2w, ln2b, w1, b1, w2, b2, le; };
struct ResOff { int64_t w, lnw, lnb; int D; };

struct Layout {
  std::vector<Entry> entries;
  int64_t nparam = 0, nstate = 0;
  int64_t dec_off = 0;   // first float of the decoder's parameters (utransformer4 .. transconv are contiguous up to nparam)
  int64_t conv1_w, conv1_b, bn_w, bn_b, tc_w, tc_b;
  int64_t rw[4] = {-1, -1, -1, -1};
  BlockOff blk[18];
  ResOff res[8];  // pm1..pm4, ps4..ps1
  bool le = true, rwave = false;
};

static int64_t alloc_f(int64_t& cur, int64_t n) {
  const int64_t o = cur;
  cur += (n + 7) & ~int64_t(7);   // 32-byte granules: the fp16 planes of a weight matrix are read 16 bytes (8 elements) at a time
  return o;
}

static void push(Layout& L, const std::string& name, int kind, int64_t off, std::initializer_list<int64_t> shp) {
  Entry e;
  e.name = name; e.kind = kind; e.offset = off; e.ndim = (int)shp.size();
  int i = 0;
  for (auto s : shp) e.shape[i++] = s;
  for (; i < 4; ++i) e.shape[i] = 1;
  L.entries.push_back(e);
}

static bool build_layout(const ral_config& c, Layout& L) {
  if (c.variant < RAL_NRA || c.variant > RAL_MLP) return false;
  const bool basic = c.variant != RAL_NRA;
  L.le = c.variant != RAL_MLP;
  L.rwave = c.variant != RAL_NRA;
  int64_t cur = 0;
  const int ld = c.leads;
  L.conv1_w = alloc_f(cur, 8 * ld * 3); L.conv1_b = alloc_f(cur, 8);
  L.bn_w = alloc_f(cur, 8); L.bn_b = alloc_f(cur, 8);
  push(L, "conv1.0.weight", RAL_PARAM, L.conv1_w, {8, ld, 3});
  push(L, "conv1.0.bias", RAL_PARAM, L.conv1_b, {8});
  push(L, "conv1.2.weight", RAL_PARAM, L.bn_w, {8});
  push(L, "conv1.2.bias", RAL_PARAM, L.bn_b, {8});
  push(L, "conv1.2.running_mean", RAL_STATE_F32, 0, {8});
  push(L, "conv1.2.running_var", RAL_STATE_F32, 8, {8});
  push(L, "conv1.2.num_batches_tracked", RAL_COUNTER_I64, 0, {});
  L.nstate = 16;
  if (L.rwave) {
    for (int i = 0; i < 4; ++i) {
      const int h = CH[i] / 4, n = 2 * RWLEN[i] - 1;
      L.rw[i] = alloc_f(cur, (int64_t)n * h);
      const std::string p = "rwattn" + std::to_string(i + 1);
      push(L, p + ".relative_position_bias_table", RAL_PARAM, L.rw[i], {n, h});
      push(L, p + ".relative_position_index", RAL_INDEX_I64, 0, {RWLEN[i], RWLEN[i]});
    }
  }
  auto add_block = [&](int bi, const std::string& pre, int C) {
    BlockOff& b = L.blk[bi];
    b.wqkv = alloc_f(cur, 3 * C * C); b.bqkv = alloc_f(cur, 3 * C);
    b.wp = alloc_f(cur, C * C); b.bp = alloc_f(cur, C);
    b.ln1w = alloc_f(cur, C); b.ln1b = alloc_f(cur, C); b.ln2w = alloc_f(cur, C); b.ln2b = alloc_f(cur, C);
    b.w1 = alloc_f(cur, 4 * C * C); b.b1 = alloc_f(cur, 4 * C);
    b.w2 = alloc_f(cur, 4 * C * C); b.b2 = alloc_f(cur, C);
    b.le = L.le ? alloc_f(cur, 3) : -1;
    push(L, pre + "attn.qkv_proj.to_q.weight", RAL_PARAM, b.wqkv, {C, C});
    push(L, pre + "attn.qkv_proj.to_q.bias", RAL_PARAM, b.bqkv, {C});
    push(L, pre + "attn.qkv_proj.to_kv.weight", RAL_PARAM, b.wqkv + C * C, {2 * C, C});
    push(L, pre + "attn.qkv_proj.to_kv.bias", RAL_PARAM, b.bqkv + C, {2 * C});
    push(L, pre + "attn.proj.weight", RAL_PARAM, b.wp, {C, C});
    push(L, pre + "attn.proj.bias", RAL_PARAM, b.bp, {C});
    push(L, pre + "norm1.weight", RAL_PARAM, b.ln1w, {C});
    push(L, pre + "norm1.bias", RAL_PARAM, b.ln1b, {C});
    push(L, pre + "norm2.weight", RAL_PARAM, b.ln2w, {C});
    push(L, pre + "norm2.bias", RAL_PARAM, b.ln2b, {C});
    push(L, pre + "mlp.fc1.weight", RAL_PARAM, b.w1, {4 * C, C});
    push(L, pre + "mlp.fc1.bias", RAL_PARAM, b.b1, {4 * C});
    push(L, pre + "mlp.fc2.weight", RAL_PARAM, b.w2, {C, 4 * C});
    push(L, pre + "mlp.fc2.bias", RAL_PARAM, b.b2, {C});
    if (L.le) push(L, pre + "mlp.leconv.partial_conv3.weight", RAL_PARAM, b.le, {1, 1, 3});
  };
  auto add_stage = [&](int si) {
    for (int i = 0; i < 2; ++i) {
      const std::string pre = std::string(STAGES[si].name) + (basic ? ".blocks." : ".") + std::to_string(i) + ".";
      add_block(si * 2 + i, pre, CH[STAGES[si].level]);
    }
  };
  auto add_res = [&](int ri, const std::string& name, int D) {
    ResOff& r = L.res[ri];
    r.D = D;
    r.w = alloc_f(cur, (int64_t)D * D); r.lnw = alloc_f(cur, D); r.lnb = alloc_f(cur, D);
    push(L, name + ".reduction.weight", RAL_PARAM, r.w, {D, D});
    push(L, name + ".norm.weight", RAL_PARAM, r.lnw, {D});
    push(L, name + ".norm.bias", RAL_PARAM, r.lnb, {D});
  };
  add_stage(0); add_res(0, "pm1", 16);
  add_stage(1); add_res(1, "pm2", 32);
  add_stage(2); add_res(2, "pm3", 64);
  add_stage(3); add_res(3, "pm4", 128);
  add_stage(4);
  add_stage(5); add_res(4, "ps4", 64);
  add_stage(6); add_res(5, "ps3", 32);
  add_stage(7); add_res(6, "ps2", 16);
  add_stage(8); add_res(7, "ps1", 8);
  L.tc_w = alloc_f(cur, ld * 8 * 3); L.tc_b = alloc_f(cur, ld);
  push(L, "transconv.0.weight", RAL_PARAM, L.tc_w, {ld, 8, 3});
  push(L, "transconv.0.bias", RAL_PARAM, L.tc_b, {ld});
  L.nparam = cur;
  L.dec_off = cur;
  for (const Entry& e : L.entries)
    if (e.kind == RAL_PARAM && e.name.rfind("utransformer4.", 0) == 0 && e.offset < L.dec_off) L.dec_off = e.offset;
  return true;
}

static int check_cfg(const ral_config* c) {
  if (!c) return fail("null config");
  if (c->variant == RAL_UNET) return unet_check_cfg(c, g_err, sizeof(g_err));
  if (c->variant == RAL_ACDAE) return acdae_check_cfg(c, g_err, sizeof(g_err));
  if (c->variant == RAL_DANET) return danet_check_cfg(c, g_err, sizeof(g_err));
  if (c->variant < 0 || c->variant > RAL_DANET) return fail("unknown variant %d", c->variant);
  if (c->leads != 1 && c->leads != 2) return fail("leads must be 1 or 2 (got %d); use the 12-lead adapter above it", c->leads);
  // the reference's own constraint (raletransformer.py:170,448-450): four PatchMerging halvings -> a multiple of 16 (its positional
  // table ends at 1000; 1024 is accepted here).  The R-wave variants also need their (32, 16, 8, 4)-token windows to fit.
  if (c->L <= 0 || c->L % 16 != 0 || c->L > 1024) return fail("L must be a multiple of 16 and <= 1024 (got %d)", c->L);
  if (c->variant != RAL_NRA && c->L < 32) return fail("L must be at least 32 for the R-wave variants (got %d)", c->L);
  if (c->max_batch <= 0) return fail("max_batch must be positive");
  return 0;
}

// positional encoding in the reference's op order, fp32 (raletransformer.py:172-181)
static void fill_pe(float* P, int n, int C) {
  for (int i = 0; i < C / 2; ++i) {
    const float den = powf(10000.0f, (float)(2 * i) / (float)C);
    for (int p = 0; p < n; ++p) {
      const float X = (float)p / den;
      P[(size_t)p * C + 2 * i] = sinf(X);
      P[(size_t)p * C + 2 * i + 1] = cosf(X);
    }
  }
}

// ---------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------
struct BlockAct { float *in, *qkv, *o, *lse, *x1, *upre, *out; };

// Sets of per-block backward temporaries (dx1, do, dqkv, du_pre, ...): block i of a lane's chain may start once the
// weight-gradient kernels of block i - dw_sets() have finished with theirs.  At the wide levels a block's four
// weight-gradient launches take about as long as its chain, so with two sets the chain waits for them; with more it runs
// ahead and the side stream catches up under the narrow levels, whose weight gradients are fused into the chain kernels.
// Measured at batch 2048 (ms per step): 2 sets 17.09, 3: 17.08, 4: 17.04, 6: 17.00, 8: 16.99; a set is 9.6 E floats
// (0.32 GB at batch 2048).  RAL_DW_SETS overrides (2 .. MAX_SETS).
#define MAX_SETS 8
static int dw_sets() {
  static const int n = [] {
    int k = (int)ral_knob("DW_SETS", 6);
    return k < 2 ? 2 : (k > MAX_SETS ? MAX_SETS : k);
  }();
  return n;
}

struct RalModel {
  ral_config cfg;
  Layout lay;
  int L, Lp, E1;  // L: samples per window; Lp: token slots per window at level 0 = L rounded up to a multiple of 256 (every level then
                  // has a multiple of 16 slots; the slots past L >> level do not exist for the model: masked keys, zero conv halos, no
                  // gradient); E1 = 8 * Lp floats per window per stage tensor
  char* slab = nullptr;
  size_t slab_bytes = 0;
  std::map<std::string, std::pair<float*, int64_t>> dbg;
  // bound buffers
  float *params = nullptr, *grads = nullptr, *am = nullptr, *av = nullptr, *state = nullptr;
  double* bn_sums = nullptr;
  // workspace
  float* pe[5];
  float *xin, *a0, *x0, *ss;
  BlockAct act[18];
  float* res_out[8];  // p1..p4 (pm), u3,u2,u1,u0 (ps)
  float* xmid;
  // backward temporaries
  // every gradient tensor has its own buffer (no ping-pong): the weight-gradient kernels run on a side
  // stream and read them long after the data-gradient chain has moved on
  float *gy[18], *gin[9], *du0, *dz0;
  float *dx1[MAX_SETS], *dohm[MAX_SETS], *dqkv[MAX_SETS], *dupre[MAX_SETS], *a2c0[MAX_SETS], *astat[MAX_SETS];   // per-block temporaries, dw_sets() sets (side-stream overlap)
  void* lanes = nullptr;   // LaneSet
  int n_lanes = 2;
  bool side_stream = true;
  bool attn_f16 = true;      // attention score tiles (S, dP) as fp16-pair products - only while f16_split > 0 (option "attn_f16"; RAL_ATTN_F16=0)
  bool narrow_f16 = true;    // the strip kernel of the narrow levels' MLP forward on fp16-pair products - only while f16_split > 0 (option "narrow_f16")
  int f16_split = 64;         // narrowest width whose Linear layers (q/k/v projection, proj, fc1, fc2 of the forward) run as
                              // two-piece fp16 products on the f16 matrix cores (0 = none: fp32 MFMA everywhere)
  bool want_dw = true;      // false inside ral_backward_input: frozen weights, data gradients only
  int dec_lanes = 0; bool dec_side = false;   // lanes / side streams that carried the last backward (bucket events)
  hipEvent_t ev_bwd_done = nullptr;          // recorded at the end of ral_backward_end
  // weight preparation under the stem: the split planes of this forward (and, in training, the transposes + their planes for
  // its backward) are formed on lane 0's weight-gradient stream - idle during a forward - while the caller's stream runs the
  // stem conv / BatchNorm; fwd_end / bwd_begin wait for the events instead of launching the kernels (side_stream = 0: inline)
  hipEvent_t ev_prep_go = nullptr, ev_prep_fwd = nullptr, ev_prep_bwd = nullptr;
  bool prep_fwd = false, prep_bwd = false;
  // option "static_params" (the host mirror sets it in eval mode): the caller promises not to touch the parameters until it says so
  // again, so the activation scales and the weight planes an eval forward formed stay valid for the next one - a forward then has no
  // preparation kernels at all (they were ~0.1 ms of a 3.6 ms inference forward)
  bool static_params = false, planes_valid = false, skip_prep = false;
  bool sums_clean = false;   // bn_sums[0, 32) were zeroed by the optimiser kernel of the previous step (no fill kernel in front of the stem)
  bool prep_stale = false;   // parameters or arithmetic options changed after the preparation was queued: the backward re-builds its planes
  bool bwd_recorded = false;
  float* paramsT = nullptr;   // transposed copies of the weight matrices (same offsets), refreshed per backward
  unsigned short* wh = nullptr;             // tiled split planes of the wide levels' weight matrices (a matrix at twice its float
                                            // offset), re-written every forward from the descriptors below
  float* ascale = nullptr;                  // device: ASC_N activation scales per transformer block (k_act_scales, once per forward)
  int* adesc = nullptr;                     // device: 12 ints per block (its descriptor)
  int* wdesc = nullptr;                     // device: int4 {offset, rows, columns, first work item} per matrix
  int ndesc = 0, nwork = 0;
  unsigned* gmax = nullptr;                 // (18 blocks x 4 lanes x 4) largest-magnitude bits of dx2 / du / dx1 / dqkv per block and lane: scales of the split weight-gradient products; zeroed per backward
  unsigned short* whT = nullptr;            // the same for the transposed matrices of the backward (from paramsT), training only
  int* wdescT = nullptr;
  int ndescT = 0, nworkT = 0;
  void* tdesc = nullptr; int tn = 0, ttotal = 0;
  const float* last_x = nullptr;
  int last_B = 0;
  int nch_f[5], nch_b[5], hg_f[5], hg_b[5];
  int dw_ksplit[5] = {256, 256, 256, 256, 256};
  // optional in-library kernel timing (bench.py roofline leg): hipEvent pairs around the
  // launches of ONE selected kernel kind, on the stream the kernels run on
  int prof_kind = -1;                       // a kernel kind, -1 = off, K_ALL = every kind (ral_profile_timeline)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  std::vector<std::pair<int, hipStream_t>> prof_meta;   // (kind, stream) of each recorded pair
  size_t prof_used = 0;
};

enum { K_QKV_FWD = 0, K_ATTN_FWD, K_MLP_FWD, K_MLP_BWD, K_ATTN_BWD, K_QKV_BWD, K_DW, K_RES_FWD, K_RES_BWD, K_STEM, K_NKINDS };
static const char* KIND_NAMES[K_NKINDS] = {"qkv_fwd", "attn_fwd", "mlp_fwd", "mlp_bwd", "attn_bwd", "qkv_bwd",
                                           "dw", "resample_fwd", "resample_bwd", "stem"};

enum { K_ALL = 1000 };
struct ProfScope {
  RalModel* m; hipStream_t s; bool on;
  ProfScope(RalModel* m_, int kind, hipStream_t s_) : m(m_), s(s_), on(m_->prof_kind == kind || m_->prof_kind == K_ALL) {
    if (!on) return;
    if (m->prof_used == m->prof_ev.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {   // no timing for this launch
        if (a) (void)hipEventDestroy(a);
        on = false;
        return;
      }
      m->prof_ev.push_back({a, b});
      m->prof_meta.push_back({kind, s});
    }
    m->prof_meta[m->prof_used] = {kind, s};
    EV(hipEventRecord(m->prof_ev[m->prof_used].first, s));
  }
  ~ProfScope() {
    if (!on) return;
    EV(hipEventRecord(m->prof_ev[m->prof_used].second, s));
    m->prof_used++;
  }
};

struct ral_handle {
  int kind;  // 0 ralenet, 1 unet, 2 acdae, 3 danet
  RalModel* m;
  UNetModel* u;
  AcdaeModel* a = nullptr;
  DanetModel* d = nullptr;
};

static size_t plan_workspace(const ral_config& c, RalModel* m /* may be null: size only */, char* base) {
  size_t cur = 0;
  const int Lp = (c.L + 255) / 256 * 256;      // token slots per window (RalModel::Lp)
  const size_t B = c.max_batch, E = (size_t)8 * Lp * B;
  auto take = [&](const char* name, size_t nfloat) -> float* {
    float* p = base ? reinterpret_cast<float*>(base + cur) : nullptr;
    if (m && base) m->dbg[name] = {p, (int64_t)nfloat};
    cur += ((nfloat * sizeof(float)) + 255) & ~size_t(255);
    return p;
  };
  RalModel tmp_;
  RalModel& M = m ? *m : tmp_;
  for (int l = 0; l < 5; ++l) M.pe[l] = take(("pe" + std::to_string(l)).c_str(), (size_t)(Lp >> l) * CH[l]);
  M.a0 = take("a0", E); M.x0 = take("x0", E); M.ss = take("bn_ss", 64);
  const bool tr = c.train != 0;
  float *sh_qkv = nullptr, *sh_o = nullptr;
  if (!tr) { sh_qkv = take("qkv", 3 * E); sh_o = take("o", E); }
  for (int b = 0; b < 18; ++b) {
    const std::string p = "blk" + std::to_string(b) + ".";
    BlockAct& a = M.act[b];
    a.qkv = tr ? take((p + "qkv").c_str(), 3 * E) : sh_qkv;
    a.o = tr ? take((p + "o").c_str(), E) : sh_o;
    a.lse = tr ? take((p + "lse").c_str(), E / 4) : nullptr;
    a.x1 = tr ? take((p + "x1").c_str(), E) : nullptr;
    // u_pre is kept only where the backward reads it: the narrow levels re-compute it (k_mlp_bwd_s)
    const int lvl = STAGES[b / 2].level;
    a.upre = (tr && !mlp_bwd_is_fused(CH[lvl], Lp >> lvl)) ? take((p + "upre").c_str(), 4 * E) : nullptr;
    a.out = take((p + "out").c_str(), E);
  }
  { Layout L_; build_layout(c, L_); M.wh = reinterpret_cast<unsigned short*>(take("wh", (size_t)L_.nparam)); }
  M.wdesc = reinterpret_cast<int*>(take("wdesc", 4 * 64));
  M.ascale = take("ascale", 18 * ASC_N);
  M.adesc = reinterpret_cast<int*>(take("adesc", 18 * 12));
  static const char* RN[8] = {"p1", "p2", "p3", "p4", "u3", "u2", "u1", "u0"};
  for (int r = 0; r < 8; ++r) M.res_out[r] = take(RN[r], E);
  M.xmid = take("xmid", E);
  if (tr) {
    for (int i = 0; i < 18; ++i) M.gy[i] = (i == 9) ? nullptr : take(("gy" + std::to_string(i)).c_str(), E);
    for (int i = 0; i < 9; ++i) M.gin[i] = take(("gin" + std::to_string(i)).c_str(), E);
    M.gy[9] = M.gin[5];   // x_mid = transformer(x4) + x4: the stage-4 output gradient IS g x_mid
    M.du0 = take("du0", E);
    for (int k = 0; k < dw_sets(); ++k) {
      M.dx1[k] = take(("dx1_" + std::to_string(k)).c_str(), E); M.dohm[k] = take(("do_" + std::to_string(k)).c_str(), E);
      M.dqkv[k] = take(("dqkv_" + std::to_string(k)).c_str(), 3 * E); M.dupre[k] = take(("dupre_" + std::to_string(k)).c_str(), 4 * E);
      M.a2c0[k] = take(("a2c0_" + std::to_string(k)).c_str(), E / 8);
      // attention backward scratch: (B, H, N, 2) = E / 2 floats from sweep Q to sweep KV, then (B, 2, H, 64) table-gradient
      // partials (H <= 16 on the levels that have a table: at most 2048 floats per window)
      M.astat[k] = take(("astat_" + std::to_string(k)).c_str(), E / 2 + (size_t)B * 2048);
    }
    M.dz0 = take("dz0", E);
    Layout L_; build_layout(c, L_);
    M.paramsT = take("paramsT", (size_t)L_.nparam);
    M.whT = reinterpret_cast<unsigned short*>(take("whT", (size_t)L_.nparam));
    M.wdescT = reinterpret_cast<int*>(take("wdescT", 4 * 64));
    M.gmax = reinterpret_cast<unsigned*>(take("gmax", 18 * 4 * 4));
    M.tdesc = take("tdesc", 4 * 128);
  }
  return cur;
}

static size_t env_size(const char* name, size_t dflt) { return (size_t)ral_knob(name + 4, (long long)dflt); }   // (name = "RAL_<KNOB>")

// head-group size of the attention kernels: the largest power-of-two fraction of the heads whose tiles fit the LDS budget
static int attn_head_group(int N, int H, int Len, bool bwd) {
  const size_t budget = bwd ? env_size("RAL_ATTN_BWD_LDS", 78 * 1024) : env_size("RAL_ATTN_FWD_LDS", 72 * 1024);
  int hg = H;
  while (hg > 1 && (bwd ? attn_bwd_lds(N, hg, Len) : attn_fwd_lds(N, hg, Len)) > budget) hg /= 2;
  return hg;
}

static void choose_tiling(RalModel* m) {
  // tuning knobs (bytes of LDS a workgroup may take; fewer bytes = more hidden chunks / smaller head
  // groups but more co-resident workgroups per CU).  Environment overrides are for experiments only.
  const size_t budget = env_size("RAL_MLP_LDS", 78000);
  // split-K workgroups of a fully sliced weight-gradient product per channel width {8,16,32,64,128} (products with
  // fewer slices get more, up to RAL_DW_MINWG workgroups per launch - see ral_dw.hip for why that is 192)
  static const int KS_DEFAULT[5] = {128, 128, 128, 64, 32};
  for (int l = 0; l < 5; ++l) m->dw_ksplit[l] = KS_DEFAULT[l];
  {   // DW_KSPLIT_<width>: split-K workgroups of one channel width
    static const char* const KSN[5] = {"DW_KSPLIT_8", "DW_KSPLIT_16", "DW_KSPLIT_32", "DW_KSPLIT_64", "DW_KSPLIT_128"};
    for (int l = 0; l < 5; ++l) { const int v = (int)ral_knob(KSN[l], m->dw_ksplit[l]); if (v > 0) m->dw_ksplit[l] = v; }
  }
  set_dw_lds_budget(env_size("RAL_DW_LDS", 76 * 1024));
  for (int l = 0; l < 5; ++l) {
    const int C = CH[l], N = m->Lp >> l, H = C / 4;
    int n = 1;
    while (n < 4 && mlp_fwd_lds(C, N, n) > budget) n *= 2;
    m->nch_f[l] = n;
    n = 1;
    while (n < 4 && mlp_bwd_lds(C, N, n) > budget) n *= 2;
    m->nch_b[l] = n;
    const int Len = l < 4 ? RWLEN[l] : 0;
    m->hg_f[l] = attn_head_group(N, H, Len, false);
    m->hg_b[l] = attn_head_group(N, H, Len, true);
  }
}

static BlockP block_ptrs(const BlockOff& o, float* base) {
  BlockP p;
  p.wqkv = base + o.wqkv; p.bqkv = base + o.bqkv; p.wp = base + o.wp; p.bp = base + o.bp;
  p.ln1w = base + o.ln1w; p.ln1b = base + o.ln1b; p.ln2w = base + o.ln2w; p.ln2b = base + o.ln2b;
  p.w1 = base + o.w1; p.b1 = base + o.b1; p.w2 = base + o.w2; p.b2 = base + o.b2;
  p.le = o.le >= 0 ? base + o.le : nullptr;
  p.asc = nullptr;
  return p;
}

// ---------------------------------------------------------------------------------
// lanes: the windows of a batch are independent through the whole transformer stack, so a batch is cut
// into NL contiguous micro-batches ("lanes") that run the same kernel chain on separate HIP streams.
// Kernels of different kinds (VALU-bound attention, MFMA/latency-bound projections) then share the CUs.
// Every tensor is batch-major, so a lane is just a pointer offset of w0 windows.
// ---------------------------------------------------------------------------------
struct Lane {
  int w0 = 0, B = 0;
  hipStream_t s = nullptr, s2 = nullptr;     // chain stream, weight-gradient side stream
  hipEvent_t ev_ready[MAX_SETS] = {}, ev_done[MAX_SETS] = {}, ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_dec_main = nullptr, ev_dec_side = nullptr;   // decoder half of the gradients complete (this lane)
  bool dw_pending[MAX_SETS] = {};
  int bwd_count = 0;
};
#ifndef MAX_LANES
#define MAX_LANES 4
#endif
struct LaneSet { Lane l[MAX_LANES]; int n = 1; hipStream_t own[MAX_LANES] = {}; };
static LaneSet* lanes_of(RalModel* m);

template <class T> static inline T* woff(T* p, int w0, size_t per_window) { return p ? p + (size_t)w0 * per_window : p; }

// ---------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------
static void run_block_fwd(RalModel* m, int bi, const float* in, bool training, const Lane& ln, const float* addend = nullptr, float* sum_out = nullptr) {
  const int si = bi / 2, l = STAGES[si].level, C = CH[l], N = m->Lp >> l, H = C / 4;
  const size_t E1 = m->E1;
  BlockP w = block_ptrs(m->lay.blk[bi], m->params);
  w.asc = m->f16_split > 0 ? m->ascale + ASC_N * bi : nullptr;
  BlockAct& a = m->act[bi];
  a.in = const_cast<float*>(in);   // base pointer (window 0); lanes offset it
  const float* table = nullptr;
  int Len = 0;
  if (m->lay.rwave && STAGES[si].rw) {
    table = m->params + m->lay.rw[STAGES[si].rw - 1];
    Len = RWLEN[STAGES[si].rw - 1];
  }
  const int w0 = ln.w0, B = ln.B;
  hipStream_t s = ln.s;
  const bool split = m->f16_split > 0 && C >= m->f16_split;
  const float* x = woff(in, w0, E1);
  float* qkv = woff(a.qkv, w0, 3 * E1);
  float* o = woff(a.o, w0, E1);
  { ProfScope p(m, K_QKV_FWD, s); launch_qkv_fwd(C, x, m->pe[l], w, split ? m->wh + 2 * m->lay.blk[bi].wqkv : nullptr, qkv, N, B, s); }
  const int NE = m->L >> l;   // existing tokens of the N slots (NE < N: a window length that is not a multiple of 256)
  { ProfScope p(m, K_ATTN_FWD, s);
    launch_attn_fwd(qkv, o, training ? woff(a.lse, w0, E1 / 4) : nullptr, table, N, H, m->hg_f[l], Len, B,
                    (m->f16_split > 0 && m->attn_f16) ? 1 : 0, s, NE); }
  { ProfScope p(m, K_MLP_FWD, s);
    launch_mlp_fwd(C, m->nch_f[l], x, o, w, m->params, split ? m->wh : nullptr, training ? woff(a.x1, w0, E1) : nullptr,
                   (training && !mlp_bwd_is_fused(C, N)) ? woff(a.upre, w0, 4 * E1) : nullptr,
                   woff(a.out, w0, E1), N, B, (m->f16_split > 0 && m->narrow_f16) ? 1 : 0, s, NE, woff(addend, w0, E1), woff(sum_out, w0, E1)); }
}

static const float* run_stage_fwd(RalModel* m, int si, const float* in, bool training, const Lane& ln) {
  run_block_fwd(m, si * 2, in, training, ln);
  run_block_fwd(m, si * 2 + 1, m->act[si * 2].out, training, ln);
  return m->act[si * 2 + 1].out;
}

static void run_res_fwd(RalModel* m, int ri, const float* in, const float* skip, const Lane& ln) {
  const ResOff& r = m->lay.res[ri];
  const int T = m->E1 / r.D, Tv = 8 * m->L / r.D;  // output token slots per window, and how many of them exist
  ProfScope p(m, K_RES_FWD, ln.s);
  launch_resample_fwd(r.D, ri >= 4, woff(in, ln.w0, m->E1), m->params + r.w, m->params + r.lnw, m->params + r.lnb,
                      woff(skip, ln.w0, m->E1), woff(m->res_out[ri], ln.w0, m->E1), T, Tv, ln.B, ln.s);
}

// split [0, B) over the lanes; lane 0 runs on the caller's stream
static int plan_lanes(RalModel* m, int B, hipStream_t s) {
  LaneSet* L = lanes_of(m);
  int n = m->n_lanes;
  if (B < 64 * n || B % n != 0) n = 1;
  for (int i = 0; i < n; ++i) {
    Lane& ln = L->l[i];
    ln.w0 = i * (B / n); ln.B = B / n;
    ln.s = i == 0 ? s : L->own[i];
    ln.bwd_count = 0;
    for (int k = 0; k < MAX_SETS; ++k) ln.dw_pending[k] = false;
  }
  L->n = n;
  return n;
}

static void fork_lanes(RalModel* m, hipStream_t s) {     // lanes 1.. start after everything queued on s so far
  LaneSet* L = lanes_of(m);
  if (L->n < 2) return;
  EV(hipEventRecord(L->l[0].ev_fork, s));
  for (int i = 1; i < L->n; ++i) EV(hipStreamWaitEvent(L->l[i].s, L->l[0].ev_fork, 0));
}
static void join_lanes(RalModel* m, hipStream_t s) {     // s continues after every lane has finished
  LaneSet* L = lanes_of(m);
  for (int i = 1; i < L->n; ++i) {
    EV(hipEventRecord(L->l[i].ev_join, L->l[i].s));
    EV(hipStreamWaitEvent(s, L->l[i].ev_join, 0));
  }
}

static int fwd_begin(RalModel* m, const float* x, int B, int training, hipStream_t s) {
  sched_reset();
  if (!m->params || !m->state) return fail("ral_bind was not called");
  if (B <= 0 || B > m->cfg.max_batch) return fail("batch %d outside (0, max_batch=%d]", B, m->cfg.max_batch);
  if (training && !m->cfg.train) return fail("handle was created with train=0");
  if (training && !m->bn_sums) return fail("training forward needs bn_sums bound");
  const Layout& Y = m->lay;
  m->last_x = x; m->last_B = B;
  m->prep_fwd = m->prep_bwd = false;
  m->prep_stale = false;
  m->skip_prep = !training && m->static_params && m->planes_valid;
  if (training) m->planes_valid = false;                         // (an optimiser step will follow)
  else if (m->static_params) m->planes_valid = true;             // (what this forward prepares stays)
  static const bool prep_on = ral_knob("PREP_OVERLAP", 1) != 0;
  if (!m->skip_prep && prep_on && m->side_stream && m->ev_prep_go) {
    hipStream_t ps = lanes_of(m)->l[0].s2;
    HIP_OK(hipEventRecord(m->ev_prep_go, s));            // (the parameters are final: everything queued on s so far has run)
    HIP_OK(hipStreamWaitEvent(ps, m->ev_prep_go, 0));
    if (m->f16_split > 0) {
      launch_act_scales(m->params, m->adesc, m->ascale, 18, ps);
      launch_tile_planes(m->params, m->wh, m->wdesc, m->ndesc, m->nwork, 0, ps);
    }
    HIP_OK(hipEventRecord(m->ev_prep_fwd, ps));
    m->prep_fwd = true;
    if (training) {
      launch_transpose_mats(m->params, m->paramsT, m->tdesc, m->tn, m->ttotal, ps);
      if (m->f16_split > 0) launch_tile_planes(m->paramsT, m->whT, m->wdescT, m->ndescT, m->nworkT, 1, ps);
      HIP_OK(hipEventRecord(m->ev_prep_bwd, ps));
      m->prep_bwd = true;
    }
  }
  if (training) {
    if (!m->sums_clean) HIP_OK(hipMemsetAsync(m->bn_sums, 0, 64 * sizeof(double), s));
    m->sums_clean = false;
    launch_conv1_fwd(m->cfg.leads, 0, x, m->params + Y.conv1_w, m->params + Y.conv1_b, m->a0, m->bn_sums, nullptr,
                     nullptr, nullptr, nullptr, m->L, m->Lp, B, s);
  } else {
    launch_conv1_fwd(m->cfg.leads, 1, x, m->params + Y.conv1_w, m->params + Y.conv1_b, m->x0, nullptr,
                     m->params + Y.bn_w, m->params + Y.bn_b, m->state, m->state + 8, m->L, m->Lp, B, s);
  }
  return 0;
}

static int fwd_end(RalModel* m, float* y, int B, int64_t global_windows, int training, hipStream_t s) {
  sched_reset();
  const Layout& Y = m->lay;
  if (training) {
    launch_bn_train8(m->bn_sums, (double)global_windows * m->L, m->params + Y.bn_w, m->params + Y.bn_b, m->ss,
                     m->state, m->state + 8, m->a0, m->x0, (size_t)B * m->Lp, s);
  }
  const bool tr = training != 0;
  // (the activation scales and the split planes of the wide levels' weights: formed under the stem on an idle stream.  ONE join, here,
  // in front of the fork: waiting for the planes only in front of the first wide level - an edge from the preparation stream into each
  // lane - gained nothing measurable on the eager step, and the hipGraph replay of the forward measured 457 k windows/s against 541 k
  // eager with it in the round-6 collection (572 k / 563 k without it))
  if (m->prep_fwd) { HIP_OK(hipStreamWaitEvent(s, m->ev_prep_fwd, 0)); m->prep_fwd = false; }
  else if (m->f16_split > 0 && !m->skip_prep) {
    launch_act_scales(m->params, m->adesc, m->ascale, 18, s);
    launch_tile_planes(m->params, m->wh, m->wdesc, m->ndesc, m->nwork, 0, s);
  }
  const int nl = plan_lanes(m, B, s);
  fork_lanes(m, s);
  LaneSet* LS = lanes_of(m);
  // issue stage by stage, alternating lanes, so that the lanes progress together
  const float* cur = m->x0;
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < nl; ++k) run_stage_fwd(m, i, cur, tr, LS->l[k]);
    for (int k = 0; k < nl; ++k) run_res_fwd(m, i, m->act[2 * i + 1].out, nullptr, LS->l[k]);
    cur = m->res_out[i];
  }
  for (int k = 0; k < nl; ++k) {
    const Lane& ln = LS->l[k];
    // x_mid = transformer(x4) + x4 (raletransformer.py:659): the second output of the bottleneck's last MLP kernel
    run_block_fwd(m, 8, cur, tr, ln);
    run_block_fwd(m, 9, m->act[8].out, tr, ln, m->res_out[3], m->xmid);
  }
  cur = m->xmid;
  for (int i = 0; i < 4; ++i) {
    for (int k = 0; k < nl; ++k) run_stage_fwd(m, 5 + i, cur, tr, LS->l[k]);
    for (int k = 0; k < nl; ++k) run_res_fwd(m, 4 + i, m->act[2 * (5 + i) + 1].out, i < 3 ? m->res_out[2 - i] : nullptr, LS->l[k]);
    cur = m->res_out[4 + i];
  }
  for (int k = 0; k < nl; ++k) {
    const Lane& ln = LS->l[k];
    launch_final_fwd(m->cfg.leads, woff(cur, ln.w0, m->E1), woff(m->x0, ln.w0, m->E1), m->params + Y.tc_w, m->params + Y.tc_b,
                     woff(y, ln.w0, (size_t)m->cfg.leads * m->L), m->L, m->Lp, ln.B, ln.s);
  }
  join_lanes(m, s);
  HIP_OK(hipGetLastError());
  return sched_check();
}

// ---------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------
// one block: dy (grad of block output) -> dx (grad of block input) [+ extra]; all pointers are window-0 bases
static void run_block_bwd(RalModel* m, int bi, const float* dy, const float* extra, float* dx, Lane& ln) {
  const int si = bi / 2, l = STAGES[si].level, C = CH[l], N = m->Lp >> l, H = C / 4;
  const size_t E1 = m->E1;
  BlockP w = block_ptrs(m->lay.blk[bi], m->params);
  w.asc = m->f16_split > 0 ? m->ascale + ASC_N * bi : nullptr;
  const BlockP g = block_ptrs(m->lay.blk[bi], m->grads);
  const BlockP wt = block_ptrs(m->lay.blk[bi], m->paramsT);
  BlockAct& a = m->act[bi];
  const float* table = nullptr;
  float* gtable = nullptr;
  int Len = 0;
  if (m->lay.rwave && STAGES[si].rw) {
    table = m->params + m->lay.rw[STAGES[si].rw - 1];
    gtable = m->grads + m->lay.rw[STAGES[si].rw - 1];
    Len = RWLEN[STAGES[si].rw - 1];
  }
  const int w0 = ln.w0, B = ln.B;
  hipStream_t s = ln.s;
  const int k = ln.bwd_count++ % dw_sets();            // temporary set of this block
  const bool side = m->side_stream && m->want_dw;
  hipStream_t sd = side ? ln.s2 : s;                   // stream of the weight-gradient kernels
  if (side && ln.dw_pending[k]) EV(hipStreamWaitEvent(s, ln.ev_done[k], 0));   // set k free again?
  const float* dyw = woff(dy, w0, E1);
  float *dupre = woff(m->dupre[k], w0, 4 * E1), *dx1 = woff(m->dx1[k], w0, E1), *dohm = woff(m->dohm[k], w0, E1),
        *dqkv = woff(m->dqkv[k], w0, 3 * E1), *a2c0 = woff(m->a2c0[k], w0, m->Lp);   // (per-window stride L at every level: the lanes run different levels concurrently)
  const float *x1 = woff(a.x1, w0, E1), *upre = woff(a.upre, w0, 4 * E1), *qkv = woff(a.qkv, w0, 3 * E1),
              *o = woff(a.o, w0, E1), *lse = woff(a.lse, w0, E1 / 4), *xin = woff(a.in, w0, E1);
  bool fused_mlp_dw;
  const bool splitb = m->f16_split > 0 && C >= m->f16_split;
  // maxima of this block's gradient tensors (this lane's windows), for the split weight-gradient products: only when both
  // split data-gradient kernels take the shape (they publish them)
  unsigned* gmax = (splitb && m->want_dw && !mlp_bwd_is_fused(C, N) && mlp_bwd_h_nch(C, N) && qkv_bwd_uses_f16(C, N)) ? m->gmax + ((size_t)bi * 4 + (&ln - lanes_of(m)->l)) * 4 : nullptr;
  const int NE = m->L >> l;
  { ProfScope p(m, K_MLP_BWD, s); fused_mlp_dw = launch_mlp_bwd(C, m->nch_b[l], dyw, x1, upre, w, wt, m->paramsT, splitb ? m->whT : nullptr, gmax, g, dupre, dx1, dohm, a2c0, N, B, m->want_dw, s,
                                                                  (m->f16_split > 0 && m->narrow_f16) ? 1 : 0, NE); }
  // the R-wave table gradient leaves the one-sweep attention kernels as one row of partials per workgroup; the kernel that adds
  // the rows up runs with the block's weight-gradient kernels (side stream), not on the chain: nothing on the chain reads it
  AttnTabReduce tabred{nullptr, nullptr, 0, 0};
  static const bool tab_side = ral_knob("TABRED_SIDE", 1) != 0;
  { ProfScope p(m, K_ATTN_BWD, s);
    if (side && tab_side) attn_tab_defer_to(&tabred);
    // (each lane's scratch: its share of the stat2 region followed by its share of the partials region)
    float* scratch = m->astat[k] + (size_t)w0 * (E1 / 2 + 2048);
    launch_attn_bwd(qkv, o, dohm, lse, table, gtable, dqkv, scratch, (size_t)B * (E1 / 2 + 2048), N, H, m->hg_b[l], Len, B,
                    (m->f16_split > 0 && m->attn_f16) ? 1 : 0, s, NE);
    attn_tab_defer_to(nullptr); }
  bool fused_qkv_dw;
  { ProfScope p(m, K_QKV_BWD, s);
    fused_qkv_dw = launch_qkv_bwd(C, dqkv, xin, m->pe[l], dx1, woff(extra, w0, E1), w, wt, m->paramsT, splitb ? m->whT : nullptr, gmax, g, woff(dx, w0, E1), N, B, m->want_dw, s); }
  if (!m->want_dw) return;
  if (side) {
    EV(hipEventRecord(ln.ev_ready[k], s));
    EV(hipStreamWaitEvent(sd, ln.ev_ready[k], 0));
  }
  if (tabred.ntab > 0) launch_attn_tpart_reduce(tabred.tpart, tabred.gtable, tabred.ntab, tabred.nrow, sd);
  { ProfScope p(m, K_DW, sd);
#ifdef RAL_DIAG   // diagnostic builds only (make VARIANT=diag EXTRA=-DRAL_DIAG; WRONG gradients): what the weight-gradient kernels of the levels C >= RAL_DIAG_SKIP_WIDE_DW cost the step
    static const int skipw = (int)ral_knob("DIAG_SKIP_WIDE_DW", 0);
    if (!(skipw && C >= skipw))
#endif
    launch_block_dw(C, dyw, upre, w.le ? a2c0 : nullptr, dupre, x1, dx1, o, dqkv, xin, m->pe[l], w, g, N, B, m->dw_ksplit[l], fused_mlp_dw, gmax, sd, fused_qkv_dw); }
  if (side) { EV(hipEventRecord(ln.ev_done[k], sd)); ln.dw_pending[k] = true; }
}

// stage: grad of stage output `dy` -> grad of stage input written to `dx` (+extra). Uses `tmp` between blocks.
static void run_stage_bwd(RalModel* m, int si, const float* dy, const float* extra, float* tmp, float* dx, Lane& ln) {
  run_block_bwd(m, si * 2 + 1, dy, nullptr, tmp, ln);
  run_block_bwd(m, si * 2, tmp, extra, dx, ln);
}

static void run_res_bwd(RalModel* m, int ri, const float* dy, const float* in, float* dx, Lane& ln) {
  const ResOff& r = m->lay.res[ri];
  const int T = m->E1 / r.D, Tv = 8 * m->L / r.D;
  const size_t E1 = m->E1;
  hipStream_t s = ln.s;
  { ProfScope p(m, K_RES_BWD, s);
    launch_resample_bwd(r.D, ri >= 4, woff(dy, ln.w0, E1), woff(in, ln.w0, E1), m->paramsT + r.w, m->params + r.lnw,
                        m->grads + r.lnw, m->grads + r.lnb, woff(dx, ln.w0, E1), T, Tv, ln.B, s); }
  if (!m->want_dw) return;
  int lvl = 0;
  while ((8 << lvl) < r.D) ++lvl;
  hipStream_t sd = s;
  if (m->side_stream) {   // dy was produced on s: fork the weight-gradient product to the side stream
    EV(hipEventRecord(ln.ev_fork, s));
    EV(hipStreamWaitEvent(ln.s2, ln.ev_fork, 0));
    sd = ln.s2;
  }
  launch_resample_dw(r.D, ri >= 4, woff(dy, ln.w0, E1), woff(in, ln.w0, E1), m->params + r.lnw, m->params + r.lnb,
                     m->grads + r.w, T, Tv, ln.B, m->dw_ksplit[lvl], sd);
}

static int bwd_begin(RalModel* m, const float* dy, int B, hipStream_t s) {
  sched_reset();
  if (!m->cfg.train) return fail("handle was created with train=0");
  if (!m->grads || !m->bn_sums) return fail("ral_bind: grads / bn_sums not bound");
  if (B != m->last_B) return fail("backward batch %d != forward batch %d", B, m->last_B);
  const Layout& Y = m->lay;
  launch_zero_bwd(m->grads, (size_t)Y.nparam, m->bn_sums + 32, 32, m->gmax, 18 * 4 * 4, s);   // gradient buffer, backward BatchNorm sums, gradient maxima
  const bool prep_ok = m->prep_bwd && !m->prep_stale;
  if (m->prep_bwd) { HIP_OK(hipStreamWaitEvent(s, m->ev_prep_bwd, 0)); m->prep_bwd = false; }   // transposes + their planes: formed during the forward
  if (!prep_ok) {   // (... or formed from parameters / for options that have changed since: after them, again, from what is bound now)
    launch_transpose_mats(m->params, m->paramsT, m->tdesc, m->tn, m->ttotal, s);
    if (m->f16_split > 0) {
      launch_tile_planes(m->paramsT, m->whT, m->wdescT, m->ndescT, m->nworkT, 1, s);
      if (m->prep_stale) launch_act_scales(m->params, m->adesc, m->ascale, 18, s);   // (the weight-gradient products scale their activation operands)
    }
  }
  float** gy = m->gy; float** gin = m->gin;
  const int nl = plan_lanes(m, B, s);
  LaneSet* LS = lanes_of(m);
  fork_lanes(m, s);
  // output conv: dy -> d(u0 + x0), each lane its windows
  for (int k = 0; k < nl; ++k) {
    const Lane& ln = LS->l[k];
    launch_final_bwd(m->cfg.leads, woff(dy, ln.w0, (size_t)m->cfg.leads * m->L), woff(m->res_out[7], ln.w0, m->E1), woff(m->x0, ln.w0, m->E1),
                     m->params + Y.tc_w, m->grads + Y.tc_w, m->grads + Y.tc_b, woff(m->du0, ln.w0, m->E1), m->L, m->Lp, ln.B, ln.s);
  }
  const bool side = m->side_stream && m->want_dw;
  if (side)   // side streams start after the gradient buffer has been zeroed
    for (int k = 0; k < nl; ++k) {
      EV(hipEventRecord(LS->l[k].ev_fork, LS->l[k].s));
      EV(hipStreamWaitEvent(LS->l[k].s2, LS->l[k].ev_fork, 0));
    }
#define EACH_LANE(stmt) for (int k_ = 0; k_ < nl; ++k_) { Lane& ln = LS->l[k_]; stmt; }
  // decoder: ps_k <- stage <- (u = ps(.) + p)
  EACH_LANE(run_res_bwd(m, 7, m->du0, m->act[17].out, gy[17], ln))
  EACH_LANE(run_stage_bwd(m, 8, gy[17], nullptr, gy[16], gin[8], ln))          // g u1
  EACH_LANE(run_res_bwd(m, 6, gin[8], m->act[15].out, gy[15], ln))
  EACH_LANE(run_stage_bwd(m, 7, gy[15], nullptr, gy[14], gin[7], ln))          // g u2
  EACH_LANE(run_res_bwd(m, 5, gin[7], m->act[13].out, gy[13], ln))
  EACH_LANE(run_stage_bwd(m, 6, gy[13], nullptr, gy[12], gin[6], ln))          // g u3
  EACH_LANE(run_res_bwd(m, 4, gin[6], m->act[11].out, gy[11], ln))
  EACH_LANE(run_stage_bwd(m, 5, gy[11], nullptr, gy[10], gin[5], ln))          // g x_mid
  // the decoder's parameters (utransformer4 .. transconv: the upper half of the flat gradient buffer) have their
  // final gradients once every lane's chain and weight-gradient stream pass this point: gradient bucket 1
  EACH_LANE(EV(hipEventRecord(ln.ev_dec_main, ln.s)); if (side) EV(hipEventRecord(ln.ev_dec_side, ln.s2));)
  m->dec_lanes = nl; m->dec_side = side;
  EACH_LANE(run_stage_bwd(m, 4, gin[5], gin[5], gy[8], gin[4], ln))            // g p4 = transformer^T(g x_mid) + g x_mid
  // encoder: pm_k <- stage, skip gradients added by the first block of each stage
  EACH_LANE(run_res_bwd(m, 3, gin[4], m->act[7].out, gy[7], ln))
  EACH_LANE(run_stage_bwd(m, 3, gy[7], gin[6], gy[6], gin[3], ln))             // g p3 (+ g u3)
  EACH_LANE(run_res_bwd(m, 2, gin[3], m->act[5].out, gy[5], ln))
  EACH_LANE(run_stage_bwd(m, 2, gy[5], gin[7], gy[4], gin[2], ln))             // g p2 (+ g u2)
  EACH_LANE(run_res_bwd(m, 1, gin[2], m->act[3].out, gy[3], ln))
  EACH_LANE(run_stage_bwd(m, 1, gy[3], gin[8], gy[2], gin[1], ln))             // g p1 (+ g u1)
  EACH_LANE(run_res_bwd(m, 0, gin[1], m->act[1].out, gy[1], ln))
  EACH_LANE(run_stage_bwd(m, 0, gy[1], m->du0, gy[0], gin[0], ln))             // g x0 (+ d u0)
#undef EACH_LANE
  if (side)   // join the side streams into their lanes, then the lanes into s
    for (int k = 0; k < nl; ++k) {
      EV(hipEventRecord(LS->l[k].ev_join, LS->l[k].s2));
      EV(hipStreamWaitEvent(LS->l[k].s, LS->l[k].ev_join, 0));
    }
  join_lanes(m, s);
  launch_bn8_bwd_stats(gin[0], m->a0, m->ss, m->bn_sums + 32, (size_t)B * m->Lp, s);
  HIP_OK(hipGetLastError());
  return sched_check();
}

static int bwd_end(RalModel* m, float* dx, int B, int64_t global_windows, hipStream_t s) {
  sched_reset();
  const Layout& Y = m->lay;
  launch_conv1_bwd(m->cfg.leads, m->gin[0], m->a0, m->last_x, m->ss, m->params + Y.bn_w, m->bn_sums + 32,
                   (double)global_windows * m->L, m->grads + Y.conv1_w, m->grads + Y.conv1_b, dx ? m->dz0 : nullptr,
                   m->L, m->Lp, B, s, m->grads + Y.bn_w, m->grads + Y.bn_b, (double)B / (double)global_windows);
  if (dx) launch_conv1_bwd_dx(m->cfg.leads, m->dz0, m->params + Y.conv1_w, dx, m->L, m->Lp, B, s);
  HIP_OK(hipEventRecord(m->ev_bwd_done, s));
  m->bwd_recorded = true;
  HIP_OK(hipGetLastError());
  return sched_check();
}

// ---------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------
static LaneSet* lanes_of(RalModel* m) { return reinterpret_cast<LaneSet*>(m->lanes); }

extern "C" {

int ral_layout_count(const ral_config* cfg) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET) return unet_layout_count(cfg);
  if (cfg->variant == RAL_ACDAE) return acdae_layout_count(cfg);
  if (cfg->variant == RAL_DANET) return danet_layout_count(cfg);
  Layout L;
  build_layout(*cfg, L);
  return (int)L.entries.size();
}

int ral_layout_entry(const ral_config* cfg, int idx, char* name, int name_cap, int32_t* kind, int64_t* offset,
                     int32_t* ndim, int64_t shape[4]) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET) return unet_layout_entry(cfg, idx, name, name_cap, kind, offset, ndim, shape);
  if (cfg->variant == RAL_ACDAE) {
    if (acdae_layout_entry(cfg, idx, name, name_cap, kind, offset, ndim, shape)) return fail("entry %d out of range", idx);
    return 0;
  }
  if (cfg->variant == RAL_DANET) {
    if (danet_layout_entry(cfg, idx, name, name_cap, kind, offset, ndim, shape)) return fail("entry %d out of range", idx);
    return 0;
  }
  Layout L;
  build_layout(*cfg, L);
  if (idx < 0 || idx >= (int)L.entries.size()) return fail("entry %d out of range", idx);
  const Entry& e = L.entries[idx];
  if ((int)e.name.size() + 1 > name_cap) return fail("name buffer too small");
  strcpy(name, e.name.c_str());
  *kind = e.kind; *offset = e.offset; *ndim = e.ndim;
  for (int i = 0; i < 4; ++i) shape[i] = e.shape[i];
  return 0;
}

int64_t ral_param_floats(const ral_config* cfg) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET) return unet_param_floats(cfg);
  if (cfg->variant == RAL_ACDAE) return acdae_param_floats(cfg);
  if (cfg->variant == RAL_DANET) return danet_param_floats(cfg);
  Layout L;
  build_layout(*cfg, L);
  return L.nparam;
}

int64_t ral_state_floats(const ral_config* cfg) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET) return unet_state_floats(cfg);
  if (cfg->variant == RAL_ACDAE) return 0;
  if (cfg->variant == RAL_DANET) return danet_state_floats(cfg);
  return 16;
}

int64_t ral_bn_sums_doubles(const ral_config* cfg) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_DANET) return 0;     // (its batch sums live in its own workspace)
  return cfg->variant == RAL_UNET ? 1280 : (cfg->variant == RAL_ACDAE ? 0 : 64);
}

int64_t ral_workspace_bytes(const ral_config* cfg) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET) return unet_workspace_bytes(cfg);
  if (cfg->variant == RAL_ACDAE) return acdae_workspace_bytes(cfg);
  if (cfg->variant == RAL_DANET) return danet_workspace_bytes(cfg);
  return (int64_t)plan_workspace(*cfg, nullptr, nullptr);
}

int ral_pe_table(const ral_config* cfg, int level, float* out_host, int64_t cap) {
  if (check_cfg(cfg)) return -1;
  if (cfg->variant == RAL_UNET || level < 0 || level > 4) return fail("no PE table for this variant/level");
  const int n = cfg->L >> level, C = CH[level];
  if (cap < (int64_t)n * C) return fail("buffer too small");
  fill_pe(out_host, n, C);
  return 0;
}

// every stream, event and allocation of a model is released here: ral_destroy and the error paths of ral_create share it
static void destroy_model(RalModel* m) {
  if (!m) return;
  if (m->lanes) {
    LaneSet* LS = reinterpret_cast<LaneSet*>(m->lanes);
    for (int i = 0; i < MAX_LANES; ++i) {
      Lane& ln = LS->l[i];
      if (LS->own[i]) (void)hipStreamDestroy(LS->own[i]);
      if (ln.s2) (void)hipStreamDestroy(ln.s2);
      hipEvent_t evs[4] = {ln.ev_fork, ln.ev_join, ln.ev_dec_main, ln.ev_dec_side};
      for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
      for (int k = 0; k < MAX_SETS; ++k) {
        if (ln.ev_ready[k]) (void)hipEventDestroy(ln.ev_ready[k]);
        if (ln.ev_done[k]) (void)hipEventDestroy(ln.ev_done[k]);
      }
    }
    delete LS;
  }
  for (auto& pr : m->prof_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  if (m->ev_bwd_done) (void)hipEventDestroy(m->ev_bwd_done);
  for (hipEvent_t e : {m->ev_prep_go, m->ev_prep_fwd, m->ev_prep_bwd}) if (e) (void)hipEventDestroy(e);
  if (m->slab) (void)hipFree(m->slab);
  delete m;
}

int ral_create(const ral_config* cfg, ral_handle** out) {
  if (check_cfg(cfg)) return -1;
  if (!out) return fail("null out");
  ral_handle* h = new ral_handle();
  if (cfg->variant == RAL_ACDAE) {
    h->kind = 2;
    h->a = acdae_create(cfg, g_err, sizeof(g_err));
    if (!h->a) { delete h; return -1; }
    *out = h;
    return 0;
  }
  if (cfg->variant == RAL_DANET) {
    h->kind = 3;
    h->d = danet_create(cfg, g_err, sizeof(g_err));
    if (!h->d) { delete h; return -1; }
    *out = h;
    return 0;
  }
  if (cfg->variant == RAL_UNET) {
    h->kind = 1;
    h->u = unet_create(cfg, g_err, sizeof(g_err));
    if (!h->u) { delete h; return -1; }
    *out = h;
    return 0;
  }
  RalModel* m = new RalModel();
  m->cfg = *cfg;
  m->L = cfg->L;
  m->Lp = (cfg->L + 255) / 256 * 256;
  m->E1 = 8 * m->Lp;
  build_layout(*cfg, m->lay);
  m->slab_bytes = plan_workspace(*cfg, nullptr, nullptr);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->slab), m->slab_bytes);
  if (e != hipSuccess) {
    fail("hipMalloc(%zu bytes) failed: %s", m->slab_bytes, hipGetErrorString(e));
    m->slab = nullptr;
    destroy_model(m); delete h;
    return -1;
  }
  plan_workspace(*cfg, m, m->slab);
  choose_tiling(m);
  {
    LaneSet* LS = new LaneSet();
    m->lanes = LS;
    m->n_lanes = ral_env_int("RAL_LANES", 2, 1, MAX_LANES);        // one of the TWO environment variables the library reads
    // the weight-gradient streams are off the critical path: lowest priority, so that their workgroups fill the
    // gaps the main chain leaves instead of competing with it (RAL_DW_PRIO=0 keeps the default priority)
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    const int pmode = (int)env_size("RAL_DW_PRIO", 0);   // 0 default, 1 lowest, 2 highest
    const bool low = pmode != 0;
    // a failed creation would leave a null stream (= the legacy default stream: the fork/join ordering would silently
    // change), so the first error fails the whole ral_create
    hipError_t first = hipSuccess;
    auto ok = [&](hipError_t r) { if (r != hipSuccess && first == hipSuccess) first = r; };
    for (int i = 0; i < MAX_LANES; ++i) {
      Lane& ln = LS->l[i];
      if (i > 0) ok(hipStreamCreateWithFlags(&LS->own[i], hipStreamNonBlocking));
      if (low) ok(hipStreamCreateWithPriority(&ln.s2, hipStreamNonBlocking, pmode == 2 ? prio_greatest : prio_least));
      else ok(hipStreamCreateWithFlags(&ln.s2, hipStreamNonBlocking));
      for (int k = 0; k < MAX_SETS; ++k) {
        ok(hipEventCreateWithFlags(&ln.ev_ready[k], hipEventDisableTiming));
        ok(hipEventCreateWithFlags(&ln.ev_done[k], hipEventDisableTiming));
      }
      ok(hipEventCreateWithFlags(&ln.ev_fork, hipEventDisableTiming));
      ok(hipEventCreateWithFlags(&ln.ev_join, hipEventDisableTiming));
      ok(hipEventCreateWithFlags(&ln.ev_dec_main, hipEventDisableTiming));
      ok(hipEventCreateWithFlags(&ln.ev_dec_side, hipEventDisableTiming));
    }
    ok(hipEventCreateWithFlags(&m->ev_bwd_done, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&m->ev_prep_go, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&m->ev_prep_fwd, hipEventDisableTiming));
    ok(hipEventCreateWithFlags(&m->ev_prep_bwd, hipEventDisableTiming));
    if (first != hipSuccess) {
      fail("stream / event creation failed: %s", hipGetErrorString(first));
      destroy_model(m); delete h;
      return -1;
    }
  }
  for (int l = 0; l < 5; ++l) {
    const int n = m->Lp >> l, C = CH[l];
    std::vector<float> P((size_t)n * C);
    fill_pe(P.data(), n, C);
    e = hipMemcpy(m->pe[l], P.data(), P.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      fail("hipMemcpy(pe) failed: %s", hipGetErrorString(e));
      destroy_model(m); delete h;
      return -1;
    }
  }
  {   // descriptors of the activation-scale kernel
    std::vector<int> d;
    for (int b = 0; b < 18; ++b) {
      const BlockOff& o = m->lay.blk[b];
      const int v[12] = {CH[STAGES[b / 2].level], (int)o.ln1w, (int)o.ln1b, (int)o.wqkv, (int)o.bqkv, (int)o.ln2w, (int)o.ln2b, (int)o.w1, (int)o.b1,
                         (int)o.le, 0, 0};
      d.insert(d.end(), v, v + 12);
    }
    e = hipMemcpy(m->adesc, d.data(), d.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("hipMemcpy(adesc) failed: %s", hipGetErrorString(e)); destroy_model(m); delete h; return -1; }
  }
  m->f16_split = (int)ral_knob("F16_SPLIT", m->f16_split);   // (a process-wide default for tests: ral_global_option; never the environment)
  if (m->L != m->Lp) m->f16_split = 0;   // padded windows run on the generic fp32-MFMA kernels (the ones that know where a window ends)
  m->attn_f16 = attn_f16_default() != 0;
  if (cfg->train) {
    m->side_stream = ral_env_int("RAL_NO_SIDE_STREAM", 0, 0, 1) == 0;   // ... and the other
    std::vector<int> d;
    int run = 0;
    auto add = [&](int64_t off, int rows, int cols) { d.push_back((int)off); d.push_back(rows); d.push_back(cols); d.push_back(run); run += rows * cols; };
    for (int b = 0; b < 18; ++b) {
      const int C = CH[STAGES[b / 2].level];
      const BlockOff& o = m->lay.blk[b];
      add(o.wqkv, 3 * C, C); add(o.wp, C, C); add(o.w1, 4 * C, C); add(o.w2, C, 4 * C);
    }
    for (int r = 0; r < 8; ++r) add(m->lay.res[r].w, m->lay.res[r].D, m->lay.res[r].D);
    m->tn = (int)d.size() / 4; m->ttotal = run;
    e = hipMemcpy(m->tdesc, d.data(), d.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("hipMemcpy(tdesc) failed: %s", hipGetErrorString(e)); destroy_model(m); delete h; return -1; }
  }
  {   // weight matrices that are re-written as tiled split planes every forward (widths with a split-operand kernel)
    std::vector<int> d;
    int run = 0;
    auto add = [&](int64_t off, int rows, int cols) { d.push_back((int)off); d.push_back(rows); d.push_back(cols); d.push_back(run); run += rows * cols / 8; };
    for (int b = 0; b < 18; ++b) {
      const int C = CH[STAGES[b / 2].level];
      if (!qkv_fwd_uses_f16(C)) continue;
      const BlockOff& o = m->lay.blk[b];
      add(o.wqkv, 3 * C, C); add(o.wp, C, C); add(o.w1, 4 * C, C); add(o.w2, C, 4 * C);
    }
    m->ndesc = (int)d.size() / 4; m->nwork = run;
    if (cfg->train) {   // transposed matrices of the backward's data-gradient products (rows x columns as they sit in paramsT)
      std::vector<int> dt;
      int runT = 0;
      auto addT = [&](int64_t off, int rows, int cols) { dt.push_back((int)off); dt.push_back(rows); dt.push_back(cols); dt.push_back(runT); runT += rows * cols / 8; };
      for (int b = 0; b < 18; ++b) {
        const int C = CH[STAGES[b / 2].level];
        if (!qkv_fwd_uses_f16(C)) continue;
        const BlockOff& o = m->lay.blk[b];
        addT(o.wqkv, C, 3 * C); addT(o.wp, C, C); addT(o.w1, C, 4 * C); addT(o.w2, 4 * C, C);
      }
      m->ndescT = (int)dt.size() / 4; m->nworkT = runT;
      if (m->ndescT) {
        e = hipMemcpy(m->wdescT, dt.data(), dt.size() * sizeof(int), hipMemcpyHostToDevice);
        if (e != hipSuccess) { fail("hipMemcpy(wdescT) failed: %s", hipGetErrorString(e)); destroy_model(m); delete h; return -1; }
      }
    }
    if (m->ndesc > 64) { fail("too many split-plane descriptors"); destroy_model(m); delete h; return -1; }
    if (m->ndesc) {
      e = hipMemcpy(m->wdesc, d.data(), d.size() * sizeof(int), hipMemcpyHostToDevice);
      if (e != hipSuccess) { fail("hipMemcpy(wdesc) failed: %s", hipGetErrorString(e)); destroy_model(m); delete h; return -1; }
    }
  }
  h->m = m;
  *out = h;
  return 0;
}

int ral_destroy(ral_handle* h) {
  if (!h) return 0;
  destroy_model(h->m);
  if (h->u) unet_destroy(h->u);
  if (h->a) acdae_destroy(h->a);
  if (h->d) danet_destroy(h->d);
  delete h;
  return 0;
}

int ral_bind(ral_handle* h, float* params, float* grads, float* adam_m, float* adam_v, float* state, double* bn_sums) {
  if (!h) return fail("null handle");
  if (h->kind == 1) return unet_bind(h->u, params, grads, adam_m, adam_v, state, bn_sums);
  if (h->kind == 2) return acdae_bind(h->a, params, grads, adam_m, adam_v);
  if (h->kind == 3) return danet_bind(h->d, params, grads, adam_m, adam_v, state);
  RalModel* m = h->m;
  m->prep_stale = true;
  m->sums_clean = false;
  m->planes_valid = false;
  m->params = params; m->grads = grads; m->am = adam_m; m->av = adam_v; m->state = state; m->bn_sums = bn_sums;
  return 0;
}

int ral_forward_begin(ral_handle* h, const float* x, int B, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind == 2) return fail("ACDAE has no BatchNorm: use ral_forward");
  if (h->kind == 1 || h->kind == 3) return fail("U-Net / DANet have a BatchNorm per layer: use ral_forward (per-rank statistics)");
  return fwd_begin(h->m, x, B, 1, (hipStream_t)s);
}

int ral_forward_end(ral_handle* h, float* y, int B, int64_t global_windows, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind == 2) return fail("ACDAE has no BatchNorm: use ral_forward");
  if (h->kind == 1 || h->kind == 3) return fail("U-Net / DANet have a BatchNorm per layer: use ral_forward (per-rank statistics)");
  return fwd_end(h->m, y, B, global_windows, 1, (hipStream_t)s);
}

int ral_forward(ral_handle* h, const float* x, float* y, int B, int training, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind == 1) return unet_forward(h->u, x, y, B, training, (hipStream_t)s, g_err, sizeof(g_err));
  if (h->kind == 2) return acdae_forward(h->a, x, y, B, (hipStream_t)s, g_err, sizeof(g_err));
  if (h->kind == 3) return danet_forward(h->d, x, y, B, training, (hipStream_t)s, g_err, sizeof(g_err));
  if (fwd_begin(h->m, x, B, training, (hipStream_t)s)) return -1;
  return fwd_end(h->m, y, B, B, training, (hipStream_t)s);
}

int ral_loss(ral_handle* h, const float* pred, const float* target, int B, int64_t global_windows, float* dy,
             float* snr, float* rmse, double* loss_sum, ral_stream s) {
  if (!h) return fail("null handle");
  const ral_config& c = h->kind == 1 ? unet_public(h->u)->cfg
                        : (h->kind == 2 ? acdae_public(h->a)->cfg : (h->kind == 3 ? danet_public(h->d)->cfg : h->m->cfg));
  const int n = c.leads * c.L;
  const float gscale = (float)(2.0 / ((double)global_windows * n));
  launch_loss(pred, target, dy, snr, rmse, loss_sum, n, B, gscale, (hipStream_t)s);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_loss_mean(const float* pred, const float* target, int n, int B, int64_t global_windows, float* dy, float* snr,
                  float* rmse, double* loss_mean, double* scratch2, ral_stream s) {
  if (!pred || !target || !loss_mean || !scratch2) return fail("ral_loss_mean: null pointer");
  if (n <= 0 || B <= 0 || global_windows <= 0) return fail("ral_loss_mean: n, B and global_windows must be positive");
  launch_loss(pred, target, dy, snr, rmse, scratch2, n, B, (float)(2.0 / ((double)global_windows * n)), (hipStream_t)s, loss_mean,
              1.0 / (double)global_windows);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_loss_means(const float* pred, const float* target, int n, int B, int64_t global_windows, float* dy, float* snr,
                   float* rmse, double* means3, double* scratch64, ral_stream s) {
  if (!pred || !target || !means3 || !scratch64) return fail("ral_loss_means: null pointer");
  if (n <= 0 || B <= 0 || global_windows <= 0) return fail("ral_loss_means: n, B and global_windows must be positive");
  launch_loss(pred, target, dy, snr, rmse, scratch64, n, B, (float)(2.0 / ((double)global_windows * n)), (hipStream_t)s, means3,
              1.0 / (double)global_windows, 1);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_forward_loss_means(ral_handle* h, const float* x, const float* target, float* y, int B, float* dy, float* snr,
                           float* rmse, double* means3, double* scratch64, ral_stream s) {
  if (!h) return fail("null handle");
  if (!x || !target || !y || !means3 || !scratch64) return fail("ral_forward_loss_means: null pointer");
  if (h->kind == 1 && dy && unet_public(h->u)->cfg.train)      // U-Net: the output BatchNorm, the loss and the backward's first sums in one kernel
    return unet_forward_loss(h->u, x, target, y, B, dy, snr, rmse, scratch64, means3, 1.0 / (double)B, 1, (hipStream_t)s, g_err, sizeof(g_err));
  if (ral_forward(h, x, y, B, 1, s)) return -1;
  const ral_config& c = h->kind == 1 ? unet_public(h->u)->cfg
                        : (h->kind == 2 ? acdae_public(h->a)->cfg : (h->kind == 3 ? danet_public(h->d)->cfg : h->m->cfg));
  return ral_loss_means(y, target, c.leads * c.L, B, B, dy, snr, rmse, means3, scratch64, s);
}

int ral_loss_flat(const float* pred, const float* target, int n, int B, int64_t global_windows, float* dy, float* snr,
                  float* rmse, double* loss_sum, ral_stream s) {
  if (!pred || !target) return fail("ral_loss_flat: null pointer");
  if (n <= 0 || B <= 0 || global_windows <= 0) return fail("ral_loss_flat: n, B and global_windows must be positive");
  launch_loss(pred, target, dy, snr, rmse, loss_sum, n, B, (float)(2.0 / ((double)global_windows * n)), (hipStream_t)s);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_backward_begin(ral_handle* h, const float* dy, int B, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind != 0) return fail("U-Net / ACDAE: use ral_backward");
  return bwd_begin(h->m, dy, B, (hipStream_t)s);
}

int ral_backward_end(ral_handle* h, float* dx, int B, int64_t global_windows, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind != 0) return fail("U-Net / ACDAE: use ral_backward");
  return bwd_end(h->m, dx, B, global_windows, (hipStream_t)s);
}

// ---- U-Net, one stage at a time (sync-BatchNorm under data parallelism) ----
int ral_unet_stage_bn(int si) { return unet_stage_bn(si); }
#define UNET_ONLY if (!h) return fail("null handle"); if (h->kind != 1) return fail("U-Net handles only")
int ral_unet_forward_stage(ral_handle* h, const float* x, int B, int training, int si, int64_t global_windows, ral_stream s) {
  UNET_ONLY;
  return unet_forward_stage(h->u, x, B, training, si, global_windows, (hipStream_t)s, g_err, sizeof(g_err));
}
int ral_unet_forward_finish(ral_handle* h, float* y, int B, int training, int64_t global_windows, ral_stream s) {
  UNET_ONLY;
  return unet_forward_finish(h->u, y, B, training, global_windows, (hipStream_t)s, g_err, sizeof(g_err));
}
int ral_unet_backward_start(ral_handle* h, const float* dy, int B, int64_t global_windows, ral_stream s) {
  UNET_ONLY;
  return unet_backward_start(h->u, dy, B, global_windows, (hipStream_t)s, g_err, sizeof(g_err));
}
int ral_unet_backward_stage(ral_handle* h, int B, int si, int64_t global_windows, ral_stream s) {
  UNET_ONLY;
  return unet_backward_stage(h->u, B, si, global_windows, (hipStream_t)s, g_err, sizeof(g_err));
}
int ral_unet_backward_finish(ral_handle* h, int B, int64_t global_windows, ral_stream s) {
  UNET_ONLY;
  return unet_backward_finish(h->u, B, global_windows, (hipStream_t)s, g_err, sizeof(g_err));
}
#undef UNET_ONLY

int ral_grad_bucket(ral_handle* h, int k, int64_t* offset, int64_t* count) {
  if (!h || h->kind != 0) return fail("gradient buckets: RA-LENet handles only");
  if (k < 0 || k > 1 || !offset || !count) return fail("gradient bucket index must be 0 or 1");
  const Layout& Y = h->m->lay;
  *offset = k == 0 ? 0 : Y.dec_off;
  *count = k == 0 ? Y.dec_off : Y.nparam - Y.dec_off;
  return 0;
}

int ral_grad_bucket_wait(ral_handle* h, int k, ral_stream s) {
  if (!h || h->kind != 0) return fail("gradient buckets: RA-LENet handles only");
  RalModel* m = h->m;
  if (k == 1) {
    if (m->dec_lanes <= 0) return fail("bucket 1 is available after ral_backward_begin");
    LaneSet* LS = lanes_of(m);
#ifdef RAL_DIAG   // diagnostic builds only (dropping a wait is a data race): 1 chains only, 2 side streams only, 0 none
    static const int dbg = (int)ral_knob("BUCKET_WAIT", 3);
#else
    constexpr int dbg = 3;
#endif
    for (int i = 0; i < m->dec_lanes; ++i) {
      if (dbg & 1) HIP_OK(hipStreamWaitEvent((hipStream_t)s, LS->l[i].ev_dec_main, 0));
      if (m->dec_side && (dbg & 2)) HIP_OK(hipStreamWaitEvent((hipStream_t)s, LS->l[i].ev_dec_side, 0));
    }
    return 0;
  }
  if (k == 0) {
    if (!m->bwd_recorded) return fail("bucket 0 is available after ral_backward_end");
    HIP_OK(hipStreamWaitEvent((hipStream_t)s, m->ev_bwd_done, 0));
    return 0;
  }
  return fail("gradient bucket index must be 0 or 1");
}

int ral_backward(ral_handle* h, const float* dy, float* dx, int B, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind == 1) return unet_backward(h->u, dy, dx, B, (hipStream_t)s, g_err, sizeof(g_err));   // (dy = NULL: the gradient ral_forward_loss_means wrote)
  if (!dy) return fail("ral_backward: dy is NULL");
  if (h->kind == 2) return acdae_backward(h->a, dy, dx, B, (hipStream_t)s, g_err, sizeof(g_err));
  if (h->kind == 3) return danet_backward(h->d, dy, dx, B, (hipStream_t)s, g_err, sizeof(g_err));
  if (bwd_begin(h->m, dy, B, (hipStream_t)s)) return -1;
  return bwd_end(h->m, dx, B, B, (hipStream_t)s);
}

int ral_backward_input(ral_handle* h, const float* dy, float* dx, int B, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind != 0) return fail("ral_backward_input: RA-LENet handles only");
  if (!dx) return fail("ral_backward_input: dx is the only result, it cannot be NULL");
  RalModel* m = h->m;
  m->want_dw = false;
  int rc = bwd_begin(m, dy, B, (hipStream_t)s);
  if (!rc) rc = bwd_end(m, dx, B, B, (hipStream_t)s);
  m->want_dw = true;
  return rc;
}

int ral_backward_input_begin(ral_handle* h, const float* dy, int B, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind != 0) return fail("ral_backward_input_begin: RA-LENet handles only");
  RalModel* m = h->m;
  m->want_dw = false;
  const int rc = bwd_begin(m, dy, B, (hipStream_t)s);
  m->want_dw = true;
  return rc;
}

int ral_backward_input_end(ral_handle* h, float* dx, int B, int64_t global_windows, ral_stream s) {
  if (!h) return fail("null handle");
  if (h->kind != 0) return fail("ral_backward_input_end: RA-LENet handles only");
  if (!dx) return fail("ral_backward_input_end: dx is the only result, it cannot be NULL");
  return bwd_end(h->m, dx, B, global_windows, (hipStream_t)s);
}

int ral_adam_step(ral_handle* h, double lr, double beta1, double beta2, double eps, int step, float grad_scale,
                  ral_stream s) {
  if (!h) return fail("null handle");
  float *p, *g, *am, *av;
  int64_t n;
  if (h->kind == 1) { UNetPublic* u = unet_public(h->u); p = u->params; g = u->grads; am = u->am; av = u->av; n = u->nparam; }
  else if (h->kind == 2) { AcdaePublic* u = acdae_public(h->a); p = u->params; g = u->grads; am = u->am; av = u->av; n = u->nparam; }
  else if (h->kind == 3) { DanetPublic* u = danet_public(h->d); p = u->params; g = u->grads; am = u->am; av = u->av; n = u->nparam; }
  else { p = h->m->params; g = h->m->grads; am = h->m->am; av = h->m->av; n = h->m->lay.nparam; h->m->prep_stale = true; h->m->planes_valid = false; }
  if (!p || !g || !am || !av) return fail("ral_bind: params/grads/adam buffers not bound");
  if (step < 1) return fail("step is 1-based");
  double* zero64 = nullptr;
  if (h->kind == 0 && h->m->bn_sums) { zero64 = h->m->bn_sums; h->m->sums_clean = true; }   // (the next step's stem starts from cleared sums)
  launch_adam(p, g, am, av, (size_t)n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)s, zero64);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_set_option(ral_handle* h, const char* key, int value) {
  if (!h || !key) return fail("null handle or key");
  if (h->kind == 1) {
    if (unet_set_option(h->u, key, value)) return fail("unknown U-Net option %s", key);
    return 0;
  }
  if (h->kind != 0) return fail("no options for this handle");
  RalModel* m = h->m;
  m->prep_stale = true;   // (weight planes queued by a forward were formed for the options of that moment)
  m->planes_valid = false;
  if (!strcmp(key, "static_params")) { m->static_params = value != 0; return 0; }   // (setting it again = "the parameters have changed")
  if (!strcmp(key, "lanes")) { m->n_lanes = value < 1 ? 1 : (value > MAX_LANES ? MAX_LANES : value); return 0; }
  if (!strcmp(key, "side_stream")) { m->side_stream = value != 0; return 0; }
  if (!strcmp(key, "f16_split")) {
    if (value > 0 && m->L != m->Lp) return fail("f16_split > 0 needs a window length that is a multiple of 256 (L = %d runs on padded windows, fp32 MFMA)", m->L);
    m->f16_split = value; return 0;
  }
  if (!strcmp(key, "attn_f16")) { m->attn_f16 = value != 0; return 0; }
  if (!strcmp(key, "narrow_f16")) { m->narrow_f16 = value != 0; return 0; }
  return fail("unknown option %s", key);
}

int ral_profile_select(ral_handle* h, const char* kind) {
  if (!h || h->kind != 0) return fail("profiling is available for RA-LENet handles only");
  RalModel* m = h->m;
  m->prof_used = 0;
  m->prof_kind = -1;
  if (!kind || !*kind) return 0;
  if (!strcmp(kind, "*")) { m->prof_kind = K_ALL; return 0; }
  for (int k = 0; k < K_NKINDS; ++k)
    if (!strcmp(kind, KIND_NAMES[k])) { m->prof_kind = k; return 0; }
  return fail("unknown kernel kind %s", kind);
}

int ral_profile_read(ral_handle* h, double* total_ms, int64_t* launches) {
  if (!h || h->kind != 0) return fail("profiling is available for RA-LENet handles only");
  RalModel* m = h->m;
  double tot = 0.0;
  for (size_t i = 0; i < m->prof_used; ++i) {
    float ms = 0.f;
    HIP_OK(hipEventSynchronize(m->prof_ev[i].second));
    HIP_OK(hipEventElapsedTime(&ms, m->prof_ev[i].first, m->prof_ev[i].second));
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)m->prof_used;
  m->prof_used = 0;
  return 0;
}

int ral_profile_timeline(ral_handle* h, double* rows, int64_t cap_rows, int64_t* nrows) {
  if (!h || h->kind != 0) return fail("profiling is available for RA-LENet handles only");
  RalModel* m = h->m;
  *nrows = (int64_t)m->prof_used;
  if ((int64_t)m->prof_used > cap_rows) return fail("timeline: %zu rows, room for %lld", m->prof_used, (long long)cap_rows);
  std::vector<hipStream_t> streams;
  for (size_t i = 0; i < m->prof_used; ++i) {
    HIP_OK(hipEventSynchronize(m->prof_ev[i].second));
    float t0 = 0.f, t1 = 0.f;
    HIP_OK(hipEventElapsedTime(&t0, m->prof_ev[0].first, m->prof_ev[i].first));
    HIP_OK(hipEventElapsedTime(&t1, m->prof_ev[0].first, m->prof_ev[i].second));
    size_t si = 0;
    while (si < streams.size() && streams[si] != m->prof_meta[i].second) ++si;
    if (si == streams.size()) streams.push_back(m->prof_meta[i].second);
    rows[4 * i] = m->prof_meta[i].first; rows[4 * i + 1] = (double)si; rows[4 * i + 2] = t0; rows[4 * i + 3] = t1;
  }
  m->prof_used = 0;
  return 0;
}

int ral_conv13_forward(const float* x, const float* w, const float* b, float* y, int B, int cin, int cout, int L,
                       int lrelu, ral_stream s) {
  if (launch_conv13_fwd(x, w, b, y, B, cin, cout, L, lrelu, (hipStream_t)s)) return fail("conv13: unsupported channel counts %d -> %d", cin, cout);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_conv13_backward(const float* x, const float* y, const float* dy, const float* w, float* gw, float* gb,
                        float* dx, int B, int cin, int cout, int L, int lrelu, ral_stream s) {
  if (launch_conv13_bwd(x, y, dy, w, gw, gb, dx, B, cin, cout, L, lrelu, (hipStream_t)s)) return fail("conv13: unsupported channel counts %d -> %d", cin, cout);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int step, float grad_scale, ral_stream s) {
  if (n % 4) return fail("flat Adam needs a multiple of 4 floats");
  launch_adam(p, g, m, v, (size_t)n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)s);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_prep_windows(const float* sig, const float* noise, int64_t T, int leads, int L, double snr_db, double* sums,
                     float* noisy, float* clean, ral_stream s) {
  if (!sig || !noise || !sums || !noisy || !clean) return fail("prep_windows: null pointer");
  if (launch_prep_windows(sig, noise, (long long)T, leads, L, snr_db, sums, noisy, clean, (hipStream_t)s))
    return fail("prep_windows: need 1 <= leads <= 16 and T a positive multiple of L (T=%lld leads=%d L=%d)", (long long)T, leads, L);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_stream_windows(const float* rec, int64_t R, int64_t T, int leads, int L, int hop, int64_t w0, int nw, float* win,
                       float* stats, ral_stream s) {
  if (!rec || !win || !stats) return fail("stream_windows: null pointer");
  if (launch_stream_windows(rec, (long long)R, (long long)T, leads, L, hop, (long long)w0, nw, win, stats, (hipStream_t)s))
    return fail("stream_windows: need R >= 1, T >= L, 1 <= hop <= L, L a multiple of 64 and <= 2048, a window range inside the records "
                "(R=%lld T=%lld leads=%d L=%d hop=%d w0=%lld nw=%d)", (long long)R, (long long)T, leads, L, hop, (long long)w0, nw);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_stream_stitch(const float* y, const float* stats, int64_t R, int64_t T, int leads, int L, int hop, float* out,
                      ral_stream s) {
  if (!y || !stats || !out) return fail("stream_stitch: null pointer");
  if (launch_stream_stitch(y, stats, (long long)R, (long long)T, leads, L, hop, out, (hipStream_t)s))
    return fail("stream_stitch: need R >= 1, T >= L, 1 <= hop <= L with L - hop even (R=%lld T=%lld leads=%d L=%d hop=%d)",
                (long long)R, (long long)T, leads, L, hop);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_wavelet_denoise(const float* x, float* y, int64_t rows, int L, float threshold, ral_stream s) {
  if (!x || !y) return fail("wavelet_denoise: null pointer");
  if (!(threshold >= 0.f)) return fail("wavelet_denoise: the threshold factor must be non-negative (got %g)", (double)threshold);
  if (launch_wavelet_denoise(x, y, (long long)rows, L, threshold, (hipStream_t)s))
    return fail("wavelet_denoise: need rows >= 0 and an even record length <= 8192 (rows=%lld L=%d)", (long long)rows, L);
  HIP_OK(hipGetLastError());
  return 0;
}

static int check_attn_args(int N, int H, int Len, int B) {
  if (N < 16 || N % 16 != 0 || N > 1024) return fail("attention: N must be a multiple of 16 in [16, 1024] (got %d)", N);
  if (H < 1 || (H & (H - 1)) != 0 || H > 32) return fail("attention: H must be a power of two <= 32 (got %d)", H);
  if (Len < 0 || Len > N || ((N - Len) & 1)) return fail("attention: the R-wave window (Len=%d) must fit N=%d and be centred", Len, N);
  if (B < 1) return fail("attention: B must be positive");
  return 0;
}

int ral_attention_forward(const float* qkv, float* o, float* lse, const float* table, int N, int H, int Len, int B,
                          ral_stream s) {
  if (!qkv || !o) return fail("attention: null pointer");
  if (check_attn_args(N, H, table ? Len : 0, B)) return -1;
  launch_attn_fwd(qkv, o, lse, table, N, H, attn_head_group(N, H, table ? Len : 0, false), table ? Len : 0, B, attn_f16_default(), (hipStream_t)s);
  HIP_OK(hipGetLastError());
  return 0;
}

int64_t ral_attention_backward_scratch_floats(int N, int H, int Len, int has_table, int B) {
  if (check_attn_args(N, H, has_table ? Len : 0, B)) return -1;
  return (int64_t)attn_bwd_scratch_floats(N, H, has_table ? Len : 0, has_table != 0, B);
}

int ral_attention_backward(const float* qkv, const float* o, const float* d_o, const float* lse, const float* table,
                           float* gtable, float* dqkv, float* scratch, int64_t scratch_floats, int N, int H, int Len,
                           int B, ral_stream s) {
  if (!qkv || !o || !d_o || !lse || !dqkv || (table && !gtable)) return fail("attention: null pointer");
  if (check_attn_args(N, H, table ? Len : 0, B)) return -1;
  // scratch of the two-sweep scalar-path kernels ((B, H, N, 2) floats + table-gradient partials): owned by the caller,
  // so the entry point holds no state, never allocates and can be captured into a hipGraph
  const int64_t need = ral_attention_backward_scratch_floats(N, H, Len, table != nullptr, B);
  if (need > 0 && (!scratch || scratch_floats < need))
    return fail("attention backward: this shape needs %lld floats of scratch (ral_attention_backward_scratch_floats), got %lld",
                (long long)need, (long long)(scratch ? scratch_floats : 0));
  launch_attn_bwd(qkv, o, d_o, lse, table, gtable, dqkv, need > 0 ? scratch : nullptr, need > 0 ? (size_t)scratch_floats : 0, N, H,
                  attn_head_group(N, H, table ? Len : 0, true), table ? Len : 0, B, attn_f16_default(), (hipStream_t)s);
  HIP_OK(hipGetLastError());
  return 0;
}

int ral_debug_tensor(ral_handle* h, const char* name, float** ptr, int64_t* numel) {
  if (!h || h->kind != 0) return fail("no debug tensors for this handle");
  auto it = h->m->dbg.find(name);
  if (it == h->m->dbg.end()) return fail("unknown tensor %s", name);
  *ptr = it->second.first;
  *numel = it->second.second;
  return 0;
}

}  // extern "C"
