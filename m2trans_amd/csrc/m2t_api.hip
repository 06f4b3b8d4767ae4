// m2t_api.hip -- the C ABI (include/m2t.h): plan, whole-model forward / backward launch
// sequences, optimiser and the stand-alone operators.  Host code only; every kernel lives in
// the k_*.hip files.  The launch sequence restates M2Trans.forward / CFTM.forward
// (models/M2Trans_network.py:58-76,132-164) and train.py:199-210.
#include <map>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include "m2t_kernels.h"
#include <cstdlib>
#include "../../include/m2t.h"

static thread_local std::string g_err;
int m2t_set_hip_error(hipError_t e, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d", (int)e, hipGetErrorString(e), file, line);
  g_err = buf;
  return (int)e;
}
int m2t_set_error(int code, const char* msg) { g_err = msg; return code; }

int m2t_ensure_dynamic_lds(const void* kernel, int bytes) {
  static thread_local std::map<std::pair<int, const void*>, int> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  auto it = done.find({dev, kernel});
  if (it != done.end() && it->second >= bytes) return 0;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  done[{dev, kernel}] = bytes;
  return 0;
}

// ---- optional per-kernel timing with HIP events on the launch stream -------------------------
// The enable mask and the event pool are process-wide (an atomic and a mutex-protected pool): the C ABI is entered from
// the caller's thread for m2t_forward and from the autograd engine's worker thread for m2t_backward (model(x);
// loss.backward()), and both must land in the same table.  Only the "a dispatch-timed scope is open" state is per thread.
#include <atomic>
#include <mutex>
namespace {
struct ProfRec { hipEvent_t a, b; int cat; };
struct ProfState {
  std::atomic<unsigned long long> mask{0};
  std::atomic<int> every{1};              // dispatch-timed categories: events ride on one launch in `every` (m2t_profile_sample_every)
  std::mutex mu;
  std::vector<ProfRec> pool;
  size_t used = 0;
  long long seen[64] = {0};               // launches of each category since m2t_profile_enable (under mu)
};
ProfState g_prof;
struct ProfOpen { long long slot = -1; bool taken = false; };
thread_local ProfOpen g_open;             // the record of the scope this thread has open
long long prof_claim(int cat) {           // next free record, or -1 (mask off / not a sampled launch / pool exhausted)
  if (!((g_prof.mask.load(std::memory_order_relaxed) >> cat) & 1ull)) return -1;
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if ((M2T_PROF_DISPATCH_CATS >> cat) & 1ull) {
    // an event-carrying dispatch costs ~10 us of launch path (measured: 16 timed launches per step = +2.3 % on the step);
    // timing a uniform 1-in-N sample of a category's launches keeps the average and most of the step
    const int n = g_prof.every.load(std::memory_order_relaxed);
    if (n > 1 && (g_prof.seen[cat]++ % n) != 0) return -1;
  }
  if (g_prof.used >= g_prof.pool.size()) return -1;
  g_prof.pool[g_prof.used].cat = -1;      // becomes `cat` once both events are on a stream
  return (long long)g_prof.used++;
}
}
void m2t_prof_begin(int cat, hipStream_t st) {
  g_open.slot = prof_claim(cat);
  g_open.taken = false;
  if (g_open.slot < 0) return;
  if ((M2T_PROF_DISPATCH_CATS >> cat) & 1ull) return;           // the launcher takes the events (m2t_prof_take)
  (void)hipEventRecord(g_prof.pool[(size_t)g_open.slot].a, st);
}
namespace { thread_local hipEvent_t g_fork_armed = nullptr; }
hipEvent_t m2t_fork_take() {
  if (!g_fork_armed || (g_open.slot >= 0 && !g_open.taken)) return nullptr;     // a timing pair goes first; the fork then falls back to a record
  hipEvent_t e = g_fork_armed;
  g_fork_armed = nullptr;
  return e;
}
bool m2t_prof_take(hipEvent_t* a, hipEvent_t* b) {
  if (g_open.slot < 0 || g_open.taken) return false;
  g_open.taken = true;
  *a = g_prof.pool[(size_t)g_open.slot].a; *b = g_prof.pool[(size_t)g_open.slot].b;
  return true;
}
void m2t_prof_end(int cat, hipStream_t st) {
  if (g_open.slot < 0) return;
  ProfRec& r = g_prof.pool[(size_t)g_open.slot];
  if ((M2T_PROF_DISPATCH_CATS >> cat) & 1ull) {
    if (g_open.taken) r.cat = cat;         // a scope whose launcher did not take the events stays unlabelled (dropped)
  } else {
    (void)hipEventRecord(r.b, st);
    r.cat = cat;
  }
  g_open.slot = -1;
  g_open.taken = false;
}
extern "C" int m2t_profile_enable(unsigned long long category_mask) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  if (category_mask && g_prof.pool.empty()) {
    g_prof.pool.resize(16384);
    for (auto& r : g_prof.pool) {
      // timing-only events: without the system-scope fence a default event carries, whose L2 write-back lengthens the
      // measured kernel and the one behind it (rocprofv3 of the same step: 46 vs 31 us for a sampled C = 256 attention backward
      // launch, 54 vs 44 us for its successor).  m2t_profile_read is only called after the streams were synchronised.
      if (hipEventCreateWithFlags(&r.a, hipEventDisableSystemFence) != hipSuccess ||
          hipEventCreateWithFlags(&r.b, hipEventDisableSystemFence) != hipSuccess)
        return m2t_set_error(M2T_ERR_STATE, "m2t_profile_enable: hipEventCreate failed");
      r.cat = -1;
    }
  }
  g_prof.mask.store(category_mask, std::memory_order_relaxed);
  g_prof.used = 0;
  for (auto& v : g_prof.seen) v = 0;
  return 0;
}
extern "C" int m2t_profile_sample_every(int n) {
  if (n < 1) return m2t_set_error(M2T_ERR_ARG, "m2t_profile_sample_every: n >= 1");
  g_prof.every.store(n, std::memory_order_relaxed);
  return 0;
}
// total milliseconds and launch count of one category since m2t_profile_enable, over every thread that launched; the
// caller must have synchronised the streams
extern "C" int m2t_profile_read(int cat, double* total_ms, long long* count) {
  std::lock_guard<std::mutex> lk(g_prof.mu);
  double t = 0.0; long long n = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    if (g_prof.pool[i].cat != cat) continue;
    float ms = 0.f;
    hipError_t e = hipEventElapsedTime(&ms, g_prof.pool[i].a, g_prof.pool[i].b);
    if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
    t += ms; ++n;
  }
  if (total_ms) *total_ms = t;
  if (count) *count = n;
  return 0;
}

struct WsTensor { size_t off; size_t n; };   // byte offset, element count

struct m2t_plan {
  int B, H0, W0, H, W, scale, nb, dt;
  size_t esz;
  long long P;                       // padded LR pixels per image
  int Hs, Ws, Hsp, Wsp;              // SR size (cropped) and padded SR size
  std::vector<std::string> pnames;
  std::map<std::string, long long> poff, pnum;
  long long nparams = 0;
  std::map<std::string, WsTensor> ws;
  size_t ws_bytes = 0;
  std::vector<m2t_pack_desc> descs;
  std::vector<char> desc_is_qkv;             // per descriptor: a packed form of a qkv_conv.weight
  std::vector<int> pack_blocks;              // (descriptor, chunk) pairs: one workgroup of the packing kernel each.  Order: every
                                             // descriptor that is always needed, then the plain copies of the C >= 64 qkv weights (read only by the
                                             // unfused forward GEMM), then their transposes (read only by the unfused data-gradient GEMM)
  int pack_nb_base = 0, pack_nb_copy = 0, pack_nb_tr = 0;   // workgroups of the three groups (round 5: the default bf16 path skips the last two:
                                             // 6.3 M of the 14.2 M packed elements, the transposes being the slowest gathers of the kernel)
  std::map<std::string, long long> pk;       // packed weight offsets (elements of T)
  long long npacked = 0;
  bool have_seed = false, have_acts = false;
  // m2t_l1_loss_deferred: the loss and the seed are produced inside the next m2t_backward (round 5)
  bool l1_deferred = false;
  const float* l1_hr = nullptr; float* l1_loss_out = nullptr; float l1_sc = 0.f, l1_R = 0.f;
  int use_fp32_fast = 1;               // fp32: the v_mfma_f32_32x32x2_f32 GEMM / qkv weight-gradient kernels of round 5 (k_gemm.hip); 0 = the 16x16x4 kernels
  int use_fused_l1 = 1;                // bf16 x4: the clamp + L1 seed inside the fused tail backward when the loss was requested through m2t_l1_loss_deferred
  int use_tail_bwd32 = 1;              // bf16 x4, recomputing fused tail backward: the 32x32x16-MFMA kernel of round 6 (k_tail_bwd.hip); 0 = the 16x16x32 kernel
  // ---- options (m2t_set_option; include/m2t.h documents each) ----
  bool use_side = true;
  bool debug_skip_side = false;        // timing experiments only: skip every parameter-gradient kernel (results are WRONG)
  bool use_fused_tail_bwd = true;      // x4 bf16: k_tail_bwd.hip instead of four HR kernels                              } option "fused_tail":
  bool use_fused_tail_fwd = true;      // x4 bf16: tail.3 expansion + GELU + tail conv in one kernel, gelu(t2) / gelu'(t2) never stored } 0 / 1 / 2 / 3
  bool use_stream_tail_fwd = true;     // ... as the row-streaming kernel (k_tail_stream.hip, round 4: 152 vs 229 us, same bits); 2 = the 16x16-tile kernel
  bool use_stream_tail_bwd = false;    // x4: the row-streaming BACKWARD (k_tail_bwd_stream.hip) instead of the tile kernel: option value 4, kept for
                                       // A/B -- same data gradient bits, 580 against 377 us stand-alone at batch 16
                                       // (with bf16 mode's exp2 / rcp GELU: 5.30 vs 5.34 ms per step and 1.6 GB less HBM traffic;
                                       // with the erf form of round 2 it was 1 % slower)
  int use_fused_prep_bwd = 1;          // bf16: branch_prep_bwd of branch 4 inside the attention backward of branch 3 (round 4)
  int use_fused_prep_fwd = 1;          // bf16 C = 64 / 256 branches: branch_prep inside the fused forward attention kernel (round 4)
  int fused_attn_fwd2 = 0;             // bf16 C = 256 forward branch with branch_prep inside: the two-windows-per-CU kernels (k_attn_fwd2.hip, round 5):
                                       // 0 = never (default: measured equal to the one-window kernel within 2 %, profiles/README.md), 1 = 4-wave workgroups
                                       // (two per CU), 2 = 8-wave workgroups of two windows, -1 = variant 1 when the branch has more windows than CUs (256)
  int fwd2_variant(long long nwin, int h, int w) const {
    if (fused_attn_fwd2 == 0 || (fused_attn_fwd2 < 0 && nwin <= 256)) return 0;
    const bool even = (((h / 8) * (w / 8)) & 1) == 0;
    if (fused_attn_fwd2 == 2 && even) return 2;
    return 1;
  }
  int use_fused_norm_red = 0;          // bf16 with the C = 16 prep kernel: the first stage of the InstanceNorm backward reduction rides in that launch (round 5;
                                       // measured SLOWER, -2.2 % on the step: both roles are memory-heavy, profiles/README.md -- kept for A/B)
  int fork_on_kernel = 1;              // a fork event rides on the dispatch it follows (its stop event) instead of a marker packet behind it:
                                       // same-box A/B 4.757 -> 4.726 ms (config 1), 8.536 -> 8.469 (config 3); not under stream capture
  int gate_branch = -1;                // side-stream gate: -1 ungated (a branch's side work follows its attention launch), else the branch (3..0)
                                       // behind whose attention launch a block's parameter-gradient work is released.  Same-box A/B (config 1),
                                       // end of round 3 (one buffer-set wait per block instead of one per branch): ungated 4.84 ms, 2 = 4.95,
                                       // 0 = 5.10, 3 = 5.10, 1 = 5.14, one stream 5.35; batch 32: ungated 8.78, 2 = 8.76.  (While the main chain
                                       // still waited for the side stream once per BRANCH the gate paid: round 2: 1 = 5.49, 2 = 5.60, ungated 5.64)
  bool use_resident_attn_bwd = true;   // bf16: whole-window-resident attention backward (k_attn_res.hip)         } option "attn_bwd":
  bool use_fused_conv_bwd = true;      // bf16 conv3x3 64 -> 64 backward: data + weight / bias gradient in one row-streaming pass (k_conv.hip)
  int use_conv_rows = 1;               // bf16 conv3x3 64 -> 64: row-streaming LDS-DMA kernel (k_conv.hip); 0 = the tile kernel
  int wgrad_big_tiles = -1;            // C = 256 qkv weight gradient: 128 x 128 output tiles (k_gemm.hip); value = target workgroups,
                                       // 0 = off, -1 = auto: 256 from 24 576 rows on (batch 32: 9.82 vs 9.93 ms; batch 16: 5.53 vs 5.49)
  bool use_fused_qkv_dgrad = true;     // bf16, C = 64 / 256: projection data gradient inside that kernel                } 0 .. 3
  bool use_c16_prep = true;            // bf16, C = 16: overlap-add + projection data gradient + branch_prep_bwd in one kernel }
  int use_fused_c16_fwd = 2;           // bf16, C = 16 branch: norm apply + qkv projection + attention + residual in one kernel (k_attn_c16.hip);
                                       // 2: ... and qkv1 is not stored: the wave-per-window backward recomputes it from d1 (needs attn_bwd >= 1)
  // x2 / x3, bf16: expansion + PixelShuffle + GELU + tail conv as ONE row-streaming forward kernel and ONE recomputing backward kernel
  // (k_tail_stream.hip, k_tail_bwd_stream.hip; option "fused_tail" >= 1): gelu(t) / gelu'(t) are never stored
  bool stream_tail_x23() const { return dt != M2T_F32 && scale != 4 && use_fused_tail_bwd; }
  bool c16_recompute() const { return dt != M2T_F32 && use_fused_c16_fwd == 2 && use_resident_attn_bwd; }
  int use_fused_attn_fwd = 2;          // bf16, C = 64 / 256: qkv projection + attention + epilogue in one kernel (k_attn_fused.hip);
                                       // 2: ... and qkv2 (C = 64) is not stored: the resident backward recomputes it from d2 (needs attn_bwd = 2)
  bool c64_recompute() const { return dt != M2T_F32 && use_fused_attn_fwd == 2 && use_resident_attn_bwd && use_fused_qkv_dgrad; }
  // deferred, batched parameter-gradient reductions (m2t_backward): slabs live in the "arena" workspace
  // region; the descriptor table is identical every step, so it is uploaded once
  std::vector<m2t_red_desc> red_descs;
  bool red_uploaded = false;
  size_t arena_floats = 0;
  hipStream_t side = nullptr;
  // gradient buckets: [lo, hi) float ranges of the flat gradient buffer in the order m2t_backward completes them
  // (tail, block pairs from the last to the first, head); one event each, recorded behind the bucket's reduction
  std::vector<std::pair<long long, long long>> buckets;
  std::vector<hipEvent_t> bucket_events;
  std::vector<hipEvent_t> events;
  int ensure_side(hipStream_t caller) {
    if (side) return 0;
    // (a CU-masked side stream and a side stream of LOWER priority than the caller's were both measured and are slower: profiles/README.md,
    //  round 2.)  The side stream takes the CALLER'S priority: a caller that runs the step on a high-priority stream beside other work of
    //  its own (the MedCLIP encoder of configs[2] on a normal-priority stream, round 5) gets both halves of the backward pass ahead of it
    int prio = 0;
    if (hipStreamGetPriority(caller, &prio) != hipSuccess) prio = 0;
    if (hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio) != hipSuccess) return -1;
    events.resize(192);
    for (auto& e : events)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
    bucket_events.resize(buckets.size());
    for (auto& e : bucket_events)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
    return 0;
  }
  ~m2t_plan() {
    for (auto e : events) if (e) (void)hipEventDestroy(e);
    for (auto e : bucket_events) if (e) (void)hipEventDestroy(e);
    // the caller releases the workspace right after this: nothing of the plan's own stream may still be running in it
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
  }

  void add_param(const std::string& n, long long cnt) { pnames.push_back(n); poff[n] = nparams; pnum[n] = cnt; nparams += cnt; }
  size_t add_ws(const std::string& n, size_t elems, size_t es) {
    ws_bytes = (ws_bytes + 255) & ~(size_t)255;
    ws[n] = WsTensor{ws_bytes, elems};
    ws_bytes += elems * es;
    return ws[n].off;
  }
  long long add_pack(const std::string& n, const std::string& src, int kind, long long cnt, int d0, int d1, int d2) {
    npacked = (npacked + 7) & ~7LL;
    m2t_pack_desc d;
    d.src_off = poff.at(src); d.dst_off = npacked; d.n = cnt; d.kind = kind; d.d0 = d0; d.d1 = d1; d.d2 = d2;
    descs.push_back(d);
    desc_is_qkv.push_back(src.find("qkv_conv.weight") != std::string::npos ? 1 : 0);
    pk[n] = npacked;
    npacked += cnt;
    return pk[n];
  }
};

static const int BR_C[4] = {16, 64, 256, 256};
static const int BR_L[4] = {0, 1, 2, 2};

extern "C" int m2t_version(void) { return 100; }
extern "C" const char* m2t_last_error_string(void) { return g_err.c_str(); }

extern "C" int m2t_plan_create(m2t_plan** out, int B, int H0, int W0, int scale, int n_blocks, int dtype) {
  if (!out || B < 1 || H0 < 2 || W0 < 2 || (scale != 2 && scale != 3 && scale != 4) || n_blocks < 1 ||
      (dtype != M2T_F32 && dtype != M2T_BF16))
    return m2t_set_error(M2T_ERR_ARG, "m2t_plan_create: bad argument");
  m2t_plan* p = new m2t_plan();
  p->B = B; p->H0 = H0; p->W0 = W0; p->scale = scale; p->nb = n_blocks; p->dt = dtype;
  p->esz = (dtype == M2T_F32) ? 4 : 2;
  p->H = (H0 + 31) / 32 * 32;
  p->W = (W0 + 31) / 32 * 32;
  if (p->H - H0 >= H0 || p->W - W0 >= W0) { delete p; return m2t_set_error(M2T_ERR_ARG, "m2t_plan_create: reflect pad needs pad < size"); }
  p->P = (long long)p->H * p->W;
  p->Hs = H0 * scale; p->Ws = W0 * scale; p->Hsp = p->H * scale; p->Wsp = p->W * scale;
  const int s = scale;
  // ---- parameters: trainable tensors in the reference's registration order ----
  p->add_param("head.weight", 64 * 3 * 9);
  p->add_param("head.bias", 64);
  for (int b = 0; b < n_blocks; ++b) {
    for (int i = 0; i < 4; ++i) {
      const int C = BR_C[i];
      const std::string pre = "body." + std::to_string(b) + ".attn" + std::to_string(i + 1) + ".";
      p->add_param(pre + "rel_h", 10 * C / 2);
      p->add_param(pre + "rel_w", 10 * C / 2);
      p->add_param(pre + "qkv_conv.weight", 3LL * C * C);
    }
    const std::string pre = "body." + std::to_string(b) + ".feed_forward.0.";
    p->add_param(pre + "weight", 64 * 64 * 9);
    p->add_param(pre + "bias", 64);
  }
  if (s == 4) {
    p->add_param("tail.0.weight", 256 * 64); p->add_param("tail.0.bias", 256);
    p->add_param("tail.3.weight", 256 * 64); p->add_param("tail.3.bias", 256);
    p->add_param("tail.6.weight", 3 * 64 * 9);
  } else {
    p->add_param("tail.0.weight", 64LL * s * s * 64); p->add_param("tail.0.bias", 64 * s * s);
    p->add_param("tail.3.weight", 3 * 64 * 9);
  }
  // ---- packed weights (element type T) ----
  for (int b = 0; b < n_blocks; ++b) {
    for (int i = 0; i < 4; ++i) {
      const int C = BR_C[i];
      const std::string pre = "body." + std::to_string(b) + ".attn" + std::to_string(i + 1) + ".";
      const std::string k = "b" + std::to_string(b) + ".w" + std::to_string(i + 1);
      p->add_pack(k, pre + "qkv_conv.weight", M2T_PACK_COPY, 3LL * C * C, 0, 0, 0);
      p->add_pack(k + "T", pre + "qkv_conv.weight", M2T_PACK_TRANSPOSE, 3LL * C * C, 3 * C, C, 0);
      if (C >= 64) p->add_pack(k + "F", pre + "qkv_conv.weight", M2T_PACK_FRAG16, 3LL * C * C, 3 * C, C, 0);   // k_attn_fused.hip
      if (C >= 64) p->add_pack(k + "TF", pre + "qkv_conv.weight", M2T_PACK_FRAG16_T, 3LL * C * C, C, 3 * C, 0);  // Wqkv^T fragments: fused data gradient (k_attn_res.hip)
    }
    const std::string pre = "body." + std::to_string(b) + ".feed_forward.0.weight";
    p->add_pack("b" + std::to_string(b) + ".wf", pre, M2T_PACK_CONV3, 64 * 64 * 9, 64, 64, 0);
    p->add_pack("b" + std::to_string(b) + ".wfT", pre, M2T_PACK_CONV3_T, 64 * 64 * 9, 64, 64, 0);
    p->add_pack("b" + std::to_string(b) + ".wfR", pre, M2T_PACK_CONV3_ROWS, 64 * 64 * 9, 64, 64, 0);      // conv3x3_c64_rows_kernel (bf16)
    p->add_pack("b" + std::to_string(b) + ".wfTR", pre, M2T_PACK_CONV3_ROWS_T, 64 * 64 * 9, 64, 64, 0);
  }
  {
    const int r0 = (s == 4) ? 2 : s;
    p->add_pack("t0", "tail.0.weight", M2T_PACK_SHUF_ROWS, 64LL * r0 * r0 * 64, 64, r0 * r0, 64);
    p->add_pack("t0T", "tail.0.weight", M2T_PACK_SHUF_ROWS_T, 64LL * r0 * r0 * 64, 64, r0 * r0, 64);
    if (s == 4) {
      p->add_pack("t3", "tail.3.weight", M2T_PACK_SHUF_ROWS, 256 * 64, 64, 4, 64);
      p->add_pack("t3T", "tail.3.weight", M2T_PACK_SHUF_ROWS_T, 256 * 64, 64, 4, 64);
    }
  }
  // ---- workspace ----
  const size_t es = p->esz;
  const long long BP = (long long)B * p->P;
  p->add_ws("zero_page", 256, 1);          // source of every out-of-image pixel the LDS-DMA kernels stage (k_conv.hip)
  p->add_ws("pack_descs", p->descs.size() * sizeof(m2t_pack_desc), 1);
  for (int group = 0; group < 3; ++group) {
    int nb_g = 0;
    for (size_t i = 0; i < p->descs.size(); ++i) {
      const bool big = p->descs[i].n >= 3LL * 64 * 64 && (p->descs[i].kind == M2T_PACK_COPY || p->descs[i].kind == M2T_PACK_TRANSPOSE) && p->desc_is_qkv[i];
      const int gi = !big ? 0 : (p->descs[i].kind == M2T_PACK_COPY ? 1 : 2);
      if (gi != group) continue;
      for (long long c = 0; c * M2T_PACK_CHUNK < p->descs[i].n; ++c) { p->pack_blocks.push_back((int)i); p->pack_blocks.push_back((int)c); ++nb_g; }
    }
    (group == 0 ? p->pack_nb_base : (group == 1 ? p->pack_nb_copy : p->pack_nb_tr)) = nb_g;
  }
  p->add_ws("pack_blocks", p->pack_blocks.size() * sizeof(int), 1);
  p->add_ws("packed", p->npacked, es);
  for (int b = 0; b <= n_blocks; ++b) p->add_ws("X" + std::to_string(b), BP * 64, es);
  for (int b = 0; b < n_blocks; ++b) {
    const std::string k = "b" + std::to_string(b) + ".";
    p->add_ws(k + "mean", B * 64, 4);
    p->add_ws(k + "rstd", B * 64, 4);
    p->add_ws(k + "xc", BP * 64, es);
    for (int i = 0; i < 4; ++i) {
      p->add_ws(k + "d" + std::to_string(i + 1), BP * 16, es);
      p->add_ws(k + "qkv" + std::to_string(i + 1), BP * 48, es);
    }
  }
  p->add_ws("xin", BP * 16, es);
  p->add_ws("a", BP * 16, es);
  p->add_ws("norm_part", (size_t)B * 8 * M2T_NORM_SPLIT * 64 * 3, 4);      // (the conv epilogue leaves up to 256 partials per image)
  p->add_ws("norm_s", (size_t)B * 64 * 2, 4);
  p->add_ws("vring", window_attn_fwd2_vring_elems(B, p->H / 4, p->W / 4), es);      // v rows of the ring keys (k_attn_fwd2.hip)
  p->add_ws("norm_part0", (size_t)B * p->H * (p->W / 16) * 32, 4);      // per-tile plane-0 partials of the InstanceNorm backward (fused_norm_red)
  const int r0 = (s == 4) ? 2 : s;
  // tail activations: gelu(t) and gelu'(t) of each expansion (the pre-activation t itself is never needed again)
  p->add_ws("t1act", BP * r0 * r0 * 64, es);
  p->add_ws("t1der", BP * r0 * r0 * 64, es);
  if (s == 4) { p->add_ws("t2act", BP * 16 * 64, es); p->add_ws("t2der", BP * 16 * 64, es); }
  p->add_ws("srpre", (size_t)B * 3 * p->Hsp * p->Wsp, 4);
  p->add_ws("loss_part", M2T_LOSS_BLOCKS, 4);
  // backward
  p->add_ws("gpre", (size_t)B * 3 * p->Hsp * p->Wsp, 4);
  if (s == 4) p->add_ws("g_t2pre", BP * 16 * 64, es);
  p->add_ws("g_t1pre", BP * r0 * r0 * 64, es);
  p->add_ws("gT", BP * 64, es);
  p->add_ws("gA", BP * 64, es);
  p->add_ws("gB", BP * 64, es);
  p->add_ws("gxc", BP * 64, es);
  p->add_ws("gn", BP * 64, es);
  p->add_ws("ga", BP * 16, es);
  p->add_ws("gd", BP * 16, es);
  p->add_ws("gd2", BP * 16, es);       // second set: a branch's attention backward reads the previous branch's rows while it writes its own (fused_prep_bwd)
  p->add_ws("gdwin2", BP * 9, es);
  p->add_ws("gdwin", BP * 9, es);      // ring rows of the fused projection data gradient: [windows][36][C], windows * C = BP / 4
  p->add_ws("head_cols", BP * 32, es);
  for (int i = 0; i < 4; ++i) {     // TWO sets per branch (even / odd blocks): the side stream may lag the main chain by two blocks, and
    for (const char* set : {"", "b"}) {   // the main chain waits for it once per block instead of once per branch
      p->add_ws("gqkv" + std::to_string(i) + set, BP * 48, es);
      p->add_ws("win" + std::to_string(i) + set, BP * 50, es);
      p->add_ws("relw" + std::to_string(i) + set, (size_t)(BP / 64) * 10 * 16, 4);
    }
  }
  p->add_ws("rel_part", 32 * 10 * 256, 4);
  {
    // arena: every slab set of one backward pass (see m2t_backward); sized from the launchers' slab rules
    size_t per_block = (size_t)256 * 9 * 64 * 64 + (size_t)256 * 64;                 // conv wgrad + ff bias partials
    per_block += (size_t)1024 * 768 + (size_t)512 * 12288 + 2 * (size_t)32 * 196608; // qkv wgrads (upper bounds)
    per_block += 4 * (size_t)32 * 2560;                                               // rel-pos partials
    size_t tail = 2 * (size_t)256 * (16384 + 36864) + (size_t)1024 * 2048 + 4 * (size_t)256 * 768 + (size_t)256 * 1728 * 2;
    p->arena_floats = per_block * n_blocks + tail + (size_t)512 * (64 * 32 + 64) + (1u << 20);
    p->add_ws("arena", p->arena_floats, 4);
    p->add_ws("red_descs", 512 * sizeof(m2t_red_desc), 1);
  }
  p->add_ws("col_part", (size_t)256 * 768, 4);
  p->ws_bytes = (p->ws_bytes + 255) & ~(size_t)255;
  *out = p;
  {
    // gradient buckets in completion order (see m2t_backward): the tail, then the blocks in the groups the
    // deferred reductions are flushed in (after every even block index, walking from the last block to the first),
    // then whatever precedes the lowest flushed block (the head).  state_dict order makes each a contiguous range.
    auto first_of = [&](const std::string& prefix) {
      long long lo = p->nparams;
      for (const auto& n : p->pnames) if (n.rfind(prefix, 0) == 0) lo = std::min(lo, p->poff.at(n));
      return lo;
    };
    long long hi = p->nparams;
    const long long tail_lo = first_of("tail.");
    p->buckets.push_back({tail_lo, hi});
    hi = tail_lo;
    for (int b = n_blocks - 1; b >= 0; --b)
      if ((b & 1) == 0) {
        const long long lo = first_of("body." + std::to_string(b) + ".");
        p->buckets.push_back({lo, hi});
        hi = lo;
      }
    p->buckets.push_back({0, hi});
  }
  return 0;
}
extern "C" void m2t_plan_destroy(m2t_plan* p) { delete p; }

extern "C" long long m2t_plan_query(const m2t_plan* p, const char* key) {
  if (!p || !key) return -1;
  const std::string k(key);
  if (k == "workspace_bytes") return (long long)p->ws_bytes;
  if (k == "num_params") return p->nparams;
  if (k == "num_param_tensors") return (long long)p->pnames.size();
  if (k == "padded_h") return p->H;
  if (k == "padded_w") return p->W;
  if (k == "grad_buckets") return (long long)p->buckets.size();
  if (k.rfind("grad_bucket_lo:", 0) == 0) { const size_t i = (size_t)atoll(k.c_str() + 15); return i < p->buckets.size() ? p->buckets[i].first : -1; }
  if (k.rfind("grad_bucket_hi:", 0) == 0) { const size_t i = (size_t)atoll(k.c_str() + 15); return i < p->buckets.size() ? p->buckets[i].second : -1; }
  if (k.rfind("param:", 0) == 0) { auto it = p->poff.find(k.substr(6)); return it == p->poff.end() ? -1 : it->second; }
  if (k.rfind("numel:", 0) == 0) { auto it = p->pnum.find(k.substr(6)); return it == p->pnum.end() ? -1 : it->second; }
  if (k.rfind("ws:", 0) == 0) { auto it = p->ws.find(k.substr(3)); return it == p->ws.end() ? -1 : (long long)it->second.off; }
  if (k.rfind("wsn:", 0) == 0) { auto it = p->ws.find(k.substr(4)); return it == p->ws.end() ? -1 : (long long)it->second.n; }
  if (k.rfind("packed:", 0) == 0) { auto it = p->pk.find(k.substr(7)); return it == p->pk.end() ? -1 : it->second; }
  // which stored tensors the current options leave unwritten (tests read the workspace by name)
  if (k.rfind("opt:", 0) == 0) {        // the options in force (profile.py prices the kernels that actually run)
    const std::string o = k.substr(4);
    if (o == "side_stream") return p->use_side;
    // (the EFFECTIVE state, like the keys below: an option whose precondition is off did not run)
    if (o == "fork_on_kernel") return p->use_side && p->fork_on_kernel;
    if (o == "fp32_fast") return p->dt == M2T_F32 && p->use_fp32_fast;
    if (o == "tail_bwd_mfma32") return p->dt != M2T_F32 && p->scale == 4 && p->use_tail_bwd32 && p->use_fused_tail_bwd && p->use_fused_tail_fwd && !p->use_stream_tail_bwd;
    if (o == "fused_l1") return p->dt != M2T_F32 && p->scale == 4 && p->use_fused_l1 && p->use_fused_tail_bwd && p->use_fused_tail_fwd && !p->use_stream_tail_bwd;
    if (o == "fused_attn_fwd2") {      // effective: would the C = 256 branches run k_attn_fwd2.hip
      const bool eligible = p->dt != M2T_F32 && p->use_fused_attn_fwd != 0 && p->use_fused_prep_fwd;
      const long long nwin = (long long)p->B * (p->H / 32) * (p->W / 32);
      return eligible ? p->fwd2_variant(nwin, p->H / 4, p->W / 4) : 0;
    }
    if (o == "fused_norm_red") return p->dt != M2T_F32 && p->use_resident_attn_bwd && p->use_fused_qkv_dgrad && p->use_c16_prep && p->use_fused_norm_red;
    if (o == "fused_prep_fwd") return p->dt != M2T_F32 && p->use_fused_attn_fwd != 0 && p->use_fused_prep_fwd;
    if (o == "fused_prep_bwd") return p->dt != M2T_F32 && p->use_resident_attn_bwd && p->use_fused_qkv_dgrad && p->use_fused_prep_bwd;
    if (o == "gate_branch") return p->gate_branch + 1000;      // (offset: -1 is the "unknown key" value of this function)
    if (o == "wgrad_big_tiles") return p->wgrad_big_tiles + 1000;
    if (o == "fused_tail") {
      if (p->dt == M2T_F32) return 0;
      if (p->scale != 4) return p->stream_tail_x23() ? 3 : 0;      // x2 / x3: the row-streaming pair or the plain kernels
      return p->use_fused_tail_bwd ? (p->use_fused_tail_fwd ? (p->use_stream_tail_fwd ? (p->use_stream_tail_bwd ? 4 : 3) : 2) : 1) : 0;
    }
    if (o == "attn_bwd") return p->dt == M2T_F32 ? 0 : (p->use_resident_attn_bwd ? (p->use_fused_qkv_dgrad ? (p->use_c16_prep ? 3 : 2) : 1) : 0);
    if (o == "conv_rows") return p->dt != M2T_F32 ? p->use_conv_rows : 0;
    if (o == "fused_conv_bwd") return p->dt != M2T_F32 && p->use_fused_conv_bwd && conv3x3_c64_bwd_fusable(p->B, p->H, p->W);
    if (o == "fused_attn_fwd") return p->dt != M2T_F32 ? (p->use_fused_attn_fwd == 2 && !p->c64_recompute() ? 1 : p->use_fused_attn_fwd) : 0;
    if (o == "fused_c16_fwd") return p->dt != M2T_F32 ? (p->use_fused_c16_fwd == 2 && !p->use_resident_attn_bwd ? 1 : p->use_fused_c16_fwd) : 0;
    if (o == "fused_qkv_dgrad") return p->use_fused_qkv_dgrad && p->use_resident_attn_bwd && p->dt != M2T_F32;
    if (o == "debug_skip_side") return p->debug_skip_side;
    return -1;
  }
  if (k == "stores_qkv2") return p->c64_recompute() ? 0 : 1;
  if (k == "stores_qkv1") return (p->use_fused_c16_fwd != 0 && p->c16_recompute()) ? 0 : 1;
  if (k == "stores_t1") return p->stream_tail_x23() ? 0 : 1;
  if (k == "stores_t2") return (p->scale == 4 && !(p->dt != M2T_F32 && p->use_fused_tail_fwd && p->use_fused_tail_bwd)) ? 1 : 0;
  return -1;
}

#define WSP(name) ((char*)workspace + p->ws.at(name).off)
#define CK(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

extern "C" int m2t_plan_init_workspace(m2t_plan* p, void* workspace, void* stream) {
  if (!p || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_plan_init_workspace: null");
  hipError_t e = hipMemcpyAsync(WSP("pack_descs"), p->descs.data(), p->descs.size() * sizeof(m2t_pack_desc),
                                hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  e = hipMemcpyAsync(WSP("pack_blocks"), p->pack_blocks.data(), p->pack_blocks.size() * sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  e = hipMemsetAsync(WSP("zero_page"), 0, 256, (hipStream_t)stream);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  e = hipStreamSynchronize((hipStream_t)stream);   // the host table may be freed/moved afterwards
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  p->have_acts = p->have_seed = false;
  return 0;
}

static inline char* packed_ptr(const m2t_plan* p, void* workspace, const std::string& k) {
  return (char*)workspace + p->ws.at("packed").off + p->pk.at(k) * p->esz;
}

extern "C" int m2t_forward(m2t_plan* p, const float* params, const float* x, float* sr, float rgb_range,
                           int keep_activations, void* workspace, void* stream) {
  if (!p || !params || !x || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  g_m2t_f32_fast = p->use_fp32_fast;        // thread-local switch the fp32 launchers of k_gemm.hip read; back to its default on every exit path
  struct F32FastGuard { ~F32FastGuard() { g_m2t_f32_fast = 1; } } f32_fast_guard;
  const int dt = p->dt, B = p->B, H = p->H, W = p->W, s = p->scale;
  const long long BP = (long long)B * p->P;
  (void)keep_activations;   // v1 keeps every activation in the workspace either way
  // (re-packing on the side stream under the head conv was measured: -0.4 %, both kernels are bound by workgroup launch rate)
  {
    // the plain / transposed copies of the C >= 64 qkv weights are read only by the unfused projection GEMMs (forward / data gradient):
    // with the fused attention kernels in force (the bf16 default) they are not packed.  A change of either option invalidates the
    // activations (m2t_set_option), so the backward pass always meets the packs of the options it runs under
    const bool need_copy = dt == M2T_F32 || p->use_fused_attn_fwd == 0;
    const bool need_tr = dt == M2T_F32 || !(p->use_fused_qkv_dgrad && p->use_resident_attn_bwd);
    const char* blocks = (const char*)WSP("pack_blocks");
    const int n0 = p->pack_nb_base + (need_copy ? p->pack_nb_copy : 0) + ((need_copy && need_tr) ? p->pack_nb_tr : 0);
    CK(launch_pack(dt, params, WSP("packed"), (const m2t_pack_desc*)WSP("pack_descs"), blocks, n0, st));
    if (need_tr && !need_copy)
      CK(launch_pack(dt, params, WSP("packed"), (const m2t_pack_desc*)WSP("pack_descs"), blocks + (size_t)(p->pack_nb_base + p->pack_nb_copy) * 2 * sizeof(int), p->pack_nb_tr, st));
  }
  CK(launch_head_conv_fwd(dt, x, params + p->poff.at("head.weight"), params + p->poff.at("head.bias"), WSP("X0"), B,
                          p->H0, p->W0, H, W, st));
  int stat_partials = 0;
  for (int b = 0; b < p->nb; ++b) {
    const std::string k = "b" + std::to_string(b) + ".";
    const std::string pre = "body." + std::to_string(b) + ".";
    void* X = WSP("X" + std::to_string(b));
    float* mean = (float*)WSP(k + "mean");
    float* rstd = (float*)WSP(k + "rstd");
    void* xc = WSP(k + "xc");
    // statistics of the block input: left as per-segment partials by the previous block's conv (bf16, row-streaming), else two stages
    if (stat_partials > 0) CK(launch_instnorm_finalize((const float*)WSP("norm_part"), mean, rstd, B, stat_partials, st));
    else CK(launch_instnorm_stats(dt, X, mean, rstd, (float*)WSP("norm_part"), B, (int)p->P, st));
    for (int i = 0; i < 4; ++i) {
      const int C = BR_C[i], L = BR_L[i];
      const int h = H >> L, w = W >> L;
      const long long M = (long long)B * h * w;
      const std::string an = pre + "attn" + std::to_string(i + 1) + ".";
      void* d = WSP(k + "d" + std::to_string(i + 1));
      void* qkv = WSP(k + "qkv" + std::to_string(i + 1));
      const float* rh = params + p->poff.at(an + "rel_h");
      const float* rw = params + p->poff.at(an + "rel_w");
      {
        void* xc_i = (char*)xc + (size_t)i * BP * 16 * p->esz;       // chunk i of the P64 concat buffer: a dense plane
        if (dt != M2T_F32 && p->use_fused_c16_fwd != 0 && i == 0) {
          // x1 = attn1(norm(x)[chunk 0]) + norm(x)[chunk 0] (:135-139): one launch, d1 and qkv1 written for the backward
          CK(launch_window_attn_fused_c16_fwd(X, mean, rstd, packed_ptr(p, workspace, k + "w1"), rh, rw, d, p->c16_recompute() ? nullptr : qkv, xc_i, 16, 0, B, h, w, st));
          continue;
        }
        if (dt != M2T_F32 && p->use_fused_attn_fwd != 0 && p->use_fused_prep_fwd && C >= 64 && i >= 1) {
          // branch_prep (norm apply + mix + DWT^L), the qkv projection, the window attention and IWT^L / residual in one kernel
          const void* xn_i = (const char*)X + (size_t)i * BP * 16 * p->esz;
          const void* xprev = (const char*)xc + (size_t)(i - 1) * BP * 16 * p->esz;
          const long long nwin = (long long)B * (h / 8) * (w / 8);
          const int v2 = (C == 256 && L == 2) ? p->fwd2_variant(nwin, h, w) : 0;
          if (v2 != 0) {
            // more windows than CUs: the kernels that put two windows on a CU (k_attn_fwd2.hip)
            CK(launch_window_attn_fused_prep_fwd2(xn_i, xprev, mean, rstd, i, WSP("xin"), d, packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "F"),
                                                  rh, rw, qkv, xc_i, WSP("vring"), B, h, w, v2, st));
            continue;
          }
          CK(launch_window_attn_fused_prep_fwd(xn_i, xprev, mean, rstd, i, WSP("xin"), d, packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "F"),
                                               rh, rw, (C == 64 && p->c64_recompute()) ? nullptr : qkv, xc_i, B, h, w, C, L, st));
          continue;
        }
        CK(launch_branch_prep(dt, L, X, mean, rstd, xc, i, WSP("xin"), d, B, H, W, st));
        if (dt != M2T_F32 && p->use_fused_attn_fwd != 0 && C >= 64) {
          // qkv projection + window attention + IWT^L / residual in one kernel; qkv is written for the backward pass unless that
          // recomputes it (C = 64)
          CK(launch_window_attn_fused_fwd(d, packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "F"), rh, rw,
                                          (C == 64 && p->c64_recompute()) ? nullptr : qkv, xc_i, 16, 0,
                                          WSP("xin"), 16, B, h, w, C, L, st));
          continue;
        }
        m2t_gemm_args ga{};
        ga.A = d; ga.lda = C; ga.W = packed_ptr(p, workspace, k + "w" + std::to_string(i + 1));
        ga.Y = qkv; ga.ldy = 3 * C; ga.M = M; ga.N = 3 * C; ga.K = C;
        { M2TProfScope ps(M2T_PROF_GEMM_QKV, st); CK(launch_gemm_nt(dt, M2T_A_PLAIN, M2T_E_PLAIN, ga, st)); }
        if (i == 0) {
          // x1 = attn1(x1) + x1 written straight into the concat buffer (:139,163)
          CK(launch_window_attn_fwd(dt, qkv, rh, rw, xc_i, 16, 0, d, 16, B, h, w, C, st));
        } else {
          // x_k = IWT^L(attn_k(.)) + x_k_in written straight into the concat buffer (:145,153,161,163)
          CK(launch_window_attn_fwd(dt, qkv, rh, rw, xc_i, 16, 0, WSP("xin"), 16, B, h, w, C, st, L));
        }
      }
    }
    // x = feed_forward(xc) + x (:164); the last block also folds in `res + x` (:70)
    { M2TProfScope ps(M2T_PROF_CONV3_FWD, st);
      const int variant = p->use_conv_rows ? 0 : 1;
      stat_partials = (b < p->nb - 1) ? conv3x3_c64_stat_partials(dt, B, H, W, variant) : 0;
      CK(launch_conv3x3_c64(dt, xc, packed_ptr(p, workspace, k + "wf"), params + p->poff.at(pre + "feed_forward.0.bias"), X,
                            (b == p->nb - 1) ? WSP("X0") : nullptr, WSP("X" + std::to_string(b + 1)), B, H, W, st,
                            packed_ptr(p, workspace, k + "wfR"), WSP("zero_page"), variant, stat_partials > 0 ? (float*)WSP("norm_part") : nullptr)); }
  }
  void* Y = WSP("X" + std::to_string(p->nb));
  const int r0 = (s == 4) ? 2 : s;
  const float* wlast = params + p->poff.at(s == 4 ? "tail.6.weight" : "tail.3.weight");
  if (p->stream_tail_x23()) {
    M2TProfScope ps(M2T_PROF_TAIL_FWD_FUSED, st);
    CK(launch_tail_fwd_stream(Y, 1, packed_ptr(p, workspace, "t0"), params + p->poff.at("tail.0.bias"), wlast, (float*)WSP("srpre"), B, H, W, r0, 0, st));
  } else {
  { M2TProfScope ps(M2T_PROF_TAIL_GEMM, st);
    CK(launch_tail_expand(dt, Y, packed_ptr(p, workspace, "t0"), params + p->poff.at("tail.0.bias"), WSP("t1act"), WSP("t1der"), BP, H, W, r0, true, st)); }
  const void* last_act = WSP("t1act");
  if (s == 4 && dt != M2T_F32 && p->use_fused_tail_fwd && p->use_fused_tail_bwd) {
    M2TProfScope ps(M2T_PROF_TAIL_FWD_FUSED, st);
    if (p->use_stream_tail_fwd)
      CK(launch_tail_fwd_stream(WSP("t1act"), 0, packed_ptr(p, workspace, "t3"), params + p->poff.at("tail.3.bias"), wlast, (float*)WSP("srpre"),
                                B, 2 * H, 2 * W, 2, 0, st));
    else
      CK(launch_tail_fwd_fused(WSP("t1act"), packed_ptr(p, workspace, "t3"), params + p->poff.at("tail.3.bias"), wlast, (float*)WSP("srpre"),
                               B, p->Hsp, p->Wsp, st));
  } else {
  if (s == 4) {
    { M2TProfScope ps(M2T_PROF_TAIL_GEMM, st);
      CK(launch_tail_expand(dt, WSP("t1act"), packed_ptr(p, workspace, "t3"), params + p->poff.at("tail.3.bias"), WSP("t2act"), WSP("t2der"),
                            BP * 4, 2 * H, 2 * W, 2, false, st)); }
    last_act = WSP("t2act");
  }
  { M2TProfScope ps(M2T_PROF_FINAL_FWD, st); CK(launch_final_conv_fwd(dt, last_act, wlast, (float*)WSP("srpre"), B, p->Hsp, p->Wsp, st)); }
  }
  }
  if (sr)
    CK(launch_clamp_l1((const float*)WSP("srpre"), nullptr, sr, nullptr, nullptr, nullptr, B, p->Hsp, p->Wsp, p->Hs,
                       p->Ws, rgb_range, 0.f, 0.f, st));
  p->have_acts = true;
  p->have_seed = false;
  p->l1_deferred = false;
  return 0;
}

extern "C" int m2t_l1_loss(m2t_plan* p, const float* hr, float lambda_l1, double divisor, float rgb_range,
                           float* loss_out, void* workspace, void* stream) {
  if (!p || !hr || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_l1_loss: null argument");
  if (!p->have_acts) return m2t_set_error(M2T_ERR_STATE, "m2t_l1_loss: call m2t_forward first");
  const float sc = (float)((double)lambda_l1 / divisor);
  CK(launch_clamp_l1((const float*)WSP("srpre"), hr, nullptr, (float*)WSP("gpre"), (float*)WSP("loss_part"), loss_out,
                     p->B, p->Hsp, p->Wsp, p->Hs, p->Ws, rgb_range, sc, sc, (hipStream_t)stream));
  p->have_seed = true;
  p->l1_deferred = false;
  return 0;
}

// The same loss and seed, produced INSIDE the next m2t_backward: on the bf16 x4 path the clamp + L1 seed are taken by the fused tail
// backward while it stages its g(sr) halo (the pre-clamp output is read there instead of a stored seed: one 150 MB pass and two
// launches fewer per step); everywhere else m2t_backward simply runs m2t_l1_loss's kernel first.  hr must stay valid until then.
extern "C" int m2t_l1_loss_deferred(m2t_plan* p, const float* hr, float lambda_l1, double divisor, float rgb_range,
                                    float* loss_out, void* workspace, void* stream) {
  (void)stream;
  if (!p || !hr || !workspace || !loss_out) return m2t_set_error(M2T_ERR_ARG, "m2t_l1_loss_deferred: null argument");
  if (!p->have_acts) return m2t_set_error(M2T_ERR_STATE, "m2t_l1_loss_deferred: call m2t_forward first");
  p->l1_hr = hr; p->l1_loss_out = loss_out; p->l1_sc = (float)((double)lambda_l1 / divisor); p->l1_R = rgb_range;
  p->l1_deferred = true;
  p->have_seed = true;
  return 0;
}

// upstream gradient -> gradient of the padded pre-clamp output (clamp mask, zero outside the crop)
__global__ void __launch_bounds__(256) seed_from_grad_kernel(const float* __restrict__ pre, const float* __restrict__ gsr,
                                                             float* __restrict__ gpre, int B, int Hp, int Wp, int Hs, int Ws, float R) {
  const long long total = (long long)B * 3 * Hp * Wp;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % Wp);
    long long q = t / Wp;
    const int y = (int)(q % Hp);
    const long long bc = q / Hp;
    float g = 0.f;
    if (y < Hs && x < Ws) {
      const float v = pre[t];
      if (v >= 0.f && v <= R) g = gsr[(bc * Hs + y) * Ws + x];
    }
    gpre[t] = g;
  }
}
extern "C" int m2t_set_output_grad(m2t_plan* p, const float* g_sr, float rgb_range, void* workspace, void* stream) {
  if (!p || !g_sr || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_set_output_grad: null argument");
  if (!p->have_acts) return m2t_set_error(M2T_ERR_STATE, "m2t_set_output_grad: call m2t_forward first");
  const long long total = (long long)p->B * 3 * p->Hsp * p->Wsp;
  hipLaunchKernelGGL(seed_from_grad_kernel, dim3((unsigned)std::min<long long>(ceil_divll(total, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, (const float*)WSP("srpre"), g_sr, (float*)WSP("gpre"), p->B, p->Hsp, p->Wsp,
                     p->Hs, p->Ws, rgb_range);
  M2T_LAUNCH_CHECK();
  p->have_seed = true;
  p->l1_deferred = false;
  return 0;
}

// Two-stream backward.  The data-gradient chain (what the next kernel needs) runs on the
// caller's stream; everything that only produces PARAMETER gradients (weight/bias gradients,
// slab reductions, rel-pos reductions) runs on the plan's side stream, forked/joined with
// events, because neither class of kernel fills 256 CUs on its own at these sizes.
// Hazards are closed explicitly: gqkv/relw are double-buffered and re-used only after the
// side stream's consumer of two branches ago has finished; a block's gy buffer is rewritten
// only after the side stream's conv-wgrad of the following block has read it.
extern "C" int m2t_backward(m2t_plan* p, const float* params, const float* x, float* grads, void* workspace,
                            void* stream) {
  if (!p || !params || !x || !grads || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_backward: null argument");
  if (!p->have_acts || !p->have_seed)
    return m2t_set_error(M2T_ERR_STATE, "m2t_backward: needs m2t_forward and a seed (m2t_l1_loss / m2t_set_output_grad)");
  g_m2t_f32_fast = p->use_fp32_fast;
  struct F32FastGuard { ~F32FastGuard() { g_m2t_f32_fast = 1; } } f32_fast_guard;
  hipStream_t st = (hipStream_t)stream;
  if (p->ensure_side(st) != 0) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: cannot create the side stream / events");
  hipStream_t sd = p->use_side ? p->side : st;
  size_t evi = 0;
  auto next_event = [&]() -> hipEvent_t { return p->events[(evi++) % p->events.size()]; };
  // option "fork_on_kernel": arm_fork() in front of the launch the fork follows; the event then rides on that dispatch as its stop
  // event (no marker packet between the kernel and its successor on the main stream).  fork() records as usual if nothing took it.
  hipEvent_t armed = nullptr;
  // the armed event is thread-local state that the NEXT timed launch of this thread takes: nothing armed by an earlier call (one that
  // returned early between arm_fork() and its launch) may leak into this pass, and nothing armed here may outlive it on any exit path
  g_fork_armed = nullptr;
  struct ForkArmGuard { ~ForkArmGuard() { g_fork_armed = nullptr; } } fork_arm_guard;
  hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cap_status);
  const bool fork_on_kernel = p->fork_on_kernel && cap_status == hipStreamCaptureStatusNone;   // an event-carrying dispatch is not a graph node
  auto arm_fork = [&]() {
    if (sd == st || !fork_on_kernel) return;
    armed = next_event();
    g_fork_armed = armed;
  };
  auto fork = [&]() {            // side stream continues from this point of the main stream
    if (sd == st) return;
    hipEvent_t e;
    if (armed && g_fork_armed == nullptr) e = armed;            // taken by the launch
    else { e = armed ? armed : next_event(); g_fork_armed = nullptr; (void)hipEventRecord(e, st); }
    armed = nullptr;
    (void)hipStreamWaitEvent(sd, e, 0);
  };
  auto side_marker = [&]() -> hipEvent_t {
    if (sd == st) return nullptr;
    hipEvent_t e = next_event();
    (void)hipEventRecord(e, sd);
    return e;
  };
  auto main_wait = [&](hipEvent_t e) { if (e) (void)hipStreamWaitEvent(st, e, 0); };

  const int dt = p->dt, B = p->B, H = p->H, W = p->W, s = p->scale;
  const long long BP = (long long)B * p->P;
  // ---- slab arena + deferred reductions (all on the side stream) ----
  float* arena = (float*)WSP("arena");
  const m2t_red_desc* descs_dev = (const m2t_red_desc*)WSP("red_descs");
  size_t arena_top = 0;
  std::vector<m2t_red_desc> descs;
  size_t flushed = 0;
  bool overflow = false;
  // returns nullptr (and the caller returns M2T_ERR_STATE through ARENA) BEFORE any kernel could write past the arena
  auto arena_alloc = [&](size_t nfloats) -> float* {
    arena_top = (arena_top + 63) & ~(size_t)63;
    if (arena_top + nfloats > p->arena_floats) { overflow = true; return nullptr; }
    float* ptr = arena + arena_top;
    arena_top += nfloats;
    return ptr;
  };
#define ARENA(var, nfloats)                                                                              \
  float* var = arena_alloc(nfloats);                                                                     \
  if (!var) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: slab arena too small for this plan")
  auto defer = [&](const float* slab, long long dst_off, int ns, long long n, int perm, int p0, int p1, int p2) {
    m2t_red_desc d;
    d.src_off = (long long)(slab - arena); d.dst_off = dst_off; d.n = n; d.ns = ns; d.perm = perm; d.p0 = p0; d.p1 = p1; d.p2 = p2; d.pad_ = 0;
    descs.push_back(d);
  };
  size_t bucket_i = 0;
  auto mark_bucket_on = [&](hipStream_t rs) {   // the gradient range of bucket_i is final on stream rs from here on
    if (bucket_i < p->bucket_events.size() && p->red_uploaded) (void)hipEventRecord(p->bucket_events[bucket_i], rs);
    ++bucket_i;
  };
  auto mark_bucket = [&]() { mark_bucket_on(sd); };
  hipStream_t flush_stream = sd;
  auto flush = [&]() -> int {        // one launch reduces everything deferred since the last flush
    if (overflow) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: slab arena too small");
    if (descs.size() > 512) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: too many deferred reductions");
    if (!p->red_uploaded) return 0;  // first call: table not on the device yet, reduced at the end
    const int cnt = (int)(descs.size() - flushed);
    int rc = launch_multi_reduce(arena, grads, descs_dev + flushed, cnt, flush_stream);
    flushed = descs.size();
    return rc;
  };
  float* relp = nullptr;
  const float* gpre = (const float*)WSP("gpre");
  int ns = 0;
  const int r0 = (s == 4) ? 2 : s;
  // ---- tail ----
  const std::string wl = (s == 4) ? "tail.6.weight" : "tail.3.weight";
  const void* last_act = (s == 4) ? WSP("t2act") : WSP("t1act");
  const void* last_der = (s == 4) ? WSP("t2der") : WSP("t1der");
  void* g_last = (s == 4) ? WSP("g_t2pre") : WSP("g_t1pre");
  const bool skip = p->debug_skip_side;
  hipStream_t tws = sd;      // tail weight gradients: side stream (same-box A/B: +0.5 % over the main stream)
  const bool fused_tail = p->use_fused_tail_bwd && s == 4 && dt != M2T_F32;
  const bool stream_x23 = p->stream_tail_x23();
  // a deferred L1 loss (m2t_l1_loss_deferred): inside the fused tail backward where that kernel runs in its recomputing form,
  // otherwise by m2t_l1_loss's own kernel, here -- on the main stream IN FRONT OF THE FIRST FORK: on the unfused tail path the tail
  // conv's weight gradient reads the seed on the side stream, which only orders itself behind what the fork event covers (round 5
  // enqueued it behind the fork: the side stream could read gpre while it was being written, or the previous step's seed)
  const bool l1_in_tail = p->l1_deferred && p->use_fused_l1 && fused_tail && p->use_fused_tail_fwd && !(p->use_stream_tail_bwd && p->use_fused_tail_fwd);
  if (p->l1_deferred && !l1_in_tail)
    CK(launch_clamp_l1((const float*)WSP("srpre"), p->l1_hr, nullptr, (float*)WSP("gpre"), (float*)WSP("loss_part"), p->l1_loss_out,
                       p->B, p->Hsp, p->Wsp, p->Hs, p->Ws, p->l1_R, p->l1_sc, p->l1_sc, st));
  fork();
  hipEvent_t im2col_done = nullptr;           // head_cols is produced on the side stream; the head weight gradient may run on the main one
  if (!skip) { CK(launch_head_im2col(dt, x, WSP("head_cols"), B, p->H0, p->W0, H, W, sd)); im2col_done = side_marker(); }
  if (stream_x23) {
    // x2 / x3: the whole tail backward in one row-streaming launch (k_tail_bwd_stream.hip): g(body output) straight into gT
    const int N0 = 64 * r0 * r0;
    const int nb = tail_bwd_stream_blocks(B, H, W, r0);
    ARENA(swf, (size_t)nb * 32 * 64);
    ARENA(sw0, (size_t)nb * N0 * 64);
    ARENA(sb0, (size_t)nb * N0);
    { M2TProfScope ps(M2T_PROF_FINAL_DGRAD, st);
      CK(launch_tail_bwd_stream(gpre, params + p->poff.at(wl), WSP("X" + std::to_string(p->nb)), nullptr, packed_ptr(p, workspace, "t0T"),
                                params + p->poff.at("tail.0.bias"), WSP("gT"), swf, sw0, sb0, &ns, B, H, W, r0, 1, st)); }
    defer(swf, p->poff.at(wl), ns, 32 * 64, 3, 0, 0, 0);
    defer(sw0, p->poff.at("tail.0.weight"), ns, (long long)N0 * 64, 2, 64, r0 * r0, 64);
    defer(sb0, p->poff.at("tail.0.bias"), ns, N0, 2, 64, r0 * r0, 1);
  } else if (fused_tail) {
    // one pass over the high-resolution tensors (k_tail_bwd.hip): tail conv dgrad + wgrad, GELU', tail.3 dgrad + wgrad
    const bool sbwd = p->use_stream_tail_bwd && p->use_fused_tail_fwd;      // (the streaming form always recomputes)
    const int nb = sbwd ? tail_bwd_stream_blocks(B, 2 * H, 2 * W, 2) : tail_bwd_fused_blocks(B, p->Hsp, p->Wsp);
    ARENA(swf, (size_t)nb * 32 * 64);
    ARENA(sw3, (size_t)nb * 256 * 64);
    ARENA(sb3, (size_t)nb * 256);
    { M2TProfScope ps(M2T_PROF_FINAL_DGRAD, st);
      const bool rc = p->use_fused_tail_fwd;       // the forward did not store gelu(t2) / gelu'(t2): recompute per tile
      if (sbwd)
        CK(launch_tail_bwd_stream(gpre, params + p->poff.at(wl), WSP("t1act"), WSP("t1der"), packed_ptr(p, workspace, "t3T"),
                                  params + p->poff.at("tail.3.bias"), WSP("g_t1pre"), swf, sw3, sb3, &ns, B, 2 * H, 2 * W, 2, 0, st));
      else
      CK(launch_tail_bwd_fused(gpre, params + p->poff.at(wl), rc ? nullptr : WSP("t2act"), rc ? nullptr : WSP("t2der"), WSP("t1act"),
                               WSP("t1der"), packed_ptr(p, workspace, "t3T"), params + p->poff.at("tail.3.bias"), WSP("g_t1pre"), swf, sw3,
                               sb3, &ns, B, p->Hsp, p->Wsp, st, l1_in_tail ? (const float*)WSP("srpre") : nullptr, p->l1_hr,
                               (float*)WSP("loss_part"), p->Hs, p->Ws, p->l1_R, p->l1_sc, p->use_tail_bwd32 ? 32 : 16)); }
    if (l1_in_tail) CK(launch_loss_finish((const float*)WSP("loss_part"), ns, p->l1_sc, p->l1_loss_out, st));
    defer(swf, p->poff.at(wl), ns, 32 * 64, 3, 0, 0, 0);
    defer(sw3, p->poff.at("tail.3.weight"), ns, 256 * 64, 2, 64, 4, 64);
    defer(sb3, p->poff.at("tail.3.bias"), ns, 256, 2, 64, 4, 1);
  } else {
  if (!skip) {
    ARENA(slabs, (size_t)1024 * 32 * 64);
    { M2TProfScope ps(M2T_PROF_FINAL_WGRAD, tws); CK(launch_final_conv_wgrad(dt, gpre, last_act, slabs, &ns, B, p->Hsp, p->Wsp, tws)); }
    defer(slabs, p->poff.at(wl), ns, 32 * 64, 3, 0, 0, 0);
  }
  { M2TProfScope ps(M2T_PROF_FINAL_DGRAD, st); CK(launch_final_conv_dgrad(dt, gpre, params + p->poff.at(wl), last_der, g_last, B, p->Hsp, p->Wsp, st)); }
  if (s == 4) {
    // tail.3: u = t1act W3^T + b3 (t1act = gelu(t1)), shuffled; g(t1) = (g_u W3) * t1der
    fork();
    ARENA(slabs, (size_t)wgrad_slab_count(BP * 4, 256, 64) * 256 * 64);
    ARENA(colp, (size_t)wgrad_slab_count(BP * 4, 256, 64) * 256);
    m2t_wgrad_args wa{};
    if (!skip) {
    wa.G = WSP("g_t2pre"); wa.gmode = M2T_A_UNSHUF; wa.X = WSP("t1act"); wa.ldx = 64; wa.xmode = M2T_A_PLAIN;
    wa.slabs = slabs; wa.bias_slabs = colp; wa.M = BP * 4; wa.N = 256; wa.K = 64; wa.H = 2 * H; wa.Wd = 2 * W; wa.r = 2; wa.C = 64;
    { M2TProfScope ps(M2T_PROF_TAIL_WGRAD, tws); CK(launch_wgrad_tn(dt, wa, &ns, tws)); }
    defer(slabs, p->poff.at("tail.3.weight"), ns, 256 * 64, 2, 64, 4, 64);
    defer(colp, p->poff.at("tail.3.bias"), ns, 256, 2, 64, 4, 1);   // bias gradient rode along in the wgrad kernel
    }
    m2t_gemm_args ga{};
    ga.A = WSP("g_t2pre"); ga.W = packed_ptr(p, workspace, "t3T"); ga.Y = WSP("g_t1pre"); ga.ldy = 64;
    ga.aux = WSP("t1der"); ga.ldaux = 64; ga.M = BP * 4; ga.N = 64; ga.K = 256;
    ga.H = 2 * H; ga.Wd = 2 * W; ga.r = 2; ga.C = 64;
    { M2TProfScope ps(M2T_PROF_TAIL_GEMM, st); CK(launch_gemm_nt(dt, M2T_A_UNSHUF, M2T_E_GELU_GRAD, ga, st)); }
  }
  }
  void* Y = WSP("X" + std::to_string(p->nb));
  if (!stream_x23) {
    const int N0 = 64 * r0 * r0;
    fork();
    ARENA(slabs, (size_t)wgrad_slab_count(BP, N0, 64) * N0 * 64);
    ARENA(colp, (size_t)wgrad_slab_count(BP, N0, 64) * N0);
    m2t_wgrad_args wa{};
    wa.G = WSP("g_t1pre"); wa.gmode = M2T_A_UNSHUF; wa.X = Y; wa.ldx = M2T_LD_P64; wa.xmode = M2T_A_PLAIN;
    wa.slabs = slabs; wa.bias_slabs = colp; wa.M = BP; wa.N = N0; wa.K = 64; wa.H = H; wa.Wd = W; wa.r = r0; wa.C = 64;
    if (!skip) {
    { M2TProfScope ps(M2T_PROF_TAIL_WGRAD, tws); CK(launch_wgrad_tn(dt, wa, &ns, tws)); }
    defer(slabs, p->poff.at("tail.0.weight"), ns, (long long)N0 * 64, 2, 64, r0 * r0, 64);
    defer(colp, p->poff.at("tail.0.bias"), ns, N0, 2, 64, r0 * r0, 1);
    }
    m2t_gemm_args ga{};
    ga.A = WSP("g_t1pre"); ga.W = packed_ptr(p, workspace, "t0T"); ga.Y = WSP("gT"); ga.ldy = M2T_LD_P64;
    ga.M = BP; ga.N = 64; ga.K = N0; ga.H = H; ga.Wd = W; ga.r = r0; ga.C = 64;
    { M2TProfScope ps(M2T_PROF_TAIL_GEMM, st); CK(launch_gemm_nt(dt, M2T_A_UNSHUF, M2T_E_PLAIN, ga, st)); }
  }
  if (fused_tail || stream_x23) fork();     // the reduction (side stream) follows the main-stream producer
  CK(flush());
  mark_bucket();
  // ---- body, last block first.  gy = gradient of X[b+1] ----
  void* gy = WSP("gT");
  void* gnext[2] = {WSP("gA"), WSP("gB")};
  void* gqkv_sets[2][4]; float* relw_sets[2][4]; void* win_sets[2][4];
  for (int i = 0; i < 4; ++i) {
    gqkv_sets[0][i] = WSP("gqkv" + std::to_string(i)); gqkv_sets[1][i] = WSP("gqkv" + std::to_string(i) + "b");
    relw_sets[0][i] = (float*)WSP("relw" + std::to_string(i)); relw_sets[1][i] = (float*)WSP("relw" + std::to_string(i) + "b");
    win_sets[0][i] = WSP("win" + std::to_string(i)); win_sets[1][i] = WSP("win" + std::to_string(i) + "b");
  }
  // Every event recorded on / waited for by the main stream costs a few microseconds of idle between two dependent kernels (the
  // backward's launches sit 4-7 us apart where such an operation lies between them, 0-1 us where none does): the block's side work
  // is therefore released with TWO forks (at the gate, and after the last branch), and its buffers are protected by ONE wait per
  // block -- for the side work of the block two back, which used the same buffer set.
  hipEvent_t block_done[2] = {nullptr, nullptr};
  hipEvent_t conv_done_prev = nullptr;      // side finished reading gy of the previously processed block
  // Side-stream schedule.  The two C = 256 attention kernels that open a block need a whole CU's LDS per workgroup
  // (k_attn_res.hip): any concurrent parameter-gradient kernel starves them until it has drained (measured:
  // 70 us instead of 25 us per launch); the C = 64 / C = 16 attention kernels run 2x slower next to a weight-
  // gradient GEMM.  So the block's side work is GATED behind the attention launch of branch `gate_branch`
  // (default 2 = behind both C = 256 launches): the conv / qkv weight gradients then run under the halo gathers, the
  // C = 64 / C = 16 attention (which lose less than the step gains), the data-gradient GEMMs and the norm backward.  Each branch has its own gqkv / win / relw
  // buffers, so the lag is harmless.
  const bool gated = p->gate_branch >= 0 && sd != st;
  const int gate = p->gate_branch;          // branch index after whose attention launch the block's side work is released
  for (int b = p->nb - 1; b >= 0; --b) {
    const std::string k = "b" + std::to_string(b) + ".";
    const std::string pre = "body." + std::to_string(b) + ".";
    void* X = WSP("X" + std::to_string(b));
    float* mean = (float*)WSP(k + "mean");
    float* rstd = (float*)WSP(k + "rstd");
    void* xc = WSP(k + "xc");
    void* gxc = WSP("gxc");
    void* gn = WSP("gn");
    bool norm_prered = false;                  // the InstanceNorm backward's first reduction stage rode in the C = 16 prep launch
    void* gy_blk = gy;
    void** gqkv_buf = gqkv_sets[b & 1]; float** relw_buf = relw_sets[b & 1]; void** win_buf = win_sets[b & 1];
    main_wait(block_done[b & 1]);              // the side consumers of this buffer set (block b + 2) are done
    std::vector<int> pending;                  // branches whose side work waits for the block's second fork
    // feed_forward conv: weight / bias gradients on the side stream, data gradient on the main one
    auto side_conv = [&]() -> int {
      if (skip) return 0;
      ARENA(slabs, (size_t)256 * 9 * 64 * 64);
      ARENA(colp, (size_t)256 * 64);
      { M2TProfScope ps(M2T_PROF_CONV3_WGRAD, sd); CK(launch_conv3x3_c64_wgrad(dt, xc, gy_blk, slabs, colp, &ns, B, H, W, sd)); }
      defer(slabs, p->poff.at(pre + "feed_forward.0.weight"), ns, 9 * 64 * 64, 1, 64, 64, 0);
      defer(colp, p->poff.at(pre + "feed_forward.0.bias"), ns, 64, 0, 0, 0, 0);     // bias gradient rode along
      return 0;
    };
    auto fused_dgrad = [&](int i) -> bool {    // projection data gradient inside the attention backward kernel (k_attn_res.hip)
      return dt != M2T_F32 && p->use_fused_qkv_dgrad && p->use_resident_attn_bwd && BR_C[i] >= 64;
    };
    auto side_branch = [&](int i) -> int {     // qkv weight gradient + rel-pos partial reduction of branch i
      if (skip) return 0;
      const int C = BR_C[i], L = BR_L[i];
      const int h = H >> L, w = W >> L;
      const long long M = (long long)B * h * w;
      const std::string an = pre + "attn" + std::to_string(i + 1) + ".";
      ARENA(slabs, (size_t)wgrad_slab_count(M, 3 * C, C) * 3 * C * C);
      // fused data gradient: the main chain never reads gqkv, so the overlap-add of dK|dV happens here, off the critical path
      if (fused_dgrad(i)) CK(launch_halo_gather(dt, win_buf[i], gqkv_buf[i], B, h, w, 2 * C, 3 * C, C, sd));
      m2t_wgrad_args wa{};
      wa.G = gqkv_buf[i]; wa.ldg = 3 * C; wa.gmode = M2T_A_PLAIN; wa.X = WSP(k + "d" + std::to_string(i + 1)); wa.ldx = C; wa.xmode = M2T_A_PLAIN;
      wa.slabs = slabs; wa.M = M; wa.N = 3 * C; wa.K = C; wa.H = h; wa.Wd = w; wa.r = 1; wa.C = C; wa.halo_win = win_buf[i];
      wa.big_tiles = p->wgrad_big_tiles >= 0 ? p->wgrad_big_tiles : (M >= 24576 ? 256 : 0);
      { M2TProfScope ps(M2T_PROF_WGRAD_QKV, sd); CK(launch_wgrad_tn(dt, wa, &ns, sd)); }
      defer(slabs, p->poff.at(an + "qkv_conv.weight"), ns, 3LL * C * C, 0, 0, 0, 0);
      ARENA(relp, (size_t)32 * 10 * C);
      int nsp = 0;
      CK(launch_rel_reduce1(relw_buf[i], relp, (int)(M / 64), C, &nsp, sd));
      defer(relp, p->poff.at(an + "rel_h"), nsp, 10LL * C, 4, C, 0, 0);     // rel_h then rel_w are adjacent parameters
      return 0;
    };
    hipEvent_t conv_done = nullptr;
    // bf16: both gradients in one pass over gy on the main stream (the partials still reduce on the side stream: every block
    // forks at least once after this launch and before its flush)
    const bool fuse_conv = dt != M2T_F32 && p->use_fused_conv_bwd && !skip && conv3x3_c64_bwd_fusable(B, H, W);
    if (fuse_conv) {
      ARENA(slabs, (size_t)256 * 9 * 64 * 64);
      ARENA(colp, (size_t)256 * 64);
      { M2TProfScope ps(M2T_PROF_CONV3_BWD, st);
        CK(launch_conv3x3_c64_bwd_fused(gy, xc, packed_ptr(p, workspace, k + "wfTR"), gxc, slabs, colp, &ns, WSP("zero_page"), B, H, W, st)); }
      defer(slabs, p->poff.at(pre + "feed_forward.0.weight"), ns, 9 * 64 * 64, 6, 64, 64, 0);    // slabs come as [tap][ic][oc]
      defer(colp, p->poff.at(pre + "feed_forward.0.bias"), ns, 64, 0, 0, 0, 0);
    } else if (!gated) {
      fork();
      CK(side_conv());
      conv_done = side_marker();
    }
    if (!fuse_conv)
    { M2TProfScope ps(M2T_PROF_CONV3_DGRAD, st); CK(launch_conv3x3_c64(dt, gy, packed_ptr(p, workspace, k + "wfT"), nullptr, nullptr, nullptr, gxc, B, H, W, st,
                                                                        packed_ptr(p, workspace, k + "wfTR"), WSP("zero_page"), p->use_conv_rows ? 0 : 1)); }
    for (int i = 3; i >= 0; --i) {
      const int C = BR_C[i], L = BR_L[i];
      const int h = H >> L, w = W >> L;
      const long long M = (long long)B * h * w;
      const std::string an = pre + "attn" + std::to_string(i + 1) + ".";
      const void* qkv = WSP(k + "qkv" + std::to_string(i + 1));
      const float* rh = params + p->poff.at(an + "rel_h");
      const float* rw = params + p->poff.at(an + "rel_w");
      void* gqkv = gqkv_buf[i];
      void* win = win_buf[i];
      float* relw = relw_buf[i];
      // gradient of IWT^L is DWT^L: applied while the kernel loads g_xc[chunk i]
      // dK|dV stay window-major in `win`; the fused tail kernel gathers them once per row, writes them back
      // into gqkv for the weight-gradient GEMM, multiplies by Wqkv and applies IWT / branch mixing.
      // (gathering inside the TILED GEMM / wgrad loaders, M2T_A_HALO, was measured slower: the gather is then
      //  repeated once per column-block.)
      const void* gxc_i = (const char*)gxc + (size_t)i * BP * 16 * p->esz;       // chunk i of the P64 gradient: a dense plane
      // bf16 C = 16 with "attn_bwd" = 3: the overlap-add, the projection data gradient and branch_prep_bwd are one kernel behind
      // the attention backward; it completes dK|dV in gqkv, so the branch's side work is released after it
      const bool c16_prep = dt != M2T_F32 && C == 16 && p->use_c16_prep && p->use_fused_qkv_dgrad && p->use_resident_attn_bwd;
      if (!gated && fused_dgrad(i)) arm_fork();
      // branch 4's branch_prep_bwd runs inside branch 3's attention backward (same level, same window grid): branch 4 then leaves its
      // own-window g_d rows in the second buffer set, and its prep launch is skipped below
      const bool pb_consumer = i == 2 && fused_dgrad(2) && fused_dgrad(3) && p->use_fused_prep_bwd && BR_L[3] == BR_L[2];
      const bool pb_producer = i == 3 && fused_dgrad(2) && fused_dgrad(3) && p->use_fused_prep_bwd && BR_L[3] == BR_L[2];
      if (fused_dgrad(i)) {
        M2TProfScope ps(C == 64 ? M2T_PROF_ATTN_BWD_64 : M2T_PROF_ATTN_BWD_256, st);
        const bool rc64 = C == 64 && p->c64_recompute();
        CK(launch_window_attn_bwd_resident(qkv, rh, rw, gxc_i, 16, 0, gqkv, win, relw, B, h, w, C, L, st,
                                           packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "TF"),
                                           pb_producer ? WSP("gd2") : WSP("gd"), pb_producer ? WSP("gdwin2") : WSP("gdwin"),
                                           rc64 ? WSP(k + "d" + std::to_string(i + 1)) : nullptr,
                                           rc64 ? packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "F") : nullptr,
                                           pb_consumer ? WSP("gd2") : nullptr, pb_consumer ? WSP("gdwin2") : nullptr,
                                           pb_consumer ? (const void*)((const char*)gxc + (size_t)(i + 1) * BP * 16 * p->esz) : nullptr,
                                           pb_consumer ? (void*)((char*)gn + (size_t)(i + 1) * BP * 16 * p->esz) : nullptr));
      } else if (C == 16 && p->c16_recompute()) {
        // qkv1 was not stored: recomputed inside the kernel from d1 (identical bits); then the halo overlap-add as usual
        { M2TProfScope ps(M2T_PROF_ATTN_BWD_16, st);
          CK(launch_window_attn_bwd_c16(nullptr, rh, rw, gxc_i, 16, 0, gqkv, win, relw, B, h, w, st, WSP(k + "d1"), packed_ptr(p, workspace, k + "w1"))); }
        if (!c16_prep) CK(launch_halo_gather(dt, win, gqkv, B, h, w, 2 * C, 3 * C, C, st));
      } else {
        CK(launch_window_attn_bwd(dt, qkv, rh, rw, gxc_i, 16, 0, gqkv, win, relw, B, h, w, C, st, L, !c16_prep, p->use_resident_attn_bwd));
      }
      auto release_side = [&]() -> int {
        if (!gated) {
          fork();
          CK(side_branch(i));
          if (i == 0) block_done[b & 1] = side_marker();
        } else if (i == gate) {
          fork();                              // the gate: the LDS-hungry attention kernels of this block are on their way
          // gated branches first, then the block's conv weight gradient (512 threads, 80 KB of LDS: +1.9 % over putting it
          // first, where it met the C = 64 attention)
          for (int j = 3; j >= gate; --j) CK(side_branch(j));
          if (!fuse_conv) {
            CK(side_conv());
            conv_done = side_marker();
          }
          if (i == 0) block_done[b & 1] = side_marker();
        } else if (i < gate) {
          pending.push_back(i);                // released together behind the last branch: one fork instead of one per branch
          if (i == 0) {
            fork();
            for (int j : pending) CK(side_branch(j));
            block_done[b & 1] = side_marker();
          }
        }
        return 0;
      };
      if (!c16_prep) CK(release_side());
      if (pb_producer) {
        // (nothing: the next attention backward applies this branch's branch_prep_bwd while it loads its output gradient)
      } else if (fused_dgrad(i)) {
        // own-window products are in gd; add the ring rows of the (<= 3) neighbouring windows to the border pixels
        CK(launch_branch_prep_bwd(dt, L, WSP("gd"), gxc, gn, i, B, H, W, st, WSP("gdwin")));
      } else if (c16_prep) {
        if (!gated) arm_fork();
        if (p->use_fused_norm_red) {
          CK(launch_c16_dgrad_prep(gqkv, win, packed_ptr(p, workspace, k + "w1T"), gxc, gn, B, H, W, st, X, mean, rstd,
                                   (float*)WSP("norm_part"), (float*)WSP("norm_part0")));
          norm_prered = true;
        } else {
          CK(launch_c16_dgrad_prep(gqkv, win, packed_ptr(p, workspace, k + "w1T"), gxc, gn, B, H, W, st));
        }
        CK(release_side());
      } else {
        m2t_gemm_args ga{};
        ga.A = gqkv; ga.lda = 3 * C; ga.W = packed_ptr(p, workspace, k + "w" + std::to_string(i + 1) + "T");
        ga.Y = WSP("gd"); ga.ldy = C; ga.M = M; ga.N = C; ga.K = 3 * C; ga.H = h; ga.Wd = w; ga.r = 1; ga.C = C; ga.halo_win = win;
        { M2TProfScope ps(M2T_PROF_GEMM_QKV_DGRAD, st); CK(launch_gemm_nt(dt, M2T_A_PLAIN, M2T_E_PLAIN, ga, st)); }
        CK(launch_branch_prep_bwd(dt, L, WSP("gd"), gxc, gn, i, B, H, W, st));
      }
    }
    // block 0: the head's g(res) = g(X0) + g(Y) joins in the same pass and lands where the head weight gradient reads it
    void* gx = (b == 0) ? WSP("gxc") : gnext[b & 1];
    // gx's buffer was the gy of block b+1: its conv-wgrad / colsum on the side stream must be done
    main_wait(conv_done_prev);
    CK(launch_instnorm_bwd(dt, gn, X, mean, rstd, gy, gx, (float*)WSP("norm_part"), (float*)WSP("norm_s"), B, (int)p->P, st,
                           norm_prered ? (const float*)WSP("norm_part0") : nullptr, H * (W / 16), (b == 0) ? WSP("gT") : nullptr));
    conv_done_prev = conv_done;
    gy = gx;
    if ((b & 1) == 0) { CK(flush()); mark_bucket(); }
    else if (b == 1) CK(flush());      // (the step's last pair: its first half is reduced under block 0, so that what is left after the
                                       //  last data-gradient kernel is short -- the main stream idles until it is done)
  }
  // head: g(res) = g(X0) from the chain + g(Y) from `res + x`: added inside block 0's InstanceNorm backward (it wrote gxc)
  if (p->nb == 0) CK(launch_add(dt, gy, WSP("gT"), WSP("gxc"), BP * 64, st));
  // The end of the step is a serial chain: head weight gradient -> its reduction -> (the caller's) Adam.  In steady state it runs on the
  // MAIN stream behind the last data-gradient kernel: handing it to the side stream and back cost two cross-stream waits and a
  // queue position behind the last block pair's reduction (96 us between the last backward kernel and Adam, measured).
  const bool tail_on_main = p->red_uploaded && sd != st;
  hipStream_t hs = tail_on_main ? st : sd;
  if (!tail_on_main) fork();
  if (!skip) {
    // head conv: im2col'd input (made at the start of this backward, off the critical path) x output gradient
    const int nsl = wgrad_slab_count(BP, 64, 32);
    ARENA(slabs, (size_t)nsl * 64 * 32);
    ARENA(colp, (size_t)nsl * 64);
    m2t_wgrad_args wa{};
    wa.G = WSP("gxc"); wa.ldg = M2T_LD_P64; wa.gmode = M2T_A_PLAIN; wa.X = WSP("head_cols"); wa.ldx = 32; wa.xmode = M2T_A_PLAIN;
    wa.slabs = slabs; wa.bias_slabs = colp; wa.M = BP; wa.N = 64; wa.K = 32; wa.H = H; wa.Wd = W; wa.r = 1; wa.C = 64;
    if (hs == st) main_wait(im2col_done);      // (long since complete; the wait closes the hazard for every n_blocks)
    CK(launch_wgrad_tn(dt, wa, &ns, hs));
    defer(slabs, p->poff.at("head.weight"), ns, 64 * 32, 5, 32, 27, 0);
    defer(colp, p->poff.at("head.bias"), ns, 64, 0, 0, 0, 0);
  }
  const bool first_backward = !p->red_uploaded;
  if (!p->red_uploaded) {
    // first backward of this plan: publish the (step-invariant) descriptor table, then reduce everything
    if (overflow) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: slab arena too small");
    if (descs.size() > 512) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: too many deferred reductions");
    p->red_descs = descs;
    hipError_t e = hipMemcpyAsync(WSP("red_descs"), p->red_descs.data(), descs.size() * sizeof(m2t_red_desc), hipMemcpyHostToDevice, sd);
    if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
    p->red_uploaded = true;
    flushed = 0;
  } else if (p->red_descs.size() != descs.size()) {
    return m2t_set_error(M2T_ERR_STATE, "m2t_backward: reduction table changed between steps");
  }
  if (tail_on_main && !first_backward) {
    main_wait(side_marker());        // join first: everything the side stream still reduces; then the head's reduction on this stream
    flush_stream = st;
    CK(flush());
    mark_bucket_on(st);
  } else {
    CK(flush());
    mark_bucket();
  }
  if (first_backward)                // nothing was reduced before this point: every bucket completes here
    for (auto e : p->bucket_events) (void)hipEventRecord(e, sd);
  if (bucket_i != p->buckets.size()) return m2t_set_error(M2T_ERR_STATE, "m2t_backward: gradient bucket table out of step");
  if (!(tail_on_main && !first_backward)) main_wait(side_marker());          // join: every gradient is complete in main-stream order
  p->have_seed = false;
  p->l1_deferred = false;
  return 0;
}

extern "C" int m2t_stream_wait_bucket(m2t_plan* p, int bucket, void* stream) {
  if (!p || bucket < 0 || (size_t)bucket >= p->buckets.size()) return m2t_set_error(M2T_ERR_ARG, "m2t_stream_wait_bucket: bad bucket");
  if (p->bucket_events.size() != p->buckets.size()) return m2t_set_error(M2T_ERR_STATE, "m2t_stream_wait_bucket: call m2t_backward first");
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, p->bucket_events[bucket], 0);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  return 0;
}

extern "C" int m2t_set_option(m2t_plan* p, const char* key, long long value) {
  if (!p || !key) return m2t_set_error(M2T_ERR_ARG, "m2t_set_option: null");
  p->red_uploaded = false;     // the deferred-reduction table depends on the schedule: rebuild it on the next backward
  const std::string k(key);
  if (k == "side_stream") { p->use_side = (value != 0); return 0; }
  if (k == "fork_on_kernel") { p->fork_on_kernel = value != 0; return 0; }
  if (k == "fused_prep_fwd") { p->use_fused_prep_fwd = value != 0; return 0; }
  if (k == "fused_prep_bwd") { p->use_fused_prep_bwd = value != 0; return 0; }
  if (k == "fused_norm_red") { p->use_fused_norm_red = value != 0; return 0; }
  if (k == "fused_l1") { p->use_fused_l1 = value != 0; return 0; }
  if (k == "tail_bwd_mfma32") { p->use_tail_bwd32 = value != 0; return 0; }
  if (k == "fp32_fast") { p->use_fp32_fast = value != 0; return 0; }
  if (k == "fused_attn_fwd2") { if (value < -1 || value > 2) return m2t_set_error(M2T_ERR_ARG, "fused_attn_fwd2: -1 .. 2"); p->fused_attn_fwd2 = (int)value; return 0; }
  if (k == "gate_branch") { if (value < -1 || value > 3) return m2t_set_error(M2T_ERR_ARG, "gate_branch: -1..3"); p->gate_branch = (int)value; return 0; }
  if (k == "wgrad_big_tiles") { p->wgrad_big_tiles = (int)value; return 0; }
  if (k == "fused_tail") {
    if (value < 0 || value > 4) return m2t_set_error(M2T_ERR_ARG, "fused_tail: 0..4");
    p->use_fused_tail_bwd = value >= 1; p->use_fused_tail_fwd = value >= 2; p->use_stream_tail_fwd = value >= 3; p->use_stream_tail_bwd = value == 4;
    p->have_acts = false; return 0;
  }
  if (k == "attn_bwd") {
    if (value < 0 || value > 3) return m2t_set_error(M2T_ERR_ARG, "attn_bwd: 0..3");
    // (which of qkv1 / qkv2 the forward stores depends on this option: activations of a forward run under another value are unusable)
    p->use_resident_attn_bwd = value >= 1; p->use_fused_qkv_dgrad = value >= 2; p->use_c16_prep = value == 3; p->have_acts = false; return 0;
  }
  if (k == "conv_rows") { if (value < 0 || value > 1) return m2t_set_error(M2T_ERR_ARG, "conv_rows: 0 / 1"); p->use_conv_rows = (int)value; return 0; }
  if (k == "fused_conv_bwd") { p->use_fused_conv_bwd = (value != 0); return 0; }
  if (k == "fused_attn_fwd") { if (value < 0 || value > 2) return m2t_set_error(M2T_ERR_ARG, "fused_attn_fwd: 0..2"); p->use_fused_attn_fwd = (int)value; p->have_acts = false; return 0; }
  if (k == "fused_c16_fwd") { if (value < 0 || value > 2) return m2t_set_error(M2T_ERR_ARG, "fused_c16_fwd: 0..2"); p->use_fused_c16_fwd = (int)value; p->have_acts = false; return 0; }
  if (k == "debug_skip_side") { p->debug_skip_side = (value != 0); return 0; }
  return m2t_set_error(M2T_ERR_ARG, "m2t_set_option: unknown key");
}

extern "C" int m2t_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                             float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || n <= 0 || step < 1)
    return m2t_set_error(M2T_ERR_ARG, "m2t_adam_step: bad argument");
  return launch_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale, (hipStream_t)stream);
}

// ---- stand-alone operators ---------------------------------------------------------------
extern "C" int m2t_dwt(int dtype, int levels, const void* src, void* dst, int B, int H, int W, int C, void* stream) {
  return launch_dwt(dtype, levels, src, C, 0, dst, C << (2 * levels), 0, B, H, W, C, false, (hipStream_t)stream);
}
extern "C" int m2t_iwt(int dtype, int levels, const void* src, void* dst, int B, int H, int W, int C, void* stream) {
  // src [B][H/S][W/S][C * 4^l] -> dst [B][H][W][C]
  return launch_dwt(dtype, levels, src, C << (2 * levels), 0, dst, C, 0, B, H, W, C, true, (hipStream_t)stream);
}
extern "C" int m2t_pixel_shuffle(const float* in, float* out, int B, int C, int H, int W, int r, void* stream) {
  return launch_pixel_shuffle_nchw(in, out, B, C, H, W, r, 0, (hipStream_t)stream);
}
extern "C" int m2t_pixel_unshuffle(const float* in, float* out, int B, int C, int H, int W, int r, void* stream) {
  // in [B][C][H*r][W*r] -> out [B][C*r*r][H][W]
  return launch_pixel_shuffle_nchw(in, out, B, C, H, W, r, 1, (hipStream_t)stream);
}
extern "C" int m2t_to_nhwc(int dtype, const float* nchw, void* nhwc, int B, int C, int HW, void* stream) {
  return launch_layout(dtype, nchw, nhwc, nullptr, B, C, HW, 0, (hipStream_t)stream);
}
extern "C" int m2t_to_nchw(int dtype, const void* nhwc, float* nchw, int B, int C, int HW, void* stream) {
  return launch_layout(dtype, nullptr, const_cast<void*>(nhwc), nchw, B, C, HW, 1, (hipStream_t)stream);
}
extern "C" int m2t_window_attention_fwd(int dtype, const void* qkv, const float* rel_h, const float* rel_w, void* out,
                                        int B, int h, int w, int C, void* stream) {
  return launch_window_attn_fwd(dtype, qkv, rel_h, rel_w, out, C, 0, nullptr, 0, B, h, w, C, (hipStream_t)stream);
}
extern "C" size_t m2t_window_attention_bwd_scratch_bytes(int dtype, int B, int h, int w, int C) {
  const size_t es = (dtype == M2T_F32) ? 4 : 2;
  const size_t nwin = (size_t)B * (h / 8) * (w / 8);
  return ((nwin * 100 * 2 * C * es + 255) & ~(size_t)255) + ((nwin * 10 * C * 4 + 255) & ~(size_t)255) + (size_t)32 * 10 * C * 4;
}
extern "C" int m2t_window_attention_bwd(int dtype, const void* qkv, const float* rel_h, const float* rel_w,
                                        const void* gout, void* gqkv, float* grel_h, float* grel_w, void* scratch, int B,
                                        int h, int w, int C, void* stream) {
  const size_t es = (dtype == M2T_F32) ? 4 : 2;
  const size_t nwin = (size_t)B * (h / 8) * (w / 8);
  const size_t woff = (nwin * 100 * 2 * C * es + 255) & ~(size_t)255;
  const size_t roff = woff + ((nwin * 10 * C * 4 + 255) & ~(size_t)255);
  int rc = launch_window_attn_bwd(dtype, qkv, rel_h, rel_w, gout, C, 0, gqkv, scratch, (float*)((char*)scratch + woff),
                                  B, h, w, C, (hipStream_t)stream, 0, true);
  if (rc) return rc;
  return launch_rel_reduce((float*)((char*)scratch + woff), (float*)((char*)scratch + roff), grel_h, grel_w, (int)nwin, C,
                           (hipStream_t)stream);
}
