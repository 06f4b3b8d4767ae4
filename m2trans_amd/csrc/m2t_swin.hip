// m2t_swin.hip -- C ABI of the MedCLIP image tower (Swin-T 224) + the SemanticLoss value
// (include/m2t.h, "SemanticLoss" section).  Restates the arithmetic behind losses.py:53-79;
// forward only (the reference runs it under torch.no_grad()).  Host code; kernels in k_swin.hip
// and k_gemm.hip.
#include <map>
#include <string>
#include <vector>
#include "m2t_kernels.h"
#include <cstring>
#include "../../include/m2t.h"

namespace {
const int DEPTHS[4] = {2, 2, 6, 2};
const int HEADS[4] = {3, 6, 12, 24};
struct Ws { size_t off, n; };
}

struct m2t_swin {
  int max_images, dt;
  size_t esz;
  std::vector<std::string> pnames;
  std::map<std::string, long long> poff, pnum;
  long long nparams = 0;
  std::map<std::string, long long> pk;     // packed (T) offsets in elements
  long long npacked = 0;
  std::map<std::string, long long> fb;     // fp32 side buffers (fused qkv biases) offsets in floats
  long long nfb = 0;
  std::map<std::string, Ws> ws;
  size_t ws_bytes = 0;
  const float* weights = nullptr;          // caller's flat fp32 weights (device), set by load_weights
  // crop tables travel through a small ring of PINNED host slots, so the upload is a true asynchronous copy and the call returns
  // without waiting for the stream (round 3: the hipStreamSynchronize that protected a pageable temporary made every
  // SemanticLoss step wait for the model's forward pass -- 8 of the 11 ms of configs[2] were host stall)
  static constexpr int NPIN = 8;
  int* pinned = nullptr;                   // [NPIN][3 * max_images]
  hipEvent_t pin_ev[NPIN] = {};            // slot k's copy has been consumed
  bool pin_used[NPIN] = {};
  int pin_next = 0;
  bool fused_mlp = true;                   // bf16, stages 1 / 2: LayerNorm + fc1 + GELU + fc2 + residual in one kernel (k_swin.hip)
  void add_param(const std::string& n, long long c) { pnames.push_back(n); poff[n] = nparams; pnum[n] = c; nparams += c; }
  void add_pack(const std::string& n, long long c) { npacked = (npacked + 7) & ~7LL; pk[n] = npacked; npacked += c; }
  void add_ws(const std::string& n, size_t elems, size_t es) {
    ws_bytes = (ws_bytes + 255) & ~(size_t)255;
    ws[n] = Ws{ws_bytes, elems};
    ws_bytes += elems * es;
  }
};

extern "C" int m2t_swin_create(m2t_swin** out, int max_images, int dtype) {
  if (!out || max_images < 1 || (dtype != M2T_F32 && dtype != M2T_BF16)) return m2t_set_error(M2T_ERR_ARG, "m2t_swin_create: bad argument");
  m2t_swin* p = new m2t_swin();
  p->max_images = max_images; p->dt = dtype; p->esz = (dtype == M2T_F32) ? 4 : 2;
  // parameter inventory: HF swin-tiny checkpoint names (transformers 4.24), + the MedCLIP projection
  p->add_param("embeddings.patch_embeddings.projection.weight", 96 * 48);
  p->add_param("embeddings.patch_embeddings.projection.bias", 96);
  p->add_param("embeddings.norm.weight", 96);
  p->add_param("embeddings.norm.bias", 96);
  p->add_pack("pe", 96 * 48);
  for (int s = 0; s < 4; ++s) {
    const long long C = 96LL << s;
    for (int j = 0; j < DEPTHS[s]; ++j) {
      const std::string b = "encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(j) + ".";
      p->add_param(b + "layernorm_before.weight", C);
      p->add_param(b + "layernorm_before.bias", C);
      for (const char* nm : {"query", "key", "value"}) {
        p->add_param(b + "attention.self." + nm + ".weight", C * C);
        p->add_param(b + "attention.self." + nm + ".bias", C);
      }
      p->add_param(b + "attention.self.relative_position_bias_table", 169LL * HEADS[s]);
      p->add_param(b + "attention.output.dense.weight", C * C);
      p->add_param(b + "attention.output.dense.bias", C);
      p->add_param(b + "layernorm_after.weight", C);
      p->add_param(b + "layernorm_after.bias", C);
      p->add_param(b + "intermediate.dense.weight", 4 * C * C);
      p->add_param(b + "intermediate.dense.bias", 4 * C);
      p->add_param(b + "output.dense.weight", 4 * C * C);
      p->add_param(b + "output.dense.bias", C);
      p->add_pack(b + "qkv", 3 * C * C);
      p->add_pack(b + "o", C * C);
      p->add_pack(b + "fc1", 4 * C * C);
      p->add_pack(b + "fc2", 4 * C * C);
      if (s < 2) { p->add_pack(b + "fc1F", 4 * C * C); p->add_pack(b + "fc2F", 4 * C * C); }   // MFMA fragment order: fused MLP of stages 1 / 2
      p->fb[b + "qkv_bias"] = p->nfb; p->nfb += 3 * C;
    }
    if (s < 3) {
      const std::string d = "encoder.layers." + std::to_string(s) + ".downsample.";
      p->add_param(d + "norm.weight", 4 * C);
      p->add_param(d + "norm.bias", 4 * C);
      p->add_param(d + "reduction.weight", 8 * C * C);
      p->add_pack(d + "red", 8 * C * C);
    }
  }
  p->add_param("layernorm.weight", 768);
  p->add_param("layernorm.bias", 768);
  p->add_param("projection_head.weight", 512 * 768);
  const size_t n = (size_t)max_images, es = p->esz;
  const size_t tok = n * 3136 * 96;             // elements of the widest token tensor at every stage
  p->add_ws("packed", (size_t)p->npacked, es);
  p->add_ws("fbias", (size_t)p->nfb, 4);
  p->add_ws("crops", n * 3, 4);
  p->add_ws("A0", n * 3136 * 48, es);
  p->add_ws("X", tok, es);
  p->add_ws("Hn", tok, es);
  p->add_ws("QKV", tok * 3, es);
  p->add_ws("AO", tok, es);
  p->add_ws("MH", tok * 4, es);
  p->add_ws("emb", n * 512, 4);
  p->ws_bytes = (p->ws_bytes + 255) & ~(size_t)255;
  *out = p;
  return 0;
}
extern "C" void m2t_swin_destroy(m2t_swin* p) {
  if (!p) return;
  if (p->pinned) {
    (void)hipHostFree(p->pinned);
    for (auto& e : p->pin_ev) if (e) (void)hipEventDestroy(e);
  }
  delete p;
}
extern "C" long long m2t_swin_query(const m2t_swin* p, const char* key) {
  if (!p || !key) return -1;
  const std::string k(key);
  if (k == "workspace_bytes") return (long long)p->ws_bytes;
  if (k == "num_params") return p->nparams;
  if (k == "num_param_tensors") return (long long)p->pnames.size();
  if (k == "max_images") return p->max_images;
  if (k.rfind("param:", 0) == 0) { auto it = p->poff.find(k.substr(6)); return it == p->poff.end() ? -1 : it->second; }
  if (k.rfind("numel:", 0) == 0) { auto it = p->pnum.find(k.substr(6)); return it == p->pnum.end() ? -1 : it->second; }
  if (k.rfind("name:", 0) == 0) return -1;
  if (k.rfind("ws:", 0) == 0) { auto it = p->ws.find(k.substr(3)); return it == p->ws.end() ? -1 : (long long)it->second.off; }
  return -1;
}
// i-th parameter name in flat order (so the Python side never duplicates the inventory)
extern "C" const char* m2t_swin_param_name(const m2t_swin* p, int i) {
  if (!p || i < 0 || i >= (int)p->pnames.size()) return nullptr;
  return p->pnames[i].c_str();
}

#define SWP(name) ((char*)workspace + p->ws.at(name).off)
#define CKS(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)
static inline char* spk(const m2t_swin* p, void* workspace, const std::string& k) {
  return (char*)workspace + p->ws.at("packed").off + p->pk.at(k) * p->esz;
}

// one-time: convert the frozen fp32 weights to the element type / fused layouts the kernels read
extern "C" int m2t_swin_load_weights(m2t_swin* p, const float* weights, void* workspace, void* stream) {
  if (!p || !weights || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_swin_load_weights: null");
  hipStream_t st = (hipStream_t)stream;
  const int dt = p->dt;
  p->weights = weights;
  float* fbias = (float*)SWP("fbias");
  CKS(launch_convert(dt, weights + p->poff.at("embeddings.patch_embeddings.projection.weight"), spk(p, workspace, "pe"), 96 * 48, st));
  for (int s = 0; s < 4; ++s) {
    const long long C = 96LL << s;
    for (int j = 0; j < DEPTHS[s]; ++j) {
      const std::string b = "encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(j) + ".";
      int part = 0;
      for (const char* nm : {"query", "key", "value"}) {
        CKS(launch_convert(dt, weights + p->poff.at(b + "attention.self." + nm + ".weight"),
                           spk(p, workspace, b + "qkv") + (size_t)part * C * C * p->esz, C * C, st));
        CKS(launch_convert(M2T_F32, weights + p->poff.at(b + "attention.self." + nm + ".bias"),
                           fbias + p->fb.at(b + "qkv_bias") + part * C, C, st));
        ++part;
      }
      CKS(launch_convert(dt, weights + p->poff.at(b + "attention.output.dense.weight"), spk(p, workspace, b + "o"), C * C, st));
      CKS(launch_convert(dt, weights + p->poff.at(b + "intermediate.dense.weight"), spk(p, workspace, b + "fc1"), 4 * C * C, st));
      CKS(launch_convert(dt, weights + p->poff.at(b + "output.dense.weight"), spk(p, workspace, b + "fc2"), 4 * C * C, st));
      if (s < 2) {
        CKS(launch_frag16_pack(dt, weights + p->poff.at(b + "intermediate.dense.weight"), spk(p, workspace, b + "fc1F"), (int)(4 * C), (int)C, st));
        CKS(launch_frag16_pack(dt, weights + p->poff.at(b + "output.dense.weight"), spk(p, workspace, b + "fc2F"), (int)C, (int)(4 * C), st));
      }
    }
    if (s < 3) {
      const std::string d = "encoder.layers." + std::to_string(s) + ".downsample.";
      CKS(launch_convert(dt, weights + p->poff.at(d + "reduction.weight"), spk(p, workspace, d + "red"), 8 * C * C, st));
    }
  }
  return 0;
}

static int swin_gemm(int dt, int emode, const void* A, int K, const void* W, void* Y, int N, long long M, const float* bias,
                     const void* aux, hipStream_t st) {
  m2t_gemm_args ga{};
  ga.A = A; ga.lda = K; ga.W = W; ga.Y = Y; ga.ldy = N; ga.bias = bias; ga.aux = aux; ga.ldaux = N;
  ga.M = M; ga.N = N; ga.K = K; ga.H = 1; ga.Wd = 1; ga.r = 1; ga.C = 64;
  return launch_gemm_nt(dt, M2T_A_PLAIN, emode, ga, st);
}

// encode n (<= max_images) 224x224 crops: src [n_src][3][Hs][Ws] fp32 NCHW (device), crops_host [n][3] =
// (source index, y0, x0) -> emb [n][512] (unit norm, fp32, device)
extern "C" int m2t_swin_encode(m2t_swin* p, const float* src, int n_src, int Hs, int Ws, const int* crops_host, int n,
                               float* emb, void* workspace, void* stream) {
  return m2t_swin_encode_pair(p, src, n_src, nullptr, 0, Hs, Ws, crops_host, n, emb, workspace, stream);
}

// the same with the source images in TWO tensors (indices 0 .. n_a - 1 in src_a, n_a .. n_a + n_b - 1 in src_b): the SR and
// HR batches of SemanticLoss.batch are encoded in one pass without concatenating them first
extern "C" int m2t_swin_encode_pair(m2t_swin* p, const float* src, int n_a, const float* src_b, int n_b, int Hs, int Ws,
                                    const int* crops_host, int n, float* emb, void* workspace, void* stream) {
  const int n_src = n_a + n_b;
  if (!p || !src || !crops_host || !emb || !workspace || n_a < 1 || n_b < 0 || (n_b > 0 && !src_b))
    return m2t_set_error(M2T_ERR_ARG, "m2t_swin_encode: null / bad source counts");
  if (!p->weights) return m2t_set_error(M2T_ERR_STATE, "m2t_swin_encode: call m2t_swin_load_weights first");
  if (n < 1 || n > p->max_images) return m2t_set_error(M2T_ERR_ARG, "m2t_swin_encode: n out of range");
  for (int i = 0; i < n; ++i) {
    const int si = crops_host[3 * i], y0 = crops_host[3 * i + 1], x0 = crops_host[3 * i + 2];
    if (si < 0 || si >= n_src || y0 < 0 || x0 < 0 || y0 + 224 > Hs || x0 + 224 > Ws)
      return m2t_set_error(M2T_ERR_ARG, "m2t_swin_encode: crop outside its source image");
  }
  hipStream_t st = (hipStream_t)stream;
  const int dt = p->dt;
  const float* wt = p->weights;
  hipError_t e = hipSuccess;
  if (!p->pinned) {
    e = hipHostMalloc((void**)&p->pinned, sizeof(int) * 3 * (size_t)p->max_images * m2t_swin::NPIN, hipHostMallocDefault);
    if (e != hipSuccess) { p->pinned = nullptr; return m2t_set_hip_error(e, __FILE__, __LINE__); }
    for (auto& ev : p->pin_ev) {
      e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
    }
  }
  const int slot = p->pin_next;
  p->pin_next = (slot + 1) % m2t_swin::NPIN;
  if (p->pin_used[slot]) {             // eight encodes ago: long done unless the caller never synchronises
    e = hipEventSynchronize(p->pin_ev[slot]);
    if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  }
  int* stage = p->pinned + (size_t)slot * 3 * p->max_images;
  memcpy(stage, crops_host, sizeof(int) * 3 * n);      // crops_host may be a temporary of the caller
  e = hipMemcpyAsync(SWP("crops"), stage, sizeof(int) * 3 * n, hipMemcpyHostToDevice, st);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  e = hipEventRecord(p->pin_ev[slot], st);
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  p->pin_used[slot] = true;
  void *X = SWP("X"), *Hn = SWP("Hn"), *QKV = SWP("QKV"), *AO = SWP("AO"), *MH = SWP("MH");
  const float* fbias = (const float*)SWP("fbias");
  CKS(launch_swin_patchify(dt, src, src_b, n_a, Hs, Ws, (const int*)SWP("crops"), n, SWP("A0"), st));
  long long M = (long long)n * 3136;
  CKS(swin_gemm(dt, M2T_E_BIAS, SWP("A0"), 48, spk(p, workspace, "pe"), X, 96, M,
                wt + p->poff.at("embeddings.patch_embeddings.projection.bias"), nullptr, st));
  CKS(launch_layernorm(dt, X, wt + p->poff.at("embeddings.norm.weight"), wt + p->poff.at("embeddings.norm.bias"), X, M, 96, st));
  int H = 56;
  for (int s = 0; s < 4; ++s) {
    const int C = 96 << s;
    for (int j = 0; j < DEPTHS[s]; ++j) {
      const std::string b = "encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(j) + ".";
      const int shift = (j % 2 == 0 || H <= 7) ? 0 : 3;
      CKS(launch_layernorm(dt, X, wt + p->poff.at(b + "layernorm_before.weight"), wt + p->poff.at(b + "layernorm_before.bias"), Hn, M, C, st));
      CKS(swin_gemm(dt, M2T_E_BIAS, Hn, C, spk(p, workspace, b + "qkv"), QKV, 3 * C, M, fbias + p->fb.at(b + "qkv_bias"), nullptr, st));
      CKS(launch_swin_attn(dt, QKV, wt + p->poff.at(b + "attention.self.relative_position_bias_table"), AO, n, H, H, C, HEADS[s], shift, st));
      CKS(swin_gemm(dt, M2T_E_BIAS_RESID, AO, C, spk(p, workspace, b + "o"), X, C, M, wt + p->poff.at(b + "attention.output.dense.bias"), X, st));
      if (dt != M2T_F32 && s < 2 && p->fused_mlp) {
        // LayerNorm + fc1 + GELU + fc2 + residual in one kernel: the 4C-wide hidden tensor stays in LDS
        CKS(launch_swin_mlp_fused(X, wt + p->poff.at(b + "layernorm_after.weight"), wt + p->poff.at(b + "layernorm_after.bias"),
                                  spk(p, workspace, b + "fc1F"), wt + p->poff.at(b + "intermediate.dense.bias"), spk(p, workspace, b + "fc2F"),
                                  wt + p->poff.at(b + "output.dense.bias"), M, C, st));
        continue;
      }
      CKS(launch_layernorm(dt, X, wt + p->poff.at(b + "layernorm_after.weight"), wt + p->poff.at(b + "layernorm_after.bias"), Hn, M, C, st));
      CKS(swin_gemm(dt, M2T_E_BIAS_GELU, Hn, C, spk(p, workspace, b + "fc1"), MH, 4 * C, M, wt + p->poff.at(b + "intermediate.dense.bias"), nullptr, st));
      CKS(swin_gemm(dt, M2T_E_BIAS_RESID, MH, 4 * C, spk(p, workspace, b + "fc2"), X, C, M, wt + p->poff.at(b + "output.dense.bias"), X, st));
    }
    if (s < 3) {
      const std::string d = "encoder.layers." + std::to_string(s) + ".downsample.";
      CKS(launch_swin_merge_gather(dt, X, Hn, n, H, H, C, st));
      M /= 4; H /= 2;
      CKS(launch_layernorm(dt, Hn, wt + p->poff.at(d + "norm.weight"), wt + p->poff.at(d + "norm.bias"), Hn, M, 4 * C, st));
      CKS(swin_gemm(dt, M2T_E_PLAIN, Hn, 4 * C, spk(p, workspace, d + "red"), X, 2 * C, M, nullptr, nullptr, st));
    }
  }
  CKS(launch_layernorm(dt, X, wt + p->poff.at("layernorm.weight"), wt + p->poff.at("layernorm.bias"), Hn, M, 768, st));
  CKS(launch_swin_head(dt, Hn, wt + p->poff.at("projection_head.weight"), emb, n, st));
  return 0;
}

extern "C" int m2t_semantic_loss(const float* emb, const float* text, int B, int n_patches, float* per_sample, float* total,
                                 void* stream) {
  if (!emb || !text || B < 1 || n_patches < 1) return m2t_set_error(M2T_ERR_ARG, "m2t_semantic_loss: bad argument");
  return launch_semantic_loss(emb, text, B, n_patches, per_sample, total, (hipStream_t)stream);
}
extern "C" int m2t_bicubic_resize(const float* src, float* dst, int NC, int Hin, int Win, int Hout, int Wout, void* stream) {
  if (!src || !dst || NC < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) return m2t_set_error(M2T_ERR_ARG, "m2t_bicubic_resize: bad argument");
  return launch_bicubic_resize(src, dst, NC, Hin, Win, Hout, Wout, (hipStream_t)stream);
}
