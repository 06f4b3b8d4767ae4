// m2t_text.hip -- the MedCLIP TEXT tower behind SemanticLoss (losses.py:22-25,64-65,74): BERT-base
// (emilyalsentzer/Bio_ClinicalBERT geometry: 12 layers, 768 hidden, 12 heads, 3072 intermediate, vocab 28 996, 512
// positions, 2 token types, LayerNorm eps 1e-12, erf GELU) -> mean over the tokens of hidden states 1, 2 and -1 ->
// Linear(768, 512, bias = False) -> L2 normalisation (encode_text of the un-vendored `medclip` package, restated from its
// published source; PARITY UNPINNED: neither the package nor its checkpoint is in the reference tree -- the oracle
// (oracle/text_oracle.py) is checked against transformers.BertModel with random weights).
// Forward only and off the training step's critical path: the reference evaluates it under torch.no_grad() once per
// sample (losses.py:63-65), and because it passes token_type_ids in the input_ids slot (losses.py:65) the result
// depends only on the caption's token COUNT -- m2trans_amd/losses.py caches one embedding per count.
// Built from the library's GEMM (bias / bias + GELU / bias + residual epilogues) and LayerNorm launchers plus three
// small kernels here (embedding + LayerNorm, masked multi-head attention for short sequences, pooling + projection).
#include <map>
#include <string>
#include <vector>
#include "m2t_kernels.h"
#include "../../include/m2t.h"

namespace {
constexpr int TX_H = 768, TX_HEADS = 12, TX_DH = 64, TX_FF = 3072, TX_LAYERS = 12, TX_VOCAB = 28996, TX_POS = 512, TX_TYPES = 2;
struct Ws { size_t off, n; };

// embeddings(input_ids, position 0..len-1, token type 0) + LayerNorm(eps 1e-12): one wave per token row
template <typename T>
__global__ void __launch_bounds__(256) text_embed_kernel(const int* __restrict__ ids, const float* __restrict__ word,
                                                         const float* __restrict__ pos, const float* __restrict__ type,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         T* __restrict__ out, int nrows, int len) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= nrows) return;
  const int id = min(max(ids[row], 0), TX_VOCAB - 1), t = row % len;
  float v[12];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int c = lane + 64 * j;
    v[j] = word[(long long)id * TX_H + c] + type[c] + pos[(long long)t * TX_H + c];      // HF order: inputs + token types, + positions
    s += v[j];
  }
  s = wave_sum(s);
  const float mean = s * (1.0f / TX_H);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 12; ++j) { const float d = v[j] - mean; q += d * d; }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q * (1.0f / TX_H) + 1e-12f);
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int c = lane + 64 * j;
    out[(long long)row * TX_H + c] = from_f<T>((v[j] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// BertSelfAttention for short sequences (len <= 128): one workgroup per (sequence, head), one wave per query row.
// qkv [rows][3 * 768] (q | k | v); scores = q . k / 8 + (mask ? 0 : -inf); softmax; out [rows][768]
template <typename T>
__global__ void __launch_bounds__(256) text_attn_kernel(const T* __restrict__ qkv, const int* __restrict__ mask, T* __restrict__ out, int len) {
  __shared__ float Ks[128][TX_DH + 1], Vs[128][TX_DH + 1];
  __shared__ float Ps[4][128];
  const int seq = blockIdx.x / TX_HEADS, head = blockIdx.x % TX_HEADS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long r0 = (long long)seq * len;
  for (int i = threadIdx.x; i < len * TX_DH; i += 256) {
    const int t = i / TX_DH, d = i % TX_DH;
    Ks[t][d] = to_f(qkv[(r0 + t) * (3 * TX_H) + TX_H + head * TX_DH + d]);
    Vs[t][d] = to_f(qkv[(r0 + t) * (3 * TX_H) + 2 * TX_H + head * TX_DH + d]);
  }
  __syncthreads();
  for (int qi = wv; qi < len; qi += 4) {
    const float qd = to_f(qkv[(r0 + qi) * (3 * TX_H) + head * TX_DH + lane]);      // lane = head dimension
    float sc[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int key = lane + 64 * h;
      float a = 0.f;
      for (int d = 0; d < TX_DH; ++d) a += __shfl(qd, d) * ((key < len) ? Ks[key][d] : 0.f);
      sc[h] = (key < len && mask[r0 + key] != 0) ? a * 0.125f : -3.0e38f;
    }
    float mx = fmaxf(sc[0], sc[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float e0 = (sc[0] > -1.0e38f) ? __expf(sc[0] - mx) : 0.f, e1 = (sc[1] > -1.0e38f) ? __expf(sc[1] - mx) : 0.f;
    const float sum = wave_sum(e0 + e1);
    const float inv = (sum > 0.f) ? 1.0f / sum : 0.f;
    Ps[wv][lane] = e0 * inv;
    Ps[wv][lane + 64] = e1 * inv;
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int key = 0; key < len; ++key) o += Ps[wv][key] * Vs[key][lane];
    out[(r0 + qi) * TX_H + head * TX_DH + lane] = from_f<T>(o);
    __builtin_amdgcn_wave_barrier();
  }
}

// pooled[seq][c] (+)= mean over the tokens of H[seq][t][c]  (UNMASKED mean, as the package takes it: `.mean(2)`)
template <typename T>
__global__ void __launch_bounds__(256) text_pool_kernel(const T* __restrict__ Hs, float* __restrict__ pooled, int len, int accumulate) {
  const int seq = blockIdx.x;
  for (int c = threadIdx.x; c < TX_H; c += 256) {
    float a = 0.f;
    for (int t = 0; t < len; ++t) a += to_f(Hs[((long long)seq * len + t) * TX_H + c]);
    a /= (float)len;
    pooled[(long long)seq * TX_H + c] = accumulate ? pooled[(long long)seq * TX_H + c] + a : a;
  }
}
// emb[seq] = normalise(proj [512][768] . (pooled[seq] / 3))
__global__ void __launch_bounds__(512) text_head_kernel(const float* __restrict__ pooled, const float* __restrict__ proj, float* __restrict__ emb) {
  __shared__ float xs[TX_H];
  __shared__ float part[8];
  const int seq = blockIdx.x, o = threadIdx.x;
  for (int c = threadIdx.x; c < TX_H; c += 512) xs[c] = pooled[(long long)seq * TX_H + c] * (1.0f / 3.0f);
  __syncthreads();
  float a = 0.f;
  for (int c = 0; c < TX_H; ++c) a += proj[(long long)o * TX_H + c] * xs[c];
  const float ss = wave_sum(a * a);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) tot += part[i];
  emb[(long long)seq * 512 + o] = a / sqrtf(tot);
}
}  // namespace

struct m2t_text {
  int max_seqs, max_len, dt;
  size_t esz;
  std::vector<std::string> pnames;
  std::map<std::string, long long> poff, pnum;
  long long nparams = 0;
  std::map<std::string, long long> pk;     // packed (T) weight offsets in elements
  long long npacked = 0;
  std::map<std::string, long long> fb;     // fused fp32 qkv biases
  long long nfb = 0;
  std::map<std::string, Ws> ws;
  size_t ws_bytes = 0;
  const float* weights = nullptr;
  void add_param(const std::string& n, long long c) { pnames.push_back(n); poff[n] = nparams; pnum[n] = c; nparams += c; }
  void add_pack(const std::string& n, long long c) { npacked = (npacked + 7) & ~7LL; pk[n] = npacked; npacked += c; }
  void add_ws(const std::string& n, size_t elems, size_t es) {
    ws_bytes = (ws_bytes + 255) & ~(size_t)255;
    ws[n] = Ws{ws_bytes, elems};
    ws_bytes += elems * es;
  }
};

extern "C" int m2t_text_create(m2t_text** out, int max_seqs, int max_len, int dtype) {
  if (!out || max_seqs < 1 || max_len < 1 || max_len > 128 || (dtype != M2T_F32 && dtype != M2T_BF16))
    return m2t_set_error(M2T_ERR_ARG, "m2t_text_create: bad argument (1 <= max_len <= 128)");
  m2t_text* p = new m2t_text();
  p->max_seqs = max_seqs; p->max_len = max_len; p->dt = dtype; p->esz = (dtype == M2T_F32) ? 4 : 2;
  // parameter inventory: HF BertModel checkpoint names (transformers 4.24; `pooler.*` is not used by encode_text and is
  // not part of the buffer) + the MedCLIP text projection
  p->add_param("embeddings.word_embeddings.weight", (long long)TX_VOCAB * TX_H);
  p->add_param("embeddings.position_embeddings.weight", (long long)TX_POS * TX_H);
  p->add_param("embeddings.token_type_embeddings.weight", (long long)TX_TYPES * TX_H);
  p->add_param("embeddings.LayerNorm.weight", TX_H);
  p->add_param("embeddings.LayerNorm.bias", TX_H);
  for (int l = 0; l < TX_LAYERS; ++l) {
    const std::string b = "encoder.layer." + std::to_string(l) + ".";
    for (const char* nm : {"query", "key", "value"}) {
      p->add_param(b + "attention.self." + nm + ".weight", (long long)TX_H * TX_H);
      p->add_param(b + "attention.self." + nm + ".bias", TX_H);
    }
    p->add_param(b + "attention.output.dense.weight", (long long)TX_H * TX_H);
    p->add_param(b + "attention.output.dense.bias", TX_H);
    p->add_param(b + "attention.output.LayerNorm.weight", TX_H);
    p->add_param(b + "attention.output.LayerNorm.bias", TX_H);
    p->add_param(b + "intermediate.dense.weight", (long long)TX_FF * TX_H);
    p->add_param(b + "intermediate.dense.bias", TX_FF);
    p->add_param(b + "output.dense.weight", (long long)TX_H * TX_FF);
    p->add_param(b + "output.dense.bias", TX_H);
    p->add_param(b + "output.LayerNorm.weight", TX_H);
    p->add_param(b + "output.LayerNorm.bias", TX_H);
    p->add_pack(b + "qkv", 3LL * TX_H * TX_H);
    p->add_pack(b + "o", (long long)TX_H * TX_H);
    p->add_pack(b + "fc1", (long long)TX_FF * TX_H);
    p->add_pack(b + "fc2", (long long)TX_H * TX_FF);
    p->fb[b + "qkv_bias"] = p->nfb; p->nfb += 3 * TX_H;
  }
  p->add_param("projection_head.weight", 512LL * TX_H);
  const size_t rows = (size_t)max_seqs * max_len, es = p->esz;
  p->add_ws("packed", (size_t)p->npacked, es);
  p->add_ws("fbias", (size_t)p->nfb, 4);
  p->add_ws("ids", rows, 4);
  p->add_ws("mask", rows, 4);
  p->add_ws("X", rows * TX_H, es);
  p->add_ws("Y", rows * TX_H, es);
  p->add_ws("QKV", rows * 3 * TX_H, es);
  p->add_ws("AO", rows * TX_H, es);
  p->add_ws("MH", rows * TX_FF, es);
  p->add_ws("pooled", (size_t)max_seqs * TX_H, 4);
  p->ws_bytes = (p->ws_bytes + 255) & ~(size_t)255;
  *out = p;
  return 0;
}
extern "C" void m2t_text_destroy(m2t_text* p) { delete p; }
extern "C" long long m2t_text_query(const m2t_text* p, const char* key) {
  if (!p || !key) return -1;
  const std::string k(key);
  if (k == "workspace_bytes") return (long long)p->ws_bytes;
  if (k == "num_params") return p->nparams;
  if (k == "num_param_tensors") return (long long)p->pnames.size();
  if (k == "max_seqs") return p->max_seqs;
  if (k == "max_len") return p->max_len;
  if (k.rfind("param:", 0) == 0) { auto it = p->poff.find(k.substr(6)); return it == p->poff.end() ? -1 : it->second; }
  if (k.rfind("numel:", 0) == 0) { auto it = p->pnum.find(k.substr(6)); return it == p->pnum.end() ? -1 : it->second; }
  return -1;
}
extern "C" const char* m2t_text_param_name(const m2t_text* p, int i) {
  if (!p || i < 0 || i >= (int)p->pnames.size()) return nullptr;
  return p->pnames[i].c_str();
}

#define TXP(name) ((char*)workspace + p->ws.at(name).off)
#define CKX(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)
static inline char* tpk(const m2t_text* p, void* workspace, const std::string& k) {
  return (char*)workspace + p->ws.at("packed").off + p->pk.at(k) * p->esz;
}

// once per weight set: the frozen fp32 weights -> element type T, q | k | v fused into one [2304][768] matrix per layer
extern "C" int m2t_text_load_weights(m2t_text* p, const float* weights, void* workspace, void* stream) {
  if (!p || !weights || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_text_load_weights: null");
  hipStream_t st = (hipStream_t)stream;
  const int dt = p->dt;
  p->weights = weights;
  float* fbias = (float*)TXP("fbias");
  for (int l = 0; l < TX_LAYERS; ++l) {
    const std::string b = "encoder.layer." + std::to_string(l) + ".";
    int part = 0;
    for (const char* nm : {"query", "key", "value"}) {
      CKX(launch_convert(dt, weights + p->poff.at(b + "attention.self." + nm + ".weight"),
                         tpk(p, workspace, b + "qkv") + (size_t)part * TX_H * TX_H * p->esz, (long long)TX_H * TX_H, st));
      CKX(launch_convert(M2T_F32, weights + p->poff.at(b + "attention.self." + nm + ".bias"), fbias + p->fb.at(b + "qkv_bias") + part * TX_H, TX_H, st));
      ++part;
    }
    CKX(launch_convert(dt, weights + p->poff.at(b + "attention.output.dense.weight"), tpk(p, workspace, b + "o"), (long long)TX_H * TX_H, st));
    CKX(launch_convert(dt, weights + p->poff.at(b + "intermediate.dense.weight"), tpk(p, workspace, b + "fc1"), (long long)TX_FF * TX_H, st));
    CKX(launch_convert(dt, weights + p->poff.at(b + "output.dense.weight"), tpk(p, workspace, b + "fc2"), (long long)TX_H * TX_FF, st));
  }
  return 0;
}

static int text_gemm(int dt, int emode, const void* A, int K, const void* W, void* Y, int N, long long M, const float* bias,
                     const void* aux, hipStream_t st) {
  m2t_gemm_args ga{};
  ga.A = A; ga.lda = K; ga.W = W; ga.Y = Y; ga.ldy = N; ga.bias = bias; ga.aux = aux; ga.ldaux = N;
  ga.M = M; ga.N = N; ga.K = K; ga.H = 1; ga.Wd = 1; ga.r = 1; ga.C = 64;
  return launch_gemm_nt(dt, M2T_A_PLAIN, emode, ga, st);
}

// medmodel.encode_text(input_ids, attention_mask) (losses.py:65,74): ids_host / mask_host int[n][len] in HOST memory
// (the reference passes `token_type_ids` as input_ids: the caller reproduces that by passing those values here)
// -> emb [n][512] fp32 on the device, unit L2 norm
extern "C" int m2t_text_encode(m2t_text* p, const int* ids_host, const int* mask_host, int n, int len, float* emb, void* workspace,
                               void* stream) {
  if (!p || !ids_host || !mask_host || !emb || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_text_encode: null argument");
  if (!p->weights) return m2t_set_error(M2T_ERR_STATE, "m2t_text_encode: call m2t_text_load_weights first");
  if (n < 1 || n > p->max_seqs || len < 1 || len > p->max_len) return m2t_set_error(M2T_ERR_ARG, "m2t_text_encode: n / len out of range");
  hipStream_t st = (hipStream_t)stream;
  const int dt = p->dt;
  const float* wt = p->weights;
  const int rows = n * len;
  hipError_t e = hipMemcpyAsync(TXP("ids"), ids_host, sizeof(int) * rows, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(TXP("mask"), mask_host, sizeof(int) * rows, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);       // the host arrays may be temporaries
  if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  void *X = TXP("X"), *Y = TXP("Y"), *QKV = TXP("QKV"), *AO = TXP("AO"), *MH = TXP("MH");
  float* pooled = (float*)TXP("pooled");
  const float* fbias = (const float*)TXP("fbias");
  const int* ids = (const int*)TXP("ids");
  const int* mask = (const int*)TXP("mask");
#define TX_LAUNCH(kern, grid, block, ...)                                                                 \
  do {                                                                                                    \
    if (dt == M2T_F32) hipLaunchKernelGGL(kern<float>, grid, block, 0, st, __VA_ARGS__);                  \
    else hipLaunchKernelGGL(kern<bf16_t>, grid, block, 0, st, __VA_ARGS__);                               \
    M2T_LAUNCH_CHECK();                                                                                   \
  } while (0)
  if (dt == M2T_F32)
    hipLaunchKernelGGL(text_embed_kernel<float>, dim3((rows + 3) / 4), dim3(256), 0, st, ids, wt + p->poff.at("embeddings.word_embeddings.weight"),
                       wt + p->poff.at("embeddings.position_embeddings.weight"), wt + p->poff.at("embeddings.token_type_embeddings.weight"),
                       wt + p->poff.at("embeddings.LayerNorm.weight"), wt + p->poff.at("embeddings.LayerNorm.bias"), (float*)X, rows, len);
  else
    hipLaunchKernelGGL(text_embed_kernel<bf16_t>, dim3((rows + 3) / 4), dim3(256), 0, st, ids, wt + p->poff.at("embeddings.word_embeddings.weight"),
                       wt + p->poff.at("embeddings.position_embeddings.weight"), wt + p->poff.at("embeddings.token_type_embeddings.weight"),
                       wt + p->poff.at("embeddings.LayerNorm.weight"), wt + p->poff.at("embeddings.LayerNorm.bias"), (bf16_t*)X, rows, len);
  M2T_LAUNCH_CHECK();
  for (int l = 0; l < TX_LAYERS; ++l) {
    const std::string b = "encoder.layer." + std::to_string(l) + ".";
    CKX(text_gemm(dt, M2T_E_BIAS, X, TX_H, tpk(p, workspace, b + "qkv"), QKV, 3 * TX_H, rows, fbias + p->fb.at(b + "qkv_bias"), nullptr, st));
    if (dt == M2T_F32) hipLaunchKernelGGL(text_attn_kernel<float>, dim3(n * TX_HEADS), dim3(256), 0, st, (const float*)QKV, mask, (float*)AO, len);
    else hipLaunchKernelGGL(text_attn_kernel<bf16_t>, dim3(n * TX_HEADS), dim3(256), 0, st, (const bf16_t*)QKV, mask, (bf16_t*)AO, len);
    M2T_LAUNCH_CHECK();
    // BertSelfOutput: LayerNorm(dense(attention) + hidden); BertOutput: LayerNorm(dense(gelu(dense(h))) + h)
    CKX(text_gemm(dt, M2T_E_BIAS_RESID, AO, TX_H, tpk(p, workspace, b + "o"), Y, TX_H, rows, wt + p->poff.at(b + "attention.output.dense.bias"), X, st));
    CKX(launch_layernorm(dt, Y, wt + p->poff.at(b + "attention.output.LayerNorm.weight"), wt + p->poff.at(b + "attention.output.LayerNorm.bias"), X, rows, TX_H, st, 1e-12f));
    CKX(text_gemm(dt, M2T_E_BIAS_GELU, X, TX_H, tpk(p, workspace, b + "fc1"), MH, TX_FF, rows, wt + p->poff.at(b + "intermediate.dense.bias"), nullptr, st));
    CKX(text_gemm(dt, M2T_E_BIAS_RESID, MH, TX_FF, tpk(p, workspace, b + "fc2"), Y, TX_H, rows, wt + p->poff.at(b + "output.dense.bias"), X, st));
    CKX(launch_layernorm(dt, Y, wt + p->poff.at(b + "output.LayerNorm.weight"), wt + p->poff.at(b + "output.LayerNorm.bias"), X, rows, TX_H, st, 1e-12f));
    // hidden_states[l + 1] = X: the package pools hidden states 1, 2 and -1
    if (l == 0 || l == 1 || l == TX_LAYERS - 1) {
      if (dt == M2T_F32) hipLaunchKernelGGL(text_pool_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)X, pooled, len, l == 0 ? 0 : 1);
      else hipLaunchKernelGGL(text_pool_kernel<bf16_t>, dim3(n), dim3(256), 0, st, (const bf16_t*)X, pooled, len, l == 0 ? 0 : 1);
      M2T_LAUNCH_CHECK();
    }
  }
  hipLaunchKernelGGL(text_head_kernel, dim3(n), dim3(512), 0, st, pooled, wt + p->poff.at("projection_head.weight"), emb);
  M2T_LAUNCH_CHECK();
  return 0;
}
