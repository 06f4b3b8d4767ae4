// m2t_rlutrans.hip -- util/rlutrans.py TransBlock.forward behind the C ABI.
//
//   x = x + EffAttention(LayerNorm(x))        util/rlutrans.py:85, 47-67
//   x = x + Mlp(LayerNorm(x))                 util/rlutrans.py:86, 21-27
// dim = 64, 8 heads of 8, Mlp hidden = dim / 4 with ReLU; EffAttention = reduce (64 -> 64, no bias), qkv (64 -> 192, no
// bias), softmax attention INSIDE token chunks of length floor(N / 16) (:53-55 -- a 17th, shorter chunk when N is not a
// multiple of 16), proj (64 -> 64 + bias).  The reference never calls this block (dead code): it is built
// from the library's existing LayerNorm and GEMM kernels plus one small attention kernel, forward only.
#include "../../include/m2t.h"
#include "m2t_kernels.h"

namespace {

constexpr int TB_DIM = 64, TB_HEADS = 8, TB_HD = 8, TB_HID = 16;
// flat parameter buffer = state_dict order of TransBlock(n_feat=64, dim=64)
constexpr long long OFF_REDUCE = 0, OFF_QKV = OFF_REDUCE + 64 * 64, OFF_PROJ_W = OFF_QKV + 192 * 64, OFF_PROJ_B = OFF_PROJ_W + 64 * 64,
                    OFF_N1W = OFF_PROJ_B + 64, OFF_N1B = OFF_N1W + 64, OFF_FC1W = OFF_N1B + 64, OFF_FC1B = OFF_FC1W + 16 * 64,
                    OFF_FC2W = OFF_FC1B + 16, OFF_FC2B = OFF_FC2W + 64 * 16, OFF_N2W = OFF_FC2B + 64, OFF_N2B = OFF_N2W + 64,
                    TB_NPARAMS = OFF_N2B + 64;
static_assert(TB_NPARAMS == 22928, "parameter count of TransBlock(dim=64) (SURVEY A17)");

// one thread per (token, head): scores against the keys of the token's chunk, softmax (max, then sum -- the order of
// torch.softmax), weighted sum of the values.  qkv [M][192] = q | k | v, each [head][8]; out [M][64] = [head][8]
template <typename T>
__global__ void __launch_bounds__(256) transblock_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, long long M, int N, int L) {
  const long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t >= M * TB_HEADS) return;
  const int head = (int)(t & 7);
  const long long tok = t >> 3;
  const long long b = tok / N;
  const int n = (int)(tok - b * N);
  const int j0 = (n / L) * L, j1 = min(N, j0 + L);
  const float scale = 0.35355339059327379f;       // head_dim ** -0.5, head_dim = 8
  float q[TB_HD];
#pragma unroll
  for (int d = 0; d < TB_HD; ++d) q[d] = to_f(qkv[tok * 192 + head * 8 + d]);
  float mx = -3.0e38f;
  for (int j = j0; j < j1; ++j) {
    const T* kp = qkv + (b * N + j) * 192 + 64 + head * 8;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < TB_HD; ++d) s += q[d] * to_f(kp[d]);
    mx = fmaxf(mx, s * scale);
  }
  float sum = 0.f, acc[TB_HD];
#pragma unroll
  for (int d = 0; d < TB_HD; ++d) acc[d] = 0.f;
  for (int j = j0; j < j1; ++j) {
    const T* kp = qkv + (b * N + j) * 192 + 64 + head * 8;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < TB_HD; ++d) s += q[d] * to_f(kp[d]);
    const float e = __expf(s * scale - mx);
    sum += e;
#pragma unroll
    for (int d = 0; d < TB_HD; ++d) acc[d] += e * to_f(kp[64 + d]);
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int d = 0; d < TB_HD; ++d) out[tok * 64 + head * 8 + d] = from_f<T>(acc[d] * inv);
}

template <typename T> __global__ void __launch_bounds__(256) tb_to_f32_kernel(const T* __restrict__ s, float* __restrict__ d, long long n) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) d[t] = to_f(s[t]);
}

struct TbLayout {
  size_t es, wts, X, Hn, R, QKV, AO, X1, H1, total;
  TbLayout(long long M, int dt) {
    es = dt == M2T_F32 ? 4 : 2;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    wts = o; o += al((size_t)TB_NPARAMS * es);
    X = o; o += al((size_t)M * 64 * es);
    Hn = o; o += al((size_t)M * 64 * es);
    R = o; o += al((size_t)M * 64 * es);
    QKV = o; o += al((size_t)M * 192 * es);
    AO = o; o += al((size_t)M * 64 * es);
    X1 = o; o += al((size_t)M * 64 * es);
    H1 = o; o += al((size_t)M * 16 * es);
    total = o;
  }
};

int tb_gemm(int dt, int emode, const void* A, int K, const void* W, void* Y, int N, long long M, const float* bias, const void* aux,
            hipStream_t st) {
  m2t_gemm_args ga{};
  ga.A = A; ga.lda = K; ga.W = W; ga.Y = Y; ga.ldy = N; ga.bias = bias; ga.aux = aux; ga.ldaux = N;
  ga.M = M; ga.N = N; ga.K = K; ga.H = 1; ga.Wd = 1; ga.r = 1; ga.C = 64;
  return launch_gemm_nt(dt, M2T_A_PLAIN, emode, ga, st);
}

}  // namespace

#define CKT(call) do { int rc__ = (call); if (rc__) return rc__; } while (0)

extern "C" size_t m2t_transblock_workspace_bytes(int B, int N, int dtype) {
  if (B < 1 || N < 1) return 0;
  return TbLayout((long long)B * N, dtype).total;
}

extern "C" int m2t_transblock_forward(const float* params, const float* x, float* y, int B, int N, int dtype, void* workspace,
                                      void* stream) {
  if (!params || !x || !y || !workspace) return m2t_set_error(M2T_ERR_ARG, "m2t_transblock_forward: null argument");
  if (dtype != M2T_F32 && dtype != M2T_BF16) return m2t_set_error(M2T_ERR_ARG, "m2t_transblock_forward: dtype must be M2T_F32 or M2T_BF16");
  if (B < 1 || N < 16) return m2t_set_error(M2T_ERR_ARG, "m2t_transblock_forward: needs B >= 1 and N >= 16 (the reference splits the tokens into chunks of N // 16)");
  hipStream_t st = (hipStream_t)stream;
  const long long M = (long long)B * N;
  const TbLayout lo(M, dtype);
  char* ws = (char*)workspace;
  const int dt = dtype;
  auto W = [&](long long off) { return (void*)(ws + lo.wts + (size_t)off * lo.es); };
  // weights and input in the compute element type (biases / LayerNorm affine stay fp32, read from `params`)
  CKT(launch_convert(dt, params, ws + lo.wts, TB_NPARAMS, st));
  CKT(launch_convert(dt, x, ws + lo.X, M * 64, st));
  void *X = ws + lo.X, *Hn = ws + lo.Hn, *R = ws + lo.R, *QKV = ws + lo.QKV, *AO = ws + lo.AO, *X1 = ws + lo.X1, *H1 = ws + lo.H1;
  CKT(launch_layernorm(dt, X, params + OFF_N1W, params + OFF_N1B, Hn, M, 64, st));                         // :85 norm1
  CKT(tb_gemm(dt, M2T_E_PLAIN, Hn, 64, W(OFF_REDUCE), R, 64, M, nullptr, nullptr, st));                   // :48 reduce
  CKT(tb_gemm(dt, M2T_E_PLAIN, R, 64, W(OFF_QKV), QKV, 192, M, nullptr, nullptr, st));                    // :50 qkv
  {
    const int L = N / 16;                                                                                  // :53-55
    const long long nt = M * TB_HEADS;
    if (dt == M2T_F32) hipLaunchKernelGGL(transblock_attn_kernel<float>, dim3((unsigned)ceil_divll(nt, 256)), dim3(256), 0, st, (const float*)QKV, (float*)AO, M, N, L);
    else hipLaunchKernelGGL(transblock_attn_kernel<bf16_t>, dim3((unsigned)ceil_divll(nt, 256)), dim3(256), 0, st, (const bf16_t*)QKV, (bf16_t*)AO, M, N, L);
    M2T_LAUNCH_CHECK();
  }
  CKT(tb_gemm(dt, M2T_E_BIAS_RESID, AO, 64, W(OFF_PROJ_W), X1, 64, M, params + OFF_PROJ_B, X, st));       // :66 proj, :85 + x
  CKT(launch_layernorm(dt, X1, params + OFF_N2W, params + OFF_N2B, Hn, M, 64, st));                        // :86 norm2
  CKT(tb_gemm(dt, M2T_E_BIAS_RELU, Hn, 64, W(OFF_FC1W), H1, TB_HID, M, params + OFF_FC1B, nullptr, st));   // :22-23 fc1 + ReLU
  CKT(tb_gemm(dt, M2T_E_BIAS_RESID, H1, TB_HID, W(OFF_FC2W), R, 64, M, params + OFF_FC2B, X1, st));       // :25 fc2, :86 + x
  if (dt == M2T_F32) {
    hipError_t e = hipMemcpyAsync(y, R, (size_t)M * 64 * 4, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return m2t_set_hip_error(e, __FILE__, __LINE__);
  } else {
    hipLaunchKernelGGL(tb_to_f32_kernel<bf16_t>, dim3((unsigned)std::min<long long>(ceil_divll(M * 64, 256), 4096)), dim3(256), 0, st, (const bf16_t*)R, y, M * 64);
    M2T_LAUNCH_CHECK();
  }
  return 0;
}
