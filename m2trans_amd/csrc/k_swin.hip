// k_swin.hip -- forward kernels of the MedCLIP image tower (Swin-T, 224x224) used by the
// reference's SemanticLoss (losses.py:22-25,68-69; medclip wraps HF swin-tiny-patch4-window7-224).
// Forward only: the regulariser runs under torch.no_grad() (losses.py:63) and carries no gradient.
// The dense layers (patch projection, qkv, o_proj, MLP, patch-merging reduction) are gemm_nt
// launches (k_gemm.hip); this file holds what sits between them.
#include "m2t_kernels.h"

// ---------------------------------------------------------------------------------------
// crop + patchify: out[(n, ph, pw)][c*16 + ky*4 + kx] = src[idx_n][c][y0_n + 4 ph + ky][x0_n + 4 pw + kx]
// (createNRandompatches crops, losses.py:38-39, fused with the im2col of the 4x4/4 patch conv)
// crops: int [n][3] = (source image index, y0, x0), device memory
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) swin_patchify_kernel(const float* __restrict__ src, const float* __restrict__ src_b, int n_a,
                                                            int Hs, int Ws, const int* __restrict__ crops, int n, T* __restrict__ out) {
  // source images 0 .. n_a - 1 live in `src`, the rest in `src_b` (the SR batch and the HR batch need not be concatenated)
  const long long total = (long long)n * 3136 * 6;     // 6 vectors of 8 per patch row (48)
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t % 6);
    const long long row = t / 6;
    const int pw = (int)(row % 56), ph = (int)((row / 56) % 56), im = (int)(row / 3136);
    const int si = crops[im * 3], y0 = crops[im * 3 + 1], x0 = crops[im * 3 + 2];
    const float* base = (si < n_a) ? src + (long long)si * 3 * Hs * Ws : src_b + (long long)(si - n_a) * 3 * Hs * Ws;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = v * 8 + e, c = k >> 4, ky = (k >> 2) & 3, kx = k & 3;
      o[e] = base[((long long)c * Hs + y0 + 4 * ph + ky) * Ws + x0 + 4 * pw + kx];
    }
    store8f(out + row * 48 + v * 8, o);
  }
}
int launch_swin_patchify(int dt, const float* src, const float* src_b, int n_a, int Hs, int Ws, const int* crops, int n, void* out,
                         hipStream_t st) {
  const long long total = (long long)n * 3136 * 6;
  const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(swin_patchify_kernel<float>, dim3(g), dim3(256), 0, st, src, src_b, n_a, Hs, Ws, crops, n, (float*)out);
  else hipLaunchKernelGGL(swin_patchify_kernel<bf16_t>, dim3(g), dim3(256), 0, st, src, src_b, n_a, Hs, Ws, crops, n, (bf16_t*)out);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// LayerNorm over the channel axis (eps 1e-5, affine), fp32 statistics.  A row is held in registers by a group of G lanes
// (8 channels = one 16-byte access per lane and vector, NV vectors per lane): one read, two-pass mean / variance on the
// registers, one write.  G = 16 / 32 / 64 lanes for C <= 128 / 256 / 512 (a wave takes 4 / 2 / 1 rows at once),
// G = 64 with NV = 2, 3 for C <= 1024 / 1536.  (The first version, one wave per row with 2-byte accesses and three
// passes over the row, ran at 1.2 TB/s: 65 us for the 200 704 x 96 map of stage 1.)
// ---------------------------------------------------------------------------------------
template <typename T, int G, int NV>
__global__ void __launch_bounds__(256) layernorm_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ y, long long M, int C,
                                                        float eps) {
  constexpr int RPW = 64 / G;                      // rows per wave
  const int lane = threadIdx.x & 63, gl = lane % G, sub = lane / G;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
  float ga[NV][8], be[NV][8];
  bool act[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int c0 = (gl + v * G) * 8;
    act[v] = c0 < C;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ga[v][e] = act[v] ? gamma[c0 + e] : 0.f; be[v][e] = act[v] ? beta[c0 + e] : 0.f; }
  }
  const float invC = 1.0f / (float)C;
  constexpr int U = (NV == 1) ? 4 : 2;             // row groups per trip: U * NV 16-byte loads in flight per lane
  for (long long r0 = wave * RPW; r0 < M; r0 += nw * RPW * U) {
    float v8[U][NV][8];
    long long rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      rr[u] = min(r0 + u * nw * RPW + sub, M - 1);   // clamped: the load stays unconditional, the store is masked
#pragma unroll
      for (int v = 0; v < NV; ++v) load8f(x + rr[u] * C + min((gl + v * G) * 8, C - 8), v8[u][v]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) { v8[u][v][e] = act[v] ? v8[u][v][e] : 0.f; s += v8[u][v][e]; }
#pragma unroll
      for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o);
      const float mean = s * invC;
      float q = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = act[v] ? v8[u][v][e] - mean : 0.f; q += d * d; }
#pragma unroll
      for (int o = 1; o < G; o <<= 1) q += __shfl_xor(q, o);
      const float rstd = 1.0f / sqrtf(q * invC + eps);
      if (r0 + u * nw * RPW + sub < M) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
          if (act[v]) {
            float o8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o8[e] = (v8[u][v][e] - mean) * rstd * ga[v][e] + be[v][e];
            store8f(y + rr[u] * C + (gl + v * G) * 8, o8);
          }
      }
    }
  }
}
int launch_layernorm(int dt, const void* x, const float* gamma, const float* beta, void* y, long long M, int C, hipStream_t st, float eps) {
  if (C % 8 || C > 1536) return m2t_set_error(-2, "layernorm: C must be a multiple of 8, at most 1536");
#define LN_GO(T_, G_, NV_)                                                                                           \
  {                                                                                                                  \
    const int g = (int)std::min<long long>(ceil_divll(M, 4 * (64 / G_)), 2048);   /* gamma / beta are re-read per thread */ \
    hipLaunchKernelGGL((layernorm_kernel<T_, G_, NV_>), dim3(g), dim3(256), 0, st, (const T_*)x, gamma, beta, (T_*)y, M, C, eps); \
  }
#define LN_T(T_)                                                                                                     \
  if (C <= 128) LN_GO(T_, 16, 1) else if (C <= 256) LN_GO(T_, 32, 1) else if (C <= 512) LN_GO(T_, 64, 1)             \
  else if (C <= 1024) LN_GO(T_, 64, 2) else LN_GO(T_, 64, 3)
  if (dt == M2T_F32) { LN_T(float) } else { LN_T(bf16_t) }
#undef LN_T
#undef LN_GO
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// (shifted) 7x7 window multi-head attention, head_dim 32.
// qkv [nimg*H*W][3C] (q | k | v, each heads x 32) in image token order; the cyclic shift and the
// window partition are index math on the loads/stores (no roll / partition copies).
// One workgroup per (window, head): 49 tokens padded to 64; wave w owns queries 16w..16w+15.
//   score = q.k * 32^-1/2 + table[(dh+6)*13 + (dw+6)][head] (+ -100 across cyclic-shift regions)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) swin_attn_kernel(const T* __restrict__ qkv, const float* __restrict__ table,
                                                        T* __restrict__ out, int H, int W, int C, int heads, int shift) {
  __shared__ __attribute__((aligned(16))) T Ks[64][40];
  __shared__ __attribute__((aligned(16))) T Vs[64][40];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int head = blockIdx.y;
  const int nwx = W / 7, nwy = H / 7;
  const int wi = blockIdx.x;
  const int wx = wi % nwx, wy = (wi / nwx) % nwy, im = wi / (nwx * nwy);
  auto token_of = [&](int pos) -> long long {        // window position -> token index in the un-shifted image
    const int py = pos / 7, px = pos - py * 7;
    int y = wy * 7 + py + shift, x = wx * 7 + px + shift;
    if (y >= H) y -= H;
    if (x >= W) x -= W;
    return ((long long)im * H + y) * W + x;
  };
  auto region_of = [&](int pos) -> int {             // region id in the ROLLED frame (shift mask)
    const int py = pos / 7, px = pos - py * 7;
    const int y = wy * 7 + py, x = wx * 7 + px;
    const int rh = (y >= H - 7) + (y >= H - shift), rw = (x >= W - 7) + (x >= W - shift);
    return rh * 3 + rw;
  };
  // stage K and V rows of this head (rows >= 49 zero)
  {
    const int key = tid >> 2, cv = tid & 3;
    Frag8<T> kf = frag_zero<T>(), vf = frag_zero<T>();
    if (key < 49) {
      const T* base = qkv + token_of(key) * (3 * C) + head * 32 + cv * 8;
      kf = load8(base + C);
      vf = load8(base + 2 * C);
    }
    store8(&Ks[key][cv * 8], kf);
    store8(&Vs[key][cv * 8], vf);
  }
  __syncthreads();
  const int q = 16 * wv + lr;
  const bool qok = q < 49;
  const long long qtok = qok ? token_of(q) : 0;
  Frag8<T> qf = frag_zero<T>();
  if (qok) qf = load8(qkv + qtok * (3 * C) + head * 32 + 8 * g);
  f32x4 s[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const Frag8<T> kf = load8(&Ks[16 * t + lr][8 * g]);
    mma16(s[t], kf, qf);
  }
  const float scale = 0.17677669529663687f;   // 32^-0.5
  const int qy = q / 7, qx = q - qy * 7;
  const int qreg = (shift > 0 && qok) ? region_of(q) : 0;
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;
      float v = -3.0e38f;
      if (key < 49 && qok) {
        const int ky = key / 7, kx = key - ky * 7;
        v = s[t][r] * scale + table[((qy - ky + 6) * 13 + (qx - kx + 6)) * heads + head];
        if (shift > 0 && region_of(key) != qreg) v += -100.0f;
      }
      s[t][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = fmaxf(mx, __shfl_xor(mx, 16));
  mx = fmaxf(mx, __shfl_xor(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = 16 * t + 4 * g + r;
      const float e = (key < 49 && qok) ? __expf(s[t][r] - mx) : 0.f;
      s[t][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 16);
  sum += __shfl_xor(sum, 32);
  const float inv = qok ? 1.0f / sum : 0.f;
  Frag8<T> pf[2];
#pragma unroll
  for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
    for (int j = 0; j < 8; ++j) pf[c4].set(j, s[2 * c4 + (j >> 2)][j & 3] * inv);
  f32x4 o[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const Frag8<T> vf = load8_tr(&Vs[32 * c4 + 4 * g][16 * mt], &Vs[32 * c4 + 16 + 4 * g][16 * mt], 40, lane);
      mma16(o[mt], vf, pf[c4]);
    }
  if (qok) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float v[4] = {o[mt][0], o[mt][1], o[mt][2], o[mt][3]};
      store4(out + qtok * C + head * 32 + 16 * mt + 4 * g, v);
    }
  }
}
int launch_swin_attn(int dt, const void* qkv, const float* bias_table, void* out, int nimg, int H, int W, int C, int heads,
                     int shift, hipStream_t st) {
  if (H % 7 || W % 7 || C != heads * 32) return m2t_set_error(-2, "swin_attn: grid must be a multiple of 7 and head_dim 32");
  dim3 grid(nimg * (H / 7) * (W / 7), heads);
  if (dt == M2T_F32) hipLaunchKernelGGL(swin_attn_kernel<float>, grid, dim3(256), 0, st, (const float*)qkv, bias_table, (float*)out, H, W, C, heads, shift);
  else hipLaunchKernelGGL(swin_attn_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)qkv, bias_table, (bf16_t*)out, H, W, C, heads, shift);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// patch merging gather: [n][H][W][C] -> [n][H/2][W/2][4C], concat order (r0,c0),(r1,c0),(r0,c1),(r1,c1)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) swin_merge_gather_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int H, int W, int C) {
  const int cv = C / 8;
  const long long total = (long long)n * (H / 2) * (W / 2) * 4 * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t % cv);
    long long r = t / cv;
    const int part = (int)(r & 3); r >>= 2;
    const int ox = (int)(r % (W / 2)); r /= (W / 2);
    const int oy = (int)(r % (H / 2));
    const int im = (int)(r / (H / 2));
    const int row = part & 1, col = part >> 1;
    const Frag8<T> f = load8(x + (((long long)im * H + 2 * oy + row) * W + 2 * ox + col) * C + v * 8);
    store8(y + (((long long)im * (H / 2) + oy) * (W / 2) + ox) * (4 * C) + part * C + v * 8, f);
  }
}
int launch_swin_merge_gather(int dt, const void* x, void* y, int nimg, int H, int W, int C, hipStream_t st) {
  const long long total = (long long)nimg * (H / 2) * (W / 2) * 4 * (C / 8);
  const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(swin_merge_gather_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, nimg, H, W, C);
  else hipLaunchKernelGGL(swin_merge_gather_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, nimg, H, W, C);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// head: mean over the 49 tokens of the final-LayerNorm output -> Linear(768,512,no bias) -> L2 normalise
// one workgroup per image
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) swin_head_kernel(const T* __restrict__ x, const float* __restrict__ proj, float* __restrict__ emb) {
  __shared__ float pooled[768];
  __shared__ float e[512];
  __shared__ float red[4];
  const int im = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < 768; c += 256) {
    float s = 0.f;
    for (int t = 0; t < 49; ++t) s += to_f(x[((long long)im * 49 + t) * 768 + c]);
    pooled[c] = s / 49.0f;
  }
  __syncthreads();
  float ss = 0.f;
  for (int o = tid; o < 512; o += 256) {
    const float* w = proj + (long long)o * 768;
    float a = 0.f;
    for (int c = 0; c < 768; ++c) a = fmaf(pooled[c], w[c], a);
    e[o] = a;
    ss += a * a;
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float inv = 1.0f / sqrtf(red[0] + red[1] + red[2] + red[3]);
  for (int o = tid; o < 512; o += 256) emb[(long long)im * 512 + o] = e[o] * inv;
}
int launch_swin_head(int dt, const void* x, const float* proj, float* emb, int nimg, hipStream_t st) {
  if (dt == M2T_F32) hipLaunchKernelGGL(swin_head_kernel<float>, dim3(nimg), dim3(256), 0, st, (const float*)x, proj, emb);
  else hipLaunchKernelGGL(swin_head_kernel<bf16_t>, dim3(nimg), dim3(256), 0, st, (const bf16_t*)x, proj, emb);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// SemanticLoss value (losses.py:71-79): emb [2B][512] (x embeddings then y embeddings, already unit
// norm), text [B][512] (normalised here): per_sample[i] = |x_i.t_i - y_i.t_i| / n_patches; total = sum
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) semantic_loss_kernel(const float* __restrict__ emb, const float* __restrict__ text, int B,
                                                            float inv_np, float* __restrict__ per_sample, float* __restrict__ total) {
  __shared__ float vals[256];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float tot = 0.f;
  for (int i = wv; i < B; i += 4) {
    float dx = 0.f, dy = 0.f, tt = 0.f;
    for (int c = lane; c < 512; c += 64) {
      const float t = text[(long long)i * 512 + c];
      dx += emb[(long long)i * 512 + c] * t;
      dy += emb[(long long)(B + i) * 512 + c] * t;
      tt += t * t;
    }
    dx = wave_sum(dx); dy = wave_sum(dy); tt = wave_sum(tt);
    const float v = fabsf(dx - dy) / sqrtf(tt) * inv_np;
    if (lane == 0 && per_sample) per_sample[i] = v;
    tot += v;
  }
  if (lane == 0) vals[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0 && total) total[0] = vals[0] + vals[1] + vals[2] + vals[3];
}
int launch_semantic_loss(const float* emb, const float* text, int B, int n_patches, float* per_sample, float* total, hipStream_t st) {
  hipLaunchKernelGGL(semantic_loss_kernel, dim3(1), dim3(256), 0, st, emb, text, B, 1.0f / (float)n_patches, per_sample, total);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// bicubic resize, align_corners=True, A = -0.75 (torch F.interpolate 'bicubic'; losses.py:53-54).
// Only reaches the returned value when N_patches == 1 (otherwise its result is discarded).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ void __launch_bounds__(256) bicubic_kernel(const float* __restrict__ src, float* __restrict__ dst, int NC, int Hin, int Win,
                                                      int Hout, int Wout) {
  const float A = -0.75f;
  const float sy = (Hout > 1) ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
  const float sx = (Wout > 1) ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
  const long long total = (long long)NC * Hout * Wout;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int ox = (int)(t % Wout), oy = (int)((t / Wout) % Hout);
    const long long nc = t / ((long long)Wout * Hout);
    const float fy = sy * oy, fx = sx * ox;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    const float ty = fy - iy, tx = fx - ix;
    const float wy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    const float wx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    const float* sp = src + nc * Hin * Win;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), Hin - 1);
      float row = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), Win - 1);
        row += wx[b] * sp[(long long)yy * Win + xx];
      }
      acc += wy[a] * row;
    }
    dst[t] = acc;
  }
}
int launch_bicubic_resize(const float* src, float* dst, int NC, int Hin, int Win, int Hout, int Wout, hipStream_t st) {
  const long long total = (long long)NC * Hout * Wout;
  hipLaunchKernelGGL(bicubic_kernel, dim3((unsigned)std::min<long long>(ceil_divll(total, 256), 4096)), dim3(256), 0, st, src, dst, NC, Hin, Win, Hout, Wout);
  M2T_LAUNCH_CHECK();
  return 0;
}

// fp32 -> T copy (weight packing for the frozen tower)
template <typename T>
__global__ void __launch_bounds__(256) convert_kernel(const float* __restrict__ s, T* __restrict__ d, long long n) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) d[t] = from_f<T>(s[t]);
}
int launch_convert(int dt, const float* src, void* dst, long long n, hipStream_t st) {
  const int g = (int)std::min<long long>(ceil_divll(n, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(convert_kernel<float>, dim3(g), dim3(256), 0, st, src, (float*)dst, n);
  else hipLaunchKernelGGL(convert_kernel<bf16_t>, dim3(g), dim3(256), 0, st, src, (bf16_t*)dst, n);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// Fused Swin MLP of stages 1 / 2 (bf16):  x <- x + fc2(gelu(fc1(LayerNorm(x)) + b1)) + b2      (modeling_swin: layernorm_after,
// intermediate.dense + GELU, output.dense, residual) for 64 token rows per workgroup.
// Unfused the 4C-wide hidden tensor goes through HBM twice (154 MB per layer at stage 1, batch 64) and the LayerNorm is a
// pass of its own: 23 + 70 + 60 us per layer.  Here the normalised rows (64 x C) and the hidden tile (64 x 4C, 50 / 99 KB)
// live in LDS; W1 / W2 arrive as pre-packed MFMA A-fragments (FRAG16 order) straight from L2, every fragment is loaded
// by exactly one wave and reused over the four 16-row tiles.  Six waves: wave w owns 4C/96 channel tiles of fc1 and C/96 of
// fc2 (C = 96: 4 and 1; C = 192: 8 and 2), so both products are balanced.
// ---------------------------------------------------------------------------------------
template <int C> struct MlpCfg {
  static constexpr int H4 = 4 * C, LDX = C + 8, LDH = H4 + 8;
  static constexpr int NCT1 = H4 / 16 / 6, NCT2 = C / 16 / 6, KS1 = C / 32, KS2 = H4 / 32;
  static constexpr size_t szX = sizeof(bf16_t) * 64 * LDX, szH = sizeof(bf16_t) * 64 * LDH, total = szX + szH;
  static_assert(C % 96 == 0, "six waves split the channel tiles of both products evenly for C = 96 k");
};
template <int C>
__global__ void __launch_bounds__(384) swin_mlp_fused_kernel(bf16_t* __restrict__ X, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const bf16_t* __restrict__ w1f,
                                                             const float* __restrict__ b1, const bf16_t* __restrict__ w2f,
                                                             const float* __restrict__ b2, long long M) {
  using T = bf16_t;
  using Cfg = MlpCfg<C>;
  constexpr int LDX = Cfg::LDX, LDH = Cfg::LDH, NCT1 = Cfg::NCT1, NCT2 = Cfg::NCT2, KS1 = Cfg::KS1, KS2 = Cfg::KS2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Hn)[LDX] = reinterpret_cast<T(*)[LDX]>(smem);
  T(*Hd)[LDH] = reinterpret_cast<T(*)[LDH]>(smem + Cfg::szX);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 64;
  // ---- the first fragments of W1 are on their way while the rows are normalised ----
  Frag8<T> wa[NCT1];
#pragma unroll
  for (int c = 0; c < NCT1; ++c) wa[c] = load8(w1f + (((long long)(wv * NCT1 + c) * KS1 + 0) * 64 + lane) * 8);
  // ---- LayerNorm: 4 lanes per row (threads 0..255), each C/4 channels as 16-byte vectors; fp32 two-pass ----
  if (tid < 256) {
    constexpr int NV = C / 32;                      // vectors per lane
    const int row = tid >> 2, part = tid & 3;
    const long long m = min(m0 + row, M - 1);
    float v[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      load8f(X + m * C + (part + 4 * i) * 8, v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s * (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2);
    const float rstd = 1.0f / sqrtf(q * (1.0f / (float)C) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c0 = (part + 4 * i) * 8;
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gamma[c0 + e] + beta[c0 + e];
      store8f(&Hn[row][c0], o);
    }
  }
  __syncthreads();
  // ---- hidden^T = W1 Hn^T: wave wv owns channel tiles wv*NCT1 .. ; all four row tiles ----
  {
    f32x4 acc[NCT1][4];
#pragma unroll
    for (int c = 0; c < NCT1; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[c][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      Frag8<T> a[NCT1];
#pragma unroll
      for (int c = 0; c < NCT1; ++c) a[c] = wa[c];
      if (ks + 1 < KS1) {
#pragma unroll
        for (int c = 0; c < NCT1; ++c) wa[c] = load8(w1f + (((long long)(wv * NCT1 + c) * KS1 + ks + 1) * 64 + lane) * 8);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const Frag8<T> bfr = load8(&Hn[16 * r + lr][32 * ks + 8 * g]);
#pragma unroll
        for (int c = 0; c < NCT1; ++c) mma16(acc[c][r], a[c], bfr);
      }
    }
    // lane (row 16 r + lr, g) holds hidden channels 16 ct + 4 g .. + 3
#pragma unroll
    for (int c = 0; c < NCT1; ++c) {
      const int ch = 16 * (wv * NCT1 + c) + 4 * g;
      const float bb[4] = {b1[ch], b1[ch + 1], b1[ch + 2], b1[ch + 3]};     // (parameter offsets are not 16-byte aligned)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = gelu_erf(acc[c][r][e] + bb[e]);
        store4(&Hd[16 * r + lr][ch], o);
      }
    }
  }
  // ---- y^T = W2 hidden^T: wave wv owns channel tiles wv*NCT2 .. ; W2 fragments through a 4-deep ring ----
  constexpr int DEPTH = 4;
  Frag8<T> wb[NCT2][DEPTH];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int c = 0; c < NCT2; ++c) wb[c][d] = load8(w2f + (((long long)(wv * NCT2 + c) * KS2 + d) * 64 + lane) * 8);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  f32x4 acc2[NCT2][4];
#pragma unroll
  for (int c = 0; c < NCT2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc2[c][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KS2; ++ks) {
    Frag8<T> a[NCT2];
#pragma unroll
    for (int c = 0; c < NCT2; ++c) a[c] = wb[c][ks % DEPTH];
    if (ks + DEPTH < KS2) {
#pragma unroll
      for (int c = 0; c < NCT2; ++c) wb[c][ks % DEPTH] = load8(w2f + (((long long)(wv * NCT2 + c) * KS2 + ks + DEPTH) * 64 + lane) * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const Frag8<T> bfr = load8(&Hd[16 * r + lr][32 * ks + 8 * g]);
#pragma unroll
      for (int c = 0; c < NCT2; ++c) mma16(acc2[c][r], a[c], bfr);
    }
  }
#pragma unroll
  for (int c = 0; c < NCT2; ++c) {
    const int ch = 16 * (wv * NCT2 + c) + 4 * g;
    const float bb[4] = {b2[ch], b2[ch + 1], b2[ch + 2], b2[ch + 3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long m = m0 + 16 * r + lr;
      if (m < M) {
        float p[4], o[4];
        load4(X + m * C + ch, p);                 // the residual: the un-normalised input row (read-modify-write by one lane)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc2[c][r][e] + bb[e] + p[e];
        store4(X + m * C + ch, o);
      }
    }
  }
}
// src [N][K] fp32 -> MFMA A-fragment order [N/16][K/32][64 lanes][8] (see M2T_PACK_FRAG16)
template <typename T>
__global__ void __launch_bounds__(256) frag16_pack_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int K) {
  const long long total = (long long)N * K;
  const int nks = K >> 5;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(e & 7), l = (int)((e >> 3) & 63);
    const long long f = e >> 9;
    const int ks = (int)(f % nks), tile = (int)(f / nks);
    dst[e] = from_f<T>(src[(long long)(16 * tile + (l & 15)) * K + 32 * ks + 8 * (l >> 4) + j]);
  }
}
int launch_frag16_pack(int dt, const float* src, void* dst, int N, int K, hipStream_t st) {
  if (N % 16 || K % 32) return m2t_set_error(-2, "frag16_pack: N must be a multiple of 16 and K of 32");
  const int g = (int)std::min<long long>(ceil_divll((long long)N * K, 256), 2048);
  if (dt == M2T_F32) hipLaunchKernelGGL(frag16_pack_kernel<float>, dim3(g), dim3(256), 0, st, src, (float*)dst, N, K);
  else hipLaunchKernelGGL(frag16_pack_kernel<bf16_t>, dim3(g), dim3(256), 0, st, src, (bf16_t*)dst, N, K);
  M2T_LAUNCH_CHECK();
  return 0;
}
// bf16, C = 96 / 192: X [M][C] updated in place; w1f / w2f = fc1 [4C][C] / fc2 [C][4C] weights in FRAG16 order
int launch_swin_mlp_fused(void* X, const float* gamma, const float* beta, const void* w1f, const float* b1, const void* w2f,
                          const float* b2, long long M, int C, hipStream_t st) {
  const unsigned nblk = (unsigned)ceil_divll(M, 64);
  if (C == 96) {
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)swin_mlp_fused_kernel<96>, (int)MlpCfg<96>::total)) return rc__;
    hipLaunchKernelGGL(swin_mlp_fused_kernel<96>, dim3(nblk), dim3(384), MlpCfg<96>::total, st, (bf16_t*)X, gamma, beta, (const bf16_t*)w1f,
                       b1, (const bf16_t*)w2f, b2, M);
  } else if (C == 192) {
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)swin_mlp_fused_kernel<192>, (int)MlpCfg<192>::total)) return rc__;
    hipLaunchKernelGGL(swin_mlp_fused_kernel<192>, dim3(nblk), dim3(384), MlpCfg<192>::total, st, (bf16_t*)X, gamma, beta, (const bf16_t*)w1f,
                       b1, (const bf16_t*)w2f, b2, M);
  } else {
    return m2t_set_error(-2, "swin_mlp_fused: C must be 96 or 192");
  }
  M2T_LAUNCH_CHECK();
  return 0;
}
