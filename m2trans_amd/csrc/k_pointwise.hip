// k_pointwise.hip -- the HBM-bound kernels of the M2Trans step (gfx950).
//
// Everything here moves bytes with little arithmetic: Haar DWT/IWT fused with the
// InstanceNorm apply / branch mixing / residuals, InstanceNorm statistics, pixel shuffle,
// column sums, clamp + L1 (+ its backward seed), the weight packer and fused Adam.  Loads
// and stores are 8/16-byte vectors along the contiguous channel axis of the NHWC tensors.
//
// Reference arithmetic restated (paths in /root/reference):
//   DWT / IWT ............... models/M2Trans_network.py:198-237
//   InstanceNorm2d .......... models/M2Trans_network.py:127,135
//   branch mixing/residuals . models/M2Trans_network.py:137-163
//   PixelShuffle ............ models/M2Trans_network.py:43,46,53
//   clamp, crop, L1 ......... models/M2Trans_network.py:74-76, train.py:76,199
//   Adam .................... train.py:81,210
#include "m2t_kernels.h"
#include "m2t_window.h"
#include "m2t_gemm_load.h"
#include "m2t_haar.h"
#include "m2t_instnorm.h"

// =======================================================================================
// generic L-level DWT / IWT on an NHWC tensor slice (operator API + bit-exact tests)
//   src [B][H][W][lds] (channels c0..c0+C)  ->  dst [B][H/S][W/S][ldd] (channels d0 + band*C + c)
// one thread = one pixel block x 4 channels
// =======================================================================================
template <typename T, int L>
__global__ void __launch_bounds__(256) dwt_kernel(const T* __restrict__ src, int lds_, int c0, T* __restrict__ dst,
                                                  int ldd, int d0, int B, int H, int W, int C) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N;
  const int cg = C / 4;
  const int hb = H / S, wb = W / S;
  const long long total = (long long)B * hb * wb * cg;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    long long r = t / cg;
    const int j = (int)(r % wb); r /= wb;
    const int i = (int)(r % hb);
    const int b = (int)(r / hb);
    float v[4][S][S];
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        float q[4];
        load4(src + (((long long)b * H + (i * S + y)) * W + (j * S + x)) * lds_ + c0 + g * 4, q);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c][y][x] = q[c];
      }
    float o[4][N];
#pragma unroll
    for (int c = 0; c < 4; ++c) Haar<L>::fwd(v[c], o[c]);
    T* dp = dst + (((long long)b * hb + i) * wb + j) * ldd + d0 + g * 4;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float q[4] = {o[0][n], o[1][n], o[2][n], o[3][n]};
      store4(dp + n * C, q);
    }
  }
}

template <typename T, int L>
__global__ void __launch_bounds__(256) iwt_kernel(const T* __restrict__ src, int lds_, int c0, T* __restrict__ dst,
                                                  int ldd, int d0, int B, int H, int W, int C) {
  // src [B][H/S][W/S][lds] holds N*C channels at c0; dst [B][H][W][ldd] gets C channels at d0
  constexpr int S = Haar<L>::S, N = Haar<L>::N;
  const int cg = C / 4;
  const int hb = H / S, wb = W / S;
  const long long total = (long long)B * hb * wb * cg;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t % cg);
    long long r = t / cg;
    const int j = (int)(r % wb); r /= wb;
    const int i = (int)(r % hb);
    const int b = (int)(r / hb);
    float o[4][N];
    const T* sp = src + (((long long)b * hb + i) * wb + j) * lds_ + c0 + g * 4;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float q[4];
      load4(sp + n * C, q);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c][n] = q[c];
    }
    float v[4][S][S];
#pragma unroll
    for (int c = 0; c < 4; ++c) Haar<L>::inv(o[c], v[c]);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        float q[4] = {v[0][y][x], v[1][y][x], v[2][y][x], v[3][y][x]};
        store4(dst + (((long long)b * H + (i * S + y)) * W + (j * S + x)) * ldd + d0 + g * 4, q);
      }
  }
}

static inline int grid_for(long long total, int block = 256, int cap = 256 * 16) {
  long long g = ceil_divll(total, block);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

template <typename T>
int launch_dwt_t(int L, const T* src, int lds_, int c0, T* dst, int ldd, int d0, int B, int H, int W, int C,
                 bool inverse, hipStream_t st) {
  const int S = 1 << L;
  const long long total = (long long)B * (H / S) * (W / S) * (C / 4);
  const int g = grid_for(total);
  if (!inverse) {
    if (L == 1) hipLaunchKernelGGL((dwt_kernel<T, 1>), dim3(g), dim3(256), 0, st, src, lds_, c0, dst, ldd, d0, B, H, W, C);
    else if (L == 2) hipLaunchKernelGGL((dwt_kernel<T, 2>), dim3(g), dim3(256), 0, st, src, lds_, c0, dst, ldd, d0, B, H, W, C);
    else return m2t_set_error(-2, "dwt: levels must be 1 or 2");
  } else {
    if (L == 1) hipLaunchKernelGGL((iwt_kernel<T, 1>), dim3(g), dim3(256), 0, st, src, lds_, c0, dst, ldd, d0, B, H, W, C);
    else if (L == 2) hipLaunchKernelGGL((iwt_kernel<T, 2>), dim3(g), dim3(256), 0, st, src, lds_, c0, dst, ldd, d0, B, H, W, C);
    else return m2t_set_error(-2, "iwt: levels must be 1 or 2");
  }
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_dwt(int dt, int L, const void* src, int lds_, int c0, void* dst, int ldd, int d0, int B, int H, int W,
               int C, bool inverse, hipStream_t st) {
  if (C % 4 || H % (1 << L) || W % (1 << L)) return m2t_set_error(-2, "dwt: C%4 or H,W not divisible");
  if (dt == M2T_F32) return launch_dwt_t<float>(L, (const float*)src, lds_, c0, (float*)dst, ldd, d0, B, H, W, C, inverse, st);
  return launch_dwt_t<bf16_t>(L, (const bf16_t*)src, lds_, c0, (bf16_t*)dst, ldd, d0, B, H, W, C, inverse, st);
}

// =======================================================================================
// pixel shuffle / unshuffle on NCHW fp32 (the reference layout): pure index permutation,
// bit-exact.  out[b,c,h*r+i,w*r+j] = in[b,c*r*r+i*r+j,h,w]   (nn.PixelShuffle)
// =======================================================================================
__global__ void __launch_bounds__(256) pixel_shuffle_nchw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                  int B, int C, int H, int W, int r, int inverse) {
  const long long total = (long long)B * C * r * r * H * W;   // C = channels AFTER shuffle
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    // t indexes the shuffled tensor [B][C][H*r][W*r]
    const int ow = (int)(t % (W * r));
    long long q = t / (W * r);
    const int oh = (int)(q % (H * r)); q /= (H * r);
    const int c = (int)(q % C);
    const int b = (int)(q / C);
    const int h = oh / r, i = oh % r, w = ow / r, j = ow % r;
    const long long s = (((long long)b * C * r * r + (c * r * r + i * r + j)) * H + h) * W + w;
    if (!inverse) out[t] = in[s]; else out[s] = in[t];
  }
}
int launch_pixel_shuffle_nchw(const float* in, float* out, int B, int C, int H, int W, int r, int inverse,
                              hipStream_t st) {
  const long long total = (long long)B * C * r * r * H * W;
  hipLaunchKernelGGL(pixel_shuffle_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, st, in, out, B, C, H, W, r, inverse);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// InstanceNorm statistics: per (b,c) mean and 1/sqrt(var+eps) over the P pixels of an
// NHWC [B][P][64] tensor.  Two deterministic stages, Welford/Chan merges in fp32.
// Stage 1: grid (nsplit, B); a block of 256 threads = 8 channel-groups(8 ch) x 32 pixel lanes.
// Also used for the backward reductions (sum g, sum g*xhat) through kernel below.
// =======================================================================================
struct Wf { float n, mean, m2; };
__device__ __forceinline__ Wf wf_merge(const Wf& a, const Wf& b) {
  Wf r;
  r.n = a.n + b.n;
  if (r.n == 0.f) { r.mean = 0.f; r.m2 = 0.f; return r; }
  const float d = b.mean - a.mean;
  const float f = b.n / r.n;
  r.mean = a.mean + d * f;
  r.m2 = a.m2 + b.m2 + d * d * a.n * f;
  return r;
}

template <typename T>
__global__ void __launch_bounds__(256) instnorm_stats1_kernel(const T* __restrict__ x, float* __restrict__ part, int P,
                                                              int nsplit) {
  // part [B][nsplit][64][3] = (n, mean, M2).  A thread owns 8 channels of every 32nd pixel of the split; it takes
  // them 16 pixels at a time: all loads first, then an exact two-pass (mean, then squared deviations) on the
  // registers -- no per-element division -- and one Welford merge per 16 pixels.
  const int b = blockIdx.y, sp = blockIdx.x;
  const long long npix = (long long)gridDim.y * P;          // x is P64
  const int cgp = threadIdx.x & 7, pl = threadIdx.x >> 3;   // 8 channel groups x 32 pixel lanes
  const int per = ceil_div(P, nsplit);
  const int p0 = sp * per, p1 = min(P, p0 + per);
  Wf w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { w[c].n = 0.f; w[c].mean = 0.f; w[c].m2 = 0.f; }
  for (int pb = p0 + pl; pb < p1; pb += 32 * 16) {
    float v[16][8];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = pb + 32 * i;
      if (p < p1) { load8f(x + p64(npix, (long long)b * P + p, cgp * 8), v[i]); ++cnt; }
      else {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[i][c] = 0.f;
      }
    }
    const float fn = (float)cnt, inv = 1.0f / fn;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) sum += v[i][c];            // invalid slots hold 0
      Wf q; q.n = fn; q.mean = sum * inv; q.m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = v[i][c] - q.mean; q.m2 += (i < cnt) ? d * d : 0.f; }
      w[c] = wf_merge(w[c], q);
    }
  }
  // block merge without divisions in the tree: N = sum n_i, mean = sum n_i mean_i / N, then
  // M2 = sum (M2_i + n_i (mean_i - mean)^2); the 32 pixel lanes of a channel group are 8 lanes per wave
  // (xor 8, 16, 32) x 4 waves (LDS), all in a fixed order
  __shared__ float sh[2][4][8][8];
  const int wvi = threadIdx.x >> 6;
  float nn[8], nm[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    nn[c] = w[c].n; nm[c] = w[c].n * w[c].mean;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { nn[c] += __shfl_xor(nn[c], o); nm[c] += __shfl_xor(nm[c], o); }
  }
  if ((threadIdx.x & 63) < 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) { sh[0][wvi][cgp][c] = nn[c]; sh[1][wvi][cgp][c] = nm[c]; }
  }
  __syncthreads();
  float N[8], mean_b[8], dev[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    N[c] = (sh[0][0][cgp][c] + sh[0][1][cgp][c]) + (sh[0][2][cgp][c] + sh[0][3][cgp][c]);
    mean_b[c] = ((sh[1][0][cgp][c] + sh[1][1][cgp][c]) + (sh[1][2][cgp][c] + sh[1][3][cgp][c])) / fmaxf(N[c], 1.f);
    const float d = w[c].mean - mean_b[c];
    dev[c] = w[c].m2 + w[c].n * d * d;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) dev[c] += __shfl_xor(dev[c], o);
  }
  __syncthreads();
  if ((threadIdx.x & 63) < 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) sh[0][wvi][cgp][c] = dev[c];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float* o = part + (((long long)b * nsplit + sp) * 64 + cgp * 8 + c) * 3;
      o[0] = N[c]; o[1] = mean_b[c];
      o[2] = (sh[0][0][cgp][c] + sh[0][1][cgp][c]) + (sh[0][2][cgp][c] + sh[0][3][cgp][c]);
    }
  }
}
__global__ void __launch_bounds__(64) instnorm_stats2_kernel(const float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd,
                                       int nsplit, float eps) {
  // every partial is fetched before the first merge (independent loads in flight), then a fixed pairwise tree:
  // depth 5 instead of a 32-step dependent chain of loads and divisions
  const int b = blockIdx.x, ch = threadIdx.x;   // 64 threads
  Wf w[M2T_NORM_SPLIT];
#pragma unroll
  for (int s = 0; s < M2T_NORM_SPLIT; ++s) {
    w[s].n = 0.f; w[s].mean = 0.f; w[s].m2 = 0.f;
    if (s < nsplit) {
      const float* o = part + (((long long)b * nsplit + s) * 64 + ch) * 3;
      w[s].n = o[0]; w[s].mean = o[1]; w[s].m2 = o[2];
    }
  }
  static_assert(M2T_NORM_SPLIT == 32, "the fixed tree below is written for 32 partials");
#pragma unroll
  for (int s = 0; s < 32; s += 2) w[s] = wf_merge(w[s], w[s + 1]);
#pragma unroll
  for (int s = 0; s < 32; s += 4) w[s] = wf_merge(w[s], w[s + 2]);
#pragma unroll
  for (int s = 0; s < 32; s += 8) w[s] = wf_merge(w[s], w[s + 4]);
#pragma unroll
  for (int s = 0; s < 32; s += 16) w[s] = wf_merge(w[s], w[s + 8]);
  const Wf r = wf_merge(w[0], w[16]);
  mean[b * 64 + ch] = r.mean;
  rstd[b * 64 + ch] = 1.0f / sqrtf(r.m2 / r.n + eps);   // biased variance
}
int launch_instnorm_stats(int dt, const void* x, float* mean, float* rstd, float* part, int B, int P, hipStream_t st) {
  const int nsplit = M2T_NORM_SPLIT;
  if (dt == M2T_F32) hipLaunchKernelGGL(instnorm_stats1_kernel<float>, dim3(nsplit, B), dim3(256), 0, st, (const float*)x, part, P, nsplit);
  else hipLaunchKernelGGL(instnorm_stats1_kernel<bf16_t>, dim3(nsplit, B), dim3(256), 0, st, (const bf16_t*)x, part, P, nsplit);
  M2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(instnorm_stats2_kernel, dim3(B), dim3(64), 0, st, part, mean, rstd, nsplit, 1e-5f);
  M2T_LAUNCH_CHECK();
  return 0;
}

// The second stage alone, for partials written by another kernel (the row-streaming conv leaves the statistics of its output as
// `part` [B][nsplit <= 64][64][3] = (n, mean, M2), k_conv.hip).  One 256-thread block per image: thread (channel, quarter) takes
// every 4th partial; N = sum n_i, mean = sum n_i mean_i / N, M2 = sum (M2_i + n_i (mean_i - mean)^2) -- no division per merge, all
// loads in flight at once, two LDS exchanges, every sum in a fixed order.  (The 64-thread pair tree this replaces took 15 us per
// launch, more than the 12.5 us pass over the map that the epilogue partials had saved.)
// NI partials per thread: 16 covers nsplit <= 64 (128 x 128 maps), 64 covers nsplit <= 256 (256 x 256); same sums in the same order
template <int NI>
__global__ void __launch_bounds__(256) instnorm_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean,
                                                                float* __restrict__ rstd, int nsplit, float eps) {
  const int b = blockIdx.x, ch = threadIdx.x & 63, q = threadIdx.x >> 6;
  __shared__ float sh[3][4][64];
  float n[NI], m[NI], v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int s = q + 4 * i;
    n[i] = 0.f; m[i] = 0.f; v[i] = 0.f;
    if (s < nsplit) {
      const float* o = part + (((long long)b * nsplit + s) * 64 + ch) * 3;
      n[i] = o[0]; m[i] = o[1]; v[i] = o[2];
    }
  }
  float sn = 0.f, sm = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) { sn += n[i]; sm = fmaf(n[i], m[i], sm); }
  sh[0][q][ch] = sn; sh[1][q][ch] = sm;
  __syncthreads();
  const float N = (sh[0][0][ch] + sh[0][1][ch]) + (sh[0][2][ch] + sh[0][3][ch]);
  const float mu = ((sh[1][0][ch] + sh[1][1][ch]) + (sh[1][2][ch] + sh[1][3][ch])) / fmaxf(N, 1.f);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) { const float d = m[i] - mu; sq += fmaf(n[i] * d, d, v[i]); }
  sh[2][q][ch] = sq;
  __syncthreads();
  if (q == 0) {
    const float M2 = (sh[2][0][ch] + sh[2][1][ch]) + (sh[2][2][ch] + sh[2][3][ch]);
    mean[b * 64 + ch] = mu;
    rstd[b * 64 + ch] = 1.0f / sqrtf(M2 / N + eps);   // biased variance
  }
}
int launch_instnorm_finalize(const float* part, float* mean, float* rstd, int B, int nsplit, hipStream_t st) {
  if (nsplit < 1 || nsplit > 8 * M2T_NORM_SPLIT) return m2t_set_error(-2, "instnorm_finalize: 1 .. 256 partials per image");
  if (nsplit <= 64) hipLaunchKernelGGL(instnorm_finalize_kernel<16>, dim3(B), dim3(256), 0, st, part, mean, rstd, nsplit, 1e-5f);
  else hipLaunchKernelGGL(instnorm_finalize_kernel<64>, dim3(B), dim3(256), 0, st, part, mean, rstd, nsplit, 1e-5f);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// branch_prep<L>: the input side of CFTM branch k (k = chunk index 0..3, L = DWT levels)
//   xin = norm(x)[chunk k]                      (k = 0)         (:135-139)
//   xin = (norm(x)[chunk k] + xc[chunk k-1]) / 2 (k >= 1)        (:141,147,155)
//   d   = DWT^L(xin)                                              (:143,149-150,157-158)
// x, xc: [B][H][W][64];  xin: [B][H][W][16];  d: [B][H/S][W/S][16*4^L]
// one thread = one (2^L)^2 pixel block x 4 channels
// =======================================================================================
// CH channels per thread: 4 (fp32 parity path: 16-byte accesses) or 8 (bf16: 16-byte accesses instead of 8-byte ones --
// the kernel is bound by memory INSTRUCTIONS, not bytes: 2.9 TB/s with 8-byte accesses)
template <int CH, typename T> __device__ __forceinline__ void load_ch(const T* p, float (&o)[CH]) {
  if constexpr (CH == 4) load4(p, o); else load8f(p, o);
}
template <int CH, typename T> __device__ __forceinline__ void store_ch(T* p, const float (&o)[CH]) {
  if constexpr (CH == 4) store4(p, o); else store8f(p, o);
}
template <typename T, int L, int CH>
__global__ void __launch_bounds__(256) branch_prep_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const T* __restrict__ xc,
                                                          int k, T* __restrict__ xin, T* __restrict__ d, int B, int H, int W) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N, G = 16 / CH;
  const int hb = H / S, wb = W / S;
  const long long npix = (long long)B * H * W;
  const int total = B * hb * wb * G;              // < 2^31: 32-bit index math (64-bit division is a software loop)
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int g = t % G;
    int r = t / G;
    const int j = r % wb; r /= wb;
    const int i = r % hb;
    const int b = r / hb;
    float mu[CH], rs[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { mu[c] = mean[b * 64 + k * 16 + g * CH + c]; rs[c] = rstd[b * 64 + k * 16 + g * CH + c]; }
    float v[CH][S][S];
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) {
        const long long pix = ((long long)b * H + (i * S + y)) * W + (j * S + xx);
        float q[CH];
        load_ch<CH>(x + ((long long)k * npix + pix) * 16 + g * CH, q);            // P64: chunk k is a dense plane
#pragma unroll
        for (int c = 0; c < CH; ++c) q[c] = (q[c] - mu[c]) * rs[c];
        if (k > 0) {
          float p[CH];
          load_ch<CH>(xc + ((long long)(k - 1) * npix + pix) * 16 + g * CH, p);
#pragma unroll
          for (int c = 0; c < CH; ++c) q[c] = (q[c] + p[c]) * 0.5f;
        }
        if (L > 0) {
          // keep the stored (rounded) value and the transformed value identical
          store_ch<CH>(xin + pix * 16 + g * CH, q);
          if (sizeof(T) == 2) {
#pragma unroll
            for (int c = 0; c < CH; ++c) q[c] = to_f(from_f<T>(q[c]));
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c][y][xx] = q[c];
      }
    float o[CH][N];
#pragma unroll
    for (int c = 0; c < CH; ++c) Haar<L>::fwd(v[c], o[c]);
    T* dp = d + (((long long)b * hb + i) * wb + j) * (16 * N) + g * CH;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float q[CH];
#pragma unroll
      for (int c = 0; c < CH; ++c) q[c] = o[c][n];
      store_ch<CH>(dp + n * 16, q);
    }
  }
}
// ---------------------------------------------------------------------------------------
// bf16, L = 2 (the two C = 256 branches): tiled through LDS.  The register kernel above gives each thread a 4x4 pixel
// block, so one wave-load touches 32 different 128-byte lines and uses 32 bytes of each (the rest comes back three
// loads later, after the 16-32 KB L1 has turned over): 14 us for 34 MB.  Here a workgroup owns 4 rows x TX pixels:
// phase A reads x / xc and writes xin with consecutive lanes on consecutive 16 bytes, phase B does the Haar
// butterflies per (block, channel) out of LDS, phase C writes the d rows as one contiguous run.  Same fp32 operations
// and rounding points as the register kernel -> identical bits.
// ---------------------------------------------------------------------------------------
template <int L, int TX>     // L = 1 (C = 64 branch: 2 rows x TX pixels per workgroup) or 2 (C = 256 branches: 4 rows x TX pixels)
__global__ void __launch_bounds__(256) branch_prep_tiled_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const bf16_t* __restrict__ xc, int k,
                                                                bf16_t* __restrict__ xin, bf16_t* __restrict__ d, int B, int H, int W) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N, BW = 16 * N;
  constexpr int NB = TX / S;                       // S x S blocks per tile
  constexpr int NV = S * TX * 2;                   // 16-byte vectors of the full-resolution tile (S rows x TX px x 2 halves)
  __shared__ __attribute__((aligned(16))) bf16_t Q[S][TX][16];
  __shared__ __attribute__((aligned(16))) bf16_t D[NB][BW];
  const int tid = threadIdx.x;
  const int tpr = W / TX;                          // tiles per image row
  const int t = blockIdx.x;                        // tile = (b, block row i, tile column)
  const int tc = t % tpr, i = (t / tpr) % (H / S), b = t / (tpr * (H / S));
  const long long npix = (long long)B * H * W;
  const int x0 = tc * TX;
  // ---- phase A ----
  for (int v = tid; v < NV; v += 256) {
    const int row = v / (2 * TX), cv = v % (2 * TX), px = cv >> 1, half = cv & 1;
    const long long pix = ((long long)b * H + S * i + row) * W + x0 + px;
    float q[8];
    load8f(x + ((long long)k * npix + pix) * 16 + half * 8, q);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ch = b * 64 + k * 16 + half * 8 + c;
      q[c] = (q[c] - mean[ch]) * rstd[ch];
    }
    float p[8];
    load8f(xc + ((long long)(k - 1) * npix + pix) * 16 + half * 8, p);        // k >= 1 for the L >= 1 branches
#pragma unroll
    for (int c = 0; c < 8; ++c) q[c] = (q[c] + p[c]) * 0.5f;
    store8f(xin + pix * 16 + half * 8, q);
    store8f(&Q[row][px][half * 8], q);             // the transformed value is the stored (rounded) one
  }
  __syncthreads();
  // ---- phase B: (block, channel) items ----
  for (int it = tid; it < NB * 16; it += 256) {
    const int blk = it >> 4, ch = it & 15;
    float vv[S][S];
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) vv[y][xx] = to_f(Q[y][S * blk + xx][ch]);
    float o[N];
    Haar<L>::fwd(vv, o);
#pragma unroll
    for (int n = 0; n < N; ++n) D[blk][n * 16 + ch] = from_f<bf16_t>(o[n]);
  }
  __syncthreads();
  // ---- phase C: NB consecutive d rows = one contiguous run of NB * 2 BW bytes ----
  bf16_t* dp = d + (((long long)b * (H / S) + i) * (W / S) + x0 / S) * BW;
  for (int v = tid; v < NB * (BW / 8); v += 256) store8(dp + v * 8, load8(&D[0][0] + v * 8));
}

template <typename T>
int launch_branch_prep_t(int L, const T* x, const float* mean, const float* rstd, const T* xc, int k, T* xin, T* d,
                         int B, int H, int W, hipStream_t st) {
  constexpr int CH = sizeof(T) == 2 ? 8 : 4;
  if ((long long)B * H * W * 4 >= (1LL << 31)) return m2t_set_error(-2, "branch_prep: B*H*W too large for 32-bit indexing");
  const int S = 1 << L;
  const int g = grid_for((long long)B * (H / S) * (W / S) * (16 / CH));
  if (L == 0) hipLaunchKernelGGL((branch_prep_kernel<T, 0, CH>), dim3(g), dim3(256), 0, st, x, mean, rstd, xc, k, xin, d, B, H, W);
  else if (L == 1) hipLaunchKernelGGL((branch_prep_kernel<T, 1, CH>), dim3(g), dim3(256), 0, st, x, mean, rstd, xc, k, xin, d, B, H, W);
  else hipLaunchKernelGGL((branch_prep_kernel<T, 2, CH>), dim3(g), dim3(256), 0, st, x, mean, rstd, xc, k, xin, d, B, H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_branch_prep(int dt, int L, const void* x, const float* mean, const float* rstd, const void* xc, int k,
                       void* xin, void* d, int B, int H, int W, hipStream_t st) {
  if (dt == M2T_F32) return launch_branch_prep_t<float>(L, (const float*)x, mean, rstd, (const float*)xc, k, (float*)xin, (float*)d, B, H, W, st);
  if (L == 2 && k >= 1 && W % 32 == 0 && H % 4 == 0) {
    if (W % 64 == 0) hipLaunchKernelGGL((branch_prep_tiled_kernel<2, 64>), dim3(B * (H / 4) * (W / 64)), dim3(256), 0, st, (const bf16_t*)x, mean, rstd,
                                        (const bf16_t*)xc, k, (bf16_t*)xin, (bf16_t*)d, B, H, W);
    else hipLaunchKernelGGL((branch_prep_tiled_kernel<2, 32>), dim3(B * (H / 4) * (W / 32)), dim3(256), 0, st, (const bf16_t*)x, mean, rstd,
                            (const bf16_t*)xc, k, (bf16_t*)xin, (bf16_t*)d, B, H, W);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  if (L == 1 && k >= 1 && W % 32 == 0 && H % 2 == 0) {
    if (W % 64 == 0) hipLaunchKernelGGL((branch_prep_tiled_kernel<1, 64>), dim3(B * (H / 2) * (W / 64)), dim3(256), 0, st, (const bf16_t*)x, mean, rstd,
                                        (const bf16_t*)xc, k, (bf16_t*)xin, (bf16_t*)d, B, H, W);
    else hipLaunchKernelGGL((branch_prep_tiled_kernel<1, 32>), dim3(B * (H / 2) * (W / 32)), dim3(256), 0, st, (const bf16_t*)x, mean, rstd,
                            (const bf16_t*)xc, k, (bf16_t*)xin, (bf16_t*)d, B, H, W);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  return launch_branch_prep_t<bf16_t>(L, (const bf16_t*)x, mean, rstd, (const bf16_t*)xc, k, (bf16_t*)xin, (bf16_t*)d, B, H, W, st);
}

// =======================================================================================
// branch_post<L>: xc[chunk k] = IWT^L(a) + xin       (:145,153,161)
// =======================================================================================
template <typename T, int L>
__global__ void __launch_bounds__(256) branch_post_kernel(const T* __restrict__ a, const T* __restrict__ xin,
                                                          T* __restrict__ xc, int k, int B, int H, int W) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N;
  const int hb = H / S, wb = W / S;
  const long long total = (long long)B * hb * wb * 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int g = (int)(t & 3);
    long long r = t >> 2;
    const int j = (int)(r % wb); r /= wb;
    const int i = (int)(r % hb);
    const int b = (int)(r / hb);
    float o[4][N];
    const T* sp = a + (((long long)b * hb + i) * wb + j) * (16 * N) + g * 4;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float q[4];
      load4(sp + n * 16, q);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c][n] = q[c];
    }
    float v[4][S][S];
#pragma unroll
    for (int c = 0; c < 4; ++c) Haar<L>::inv(o[c], v[c]);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) {
        const long long pix = ((long long)b * H + (i * S + y)) * W + (j * S + xx);
        float p[4];
        load4(xin + pix * 16 + g * 4, p);
        float q[4] = {v[0][y][xx] + p[0], v[1][y][xx] + p[1], v[2][y][xx] + p[2], v[3][y][xx] + p[3]};
        store4(xc + ((long long)k * B * H * W + pix) * 16 + g * 4, q);
      }
  }
}
int launch_branch_post(int dt, int L, const void* a, const void* xin, void* xc, int k, int B, int H, int W,
                       hipStream_t st) {
  const int S = 1 << L;
  const int g = grid_for((long long)B * (H / S) * (W / S) * 4);
#define BP(T_, L_) hipLaunchKernelGGL((branch_post_kernel<T_, L_>), dim3(g), dim3(256), 0, st, (const T_*)a, (const T_*)xin, (T_*)xc, k, B, H, W)
  if (dt == M2T_F32) { if (L == 1) BP(float, 1); else BP(float, 2); }
  else { if (L == 1) BP(bf16_t, 1); else BP(bf16_t, 2); }
#undef BP
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// backward of branch_post: ga = DWT^L(g_xc[chunk k])   (IWT^T = DWT, orthonormal Haar)
// =======================================================================================
int launch_branch_post_bwd(int dt, int L, const void* gxc, int k, void* ga, int B, int H, int W, hipStream_t st) {
  // source = channels k*16.. of the 64-wide g_xc; destination = dense [.,16*4^L]
  return launch_dwt(dt, L, gxc, 64, k * 16, ga, 16 << (2 * L), 0, B, H, W, 16, false, st);
}

// =======================================================================================
// backward of branch_prep<L> for branch k:
//   g_xin = IWT^L(g_d) + g_xc[chunk k]                 (k >= 1, L >= 1)
//   g_n[chunk k]     = g_xin / 2 ;   g_xc[chunk k-1] += g_xin / 2
// and for k = 0 (L = 0):   g_n[chunk 0] = g_d + g_xc[chunk 0]
// g_n, g_xc: P64 ([4][B*H*W][16])
// =======================================================================================
// gdwin != nullptr: g_d holds only the own-window products of the fused projection data gradient (k_attn_res.hip); the
// ring rows of the neighbouring windows ([window][36][16 N]) are added to the border pixels while the row is loaded
// (same fp32 adds in the same order and one rounding to T, as halo_gather + a plain load would give)
template <typename T, int L, int CH>
__global__ void __launch_bounds__(256) branch_prep_bwd_kernel(const T* __restrict__ gd, T* __restrict__ gxc,
                                                              T* __restrict__ gn, int k, int B, int H, int W,
                                                              const T* __restrict__ gdwin) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N, G = 16 / CH;
  const int hb = H / S, wb = W / S;
  const long long npix = (long long)B * H * W;          // gxc, gn are P64
  const int total = B * hb * wb * G;              // < 2^31: 32-bit index math (64-bit division is a software loop)
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int g = t % G;
    int r = t / G;
    const int j = r % wb; r /= wb;
    const int i = r % hb;
    const int b = r / hb;
    float o[CH][N];
    const T* sp = gd + (((long long)b * hb + i) * wb + j) * (16 * N) + g * CH;
    long long hoff[3];
    const int nsrc = gdwin ? halo_sources(b, i, j, hb / 8, wb / 8, 16 * N, hoff) : 0;
#pragma unroll
    for (int n = 0; n < N; ++n) {
      float q[CH];
      load_ch<CH>(sp + n * 16, q);
      if (nsrc) {
        for (int a = 0; a < nsrc; ++a) {
          float rr[CH];
          load_ch<CH>(gdwin + hoff[a] + n * 16 + g * CH, rr);
#pragma unroll
          for (int c = 0; c < CH; ++c) q[c] += rr[c];
        }
        if (sizeof(T) == 2) {
#pragma unroll
          for (int c = 0; c < CH; ++c) q[c] = to_f(from_f<T>(q[c]));
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c) o[c][n] = q[c];
    }
    float v[CH][S][S];
#pragma unroll
    for (int c = 0; c < CH; ++c) Haar<L>::inv(o[c], v[c]);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) {
        const long long pix = ((long long)b * H + (i * S + y)) * W + (j * S + xx);
        float p[CH];
        load_ch<CH>(gxc + ((long long)k * npix + pix) * 16 + g * CH, p);
        float q[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) q[c] = v[c][y][xx] + p[c];
        if (k == 0) {
          store_ch<CH>(gn + pix * 16 + g * CH, q);
        } else {
          float pp[CH];
          load_ch<CH>(gxc + ((long long)(k - 1) * npix + pix) * 16 + g * CH, pp);
#pragma unroll
          for (int c = 0; c < CH; ++c) { q[c] *= 0.5f; pp[c] += q[c]; }
          store_ch<CH>(gn + ((long long)k * npix + pix) * 16 + g * CH, q);
          store_ch<CH>(gxc + ((long long)(k - 1) * npix + pix) * 16 + g * CH, pp);
        }
      }
  }
}
// bf16, L = 1 / 2 backward, tiled through LDS like branch_prep_tiled_kernel (the register kernel: 14 us for 42 MB at L = 2):
// phase A stages the g_d rows, phase B the inverse butterflies per (block, channel), phase C the full-resolution
// read-modify-writes with consecutive lanes on consecutive 16 bytes.
template <int L, int TX>     // L = 1 (C = 64 branch: 2 rows x TX pixels per workgroup) or 2 (C = 256 branches: 4 rows x TX pixels)
__global__ void __launch_bounds__(256) branch_prep_bwd_tiled_kernel(const bf16_t* __restrict__ gd, bf16_t* __restrict__ gxc,
                                                                    bf16_t* __restrict__ gn, int k, int B, int H, int W,
                                                                    const bf16_t* __restrict__ gdwin) {
  constexpr int S = Haar<L>::S, N = Haar<L>::N, BW = 16 * N;       // block edge, sub-bands, values per block of g_d
  constexpr int NB = TX / S, VB = BW / 8, NV = S * TX * 2;         // blocks per tile, 16-byte vectors per block, full-resolution vectors
  __shared__ __attribute__((aligned(16))) bf16_t D[NB][BW];
  __shared__ __attribute__((aligned(16))) float V[S][TX][16];
  const int tid = threadIdx.x;
  const int tpr = W / TX;
  const int t = blockIdx.x;
  const int tc = t % tpr, i = (t / tpr) % (H / S), b = t / (tpr * (H / S));
  const long long npix = (long long)B * H * W;
  const int x0 = tc * TX;
  const bf16_t* sp = gd + (((long long)b * (H / S) + i) * (W / S) + x0 / S) * BW;
  // the read-modify-write operands of phase C are requested first: their HBM round trip then runs under phases A and B instead
  // of behind them (the workgroup is three dependent memory phases long; this removes one of its two exposed round trips)
  constexpr int CIT = (NV + 255) / 256;
  Frag8<bf16_t> pc[CIT], ppc[CIT];
#pragma unroll
  for (int it = 0; it < CIT; ++it) {
    const int v = min(tid + it * 256, NV - 1);
    const int row = v / (2 * TX), cv = v % (2 * TX), px = cv >> 1, half = cv & 1;
    const long long pix = ((long long)b * H + S * i + row) * W + x0 + px;
    pc[it] = load8(gxc + ((long long)k * npix + pix) * 16 + half * 8);
    ppc[it] = load8(gxc + ((long long)(k - 1) * npix + pix) * 16 + half * 8);
  }
  if (!gdwin) {
    for (int v = tid; v < NB * VB; v += 256) store8(&D[0][0] + v * 8, load8(sp + v * 8));
  } else {
    for (int v = tid; v < NB * VB; v += 256) {
      const int blk = v / VB, cv = v % VB;
      float q[8];
      load8f(sp + v * 8, q);
      long long hoff[3];
      const int nsrc = halo_sources(b, i, x0 / S + blk, H / (8 * S), W / (8 * S), BW, hoff);
      for (int a = 0; a < nsrc; ++a) {
        float rr[8];
        load8f(gdwin + hoff[a] + cv * 8, rr);
#pragma unroll
        for (int c = 0; c < 8; ++c) q[c] += rr[c];
      }
      store8f(&D[0][0] + v * 8, q);
    }
  }
  __syncthreads();
  for (int it = tid; it < NB * 16; it += 256) {
    const int blk = it >> 4, ch = it & 15;
    float o[N];
#pragma unroll
    for (int n = 0; n < N; ++n) o[n] = to_f(D[blk][n * 16 + ch]);
    float vv[S][S];
    Haar<L>::inv(o, vv);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) V[y][S * blk + xx][ch] = vv[y][xx];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < CIT; ++it) {
    const int v = tid + it * 256;
    if (v < NV) {
      const int row = v / (2 * TX), cv = v % (2 * TX), px = cv >> 1, half = cv & 1;
      const long long pix = ((long long)b * H + S * i + row) * W + x0 + px;
      float pp[8], q[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        q[c] = (V[row][px][half * 8 + c] + pc[it].get(c)) * 0.5f;
        pp[c] = ppc[it].get(c) + q[c];
      }
      store8f(gn + ((long long)k * npix + pix) * 16 + half * 8, q);
      store8f(gxc + ((long long)(k - 1) * npix + pix) * 16 + half * 8, pp);
    }
  }
}

int launch_branch_prep_bwd(int dt, int L, const void* gd, void* gxc, void* gn, int k, int B, int H, int W,
                           hipStream_t st, const void* gdwin) {
  if ((long long)B * H * W * 4 >= (1LL << 31)) return m2t_set_error(-2, "branch_prep_bwd: B*H*W too large for 32-bit indexing");
  if (dt != M2T_F32 && L == 2 && k >= 1 && W % 32 == 0 && H % 4 == 0) {
    if (W % 64 == 0) hipLaunchKernelGGL((branch_prep_bwd_tiled_kernel<2, 64>), dim3(B * (H / 4) * (W / 64)), dim3(256), 0, st, (const bf16_t*)gd,
                                        (bf16_t*)gxc, (bf16_t*)gn, k, B, H, W, (const bf16_t*)gdwin);
    else hipLaunchKernelGGL((branch_prep_bwd_tiled_kernel<2, 32>), dim3(B * (H / 4) * (W / 32)), dim3(256), 0, st, (const bf16_t*)gd, (bf16_t*)gxc,
                            (bf16_t*)gn, k, B, H, W, (const bf16_t*)gdwin);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  if (dt != M2T_F32 && L == 1 && k >= 1 && W % 32 == 0 && H % 2 == 0) {      // C = 64 branch: the same staging, two rows per workgroup
    if (W % 64 == 0) hipLaunchKernelGGL((branch_prep_bwd_tiled_kernel<1, 64>), dim3(B * (H / 2) * (W / 64)), dim3(256), 0, st, (const bf16_t*)gd,
                                        (bf16_t*)gxc, (bf16_t*)gn, k, B, H, W, (const bf16_t*)gdwin);
    else hipLaunchKernelGGL((branch_prep_bwd_tiled_kernel<1, 32>), dim3(B * (H / 2) * (W / 32)), dim3(256), 0, st, (const bf16_t*)gd, (bf16_t*)gxc,
                            (bf16_t*)gn, k, B, H, W, (const bf16_t*)gdwin);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  const int S = 1 << L;
  const int CH = dt == M2T_F32 ? 4 : 8;
  const int g = grid_for((long long)B * (H / S) * (W / S) * (16 / CH));
#define BP(T_, L_, CH_) hipLaunchKernelGGL((branch_prep_bwd_kernel<T_, L_, CH_>), dim3(g), dim3(256), 0, st, (const T_*)gd, (T_*)gxc, (T_*)gn, k, B, H, W, (const T_*)gdwin)
  if (dt == M2T_F32) { if (L == 0) BP(float, 0, 4); else if (L == 1) BP(float, 1, 4); else BP(float, 2, 4); }
  else { if (L == 0) BP(bf16_t, 0, 8); else if (L == 1) BP(bf16_t, 1, 8); else BP(bf16_t, 2, 8); }
#undef BP
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// InstanceNorm backward
//   g_x = rstd * (g_n - mean_p(g_n) - xhat * mean_p(g_n * xhat)) + g_res
// stage 1/2: s1[b,c] = sum_p g_n, s2[b,c] = sum_p g_n*xhat ; stage 3: apply (+ residual grad,
// the "+ x" of the feed-forward conv, :164)
// =======================================================================================
template <typename T, int NCG = 8>
__global__ void __launch_bounds__(256) instnorm_bwd_red1_kernel(const T* __restrict__ gn, const T* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ part, int P, int nsplit) {
  // part [B][nsplit][64][2]; the body is shared with c16_dgrad_prep_kernel (m2t_instnorm.h).  NCG < 8: the 8 channel groups are split over
  // gridDim.z = 8 / NCG workgroups (256^2 maps at batch 8 gave only 32 x 8 = 256 workgroups of a pure streaming kernel: 3.4 TB/s).  Which
  // form runs depends on the IMAGE size only, so an image's sums still do not depend on the batch it is in.
  __shared__ float sh[256][8][2];
  instnorm_bwd_red1_body<T, 0, NCG>(gn, x, mean, rstd, part, P, nsplit, blockIdx.y, blockIdx.x, gridDim.y, sh, NCG * (int)blockIdx.z);
}
__global__ void __launch_bounds__(64) instnorm_bwd_red2_kernel(const float* __restrict__ part, float* __restrict__ s, int nsplit, float invP) {
  // all 32 partials of a channel are fetched before the first add (one round trip instead of four dependent ones:
  // 18 -> 6 us on the critical path of every block), then a fixed pairwise tree
  static_assert(M2T_NORM_SPLIT == 32, "the fixed tree below is written for 32 partials");
  const int b = blockIdx.x, ch = threadIdx.x;
  const float* o = part + ((long long)b * nsplit * 64 + ch) * 2;
  float2 v[M2T_NORM_SPLIT];
#pragma unroll
  for (int q = 0; q < M2T_NORM_SPLIT; ++q)
    v[q] = (q < nsplit) ? *reinterpret_cast<const float2*>(o + (long long)q * 128) : make_float2(0.f, 0.f);
#pragma unroll
  for (int st = 1; st < M2T_NORM_SPLIT; st <<= 1)
#pragma unroll
    for (int q = 0; q < M2T_NORM_SPLIT; q += 2 * st) { v[q].x += v[q + st].x; v[q].y += v[q + st].y; }
  s[(b * 64 + ch) * 2 + 0] = v[0].x * invP;
  s[(b * 64 + ch) * 2 + 1] = v[0].y * invP;
}
// Second stage when the first one ran inside c16_dgrad_prep_kernel (round 5, option "fused_norm_red"): channels 16 .. 63 come as the
// usual M2T_NORM_SPLIT partials (that kernel's extra workgroups), channels 0 .. 15 as one partial PER 16-PIXEL TILE of the plane-0
// producer, part0 [tile][2][16] with the image's tiles contiguous.  One workgroup per image; fixed orders throughout.
__global__ void __launch_bounds__(256) instnorm_bwd_red2x_kernel(const float* __restrict__ part, const float* __restrict__ part0,
                                                                 float* __restrict__ s, int nsplit, int tiles, float invP) {
  static_assert(M2T_NORM_SPLIT == 32, "the fixed tree below is written for 32 partials");
  __shared__ float sl[8][32];
  const int b = blockIdx.x, t = threadIdx.x;
  {
    const int v = t & 31, slice = t >> 5;                     // value v = which * 16 + channel; slice of the image's tiles
    const float* o = part0 + (long long)b * tiles * 32 + v;
    float a = 0.f;
    int i = slice;
    for (; i + 56 < tiles; i += 64) {                          // eight loads in flight, added in tile order
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = o[(long long)(i + 8 * u) * 32];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += q[u];
    }
    for (; i < tiles; i += 8) a += o[(long long)i * 32];
    sl[slice][v] = a;
  }
  __syncthreads();
  if (t < 32) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += sl[u][t];
    s[(b * 64 + (t & 15)) * 2 + (t >> 4)] = a * invP;
  } else if (t >= 64 && t < 64 + 48) {
    const int ch = 16 + (t - 64);
    const float* o = part + ((long long)b * nsplit * 64 + ch) * 2;
    float2 v[M2T_NORM_SPLIT];
#pragma unroll
    for (int q = 0; q < M2T_NORM_SPLIT; ++q)
      v[q] = (q < nsplit) ? *reinterpret_cast<const float2*>(o + (long long)q * 128) : make_float2(0.f, 0.f);
#pragma unroll
    for (int st = 1; st < M2T_NORM_SPLIT; st <<= 1)
#pragma unroll
      for (int q = 0; q < M2T_NORM_SPLIT; q += 2 * st) { v[q].x += v[q + st].x; v[q].y += v[q + st].y; }
    s[(b * 64 + ch) * 2 + 0] = v[0].x * invP;
    s[(b * 64 + ch) * 2 + 1] = v[0].y * invP;
  }
}
// gres2 != nullptr (block 0): a second residual gradient joins in the same pass -- g(res) = g(X0) + g(Y) of `res + x`
// (models/M2Trans_network.py:70), which used to be an add_kernel launch of its own behind the last block (round 5)
// (R2 is a template argument: a load under a run-time branch -- even a uniform one -- is followed by s_waitcnt vmcnt(0), which cost
//  every launch 7 us when the second residual was a nullable pointer)
template <typename T, bool R2>
__global__ void __launch_bounds__(256) instnorm_bwd_apply_kernel(const T* __restrict__ gn, const T* __restrict__ x,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 const float* __restrict__ s, const T* __restrict__ gres,
                                                                 T* __restrict__ gx, int B, int P, const T* __restrict__ gres2) {
  const int total = B * P * 8;                    // < 2^31 (checked by the launcher): 32-bit index math
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int cgp = t & 7;
    const int pix = t >> 3;
    const int b = pix / P;
    float g[8], v[8], r[8], r2[8];
    const long long o64 = p64((long long)B * P, pix, cgp * 8);     // all tensors are P64
    load8f(gn + o64, g);
    load8f(x + o64, v);
    load8f(gres + o64, r);
    if constexpr (R2) load8f(gres2 + o64, r2);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ch = b * 64 + cgp * 8 + c;
      const float rs = rstd[ch];
      const float xh = (v[c] - mean[ch]) * rs;
      r[c] += rs * (g[c] - s[ch * 2] - xh * s[ch * 2 + 1]);
      // (the unfused chain stored this sum in T before the add: the same rounding point is kept, so the bits are the add_kernel's)
      if constexpr (R2) r[c] = to_f(from_f<T>(r[c])) + r2[c];
    }
    store8f(gx + o64, r);
  }
}
// part0 != nullptr (bf16): the first stage already ran (c16_dgrad_prep_kernel with its norm arguments): `part` holds channels 16 .. 63,
// part0 [B * tiles0][2][16] the per-tile partials of channels 0 .. 15 (tiles0 tiles per image)
int launch_instnorm_bwd(int dt, const void* gn, const void* x, const float* mean, const float* rstd, const void* gres,
                        void* gx, float* part, float* s, int B, int P, hipStream_t st, const float* part0, int tiles0, const void* gres2) {
  if ((long long)B * P * 8 >= (1LL << 31)) return m2t_set_error(-2, "instnorm_bwd: B*P too large for 32-bit indexing");
  const int nsplit = M2T_NORM_SPLIT;
  if (part0) {
    if (dt == M2T_F32 || tiles0 < 1) return m2t_set_error(-2, "instnorm_bwd: pre-reduced partials are a bf16 path");
    hipLaunchKernelGGL(instnorm_bwd_red2x_kernel, dim3(B), dim3(256), 0, st, (const float*)part, part0, s, nsplit, tiles0, 1.0f / (float)P);
    M2T_LAUNCH_CHECK();
    const int g = grid_for((long long)B * P * 8);
    if (gres2) hipLaunchKernelGGL((instnorm_bwd_apply_kernel<bf16_t, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)gn, (const bf16_t*)x, mean, rstd, s, (const bf16_t*)gres, (bf16_t*)gx, B, P, (const bf16_t*)gres2);
    else hipLaunchKernelGGL((instnorm_bwd_apply_kernel<bf16_t, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)gn, (const bf16_t*)x, mean, rstd, s, (const bf16_t*)gres, (bf16_t*)gx, B, P, (const bf16_t*)nullptr);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  if (dt == M2T_F32) hipLaunchKernelGGL(instnorm_bwd_red1_kernel<float>, dim3(nsplit, B), dim3(256), 0, st, (const float*)gn, (const float*)x, mean, rstd, part, P, nsplit);
  else if (P >= 32768) hipLaunchKernelGGL((instnorm_bwd_red1_kernel<bf16_t, 2>), dim3(nsplit, B, 4), dim3(256), 0, st, (const bf16_t*)gn, (const bf16_t*)x, mean, rstd, part, P, nsplit);
  else hipLaunchKernelGGL(instnorm_bwd_red1_kernel<bf16_t>, dim3(nsplit, B), dim3(256), 0, st, (const bf16_t*)gn, (const bf16_t*)x, mean, rstd, part, P, nsplit);
  M2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(instnorm_bwd_red2_kernel, dim3(B), dim3(64), 0, st, part, s, nsplit, 1.0f / (float)P);
  M2T_LAUNCH_CHECK();
  const int g = grid_for((long long)B * P * 8);
#define M2T_APPLY(T_, R2_) hipLaunchKernelGGL((instnorm_bwd_apply_kernel<T_, R2_>), dim3(g), dim3(256), 0, st, (const T_*)gn, (const T_*)x, mean, rstd, s, (const T_*)gres, (T_*)gx, B, P, (const T_*)gres2)
  if (dt == M2T_F32) { if (gres2) M2T_APPLY(float, true); else M2T_APPLY(float, false); }
  else { if (gres2) M2T_APPLY(bf16_t, true); else M2T_APPLY(bf16_t, false); }
#undef M2T_APPLY
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// elementwise add:  out = a + b   (res + x at :70, and gradient joins)
// =======================================================================================
template <typename T>
__global__ void __launch_bounds__(256) add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n8) {
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n8; t += (long long)gridDim.x * blockDim.x) {
    float x[8], y[8];
    load8f(a + t * 8, x);
    load8f(b + t * 8, y);
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] += y[c];
    store8f(o + t * 8, x);
  }
}
int launch_add(int dt, const void* a, const void* b, void* o, long long n, hipStream_t st) {
  const long long n8 = n / 8;
  if (dt == M2T_F32) hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n8)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)o, n8);
  else hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n8)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)o, n8);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// column sums of a [M][N] matrix (bias gradients): out[n] = sum_m a[m][n]; optional
// un-shuffle gather view (tail bias).  Two stages through `part` [nblk][N].
// =======================================================================================
template <typename T, int AMODE>
__global__ void __launch_bounds__(256) colsum1_kernel(const T* __restrict__ a, int lda, float* __restrict__ part, long long M,
                                                      int N, int rows_per_block, ShufGeom sg) {
  // thread: column group of 8 = threadIdx.x % (N/8); row lane = threadIdx.x / (N/8)
  const int ng = N / 8;
  const int lanes = 256 / ng;
  const int cgp = threadIdx.x % ng, rl = threadIdx.x / ng;
  const long long m0 = (long long)blockIdx.x * rows_per_block;
  const long long m1 = min(M, m0 + rows_per_block);
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < lanes)
    for (long long m = m0 + rl; m < m1; m += lanes) {
      const Frag8<T> f = gemm_load_a<T, AMODE>(a, lda, m, cgp * 8, sg);
#pragma unroll
      for (int c = 0; c < 8; ++c) s[c] += f.get(c);
    }
  __shared__ float sh[256][8];
#pragma unroll
  for (int c = 0; c < 8; ++c) sh[threadIdx.x][c] = s[c];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += 256) {
    const int g = n >> 3, c = n & 7;
    float acc = 0.f;
    for (int l = 0; l < lanes; ++l) acc += sh[l * ng + g][c];
    part[(long long)blockIdx.x * N + n] = acc;
  }
}
__global__ void __launch_bounds__(256) colsum2_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int N,
                                                      int accumulate) {
  // 32 columns x 8 lanes per block; each lane sums every 8th partial, then an ordered 8-way combine
  const int n = blockIdx.x * 32 + (threadIdx.x & 31), ln = threadIdx.x >> 5;
  float acc = 0.f;
  if (n < N)
    for (int b = ln; b < nblk; b += 8) acc += part[(long long)b * N + n];
  __shared__ float sh[8][32];
  sh[ln][threadIdx.x & 31] = acc;
  __syncthreads();
  if (ln == 0 && n < N) {
    float a = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) a += sh[l][threadIdx.x & 31];
    if (accumulate) out[n] += a; else out[n] = a;
  }
}
int launch_colsum(int dt, const void* a, int lda, long long M, int N, float* part, int max_part_blocks, float* out,
                  int accumulate, hipStream_t st, int unshuf, int gH, int gW, int gr, int gC, int* nblk_out) {
  if (N % 8 || N / 8 > 256) return m2t_set_error(-2, "colsum: bad N");
  ShufGeom sg{gH, gW, gr, gC};
  int nblk = (int)std::min<long long>(max_part_blocks, ceil_divll(M, 64));
  if (nblk < 1) nblk = 1;
  const int rpb = (int)ceil_divll(M, nblk);
  nblk = (int)ceil_divll(M, rpb);
  if (dt == M2T_F32) {
    if (unshuf) hipLaunchKernelGGL((colsum1_kernel<float, M2T_A_UNSHUF>), dim3(nblk), dim3(256), 0, st, (const float*)a, lda, part, M, N, rpb, sg);
    else hipLaunchKernelGGL((colsum1_kernel<float, M2T_A_PLAIN>), dim3(nblk), dim3(256), 0, st, (const float*)a, lda, part, M, N, rpb, sg);
  } else {
    if (unshuf) hipLaunchKernelGGL((colsum1_kernel<bf16_t, M2T_A_UNSHUF>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)a, lda, part, M, N, rpb, sg);
    else hipLaunchKernelGGL((colsum1_kernel<bf16_t, M2T_A_PLAIN>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)a, lda, part, M, N, rpb, sg);
  }
  M2T_LAUNCH_CHECK();
  if (nblk_out) { *nblk_out = nblk; return 0; }      // partials only: the caller reduces them later (batched)
  hipLaunchKernelGGL(colsum2_kernel, dim3(ceil_div(N, 32)), dim3(256), 0, st, part, out, nblk, N, accumulate);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// slab reduction for weight gradients: out[perm(e)] = sum_s slab[s][e]
//   perm 0: identity
//   perm 1: conv3x3 packed [tap][O][I] -> torch [O][I][3][3]
//   perm 2: shuffled rows n' = sub*C + c (sub = i*r+j) -> torch row c*r*r + sub; rows of K
//   perm 3 .. 6: see red_dest
// =======================================================================================
// Batched form: ONE launch reduces many slab sets (every weight / bias / rel-pos gradient of a group of
// blocks), driven by a descriptor table.  grid (x = element chunks, y = descriptor).
__device__ __forceinline__ long long red_dest(long long e, int perm, int p0, int p1, int p2, bool& skip) {
  skip = false;
  if (perm == 1) {           // conv3x3 packed [tap][O][I] -> torch [O][I][3][3]; p0 = O, p1 = I
    const int i = (int)(e % p1); const int o = (int)((e / p1) % p0); const int tap = (int)(e / ((long long)p0 * p1));
    return ((long long)o * p1 + i) * 9 + tap;
  } else if (perm == 2) {    // shuffled rows n' = sub*C + c -> torch row c*rr + sub; p0 = C, p1 = rr, p2 = K
    const int kk = (int)(e % p2); const int np = (int)(e / p2);
    const int sub = np / p0, c = np % p0;
    return ((long long)c * p1 + sub) * p2 + kk;
  } else if (perm == 3) {    // tail conv slab [32 (tap*3+oc, 27 used)][64 ic] -> torch [3][64][3][3]
    const int ic = (int)(e & 63), nn = (int)(e >> 6);
    if (nn >= 27) { skip = true; return 0; }
    return ((long long)(nn % 3) * 64 + ic) * 9 + nn / 3;
  } else if (perm == 5) {    // rows padded from p1 to p0 columns (head conv: [64][32] -> [64][27])
    const int kk = (int)(e % p0);
    if (kk >= p1) { skip = true; return 0; }
    return (e / p0) * p1 + kk;
  } else if (perm == 6) {    // conv3x3 [tap][I][O] (fused backward kernel) -> torch [O][I][3][3]; p0 = O, p1 = I
    const int o = (int)(e % p0); const int i = (int)((e / p0) % p1); const int tap = (int)(e / ((long long)p0 * p1));
    return ((long long)o * p1 + i) * 9 + tap;
  } else if (perm == 4) {    // rel-pos [10][C] -> rel_h [10][C/2] followed by rel_w [10][C/2]; p0 = C
    const int c = (int)(e % p0), i = (int)(e / p0);
    return (c < p0 / 2) ? ((long long)i * (p0 / 2) + c) : ((long long)10 * (p0 / 2) + (long long)i * (p0 / 2) + (c - p0 / 2));
  }
  return e;
}
__global__ void __launch_bounds__(256) multi_reduce_kernel(const float* __restrict__ arena, float* __restrict__ grads,
                                                           const m2t_red_desc* __restrict__ descs) {
  // 64 consecutive elements per chunk; the four waves split the slabs (wave g takes s = g, g+4, ...), each
  // with four independent partial sums in flight, then a fixed-order combine through LDS -> deterministic
  const m2t_red_desc d = descs[blockIdx.y];
  const float* slab = arena + d.src_off;
  float* out = grads + d.dst_off;
  __shared__ float sh[4][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long long nchunk = (d.n + 63) / 64;
  for (long long ch = blockIdx.x; ch < nchunk; ch += gridDim.x) {
    const long long e = ch * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < d.n) {
      int s = g;
      float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
      for (; s + 28 < d.ns; s += 32) {           // eight loads in flight per lane
        a0 += slab[(long long)s * d.n + e];
        a1 += slab[(long long)(s + 4) * d.n + e];
        a2 += slab[(long long)(s + 8) * d.n + e];
        a3 += slab[(long long)(s + 12) * d.n + e];
        b0 += slab[(long long)(s + 16) * d.n + e];
        b1 += slab[(long long)(s + 20) * d.n + e];
        b2 += slab[(long long)(s + 24) * d.n + e];
        b3 += slab[(long long)(s + 28) * d.n + e];
      }
      a0 += b0; a1 += b1; a2 += b2; a3 += b3;
      for (; s + 12 < d.ns; s += 16) {
        a0 += slab[(long long)s * d.n + e];
        a1 += slab[(long long)(s + 4) * d.n + e];
        a2 += slab[(long long)(s + 8) * d.n + e];
        a3 += slab[(long long)(s + 12) * d.n + e];
      }
      for (; s < d.ns; s += 4) a0 += slab[(long long)s * d.n + e];
    }
    sh[g][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && e < d.n) {
      bool skip;
      const long long dst = red_dest(e, d.perm, d.p0, d.p1, d.p2, skip);
      if (!skip) out[dst] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
    }
    __syncthreads();
  }
}
int launch_multi_reduce(const float* arena, float* grads, const m2t_red_desc* descs, int ndesc, hipStream_t st) {
  if (ndesc <= 0) return 0;
  hipLaunchKernelGGL(multi_reduce_kernel, dim3(256, ndesc), dim3(256), 0, st, arena, grads, descs);
  M2T_LAUNCH_CHECK();
  return 0;
}


// =======================================================================================
// clamp + crop + L1 (+ backward seed)                    (:74-76, train.py:199)
//   pre  [B][3][Hp][Wp] fp32  (padded-size pre-clamp network output, NCHW)
//   sr   [B][3][Hs][Ws] fp32  = clamp(pre, 0, R) cropped
//   loss partial sums of |sr - hr|  -> part[blocks]  (sum finished by loss_finish)
//   gpre [B][3][Hp][Wp] fp32  = scale * sign(sr - hr) * [0 <= pre <= R] inside the crop, 0 outside
// =======================================================================================
__global__ void __launch_bounds__(256) clamp_l1_kernel(const float* __restrict__ pre, const float* __restrict__ hr,
                                                       float* __restrict__ sr, float* __restrict__ gpre,
                                                       float* __restrict__ part, int B, int Hp, int Wp, int Hs, int Ws,
                                                       float R, float gscale) {
  const long long total = (long long)B * 3 * Hp * Wp;
  float acc = 0.f;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % Wp);
    long long q = t / Wp;
    const int y = (int)(q % Hp);
    const long long bc = q / Hp;
    float g = 0.f;
    if (y < Hs && x < Ws) {
      const float v = pre[t];
      const float c = fminf(fmaxf(v, 0.f), R);
      const long long o = (bc * Hs + y) * Ws + x;
      if (sr) sr[o] = c;
      if (hr) {
        const float d = c - hr[o];
        acc += fabsf(d);
        const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
        g = (v >= 0.f && v <= R) ? sg * gscale : 0.f;
      }
    }
    if (gpre) gpre[t] = g;
  }
  acc = wave_sum(acc);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && part) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
// four consecutive x per thread (Wp, Ws multiples of 4): 16-byte accesses and one index decomposition per quad -- the
// scalar kernel above is bound by memory instructions and 64-bit divisions (64 us for 150 MB)
__global__ void __launch_bounds__(256) clamp_l1_vec4_kernel(const float* __restrict__ pre, const float* __restrict__ hr,
                                                            float* __restrict__ sr, float* __restrict__ gpre,
                                                            float* __restrict__ part, int B, int Hp, int Wp, int Hs, int Ws,
                                                            float R, float gscale) {
  const int wq = Wp >> 2;
  const long long total = (long long)B * 3 * Hp * wq;
  float acc = 0.f;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int xq = (int)(t % wq);
    const int rowi = (int)(t / wq);                 // (b*3 + c) * Hp + y  < 2^31
    const int y = rowi % Hp, bc = rowi / Hp;
    f32x4 g = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (y < Hs && 4 * xq < Ws) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(pre + 4 * t);
      const long long o = ((long long)bc * Hs + y) * Ws + 4 * xq;
      f32x4 c;
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = fminf(fmaxf(v[i], 0.f), R);
      if (sr) *reinterpret_cast<f32x4*>(sr + o) = c;
      if (hr) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hr + o);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = c[i] - hv[i];
          acc += fabsf(d);
          const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
          g[i] = (v[i] >= 0.f && v[i] <= R) ? sg * gscale : 0.f;
        }
      }
    }
    if (gpre) *reinterpret_cast<f32x4*>(gpre + 4 * t) = g;
  }
  acc = wave_sum(acc);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && part) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void loss_finish_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ loss) {
  // single block, deterministic order
  __shared__ float sh[256];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = sh[0] * scale;
}
int launch_clamp_l1(const float* pre, const float* hr, float* sr, float* gpre, float* part, float* loss, int B, int Hp,
                    int Wp, int Hs, int Ws, float R, float loss_scale, float gscale, hipStream_t st) {
  const bool vec = (Wp % 4 == 0) && (Ws % 4 == 0) && ((long long)B * 3 * Hp < (1LL << 31));
  const int g = std::min(M2T_LOSS_BLOCKS, grid_for((long long)B * 3 * Hp * Wp / (vec ? 4 : 1)));
  if (vec) hipLaunchKernelGGL(clamp_l1_vec4_kernel, dim3(g), dim3(256), 0, st, pre, hr, sr, gpre, part, B, Hp, Wp, Hs, Ws, R, gscale);
  else hipLaunchKernelGGL(clamp_l1_kernel, dim3(g), dim3(256), 0, st, pre, hr, sr, gpre, part, B, Hp, Wp, Hs, Ws, R, gscale);
  M2T_LAUNCH_CHECK();
  if (loss) {
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, part, g, loss_scale, loss);
    M2T_LAUNCH_CHECK();
  }
  return 0;
}

int launch_loss_finish(const float* part, int n, float loss_scale, float* loss, hipStream_t st) {
  if (n < 1 || n > M2T_LOSS_BLOCKS) return m2t_set_error(-2, "loss_finish: 1 .. M2T_LOSS_BLOCKS partials");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, part, n, loss_scale, loss);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// fused multi-tensor Adam over the flat parameter buffer   (train.py:81,210)
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// g is pre-multiplied by gscale (1/world_size after a SUM all-reduce).
// =======================================================================================
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale) {
  const long long n4 = n / 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n4; t += (long long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[t];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[t];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[t];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[t];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float gi = gg[i] * gscale;
      mm[i] = b1 * mm[i] + (1.f - b1) * gi;
      vv[i] = b2 * vv[i] + (1.f - b2) * gi * gi;
      const float denom = sqrtf(vv[i]) / bc2_sqrt + eps;
      pp[i] = pp[i] - (lr / bc1) * (mm[i] / denom);
    }
    reinterpret_cast<f32x4*>(p)[t] = pp;
    reinterpret_cast<f32x4*>(m)[t] = mm;
    reinterpret_cast<f32x4*>(v)[t] = vv;
  }
  // tail (n not multiple of 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long long i = n4 * 4 + threadIdx.x;
    const float gi = g[i] * gscale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] = p[i] - (lr / bc1) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
  }
}
int launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                int step, float gscale, hipStream_t st) {
  const float bc1 = 1.f - powf(b1, (float)step);
  const float bc2 = 1.f - powf(b2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, bc1,
                     sqrtf(bc2), gscale);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// weight packer: ONE launch converts every fp32 master tensor into the element type and
// layouts the kernels want, driven by a descriptor table built at plan creation.
// =======================================================================================
// source index of packed element e of descriptor d (the layouts are documented in m2t_kernels.h)
__device__ __forceinline__ long long pack_src_index(const m2t_pack_desc& d, int e) {
  switch (d.kind) {
    case M2T_PACK_COPY: return e;
    case M2T_PACK_TRANSPOSE: {        // src [d0][d1] -> dst [d1][d0]
      const int r = e / d.d0, c = e % d.d0;   // dst row r (0..d1), col c (0..d0)
      return (long long)c * d.d1 + r;
    }
    case M2T_PACK_CONV3: {            // src [O=d0][I=d1][9] -> dst [tap][O][I]
      const int i = e % d.d1; const int oo = (e / d.d1) % d.d0; const int tap = e / (d.d0 * d.d1);
      return ((long long)oo * d.d1 + i) * 9 + tap;
    }
    case M2T_PACK_CONV3_T: {          // src [O=d0][I=d1][9] -> dst [tap'][I][O], tap' = 8 - tap (flipped kernel)
      const int oo = e % d.d0; const int i = (e / d.d0) % d.d1; const int tp = e / (d.d0 * d.d1);
      return ((long long)oo * d.d1 + i) * 9 + (8 - tp);
    }
    case M2T_PACK_SHUF_ROWS: {        // src [C*rr][K=d2] (row c*rr+sub) -> dst [sub*C + c][K]; d0 = C, d1 = rr
      const int kk = e % d.d2; const int np = e / d.d2;
      const int sub = np / d.d0, c = np % d.d0;
      return ((long long)c * d.d1 + sub) * d.d2 + kk;
    }
    case M2T_PACK_SHUF_ROWS_T: {      // src [C*rr][K] -> dst [K][sub*C + c]
      const int np = e % (d.d0 * d.d1); const int kk = e / (d.d0 * d.d1);
      const int sub = np / d.d0, c = np % d.d0;
      return ((long long)c * d.d1 + sub) * d.d2 + kk;
    }
    case M2T_PACK_FRAG16: {           // src [N=d0][K=d1] -> [N/16][K/32][64][8] (see m2t_kernels.h)
      const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;
      const int nks = d.d1 >> 5;
      const int ks = f % nks, tile = f / nks;
      return (long long)(16 * tile + (l & 15)) * d.d1 + 32 * ks + 8 * (l >> 4) + j;
    }
    case M2T_PACK_CONV3_ROWS:         // src torch [O=64][I=64][3][3] -> the A-fragments of conv3x3_c64_rows_kernel (k_conv.hip):
    case M2T_PACK_CONV3_ROWS_T: {     // [tap][half][kc][nt][64 lanes][8]; _T: the data-gradient weights (flipped taps, O <-> I)
      const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;
      const int nt = f & 1, kc = (f >> 1) & 1, h = (f >> 2) & 1, tap = f >> 3;
      const int row = 32 * h + 8 * ((l & 15) >> 2) + 4 * nt + (l & 3);     // output channel of the product
      const int k = 32 * kc + 8 * (l >> 4) + j;                             // contraction channel
      return (d.kind == M2T_PACK_CONV3_ROWS) ? ((long long)row * 64 + k) * 9 + tap : ((long long)k * 64 + row) * 9 + (8 - tap);
    }
    case M2T_PACK_FRAG16_T: {         // src [K=d1][N=d0] -> fragments of the transpose [N/16][K/32][64][8]
      const int j = e & 7, l = (e >> 3) & 63, f = e >> 9;
      const int nks = d.d1 >> 5;
      const int ks = f % nks, tile = f / nks;
      return (long long)(32 * ks + 8 * (l >> 4) + j) * d.d0 + 16 * tile + (l & 15);
    }
  }
  return e;
}
// Eight CONSECUTIVE packed elements whose index differs only in the innermost packed dimension come from source elements a fixed
// stride apart; returns that stride, or 0 when the descriptor's innermost dimension is not a multiple of 8 (element-wise path).
__device__ __forceinline__ long long pack_run_stride(const m2t_pack_desc& d) {
  switch (d.kind) {
    case M2T_PACK_COPY: return 1;
    case M2T_PACK_TRANSPOSE: return (d.d0 % 8 == 0) ? d.d1 : 0;
    case M2T_PACK_CONV3: return (d.d1 % 8 == 0) ? 9 : 0;
    case M2T_PACK_CONV3_T: return (d.d0 % 8 == 0) ? (long long)d.d1 * 9 : 0;
    case M2T_PACK_SHUF_ROWS: return (d.d2 % 8 == 0) ? 1 : 0;
    case M2T_PACK_SHUF_ROWS_T: return (d.d0 % 8 == 0) ? (long long)d.d1 * d.d2 : 0;
    case M2T_PACK_FRAG16: return 1;
    case M2T_PACK_CONV3_ROWS: return 9;
    case M2T_PACK_CONV3_ROWS_T: return 64 * 9;
    case M2T_PACK_FRAG16_T: return d.d0;
  }
  return 0;
}
template <typename T>
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ master, T* __restrict__ packed,
                                                   const m2t_pack_desc* __restrict__ descs, const int2* __restrict__ blocks) {
  // one workgroup per M2T_PACK_CHUNK output elements: blocks[i] = (descriptor, chunk).  (16 workgroups per descriptor --
  // 48 dependent gathers per thread on the 196 608-element attention weights -- took 98 us at the head of every step)
  // Round 5: a thread converts EIGHT consecutive packed elements -- one index decomposition (the divisions by run-time extents were
  // most of the kernel's instructions), eight strided source loads, one 16-byte (bf16) store: 35.7 -> ~12 us at the head of the step.
  const int2 bk = blocks[blockIdx.x];
  const m2t_pack_desc d = descs[bk.x];
  const float* s = master + d.src_off;
  T* o = packed + d.dst_off;
  const int n = (int)d.n;                  // every packed tensor has < 2^31 elements: 32-bit index math (64-bit division is a software loop)
  const int e0 = bk.y * M2T_PACK_CHUNK, e1 = min(n, e0 + M2T_PACK_CHUNK);
  const long long stride = pack_run_stride(d);
  if (stride != 0 && (n & 7) == 0 && ((d.dst_off * sizeof(T)) & 15) == 0) {
    for (int e = e0 + 8 * threadIdx.x; e < e1; e += 8 * 256) {
      const float* sp = s + pack_src_index(d, e);
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = sp[j * stride];
      store8f(o + e, v);
    }
    return;
  }
  for (int e = e0 + threadIdx.x; e < e1; e += 256) o[e] = from_f<T>(s[pack_src_index(d, e)]);
}
int launch_pack(int dt, const float* master, void* packed, const m2t_pack_desc* descs, const void* blocks, int nblocks, hipStream_t st) {
  if (dt == M2T_F32) hipLaunchKernelGGL(pack_kernel<float>, dim3(nblocks), dim3(256), 0, st, master, (float*)packed, descs, (const int2*)blocks);
  else hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(nblocks), dim3(256), 0, st, master, (bf16_t*)packed, descs, (const int2*)blocks);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// layout converters for the operator API / tests: NCHW fp32 <-> NHWC T
// =======================================================================================
template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int C, int HW, int inverse,
                                                           float* __restrict__ in_w) {
  const long long total = (long long)B * C * HW;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    // t indexes NHWC
    const int c = (int)(t % C);
    const long long q = t / C;
    const int p = (int)(q % HW);
    const int b = (int)(q / HW);
    const long long s = ((long long)b * C + c) * HW + p;
    if (!inverse) out[t] = from_f<T>(in[s]); else in_w[s] = to_f(out[t]);
  }
}
int launch_layout(int dt, const float* nchw, void* nhwc, float* nchw_out, int B, int C, int HW, int inverse, hipStream_t st) {
  const int g = grid_for((long long)B * C * HW);
  if (dt == M2T_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(g), dim3(256), 0, st, nchw, (float*)nhwc, B, C, HW, inverse, nchw_out);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(g), dim3(256), 0, st, nchw, (bf16_t*)nhwc, B, C, HW, inverse, nchw_out);
  M2T_LAUNCH_CHECK();
  return 0;
}
