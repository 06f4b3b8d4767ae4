// k_metrics.hip -- the evaluation metrics of the reference's test loop, on the device.
//
//   test.py:101-113 / train.py:299-312:  Y = utils.rgb_to_ycbcr(img)[:, 0:1]  (utils.py:121-146, including its
//   /255 on inputs that are already in [0,1]); crop `scale` pixels on every side; x255 when rgb_range == 1;
//   utils.calc_psnr (utils.py:179-184): mean(((sr - hr)/255)^2) in fp64;  utils.calc_ssim (utils.py:232-234):
//   pytorch_msssim.ssim defaults (11-tap Gaussian sigma 1.5, VALID, data_range 255, K = (0.01, 0.03)).
//
// Y is evaluated in fp32 with exactly the reference's operation order (no contraction) so the two Y planes are
// bit-identical to the reference's; everything after that is fp64.  The reference filters in fp32, where
// sigma = E[x^2] - mu^2 cancels ~5 of 7 digits (x^2 ~ 5e4, sigma of smooth images ~1): its own value moves by up
// to 1e-3 with the summation order of the convolution backend.  fp64 here is the value of the formula itself.
//
// Layout: sr, hr NCHW fp32 [B,3,H,W] as the reference's eval loop holds them.  One workgroup = one 32x32 tile of
// the SSIM map (42x42 Y samples in LDS, vertical pass then horizontal pass like the dependency).  Partial sums go
// to a scratch array and are folded by one workgroup per image in a fixed order: deterministic, no atomics.
#include "m2t_common.h"
#include "m2t_kernels.h"
#include "../../include/m2t.h"
#include <math.h>

namespace {

constexpr int WIN = 11;
constexpr int TS = 32;                 // SSIM map tile edge
constexpr int TI = TS + WIN - 1;       // 42 input samples per tile edge

struct SsimWin { double g[WIN]; };

__device__ __forceinline__ float y_of(const float* img, long long plane, long long off, int times255) {
  // image / 255. ; 65.481 * r + 128.553 * g + 24.966 * b + 16.0   (left to right, fp32, no fma)
  const float r = __fdiv_rn(img[off], 255.f);
  const float g = __fdiv_rn(img[plane + off], 255.f);
  const float b = __fdiv_rn(img[2 * plane + off], 255.f);
  float y = __fmul_rn(65.481f, r);
  y = __fadd_rn(y, __fmul_rn(128.553f, g));
  y = __fadd_rn(y, __fmul_rn(24.966f, b));
  y = __fadd_rn(y, 16.0f);
  return times255 ? __fmul_rn(y, 255.f) : y;
}

__device__ __forceinline__ double block_sum_256(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// squared error of the Y planes: grid (chunks, B); partial[b][chunk]
__global__ __launch_bounds__(256) void eval_mse_y_kernel(const float* __restrict__ sr, const float* __restrict__ hr, int H, int W,
                                                         int crop, int times255, double* __restrict__ partial) {
  __shared__ double red[4];
  const int b = blockIdx.y, Hc = H - 2 * crop, Wc = W - 2 * crop;
  const long long plane = (long long)H * W, n = (long long)Hc * Wc;
  const float* s = sr + (long long)b * 3 * plane;
  const float* h = hr + (long long)b * 3 * plane;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / Wc), x = (int)(i - (long long)y * Wc);
    const long long off = (long long)(y + crop) * W + (x + crop);
    const double d = ((double)y_of(s, plane, off, times255) - (double)y_of(h, plane, off, times255)) / 255.0;
    acc += d * d;
  }
  const double t = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[(long long)b * gridDim.x + blockIdx.x] = t;
}

// SSIM map tile: grid (tiles_x, tiles_y, B); nh / nw = taps along H / W (11, or 1 when the axis is shorter than the window)
__global__ __launch_bounds__(256) void eval_ssim_y_kernel(const float* __restrict__ sr, const float* __restrict__ hr, int H, int W,
                                                          int crop, int times255, int nh, int nw, SsimWin win,
                                                          double* __restrict__ partial) {
  __shared__ double X[TI][TI + 1], Y[TI][TI + 1];
  __shared__ double V[5][TS][TI + 1];          // vertical pass of x, y, xx, yy, xy
  __shared__ double red[4];
  const int b = blockIdx.z, Hc = H - 2 * crop, Wc = W - 2 * crop;
  const int Hm = Hc - nh + 1, Wm = Wc - nw + 1;                       // SSIM map size
  const int y0 = blockIdx.y * TS, x0 = blockIdx.x * TS;
  const int th = min(TS, Hm - y0), tw = min(TS, Wm - x0);             // outputs of this tile
  const int ih = th + nh - 1, iw = tw + nw - 1;                       // inputs of this tile
  const long long plane = (long long)H * W;
  const float* s = sr + (long long)b * 3 * plane;
  const float* h = hr + (long long)b * 3 * plane;
  for (int i = threadIdx.x; i < ih * iw; i += 256) {
    const int r = i / iw, c = i - r * iw;
    const long long off = (long long)(y0 + r + crop) * W + (x0 + c + crop);
    X[r][c] = (double)y_of(s, plane, off, times255);
    Y[r][c] = (double)y_of(h, plane, off, times255);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < th * iw; i += 256) {
    const int r = i / iw, c = i - r * iw;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    for (int t = 0; t < nh; ++t) {
      const double g = nh == 1 ? 1.0 : win.g[t], x = X[r + t][c], y = Y[r + t][c];
      a0 += g * x; a1 += g * y; a2 += g * (x * x); a3 += g * (y * y); a4 += g * (x * y);
    }
    V[0][r][c] = a0; V[1][r][c] = a1; V[2][r][c] = a2; V[3][r][c] = a3; V[4][r][c] = a4;
  }
  __syncthreads();
  const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
  double acc = 0.0;
  for (int i = threadIdx.x; i < th * tw; i += 256) {
    const int r = i / tw, c = i - r * tw;
    double m[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      double a = 0;
      for (int t = 0; t < nw; ++t) a += (nw == 1 ? 1.0 : win.g[t]) * V[q][r][c + t];
      m[q] = a;
    }
    const double mu1 = m[0], mu2 = m[1];
    const double s1 = m[2] - mu1 * mu1, s2 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
    const double cs = (2.0 * s12 + C2) / (s1 + s2 + C2);
    acc += ((2.0 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs;
  }
  const double t = block_sum_256(acc, red);
  if (threadIdx.x == 0)
    partial[((long long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = t;
}

// out[b] = { mean squared error, mean SSIM }: one workgroup per image, fixed order
__global__ __launch_bounds__(256) void eval_finalize_kernel(const double* __restrict__ pm, int nm, double cm,
                                                            const double* __restrict__ ps, int ns, double cs,
                                                            double* __restrict__ out) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nm; i += 256) a += pm[(long long)b * nm + i];
  for (int i = threadIdx.x; i < ns; i += 256) c += ps[(long long)b * ns + i];
  const double ta = block_sum_256(a, red);
  const double tc = block_sum_256(c, red);
  if (threadIdx.x == 0) { out[2 * b] = ta / cm; out[2 * b + 1] = tc / cs; }
}

constexpr int MSE_CHUNKS = 64;

struct Geo { int Hc, Wc, nh, nw, Hm, Wm, tx, ty; };
static bool geometry(int H, int W, int crop, Geo* g) {
  if (crop < 0 || H - 2 * crop < 1 || W - 2 * crop < 1) return false;
  g->Hc = H - 2 * crop; g->Wc = W - 2 * crop;
  g->nh = g->Hc >= WIN ? WIN : 1; g->nw = g->Wc >= WIN ? WIN : 1;
  g->Hm = g->Hc - g->nh + 1; g->Wm = g->Wc - g->nw + 1;
  g->tx = (g->Wm + TS - 1) / TS; g->ty = (g->Hm + TS - 1) / TS;
  return true;
}

}  // namespace

extern "C" size_t m2t_eval_metrics_scratch_bytes(int B, int H, int W, int crop) {
  Geo g;
  if (B < 1 || !geometry(H, W, crop, &g)) return 0;
  return sizeof(double) * (size_t)B * ((size_t)MSE_CHUNKS + (size_t)g.tx * g.ty);
}

extern "C" int m2t_eval_metrics(const float* sr, const float* hr, int B, int H, int W, int crop, float rgb_range,
                                const float* window_host, void* scratch, double* out, void* stream) {
  Geo g;
  if (!sr || !hr || !scratch || !out || B < 1 || B > 65535 || !geometry(H, W, crop, &g))
    return m2t_set_error(M2T_ERR_ARG, "m2t_eval_metrics: bad argument (need H, W > 2*crop)");
  hipStream_t st = (hipStream_t)stream;
  const int times255 = rgb_range == 1.0f ? 1 : 0;
  // the dependency's window: exp(-(i-5)^2 / (2 sigma^2)) in fp32, normalised in fp32
  SsimWin win;
  float gf[WIN], sum = 0.f;
  for (int i = 0; i < WIN; ++i) { const float c = (float)(i - WIN / 2); gf[i] = (float)exp(-(double)(c * c) / (2.0 * 1.5 * 1.5)); }
  for (int i = 0; i < WIN; ++i) sum += gf[i];
  for (int i = 0; i < WIN; ++i) win.g[i] = window_host ? (double)window_host[i] : (double)(gf[i] / sum);
  double* pm = (double*)scratch;
  double* ps = pm + (size_t)B * MSE_CHUNKS;
  eval_mse_y_kernel<<<dim3(MSE_CHUNKS, B), 256, 0, st>>>(sr, hr, H, W, crop, times255, pm);
  M2T_LAUNCH_CHECK();
  eval_ssim_y_kernel<<<dim3(g.tx, g.ty, B), 256, 0, st>>>(sr, hr, H, W, crop, times255, g.nh, g.nw, win, ps);
  M2T_LAUNCH_CHECK();
  eval_finalize_kernel<<<B, 256, 0, st>>>(pm, MSE_CHUNKS, (double)g.Hc * g.Wc, ps, g.tx * g.ty, (double)g.Hm * g.Wm, out);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// GMSD of the eval loop: piq.gmsd(hr, sr, data_range=1., reduction='none') (test.py:98; Xue et al. 2014 as the `piq`
// package implements it -- PARITY UNPINNED: `piq` is not vendored by the reference nor installed here, the oracle in
// tests/test_metrics.py restates the published algorithm in fp64):
//   gray = 0.299 R + 0.587 G + 0.114 B (rgb2yiq Y) of x / data_range; zero-pad bottom / right to even size;
//   2 x 2 average pooling; Prewitt gradients ([-1 0 1] x 3 / 3 and its transpose, zero padding 1);
//   gm = sqrt(gx^2 + gy^2); gms = (2 gm_x gm_y + t) / (gm_x^2 + gm_y^2 + t), t = 170 / 255^2;
//   GMSD = sqrt(mean((gms - mean(gms))^2)) over the pooled map.
// The map arithmetic is fp32 like the dependency's; the two spatial moments are accumulated in fp64.
// ---------------------------------------------------------------------------------------
namespace {
constexpr int GMSD_CHUNKS = 256;
__device__ __forceinline__ float gmsd_pooled(const float* img, long long plane, int H, int W, int py, int px, int hp, int wp, float inv_range) {
  if (py < 0 || py >= hp || px < 0 || px >= wp) return 0.f;            // zero padding of the Prewitt convolution
  float a = 0.f;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int y = 2 * py + dy, x = 2 * px + dx;
      float g = 0.f;
      if (y < H && x < W) {                                              // the pad row / column of an odd-sized image is zero
        const long long o = (long long)y * W + x;
        g = 0.299f * (img[o] * inv_range) + 0.587f * (img[plane + o] * inv_range) + 0.114f * (img[2 * plane + o] * inv_range);
      }
      a += g;
    }
  return a * 0.25f;
}
__global__ __launch_bounds__(256) void eval_gmsd_kernel(const float* __restrict__ xr, const float* __restrict__ yr, int H, int W, int hp, int wp,
                                                        float inv_range, double* __restrict__ partial) {
  __shared__ double red[4];
  const int b = blockIdx.y;
  const long long plane = (long long)H * W, n = (long long)hp * wp;
  const float* xs = xr + (long long)b * 3 * plane;
  const float* ys = yr + (long long)b * 3 * plane;
  const float t = 170.0f / (255.0f * 255.0f);
  double s1 = 0.0, s2 = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int py = (int)(i / wp), px = (int)(i % wp);
    float gm[2];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      const float* img = im ? ys : xs;
      float v[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) v[dy][dx] = gmsd_pooled(img, plane, H, W, py + dy - 1, px + dx - 1, hp, wp, inv_range);
      const float gx = ((v[0][2] - v[0][0]) + (v[1][2] - v[1][0]) + (v[2][2] - v[2][0])) * (1.0f / 3.0f);
      const float gy = ((v[2][0] - v[0][0]) + (v[2][1] - v[0][1]) + (v[2][2] - v[0][2])) * (1.0f / 3.0f);
      gm[im] = sqrtf(gx * gx + gy * gy);
    }
    const float gms = (2.0f * gm[0] * gm[1] + t) / (gm[0] * gm[0] + gm[1] * gm[1] + t);
    s1 += (double)gms;
    s2 += (double)gms * (double)gms;
  }
  const double a = block_sum_256(s1, red);
  const double q = block_sum_256(s2, red);
  if (threadIdx.x == 0) { partial[((long long)b * GMSD_CHUNKS + blockIdx.x) * 2] = a; partial[((long long)b * GMSD_CHUNKS + blockIdx.x) * 2 + 1] = q; }
}
__global__ __launch_bounds__(256) void eval_gmsd_finalize_kernel(const double* __restrict__ partial, double n, double* __restrict__ out) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  const double a = block_sum_256(partial[((long long)b * GMSD_CHUNKS + threadIdx.x) * 2], red);
  const double q = block_sum_256(partial[((long long)b * GMSD_CHUNKS + threadIdx.x) * 2 + 1], red);
  if (threadIdx.x == 0) {
    const double mean = a / n;
    out[b] = sqrt(fmax(q / n - mean * mean, 0.0));
  }
}
}  // namespace

extern "C" size_t m2t_eval_gmsd_scratch_bytes(int B) { return B < 1 ? 0 : sizeof(double) * 2 * (size_t)B * GMSD_CHUNKS; }
extern "C" int m2t_eval_gmsd(const float* x, const float* y, int B, int H, int W, float data_range, void* scratch, double* out, void* stream) {
  if (!x || !y || !scratch || !out || B < 1 || B > 65535 || H < 2 || W < 2 || !(data_range > 0.f))
    return m2t_set_error(M2T_ERR_ARG, "m2t_eval_gmsd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int hp = (H + (H & 1 ? 1 : (W & 1))) / 2, wp = (W + (W & 1 ? 1 : (H & 1))) / 2;   // piq pads BOTH dims by max(H % 2, W % 2)
  eval_gmsd_kernel<<<dim3(GMSD_CHUNKS, B), 256, 0, st>>>(x, y, H, W, hp, wp, 1.0f / data_range, (double*)scratch);
  M2T_LAUNCH_CHECK();
  eval_gmsd_finalize_kernel<<<B, 256, 0, st>>>((const double*)scratch, (double)hp * wp, out);
  M2T_LAUNCH_CHECK();
  return 0;
}
