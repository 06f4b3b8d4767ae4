// k_fsim.hip -- FSIMc of the eval loop on the device (test.py:95-96: piq.fsim(hr, sr, data_range=1., reduction='none');
// piq 0.8.0 per environment.yml:118, absent from the reference tree and from this image: the published algorithm -- Zhang et al.,
// IEEE TIP 2011, phase congruency after Kovesi's phasecong2 as piq/fsim.py implements it -- is restated in
// oracle/fsim_oracle.py, which these kernels are tested against; PARITY UNPINNED).
//
// Everything in fp64 (the reference runs piq in fp32; the eps regulariser keeps piq's float32 value).  The 2-D transforms are
// explicit DFTs, one line per workgroup (line + twiddle table in LDS, every thread sums one output frequency): the pooled eval images
// are ~256 x 256, 50 transforms per image pair = 13 GFLOP -- milliseconds, off every hot path, any size (no power-of-two FFT).
//   prep      x / range * 255, k x k average pooling (k = max(1, round(min(H, W) / 256))), RGB -> YIQ
//   filters   4 orientations x 4 scales of log-Gabor x angular spread x low-pass, zero frequency at the corner
//   consts    per orientation: sum F(scale 0)^2; sum_s sum (ifft F_s)^2 h w; sum_{s<t} sum (ifft F_s)(ifft F_t) h w
//   pc        even / odd responses = ifft2(fft2(Y) F); per orientation energy, amplitude sums, scale-0 amplitude^2
//   median    torch.median (lower median) of the scale-0 amplitude^2 per orientation by a 64-bit radix select -> noise threshold T
//   final     PC maps, Scharr gradient maps, similarity maps, chroma terms, sum(S PCm) / sum(PCm)
#include "m2t_kernels.h"
#include "../../include/m2t.h"

namespace {

typedef double2 cplx;
constexpr int FS_NF = 16, FS_NO = 4, FS_NS = 4;
constexpr double FS_EPS = 1.1920928955078125e-07;        // torch.finfo(torch.float32).eps

__device__ __forceinline__ double block_sum(double v, double* red) {      // 256 threads; every thread receives the sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// normalised frequency coordinate of index i of an axis of length n AFTER piq's ifftshift (roll by -(n / 2))
__device__ __forceinline__ double fs_coord(int i, int n) {
  const int j = (i + n / 2) % n;                            // index before the roll
  return (n & 1) ? ((double)j - (n - 1) / 2.0) / (n - 1) : ((double)j - n / 2) / n;
}

__global__ void __launch_bounds__(256) fsim_prep_kernel(const float* __restrict__ img, double* __restrict__ yiq, int H, int W, int h, int w, int k,
                                                        double scale) {
  const long long n = (long long)h * w, plane = (long long)H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int py = (int)(i / w), px = (int)(i % w);
    double c[3] = {0.0, 0.0, 0.0};
    for (int dy = 0; dy < k; ++dy)
      for (int dx = 0; dx < k; ++dx) {
        const long long o = (long long)(py * k + dy) * W + px * k + dx;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch] += (double)img[ch * plane + o] * scale;
      }
    const double inv = 1.0 / (k * k);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) c[ch] *= inv;
    yiq[i] = 0.299 * c[0] + 0.587 * c[1] + 0.114 * c[2];
    yiq[n + i] = 0.5959 * c[0] - 0.2746 * c[1] - 0.3213 * c[2];
    yiq[2 * n + i] = 0.2115 * c[0] - 0.5227 * c[1] + 0.3112 * c[2];
  }
}

__global__ void __launch_bounds__(256) fsim_twiddle_kernel(cplx* __restrict__ tw, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    double s, c;
    sincospi(2.0 * (double)i / (double)n, &s, &c);
    tw[i] = make_double2(c, s);                              // exp(+2 pi i k / n); the forward transform conjugates
  }
}

__global__ void __launch_bounds__(256) fsim_filter_kernel(double* __restrict__ F, int h, int w) {
  const long long n = (long long)h * w;
  const double theta_sigma = M_PI / (FS_NO * 1.2);
  const double lsf = 2.0 * log(0.55) * log(0.55);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int u = (int)(i / w), v = (int)(i % w);
    const double gx = fs_coord(u, h), gy = fs_coord(v, w);
    double radius = sqrt(gx * gx + gy * gy);
    const double lp = 1.0 / (1.0 + pow(radius / 0.45, 30.0));
    const double theta = atan2(-gy, gx);
    if (i == 0) radius = 1.0;
    const double st = sin(theta), ct = cos(theta);
    double lg[FS_NS];
#pragma unroll
    for (int s = 0; s < FS_NS; ++s) {
      const double omega0 = 1.0 / (6.0 * (double)(1 << s));
      const double l = log(radius / omega0);
      lg[s] = (i == 0) ? 0.0 : exp(-(l * l) / lsf) * lp;
    }
#pragma unroll
    for (int o = 0; o < FS_NO; ++o) {
      const double angl = o * M_PI / FS_NO;
      const double ds = st * cos(angl) - ct * sin(angl), dc = ct * cos(angl) + st * sin(angl);
      const double dth = fabs(atan2(ds, dc));
      const double spread = exp(-(dth * dth) / (2.0 * theta_sigma * theta_sigma));
#pragma unroll
      for (int s = 0; s < FS_NS; ++s) F[(long long)(o * FS_NS + s) * n + i] = spread * lg[s];
    }
  }
}

// DFT along one axis: line l of batch item blockIdx.y, N points at stride `es` starting at base + (l / lpo) * os + (l % lpo) * ls.
// in_re / in_im (in_im may be null: real input), optional real multiplier `mul` with the same indexing plus mul_bstride per batch
// item; out: complex, same geometry.  sign = -1 forward (conjugated twiddles), +1 inverse; `scale` multiplies the result.
struct DftArgs {
  const double* in_re; const double* in_im; long long in_bstride; int in_cplx;       // in_cplx: interleaved (re, im) input in in_re
  const double* mul; long long mul_bstride;
  cplx* out; long long out_bstride;
  const cplx* tw;
  int N, nlines, lpo; long long os, ls, es; double sign, scale;
};
__global__ void __launch_bounds__(256) fsim_dft_kernel(DftArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cplx* line = reinterpret_cast<cplx*>(smem);
  cplx* tws = line + a.N;
  const int l = blockIdx.x, bi = blockIdx.y;
  const long long base = (long long)(l / a.lpo) * a.os + (long long)(l % a.lpo) * a.ls;
  for (int x = threadIdx.x; x < a.N; x += 256) {
    const long long idx = base + (long long)x * a.es;
    double re, im;
    if (a.in_cplx) { const cplx v = reinterpret_cast<const cplx*>(a.in_re)[(long long)bi * a.in_bstride + idx]; re = v.x; im = v.y; }
    else { re = a.in_re[(long long)bi * a.in_bstride + idx]; im = a.in_im ? a.in_im[(long long)bi * a.in_bstride + idx] : 0.0; }
    if (a.mul) { const double m = a.mul[(long long)bi * a.mul_bstride + idx]; re *= m; im *= m; }
    line[x] = make_double2(re, im);
    const cplx t = a.tw[x];
    tws[x] = make_double2(t.x, a.sign * t.y);
  }
  __syncthreads();
  for (int u = threadIdx.x; u < a.N; u += 256) {
    double sr = 0.0, si = 0.0;
    int k = 0;
    for (int x = 0; x < a.N; ++x) {
      const cplx v = line[x], t = tws[k];
      sr += v.x * t.x - v.y * t.y;
      si += v.x * t.y + v.y * t.x;
      k += u;
      if (k >= a.N) k -= a.N;
    }
    a.out[(long long)bi * a.out_bstride + base + (long long)u * a.es] = make_double2(sr * a.scale, si * a.scale);
  }
}

// per orientation o (blockIdx.x): em_n = sum F[o, 0]^2, sum_an2 = sum_s sum fi^2, sum_ai_aj = sum_{s < t} sum fi_s fi_t, fi = Re(ifft2 F) sqrt(h w)
__global__ void __launch_bounds__(256) fsim_consts_kernel(const double* __restrict__ F, const cplx* __restrict__ fi, long long n, double root_hw,
                                                          double* __restrict__ consts) {
  __shared__ double red[4];
  const int o = blockIdx.x;
  double em = 0.0, an2 = 0.0, aiaj = 0.0;
  for (long long i = threadIdx.x; i < n; i += 256) {
    const double f0 = F[(long long)(o * FS_NS) * n + i];
    em += f0 * f0;
    double v[FS_NS];
#pragma unroll
    for (int s = 0; s < FS_NS; ++s) { v[s] = fi[(long long)(o * FS_NS + s) * n + i].x * root_hw; an2 += v[s] * v[s]; }
#pragma unroll
    for (int s = 0; s < FS_NS - 1; ++s)
#pragma unroll
      for (int t = s + 1; t < FS_NS; ++t) aiaj += v[s] * v[t];
  }
  em = block_sum(em, red); an2 = block_sum(an2, red); aiaj = block_sum(aiaj, red);
  if (threadIdx.x == 0) { consts[o * 3] = em; consts[o * 3 + 1] = an2; consts[o * 3 + 2] = aiaj; }
}

// per pixel: energy[o] (before the noise threshold), the sum of all 16 amplitudes, the scale-0 amplitude^2 of each orientation
__global__ void __launch_bounds__(256) fsim_pc_accum_kernel(const cplx* __restrict__ eo, long long n, double* __restrict__ energy, double* __restrict__ ansum,
                                                            double* __restrict__ e2) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    double at = 0.0;
#pragma unroll
    for (int o = 0; o < FS_NO; ++o) {
      cplx v[FS_NS];
      double se = 0.0, so = 0.0;
#pragma unroll
      for (int s = 0; s < FS_NS; ++s) { v[s] = eo[(long long)(o * FS_NS + s) * n + i]; se += v[s].x; so += v[s].y; at += sqrt(v[s].x * v[s].x + v[s].y * v[s].y); }
      const double xe = sqrt(se * se + so * so) + FS_EPS;
      const double me = se / xe, mo = so / xe;
      double en = 0.0;
#pragma unroll
      for (int s = 0; s < FS_NS; ++s) en += v[s].x * me + v[s].y * mo - fabs(v[s].x * mo - v[s].y * me);
      energy[(long long)o * n + i] = en;
      e2[(long long)o * n + i] = v[0].x * v[0].x + v[0].y * v[0].y;
    }
    ansum[i] = at;
  }
}

// lower median (torch.median) of n non-negative doubles by radix select on the bit pattern, 8 bits per pass; then the noise
// threshold T of the orientation.  One workgroup per orientation.
__global__ void __launch_bounds__(256) fsim_threshold_kernel(const double* __restrict__ e2, long long n, const double* __restrict__ consts,
                                                             double* __restrict__ T) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long prefix_s;
  __shared__ long long rank_s;
  const int o = blockIdx.x;
  const unsigned long long* v = reinterpret_cast<const unsigned long long*>(e2 + (long long)o * n);
  if (threadIdx.x == 0) { prefix_s = 0ull; rank_s = (n - 1) / 2; }
  __syncthreads();
  for (int pass = 7; pass >= 0; --pass) {
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long prefix = prefix_s;
    const unsigned long long himask = pass == 7 ? 0ull : (~0ull << (8 * (pass + 1)));
    for (long long i = threadIdx.x; i < n; i += 256) {
      const unsigned long long b = v[i];
      if ((b & himask) == prefix) atomicAdd(&hist[(unsigned)((b >> (8 * pass)) & 255ull)], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      long long r = rank_s;
      int d = 0;
      for (; d < 256; ++d) { if (r < (long long)hist[d]) break; r -= hist[d]; }
      rank_s = r;
      prefix_s = prefix | ((unsigned long long)d << (8 * pass));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double median = __longlong_as_double((long long)prefix_s);
    const double mean_e2n = -median / log(0.5);
    const double noise_power = mean_e2n / consts[o * 3];
    const double ne2 = 2.0 * noise_power * consts[o * 3 + 1] + 4.0 * noise_power * consts[o * 3 + 2];
    const double tau = sqrt(ne2 / 2.0);
    T[o] = (tau * sqrt(M_PI / 2.0) + 2.0 * sqrt((2.0 - M_PI / 2.0) * tau * tau)) / 1.7;
  }
}

__device__ __forceinline__ double fs_sim(double a, double b, double c) { return (2.0 * a * b + c) / (a * a + b * b + c); }
__device__ __forceinline__ double fs_grad(const double* __restrict__ y, int py, int px, int h, int w) {
  double gx = 0.0, gy = 0.0;
  const double k[3][3] = {{-3.0 / 16, 0.0, 3.0 / 16}, {-10.0 / 16, 0.0, 10.0 / 16}, {-3.0 / 16, 0.0, 3.0 / 16}};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int yy = py + a - 1, xx = px + b - 1;
      const double v = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? y[(long long)yy * w + xx] : 0.0;
      gx += k[a][b] * v;
      gy += k[b][a] * v;
    }
  return sqrt(gx * gx + gy * gy);
}
// yiq: [2 images][3][n]; energy [2][4][n]; ansum [2][n]; T [2][4] -> partial sums (score, pc_max) per workgroup
__global__ void __launch_bounds__(256) fsim_final_kernel(const double* __restrict__ yiq, const double* __restrict__ energy, const double* __restrict__ ansum,
                                                         const double* __restrict__ T, int h, int w, double* __restrict__ partial) {
  __shared__ double red[4];
  const long long n = (long long)h * w;
  double ssum = 0.0, psum = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int py = (int)(i / w), px = (int)(i % w);
    double pc[2], gm[2];
#pragma unroll
    for (int im = 0; im < 2; ++im) {
      double e = 0.0;
#pragma unroll
      for (int o = 0; o < FS_NO; ++o) e += fmax(energy[((long long)im * FS_NO + o) * n + i] - T[im * FS_NO + o], 0.0);
      pc[im] = (e + FS_EPS) / (ansum[(long long)im * n + i] + FS_EPS);
      gm[im] = fs_grad(yiq + (long long)im * 3 * n, py, px, h, w);
    }
    const double pcm = fmax(pc[0], pc[1]);
    double sc = fs_sim(gm[0], gm[1], 160.0) * fs_sim(pc[0], pc[1], 0.85) * pcm;
    const double si = fs_sim(yiq[n + i], yiq[3 * n + n + i], 200.0), sq = fs_sim(yiq[2 * n + i], yiq[3 * n + 2 * n + i], 200.0);
    sc *= pow(fabs(si * sq), 0.03);
    ssum += sc;
    psum += pcm;
  }
  ssum = block_sum(ssum, red);
  psum = block_sum(psum, red);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = ssum; partial[2 * blockIdx.x + 1] = psum; }
}
__global__ void __launch_bounds__(256) fsim_finish_kernel(const double* __restrict__ partial, int nblk, double* __restrict__ out) {
  __shared__ double red[4];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) { a += partial[2 * i]; b += partial[2 * i + 1]; }
  a = block_sum(a, red);
  b = block_sum(b, red);
  if (threadIdx.x == 0) out[0] = a / b;
}

constexpr int FS_BLOCKS = 256;
struct FsLayout {
  int h, w, k; long long n;
  size_t yiq, F, tw_h, tw_w, imfft, tmp, eo, energy, ansum, e2, consts, T, partial, total;
};
FsLayout fs_layout(int H, int W) {
  FsLayout L;
  const double r = (double)std::min(H, W) / 256.0;
  L.k = std::max(1, (int)nearbyint(r));                          // Python round(): half to even, as nearbyint in the default rounding mode
  L.h = H / L.k; L.w = W / L.k; L.n = (long long)L.h * L.w;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  L.yiq = take(sizeof(double) * 2 * 3 * L.n);
  L.F = take(sizeof(double) * FS_NF * L.n);
  L.tw_h = take(sizeof(cplx) * L.h);
  L.tw_w = take(sizeof(cplx) * L.w);
  L.imfft = take(sizeof(cplx) * L.n);
  L.tmp = take(sizeof(cplx) * FS_NF * L.n);
  L.eo = take(sizeof(cplx) * FS_NF * L.n);
  L.energy = take(sizeof(double) * 2 * FS_NO * L.n);
  L.ansum = take(sizeof(double) * 2 * L.n);
  L.e2 = take(sizeof(double) * FS_NO * L.n);
  L.consts = take(sizeof(double) * FS_NO * 3);
  L.T = take(sizeof(double) * 2 * FS_NO);
  L.partial = take(sizeof(double) * 2 * FS_BLOCKS);
  L.total = o;
  return L;
}

}  // namespace

extern "C" size_t m2t_eval_fsim_scratch_bytes(int H, int W) {
  if (H < 8 || W < 8) return 0;
  return fs_layout(H, W).total;
}

// x, y: fp32 NCHW [B][3][H][W] device tensors in [0, data_range] (test.py:95 passes hr, sr); out: fp64 [B] FSIMc per image pair.
extern "C" int m2t_eval_fsim(const float* x, const float* y, int B, int H, int W, float data_range, void* scratch, double* out, void* stream) {
  if (!x || !y || !scratch || !out || B < 1 || H < 8 || W < 8 || !(data_range > 0.f)) return m2t_set_error(M2T_ERR_ARG, "m2t_eval_fsim: bad argument");
  const FsLayout L = fs_layout(H, W);
  if (L.h > 4096 || L.w > 4096) return m2t_set_error(M2T_UNSUPPORTED, "m2t_eval_fsim: pooled image larger than 4096 (DFT line must fit in LDS)");
  hipStream_t st = (hipStream_t)stream;
  char* S = (char*)scratch;
  double* yiq = (double*)(S + L.yiq); double* F = (double*)(S + L.F);
  cplx* tw_h = (cplx*)(S + L.tw_h); cplx* tw_w = (cplx*)(S + L.tw_w);
  cplx* imfft = (cplx*)(S + L.imfft); cplx* tmp = (cplx*)(S + L.tmp); cplx* eo = (cplx*)(S + L.eo);
  double* energy = (double*)(S + L.energy); double* ansum = (double*)(S + L.ansum); double* e2 = (double*)(S + L.e2);
  double* consts = (double*)(S + L.consts); double* T = (double*)(S + L.T); double* partial = (double*)(S + L.partial);
  const int h = L.h, w = L.w; const long long n = L.n;
  const int nb = (int)std::min<long long>(FS_BLOCKS, (n + 255) / 256);
  const size_t sh_w = sizeof(cplx) * 2 * w, sh_h = sizeof(cplx) * 2 * h;
  if (int rc = m2t_ensure_dynamic_lds((const void*)fsim_dft_kernel, (int)std::max(sh_w, sh_h))) return rc;
  // 2-D transform of `batch` maps: rows (axis 1) then columns (axis 0)
  auto dft2 = [&](const double* in_re, long long in_bs, int in_cplx, const double* mul, long long mul_bs, cplx* mid, cplx* dst, int batch, double sign,
                  double scale) -> int {
    DftArgs a{in_re, nullptr, in_bs, in_cplx, mul, mul_bs, mid, n, tw_w, w, h, h, 0, (long long)w, 1, sign, 1.0};
    hipLaunchKernelGGL(fsim_dft_kernel, dim3(h, batch), dim3(256), sh_w, st, a);
    DftArgs b{reinterpret_cast<const double*>(mid), nullptr, n, 1, nullptr, 0, dst, n, tw_h, h, w, w, 0, 1, (long long)w, sign, scale};
    hipLaunchKernelGGL(fsim_dft_kernel, dim3(w, batch), dim3(256), sh_h, st, b);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : m2t_set_hip_error(e, __FILE__, __LINE__);
  };
  hipLaunchKernelGGL(fsim_twiddle_kernel, dim3((h + 255) / 256), dim3(256), 0, st, tw_h, h);
  hipLaunchKernelGGL(fsim_twiddle_kernel, dim3((w + 255) / 256), dim3(256), 0, st, tw_w, w);
  hipLaunchKernelGGL(fsim_filter_kernel, dim3(nb), dim3(256), 0, st, F, h, w);
  // filter constants: ifft2 of the 16 (real) filters
  if (int rc = dft2(F, n, 0, nullptr, 0, tmp, eo, FS_NF, 1.0, 1.0 / (double)n)) return rc;
  hipLaunchKernelGGL(fsim_consts_kernel, dim3(FS_NO), dim3(256), 0, st, F, eo, n, sqrt((double)n), consts);
  const double scale = 255.0 / (double)data_range;
  const long long plane3 = 3LL * H * W;
  for (int b = 0; b < B; ++b) {
    for (int im = 0; im < 2; ++im) {
      const float* src = (im ? y : x) + (long long)b * plane3;
      hipLaunchKernelGGL(fsim_prep_kernel, dim3(nb), dim3(256), 0, st, src, yiq + (long long)im * 3 * n, H, W, h, w, L.k, scale);
      if (int rc = dft2(yiq + (long long)im * 3 * n, 0, 0, nullptr, 0, tmp, imfft, 1, -1.0, 1.0)) return rc;
      // even / odd responses of the 16 filters: ifft2(imfft * F_f), the product formed while the rows are loaded
      if (int rc = dft2(reinterpret_cast<const double*>(imfft), 0, 1, F, n, tmp, eo, FS_NF, 1.0, 1.0 / (double)n)) return rc;
      hipLaunchKernelGGL(fsim_pc_accum_kernel, dim3(nb), dim3(256), 0, st, eo, n, energy + (long long)im * FS_NO * n, ansum + (long long)im * n, e2);
      hipLaunchKernelGGL(fsim_threshold_kernel, dim3(FS_NO), dim3(256), 0, st, e2, n, consts, T + im * FS_NO);
    }
    hipLaunchKernelGGL(fsim_final_kernel, dim3(nb), dim3(256), 0, st, yiq, energy, ansum, T, h, w, partial);
    hipLaunchKernelGGL(fsim_finish_kernel, dim3(1), dim3(256), 0, st, partial, nb, out + b);
  }
  M2T_LAUNCH_CHECK();
  return 0;
}
