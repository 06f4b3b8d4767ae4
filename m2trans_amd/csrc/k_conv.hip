// k_conv.hip -- the 3x3 convolutions of M2Trans on gfx950.
//
//   head   3->64  reflect pad, bias   (models/M2Trans_network.py:34,63)   VALU, K = 27
//   ff     64->64 zero pad, bias, +x  (models/M2Trans_network.py:124-126,164) MFMA implicit GEMM
//          (also its data gradient, with the flipped/transposed packed weights) + MFMA wgrad
//   tail   64->3  reflect pad, no bias, input GELU fused on load (:44-48,54-55)  VALU, N = 3
//
// NHWC activations; a workgroup stages a pixel tile + halo once in LDS and re-uses it for
// all 9 taps.
#include "m2t_kernels.h"

// =======================================================================================
// head conv forward
// x: NCHW fp32 [B][3][H0][W0]; logical image = x reflect-padded right/bottom to [H][W]
// (check_image_size, :78-86) and then reflect-padded by 1 for the conv.
// =======================================================================================
__device__ __forceinline__ int head_src(int i, int n, int n0) {
  i = reflect_idx(i, n);                 // the conv's own reflect padding on the padded image
  return (i < n0) ? i : (2 * n0 - 2 - i);  // the right/bottom reflect pad to a multiple of 32
}

// One thread = TWO horizontally adjacent pixels x 16 output channels (round 5; one pixel before): a broadcast LDS read of 16 weights
// now feeds 32 FMAs instead of 16 (the kernel issued 108 ds_read_b128 for 432 FMAs per thread), the two pixels share two of their
// four input columns (36 loads for two pixels instead of 54), and half as many workgroups stage the 27 x 64 weights.  Per output
// the operation order is unchanged -- bias, then (ic, ky, kx) with fmaf -- so the bits are the same.
template <typename T>
__global__ void __launch_bounds__(256) head_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out, int B,
                                                            int H0, int W0, int H, int W) {
  __shared__ float ws[27][64];   // [ic*9 + tap][oc]
  __shared__ float bs[64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) {
    const int oc = i / 27, q = i % 27;
    ws[q][oc] = w[i];            // torch [oc][ic][ky][kx] flattened = oc*27 + (ic*9 + tap)
  }
  if (threadIdx.x < 64) bs[threadIdx.x] = bias[threadIdx.x];
  __syncthreads();
  const int Wh = W >> 1;                                   // pixel pairs per row (W is a multiple of 32)
  const long long total = (long long)B * H * Wh * 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int og = (int)(t & 3);
    const long long pp = t >> 2;
    const int xx = 2 * (int)(pp % Wh);
    const long long q = pp / Wh;
    const int yy = (int)(q % H);
    const int b = (int)(q / H);
    // the 3 x 4 input patch of the pair per channel: columns xx - 1 .. xx + 2
    int sx[4], sy[3];
#pragma unroll
    for (int c = 0; c < 4; ++c) sx[c] = head_src(xx + c - 1, W, W0);
#pragma unroll
    for (int r = 0; r < 3; ++r) sy[r] = head_src(yy + r - 1, H, H0);
    float v[3][3][4];
#pragma unroll
    for (int ic = 0; ic < 3; ++ic)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[ic][r][c] = x[(((long long)b * 3 + ic) * H0 + sy[r]) * W0 + sx[c]];
    float acc[2][16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = bs[og * 16 + e]; acc[1][e] = acc[0][e]; }
#pragma unroll
    for (int ic = 0; ic < 3; ++ic)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wr = &ws[ic * 9 + ky * 3 + kx][og * 16];
          const float v0 = v[ic][ky][kx], v1 = v[ic][ky][kx + 1];
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const float we = wr[e];
            acc[0][e] = fmaf(v0, we, acc[0][e]);
            acc[1][e] = fmaf(v1, we, acc[1][e]);
          }
        }
    const long long pix = ((long long)b * H + yy) * W + xx;
    T* o = out + ((long long)og * B * H * W + pix) * 16;                  // P64: chunk og is a dense plane; the pair is 64 contiguous bytes (bf16)
    store16f(o, acc[0]);
    store16f(o + 16, acc[1]);
  }
}
int launch_head_conv_fwd(int dt, const float* x, const float* w, const float* b, void* out, int B, int H0, int W0,
                         int H, int W, hipStream_t st) {
  if (W & 1) return m2t_set_error(-2, "head_conv_fwd: the padded width must be even");
  const long long total = (long long)B * H * (W / 2) * 4;
  const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(head_conv_fwd_kernel<float>, dim3(g), dim3(256), 0, st, x, w, b, (float*)out, B, H0, W0, H, W);
  else hipLaunchKernelGGL(head_conv_fwd_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, w, b, (bf16_t*)out, B, H0, W0, H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}

// im2col of the (padded, reflect-extended) head input: cols[pixel][32], column k = ic*9 + ky*3 + kx (27 used,
// the rest 0), so the head weight gradient is one [64 x M] x [M x 32] product for the generic wgrad GEMM
// (gradient rows on the MFMA rows, contraction over pixels) and the bias gradient rides along in it.
template <typename T>
__global__ void __launch_bounds__(256) head_im2col_kernel(const float* __restrict__ x, T* __restrict__ cols, int B, int H0,
                                                          int W0, int H, int W) {
  const long long total = (long long)B * H * W * 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(t & 3);
    const long long pix = t >> 2;
    const int xx = (int)(pix % W);
    const long long q = pix / W;
    const int yy = (int)(q % H);
    const int b = (int)(q / H);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * j + e;
      v[e] = 0.f;
      if (k < 27) {
        const int ic = k / 9, tap = k - 9 * ic, ky = tap / 3, kx = tap - 3 * ky;
        const int sy = head_src(yy + ky - 1, H, H0), sx = head_src(xx + kx - 1, W, W0);
        v[e] = x[(((long long)b * 3 + ic) * H0 + sy) * W0 + sx];
      }
    }
    store8f(cols + pix * 32 + 8 * j, v);
  }
}
int launch_head_im2col(int dt, const float* x, void* cols, int B, int H0, int W0, int H, int W, hipStream_t st) {
  const long long total = (long long)B * H * W * 4;
  const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(head_im2col_kernel<float>, dim3(g), dim3(256), 0, st, x, (float*)cols, B, H0, W0, H, W);
  else hipLaunchKernelGGL(head_im2col_kernel<bf16_t>, dim3(g), dim3(256), 0, st, x, (bf16_t*)cols, B, H0, W0, H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// 64 -> 64 3x3 conv, zero padding: implicit GEMM on the matrix cores.  All feature maps are P64.
// Workgroup = 8 x 16 output pixels x 64 output channels; 4 waves, wave w owns pixel rows
// 2w, 2w+1 (two 16-pixel m-tiles) x 4 channel tiles.  LDS: input halo tile 10x18x64 and the
// current tap's 64x64 weight slice.  Lane (pixel = lane&15, g = lane>>4) ends with 16
// consecutive output channels at 16 g.
// =======================================================================================
#ifndef M2T_CONV_STAMP
#define M2T_CONV_STAMP(i) do { } while (0)      // scratch/bench_conv.hip defines it to record s_memtime per phase
#endif
#define C3_TH 8
#define C3_TW 16
#define C3_LD 72   // 64 channels + 8 pad

template <typename T>
__global__ void __launch_bounds__(256) conv3x3_c64_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                          const float* __restrict__ bias, const T* __restrict__ res1,
                                                          const T* __restrict__ res2, T* __restrict__ y, int H, int W) {
  __shared__ __attribute__((aligned(16))) T Xs[(C3_TH + 2) * (C3_TW + 2)][C3_LD];
  __shared__ __attribute__((aligned(16))) T Ws[64][C3_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * C3_TW, y0 = blockIdx.y * C3_TH, b = blockIdx.z;
  const long long npix = (long long)gridDim.z * H * W;      // x, res1, res2, y are P64: [4][npix][16]
  const long long pb = (long long)b * H * W;

  // stage the halo tile (zero outside the image): every load first, then the LDS stores
  {
    constexpr int TOT = (C3_TH + 2) * (C3_TW + 2) * 8, ITEMS = (TOT + 255) / 256;
    Frag8<T> f[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx & 7, p = idx >> 3;
      const int py = p / (C3_TW + 2), px = p - py * (C3_TW + 2);
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      f[it] = frag_zero<T>();
      if (idx < TOT && gy >= 0 && gy < H && gx >= 0 && gx < W) f[it] = load8(x + p64(npix, pb + (long long)gy * W + gx, cv * 8));
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      if (idx < TOT) store8(&Xs[idx >> 3][(idx & 7) * 8], f[it]);
    }
  }
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // weight slice of the next tap is fetched into registers while the current tap is multiplied
  Frag8<T> wreg[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int idx = tid + it * 256;
    wreg[it] = load8(wp + ((long long)(idx >> 3)) * 64 + (idx & 7) * 8);
  }
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    __syncthreads();   // previous tap's reads of Ws done (and, tap 0, Xs staged)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      store8(&Ws[idx >> 3][(idx & 7) * 8], wreg[it]);
    }
    __syncthreads();
    if (tap < 8) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;
        wreg[it] = load8(wp + ((long long)(tap + 1) * 64 + (idx >> 3)) * 64 + (idx & 7) * 8);
      }
    }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      Frag8<T> xf[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        xf[mt] = load8(&Xs[(2 * wv + mt + ky) * (C3_TW + 2) + lr + kx][kc * 32 + g * 8]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int nl = 16 * (lr >> 2) + 4 * nt + (lr & 3);
        const Frag8<T> wf = load8(&Ws[nl][kc * 32 + g * 8]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma16(acc[mt][nt], wf, xf[mt]);
      }
    }
  }
  // epilogue
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int gy = y0 + 2 * wv + mt, gx = x0 + lr;
    const long long off = ((long long)g * npix + pb + (long long)gy * W + gx) * 16;      // channels 16 g .. = plane g
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
    if (bias) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bias[16 * g + e];
    }
    if (res1) {
      float p[16];
      load16f(res1 + off, p);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += p[e];
    }
    if (res2) {
      float p[16];
      load16f(res2 + off, p);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += p[e];
    }
    store16f(y + off, v);
  }
}
// ---------------------------------------------------------------------------------------
// Pipelined variant: same tile, LDS footprint (35 KB -> 4 workgroups per CU), lane mapping and accumulation order as
// conv3x3_c64_kernel (identical bits), but a workgroup walks `tiles_per_block` consecutive tiles and fetches the NEXT
// tile's halo into registers before it starts the current tile's taps.  With one tile per workgroup all ~1024
// co-resident workgroups move in lockstep (everyone loads, then everyone multiplies, then everyone stores: two such
// rounds at B = 16), so HBM idles during the MFMA phase and vice versa; here round k's stores and round k+1's loads
// run under the taps.  Tiles are handed out XCD-aware (see below).  MEASURED (B = 16, 128x128, forward / data gradient
// under the side stream): one tile per workgroup 37.7 / 42.9 us; 2048 workgroups XCD-aware 35.6 / 42.0 us; 1024
// workgroups x 2 tiles XCD-aware 34.3 / 46.9 us; 512 x 4: 43.6 / 48.4 us -- the pipelining itself buys little (the
// co-resident workgroups drift out of phase on their own), the L2-local tile order ~5 %.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256, 4) conv3x3_c64_pipe_kernel(const T* __restrict__ x, const T* __restrict__ wp,
                                                               const float* __restrict__ bias, const T* __restrict__ res1,
                                                               const T* __restrict__ res2, T* __restrict__ y, int B, int H, int W,
                                                               int tiles_per_block, int xcd_order) {
  __shared__ __attribute__((aligned(16))) T Xs[(C3_TH + 2) * (C3_TW + 2)][C3_LD];
  __shared__ __attribute__((aligned(16))) T Ws[64][C3_LD];
  constexpr int TOT = (C3_TH + 2) * (C3_TW + 2) * 8, ITEMS = (TOT + 255) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int tw = W / C3_TW, th = H / C3_TH;
  const int ntiles = B * th * tw;                       // < 2^31: 32-bit tile arithmetic (64-bit division is a software loop)
  const long long npix = (long long)B * H * W;
  // XCD-aware order: workgroup id i runs on XCD i % 8 (round-robin dispatch), so give each XCD one contiguous run of
  // tiles (whole images at B >= 8): neighbouring tiles then share their halo columns / rows in ONE L2
  int chunk = blockIdx.x;
  if (xcd_order) chunk = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int t0 = chunk * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
  Frag8<T> f[ITEMS];
  unsigned fvalid = 0;                    // bit it: item `it` of f lies inside the image (else the halo is zero)
  auto fetch = [&](int t) {               // branch-free: clamped address, validity applied at the LDS store
    const int tx = t % tw, q = t / tw;
    const int ty = q % th;
    const long long pb = (long long)(q / th) * H * W;
    const int x0 = tx * C3_TW, y0 = ty * C3_TH;
    fvalid = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx & 7, p = min(idx >> 3, (C3_TH + 2) * (C3_TW + 2) - 1);
      const int py = p / (C3_TW + 2), px = p - py * (C3_TW + 2);
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      if (idx < TOT && gy >= 0 && gy < H && gx >= 0 && gx < W) fvalid |= 1u << it;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      f[it] = load8(x + p64(npix, pb + (long long)cy * W + cx, cv * 8));
    }
  };
  M2T_CONV_STAMP(0);
  if (t0 < t1) fetch(t0);
  for (int t = t0; t < t1; ++t) {
    const int tx = t % tw, q = t / tw;
    const int ty = q % th;
    const long long pb = (long long)(q / th) * H * W;
    const int x0 = tx * C3_TW, y0 = ty * C3_TH;
    if (t > t0) __syncthreads();          // every wave is done with the previous tile's Xs
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      if (idx < TOT) store8(&Xs[idx >> 3][(idx & 7) * 8], ((fvalid >> it) & 1u) ? f[it] : frag_zero<T>());
    }
    M2T_CONV_STAMP(1);
    if (t + 1 < t1) fetch(t + 1);         // in flight under this tile's 576 MFMAs
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag8<T> wreg[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      wreg[it] = load8(wp + ((long long)(idx >> 3)) * 64 + (idx & 7) * 8);
    }
    #pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = tid + it * 256;
        store8(&Ws[idx >> 3][(idx & 7) * 8], wreg[it]);
      }
      __syncthreads();
      if (tap < 8) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int idx = tid + it * 256;
          wreg[it] = load8(wp + ((long long)(tap + 1) * 64 + (idx >> 3)) * 64 + (idx & 7) * 8);
        }
      }
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        Frag8<T> xf[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          xf[mt] = load8(&Xs[(2 * wv + mt + ky) * (C3_TW + 2) + lr + kx][kc * 32 + g * 8]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int nl = 16 * (lr >> 2) + 4 * nt + (lr & 3);
          const Frag8<T> wf = load8(&Ws[nl][kc * 32 + g * 8]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) mma16(acc[mt][nt], wf, xf[mt]);
        }
      }
    }
    M2T_CONV_STAMP(2);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int gy = y0 + 2 * wv + mt, gx = x0 + lr;
      const long long off = ((long long)g * npix + pb + (long long)gy * W + gx) * 16;
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
      if (bias) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += bias[16 * g + e];
      }
      if (res1) {
        float p[16];
        load16f(res1 + off, p);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += p[e];
      }
      if (res2) {
        float p[16];
        load16f(res2 + off, p);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += p[e];
      }
      store16f(y + off, v);
    }
    M2T_CONV_STAMP(3);
  }
}
// ---------------------------------------------------------------------------------------
// Row-streaming kernel (bf16, round 3; the default): a workgroup owns a 32-pixel-wide column strip segment and walks
// DOWN it two output rows per step.  What bounded the tile kernels above: with one tile per workgroup all ~1024
// co-resident workgroups run in lockstep -- everyone loads, everyone multiplies, everyone fetches residuals, everyone
// stores -- so the memory system idles during the products and vice versa, and every tile re-reads a 10 x 18 halo for
// 8 x 16 outputs (1.41x).  Here
//   * the input rows AND the residual rows stream through LDS rings by LDS-DMA (global_load_lds_dwordx4: no registers, no
//     ds_write, requests stay in flight across barriers): the row pair D steps ahead and the residual rows DR steps ahead
//     are always outstanding; each input row is fetched once per segment (read amplification (RS + 2) / RS x 34 / 32 =
//     1.13 .. 1.2 instead of 1.41);
//   * out-of-image pixels (zero padding of the conv, models/M2Trans_network.py:124-126) come from a zero page in HBM:
//     the source address is selected per lane, no branch around a load;
//   * the packed weights stay in REGISTERS for the whole segment: wave w owns the 32 output channels of half (w & 1)
//     (two permuted 16-row MFMA tiles, so a lane ends with 8 consecutive channels = one 16-byte store) of output row
//     (w >> 1) of the step: 36 A-fragments; per k-step two B-fragment reads from LDS feed four MFMAs;
//   * one barrier per step; stores drain under the next steps.
// The LDS image of a row is the linear P64 image the DMA writes ([plane][36 pixels][16 channels]: 32-byte pixel stride,
// conflict-free for the 16-byte B-fragment reads).  Same products in the same order as conv3x3_c64_kernel
// (tap-major, 32-deep k-steps, bias, res1, res2, one rounding): identical bits.
// The DMA is issued from inline asm ON PURPOSE: with __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2) drains every
// outstanding DMA (s_waitcnt vmcnt(0)) at the next use of any register a plain global load wrote, which serialises the
// ring.  Hidden in asm the compiler neither counts nor waits for it; the loop below contains no compiler-visible load
// at all (weights are forced complete before it, residuals come through LDS), and the single wait per step is
// "all but the n youngest" with n from a running count of what this wave has issued (vmcnt retires in order and counts
// DMA and stores alike).
// ---------------------------------------------------------------------------------------
#ifndef C3R_STAMP
#define C3R_STAMP(i) do { } while (0)   // scratch/bench_conv_rows.hip -DSTAMPS records s_memtime per phase (wave 0 lane 0; counted in `issued`)
#endif
#define C3R_SW 32                       // strip width (output pixels)
#define C3R_PXP 36                      // pixels per LDS plane-row: 34 real (1-pixel halo each side) + 2 padding
#define C3R_ROWB (4 * C3R_PXP * 32)     // bytes of one ring row (4 planes x 36 px x 16 ch x 2 B) = 4608; a row pair = 9 KiB = 9 DMAs
#define C3R_RESB (2 * 4 * C3R_SW * 32)  // residual rows of one step: 2 rows x 4 planes x 32 px x 32 B = 8 KiB = 8 DMAs

__device__ __forceinline__ void c3r_wait_vm(int n) {       // s_waitcnt vmcnt(n): all but the n youngest are done
  switch (n) {
#define C3R_W(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    C3R_W(0) C3R_W(1) C3R_W(2) C3R_W(3) C3R_W(4) C3R_W(5) C3R_W(6) C3R_W(7) C3R_W(8) C3R_W(9) C3R_W(10) C3R_W(11) C3R_W(12)
    C3R_W(13) C3R_W(14) C3R_W(15) C3R_W(16) C3R_W(17) C3R_W(18) C3R_W(19) C3R_W(20) C3R_W(21) C3R_W(22) C3R_W(23) C3R_W(24)
    C3R_W(25) C3R_W(26) C3R_W(27) C3R_W(28) C3R_W(29) C3R_W(30) C3R_W(31) C3R_W(32) C3R_W(33) C3R_W(34) C3R_W(35) C3R_W(36)
    C3R_W(37) C3R_W(38) C3R_W(39) C3R_W(40) C3R_W(41) C3R_W(42) C3R_W(43) C3R_W(44) C3R_W(45) C3R_W(46) C3R_W(47) C3R_W(48)
#undef C3R_W
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;    // (more than 48 younger operations never happens at D <= 4)
  }
}
// one 1-KiB LDS-DMA piece: lane l's 16 bytes at gsrc land at LDS byte address lds_dst + 16 l (lds_dst wave-uniform).
// M0 is compiler-reserved and not preserved around an asm statement: saved and restored inside the same statement.
__device__ __forceinline__ void c3r_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NRES, int D, int DR, bool PIPE, bool ST>   // residual tensors (0: data gradient, 1: forward, 2: last block); DMA depth of the input
                                                 // rows / residual rows, in steps; PIPE: epilogue of step s - 1 under the products of step s;
                                                 // ST: InstanceNorm statistics of the output (the next block's input) as per-segment partials
__global__ void __launch_bounds__(256, 2) conv3x3_c64_rows_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                                  const float* __restrict__ bias, const bf16_t* __restrict__ res1,
                                                                  const bf16_t* __restrict__ res2, bf16_t* __restrict__ y,
                                                                  const bf16_t* __restrict__ zero_page, int B, int H, int W, int RS,
                                                                  float* __restrict__ stat_part) {
  using T = bf16_t;
  static_assert(DR >= 1 && DR <= D, "the residual rows are issued no earlier than the input rows they go with");
  constexpr int NR = 2 * D + 4;                       // ring rows: a pair in flight never overwrites a row a slower wave still reads
  constexpr int NS = DR + 1;                          // residual slots
  constexpr int NM = D + 2;                           // issue marks kept
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* ring = smem;
  unsigned char* rring = smem + NR * C3R_ROWB;
  float* bias_s = reinterpret_cast<float*>(smem + NR * C3R_ROWB + ((NRES > 0) ? NRES : 0) * NS * C3R_RESB);
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
  const unsigned rring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)rring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int nseg = H / RS, nsx = W / C3R_SW;
  const int L = xcd_block_index();                    // vertically adjacent segments (shared halo rows) behind one L2
  const int seg = L % nseg, sx = (L / nseg) % nsx, b = L / (nseg * nsx);
  const int x0 = sx * C3R_SW, y0 = seg * RS;
  const long long npix = (long long)B * H * W;
  const long long pb = (long long)b * H * W;
  const int NQ = RS / 2 + 1;                          // row pairs of the segment: pair k = input rows y0 + 2k - 1, y0 + 2k
  const int nsteps = RS / 2;

  if (tid < 64) bias_s[tid] = bias ? bias[tid] : 0.f;
  int issued = 0;                                     // vector-memory operations this wave has issued (restarts after the prologue)
  C3R_STAMP(0);

  // ---- DMA descriptors.  Input rows: instruction i (0..8) of a pair is issued by wave i % 4; chunk c = 64 i + lane ----
  const int ndma = (wv == 0) ? 3 : 2;
  long long src_off[3];                               // element offset of this lane's chunk at image row 0 (valid columns only)
  int src_rr[3];                                      // row of the pair (0 / 1); -1: a padding / out-of-image column (always the zero page)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = 64 * (wv + 4 * j) + lane;           // (j == 2 is used by wave 0 only)
    const int rr = c / (4 * 2 * C3R_PXP), rem = c % (4 * 2 * C3R_PXP);
    const int pl = rem / (2 * C3R_PXP), rem2 = rem % (2 * C3R_PXP);
    const int px = rem2 >> 1, hf = rem2 & 1;
    const int col = x0 - 1 + px;
    const bool ok = px < C3R_SW + 2 && col >= 0 && col < W;
    src_rr[j] = ok ? rr : -1;
    src_off[j] = ((long long)pl * npix + pb + (ok ? col : 0)) * 16 + hf * 8;
  }
  // residual rows: instruction i (0..7) of a step is issued by wave i % 4; chunk c = 64 i + lane = (row, plane, pixel, half)
  long long rsrc_off[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = 64 * (wv + 4 * j) + lane;
    const int rr = c >> 8, pl = (c >> 6) & 3, px = (c & 63) >> 1, hf = c & 1;
    rsrc_off[j] = ((long long)pl * npix + pb + (long long)(y0 + rr) * W + x0 + px) * 16 + hf * 8;
  }
  auto issue_pair = [&](int k) {                      // k >= NQ: a dummy pair (zero page -> a ring slot nobody reads): keeps the cadence
    const int row0 = y0 + 2 * k - 1;
    const unsigned dst = ring_lds + ((2 * k) % NR) * C3R_ROWB;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < ndma) {
        const int row = row0 + (src_rr[j] > 0 ? 1 : 0);
        const bool ok = src_rr[j] >= 0 && k < NQ && row >= 0 && row < H;
        const T* src = ok ? (x + src_off[j] + (long long)row * W * 16) : zero_page;
        c3r_dma16(src, dst + 1024 * (wv + 4 * j));
      }
    }
    issued += ndma;
  };
  auto issue_res = [&](int t) {                       // residual rows of step t (t >= nsteps: the rows of the last step again, never read)
    if constexpr (NRES > 0) {
      const long long roff = (long long)(2 * min(t, nsteps - 1)) * W * 16;
      const unsigned dst = rring_lds + (t % NS) * (NRES * C3R_RESB);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        c3r_dma16(res1 + rsrc_off[j] + roff, dst + 1024 * (wv + 4 * j));
        if constexpr (NRES > 1) c3r_dma16(res2 + rsrc_off[j] + roff, dst + C3R_RESB + 1024 * (wv + 4 * j));
      }
      issued += 2 * NRES;
    }
  };

  // ---- prologue: residual rows of steps 0 .. DR - 1 and row pairs 0 .. D go out first, the weights behind them ----
#pragma unroll
  for (int t = 0; t < DR; ++t) issue_res(t);
#pragma unroll
  for (int k = 0; k <= D; ++k) issue_pair(k);
  // ---- the wave's weights: A-fragments of output channels 32 (wv & 1) + 8 (lr >> 2) + 4 nt + (lr & 3), all 18 k-steps.
  // wp is in M2T_PACK_CONV3_ROWS order: every wave load is one contiguous 1 KB (16 rows x 64 B per load -- the shape of the
  // [tap][oc][ic] layout -- kept the address unit busy for ~20 k cycles per workgroup: 10 us of a 30 us kernel) ----
  Frag8<T> wreg[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        wreg[tap][kc][nt] = load8(wp + ((((tap * 2 + (wv & 1)) * 2 + kc) * 2 + nt) * 64 + lane) * 8);

  // every weight fragment is in its registers before the loop: the compiler places its waits for these plain loads HERE
  // (vmcnt retires in order, so the DMAs issued before them have landed as well) and has no plain load left to wait for
  // inside the loop, where such a wait would drain the ring
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) asm volatile("" :: "v"(wreg[tap][kc][nt].v));
  issued = 0;                                         // nothing is outstanding any more: the count restarts
  C3R_STAMP(1);
  int mark[NM];                                       // mark[t % NM] = `issued` once everything step t reads from LDS has been issued
#pragma unroll
  for (int t = 0; t < NM; ++t) mark[t] = 0;           // steps 0 .. D - 1 read what the prologue fetched

  const int rrw = wv >> 1;                            // this wave's output row within the step
  const int laneA = (g >> 1) * (C3R_PXP * 32) + lr * 32 + (g & 1) * 16;       // B-fragment byte offset inside a ring row
  const int plane_o = 2 * (wv & 1) + (g >> 1);                                  // output plane of this lane's 8 channels
  const long long obase = ((long long)plane_o * npix + pb) * 16 + (g & 1) * 8;
  const int laneR = rrw * (4 * C3R_SW * 32) + plane_o * (C3R_SW * 32) + lr * 32 + (g & 1) * 16;   // residual bytes of (row, plane, pixel, half)

  auto top = [&](int s) {
    // the row pair s + 1 and the residual rows of step s have landed: everything this wave issued up to mark[s] is done ...
    c3r_wait_vm(issued - mark[s % NM]);
    lds_barrier();                                    // ... and so has every other wave's share; all waves are done with step s - 1
    C3R_STAMP(2 + 3 * s);
    // (marks only grow, so the later of a step's two fetches -- its residual rows when DR < D -- is the one that stays)
    issue_res(s + DR);
    if constexpr (NRES > 0) mark[(s + DR) % NM] = issued;        // step s + DR: its residual rows are out (its row pairs went earlier)
    issue_pair(s + 1 + D);
    if constexpr (NRES == 0 || DR == D) mark[(s + D) % NM] = issued;   // step s + D reads pairs s + D, s + D + 1
  };
  auto tap_products = [&](int tap, f32x4 (&acc)[2][2], const int (&rowoff)[3]) {
    const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      Frag8<T> xf[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        xf[mt] = load8(reinterpret_cast<const T*>(ring + rowoff[ky] + laneA + kc * (2 * C3R_PXP * 32) + (16 * mt + kx) * 32));
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mma16(acc[mt][nt], wreg[tap][kc][nt], xf[mt]);
    }
  };
  constexpr int NRR = (NRES > 0) ? NRES : 1;
  auto read_res = [&](int s, Frag8<T> (&rv)[NRR][2]) {           // this lane's residual rows of step s: LDS -> registers
    if constexpr (NRES > 0) {
      const unsigned char* rslot = rring + (s % NS) * (NRES * C3R_RESB) + laneR;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        rv[0][mt] = load8(reinterpret_cast<const T*>(rslot + mt * (16 * 32)));
        if constexpr (NRES > 1) rv[1][mt] = load8(reinterpret_cast<const T*>(rslot + C3R_RESB + mt * (16 * 32)));
      }
    }
  };
  // ST: this lane's 8 channels over its 2 x nsteps pixels, as sums of (v - K) and (v - K)^2 of the STORED (rounded) values around the
  // lane's first value K -- what instnorm_stats1_kernel would read back from HBM, without the cancellation of plain sums of squares
  // Every 16 output rows (8 steps) a wave leaves one partial (n, mean, M2) per channel for its 8 rows x 32 pixels: the partials and
  // their order do not depend on the segment length, hence not on the batch size (bitwise batch invariance of the step).
  float stk[8], st1[8], st2[8];
  bool st_first = true;
#pragma unroll
  for (int e = 0; e < 8; ++e) { stk[e] = 0.f; st1[e] = 0.f; st2[e] = 0.f; }
  auto stat_flush = [&](int sub) {                    // sub: index of the 16-row group inside the segment
    if constexpr (ST) {
      // lane -> (n, mean, M2) of its 16 values; the 16 pixel lanes of a channel group (same g) merge by xor butterflies: equal
      // counts, so mean = the average of the means and M2 = sum (M2_i + n_i (mean_i - mean)^2), every sum in a fixed order
      const float nl = 16.0f, inl = 1.0f / 16.0f;
      float mean_w[8], m2_w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dm = st1[e] * inl;
        const float mean_l = stk[e] + dm;
        const float m2_l = fmaxf(st2[e] - st1[e] * dm, 0.f);
        float sm = mean_l;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) sm += __shfl_xor(sm, o);
        mean_w[e] = sm * (1.0f / 16.0f);
        const float dd = mean_l - mean_w[e];
        float q = fmaf(nl * dd, dd, m2_l);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
        m2_w[e] = q;
      }
      if (lr == 0) {
        // part [image][strip][16-row group][row parity][64][3]: this lane's 8 channels = 24 consecutive floats, six 16-byte stores
        const int idx = ((sx * (H / 16) + y0 / 16 + sub) * 2 + rrw);
        float* o = stat_part + (((long long)b * (nsx * (H / 16) * 2) + idx) * 64 + 32 * (wv & 1) + 8 * g) * 3;
        float t[24];
#pragma unroll
        for (int e = 0; e < 8; ++e) { t[3 * e] = 256.0f; t[3 * e + 1] = mean_w[e]; t[3 * e + 2] = m2_w[e]; }
#pragma unroll
        for (int v4 = 0; v4 < 6; ++v4) {
          const float q4[4] = {t[4 * v4], t[4 * v4 + 1], t[4 * v4 + 2], t[4 * v4 + 3]};
          store4(o + 4 * v4, q4);
        }
      }
      issued += 6;                                    // (exec-masked stores still count)
      st_first = true;
#pragma unroll
      for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }
    }
  };
  // epilogue of one 16-pixel tile: + bias + res1 + res2 (this order), one rounding, one 16-byte store
  auto epi_mt = [&](int mt, const f32x4 (&acc)[2][2], const Frag8<T> (&rv)[NRR][2], int orow) {
    float v[8];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
    if (bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(&bias_s[32 * (wv & 1) + 8 * g]);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(&bias_s[32 * (wv & 1) + 8 * g + 4]);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if constexpr (NRES > 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[0][mt].get(e);
    }
    if constexpr (NRES > 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[1][mt].get(e);
    }
    if constexpr (ST) {
      Frag8<T> f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f.set(e, v[e]);
      store8(y + obase + ((long long)orow * W + x0 + 16 * mt + lr) * 16, f);
      if (st_first) {
#pragma unroll
        for (int e = 0; e < 8; ++e) stk[e] = f.get(e);
        st_first = false;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dv = f.get(e) - stk[e]; st1[e] += dv; st2[e] = fmaf(dv, dv, st2[e]); }
    } else {
      store8f(y + obase + ((long long)orow * W + x0 + 16 * mt + lr) * 16, v);
    }
  };

  if constexpr (!PIPE) {
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      top(s);
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int rowoff[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) rowoff[ky] = ((2 * s + rrw + ky) % NR) * C3R_ROWB;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) tap_products(tap, acc, rowoff);
      C3R_STAMP(3 + 3 * s);
      Frag8<T> rv[NRR][2];
      read_res(s, rv);
      epi_mt(0, acc, rv, y0 + 2 * s + rrw);
      epi_mt(1, acc, rv, y0 + 2 * s + rrw);
      issued += 2;
      if ((s & 7) == 7) stat_flush(s >> 3);
      C3R_STAMP(4 + 3 * s);
    }
  } else {
    // software-pipelined: the epilogue of step s - 1 (16 conversions, 2 stores per lane, ~900 cycles on its own) is issued BETWEEN
    // the MFMAs of step s, whose matrix pipe leaves half of the wave's issue slots free; its accumulators and residual rows
    // wait in registers across the barrier
    f32x4 pacc[2][2];
    Frag8<T> prv[NRR][2];
#pragma unroll 1
    for (int s = 0; s < nsteps; ++s) {
      top(s);
      f32x4 acc[2][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      int rowoff[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) rowoff[ky] = ((2 * s + rrw + ky) % NR) * C3R_ROWB;
      const int porow = y0 + 2 * (s - 1) + rrw;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        tap_products(tap, acc, rowoff);
        if (s > 0 && tap == 1) epi_mt(0, pacc, prv, porow);
        if (s > 0 && tap == 4) epi_mt(1, pacc, prv, porow);
      }
      if (s > 0) issued += 2;
      if (s > 0 && ((s - 1) & 7) == 7) stat_flush((s - 1) >> 3);
      C3R_STAMP(3 + 3 * s);
      read_res(s, prv);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) pacc[mt][nt] = acc[mt][nt];
      C3R_STAMP(4 + 3 * s);
    }
    epi_mt(0, pacc, prv, y0 + 2 * (nsteps - 1) + rrw);
    epi_mt(1, pacc, prv, y0 + 2 * (nsteps - 1) + rrw);
    stat_flush((nsteps - 1) >> 3);
  }
  // (the dummy pairs / residual rows still in flight target LDS only; the wave ends when its counter drains)
}

// ---------------------------------------------------------------------------------------
// Fused BACKWARD of the 64 -> 64 3x3 conv (bf16, round 3): data gradient AND weight / bias gradient in ONE pass over the
// output gradient (the autograd of models/M2Trans_network.py:124-126,164).  Separately the two kernels read gy twice,
// run on two streams and fight for registers and bandwidth (stand-alone 24.7 + 31.3 us, ~80 us of kernel time where they
// meet inside the step); together they are 38.7 GFLOP on ~140 MB: on the matrix-core side of the ridge.
//   * same row streaming as conv3x3_c64_rows_kernel: a workgroup owns a 32-pixel-wide strip segment and walks down it FOUR
//     rows per step; gy and x rows (1-pixel halo, zero page outside the image) arrive by LDS-DMA in two rings, quads of
//     rows D steps ahead;
//   * eight waves, two roles, one of each per SIMD (waves w and w + 4 share a SIMD):
//       waves 0-3  data gradient: wave (half, row pair) = conv3x3_c64_rows_kernel's wave with the flipped / transposed
//                  weights in registers (M2T_PACK_CONV3_ROWS_T) -- the same products in the same order, identical bits;
//       waves 4-7  weight gradient: wave v owns input-channel tile v for all nine taps and all four output-channel tiles
//                  (36 accumulator tiles); per row of the step: four gy^T fragments and nine shifted x fragments, both by
//                  transposing LDS reads (ds_read_b64_tr_b16) of the linear P64 row image (a pixel's 16 channels of a plane
//                  are 32 contiguous bytes: exactly the 4 x 16 block the instruction transposes), k-slot (g, j) <-> pixel
//                  4 g + (j & 3) + 16 (j >> 2) so that the two lane halves of a read hit disjoint banks; the bias gradient
//                  is one more MFMA per row against a ones operand;
//   * the weight-gradient accumulators stay in registers over ALL segments a workgroup processes and leave as one fp32
//     slab [9][64][64] (+ [64]) per workgroup, reduced afterwards in a fixed order (no atomics).
// ---------------------------------------------------------------------------------------
#define C3B_ROWS 4                       // output rows per step
#ifndef C3B_STAMP
#define C3B_STAMP(i) do { } while (0)   // scratch/bench_conv_bwd.hip -DSTAMPS: s_memtime per phase (waves 0 and 4, lane 0)
#define C3B_STEP_WAIT(n) c3r_wait_vm(n)
#endif
__device__ __forceinline__ Frag8<bf16_t> c3b_tr(const unsigned char* p) {
  // p = this lane's address (see c3b_tr_lane): elements 0..3 = pixels +0..3, 4..7 = pixels +16..19 of the fragment's first pixel,
  // channel (lane & 15) of the plane
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 16 * 32));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
__device__ __forceinline__ int c3b_tr_lane(int lane) {        // lane's byte offset inside the 4-pixel x 16-channel block it helps transpose
  const int i = lane & 15;
  return (i >> 2) * 32 + (i & 3) * 8;
}

template <int D>
__global__ void __launch_bounds__(512) conv3x3_c64_bwd_rows_kernel(const bf16_t* __restrict__ gy, const bf16_t* __restrict__ x,
                                                                   const bf16_t* __restrict__ wp, bf16_t* __restrict__ gx,
                                                                   float* __restrict__ slabs, float* __restrict__ bias_slabs,
                                                                   const bf16_t* __restrict__ zero_page, int B, int H, int W, int RS) {
  using T = bf16_t;
  constexpr int NR = C3B_ROWS * (D + 2);              // ring rows per tensor: quads s, s + 1 in use, D more in flight
  static_assert((NR & (NR - 1)) == 0, "ring row indices wrap with a mask");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* gring = smem;
  unsigned char* xring = smem + NR * C3R_ROWB;
  const unsigned gring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)gring;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const int nseg = H / RS, nsx = W / C3R_SW;
  const int nsegs = B * nsx * nseg;
  const long long npix = (long long)B * H * W;
  const int nsteps = RS / C3B_ROWS;
  const int nquads = nsteps + 1;                      // quad k = input rows y0 + 4 k - 1 .. y0 + 4 k + 2

  // DMA: the four data-gradient waves issue all 36 pieces of a quad pair (piece i = tensor i / 18, 1-KiB piece i % 18 of its 4-row
  // image [row][plane][36 pixels][32 B]; wave w issues i = w + 4 j, j = 0 .. 8): an LDS-DMA instruction holds its wave's issue for
  // 100+ cycles, which the weight-gradient wave of the same SIMD fills with products.  A piece's 64 chunks of 16 bytes cross at most
  // one plane-row boundary (72 chunks per plane-row), so a chunk's (row, plane) is one of two wave-uniform pairs, chosen per lane by
  // one bit: everything but two bits per piece (that choice, and "my pixel is inside the image row") stays in scalar registers.
  constexpr int NP = 9;                               // pieces per data-gradient wave and quad pair
  unsigned lanebits = 0;                              // bit 2 j: second (row, plane) pair; bit 2 j + 1: column valid (per segment)
  const int lane16 = lane * 16;
  int x0 = 0, y0 = 0;
  long long sbase = 0;                                // element offset of (image, row y0 - 1, column x0 - 1)
  C3B_STAMP(0);
  auto issue_piece = [&](int k, int j) {              // k >= nquads: zero-page pieces into ring slots nobody reads (keeps the cadence)
    const int slot = ((C3B_ROWS * k) & (NR - 1)) * C3R_ROWB;
    const int i = wv + 4 * j;
    const bool second = i >= 18;                      // which tensor
    const int ii = second ? i - 18 : i;
    const int c0 = 64 * ii, q0 = c0 / 72, rem0 = c0 - 72 * q0;
    // wave-uniform: byte offset from the tensor base and row validity of the two candidate (row, plane) pairs
    unsigned boff[2]; bool rok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = q0 + h, rr = q >> 2, pl = q & 3;
      const int row = y0 + C3B_ROWS * k - 1 + rr;
      rok[h] = k < nquads && (unsigned)row < (unsigned)H && row <= y0 + RS;      // rows beyond the bottom halo row are never read
      boff[h] = (unsigned)((sbase + (long long)pl * npix * 16 + ((long long)C3B_ROWS * k + rr) * W * 16 + 8 * (rem0 - 72 * h)) * 2);
    }
    const bool hi = (lanebits >> (2 * j)) & 1, cok = (lanebits >> (2 * j + 1)) & 1;
    const bool ok = cok && (hi ? rok[1] : rok[0]);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(second ? x : gy) + ((hi ? boff[1] : boff[0]) + (unsigned)lane16);
    // (one zero page for the whole chip is not a hot spot: a private 256-byte page per workgroup changed nothing, 46.4 vs 47.3 us)
    c3r_dma16(ok ? reinterpret_cast<const T*>(src) : zero_page, gring_lds + (second ? NR * C3R_ROWB : 0) + 1024 * ii + slot);
  };
  auto segment_columns = [&]() {                      // per segment: which lanes' pixels lie inside the image row
    lanebits = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int c0 = 64 * ((wv + 4 * j) % 18);
      const int rem0 = c0 - 72 * (c0 / 72);
      const int hi = lane >= 72 - rem0 ? 1 : 0;       // lanes from here on belong to the next plane-row
      const int px = (rem0 + lane - 72 * hi) >> 1;
      lanebits |= (unsigned)(hi | ((px < C3R_SW + 2 && (unsigned)(x0 - 1 + px) < (unsigned)W) ? 2 : 0)) << (2 * j);
    }
  };
  auto segment_begin = [&](int sg) {
    const int seg = sg % nseg, sx = (sg / nseg) % nsx, b = sg / (nseg * nsx);
    x0 = sx * C3R_SW; y0 = seg * RS;
    sbase = (((long long)b * H + (y0 - 1)) * W + (x0 - 1)) * 16;
    if (wv < 4) {
      segment_columns();
#pragma unroll
      for (int k = 0; k <= D; ++k)
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_piece(k, j);
    }
  };
  // step s reads quads s and s + 1.  Everything a data-gradient wave issued after its last piece of quad s + 1 may still be in
  // flight: the D - 1 younger quads and the four stores of each of the min(s, D) steps since -- the segment's first D + 1 quads go
  // out together, so step 0 starts when two of them are in.  The pieces of quad s + 1 + D (whose ring slot, quad s - 1's, the
  // barrier has just freed) are issued INSIDE the step's MFMA stream, one per fragment group.
  auto step_begin = [&](int s, bool dma_wave) {
    if (dma_wave) C3B_STEP_WAIT((D - 1) * NP + min(s, D) * 4);
    // (no LDS-counter wait: the only LDS reads in flight here are the next step's prefetches, which read quad s + 1)
    __builtin_amdgcn_s_barrier();                     // every wave's pieces of quads s, s + 1 are in; all waves are done with quad s - 1
    asm volatile("" ::: "memory");
    C3B_STAMP(2 + 3 * s);
  };

  // The two roles run separate copies of the loop (same barrier count) so that neither carries the other's registers.
  if (wv < 4) {
    // ---- data gradient: wave (half, rp) -- rows 2 rp, 2 rp + 1 of the step, 32 output channels (= conv input channels) ----
    // The four input rows i of the row pair are walked once: group (i, kx) = four B fragments (kc, mt), used by output row r with
    // ky = i - r: per accumulator the products still arrive in (tap, kc) order, as in conv3x3_c64_rows_kernel.
    const int half = wv & 1, rp = (wv >> 1) & 1;
    const int laneA = (g >> 1) * (C3R_PXP * 32) + lr * 32 + (g & 1) * 16;        // B-fragment byte offset in a ring row
    const int plane_o = 2 * half + (g >> 1);
    Frag8<T> wreg[9][2][2];                           // (in flight under the first segment's quads; first use below)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          wreg[tap][kc][nt] = load8(wp + ((((tap * 2 + half) * 2 + kc) * 2 + nt) * 64 + lane) * 8);
    Frag8<T> xf[2][2][2];                             // [buffer][kc][mt]
    auto load_group = [&](int buf, int ring_row, int kx) {
      const unsigned char* p = gring + (ring_row & (NR - 1)) * C3R_ROWB + laneA + kx * 32;
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xf[buf][kc][mt] = load8(reinterpret_cast<const T*>(p + kc * (2 * C3R_PXP * 32) + mt * (16 * 32)));
    };
    for (int sg = blockIdx.x; sg < nsegs; sg += gridDim.x) {
      segment_begin(sg);
      C3B_STAMP(1);
      const long long obase = sbase + (long long)plane_o * npix * 16 + ((long long)W + 1) * 16 + (g & 1) * 8;   // (row y0, column x0)
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        step_begin(s, true);
        if (s == 0) {                                 // (the weights went out before the first quads: in vmcnt order they are in too)
#pragma unroll
          for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt) asm volatile("" :: "v"(wreg[tap][kc][nt].v));
        }
        const int r0 = C3B_ROWS * s + 2 * rp;         // ring row of input row i = 0 of the pair
        if (s == 0) load_group(0, r0, 0);             // (later steps: prefetched at the end of the step before)
        f32x4 acc[2][2][2];                           // [output row of the pair][mt][nt]
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[r][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto store_row = [&](int r) {
          const long long o = obase + ((long long)(C3B_ROWS * s + 2 * rp + r) * W + lr) * 16;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            float v[8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int q = 0; q < 4; ++q) v[4 * nt + q] = acc[r][mt][nt][q];
            store8f(gx + o + mt * (16 * 16), v);
          }
        };
#pragma unroll
        for (int gi = 0; gi < 12; ++gi) {
          const int i = gi / 3, kx = gi - 3 * i;
          if (gi < 11) load_group((gi + 1) & 1, r0 + (gi + 1) / 3, (gi + 1) % 3);
          else load_group(0, r0 + C3B_ROWS, 0);       // the next step's first group (quad s + 1: landed)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int ky = i - r;
            if (ky < 0 || ky > 2) continue;
            const int tap = 3 * ky + kx;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
              for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) mma16(acc[r][mt][nt], wreg[tap][kc][nt], xf[gi & 1][kc][mt]);
          }
          if (gi < NP) issue_piece(s + 1 + D, gi);
          __builtin_amdgcn_sched_barrier(0);
          if (gi == 8) store_row(0);                  // its last products were input row 2's
        }
        store_row(1);
        C3B_STAMP(4 + 3 * s);
      }
      c3r_wait_vm(0);                                 // the dummy quads still in flight target the rings
      __builtin_amdgcn_s_barrier();                   // and every wave is done reading them
    }
  } else {
    // ---- weight gradient: wave v owns input-channel tile v; a step = four rows of 32 pixels = four 32-deep contraction chunks.
    // The operand stream runs ahead: the x fragment XB taps on loads into the registers a tap's MFMAs have just released, the next
    // row's gy^T fragments after the row's last MFMA (the data-gradient wave of the SIMD fills that gap).
    const int vt = wv & 3;
    const int trl = c3b_tr_lane(lane);
    f32x4 wacc[9][4], bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < 4; ++o) wacc[tap][o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    Frag8<T> ones;
#pragma unroll
    for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);
    constexpr int XB = 6;                             // x fragments in flight (positions 9 rl + tap of a step, XB divides 36)
    static_assert((9 * C3B_ROWS) % XB == 0, "the x-fragment ring must close over a step");
    Frag8<T> ga[4], gb, xb[XB];
    // row R of the segment (R = 4 s + rl): gy lives in ring row R + 1, pixel p of the strip in ring pixel p + 1
    auto load_g = [&](int R) {
      const unsigned char* p = gring + ((R + 1) & (NR - 1)) * C3R_ROWB + (1 + 4 * g) * 32 + trl;
#pragma unroll
      for (int o = 0; o < 4; ++o) ga[o] = c3b_tr(p + o * (C3R_PXP * 32));
      gb = c3b_tr(p + vt * (C3R_PXP * 32));           // the bias product's operand = ga[vt] (a register select over ga[] would go through scratch)
    };
    auto load_x = [&](int n, int R0) {                // position n = 9 rl + tap of the step whose first row is R0 (n >= 36: the next step's)
      const int rl = n / 9, tap = n - 9 * rl;         // x at (row + ky - 1, pixel + kx - 1): ring row R + ky, ring pixel 4 g + kx
      const int ky = tap / 3, kx = tap - 3 * ky;
      xb[n % XB] = c3b_tr(xring + ((R0 + rl + ky) & (NR - 1)) * C3R_ROWB + vt * (C3R_PXP * 32) + (4 * g + kx) * 32 + trl);
    };
    for (int sg = blockIdx.x; sg < nsegs; sg += gridDim.x) {
      segment_begin(sg);
      C3B_STAMP(1);
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        step_begin(s, false);
        if (s == 0) {
          load_g(0);
#pragma unroll
          for (int n = 0; n < XB; ++n) load_x(n, 0);
        }
#pragma unroll
        for (int rl = 0; rl < C3B_ROWS; ++rl) {
          const int R = C3B_ROWS * s + rl;
          mma16(bacc, gb, ones);
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const int n = 9 * rl + tap;
#pragma unroll
            for (int o = 0; o < 4; ++o) mma16(wacc[tap][o], ga[o], xb[n % XB]);
            load_x(n + XB, C3B_ROWS * s);             // (row 4 of the step = the next step's row 0: quad s + 1, landed)
            __builtin_amdgcn_sched_barrier(0);
          }
          load_g(R + 1);
        }
        C3B_STAMP(4 + 3 * s);
      }
      __builtin_amdgcn_s_barrier();
    }
    // slab [tap][ic][oc]: a lane's four accumulator rows are four consecutive output channels
    float* out = slabs + (long long)blockIdx.x * (9 * 64 * 64);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float v[4] = {wacc[tap][o][0], wacc[tap][o][1], wacc[tap][o][2], wacc[tap][o][3]};
        store4(out + ((long long)tap * 64 + 16 * vt + lr) * 64 + 16 * o + 4 * g, v);
      }
    if (lr == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_slabs[(long long)blockIdx.x * 64 + 16 * vt + 4 * g + r] = bacc[r];
    }
  }
  C3B_STAMP(63);
}

// bf16 takes the pipelined kernel: at most this many workgroups (8 per CU: with more tiles a workgroup walks several)
constexpr int C3_PIPE_MAX_BLOCKS = 2048;

// segment length of the row-streaming kernel: >= 512 workgroups (2 per CU) when the map allows it, segments of 16..64 rows
static int c3r_force_rs = 0;       // scratch/bench_conv_rows.hip only: overrides the segment length
static int c3r_rows_per_segment(int B, int H, int W) {
  if (c3r_force_rs > 0) return (H % c3r_force_rs == 0) ? c3r_force_rs : 0;
  const long long strips = (long long)B * (W / C3R_SW);
  int rs = 64;
  while (rs > 16 && (H % rs != 0 || strips * (H / rs) < 512)) rs >>= 1;
  return (H % rs == 0) ? rs : 0;
}
template <int NRES, int D, int DR, bool PIPE = false>
static int go_c3r(const void* x, const void* wp /* M2T_PACK_CONV3_ROWS(_T) */, const float* bias, const void* res1, const void* res2, void* y, const void* zero_page,
                  int B, int H, int W, int rs, hipStream_t st, float* stat_part = nullptr) {
  constexpr int NR = 2 * D + 4;
  const size_t sh = (size_t)NR * C3R_ROWB + (size_t)NRES * (DR + 1) * C3R_RESB + 64 * sizeof(float);
  const int nblk = B * (W / C3R_SW) * (H / rs);
  if constexpr (NRES == 1) {
    if (stat_part) {
      if (int rc__ = m2t_ensure_dynamic_lds((const void*)conv3x3_c64_rows_kernel<NRES, D, DR, PIPE, true>, (int)sh)) return rc__;
      M2T_LAUNCH_TIMED((conv3x3_c64_rows_kernel<NRES, D, DR, PIPE, true>), dim3(nblk), dim3(256), sh, st, (const bf16_t*)x, (const bf16_t*)wp, bias,
                       (const bf16_t*)res1, (const bf16_t*)res2, (bf16_t*)y, (const bf16_t*)zero_page, B, H, W, rs, stat_part);
      return 0;
    }
  }
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)conv3x3_c64_rows_kernel<NRES, D, DR, PIPE, false>, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED((conv3x3_c64_rows_kernel<NRES, D, DR, PIPE, false>), dim3(nblk), dim3(256), sh, st, (const bf16_t*)x, (const bf16_t*)wp, bias,
                   (const bf16_t*)res1, (const bf16_t*)res2, (bf16_t*)y, (const bf16_t*)zero_page, B, H, W, rs, (float*)nullptr);
  return 0;
}

// the rows kernel can leave the InstanceNorm statistics of y as <= 8 M2T_NORM_SPLIT per-image partials (64 at 128 x 128, 256 at 256 x 256 -- the
// BASELINE sizes; round 3 capped this at 64 and configs[4] paid a 30 us statistics pass per block): number of partials, 0 = not available
int conv3x3_c64_stat_partials(int dt, int B, int H, int W, int variant) {
  if (dt == M2T_F32 || variant == 1 || W % C3R_SW || (long long)B * H * W * 64 >= (1LL << 31)) return 0;
  if (c3r_rows_per_segment(B, H, W) <= 0) return 0;
  const int n = (W / C3R_SW) * (H / 16) * 2;          // one per wave and 16 output rows: independent of the segment length
  return n <= 8 * M2T_NORM_SPLIT ? n : 0;
}
int launch_conv3x3_c64(int dt, const void* x, const void* wp, const float* bias, const void* res1, const void* res2,
                       void* y, int B, int H, int W, hipStream_t st, const void* wrows, const void* zero_page, int variant, float* stat_part) {
  if (H % C3_TH || W % C3_TW) return m2t_set_error(-2, "conv3x3_c64: H%8 or W%16");
  if (stat_part && !(res1 && !res2 && conv3x3_c64_stat_partials(dt, B, H, W, variant) > 0))
    return m2t_set_error(-2, "conv3x3_c64: statistics partials need the one-residual row-streaming kernel (conv3x3_c64_stat_partials)");
  const long long ntiles = (long long)B * (H / C3_TH) * (W / C3_TW);
  if (res2 && !res1) return m2t_set_error(-2, "conv3x3_c64: res2 without res1");
  if (dt != M2T_F32 && wrows && zero_page && variant != 1 && W % C3R_SW == 0 && (long long)B * H * W * 64 < (1LL << 31)) {
    const int rs = c3r_rows_per_segment(B, H, W);
    if (rs > 0) {
      int rc;
      // LDS per workgroup (two per CU): ring 46 080 B + residual slots: <= 79 104 B
      // DMA depth 2 (ring of 8 rows = 36 KB + residual slots; LDS per workgroup <= 61 KB).  (Depth 3 and the pipelined-epilogue
      // form measured 29.4 / 28.8 us against 27.8 stand-alone and tied inside the step: their dispatch was retired in round 4,
      // numbers in profiles/README.md)
      if (res2) rc = go_c3r<2, 2, 1>(x, wrows, bias, res1, res2, y, zero_page, B, H, W, rs, st);
      else if (res1) rc = go_c3r<1, 2, 2>(x, wrows, bias, res1, res2, y, zero_page, B, H, W, rs, st, stat_part);
      else rc = go_c3r<0, 2, 1>(x, wrows, bias, res1, res2, y, zero_page, B, H, W, rs, st);
      if (rc) return rc;
      M2T_LAUNCH_CHECK();
      return 0;
    }
  }
  if (dt != M2T_F32) {
    const int tpb = (int)ceil_divll(ntiles, C3_PIPE_MAX_BLOCKS);
    const int nblk = (int)ceil_divll(ntiles, tpb);
    const int xcd_order = (nblk % 8 == 0 && (long long)nblk * tpb == ntiles) ? 1 : 0;
    M2T_LAUNCH_TIMED(conv3x3_c64_pipe_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, bias,
                     (const bf16_t*)res1, (const bf16_t*)res2, (bf16_t*)y, B, H, W, tpb, xcd_order);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(W / C3_TW, H / C3_TH, B);
  M2T_LAUNCH_TIMED(conv3x3_c64_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)wp, bias, (const float*)res1, (const float*)res2, (float*)y, H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// 64 -> 64 3x3 conv weight gradient:  dW[tap][oc][ic] = sum_p gy[p][oc] * x[p + tap][ic]
// Workgroup sweeps tiles of TH x 16 pixels.  Both operands are contracted over PIXELS, so they are needed
// pixel-contiguous per channel: the tiles are staged row-major exactly as they lie in HBM (16-byte stores:
// gy as Gs[m][oc], the zero-padded x halo as Xs[(row+1)*18 + col+1][ic]) and transposed by the LDS read
// (ds_read_b64_tr_b16): 8 consecutive pixels of a tile row are 8 consecutive rows of either array, at any tap
// offset.  8 waves: wave (op, tq) owns two oc tiles for all 4 ic tiles and 2-3 taps (24 accumulators).
// Output: fp32 slabs [nblk][9][64][64] (+ bias slabs [nblk][64]: the bias gradient is one extra MFMA per step
// against a ones operand), summed (and permuted to torch layout) by the batched reduction.  x, gy are P64.
// =======================================================================================
template <typename T, int TH>
__global__ void __launch_bounds__(512) conv3x3_c64_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                float* __restrict__ slabs, float* __restrict__ bias_slabs, int B,
                                                                int H, int W, int tiles_per_block) {
  constexpr int MT = TH * 16;              // pixels per tile
  constexpr int HX = (TH + 2) * 18;        // halo pixels
  constexpr int LD = 72;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Gs)[LD] = reinterpret_cast<T(*)[LD]>(smem);                          // [MT][72]
  T(*Xs)[LD] = reinterpret_cast<T(*)[LD]>(smem + sizeof(T) * MT * LD);    // [HX][72]
  // 8 waves: wave (op, tq) owns the oc tiles 2 op, 2 op + 1 and the taps of quarter tq = {0,1,2} {3,4} {5,6} {7,8}:
  // every transposed x operand feeds two MFMAs, which halves the LDS read traffic that bounds this kernel
  const int tid = threadIdx.x, lane = tid & 63, op = (tid >> 6) & 1, tq = tid >> 7;
  const int tap0 = (tq == 0) ? 0 : 2 * tq + 1, ntap = (tq == 0) ? 3 : 2;
  const int g = lane >> 4;
  const int tw = W / 16, th = H / TH;
  const long long ntiles = (long long)B * th * tw;
  const long long npix = (long long)B * H * W;              // x, gy are P64
  f32x4 acc[3][2][4], accb[2];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][o][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag8<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);

  const long long t0 = (long long)blockIdx.x * tiles_per_block;
  const long long t1 = min(ntiles, t0 + tiles_per_block);
  constexpr int NG = MT * 8 / 512;
  constexpr int TOTX = HX * 8, NX = (TOTX + 511) / 512;
  Frag8<T> fg[NG], fx[NX];
  auto fetch = [&](long long t) {          // all global loads of a tile, into registers
    const int tx = (int)(t % tw);
    const long long q = t / tw;
    const int ty = (int)(q % th);
    const int b = (int)(q / th);
    const int x0 = tx * 16, y0 = ty * TH;
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int idx = tid + it * 512;
      const int m = idx >> 3, cv = idx & 7;
      fg[it] = load8(gy + p64(npix, ((long long)b * H + y0 + (m >> 4)) * W + x0 + (m & 15), cv * 8));
    }
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = tid + it * 512;
      const int p = idx >> 3, cv = idx & 7;
      const int row = p / 18, xs = p - row * 18 - 1;        // xs in [-1, 16]
      const int gyy = y0 + row - 1, gxx = x0 + xs;
      fx[it] = frag_zero<T>();
      if (idx < TOTX && gyy >= 0 && gyy < H && gxx >= 0 && gxx < W)
        fx[it] = load8(x + p64(npix, ((long long)b * H + gyy) * W + gxx, cv * 8));
    }
  };
  if (t0 < t1) fetch(t0);
  for (long long t = t0; t < t1; ++t) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NG; ++it) {
      const int idx = tid + it * 512;
      store8(&Gs[idx >> 3][(idx & 7) * 8], fg[it]);
    }
#pragma unroll
    for (int it = 0; it < NX; ++it) {
      const int idx = tid + it * 512;
      if (idx < TOTX) store8(&Xs[idx >> 3][(idx & 7) * 8], fx[it]);
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);          // the next tile's loads fly under this tile's products
#pragma unroll 1
    for (int ch = 0; ch < MT / 32; ++ch) {
      // k-slot (g, j) <-> pixel m = 32 ch + 8 g + j : row = 2 ch + (g >> 1), col = 8 (g & 1) + j
      const int m0 = 32 * ch + 8 * g;
      Frag8<T> gf[2];
#pragma unroll
      for (int o = 0; o < 2; ++o) gf[o] = load8_tr(&Gs[m0][16 * (2 * op + o)], &Gs[m0 + 4][16 * (2 * op + o)], LD, lane);
      if (tq == 0) {                         // bias gradient: row sums of gy^T ride along (every column equal)
        mma16(accb[0], gf[0], ones);
        mma16(accb[1], gf[1], ones);
      }
#pragma unroll
      for (int tl = 0; tl < 3; ++tl) {
        if (tl < ntap) {
          const int tap = tap0 + tl;
          const int ky = tap / 3, kx = tap - 3 * ky;
          const int hp = (2 * ch + (g >> 1) + ky) * 18 + 8 * (g & 1) + kx;
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const Frag8<T> xf = load8_tr(&Xs[hp][16 * it], &Xs[hp + 4][16 * it], LD, lane);
            mma16(acc[tl][0][it], gf[0], xf);
            mma16(acc[tl][1][it], gf[1], xf);
          }
        }
      }
    }
  }
  const int lr = lane & 15;
  float* out = slabs + (long long)blockIdx.x * (9 * 64 * 64);
#pragma unroll
  for (int tl = 0; tl < 3; ++tl) {
    if (tl < ntap) {
      const int tap = tap0 + tl;
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            out[((long long)tap * 64 + 16 * (2 * op + o) + 4 * g + r) * 64 + 16 * it + lr] = acc[tl][o][it][r];
    }
  }
  if (lr == 0 && tq == 0) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) bias_slabs[(long long)blockIdx.x * 64 + 16 * (2 * op + o) + 4 * g + r] = accb[o][r];
  }
}
int launch_conv3x3_c64_wgrad(int dt, const void* x, const void* gy, float* slabs, float* bias_slabs, int* nslab, int B, int H, int W,
                             hipStream_t st) {
  if (W % 16 || H % 8) return m2t_set_error(-2, "conv3x3_c64_wgrad: H%8 or W%16");
  const int TH = (dt == M2T_F32) ? 4 : 8;
  const long long ntiles = (long long)B * (H / TH) * (W / 16);
  int nblk = (int)std::min<long long>(256, ntiles);
  const int tpb = (int)ceil_divll(ntiles, nblk);
  nblk = (int)ceil_divll(ntiles, tpb);
  if (dt == M2T_F32) {
    const size_t sh = sizeof(float) * 72 * (4 * 16 + 6 * 18);
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_wgrad_kernel<float, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    M2T_LAUNCH_TIMED((conv3x3_c64_wgrad_kernel<float, 4>), dim3(nblk), dim3(512), sh, st, (const float*)x, (const float*)gy, slabs, bias_slabs, B, H, W, tpb);
  } else {
    const size_t sh = sizeof(bf16_t) * 72 * (8 * 16 + 10 * 18);
    (void)hipFuncSetAttribute((const void*)conv3x3_c64_wgrad_kernel<bf16_t, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    M2T_LAUNCH_TIMED((conv3x3_c64_wgrad_kernel<bf16_t, 8>), dim3(nblk), dim3(512), sh, st, (const bf16_t*)x, (const bf16_t*)gy, slabs, bias_slabs, B, H, W, tpb);
  }
  M2T_LAUNCH_CHECK();
  *nslab = nblk;
  return 0;
}

// fused backward (bf16): gx = data gradient, slabs [n][9][64][64] / bias_slabs [n][64] = weight / bias gradient partials,
// *nslab = n <= 256 workgroups.  wrows_t: the weight in M2T_PACK_CONV3_ROWS_T order.  Returns M2T_UNSUPPORTED when the shape
// has no strip decomposition (the caller then uses the two separate kernels).
bool conv3x3_c64_bwd_fusable(int B, int H, int W) { return W % C3R_SW == 0 && H % 16 == 0 && (long long)B * H * W * 64 < (1LL << 31); }
int launch_conv3x3_c64_bwd_fused(const void* gy, const void* x, const void* wrows_t, void* gx, float* slabs, float* bias_slabs, int* nslab,
                                 const void* zero_page, int B, int H, int W, hipStream_t st) {
  if (!conv3x3_c64_bwd_fusable(B, H, W)) return M2T_UNSUPPORTED;
  // segments of 16 .. 64 rows, at least 256 of them when the map allows; a workgroup walks every 256th segment
  const long long strips = (long long)B * (W / C3R_SW);
  int rs = 64;
  while (rs > 16 && (H % rs != 0 || strips * (H / rs) < 256)) rs >>= 1;
  if (H % rs) return M2T_UNSUPPORTED;
  const long long nsegs = strips * (H / rs);
  const int nblk = (int)std::min<long long>(256, nsegs);
  constexpr int D = 2;
  const size_t sh = (size_t)2 * C3B_ROWS * (D + 2) * C3R_ROWB;
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)conv3x3_c64_bwd_rows_kernel<D>, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED((conv3x3_c64_bwd_rows_kernel<D>), dim3(nblk), dim3(512), sh, st, (const bf16_t*)gy, (const bf16_t*)x, (const bf16_t*)wrows_t,
                   (bf16_t*)gx, slabs, bias_slabs, (const bf16_t*)zero_page, B, H, W, rs);
  M2T_LAUNCH_CHECK();
  *nslab = nblk;
  return 0;
}

// =======================================================================================
// tail conv 64 -> 3, reflect padding, no bias; input = the activation gelu(t) stored by tail_expand (`tpre`
// below names that tensor; `tder` the stored derivative gelu'(t)).
// N = 3 is too thin for a direct MFMA mapping, so the 9 taps are folded into the N axis:
//     Y[p][(tap,oc)] = sum_ic act[p][ic] * w[oc][ic][tap]          (K = 64, N = 27 -> 32)
//     out[o][oc]     = sum_tap Y[o + off(tap)][(tap,oc)]            (9 shifted adds from LDS)
// and symmetrically for the data gradient (K = 27 -> 32 gathered gradient taps, N = 64) and the
// weight gradient (contraction over the tile's halo pixels, transposing LDS reads).
// Workgroup = 16 x 16 output pixels (+1 halo).  Output / gradient tensors are NCHW fp32.
// =======================================================================================
#define FC_T 16
#define FC_HP ((FC_T + 2) * (FC_T + 2))   // 324 halo pixels
#define FC_HPP 352                        // padded to 11 contraction chunks of 32
#define FC_LD 72

// halo tile of the stored activation gelu(t) (reflect addressing), rows >= 324 zero
template <typename T, int ROWS = FC_HPP>
__device__ __forceinline__ void final_stage_act(T (*As)[FC_LD], const T* __restrict__ tb, int y0, int x0, int H, int W, int tid) {
  constexpr int ITEMS = (ROWS * 8 + 255) / 256;   // 11
  Frag8<T> f[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx & 7, p = idx >> 3;
    f[it] = frag_zero<T>();
    if (p < FC_HP) {
      const int py = p / (FC_T + 2), px = p - py * (FC_T + 2);
      const int gy = reflect_idx(y0 + py - 1, H), gx = reflect_idx(x0 + px - 1, W);
      f[it] = load8(tb + ((long long)gy * W + gx) * 64 + cv * 8);
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx & 7, p = idx >> 3;
    if (p < ROWS) store8(&As[p][cv * 8], f[it]);
  }
}

// Persistent: a workgroup keeps the 27 x 64 weight tile in LDS and sweeps tiles dealt round-robin; the next tile's
// halo (11 vectors per thread) is in flight while the current one is multiplied and summed.
template <typename T>
__global__ void __launch_bounds__(256) final_conv_fwd_kernel(const T* __restrict__ tpre, const float* __restrict__ w,
                                                             float* __restrict__ out, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*As)[FC_LD] = reinterpret_cast<T(*)[FC_LD]>(smem);                                   // [336][72] (21 pixel tiles)
  constexpr size_t szAY = (sizeof(T) * 336 * FC_LD > sizeof(float) * 336 * 33) ? sizeof(T) * 336 * FC_LD : sizeof(float) * 336 * 33;
  T(*Ws)[FC_LD] = reinterpret_cast<T(*)[FC_LD]>(smem + szAY);      // [32][72]
  float(*Ys)[33] = reinterpret_cast<float(*)[33]>(smem);   // [336][33] fp32, ALIASES As once the products are done
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int tw = W / FC_T, th = H / FC_T;
  const long long ntiles = (long long)B * th * tw;
  const long long hw = (long long)H * W;
  for (int i = tid; i < 32 * 64; i += 256) {
    const int n = i >> 6, ic = i & 63;              // n = tap*3 + oc
    float v = 0.f;
    if (n < 27) v = w[((n % 3) * 64 + ic) * 9 + n / 3];
    Ws[n][ic] = from_f<T>(v);
  }
  constexpr int ITEMS = (336 * 8 + 255) / 256;      // 11
  Frag8<T> f[ITEMS];
  auto fetch = [&](long long t) {
    const int x0 = (int)(t % tw) * FC_T;
    const long long q = t / tw;
    const int y0 = (int)(q % th) * FC_T;
    const T* tb = tpre + (q / th) * hw * 64;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx & 7, p = idx >> 3;
      f[it] = frag_zero<T>();
      if (p < FC_HP) {
        const int py = p / (FC_T + 2), px = p - py * (FC_T + 2);
        const int gy = reflect_idx(y0 + py - 1, H), gx = reflect_idx(x0 + px - 1, W);
        f[it] = load8(tb + ((long long)gy * W + gx) * 64 + cv * 8);
      }
    }
  };
  // tiles are dealt round-robin over LOGICAL workgroup indices (XCD-aware): in every round an XCD owns a contiguous
  // run of tiles, so horizontally adjacent tiles share their halo columns in one L2
  const int lb = xcd_block_index();
  if ((long long)lb < ntiles) fetch(lb);
  for (long long t = lb; t < ntiles; t += gridDim.x) {
    const int x0 = (int)(t % tw) * FC_T;
    const long long q = t / tw;
    const int y0 = (int)(q % th) * FC_T;
    const long long b = q / th;
    __syncthreads();      // the previous tile's Ys reads are done (first trip: Ws staged)
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx & 7, p = idx >> 3;
      if (p < 336) store8(&As[p][cv * 8], f[it]);
    }
    __syncthreads();
    if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
    // Y^T tile products: rows n (2 tiles), cols = halo pixels (21 tiles of 16: wave wv takes wv, wv+4, ...)
    f32x4 acc[6][2];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int mt = wv + 4 * j;
      if (mt < 21) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const Frag8<T> xf = load8(&As[16 * mt + lr][32 * kc + 8 * g]);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const Frag8<T> wf = load8(&Ws[16 * nt + lr][32 * kc + 8 * g]);
            mma16(acc[j][nt], wf, xf);
          }
        }
      }
    }
    __syncthreads();      // every wave is done reading As: its memory becomes the Y tile
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int mt = wv + 4 * j;
      if (mt < 21) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ys[16 * mt + lr][16 * nt + 4 * g + r] = acc[j][nt][r];
      }
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const float* yp = &Ys[(ty + ky) * (FC_T + 2) + tx + kx][tap * 3];
      a0 += yp[0]; a1 += yp[1]; a2 += yp[2];
    }
    const long long o = b * 3 * hw + (long long)(y0 + ty) * W + x0 + tx;
    out[o] = a0;
    out[o + hw] = a1;
    out[o + 2 * hw] = a2;
  }
}
template <typename T> static size_t final_fwd_smem() {
  const size_t a = sizeof(T) * 336 * FC_LD, y = sizeof(float) * 336 * 33;   // Ys aliases As; Ws sits behind the larger of the two
  // (bf16: 48 384 + 4 608 B -> three workgroups per CU)
  return std::max(a, y) + sizeof(T) * 32 * FC_LD;
}
int launch_final_conv_fwd(int dt, const void* tpre, const float* w, float* out, int B, int H, int W, hipStream_t st) {
  if (H % FC_T || W % FC_T) return m2t_set_error(-2, "final_conv: H,W must be multiples of 16");
  const long long ntiles = (long long)B * (H / FC_T) * (W / FC_T);
  if (dt == M2T_F32) {
    const size_t sh = final_fwd_smem<float>();
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)final_conv_fwd_kernel<float>, (int)sh)) return rc__;
    const int grid = (int)std::min<long long>(ntiles, 256);        // fp32: 101 KB of LDS, one workgroup per CU
    M2T_LAUNCH_TIMED(final_conv_fwd_kernel<float>, dim3(grid), dim3(256), sh, st, (const float*)tpre, w, out, B, H, W);
  } else {
    const size_t sh = final_fwd_smem<bf16_t>();
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)final_conv_fwd_kernel<bf16_t>, (int)sh)) return rc__;
    const int grid = (int)std::min<long long>(ntiles, 768);        // bf16: 53 KB of LDS, three workgroups per CU
    M2T_LAUNCH_TIMED(final_conv_fwd_kernel<bf16_t>, dim3(grid), dim3(256), sh, st, (const bf16_t*)tpre, w, out, B, H, W);
  }
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// tail conv data gradient (+ GELU backward):
//   g_act[p][ic] = sum_{(tap,oc)} Geff[p][(tap,oc)] * w[oc][ic][tap],   g_t = g_act * gelu'(t)  (tpre = stored derivative)
// Geff gathers gout at the output positions that read input pixel p through `tap`, INCLUDING the
// reads that reached p through the reflect padding (rows/cols 1 and n-2 also serve the padded
// ring positions -1 and n).  GEMM: M = 256 tile pixels, K = 27 -> 32, N = 64.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) final_conv_dgrad_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                               const T* __restrict__ tpre, T* __restrict__ gt, int H, int W) {
  __shared__ float Gs[3][FC_HP];                                     // gout halo tile (0 outside the image)
  __shared__ __attribute__((aligned(16))) T Ge[256][40];            // Geff [pixel][(tap,oc) padded to 32]
  __shared__ __attribute__((aligned(16))) T Wt[64][40];             // [ic][(tap,oc)]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * FC_T, y0 = blockIdx.y * FC_T, b = blockIdx.z;
  const long long hw = (long long)H * W;
  for (int i = tid; i < 3 * FC_HP; i += 256) {
    const int oc = i / FC_HP, p = i - oc * FC_HP;
    const int py = p / (FC_T + 2), px = p - py * (FC_T + 2);
    const int gy = y0 + py - 1, gx = x0 + px - 1;
    Gs[oc][p] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? gout[((long long)b * 3 + oc) * hw + (long long)gy * W + gx] : 0.f;
  }
  for (int i = tid; i < 64 * 32; i += 256) {
    const int ic = i >> 5, n = i & 31;
    float v = 0.f;
    if (n < 27) v = w[((n % 3) * 64 + ic) * 9 + n / 3];
    Wt[ic][n] = from_f<T>(v);
  }
  __syncthreads();
  {
    // one thread per tile pixel builds its 27 gathered taps
    const int ty = tid >> 4, tx = tid & 15;
    const int yy = y0 + ty, xx = x0 + tx;
    int pys[2], pxs[2], npy = 1, npx = 1;
    pys[0] = yy; pxs[0] = xx;
    if (yy == 1) pys[npy++] = -1;
    if (yy == H - 2 && npy < 2) pys[npy++] = H;
    if (xx == 1) pxs[npx++] = -1;
    if (xx == W - 2 && npx < 2) pxs[npx++] = W;
#pragma unroll 1
    for (int oc = 0; oc < 3; ++oc) {          // one output channel at a time keeps the live set small
      float ge[9];
#pragma unroll
      for (int i = 0; i < 9; ++i) ge[i] = 0.f;
      for (int a = 0; a < npy; ++a)
        for (int c = 0; c < npx; ++c) {
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int oy = pys[a] - ky + 1;
            if (oy < 0 || oy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int ox = pxs[c] - kx + 1;
              if (ox < 0 || ox >= W) continue;
              ge[ky * 3 + kx] += Gs[oc][(oy - y0 + 1) * (FC_T + 2) + (ox - x0 + 1)];   // always inside the +-1 halo
            }
          }
        }
#pragma unroll
      for (int i = 0; i < 9; ++i) Ge[tid][i * 3 + oc] = from_f<T>(ge[i]);
    }
#pragma unroll
    for (int i = 27; i < 32; ++i) Ge[tid][i] = from_f<T>(0.f);
  }
  __syncthreads();
  // wave wv: pixel tiles 4 wv .. 4 wv + 3 (= tile rows), all 4 channel tiles; A rows = ic (permuted so a
  // lane ends with 16 consecutive channels), B cols = pixels
#pragma unroll 1
  for (int mtl = 0; mtl < 4; ++mtl) {
    const int mt = 4 * wv + mtl;
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const Frag8<T> xf = load8(&Ge[16 * mt + lr][8 * g]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int nl = 16 * (lr >> 2) + 4 * nt + (lr & 3);
      const Frag8<T> wf = load8(&Wt[nl][8 * g]);
      mma16(acc[nt], wf, xf);
    }
    const int pix_t = 16 * mt + lr;
    const long long pix = ((long long)b * H + y0 + (pix_t >> 4)) * W + x0 + (pix_t & 15);
    float v[16], pp[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[nt][r];
    load16f(tpre + pix * 64 + 16 * g, pp);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] *= pp[e];          // stored gelu'(t)
    store16f(gt + pix * 64 + 16 * g, v);
  }
}
// ---------------------------------------------------------------------------------------
// fp32 tail conv data gradient (+ GELU backward) on the VALU (round 5): K = 27 padded to 32 on exact-fp32 MFMA plus a per-pixel
// gather loop left the kernel above at 696 us for 2.1 GB of traffic.  Here a wave owns 64 pixels of a row, LANE = PIXEL: the 27
// gathered gradient values Geff[p][(tap, oc)] are per-lane registers built from 27 coalesced, masked row loads (the reads that reached
// p through the reflect padding are the column x -+ 1 / row y -+ 1 values already in hand: rows / columns 1 and n - 2 also serve
// the padded ring), the weights of one input channel are wave-uniform (scalar loads), 27 v_fmac per (pixel, channel); the
// [pixel][channel] tile turns through LDS (33-float rows, 32 channels at a time) so that gelu'(t) is read and g(t) written as
// 128-byte row pieces.  Same sums as final_conv_dgrad_kernel in another fp32 order.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) final_conv_dgrad_f32_kernel(const float* __restrict__ gout, const float* __restrict__ w,
                                                                  const float* __restrict__ der, float* __restrict__ gt, int B, int H,
                                                                  int W, int nunits) {
  __shared__ float T[4][64][33];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u = blockIdx.x * 4 + wv;
  if (u >= nunits) return;                 // (no workgroup barrier below: the waves are independent)
  const int segs = (W + 63) >> 6;
  const int seg = u % segs, q = u / segs, y = q % H, b = q / H;
  const int x0 = seg << 6, x = x0 + lane;
  const long long hw = (long long)H * W;
  // E[row r = y - 1 + j][kx][oc]: the column-combined values of the three source rows (zero outside the image)
  float E[3][3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int yy = y - 1 + j;
    const bool vr = yy >= 0 && yy < H;
#pragma unroll
    for (int oc = 0; oc < 3; ++oc) {
      const float* gr = gout + ((long long)b * 3 + oc) * hw + (long long)(vr ? yy : 0) * W;
      const float vm = (vr && x - 1 >= 0 && x - 1 < W) ? gr[x - 1] : 0.f;
      const float v0 = (vr && x < W) ? gr[x] : 0.f;
      const float vp = (vr && x + 1 < W) ? gr[x + 1] : 0.f;
      // output column ox = x - kx + 1; x == 1 also serves the padded column -1 (kx = 0 -> ox = 0 = x - 1), x == W - 2 the column W
      E[j][0][oc] = vp + (x == 1 ? vm : 0.f);
      E[j][1][oc] = v0;
      E[j][2][oc] = vm + (x == W - 2 ? vp : 0.f);
    }
  }
  // Geff[ky][kx][oc]: output row oy = y - ky + 1 -> source row index j = 2 - ky; y == 1 also serves the padded row -1 (ky = 0 -> oy = 0),
  // y == H - 2 the row H (ky = 2 -> oy = H - 1)
  float G[27];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int oc = 0; oc < 3; ++oc) {
      G[(0 * 3 + kx) * 3 + oc] = E[2][kx][oc] + (y == 1 ? E[0][kx][oc] : 0.f);
      G[(1 * 3 + kx) * 3 + oc] = E[1][kx][oc];
      G[(2 * 3 + kx) * 3 + oc] = E[0][kx][oc] + (y == H - 2 ? E[2][kx][oc] : 0.f);
    }
  const long long prow = ((long long)b * H + y) * W + x0;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
#pragma unroll 2
    for (int icl = 0; icl < 32; ++icl) {
      const int ic = 32 * half + icl;
      float sacc = 0.f;
#pragma unroll
      for (int oc = 0; oc < 3; ++oc) {
        const float* wr = w + (oc * 64 + ic) * 9;       // w[oc][ic][tap]: nine contiguous values, wave-uniform
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) sacc = fmaf(G[tap * 3 + oc], wr[tap], sacc);
      }
      T[wv][lane][icl] = sacc;
    }
    // (one wave writes and reads its own tile: LDS operations of a wave complete in order)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pi = 8 * j + (lane >> 3), c4 = (lane & 7) * 4;
      if (x0 + pi < W) {
        const long long off = (prow + pi) * 64 + 32 * half + c4;
        const f32x4 d = *reinterpret_cast<const f32x4*>(der + off);
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = T[wv][pi][c4 + i] * d[i];
        *reinterpret_cast<f32x4*>(gt + off) = v;
      }
    }
  }
}

int launch_final_conv_dgrad(int dt, const float* gout, const float* w, const void* tpre, void* gtpre, int B, int H, int W,
                            hipStream_t st) {
  if (H % FC_T || W % FC_T) return m2t_set_error(-2, "final_conv: H,W must be multiples of 16");
  if (dt == M2T_F32 && g_m2t_f32_fast && (long long)B * H * W * 64 < (1LL << 31)) {
    const int nunits = B * H * ((W + 63) / 64);
    M2T_LAUNCH_TIMED(final_conv_dgrad_f32_kernel, dim3((nunits + 3) / 4), dim3(256), 0, st, gout, w, (const float*)tpre, (float*)gtpre, B, H, W, nunits);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  dim3 grid(W / FC_T, H / FC_T, B);
  if (dt == M2T_F32) M2T_LAUNCH_TIMED(final_conv_dgrad_kernel<float>, grid, dim3(256), 0, st, gout, w, (const float*)tpre, (float*)gtpre, H, W);
  else M2T_LAUNCH_TIMED(final_conv_dgrad_kernel<bf16_t>, grid, dim3(256), 0, st, gout, w, (const bf16_t*)tpre, (bf16_t*)gtpre, H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------
// tail conv weight gradient: dW[oc][ic][tap] = sum_{output pixels o} gout[oc][o] * act[refl(o + off(tap))][ic]
// Per 16x16 tile and tap: a [16 (3 used) x 256] x [256 x 64] product whose contraction runs over the
// tile's output pixels; the gradient operand is the row-major [oc][o] tile, the activation operand comes
// from the row-major halo tile by transposing LDS reads at the tap's offset (8 consecutive pixels of a
// tile row = 8 consecutive halo rows).  Wave w owns channel tile w for all 9 taps (9 accumulators);
// a workgroup sweeps several tiles before writing its slab.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) final_conv_wgrad_kernel(const float* __restrict__ gout, const T* __restrict__ tpre,
                                                               float* __restrict__ slabs, int B, int H, int W,
                                                               int tiles_per_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*As)[FC_LD] = reinterpret_cast<T(*)[FC_LD]>(smem);                                          // [352][72] act
  T(*Gs)[FC_T * FC_T + 8] = reinterpret_cast<T(*)[FC_T * FC_T + 8]>(smem + sizeof(T) * FC_HPP * FC_LD);   // [16][264], rows 3.. zero
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int tw = W / FC_T, th = H / FC_T;
  const long long ntiles = (long long)B * th * tw;
  const long long hw = (long long)H * W;
  f32x4 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < 13 * (FC_T * FC_T + 8); i += 256) Gs[3 + i / (FC_T * FC_T + 8)][i % (FC_T * FC_T + 8)] = from_f<T>(0.f);
  const long long t0 = (long long)blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
  for (long long t = t0; t < t1; ++t) {
    const int tx = (int)(t % tw);
    const long long q = t / tw;
    const int ty = (int)(q % th);
    const int b = (int)(q / th);
    const int x0 = tx * FC_T, y0 = ty * FC_T;
    __syncthreads();
    final_stage_act<T>(As, tpre + (long long)b * hw * 64, y0, x0, H, W, tid);
    for (int i = tid; i < 3 * FC_T * FC_T; i += 256) {
      const int oc = i >> 8, p = i & 255;
      Gs[oc][p] = from_f<T>(gout[((long long)b * 3 + oc) * hw + (long long)(y0 + (p >> 4)) * W + x0 + (p & 15)]);
    }
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < 8; ++ch) {
      // contraction slot (g, j) <-> output pixel o = 32 ch + 8 g + j : tile row 2 ch + (g >> 1), col 8 (g & 1) + j
      const Frag8<T> gf = load8(&Gs[lr][32 * ch + 8 * g]);
      const int orow = 2 * ch + (g >> 1), ocol = 8 * (g & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int hp = (orow + tap / 3) * (FC_T + 2) + ocol + tap % 3;       // halo row of the slot's first pixel
        const Frag8<T> af = load8_tr(&As[hp][16 * wv], &As[hp + 4][16 * wv], FC_LD, lane);
        mma16(acc[tap], gf, af);
      }
    }
  }
  // slab [32 n = tap*3+oc][64 ic]; lane (ic = 16 wv + lr, g) holds rows oc = 4 g + r (only g == 0, r < 3 are real)
  float* out = slabs + (long long)blockIdx.x * (32 * 64);
  if (g == 0) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 3; ++r) out[(tap * 3 + r) * 64 + 16 * wv + lr] = acc[tap][r];
  }
  if (tid < 5 * 64) out[27 * 64 + tid] = 0.f;
}
// ---------------------------------------------------------------------------------------
// fp32 tail conv weight gradient on the VALU (round 5).  The MFMA kernel above pads the 3 output channels to a 16-row tile: in fp32,
// where the matrix rate EQUALS the vector rate, 13 of every 16 products are wasted and the kernel sits at its (padded) MFMA bound,
// 794 us at batch 16 for 1.07 GB of activations.  Here a wave owns 4 output rows x 64 pixels, lane = input channel: the gradient
// values of a pixel are wave-uniform (scalar loads, SGPR operands of v_fmac), the activation rows are 256-byte coalesced row loads
// kept in a sliding 6 x 3 register window (1.5 loads per output pixel), 27 accumulators per lane.  Persistent workgroups; the four
// waves fold through LDS into the slab [32 = tap * 3 + oc | 5 zero rows][64] of final_conv_wgrad_kernel.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) final_conv_wgrad_f32_kernel(const float* __restrict__ gout, const float* __restrict__ act,
                                                                  float* __restrict__ slabs, int B, int H, int W, int units_per_wave) {
  __shared__ float R[4][27][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int segs = (W + 63) >> 6, rgs = H >> 2;
  const int nunits = B * rgs * segs;
  const int wave_id = blockIdx.x * 4 + wv;
  const int u0 = wave_id * units_per_wave, u1 = min(nunits, u0 + units_per_wave);
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
#pragma unroll 1
  for (int u = u0; u < u1; ++u) {
    const int seg = u % segs, q = u / segs, yg = q % rgs, b = q / rgs;
    const int x0 = seg << 6, x1 = min(W, x0 + 64), y0 = yg << 2;
    const float* ab = act + (long long)b * H * W * 64 + lane;
    const float* gb = gout + (long long)b * 3 * H * W + (long long)y0 * W;
    long long rowo[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) rowo[j] = (long long)reflect_idx(y0 - 1 + j, H) * W * 64;
    // columns x - 1 .. x + 4 of the six rows; four pixels per iteration (W % 16 == 0: a segment is a whole number of quads), the gradient
    // values of a quad through one 16-byte scalar load per (row, channel)
    float c[6][6];
    {
      const long long xa = (long long)reflect_idx(x0 - 1, W) * 64, xb = (long long)x0 * 64;
#pragma unroll
      for (int j = 0; j < 6; ++j) { c[0][j] = ab[rowo[j] + xa]; c[1][j] = ab[rowo[j] + xb]; }
    }
#pragma unroll 1
    for (int x = x0; x < x1; x += 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const long long xr = (long long)reflect_idx(x + 1 + i, W) * 64;
#pragma unroll
        for (int j = 0; j < 6; ++j) c[2 + i][j] = ab[rowo[j] + xr];
      }
      f32x4 gq[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int oc = 0; oc < 3; ++oc) gq[r][oc] = *reinterpret_cast<const f32x4*>(gb + (long long)(oc * H + r) * W + x);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g0 = gq[r][0][i], g1 = gq[r][1][i], g2 = gq[r][2][i];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float a0 = c[i][r + ky], a1 = c[i + 1][r + ky], a2 = c[i + 2][r + ky];
            float* ak = &acc[ky * 9];
            ak[0] = fmaf(g0, a0, ak[0]); ak[1] = fmaf(g1, a0, ak[1]); ak[2] = fmaf(g2, a0, ak[2]);
            ak[3] = fmaf(g0, a1, ak[3]); ak[4] = fmaf(g1, a1, ak[4]); ak[5] = fmaf(g2, a1, ak[5]);
            ak[6] = fmaf(g0, a2, ak[6]); ak[7] = fmaf(g1, a2, ak[7]); ak[8] = fmaf(g2, a2, ak[8]);
          }
        }
#pragma unroll
      for (int j = 0; j < 6; ++j) { c[0][j] = c[4][j]; c[1][j] = c[5][j]; }
    }
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) R[wv][k][lane] = acc[k];
  __syncthreads();
  float* out = slabs + (long long)blockIdx.x * (32 * 64);
  for (int i = tid; i < 32 * 64; i += 256) {
    const int k = i >> 6, c = i & 63;
    out[i] = (k < 27) ? (R[0][k][c] + R[1][k][c]) + (R[2][k][c] + R[3][k][c]) : 0.f;
  }
}
template <typename T> static size_t final_wgrad_smem() {
  return sizeof(T) * (FC_HPP * FC_LD + 16 * (FC_T * FC_T + 8));
}
int launch_final_conv_wgrad(int dt, const float* gout, const void* tpre, float* slabs, int* nslab, int B, int H, int W,
                            hipStream_t st) {
  if (H % FC_T || W % FC_T) return m2t_set_error(-2, "final_conv: H,W must be multiples of 16");
  const long long ntiles = (long long)B * (H / FC_T) * (W / FC_T);
  int nblk = (int)std::min<long long>(1024, ntiles);
  const int tpb = (int)ceil_divll(ntiles, nblk);
  nblk = (int)ceil_divll(ntiles, tpb);
  if (dt == M2T_F32 && g_m2t_f32_fast && H % 4 == 0 && (long long)B * H * W * 64 < (1LL << 31)) {
    const int nunits = B * (H / 4) * ((W + 63) / 64);
    int nb = std::min(1024, (nunits + 3) / 4);
    const int upw = (nunits + nb * 4 - 1) / (nb * 4);
    nb = (nunits + upw * 4 - 1) / (upw * 4);
    M2T_LAUNCH_TIMED(final_conv_wgrad_f32_kernel, dim3(nb), dim3(256), 0, st, gout, (const float*)tpre, slabs, B, H, W, upw);
    M2T_LAUNCH_CHECK();
    *nslab = nb;
    return 0;
  }
  if (dt == M2T_F32) {
    const size_t sh = final_wgrad_smem<float>();
    (void)hipFuncSetAttribute((const void*)final_conv_wgrad_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    M2T_LAUNCH_TIMED(final_conv_wgrad_kernel<float>, dim3(nblk), dim3(256), sh, st, gout, (const float*)tpre, slabs, B, H, W, tpb);
  } else {
    const size_t sh = final_wgrad_smem<bf16_t>();
    (void)hipFuncSetAttribute((const void*)final_conv_wgrad_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    M2T_LAUNCH_TIMED(final_conv_wgrad_kernel<bf16_t>, dim3(nblk), dim3(256), sh, st, gout, (const bf16_t*)tpre, slabs, B, H, W, tpb);
  }
  M2T_LAUNCH_CHECK();
  *nslab = nblk;
  return 0;
}
