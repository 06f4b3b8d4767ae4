// m2t_instnorm.h -- the first reduction stage of the InstanceNorm backward (models/M2Trans_network.py:127,135 under autograd),
// shared by its own kernel (k_pointwise.hip) and by c16_dgrad_prep_kernel (k_attn_c16.hip), which runs it as extra workgroups
// beside its own tiles (round 5):
//   s1[b, c] = sum_p g_n[b, p, c] ,  s2[b, c] = sum_p g_n[b, p, c] * xhat[b, p, c]      xhat = (x - mean) * rstd
// One workgroup (256 threads) sums the pixels [p0, p1) of image b for the channel groups CG0 .. CG0 + NCG - 1 (8 channels each)
// and leaves part[b][sp][64][2]; every lane accumulates its pixels in a fixed order and the lanes are folded in a fixed order,
// so the partial depends on nothing but the image (bitwise batch invariance).
#pragma once
#include "m2t_common.h"

template <typename T, int CG0, int NCG>
__device__ __forceinline__ void instnorm_bwd_red1_body(const T* __restrict__ gn, const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd, float* __restrict__ part, int P, int nsplit, int b,
                                                       int sp, int nimg, float (*sh)[8][2], int cg_off = 0) {
  constexpr int NPL = 256 / NCG;                            // pixel lanes (32 for all 64 channels, 42 for the 48 channels of planes 1 .. 3)
  const long long npix = (long long)nimg * P;               // gn, x are P64
  const int tid = threadIdx.x;
  const bool on = tid < NCG * NPL;
  const int cgp = CG0 + cg_off + (on ? tid % NCG : 0), pl = on ? tid / NCG : 0;      // cg_off: this workgroup's first channel group (a launch that splits the channels over gridDim.z)
  const int per = ceil_div(P, nsplit);
  const int p0 = sp * per, p1 = on ? min(P, p0 + per) : 0;
  float mu[8], rs[8], s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { mu[c] = mean[b * 64 + cgp * 8 + c]; rs[c] = rstd[b * 64 + cgp * 8 + c]; s1[c] = 0.f; s2[c] = 0.f; }
  // four pixels per trip: eight 16-byte loads in flight per lane before the first use
  int p = p0 + pl;
  for (; p + 3 * NPL < p1; p += 4 * NPL) {
    float g[4][8], v[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      load8f(gn + p64(npix, (long long)b * P + p + NPL * i, cgp * 8), g[i]);
      load8f(x + p64(npix, (long long)b * P + p + NPL * i, cgp * 8), v[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 8; ++c) { s1[c] += g[i][c]; s2[c] += g[i][c] * ((v[i][c] - mu[c]) * rs[c]); }
  }
  for (; p < p1; p += NPL) {
    float g[8], v[8];
    load8f(gn + p64(npix, (long long)b * P + p, cgp * 8), g);
    load8f(x + p64(npix, (long long)b * P + p, cgp * 8), v);
#pragma unroll
    for (int c = 0; c < 8; ++c) { s1[c] += g[c]; s2[c] += g[c] * ((v[c] - mu[c]) * rs[c]); }
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) { sh[tid][c][0] = s1[c]; sh[tid][c][1] = s2[c]; }
  __syncthreads();
  if (tid < 8 * NCG) {
    const int gi = tid >> 3, c = tid & 7;                   // channel group gi of this workgroup, channel c of it
    float a1 = 0.f, a2 = 0.f;
    for (int l = 0; l < NPL; ++l) { a1 += sh[l * NCG + gi][c][0]; a2 += sh[l * NCG + gi][c][1]; }
    float* o = part + (((long long)b * nsplit + sp) * 64 + (CG0 + cg_off + gi) * 8 + c) * 2;
    o[0] = a1; o[1] = a2;
  }
}
