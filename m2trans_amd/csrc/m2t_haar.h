// m2t_haar.h -- L-level Haar DWT / IWT on registers (models/M2Trans_network.py:198-237), shared by the
// pointwise kernels and by the attention kernels that fuse the transforms into their loads / stores.
#pragma once
#include "m2t_common.h"

// =======================================================================================
// Haar butterflies on registers (association order = the reference's, so fp32 is bit-exact)
// =======================================================================================
__device__ __forceinline__ void haar2_fwd(float a, float b, float c, float d, float (&o)[4]) {
  // a=(even r,even c) b=(odd r,even c) c=(even r,odd c) d=(odd r,odd c)  (:203-207)
  o[0] = 0.5f * (((a + b) + c) + d);
  o[1] = 0.5f * (((-a - b) + c) + d);
  o[2] = 0.5f * (((-a + b) - c) + d);
  o[3] = 0.5f * (((a - b) - c) + d);
}
__device__ __forceinline__ void haar2_inv(float ll, float hl, float lh, float hh, float& a, float& b,
                                          float& c, float& d) {
  a = 0.5f * (((ll - hl) - lh) + hh);   // even r, even c   (:225)
  b = 0.5f * (((ll - hl) + lh) - hh);   // odd r,  even c   (:227)
  c = 0.5f * (((ll + hl) - lh) - hh);   // even r, odd c    (:229)
  d = 0.5f * (((ll + hl) + lh) + hh);   // odd r,  odd c    (:231)
}

// L-level transform of one channel of a (2^L x 2^L) pixel block.
// in: v[y][x];  out: o[band index], band index = band_L * 4^(L-1) + ... + band_1 (band-major
// nesting exactly as repeated torch.cat((LL,HL,LH,HH),1) produces).
template <int L> struct Haar;
template <> struct Haar<0> {
  static constexpr int S = 1, N = 1;
  __device__ static __forceinline__ void fwd(const float (&v)[1][1], float (&o)[1]) { o[0] = v[0][0]; }
  __device__ static __forceinline__ void inv(const float (&o)[1], float (&v)[1][1]) { v[0][0] = o[0]; }
};
template <> struct Haar<1> {
  static constexpr int S = 2, N = 4;
  __device__ static __forceinline__ void fwd(const float (&v)[2][2], float (&o)[4]) {
    haar2_fwd(v[0][0], v[1][0], v[0][1], v[1][1], o);
  }
  __device__ static __forceinline__ void inv(const float (&o)[4], float (&v)[2][2]) {
    haar2_inv(o[0], o[1], o[2], o[3], v[0][0], v[1][0], v[0][1], v[1][1]);
  }
};
template <> struct Haar<2> {
  static constexpr int S = 4, N = 16;
  __device__ static __forceinline__ void fwd(const float (&v)[4][4], float (&o)[16]) {
    float t[2][2][4];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
        haar2_fwd(v[2 * I][2 * J], v[2 * I + 1][2 * J], v[2 * I][2 * J + 1], v[2 * I + 1][2 * J + 1], t[I][J]);
#pragma unroll
    for (int b1 = 0; b1 < 4; ++b1) {
      float r[4];
      haar2_fwd(t[0][0][b1], t[1][0][b1], t[0][1][b1], t[1][1][b1], r);
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) o[b2 * 4 + b1] = r[b2];
    }
  }
  __device__ static __forceinline__ void inv(const float (&o)[16], float (&v)[4][4]) {
    float t[2][2][4];
#pragma unroll
    for (int b1 = 0; b1 < 4; ++b1)
      haar2_inv(o[0 * 4 + b1], o[1 * 4 + b1], o[2 * 4 + b1], o[3 * 4 + b1], t[0][0][b1], t[1][0][b1],
                t[0][1][b1], t[1][1][b1]);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
        haar2_inv(t[I][J][0], t[I][J][1], t[I][J][2], t[I][J][3], v[2 * I][2 * J], v[2 * I + 1][2 * J],
                  v[2 * I][2 * J + 1], v[2 * I + 1][2 * J + 1]);
  }
};

// The forward transforms on two channels at once (f32x2 -> v_pk_add_f32 / v_pk_mul_f32: IEEE results per component, same
// association order as above, so the bits are those of the scalar form).  Used where the butterflies sit inside an attention kernel.
__device__ __forceinline__ void haar2_fwd2(f32x2 a, f32x2 b, f32x2 c, f32x2 d, f32x2 (&o)[4]) {
  // -a - b = -(a + b), -a + b = -(a - b), -t - c = -(t + c) and -s + c = c - s hold exactly in IEEE arithmetic (rounding is sign-symmetric),
  // so two shared sums replace the negations: 10 packed adds instead of 12 adds + negation moves
  const f32x2 hf = {0.5f, 0.5f};
  const f32x2 s = a + b, t = a - b;
  o[0] = hf * ((s + c) + d);
  o[1] = hf * ((c - s) + d);
  o[2] = hf * (d - (t + c));
  o[3] = hf * ((t - c) + d);
}
template <int L> struct Haar2;
template <> struct Haar2<1> {
  __device__ static __forceinline__ void fwd(const f32x2 (&v)[2][2], f32x2 (&o)[4]) { haar2_fwd2(v[0][0], v[1][0], v[0][1], v[1][1], o); }
};
template <> struct Haar2<2> {
  __device__ static __forceinline__ void fwd(const f32x2 (&v)[4][4], f32x2 (&o)[16]) {
    f32x2 t[2][2][4];
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
        haar2_fwd2(v[2 * I][2 * J], v[2 * I + 1][2 * J], v[2 * I][2 * J + 1], v[2 * I + 1][2 * J + 1], t[I][J]);
#pragma unroll
    for (int b1 = 0; b1 < 4; ++b1) {
      f32x2 r[4];
      haar2_fwd2(t[0][0][b1], t[1][0][b1], t[0][1][b1], t[1][1][b1], r);
#pragma unroll
      for (int b2 = 0; b2 < 4; ++b2) o[b2 * 4 + b1] = r[b2];
    }
  }
};

// inverse butterflies on two channels at once: the association order of haar2_inv with its two shared differences / sums named
__device__ __forceinline__ void haar2_inv2(f32x2 ll, f32x2 hl, f32x2 lh, f32x2 hh, f32x2& a, f32x2& b, f32x2& c, f32x2& d) {
  const f32x2 hf = {0.5f, 0.5f};
  const f32x2 m = ll - hl, p = ll + hl;
  a = hf * ((m - lh) + hh);
  b = hf * ((m + lh) - hh);
  c = hf * ((p - lh) - hh);
  d = hf * ((p + lh) + hh);
}
template <> struct Haar2<0> {};
struct Haar2Inv2 {      // L = 2
  __device__ static __forceinline__ void inv(const f32x2 (&o)[16], f32x2 (&v)[4][4]) {
    f32x2 t[2][2][4];
#pragma unroll
    for (int b1 = 0; b1 < 4; ++b1)
      haar2_inv2(o[0 * 4 + b1], o[1 * 4 + b1], o[2 * 4 + b1], o[3 * 4 + b1], t[0][0][b1], t[1][0][b1], t[0][1][b1], t[1][1][b1]);
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
      for (int J = 0; J < 2; ++J)
        haar2_inv2(t[I][J][0], t[I][J][1], t[I][J][2], t[I][J][3], v[2 * I][2 * J], v[2 * I + 1][2 * J], v[2 * I][2 * J + 1], v[2 * I + 1][2 * J + 1]);
  }
};
