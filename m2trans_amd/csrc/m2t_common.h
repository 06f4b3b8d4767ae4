// m2t_common.h -- shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Conventions used by every kernel in this directory
//   * activations live in HBM as NHWC ("pixel-major": [B][H][W][C], channels contiguous)
//     in element type T = float (parity mode) or __bf16 (throughput mode);
//   * all accumulation, statistics, softmax and loss arithmetic is fp32;
//   * contractions run on the matrix cores through ONE abstraction, a 16x16x32 tile
//     product: one v_mfma_f32_16x16x32_bf16 for bf16, eight v_mfma_f32_16x16x4_f32 for
//     f32 (exact fp32, a k-ordered fmaf chain).  Both consume the same per-lane operand
//     shape, "8 consecutive k for (row|col = lane&15), k-group = lane>>4", so a kernel is
//     written once and instantiated for both element types.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define M2T_WAVE 64

enum m2t_dtype { M2T_F32 = 0, M2T_BF16 = 1 };

// ---------------------------------------------------------------------------------------
// 8-element operand fragment
// ---------------------------------------------------------------------------------------
template <typename T> struct Frag8;

template <> struct __attribute__((aligned(16))) Frag8<float> {
  float v[8];
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
  __device__ __forceinline__ float v_elem(int i) const { return v[i]; }
};
template <> struct __attribute__((aligned(16))) Frag8<bf16_t> {
  bf16x8 v;
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16_t)x; }
  __device__ __forceinline__ bf16_t v_elem(int i) const { return v[i]; }
};

template <typename T> __device__ __forceinline__ Frag8<T> frag_zero() {
  Frag8<T> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.set(i, 0.f);
  return f;
}

// 8 consecutive elements from global or LDS (p must be 16-byte aligned)
__device__ __forceinline__ Frag8<float> load8(const float* p) {
  Frag8<float> f;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
  f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
  f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
  return f;
}
__device__ __forceinline__ Frag8<bf16_t> load8(const bf16_t* p) {
  Frag8<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8*>(p);
  return f;
}
__device__ __forceinline__ void store8(float* p, const Frag8<float>& f) {
  f32x4 a = {f.v[0], f.v[1], f.v[2], f.v[3]};
  f32x4 b = {f.v[4], f.v[5], f.v[6], f.v[7]};
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}
__device__ __forceinline__ void store8(bf16_t* p, const Frag8<bf16_t>& f) {
  *reinterpret_cast<bf16x8*>(p) = f.v;
}

// 4 consecutive elements
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
  o[0] = (float)a[0]; o[1] = (float)a[1]; o[2] = (float)a[2]; o[3] = (float)a[3];
}
__device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
  f32x4 a = {o[0], o[1], o[2], o[3]};
  *reinterpret_cast<f32x4*>(p) = a;
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&o)[4]) {
  bf16x4 a = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
  *reinterpret_cast<bf16x4*>(p) = a;
}

template <typename T> __device__ __forceinline__ void load8f(const T* p, float (&o)[8]) {
  const Frag8<T> f = load8(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = f.get(i);
}
template <typename T> __device__ __forceinline__ void store8f(T* p, const float (&o)[8]) {
  Frag8<T> f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f.set(i, o[i]);
  store8(p, f);
}
// 16 consecutive elements from fp32 registers
template <typename T> __device__ __forceinline__ void store16f(T* p, const float (&o)[16]) {
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = o[i]; b[i] = o[8 + i]; }
  store8f(p, a);
  store8f(p + 8, b);
}
template <typename T> __device__ __forceinline__ void load16f(const T* p, float (&o)[16]) {
  float a[8], b[8];
  load8f(p, a);
  load8f(p + 8, b);
#pragma unroll
  for (int i = 0; i < 8; ++i) { o[i] = a[i]; o[8 + i] = b[i]; }
}

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

// ---------------------------------------------------------------------------------------
// 16x16x32 tile product:  D[i][j] += sum_k A[i][k] * B[k][j]
//   lane l supplies A[i = l&15][k = 8*(l>>4) + 0..7]  and  B[k = 8*(l>>4) + 0..7][j = l&15]
//   lane l receives D[i = 4*(l>>4) + r][j = l&15] in acc[r], r = 0..3
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void mma16(f32x4& acc, const Frag8<bf16_t>& a, const Frag8<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma16(f32x4& acc, const Frag8<float>& a, const Frag8<float>& b) {
  // eight exact-fp32 16x16x4 products; product j contracts over the four k = 8g + j, g = 0..3
#pragma unroll
  for (int j = 0; j < 8; ++j)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------
// misc math
// ---------------------------------------------------------------------------------------
// erf(x / sqrt 2) and exp(-x^2 / 2) from ONE exponential: Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 on erf), t = 1 / (1 + p u), u = |x| / sqrt 2.  ~12 VALU instructions instead
// of the ~30 of libm's erff; the tail tensors evaluate GELU on 16x more pixels than the body.
__device__ __forceinline__ void erf_parts(float x, float& erfv, float& ex) {
  const float u = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, u, 1.0f));
  ex = __expf(-u * u);
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  erfv = copysignf(1.0f - poly * ex, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
  float e, ex;
  erf_parts(x, e, ex);
  return 0.5f * x * (1.0f + e);
}
// GELU and its derivative from one erf / exp evaluation
__device__ __forceinline__ void gelu_erf_both(float x, float& act, float& der) {
  float e, ex;
  erf_parts(x, e, ex);
  const float cdf = 0.5f * (1.0f + e);
  act = x * cdf;
  der = cdf + x * (0.39894228040143267794f * ex);
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  float e, ex;
  erf_parts(x, e, ex);
  return 0.5f * (1.0f + e) + x * (0.39894228040143267794f * ex);
}

// bf16 mode's GELU: the results are stored with 8 significant bits, so the erf behind GELU -- what the high-resolution tail
// kernels are bound by -- is replaced by one exp2 and one rcp:
//     Phi(t) ~ sigma(t (a + b t^2 + c t^4)),   gelu(t) = t Phi(t),   gelu'(t) = s + t s (1 - s) (a + 3 b t^2 + 5 c t^4)
// with (a, b, c) fitted (minimax over |t| <= 9, tools/fit_gelu.py) so that |gelu - exact| <= 3.8e-5 and |gelu' - exact| <=
// 9.3e-5: 1 % and 2.4 % of a bf16 ulp at 1.0.  c < 0, so the polynomial is evaluated at t clamped to [-8, 8] (sigma is 0 / 1
// to 1e-12 there).  fp32 parity mode keeps the erf form above.
#define M2T_GELU_A 1.59484566f
#define M2T_GELU_B 7.40076173e-2f
#define M2T_GELU_C -6.95025211e-4f
// the same coefficients times -log2(e): exp(-z) = exp2(t (A2 + B2 t^2 + C2 t^4)) without a separate scaling multiply
#define M2T_GELU_A2 (-1.4426950408889634f * M2T_GELU_A)
#define M2T_GELU_B2 (-1.4426950408889634f * M2T_GELU_B)
#define M2T_GELU_C2 (-1.4426950408889634f * M2T_GELU_C)
// Two values at once.  A wave64 fp32 VALU instruction takes 4 cycles on gfx950 and the packed forms (v_pk_mul_f32, v_pk_add_f32,
// v_pk_fma_f32) process two values per lane in the same 4: the tail kernels are bound by exactly this arithmetic (measured with
// s_memtime stamps, round 4), so the polynomial part is written on float2 -- 3 packed + 3 scalar (med3, exp2, rcp) issue slots per
// value instead of 9 scalar ones; with the derivative 5.5 + 3 instead of 15.  Element-wise IEEE operations: the scalar wrappers
// below give the same bits.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void gelu_fast_core2(f32x2 x, f32x2& tc, f32x2& t2, f32x2& e, f32x2& s) {
  tc[0] = __builtin_amdgcn_fmed3f(x[0], -8.0f, 8.0f);
  tc[1] = __builtin_amdgcn_fmed3f(x[1], -8.0f, 8.0f);
  t2 = tc * tc;
  const f32x2 u = pk_fma(pk_fma((f32x2){M2T_GELU_C2, M2T_GELU_C2}, t2, (f32x2){M2T_GELU_B2, M2T_GELU_B2}), t2, (f32x2){M2T_GELU_A2, M2T_GELU_A2});
  const f32x2 z = tc * u;                                                          // -log2(e) t u(t^2)
  e[0] = __builtin_amdgcn_exp2f(z[0]);                                             // exp(-t u)
  e[1] = __builtin_amdgcn_exp2f(z[1]);
  const f32x2 d = e + (f32x2){1.0f, 1.0f};
  s[0] = __builtin_amdgcn_rcpf(d[0]);                                              // sigma(t u)
  s[1] = __builtin_amdgcn_rcpf(d[1]);
}
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
  f32x2 tc, t2, e, s;
  gelu_fast_core2(x, tc, t2, e, s);
  return x * s;
}
__device__ __forceinline__ void gelu_fast_both2(f32x2 x, f32x2& act, f32x2& der) {
  f32x2 tc, t2, e, s;
  gelu_fast_core2(x, tc, t2, e, s);
  act = x * s;
  const f32x2 du = pk_fma(pk_fma((f32x2){5.0f * M2T_GELU_C, 5.0f * M2T_GELU_C}, t2, (f32x2){3.0f * M2T_GELU_B, 3.0f * M2T_GELU_B}), t2,
                          (f32x2){M2T_GELU_A, M2T_GELU_A});
  der = pk_fma(act * (e * s), du, s);                                              // s + t s (1 - s) u'(t),  1 - s = e s
}
__device__ __forceinline__ void gelu_fast_both(float x, float& act, float& der) {
  f32x2 a, d;
  gelu_fast_both2((f32x2){x, x}, a, d);
  act = a[0]; der = d[0];
}
__device__ __forceinline__ float gelu_fast(float x) { return gelu_fast2((f32x2){x, x})[0]; }
// the tail's activation by storage type: exact (erf) for fp32 parity mode, the approximation for bf16 storage
template <typename T> __device__ __forceinline__ void gelu_tail_both(float x, float& act, float& der) {
  if constexpr (sizeof(T) == 2) gelu_fast_both(x, act, der); else gelu_erf_both(x, act, der);
}
template <typename T> __device__ __forceinline__ float gelu_tail(float x) {
  if constexpr (sizeof(T) == 2) return gelu_fast(x); else return gelu_erf(x);
}
// four values (one accumulator tile's registers of a lane) + their bias: act (and der) by storage type
template <typename T> __device__ __forceinline__ void gelu_tail4(const f32x4& acc, const float (&bias)[4], float (&act)[4]) {
  if constexpr (sizeof(T) == 2) {
    const f32x2 a0 = gelu_fast2((f32x2){acc[0], acc[1]} + (f32x2){bias[0], bias[1]});
    const f32x2 a1 = gelu_fast2((f32x2){acc[2], acc[3]} + (f32x2){bias[2], bias[3]});
    act[0] = a0[0]; act[1] = a0[1]; act[2] = a1[0]; act[3] = a1[1];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) act[r] = gelu_erf(acc[r] + bias[r]);
  }
}
// eight values = the two accumulator tiles whose rows interleave to 8 consecutive channels (k_tail_stream.hip), in STAGES over the
// four pairs: independent transcendentals sit next to each other, so hipcc needs no s_nop behind each v_exp / v_rcp (a wait
// state is due before a VALU instruction reads a transcendental's result; in the pair-by-pair form it filled them with 66 s_nop
// per step of the row-streaming kernel, which is bound by instruction issue).  Same element-wise operations: same bits.
template <typename T> __device__ __forceinline__ void gelu_tail8(const f32x4& lo, const f32x4& hi, const f32x4& blo, const f32x4& bhi, float (&act)[8]) {
  if constexpr (sizeof(T) == 2) {
    f32x2 x[4] = {(f32x2){lo[0], lo[1]} + (f32x2){blo[0], blo[1]}, (f32x2){lo[2], lo[3]} + (f32x2){blo[2], blo[3]},
                  (f32x2){hi[0], hi[1]} + (f32x2){bhi[0], bhi[1]}, (f32x2){hi[2], hi[3]} + (f32x2){bhi[2], bhi[3]}};
    f32x2 z[4], e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -8.0f, 8.0f);
      z[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -8.0f, 8.0f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 t2 = z[i] * z[i];
      const f32x2 u = pk_fma(pk_fma((f32x2){M2T_GELU_C2, M2T_GELU_C2}, t2, (f32x2){M2T_GELU_B2, M2T_GELU_B2}), t2, (f32x2){M2T_GELU_A2, M2T_GELU_A2});
      z[i] = z[i] * u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i][0] = __builtin_amdgcn_exp2f(z[i][0]); e[i][1] = __builtin_amdgcn_exp2f(z[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) e[i] = e[i] + (f32x2){1.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i][0] = __builtin_amdgcn_rcpf(e[i][0]); z[i][1] = __builtin_amdgcn_rcpf(e[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const f32x2 a = x[i] * z[i]; act[2 * i] = a[0]; act[2 * i + 1] = a[1]; }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) { act[r] = gelu_erf(lo[r] + blo[r]); act[4 + r] = gelu_erf(hi[r] + bhi[r]); }
  }
}
// ... and with the derivative (the recomputing backward kernels): act = gelu(x), der = gelu'(x), staged the same way
template <typename T> __device__ __forceinline__ void gelu_tail_both8(const f32x4& lo, const f32x4& hi, const f32x4& blo, const f32x4& bhi, float (&act)[8],
                                                                     float (&der)[8]) {
  if constexpr (sizeof(T) == 2) {
    f32x2 x[4] = {(f32x2){lo[0], lo[1]} + (f32x2){blo[0], blo[1]}, (f32x2){lo[2], lo[3]} + (f32x2){blo[2], blo[3]},
                  (f32x2){hi[0], hi[1]} + (f32x2){bhi[0], bhi[1]}, (f32x2){hi[2], hi[3]} + (f32x2){bhi[2], bhi[3]}};
    f32x2 t2[4], z[4], e[4], s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      z[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -8.0f, 8.0f);
      z[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -8.0f, 8.0f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t2[i] = z[i] * z[i];
      const f32x2 u = pk_fma(pk_fma((f32x2){M2T_GELU_C2, M2T_GELU_C2}, t2[i], (f32x2){M2T_GELU_B2, M2T_GELU_B2}), t2[i], (f32x2){M2T_GELU_A2, M2T_GELU_A2});
      z[i] = z[i] * u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { e[i][0] = __builtin_amdgcn_exp2f(z[i][0]); e[i][1] = __builtin_amdgcn_exp2f(z[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = e[i] + (f32x2){1.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) { s[i][0] = __builtin_amdgcn_rcpf(z[i][0]); s[i][1] = __builtin_amdgcn_rcpf(z[i][1]); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x2 a = x[i] * s[i];
      const f32x2 du = pk_fma(pk_fma((f32x2){5.0f * M2T_GELU_C, 5.0f * M2T_GELU_C}, t2[i], (f32x2){3.0f * M2T_GELU_B, 3.0f * M2T_GELU_B}), t2[i],
                              (f32x2){M2T_GELU_A, M2T_GELU_A});
      const f32x2 d = pk_fma(a * (e[i] * s[i]), du, s[i]);
      act[2 * i] = a[0]; act[2 * i + 1] = a[1];
      der[2 * i] = d[0]; der[2 * i + 1] = d[1];
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) { gelu_erf_both(lo[r] + blo[r], act[r], der[r]); gelu_erf_both(hi[r] + bhi[r], act[4 + r], der[4 + r]); }
  }
}
template <typename T> __device__ __forceinline__ void gelu_tail_both4(const f32x4& acc, const float (&bias)[4], float (&act)[4], float (&der)[4]) {
  if constexpr (sizeof(T) == 2) {
    f32x2 a0, d0, a1, d1;
    gelu_fast_both2((f32x2){acc[0], acc[1]} + (f32x2){bias[0], bias[1]}, a0, d0);
    gelu_fast_both2((f32x2){acc[2], acc[3]} + (f32x2){bias[2], bias[3]}, a1, d1);
    act[0] = a0[0]; act[1] = a0[1]; act[2] = a1[0]; act[3] = a1[1];
    der[0] = d0[0]; der[1] = d0[1]; der[2] = d1[0]; der[3] = d1[1];
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) gelu_erf_both(acc[r] + bias[r], act[r], der[r]);
  }
}

__device__ __forceinline__ int reflect_idx(int i, int n) {   // torch 'reflect' (no edge repeat)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// ---------------------------------------------------------------------------------------
// Transposed operand read (bf16): ds_read_b64_tr_b16.
// The tile sits ROW-major in LDS as [contraction index][row|col index] (exactly as it was
// copied from HBM with 16-byte vectors); the hardware hands lane (i = lane&15, g = lane>>4)
// the column i of four consecutive LDS rows.  Two reads give the 8-element operand:
//   elements 0..3 = rows r0 + 0..3, elements 4..7 = rows r1 + 0..3, all at column c0 + i.
// p0 / p1 point at &tile[r0][c0] / &tile[r1][c0]; ld = row stride in elements (multiple of 4).
// EXEC must be all ones (every lane of the wave calls this).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ Frag8<bf16_t> load8_tr(const bf16_t* p0, const bf16_t* p1, int ld, int lane) {
  const int i = lane & 15, q = i >> 2, pp = i & 3;
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p0 + q * ld + 4 * pp));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p1 + q * ld + 4 * pp));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// fp32 has no transposing read: gather the 8 elements one by one (parity mode only)
__device__ __forceinline__ Frag8<float> load8_tr(const float* p0, const float* p1, int ld, int lane) {
  const int i = lane & 15;
  Frag8<float> f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f.v[e] = p0[e * ld + i];
    f.v[4 + e] = p1[e * ld + i];
  }
  return f;
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0) + s_barrier): for kernels whose waves exchange
// data through LDS alone.  __syncthreads() is a workgroup-scope fence + barrier; the fence may add vmcnt(0) (it does
// whenever LDS-DMA or stores are pending in hipcc's model), draining prefetches that were meant to stay in flight.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Workgroup i of a 1-D grid is dispatched to XCD i % 8 (round-robin).  Returns a LOGICAL index such that each XCD
// owns one contiguous run of logical indices, so spatial neighbours (tiles / windows of the same image, which share
// halo rows) sit behind ONE L2.  Identity when the grid is not a multiple of 8.
__device__ __forceinline__ int xcd_block_index() {
  const int n = gridDim.x, i = blockIdx.x;
  return (n & 7) ? i : (i & 7) * (n >> 3) + (i >> 3);
}

// host-side launch check
#define M2T_LAUNCH_CHECK()                                  \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return m2t_set_hip_error(e__, __FILE__, __LINE__); \
  } while (0)

int m2t_set_hip_error(hipError_t e, const char* file, int line);
int m2t_set_error(int code, const char* msg);
// Raises the dynamic-LDS limit of `kernel` to `bytes` on the CURRENT device, once per (calling thread, device, kernel):
// the attribute is per device, and the cache is thread-local, so the library keeps no process-global mutable state
// (two host threads driving two GPUs each set it for their own device).  Returns 0 or the hipError_t.
int m2t_ensure_dynamic_lds(const void* kernel, int bytes);

// ---------------------------------------------------------------------------------------
// "P64": the chunk-planar layout of every 64-channel low-resolution feature map (X_b, xc, their gradients):
// [4 chunks][npix][16 channels].  Each CFTM branch works on ONE 16-channel chunk (models/M2Trans_network.py:
// 129-161); with channel-interleaved rows [npix][64] every such access fetched the whole 128-byte row for 32
// useful bytes (rocprof FETCH_SIZE: 33.6 MB per branch_prep launch for 8.4 MB of input).
// ---------------------------------------------------------------------------------------
__host__ __device__ static inline long long p64(long long npix, long long pix, int c) {
  return ((long long)(c >> 4) * npix + pix) * 16 + (c & 15);
}

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ static inline long long ceil_divll(long long a, long long b) { return (a + b - 1) / b; }
