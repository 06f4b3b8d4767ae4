// m2t_gemm_load.h -- A-side row loaders shared by the GEMM, wgrad and column-sum kernels.
#pragma once
#include "m2t_kernels.h"

struct ShufGeom { int H, W, r, C; const void* aux = nullptr; long long npix = 0; };   // npix: rows of a P64 operand

// logical A row m, logical column k..k+7  ->  8 elements
template <typename T, int AMODE>
__device__ __forceinline__ Frag8<T> gemm_load_a(const T* __restrict__ A, int lda, long long m, int k, const ShufGeom& sg) {
  if (AMODE == M2T_A_UNSHUF) {
    // A is the shuffled tensor [B][H*r][W*r][C]; row m = (b,h,w); column k = sub*C + c, sub = i*r + j
    const int sub = k / sg.C, c = k - sub * sg.C;
    const int i = sub / sg.r, j = sub - i * sg.r;
    const int w = (int)(m % sg.W);
    const long long q = m / sg.W;
    const int h = (int)(q % sg.H);
    const long long b = q / sg.H;
    const long long pix = (b * sg.H * sg.r + (h * sg.r + i)) * ((long long)sg.W * sg.r) + (w * sg.r + j);
    return load8(A + pix * sg.C + c);
  }
  if (lda == M2T_LD_P64) return load8(A + p64(sg.npix, m, k));
  Frag8<T> f = load8(A + m * lda + k);
  if (AMODE == M2T_A_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.set(e, gelu_erf(f.get(e)));
  }
  return f;
}

