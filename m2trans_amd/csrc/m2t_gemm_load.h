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
  if (AMODE == M2T_A_HALO) {
    // A = gqkv [B][h][w][3C] whose q part is final; the k|v parts are gathered here from the per-window
    // scratch `win` [B*L][100][2C] of the attention backward: a pixel sums the rows of the (<= 4) windows
    // whose 10x10 key neighbourhood covers it (the overlap-add of F.unfold's backward, as a gather).
    const int C = sg.C, h = sg.H, w = sg.W;
    if (k < C) return load8(A + m * lda + k);
    const T* win = reinterpret_cast<const T*>(sg.aux);
    const int kk = k - C;
    const int x = (int)(m % w);
    const long long q = m / w;
    const int y = (int)(q % h);
    const long long b = q / h;
    const int nh = h >> 3, nw = w >> 3;
    int wys[2], krs[2], ny = 1, wxs[2], kcs[2], nx = 1;
    wys[0] = y >> 3; krs[0] = (y & 7) + 1;
    if ((y & 7) == 0 && wys[0] > 0) { wys[1] = wys[0] - 1; krs[1] = 9; ny = 2; }
    else if ((y & 7) == 7 && wys[0] < nh - 1) { wys[1] = wys[0] + 1; krs[1] = 0; ny = 2; }
    wxs[0] = x >> 3; kcs[0] = (x & 7) + 1;
    if ((x & 7) == 0 && wxs[0] > 0) { wxs[1] = wxs[0] - 1; kcs[1] = 9; nx = 2; }
    else if ((x & 7) == 7 && wxs[0] < nw - 1) { wxs[1] = wxs[0] + 1; kcs[1] = 0; nx = 2; }
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int c = 0; c < nx; ++c) {
        const long long wi = (b * nh + wys[a]) * nw + wxs[c];
        const Frag8<T> f = load8(win + (wi * 100 + krs[a] * 10 + kcs[c]) * (2 * C) + kk);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += f.get(e);
      }
    Frag8<T> r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.set(e, acc[e]);
    return r;
  }
  if (lda == M2T_LD_P64) return load8(A + p64(sg.npix, m, k));
  Frag8<T> f = load8(A + m * lda + k);
  if (AMODE == M2T_A_GELU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.set(e, gelu_erf(f.get(e)));
  }
  return f;
}

