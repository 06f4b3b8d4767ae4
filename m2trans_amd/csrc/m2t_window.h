// m2t_window.h -- geometry of the 8x8-query / 10x10-key halo windows (models/M2Trans_network.py:310-317),
// shared by the attention kernels and the fused branch kernel.
#pragma once
#include "m2t_common.h"

#define WA_NK 100
#define WA_KT 7          // key tiles that can hold real keys (112)
#define WA_KR 128        // key rows staged (4 contraction chunks of 32)
#define WA_QP 72         // 64 queries + 8 pad

struct WinGeom {
  int h, w, nw, nh;
  int b, wy, wx;
  int wi;                 // logical window index (XCD-aware, see xcd_block_index)
  __device__ __forceinline__ bool key_pixel(int key, long long& pix) const {
    const int kr = key / 10, kc = key - kr * 10;
    const int y = 8 * wy + kr - 1, x = 8 * wx + kc - 1;
    pix = ((long long)b * h + y) * w + x;
    return (y >= 0 && y < h && x >= 0 && x < w);
  }
  __device__ __forceinline__ long long query_pixel(int q) const {
    return ((long long)b * h + 8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
  }
};
// dK|dV rows of a window's backward: the 64 keys that are the window's OWN pixels (kr, kc in 1..8) go straight to
// their row of gqkv ([pixel][3C], columns C..3C); the 36 ring keys go to the per-window scratch `win`
// [window][36][2C] and are added by halo_gather to the border pixels of the neighbouring windows.
#define WA_RING 36
__device__ __forceinline__ int ring_index(int kr, int kc) {      // kr or kc in {0, 9}
  return (kr == 0) ? kc : ((kr == 9) ? 10 + kc : ((kc == 0) ? 19 + kr : 27 + kr));
}
// destination row (element offset of its dK|dV part) of key (kr,kc) of window (b,wy,wx); false: phantom key outside
// the image whose dK|dV is dropped (the gradient of zero padding) -- only possible for ring keys
template <typename T>
__device__ __forceinline__ T* dkv_row(T* gqkv, T* win, long long wi, int b, int wy, int wx, int h, int w, int C, int key) {
  const int kr = key / 10, kc = key - kr * 10;
  if (kr >= 1 && kr <= 8 && kc >= 1 && kc <= 8)
    return gqkv + (((long long)b * h + 8 * wy + kr - 1) * w + 8 * wx + kc - 1) * (3 * C) + C;
  return win + (wi * WA_RING + ring_index(kr, kc)) * (2 * C);
}

// The (<= 3) neighbouring windows whose 10x10 neighbourhood covers pixel (y, x) of image b, as element offsets of their
// ring rows in a [window][36][rw] scratch (same order as halo_gather_kernel adds them: row-neighbour column first).
__device__ __forceinline__ int halo_sources(int b, int y, int x, int nh, int nw, int rw, long long (&off)[3]) {
  // branch-free: the three candidates -- A = the column neighbour of the pixel's own window row, B = the row neighbour, C = the diagonal one --
  // are all computed (their indices are valid whether they exist or not) and packed by selects.  As two nested loops with data-dependent
  // trip counts this cost the attention backward's prep role 4.7 k cycles per workgroup (profiles/r06_attn_phase0_integer_work.txt)
  const int wy0 = y >> 3, wx0 = x >> 3, py = y & 7, px = x & 7;
  const bool up = py == 0 && wy0 > 0, dn = py == 7 && wy0 < nh - 1;
  const bool lf = px == 0 && wx0 > 0, rt = px == 7 && wx0 < nw - 1;
  const bool hasy = up || dn, hasx = lf || rt;
  const int wy1 = wy0 + (dn ? 1 : 0) - (up ? 1 : 0), kr1 = up ? 9 : 0;
  const int wx1 = wx0 + (rt ? 1 : 0) - (lf ? 1 : 0), kc1 = lf ? 9 : 0;
  const int kr0 = py + 1, kc0 = px + 1;
  const int row0 = (b * nh + wy0) * nw, row1 = (b * nh + wy1) * nw;          // window counts stay far below 2^31 / 36
  const int eA = (row0 + wx1) * WA_RING + ((kc1 == 0) ? 19 + kr0 : 27 + kr0); // ring_index(kr0 in 1..8, kc1 in {0, 9})
  const int eB = (row1 + wx0) * WA_RING + ((kr1 == 0) ? kc0 : 10 + kc0);      // ring_index(kr1 in {0, 9}, kc0)
  const int eC = (row1 + wx1) * WA_RING + ((kr1 == 0) ? kc1 : 10 + kc1);
  off[0] = (long long)(hasx ? eA : eB) * rw;
  off[1] = (long long)eB * rw;                                               // (read only when both neighbours exist)
  off[2] = (long long)eC * rw;
  return (hasx ? 1 : 0) + (hasy ? 1 : 0) + ((hasx && hasy) ? 1 : 0);
}

__device__ __forceinline__ WinGeom make_geom(int h, int w) {
  WinGeom g;
  g.h = h; g.w = w; g.nw = w / 8; g.nh = h / 8;
#ifdef M2T_WIN_HOT       // scratch/bench_*.hip knock-out (results WRONG): every workgroup works on one of M2T_WIN_HOT windows -- the kernel's
  const int wi = xcd_block_index() % (M2T_WIN_HOT);      // loads and stores hit L2, its HBM traffic all but disappears, its instruction stream stays
#else
  const int wi = xcd_block_index();
#endif
  g.wi = wi;
  g.wx = wi % g.nw;
  const int q = wi / g.nw;
  g.wy = q % g.nh;
  g.b = q / g.nh;
  return g;
}

