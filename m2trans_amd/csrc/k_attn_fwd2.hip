// k_attn_fwd2.hip -- the fused C = 256 forward branch (branch_prep + qkv projection + window attention + IWT^2 / residual,
// models/M2Trans_network.py:141-161,281,307-332) built for TWO windows per CU (round 5; option "fused_attn_fwd2", off by default).
//
// window_attn_fused_fwd_kernel<256, 2, 8, true> (k_attn_fused.hip) keeps a whole window in one 8-wave workgroup: 155 KB of LDS
// and every K | V | Q output tile in accumulators (295 KB of registers), i.e. ONE window per CU, whose phases -- load burst,
// projection, stores, softmax, epilogue -- run one after the other with nothing to overlap them.  Two such windows cannot share a
// CU: their accumulators alone (590 KB) exceed the register file (512 KB).  This kernel trades LDS operand traffic for capacity:
//
//   * a window = 4 waves, <= 256 registers, 79.6 KB of LDS.  TW = 1: one window per workgroup, two workgroups per CU.  TW = 2: two
//     neighbouring windows per 8-wave workgroup (waves 0..3 / 4..7), one workgroup per CU, shared barriers;
//   * the projection runs in four OUTPUT-channel chunks of 64: wave u of a window owns channel tile 4 c + u of K, V and Q (18
//     accumulator tiles), streams its three weight fragments per k-step from L2 through a 4-deep ring and reads the x rows from LDS
//     one k-step ahead (double-buffered by name) -- every x fragment feeds 3 (query tiles) or 2 MFMAs;
//   * K^ (+ rel-pos) and q of a chunk go through a 24 KB LDS buffer and are consumed at once: S^T += K^_c q_c^T with wave u = query
//     tile u over ALL 112 keys, so the scores, the softmax and P^T never leave the registers (no statistics merge, and the
//     accumulator layout of S^T is the B-operand layout of O^T = V^T P^T);
//   * v leaves for HBM as it is produced (own pixels: the qkv tensor the backward reads; ring keys: a per-window scratch) and
//     comes back chunk by chunk through the same LDS buffer for P V -- an L2 hit, 51 KB per window;
//   * O^T (all 256 channels of the wave's 16 queries) stays in registers: a lane holds the 16 bands of 4 base channels, i.e. the
//     whole 4 x 4 pixel block of IWT^2 -- the epilogue needs no staging.
// Same products in the same k order as the 8-wave kernel: q | k | v are bit-identical; S^T accumulates the same 8 k-steps in the
// same order (bit-identical scores); the softmax is the single-pass form and P V sums the keys in another slot order: the output
// agrees to fp32 summation order (rare 1-ulp flips of the bf16 result).
//
// MEASURED (scratch/bench_fwd2.hip, batch 32, 512 windows; profiles/r05_attn_fwd2_stamps.txt): 59.3 us (TW = 1) / 64.9 us (TW = 2)
// against 61.7 us for the 8-wave kernel -- a tie inside the step.  Per window (s_memtime, TW = 1): phase 0 27.5 k cycles, phase A
// 51 k, softmax + P V + epilogue 19 k = 102 k, i.e. exactly what the 8-wave kernel takes for TWO windows one after the other
// (2 x 51 k): the co-resident windows start in the same cycle and stay in the same phase (512 windows = one round of the chip), so
// they compete for the same pipe at every moment instead of filling each other's waits; the weight stream (knocked out: phase A
// 43 k) is not what bounds it, and SQ_VALU_MFMA_BUSY_CYCLES is 18 % of the launch for both kernels.  What the experiment cost in
// hipcc lessons is in DESIGN.md ("Round 5: the two-windows-per-CU forward kernel"): FLAT accesses from laundered pointers, address
// registers kept alive across unrolled chunks, non-temporal stores (slower).
#include "m2t_kernels.h"
#include "m2t_haar.h"
#include "m2t_window.h"

#ifndef M2T_FWD2_STAMP
#define M2T_FWD2_STAMP(i) do { } while (0)      // scratch/bench_fwd2.hip defines it to record s_memtime per phase
#endif

#ifndef M2T_FWD2_NT
#define M2T_FWD2_NT 0                           // streaming accesses marked non-temporal (scratch/bench_fwd2.hip builds both)
#endif

namespace {

// Streaming traffic of the kernel -- the block loads of phase 0, the q | k | d | xin | out stores -- is marked NON-TEMPORAL: with two
// windows per CU in different phases it flows through the XCD's 4 MB L2 all the time (~350 KB per window) and evicts the 393 KB of
// weight fragments every window streams, which then come over the fabric instead of out of L2.  (v is stored normally: the kernel
// reads it back.)
template <typename P, typename V> __device__ __forceinline__ void f2_nt_store(P p, const V& v) {   // P: pointer to V in any address space
#if M2T_FWD2_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
template <typename V> __device__ __forceinline__ V f2_nt_load(const V* p) {
#if M2T_FWD2_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

constexpr int F2_C = 256, F2_LD = F2_C + 8, F2_CLD = 64 + 8, F2_ZR = 100;
// TW = windows per workgroup: 1 -> 4 waves, 80 KB: two workgroups per CU; 2 -> 8 waves, 157 KB: ONE workgroup per CU whose two windows
// (horizontal neighbours) share every weight fragment -- see the header
template <int TW> struct F2Cfg {
  static constexpr size_t szX = sizeof(bf16_t) * 101 * F2_LD;       // x rows (own pixels first, then the ring), row 100 = zeros
  static constexpr size_t szK = sizeof(bf16_t) * 101 * F2_CLD;      // K^ slice [key][64]; later the v chunk; row 100 = zeros
  static constexpr size_t szQ = sizeof(bf16_t) * 64 * F2_CLD;       // q slice [query][64]
  static constexpr size_t szR = sizeof(float) * 10 * 64;            // rel-pos slice [10][64]
  static constexpr size_t offX = 0, offK = TW * szX, offQ = offK + TW * szK, offR = offQ + TW * szQ;
  static constexpr size_t total = offR + szR;
  static_assert(total <= (TW == 1 ? 81920 : 163840), "LDS budget: two workgroups per CU (TW = 1) or one (TW = 2)");
  static_assert(szX % 16 == 0 && szK % 16 == 0 && szQ % 16 == 0, "16-byte carve offsets");
};

struct Fwd2Args {
  const bf16_t* xn;        // plane k of the block input X  [B][H][W][16]
  const bf16_t* xprev;     // plane k - 1 of the concat buffer xc
  const float* mean;       // [B][64]
  const float* rstd;
  bf16_t* xin;             // [B][H][W][16]   (written: own pixels; read back as the epilogue's residual)
  bf16_t* d;               // [B][h][w][256]  (written: own pixels)
  const bf16_t* wfrag;     // qkv weight, M2T_PACK_FRAG16
  const float* rel_h;      // [10][128]
  const float* rel_w;      // [10][128]
  bf16_t* qkv;             // [B][h][w][768]  (written)
  bf16_t* out;             // plane k of xc   [B][H][W][16]
  bf16_t* vring;           // [window][36][256] scratch: v of the ring keys
  int k, h, w;
  int stagger;             // TW = 1: workgroups of the odd rounds of 256 (the second workgroup of each CU) start this many x 8 k cycles late
};

// key order: 0..63 the window's own pixels (= query order), 64..99 the ring in ring_index order (branch-free)
__device__ __forceinline__ void f2_key_rc(int k, int& kr, int& kc) {
  const int r = k - 64;
  const int kr_ring = (r < 10) ? 0 : ((r < 20) ? 9 : ((r < 28) ? r - 19 : r - 27));
  const int kc_ring = (r < 10) ? r : ((r < 20) ? r - 10 : ((r < 28) ? 0 : 9));
  kr = (k < 64) ? (k >> 3) + 1 : kr_ring;
  kc = (k < 64) ? (k & 7) + 1 : kc_ring;
}
__device__ __forceinline__ bf16x4 f2_pack4(const f32x4& a) {
  bf16x4 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
  return r;
}
// transposed 8-element operand out of a [row][channel] tile: elements 0..3 = rows lo + 0..3, 4..7 = rows hi + 0..3 at column
// col0 + (lane & 15); rows >= zero_row alias the zero row
__device__ __forceinline__ Frag8<bf16_t> f2_tr8(const bf16_t* base, int ld, int lo, int hi, int col0, int lane, int zero_row) {
  const int i = lane & 15, qq = i >> 2, pp = i & 3;
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const int rlo = min(lo + qq, zero_row), rhi = min(hi + qq, zero_row);
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rlo * ld + col0 + 4 * pp));
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rhi * ld + col0 + 4 * pp));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// window `wi` (logical index) of an h x w map
__device__ __forceinline__ WinGeom f2_geom(int h, int w, int wi) {
  WinGeom g;
  g.h = h; g.w = w; g.nw = w / 8; g.nh = h / 8;
  g.wi = wi;
  g.wx = wi % g.nw;
  const int q = wi / g.nw;
  g.wy = q % g.nh;
  g.b = q / g.nh;
  return g;
}

// pointers that KEEP their address space through laundering asm statements and hand-made offsets (a generic pointer turns every
// access into a FLAT one, which counts in vmcnt AND lgkmcnt: each wait then drains both queues)
typedef const bf16_t __attribute__((address_space(3))) * f2_lds_cptr;
typedef const bf16x8 __attribute__((address_space(1))) * f2_glb_frag_cptr;
__device__ __forceinline__ Frag8<bf16_t> f2_lds_load8(f2_lds_cptr p) {
  Frag8<bf16_t> f;
  f.v = *reinterpret_cast<const bf16x8 __attribute__((address_space(3)))*>(p);
  return f;
}

template <int TW>
__global__ void __launch_bounds__(256 * TW, 2) window_attn_fwd2_kernel(Fwd2Args pa) {
  using T = bf16_t;
  using Cfg = F2Cfg<TW>;
  constexpr int C = F2_C, LD = F2_LD, CLD = F2_CLD, ZR = F2_ZR, NW = 4 * TW, NTHR = 64 * NW, NT = C / 16, NKS = C / 32;
  constexpr int NCH = 4;                                               // projection chunks of 64 output channels: wave (window, u) owns channel tile 4 c + u
  constexpr int L = 2, S = 4, NPX = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Xs)[101][LD] = reinterpret_cast<T(*)[101][LD]>(smem + Cfg::offX);         // [window][key][channel]
  T(*Kc)[101][CLD] = reinterpret_cast<T(*)[101][CLD]>(smem + Cfg::offK);       // K^ slice; phase C: v chunk
  T(*Qc)[64][CLD] = reinterpret_cast<T(*)[64][CLD]>(smem + Cfg::offQ);
  float(*RelC)[64] = reinterpret_cast<float(*)[64]>(smem + Cfg::offR);

  int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave-uniform: scalar address arithmetic
  int lr = lane & 15, g = lane >> 4;
  const int h = pa.h, w = pa.w;
  const int wi0 = TW * xcd_block_index();                               // the workgroup's first window (TW = 2: windows wi0, wi0 + 1)
  const int mywin = wv >> 2, qt = wv & 3;                               // attention role of the wave: (window, query tile)
  const WinGeom gm = f2_geom(h, w, wi0 + mywin);
  // fragment (projection p: 0 = q, 1 = k, 2 = v; channel tile NW c + wv; k-step ks): one contiguous 1 KB per wave at a wave-uniform
  // address -- a scalar base (the wave's tile 0 of q) plus COMPILE-TIME offsets, the lane's 16 bytes as the only vector part (left to
  // itself hipcc kept a 64-bit vector address per load alive from one chunk to the next: 32 registers, all spilled)
  const f2_glb_frag_cptr wbase = (f2_glb_frag_cptr)(reinterpret_cast<const bf16x8*>(pa.wfrag) + (size_t)(wv & 3) * (NKS * 64));
  auto wload = [&](int p, int c, int ks) -> Frag8<T> {                  // p, c, ks compile-time after unrolling
    Frag8<T> f;
    f2_glb_frag_cptr fp = wbase + ((p * NT + 4 * c) * NKS + ks) * 64;
    asm volatile("" : "+s"(fp));                                        // the uniform part stays a scalar pair: saddr + 32-bit lane offset
    f.v = fp[lane];
    return f;
  };
  // rel-pos slice sl (channels 64 sl .. 64 sl + 63): T[kk][ch] = ch < C/2 ? rel_h[kk][ch] : rel_w[kk][ch - C/2]
  auto rel_src = [&](int sl, int idx) -> const float* {
    const int kk = idx >> 4, c4 = (idx & 15) * 4;
    return (sl < 2) ? (pa.rel_h + kk * (C / 2) + 64 * sl + c4) : (pa.rel_w + kk * (C / 2) + 64 * (sl - 2) + c4);
  };

  // Two workgroups that start on a CU in the same cycle stay in the same phase to the end and compete for the same pipe at every
  // moment; offset by about half a window, one's MFMA phase falls on the other's load / store / softmax phases
  if (TW == 1 && pa.stagger > 0 && ((blockIdx.x >> 8) & 1)) {
    for (int i = 0; i < pa.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  M2T_FWD2_STAMP(0);
  // ======== phase 0: branch_prep of the 100 keys' 4 x 4 pixel blocks -> x rows (DWT^2 in place), xin of the own pixels ========
  {
    constexpr int NITEM1 = 100 * NPX * 2, NITEM = TW * NITEM1, PIT = (NITEM + NTHR - 1) / NTHR;      // 16-byte items: [window][key row][pixel row][key col][pixel col][half]
    const int half = tid & 1;
    const int H = h * S, W = w * S;
    // (TW = 2: the launcher guarantees an even number of windows per image, so both windows share the image's statistics)
    float mu[8], rs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mu[c] = pa.mean[gm.b * 64 + pa.k * 16 + half * 8 + c]; rs[c] = pa.rstd[gm.b * 64 + pa.k * 16 + half * 8 + c]; }
    Frag8<T> xf[PIT], pf[PIT];
    bool ok[PIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int ii = min(tid + it * NTHR, NITEM - 1);
      const int win = ii / NITEM1, i = ii - win * NITEM1;
      const WinGeom gw = f2_geom(h, w, wi0 + win);
      const int px = (i >> 1) % S, kx = (i / (2 * S)) % 10, py = (i / (20 * S)) % S, ky = i / (20 * NPX);
      const int yy = 8 * gw.wy + ky - 1, xx = 8 * gw.wx + kx - 1;
      ok[it] = yy >= 0 && yy < h && xx >= 0 && xx < w;
      const int Y = S * min(max(yy, 0), h - 1) + py, X = S * min(max(xx, 0), w - 1) + px;     // clamped: the loads are unconditional
      const long long o = (((long long)gw.b * H + Y) * W + X) * 16 + half * 8;
      xf[it].v = f2_nt_load(reinterpret_cast<const bf16x8*>(pa.xn + o));
      pf[it].v = f2_nt_load(reinterpret_cast<const bf16x8*>(pa.xprev + o));
    }
    const f32x4 rf = *reinterpret_cast<const f32x4*>(rel_src(0, min(tid, 159)));
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int ii = tid + it * NTHR;
      const int iic = min(ii, NITEM - 1);
      const int win = iic / NITEM1, ic = iic - win * NITEM1;
      const WinGeom gw = f2_geom(h, w, wi0 + win);
      const int px = (ic >> 1) % S, kx = (ic / (2 * S)) % 10, py = (ic / (20 * S)) % S, ky = ic / (20 * NPX);
      const bool own = ky >= 1 && ky <= 8 && kx >= 1 && kx <= 8;
      const int row = own ? (ky - 1) * 8 + (kx - 1) : 64 + ring_index(ky, kx);
      float q[8];
#pragma unroll
      for (int c = 0; c < 8; c += 2) {
        f32x2 v = ((f32x2){xf[it].get(c), xf[it].get(c + 1)} - (f32x2){mu[c], mu[c + 1]}) * (f32x2){rs[c], rs[c + 1]};
        v = (v + (f32x2){pf[it].get(c), pf[it].get(c + 1)}) * (f32x2){0.5f, 0.5f};
        q[c] = ok[it] ? v[0] : 0.f;
        q[c + 1] = ok[it] ? v[1] : 0.f;
      }
      if (ii < NITEM) {
        store8f(&Xs[win][row][(py * S + px) * 16 + half * 8], q);
        if (own) {
          const int Y = S * (8 * gw.wy + ky - 1) + py, X = S * (8 * gw.wx + kx - 1) + px;
          Frag8<T> xq;
#pragma unroll
          for (int c = 0; c < 8; ++c) xq.set(c, q[c]);
          f2_nt_store(reinterpret_cast<bf16x8*>(pa.xin + (((long long)gw.b * H + Y) * W + X) * 16 + half * 8), xq.v);
        }
      }
    }
    if (tid < 160) *reinterpret_cast<f32x4*>(&RelC[tid >> 4][(tid & 15) * 4]) = rf;
    if (tid < TW * 32) store8(&Xs[tid >> 5][ZR][(tid & 31) * 8], frag_zero<T>());
    if (tid >= 64 && tid < 64 + TW * 9) store8(&Kc[(tid - 64) / 9][ZR][((tid - 64) % 9) * 8], frag_zero<T>());
    lds_barrier();
    M2T_FWD2_STAMP(1);
    // DWT^2 in place, four channels per item: element (pixel p, channel c) and element (band p, channel c) share an address
    for (int ii = tid; ii < TW * 112 * 4; ii += NTHR) {
      const int win = ii / (112 * 4), i = ii - win * (112 * 4);
      const int row = 16 * (i >> 6) + (i & 15), cg = (i >> 4) & 3;
      if (row >= 100) continue;
      bf16x4 raw[NPX];
#pragma unroll
      for (int n = 0; n < NPX; ++n) raw[n] = *reinterpret_cast<const bf16x4*>(&Xs[win][row][n * 16 + 4 * cg]);
      f32x2 o[2][NPX];
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        f32x2 v[S][S];
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
          for (int xx = 0; xx < S; ++xx) v[y][xx] = (f32x2){(float)raw[y * S + xx][2 * c2], (float)raw[y * S + xx][2 * c2 + 1]};
        Haar2<L>::fwd(v, o[c2]);
      }
#pragma unroll
      for (int n = 0; n < NPX; ++n) {
        const bf16x4 ov = {(bf16_t)o[0][n][0], (bf16_t)o[0][n][1], (bf16_t)o[1][n][0], (bf16_t)o[1][n][1]};
        *reinterpret_cast<bf16x4*>(&Xs[win][row][n * 16 + 4 * cg]) = ov;
      }
    }
  }
  // the weight ring: k | v | q fragments of stream step n = 8 c + ks (chunk c, k-step ks), RD steps ahead.  (Requested only here: in
  // front of phase 0 its registers sat on top of the 104 of the block loads; the round trip it exposes once per workgroup is small)
  constexpr int RD = 4;
  Frag8<T> wr[RD][3];
  auto ring_load = [&](int n, Frag8<T> (&slot)[3]) {                    // n compile-time after unrolling
    const int c = n >> 3, ks = n & 7;
    slot[0] = wload(1, c, ks); slot[1] = wload(2, c, ks); slot[2] = wload(0, c, ks);
  };
#pragma unroll
  for (int n = 0; n < RD; ++n) ring_load(n, wr[n]);
  lds_barrier();
  M2T_FWD2_STAMP(2);

  // ======== phase A: output-channel chunks: projection (both windows per weight fragment), q | k | v out, S^T += K^_s q_s^T ========
  f32x4 s[WA_KT];                                                      // S^T of (window mywin, query tile qt): lane (query lr, g) holds keys 16 t + 4 g + r
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS read addresses as lane-dependent bases (rows lr + 16 t, and the clamped row of key tile 6) plus compile-time offsets;
  // re-laundered per chunk -- hipcc otherwise keeps one address register per (tile, k-step) alive from chunk to chunk
  f2_lds_cptr xb = (f2_lds_cptr)&Xs[mywin][lr][8 * g];
  f2_lds_cptr xb6 = (f2_lds_cptr)&Xs[mywin][min(96 + lr, ZR)][8 * g];
  f2_lds_cptr kb = (f2_lds_cptr)&Kc[mywin][lr][8 * g];
  f2_lds_cptr kb6 = (f2_lds_cptr)&Kc[mywin][min(96 + lr, ZR)][8 * g];
  f2_lds_cptr qb_ = (f2_lds_cptr)&Qc[mywin][16 * qt + lr][8 * g];
  const int u4 = wv & 3;                                               // this wave's channel tile inside a chunk

  // x fragments of a k-step: all seven requested one k-step ahead (double-buffered by name): with two waves per SIMD nothing else
  // hides the LDS latency of a fragment that only feeds two or three MFMAs (measured: the loop ran at half the MFMA rate without)
  Frag8<T> xfa[WA_KT], xfb[WA_KT];
  auto x_fetch = [&](int ks, Frag8<T> (&xf)[WA_KT]) {
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) xf[t] = f2_lds_load8(t < 6 ? xb + 16 * t * LD + 32 * ks : xb6 + 32 * ks);
  };
  x_fetch(0, xfa);
#pragma unroll
  for (int cc = 0; cc < NCH; ++cc) {
    f32x4 ak[WA_KT], av[WA_KT], aq[4];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) { ak[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; av[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t) aq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int n = 8 * cc + ks;
      // the slot the previous k-step has consumed takes stream step n - 1 + RD (no copy of the fragments: twelve registers); the
      // scheduling barrier keeps the requests HERE, in front of this k-step's MFMAs (hipcc otherwise sinks loads to their use)
#ifndef F2_DBG_NO_WEIGHT_RELOAD
      if (n >= 1 && n - 1 + RD < 8 * NCH) ring_load(n - 1 + RD, wr[(n - 1) % RD]);
#endif
      if (n + 1 < 8 * NCH) { if (n & 1) x_fetch((ks + 1) & 7, xfa); else x_fetch((ks + 1) & 7, xfb); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        const Frag8<T>& b = (n & 1) ? xfb[t] : xfa[t];
        mma16(ak[t], wr[n % RD][0], b);
        mma16(av[t], wr[n % RD][1], b);
        if (t < 4) mma16(aq[t], wr[n % RD][2], b);                      // key tiles 0..3 = the query tiles (own pixels first)
      }
    }
    // ---- outputs of the chunk: lane (key 16 t + lr, g) holds channels 16 (4 cc + u4) + 4 g .. + 3 of the wave's window ----
    // (the epilogue's index values come from laundered copies of lr / g / tid: nothing computed here is kept alive until the next
    //  chunk's epilogue; global stores go through a scalar base + a 32-bit byte offset per lane -- the launcher checks < 4 GB)
    int lre = lr, ge = g, tide = tid;
    asm volatile("" : "+v"(lre));
    asm volatile("" : "+v"(ge));
    asm volatile("" : "+v"(tide));
    const int chl = 16 * u4 + 4 * ge;                                  // chunk-local channel
    const int ch = 64 * cc + chl;
    typedef char __attribute__((address_space(1))) * glb_bptr;
    const glb_bptr qkv_b = (glb_bptr)reinterpret_cast<char*>(pa.qkv);
    const glb_bptr vr_b = (glb_bptr)reinterpret_cast<char*>(pa.vring);
    // the next chunk's rel-pos values (stored behind this chunk's first barrier)
    f32x4 rf = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (cc + 1 < NCH) rf = *reinterpret_cast<const f32x4*>(rel_src(cc + 1, min(tide, 159)));
    // every wave is done with S^T of the previous chunk (its K^ / q buffers are free) and that chunk's RelC store is visible
    if (cc > 0) lds_barrier();
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = 16 * t + lre;
      if (key < WA_NK) {
        const bf16x4 kb4 = f2_pack4(ak[t]);
        const bf16x4 vb = f2_pack4(av[t]);
        int kr, kc;
        f2_key_rc(key, kr, kc);
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(&RelC[(cc < 2) ? kr : kc][chl]);
        float kh[4] = {(float)kb4[0] + r4[0], (float)kb4[1] + r4[1], (float)kb4[2] + r4[2], (float)kb4[3] + r4[3]};
        store4(&Kc[mywin][key][chl], kh);                              // K^ = bf16(bf16(k) + rel): the rounding points of the unfused kernels
        if (key < 64) {
          const unsigned qo = ((unsigned)gm.query_pixel(key) * (3 * C) + ch) * 2u;
          f2_nt_store(reinterpret_cast<bf16x4 __attribute__((address_space(1)))*>(qkv_b + qo + 2u * C), kb4);
          *reinterpret_cast<bf16x4 __attribute__((address_space(1)))*>(qkv_b + qo + 4u * C) = vb;
          const bf16x4 qb = f2_pack4(aq[t < 4 ? t : 0]);
          *reinterpret_cast<bf16x4*>(&Qc[mywin][key][chl]) = qb;
          f2_nt_store(reinterpret_cast<bf16x4 __attribute__((address_space(1)))*>(qkv_b + qo), qb);
        } else {
          const unsigned ro = (((unsigned)gm.wi * WA_RING + (unsigned)(key - 64)) * C + ch) * 2u;
          *reinterpret_cast<bf16x4 __attribute__((address_space(1)))*>(vr_b + ro) = vb;
        }
      }
    }
    lds_barrier();                                                     // K^_c, q_c complete; every wave is done with RelC
    if (cc + 1 < NCH && tide < 160) *reinterpret_cast<f32x4*>(&RelC[tide >> 4][(tide & 15) * 4]) = rf;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      const Frag8<T> qf = f2_lds_load8(qb_ + 32 * ks2);
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) mma16(s[t], f2_lds_load8(t < 6 ? kb + 16 * t * CLD + 32 * ks2 : kb6 + 32 * ks2), qf);
    }
  }
  M2T_FWD2_STAMP(3);
  // the d rows of the windows' own pixels (x rows 0..63, intact) leave as whole rows
  {
    constexpr int VEC = C / 8, DIT = TW * 64 * VEC / NTHR;
    static_assert(TW * 64 * VEC % NTHR == 0 && DIT % 4 == 0, "d rows: whole items per thread");
#pragma unroll
    for (int it = 0; it < DIT; it += 4) {
      Frag8<T> dr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + (it + u) * NTHR;
        const int win = idx / (64 * VEC), r = idx - win * (64 * VEC);
        dr[u] = load8(&Xs[win][r / VEC][(r % VEC) * 8]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + (it + u) * NTHR;
        const int win = idx / (64 * VEC), r = idx - win * (64 * VEC);
        f2_nt_store(reinterpret_cast<bf16x8*>(pa.d + f2_geom(h, w, wi0 + win).query_pixel(r / VEC) * C + (r % VEC) * 8), dr[u].v);
      }
    }
  }
  // every v row (and xin row) this workgroup wrote must be visible to its other waves: release, barrier, acquire
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  lds_barrier();                                                       // (also: every wave is done with the last K^ / q slice)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  M2T_FWD2_STAMP(4);

  // ======== phase B: v chunk 0 and the residual rows requested, softmax over the 100 keys in registers ========
  // (index values of the phases below are derived from a laundered tid: none of them is kept alive -- or spilled -- across phase A)
  asm volatile("" : "+v"(tid));
  lane = tid & 63; lr = lane & 15; g = lane >> 4;
  constexpr int VIT = (TW * 100 * 8 + NTHR - 1) / NTHR;                // 16-byte pieces of the v chunks [window][100 keys][64 channels]
  Frag8<T> vpre[VIT];
  auto v_fetch = [&](int vc) {
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = min(tid + it * NTHR, TW * 800 - 1);
      const int win = idx / 800, r = idx - win * 800;
      const int row = r >> 3, cv = r & 7;
      const WinGeom gw = f2_geom(h, w, wi0 + win);
      const T* own = pa.qkv + gw.query_pixel(min(row, 63)) * (3 * C) + 2 * C;
      const T* rng = pa.vring + ((long long)gw.wi * WA_RING + max(row - 64, 0)) * C;
      vpre[it] = load8((row < 64 ? own : rng) + 64 * vc + 8 * cv);
    }
  };
  auto v_stage = [&]() {
#pragma unroll
    for (int it = 0; it < VIT; ++it) {
      const int idx = tid + it * NTHR;
      const int win = min(idx / 800, TW - 1), r = idx - win * 800;
      if (idx < TW * 800) store8(&Kc[win][r >> 3][(r & 7) * 8], vpre[it]);
    }
  };
  v_fetch(0);
  const int q = 16 * qt + lr;                                           // this lane's query
  const int by = 8 * gm.wy + (q >> 3), bx = 8 * gm.wx + (q & 7);
  const int H = h * S, W = w * S;
  bf16x4 resv[S][S];
#pragma unroll
  for (int yy = 0; yy < S; ++yy)
#pragma unroll
    for (int xx = 0; xx < S; ++xx) {
      const long long pix = ((long long)gm.b * H + S * by + yy) * W + S * bx + xx;
      resv[yy][xx] = *reinterpret_cast<const bf16x4*>(pa.xin + pix * 16 + 4 * g);
    }
  __builtin_amdgcn_sched_barrier(0);
  Frag8<T> pfr[4];                                                      // P^T as the B operand: slot (g, j) of k-chunk c4 <-> key 16 (2 c4 + (j >> 2)) + 4 g + (j & 3)
  {
    const float scale = rsqrtf((float)C);
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < WA_KT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * g + r;
        s[t][r] = (key < WA_NK) ? s[t][r] * scale : -3.0e38f;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < WA_KT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * t + 4 * g + r;
        const float e = (key < WA_NK) ? __expf(s[t][r] - mx) : 0.f;
        s[t][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 2 * c4 + (j >> 2);
        pfr[c4].set(j, (t < WA_KT) ? s[t < WA_KT ? t : 0][j & 3] * inv : 0.f);
      }
  }
  M2T_FWD2_STAMP(5);

  // ======== phase C: O^T = V^T P^T, four v chunks through the staging buffers ========
  f32x4 o[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int vc = 0; vc < 4; ++vc) {
    v_stage();
    if (vc + 1 < 4) v_fetch(vc + 1);
    lds_barrier();
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int m = 0; m < 4; ++m)
        mma16(o[4 * vc + m], f2_tr8(&Kc[mywin][0][0], CLD, 32 * c4 + 4 * g, 32 * c4 + 16 + 4 * g, 16 * m, lane, ZR), pfr[c4]);
    if (vc + 1 < 4) lds_barrier();
  }
  M2T_FWD2_STAMP(6);

  // ======== phase D: IWT^2 + residual, straight from the accumulators: tile mt = band mt, row 4 g + r = base channel ========
  {
    float vv[4][S][S];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bands[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) bands[mt] = o[mt][r];
      Haar<L>::inv(bands, vv[r]);
    }
#pragma unroll
    for (int yy = 0; yy < S; ++yy)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) {
        const long long pix = ((long long)gm.b * H + S * by + yy) * W + S * bx + xx;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = vv[c][yy][xx] + (float)resv[yy][xx][c];
        const bf16x4 ov = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        f2_nt_store(reinterpret_cast<bf16x4*>(pa.out + pix * 16 + 4 * g), ov);
      }
  }
  M2T_FWD2_STAMP(7);
}

}  // namespace

size_t window_attn_fwd2_vring_elems(int B, int h, int w) { return (size_t)B * (h / 8) * (w / 8) * WA_RING * F2_C; }

// bf16, C = 256, two DWT levels, k >= 1: the arguments of launch_window_attn_fused_prep_fwd plus the ring scratch
// vring [B (h/8)(w/8)][36][256].  windows_per_wg: 1 (4 waves, two workgroups per CU) or 2 (8 waves, one workgroup per CU, the weight
// stream shared by the two windows; needs an even number of windows).
int launch_window_attn_fused_prep_fwd2(const void* xn, const void* xprev, const float* mean, const float* rstd, int k, void* xin, void* d,
                                       const void* wfrag, const float* rel_h, const float* rel_w, void* qkv, void* out, void* vring,
                                       int B, int h, int w, int windows_per_wg, hipStream_t st, int stagger) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn_fwd2: h,w must be multiples of 8");
  if (!xn || !xprev || !mean || !rstd || !xin || !d || !wfrag || !qkv || !out || !vring || k < 1 || k > 3)
    return m2t_set_error(-2, "window_attn_fwd2: null argument or k outside 1..3");
  Fwd2Args pa;
  pa.xn = (const bf16_t*)xn; pa.xprev = (const bf16_t*)xprev; pa.mean = mean; pa.rstd = rstd; pa.xin = (bf16_t*)xin; pa.d = (bf16_t*)d;
  pa.wfrag = (const bf16_t*)wfrag; pa.rel_h = rel_h; pa.rel_w = rel_w; pa.qkv = (bf16_t*)qkv; pa.out = (bf16_t*)out; pa.vring = (bf16_t*)vring;
  pa.k = k; pa.h = h; pa.w = w; pa.stagger = stagger;
  const int nwin = B * (h / 8) * (w / 8);
  if ((long long)B * h * w * 3 * F2_C * 2 >= (1LL << 32)) return m2t_set_error(-2, "window_attn_fwd2: qkv tensor beyond 32-bit byte offsets");
  if (windows_per_wg != 1 && windows_per_wg != 2) return m2t_set_error(-2, "window_attn_fwd2: windows_per_wg must be 1 or 2");
  if (windows_per_wg == 2 && (((h / 8) * (w / 8)) & 1)) return m2t_set_error(-2, "window_attn_fwd2: two windows per workgroup need an even number of windows per image");
  M2TProfScope ps(M2T_PROF_ATTN_FUSED_256, st);
  if (windows_per_wg == 1) {
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_fwd2_kernel<1>, (int)F2Cfg<1>::total)) return rc__;
    M2T_LAUNCH_TIMED(window_attn_fwd2_kernel<1>, dim3(nwin), dim3(256), F2Cfg<1>::total, st, pa);
  } else {
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_fwd2_kernel<2>, (int)F2Cfg<2>::total)) return rc__;
    M2T_LAUNCH_TIMED(window_attn_fwd2_kernel<2>, dim3(nwin / 2), dim3(512), F2Cfg<2>::total, st, pa);
  }
  M2T_LAUNCH_CHECK();
  return 0;
}
