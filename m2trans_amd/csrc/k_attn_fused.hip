// k_attn_fused.hip -- TBlock.forward in ONE kernel per window (bf16, C = 64 / 256): the 1x1 qkv projection
// (models/M2Trans_network.py:281,307-308), the 8x8 / 10x10 halo window attention with the relative-position key bias
// (:310-332) and the branch epilogue xc = IWT^L(attention) + xin (:145,153,161).  q | k | v of the window's own 64
// pixels are still written out (the backward kernels recompute P from them), but they are never re-read here: the
// 3C-wide round trip through HBM between `gemm_nt_wide_kernel` and the attention kernel (50 MB per launch at C = 256,
// batch 16) and the second launch are gone.
//
// One workgroup per window: at the benchmarked size the C = 256 branches have 256 windows -- one per CU.  The K / V rows
// of the 36 halo pixels are projected redundantly (x1.56 on the K and V thirds: 100 instead of 64 rows).
//
// What bounds it (measured, scratch/bench_fused.hip): every workgroup needs the whole 3C x C weight (393 KB at C = 256)
// and all 256 CUs read it from L2 at once.  That broadcast runs at ~20-35 B/clk per CU and is latency-bound: its rate is
// (bytes in flight) / (~2.5 k cycles), so the weight stream alone costs as much as the kernel's 11 136 MFMA cycles per
// SIMD, and whatever is not overlapped with MFMAs is added to them.  So the projection is OUTPUT-STATIONARY and the
// contraction index is the OUTER loop: wave w owns output channels [C/NW w, C/NW (w+1)) of k and v for ALL 112 key rows
// in its accumulators (112 registers at C = 256, NW = 8), the weight fragments of k-step ks (M2T_PACK_FRAG16 order: one
// contiguous 1 KB per wave load, each used exactly once per workgroup) are consumed as they arrive with the next THREE
// k-steps in flight (96 KB per CU), and x comes from LDS.  q (64 rows only) follows in a second k-step loop whose
// fragments are fetched under the tail of the first.
// Tried and measured slower: weight slices resident in registers with the pixel tiles as the outer loop (the first
// 256 KB had to land before the first MFMA: 26.4 us); four waves with all of q | k | v in 288 accumulators per wave (one
// wave per SIMD cannot hide the ~1 200 VALU instructions of its epilogues, and one k-step in flight made the loop
// latency-bound: 28.6 us; kept as scratch/k_attn_fused_4wave_output_stationary.hip.txt).
//
//   phase 0  x rows of the 100 key pixels and the rel-pos table -> LDS.  Keys are ordered own-pixels-first, so key j =
//            query j for j < 64 and the first four 16-row tiles serve the q projection too.  Out-of-image keys are zero
//            rows: their projection is the zero padding F.unfold applies to k and v (:313).
//   phase 1  k-step loops: [K | V]^T tiles += W_ks X_ks^T, then Q^T likewise.
//   phase 2  k -> HBM (raw) and, + rel-pos, -> LDS Kh (bf16(bf16(k) + rel): the rounding points of the unfused kernels);
//            q -> LDS Qs and HBM; barrier; v -> LDS (over x, dead now) and HBM.
//   phase 3  S^T = K^ Q^T, softmax over the 100 keys in registers; with 8 waves the keys are split in two halves per
//            query tile and the statistics merged through LDS; P -> LDS (bf16).
//   phase 4  O^T = V^T P^T for the wave's channels x all 64 queries -> LDS (fp32, over Kh | Qs).
//   phase 5  epilogue: IWT^L over the 4^L bands of each base channel + residual, stored to the full-resolution chunk of xc.
// LDS (C = 256): x|v 53 328 + Kh 53 328 + Qs 33 792 + P 17 408 + softmax statistics 1 024 = 158 880 B.
//
// Three rules shape the memory side (each cost a factor of ~2 on its phase when broken, measured):
//  * no load under a lane-dependent branch: hipcc then waits vmcnt(0) per load (serial HBM round trips), so out-of-image
//    keys read a clamped, valid address and are zeroed by a select, and surplus slots rewrite the zero row;
//  * vmcnt retires in order and counts STORES too: a load issued after a global store cannot complete before that store
//    is acknowledged (~2 k cycles under load).  Hence the rel-pos table comes through LDS instead of per-tile loads, and
//    the residual rows of the epilogue are fetched before the first q | k | v store;
//  * barriers order LDS only (lds_barrier): __syncthreads() also drains the outstanding global stores.
#include "m2t_kernels.h"
#include "m2t_haar.h"
#include "m2t_window.h"

#ifndef M2T_FUSED_STAMP
#define M2T_FUSED_STAMP(i) do { } while (0)      // scratch/bench_fused.hip defines it to record s_memtime per phase
#endif

namespace {

// key order of this kernel: 0..63 the window's own pixels (= query order), 64..99 the ring in ring_index order.
// Branch-free (selects): the callers issue loads whose addresses depend on the result.
__device__ __forceinline__ void fused_key_rc(int k, int& kr, int& kc) {
  const int r = k - 64;
  const int kr_ring = (r < 10) ? 0 : ((r < 20) ? 9 : ((r < 28) ? r - 19 : r - 27));
  const int kc_ring = (r < 10) ? r : ((r < 20) ? r - 10 : ((r < 28) ? 0 : 9));
  kr = (k < 64) ? (k >> 3) + 1 : kr_ring;
  kc = (k < 64) ? (k & 7) + 1 : kc_ring;
}

template <int C> struct FusedCfg {
  static constexpr int LD = C + 8;          // bf16 rows of Xs|Vs / Kh / Qs
  // P [query][key].  C = 256 (8 waves, two key halves, one workgroup per CU): keys 0..127 in a region of its own, which the rel-pos table
  // borrows in phases 0-2.  C = 64 (4 waves, 1 024 windows at batch 16): the footprint decides how many windows a CU works on at once -- P has
  // 112 keys (the 16 padding keys of the last k-step are read from the zero row of Xs | Vs) and OVERLAYS Kh | Qs (dead once every wave has its
  // S^T tile: one more barrier; O overlays them too: a second one), the rel-pos table gets 2.5 KB of its own, the cross-half reduction buffer
  // is gone: 40 864 bytes, FOUR workgroups per CU instead of two = all 1 024 windows of a batch-16 launch resident in one round (round 6)
  static constexpr bool P_OVER_K = (C == 64);
  static constexpr int PKEYS = P_OVER_K ? 112 : 128;
  static constexpr int PLD = PKEYS + 8;
  static constexpr int OLD = C + 4;         // O [query][channel] fp32
  static constexpr int ZR = 100;            // the zero row of Xs|Vs and Kh (every padding key aliases it)
  static constexpr size_t szX = sizeof(bf16_t) * 101 * LD;
  static constexpr size_t szQ = sizeof(bf16_t) * 64 * LD;
  static constexpr size_t szP = sizeof(bf16_t) * 64 * PLD;
  static constexpr size_t szR = P_OVER_K ? 0 : sizeof(float) * 2 * 2 * 64;    // [max | sum][key half][query]; one key half at C = 64
  static constexpr size_t szRel = sizeof(float) * 10 * C;
  static constexpr size_t offX = 0, offK = szX, offQ = 2 * szX;
  static constexpr size_t offP = P_OVER_K ? offK : 2 * szX + szQ;
  static constexpr size_t offRel = P_OVER_K ? 2 * szX + szQ : offP;
  static constexpr size_t offR = P_OVER_K ? offRel + szRel : offP + szP;
  static constexpr size_t total = offR + szR;
  static_assert(!P_OVER_K || szP <= szX + szQ, "P overlays Kh | Qs");
  static_assert(sizeof(float) * 64 * OLD <= szX + szQ, "the fp32 O tile overlays Kh | Qs");
  static_assert(P_OVER_K || szRel <= szP, "the rel-pos table borrows P's region");
  static_assert(szX % 16 == 0 && szQ % 16 == 0 && szP % 16 == 0, "16-byte carve offsets");
};

__device__ __forceinline__ Frag8<bf16_t> tr8z(const bf16_t* base, int ld, int row_lo_base, int row_hi_base, int col0, int lane,
                                              int zero_row) {
  const int i = lane & 15, qq = i >> 2, pp = i & 3;
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const int rlo = min(row_lo_base + qq, zero_row), rhi = min(row_hi_base + qq, zero_row);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rlo * ld + col0 + 4 * pp));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rhi * ld + col0 + 4 * pp));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}

__device__ __forceinline__ bf16x4 pack4(const f32x4& a) {
  bf16x4 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
  return r;
}

// PREP: branch_prep of this branch (models/M2Trans_network.py:141-158: xin = (norm(x)[chunk k] + xc[chunk k - 1]) / 2, d = DWT^L(xin))
// runs inside phase 0 instead of in its own launch in front of this kernel: the window computes xin for the (2^L)^2 pixel blocks
// of its 100 keys (halo blocks redundantly, like their projections) from the two P64 planes -- 16-byte loads that run along
// image rows, 10 * 2^L * 32 contiguous bytes per (key row, pixel row) --, rounds it to bf16 exactly where branch_prep_tiled_kernel
// stores it, transforms every (key, channel) in place in LDS with the same Haar<L>::fwd, and writes xin (the residual of its own
// epilogue) and the d rows (the backward's operand) of its OWN pixels only.  Same operations in the same order: identical bits.
struct FusedPrepArgs {
  const bf16_t* xn;        // plane k of the block input X  [B][H][W][16]
  const bf16_t* xprev;     // plane k - 1 of the concat buffer xc
  const float* mean;       // [B][64] / [B][64] statistics of X
  const float* rstd;
  bf16_t* xin;             // [B][H][W][16] (written: own pixels)
  bf16_t* d;               // [B][h][w][C]  (written: own pixels)
  int k;
};
template <int C, int L, int NW, bool PREP = false>
__global__ void __launch_bounds__(NW * 64, (C == 64) ? 4 : 1) window_attn_fused_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wfrag,
                                                                        const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                                        bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int ldo, int oc0,
                                                                        const bf16_t* __restrict__ res, int ldr, int h, int w,
                                                                        FusedPrepArgs pa) {
  using T = bf16_t;
  using Cfg = FusedCfg<C>;
  constexpr int LD = Cfg::LD, PLD = Cfg::PLD, OLD = Cfg::OLD, ZR = Cfg::ZR;
  constexpr int NTHR = NW * 64, VEC = C / 8, NT = C / 16, NKS = C / 32, TPW = NT / NW, KH = NW / 4;
  constexpr int DEPTH = (NKS < 3) ? NKS : 3;                  // k-steps of weight fragments in flight
  static_assert(NT % NW == 0 && TPW >= 1, "channel tiles must split evenly over the waves");
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(L == 0 || C == (16 << (2 * L)), "fused IWT needs C = 16 * 4^L");
  static_assert(Cfg::total <= 163840, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Xs)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offX);                  // x rows; from phase 2 on: v rows
  T(*Kh)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offK);
  T(*Qs)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offQ);
  T(*Ps)[PLD] = reinterpret_cast<T(*)[PLD]>(smem + Cfg::offP);
  float(*RelS)[C] = reinterpret_cast<float(*)[C]>(smem + Cfg::offRel);        // phases 0-2 only (C = 256: P's region)
  float(*Os)[OLD] = reinterpret_cast<float(*)[OLD]>(smem + Cfg::offK);        // phase 4 on: overlays Kh | Qs
  float(*red)[2][64] = reinterpret_cast<float(*)[2][64]>(smem + Cfg::offR);   // [max | sum][key half][query]

  // (not const: with PREP they are re-derived from a laundered copy of tid behind phase 2 -- at 256 VGPRs every index value that is
  // live ACROSS the projection but not used in it is spilled to scratch (hipcc does not rematerialise `tid >> 2`), and its reload in the
  // softmax phase then waits, vmcnt(0), for the q | k | v stores in flight)
  int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const bf16x8* wf8 = reinterpret_cast<const bf16x8*>(wfrag);
  // fragment (projection p, channel tile ct of it, k-step ks): one contiguous 1 KB per wave
  auto wload = [&](int p, int ct, int ks) -> Frag8<T> {
    Frag8<T> f;
    f.v = wf8[((size_t)((p * NT + ct) * NKS + ks)) * 64 + lane];
    return f;
  };
  struct WKV { Frag8<T> k[TPW], v[TPW]; };
  auto wkv = [&](int ks) -> WKV {
    WKV s;
#pragma unroll
    for (int m = 0; m < TPW; ++m) { s.k[m] = wload(1, wv * TPW + m, ks); s.v[m] = wload(2, wv * TPW + m, ks); }
    return s;
  };

  M2T_FUSED_STAMP(0);
  // ---- phase 0: x rows of the 100 keys (own pixels first) and the rel-pos table -> LDS ----
  constexpr int XIT = (101 * VEC + NTHR - 1) / NTHR;          // row 100 = the zero row: written like any other row
  constexpr int RIT = (10 * C / 4 + NTHR - 1) / NTHR;         // rel table [10][C] fp32 as float4s
  WKV wbuf[DEPTH + 1];
  if constexpr (PREP) {
    static_assert(L >= 1, "branch_prep in front of an L = 0 branch is the identity on d");
    constexpr int S = 1 << L, NPX = S * S;
    constexpr int NITEM = 100 * NPX * 2, PIT = (NITEM + NTHR - 1) / NTHR;      // 16-byte items: [key row][pixel row][key col][pixel col][half]
    static_assert(NTHR % 2 == 0, "a thread's channel half is the same in every item");
    const int half = tid & 1;
    float mu[8], rs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { mu[c] = pa.mean[gm.b * 64 + pa.k * 16 + half * 8 + c]; rs[c] = pa.rstd[gm.b * 64 + pa.k * 16 + half * 8 + c]; }
    const int H = h * S, W = w * S;
    Frag8<T> xf[PIT], pf[PIT];
    bool ok[PIT];
    f32x4 rf[RIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int i = min(tid + it * NTHR, NITEM - 1);
      const int px = (i >> 1) % S, kx = (i / (2 * S)) % 10, py = (i / (20 * S)) % S, ky = i / (20 * NPX);
      const int yy = 8 * gm.wy + ky - 1, xx = 8 * gm.wx + kx - 1;
      ok[it] = yy >= 0 && yy < h && xx >= 0 && xx < w;
      const int Y = S * min(max(yy, 0), h - 1) + py, X = S * min(max(xx, 0), w - 1) + px;     // clamped: the loads are unconditional
      const long long o = (((long long)gm.b * H + Y) * W + X) * 16 + half * 8;
      xf[it] = load8(pa.xn + o);
      pf[it] = load8(pa.xprev + o);
    }
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
      const int kk = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
      const float* rp = (c4 < C / 2) ? (rel_h + kk * (C / 2) + c4) : (rel_w + kk * (C / 2) + (c4 - C / 2));
      rf[it] = *reinterpret_cast<const f32x4*>(rp);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wbuf[d] = wkv(d);
    // xin -> the key's LDS row as [pixel][16 channels] (and, for the window's own pixels, -> HBM: the epilogue's residual)
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
      const int i = tid + it * NTHR;
      const int ic = min(i, NITEM - 1);
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
      if (i < NITEM) {
        store8f(&Xs[row][(py * S + px) * 16 + half * 8], q);
        if (own) {
          const int Y = S * (8 * gm.wy + ky - 1) + py, X = S * (8 * gm.wx + kx - 1) + px;
          store8f(pa.xin + (((long long)gm.b * H + Y) * W + X) * 16 + half * 8, q);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
      *reinterpret_cast<f32x4*>(&RelS[idx / (C / 4)][(idx % (C / 4)) * 4]) = rf[it];
    }
    if (tid < VEC) { store8(&Kh[ZR][tid * 8], frag_zero<T>()); store8(&Xs[ZR][tid * 8], frag_zero<T>()); }
    M2T_FUSED_STAMP(10);
    lds_barrier();
    M2T_FUSED_STAMP(11);
    // DWT^L in place, four channels per item (8-byte LDS accesses; two-byte ones cost the same issue slot each and made this
    // loop 5 k cycles): element (pixel p, channel c) and element (band p, channel c) share an address
    // lane -> (row, channel group) so that the 32 lanes of a half-wave cover all 64 banks once: rows are 132 words apart
    // (4 banks mod 64), a lane reads 2 words, so 16 rows x 2 channel groups = 64 distinct banks
    for (int i = tid; i < 112 * 4; i += NTHR) {
      const int row = 16 * (i >> 6) + (i & 15), cg = (i >> 4) & 3;
      if (row >= 100) continue;
      bf16x4 raw[NPX];
#pragma unroll
      for (int n = 0; n < NPX; ++n) raw[n] = *reinterpret_cast<const bf16x4*>(&Xs[row][n * 16 + 4 * cg]);
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
        *reinterpret_cast<bf16x4*>(&Xs[row][n * 16 + 4 * cg]) = ov;
      }
    }
  } else {
    Frag8<T> xf[XIT];
    bool ok[XIT];
    f32x4 rf[RIT];
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = tid + it * NTHR;
      const int key = min(idx / VEC, 100), cv = idx % VEC;
      int kr, kc;
      fused_key_rc(min(key, 99), kr, kc);
      const int yy = 8 * gm.wy + kr - 1, xx = 8 * gm.wx + kc - 1;
      ok[it] = (key < 100) && yy >= 0 && yy < h && xx >= 0 && xx < w;
      const int yc = min(max(yy, 0), h - 1), xc = min(max(xx, 0), w - 1);
      xf[it] = load8(x + (((long long)gm.b * h + yc) * w + xc) * C + cv * 8);
    }
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      // T[kk][ch] = ch < C/2 ? rel_h[kk][ch] : rel_w[kk][ch - C/2]   (row kk is a key ROW for the first half of the
      // channels and a key COLUMN for the second, models/M2Trans_network.py:322-325)
      const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
      const int kk = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
      const float* rp = (c4 < C / 2) ? (rel_h + kk * (C / 2) + c4) : (rel_w + kk * (C / 2) + (c4 - C / 2));
      rf[it] = *reinterpret_cast<const f32x4*>(rp);
    }
    // the first DEPTH k-steps of weight fragments: behind x in the (in-order) return queue
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) wbuf[d] = wkv(d);
#pragma unroll
    for (int it = 0; it < XIT; ++it) {
      const int idx = tid + it * NTHR;
      const int key = min(idx / VEC, 100), cv = idx % VEC;
      store8(&Xs[key][cv * 8], ok[it] ? xf[it] : frag_zero<T>());
    }
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
      *reinterpret_cast<f32x4*>(&RelS[idx / (C / 4)][(idx % (C / 4)) * 4]) = rf[it];
    }
    if (tid < VEC) store8(&Kh[ZR][tid * 8], frag_zero<T>());
  }
  lds_barrier();
  M2T_FUSED_STAMP(1);

  // ---- phase 1a: [K | V]^T, k-step outer.  lane (pixel = lr, g) of tile t ends with channels 16 ct + 4 g .. + 3 ----
  f32x4 ak[WA_KT][TPW], av[WA_KT][TPW];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t)
#pragma unroll
    for (int m = 0; m < TPW; ++m) { ak[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; av[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  Frag8<T> wqf[NKS][TPW];               // the q fragments: fetched under the last k-steps of this loop
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + DEPTH < NKS) wbuf[(ks + DEPTH) % (DEPTH + 1)] = wkv(ks + DEPTH);
    else {
      // no k | v fragments left to fetch: the q fragments take their place in the stream
      constexpr int QPER = (NKS + DEPTH - 1) / DEPTH;
#pragma unroll
      for (int j = 0; j < QPER; ++j) {
        const int qs = (ks + DEPTH - NKS) * QPER + j;
        if (qs < NKS) {
#pragma unroll
          for (int m = 0; m < TPW; ++m) wqf[qs][m] = wload(0, wv * TPW + m, qs);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const WKV& wc = wbuf[ks % (DEPTH + 1)];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const Frag8<T> b = load8(&Xs[min(16 * t + lr, ZR)][32 * ks + 8 * g]);
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        mma16(ak[t][m], wc.k[m], b);
        mma16(av[t][m], wc.v[m], b);
      }
    }
  }
  // ---- phase 1b: Q^T (key tiles 0..3 = the query tiles) ----
  f32x4 aq[4][TPW];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < TPW; ++m) aq[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const Frag8<T> b = load8(&Xs[16 * t + lr][32 * ks + 8 * g]);
#pragma unroll
      for (int m = 0; m < TPW; ++m) mma16(aq[t][m], wqf[ks][m], b);
    }
  M2T_FUSED_STAMP(2);
  if constexpr (PREP) asm volatile("" : "+v"(lr));     // phase 2's key indices 16 t + lr: new values, not phase 1's row indices kept alive
  // the residual rows of the epilogue.  Without PREP they are fetched here, before the first global store (vmcnt retires in order).
  // With PREP they are this kernel's own xin stores of phase 0, written by other waves: every wave's stores have retired when it has
  // consumed its last weight fragment (in-order vmcnt; the release fence below states it), so the loads are issued behind the
  // NEXT barrier instead (a barrier here would make the waves that finish the projection early wait for the late ones before they
  // start their stores: +4 k cycles, measured).  They are consumed 10 k cycles later, when the q | k stores in front of them have long
  // been acknowledged.
  constexpr int ES = Haar<L>::S, EPARTS = (L == 0) ? 1 : NTHR / 256, EROWS = (L == 0) ? 1 : ES / EPARTS;
  bf16x4 resv[EROWS][(L == 0) ? 1 : ES];
  int e_item = tid & 255, e_part = tid >> 8;
  int e_q = e_item >> 2, e_cg = e_item & 3;
  auto load_residual = [&]() {
    if constexpr (L > 0) {
      const int H = h * ES, W = w * ES;
      const int by = 8 * gm.wy + (e_q >> 3), bx = 8 * gm.wx + (e_q & 7);
#pragma unroll
      for (int yy = 0; yy < EROWS; ++yy)
#pragma unroll
        for (int xx = 0; xx < ES; ++xx) {
          const long long pix = ((long long)gm.b * H + ES * by + e_part * EROWS + yy) * W + ES * bx + xx;
          resv[yy][xx] = *reinterpret_cast<const bf16x4*>(res + pix * ldr + 4 * e_cg);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (!PREP) load_residual();
  else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if constexpr (PREP) {
    // the d rows of the window's own pixels (LDS rows 0..63, intact until v overwrites them behind the next barrier) leave as whole rows
    constexpr int DIT = (64 * VEC + NTHR - 1) / NTHR;
#pragma unroll
    for (int it = 0; it < DIT; it += 2) {
      Frag8<T> dr[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = min(tid + (it + u) * NTHR, 64 * VEC - 1);
        dr[u] = load8(&Xs[idx / VEC][(idx % VEC) * 8]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = tid + (it + u) * NTHR;
        if (it + u < DIT && idx < 64 * VEC) store8(pa.d + gm.query_pixel(idx / VEC) * C + (idx % VEC) * 8, dr[u]);
      }
    }
  }

  // ---- phase 2: accumulators -> LDS / HBM ----
  // (q | k | v leave straight from the accumulators, 8 bytes per lane.  Staging them through LDS to store whole rows,
  // 16 bytes per lane, was measured SLOWER -- 27.1 vs 25.5 us: the 130 KB a workgroup writes are bound by bytes, ~10 B/clk
  // per CU = the chip's HBM write rate / 256, not by store instructions; scratch/k_attn_fused_row_stores.hip.txt)
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = 16 * t + lr;
    if (key < WA_NK) {
      int kr, kc;
      fused_key_rc(key, kr, kc);
      const long long qp = (key < 64) ? gm.query_pixel(key) : 0;
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        const int ch = 16 * (wv * TPW + m) + 4 * g;
        const bf16x4 kb = pack4(ak[t][m]);
        if (qkv && key < 64) *reinterpret_cast<bf16x4*>(qkv + qp * (3 * C) + C + ch) = kb;     // (qkv == nullptr: the backward recomputes q | k | v)
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(&RelS[(ch < C / 2) ? kr : kc][ch]);
        float kh[4] = {(float)kb[0] + r4[0], (float)kb[1] + r4[1], (float)kb[2] + r4[2], (float)kb[3] + r4[3]};
        store4(&Kh[key][ch], kh);
        if (t < 4) {
          const bf16x4 qb = pack4(aq[t][m]);
          *reinterpret_cast<bf16x4*>(&Qs[key][ch]) = qb;
          if (qkv) *reinterpret_cast<bf16x4*>(qkv + qp * (3 * C) + ch) = qb;
        }
      }
    }
  }
  lds_barrier();                        // every wave is done reading x and the rel-pos table; Kh and Qs are complete
  M2T_FUSED_STAMP(3);
  if constexpr (PREP) {
    asm volatile("" : "+v"(tid));        // a new value for the register allocator: what follows extends no live range across phase 1
    lane = tid & 63; wv = tid >> 6; lr = lane & 15; g = lane >> 4;
    e_item = tid & 255; e_part = tid >> 8; e_q = e_item >> 2; e_cg = e_item & 3;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    load_residual();
  }
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = 16 * t + lr;
    if (key < WA_NK) {
      const long long qp = (key < 64) ? gm.query_pixel(key) : 0;
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        const int ch = 16 * (wv * TPW + m) + 4 * g;
        const bf16x4 vb = pack4(av[t][m]);
        *reinterpret_cast<bf16x4*>(&Xs[key][ch]) = vb;        // row 100 of Xs stays the zero row
        if (qkv && key < 64) *reinterpret_cast<bf16x4*>(qkv + qp * (3 * C) + 2 * C + ch) = vb;
      }
    }
  }

  // ---- phase 3: S^T = K^ Q^T for (query tile qt, key tiles t0 .. t0 + NTL - 1), softmax, P -> LDS ----
  {
    const int qt = wv & 3, kh = wv >> 2;
    constexpr int NTL = (KH == 2) ? 4 : WA_KT;
    const int t0 = kh * 4;
    const int q = 16 * qt + lr;
    f32x4 s[NTL];
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) s[tl] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const Frag8<T> qf = load8(&Qs[q][32 * ks + 8 * g]);
#pragma unroll
      for (int tl = 0; tl < NTL; ++tl)       // (key tile 7 of the second half does not exist: it aliases the zero row and is masked)
        mma16(s[tl], load8(&Kh[min(16 * (t0 + tl) + lr, ZR)][32 * ks + 8 * g]), qf);
    }
    if constexpr (Cfg::P_OVER_K) lds_barrier();        // every wave has its S^T tile: Kh and Qs are dead, P may overwrite them
    // lane (q, g) holds keys 16 (t0 + tl) + 4 g + r
    const float scale = rsqrtf((float)C);
    float mx = -3.0e38f;
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * (t0 + tl) + 4 * g + r;
        s[tl][r] = (key < WA_NK) ? s[tl][r] * scale : -3.0e38f;
        mx = fmaxf(mx, s[tl][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * (t0 + tl) + 4 * g + r;
        const float e = (key < WA_NK) ? __expf(s[tl][r] - mx) : 0.f;
        s[tl][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    float inv;
    if constexpr (KH == 2) {
      if (g == 0) { red[0][kh][q] = mx; red[1][kh][q] = sum; }
      lds_barrier();
      const float m0 = red[0][0][q], m1 = red[0][1][q];
      const float mm = fmaxf(m0, m1);
      const float f0 = __expf(m0 - mm), f1 = __expf(m1 - mm);
      inv = (kh ? f1 : f0) / (red[1][0][q] * f0 + red[1][1][q] * f1);
    } else {
      inv = 1.0f / sum;
    }
    // (the rel-pos table in P's region is dead: its last reader was phase 2, before the barrier that follows it)
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
      float pv[4] = {s[tl][0] * inv, s[tl][1] * inv, s[tl][2] * inv, s[tl][3] * inv};
      store4(&Ps[q][16 * (t0 + tl) + 4 * g], pv);          // (KH == 2: tile 7 = keys 112..127 gets exact zeros)
    }
    if constexpr (KH == 1 && Cfg::PKEYS == 128) {          // (not instantiated any more: C = 64 reads its padding keys from the zero row)
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      store4(&Ps[q][112 + 4 * g], z);                      // keys 112..127: contraction padding of P V
    }
  }
  M2T_FUSED_STAMP(4);
  lds_barrier();                        // P and v are visible; Kh and Qs are dead

  // ---- phase 4: O^T = V^T P^T for this wave's channel tiles, all 64 queries ----
  {
    f32x4 o[TPW][4];
#pragma unroll
    for (int m = 0; m < TPW; ++m)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) o[m][qt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      Frag8<T> a[TPW];
#pragma unroll
      for (int m = 0; m < TPW; ++m) a[m] = tr8z(&Xs[0][0], LD, 32 * c4 + 8 * g, 32 * c4 + 8 * g + 4, 16 * (wv * TPW + m), lane, ZR);
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        // (P rows of 112 keys: the contraction padding 112..127 of the last k-step is read from the zero row of Xs | Vs instead)
        const Frag8<T> b = load8((Cfg::PKEYS == 128 || 32 * c4 + 8 * g < Cfg::PKEYS) ? &Ps[16 * qt + lr][32 * c4 + 8 * g] : &Xs[ZR][0]);
#pragma unroll
        for (int m = 0; m < TPW; ++m) mma16(o[m][qt], a[m], b);
      }
    }
    if constexpr (Cfg::P_OVER_K) lds_barrier();        // every wave is done reading P: O may overwrite it
#pragma unroll
    for (int m = 0; m < TPW; ++m)
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) *reinterpret_cast<f32x4*>(&Os[16 * qt + lr][16 * (wv * TPW + m) + 4 * g]) = o[m][qt];
  }
  M2T_FUSED_STAMP(5);
  lds_barrier();
  M2T_FUSED_STAMP(6);

  // ---- phase 5: epilogue ----
  if constexpr (L == 0) {
    for (int idx = tid; idx < 64 * VEC; idx += NTHR) {
      const int q = idx / VEC, cv = idx % VEC;
      const long long qp = gm.query_pixel(q);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = Os[q][cv * 8 + e];
      if (res) {
        float p[8];
        load8f(res + qp * ldr + cv * 8, p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p[e];
      }
      store8f(out + qp * ldo + oc0 + cv * 8, v);
    }
  } else {
    // channel = band * 16 + base channel (band-major nesting of repeated DWTs); item = (query, 4 base channels);
    // with 512 threads the two halves of the workgroup store the upper / lower rows of the (2^L)^2 block
    constexpr int S = ES, NB = Haar<L>::N;
    static_assert(NB == NT && S % EPARTS == 0, "one 16-channel tile per band");
    const int q = e_q, cg = e_cg;
    float vv[4][S][S];
    {
      f32x4 ob[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) ob[b] = *reinterpret_cast<const f32x4*>(&Os[q][16 * b + 4 * cg]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float bands[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) bands[b] = ob[b][c];
        Haar<L>::inv(bands, vv[c]);
      }
    }
    const int H = h * S, W = w * S;
    const int by = 8 * gm.wy + (q >> 3), bx = 8 * gm.wx + (q & 7);
#pragma unroll
    for (int yy = 0; yy < EROWS; ++yy)
#pragma unroll
      for (int xx = 0; xx < S; ++xx) {
        // (static indices into vv: select the row with a compile-time loop over the parts)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < EPARTS; ++pp)
          if (pp == e_part) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = vv[c][pp * EROWS + yy][xx];
          }
        const long long pix = ((long long)gm.b * H + S * by + e_part * EROWS + yy) * W + S * bx + xx;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] += (float)resv[yy][xx][c];
        store4(out + pix * ldo + oc0 + 4 * cg, v);
      }
  }
  M2T_FUSED_STAMP(7);
}

template <int C, int L, int NW, bool PREP = false>
int go_fused(const bf16_t* x, const bf16_t* wfrag, const float* rel_h, const float* rel_w, bf16_t* qkv, bf16_t* out, int ldo, int oc0,
             const bf16_t* res, int ldr, int nwin, int h, int w, hipStream_t st, FusedPrepArgs pa = FusedPrepArgs{}) {
  const size_t sh = FusedCfg<C>::total;
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_fused_fwd_kernel<C, L, NW, PREP>, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED((window_attn_fused_fwd_kernel<C, L, NW, PREP>), dim3(nwin), dim3(NW * 64), sh, st, x, wfrag, rel_h, rel_w, qkv, out, ldo,
                   oc0, res, ldr, h, w, pa);
  return 0;
}

}  // namespace

// bf16; (C, post_levels) in {(64, 1), (256, 2)} -- the two shapes of the model; anything else is an argument error.
// x [B][h][w][C]; wfrag: the qkv weight in M2T_PACK_FRAG16 order; qkv [B][h][w][3C] (written); out / res as in
// launch_window_attn_fwd (post_levels > 0: the full-resolution xc chunk / xin planes; res is then mandatory).
int launch_window_attn_fused_fwd(const void* x_, const void* wfrag_, const float* rel_h, const float* rel_w, void* qkv_, void* out_,
                                 int ldo, int oc0, const void* res_, int ldr, int B, int h, int w, int C, int post_levels,
                                 hipStream_t st) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn_fused: h,w must be multiples of 8");
  if (post_levels != 0 && !res_) return m2t_set_error(-2, "window_attn_fused: the fused IWT epilogue needs the residual");
  const bf16_t* x = (const bf16_t*)x_;
  const bf16_t* wfrag = (const bf16_t*)wfrag_;
  bf16_t* qkv = (bf16_t*)qkv_;
  bf16_t* out = (bf16_t*)out_;
  const bf16_t* res = (const bf16_t*)res_;
  const int nwin = B * (h / 8) * (w / 8);
  if (!((C == 256 && post_levels == 2) || (C == 64 && post_levels == 1)))
    return m2t_set_error(M2T_UNSUPPORTED, "window_attn_fused: unsupported (C, post_levels); built for (64, 1) and (256, 2)");
  int rc = M2T_UNSUPPORTED;
  M2TProfScope ps(C == 64 ? M2T_PROF_ATTN_FUSED_64 : M2T_PROF_ATTN_FUSED_256, st);
  if (C == 256 && post_levels == 2) rc = go_fused<256, 2, 8>(x, wfrag, rel_h, rel_w, qkv, out, ldo, oc0, res, ldr, nwin, h, w, st);
  else if (C == 64 && post_levels == 1) rc = go_fused<64, 1, 4>(x, wfrag, rel_h, rel_w, qkv, out, ldo, oc0, res, ldr, nwin, h, w, st);
  if (rc != 0) return rc;
  M2T_LAUNCH_CHECK();
  return 0;
}

// The same with branch_prep inside (bf16, k >= 1): xn = plane k of the block input, xprev = plane k - 1 of xc (both [B][H][W][16], H = h 2^L),
// mean / rstd [B][64]; xin [B][H][W][16] and d [B][h][w][C] are WRITTEN (own pixels of every window = every pixel once); out = plane k of xc.
int launch_window_attn_fused_prep_fwd(const void* xn, const void* xprev, const float* mean, const float* rstd, int k, void* xin, void* d,
                                      const void* wfrag_, const float* rel_h, const float* rel_w, void* qkv_, void* out_, int B, int h, int w,
                                      int C, int post_levels, hipStream_t st) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn_fused_prep: h,w must be multiples of 8");
  if (!xn || !xprev || !mean || !rstd || !xin || !d || k < 1 || k > 3) return m2t_set_error(-2, "window_attn_fused_prep: null argument or k outside 1..3");
  if (!((C == 256 && post_levels == 2) || (C == 64 && post_levels == 1)))
    return m2t_set_error(M2T_UNSUPPORTED, "window_attn_fused_prep: unsupported (C, post_levels); built for (64, 1) and (256, 2)");
  FusedPrepArgs pa;
  pa.xn = (const bf16_t*)xn; pa.xprev = (const bf16_t*)xprev; pa.mean = mean; pa.rstd = rstd; pa.xin = (bf16_t*)xin; pa.d = (bf16_t*)d; pa.k = k;
  const int nwin = B * (h / 8) * (w / 8);
  int rc;
  M2TProfScope ps(C == 64 ? M2T_PROF_ATTN_FUSED_64 : M2T_PROF_ATTN_FUSED_256, st);
  if (C == 256) rc = go_fused<256, 2, 8, true>(nullptr, (const bf16_t*)wfrag_, rel_h, rel_w, (bf16_t*)qkv_, (bf16_t*)out_, 16, 0, (const bf16_t*)xin, 16, nwin, h, w, st, pa);
  else rc = go_fused<64, 1, 4, true>(nullptr, (const bf16_t*)wfrag_, rel_h, rel_w, (bf16_t*)qkv_, (bf16_t*)out_, 16, 0, (const bf16_t*)xin, 16, nwin, h, w, st, pa);
  if (rc != 0) return rc;
  M2T_LAUNCH_CHECK();
  return 0;
}
