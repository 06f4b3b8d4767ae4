// k_attn_res.hip -- "resident" backward of the 8x8 / 10x10 halo window attention (bf16, C = 64 / 256).
//
// Same mathematics as window_attn_bwd_kernel (k_attn.hip; models/M2Trans_network.py:310-332 under autograd)
// but the whole window -- K^ (keys + rel-pos), V, dO and q for ALL channels -- is loaded from HBM once and
// stays in LDS for both phases, so a window pays one exposed memory latency instead of eight
// (C = 256: 4 channel chunks x 2 phases in the chunked kernel, and only one 4-wave workgroup per CU):
//   phase 0  every global load of the window is issued, then written to LDS (dO through DWT^L when the
//            branch's gradient is the DWT of a full-resolution slice);
//   phase 1  S^T = K^ Q^T and dP^T = V dO^T; with 8 waves the keys are split in two halves per query
//            tile and the softmax statistics are merged through LDS (online max/sum merge);
//   phase 2  the channel tiles are split over the waves: each wave produces dV, dK^ and dq for its 16/32
//            channels with the P / dS operands ([query][key]) shared through LDS; q (phase-1 registers) and
//            P overlay V; the outputs leave through LDS as whole 16-byte-per-lane rows.
// The relative-position gradient (dK^ summed over key columns / rows, phantom keys included) is one more
// key tile of the dK^ product: sum_j dK^[10 i + j][c] = sum_q q[q][c] * (sum_j dS[q][10 i + j]), so the
// row / column sums of dS are appended to dS as columns 128.. / 144.. and fall out of the same MFMA chain.
// LDS (C = 256): K^ 53 328 + V 53 328 + dO 33 792 + dS 21 504 + stats 1 536 = 163 488 B of 163 840.
// Measured (B=16, 32x32, C=256, MI355X): 20 us per launch against 54 us for the chunked kernel; a third of
// it is phase 0 at the HBM fair share of a CU (all 256 workgroups load in lockstep).
#include "m2t_kernels.h"
#include "m2t_haar.h"
#include "m2t_window.h"

#ifndef M2T_RES_STAMP
#define M2T_RES_STAMP(i) do { } while (0)        // scratch/bench_res.hip defines it to record s_memtime per phase
#endif
#ifndef M2T_RES_STAMP2
#define M2T_RES_STAMP2(i) do { } while (0)       // (a second set for the inside of phase 0)
#endif

namespace {

template <int C> struct ResCfg {
  static constexpr int LD = C + 8;
  static constexpr int KROWS = 101;                       // 100 keys + one zero row (all pad keys alias it)
  static constexpr int PLD = 128 + 8;                     // P   [query][key 0..127]            (keys >= 100 are 0)
  static constexpr int DLD = 160 + 8;                     // dS  [query][key 0..127 | 10 row sums @128 | 10 column sums @144]
  static constexpr size_t szK = sizeof(bf16_t) * KROWS * LD;
  static constexpr size_t szQ = sizeof(bf16_t) * 64 * LD;
  static constexpr size_t szP = sizeof(bf16_t) * 64 * PLD;
  static constexpr size_t szD = sizeof(bf16_t) * 64 * DLD;
  static constexpr bool P_IN_V = (szQ + szP <= szK);
  static constexpr size_t offV = szK;
  static constexpr size_t offDO = 2 * szK;
  static constexpr size_t offP = P_IN_V ? offV + szQ : offDO + szQ;
  static constexpr size_t offD = P_IN_V ? offDO + szQ : offP + szP;
  static constexpr size_t offRED = offD + szD;
  static constexpr size_t total = offRED + sizeof(float) * 3 * 2 * 64;
};

// transposed 8-element operand: elements 0..3 = rows row_lo_base + 0..3, 4..7 = rows row_hi_base + 0..3 at
// column col0 + (lane & 15); rows >= zero_row alias the zero row.  (see load8_tr in m2t_common.h)
__device__ __forceinline__ Frag8<bf16_t> tr8(const bf16_t* base, int ld, int row_lo_base, int row_hi_base, int col0,
                                             int lane, int zero_row) {
  const int i = lane & 15, qq = i >> 2, pp = i & 3;
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const int rlo = min(row_lo_base + qq, zero_row), rhi = min(row_hi_base + qq, zero_row);
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rlo * ld + col0 + 4 * pp));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + rhi * ld + col0 + 4 * pp));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}

// Key `key` (0..99) of the window as a 32-bit element offset of its pixel row in a [pixel][row_elems] tensor, relative to the first key
// row of the window (key_row0), clamped into the image (the loads are unconditional), and whether the key lies inside the image.
// 24-bit multiplies on purpose: a 64-bit pixel index per lane and load costs two v_mad_u64_u32 and two v_mul_lo_u32, quarter-rate
// instructions, and phase 0 of these kernels is bound by exactly that integer work (profiles/r06_attn_phase0_integer_work.txt).
struct KeyAddr { unsigned off; bool ok; int kr, kc; };
__device__ __forceinline__ int key_row0(const WinGeom& gm) { return max(8 * gm.wy - 1, 0); }
__device__ __forceinline__ KeyAddr key_addr(const WinGeom& gm, int h, int w, int key, int row_elems) {
  KeyAddr a;
  a.kr = (int)(__umul24((unsigned)key, 205u) >> 11);                             // key / 10 for key < 1024
  a.kc = key - 10 * a.kr;
  const int yy = 8 * gm.wy + a.kr - 1, xx = 8 * gm.wx + a.kc - 1;
  a.ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
  const unsigned pix = __umul24((unsigned)(min(max(yy, 0), h - 1) - key_row0(gm)), (unsigned)w) + (unsigned)min(max(xx, 0), w - 1);
  a.off = __umul24(pix, (unsigned)row_elems);
  return a;
}

// dO rows for the whole window: DOs[q][c], c over all C channels.
// L = 0: plain rows of go (ld, channel offset coff).  L = 1, 2: go is the full-resolution g_xc tensor and the
// branch gradient is DWT^L of its 16-channel slice: thread (q, 4-channel group) reads its (2^L)^2 pixel block
// once and writes every band (band-major channel order, as torch.cat((LL,HL,LH,HH),1) nests).
template <int C, int L, int NTHR>
__device__ __forceinline__ void stage_go_all(bf16_t (*dst)[C + 8], const bf16_t* __restrict__ go, int ld, int coff,
                                             const WinGeom& gm, int tid) {
  if constexpr (L == 0) {
    constexpr int VEC = C / 8;
    constexpr int ITEMS = (64 * VEC + NTHR - 1) / NTHR;
    Frag8<bf16_t> f[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * NTHR;
      f[it] = frag_zero<bf16_t>();
      if (idx < 64 * VEC) f[it] = load8(go + gm.query_pixel(idx / VEC) * ld + coff + (idx % VEC) * 8);
    }
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < 64 * VEC) store8(&dst[idx / VEC][(idx % VEC) * 8], f[it]);
    }
  } else {
    static_assert(C == (16 << (2 * L)), "fused DWT needs C = 16 * 4^L");
    constexpr int S = Haar<L>::S, N = Haar<L>::N;
    if (tid < 256) {
      const int q = tid >> 2, cg = tid & 3;
      const int H = gm.h * S, W = gm.w * S;
      const bf16_t* gwin = go + (((long long)gm.b * H + 8 * S * gm.wy) * W + 8 * S * gm.wx) * ld + coff;      // uniform; the lanes add 32-bit offsets
      const unsigned o0 = __umul24(__umul24((unsigned)(S * (q >> 3)), (unsigned)W) + S * (q & 7), (unsigned)ld) + 4 * cg;
      float v[4][S][S];
#pragma unroll
      for (int y = 0; y < S; ++y)
#pragma unroll
        for (int x = 0; x < S; ++x) {
          float t4[4];
          load4(gwin + o0 + (unsigned)((y * W + x) * ld), t4);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i][y][x] = t4[i];
        }
      float o[4][N];
#pragma unroll
      for (int i = 0; i < 4; ++i) Haar<L>::fwd(v[i], o[i]);
#pragma unroll
      for (int b = 0; b < N; ++b) {
        float t4[4] = {o[0][b], o[1][b], o[2][b], o[3][b]};
        store4(&dst[q][b * 16 + 4 * cg], t4);
      }
    }
  }
}

// The same staging in two steps -- issue() right behind the K^ loads, finish() after the S^T product -- so that the dO / V half
// of the window's load burst is still in flight while the first product runs (non-recomputing kernels).
template <int C, int L, int NTHR> struct GoStage {
  static constexpr int VEC = C / 8;
  static constexpr int ITEMS = (L == 0) ? (64 * VEC + NTHR - 1) / NTHR : 1;
  static constexpr int S = Haar<L>::S, N = Haar<L>::N;
  Frag8<bf16_t> f[ITEMS];
  bf16x4 raw[S][S];                                      // kept as loaded: a conversion here would wait for the loads before the V loads are issued
  __device__ __forceinline__ void issue(const bf16_t* __restrict__ go, int ld, int coff, const WinGeom& gm, int tid) {
    if constexpr (L == 0) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int idx = min(tid + it * NTHR, 64 * VEC - 1);
        f[it] = load8(go + gm.query_pixel(idx / VEC) * ld + coff + (idx % VEC) * 8);
      }
    } else {
      static_assert(L == 0 || C == (16 << (2 * L)), "fused DWT needs C = 16 * 4^L");
      if (tid < 256) {                                   // wave-uniform
        const int q = tid >> 2, cg = tid & 3;
        const int H = gm.h * S, W = gm.w * S;
        const bf16_t* gwin = go + (((long long)gm.b * H + 8 * S * gm.wy) * W + 8 * S * gm.wx) * ld + coff;      // uniform; the lanes add 32-bit offsets
        const unsigned o0 = __umul24(__umul24((unsigned)(S * (q >> 3)), (unsigned)W) + S * (q & 7), (unsigned)ld) + 4 * cg;
#pragma unroll
        for (int y = 0; y < S; ++y)
#pragma unroll
          for (int x = 0; x < S; ++x)
            raw[y][x] = *reinterpret_cast<const bf16x4*>(gwin + o0 + (unsigned)((y * W + x) * ld));
      }
    }
  }
  __device__ __forceinline__ void finish(bf16_t (*dst)[C + 8], int tid) {
    if constexpr (L == 0) {
#pragma unroll
      for (int it = 0; it < ITEMS; ++it) {
        const int idx = tid + it * NTHR;
        if (idx < 64 * VEC) store8(&dst[idx / VEC][(idx % VEC) * 8], f[it]);
      }
    } else {
      if (tid < 256) {
        const int q = tid >> 2, cg = tid & 3;
        float o[4][N];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v[S][S];
#pragma unroll
          for (int y = 0; y < S; ++y)
#pragma unroll
            for (int x = 0; x < S; ++x) v[y][x] = (float)raw[y][x][i];
          Haar<L>::fwd(v, o[i]);
        }
#pragma unroll
        for (int b = 0; b < N; ++b) {
          float t4[4] = {o[0][b], o[1][b], o[2][b], o[3][b]};
          store4(&dst[q][b * 16 + 4 * cg], t4);
        }
      }
    }
  }
};

// DG: the data gradient of the qkv projection, g_d = g_q Wq + g_k Wk + g_v Wv (autograd of models/M2Trans_network.py:281,
// 307), is taken in the same launch.  The product is linear in g_k | g_v, so the window multiplies ITS OWN contributions
// to all 100 keys -- they sit complete in LDS at the end of phase 2 -- and the overlap-add over neighbouring windows
// moves from dK|dV (2C wide, before a separate GEMM launch) to g_d (C wide): own pixels -> gd [pixel][C], ring keys ->
// gdwin [window][36][C].  The weight arrives as pre-packed MFMA A-fragments of Wqkv^T (M2T_PACK_FRAG16_T) straight from
// L2 through a register ring, the B operand is a plain row read of the dq / dK^ / dV rows (dq rows gathered in key order;
// ring keys read a zero row).  gqkv / win are still written: the weight-gradient GEMM (side stream) reads them.
// RC: q | k | v were NOT saved by the forward pass (window_attn_fused_fwd_kernel with qkv == nullptr); they are recomputed
// here from the branch input x [pixel][C] (kept anyway: the weight-gradient GEMM reads it) with the forward kernel's own
// products -- the same pre-packed A-fragments (M2T_PACK_FRAG16), k-step-outer, rounded to bf16 where the forward stored them --
// so every value is bit-identical to the saved one.  The x rows of the 100 keys are staged where V will live, q waits in the
// P region; the window then loads 12.8 KB (C = 64) instead of 33.6 KB and the forward writes 25 MB less per launch.
// PB: the backward of branch_prep of the NEXT branch k = i + 1 (k_pointwise.hip: g_xin = IWT^L(g_d) + g_xc[k]; g_n[k] = g_xin / 2;
// g_xc[k - 1] += g_xin / 2) runs in this kernel's phase 0 instead of in its own launch in front of it: g_xc[k - 1] = g_xc[i] is exactly
// the gradient this window loads, and the window's own pixels are exactly the blocks whose g_d rows it needs (same level, same window
// grid; the ring rows of the <= 3 neighbouring windows come from the previous launch, which is why gd / gdwin are double-buffered per
// branch).  Waves 0..3 (one (pixel, 4 channels) item per thread, as the dO staging always was) do that arithmetic -- the operations, their
// order and the rounding points of branch_prep_bwd_tiled_kernel: identical bits -- and write g_n[k], g_xc[i] and the dO rows, while waves
// 4..7 load and stage K^ | V alone (rel-pos table through LDS), so that neither role holds the other's registers.  Measured
// (scratch/bench_res.hip, batch 16): the kernel grows from 34.0 to 42.6 us and replaces an 11.1 us launch plus its fork gap; phase 0
// is 29.8 k cycles instead of 14.7 k -- the block loads are 8-byte pieces at 128-byte stride (82 KB per window at ~3 TB/s chip-wide:
// 15 k cycles before the K | V burst can start; issued together with it both arrive late: 34 k), the arithmetic (packed fp32) 3.3 k,
// the stores and the forward transform 7 k under the K | V burst.  Worth 0.35 % / 0.45 % of the step at batch 16 / 32; the C = 64 and
// C = 16 consumers (different levels, recomputing kernels) are not built.
struct ResPrepArgs {
  const bf16_t* gdn;       // own-window g_d rows of branch k  [B][h][w][C]
  const bf16_t* gdwinn;    // their ring rows                  [window][36][C]
  const bf16_t* gxk;       // plane k of g_xc                  [B][H][W][16]
  bf16_t* gnk;             // plane k of g_n (written)
};
template <int C, int L, int NW, bool DG, bool RC = false, bool PB = false>
__global__ void __launch_bounds__(NW * 64, (C == 64) ? 2 : 1) window_attn_bwd_res_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                      const float* __restrict__ rel_w, const bf16_t* __restrict__ go,
                                                                      int ldg, int gc0, bf16_t* __restrict__ gqkv,
                                                                      bf16_t* __restrict__ win, float* __restrict__ relw, int h, int w,
                                                                      const bf16_t* __restrict__ wdfrag, bf16_t* __restrict__ gd,
                                                                      bf16_t* __restrict__ gdwin, const bf16_t* __restrict__ xsrc,
                                                                      const bf16_t* __restrict__ wfrag, ResPrepArgs pb) {
  using T = bf16_t;
  using Cfg = ResCfg<C>;
  constexpr int LD = Cfg::LD, PLD = Cfg::PLD, DLD = Cfg::DLD, NTHR = NW * 64, VEC = C / 8, NT = C / 16, KH = NW / 4, TPW = NT / NW, NKC = C / 32;
  constexpr int ZR = 100;                                  // the zero row of Kh / Vs
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(NT % NW == 0 && TPW >= 1, "channel tiles must split evenly over the waves");
  static_assert(Cfg::total <= 163840, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Kh)[LD] = reinterpret_cast<T(*)[LD]>(smem);
  T(*Vs)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offV);          // phase 1
  T(*Qs)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offV);          // phase 2 (aliases Vs)
  T(*DOs)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offDO);
  T(*Pq)[PLD] = reinterpret_cast<T(*)[PLD]>(smem + Cfg::offP);        // [query][key]; inside the V region when it fits
  T(*Dq)[DLD] = reinterpret_cast<T(*)[DLD]>(smem + Cfg::offD);
  float(*red)[2][64] = reinterpret_cast<float(*)[2][64]>(smem + Cfg::offRED);   // [max | sum | delta][key half][query]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const int qt = wv & 3, kh = wv >> 2;
  const int q = 16 * qt + lr;
  const long long qpix = gm.query_pixel(q);

  M2T_RES_STAMP(0);
  Frag8<T> qreg[NKC];
  constexpr int NTL = (KH == 2) ? 4 : WA_KT;               // key tiles per wave in phase 1
  f32x4 s[NTL], dp[NTL];
  if constexpr (RC) {
    // ---- phase 0 (recompute): x rows of the 100 keys, the rel-pos table, dO; then k | v | q by MFMA ----
    static_assert(!RC || (sizeof(T) * 64 * LD <= Cfg::szP && sizeof(float) * 10 * C <= Cfg::szD && !Cfg::P_IN_V), "q / rel-pos staging regions");
    T(*Xs)[LD] = Vs;                                                             // x rows (natural key order); row 100 stays zero
    T(*Qt)[LD] = reinterpret_cast<T(*)[LD]>(smem + Cfg::offP);                   // q of the 64 queries until phase 1 has read it
    float(*RelS)[C] = reinterpret_cast<float(*)[C]>(smem + Cfg::offD);           // T[kk][ch] = ch < C/2 ? rel_h[kk][ch] : rel_w[kk][ch - C/2]
    constexpr int NKS = C / 32;
    const bf16x8* wf8 = reinterpret_cast<const bf16x8*>(wfrag);
    constexpr int XIT = (101 * VEC + NTHR - 1) / NTHR, RIT = (10 * C / 4 + NTHR - 1) / NTHR;
    // this wave's weight fragments: projection p (q, k, v), channel tiles wv * TPW + m, every k-step -- issued in front of the x
    // rows (behind their LDS writes they cost a second exposed round trip)
    Frag8<T> wq[NKS][TPW], wk[NKS][TPW], wvv[NKS][TPW];
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        wq[ks][m].v = wf8[((size_t)((0 * NT + wv * TPW + m) * NKS + ks)) * 64 + lane];
        wk[ks][m].v = wf8[((size_t)((1 * NT + wv * TPW + m) * NKS + ks)) * 64 + lane];
        wvv[ks][m].v = wf8[((size_t)((2 * NT + wv * TPW + m) * NKS + ks)) * 64 + lane];
      }
    {
      Frag8<T> xf[XIT];
      bool ok[XIT];
      f32x4 rf[RIT];
      const T* xwin = xsrc + ((long long)gm.b * h + key_row0(gm)) * w * C;        // uniform; the lanes add 32-bit offsets
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        const int idx = tid + it * NTHR;
        const int key = min(idx / VEC, 100), cv = idx % VEC;
        const KeyAddr ka = key_addr(gm, h, w, min(key, 99), C);
        ok[it] = (key < 100) && ka.ok;
        xf[it] = load8(xwin + ka.off + cv * 8);                                          // branch-free: clamped address, select below
      }
#pragma unroll
      for (int it = 0; it < RIT; ++it) {
        const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
        const int kk = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
        const float* rp = (c4 < C / 2) ? (rel_h + kk * (C / 2) + c4) : (rel_w + kk * (C / 2) + (c4 - C / 2));
        rf[it] = *reinterpret_cast<const f32x4*>(rp);
      }
      stage_go_all<C, L, NTHR>(DOs, go, ldg, gc0, gm, tid);
#pragma unroll
      for (int it = 0; it < XIT; ++it) {
        const int idx = tid + it * NTHR;
        if (idx < 101 * VEC) store8(&Xs[min(idx / VEC, 100)][(idx % VEC) * 8], ok[it] ? xf[it] : frag_zero<T>());
      }
#pragma unroll
      for (int it = 0; it < RIT; ++it) {
        const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
        *reinterpret_cast<f32x4*>(&RelS[idx / (C / 4)][(idx % (C / 4)) * 4]) = rf[it];
      }
      if (tid < VEC) store8(&Kh[ZR][tid * 8], frag_zero<T>());
    }
    __syncthreads();
    f32x4 ak[WA_KT][TPW], av[WA_KT][TPW], aq[4][TPW];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t)
#pragma unroll
      for (int m = 0; m < TPW; ++m) { ak[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; av[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < TPW; ++m) aq[t][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        const Frag8<T> bx = load8(&Xs[min(16 * t + lr, ZR)][32 * ks + 8 * g]);
#pragma unroll
        for (int m = 0; m < TPW; ++m) {
          mma16(ak[t][m], wk[ks][m], bx);
          mma16(av[t][m], wvv[ks][m], bx);
        }
      }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int qq = 16 * t + lr;                                                 // query qq = key ((qq >> 3) + 1, (qq & 7) + 1)
        const Frag8<T> bx = load8(&Xs[((qq >> 3) + 1) * 10 + (qq & 7) + 1][32 * ks + 8 * g]);
#pragma unroll
        for (int m = 0; m < TPW; ++m) mma16(aq[t][m], wq[ks][m], bx);
      }
    // K^ = bf16(bf16(k) + rel) -> Kh, q -> Qt: the rounding points of the forward kernel's stores
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = 16 * t + lr;
      if (key < WA_NK) {
        const int kr = key / 10, kc = key - kr * 10;
#pragma unroll
        for (int m = 0; m < TPW; ++m) {
          const int ch = 16 * (wv * TPW + m) + 4 * g;
          const bf16x4 kb = {(bf16_t)ak[t][m][0], (bf16_t)ak[t][m][1], (bf16_t)ak[t][m][2], (bf16_t)ak[t][m][3]};
          const f32x4 r4 = *reinterpret_cast<const f32x4*>(&RelS[(ch < C / 2) ? kr : kc][ch]);
          float kh4[4] = {(float)kb[0] + r4[0], (float)kb[1] + r4[1], (float)kb[2] + r4[2], (float)kb[3] + r4[3]};
          store4(&Kh[key][ch], kh4);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        float q4[4] = {aq[t][m][0], aq[t][m][1], aq[t][m][2], aq[t][m][3]};
        store4(&Qt[16 * t + lr][16 * (wv * TPW + m) + 4 * g], q4);
      }
    __syncthreads();                      // every wave is done reading the x rows; K^ and q are complete
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = 16 * t + lr;
      if (key < WA_NK) {
#pragma unroll
        for (int m = 0; m < TPW; ++m) {
          float v4[4] = {av[t][m][0], av[t][m][1], av[t][m][2], av[t][m][3]};
          store4(&Vs[key][16 * (wv * TPW + m) + 4 * g], v4);                        // row 100 stays the zero row
        }
      }
    }
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) qreg[kc] = load8(&Qt[q][kc * 32 + 8 * g]);
    __syncthreads();
  } else if constexpr (PB) {
  static_assert(!PB || (NW == 8 && L == 2 && C == 256 && !RC), "branch_prep_bwd inside: the C = 256 / L = 2 kernel");
  constexpr int S = 4, NB = 16;
  float(*RelS)[C] = reinterpret_cast<float(*)[C]>(smem + Cfg::offD);             // T[kk][ch] = ch < C/2 ? rel_h[kk][ch] : rel_w[kk][ch - C/2]
  T* CornS = reinterpret_cast<T*>(smem + Cfg::offD + sizeof(float) * 10 * C);   // [4 corners][2 sources][C]
  static_assert(sizeof(float) * 10 * C + sizeof(T) * 8 * C <= Cfg::szD, "rel-pos table + corner ring rows staging");
  constexpr int RIT = (10 * C / 4 + NTHR - 1) / NTHR;
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) qreg[kc] = load8(qkv + qpix * (3 * C) + kc * 32 + 8 * g);
  f32x4 rf[RIT];
#pragma unroll
  for (int it = 0; it < RIT; ++it) {
    const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
    const int kk = idx / (C / 4), c4 = (idx % (C / 4)) * 4;
    const float* rp = (c4 < C / 2) ? (rel_h + kk * (C / 2) + c4) : (rel_w + kk * (C / 2) + (c4 - C / 2));
    rf[it] = *reinterpret_cast<const f32x4*>(rp);
  }
  M2T_RES_STAMP2(0);                    // (bench: q and table loads issued)
  auto stage_rel = [&]() {
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
      const int idx = min(tid + it * NTHR, 10 * C / 4 - 1);
      *reinterpret_cast<f32x4*>(&RelS[idx / (C / 4)][(idx % (C / 4)) * 4]) = rf[it];
    }
  };
  if (wv >= 4) {
    // ---- loader role: K^ | V of the 100 keys, 256 threads ----
    constexpr int KIT2 = (WA_NK * VEC + 255) / 256;
    const int t2 = tid - 256;
    // (the K | V burst starts BEHIND the barrier that the prep waves reach once their g_d rows have arrived: issued together with
    // the prep role's loads it made both arrive late -- the window then moves 248 KB through a bandwidth-bound burst and the block
    // arithmetic could only start at its end; this way the arithmetic runs under the K | V burst)
    Frag8<T> kf[KIT2], vf[KIT2];
    {
      // the second and third ring source of the window's four corner pixels (the only pixels that have them): one 16-byte piece per
      // loader thread into CornS[corner][source - 1][C], so that the prep role does not chase them through a dependent load loop
      const int cn = t2 >> 6, a = 1 + ((t2 >> 5) & 1), vec = t2 & 31;
      const int cq = (cn & 1) * 7 + (cn >> 1) * 56;                                 // query pixels 0, 7, 56, 63
      long long ho[3];
      const int ns = halo_sources(gm.b, 8 * gm.wy + (cq >> 3), 8 * gm.wx + (cq & 7), gm.nh, gm.nw, C, ho);
      const Frag8<T> piece = load8(pb.gdwinn + (a < ns ? ho[a] : 0) + vec * 8);
      M2T_RES_STAMP2(1);                // (bench: loader role, corner piece issued)
      stage_rel();
      M2T_RES_STAMP2(2);                // (bench: loader role, table staged = q and table landed)
      store8(&CornS[(cn * 2 + (a - 1)) * C + vec * 8], a < ns ? piece : frag_zero<T>());
      M2T_RES_STAMP2(3);                // (bench: loader role, corner piece landed and staged)
    }
    lds_barrier();
    M2T_RES_STAMP(13);
    const T* kwin = qkv + ((long long)gm.b * h + key_row0(gm)) * w * (3 * C);    // uniform; the lanes add 32-bit offsets
    static_assert(256 % VEC == 0, "a thread keeps its channel piece over the iterations");
    bool kok[KIT2];
#pragma unroll
    for (int it = 0; it < KIT2; ++it) {
      const int idx = t2 + it * 256;
      const int cv = idx % VEC, key = min(idx / VEC, WA_NK - 1);
      const KeyAddr ka = key_addr(gm, h, w, key, 3 * C);                          // clamped: the loads are unconditional
      kok[it] = ka.ok;
      kf[it] = load8(kwin + ka.off + C + cv * 8);
      vf[it] = load8(kwin + ka.off + 2 * C + cv * 8);
    }
#pragma unroll
    for (int it = 0; it < KIT2; ++it) {
      const int idx = t2 + it * 256;
      const int cv = idx % VEC, key = idx / VEC;
      if (it * 256 + 255 < WA_NK * VEC || key < WA_NK) {                          // (a branch in the last iteration only)
        const int kr = (int)(__umul24((unsigned)key, 205u) >> 11), kc = key - 10 * kr;
        const bool ok = kok[it];
        const int cc = cv * 8;
        const f32x4 ra = *reinterpret_cast<const f32x4*>(&RelS[(cc < C / 2) ? kr : kc][cc]);
        const f32x4 rb = *reinterpret_cast<const f32x4*>(&RelS[(cc < C / 2) ? kr : kc][cc + 4]);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = (ok ? kf[it].get(e) : 0.f) + ra[e]; v[4 + e] = (ok ? kf[it].get(4 + e) : 0.f) + rb[e]; }
        store8f(&Kh[key][cv * 8], v);
        store8(&Vs[key][cv * 8], ok ? vf[it] : frag_zero<T>());
      }
    }
    if (t2 < VEC) {
      store8(&Kh[ZR][t2 * 8], frag_zero<T>());
      store8(&Vs[ZR][t2 * 8], frag_zero<T>());
    }
  } else {
    // ---- prep role: thread (query pixel pq = tid >> 2, 4 base channels cg): its 4x4 block of the full-resolution planes ----
    const int pq = tid >> 2, cg = tid & 3;
    const int H = gm.h * S, W = gm.w * S;
    const int by = 8 * gm.wy + (pq >> 3), bx = 8 * gm.wx + (pq & 7);
    long long hoff[3];
    const int nsrc = halo_sources(gm.b, by, bx, gm.nh, gm.nw, C, hoff);
    M2T_RES_STAMP2(1);                  // (bench: prep role, ring sources known)
    bf16x4 rd[NB], rr[NB], p3[S][S], p2[S][S];
    {
      const T* dwin = pb.gdn + (((long long)gm.b * gm.h + 8 * gm.wy) * gm.w + 8 * gm.wx) * C;                 // uniform; the lanes add 32-bit offsets
      const unsigned d0 = (__umul24((unsigned)(pq >> 3), (unsigned)gm.w) + (pq & 7)) * C + 4 * cg;
#pragma unroll
      for (int n = 0; n < NB; ++n) rd[n] = *reinterpret_cast<const bf16x4*>(dwin + d0 + n * 16);
      const T* r0p = pb.gdwinn + (nsrc > 0 ? hoff[0] : 0) + 4 * cg;               // clamped: unconditional loads, selected below
#pragma unroll
      for (int n = 0; n < NB; ++n) rr[n] = *reinterpret_cast<const bf16x4*>(r0p + n * 16);
    }
    const long long pwin = (((long long)gm.b * H + 8 * S * gm.wy) * W + 8 * S * gm.wx) * 16;                   // uniform
    const unsigned po0 = (__umul24((unsigned)(S * (pq >> 3)), (unsigned)W) + S * (pq & 7)) * 16 + 4 * cg;
#define PB_PO(y, x) (po0 + (unsigned)(((y) * W + (x)) * 16))
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        const unsigned po = PB_PO(y, x);
        p3[y][x] = *reinterpret_cast<const bf16x4*>(pb.gxk + pwin + po);
        p2[y][x] = *reinterpret_cast<const bf16x4*>(go + pwin + po);
      }
    M2T_RES_STAMP2(2);                  // (bench: prep role, every load issued)
    // two channel pairs per thread, packed fp32 (v_pk_add / v_pk_mul: IEEE per component).  g_d row of the block = own-window products
    // + the neighbours' ring rows (fp32 adds in halo_sources' order, one rounding to bf16).  The first sums sit in front of the
    // barrier on purpose: the loader waves start the K | V burst when these loads have landed
    f32x2 o2[2][NB];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const f32x2 q2 = {(float)rd[n][2 * c2], (float)rd[n][2 * c2 + 1]};
        const f32x2 sum = q2 + (f32x2){(float)rr[n][2 * c2], (float)rr[n][2 * c2 + 1]};
        o2[c2][n] = nsrc > 0 ? sum : q2;
      }
    stage_rel();
    M2T_RES_STAMP2(3);                  // (bench: prep role, table staged, at the barrier)
    lds_barrier();
    M2T_RES_STAMP(13);                  // (bench: g_d rows of the block arrived, rel-pos barrier passed)
    if (nsrc > 1) {                                                               // corner pixels only (4 of 64): rows staged by the loader waves
      const int cn = ((pq & 7) == 7 ? 1 : 0) + (pq >= 56 ? 2 : 0);
      for (int a = 1; a < nsrc; ++a) {
        const T* rp = CornS + (cn * 2 + (a - 1)) * C + 4 * cg;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
          const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(rp + n * 16);
          o2[0][n] += (f32x2){(float)r4[0], (float)r4[1]};
          o2[1][n] += (f32x2){(float)r4[2], (float)r4[3]};
        }
      }
    }
    f32x2 vq[2][S][S];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
      for (int n = 0; n < NB; ++n) o2[c2][n] = (f32x2){to_f(from_f<T>(o2[c2][n][0])), to_f(from_f<T>(o2[c2][n][1]))};
      Haar2Inv2::inv(o2[c2], vq[c2]);
    }
    M2T_RES_STAMP(12);                  // (bench: DWT^-1 of the block done)
    // g_n[k] = (IWT + g_xc[k]) / 2 ; g_xc[i] += that ; the new g_xc[i] (rounded as stored) is this window's dO before DWT^L
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        const unsigned po = PB_PO(y, x);
        f32x2 qv[2], pp[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          qv[c2] = (vq[c2][y][x] + (f32x2){(float)p3[y][x][2 * c2], (float)p3[y][x][2 * c2 + 1]}) * (f32x2){0.5f, 0.5f};
          pp[c2] = (f32x2){(float)p2[y][x][2 * c2], (float)p2[y][x][2 * c2 + 1]} + qv[c2];
        }
        const bf16x4 gv = {(bf16_t)qv[0][0], (bf16_t)qv[0][1], (bf16_t)qv[1][0], (bf16_t)qv[1][1]};
        *reinterpret_cast<bf16x4*>(pb.gnk + pwin + po) = gv;
        const bf16x4 nv = {(bf16_t)pp[0][0], (bf16_t)pp[0][1], (bf16_t)pp[1][0], (bf16_t)pp[1][1]};
        *reinterpret_cast<bf16x4*>(const_cast<T*>(go) + pwin + po) = nv;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) vq[c2][y][x] = (f32x2){(float)nv[2 * c2], (float)nv[2 * c2 + 1]};
      }
    M2T_RES_STAMP(14);                  // (bench: full-resolution planes read, combined, stored)
    f32x2 ob[2][NB];
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) Haar2<L>::fwd(vq[c2], ob[c2]);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const bf16x4 dv = {(bf16_t)ob[0][n][0], (bf16_t)ob[0][n][1], (bf16_t)ob[1][n][0], (bf16_t)ob[1][n][1]};
      *reinterpret_cast<bf16x4*>(&DOs[pq][n * 16 + 4 * cg]) = dv;
    }
  }
  M2T_RES_STAMP(10);                    // (bench: this wave's phase-0 role done)
  __syncthreads();
  } else {
  // ---- phase 0: every global load of the window is issued -- q first (phase 1 needs it in registers; issued behind the LDS
  // writes it cost a second exposed round trip), K^ | rel, V, dO -- then written to LDS.  All 256 workgroups run this phase at the
  // same time and the chip's memory system bounds it: 266 KB of lane requests per workgroup in 16 k cycles = 16 B/clk per CU, the
  // rate at which 256 CUs streaming together are served (scratch/bench_ta.hip: 15 B/clk per CU; a CU alone on L2-resident data
  // gets 62).  A split that starts S^T on K^ while V / dO are still arriving gained nothing, and neither did the rel-pos table
  // through LDS (10 KB once instead of two 16-byte loads beside every K load: those hit L1 / L2, the extra barrier cost 1 us --
  // profiles/r06_attn_phase0_integer_work.txt) ----
  constexpr int KIT = (WA_NK * VEC + NTHR - 1) / NTHR;
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) qreg[kc] = load8(qkv + qpix * (3 * C) + kc * 32 + 8 * g);
  Frag8<T> kf[KIT], vf[KIT];
  f32x4 r0[KIT], r1[KIT];
  bool kok[KIT];
  unsigned koff[KIT];
  const T* kwin = qkv + ((long long)gm.b * h + key_row0(gm)) * w * (3 * C);      // uniform; the lanes add 32-bit offsets
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int idx = tid + it * NTHR;
    const int cv = idx % VEC, key = min(idx / VEC, WA_NK - 1);
    const KeyAddr ka = key_addr(gm, h, w, key, 3 * C);                            // clamped: the load is unconditional
    kok[it] = ka.ok;
    koff[it] = ka.off + cv * 8;
    kf[it] = load8(kwin + koff[it] + C);
    const int cc = cv * 8;
    const float* rp = (cc < C / 2) ? (rel_h + ka.kr * (C / 2) + cc) : (rel_w + ka.kc * (C / 2) + (cc - C / 2));
    r0[it] = *reinterpret_cast<const f32x4*>(rp);
    r1[it] = *reinterpret_cast<const f32x4*>(rp + 4);
  }
#pragma unroll
  for (int it = 0; it < KIT; ++it) vf[it] = load8(kwin + koff[it] + 2 * C);
  GoStage<C, L, NTHR> gos;
  gos.issue(go, ldg, gc0, gm, tid);
  M2T_RES_STAMP(10);                    // (bench: every load of the window issued)
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int idx = tid + it * NTHR;
    const int cv = idx % VEC, key = idx / VEC;
    if (it * NTHR + NTHR - 1 < WA_NK * VEC || key < WA_NK) {                      // (a branch in the last iteration only)
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = (kok[it] ? kf[it].get(e) : 0.f) + r0[it][e]; v[4 + e] = (kok[it] ? kf[it].get(4 + e) : 0.f) + r1[it][e]; }
      store8f(&Kh[key][cv * 8], v);
    }
  }
  M2T_RES_STAMP(11);                    // (bench: K^ rows arrived, rel-pos added, staged)
  gos.finish(DOs, tid);
  M2T_RES_STAMP(12);                    // (bench: dO through DWT^L staged)
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int idx = tid + it * NTHR;
    const int cv = idx % VEC, key = idx / VEC;
    if (key < WA_NK) store8(&Vs[key][cv * 8], kok[it] ? vf[it] : frag_zero<T>());
  }
  if (tid < VEC) {
    store8(&Kh[ZR][tid * 8], frag_zero<T>());
    store8(&Vs[ZR][tid * 8], frag_zero<T>());
  }
  __syncthreads();
  }
  {
  M2T_RES_STAMP(1);
  // ---- phase 1: S^T = K^ Q^T and dP^T = V dO^T for (query tile qt, key tiles t0 .. t0 + NTL - 1) ----
  const int t0 = kh * 4;
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) { s[tl] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[tl] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) {
    const Frag8<T> gf = load8(&DOs[q][kc * 32 + 8 * g]);
#pragma unroll
    for (int tl = 0; tl < NTL; ++tl) {
      const int row = min(16 * (t0 + tl) + lr, ZR);
      const Frag8<T> kf = load8(&Kh[row][kc * 32 + 8 * g]);
      mma16(s[tl], kf, qreg[kc]);
      const Frag8<T> vf = load8(&Vs[row][kc * 32 + 8 * g]);
      mma16(dp[tl], vf, gf);
    }
  }
  }
  const int t0 = kh * 4;
  M2T_RES_STAMP(2);
  // ---- softmax over the 100 keys and dS ----
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
    __syncthreads();                    // also: every wave is done with the Kh / Vs reads of phase 1
    const float m0 = red[0][0][q], m1 = red[0][1][q];
    const float mm = fmaxf(m0, m1);
    const float f0 = __expf(m0 - mm), f1 = __expf(m1 - mm);
    const float tot = red[1][0][q] * f0 + red[1][1][q] * f1;
    inv = (kh ? f1 : f0) / tot;
  } else {
    __syncthreads();
    inv = 1.0f / sum;
  }
  float delta = 0.f;
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[tl][r] *= inv;                  // P
      delta += s[tl][r] * dp[tl][r];
    }
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
  if constexpr (KH == 2) {
    if (g == 0) red[2][kh][q] = delta;
    __syncthreads();
    delta = red[2][0][q] + red[2][1][q];
  }
  // P and dS as [query][key]: a lane's four keys of a tile are contiguous (masked keys give exact zeros)
#pragma unroll
  for (int tl = 0; tl < NTL; ++tl) {
    float pv[4], dv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { pv[r] = s[tl][r]; dv[r] = s[tl][r] * (dp[tl][r] - delta) * scale; }
    store4(&Pq[q][16 * (t0 + tl) + 4 * g], pv);
    store4(&Dq[q][16 * (t0 + tl) + 4 * g], dv);
  }
  if constexpr (KH == 1) {              // keys 112..127 (contraction padding of dq)
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    store4(&Dq[q][112 + 4 * g], z);
  }
  if (kh == 0) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) store8(&Qs[q][kc * 32 + 8 * g], qreg[kc]);
  }
  __syncthreads();
  // row / column sums of dS (over key columns j -> @128 + i, over key rows -> @144 + i) as one more MFMA product:
  // sums[n][q] = sum_key Ind[n][key] dS[q][key] with the 0 / 1 indicator built in registers (n-tile 0: key / 10 == n, n-tile 1:
  // key % 10 == n; pad keys carry dS = 0).  The scalar loop it replaces (20 two-byte LDS reads per item, 20 of 64 lanes busy)
  // was 2.5 k (C = 256) / 5 k (C = 64) cycles of every window; a sum of <= 10 bf16 values is exact in fp32 in either order
  // unless their exponents are > 16 bits apart.
  {
    constexpr int JN = (NW == 8) ? 1 : 2;                    // n-tiles per wave: 8 waves = 4 query tiles x 2 n-tiles
    const int sqt = (NW == 8) ? (wv & 3) : wv, nt0 = (NW == 8) ? (wv >> 2) : 0;
    f32x4 sacc[JN];
#pragma unroll
    for (int jn = 0; jn < JN; ++jn) sacc[jn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const Frag8<T> bd = load8(&Dq[16 * sqt + lr][32 * ks + 8 * g]);
#pragma unroll
      for (int jn = 0; jn < JN; ++jn) {
        // key = 32 ks + 8 g + e.  rows: key / 10 == lr <=> (unsigned)(key - 10 lr) < 10; columns: key % 10 == lr <=>
        // (8 g - lr) mod 10 == (10 - (32 ks + e) % 10) % 10, a compile-time constant per element
        const bool rows = (nt0 + jn) == 0;
        const int rbase = 8 * g - 10 * lr, cm0 = (8 * g - lr + 20) % 10;
        unsigned pk[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
          const int c0 = 32 * ks + 2 * e2, c1 = c0 + 1;
          bool h0, h1;
          if (rows) { h0 = (unsigned)(rbase + c0) < 10u; h1 = (unsigned)(rbase + c1) < 10u; }
          else { h0 = cm0 == (10 - c0 % 10) % 10; h1 = cm0 == (10 - c1 % 10) % 10; }
          pk[e2] = (h0 ? 0x3F80u : 0u) | (h1 ? 0x3F800000u : 0u);
        }
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 pv = {pk[0], pk[1], pk[2], pk[3]};
        Frag8<T> ai;
        ai.v = __builtin_bit_cast(bf16x8, pv);
        mma16(sacc[jn], ai, bd);
      }
    }
#pragma unroll
    for (int jn = 0; jn < JN; ++jn) {
      float v4[4] = {sacc[jn][0], sacc[jn][1], sacc[jn][2], sacc[jn][3]};
      store4(&Dq[16 * sqt + lr][128 + 16 * (nt0 + jn) + 4 * g], v4);
    }
  }
  __syncthreads();

  M2T_RES_STAMP(3);
  // ---- phase 2: wave wv owns channel tiles wv*TPW .. wv*TPW + TPW - 1 ----
  // The 16x16 output tiles (4 channels x 1 row per lane) go through LDS and leave as whole rows, 16 bytes per
  // lane: dV -> the V region (q, P are dead once dV and dK^ are computed), dK^ -> the K^ region and dq -> the dO
  // region (K^ is dead once dq is computed).
  T(*VOUT)[LD] = Vs;
  T(*KOUT)[LD] = Kh;
  T(*QOUT)[LD] = DOs;
  const int mt0 = wv * TPW;
  // dV^T [c][key] = sum_q dO[q][c] P[q][key] ; dK^^T [c][key] = sum_q q[q][c] dS[q][key]; tile 7 of dK^ = the
  // rel-pos sums of this tile's channel half
  f32x4 av[TPW][WA_KT], ak[TPW][WA_KT + 1];
#pragma unroll
  for (int m = 0; m < TPW; ++m) {
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) av[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t <= WA_KT; ++t) ak[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    Frag8<T> a[TPW];
#pragma unroll
    for (int m = 0; m < TPW; ++m) a[m] = tr8(&DOs[0][0], LD, 32 * kc + 8 * g, 32 * kc + 8 * g + 4, 16 * (mt0 + m), lane, 64);
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const Frag8<T> b = tr8(&Pq[0][0], PLD, 32 * kc + 8 * g, 32 * kc + 8 * g + 4, 16 * t, lane, 64);
#pragma unroll
      for (int m = 0; m < TPW; ++m) mma16(av[m][t], a[m], b);
    }
  }
#pragma unroll
  for (int kc = 0; kc < 2; ++kc) {
    Frag8<T> a[TPW];
#pragma unroll
    for (int m = 0; m < TPW; ++m) a[m] = tr8(&Qs[0][0], LD, 32 * kc + 8 * g, 32 * kc + 8 * g + 4, 16 * (mt0 + m), lane, 64);
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const Frag8<T> b = tr8(&Dq[0][0], DLD, 32 * kc + 8 * g, 32 * kc + 8 * g + 4, 16 * t, lane, 64);
#pragma unroll
      for (int m = 0; m < TPW; ++m) mma16(ak[m][t], a[m], b);
    }
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
      const int col0 = (16 * (mt0 + m) < C / 2) ? 128 : 144;
      const Frag8<T> b = tr8(&Dq[0][0], DLD, 32 * kc + 8 * g, 32 * kc + 8 * g + 4, col0, lane, 64);
      mma16(ak[m][WA_KT], a[m], b);
    }
  }
  M2T_RES_STAMP(4);
  if (lr < 10) {
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
      float v[4] = {ak[m][WA_KT][0], ak[m][WA_KT][1], ak[m][WA_KT][2], ak[m][WA_KT][3]};
      store4(relw + ((long long)gm.wi * 10 + lr) * C + 16 * (mt0 + m) + 4 * g, v);
    }
  }
  __syncthreads();                      // every wave is done with dO, q and P
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = 16 * t + lr;
    if (key < WA_NK) {
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        float u[4] = {av[m][t][0], av[m][t][1], av[m][t][2], av[m][t][3]};
        store4(&VOUT[key][16 * (mt0 + m) + 4 * g], u);
      }
    }
  }
  // DG: the first DEPTH k-steps of the weight stream are requested here, in front of the dq product, so that their L2 round trip
  // runs under it
  constexpr int NK3 = 3 * C / 32, DEPTH = (NK3 < 8) ? NK3 : 8;     // 8 fragments per tile in flight: the L2 stream is latency-bound
  Frag8<T> wb[TPW][DG ? DEPTH : 1];
  auto wfetch = [&](int ks, int slot) {
#pragma unroll
    for (int m = 0; m < TPW; ++m) wb[m][slot] = load8(wdfrag + (((long long)(mt0 + m) * NK3 + ks) * 64 + lane) * 8);
  };
  if constexpr (DG) {
#pragma unroll
    for (int ks = 0; ks < DEPTH; ++ks) wfetch(ks, ks);
    __builtin_amdgcn_sched_barrier(0);
  }
  // dq^T [c][q] = sum_keys K^[key][c] dS[q][key]
  f32x4 o[TPW][4];
#pragma unroll
  for (int m = 0; m < TPW; ++m)
#pragma unroll
    for (int t = 0; t < 4; ++t) o[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4) {
    Frag8<T> a[TPW];
#pragma unroll
    for (int m = 0; m < TPW; ++m) a[m] = tr8(&Kh[0][0], LD, 32 * c4 + 8 * g, 32 * c4 + 8 * g + 4, 16 * (mt0 + m), lane, ZR);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const Frag8<T> b = load8(&Dq[16 * t + lr][32 * c4 + 8 * g]);
#pragma unroll
      for (int m = 0; m < TPW; ++m) mma16(o[m][t], a[m], b);
    }
  }
  __syncthreads();                      // every wave is done with K^; the dV rows are complete
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = 16 * t + lr;
    if (key < WA_NK) {
#pragma unroll
      for (int m = 0; m < TPW; ++m) {
        float v[4] = {ak[m][t][0], ak[m][t][1], ak[m][t][2], ak[m][t][3]};
        store4(&KOUT[key][16 * (mt0 + m) + 4 * g], v);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int m = 0; m < TPW; ++m) {
      float v[4] = {o[m][t][0], o[m][t][1], o[m][t][2], o[m][t][3]};
      store4(&QOUT[16 * t + lr][16 * (mt0 + m) + 4 * g], v);
    }
  M2T_RES_STAMP(5);
  if constexpr (DG) {
    // Row order of this product: the window's OWN 64 pixels first (row tiles 0..3, query order), then the 36 ring keys (tiles 4..6, ring_index
    // order, 12 pad rows on the zero row).  Ring keys have no dq row, so the dq part of the k range runs on four row tiles instead of seven
    // (-14 % of the phase's MFMAs; in key order every tile mixes own and ring keys and the ring lanes multiplied zeros).
    int krow[WA_KT], qrow[WA_KT];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      if (t < 4) {
        const int qq = 16 * t + lr;
        qrow[t] = qq;
        krow[t] = 10 * ((qq >> 3) + 1) + (qq & 7) + 1;
      } else {
        const int r = 16 * (t - 4) + lr;                          // ring_index^-1: rows 0 and 9 (10 keys each), then columns 0 and 9 of rows 1..8
        const int kr = (r < 10) ? 0 : ((r < 20) ? 9 : ((r < 28) ? r - 19 : r - 27));
        const int kc = (r < 10) ? r : ((r < 20) ? r - 10 : ((r < 28) ? 0 : 9));
        qrow[t] = 0;                                              // (never read: the dq part skips these tiles)
        krow[t] = (r < WA_RING) ? 10 * kr + kc : ZR;
      }
    }
    f32x4 ad[TPW][WA_KT];
#pragma unroll
    for (int m = 0; m < TPW; ++m)
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) ad[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    lds_barrier();                        // dq, dK^, dV rows are complete in LDS
    // B operands (7 row reads per k-step, each feeding only TPW MFMAs) are read one k-step ahead, so the LDS round
    // trip of step ks + 1 runs under the MFMAs of step ks
    auto bfetch = [&](int ks, Frag8<T> (&bf)[WA_KT]) {
      const int part = ks / NKC, kk = ks - part * NKC;
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        if (part == 0 && t >= 4) continue;                        // ring rows: no dq
        const T* rp = (part == 0) ? &QOUT[qrow[t]][0] : ((part == 1) ? &KOUT[krow[t]][0] : &VOUT[krow[t]][0]);
        bf[t] = load8(rp + 32 * kk + 8 * g);
      }
    };
    // The dq | dK^ | dV rows leave for gqkv / win WHILE the product runs: item j = tid + slot * NTHR (V rows, then K^ rows,
    // then dq rows, 16 bytes each) is read from LDS in k-step `slot / SPS` and stored one k-step later, so the 134 KB
    // (C = 256) that used to leave in a burst behind the last MFMA ride under the weight stream instead (stamps,
    // scratch/bench_res.hip: the store phase was 9 k of the workgroup's 62 k cycles).  Branch-free addresses: a lane's item
    // may belong to a different part than its neighbour's.
    constexpr int NV_KV = WA_NK * VEC, NITEMS = 2 * NV_KV + 64 * VEC, NSLOT = (NITEMS + NTHR - 1) / NTHR;
    constexpr int SPS = (NSLOT + NK3 - 2) / (NK3 - 1);            // slots per k-step (the last k-step only stores)
    const int gb = gm.b, gwy = gm.wy, gwx = gm.wx, gwi = gm.wi;
    const T* const vbase = &VOUT[0][0];
    const T* const kbase = &KOUT[0][0];
    const T* const qbase = &QOUT[0][0];
#define M2T_RES_ITEM(slot, SRC, DST)                                                                                              \
    do {                                                                                                                          \
      const int j__ = min(tid + (slot) * NTHR, NITEMS - 1);                                                                       \
      const int part__ = (j__ >= 2 * NV_KV) ? 2 : ((j__ >= NV_KV) ? 1 : 0);                                                       \
      const int jj__ = j__ - part__ * NV_KV;                                                                                      \
      const int row__ = jj__ / VEC, cv__ = jj__ - row__ * VEC;                                                                    \
      const T* sb__ = (part__ == 0) ? vbase : ((part__ == 1) ? kbase : qbase);                                                    \
      const int kr__ = row__ / 10, kc__ = row__ - kr__ * 10;                                                                      \
      const bool own__ = (part__ == 2) || (kr__ >= 1 && kr__ <= 8 && kc__ >= 1 && kc__ <= 8);                                     \
      const int py__ = (part__ == 2) ? (row__ >> 3) : kr__ - 1, px__ = (part__ == 2) ? (row__ & 7) : kc__ - 1;                    \
      const long long pixoff__ = (((long long)gb * h + 8 * gwy + py__) * w + 8 * gwx + px__) * (3 * C) +                          \
                                 ((part__ == 2) ? 0 : ((part__ == 1) ? C : 2 * C));                                               \
      const long long winoff__ = ((long long)gwi * WA_RING + ring_index(kr__, kc__)) * (2 * C) + ((part__ == 1) ? 0 : C);         \
      SRC = sb__ + row__ * LD + cv__ * 8;                                                                                         \
      DST = (own__ ? gqkv + pixoff__ : win + winoff__) + cv__ * 8;                                                                \
    } while (0)
    static_assert(SPS <= 2, "two staging registers");
    Frag8<T> stg0 = frag_zero<T>(), stg1 = frag_zero<T>();
    T* sdst0 = gqkv;
    T* sdst1 = gqkv;
    Frag8<T> bq[2][WA_KT];
    bfetch(0, bq[0]);
#pragma unroll
    for (int ks = 0; ks < NK3; ++ks) {
      Frag8<T> a[TPW];
#pragma unroll
      for (int m = 0; m < TPW; ++m) a[m] = wb[m][ks % DEPTH];
      if (ks + DEPTH < NK3) wfetch(ks + DEPTH, ks % DEPTH);
      if (ks >= 1) {                                              // the items read in the previous k-step
        const int sl0 = (ks - 1) * SPS, sl1 = sl0 + 1;
        if (sl0 < NSLOT && ((sl0 + 1) * NTHR <= NITEMS || tid + sl0 * NTHR < NITEMS)) store8(sdst0, stg0);
        if (SPS == 2 && sl1 < NSLOT && ((sl1 + 1) * NTHR <= NITEMS || tid + sl1 * NTHR < NITEMS)) store8(sdst1, stg1);
      }
      if (ks + 1 < NK3) bfetch(ks + 1, bq[(ks + 1) & 1]);
      if (ks + 1 < NK3) {
        const int sl0 = ks * SPS, sl1 = sl0 + 1;
        if (sl0 < NSLOT) { const T* sp; M2T_RES_ITEM(sl0, sp, sdst0); stg0 = load8(sp); }
        if (SPS == 2 && sl1 < NSLOT) { const T* sp; M2T_RES_ITEM(sl1, sp, sdst1); stg1 = load8(sp); }
      }
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetches HERE: hipcc otherwise sinks them to just before their use
#pragma unroll
      for (int t = 0; t < ((ks < NKC) ? 4 : WA_KT); ++t)
#pragma unroll
        for (int m = 0; m < TPW; ++m) mma16(ad[m][t], a[m], bq[ks & 1][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef M2T_RES_ITEM
    static_assert(SPS * (NK3 - 1) >= NSLOT, "every slot has a k-step");
  M2T_RES_STAMP(6);
    // lane (row 16 t + lr of the order above, g) holds channels 16 (mt0 + m) + 4 g .. + 3 of g_d for that key
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = krow[t];
      long long pix;
      if (key < WA_NK && gm.key_pixel(key, pix)) {                // (ring keys outside the image are dropped: the gradient of zero padding)
        T* dst = (t < 4) ? gd + pix * C : gdwin + ((long long)gm.wi * WA_RING + 16 * (t - 4) + lr) * C;
#pragma unroll
        for (int m = 0; m < TPW; ++m) {
          float v[4] = {ad[m][t][0], ad[m][t][1], ad[m][t][2], ad[m][t][3]};
          store4(dst + 16 * (mt0 + m) + 4 * g, v);
        }
      }
    }
  }
  if constexpr (!DG) {
  for (int idx = tid; idx < WA_NK * VEC; idx += NTHR) {
    const int key = idx / VEC, cv = idx % VEC;
    store8(dkv_row(gqkv, win, (long long)gm.wi, gm.b, gm.wy, gm.wx, h, w, C, key) + C + cv * 8, load8(&VOUT[key][cv * 8]));
  }
  __syncthreads();
  for (int idx = tid; idx < WA_NK * VEC; idx += NTHR) {
    const int key = idx / VEC, cv = idx % VEC;
    store8(dkv_row(gqkv, win, (long long)gm.wi, gm.b, gm.wy, gm.wx, h, w, C, key) + cv * 8, load8(&KOUT[key][cv * 8]));
  }
  for (int idx = tid; idx < 64 * VEC; idx += NTHR) {
    const int row = idx / VEC, cv = idx % VEC;
    store8(gqkv + gm.query_pixel(row) * (3 * C) + cv * 8, load8(&QOUT[row][cv * 8]));
  }
  }
  M2T_RES_STAMP(7);
}

// ---------------------------------------------------------------------------------------
// Resident FORWARD (bf16, C = 64 / 256): K^ and V of the whole window are loaded once (one exposed latency instead
// of 2 x C/64 chunk stages), q comes straight from HBM in operand layout, S^T / P^T never leave the registers
// (the accumulator layout of S^T is the B-operand layout of O^T = V^T P^T), and the branch epilogue
// xc = IWT^L(O) + xin is applied on the accumulators exactly as in window_attn_fwd_kernel.
// ---------------------------------------------------------------------------------------
template <int C, int L>
__global__ void __launch_bounds__(256) window_attn_fwd_res_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                  const float* __restrict__ rel_w, bf16_t* __restrict__ out, int ldo,
                                                                  int oc0, const bf16_t* __restrict__ res, int ldr, int h, int w) {
  using T = bf16_t;
  static_assert(L == 0 || C == (16 << (2 * L)), "fused IWT needs C = 16 * 4^L");
  constexpr int LD = C + 8, VEC = C / 8, NT = C / 16, NKC = C / 32, ZR = 100;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Kh)[LD] = reinterpret_cast<T(*)[LD]>(smem);
  T(*Vs)[LD] = reinterpret_cast<T(*)[LD]>(smem + sizeof(T) * 101 * LD);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const int q = 16 * wv + lr;
  const long long qpix = gm.query_pixel(q);
  {
    constexpr int KIT = (WA_NK * VEC + 255) / 256;
    Frag8<T> kf[KIT], vf[KIT];
    f32x4 r0[KIT], r1[KIT];
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx % VEC, key = idx / VEC;
      kf[it] = frag_zero<T>();
      vf[it] = frag_zero<T>();
      r0[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      r1[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (key < WA_NK) {
        long long pix;
        if (gm.key_pixel(key, pix)) {
          kf[it] = load8(qkv + pix * (3 * C) + C + cv * 8);
          vf[it] = load8(qkv + pix * (3 * C) + 2 * C + cv * 8);
        }
        const int kr = key / 10, kc = key - kr * 10;
        const int cc = cv * 8;
        const float* rp = (cc < C / 2) ? (rel_h + kr * (C / 2) + cc) : (rel_w + kc * (C / 2) + (cc - C / 2));
        r0[it] = *reinterpret_cast<const f32x4*>(rp);
        r1[it] = *reinterpret_cast<const f32x4*>(rp + 4);
      }
    }
#pragma unroll
    for (int it = 0; it < KIT; ++it) {
      const int idx = tid + it * 256;
      const int cv = idx % VEC, key = idx / VEC;
      if (key < WA_NK) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = kf[it].get(e) + r0[it][e]; v[4 + e] = kf[it].get(4 + e) + r1[it][e]; }
        store8f(&Kh[key][cv * 8], v);
        store8(&Vs[key][cv * 8], vf[it]);
      }
    }
    if (tid < VEC) {
      store8(&Kh[ZR][tid * 8], frag_zero<T>());
      store8(&Vs[ZR][tid * 8], frag_zero<T>());
    }
  }
  Frag8<T> qreg[NKC];
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc) qreg[kc] = load8(qkv + qpix * (3 * C) + kc * 32 + 8 * g);
  __syncthreads();
  // ---- S^T = K^ Q^T ----
  f32x4 s[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) mma16(s[t], load8(&Kh[min(16 * t + lr, ZR)][kc * 32 + 8 * g]), qreg[kc]);
  // ---- softmax over the 100 real keys; lane (q, g) holds keys 16 t + 4 g + r ----
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
  // P^T as the B operand: k-chunk c4 covers key tiles 2c4, 2c4+1; slot (g, j) <-> key 16(2c4 + (j>>2)) + 4g + (j&3)
  Frag8<T> pf[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 2 * c4 + (j >> 2);
      pf[c4].set(j, (t < WA_KT) ? s[t < WA_KT ? t : 0][j & 3] * inv : 0.f);
    }
  // ---- O^T = V^T P^T: V^T fragments by transposing LDS reads in the same key order (rows >= 100 alias the zero row) ----
  f32x4 o[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) mma16(o[mt], tr8(&Vs[0][0], LD, 32 * c4 + 4 * g, 32 * c4 + 16 + 4 * g, 16 * mt, lane, ZR), pf[c4]);
  if constexpr (L == 0) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      float v[4] = {o[mt][0], o[mt][1], o[mt][2], o[mt][3]};
      const int cc = 16 * mt + 4 * g;
      if (res) {
        float p[4];
        load4(res + qpix * ldr + cc, p);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p[e];
      }
      store4(out + qpix * ldo + oc0 + cc, v);
    }
  } else {
    // channel = band * 16 + base channel (band-major nesting of repeated DWTs): tile mt = band mt, row = base channel
    constexpr int S = Haar<L>::S, NB = Haar<L>::N;
    static_assert(NB == NT, "one 16-channel tile per band");
    float vv[4][S][S];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bands[NB];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) bands[mt] = o[mt][r];
      Haar<L>::inv(bands, vv[r]);
    }
    const int H = h * S, W = w * S;
    const int by = 8 * gm.wy + (q >> 3), bx = 8 * gm.wx + (q & 7);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        const long long pix = ((long long)gm.b * H + S * by + y) * W + S * bx + x;
        float p[4];
        load4(res + pix * ldr + 4 * g, p);
        float v[4] = {vv[0][y][x] + p[0], vv[1][y][x] + p[1], vv[2][y][x] + p[2], vv[3][y][x] + p[3]};
        store4(out + pix * ldo + oc0 + 4 * g, v);
      }
  }
}

template <int C, int L>
int go_fwd_res(const bf16_t* qkv, const float* rel_h, const float* rel_w, bf16_t* out, int ldo, int oc0, const bf16_t* res, int ldr,
               int nwin, int h, int w, hipStream_t st) {
  const size_t sh = sizeof(bf16_t) * 2 * 101 * (C + 8);
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_fwd_res_kernel<C, L>, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED((window_attn_fwd_res_kernel<C, L>), dim3(nwin), dim3(256), sh, st, qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, h, w);
  return 0;
}

template <int C, int L, int NW, bool DG = false, bool RC = false, bool PB = false>
int go_res(const bf16_t* qkv, const float* rel_h, const float* rel_w, const bf16_t* gout, int ldg, int gc0, bf16_t* gqkv,
           bf16_t* win, float* relw, int nwin, int h, int w, hipStream_t st, const bf16_t* wdfrag = nullptr, bf16_t* gd = nullptr,
           bf16_t* gdwin = nullptr, const bf16_t* xsrc = nullptr, const bf16_t* wfrag = nullptr, ResPrepArgs pb = ResPrepArgs{}) {
  const size_t sh = ResCfg<C>::total;
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_bwd_res_kernel<C, L, NW, DG, RC, PB>, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED((window_attn_bwd_res_kernel<C, L, NW, DG, RC, PB>), dim3(nwin), dim3(NW * 64), sh, st, qkv, rel_h, rel_w, gout, ldg, gc0,
                     gqkv, win, relw, h, w, wdfrag, gd, gdwin, xsrc, wfrag, pb);
  return 0;
}

}  // namespace

// bf16, C in {64, 256}, dwt_levels in {0, L(C)}; returns M2T_UNSUPPORTED otherwise (the caller then uses the chunked kernel)
int launch_window_attn_bwd_resident(const void* qkv_, const float* rel_h, const float* rel_w, const void* gout_, int ldg, int gc0,
                                    void* gqkv_, void* win_, float* relw, int B, int h, int w, int C, int dwt_levels, hipStream_t st,
                                    const void* wdfrag, void* gd, void* gdwin, const void* xsrc, const void* wfrag,
                                    const void* pb_gd, const void* pb_gdwin, const void* pb_gxk, void* pb_gnk) {
  const bf16_t* qkv = (const bf16_t*)qkv_;
  const bf16_t* gout = (const bf16_t*)gout_;
  bf16_t* gqkv = (bf16_t*)gqkv_;
  bf16_t* win = (bf16_t*)win_;
  const int nwin = B * (h / 8) * (w / 8);
  int rc = M2T_UNSUPPORTED;
  if (wdfrag) {      // fused projection data gradient: the two shapes of the model
    if (!gd || !gdwin) return m2t_set_error(-2, "window_attn_bwd_resident: fused data gradient needs gd and gdwin");
    if (C == 256 && dwt_levels == 2 && pb_gd) {
      // branch_prep_bwd of the next branch inside: gout is then READ AND WRITTEN (g_xc[i] += ...), a dense 16-channel plane
      if (!pb_gdwin || !pb_gxk || !pb_gnk || ldg != 16 || gc0 != 0 || pb_gd == gd || pb_gdwin == gdwin)
        return m2t_set_error(-2, "window_attn_bwd_resident: fused branch_prep_bwd needs its four operands, a dense g_xc plane and gd / gdwin buffers other than the ones it writes");
      ResPrepArgs pb;
      pb.gdn = (const bf16_t*)pb_gd; pb.gdwinn = (const bf16_t*)pb_gdwin; pb.gxk = (const bf16_t*)pb_gxk; pb.gnk = (bf16_t*)pb_gnk;
      rc = go_res<256, 2, 8, true, false, true>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st, (const bf16_t*)wdfrag, (bf16_t*)gd,
                                                (bf16_t*)gdwin, nullptr, nullptr, pb);
    } else if (C == 256 && dwt_levels == 2)
      rc = go_res<256, 2, 8, true>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st, (const bf16_t*)wdfrag, (bf16_t*)gd, (bf16_t*)gdwin);
    else if (C == 64 && dwt_levels == 1 && xsrc && wfrag)     // q | k | v not saved: recomputed from the branch input
      rc = go_res<64, 1, 4, true, true>(nullptr, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st, (const bf16_t*)wdfrag, (bf16_t*)gd,
                                        (bf16_t*)gdwin, (const bf16_t*)xsrc, (const bf16_t*)wfrag);
    else if (C == 64 && dwt_levels == 1)
      rc = go_res<64, 1, 4, true>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st, (const bf16_t*)wdfrag, (bf16_t*)gd, (bf16_t*)gdwin);
    if (rc != 0) return rc;
    M2T_LAUNCH_CHECK();
    return 0;
  }
  if (C == 256 && dwt_levels == 2) rc = go_res<256, 2, 8>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st);
  else if (C == 256 && dwt_levels == 0) rc = go_res<256, 0, 8>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st);
  else if (C == 64 && dwt_levels == 1) rc = go_res<64, 1, 4>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st);
  else if (C == 64 && dwt_levels == 0) rc = go_res<64, 0, 4>(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, nwin, h, w, st);
  if (rc != 0) return rc;
  M2T_LAUNCH_CHECK();
  return 0;
}

// forward: bf16, (C, post_levels) in {(64,0), (64,1), (256,0), (256,2)}; M2T_UNSUPPORTED otherwise
int launch_window_attn_fwd_resident(const void* qkv_, const float* rel_h, const float* rel_w, void* out_, int ldo, int oc0,
                                    const void* res_, int ldr, int B, int h, int w, int C, int post_levels, hipStream_t st) {
  const bf16_t* qkv = (const bf16_t*)qkv_;
  bf16_t* out = (bf16_t*)out_;
  const bf16_t* res = (const bf16_t*)res_;
  const int nwin = B * (h / 8) * (w / 8);
  int rc = M2T_UNSUPPORTED;
  if (C == 256 && post_levels == 2) rc = go_fwd_res<256, 2>(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, nwin, h, w, st);
  else if (C == 256 && post_levels == 0) rc = go_fwd_res<256, 0>(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, nwin, h, w, st);
  else if (C == 64 && post_levels == 1) rc = go_fwd_res<64, 1>(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, nwin, h, w, st);
  else if (C == 64 && post_levels == 0) rc = go_fwd_res<64, 0>(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, nwin, h, w, st);
  if (rc != 0) return rc;
  M2T_LAUNCH_CHECK();
  return 0;
}
