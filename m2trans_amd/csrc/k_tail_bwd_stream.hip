// k_tail_bwd_stream.hip -- backward of the tail's last stage as a row-streaming kernel (bf16; round 4).
//
//     t  = PixelShuffle(R)(W a + b),  act = gelu(t),  d = gelu'(t)    recomputed (the forward stores neither)
//     g(act) = conv3x3^T(g(sr))                      tail conv data gradient, reflect padding folded into Geff
//     dWf   += Geff^T act                             tail conv weight gradient
//     g(t)   = g(act) * d ;  g(u) = pixel_unshuffle(g(t))
//     g(a)   = g(u) W  (* gelu'(t1) for x4)           expansion data gradient
//     dW    += g(u)^T a ,  db += sum g(u)
//
// Two uses (the mirror of k_tail_stream.hip):
//   TAIL0 = false, R = 2: x4's tail.3 stage (models/M2Trans_network.py:45-48 under autograd): a = gelu(t1) and gelu'(t1), NHWC on the
//          mid-resolution map, g(t1) out.  Bit-identical data gradient to the tile kernel (k_tail_bwd.hip) but SLOWER (580 against
//          377 us at batch 16: two waves per SIMD cannot hide the dependent MFMA -> GELU -> MFMA chain of a wave): not dispatched,
//          the tile kernel stays the x4 path.
//   TAIL0 = true, R = 2 / 3: x2's / x3's whole tail (:52-55): a = the body output (P64 planes), g(body output) out (P64).  Replaces
//          final_conv_dgrad + final_conv_wgrad + gemm_nt + wgrad_tn, which read and wrote the 64-channel HR tensors gelu(t), gelu'(t)
//          and g(t) (x3 at 256x256 LR, batch 8: 1.6 ms of kernels -> one launch).
//
// Decomposition: a workgroup of R^2 waves owns a strip of 16 input pixels and walks down it one row per step; wave v owns sub-pixel
// position v = (sy, sx) of the shuffle, all 64 channels.  Phase A, per wave, with nothing but operands staged through LDS: input row
// -> 8 MFMAs (t; weight fragments out of W^T in LDS by transposing reads, rows permuted so that two accumulator tiles hold 8
// consecutive channels of the lane's pixel) -> bias + GELU and GELU' in registers -> the lane's eight Geff entries from a ring of
// g(sr) rows in LDS -> 4 MFMAs (g(act) in the accumulator layout of d) -> g(t) = g(act) d.  g(t), act and Geff go to LDS once,
// row-major [pixel][channel].  Phase B (after a barrier): every wave accumulates dW / db of its own sub-pixel position (K = 16 MFMAs
// against transposing LDS reads); the first eight tile owners accumulate dWf over all positions; waves 0 .. 3 compute one
// 16-channel tile of g(a) each.  Two barriers per step; workgroups are persistent and leave ONE slab of partial parameter
// gradients each.  The loop is unrolled by four (ring slots, the input double buffer compile-time); every load is unconditional
// (hipcc's vmcnt accounting, see k_tail_stream.hip).
#include <type_traits>
#include "m2t_kernels.h"

#ifndef BS_STAMP
#define BS_STAMP(i) do { } while (0)       // scratch/bench_tail.hip: s_memtime per phase
#endif

namespace {

constexpr int BS_LDG = 72;      // bf16 row stride of the G / A2 / A1 tiles (64 channels + 8)
constexpr int BS_LDE = 40;      // ... of the GE tiles and Wt (32 entries + 8)
constexpr int BS_RROW = 64;     // floats per row of the g(sr) ring (16 R + 2 used; the rest stays zero: padding k-slots read there)

template <int R> struct BSCfg {
  static constexpr int NW = R * R;                  // producer waves = sub-pixel positions
  static constexpr int NWT = (R == 2) ? 4 : 16;     // waves per workgroup; R = 3: 9 producers + 7 helper waves that own the dW accumulators
  static constexpr int NTHR = 64 * NWT;
  static constexpr int LDW3 = 64 * NW + 8;          // bf16 row stride of W^T [k][n']
  static constexpr int RING = 4 * R, RSLOT = RING + 2;   // ring rows per channel plane + 2 mirrors of rows 0, 1
  static constexpr int NCOL = 16 * R + 2;           // ring columns in use
  static constexpr int FT = (NW == 4) ? 2 : 1;      // dWf tiles per owning wave
  static constexpr size_t oWT = 0;
  static constexpr size_t oW3 = oWT + sizeof(bf16_t) * 64 * BS_LDE;
  static constexpr size_t oB = oW3 + sizeof(bf16_t) * 64 * LDW3;
  static constexpr size_t oR = oB + sizeof(float) * 4 * NW * 16;
  static constexpr size_t oG = oR + sizeof(float) * 3 * RSLOT * BS_RROW;
  static constexpr size_t oA2 = oG + sizeof(bf16_t) * NW * 16 * BS_LDG;
  static constexpr size_t oGE = oA2 + sizeof(bf16_t) * NW * 16 * BS_LDG;
  static constexpr size_t oA1 = oGE + sizeof(bf16_t) * NW * 16 * BS_LDE;
  static constexpr size_t total = oA1 + sizeof(bf16_t) * 16 * BS_LDG;
  static_assert(R == 2 || R == 3, "R = 2 or 3");
  static_assert(total <= 160 * 1024, "LDS budget");
  // (the helper-wave layout of R = 3 writes g(a) as P64 planes: it exists for TAIL0 only)
};

struct BSArgs {
  const float* gout;      // g(sr)  fp32 [B][3][H][W]
  const float* wf;        // tail conv weight fp32 [3][64][3][3]
  const bf16_t* a;        // expansion input: NHWC [B][Hi][Wi][64] (x4: gelu(t1)) or P64 planes (TAIL0: the body output)
  const bf16_t* d1;       // x4: gelu'(t1), NHWC
  const bf16_t* w3t;      // packed expansion weight^T [64 k][64 R^2 n'], n' = sub * 64 + c
  const float* b3;        // expansion bias fp32, torch order (c * R^2 + sub)
  bf16_t* ga;             // g(a): NHWC (x4: g(t1)) or P64 planes (TAIL0)
  float* slab_wf;         // [grid][32][64]          ((tap*3+oc) x ic)
  float* slab_w3;         // [grid][64 R^2][64]      (n' x k)
  float* slab_b3;         // [grid][64 R^2]
  int B, Hi, Wi;
  int nstrip, nseg, rows, ntask;      // rows per segment: a multiple of 4 that divides Hi
};

__device__ __forceinline__ void bs_mma4(f32x4& acc, bf16x4 a, bf16x4 b) { acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0); }
// transposing read: the lane passes the address of ITS 8-byte piece (row r0 + (i >> 2), columns c0 + 4 (i & 3) ..) and receives
// column c0 + i of rows r0 .. r0 + 3 (i = lane & 15)
__device__ __forceinline__ bf16x4 bs_tr4(const bf16_t* p) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
}

template <int R, bool TAIL0>
__global__ void __launch_bounds__(BSCfg<R>::NTHR, (R == 2 ? 2 : 4)) tail_bwd_stream_kernel(BSArgs p) {
  using T = bf16_t;
  using Cfg = BSCfg<R>;
  constexpr int NW = Cfg::NW, NWT = Cfg::NWT, NTHR = Cfg::NTHR, LDW3 = Cfg::LDW3, RSLOT = Cfg::RSLOT, RING = Cfg::RING, NCOL = Cfg::NCOL, FT = Cfg::FT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Wt)[BS_LDE] = reinterpret_cast<T(*)[BS_LDE]>(smem + Cfg::oWT);                         // [ic][(tap,oc) -> 32]
  T(*W3s)[LDW3] = reinterpret_cast<T(*)[LDW3]>(smem + Cfg::oW3);                            // W^T [k][n']
  f32x4* Bs = reinterpret_cast<f32x4*>(smem + Cfg::oB);                                     // bias [(wave * 4 + 2 kc + half) * 4 + g]
  float* Gr = reinterpret_cast<float*>(smem + Cfg::oR);                                     // g(sr) ring [oc][slot][col]
  T(*G)[16][BS_LDG] = reinterpret_cast<T(*)[16][BS_LDG]>(smem + Cfg::oG);                   // g(t)   [sub][pixel][c]
  T(*A2)[16][BS_LDG] = reinterpret_cast<T(*)[16][BS_LDG]>(smem + Cfg::oA2);                 // act    [sub][pixel][c]
  T(*GE)[16][BS_LDE] = reinterpret_cast<T(*)[16][BS_LDE]>(smem + Cfg::oGE);                 // Geff   [sub][pixel][(tap,oc)]
  T(*A1)[BS_LDG] = reinterpret_cast<T(*)[BS_LDG]>(smem + Cfg::oA1);                         // input  [pixel][k]
  // (wv through readfirstlane: the wave index, the sub-pixel position and everything derived from them are wave-uniform and live in
  //  SGPRs -- the address arithmetic they feed is scalar and does not compete with the accumulators for the 128 / 256 vector registers)
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  // R = 2: every wave is a producer and owns the dW / db tiles of its sub-pixel position (88 accumulator registers, 256 per wave,
  // two workgroups per CU).  R = 3: nine producers cannot carry 16 dW tiles each at three waves per SIMD (168 registers), so the
  // workgroup has 16 waves (128 registers): waves 0 .. 8 produce and keep db (+ dWf, g(a)), waves 9 .. 15 own the 144 dW tiles
  constexpr bool HELP = NWT > NW;
  const bool prod = wv < NW;
  const int sub = prod ? wv : 0;
  const int sy = sub / R, sx = sub - sy * R;
  const int Hi = p.Hi, Wi = p.Wi, H = R * Hi, W = R * Wi;
  const long long hw = (long long)H * W, npix = (long long)p.B * Hi * Wi;
  const int tq = lr >> 2, tp = lr & 3;                              // transposing-read lane geometry

  // ---- constants: Wt, W^T, bias (LDS), a zeroed ring ----
  for (int i = tid; i < 64 * 32; i += NTHR) {
    const int ic = i >> 5, n = i & 31;
    float v = 0.f;
    if (n < 27) v = p.wf[((n % 3) * 64 + ic) * 9 + n / 3];
    Wt[ic][n] = from_f<T>(v);
  }
  for (int i = tid; i < 64 * 8 * NW; i += NTHR) {
    const int k = i / (8 * NW), cv = i - k * (8 * NW);
    store8(&W3s[k][cv * 8], load8(p.w3t + (long long)k * (64 * NW) + cv * 8));
  }
  for (int i = tid; i < NW * 64; i += NTHR) {
    const int w = i >> 6, q = (i >> 2) & 15, r = i & 3;              // q = (2 kc + half) * 4 + g
    const int kc = q >> 3, hf = (q >> 2) & 1, gg = q & 3;
    reinterpret_cast<float*>(Bs)[i] = p.b3[(32 * kc + 8 * gg + 4 * hf + r) * NW + w];
  }
  for (int i = tid; i < 3 * RSLOT * BS_RROW; i += NTHR) Gr[i] = 0.f;
  // parameter-gradient accumulators of the whole launch.  accW: R = 2: [tau][kappa] of the wave's own position; R = 3 (helper h =
  // wv - 9): n' tiles T = h, h + 7, .. (T = 4 sub + tau), up to six of them
  // (R = 3: the helper waves run their OWN loop below -- separate code, so that the producers' registers do not carry the helpers'
  //  96 accumulator registers and vice versa; both loops execute the same barriers)
  if constexpr (HELP) {
    if (!prod) {
      constexpr int NTW = 6;
      f32x4 aW[NTW][4];
#pragma unroll
      for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) aW[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int h = wv - NW;
      __syncthreads();
      for (int task = xcd_block_index(); task < p.ntask; task += gridDim.x) {
        const int strip = task % p.nstrip, seg = (task / p.nstrip) % p.nseg, b = task / (p.nstrip * p.nseg);
        const long long pixt = ((long long)b * Hi + p.rows * seg) * Wi + 16 * strip;         // wave-uniform: the lane's pixel is added as a 32-bit offset
        lds_barrier();                                            // (the producers' prologue barrier)
        for (int s = 0; s < p.rows; ++s) {
          lds_barrier();                                          // phase A of the producers is complete
          bf16x4 bk[4];
#pragma unroll
          for (int kp = 0; kp < 4; ++kp) bk[kp] = bs_tr4(&A1[4 * g + tq][16 * kp + 4 * tp]);
#pragma unroll
          for (int i = 0; i < NTW; ++i) {
            const int Tn = h + 7 * i;                             // n' tile: position Tn >> 2, channel tile Tn & 3
            if (Tn < 4 * NW) {
              const bf16x4 an = bs_tr4(&G[Tn >> 2][4 * g + tq][16 * (Tn & 3) + 4 * tp]);
#pragma unroll
              for (int kp = 0; kp < 4; ++kp) bs_mma4(aW[i][kp], an, bk[kp]);
            }
          }
          if (h < 4) {
            // g(a) of the row, channel tile h: contraction over the 64 R^2 values g(u) of a pixel (the helpers have the time: the
            // producers' waves 0 .. 3 would add it to their dWf share)
            f32x4 accD = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int kc8 = 0; kc8 < 2 * NW; ++kc8)
              mma16(accD, load8(&W3s[16 * h + lr][32 * kc8 + 8 * g]), load8(&G[kc8 >> 1][lr][32 * (kc8 & 1) + 8 * g]));
            float v[4] = {accD[0], accD[1], accD[2], accD[3]};
            int lo = 16 * lr + 4 * g;
            asm volatile("" : "+v"(lo));                          // (or hipcc keeps p.ga + lo as a 64-bit pair across the loop -- and spills it)
            store4(p.ga + ((long long)h * npix + pixt + (long long)s * Wi) * 16 + lo, v);          // TAIL0: P64, plane = channel tile
          }
          lds_barrier();
        }
      }
      float* out = p.slab_w3 + (long long)blockIdx.x * (64 * NW * 64);
      int ln2 = lane;
      asm volatile("" : "+v"(ln2));                               // (the slab addresses are computed here, not kept across the loop)
      const int lr2 = ln2 & 15, g2 = ln2 >> 4;
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int Tn = h + 7 * i;
        if (Tn < 4 * NW) {
#pragma unroll
          for (int kp = 0; kp < 4; ++kp)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)(16 * Tn + 4 * g2 + r) * 64 + 16 * kp + lr2] = aW[i][kp][r];
        }
      }
      return;
    }
  }
  // (producer-only values from here on: computed behind the helpers' branch so that none of them is live across it)
  bf16x4 ones4;
#pragma unroll
  for (int e = 0; e < 4; ++e) ones4[e] = (T)1.0f;
  // this lane's eight Geff sources: k-slot 8 g + j = (tap, oc) = (n / 3, n % 3); pixel row R r + sy, column R (ci + lr) + sx;
  // source row R r + sy - ky + 1 = window row sy - ky + 2 of the step's R + 2 ring rows, ring column R lr + sx - kx + 2.
  // k-slots 27 .. 31 read a column the ring never writes (zero)
  int gsrc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = 8 * g + j;
    const int tap = n / 3, oc = n - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
    gsrc[j] = (n < 27) ? ((oc * RSLOT + (sy - ky + 2)) * BS_RROW + R * lr + sx - kx + 2) : 60;
  }
  constexpr int NTW = HELP ? 1 : 4;
  f32x4 accW[NTW][4], accB[4], accF[FT];
#pragma unroll
  for (int a = 0; a < NTW; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) accW[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a) accB[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < FT; ++a) accF[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // staging geometry of the ring: thread -> (oc, one of the R new rows, column)
  constexpr int ST_ROW = R * NCOL;                                   // values per channel and step
  const int st_oc = min(tid / ST_ROW, 2), st_rem = tid % ST_ROW, st_rs = st_rem / NCOL, st_col = st_rem % NCOL;
  const bool st_on = tid < 3 * ST_ROW;
  constexpr int NPT = 64 * NW;                                       // producer threads (the helper waves of R = 3 stage nothing)
  static_assert(3 * ST_ROW <= NPT && 3 * (R + 2) * NCOL <= 2 * NPT, "ring staging: one value per producer thread and step, two in the prologue");

  for (int task = xcd_block_index(); task < p.ntask; task += gridDim.x) {
    const int strip = task % p.nstrip;
    const int seg = (task / p.nstrip) % p.nseg;
    const int b = task / (p.nstrip * p.nseg);
    const int ci = 16 * strip;
    const int ri = p.rows * seg;                                  // (the launcher guarantees Hi % rows == 0)
    const float* gbase = p.gout + (long long)b * 3 * hw;
    // ring row fetch: HR row y, ring column col (image column R ci - 1 + col), channel oc; 0 outside the image
    auto g_fetch = [&](int oc, int y, int col) -> float {
      const int x = R * ci - 1 + col;
      return gbase[(long long)oc * hw + (long long)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)];
    };
    auto g_inside = [&](int y, int col) -> bool {
      const int x = R * ci - 1 + col;
      return y >= 0 && y < H && x >= 0 && x < W;
    };
    // ---- task prologue: the first R + 2 ring rows (HR rows R ri - 1 .. -> slots 0 .., mirrors of slots 0, 1), first input row ----
    {
      constexpr int PN = 3 * (R + 2) * NCOL;
      float v[2];
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int i = min(tid + it * NPT, PN - 1);
        const int oc = i / ((R + 2) * NCOL), rem = i - oc * ((R + 2) * NCOL), t = rem / NCOL, col = rem - t * NCOL;
        v[it] = g_fetch(oc, R * ri - 1 + t, col);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int i = tid + it * NPT;
        if (i < PN) {
          const int oc = i / ((R + 2) * NCOL), rem = i - oc * ((R + 2) * NCOL), t = rem / NCOL, col = rem - t * NCOL;
          const float x = g_inside(R * ri - 1 + t, col) ? v[it] : 0.f;
          Gr[(oc * RSLOT + t) * BS_RROW + col] = x;
          if (t < 2) Gr[(oc * RSLOT + RING + t) * BS_RROW + col] = x;
        }
      }
    }
    const T* abase[2];
    const long long pix0 = ((long long)b * Hi + ri) * Wi + ci + lr;
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      const int c0 = 32 * ki + 8 * g;
      abase[ki] = TAIL0 ? p.a + ((long long)(c0 >> 4) * npix + pix0) * 16 + (c0 & 15) : p.a + pix0 * 64 + c0;
    }
    const int smax = Hi - 1 - ri;
    const int rstride = TAIL0 ? Wi * 16 : Wi * 64;
    auto a_load = [&](int s, Frag8<T> (&f)[2]) {
      const long long off = (long long)min(s, smax) * rstride;
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) f[ki] = load8(abase[ki] + off);
    };
    Frag8<T> fa[2], fb[2];
    a_load(0, fa);
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) asm volatile("" : "+v"(fa[ki].v));
    lds_barrier();

    // one step: input row r = ri + s, unrolled copy CC = s % 4
    auto step = [&](int s, auto cc_tag, Frag8<T> (&cur)[2], Frag8<T> (&nxt)[2]) {
      constexpr int CC = decltype(cc_tag)::value;
      const int r = ri + s;
      // R = 3 runs at 128 registers: left alone, hipcc hoists every lane-dependent LDS address of the four unrolled steps out of the
      // row loop (~100 values) and spills 70 of them -- and scratch reloads are VMEM operations that retire in order with the row
      // prefetches.  A laundered lane index per step keeps the address arithmetic inside the step (a few dozen integer operations)
      int ln_ = lane;
      if constexpr (R == 3) asm volatile("" : "+v"(ln_));
      const int lr = ln_ & 15, g = ln_ >> 4, tq = lr >> 2, tp = lr & 3;
      // ---- phase A ----
      BS_STAMP(0);
      a_load(s + 1, nxt);
      // the ring rows the NEXT step adds (HR rows R r + R + 1 .. R r + 2 R): fetched now, written in phase B
      const float gq = g_fetch(st_oc, R * r + R + 1 + st_rs, st_col);
      // x4: gelu'(t1) of this row, channel tile wv: waves 0 .. 3 compute one 16-channel tile of g(a) each in phase B
      bf16x4 rd1 = {(T)0.f, (T)0.f, (T)0.f, (T)0.f};
      if constexpr (!TAIL0) rd1 = *reinterpret_cast<const bf16x4*>(p.d1 + (((long long)b * Hi + min(r, Hi - 1)) * Wi + ci + lr) * 64 + 16 * (wv & 3) + 4 * g);
      if (wv == 0) {                                             // the input row, row-major, for the transposed operand of dW
        store8(&A1[lr][8 * g], cur[0]);
        store8(&A1[lr][32 + 8 * g], cur[1]);
      }
      BS_STAMP(1);
      // Geff of the lane's pixel: eight ring reads (fast path), or the generic gather with the reflect folds on the image border
      Frag8<T> geff;
      {
        float ge[8];
        const bool border = r == 0 || r == Hi - 1 || ci == 0 || ci + 16 == Wi;
        // the eight ring reads are issued unconditionally (they overlap the MFMAs); the rare border step then overrides them
#pragma unroll
        for (int j = 0; j < 8; ++j) ge[j] = Gr[gsrc[j] + R * CC * BS_RROW];
        if (border) {
          const int yy = R * r + sy, xx = R * (ci + lr) + sx;
          const int ey = (yy == 1) ? -1 : ((yy == H - 2) ? H : yy);
          const int ex = (xx == 1) ? -1 : ((xx == W - 2) ? W : xx);
          auto ring = [&](int oc, int y, int x) -> float {      // y in the step's R + 2 rows, x in the strip's columns
            return Gr[(oc * RSLOT + (y - (R * r - 1)) + R * CC) * BS_RROW + (x - (R * ci - 1))];
          };
          if constexpr (R == 3) {
            // Round 6: the reflect terms without divergent loops.  Pixel row 1 also receives, through tap ky = 0, the gradient of output row 0
            // (row H - 2 through ky = 2 that of row H - 1; columns likewise): in ring coordinates the mirrored source of tap (ky, kx) is the
            // entry the OPPOSITE tap reads (row yy + ky - 1, column xx + kx - 1), always inside the step's rows / the strip's columns, so the
            // three extra reads per value are issued unconditionally and selected afterwards.  (Rounds 4-5 walked them in a `combo` loop with
            // one serialized, divergent LDS read per branch -- every step of the image's first and last strip, 12.5 % of the x3 tasks.)
            const bool r0 = yy == 1, r2 = yy == H - 2, c0 = xx == 1, c2 = xx == W - 2;
            // (two values at a time, pinned: the R = 3 kernel runs at 128 registers)
  #pragma unroll
            for (int j2 = 0; j2 < 8; j2 += 2) {
              float vb[2], vr[2], vc[2], vk[2];
              int kyv[2], kxv[2];
  #pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int n = min(8 * g + j2 + u, 26);
                const int tap = n / 3, oc = n - 3 * tap, ky = tap / 3, kx = tap - 3 * ky;
                kyv[u] = ky; kxv[u] = kx;
                vb[u] = ring(oc, yy - ky + 1, xx - kx + 1);
                vr[u] = ring(oc, yy + ky - 1, xx - kx + 1);
                vc[u] = ring(oc, yy - ky + 1, xx + kx - 1);
                vk[u] = ring(oc, yy + ky - 1, xx + kx - 1);
              }
              __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
              for (int u = 0; u < 2; ++u) {
                const bool rm = (kyv[u] == 0 && r0) || (kyv[u] == 2 && r2);
                const bool cm = (kxv[u] == 0 && c0) || (kxv[u] == 2 && c2);
                float v = vb[u];
                v += rm ? vr[u] : 0.f;              // (the order the loop added them in: mirrored row, mirrored column, both)
                v += cm ? vc[u] : 0.f;
                v += (rm && cm) ? vk[u] : 0.f;
                ge[j2 + u] = (8 * g + j2 + u < 27) ? v : 0.f;
              }
              __builtin_amdgcn_sched_barrier(0);
            }
        
          } else {
            // (R = 2 -- x2, and the x4 A/B variant -- sits at the 256-register limit: the walk of rounds 4-5)
  #pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int n = 8 * g + j;
              const int tap = n / 3, oc = min(n - 3 * tap, 2), ky = min(tap / 3, 2), kx = tap - 3 * (tap / 3);
              float v = 0.f;
              if (n < 27) {
                v = ring(oc, yy - ky + 1, xx - kx + 1);
  #pragma unroll 1
                for (int combo = 1; combo < 4; ++combo) {
                  if ((combo & 1) && ey == yy) continue;
                  if ((combo & 2) && ex == xx) continue;
                  const int oy = ((combo & 1) ? ey : yy) - ky + 1, oxx = ((combo & 2) ? ex : xx) - kx + 1;
                  if (oy >= 0 && oy < H && oxx >= 0 && oxx < W) v += ring(oc, oy, oxx);
                }
              }
              ge[j] = v;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) geff.set(j, ge[j]);
      }
      store8(&GE[wv][lr][8 * g], geff);
      BS_STAMP(2);
      // bias + GELU, GELU'; g(act) = Wf^T Geff in the same accumulator layout; g(t) = g(act) d with d rounded to bf16 first (the
      // stored-activation path keeps gelu'(t) in bf16: same bits).  Four values at a time: registers
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        // t of the two 16-channel tiles of this half (kept per half: the four accumulator tiles at once cost 8 more registers)
        f32x4 acc[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          acc[hf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ki = 0; ki < 2; ++ki) {
            // A fragment W[(wv, channel 32 kc + 8 (i >> 2) + 4 hf + (i & 3))][32 ki + 8 g ..] out of W^T by two transposing reads: lane
            // (q, pp) fetches row 32 ki + 8 g (+ 4) + q, columns 32 kc + 8 pp + 4 hf .. + 3, whose four values go to lanes 4 pp + 0 .. 3
            // (32 registers of resident fragments do not fit beside the accumulator registers)
            const T* wp0 = &W3s[32 * ki + 8 * g + tq][64 * wv + 32 * kc + 8 * tp + 4 * hf];
            Frag8<T> wa;
            wa.v = __builtin_shufflevector(bs_tr4(wp0), bs_tr4(wp0 + 4 * LDW3), 0, 1, 2, 3, 4, 5, 6, 7);
            mma16(acc[hf], wa, cur[ki]);
          }
        }
        Frag8<T> fg, f2;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          float a2v[4], d2v[4];
          const f32x4 bias = Bs[(wv * 4 + 2 * kc + hf) * 4 + g];
          const float b4[4] = {bias[0], bias[1], bias[2], bias[3]};
          gelu_tail_both4<T>(acc[hf], b4, a2v, d2v);
          f32x4 ga = (f32x4){0.f, 0.f, 0.f, 0.f};
          const int chl = 32 * kc + 8 * (lr >> 2) + 4 * hf + (lr & 3);
          mma16(ga, load8(&Wt[chl][8 * g]), geff);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            fg.set(4 * hf + e, ga[e] * (float)(T)d2v[e]);
            f2.set(4 * hf + e, a2v[e]);
          }
        }
        store8(&G[wv][lr][32 * kc + 8 * g], fg);
        store8(&A2[wv][lr][32 * kc + 8 * g], f2);
      }
      BS_STAMP(3);
      lds_barrier();
      BS_STAMP(4);
      // ---- phase B ----
      // the next step's new ring rows -> slots R CC + R + 2 + j (mod 4 R; slots 0, 1 are mirrored behind the ring)
      if (st_on) {
        const int slot = (R * CC + R + 2 + st_rs) % RING;
        const float x = g_inside(R * r + R + 1 + st_rs, st_col) ? gq : 0.f;
        Gr[(st_oc * RSLOT + slot) * BS_RROW + st_col] = x;
        if (slot < 2) Gr[(st_oc * RSLOT + RING + slot) * BS_RROW + st_col] = x;
      }
      // dW / db: contraction over the 16 pixels of a position, n' tile (channels 16 tau ..) x k tile kappa
      if constexpr (!HELP) {
        bf16x4 bk[4];
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) bk[kp] = bs_tr4(&A1[4 * g + tq][16 * kp + 4 * tp]);
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) {
          const bf16x4 an = bs_tr4(&G[wv][4 * g + tq][16 * tau + 4 * tp]);
#pragma unroll
          for (int kp = 0; kp < 4; ++kp) bs_mma4(accW[tau][kp], an, bk[kp]);
          bs_mma4(accB[tau], an, ones4);
        }
      } else {
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) bs_mma4(accB[tau], bs_tr4(&G[wv][4 * g + tq][16 * tau + 4 * tp]), ones4);
      }
      BS_STAMP(5);
      // dWf: wave w owns ((tap, oc) tile, channel tile) = R = 2: (0, w), (1, w); R = 3: (w >> 2, w & 3) for w < 8.  Contraction over
      // the 16 R^2 HR pixels of the step
      if (NW == 4 || wv < 8) {
        const int ct = (NW == 4) ? wv : (wv & 3);
        constexpr int UNR_F = (NW == 4) ? 4 : 1;
#pragma unroll UNR_F
        for (int sb = 0; sb < NW; ++sb) {                         // (R = 3: rolled -- 27 hoisted LDS reads do not fit 128 registers)
          const bf16x4 bc = bs_tr4(&A2[sb][4 * g + tq][16 * ct + 4 * tp]);
#pragma unroll
          for (int f = 0; f < FT; ++f) {
            const int nt = (NW == 4) ? f : (wv >> 2);
            bs_mma4(accF[f], bs_tr4(&GE[sb][4 * g + tq][16 * nt + 4 * tp]), bc);
          }
        }
      }
      BS_STAMP(6);
      // g(a) of the row, channel tile wv (waves 0 .. 3): sub-pixel position major, then the two channel halves (the tile kernel's order)
      if (!HELP && wv < 4) {
        f32x4 accD = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int UNR_D = (NW == 4) ? 8 : 2;
#pragma unroll UNR_D
        for (int kc8 = 0; kc8 < 2 * NW; ++kc8)
          mma16(accD, load8(&W3s[16 * wv + lr][32 * kc8 + 8 * g]), load8(&G[kc8 >> 1][lr][32 * (kc8 & 1) + 8 * g]));
        float v[4];
        const long long pix = ((long long)b * Hi + r) * Wi + ci + lr;
        if constexpr (TAIL0) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = accD[e];
          store4(p.ga + ((long long)wv * npix + pix) * 16 + 4 * g, v);          // P64: plane = channel tile
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = accD[e] * (float)rd1[e];
          store4(p.ga + pix * 64 + 16 * wv + 4 * g, v);
        }
      }
      BS_STAMP(7);
      lds_barrier();
      BS_STAMP(8);
    };
    for (int s = 0; s < p.rows; s += 4) {
      step(s, std::integral_constant<int, 0>{}, fa, fb);
      step(s + 1, std::integral_constant<int, 1>{}, fb, fa);
      step(s + 2, std::integral_constant<int, 2>{}, fa, fb);
      step(s + 3, std::integral_constant<int, 3>{}, fb, fa);
    }
  }

  // ---- slabs ----
  if (NW == 4 || wv < 8) {
    float* out = p.slab_wf + (long long)blockIdx.x * (32 * 64);
    const int ct = (NW == 4) ? wv : (wv & 3);
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const int nt = (NW == 4) ? f : (wv >> 2);
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * nt + 4 * g + r) * 64 + 16 * ct + lr] = accF[f][r];
    }
  }
  {
    float* out = p.slab_w3 + (long long)blockIdx.x * (64 * NW * 64);
    float* outb = p.slab_b3 + (long long)blockIdx.x * (64 * NW);
    if constexpr (!HELP) {
#pragma unroll
      for (int tau = 0; tau < 4; ++tau)
#pragma unroll
        for (int kp = 0; kp < 4; ++kp)
#pragma unroll
          for (int r = 0; r < 4; ++r) out[(long long)(64 * wv + 16 * tau + 4 * g + r) * 64 + 16 * kp + lr] = accW[tau][kp][r];
    }
    if (lr == 0) {
#pragma unroll
      for (int tau = 0; tau < 4; ++tau)
#pragma unroll
        for (int r = 0; r < 4; ++r) outb[64 * wv + 16 * tau + 4 * g + r] = accB[tau][r];
    }
  }
}

int bs_rows(int Hi) { return Hi % 32 == 0 ? 32 : (Hi % 16 == 0 ? 16 : (Hi % 8 == 0 ? 8 : 4)); }      // rows per segment: divides Hi, multiple of 4

}  // namespace

// Hi, Wi: size of the expansion's input map; R: shuffle factor
int tail_bwd_stream_blocks(int B, int Hi, int Wi, int R) {
  const int rows = bs_rows(Hi);
  const long long ntask = (long long)B * (Wi / 16) * (Hi / rows);
  return (int)std::min<long long>(R == 2 ? 512 : 256, ntask);
}

// bf16.  tail0 = 0 (R = 2 only): x4's tail.3 stage, a / d1 / ga NHWC [B][Hi][Wi][64].  tail0 = 1 (R = 2, 3): x2's / x3's tail, a and ga
// P64 planes of the LR map, d1 unused.  w3t: packed expansion weight^T [64][64 R^2]; b3: its bias (torch order); wf: tail conv weight.
// Hi % 4 == 0, Wi % 16 == 0.  Slabs: tail_bwd_stream_blocks() of each kind ([32][64], [64 R^2][64], [64 R^2]).
int launch_tail_bwd_stream(const float* gout, const float* wf, const void* a, const void* d1, const void* w3t, const float* b3, void* ga,
                           float* slab_wf, float* slab_w3, float* slab_b3, int* nslab_out, int B, int Hi, int Wi, int R, int tail0,
                           hipStream_t st) {
  if (Hi % 4 || Wi % 16 || Hi < 4) return m2t_set_error(-2, "tail_bwd_stream: Hi % 4 or Wi % 16");
  if (!((R == 2) || (R == 3 && tail0))) return m2t_set_error(M2T_UNSUPPORTED, "tail_bwd_stream: R = 2, or R = 3 with tail0");
  if (!tail0 && !d1) return m2t_set_error(-2, "tail_bwd_stream: the x4 stage needs gelu'(t1)");
  BSArgs p{gout, wf, (const bf16_t*)a, (const bf16_t*)d1, (const bf16_t*)w3t, b3, (bf16_t*)ga, slab_wf, slab_w3, slab_b3, B, Hi, Wi, 0, 0, bs_rows(Hi), 0};
  p.nstrip = Wi / 16;
  p.nseg = Hi / p.rows;
  p.ntask = B * p.nstrip * p.nseg;
  const int nblk = tail_bwd_stream_blocks(B, Hi, Wi, R);
#define BS_GO(RR, TT)                                                                                                   \
  do {                                                                                                                  \
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_bwd_stream_kernel<RR, TT>, (int)BSCfg<RR>::total)) return rc__; \
    M2T_LAUNCH_TIMED((tail_bwd_stream_kernel<RR, TT>), dim3(nblk), dim3(BSCfg<RR>::NTHR), BSCfg<RR>::total, st, p);        \
  } while (0)
  if (R == 2 && !tail0) BS_GO(2, false);
  else if (R == 2) BS_GO(2, true);
  else BS_GO(3, true);
#undef BS_GO
  M2T_LAUNCH_CHECK();
  *nslab_out = nblk;
  return 0;
}
