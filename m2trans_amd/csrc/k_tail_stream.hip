// k_tail_stream.hip -- the tail's last stage, FORWARD, as a row-streaming kernel (bf16; round 4):
//
//     t   = PixelShuffle(R)(conv1x1 64 -> 64 R^2 (a) + bias)      (models/M2Trans_network.py:45-46 for x4: R = 2 on the mid-resolution
//     act = gelu(t)                                                 map; :52-53 for x2 / x3: R = scale on the LR map)
//     sr' = conv3x3 64 -> 3, reflect padding, no bias (act)        (:48 / :55)
//
// It replaces tail_fwd_fused_kernel (k_tail_fwd.hip: 16x16 output tiles, the 64-channel activation block staged in LDS, five
// workgroup barriers per tile, 1.56x halo recomputation of the GELU: 263 us at batch 16) and, for x2 / x3, the pair
// tail_expand_kernel + final_conv_fwd_kernel (which wrote and re-read the 64-channel HR activation).
//
// Decomposition.  A workgroup owns a strip of 16 input pixels (16 R output columns) x `rows` input rows and walks down it one input
// row per step.  Wave v of its R^2 waves owns sub-pixel position v = (sy, sx) of the shuffle, all 64 channels:
//   * the step's 16 input pixels are this wave's MFMA B operand straight from global memory (a lane's 8 consecutive channels of its
//     pixel = one 16-byte load; the R^2 waves of the workgroup read the same 2 KB, L1 hits), prefetched two steps ahead;
//   * t^T [channel][pixel] = W a^T: 8 MFMAs against the wave's 8 weight fragments (32 registers, loaded once).  The ROWS of the four
//     16-channel tiles are permuted -- row 4 g + r of tile (kc, half) is channel 32 kc + 8 g + 4 half + r -- so that a lane ends up
//     with channels 32 kc + 8 g + 0 .. 7 of its pixel in the two tiles of kc: after bias + GELU + rounding that IS the B operand
//     of the tail conv's products (8 consecutive k per lane).  The activation never touches LDS;
//   * Y^T [(tap, oc)][pixel] = Wf act^T: 6 MFMAs.  The rows of Wf are ordered so that accumulator tile nt of lane (pixel, g) is
//     the float4 (oc 0, oc 1, oc 2, 0) of tap 4 nt + g (third tile: tap 8 in the g = 0 lanes): one 16-byte LDS store per tile into
//     a ring of Y rows [slot 3][sub-row R][tap 9][pixel] (fp32);
//   * ONE workgroup barrier per step; then ONE wave (R = 3: three), rotating, sums the nine taps of the R output rows whose three Y
//     rows are complete -- nine 16-byte LDS reads and 27 additions per output pixel, in the tap order of final_conv_fwd_kernel --
//     and stores them, while the other waves start the next step (the ring's third slot makes that safe with one barrier).
// What bounds it is instruction ISSUE (measured with s_memtime and knock-out builds, scratch/bench_tail.hip): four waves per SIMD
// issue one instruction per ~2.8 cycles in total, so the step costs its instruction count -- the GELU's 2 v_med3 + 2 v_exp +
// 2 v_rcp + 7 packed fp32 operations per pair of values first of all.  The GELU is evaluated once per activation of the strip:
// the halo is one input column left and right of 14-15 owned ones and one input row above and below a segment, x1.1 instead of
// the tile kernel's x1.56.  Same operand fragments, k order, bias add, GELU and tap order as the kernels it replaces: identical bits.
//
// Two rules of hipcc's s_waitcnt insertion shaped the loop (each cost 30 % when broken): (1) it counts vmcnt over straight-line
// code only -- with the prefetch under `if (s + 2 < rows)` it fell back to vmcnt(0) in front of the first MFMA of every step and so
// waited for the prefetch it had just issued: every load of the loop is unconditional (clamped addresses), the loop is unrolled by
// three with the buffers rotating by NAME (a register copy of a just-loaded fragment waits for it) and `rows` is a multiple of 3;
// (2) the loop header merges the prologue's pending-load state with the back edge's: everything the prologue loads is pinned as
// arrived (an empty asm with a "+v" constraint) before the loop.
#include <type_traits>
#include "m2t_kernels.h"

namespace {

constexpr int TS_LDW = 72;                       // bf16 row stride of the tail conv weight image in LDS

template <int R> struct TSCfg {
  static constexpr int NW = R * R;               // waves = sub-pixel positions
  static constexpr int NPX = 16 * R;             // output pixels per Y row
  static constexpr int RPW = 64 / NPX;           // output rows one consumer wave sums (R = 2: 2, R = 3: 1)
  static constexpr int UPS = R / RPW;            // consumer waves per step (R = 2: 1, R = 3: 3)
  static constexpr size_t szW = sizeof(bf16_t) * 48 * TS_LDW;
  static constexpr size_t szB = sizeof(float) * 4 * NW * 16;               // bias [wave][kc][half][g] float4
  static constexpr size_t szY = sizeof(float) * 4 * 3 * R * 9 * NPX;       // [slot 3][sub-row R][tap 9][pixel] float4
  static constexpr size_t total = szW + szB + szY;
  static_assert(R == 2 || R == 3, "R = 2 or 3");
};

struct TSArgs {
  const bf16_t* a;        // input [B][Hi][Wi][64] (NHWC) or P64 planes
  const bf16_t* wp;       // packed expansion weight rows [sub * 64 + c][64]   (M2T_PACK_SHUF_ROWS)
  const float* bias;      // expansion bias, torch order [c * R^2 + sub]
  const float* wf;        // tail conv weight fp32 [3][64][3][3]
  float* out;             // fp32 NCHW [B][3][R Hi][R Wi]
  int B, Hi, Wi;
  int nstrip, nseg, rows; // strips of 16 input columns every 15; segments of `rows` input rows every rows - 1; rows % 3 == 0
};

template <int R, bool P64IN>
__global__ void __launch_bounds__(64 * R * R, (R == 2 ? 4 : 3)) tail_fwd_stream_kernel(TSArgs p) {
  using T = bf16_t;
  using Cfg = TSCfg<R>;
  constexpr int NW = Cfg::NW, NPX = Cfg::NPX, NTHR = 64 * NW, RPW = Cfg::RPW, UPS = Cfg::UPS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Wfs)[TS_LDW] = reinterpret_cast<T(*)[TS_LDW]>(smem);                                  // [48 rows, see below][ic]
  f32x4* Bs = reinterpret_cast<f32x4*>(smem + Cfg::szW);                                   // [(wave * 4 + 2 kc + half) * 4 + g]: bias of accumulator rows 4 g + 0 .. 3
  f32x4* Ys = reinterpret_cast<f32x4*>(smem + Cfg::szW + Cfg::szB);                        // [((slot * R + sub-row) * 9 + tap) * NPX + pixel]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int sy = wv / R, sx = wv - sy * R;
  const int Hi = p.Hi, Wi = p.Wi, H = R * Hi, W = R * Wi;
  const long long npix = (long long)p.B * Hi * Wi, hw = (long long)H * W;

  // ---- task geometry (XCD-aware order: consecutive logical indices = neighbouring strips of one image) ----
  const int task = xcd_block_index();
  const int strip = task % p.nstrip;
  const int seg = (task / p.nstrip) % p.nseg;
  const int b = task / (p.nstrip * p.nseg);
  const int ci = min(15 * strip, Wi - 16);                       // first input column of the strip
  const int ri = min((p.rows - 1) * seg, Hi - p.rows);           // first input row of the segment
  // output ranges: a pixel is written by the strip / segment that holds all three of its Y columns / rows
  const int ox0 = strip == 0 ? 0 : R * min(15 * (strip - 1), Wi - 16) + NPX - 1;
  const int ox1 = (ci + 16 == Wi) ? W : R * ci + NPX - 1;
  const int oy0 = seg == 0 ? 0 : R * min((p.rows - 1) * (seg - 1), Hi - p.rows) + R * p.rows - 1;
  const int oy1 = (ri + p.rows == Hi) ? H : R * (ri + p.rows) - 1;

  // ---- per-wave constants: weight fragments, bias ----
  Frag8<T> w3f[2][2][2];                                         // [kc_out][half][kc_in]
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int chl = 32 * kc + 8 * (lr >> 2) + 4 * hf + (lr & 3);         // the channel lane lr supplies as MFMA row lr
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) w3f[kc][hf][ki] = load8(p.wp + (long long)(wv * 64 + chl) * 64 + 32 * ki + 8 * g);
    }
  // the bias of accumulator row 4 g + r of tile (kc, half) = channel 32 kc + 8 g + 4 half + r: in LDS (16 registers otherwise)
  for (int i = tid; i < NW * 64; i += NTHR) {
    const int w = i >> 6, q = (i >> 2) & 15, r = i & 3;           // q = (2 kc + half) * 4 + g
    const int kc = q >> 3, hf = (q >> 2) & 1, gg = q & 3;
    reinterpret_cast<float*>(Bs)[i] = p.bias[(32 * kc + 8 * gg + 4 * hf + r) * NW + w];
  }
  // tail conv weight rows in accumulator order: row 16 nt + 4 gg + r = (tap 4 nt + gg, oc r) for nt < 2, (tap 8, oc r) for
  // nt = 2, gg = 0; r = 3 and the rest of tile 2 are zero rows
  for (int i = tid; i < 48 * 64; i += NTHR) {
    const int row = i >> 6, ic = i & 63;
    const int nt = row >> 4, gg = (row >> 2) & 3, r = row & 3;
    const int tap = nt < 2 ? 4 * nt + gg : (gg == 0 ? 8 : -1);
    float v = 0.f;
    if (tap >= 0 && r < 3) v = p.wf[(r * 64 + ic) * 9 + tap];
    Wfs[row][ic] = from_f<T>(v);
  }
  // the step's input fragments: pixel (row, ci + lr), channels 32 ki + 8 g .. + 7; rows clamped to the image (a clamped row is never
  // used).  EVERY load of the loop is unconditional, see the header
  const T* abase[2];
#pragma unroll
  for (int ki = 0; ki < 2; ++ki) {
    const long long pix = ((long long)b * Hi + ri) * Wi + ci + lr;
    const int c0 = 32 * ki + 8 * g;
    abase[ki] = P64IN ? p.a + ((long long)(c0 >> 4) * npix + pix) * 16 + (c0 & 15) : p.a + pix * 64 + c0;
  }
  const int smax = Hi - 1 - ri;
  const int rstride = P64IN ? Wi * 16 : Wi * 64;
  auto a_load = [&](int s, Frag8<T> (&f)[2]) {
    const long long off = (long long)min(s, smax) * rstride;
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) f[ki] = load8(abase[ki] + off);
  };
  Frag8<T> fa[2], fb[2], fc[2];
  a_load(0, fa);
  a_load(1, fb);
  // pin everything the prologue loaded as arrived (header, rule 2)
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) asm volatile("" : "+v"(w3f[kc][hf][ki].v));
#pragma unroll
  for (int ki = 0; ki < 2; ++ki) { asm volatile("" : "+v"(fa[ki].v)); asm volatile("" : "+v"(fb[ki].v)); }
  lds_barrier();

  // ---- the consumer's lane geometry: lane = (row of the wave's RPW rows, pixel of the strip) ----
  const int c_lrow = min(lane / NPX, RPW - 1), c_px = lane % NPX;
  const int ox = R * ci + c_px;
  const bool c_on = lane < RPW * NPX && ox >= ox0 && ox < ox1;
  int cpx[3];                                                    // strip-local pixel of the three (reflected) columns
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) cpx[kx] = min(max(reflect_idx(ox + kx - 1, W) - R * ci, 0), NPX - 1);
  float* const obase = p.out + (long long)b * 3 * hw + ox;

  auto sum_store = [&](int y, const int (&rowb)[3], bool ok) {   // rowb[ky]: ring index of Y row y + ky - 1, tap 3 ky, pixel 0
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const f32x4 v = Ys[rowb[ky] + kx * NPX + cpx[kx]];
        s0 += v[0]; s1 += v[1]; s2 += v[2];
      }
    if (ok) {
      float* o = obase + (long long)y * W;
      o[0] = s0; o[hw] = s1; o[2 * hw] = s2;
    }
  };
  // any output row whose Y rows are in the ring: reflected rows, ring slots by division (the image's first and last rows)
  auto consume_generic = [&](int y, bool on) {
    int rowb[3];
    const int yc = min(max(y, 0), H - 1);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ry = reflect_idx(yc + ky - 1, H);
      const int ir = ry / R;
      rowb[ky] = (((max(ir - ri, 0) % 3) * R + (ry - ir * R)) * 9 + 3 * ky) * NPX;
    }
    sum_store(yc, rowb, on && c_on && y >= oy0 && y < oy1);
  };

  // one step; `cur` holds input row ri + s, `nxt` (the buffer of row ri + s - 1, free now) receives row ri + s + 2; SLOT = s % 3
  auto step = [&](int s, auto slot_tag, Frag8<T> (&cur)[2], Frag8<T> (&nxt)[2]) {
    constexpr int SLOT = decltype(slot_tag)::value, PREV = (SLOT + 2) % 3;
    a_load(s + 2, nxt);
    // ---- expansion: four channel tiles of this sub-pixel position, 16 pixels ----
    f32x4 acc[2][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        acc[kc][hf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) mma16(acc[kc][hf], w3f[kc][hf][ki], cur[ki]);
      }
    // ---- bias + GELU; the two tiles of kc give the lane channels 32 kc + 8 g .. + 7 of its pixel: the conv's B operand ----
    f32x4 y[3] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      float v8[8];
#if defined(TS_KO) && (TS_KO & 1)       // scratch/bench_tail.hip knock-out (results WRONG): the activation replaced by bias add
      { const f32x4 b0 = Bs[(wv * 4 + 2 * kc) * 4 + g], b1 = Bs[(wv * 4 + 2 * kc + 1) * 4 + g];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v8[e] = acc[kc][0][e] + b0[e]; v8[4 + e] = acc[kc][1][e] + b1[e]; } }
#else
      gelu_tail8<T>(acc[kc][0], acc[kc][1], Bs[(wv * 4 + 2 * kc) * 4 + g], Bs[(wv * 4 + 2 * kc + 1) * 4 + g], v8);
#endif
      Frag8<T> bf;
#pragma unroll
      for (int e = 0; e < 8; ++e) bf.set(e, v8[e]);
#pragma unroll
      for (int nt = 0; nt < 3; ++nt) mma16(y[nt], load8(&Wfs[16 * nt + lr][32 * kc + 8 * g]), bf);     // (A fragments from LDS: registers matter more)
    }
    // ---- Y row (input row ri + s, sub-row sy), pixel R lr + sx: taps g, 4 + g and (g = 0) 8 ----
    {
      f32x4* yb = Ys + (SLOT * R + sy) * 9 * NPX + R * lr + sx;
      yb[g * NPX] = y[0];
      yb[(4 + g) * NPX] = y[1];
      if (g == 0) yb[8 * NPX] = y[2];
    }
#if !(defined(TS_KO) && (TS_KO & 2))    // knock-out 2: no barrier (and so no ordering between producers and the consumer)
    lds_barrier();
#endif
    // ---- output rows R (ri + s) - 1 .. R (ri + s) + R - 2: their three Y rows are complete.  Unit u (RPW rows) -> wave (s UPS + u) % NW ----
#pragma unroll
    for (int u = 0; u < UPS; ++u) {
#if defined(TS_KO) && (TS_KO & 4)       // knock-out 4: no consumer (tap sums + stores)
      if (wv == 77) {
#else
      if (wv == (s * UPS + u) % NW) {
#endif
        const int j = u * RPW + c_lrow;                          // row R (ri + s) - 1 + j
        const int yrow = R * (ri + s) - 1 + j;
        if (s > 0) {
          // rows y - 1 .. y + 1 relative to the first row of the previous slot: R - 2 + j + ky  (< R: previous slot)
          int rowb[3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int rel = R - 2 + j + ky;
            rowb[ky] = (((rel < R ? PREV : SLOT) * R + (rel < R ? rel : rel - R)) * 9 + 3 * ky) * NPX;
          }
          sum_store(yrow, rowb, c_on && yrow >= oy0 && yrow < oy1);
        } else {
          consume_generic(yrow, ri == 0 && yrow >= 0);           // the image's first rows (row -1 reflects onto row 1)
        }
      }
    }
  };
  for (int s = 0; s < p.rows; s += 3) {       // rows is a multiple of 3 (launcher): no branch around a load, see the header
    step(s, std::integral_constant<int, 0>{}, fa, fc);
    step(s + 1, std::integral_constant<int, 1>{}, fb, fa);
    step(s + 2, std::integral_constant<int, 2>{}, fc, fb);
  }
  // the image's last row (its lower neighbour is the reflected row H - 2); every Y row was complete at the last barrier
  if (wv == 0 && ri + p.rows == Hi && c_lrow == 0) consume_generic(H - 1, true);
}

}  // namespace

// rows per segment for an input of Hi rows: segments of ~`target` rows (one halo row above and below), evenly sized, a multiple of 3
static int tail_stream_rows(int Hi, int target) {
  int rows = Hi;
  if (Hi > target) {
    const int nseg = (Hi + target - 2) / (target - 1);
    rows = (Hi - 1 + nseg - 1) / nseg + 1;
  }
  rows = (rows + 2) / 3 * 3;                // the kernel's step loop is unrolled by three
  if (rows > Hi) rows = Hi / 3 * 3;         // (then two overlapping segments cover the image)
  return rows;
}

#ifdef TS_BENCH_HOOKS
static int g_ts_pad_lds = 0;      // scratch/bench_tail.hip: extra dynamic LDS per workgroup, to force a lower occupancy
#define TS_PAD g_ts_pad_lds
#else
#define TS_PAD 0
#endif

// bf16.  R = 2: x4's tail.3 stage (a = gelu(t1), NHWC [B][Hi][Wi][64]) or x2's tail.0 stage (a = the body output, P64 planes);
// R = 3: x3's tail.0 stage.  wp / bias / wf as for launch_tail_fwd_fused.  Hi, Wi >= 16.
int launch_tail_fwd_stream(const void* a, int a_is_p64, const void* wp, const float* bias, const float* wf, float* out, int B, int Hi,
                           int Wi, int R, int seg_rows, hipStream_t st) {
  if (Hi < 16 || Wi < 16) return m2t_set_error(-2, "tail_fwd_stream: input smaller than 16 x 16");
  if (R != 2 && R != 3) return m2t_set_error(M2T_UNSUPPORTED, "tail_fwd_stream: R must be 2 or 3");
  TSArgs p{(const bf16_t*)a, (const bf16_t*)wp, bias, wf, out, B, Hi, Wi, 0, 0, 0};
  p.rows = tail_stream_rows(Hi, seg_rows > 2 ? seg_rows : 32);
  p.nstrip = (Wi - 16 + 14) / 15 + 1;
  p.nseg = (Hi - p.rows + p.rows - 2) / (p.rows - 1) + 1;
  const int ntask = B * p.nstrip * p.nseg;
#define TS_GO(RR, PP)                                                                                                    \
  do {                                                                                                                   \
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_fwd_stream_kernel<RR, PP>, (int)TSCfg<RR>::total + TS_PAD)) return rc__; \
    M2T_LAUNCH_TIMED((tail_fwd_stream_kernel<RR, PP>), dim3(ntask), dim3(64 * RR * RR), TSCfg<RR>::total + TS_PAD, st, p); \
  } while (0)
  if (R == 2 && !a_is_p64) TS_GO(2, false);
  else if (R == 2) TS_GO(2, true);
  else if (!a_is_p64) TS_GO(3, false);
  else TS_GO(3, true);
#undef TS_GO
  M2T_LAUNCH_CHECK();
  return 0;
}
