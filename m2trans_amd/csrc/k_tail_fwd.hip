// k_tail_fwd.hip -- the x4 tail's high-resolution half of the FORWARD pass in one kernel (bf16):
//
//     t2  = PixelShuffle(2)(conv1x1 64 -> 256 (a1) + b3)         (models/M2Trans_network.py:45-46)
//     a2  = gelu(t2)                                              (:47)
//     sr' = conv3x3 64 -> 3, reflect padding, no bias (a2)        (:48)
//
// Unfused (tail_expand_kernel + final_conv_fwd_kernel) the 64-channel 512x512 tensor a2 is written once, together with
// gelu'(t2) for the backward pass, and read once: 1.07 GB + 0.54 GB per step at batch 16 -- the largest single item of
// the step's HBM traffic -- for 134 MB of input and 50 MB of output.  Here a2 exists only in LDS: per 16x16 output
// tile a workgroup stages the 10x10 mid-resolution pixels of a1 = gelu(t1) that cover the tile and its 1-pixel halo
// (reflected at the image border, so every source lies inside the image), expands them (224 MFMAs), applies GELU
// (the activation is what the kernel is bound by: 25 600 evaluations per tile on the VALU, the halo costs 1.56x; in bf16 mode
// it is the exp2 / rcp form of m2t_common.h, 11 issue slots instead of erf's 14 + a division), and feeds
// the 18x18 halo pixels to the tail conv exactly as final_conv_fwd_kernel does.  Two 512-thread workgroups share a CU (78 KB of LDS
// and 124 VGPRs each: the tail.3 weight fragments are 16 registers per wave, not 36.8 KB of LDS), so one workgroup's GELU phase
// runs under the other's MFMA / LDS phases: 341 -> 279 us per launch at batch 16.  The backward pass recomputes a2 and
// gelu'(t2) the same way (k_tail_bwd.hip, RC variant).  Operand fragments, k order, bias add, gelu and the tap
// summation are those of the two kernels it replaces: identical bits.
#include "m2t_kernels.h"

namespace {

constexpr int TF_T = 16;                         // output tile edge
constexpr int TF_MB = TF_T / 2 + 2;              // 10: edge of the mid-resolution block
constexpr int TF_NM = TF_MB * TF_MB;             // 100 mid pixels (7 MFMA column tiles)
constexpr int TF_BE = 2 * TF_MB;                 // 20: edge of the a2 block
constexpr int TF_HP = (TF_T + 2) * (TF_T + 2);   // 324 halo pixels of the conv (21 tiles of 16, last one partly padding)
constexpr int TF_LD = 72;
constexpr size_t TF_SZ_WF = sizeof(bf16_t) * 32 * TF_LD;
constexpr size_t TF_SZ_A1 = sizeof(bf16_t) * 112 * TF_LD, TF_SZ_A2 = sizeof(bf16_t) * TF_BE * TF_BE * TF_LD;
constexpr size_t TF_SMEM = TF_SZ_WF + TF_SZ_A1 + TF_SZ_A2;      // 78 336 B: TWO workgroups per CU (the tail.3 weight fragments live in registers)
static_assert(2 * TF_SMEM <= 160 * 1024, "two workgroups per CU");
static_assert(sizeof(float) * 336 * 33 <= TF_SZ_A2, "the fp32 product tile overlays the a2 block");

__global__ void __launch_bounds__(512, 4) tail_fwd_fused_kernel(const bf16_t* __restrict__ a1, const bf16_t* __restrict__ w3p,
                                                             const float* __restrict__ b3, const float* __restrict__ wf,
                                                             float* __restrict__ out, int B, int H, int W) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Wfs)[TF_LD] = reinterpret_cast<T(*)[TF_LD]>(smem);                                          // [(tap, oc) -> 32][ic]
  T(*A1b)[TF_LD] = reinterpret_cast<T(*)[TF_LD]>(smem + TF_SZ_WF);                               // [100 mid pixels -> 112][k]
  T(*A2b)[TF_LD] = reinterpret_cast<T(*)[TF_LD]>(smem + TF_SZ_WF + TF_SZ_A1);                    // [20 x 20 block pixels][c]
  float(*Ys)[33] = reinterpret_cast<float(*)[33]>(smem + TF_SZ_WF + TF_SZ_A1);                   // [336][33], overlays A2b
  const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int Hm = H / 2, Wm = W / 2;
  const int tw = W / TF_T, th = H / TF_T;
  const long long ntiles = (long long)B * th * tw;
  const long long hw = (long long)H * W;
  // this wave's tail.3 weight fragments (n' tiles 2 w8, 2 w8 + 1; [256 n'][64 k] packed rows): 16 registers for the whole kernel --
  // in LDS they were 36.8 KB, the difference between one and two workgroups per CU
  Frag8<T> w3f[2][2];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) w3f[o][kc] = load8(w3p + (long long)(16 * (2 * w8 + o) + lr) * 64 + 32 * kc + 8 * g);
  for (int i = tid; i < 32 * 64; i += 512) {
    const int n = i >> 6, ic = i & 63;              // n = tap * 3 + oc
    float v = 0.f;
    if (n < 27) v = wf[((n % 3) * 64 + ic) * 9 + n / 3];
    Wfs[n][ic] = from_f<T>(v);
  }
  for (int i = tid; i < 12 * 8; i += 512) store8(&A1b[100 + (i >> 3)][(i & 7) * 8], frag_zero<T>());    // MFMA padding rows
  // tiles are dealt round-robin over LOGICAL workgroup indices (XCD-aware: in every round an XCD owns a contiguous run)
  const long long lb = xcd_block_index();
  for (long long t = lb; t < ntiles; t += gridDim.x) {
    const int x0 = (int)(t % tw) * TF_T;
    const long long q = t / tw;
    const int y0 = (int)(q % th) * TF_T;
    const long long b = q / th;
    // the 10 x 10 mid pixels: clamped coordinates (a clamped pixel is never used).  No cross-tile prefetch in registers: the second
    // workgroup of the CU is what runs while this one waits (and the eight registers are what keeps the kernel at 128)
    Frag8<T> ra[2];
    {
      const T* ab = a1 + b * (long long)Hm * Wm * 64;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int idx = min(tid + it * 512, TF_NM * 8 - 1);
        const int m = idx >> 3, cv = idx & 7;
        const int my = min(max(y0 / 2 - 1 + m / TF_MB, 0), Hm - 1), mx = min(max(x0 / 2 - 1 + m % TF_MB, 0), Wm - 1);
        ra[it] = load8(ab + ((long long)my * Wm + mx) * 64 + cv * 8);
      }
    }
    lds_barrier();          // the previous tile's readers of A1b / Ys are done (first trip: weights staged)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 512;
      if (idx < TF_NM * 8) store8(&A1b[idx >> 3][(idx & 7) * 8], ra[it]);
    }
    lds_barrier();
    // ---- t2^T [n'][m] = W3 a1^T + b3, a2 = gelu(t2) -> the 20 x 20 block.  wave w8: n' tiles 2 w8, 2 w8 + 1 ----
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int nt = 2 * w8 + o, sub = nt >> 2, ct = nt & 3;
      f32x4 acc[7];
#pragma unroll
      for (int mt = 0; mt < 7; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
        for (int mt = 0; mt < 7; ++mt) mma16(acc[mt], w3f[o][kc], load8(&A1b[16 * mt + lr][32 * kc + 8 * g]));
      }
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = b3[(16 * ct + 4 * g + r) * 4 + sub];
#pragma unroll
      for (int mt = 0; mt < 7; ++mt) {
        const int m = 16 * mt + lr;
        if (m < TF_NM) {
          float av[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) av[r] = gelu_tail<T>(acc[mt][r] + bv[r]);
          const int row = (2 * (m / TF_MB) + (sub >> 1)) * TF_BE + 2 * (m % TF_MB) + (sub & 1);
          store4(&A2b[row][16 * ct + 4 * g], av);
        }
      }
    }
    lds_barrier();
    // ---- tail conv: Y^T [(tap, oc)][halo pixel] = Wf a2^T over the 324 halo pixels (21 tiles of 16) ----
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      acc[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[j][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int mt = w8 + 8 * j;
      if (mt < 21) {
        const int p = min(16 * mt + lr, TF_HP - 1);
        const int py = p / (TF_T + 2), px = p - py * (TF_T + 2);
        const int ry = reflect_idx(y0 + py - 1, H), rx = reflect_idx(x0 + px - 1, W);
        const int brow = (ry - (y0 - 2)) * TF_BE + (rx - (x0 - 2));
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const Frag8<T> xf = load8(&A2b[brow][32 * kc + 8 * g]);
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) mma16(acc[j][nt], load8(&Wfs[16 * nt + lr][32 * kc + 8 * g]), xf);
        }
      }
    }
    lds_barrier();          // every wave is done reading the a2 block: its memory becomes the product tile
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int mt = w8 + 8 * j;
      if (mt < 21) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ys[16 * mt + lr][16 * nt + 4 * g + r] = acc[j][nt][r];
      }
    }
    lds_barrier();
    if (tid < 256) {
      const int ty = tid >> 4, tx = tid & 15;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const float* yp = &Ys[(ty + ky) * (TF_T + 2) + tx + kx][tap * 3];
        s0 += yp[0]; s1 += yp[1]; s2 += yp[2];
      }
      const long long o = b * 3 * hw + (long long)(y0 + ty) * W + x0 + tx;
      out[o] = s0;
      out[o + hw] = s1;
      out[o + 2 * hw] = s2;
    }
  }
}

}  // namespace

// bf16, x4 tail.  a1 = gelu(t1) [B][H/2][W/2][64]; w3p = packed tail.3 weight rows [sub * 64 + c][64] (M2T_PACK_SHUF_ROWS);
// b3 = tail.3 bias fp32 in torch order; wf = tail.6 weight fp32 [3][64][3][3]; out fp32 NCHW [B][3][H][W].  H, W multiples of 16.
int launch_tail_fwd_fused(const void* a1, const void* w3p, const float* b3, const float* wf, float* out, int B, int H, int W,
                          hipStream_t st) {
  if (H % 32 || W % 32) return m2t_set_error(-2, "tail_fwd_fused: H, W must be multiples of 32");
  const long long ntiles = (long long)B * (H / TF_T) * (W / TF_T);
  const int nblk = (int)std::min<long long>(512, ntiles);         // two workgroups per CU
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_fwd_fused_kernel, (int)TF_SMEM)) return rc__;
  M2T_LAUNCH_TIMED(tail_fwd_fused_kernel, dim3(nblk), dim3(512), TF_SMEM, st, (const bf16_t*)a1, (const bf16_t*)w3p, b3, wf, out, B,
                   H, W);
  M2T_LAUNCH_CHECK();
  return 0;
}
