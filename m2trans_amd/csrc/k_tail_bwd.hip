// k_tail_bwd.hip -- the backward of the x4 tail's high-resolution half in ONE pass over the HR tensors (bf16):
//
//     g(a2)  = conv3x3^T(g(sr))                      tail conv data gradient          (models/M2Trans_network.py:48)
//     dWf   += g(sr) (*) a2                           tail conv weight gradient
//     g(t2)  = g(a2) * gelu'(t2)                      GELU backward                   (:46)
//     g(u)   = pixel_unshuffle(g(t2))                 PixelShuffle backward           (:45)
//     g(t1)  = (g(u) W3) * gelu'(t1)                  tail.3 data gradient + GELU     (:44,43)
//     dW3   += g(u)^T a1 ,  db3 += sum g(u)           tail.3 weight / bias gradient
//
// Unfused these are four kernels that write g(t2) (537 MB at batch 16) once and read it twice; here g(t2) lives
// only in LDS.  Per 16x16 HR tile (= 8x8 pixels of the 2x2-shuffled mid-resolution map) a workgroup reads the
// stored activation a2 = gelu(t2) and derivative gelu'(t2) once (tail_expand wrote both), the 18x18 halo of
// g(sr), and the a1 = gelu(t1) / gelu'(t1) tiles, and writes the g(t1) tile.  The three parameter gradients
// accumulate in registers over the workgroup's strip of tiles and leave as one fp32 slab each (deterministic
// reduction afterwards, no atomics).  HBM per step: 1.5 GB instead of 3.3 GB; measured 318 us (4.8 TB/s)
// against 804 us for the four kernels (B = 16, 512x512 HR).
//
// The reflect padding of the tail conv is folded into "Geff" exactly as in final_conv_dgrad_kernel (k_conv.hip):
// Geff[q][(tap,oc)] gathers g(sr) at the output positions that read input pixel q through `tap`, including the
// reads that reached q through the padding ring; then g(a2) = Geff Wf and dWf = Geff^T a2 (contraction over q).
#include "m2t_kernels.h"
#include <type_traits>

#ifndef M2T_TAIL_STAMP
#define M2T_TAIL_STAMP(i) do { } while (0)       // scratch/bench_tail.hip defines it to record s_memtime per phase
#endif
#ifndef M2T_TAIL_STAMP2
#define M2T_TAIL_STAMP2(i) do { (void)tile_it; } while (0)   // the 32x32x16 kernel's stamps (a loop counter instead of a division per stamp)
#endif

namespace {

constexpr int TB_T = 16;                       // HR tile edge
constexpr int TB_HP = (TB_T + 2) * (TB_T + 2); // halo pixels of g(sr)
constexpr int TB_LD = 72;

struct TailBwdArgs {
  const float* gout;      // g(sr)   fp32 [B][3][H][W]
  const float* wf;        // tail conv weight fp32 [3][64][3][3]
  const bf16_t* act;      // a2 = gelu(t2)   [B][H][W][64]
  const bf16_t* der;      // gelu'(t2)       [B][H][W][64]
  const bf16_t* a1;       // a1 = gelu(t1)   [B][H/2][W/2][64]
  const bf16_t* d1;       // gelu'(t1)       [B][H/2][W/2][64]
  const bf16_t* w3t;      // packed tail.3 weight^T [64 k][256 n'], n' = sub*64 + c
  const float* b3;        // tail.3 bias fp32 [256] in torch order (c*4 + sub); used by the recomputing variant
  bf16_t* gt1;            // g(t1)           [B][H/2][W/2][64]
  float* slab_wf;         // [nblk][32][64]   ((tap*3+oc) x ic)
  float* slab_w3;         // [nblk][256][64]  (n' x k)
  float* slab_b3;         // [nblk][256]
  int B, H, W;
  // L1 (round 5): the clamp + L1 seed of train.py:199 (clamp_l1_vec4_kernel) is taken while the g(sr) halo is staged -- gout is then
  // unused: g(sr) = sign(clamp(pre) - hr) * gscale where 0 <= pre <= R inside the crop, 0 elsewhere -- and the workgroup leaves the
  // sum of |clamp(pre) - hr| over ITS tiles' own pixels in loss_part[blockIdx] (fixed tile assignment: deterministic)
  const float* pre;       // pre-clamp output fp32 [B][3][H][W] (padded size)
  const float* hr;        // target fp32 [B][3][Hs][Ws] (cropped size)
  float* loss_part;       // [nblk]
  int Hs, Ws;
  float R, gscale;
};

__device__ __forceinline__ Frag8<bf16_t> tr_rows(const bf16_t* lo, const bf16_t* hi) {
  // lo / hi: THIS lane's 8-byte pieces (row r + (i >> 2), columns c0 + 4 (i & 3) ..) of the two 4-row groups
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)lo);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)hi);
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// HR tile row of (mid pixel m = my*8 + mx, sub = i*2 + j)
__device__ __forceinline__ int hr_row(int m, int sub) { return (2 * (m >> 3) + (sub >> 1)) * TB_T + 2 * (m & 7) + (sub & 1); }

// RC (recompute): a2 = gelu(t2) and gelu'(t2) are NOT read from HBM (the forward then never stores them: 1.07 GB per
// step at batch 16) but recomputed per tile from the a1 tile that is staged anyway: t2 = W3 a1 + b3 is 128 MFMAs per tile,
// the two erf-based functions 32 elements per thread.  Same operand fragments, k order, bias add and gelu_erf_both as
// tail_expand_kernel (k_gemm.hip), so the recomputed values are the bits the forward would have stored.  gelu'(t2) is
// written where g(t2) goes (the product is formed in place), so the LDS footprint does not grow.
template <bool RC, bool L1 = false>
__global__ void __launch_bounds__(512) tail_bwd_fused_kernel(TailBwdArgs a) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float(*Gs)[TB_HP] = reinterpret_cast<float(*)[TB_HP]>(smem);                 // [3][324] g(sr) halo (0 outside the image)
  size_t off = sizeof(float) * 3 * TB_HP;
  T(*Ge)[40] = reinterpret_cast<T(*)[40]>(smem + off);  off += sizeof(T) * 256 * 40;      // Geff [pixel][(tap,oc) -> 32]
  T(*Wt)[40] = reinterpret_cast<T(*)[40]>(smem + off);  off += sizeof(T) * 64 * 40;       // [ic][(tap,oc)]
  T(*A2)[TB_LD] = reinterpret_cast<T(*)[TB_LD]>(smem + off);  off += sizeof(T) * 256 * TB_LD;   // a2 tile [pixel][c]
  T(*Gz)[TB_LD] = reinterpret_cast<T(*)[TB_LD]>(smem + off);  off += sizeof(T) * 256 * TB_LD;   // g(t2) tile [pixel][c]
  T(*A1)[TB_LD] = reinterpret_cast<T(*)[TB_LD]>(smem + off);  off += sizeof(T) * 64 * TB_LD;    // a1 tile [mid pixel][k]
  T(*W3s)[264] = reinterpret_cast<T(*)[264]>(smem + off);                                      // W3^T [k][n'] (whole strip)

  const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: SGPR arithmetic
  const int lr = lane & 15, g = lane >> 4;
  const int H = a.H, W = a.W, Hm = H / 2, Wm = W / 2;
  const int tw = W / TB_T, th = H / TB_T;
  const long long ntiles = (long long)a.B * th * tw;
  const long long hw = (long long)H * W;
  // tiles are dealt round-robin (tile = block + i * grid): border tiles, which cost more, spread over all workgroups
  // (LOGICAL workgroup index, XCD-aware: in every round an XCD owns a contiguous run of tiles -> halo columns shared in one L2)
  const long long t0 = xcd_block_index(), tstep = gridDim.x, t1 = ntiles;

  for (int i = tid; i < 64 * 32; i += 512) {
    const int ic = i >> 5, n = i & 31;
    float v = 0.f;
    if (n < 27) v = a.wf[((n % 3) * 64 + ic) * 9 + n / 3];
    Wt[ic][n] = from_f<T>(v);
  }
  // tail.3 data gradient: wave (kt = w8 & 3, mh = w8 >> 2) owns output channels 16 kt .. of mid tiles 2 mh, 2 mh + 1;
  // W3^T stays in LDS for the whole strip (in registers it pushed the prefetch registers into scratch: +37 % time)
  const int kt = w8 & 3, mh = w8 >> 2;
  for (int i = tid; i < 64 * 32; i += 512) store8(&W3s[i >> 5][(i & 31) * 8], load8(a.w3t + (long long)(i >> 5) * 256 + (i & 31) * 8));
  Frag8<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);
  // strip accumulators
  f32x4 accF = (f32x4){0.f, 0.f, 0.f, 0.f};      // dWf tile: rows (tap,oc) 16 (w8 >> 2) .., cols ic 16 (w8 & 3) ..
  f32x4 accW[2][4], accB[2];                     // dW3 rows n' 16 (2 w8 + o) .., cols k 16 kt2 .. ; db3
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    accB[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) accW[o][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // the recompute's bias values, once (a global load inside the tile loop is followed by s_waitcnt vmcnt(0): vmcnt retires in order,
  // so it drained the NEXT tile's prefetch issued just before it and exposed a full HBM round trip per tile)
  float bvr[2][4];
  if constexpr (RC) {
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int nt = 2 * w8 + o, sub = nt >> 2, ct = nt & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) bvr[o][r] = a.b3[(16 * ct + 4 * g + r) * 4 + sub];
    }
    // pin them as "arrived": without a use before the loop hipcc's waitcnt pass keeps a vmcnt(0) for them INSIDE the loop
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bvr[o][r]));
  }
  // ---- register-staged loads of a tile (raw bf16; two groups so that nothing prefetched must be copied) ----
  Frag8<T> ra2[4], ra1, rder[2][2];
  bf16x4 rd1[2];
  float rg[2], rh[2];                            // rg: g(sr) -- with L1: the pre-clamp value; rh: the target (L1 only)
  float l1acc = 0.f;
  auto tile_geom = [&](long long t, int& b, int& y0, int& x0) {
    const int tx = (int)(t % tw);
    const long long q = t / tw;
    y0 = (int)(q % th) * TB_T; x0 = tx * TB_T; b = (int)(q / th);
  };
  auto fetchA = [&](long long t) {              // staged through LDS: a2 tile, a1 tile, g(sr) halo
    int b, y0, x0;
    tile_geom(t, b, y0, x0);
    if constexpr (!RC) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 512;
        const int p = idx >> 3, cv = idx & 7;
        ra2[it] = load8(a.act + (((long long)b * H + y0 + (p >> 4)) * W + x0 + (p & 15)) * 64 + cv * 8);
      }
    }
    {
      const int m = tid >> 3, cv = tid & 7;
      ra1 = load8(a.a1 + (((long long)b * Hm + y0 / 2 + (m >> 3)) * Wm + x0 / 2 + (m & 7)) * 64 + cv * 8);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 512;
      // branch-free: a load under a lane-dependent branch is followed by a vmcnt(0) wait -- clamp the address, select afterwards
      const int ic = min(i, 3 * TB_HP - 1);
      const int oc = ic / TB_HP, p = ic - oc * TB_HP;
      const int py = p / (TB_T + 2), px = p - py * (TB_T + 2);
      const int gy = y0 + py - 1, gx = x0 + px - 1;
      // (the select happens when the value is staged, not here: using the loaded value now would wait for it)
      if constexpr (L1) {
        const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
        rg[it] = a.pre[((long long)b * 3 + oc) * hw + (long long)cy * W + cx];
        rh[it] = a.hr[(((long long)b * 3 + oc) * a.Hs + min(cy, a.Hs - 1)) * a.Ws + min(cx, a.Ws - 1)];
      } else {
        rg[it] = a.gout[((long long)b * 3 + oc) * hw + (long long)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
      }
    }
  };
  auto fetchB = [&](long long t) {              // consumed from registers: gelu'(t2) of this lane's two conv-gradient
    int b, y0, x0;                              // pixels (16 channels each), gelu'(t1) of its two mid pixels
    tile_geom(t, b, y0, x0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      if constexpr (!RC) {
        const int p = 16 * (2 * w8 + mt) + lr;
        const T* dp = a.der + (((long long)b * H + y0 + (p >> 4)) * W + x0 + (p & 15)) * 64 + 16 * g;
        rder[mt][0] = load8(dp);
        rder[mt][1] = load8(dp + 8);
      }
      const int m = 16 * (2 * mh + mt) + lr;
      rd1[mt] = *reinterpret_cast<const bf16x4*>(a.d1 + (((long long)b * Hm + y0 / 2 + (m >> 3)) * Wm + x0 / 2 + (m & 7)) * 64 + 16 * kt + 4 * g);
    }
  };
  if (t0 < t1) { fetchA(t0); fetchB(t0); }

  for (long long t = t0; t < t1; t += tstep) {
    int b, y0, x0;
    tile_geom(t, b, y0, x0);
    M2T_TAIL_STAMP(0);
    // ---- stage ----
    if constexpr (!RC) {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 512;
        store8(&A2[idx >> 3][(idx & 7) * 8], ra2[it]);
      }
    }
    store8(&A1[tid >> 3][(tid & 7) * 8], ra1);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * 512;
      if (i < 3 * TB_HP) {
        const int p = i % TB_HP;
        const int py = p / (TB_T + 2), px = p % (TB_T + 2);
        const int gy = y0 + py - 1, gx = x0 + px - 1;
        if constexpr (L1) {
          // clamp_l1_vec4_kernel's arithmetic on the staged value: the seed is 0 outside the crop and where the clamp is active
          float gv = 0.f;
          if (gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws) {
            const float v = rg[it];
            const float c = fminf(fmaxf(v, 0.f), a.R);
            const float d = c - rh[it];
            const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
            gv = (v >= 0.f && v <= a.R) ? sg * a.gscale : 0.f;
            if (py >= 1 && py <= TB_T && px >= 1 && px <= TB_T) l1acc += fabsf(d);     // the tile's OWN pixels: each pixel of the image once
          }
          Gs[i / TB_HP][p] = gv;
        } else {
          Gs[i / TB_HP][p] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? rg[it] : 0.f;      // 0 outside the image
        }
      }
    }
    __syncthreads();
    M2T_TAIL_STAMP(1);
    if (t + tstep < t1) fetchA(t + tstep);              // next tile's loads fly under this tile's products
    if constexpr (RC) {
      // ---- recompute t2^T [n'][m] = W3 [n'][k] a1^T [k][m] + b3: wave w8 owns n' tiles 2 w8, 2 w8 + 1 (one sub-pixel
      // and channel tile each), all four mid-pixel tiles; a2 -> A2, gelu'(t2) -> Gz (g(t2) is formed in place there)
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int nt = 2 * w8 + o, sub = nt >> 2, ct = nt & 3;
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const Frag8<T> wf = load8_tr(&W3s[32 * kc + 8 * g][16 * nt], &W3s[32 * kc + 8 * g + 4][16 * nt], 264, lane);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) mma16(acc[mt], wf, load8(&A1[16 * mt + lr][32 * kc + 8 * g]));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          float av[4], dv[4];
          gelu_tail_both4<T>(acc[mt], bvr[o], av, dv);
          const int row = hr_row(16 * mt + lr, sub);
          store4(&A2[row][16 * ct + 4 * g], av);
          store4(&Gz[row][16 * ct + 4 * g], dv);
        }
      }
    }
    M2T_TAIL_STAMP(2);
    // ---- Geff: one thread per tile pixel builds its 27 gathered taps (reflect ring folded in) ----
    // Only pixels in image rows / columns 1 and n-2 receive extra reads through the padding ring, so a tile that
    // does not touch the image border takes the branch-free path: tap (ky,kx) of pixel (ty,tx) is the halo
    // entry (ty - ky + 2, tx - kx + 2) (Gs is 0 outside the image).
    if (tid < 256) {
      const int ty = tid >> 4, tx = tid & 15;
      float ge[32];
#pragma unroll
      for (int i = 27; i < 32; ++i) ge[i] = 0.f;
      // the reads that reach pixel (ty,tx) directly: tap (ky,kx) <- halo entry (ty - ky + 2, tx - kx + 2); Gs is 0 outside the image
#pragma unroll
      for (int oc = 0; oc < 3; ++oc)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) ge[(ky * 3 + kx) * 3 + oc] = Gs[oc][(ty - ky + 2) * (TB_T + 2) + (tx - kx + 2)];
      // image rows / columns 1 and n-2 are also read through the reflect ring (positions -1 and n): up to three more
      // source positions for the few pixels concerned
      const int yy = y0 + ty, xx = x0 + tx;
      const int ey = (yy == 1) ? -1 : ((yy == H - 2) ? H : yy);        // the mirrored row (or yy itself: none)
      const int ex = (xx == 1) ? -1 : ((xx == W - 2) ? W : xx);
      if (ey != yy || ex != xx) {
#pragma unroll 1
        for (int combo = 1; combo < 4; ++combo) {
          const int py = (combo & 1) ? ey : yy, px = (combo & 2) ? ex : xx;
          if ((combo & 1) && ey == yy) continue;
          if ((combo & 2) && ex == xx) continue;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int oy = py - ky + 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int ox = px - kx + 1;
              if (oy >= 0 && oy < H && ox >= 0 && ox < W) {
                const int hidx = (oy - y0 + 1) * (TB_T + 2) + (ox - x0 + 1);      // always inside the +-1 halo
#pragma unroll
                for (int oc = 0; oc < 3; ++oc) ge[(ky * 3 + kx) * 3 + oc] += Gs[oc][hidx];
              }
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v8[e] = ge[8 * j + e];
        store8f(&Ge[tid][8 * j], v8);
      }
    }
    M2T_TAIL_STAMP(3);
    __syncthreads();
    M2T_TAIL_STAMP(4);
    // ---- g(t2) = (Geff Wf) * gelu'(t2): wave w8 owns pixel tiles 2 w8, 2 w8 + 1 (one k-step: 27 -> 32) ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int pt = 2 * w8 + mt;
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const Frag8<T> xf = load8(&Ge[16 * pt + lr][8 * g]);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int nl = 16 * (lr >> 2) + 4 * nt + (lr & 3);
        mma16(acc[nt], load8(&Wt[nl][8 * g]), xf);
      }
      float v[16];
      if constexpr (RC) {
        float d[16];
        load16f(&Gz[16 * pt + lr][16 * g], d);       // gelu'(t2), written by the recompute phase (same barrier as Geff)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[nt][r] * d[4 * nt + r];
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[nt][r] * rder[mt][nt >> 1].get(4 * (nt & 1) + r);
      }
      store16f(&Gz[16 * pt + lr][16 * g], v);
    }
    M2T_TAIL_STAMP(5);
    // ---- dWf += Geff^T a2 (contraction over the 256 tile pixels): wave -> ((tap,oc) tile w8 >> 2, ic tile w8 & 3) ----
    {
      const int i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int r0 = 32 * ks + 8 * g + qq;
        const Frag8<T> ga = tr_rows(&Ge[r0][16 * (w8 >> 2) + 4 * pp], &Ge[r0 + 4][16 * (w8 >> 2) + 4 * pp]);
        const Frag8<T> ab = tr_rows(&A2[r0][16 * (w8 & 3) + 4 * pp], &A2[r0 + 4][16 * (w8 & 3) + 4 * pp]);
        mma16(accF, ga, ab);
      }
    }
    M2T_TAIL_STAMP(6);
    __syncthreads();      // g(t2) tile complete
    M2T_TAIL_STAMP(7);
    // ---- g(t1)^T [k][m] = sum_n' W3^T[k][n'] g(u)[m][n'],  g(u)[m][sub*64 + c] = g(t2)[hr(m, sub)][c] ----
    {
      f32x4 accD[2];
      accD[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      accD[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 8; ++kc) {
        const int sub = kc >> 1, c0 = (kc & 1) * 32 + 8 * g;
        const Frag8<T> w3k = load8(&W3s[16 * kt + lr][32 * kc + 8 * g]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int m = 16 * (2 * mh + mt) + lr;
          mma16(accD[mt], w3k, load8(&Gz[hr_row(m, sub)][c0]));
        }
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int m = 16 * (2 * mh + mt) + lr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = accD[mt][r] * (float)rd1[mt][r];
        store4(a.gt1 + (((long long)b * Hm + y0 / 2 + (m >> 3)) * Wm + x0 / 2 + (m & 7)) * 64 + 16 * kt + 4 * g, v);
      }
    }
    M2T_TAIL_STAMP(8);
    if (t + tstep < t1) fetchB(t + tstep);              // (this tile's derivative registers are consumed)
    // ---- dW3[n'][k] += g(u)^T a1 (contraction over the 64 mid pixels), db3 += column sums of g(u) ----
    {
      const int i = lane & 15, qq = i >> 2, pp = i & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int m0 = 32 * ks + 8 * g + qq;                  // this lane's mid pixel in the two 4-row groups: m0, m0 + 4
        Frag8<T> af[4];
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) af[k2] = tr_rows(&A1[m0][16 * k2 + 4 * pp], &A1[m0 + 4][16 * k2 + 4 * pp]);
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const int nt = 2 * w8 + o, sub = nt >> 2, ct = nt & 3;
          const Frag8<T> gf = tr_rows(&Gz[hr_row(m0, sub)][16 * ct + 4 * pp], &Gz[hr_row(m0 + 4, sub)][16 * ct + 4 * pp]);
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) mma16(accW[o][k2], gf, af[k2]);
          mma16(accB[o], gf, ones);
        }
      }
    }
    M2T_TAIL_STAMP(9);
    __syncthreads();      // the tile buffers are free for the next stage
    M2T_TAIL_STAMP(10);
  }

  if constexpr (L1) {
    // (the tile buffers are free behind the loop's last barrier: eight floats of Gs carry the wave sums)
    const float ws = wave_sum(l1acc);
    float* red = reinterpret_cast<float*>(smem);
    if (lane == 0) red[w8] = ws;
    __syncthreads();
    if (tid == 0) a.loss_part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  }
  // ---- slabs ----
  {
    float* out = a.slab_wf + (long long)blockIdx.x * (32 * 64);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(16 * (w8 >> 2) + 4 * g + r) * 64 + 16 * (w8 & 3) + lr] = accF[r];
  }
  {
    float* out = a.slab_w3 + (long long)blockIdx.x * (256 * 64);
    float* outb = a.slab_b3 + (long long)blockIdx.x * 256;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(long long)(16 * (2 * w8 + o) + 4 * g + r) * 64 + 16 * k2 + lr] = accW[o][k2][r];
      if (lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) outb[16 * (2 * w8 + o) + 4 * g + r] = accB[o][r];
      }
    }
  }
}

// Geff of one tile pixel (ty, tx): part Q of NQ of its 32 gathered (tap, oc) values (27 used), written to dst[(32 / NQ) Q ..].
// The reflect ring of the tail conv (models/M2Trans_network.py:48): image rows / columns 1 and n - 2 are also read through the padding
// positions -1 and n, i.e. pixel row 1 receives, through tap ky = 0, the gradient of output row 0 as well (and row n - 2 through ky = 2
// that of row n - 1; columns likewise).  In halo coordinates the mirrored source of tap (ky, kx) is the entry the OPPOSITE tap reads
// (row ty + ky instead of ty - ky + 2), so the extra terms are up to three more reads per value at addresses that are always inside the
// halo: issued unconditionally and selected afterwards, on tiles that touch the image border only (`border` is wave-uniform).  The
// kernel of rounds 2-5 walked them in a divergent loop with one serialized LDS read per branch: ~15 k cycles per border tile.
template <int Q, int NQ> __device__ __forceinline__ void tb16_geff_part(const float (*Gs)[TB_HP], bf16_t* dst, int ty, int tx, int y0, int x0, int H, int W,
                                                                        bool border) {
  constexpr int NV = 32 / NQ;
  constexpr int P = TB_T + 2;
  float v[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int n = NV * Q + e;
    if (n < 27) {
      const int tap = n / 3, oc = n % 3, ky = tap / 3, kx = tap % 3;
      v[e] = Gs[oc][(ty - ky + 2) * P + (tx - kx + 2)];
    } else {
      v[e] = 0.f;
    }
  }
  if (border) {
    const int yy = y0 + ty, xx = x0 + tx;
    const bool r0 = yy == 1, r2 = yy == H - 2, c0 = xx == 1, c2 = xx == W - 2;
    float er[NV], ec[NV], ek[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const int n = NV * Q + e;
      er[e] = 0.f; ec[e] = 0.f; ek[e] = 0.f;
      if (n < 27) {
        const int tap = n / 3, oc = n % 3, ky = tap / 3, kx = tap % 3;
        if (ky != 1) er[e] = Gs[oc][(ty + ky) * P + (tx - kx + 2)];
        if (kx != 1) ec[e] = Gs[oc][(ty - ky + 2) * P + (tx + kx)];
        if (ky != 1 && kx != 1) ek[e] = Gs[oc][(ty + ky) * P + (tx + kx)];
      }
    }
#pragma unroll
    for (int e = 0; e < NV; ++e) {
      const int n = NV * Q + e;
      if (n < 27) {
        const int tap = n / 3, ky = tap / 3, kx = tap % 3;
        const bool rm = (ky == 0) ? r0 : ((ky == 2) ? r2 : false);
        const bool cm = (kx == 0) ? c0 : ((kx == 2) ? c2 : false);
        // (the order the divergent loop added them in: mirrored row, mirrored column, both)
        if (ky != 1) v[e] += rm ? er[e] : 0.f;
        if (kx != 1) v[e] += cm ? ec[e] : 0.f;
        if (ky != 1 && kx != 1) v[e] += (rm && cm) ? ek[e] : 0.f;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NV / 8; ++c) {
    float v8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] = v[8 * c + e];
    store8f(dst + NV * Q + 8 * c, v8);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 6: the recomputing kernel on v_mfma_f32_32x32x16_bf16 (tail_bwd32_kernel).
//
// What bounds the 16x16x32 kernel above was measured with a 16-wave partition of the same tile (scratch/k_tail_bwd_16waves_r06.hip.txt):
// with HALF the work per wave every phase took as long as before -- the phases run at the LDS rate, not at the issue rate.  A tile moves
// ~860 KB through LDS, and the ds_read_b128 that feeds a 16x16x32 operand (16 rows x 4 k-groups) is 2-way conflicted on every padded
// row-major image (its four 16-lane groups mix rows 0-3 | 12-15 of one k-group with rows 4-11 of the next): 128 of the 256 B/clk/CU.
// The 32x32x16 form needs half the operand bytes per FLOP (a 1 KB fragment feeds 32 K instead of 16 K FLOP), and its ds_read_b128
// (32 rows x 2 k-groups: every 16-lane group covers all 16 residues of the row index) is conflict-free on rows of an odd number of
// 16-byte slots.  So the kernel was rebuilt around that instruction:
//   * pixel order in LDS: p' = 64 sub + m (sub-pixel position major, mid pixel minor) instead of raster -- every operand row set is a
//     run of consecutive rows (the PixelShuffle gather hr_row() put rows 32 apart on the same banks);
//   * R  recompute t2^T = W3 a1^T + b3 (wave = 32 n' x all 64 mid pixels; the W3 fragments of a wave are strip constants in
//     registers), GELU / GELU' in registers, g(a2)^T = Wf^T Geff^T for the SAME (channel, pixel) accumulator layout, so
//     g(t2) = g(a2) gelu'(t2) is formed in registers: gelu'(t2) never goes to LDS; a2 and g(t2) are stored once;
//   * D | F + W in parallel, one wave of each kind per SIMD: waves 0-3 form g(t1)^T = W3^T g(u)^T (one 32 x 32 tile each, 16 k-steps
//     through a ring of 8 operand pairs); waves 4-7 accumulate dWf += Geff^T a2 (wave = 32 input channels x half of the tile's pixels, the
//     two halves folded behind the strip) and dW3 += g(u)^T a1 (four 32 x 32 tiles each).  The strip loop exists once per role, so neither
//     role's register allocation carries the other's accumulators;
//   * db3 from the R role's registers (per-lane sums of the rounded g(t2), one cross-lane reduction behind the strip);
//   * every operand fragment of a phase is requested before its first product (sched_barrier): with two waves per SIMD a load -> wait ->
//     MFMA chain costs an LDS round trip per k-step;
//   * the reflect-border terms of Geff without divergent loops (tb16_geff_part): the old walk cost ~15 k cycles per border tile;
//   * a1 and the g(sr) halo double-buffered: three barriers per tile instead of four.
// LDS traffic per tile ~390 KB, all row reads conflict-free.  Same roundings as the kernel above (a2, gelu'(t2), g(t2), g(t1) to bf16 at the
// same points) and the same products: g(t1), dW3 and the loss come out bit-identical (measured on every size of the tests), dWf / db3 are
// summed in another fp32 order.  Stand-alone at batch 16 (scratch/bench_tail_bwd16.hip, profiles/r06_tail_bwd_variants.txt): 346 us
// against 426 us, 658 against 805 at batch 32.  What is left is VALU issue: ~1 600 vector instructions per SIMD and tile (310 of each
// wave's 815 are the two GELU functions, 100 the bf16 conversions), ~55 % of the tile's cycles.
// ---------------------------------------------------------------------------------------------------------------------------
// At the register limit hipcc hoists every loop-invariant lane-dependent address out of the tile loop and spills some of them (round 5:
// a spilled value is reloaded by a VMEM operation that retires in order with the prefetches).  A laundered copy of the thread index per
// phase keeps the address arithmetic (a handful of VALU operations) inside the phase that uses it.
#define TB16_IDX(sfx)                                                                  \
  int tid##sfx = tid;                                                                  \
  asm volatile("" : "+v"(tid##sfx));                                                   \
  const int lane##sfx = tid##sfx & 63;                                                 \
  (void)lane##sfx

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mma32(f32x16& acc, const Frag8<bf16_t>& a, const Frag8<bf16_t>& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
// 32x32x16 operand from a CONTRACTION-major LDS image X[k][row | col] (ld elements per k row): lane (r = lane & 31, h = lane >> 5) receives
// X[k0 + 8 h + j][c0 + r], j = 0..7 (two transposing reads of a 4 x 16 block per 16-lane group).  EXEC must be all ones.
__device__ __forceinline__ Frag8<bf16_t> load8_tr32(const bf16_t* x, int ld, int k0, int c0, int lane) {
  const int i = lane & 15, q = i >> 2, pp = i & 3, gs = (lane >> 4) & 1, h = lane >> 5;
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  const bf16_t* p = x + (k0 + 8 * h + q) * ld + c0 + 16 * gs + 4 * pp;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(p + 4 * ld));
  Frag8<bf16_t> f;
  f.v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return f;
}
// accumulator register t of a 32x32 tile: row (t & 3) + 8 (t >> 2) + 4 h, column lane & 31
__device__ __forceinline__ constexpr int acc32_row(int t) { return (t & 3) + 8 * (t >> 2); }

#ifndef T32_KO
#define T32_KO 0                                 // scratch/bench_tail_bwd16.hip: knock-out experiments (results WRONG): 1 D loads, 2 D stores,
#endif                                           // 4 W loads, 8 GELU, 16 R stores, 32 Geff gather, 64 F loads

template <bool L1>
__global__ void __launch_bounds__(512) tail_bwd32_kernel(TailBwdArgs a) {
  using T = bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float(*Gs2)[3][TB_HP] = reinterpret_cast<float(*)[3][TB_HP]>(smem);          // 2 x [3][324] g(sr) halo, raster (0 outside the image)
  size_t off = sizeof(float) * 2 * 3 * TB_HP;
  T(*Ge)[40] = reinterpret_cast<T(*)[40]>(smem + off);  off += sizeof(T) * 256 * 40;      // Geff [p'][(tap,oc) -> 32]
  T(*Wt)[40] = reinterpret_cast<T(*)[40]>(smem + off);  off += sizeof(T) * 64 * 40;       // [ic][(tap,oc)]
  T(*A2)[TB_LD] = reinterpret_cast<T(*)[TB_LD]>(smem + off);  off += sizeof(T) * 256 * TB_LD;   // a2 [p'][c]
  T(*Gz)[TB_LD] = reinterpret_cast<T(*)[TB_LD]>(smem + off);  off += sizeof(T) * 256 * TB_LD;   // g(t2) [p'][c]
  T(*A12)[64][TB_LD] = reinterpret_cast<T(*)[64][TB_LD]>(smem + off);  off += sizeof(T) * 2 * 64 * TB_LD;    // 2 x a1 [m][k]
  T(*W3s)[264] = reinterpret_cast<T(*)[264]>(smem + off);  off += sizeof(T) * 64 * 264;         // W3^T [k][n'] (whole strip)
  float* B3s = reinterpret_cast<float*>(smem + off);                                            // tail.3 bias [n' = 64 sub + c]

  const int tid = threadIdx.x, lane = tid & 63, w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, Hm = H / 2, Wm = W / 2;
  const int tw = W / TB_T, th = H / TB_T;
  const long long hw = (long long)H * W;
  const int t0 = (int)xcd_block_index(), tstep = (int)gridDim.x, t1 = a.B * th * tw;

  for (int i = tid; i < 64 * 32; i += 512) {
    const int ic = i >> 5, n = i & 31;
    float v = 0.f;
    if (n < 27) v = a.wf[((n % 3) * 64 + ic) * 9 + n / 3];
    Wt[ic][n] = from_f<T>(v);
  }
  for (int i = tid; i < 64 * 32; i += 512) store8(&W3s[i >> 5][(i & 31) * 8], load8(a.w3t + (long long)(i >> 5) * 256 + (i & 31) * 8));
  if (tid < 256) B3s[tid] = a.b3[(tid & 63) * 4 + (tid >> 6)];          // torch order c * 4 + sub -> n' order
  __syncthreads();
  // ---- strip constants of this wave's R role: n' tile w8 = (sub, 32 channels cb ..) ----
  const int sub = w8 >> 1, cb = 32 * (w8 & 1);
  Frag8<T> w3a[4], wta[2];                       // A fragments: W3[n'][k] (k-steps of 16), Wf^T[c][(tap,oc)] (two k-steps)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) w3a[ks] = load8_tr32(&W3s[0][0], 264, 16 * ks, 32 * w8, lane);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) wta[ks] = load8(&Wt[cb + r32][16 * ks + 8 * h]);
  Frag8<T> ones_f;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones_f.set(e, 1.0f);
  (void)ones_f;
  // roles of the second half of a tile
  const int f_ic = w8 & 1, f_h = (w8 >> 1) & 1;  // F (waves 4-7): input-channel tile, pixel half (sub-pixel positions 2 f_h, 2 f_h + 1)
  const int d_kt = w8 & 1, d_mt = (w8 >> 1) & 1; // D (waves 0-3): output-channel tile, mid-pixel tile
  const int w_q = w8 & 3;                        // W (waves 4-7): n' tiles 2 w_q, 2 w_q + 1, both k tiles
  auto tile_geom = [&](int t, int& b, int& y0, int& x0) {
    const int q = t / tw, tx = t - q * tw;
    b = q / th;
    y0 = (q - b * th) * TB_T; x0 = tx * TB_T;
  };
  float* red = reinterpret_cast<float*>(smem);                          // [8] wave sums
  float* fold = reinterpret_cast<float*>(smem + 1024);                  // [2 ic tiles][16][64 lanes] fp32 = 8 KB: the upper pixel half's dWf partials

  // The strip loop exists TWICE, once per role of the tile's second half (ROLE 0 = D, waves 0-3; ROLE 1 = W, waves 4-7): in one
  // loop body the register allocation of every wave would carry the other role's accumulators (64 registers of dW3 tiles) and spill.
  // Both copies execute the same barriers in the same order.
  auto strip = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    f32x16 accF, accW[ROLE == 1 ? 4 : 1];         // (accF: ROLE 1 only)
    float accB[16];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) { accF[tt] = 0.f; accB[tt] = 0.f; }
#pragma unroll
    for (int q = 0; q < (ROLE == 1 ? 4 : 1); ++q)
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) accW[q][tt] = 0.f;
    Frag8<T> ra1;
    bf16x4 rd1[4];
    float rg[2], rh[2];
    float l1acc = 0.f;
    // Global addresses: a scalar (per tile) base + a 32-bit per-lane byte offset that does not depend on the tile -- the kernel is bound
    // by VALU issue, and the 64-bit per-lane address arithmetic of the five prefetches and four stores of a tile was a quarter of a wave's
    // vector instructions.  Only the g(sr) halo of a tile that touches the image border (or the crop's) needs per-lane clamping; its offsets
    // are formed under a scalar branch that contains NO load (a load under any branch is followed by s_waitcnt vmcnt(0) at the join).
    const unsigned a1_off = (unsigned)((((tid >> 3) >> 3) * Wm + ((tid >> 3) & 7)) * 128 + (tid & 7) * 16);
    unsigned hl_off[2], hh_off[2];                 // halo entry (oc, py, px): offsets from the halo origin in pre / gout and in hr
    int hl_oc[2], hl_py[2], hl_px[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int ic = min(tid + it * 512, 3 * TB_HP - 1);
      const int oc = ic / TB_HP, p = ic - oc * TB_HP;
      const int py = p / (TB_T + 2), px = p - py * (TB_T + 2);
      hl_off[it] = (unsigned)(((long long)oc * hw + (long long)py * W + px) * 4);
      hh_off[it] = (unsigned)((((long long)oc * a.Hs + py) * a.Ws + px) * 4);
      hl_oc[it] = oc; hl_py[it] = py; hl_px[it] = px;
    }
    const unsigned d_off = (unsigned)((((32 * d_mt + r32) >> 3) * Wm + ((32 * d_mt + r32) & 7)) * 128 + (32 * d_kt + 4 * h) * 2);
    auto fetchA = [&](int t) {              // staged through LDS: a1 tile, g(sr) halo
      int b, y0, x0;
      tile_geom(t, b, y0, x0);
      ra1 = load8(reinterpret_cast<const T*>(reinterpret_cast<const char*>(a.a1 + (((long long)b * Hm + y0 / 2) * Wm + x0 / 2) * 64) + a1_off));
      // the tile's halo inside the image AND (with the loss inside) inside the crop: no clamping, the offsets are the precomputed ones
      const bool fast = y0 >= TB_T && x0 >= TB_T && y0 + TB_T + 1 <= (L1 ? a.Hs : H) && x0 + TB_T + 1 <= (L1 ? a.Ws : W);
      unsigned ol[2], oh[2];
      if (fast) {
        const unsigned tl = (unsigned)(((y0 - 1) * W + (x0 - 1)) * 4), th2 = (unsigned)(((y0 - 1) * a.Ws + (x0 - 1)) * 4);
#pragma unroll
        for (int it = 0; it < 2; ++it) { ol[it] = tl + hl_off[it]; oh[it] = th2 + hh_off[it]; }
      } else {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int gy = y0 + hl_py[it] - 1, gx = x0 + hl_px[it] - 1;
          const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
          ol[it] = (unsigned)(((long long)hl_oc[it] * hw + (long long)cy * W + cx) * 4);
          oh[it] = (unsigned)((((long long)hl_oc[it] * a.Hs + min(cy, a.Hs - 1)) * a.Ws + min(cx, a.Ws - 1)) * 4);
        }
      }
      const char* lb = reinterpret_cast<const char*>((L1 ? a.pre : a.gout) + (long long)b * 3 * hw);
      const char* hb = reinterpret_cast<const char*>(a.hr + (long long)b * 3 * a.Hs * a.Ws);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        rg[it] = *reinterpret_cast<const float*>(lb + ol[it]);
        if constexpr (L1) rh[it] = *reinterpret_cast<const float*>(hb + oh[it]);
      }
    };
    auto fetchB = [&](int t) {              // D role: gelu'(t1) of this lane's mid pixel, its 16 output channels (4 x 8 bytes)
      int b, y0, x0;
      tile_geom(t, b, y0, x0);
      const T* dp = reinterpret_cast<const T*>(reinterpret_cast<const char*>(a.d1 + (((long long)b * Hm + y0 / 2) * Wm + x0 / 2) * 64) + d_off);
#pragma unroll
      for (int j = 0; j < 4; ++j) rd1[j] = *reinterpret_cast<const bf16x4*>(dp + 8 * j);
    };
    if (t0 < t1) { fetchA(t0); if constexpr (ROLE == 0) fetchB(t0); }

    int tile_it = 0;                                     // (read by the stamps of scratch/bench_tail_bwd16.hip only)
    for (int t = t0; t < t1; t += tstep, ++tile_it) {
      int b, y0, x0;
      tile_geom(t, b, y0, x0);
      // A1 and the g(sr) halo are double-buffered: the stage of tile t + 1 may begin while slower waves still read tile t's a1 (W role);
      // the barrier behind the stage is then the only one between two tiles
      float(*Gs)[TB_HP] = Gs2[tile_it & 1];
      T(*A1)[TB_LD] = A12[tile_it & 1];
      M2T_TAIL_STAMP2(0);
      // ---- stage ----
      {
        TB16_IDX(_s);
        store8(&A1[tid_s >> 3][(tid_s & 7) * 8], ra1);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int i = tid_s + it * 512;
          if (i < 3 * TB_HP) {
            const int p = i % TB_HP;
            const int py = p / (TB_T + 2), px = p % (TB_T + 2);
            const int gy = y0 + py - 1, gx = x0 + px - 1;
            if constexpr (L1) {
              float gv = 0.f;
              if (gy >= 0 && gy < a.Hs && gx >= 0 && gx < a.Ws) {
                const float v = rg[it];
                const float c = fminf(fmaxf(v, 0.f), a.R);
                const float d = c - rh[it];
                const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
                gv = (v >= 0.f && v <= a.R) ? sg * a.gscale : 0.f;
                if (py >= 1 && py <= TB_T && px >= 1 && px <= TB_T) l1acc += fabsf(d);     // the tile's OWN pixels: each pixel of the image once
              }
              Gs[i / TB_HP][p] = gv;
            } else {
              Gs[i / TB_HP][p] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? rg[it] : 0.f;      // 0 outside the image
            }
          }
        }
      }
      __syncthreads();
      M2T_TAIL_STAMP2(1);
      if (t + tstep < t1) fetchA(t + tstep);              // next tile's loads fly under this tile's products
      M2T_TAIL_STAMP2(2);
      // ---- Geff: two threads per tile pixel (raster order: conflict-free gathers), 16 of the 32 (tap, oc) values each (the half is the
      // role's: compile-time offsets); the row goes to p' = 64 sub + m ----
      {
        TB16_IDX(_e);
        const int p = tid_e & 255;
        const int ty = p >> 4, tx = p & 15;
        const int prow = 64 * ((ty & 1) * 2 + (tx & 1)) + 8 * (ty >> 1) + (tx >> 1);
        const bool border = y0 == 0 || x0 == 0 || y0 + TB_T == H || x0 + TB_T == W;        // (scalar: a tile that touches the image border)
        if constexpr (!(T32_KO & 32)) tb16_geff_part<ROLE, 2>(Gs, &Ge[prow][0], ty, tx, y0, x0, H, W, border);
      }
      M2T_TAIL_STAMP2(3);
      __syncthreads();
      M2T_TAIL_STAMP2(4);
      // ---- R: t2^T [32 n'][64 m], GELU, g(a2)^T for the same (channel, pixel) registers, g(t2) in registers.  Every operand fragment
      // of the phase is requested before the first product: with two waves per SIMD a load -> wait -> MFMA chain per k-step costs an LDS
      // round trip (~250 cycles) per step (measured on the 16x16x32 kernel: its phases took as long with half the work per wave) ----
      {
        TB16_IDX(_r);
        const int r32_r = lane_r & 31, h_r = lane_r >> 5;
        Frag8<T> fb[2][4], fg[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) fb[mt][ks] = load8(&A1[32 * mt + r32_r][16 * ks + 8 * h_r]);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) fg[mt][ks] = load8(&Ge[64 * sub + 32 * mt + r32_r][16 * ks + 8 * h_r]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          if (mt == 1) M2T_TAIL_STAMP2(11);
          f32x16 acc, acg;
#pragma unroll
          for (int tt = 0; tt < 16; ++tt) { acc[tt] = 0.f; acg[tt] = 0.f; }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) mma32(acc, w3a[ks], fb[mt][ks]);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) mma32(acg, wta[ks], fg[mt][ks]);
          const int prow = 64 * sub + 32 * mt + r32_r;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float av[4], dv[4], gv[4];
            const f32x4 x4 = (f32x4){acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&B3s[64 * sub + cb + 8 * j + 4 * h_r]);     // (registers are scarcer than LDS reads here)
            const float b4[4] = {bq[0], bq[1], bq[2], bq[3]};
            if constexpr (T32_KO & 8) {
#pragma unroll
              for (int i = 0; i < 4; ++i) { av[i] = x4[i] + b4[i]; dv[i] = x4[i] - b4[i]; }
            } else
            gelu_tail_both4<T>(x4, b4, av, dv);
            // roundings where the other kernels store: gelu'(t2) before the product, g(t2) before it is stored and summed into db3.  Packed
            // conversions, and the rounded BITS are what is stored (one v_cvt_pk per pair and role instead of a convert / widen / convert chain)
            unsigned gq[2];
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
              const bf16x2 dp = {(T)dv[2 * i2], (T)dv[2 * i2 + 1]};
              const unsigned du = __builtin_bit_cast(unsigned, dp);
              const float g0 = acg[4 * j + 2 * i2] * __builtin_bit_cast(float, du << 16);
              const float g1 = acg[4 * j + 2 * i2 + 1] * __builtin_bit_cast(float, du & 0xffff0000u);
              const bf16x2 gp = {(T)g0, (T)g1};
              gq[i2] = __builtin_bit_cast(unsigned, gp);
              accB[4 * j + 2 * i2] += __builtin_bit_cast(float, gq[i2] << 16);
              accB[4 * j + 2 * i2 + 1] += __builtin_bit_cast(float, gq[i2] & 0xffff0000u);
            }
            if constexpr (T32_KO & 16) { if (av[0] == 123.f) store4(&A2[prow][cb + 8 * j + 4 * h_r], av); continue; }
            store4(&A2[prow][cb + 8 * j + 4 * h_r], av);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<u32x2*>(&Gz[prow][cb + 8 * j + 4 * h_r]) = (u32x2){gq[0], gq[1]};
          }
        }
      }
      M2T_TAIL_STAMP2(5);
      __syncthreads();      // a2 and g(t2) complete
      M2T_TAIL_STAMP2(6);
      // ---- F (W waves): dWf[(tap,oc)][ic] += Geff^T a2 over the 128 pixels of sub-pixel positions 2 f_h, 2 f_h + 1 ----
      if constexpr (ROLE == 1) {
        TB16_IDX(_f);
        Frag8<T> fa[8], fb[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if constexpr (T32_KO & 64) { fa[ks] = ones_f; fb[ks] = ones_f; continue; }
          fa[ks] = load8_tr32(&Ge[0][0], 40, 128 * f_h + 16 * ks, 0, lane_f);
          fb[ks] = load8_tr32(&A2[0][0], TB_LD, 128 * f_h + 16 * ks, 32 * f_ic, lane_f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) mma32(accF, fa[ks], fb[ks]);
      }
      M2T_TAIL_STAMP2(7);
      if constexpr (ROLE == 0) {
        // ---- D: g(t1)^T [32 k][32 m] = sum_n' W3^T[k][n'] g(u)[m][n'], n' = 64 sb + c ----
        TB16_IDX(_d);
        const int r32_d = lane_d & 31, h_d = lane_d >> 5;
        f32x16 acd;
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) acd[tt] = 0.f;
        // 16 k-steps through a ring of 8: the operands of step s + 8 are requested behind the product of step s
        Frag8<T> fa[8], fb[8];
        auto ldD = [&](int ks, Frag8<T>& xa, Frag8<T>& xb) {
          const int sb = ks >> 2, c0 = 16 * (ks & 3) + 8 * h_d;
          if constexpr (T32_KO & 1) { xa = ones_f; xb = ones_f; return; }
          xa = load8(&W3s[32 * d_kt + r32_d][64 * sb + c0]);
          xb = load8(&Gz[64 * sb + 32 * d_mt + r32_d][c0]);
        };
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) ldD(ks, fa[ks], fb[ks]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          mma32(acd, fa[ks], fb[ks]);
          ldD(ks + 8, fa[ks], fb[ks]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) mma32(acd, fa[ks], fb[ks]);
        M2T_TAIL_STAMP2(12);
        T* gp = reinterpret_cast<T*>(reinterpret_cast<char*>(a.gt1 + (((long long)b * Hm + y0 / 2) * Wm + x0 / 2) * 64) + d_off);
        if constexpr (!(T32_KO & 2)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = acd[4 * j + i] * (float)rd1[j][i];
          store4(gp + 8 * j, v);
        }
        M2T_TAIL_STAMP2(13);
        if (t + tstep < t1) fetchB(t + tstep);
        } else { if (acd[0] == 123.f) gp[0] = (T)1.f; }
      } else {
        // ---- W: dW3[n'][k] += g(u)^T a1 (contraction over the 64 mid pixels): n' tiles 2 w_q, 2 w_q + 1 x k tiles 0, 1 ----
        TB16_IDX(_w);
        Frag8<T> fb[4][2], fa[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if constexpr (T32_KO & 4) { fb[ks][0] = ones_f; fb[ks][1] = ones_f; fa[ks][0] = ones_f; fa[ks][1] = ones_f; continue; }
          fb[ks][0] = load8_tr32(&A1[0][0], TB_LD, 16 * ks, 0, lane_w);
          fb[ks][1] = load8_tr32(&A1[0][0], TB_LD, 16 * ks, 32, lane_w);
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            const int nt = 2 * w_q + o;                                  // n' tile: sub nt >> 1, channels 32 (nt & 1) ..
            fa[ks][o] = load8_tr32(&Gz[0][0], TB_LD, 64 * (nt >> 1) + 16 * ks, 32 * (nt & 1), lane_w);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            mma32(accW[2 * o], fa[ks][o], fb[ks][0]);
            mma32(accW[2 * o + 1], fa[ks][o], fb[ks][1]);
          }
      }
      M2T_TAIL_STAMP2(8);
      M2T_TAIL_STAMP2(9);
      M2T_TAIL_STAMP2(10);
    }

    // ---- behind the strip: loss partial, db3, the dWf partials, slabs ----
    __syncthreads();      // (every wave has left its last tile: the tile buffers are free)
    if constexpr (L1) {
      const float ws = wave_sum(l1acc);
      if (lane == 0) red[w8] = ws;
    }
    if constexpr (ROLE == 1) {
      if (f_h == 1) {
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) fold[(f_ic * 16 + tt) * 64 + lane] = accF[tt];
      }
    }
    __syncthreads();
    if constexpr (L1) {
      if (tid == 0) a.loss_part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
    if constexpr (ROLE == 1) {
      if (f_h == 0) {
        float* out = a.slab_wf + (long long)blockIdx.x * (32 * 64);
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) out[(acc32_row(tt) + 4 * h) * 64 + 32 * f_ic + r32] = accF[tt] + fold[(f_ic * 16 + tt) * 64 + lane];
      }
    }
    {
      // db3[n'] = sum over this workgroup's pixels of g(u)[.][n']: the lanes of a half hold the same 16 channels for 32 different pixels
      float* outb = a.slab_b3 + (long long)blockIdx.x * 256;
#pragma unroll
      for (int tt = 0; tt < 16; ++tt) {
        float v = accB[tt];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (r32 == 0) outb[64 * sub + cb + acc32_row(tt) + 4 * h] = v;
      }
    }
    if constexpr (ROLE == 1) {
      float* out = a.slab_w3 + (long long)blockIdx.x * (256 * 64);
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int tt = 0; tt < 16; ++tt)
            out[(long long)(32 * (2 * w_q + o) + acc32_row(tt) + 4 * h) * 64 + 32 * kt + r32] = accW[2 * o + kt][tt];
    }
  };
  if (w8 < 4) strip(std::integral_constant<int, 0>{});
  else        strip(std::integral_constant<int, 1>{});
}

constexpr size_t tail_bwd_smem() {
  return sizeof(float) * 3 * TB_HP + sizeof(bf16_t) * (256 * 40 + 64 * 40 + 2 * 256 * TB_LD + 64 * TB_LD + 64 * 264);
}
constexpr size_t tail_bwd32_smem() {      // + the second a1 tile and g(sr) halo, the bias table
  return tail_bwd_smem() + sizeof(float) * 3 * TB_HP + sizeof(bf16_t) * 64 * TB_LD + sizeof(float) * 256;
}
static_assert(tail_bwd32_smem() <= 160 * 1024, "one workgroup per CU");

}  // namespace

int tail_bwd_fused_blocks(int B, int H, int W) {
  const long long ntiles = (long long)B * (H / TB_T) * (W / TB_T);
  return (int)std::min<long long>(256, ntiles);
}

// bf16 only.  H, W: high-resolution size (multiples of 32).  nslab_out: slabs written (same count for the three sets).
int launch_tail_bwd_fused(const float* gout, const float* wf, const void* act, const void* der, const void* a1, const void* d1,
                          const void* w3t, const float* b3, void* gt1, float* slab_wf, float* slab_w3, float* slab_b3, int* nslab_out,
                          int B, int H, int W, hipStream_t st, const float* l1_pre, const float* l1_hr, float* l1_part, int Hs, int Ws,
                          float R, float gscale, int variant) {
  // act == nullptr: the forward did not store gelu(t2) / gelu'(t2); they are recomputed per tile from a1, w3t and b3
  // l1_pre != nullptr: the clamp + L1 seed is taken inside (gout unused); l1_part [tail_bwd_fused_blocks] receives the loss partials
  if (H % 32 || W % 32) return m2t_set_error(-2, "tail_bwd_fused: H, W must be multiples of 32");
  const long long ntiles = (long long)B * (H / TB_T) * (W / TB_T);
  (void)ntiles;
  const int nblk = tail_bwd_fused_blocks(B, H, W);
  const size_t sh = tail_bwd_smem();
  TailBwdArgs a{gout, wf, (const bf16_t*)act, (const bf16_t*)der, (const bf16_t*)a1, (const bf16_t*)d1, (const bf16_t*)w3t, b3,
                (bf16_t*)gt1, slab_wf, slab_w3, slab_b3, B, H, W, l1_pre, l1_hr, l1_part, Hs, Ws, R, gscale};
  if (l1_pre && (!l1_hr || !l1_part || act != nullptr)) return m2t_set_error(-2, "tail_bwd_fused: the fused L1 seed needs hr, the partial buffer and the recomputing variant");
#define TB_GO(RC_, L1_)                                                                                                    \
  do {                                                                                                                      \
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_bwd_fused_kernel<RC_, L1_>, (int)sh)) return rc__;               \
    M2T_LAUNCH_TIMED((tail_bwd_fused_kernel<RC_, L1_>), dim3(nblk), dim3(512), sh, st, a);                                   \
  } while (0)
#define TB_GO32(L1_)                                                                                                       \
  do {                                                                                                                      \
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_bwd32_kernel<L1_>, (int)tail_bwd32_smem())) return rc__;         \
    M2T_LAUNCH_TIMED((tail_bwd32_kernel<L1_>), dim3(nblk), dim3(512), tail_bwd32_smem(), st, a);                             \
  } while (0)
  if (act == nullptr) {
    if (!b3) return m2t_set_error(-2, "tail_bwd_fused: the recomputing variant needs the tail.3 bias");
    if (variant == 32) { if (l1_pre) TB_GO32(true); else TB_GO32(false); }
    else if (l1_pre) TB_GO(true, true); else TB_GO(true, false);
  } else {
    TB_GO(false, false);
  }
#undef TB_GO
#undef TB_GO32
  M2T_LAUNCH_CHECK();
  *nslab_out = nblk;
  return 0;
}
