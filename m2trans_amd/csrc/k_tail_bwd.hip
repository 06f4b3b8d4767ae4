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

#ifndef M2T_TAIL_STAMP
#define M2T_TAIL_STAMP(i) do { } while (0)       // scratch/bench_tail.hip defines it to record s_memtime per phase
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

constexpr size_t tail_bwd_smem() {
  return sizeof(float) * 3 * TB_HP + sizeof(bf16_t) * (256 * 40 + 64 * 40 + 2 * 256 * TB_LD + 64 * TB_LD + 64 * 264);
}

}  // namespace

int tail_bwd_fused_blocks(int B, int H, int W) {
  const long long ntiles = (long long)B * (H / TB_T) * (W / TB_T);
  return (int)std::min<long long>(256, ntiles);
}

// bf16 only.  H, W: high-resolution size (multiples of 32).  nslab_out: slabs written (same count for the three sets).
int launch_tail_bwd_fused(const float* gout, const float* wf, const void* act, const void* der, const void* a1, const void* d1,
                          const void* w3t, const float* b3, void* gt1, float* slab_wf, float* slab_w3, float* slab_b3, int* nslab_out,
                          int B, int H, int W, hipStream_t st, const float* l1_pre, const float* l1_hr, float* l1_part, int Hs, int Ws,
                          float R, float gscale) {
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
  if (act == nullptr) {
    if (!b3) return m2t_set_error(-2, "tail_bwd_fused: the recomputing variant needs the tail.3 bias");
    if (l1_pre) TB_GO(true, true); else TB_GO(true, false);
  } else {
    TB_GO(false, false);
  }
#undef TB_GO
  M2T_LAUNCH_CHECK();
  *nslab_out = nblk;
  return 0;
}
