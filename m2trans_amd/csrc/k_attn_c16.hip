// k_attn_c16.hip -- the 8x8 / 10x10 halo window attention of the C = 16 branch (bf16), backward and forward:
// ONE WAVE PER WINDOW, no workgroup barriers.
//
// Same mathematics as window_attn_bwd_kernel (k_attn.hip; models/M2Trans_network.py:310-332 under autograd).
// The full-resolution branch has 4096 windows per 16-patch batch and only 16 channels: a workgroup per window
// spends its time in barriers and exposed latencies (57 us per launch; this kernel: 33 us, bound by the
// issue rate of the softmax VALU work), while the arithmetic is tiny.  Here a
// 64-lane wave owns a window and four independent waves share a workgroup:
//   * with K = 16 the contraction of S^T = K^ Q^T and dP^T = V dO^T is exactly one v_mfma_f32_16x16x16_bf16,
//     whose operands (4 channels per lane) are loaded straight from HBM in operand layout -- no staging;
//   * the accumulator layout of a 16x16 tile (lane holds rows 4g..4g+3 of column l) IS the B-operand layout
//     of the next product when the contraction runs over the tile's rows: dS^T feeds dq from registers;
//   * for the products that contract over queries (dV, dK^) the P^T / dS^T tiles take one 8-byte LDS write
//     per lane ([query][key] rows) and come back through the transposing read ds_read_b64_tr_b16;
//   * K^, q and dO are also kept row-major in (wave-private) LDS so that the channel-major operands of
//     dq / dV / dK^ are one transposing read each;
//   * LDS accesses of a wave execute in order, so wave-private buffers need no barrier at all.
// 15 KB of LDS per wave: nothing here can starve (or be starved by) the parameter-gradient kernels.
#include "m2t_kernels.h"
#include "m2t_instnorm.h"
#include "m2t_window.h"

namespace {

constexpr int C16 = 16;
constexpr int C16_PLD = 120;   // P / dS rows: 112 keys + 8 pad

struct __attribute__((aligned(16))) C16WaveLds {
  bf16_t Kh[112][16];          // K^ = k + rel-pos, row-major [key][c]   (keys >= 100 are zero)
  bf16_t Qs[64][16];           // q  [query][c]
  bf16_t DOs[64][16];          // dO [query][c]
  bf16_t Pq[16][C16_PLD];      // P  of the current query tile, [query][key]
  bf16_t Dq[16][C16_PLD];      // dS of the current query tile
};
static_assert(sizeof(C16WaveLds) == 15360, "wave LDS layout");
static_assert(sizeof(float) * 16 * 113 <= sizeof(bf16_t) * (112 + 64 + 64) * 16, "KA must fit over Kh | Qs | DOs");

__device__ __forceinline__ void mma4(f32x4& acc, bf16x4 a, bf16x4 b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc, 0, 0, 0);
}
// transposing read: the lane passes the address of ITS 8-byte piece (row r0 + (i >> 2), columns 4 (i & 3) ..)
// and receives column i of rows r0 .. r0 + 3  (i = lane & 15; r0 may differ per 16-lane group)
__device__ __forceinline__ bf16x4 tr4(const bf16_t* p) {
  typedef bf16x4 __attribute__((address_space(3))) * lds_ptr;
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)p);
}
__device__ __forceinline__ bf16x4 ld4(const bf16_t* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ void st4(bf16_t* p, bf16x4 v) { *reinterpret_cast<bf16x4*>(p) = v; }
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16_t)a; v[1] = (bf16_t)b; v[2] = (bf16_t)c; v[3] = (bf16_t)d;
  return v;
}
__device__ __forceinline__ bf16x4 zero4() { return pack4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ int lr_of(int lane) { return lane & 15; }
__device__ __forceinline__ int g_of(int lane) { return lane >> 4; }
// LDS operations of one wave execute in issue order, so a wave-private buffer needs no hardware wait between
// its writes and the reads of other lanes -- only the compiler must not reorder them (a wavefront-scope fence
// would also wait for the outstanding GLOBAL stores: measured 2x slower).
__device__ __forceinline__ void wave_sync() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

__global__ void __launch_bounds__(256, 2) window_attn_bwd_c16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                  const float* __restrict__ rel_w, const bf16_t* __restrict__ go,
                                                                  int ldg, int gc0, bf16_t* __restrict__ gqkv, bf16_t* __restrict__ win,
                                                                  float* __restrict__ relw, int h, int w, int nwin,
                                                                  const bf16_t* __restrict__ dsrc, const bf16_t* __restrict__ wqkv) {
  // dsrc != nullptr: q | k | v were NOT saved by the forward pass; they are recomputed here from the branch input d
  // [pixel][16] (saved anyway: the weight-gradient GEMM reads it) and the packed weight wqkv [48][16], with the forward kernel's
  // own MFMAs (window_attn_fused_c16_fwd_kernel: out^T = W x^T, one v_mfma_f32_16x16x16_bf16 per tile, rounded to bf16 as
  // it was when stored) -- identical bits, 11 instead of 18 eight-byte loads per lane, and 25 MB less written by the forward
  // and read here per launch at batch 16
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = xcd_block_index() * 4 + wv;
  if (wi >= nwin) return;                         // no workgroup barrier anywhere: a wave may leave alone
  C16WaveLds& L = reinterpret_cast<C16WaveLds*>(smem)[wv];
  const int lr = lane & 15, g = lane >> 4;
  const int nw = w / 8, nh = h / 8;
  const int wx = wi % nw, wy = (wi / nw) % nh, b = wi / (nw * nh);
  const long long img = (long long)b * h * w;

  // ---- every global load of the window, in MFMA operand layout: row = 16 t + lr, channels 4g .. 4g+3 ----
  bf16x4 kA[WA_KT], vA[WA_KT];
  {
    bf16x4 kraw[WA_KT];
    f32x4 rel[WA_KT];
    bf16x4 xk[WA_KT], xq[4];
    if (dsrc) {                                   // wave-uniform: branch-free clamped loads of the d rows (keys, then queries)
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        const int key = min(16 * t + lr, WA_NK - 1);
        const int kr = key / 10, kc = key - kr * 10;
        const int y = 8 * wy + kr - 1, x = 8 * wx + kc - 1;
        const bool in = (16 * t + lr < WA_NK) && y >= 0 && y < h && x >= 0 && x < w;
        const int yc = min(max(y, 0), h - 1), xc = min(max(x, 0), w - 1);
        const bf16x4 v = ld4(dsrc + (img + (long long)yc * w + xc) * C16 + 4 * g);
        xk[t] = in ? v : zero4();
      }
#pragma unroll
      for (int qt = 0; qt < 4; ++qt) {
        const int q = 16 * qt + lr;
        xq[qt] = ld4(dsrc + (img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7)) * C16 + 4 * g);
      }
    }
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = 16 * t + lr;
      const int kr = key / 10, kc = key - kr * 10;
      const int y = 8 * wy + kr - 1, x = 8 * wx + kc - 1;
      kraw[t] = zero4();
      vA[t] = zero4();
      rel[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (key < WA_NK) {
        if (!dsrc && y >= 0 && y < h && x >= 0 && x < w) {
          const bf16_t* pk = qkv + (img + (long long)y * w + x) * (3 * C16) + 4 * g;
          kraw[t] = ld4(pk + C16);
          vA[t] = ld4(pk + 2 * C16);
        }
        const float* rp = (g < 2) ? (rel_h + kr * (C16 / 2) + 4 * g) : (rel_w + kc * (C16 / 2) + 4 * g - C16 / 2);
        rel[t] = *reinterpret_cast<const f32x4*>(rp);
      }
    }
    bf16x4 wA[3];                                 // rows 16 n + lr of [q | k | v], input channels 4g ..
    if (dsrc) {
#pragma unroll
      for (int n = 0; n < 3; ++n) wA[n] = ld4(wqkv + (16 * n + lr) * C16 + 4 * g);
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {           // out-of-image and padding keys: x = 0 -> k = v = 0, as F.unfold pads
        f32x4 ak = (f32x4){0.f, 0.f, 0.f, 0.f}, av = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma4(ak, wA[1], xk[t]);
        mma4(av, wA[2], xk[t]);
        kraw[t] = pack4(ak[0], ak[1], ak[2], ak[3]);
        vA[t] = pack4(av[0], av[1], av[2], av[3]);
      }
    }
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
      const int q = 16 * qt + lr;
      const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
      bf16x4 qv;
      if (dsrc) {
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
        mma4(a, wA[0], xq[qt]);
        qv = pack4(a[0], a[1], a[2], a[3]);
      } else {
        qv = ld4(qkv + qpix * (3 * C16) + 4 * g);
      }
      st4(&L.Qs[q][4 * g], pack4(0.25f * (float)qv[0], 0.25f * (float)qv[1], 0.25f * (float)qv[2], 0.25f * (float)qv[3]));   // exact
      st4(&L.DOs[q][4 * g], ld4(go + qpix * ldg + gc0 + 4 * g));
    }
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      kA[t] = pack4((float)kraw[t][0] + rel[t][0], (float)kraw[t][1] + rel[t][1], (float)kraw[t][2] + rel[t][2],
                    (float)kraw[t][3] + rel[t][3]);
      st4(&L.Kh[16 * t + lr][4 * g], kA[t]);
    }
  }
  wave_sync();
  // channel-major K^ operands (rows = channel lr, contraction = keys 16 t + 4g ..)
  bf16x4 kT[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) kT[t] = tr4(&L.Kh[16 * t + 4 * g + (lr >> 2)][4 * (lr & 3)]);

  f32x4 dvT[WA_KT], dkT[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) { dvT[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkT[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  // C^-1/2 = 1/4 is a power of two: q is pre-scaled exactly when it is staged, so S^T arrives scaled, dK^ = dS'^T q
  // needs no further factor (dS' = P (dP - delta), unscaled) and dq = dS' K^ takes the factor on its 4 outputs.
  const float scale = 0.25f;
  const f32x4 L2E = (f32x4){1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f};

#pragma unroll 1
  for (int qt = 0; qt < 4; ++qt) {
    const bf16x4 qB = ld4(&L.Qs[16 * qt + lr][4 * g]);
    const bf16x4 gB = ld4(&L.DOs[16 * qt + lr][4 * g]);
    const bf16x4 qT = tr4(&L.Qs[16 * qt + 4 * g + (lr >> 2)][4 * (lr & 3)]);
    const bf16x4 gT = tr4(&L.DOs[16 * qt + 4 * g + (lr >> 2)][4 * (lr & 3)]);
    // S^T and dP^T for this query tile: rows = keys 16 t + 4g + r, column = query lr
    f32x4 s[WA_KT], dp[WA_KT];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma4(s[t], kA[t], qB);
      mma4(dp[t], vA[t], gB);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r >= 4) s[6][r] = -3.0e38f;                  // keys 100..111 (only the last tile has any)
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float nml = -mx * 1.4426950408889634f;
    const f32x4 NM = (f32x4){nml, nml, nml, nml};
    f32x4 sum4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 x = s[t] * L2E + NM;                          // packed fma; exp(s - mx) = 2^x
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(x[r]);   // masked keys: 2^-huge = 0
      sum4 += s[t];
    }
    f32x4 del4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) del4 += s[t] * dp[t];       // sum_keys e * dP (normalised below)
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    float delta = (del4[0] + del4[1]) + (del4[2] + del4[3]);
    {                                                           // the two reductions travel together
      const float s1 = __shfl_xor(sum, 16), d1 = __shfl_xor(delta, 16);
      sum += s1; delta += d1;
      const float s2 = __shfl_xor(sum, 32), d2 = __shfl_xor(delta, 32);
      sum += s2; delta += d2;
    }
    const float inv = 1.0f / sum;
    delta *= inv;
    const f32x4 INV = (f32x4){inv, inv, inv, inv};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) s[t] *= INV;                // P
    const f32x4 DEL = (f32x4){delta, delta, delta, delta};
    // dq^T [c][q] = scale * sum_keys K^[key][c] dS'^T[key][q], straight from the accumulator layout
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 d4 = s[t] * (dp[t] - DEL);
      const bf16x4 pb = pack4(s[t][0], s[t][1], s[t][2], s[t][3]);
      const bf16x4 db = pack4(d4[0], d4[1], d4[2], d4[3]);
      st4(&L.Pq[lr][16 * t + 4 * g], pb);
      st4(&L.Dq[lr][16 * t + 4 * g], db);
      mma4(o, kT[t], db);
    }
    o *= (f32x4){scale, scale, scale, scale};
    {
      const int q = 16 * qt + lr;
      const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
      st4(gqkv + qpix * (3 * C16) + 4 * g, pack4(o[0], o[1], o[2], o[3]));
    }
    wave_sync();
    // dV^T [c][key] += dO^T P ; dK^^T [c][key] += q^T dS   (contraction over this tile's 16 queries)
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const bf16x4 pB = tr4(&L.Pq[4 * g + (lr >> 2)][16 * t + 4 * (lr & 3)]);
      const bf16x4 dB = tr4(&L.Dq[4 * g + (lr >> 2)][16 * t + 4 * (lr & 3)]);
      mma4(dvT[t], gT, pB);
      mma4(dkT[t], qT, dB);
    }
    wave_sync();
  }

  // ---- dK^ | dV rows of this window: lane holds channels 4g .. 4g+3 of key 16 t + lr ----
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = 16 * t + lr;
    if (key < WA_NK) {
      bf16_t* wp = dkv_row(gqkv, win, (long long)wi, b, wy, wx, h, w, C16, key) + 4 * g;
      st4(wp, pack4(dkT[t][0], dkT[t][1], dkT[t][2], dkT[t][3]));
      st4(wp + C16, pack4(dvT[t][0], dvT[t][1], dvT[t][2], dvT[t][3]));
    }
  }
  // ---- rel-pos gradient: dK^ summed over key columns (c < 8) / key rows (c >= 8), phantom keys included ----
  float(*KA)[113] = reinterpret_cast<float(*)[113]>(&L);
#pragma unroll
  for (int t = 0; t < WA_KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) KA[4 * g + r][16 * t + lr] = dkT[t][r];
  wave_sync();
  for (int idx = lane; idx < 10 * C16; idx += 64) {
    const int i = idx >> 4, c = idx & 15;
    float a = 0.f;
    if (c < C16 / 2) {
#pragma unroll
      for (int j = 0; j < 10; ++j) a += KA[c][i * 10 + j];
    } else {
#pragma unroll
      for (int j = 0; j < 10; ++j) a += KA[c][j * 10 + i];
    }
    relw[((long long)wi * 10 + i) * C16 + c] = a;
  }
}

// ---------------------------------------------------------------------------------------
// forward, same organisation: S^T = K^ Q^T (one MFMA per tile), softmax on the accumulators, and
// O^T = V^T P^T with P^T fed back from the accumulator layout; V is the only tensor that touches LDS
// (row-major copy, read back channel-major by ds_read_b64_tr_b16).  out[q][oc0 + c] = O (+ res).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) window_attn_fwd_c16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                     const float* __restrict__ rel_w, bf16_t* __restrict__ out, int ldo,
                                                                     int oc0, const bf16_t* __restrict__ res, int ldr, int h, int w,
                                                                     int nwin) {
  __shared__ __attribute__((aligned(16))) bf16_t VsAll[4][112][16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = xcd_block_index() * 4 + wv;
  if (wi >= nwin) return;
  bf16_t(*Vs)[16] = VsAll[wv];
  const int lr = lane & 15, g = lane >> 4;
  const int nw = w / 8, nh = h / 8;
  const int wx = wi % nw, wy = (wi / nw) % nh, b = wi / (nw * nh);
  const long long img = (long long)b * h * w;
  // every global load of the window is issued up front: q and the residual used to be loaded inside the query-tile
  // loop, four dependent round trips of ~2 us each on a wave that has nothing else to run
  bf16x4 qraw[4], rraw[4];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int q = 16 * qt + lr;
    const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
    qraw[qt] = ld4(qkv + qpix * (3 * C16) + 4 * g);
    rraw[qt] = res ? ld4(res + qpix * ldr + 4 * g) : zero4();
  }
  bf16x4 kA[WA_KT];
  {
    bf16x4 kraw[WA_KT], vraw[WA_KT];
    f32x4 rel[WA_KT];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const int key = 16 * t + lr;
      const int kr = key / 10, kc = key - kr * 10;
      const int y = 8 * wy + kr - 1, x = 8 * wx + kc - 1;
      kraw[t] = zero4();
      vraw[t] = zero4();
      rel[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (key < WA_NK) {
        if (y >= 0 && y < h && x >= 0 && x < w) {
          const bf16_t* pk = qkv + (img + (long long)y * w + x) * (3 * C16) + 4 * g;
          kraw[t] = ld4(pk + C16);
          vraw[t] = ld4(pk + 2 * C16);
        }
        const float* rp = (g < 2) ? (rel_h + kr * (C16 / 2) + 4 * g) : (rel_w + kc * (C16 / 2) + 4 * g - C16 / 2);
        rel[t] = *reinterpret_cast<const f32x4*>(rp);
      }
    }
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      kA[t] = pack4((float)kraw[t][0] + rel[t][0], (float)kraw[t][1] + rel[t][1], (float)kraw[t][2] + rel[t][2],
                    (float)kraw[t][3] + rel[t][3]);
      st4(&Vs[16 * t + lr][4 * g], vraw[t]);
    }
  }
  wave_sync();
  bf16x4 vT[WA_KT];                               // rows = channel lr, contraction = keys 16 t + 4g ..
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) vT[t] = tr4(&Vs[16 * t + 4 * g + (lr >> 2)][4 * (lr & 3)]);
  const f32x4 L2E = (f32x4){1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f};
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int q = 16 * qt + lr;
    const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
    const bf16x4 qv = qraw[qt];
    const bf16x4 qB = pack4(0.25f * (float)qv[0], 0.25f * (float)qv[1], 0.25f * (float)qv[2], 0.25f * (float)qv[3]);   // C^-1/2, exact
    f32x4 s[WA_KT];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma4(s[t], kA[t], qB);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r >= 4) s[6][r] = -3.0e38f;                  // keys 100..111
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float nml = -mx * 1.4426950408889634f;
    const f32x4 NM = (f32x4){nml, nml, nml, nml};
    f32x4 sum4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 x = s[t] * L2E + NM;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(x[r]);
      sum4 += s[t];
    }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    const f32x4 INV = (f32x4){inv, inv, inv, inv};
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 pv = s[t] * INV;
      mma4(o, vT[t], pack4(pv[0], pv[1], pv[2], pv[3]));
    }
    if (res) {
      const bf16x4 rv = rraw[qt];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += (float)rv[r];
    }
    st4(out + qpix * ldo + oc0 + 4 * g, pack4(o[0], o[1], o[2], o[3]));
  }
}


// ---------------------------------------------------------------------------------------
// fused forward of the whole C = 16 branch (models/M2Trans_network.py:135-139,281,307-332): InstanceNorm apply of
// chunk 0, the 16 -> 48 qkv projection, the window attention and the residual, one wave per window.  The unfused
// sequence is three launches (branch_prep<0>, the K = 16 GEMM, the kernel above) that move d1 and qkv through HBM
// between them.  Here a wave loads the raw x rows of its 100 keys and 64 queries in MFMA operand layout (4 channels
// per lane), normalises them in registers, and gets k, v (7 key tiles) and q (4 query tiles) with one
// v_mfma_f32_16x16x16_bf16 each: the accumulator layout of out^T = W x^T (lane = pixel, 4 consecutive output channels)
// IS the operand layout the attention part reads from HBM in the unfused kernel.  d1 (= the residual) and qkv of the
// window's own pixels are still written: the backward pass reads them.  Out-of-image halo keys are the zero padding of
// the normalised map (x^ = 0 -> k = v = 0, key = rel-pos alone).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 4) window_attn_fused_c16_fwd_kernel(const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                                           const float* __restrict__ rstd, const bf16_t* __restrict__ wqkv,
                                                                           const float* __restrict__ rel_h, const float* __restrict__ rel_w,
                                                                           bf16_t* __restrict__ d, bf16_t* __restrict__ qkv,
                                                                           bf16_t* __restrict__ out, int ldo, int oc0, int h, int w, int nwin) {
  __shared__ __attribute__((aligned(16))) bf16_t VsAll[4][112][16];
  // q and the normalised query rows (the residual) wait in wave-private LDS for the rolled query-tile loop: holding them
  // in registers across an unrolled loop costs 24 B/lane of scratch at four workgroups per CU (4096 windows = ONE round)
  __shared__ __attribute__((aligned(16))) bf16_t QsAll[4][64][16];
  __shared__ __attribute__((aligned(16))) bf16_t XrAll[4][64][16];
  // the rel-pos table (rel_h [10][8] | rel_w [10][8] fp32) per wave: read tile by tile instead of 28 registers held from the start
  __shared__ __attribute__((aligned(16))) float RelAll[4][160];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wi = xcd_block_index() * 4 + wv;
  if (wi >= nwin) return;
  bf16_t(*Vs)[16] = VsAll[wv];
  bf16_t(*Qs)[16] = QsAll[wv];
  bf16_t(*Xr)[16] = XrAll[wv];
  float* RelS = RelAll[wv];
  const int lr = lane & 15, g = lane >> 4;
  const int nw = w / 8, nh = h / 8;
  const int wx = wi % nw, wy = (wi / nw) % nh, b = wi / (nw * nh);
  const long long img = (long long)b * h * w;
  // ---- every global load of the window up front, branch-free (clamped address + select) ----
  bf16x4 xq[4], xk[WA_KT];
  bool inb[WA_KT];
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int q = 16 * qt + lr;
    xq[qt] = ld4(x + (img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7)) * C16 + 4 * g);
  }
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const int key = min(16 * t + lr, WA_NK - 1);
    const int kr = key / 10, kc = key - kr * 10;
    const int y = 8 * wy + kr - 1, xx = 8 * wx + kc - 1;
    inb[t] = (16 * t + lr < WA_NK) && y >= 0 && y < h && xx >= 0 && xx < w;
    const int yc = min(max(y, 0), h - 1), xc = min(max(xx, 0), w - 1);
    xk[t] = ld4(x + (img + (long long)yc * w + xc) * C16 + 4 * g);
  }
  {
    const int l40 = min(lane, 39);                 // lanes 0..19: rel_h, 20..39: rel_w, one float4 each (clamped, store masked)
    const f32x4 rv4 = *reinterpret_cast<const f32x4*>((l40 < 20 ? rel_h + 4 * l40 : rel_w + 4 * (l40 - 20)));
    if (lane < 40) *reinterpret_cast<f32x4*>(&RelS[4 * lane]) = rv4;
  }
  wave_sync();
  bf16x4 wA[3];                                  // rows 16 n + lr of [q | k | v], input channels 4g ..
#pragma unroll
  for (int n = 0; n < 3; ++n) wA[n] = ld4(wqkv + (16 * n + lr) * C16 + 4 * g);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + b * 64 + 4 * g);
  const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + b * 64 + 4 * g);
  // ---- InstanceNorm apply (the arithmetic of branch_prep_kernel<T, 0>): x^ = bf16((x - mean) * rstd) ----
  auto normalise = [&](bf16x4 v) {
    return pack4(((float)v[0] - mu[0]) * rs[0], ((float)v[1] - mu[1]) * rs[1], ((float)v[2] - mu[2]) * rs[2], ((float)v[3] - mu[3]) * rs[3]);
  };
#pragma unroll
  for (int qt = 0; qt < 4; ++qt) {
    const int q = 16 * qt + lr;
    const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
    const bf16x4 xn = normalise(xq[qt]);
    st4(d + qpix * C16 + 4 * g, xn);
    st4(&Xr[q][4 * g], xn);
    f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma4(a, wA[0], xn);
    const bf16x4 qr = pack4(a[0], a[1], a[2], a[3]);
    if (qkv) st4(qkv + qpix * (3 * C16) + 4 * g, qr);         // (qkv == nullptr: the backward kernel recomputes q | k | v from d)
    st4(&Qs[q][4 * g], qr);
  }
  bf16x4 kA[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) {
    const bf16x4 xn = inb[t] ? normalise(xk[t]) : zero4();
    f32x4 ak = (f32x4){0.f, 0.f, 0.f, 0.f}, av = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma4(ak, wA[1], xn);
    mma4(av, wA[2], xn);
    const bf16x4 kraw = pack4(ak[0], ak[1], ak[2], ak[3]), vraw = pack4(av[0], av[1], av[2], av[3]);
    const int key = 16 * t + lr;
    const int kr = key / 10, kc = key - kr * 10;
    if (qkv && key < WA_NK && kr >= 1 && kr <= 8 && kc >= 1 && kc <= 8) {   // the window's own pixels: saved for the backward pass
      bf16_t* pk = qkv + (img + (long long)(8 * wy + kr - 1) * w + 8 * wx + kc - 1) * (3 * C16) + 4 * g;
      st4(pk + C16, kraw);
      st4(pk + 2 * C16, vraw);
    }
    const int kk = min(key, WA_NK - 1), kr2 = kk / 10, kc2 = kk - kr2 * 10;
    const f32x4 rel = *reinterpret_cast<const f32x4*>(&RelS[(g < 2) ? kr2 * 8 + 4 * g : 80 + kc2 * 8 + 4 * g - 8]);
    kA[t] = (key < WA_NK) ? pack4((float)kraw[0] + rel[0], (float)kraw[1] + rel[1], (float)kraw[2] + rel[2], (float)kraw[3] + rel[3])
                          : zero4();
    st4(&Vs[16 * t + lr][4 * g], (key < WA_NK) ? vraw : zero4());
  }
  wave_sync();
  bf16x4 vT[WA_KT];                               // rows = channel lr, contraction = keys 16 t + 4g ..
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) vT[t] = tr4(&Vs[16 * t + 4 * g + (lr >> 2)][4 * (lr & 3)]);
  const f32x4 L2E = (f32x4){1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f, 1.4426950408889634f};
#pragma unroll 1
  for (int qt = 0; qt < 4; ++qt) {
    const int q = 16 * qt + lr;
    const long long qpix = img + (long long)(8 * wy + (q >> 3)) * w + 8 * wx + (q & 7);
    const bf16x4 qv = ld4(&Qs[q][4 * g]);
    const bf16x4 qB = pack4(0.25f * (float)qv[0], 0.25f * (float)qv[1], 0.25f * (float)qv[2], 0.25f * (float)qv[3]);   // C^-1/2, exact
    f32x4 s[WA_KT];
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      mma4(s[t], kA[t], qB);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (4 * g + r >= 4) s[6][r] = -3.0e38f;                  // keys 100..111
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(s[t][0], s[t][1])), fmaxf(s[t][2], s[t][3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float nml = -mx * 1.4426950408889634f;
    const f32x4 NM = (f32x4){nml, nml, nml, nml};
    f32x4 sum4 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 e = s[t] * L2E + NM;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[t][r] = __builtin_amdgcn_exp2f(e[r]);
      sum4 += s[t];
    }
    float sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    const f32x4 INV = (f32x4){inv, inv, inv, inv};
    f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < WA_KT; ++t) {
      const f32x4 pv = s[t] * INV;
      mma4(o, vT[t], pack4(pv[0], pv[1], pv[2], pv[3]));
    }
    const bf16x4 rv = ld4(&Xr[q][4 * g]);                      // residual = the normalised input itself (:139)
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] += (float)rv[r];
    st4(out + qpix * ldo + oc0 + 4 * g, pack4(o[0], o[1], o[2], o[3]));
  }
}


// ---------------------------------------------------------------------------------------
// C = 16 branch, what follows the attention backward on the main chain, in one pass per pixel row (bf16):
//   halo_gather     dK|dV of a window-border pixel += the ring rows of the (<= 3) neighbouring windows      (k_attn.hip)
//   gemm_nt         g_d = [dq | dK | dV] Wqkv      (M = pixels, N = 16, K = 48)                               (k_gemm.hip)
//   branch_prep_bwd g_n[chunk 0] = g_d + g_xc[chunk 0]                                                        (k_pointwise.hip)
// Three launches that move 25 + 25 + 17 MB per block at batch 16 become one that reads each gqkv row once.  A wave owns 16
// consecutive pixels of an image row: lane (pixel = l & 15, g = l >> 4) loads the 8 columns 8 g .. of the first 32 (dq | dK) and,
// for g < 2, of the last 16 (dV) of its pixel's row, adds the neighbours' ring rows in halo_gather's order (fp32, one rounding),
// writes the completed dK|dV back (the weight-gradient GEMM on the side stream reads them) and feeds the rounded values to the
// same two 32-deep MFMAs the GEMM issues (k-slot = column; the second one zero beyond column 47); the tile comes back as
// (pixel, 4 channels) per lane, is rounded like the GEMM's store, and leaves as g_n.  Bit-identical to the three kernels.
// ---------------------------------------------------------------------------------------
// NORM (round 5, option "fused_norm_red"): the first reduction stage of the InstanceNorm backward that follows this kernel on the main
// chain rides in the same launch.  Workgroups 0 .. nred - 1 run instnorm_bwd_red1_body on planes 1 .. 3 of g_n (written by the three
// earlier branches: complete) -- a bandwidth-bound role beside the issue-bound tiles, so the two overlap instead of queueing -- and
// every tile wave leaves the sums of ITS 16 pixels of plane 0 (the values it has just produced, rounded as stored) in npart0.
struct C16NormArgs {
  const bf16_t* x;          // block input X, P64 (plane 0 first)
  const float* mean;        // [B][64]
  const float* rstd;
  float* part;              // [B][M2T_NORM_SPLIT][64][2]   (channels 16 .. 63 written)
  float* part0;             // [tile][2][16]
  int nred;                 // leading workgroups in the reduction role (= B * M2T_NORM_SPLIT)
};
template <int N> __device__ __forceinline__ float row_ror_f(float v) {      // value of lane (i + N) mod 16 of the lane's row of 16
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
template <bool NORM>
__global__ void __launch_bounds__(256) c16_dgrad_prep_kernel(bf16_t* __restrict__ gqkv, const bf16_t* __restrict__ win,
                                                             const bf16_t* __restrict__ wT, const bf16_t* __restrict__ gxc,
                                                             bf16_t* __restrict__ gn, int B, int h, int w, C16NormArgs na) {
  using T = bf16_t;
  int blk = blockIdx.x, nblk = gridDim.x;
  if constexpr (NORM) {
    __shared__ float sh[256][8][2];
    if (blk < na.nred) {                                    // workgroup-uniform
      instnorm_bwd_red1_body<T, 2, 6>(gn, na.x, na.mean, na.rstd, na.part, h * w, M2T_NORM_SPLIT, blk / M2T_NORM_SPLIT, blk % M2T_NORM_SPLIT, B, sh);
      return;
    }
    blk -= na.nred;
    nblk -= na.nred;
  }
  const int lane = threadIdx.x & 63, lr = lane & 15, g = lane >> 4;
  const int tpr = w / 16;                             // tiles per image row
  const int ntile = B * h * tpr;
  const int nh = h / 8, nw = w / 8;
  // A operand: Wqkv^T rows n = lr ([16][48]): columns 8 g .. of the first 32, and 32 + 8 g .. (g < 2) of the rest
  const Frag8<T> wa0 = load8(wT + lr * 48 + 8 * g);
  const Frag8<T> wa1 = g < 2 ? load8(wT + lr * 48 + 32 + 8 * g) : frag_zero<T>();
  // this lane's two 8-column pieces of a gqkv row: piece 0 = columns 8 g (dq for g < 2, dK for g >= 2), piece 1 = 32 + 8 (g & 1)
  // (dV; lanes g >= 2 load it too -- no divergent loads -- and drop it)
  const int col0 = 8 * g, col1 = 32 + 8 * (g & 1);
  const int ring0 = g >= 2 ? 8 * (g - 2) : 0, ring1 = 16 + 8 * (g & 1);        // the pieces' columns inside a ring row [dK | dV]
  const int wave = (blk * blockDim.x + threadIdx.x) >> 6, nwave = (nblk * blockDim.x) >> 6;
  for (int t = wave; t < ntile; t += nwave) {
    const int tx = t % tpr, y = (t / tpr) % h, b = t / (tpr * h);
    const int x = 16 * tx + lr;
    const long long pix = ((long long)b * h + y) * w + x;
    T* row = gqkv + pix * 48;
    long long hoff[3];
    const int nsrc = halo_sources(b, y, x, nh, nw, 32, hoff);
    float a0[8], a1[8], r0[3][8], r1[3][8], px4[4];
    load8f(row + col0, a0);
    load8f(row + col1, a1);
    float xv[4], mu4[4], rs4[4];
    if constexpr (NORM) {                                   // (issued with the other loads of the tile)
      load4(na.x + pix * 16 + 4 * g, xv);
      *reinterpret_cast<f32x4*>(mu4) = *reinterpret_cast<const f32x4*>(na.mean + b * 64 + 4 * g);
      *reinterpret_cast<f32x4*>(rs4) = *reinterpret_cast<const f32x4*>(na.rstd + b * 64 + 4 * g);
    }
    // unconditional loads from a clamped source, selected below.  Only a corner pixel has a second and third source, and corners lie
    // on the first / last row of a window: the wave's row decides (uniformly) whether those four loads are issued at all
    const int wy = y >> 3, py = y & 7;
    const bool border_row = (py == 0 && wy > 0) || (py == 7 && wy < nh - 1);
    {
      const long long o = 0 < nsrc ? hoff[0] : 0;
      load8f(win + o + ring0, r0[0]);
      load8f(win + o + ring1, r1[0]);
    }
    load4(gxc + pix * 16 + 4 * g, px4);
    if (border_row) {
#pragma unroll
      for (int a = 1; a < 3; ++a) {
        const long long o = a < nsrc ? hoff[a] : 0;
        load8f(win + o + ring0, r0[a]);
        load8f(win + o + ring1, r1[a]);
      }
    } else {
#pragma unroll
      for (int a = 1; a < 3; ++a)
#pragma unroll
        for (int e = 0; e < 8; ++e) { r0[a][e] = 0.f; r1[a][e] = 0.f; }
    }
    const bool gather0 = g >= 2 && nsrc > 0, gather1 = g < 2 && nsrc > 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (a < nsrc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0[e] = g >= 2 ? a0[e] + r0[a][e] : a0[e]; a1[e] += r1[a][e]; }
      }
    }
    Frag8<T> f0, f1;
#pragma unroll
    for (int e = 0; e < 8; ++e) { f0.set(e, a0[e]); f1.set(e, g < 2 ? a1[e] : 0.f); }
    if (gather0) store8(row + col0, f0);
    if (gather1) store8(row + col1, f1);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    mma16(acc, wa0, f0);                              // D[n = 4 g + r][pixel lr]
    mma16(acc, wa1, f1);
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = to_f(from_f<T>(acc[r])) + px4[r];     // g_d is a stored bf16 tensor in the unfused chain
    store4(gn + pix * 16 + 4 * g, o4);
    if constexpr (NORM) {
      // s1 / s2 of the tile's 16 pixels for the lane's four channels: the stored (rounded) g_n, the operations of instnorm_bwd_red1_body,
      // then a fixed rotation tree over the 16 pixel lanes of the row (every lane ends with the total)
      float a1v[4], a2v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gq = to_f(from_f<T>(o4[r]));
        a1v[r] = gq;
        a2v[r] = gq * ((xv[r] - mu4[r]) * rs4[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a1v[r] += row_ror_f<8>(a1v[r]); a2v[r] += row_ror_f<8>(a2v[r]);
        a1v[r] += row_ror_f<4>(a1v[r]); a2v[r] += row_ror_f<4>(a2v[r]);
        a1v[r] += row_ror_f<2>(a1v[r]); a2v[r] += row_ror_f<2>(a2v[r]);
        a1v[r] += row_ror_f<1>(a1v[r]); a2v[r] += row_ror_f<1>(a2v[r]);
      }
      if (lr == 0) {
        *reinterpret_cast<f32x4*>(na.part0 + (long long)t * 32 + 4 * g) = (f32x4){a1v[0], a1v[1], a1v[2], a1v[3]};
        *reinterpret_cast<f32x4*>(na.part0 + (long long)t * 32 + 16 + 4 * g) = (f32x4){a2v[0], a2v[1], a2v[2], a2v[3]};
      }
    }
  }
}

}  // namespace

int launch_window_attn_bwd_c16(const void* qkv, const float* rel_h, const float* rel_w, const void* gout, int ldg, int gc0,
                               void* gqkv, void* win, float* relw, int B, int h, int w, hipStream_t st, const void* d, const void* wqkv) {
  if (!qkv && !(d && wqkv)) return m2t_set_error(-2, "window_attn_bwd_c16: needs the saved qkv, or d and the packed weight to recompute it");
  const int nwin = B * (h / 8) * (w / 8);
  const size_t sh = 4 * sizeof(C16WaveLds);
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)window_attn_bwd_c16_kernel, (int)sh)) return rc__;
  M2T_LAUNCH_TIMED(window_attn_bwd_c16_kernel, dim3((nwin + 3) / 4), dim3(256), sh, st, (const bf16_t*)qkv, rel_h, rel_w,
                     (const bf16_t*)gout, ldg, gc0, (bf16_t*)gqkv, (bf16_t*)win, relw, h, w, nwin, (const bf16_t*)d, (const bf16_t*)wqkv);
  M2T_LAUNCH_CHECK();
  return 0;
}

// gqkv [B*h*w][48] (dK|dV of border pixels completed in place), win [window][36][32], wT = Wqkv^T [16][48] (M2T_PACK_TRANSPOSE),
// gxc / gn: chunk 0 planes [B*h*w][16] of the P64 gradients.  w % 16 == 0.
int launch_c16_dgrad_prep(void* gqkv, const void* win, const void* wT, const void* gxc, void* gn, int B, int h, int w, hipStream_t st,
                          const void* nx, const float* mean, const float* rstd, float* npart, float* npart0) {
  if (h % 8 || w % 16) return m2t_set_error(-2, "c16_dgrad_prep: h % 8, w % 16");
  const long long ntile = (long long)B * h * (w / 16);
  if (ntile * 16 * 48 >= (1LL << 31)) return m2t_set_error(-2, "c16_dgrad_prep: too many pixels for 32-bit tile indexing");
  const int grid = (int)std::min<long long>((ntile + 3) / 4, 4096);
  C16NormArgs na{};
  if (nx) {
    if (!mean || !rstd || !npart || !npart0) return m2t_set_error(-2, "c16_dgrad_prep: the norm reduction needs mean, rstd and both partial buffers");
    na.x = (const bf16_t*)nx; na.mean = mean; na.rstd = rstd; na.part = npart; na.part0 = npart0; na.nred = B * M2T_NORM_SPLIT;
    // the reduction workgroups come FIRST in the grid: they are the long bandwidth-bound ones, the tiles fill in around them
    M2T_LAUNCH_TIMED(c16_dgrad_prep_kernel<true>, dim3(grid + na.nred), dim3(256), 0, st, (bf16_t*)gqkv, (const bf16_t*)win, (const bf16_t*)wT,
                     (const bf16_t*)gxc, (bf16_t*)gn, B, h, w, na);
  } else {
    M2T_LAUNCH_TIMED(c16_dgrad_prep_kernel<false>, dim3(grid), dim3(256), 0, st, (bf16_t*)gqkv, (const bf16_t*)win, (const bf16_t*)wT,
                     (const bf16_t*)gxc, (bf16_t*)gn, B, h, w, na);
  }
  M2T_LAUNCH_CHECK();
  return 0;
}

int launch_window_attn_fwd_c16(const void* qkv, const float* rel_h, const float* rel_w, void* out, int ldo, int oc0, const void* res,
                               int ldr, int B, int h, int w, hipStream_t st) {
  const int nwin = B * (h / 8) * (w / 8);
  M2T_LAUNCH_TIMED(window_attn_fwd_c16_kernel, dim3((nwin + 3) / 4), dim3(256), 0, st, (const bf16_t*)qkv, rel_h, rel_w,
                     (bf16_t*)out, ldo, oc0, (const bf16_t*)res, ldr, h, w, nwin);
  M2T_LAUNCH_CHECK();
  return 0;
}

// x: the P64 plane of chunk 0 of the block input [B*h*w][16]; mean / rstd [B][64]; wqkv [48][16] bf16 (M2T_PACK_COPY);
// d [B*h*w][16] and qkv [B*h*w][48] are written for the backward pass; out: chunk plane of the concat buffer.
int launch_window_attn_fused_c16_fwd(const void* x, const float* mean, const float* rstd, const void* wqkv, const float* rel_h,
                                     const float* rel_w, void* d, void* qkv, void* out, int ldo, int oc0, int B, int h, int w,
                                     hipStream_t st) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn_fused_c16_fwd: h, w must be multiples of 8");
  const int nwin = B * (h / 8) * (w / 8);
  M2TProfScope ps(M2T_PROF_ATTN_FUSED_16, st);
  M2T_LAUNCH_TIMED(window_attn_fused_c16_fwd_kernel, dim3((nwin + 3) / 4), dim3(256), 0, st, (const bf16_t*)x, mean, rstd,
                   (const bf16_t*)wqkv, rel_h, rel_w, (bf16_t*)d, (bf16_t*)qkv, (bf16_t*)out, ldo, oc0, h, w, nwin);
  M2T_LAUNCH_CHECK();
  return 0;
}
