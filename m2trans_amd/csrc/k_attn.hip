// k_attn.hip -- 8x8-query / 10x10-key halo window attention on the matrix cores (gfx950).
//
// Restates TBlock.forward after the qkv projection (models/M2Trans_network.py:310-332) and
// its autograd:
//   * one workgroup = one window; 4 waves x 16 queries; the 100 keys are padded to 7 tiles
//     of 16 (the 12 pad keys are masked out of the softmax);
//   * keys outside the image are NOT masked: they are the zero-padding of F.unfold, so their
//     key vector is the relative-position embedding alone and their value is 0 (:313-325);
//   * S^T = K^ Q^T is computed "swapped" (keys on the MFMA rows, queries on the lanes) so a
//     query's softmax row sits in the 4 lanes {q, q+16, q+32, q+48} and the probabilities are
//     directly the B operand of O^T = V^T P^T -- no LDS round trip for P;
//   * V^T / K^T / Q^T / dO^T operands are staged transposed in LDS (key- or query-contiguous);
//   * large C is processed in 64-channel chunks so the same code serves C = 16, 64, 256 in
//     fp32 and bf16 inside 160 KB of LDS.
#include "m2t_kernels.h"
#include "m2t_haar.h"

#include "m2t_window.h"

// All stagers issue EVERY global load of the tile first (fully unrolled, registers) and only then
// touch LDS: one exposed memory latency per tile instead of one per 256-thread sweep.
//
// stage the K^ chunk row-major: 128 key rows x CW channels; real keys get k + rel-pos
// (zero-padded phantom keys = rel-pos alone), rows >= 100 and channels >= CC are zero
template <typename T, int C, int CC, int CW>
__device__ __forceinline__ void stage_khat(T (*Ks)[CW + 8], const T* __restrict__ qkv, const float* __restrict__ rel_h,
                                           const float* __restrict__ rel_w, const WinGeom& gm, int c0, int tid) {
  constexpr int VEC = CW / 8;
  constexpr int ITEMS = WA_KR * VEC / 256;
  Frag8<T> kf[ITEMS];
  f32x4 r0[ITEMS], r1[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx % VEC, key = idx / VEC;
    const int c = cv * 8;
    kf[it] = frag_zero<T>();
    r0[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    r1[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (key < WA_NK && c < CC) {
      long long pix;
      if (gm.key_pixel(key, pix)) kf[it] = load8(qkv + pix * (3 * C) + C + c0 + c);
      const int kr = key / 10, kc = key - kr * 10;
      const int cc = c0 + c;
      // parameter tensors start on 64-byte boundaries of the flat buffer and cc is a multiple of 8
      const float* rp = (cc < C / 2) ? (rel_h + kr * (C / 2) + cc) : (rel_w + kc * (C / 2) + (cc - C / 2));
      r0[it] = *reinterpret_cast<const f32x4*>(rp);
      r1[it] = *reinterpret_cast<const f32x4*>(rp + 4);
    }
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx % VEC, key = idx / VEC;
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = kf[it].get(e) + r0[it][e]; v[4 + e] = kf[it].get(4 + e) + r1[it][e]; }
    store8f(&Ks[key][cv * 8], v);
  }
}
// stage a row-major chunk of 128 key rows x CW channels from channel offset `coff` of qkv (V), zero outside
template <typename T, int C, int CC, int CW>
__device__ __forceinline__ void stage_keys_rows(T (*Vs)[CW + 8], const T* __restrict__ qkv, int coff, const WinGeom& gm,
                                                int c0, int tid) {
  constexpr int VEC = CW / 8;
  constexpr int ITEMS = WA_KR * VEC / 256;
  Frag8<T> f[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx % VEC, key = idx / VEC;
    const int c = cv * 8;
    f[it] = frag_zero<T>();
    long long pix;
    if (key < WA_NK && c < CC && gm.key_pixel(key, pix)) f[it] = load8(qkv + pix * (3 * C) + coff + c0 + c);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    store8(&Vs[idx / VEC][(idx % VEC) * 8], f[it]);
  }
}
// stage a row-major chunk of the 64 query rows x CW channels: dst[q][c] = src[query pixel][coff + c0 + c]
template <typename T, int CC, int CW>
__device__ __forceinline__ void stage_query_rows(T (*dst)[CW + 8], const T* __restrict__ src, int ld, int coff,
                                                 const WinGeom& gm, int c0, int tid) {
  constexpr int VEC = CW / 8;
  constexpr int ITEMS = (64 * VEC + 255) / 256;
  Frag8<T> f[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    const int cv = idx % VEC, q = idx / VEC;
    f[it] = frag_zero<T>();
    if (idx < 64 * VEC && cv * 8 < CC) f[it] = load8(src + gm.query_pixel(q) * ld + coff + c0 + cv * 8);
  }
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * 256;
    if (idx < 64 * VEC) store8(&dst[idx / VEC][(idx % VEC) * 8], f[it]);
  }
}

// =======================================================================================
// forward
// =======================================================================================
// L = 0: out[q pixel][oc0 + c] = O (+ res).  L = 1, 2 (C = 16 * 4^L): the branch epilogue is fused in --
// xc[full-res pixels][oc0 + 0..15] = IWT^L(O) + xin  (models/M2Trans_network.py:145,153,161): a lane holds, for its
// query, every band of 4 consecutive base channels, so the inverse Haar butterflies are register-local.
template <typename T, int C, int L>
__global__ void __launch_bounds__(256) window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ rel_h,
                                                              const float* __restrict__ rel_w, T* __restrict__ out, int ldo,
                                                              int oc0, const T* __restrict__ res, int ldr, int h, int w) {
  static_assert(L == 0 || C == (16 << (2 * L)), "fused IWT needs C = 16 * 4^L");
  constexpr int CC = (C < 64) ? C : 64;       // channels per chunk
  constexpr int CW = (CC < 32) ? 32 : CC;     // staged width (MFMA k-chunk is 32)
  constexpr int NCH = C / CC;
  constexpr int NT = CC / 16;
  constexpr int LD = CW + 8;
  __shared__ __attribute__((aligned(16))) T Ks[WA_KR][LD];   // K^ chunk, then the V chunk (row-major [key][c])
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const int q = 16 * wv + lr;
  const long long qpix = gm.query_pixel(q);

  // ---- S^T = K^ Q^T ----
  f32x4 s[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * CC;
    __syncthreads();
    stage_khat<T, C, CC, CW>(Ks, qkv, rel_h, rel_w, gm, c0, tid);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < CW / 32; ++kc) {
      Frag8<T> qf = frag_zero<T>();
      if (kc * 32 + 8 * g < CC) qf = load8(qkv + qpix * (3 * C) + c0 + kc * 32 + 8 * g);
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        const Frag8<T> kf = load8(&Ks[16 * t + lr][kc * 32 + 8 * g]);
        mma16(s[t], kf, qf);
      }
    }
  }
  // ---- softmax over the 100 real keys; lane (q, g) holds keys 16 t + 4 g + r ----
  const float scale = rsqrtf((float)C);   // head_ch ** -0.5 (:311); a power of two for C = 16, 64, 256
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
  // P as the B operand: k-chunk c4 covers key tiles 2c4, 2c4+1; slot (g, j) <-> key 16(2c4 + (j>>2)) + 4g + (j&3)
  Frag8<T> pf[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 2 * c4 + (j >> 2);
      pf[c4].set(j, (t < WA_KT) ? s[t < WA_KT ? t : 0][j & 3] * inv : 0.f);
    }

  // ---- O^T = V^T P^T, one channel chunk at a time; V^T fragments by transposing LDS reads ----
  auto pv_chunk = [&](int ch, f32x4 (&o)[NT]) {
    const int c0 = ch * CC;
    __syncthreads();
    stage_keys_rows<T, C, CC, CW>(Ks, qkv, 2 * C, gm, c0, tid);
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        const Frag8<T> vf = load8_tr(&Ks[32 * c4 + 4 * g][16 * mt], &Ks[32 * c4 + 16 + 4 * g][16 * mt], LD, lane);
        mma16(o[mt], vf, pf[c4]);
      }
  };
  if constexpr (L == 0) {
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x4 o[NT];
      pv_chunk(ch, o);
      // lane (q, g) holds channels c0 + 16 mt + 4 g + r
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        float v[4] = {o[mt][0], o[mt][1], o[mt][2], o[mt][3]};
        const int cc = ch * CC + 16 * mt + 4 * g;
        if (res) {
          float p[4];
          load4(res + qpix * ldr + cc, p);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += p[e];
        }
        store4(out + qpix * ldo + oc0 + cc, v);
      }
    }
  } else {
    constexpr int S = Haar<L>::S, NB = Haar<L>::N;
    f32x4 oall[NCH][NT];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) pv_chunk(ch, oall[ch]);
    // channel = (ch * 4 + mt) * 16 + 4 g + r  =  band * 16 + base channel (band-major nesting of repeated DWTs)
    float vv[4][S][S];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bands[NB];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) bands[ch * NT + mt] = oall[ch][mt][r];
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

int launch_window_attn_fwd(int dt, const void* qkv, const float* rel_h, const float* rel_w, void* out, int ldo, int oc0,
                           const void* res, int ldr, int B, int h, int w, int C, hipStream_t st, int post_levels) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn: h,w must be multiples of 8");
  if (post_levels != 0 && !(res && ((post_levels == 1 && C == 64) || (post_levels == 2 && C == 256))))
    return m2t_set_error(-2, "window_attn: fused IWT epilogue needs (levels, C) = (1, 64) or (2, 256) and the residual");
  const int nwin = B * (h / 8) * (w / 8);
  M2TProfScope ps(C == 16 ? M2T_PROF_ATTN_FWD_16 : (C == 64 ? M2T_PROF_ATTN_FWD_64 : M2T_PROF_ATTN_FWD_256), st);
  if (dt != M2T_F32 && C == 16 && post_levels == 0)      // bf16 full-resolution branch: one wave per window
    return launch_window_attn_fwd_c16(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, B, h, w, st);
  if (dt != M2T_F32) {                                   // bf16 C = 64 / 256: whole window resident in LDS (20 -> 15 us at C = 256)
    const int rc = launch_window_attn_fwd_resident(qkv, rel_h, rel_w, out, ldo, oc0, res, ldr, B, h, w, C, post_levels, st);
    if (rc != M2T_UNSUPPORTED) return rc;
  }
#define GO(T_, C_, L_) M2T_LAUNCH_TIMED((window_attn_fwd_kernel<T_, C_, L_>), dim3(nwin), dim3(256), 0, st, (const T_*)qkv, rel_h, rel_w, (T_*)out, ldo, oc0, (const T_*)res, ldr, h, w)
#define GOT(T_)                                                                                   \
  if (post_levels == 1) GO(T_, 64, 1); else if (post_levels == 2) GO(T_, 256, 2);                   \
  else if (C == 16) GO(T_, 16, 0); else if (C == 64) GO(T_, 64, 0); else if (C == 256) GO(T_, 256, 0); \
  else return m2t_set_error(-2, "window_attn: C must be 16, 64 or 256");
  if (dt == M2T_F32) { GOT(float) } else { GOT(bf16_t) }
#undef GOT
#undef GO
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// backward: per window, recompute S and P, then
//   dP = dO V^T ; delta = rowsum(P * dP) ; dS = P * (dP - delta) * scale
//   dq = dS K^ ; dK^ = dS^T q ; dV = P^T dO
// dq goes straight to gqkv[..., 0:C]; dK^ / dV go to the window-major scratch `win`
// [B*L][100][2C] (dK^ | dV) that halo_gather sums over the <= 4 windows covering a pixel.
// The relative-position gradient of the window (dK^ summed over key columns / rows, phantom
// keys included) is reduced from the fp32 accumulators and written to relw [B*L][10][C].
// =======================================================================================
// one band of a 2x2 Haar butterfly (same association order as haar2_fwd)
__device__ __forceinline__ float haar2_fwd_band(float a, float b, float c, float d, int band) {
  switch (band) {
    case 0: return 0.5f * (((a + b) + c) + d);
    case 1: return 0.5f * (((-a - b) + c) + d);
    case 2: return 0.5f * (((-a + b) - c) + d);
    default: return 0.5f * (((a - b) - c) + d);
  }
}
// stage the 64 query rows x CW channels of the OUTPUT gradient chunk `ch` (row-major [q][c]).
// L = 0: plain rows of go.  L = 1, 2: go is the full-res g_xc tensor (ld, channel offset coff); the
// branch's gradient is DWT^L of its 16-channel slice (IWT^T = DWT), computed on load: thread (q, 4-channel
// group) reads its (2^L)^2 pixel block and writes the bands that fall into this 64-channel chunk.
template <typename T, int L, int CC, int CW>
__device__ __forceinline__ void stage_go_rows(T (*dst)[CW + 8], const T* __restrict__ go, int ld, int coff, const WinGeom& gm,
                                              int ch, int tid) {
  if constexpr (L == 0) {
    stage_query_rows<T, CC, CW>(dst, go, ld, coff, gm, ch * CC, tid);
  } else {
    constexpr int S = Haar<L>::S;
    const int q = tid >> 2, cg = tid & 3;
    const int H = gm.h * S, W = gm.w * S;
    const int by = 8 * gm.wy + (q >> 3), bx = 8 * gm.wx + (q & 7);
    float v[4][S][S];
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        float t4[4];
        load4(go + (((long long)gm.b * H + S * by + y) * W + S * bx + x) * ld + coff + 4 * cg, t4);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i][y][x] = t4[i];
      }
    if constexpr (L == 1) {
      float o[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) Haar<1>::fwd(v[i], o[i]);
#pragma unroll
      for (int b1 = 0; b1 < 4; ++b1) {
        float t4[4] = {o[0][b1], o[1][b1], o[2][b1], o[3][b1]};
        store4(&dst[q][b1 * 16 + 4 * cg], t4);
      }
    } else {
      // chunk ch = second-level band; first level on the four 2x2 quadrants, then ONE band of the second level
#pragma unroll
      for (int b1 = 0; b1 < 4; ++b1) {
        float t4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float tq[2][2];
#pragma unroll
          for (int I = 0; I < 2; ++I)
#pragma unroll
            for (int J = 0; J < 2; ++J)
              tq[I][J] = haar2_fwd_band(v[i][2 * I][2 * J], v[i][2 * I + 1][2 * J], v[i][2 * I][2 * J + 1], v[i][2 * I + 1][2 * J + 1], b1);
          t4[i] = haar2_fwd_band(tq[0][0], tq[1][0], tq[0][1], tq[1][1], ch);
        }
        store4(&dst[q][b1 * 16 + 4 * cg], t4);
      }
    }
  }
}

template <typename T, int C, int L>
__global__ void __launch_bounds__(256) window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ rel_h,
                                                              const float* __restrict__ rel_w, const T* __restrict__ go,
                                                              int ldg, int gc0, T* __restrict__ gqkv, T* __restrict__ win,
                                                              float* __restrict__ relw, int h, int w) {
  constexpr int CC = (C < 64) ? C : 64;
  constexpr int CW = (CC < 32) ? 32 : CC;
  constexpr int NCH = C / CC;
  constexpr int NT = CC / 16;
  constexpr int LD = CW + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // region A: Ks [128][LD] | Vs [128][LD] (phase 1)  /  Ks | DOs [64][LD] | Qs [64][LD] (phase 2)
  //           / KA fp32 [112][CC+1] (rel-pos reduction, aliases Ks after the chunk's products)
  // region B: PT [112][72], DST [112][72]   ([key][query])
  static_assert(L == 0 || C == (16 << (2 * L)), "fused DWT needs C = 16 * 4^L");
  constexpr size_t szK = sizeof(T) * WA_KR * LD;
  constexpr size_t szA = 2 * szK + sizeof(T) * 64 * LD;
  T(*Ks)[LD] = reinterpret_cast<T(*)[LD]>(smem);
  T(*Vs)[LD] = reinterpret_cast<T(*)[LD]>(smem + szK);                       // phase 1
  T(*Qs)[LD] = reinterpret_cast<T(*)[LD]>(smem + szK);                       // phase 2 (aliases Vs)
  T(*DOs)[LD] = reinterpret_cast<T(*)[LD]>(smem + 2 * szK);                  // output-gradient rows, both phases
  float(*KA)[CC + 1] = reinterpret_cast<float(*)[CC + 1]>(smem);
  T(*PT)[WA_QP] = reinterpret_cast<T(*)[WA_QP]>(smem + szA);
  T(*DST)[WA_QP] = reinterpret_cast<T(*)[WA_QP]>(smem + szA + sizeof(T) * (WA_KT * 16) * WA_QP);

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const int q = 16 * wv + lr;
  const long long qpix = gm.query_pixel(q);

  // ---- phase 1: S^T = K^ Q^T and dP^T = V dO^T, accumulated over channel chunks ----
  f32x4 s[WA_KT], dp[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) { s[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * CC;
    __syncthreads();
    stage_khat<T, C, CC, CW>(Ks, qkv, rel_h, rel_w, gm, c0, tid);
    stage_keys_rows<T, C, CC, CW>(Vs, qkv, 2 * C, gm, c0, tid);
    stage_go_rows<T, L, CC, CW>(DOs, go, ldg, gc0, gm, ch, tid);
    __syncthreads();
#pragma unroll
    for (int kc = 0; kc < CW / 32; ++kc) {
      Frag8<T> qf = frag_zero<T>();
      if (kc * 32 + 8 * g < CC) qf = load8(qkv + qpix * (3 * C) + c0 + kc * 32 + 8 * g);
      const Frag8<T> gf = load8(&DOs[q][kc * 32 + 8 * g]);
#pragma unroll
      for (int t = 0; t < WA_KT; ++t) {
        const Frag8<T> kf = load8(&Ks[16 * t + lr][kc * 32 + 8 * g]);
        mma16(s[t], kf, qf);
        const Frag8<T> vf = load8(&Vs[16 * t + lr][kc * 32 + 8 * g]);
        mma16(dp[t], vf, gf);
      }
    }
  }
  // ---- softmax and dS ----
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
  float delta = 0.f;
#pragma unroll
  for (int t = 0; t < WA_KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[t][r] *= inv;                       // P
      delta += s[t][r] * dp[t][r];
    }
  delta += __shfl_xor(delta, 16);
  delta += __shfl_xor(delta, 32);
#pragma unroll
  for (int t = 0; t < WA_KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) dp[t][r] = s[t][r] * (dp[t][r] - delta) * scale;   // dS (scale folded)
  // P^T and dS^T to LDS ([key][q]); dS also stays in registers as the B operand of dq
#pragma unroll
  for (int t = 0; t < WA_KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      PT[16 * t + 4 * g + r][q] = from_f<T>(s[t][r]);
      DST[16 * t + 4 * g + r][q] = from_f<T>(dp[t][r]);
    }
  Frag8<T> dsf[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 2 * c4 + (j >> 2);
      dsf[c4].set(j, (t < WA_KT) ? dp[t < WA_KT ? t : 0][j & 3] : 0.f);
    }

  // ---- phase 2: per channel chunk: dq^T = K^^T dS^T ; dV^T = dO^T P ; dK^^T = q^T dS ----
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * CC;
    __syncthreads();   // phase-1 / previous chunk's LDS reads (incl. KA) are done; PT/DST are written
    stage_khat<T, C, CC, CW>(Ks, qkv, rel_h, rel_w, gm, c0, tid);
    stage_go_rows<T, L, CC, CW>(DOs, go, ldg, gc0, gm, ch, tid);
    stage_query_rows<T, CC, CW>(Qs, qkv, 3 * C, 0, gm, c0, tid);
    __syncthreads();
    // dq for this wave's 16 queries: A = K^ read transposed (rows = channels, contraction = keys)
    {
      f32x4 o[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) o[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
          const Frag8<T> kf = load8_tr(&Ks[32 * c4 + 4 * g][16 * mt], &Ks[32 * c4 + 16 + 4 * g][16 * mt], LD, lane);
          mma16(o[mt], kf, dsf[c4]);
        }
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        float v[4] = {o[mt][0], o[mt][1], o[mt][2], o[mt][3]};
        store4(gqkv + qpix * (3 * C) + c0 + 16 * mt + 4 * g, v);
      }
    }
    // dV^T and dK^^T: wave wv owns key tiles wv and wv + 4; A = dO / q read transposed
    // (rows = channels, contraction = queries), B = P^T / dS^T rows
    f32x4 akk[2][NT];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = wv + 4 * tt;
      f32x4 av[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) { av[mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; akk[tt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      if (t < WA_KT) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
          const Frag8<T> pfr = load8(&PT[16 * t + lr][32 * kc + 8 * g]);     // B: cols = keys, k = queries
          const Frag8<T> dfr = load8(&DST[16 * t + lr][32 * kc + 8 * g]);
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) {
            const Frag8<T> gof = load8_tr(&DOs[32 * kc + 8 * g][16 * mt], &DOs[32 * kc + 8 * g + 4][16 * mt], LD, lane);
            mma16(av[mt], gof, pfr);
            const Frag8<T> qf = load8_tr(&Qs[32 * kc + 8 * g][16 * mt], &Qs[32 * kc + 8 * g + 4][16 * mt], LD, lane);
            mma16(akk[tt][mt], qf, dfr);
          }
        }
        const int key = 16 * t + lr;
        if (key < WA_NK) {
          T* wp = dkv_row(gqkv, win, (long long)gm.wi, gm.b, gm.wy, gm.wx, h, w, C, key);
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) {
            float v[4] = {akk[tt][mt][0], akk[tt][mt][1], akk[tt][mt][2], akk[tt][mt][3]};
            float u[4] = {av[mt][0], av[mt][1], av[mt][2], av[mt][3]};
            store4(wp + c0 + 16 * mt + 4 * g, v);
            store4(wp + C + c0 + 16 * mt + 4 * g, u);
          }
        }
      }
    }
    // rel-pos gradient: dK^ tile to LDS in fp32, then ordered sums over key columns / rows
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = wv + 4 * tt;
      if (t < WA_KT) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) KA[16 * t + lr][16 * mt + 4 * g + r] = akk[tt][mt][r];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 10 * CC; idx += 256) {
      const int i = idx / CC, c = idx - i * CC;
      const int cc = c0 + c;
      float a = 0.f;
      if (cc < C / 2) {
#pragma unroll
        for (int j = 0; j < 10; ++j) a += KA[i * 10 + j][c];      // row embedding: sum over columns
      } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) a += KA[j * 10 + i][c];      // column embedding: sum over rows
      }
      relw[((long long)gm.wi * 10 + i) * C + cc] = a;
    }
  }
}

// add to every window-border pixel's dK|dV (already holding its own window's contribution) the ring rows of the
// (<= 3) neighbouring windows whose 10x10 neighbourhood covers it; interior pixels are final as written.
template <typename T>
__global__ void __launch_bounds__(256) halo_gather_kernel(const T* __restrict__ win, T* __restrict__ gqkv, int B, int h,
                                                          int w, int rw, int ld, int coff) {
  // rw = elements per ring row (2C for dK|dV, C for the fused data gradient); destination rows [pixel][ld], columns coff..
  const int nv = rw / 8;
  const int nh = h / 8, nw = w / 8;
  const int total = B * nh * nw * 28 * nv;       // 28 border pixels per window; < 2^31 (checked by the launcher):
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {   // 32-bit index math --
    const int cv = t % nv;                        // five 64-bit divisions per thread cost more than its three 16-byte accesses
    int r = t / nv;
    const int bp = r % 28; r /= 28;
    const int wx0 = r % nw; r /= nw;
    const int wy0 = r % nh;
    const int b = r / nh;
    // border pixel bp of the 8x8 window: rows 0 and 7 (8 each), then columns 0 and 7 of rows 1..6 (6 each)
    int py, px;
    if (bp < 8) { py = 0; px = bp; }
    else if (bp < 16) { py = 7; px = bp - 8; }
    else if (bp < 22) { py = bp - 15; px = 0; }
    else { py = bp - 21; px = 7; }
    const int y = 8 * wy0 + py, x = 8 * wx0 + px;
    int wys[2], krs[2], ny = 1, wxs[2], kcs[2], nx = 1;
    wys[0] = wy0; krs[0] = py + 1;
    if (py == 0 && wy0 > 0) { wys[1] = wy0 - 1; krs[1] = 9; ny = 2; }
    else if (py == 7 && wy0 < nh - 1) { wys[1] = wy0 + 1; krs[1] = 0; ny = 2; }
    wxs[0] = wx0; kcs[0] = px + 1;
    if (px == 0 && wx0 > 0) { wxs[1] = wx0 - 1; kcs[1] = 9; nx = 2; }
    else if (px == 7 && wx0 < nw - 1) { wxs[1] = wx0 + 1; kcs[1] = 0; nx = 2; }
    if (ny * nx == 1) continue;                                   // image corner / edge with no neighbour
    T* dst = gqkv + (((long long)b * h + y) * w + x) * ld + coff + cv * 8;
    float acc[8];
    load8f(dst, acc);
    for (int a = 0; a < ny; ++a)
      for (int c = 0; c < nx; ++c) {
        if (a == 0 && c == 0) continue;                           // the own window wrote straight to gqkv
        const long long wi = ((long long)b * nh + wys[a]) * nw + wxs[c];
        float v[8];
        load8f(win + (wi * WA_RING + ring_index(krs[a], kcs[c])) * rw + cv * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e];
      }
    store8f(dst, acc);
  }
}

// relative-position gradients: sum the per-window partials relw [nwin][10*C] over the windows
// (two deterministic stages), then scatter to the torch layouts rel_h [1][10][1][C/2], rel_w [1][1][10][C/2]
// stage 1: workgroup = 32 columns x 8 window lanes of one split; every thread keeps 4 independent partial sums, so a
// split of 128 windows (C = 16: 4096 windows) costs 4 rounds of loads instead of 32 (30 us -> 5 us per launch)
__global__ void __launch_bounds__(256) rel_reduce1_kernel(const float* __restrict__ relw, float* __restrict__ part, int nwin,
                                                          int ncol, int win_per_split) {
  __shared__ float red[8][33];
  const int c = threadIdx.x & 31, l = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + c;
  const int w0 = blockIdx.y * win_per_split, w1 = min(nwin, w0 + win_per_split);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (col < ncol) {
    int wi = w0 + l;
    for (; wi + 24 < w1; wi += 32) {
      a0 += relw[(long long)wi * ncol + col];
      a1 += relw[(long long)(wi + 8) * ncol + col];
      a2 += relw[(long long)(wi + 16) * ncol + col];
      a3 += relw[(long long)(wi + 24) * ncol + col];
    }
    for (; wi < w1; wi += 8) a0 += relw[(long long)wi * ncol + col];
  }
  red[l][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (l == 0 && col < ncol) {
    float t = red[0][c];
#pragma unroll
    for (int j = 1; j < 8; ++j) t += red[j][c];
    part[(long long)blockIdx.y * ncol + col] = t;
  }
}
__global__ void __launch_bounds__(256) rel_reduce2_kernel(const float* __restrict__ part, float* __restrict__ grel_h,
                                                          float* __restrict__ grel_w, int nsplit, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 10 * C) return;
  float a = 0.f;
  for (int b = 0; b < nsplit; ++b) a += part[(long long)b * 10 * C + idx];
  const int i = idx / C, c = idx % C;
  if (c < C / 2) grel_h[i * (C / 2) + c] = a;        // torch rel_h [1][10][1][C/2]
  else grel_w[i * (C / 2) + (c - C / 2)] = a;        // torch rel_w [1][1][10][C/2]
}

template <typename T, int C> static size_t attn_bwd_smem() {
  constexpr int CC = (C < 64) ? C : 64;
  constexpr int CW = (CC < 32) ? 32 : CC;
  return 2 * sizeof(T) * WA_KR * (CW + 8) + sizeof(T) * 64 * (CW + 8) + 2 * sizeof(T) * (WA_KT * 16) * WA_QP;
}

template <typename T>
static int launch_window_attn_bwd_t(const T* qkv, const float* rel_h, const float* rel_w, const T* gout, int ldg, int gc0,
                                    T* gqkv, T* win, float* relw, int B, int h, int w, int C, hipStream_t st, int dwt_levels, bool gather, bool resident) {
  const int nwin = B * (h / 8) * (w / 8);
  if (dwt_levels != 0 && !((dwt_levels == 1 && C == 64) || (dwt_levels == 2 && C == 256)))
    return m2t_set_error(-2, "window_attn_bwd: fused DWT needs (levels, C) = (1, 64) or (2, 256)");
#define GO(C_, L_)                                                                                                 \
  {                                                                                                                \
    const size_t sh = attn_bwd_smem<T, C_>();                                                                      \
    (void)hipFuncSetAttribute((const void*)window_attn_bwd_kernel<T, C_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    M2T_LAUNCH_TIMED((window_attn_bwd_kernel<T, C_, L_>), dim3(nwin), dim3(256), sh, st, qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, h, w); \
  }
  m2t_prof_begin(C == 16 ? M2T_PROF_ATTN_BWD_16 : (C == 64 ? M2T_PROF_ATTN_BWD_64 : M2T_PROF_ATTN_BWD_256), st);
  int res_rc = M2T_UNSUPPORTED;
  if (resident && sizeof(T) == 2 && C == 16 && dwt_levels == 0) {
    res_rc = launch_window_attn_bwd_c16(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, B, h, w, st);
    if (res_rc != 0) return res_rc;
  } else if (resident && sizeof(T) == 2) {
    res_rc = launch_window_attn_bwd_resident(qkv, rel_h, rel_w, gout, ldg, gc0, gqkv, win, relw, B, h, w, C, dwt_levels, st, nullptr, nullptr, nullptr);
    if (res_rc != 0 && res_rc != M2T_UNSUPPORTED) return res_rc;
  }
  if (res_rc == 0) {}
  else if (dwt_levels == 1) GO(64, 1) else if (dwt_levels == 2) GO(256, 2)
  else if (C == 16) GO(16, 0) else if (C == 64) GO(64, 0) else if (C == 256) GO(256, 0)
  else return m2t_set_error(-2, "window_attn_bwd: C must be 16, 64 or 256");
#undef GO
  m2t_prof_end(C == 16 ? M2T_PROF_ATTN_BWD_16 : (C == 64 ? M2T_PROF_ATTN_BWD_64 : M2T_PROF_ATTN_BWD_256), st);
  M2T_LAUNCH_CHECK();
  if (gather) {
    const long long total = (long long)B * (h / 8) * (w / 8) * 28 * (2 * C / 8);
    if (total >= (1LL << 31)) return m2t_set_error(-2, "halo_gather: too many border vectors for 32-bit indexing");
    const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
    hipLaunchKernelGGL(halo_gather_kernel<T>, dim3(g), dim3(256), 0, st, win, gqkv, B, h, w, 2 * C, 3 * C, C);
    M2T_LAUNCH_CHECK();
  }
  return 0;
}
// the overlap-add alone: ring rows `win` [window][36][rw] are added to columns coff .. coff + rw of the border pixels' rows
// of dst [pixel][ld]
int launch_halo_gather(int dt, const void* win, void* dst, int B, int h, int w, int rw, int ld, int coff, hipStream_t st) {
  const long long total = (long long)B * (h / 8) * (w / 8) * 28 * (rw / 8);
  if (total >= (1LL << 31)) return m2t_set_error(-2, "halo_gather: too many border vectors for 32-bit indexing");
  const int g = (int)std::min<long long>(ceil_divll(total, 256), 4096);
  if (dt == M2T_F32) hipLaunchKernelGGL(halo_gather_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)win, (float*)dst, B, h, w, rw, ld, coff);
  else hipLaunchKernelGGL(halo_gather_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)win, (bf16_t*)dst, B, h, w, rw, ld, coff);
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_rel_reduce1(const float* relw, float* rel_part, int nwin, int C, int* nsplit_out, hipStream_t st) {
  int nsplit = std::min(nwin, 32);
  const int wps = ceil_div(nwin, nsplit);
  nsplit = ceil_div(nwin, wps);
  hipLaunchKernelGGL(rel_reduce1_kernel, dim3(ceil_div(10 * C, 32), nsplit), dim3(256), 0, st, relw, rel_part, nwin, 10 * C, wps);
  M2T_LAUNCH_CHECK();
  *nsplit_out = nsplit;
  return 0;
}
int launch_rel_reduce(const float* relw, float* rel_part, float* grel_h, float* grel_w, int nwin, int C, hipStream_t st) {
  int nsplit = std::min(nwin, 32);
  const int wps = ceil_div(nwin, nsplit);
  nsplit = ceil_div(nwin, wps);
  hipLaunchKernelGGL(rel_reduce1_kernel, dim3(ceil_div(10 * C, 32), nsplit), dim3(256), 0, st, relw, rel_part, nwin, 10 * C, wps);
  M2T_LAUNCH_CHECK();
  hipLaunchKernelGGL(rel_reduce2_kernel, dim3(ceil_div(10 * C, 256)), dim3(256), 0, st, rel_part, grel_h, grel_w, nsplit, C);
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_window_attn_bwd(int dt, const void* qkv, const float* rel_h, const float* rel_w, const void* gout, int ldg,
                           int gc0, void* gqkv, void* win, float* relw, int B, int h, int w, int C, hipStream_t st,
                           int dwt_levels, bool gather, bool resident) {
  if (h % 8 || w % 8) return m2t_set_error(-2, "window_attn_bwd: h,w must be multiples of 8");
  if (dt == M2T_F32)
    return launch_window_attn_bwd_t<float>((const float*)qkv, rel_h, rel_w, (const float*)gout, ldg, gc0, (float*)gqkv, (float*)win, relw, B, h, w, C, st, dwt_levels, gather, resident);
  return launch_window_attn_bwd_t<bf16_t>((const bf16_t*)qkv, rel_h, rel_w, (const bf16_t*)gout, ldg, gc0, (bf16_t*)gqkv, (bf16_t*)win, relw, B, h, w, C, st, dwt_levels, gather, resident);
}
