// k_branch.hip -- one CFTM branch of the forward pass as ONE kernel per window (gfx950):
//
//     xin = norm(x)[chunk k]  (k = 0)   |   (norm(x)[chunk k] + xc[chunk k-1]) / 2  (k >= 1)     (:135-155)
//     d   = DWT^L(xin)                                                                            (:143,149-150)
//     q|k|v = d Wqkv^T   (1x1 conv, projection of the window's 100 halo keys done in place)      (:307-308)
//     o   = softmax(q k^^T / sqrt(C)) v   (8x8 queries, 10x10 keys, rel-pos on every key)        (:310-331)
//     xc[chunk k] = IWT^L(o) + xin                                                                (:139,145,153,161)
//
// replacing four launches (branch_prep, gemm_nt, window_attn_fwd, branch_post) and the HBM round trips of
// d / a.  The projection makes the kernel a dense contraction (41 MFLOP per window at C = 256 against
// ~0.3 MB of traffic), so it is bounded by the matrix cores / LDS rather than by HBM.
// Zero-padding semantics: a key outside the image has d = 0, hence k = v = 0 and k^ = rel-pos.
//
// Work split (4 waves): the transformed tile D [128 keys][C] sits in LDS; per 64-channel output chunk, wave w
// projects output-channel tile w of K^ / Q / V for every key tile (weight fragments stay in registers, D
// fragments come from LDS); scores, softmax and PV follow window_attn_fwd_kernel.  Side outputs for the
// backward pass: d (wgrad operand), q|k|v (attention backward), xin (residual).
#include "m2t_kernels.h"
#include "m2t_haar.h"
#include "m2t_window.h"

#define BR_THREADS 512
template <typename T, int C, int L>
__global__ void __launch_bounds__(BR_THREADS)
branch_fwd_kernel(const T* __restrict__ X, const float* __restrict__ mean, const float* __restrict__ rstd,
                  T* __restrict__ xc, int k, const T* __restrict__ Wqkv, const float* __restrict__ rel_h,
                  const float* __restrict__ rel_w, T* __restrict__ xin, T* __restrict__ dout, T* __restrict__ qkv, int h, int w) {
  static_assert(C == (16 << (2 * L)), "C = 16 * 4^L");
  constexpr int S = Haar<L>::S, NB = Haar<L>::N;
  constexpr int CC = (C < 64) ? C : 64;        // output channels per chunk
  constexpr int NCH = C / CC;
  constexpr int NTC = CC / 16;                 // output-channel tiles per chunk
  constexpr int KD = (C < 32) ? 32 : C;        // contraction width of D (zero padded for C = 16)
  constexpr int KCD = KD / 32;
  constexpr int LDD = KD + 8;
  constexpr int CW = (CC < 32) ? 32 : CC;
  constexpr int LDK = CW + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*D)[LDD] = reinterpret_cast<T(*)[LDD]>(smem);                                        // [128][KD+8]
  T(*Ks)[LDK] = reinterpret_cast<T(*)[LDK]>(smem + sizeof(T) * WA_KR * LDD);             // [128][CW+8]  K^ chunk, then V chunk
  T(*Qs)[LDK] = reinterpret_cast<T(*)[LDK]>(smem + sizeof(T) * WA_KR * (LDD + LDK));     // [64][CW+8]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const WinGeom gm = make_geom(h, w);
  const int H = h * S, Wf = w * S;

  // ---- 0. zero the padding of D / Ks / Qs (rows >= 100, channels >= C resp. CC) ----
  for (int idx = tid; idx < WA_KR * (LDD / 8); idx += BR_THREADS) {
    const int row = idx / (LDD / 8), cv = idx % (LDD / 8);
    if (row >= WA_NK || cv * 8 >= C) store8(&D[row][cv * 8], frag_zero<T>());
  }
  for (int idx = tid; idx < WA_KR * (LDK / 8); idx += BR_THREADS) store8(&Ks[idx / (LDK / 8)][(idx % (LDK / 8)) * 8], frag_zero<T>());
  for (int idx = tid; idx < 64 * (LDK / 8); idx += BR_THREADS) store8(&Qs[idx / (LDK / 8)][(idx % (LDK / 8)) * 8], frag_zero<T>());
  __syncthreads();

  // ---- 1. D tile: normalise + mix + DWT^L for the 100 keys (branch pixels incl. the halo ring) ----
  for (int item = tid; item < WA_NK * 4; item += BR_THREADS) {
    const int key = item >> 2, cg = item & 3;
    const int kr = key / 10, kc = key - kr * 10;
    const int by = 8 * gm.wy + kr - 1, bx = 8 * gm.wx + kc - 1;
    const bool inside = (by >= 0 && by < h && bx >= 0 && bx < w);
    const bool interior = (kr >= 1 && kr <= 8 && kc >= 1 && kc <= 8);
    float o[4][NB];
    if (inside) {
      float mu[4], rs[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) { mu[c] = mean[gm.b * 64 + k * 16 + cg * 4 + c]; rs[c] = rstd[gm.b * 64 + k * 16 + cg * 4 + c]; }
      float v[4][S][S];
#pragma unroll
      for (int y = 0; y < S; ++y)
#pragma unroll
        for (int x = 0; x < S; ++x) {
          const long long pix = ((long long)gm.b * H + S * by + y) * Wf + S * bx + x;
          float q4[4];
          load4(X + pix * 64 + k * 16 + cg * 4, q4);
#pragma unroll
          for (int c = 0; c < 4; ++c) q4[c] = (q4[c] - mu[c]) * rs[c];
          if (k > 0) {
            float p4[4];
            load4(xc + pix * 64 + (k - 1) * 16 + cg * 4, p4);
#pragma unroll
            for (int c = 0; c < 4; ++c) q4[c] = (q4[c] + p4[c]) * 0.5f;
          }
          if (L > 0) {
            if (interior) store4(xin + pix * 16 + cg * 4, q4);
            if (sizeof(T) == 2) {
#pragma unroll
              for (int c = 0; c < 4; ++c) q4[c] = to_f(from_f<T>(q4[c]));      // transform what was stored
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c][y][x] = q4[c];
        }
#pragma unroll
      for (int c = 0; c < 4; ++c) Haar<L>::fwd(v[c], o[c]);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int n = 0; n < NB; ++n) o[c][n] = 0.f;
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      float q4[4] = {o[0][n], o[1][n], o[2][n], o[3][n]};
      store4(&D[key][n * 16 + cg * 4], q4);
      if (inside && interior) {
        const long long bp = ((long long)gm.b * h + by) * w + bx;
        store4(dout + bp * C + n * 16 + cg * 4, q4);
      }
    }
  }
  __syncthreads();

  // Projection of one 16-channel output tile for NM row tiles at once:
  //   out[row][n] = sum_c D[src row][c] * W[wrow + n][c]     (A = weight rows, B = D rows)
  // NM independent accumulator chains keep the matrix pipe busy while the D fragments arrive from LDS;
  // lane (row = lr, g) receives channels 4g..4g+3 of the tile.
  auto load_w = [&](Frag8<T> (&wf)[KCD], int wrow) {     // weight rows wrow + lr, all contraction chunks
#pragma unroll
    for (int kc = 0; kc < KCD; ++kc) {
      wf[kc] = frag_zero<T>();
      if (32 * kc + 8 * g < C) wf[kc] = load8(Wqkv + (long long)(wrow + lr) * C + 32 * kc + 8 * g);
    }
  };
  // tile ownership over 8 waves:
  //   NTC == 4: wave w owns channel tile (w & 3); key-row tiles {0..3} (w < 4) or {4..6} (w >= 4); query tiles {0,1} / {2,3}
  //   NTC == 1: the single channel tile; key-row tile w (w < 7); query tile w (w < 4)
  constexpr int KMT = (NTC == 4) ? 4 : 1;          // key-row tiles per wave
  constexpr int QMT = (NTC == 4) ? 2 : 1;          // query-row tiles per wave
  const int my_nt = (NTC == 4) ? (wv & 3) : 0;
  auto key_tile_of = [&](int i) -> int { return (NTC == 4) ? (4 * (wv >> 2) + i) : wv; };       // >= 7: skipped
  auto query_tile_of = [&](int i) -> int { return (NTC == 4) ? (2 * (wv >> 2) + i) : wv; };     // >= 4: skipped

  const bool qwave = wv < 4;                 // waves 0..3 own the 64 queries (scores, softmax, PV, epilogue)
  const int q = 16 * (wv & 3) + lr;
  const long long qpix = gm.query_pixel(q);
  // ---- 2. S^T = K^ Q^T accumulated over output-channel chunks ----
  f32x4 s[WA_KT];
#pragma unroll
  for (int t = 0; t < WA_KT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * CC;
    if (ch > 0) __syncthreads();                       // previous chunk's score products are done with Ks / Qs
    {
      Frag8<T> wk[KCD], wq[KCD];
      load_w(wk, C + c0 + 16 * my_nt);
      load_w(wq, c0 + 16 * my_nt);
      f32x4 ak[KMT], aq[QMT];
#pragma unroll
      for (int i = 0; i < KMT; ++i) ak[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < QMT; ++i) aq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KCD; ++kc) {
#pragma unroll
        for (int i = 0; i < KMT; ++i) {
          const int mt = key_tile_of(i);
          if (mt < WA_KT) mma16(ak[i], wk[kc], load8(&D[16 * mt + lr][32 * kc + 8 * g]));
        }
#pragma unroll
        for (int i = 0; i < QMT; ++i) {
          const int qt = query_tile_of(i);
          const int qq = 16 * (qt & 3) + lr;
          if (qt < 4) mma16(aq[i], wq[kc], load8(&D[((qq >> 3) + 1) * 10 + (qq & 7) + 1][32 * kc + 8 * g]));
        }
      }
#pragma unroll
      for (int i = 0; i < KMT; ++i) {
        const int mt = key_tile_of(i);
        const int key = 16 * mt + lr;
        if (mt < WA_KT && key < WA_NK) {
          float v4[4] = {ak[i][0], ak[i][1], ak[i][2], ak[i][3]};
          const int kr = key / 10, kcc = key - kr * 10;
          long long pix;
          const bool inside = gm.key_pixel(key, pix);
          if (inside && kr >= 1 && kr <= 8 && kcc >= 1 && kcc <= 8) store4(qkv + pix * (3 * C) + C + c0 + 16 * my_nt + 4 * g, v4);
          const int cc = c0 + 16 * my_nt + 4 * g;
          const float* rp = (cc < C / 2) ? (rel_h + kr * (C / 2) + cc) : (rel_w + kcc * (C / 2) + (cc - C / 2));
          const f32x4 r4 = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] += r4[e];
          store4(&Ks[key][16 * my_nt + 4 * g], v4);
        }
      }
#pragma unroll
      for (int i = 0; i < QMT; ++i) {
        const int qt = query_tile_of(i);
        if (qt < 4) {
          const int qq = 16 * qt + lr;
          float v4[4] = {aq[i][0], aq[i][1], aq[i][2], aq[i][3]};
          store4(qkv + gm.query_pixel(qq) * (3 * C) + c0 + 16 * my_nt + 4 * g, v4);
          store4(&Qs[qq][16 * my_nt + 4 * g], v4);
        }
      }
    }
    __syncthreads();
    if (qwave) {
#pragma unroll
      for (int kc = 0; kc < CW / 32; ++kc) {
        const Frag8<T> qf = load8(&Qs[q][32 * kc + 8 * g]);
#pragma unroll
        for (int t = 0; t < WA_KT; ++t) {
          const Frag8<T> kf = load8(&Ks[16 * t + lr][32 * kc + 8 * g]);
          mma16(s[t], kf, qf);
        }
      }
    }
  }
  // ---- 3. softmax over the 100 real keys; lane (q, g) holds keys 16 t + 4 g + r ----
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
  Frag8<T> pf[4];
#pragma unroll
  for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 2 * c4 + (j >> 2);
      pf[c4].set(j, (t < WA_KT) ? s[t < WA_KT ? t : 0][j & 3] * inv : 0.f);
    }

  // ---- 4. per chunk: V = D Wv^T into LDS, then O^T = V^T P^T (transposing LDS reads) ----
  f32x4 oall[NCH][NTC];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * CC;
    __syncthreads();                                   // score / previous PV products are done with Ks
    {
      Frag8<T> wvv[KCD];
      load_w(wvv, 2 * C + c0 + 16 * my_nt);
      f32x4 av[KMT];
#pragma unroll
      for (int i = 0; i < KMT; ++i) av[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < KCD; ++kc)
#pragma unroll
        for (int i = 0; i < KMT; ++i) {
          const int mt = key_tile_of(i);
          if (mt < WA_KT) mma16(av[i], wvv[kc], load8(&D[16 * mt + lr][32 * kc + 8 * g]));
        }
#pragma unroll
      for (int i = 0; i < KMT; ++i) {
        const int mt = key_tile_of(i);
        if (mt < WA_KT) {
          const int key = 16 * mt + lr;
          float v4[4] = {av[i][0], av[i][1], av[i][2], av[i][3]};       // rows >= 100 of D are zero -> V = 0 there
          const int kr = key / 10, kcc = key - kr * 10;
          long long pix;
          if (key < WA_NK && gm.key_pixel(key, pix) && kr >= 1 && kr <= 8 && kcc >= 1 && kcc <= 8)
            store4(qkv + pix * (3 * C) + 2 * C + c0 + 16 * my_nt + 4 * g, v4);
          store4(&Ks[key][16 * my_nt + 4 * g], v4);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < NTC; ++mt) oall[ch][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (qwave) {
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt) {
          const Frag8<T> vf = load8_tr(&Ks[32 * c4 + 4 * g][16 * mt], &Ks[32 * c4 + 16 + 4 * g][16 * mt], LDK, lane);
          mma16(oall[ch][mt], vf, pf[c4]);
        }
    }
  }
  if (!qwave) return;
  // ---- 5. epilogue: residual (L = 0) or IWT^L + residual, straight into the concat buffer ----
  if constexpr (L == 0) {
    float v4[4] = {oall[0][0][0], oall[0][0][1], oall[0][0][2], oall[0][0][3]};
    float p4[4];
    load4(&D[((q >> 3) + 1) * 10 + (q & 7) + 1][4 * g], p4);      // x1 = attn1(x1) + x1: the residual is d itself
#pragma unroll
    for (int e = 0; e < 4; ++e) v4[e] += p4[e];
    store4(xc + qpix * 64 + 16 * k + 4 * g, v4);
  } else {
    float vv[4][S][S];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float bands[NB];
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int mt = 0; mt < NTC; ++mt) bands[ch * NTC + mt] = oall[ch][mt][r];
      Haar<L>::inv(bands, vv[r]);
    }
    const int by = 8 * gm.wy + (q >> 3), bx = 8 * gm.wx + (q & 7);
#pragma unroll
    for (int y = 0; y < S; ++y)
#pragma unroll
      for (int x = 0; x < S; ++x) {
        const long long pix = ((long long)gm.b * H + S * by + y) * Wf + S * bx + x;
        float p4[4];
        load4(xin + pix * 16 + 4 * g, p4);
        float v4[4] = {vv[0][y][x] + p4[0], vv[1][y][x] + p4[1], vv[2][y][x] + p4[2], vv[3][y][x] + p4[3]};
        store4(xc + pix * 64 + 16 * k + 4 * g, v4);
      }
  }
}

template <typename T, int C, int L> static size_t branch_fwd_smem() {
  constexpr int CC = (C < 64) ? C : 64;
  constexpr int KD = (C < 32) ? 32 : C;
  constexpr int CW = (CC < 32) ? 32 : CC;
  return sizeof(T) * (WA_KR * (KD + 8) + WA_KR * (CW + 8) + 64 * (CW + 8));
}

// returns M2T_UNSUPPORTED when this (dtype, level) is not supported by the fused kernel (caller uses the unfused path)
int launch_branch_fwd(int dt, int L, const void* X, const float* mean, const float* rstd, void* xc, int k, const void* Wqkv,
                      const float* rel_h, const float* rel_w, void* xin, void* dout, void* qkv, int B, int h, int w,
                      hipStream_t st) {
  if (dt == M2T_F32 && L == 2) return M2T_UNSUPPORTED;       // the fp32 D tile (128 x 264 x 4 B) does not fit LDS beside K^/Q
  const int nwin = B * (h / 8) * (w / 8);
  M2TProfScope ps(L == 0 ? M2T_PROF_ATTN_FWD_16 : (L == 1 ? M2T_PROF_ATTN_FWD_64 : M2T_PROF_ATTN_FWD_256), st);
#define GO(T_, C_, L_)                                                                                                   \
  {                                                                                                                      \
    const size_t sh = branch_fwd_smem<T_, C_, L_>();                                                                     \
    (void)hipFuncSetAttribute((const void*)branch_fwd_kernel<T_, C_, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    hipLaunchKernelGGL((branch_fwd_kernel<T_, C_, L_>), dim3(nwin), dim3(BR_THREADS), sh, st, (const T_*)X, mean, rstd, (T_*)xc, k,  \
                       (const T_*)Wqkv, rel_h, rel_w, (T_*)xin, (T_*)dout, (T_*)qkv, h, w);                             \
  }
  if (dt == M2T_F32) { if (L == 0) GO(float, 16, 0) else GO(float, 64, 1) }
  else { if (L == 0) GO(bf16_t, 16, 0) else if (L == 1) GO(bf16_t, 64, 1) else GO(bf16_t, 256, 2) }
#undef GO
  M2T_LAUNCH_CHECK();
  return 0;
}

// =======================================================================================
// Backward tail of one branch as ONE kernel (replaces halo_gather + the qkv data-gradient GEMM +
// branch_prep_bwd):
//     gqkv[m][C:3C]   = sum over the <= 4 covering windows of win[.][key][0:2C]   (overlap-add as a gather;
//                       written back so the weight-gradient GEMM can read dense rows)
//     g_d[m][:]       = gqkv[m][:] . Wqkv                (K = 3C, N = C; every output channel in ONE workgroup)
//     g_xin           = IWT^L(g_d) + g_xc[chunk k]       (DWT^T = IWT; the residual path of :145,153,161)
//     k = 0 : g_n[chunk 0] = g_xin
//     k > 0 : g_n[chunk k] = g_xin / 2 ;  g_xc[chunk k-1] += g_xin / 2            (:141,147,155)
// Workgroup = 64 branch pixels x all C output channels; wave w owns pixel tile w.  A lane ends with every band
// of 4 base channels of its pixel, so the inverse Haar butterflies are register-local (as in the forward).
// The gather is paid once per row (the tiled GEMM would repeat it per column block).
// =======================================================================================
template <typename T, int C, int L>
__global__ void __launch_bounds__(256)
branch_bwd_tail_kernel(T* __restrict__ gqkv, const T* __restrict__ win, const T* __restrict__ WT /*[C][3C]*/,
                       T* __restrict__ gxc, T* __restrict__ gn, int k, int h, int w, long long M) {
  static_assert(C == (16 << (2 * L)), "C = 16 * 4^L");
  constexpr int S = Haar<L>::S, NB = Haar<L>::N;
  constexpr int K = 3 * C;
  constexpr int NKC = (K + 31) / 32;
  constexpr int NT = C / 16;
  constexpr int WIT = (C * 4 + 255) / 256;          // weight vectors per thread per chunk
  __shared__ __attribute__((aligned(16))) T As[64][40];
  __shared__ __attribute__((aligned(16))) T Ws[C][40];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * 64;
  const int nh = h >> 3, nw = w >> 3;

  // this thread's A item: row (tid >> 2), 8 columns at kv = (tid & 3) * 8 of every chunk
  const int arow = tid >> 2, akv = (tid & 3) * 8;
  const long long am = m0 + arow;
  // covering windows of the row's pixel (for the gathered k|v columns)
  int wys[2], krs[2], ny = 1, wxs[2], kcs[2], nx = 1;
  long long ab = 0;
  {
    const int x = (int)(am % w);
    const long long qq = am / w;
    const int y = (int)(qq % h);
    ab = qq / h;
    wys[0] = y >> 3; krs[0] = (y & 7) + 1;
    if ((y & 7) == 0 && wys[0] > 0) { wys[1] = wys[0] - 1; krs[1] = 9; ny = 2; }
    else if ((y & 7) == 7 && wys[0] < nh - 1) { wys[1] = wys[0] + 1; krs[1] = 0; ny = 2; }
    wxs[0] = x >> 3; kcs[0] = (x & 7) + 1;
    if ((x & 7) == 0 && wxs[0] > 0) { wxs[1] = wxs[0] - 1; kcs[1] = 9; nx = 2; }
    else if ((x & 7) == 7 && wxs[0] < nw - 1) { wxs[1] = wxs[0] + 1; kcs[1] = 0; nx = 2; }
  }
  Frag8<T> ra, rw[WIT];
  auto fetch = [&](int kc) {
    const int kk = 32 * kc + akv;
    ra = frag_zero<T>();
    if (am < M && kk < K) {
      if (kk < C) {
        ra = load8(gqkv + am * K + kk);
      } else {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int a = 0; a < ny; ++a)
          for (int c = 0; c < nx; ++c) {
            const long long wi = (ab * nh + wys[a]) * nw + wxs[c];
            const Frag8<T> f = load8(win + (wi * 100 + krs[a] * 10 + kcs[c]) * (2 * C) + (kk - C));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += f.get(e);
          }
#pragma unroll
        for (int e = 0; e < 8; ++e) ra.set(e, acc[e]);
      }
    }
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int idx = tid + it * 256;
      const int n = idx >> 2, kv = (idx & 3) * 8;
      rw[it] = frag_zero<T>();
      if (n < C && 32 * kc + kv < K) rw[it] = load8(WT + (long long)n * K + 32 * kc + kv);
    }
  };
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
#pragma unroll 1
  for (int kc = 0; kc < NKC; ++kc) {
    if (kc > 0) __syncthreads();
    store8(&As[arow][akv], ra);
    if (am < M && 32 * kc + akv >= C && 32 * kc + akv < K) store8(gqkv + am * K + 32 * kc + akv, ra);   // materialise for the wgrad
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int idx = tid + it * 256;
      if ((idx >> 2) < C) store8(&Ws[idx >> 2][(idx & 3) * 8], rw[it]);
    }
    __syncthreads();
    if (kc + 1 < NKC) fetch(kc + 1);
    const Frag8<T> af = load8(&As[16 * wv + lr][8 * g]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const Frag8<T> wf = load8(&Ws[16 * nt + lr][8 * g]);
      mma16(acc[nt], wf, af);
    }
  }
  // epilogue: lane (pixel m = m0 + 16 wv + lr, g) holds channels 16 nt + 4 g + r = band nt of base channels 4g..4g+3
  const long long m = m0 + 16 * wv + lr;
  if (m >= M) return;
  const int bx = (int)(m % w);
  const long long q2 = m / w;
  const int by = (int)(q2 % h);
  const long long b = q2 / h;
  const int H = h * S, Wf = w * S;
  float vv[4][S][S];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float bands[NB];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bands[nt] = acc[nt][r];
    Haar<L>::inv(bands, vv[r]);
  }
#pragma unroll
  for (int y = 0; y < S; ++y)
#pragma unroll
    for (int x = 0; x < S; ++x) {
      const long long pix = (b * H + S * by + y) * Wf + S * bx + x;
      float p4[4], q4[4];
      load4(gxc + pix * 64 + k * 16 + 4 * g, p4);
#pragma unroll
      for (int c = 0; c < 4; ++c) q4[c] = vv[c][y][x] + p4[c];
      if (k == 0) {
        store4(gn + pix * 64 + 4 * g, q4);
      } else {
        float pp[4];
        load4(gxc + pix * 64 + (k - 1) * 16 + 4 * g, pp);
#pragma unroll
        for (int c = 0; c < 4; ++c) { q4[c] *= 0.5f; pp[c] += q4[c]; }
        store4(gn + pix * 64 + k * 16 + 4 * g, q4);
        store4(gxc + pix * 64 + (k - 1) * 16 + 4 * g, pp);
      }
    }
}

int launch_branch_bwd_tail(int dt, int L, void* gqkv, const void* win, const void* WT, void* gxc, void* gn, int k, int B, int h,
                           int w, hipStream_t st) {
  const long long M = (long long)B * h * w;
  const int nblk = (int)((M + 63) / 64);
  M2TProfScope ps(M2T_PROF_GEMM_QKV_DGRAD, st);
#define GO(T_, C_, L_) hipLaunchKernelGGL((branch_bwd_tail_kernel<T_, C_, L_>), dim3(nblk), dim3(256), 0, st, (T_*)gqkv, (const T_*)win, (const T_*)WT, (T_*)gxc, (T_*)gn, k, h, w, M)
  if (dt == M2T_F32) { if (L == 0) GO(float, 16, 0); else if (L == 1) GO(float, 64, 1); else GO(float, 256, 2); }
  else { if (L == 0) GO(bf16_t, 16, 0); else if (L == 1) GO(bf16_t, 64, 1); else GO(bf16_t, 256, 2); }
#undef GO
  M2T_LAUNCH_CHECK();
  return 0;
}
