// k_gemm.hip -- matrix-core GEMMs over pixel-major activations (gfx950).
//
//  gemm_nt :  Y[M][N] = A[M][K] * W[N][K]^T          1x1 convs (qkv projection, tail
//             expansions) and their data gradients (with the transposed packed weight).
//             A-side variants: plain rows | GELU(rows) | pixel-UNshuffle gather;
//             epilogues: plain | +bias | +bias & pixel-shuffle scatter | * GELU'(aux).
//             (models/M2Trans_network.py:42-47,52-54,281,307 and their autograd)
//  wgrad_tn:  dW[N][K] = sum_m G[m][N] * X[m][K]      weight gradients; split over M into
//             fp32 slabs that a deterministic reduction sums (no atomics).
//
// Tile product orientation: the weight rows are the MFMA "A" operand and the activation
// rows the "B" operand, so each lane ends up with 16 CONSECUTIVE output channels of one
// pixel -> 32/64-byte vector stores along the contiguous NHWC axis.
#include <cstdlib>
#include "m2t_kernels.h"
#include "m2t_gemm_load.h"

#define GEMM_BM 128
#define GEMM_BN 64
#define GEMM_BK 32
#define GEMM_PAD 8

template <typename T, int AMODE, int EMODE>
__global__ void __launch_bounds__(256)
gemm_nt_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, T* __restrict__ Y, int ldy,
               const float* __restrict__ bias, const T* __restrict__ aux, int ldaux, T* __restrict__ Y2, long long M, int N,
               int K, ShufGeom sg) {
  __shared__ __attribute__((aligned(16))) T As[GEMM_BM][GEMM_BK + GEMM_PAD];
  __shared__ __attribute__((aligned(16))) T Ws[GEMM_BN][GEMM_BK + GEMM_PAD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  // 1-D grid, XCD-aware, column tile fastest: the workgroups that share a row strip of A are neighbours behind one L2
  const int gy = (N + GEMM_BN - 1) / GEMM_BN, L = xcd_block_index();
  const long long m0 = (long long)(L / gy) * GEMM_BM;
  const int n0 = (L % gy) * GEMM_BN;

  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // register-staged pipeline: the next k-chunk's global loads are in flight while the current one is multiplied
  Frag8<T> ra[2], rw;
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      const int row = idx >> 2, kv = idx & 3;
      const long long m = m0 + row;
      const int k = k0 + kv * 8;
      ra[it] = frag_zero<T>();
      if (m < M && k < K) ra[it] = gemm_load_a<T, AMODE>(A, lda, m, k, sg);
    }
    {
      const int row = tid >> 2, kv = tid & 3;
      const int n = n0 + row, k = k0 + kv * 8;
      rw = frag_zero<T>();
      if (n < N && k < K) rw = load8(W + (long long)n * K + k);
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += GEMM_BK) {
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;
      store8(&As[idx >> 2][(idx & 3) * 8], ra[it]);
    }
    store8(&Ws[tid >> 2][(tid & 3) * 8], rw);
    __syncthreads();
    if (k0 + GEMM_BK < K) fetch(k0 + GEMM_BK);
    // ---- 2 x 4 tile products per wave ----
    Frag8<T> xf[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xf[mt] = load8(&As[32 * wv + 16 * mt + lr][g * 8]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int nl = 16 * (lr >> 2) + 4 * nt + (lr & 3);
      const Frag8<T> wf = load8(&Ws[nl][g * 8]);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) mma16(acc[mt][nt], wf, xf[mt]);
    }
  }

  // ---- epilogue: lane (m = lr, g) holds n_local = 16 g + 4 nt + r ----
  const int nn = n0 + 16 * g;
  if (nn >= N) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const long long m = m0 + 32 * wv + 16 * mt + lr;
    if (m >= M) continue;
    float v[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r];
    if (EMODE == M2T_E_PLAIN) {
      if (ldy == M2T_LD_P64) store16f(Y + p64(M, m, nn), v);
      else store16f(Y + m * ldy + nn, v);
    } else if (EMODE == M2T_E_BIAS) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += bias[nn + e];
      store16f(Y + m * ldy + nn, v);
    } else if (EMODE == M2T_E_BIAS_SHUF) {
      // column n' = sub*C + c  <->  torch channel c*r*r + sub ; destination pixel (h*r+i, w*r+j)
      const int sub = nn / sg.C, c0 = nn - sub * sg.C;
      const int i = sub / sg.r, j = sub - i * sg.r;
      const int rr = sg.r * sg.r;
      float dv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) gelu_tail_both<T>(v[e] + bias[(c0 + e) * rr + sub], v[e], dv[e]);   // activation + derivative
      const int w = (int)(m % sg.W);
      const long long q = m / sg.W;
      const int h = (int)(q % sg.H);
      const long long b = q / sg.H;
      const long long pix = (b * sg.H * sg.r + (h * sg.r + i)) * ((long long)sg.W * sg.r) + (w * sg.r + j);
      store16f(Y + pix * sg.C + c0, v);
      store16f(Y2 + pix * sg.C + c0, dv);
    } else if (EMODE == M2T_E_BIAS_GELU) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = gelu_erf(v[e] + bias[nn + e]);
      store16f(Y + m * ldy + nn, v);
    } else if (EMODE == M2T_E_BIAS_RELU) {
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = fmaxf(v[e] + bias[nn + e], 0.f);
      store16f(Y + m * ldy + nn, v);
    } else if (EMODE == M2T_E_BIAS_RESID) {
      float p[16];
      load16f(aux + m * ldaux + nn, p);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] += (bias ? bias[nn + e] : 0.f) + p[e];
      store16f(Y + m * ldy + nn, v);
    } else {   // M2T_E_GELU_GRAD: aux holds the stored derivative gelu'(t)
      float p[16];
      load16f(aux + m * ldaux + nn, p);
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] *= p[e];
      store16f(Y + m * ldy + nn, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// gemm_nt_wide: the plain/plain case for the wide shapes (the C = 256 qkv projection 256 -> 768 and its data
// gradient 768 -> 256).  BM x 128 output tile, 64-deep stages, double-buffered LDS with ONE barrier per stage and
// the next stage's global loads in registers while the current one is multiplied: 32 (BM = 128) or 16 (BM = 64)
// MFMAs per wave per barrier instead of 8.
// ---------------------------------------------------------------------------------------
template <typename T, int BM, int EMODE>
__global__ void __launch_bounds__(256)
gemm_nt_wide_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, T* __restrict__ Y, int ldy, long long M, int N,
                    int K, const float* __restrict__ bias, const T* __restrict__ aux, int ldaux) {
  constexpr int BN = 128, BK = 64, LD = BK + 8, MT = BM / 64, NA = BM / 32, NW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*As)[BM][LD] = reinterpret_cast<T(*)[BM][LD]>(smem);
  T(*Ws)[BN][LD] = reinterpret_cast<T(*)[BN][LD]>(smem + sizeof(T) * 2 * BM * LD);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int gy = N / BN, L = xcd_block_index();         // 1-D grid, XCD-aware, column tile fastest (see gemm_nt_kernel)
  const long long m0 = (long long)(L / gy) * BM;
  const int n0 = (L % gy) * BN;
  f32x4 acc[MT][8];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag8<T> ra[NA], rw[NW];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = tid + it * 256;
      const long long m = m0 + (idx >> 3);
      ra[it] = frag_zero<T>();
      if (m < M) ra[it] = load8(A + m * lda + k0 + (idx & 7) * 8);
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = tid + it * 256;
      rw[it] = load8(W + (long long)(n0 + (idx >> 3)) * K + k0 + (idx & 7) * 8);
    }
  };
  fetch(0);
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int idx = tid + it * 256;
      store8(&As[buf][idx >> 3][(idx & 7) * 8], ra[it]);
    }
#pragma unroll
    for (int it = 0; it < NW; ++it) {
      const int idx = tid + it * 256;
      store8(&Ws[buf][idx >> 3][(idx & 7) * 8], rw[it]);
    }
    __syncthreads();      // (the buffer written next iteration was last read before this barrier)
    if (k0 + BK < K) fetch(k0 + BK);
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      Frag8<T> xf[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) xf[mt] = load8(&As[buf][16 * MT * wv + 16 * mt + lr][32 * kc + 8 * g]);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const int nl = 64 * (nt >> 2) + 16 * (lr >> 2) + 4 * (nt & 3) + (lr & 3);
        const Frag8<T> wf = load8(&Ws[buf][nl][32 * kc + 8 * g]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma16(acc[mt][nt], wf, xf[mt]);
      }
    }
  }
  // lane (m = lr, g) holds channels n0 + 64 h + 16 g + 0..15 for h = 0, 1
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = m0 + 16 * MT * wv + 16 * mt + lr;
    if (m >= M) continue;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float v[16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][4 * h + nt][r];
      const int nn = n0 + 64 * h + 16 * g;
      if constexpr (EMODE == M2T_E_BIAS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += bias[nn + e];
      } else if constexpr (EMODE == M2T_E_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = gelu_erf(v[e] + bias[nn + e]);
      } else if constexpr (EMODE == M2T_E_BIAS_RESID) {
        float p[16];
        load16f(aux + m * ldaux + nn, p);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += (bias ? bias[nn + e] : 0.f) + p[e];
      }
      store16f(Y + m * ldy + nn, v);
    }
  }
}
template <typename T, int BM, int EMODE>
static int launch_gemm_nt_wide(const m2t_gemm_args& a, hipStream_t st) {
  const size_t sh = sizeof(T) * 2 * (BM + 128) * 72;
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)gemm_nt_wide_kernel<T, BM, EMODE>, (int)sh)) return rc__;
  dim3 grid((unsigned)(ceil_divll(a.M, BM) * (a.N / 128)));
  hipLaunchKernelGGL((gemm_nt_wide_kernel<T, BM, EMODE>), grid, dim3(256), sh, st, (const T*)a.A, a.lda, (const T*)a.W, (T*)a.Y, a.ldy,
                     a.M, a.N, a.K, a.bias, (const T*)a.aux, a.ldaux);
  M2T_LAUNCH_CHECK();
  return 0;
}
template <typename T, int EMODE>
static int launch_gemm_nt_wide_bm(const m2t_gemm_args& a, hipStream_t st) {
  // enough 128-column blocks to fill the chip with BM = 128?  otherwise halve the row tile
  const long long blocks128 = ceil_divll(a.M, 128) * (a.N / 128);
  if (sizeof(T) == 2 && blocks128 >= 512) return launch_gemm_nt_wide<T, 128, EMODE>(a, st);
  return launch_gemm_nt_wide<T, 64, EMODE>(a, st);
}

// =======================================================================================
// fp32 (parity mode) GEMMs on v_mfma_f32_32x32x2_f32 (round 5).  The kernels above were shaped for bf16: in fp32 each `mma16` is a
// chain of eight DEPENDENT 16x16x4 products (40-cycle dependent latency on a 32-cycle issue) whose operands are fetched from LDS
// into the same registers right before use, one wave per SIMD -- 13 % of the fp32 MFMA rate in the C = 256 projections.  The 32x32x2
// form needs ONE operand register per lane for 4096 FLOP (half the LDS bytes per FLOP of 16x16x4), its dependent latency equals its
// issue interval (64 cycles), and a wave's 32 x 32 NB tile gives NB independent accumulators.
//   lane l = (i = l & 31, h = l >> 5):  A operand = W[n0 + i][k], B operand = X[m0 + i][k], k = 8 q + 4 h + s in step s of block q
//   (one ds_read_b128 per operand and 8-deep block; the contraction order is a fixed permutation of k, the same for every output),
//   D[row = 8 (r >> 2) + 4 h + (r & 3)][col = i] -> a lane holds 4 consecutive output channels of ONE pixel per register quad.
// =======================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
thread_local int g_m2t_f32_fast = 1;       // option fp32_fast of the plan being run (m2t_api.hip); 0 = the round-4 kernels
__device__ __forceinline__ void mfma32f(f32x16& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0); }

template <int NB>
__global__ void __launch_bounds__(256, 2)
gemm_nt_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, float* __restrict__ Y, int ldy, long long M,
                   int N, int K) {
  constexpr int BM = 128, BN = 32 * NB, BK = 32, LD = BK + 4, ROWS = BM + BN, NIT = ROWS / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float(*S)[ROWS][LD] = reinterpret_cast<float(*)[ROWS][LD]>(smem);         // [2][activation rows | weight rows][k]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int gy = N / BN, L = xcd_block_index();         // column tile fastest: the tiles sharing a row strip of A sit behind one L2
  const long long m0 = (long long)(L / gy) * BM;
  const int n0 = (L % gy) * BN;
  f32x16 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  const int srow = tid >> 3, skv = (tid & 7) * 4;       // staging slot: rows srow + 32 it, one float4 of the 32-deep stage
  f32x4 rg[NIT];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const long long m = m0 + srow + 32 * it;
      rg[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (m < M) rg[it] = *reinterpret_cast<const f32x4*>(A + m * lda + k0 + skv);
    }
#pragma unroll
    for (int it = 4; it < NIT; ++it)
      rg[it] = *reinterpret_cast<const f32x4*>(W + (long long)(n0 + srow + 32 * (it - 4)) * K + k0 + skv);
  };
  fetch(0);
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += BK, buf ^= 1) {
#pragma unroll
    for (int it = 0; it < NIT; ++it) *reinterpret_cast<f32x4*>(&S[buf][srow + 32 * it][skv]) = rg[it];
    __syncthreads();      // (the buffer written next iteration was last read before this barrier)
    if (k0 + BK < K) fetch(k0 + BK);
    const float* sx = &S[buf][32 * wv + li][4 * lh];
    const float* sw = &S[buf][BM + li][4 * lh];
    f32x4 xf[2], wf[2][NB];
    xf[0] = *reinterpret_cast<const f32x4*>(sx);
#pragma unroll
    for (int b = 0; b < NB; ++b) wf[0][b] = *reinterpret_cast<const f32x4*>(sw + 32 * b * LD);
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      if (q + 1 < BK / 8) {                              // operands of the next 8-deep block fly under this block's products
        xf[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(sx + 8 * (q + 1));
#pragma unroll
        for (int b = 0; b < NB; ++b) wf[(q + 1) & 1][b] = *reinterpret_cast<const f32x4*>(sw + 32 * b * LD + 8 * (q + 1));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int b = 0; b < NB; ++b) mfma32f(acc[b], wf[q & 1][b][s], xf[q & 1][s]);
    }
  }
  const long long m = m0 + 32 * wv + li;
  if (m < M) {
    float* y = Y + m * ldy + n0 + 4 * lh;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        *reinterpret_cast<f32x4*>(y + 32 * b + 8 * rq) = (f32x4){acc[b][4 * rq], acc[b][4 * rq + 1], acc[b][4 * rq + 2], acc[b][4 * rq + 3]};
  }
}
template <int NB>
static int launch_gemm_nt_f32_nb(const m2t_gemm_args& a, hipStream_t st) {
  const size_t sh = sizeof(float) * 2 * (128 + 32 * NB) * 36;
  if (int rc__ = m2t_ensure_dynamic_lds((const void*)gemm_nt_f32_kernel<NB>, (int)sh)) return rc__;
  dim3 grid((unsigned)(ceil_divll(a.M, 128) * (a.N / (32 * NB))));
  hipLaunchKernelGGL((gemm_nt_f32_kernel<NB>), grid, dim3(256), sh, st, (const float*)a.A, a.lda, (const float*)a.W, (float*)a.Y, a.ldy,
                     a.M, a.N, a.K);
  M2T_LAUNCH_CHECK();
  return 0;
}
static bool gemm_nt_f32_ok(int amode, int emode, const m2t_gemm_args& a) {
  return g_m2t_f32_fast && amode == M2T_A_PLAIN && emode == M2T_E_PLAIN && a.N % 32 == 0 && a.K % 32 == 0 && a.lda % 4 == 0 &&
         a.ldy > 0 && a.ldy % 4 == 0 && a.ldy != M2T_LD_P64;
}
static int launch_gemm_nt_f32(const m2t_gemm_args& a, hipStream_t st) {
  // widest column tile (fewest re-reads of the activation rows) that still gives every CU a workgroup
  const long long rows = ceil_divll(a.M, 128);
  const int nb32 = a.N / 32;
  if (nb32 % 3 == 0 && rows * (nb32 / 3) >= 256) return launch_gemm_nt_f32_nb<3>(a, st);
  if (nb32 % 2 == 0 && rows * (nb32 / 2) >= 256) return launch_gemm_nt_f32_nb<2>(a, st);
  if (rows * nb32 >= 256 || (nb32 % 3 && nb32 % 2)) return launch_gemm_nt_f32_nb<1>(a, st);
  if (nb32 % 2 == 0 && rows * (nb32 / 2) >= 128) return launch_gemm_nt_f32_nb<2>(a, st);
  return launch_gemm_nt_f32_nb<1>(a, st);
}

template <typename T>
static int launch_gemm_nt_t(int amode, int emode, const m2t_gemm_args& a, hipStream_t st) {
  if (a.K % 8 || a.N % 16) return m2t_set_error(-2, "gemm_nt: K must be a multiple of 8 and N of 16");
  if constexpr (sizeof(T) == 4) {
    if (gemm_nt_f32_ok(amode, emode, a)) return launch_gemm_nt_f32(a, st);
  }
  if (amode == M2T_A_PLAIN && a.N % 128 == 0 && a.K % 64 == 0 && a.K >= 128 && a.ldy != M2T_LD_P64) {
    if (emode == M2T_E_PLAIN) return launch_gemm_nt_wide_bm<T, M2T_E_PLAIN>(a, st);
    // the Swin-T GEMMs of stages 3 / 4 (and fc1 of stage 2): bias, bias + GELU, bias + residual epilogues
    if (sizeof(T) == 2 && emode == M2T_E_BIAS) return launch_gemm_nt_wide_bm<T, M2T_E_BIAS>(a, st);
    if (sizeof(T) == 2 && emode == M2T_E_BIAS_GELU) return launch_gemm_nt_wide_bm<T, M2T_E_BIAS_GELU>(a, st);
    if (sizeof(T) == 2 && emode == M2T_E_BIAS_RESID) return launch_gemm_nt_wide_bm<T, M2T_E_BIAS_RESID>(a, st);
  }
  dim3 grid((unsigned)(ceil_divll(a.M, GEMM_BM) * ceil_div(a.N, GEMM_BN)));
  ShufGeom sg{a.H, a.Wd, a.r, a.C, a.halo_win, a.M};
#define GO(AM, EM)                                                                                              \
  hipLaunchKernelGGL((gemm_nt_kernel<T, AM, EM>), grid, dim3(256), 0, st, (const T*)a.A, a.lda, (const T*)a.W, \
                     (T*)a.Y, a.ldy, a.bias, (const T*)a.aux, a.ldaux, (T*)a.Y2, a.M, a.N, a.K, sg)
  if (amode == M2T_A_PLAIN && emode == M2T_E_PLAIN) GO(M2T_A_PLAIN, M2T_E_PLAIN);
  else if (amode == M2T_A_PLAIN && emode == M2T_E_BIAS) GO(M2T_A_PLAIN, M2T_E_BIAS);
  else if (amode == M2T_A_PLAIN && emode == M2T_E_BIAS_SHUF) GO(M2T_A_PLAIN, M2T_E_BIAS_SHUF);
  else if (amode == M2T_A_PLAIN && emode == M2T_E_BIAS_GELU) GO(M2T_A_PLAIN, M2T_E_BIAS_GELU);
  else if (amode == M2T_A_PLAIN && emode == M2T_E_BIAS_RESID) GO(M2T_A_PLAIN, M2T_E_BIAS_RESID);
  else if (amode == M2T_A_PLAIN && emode == M2T_E_BIAS_RELU) GO(M2T_A_PLAIN, M2T_E_BIAS_RELU);
  else if (amode == M2T_A_UNSHUF && emode == M2T_E_PLAIN) GO(M2T_A_UNSHUF, M2T_E_PLAIN);
  else if (amode == M2T_A_UNSHUF && emode == M2T_E_GELU_GRAD) GO(M2T_A_UNSHUF, M2T_E_GELU_GRAD);
  else return m2t_set_error(-2, "gemm_nt: unsupported (A mode, epilogue) combination");
#undef GO
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_gemm_nt(int dt, int amode, int emode, const m2t_gemm_args& a, hipStream_t st) {
  if (dt == M2T_F32) return launch_gemm_nt_t<float>(amode, emode, a, st);
  return launch_gemm_nt_t<bf16_t>(amode, emode, a, st);
}

// =======================================================================================
// tail_expand: the two upsampler 1x1 convs (models/M2Trans_network.py:42-47,52-54) as one specialised
// kernel:  t[(b, r h + i, r w + j)][c] = bias[c r^2 + i r + j] + sum_k act(x[(b,h,w)][k]) W[(i r + j) 64 + c][k]
// K = 64, N = 64 r^2 (packed rows already in sub-pixel-major order).  A persistent workgroup keeps the WHOLE
// weight matrix in LDS and sweeps 128-row tiles: the activation tile is staged once (GELU applied once, not
// once per 64-column block), the next tile's global loads fly while the current one is multiplied, and each
// 64-column group is exactly one sub-pixel, so a lane stores 16 consecutive channels of one output pixel.
// Every expansion of the reference's tail is followed by GELU (:43-46,52-53), and the erf behind it is what
// bounds the high-resolution kernels (VALU, not HBM): so the epilogue evaluates it ONCE per element and stores
// both the activation Y = gelu(t) and the derivative Yd = gelu'(t); the consumers (next expansion, tail conv,
// their weight / data gradients) read those instead of re-evaluating erf on every (halo-inflated) pass.
// =======================================================================================
template <typename T, int NSUB>
__global__ void __launch_bounds__(256)
tail_expand_kernel(const T* __restrict__ X, const T* __restrict__ Wp, const float* __restrict__ bias, T* __restrict__ Y,
                   T* __restrict__ Yd, long long M, int H, int Wd, int r, int tiles_per_block, int x_p64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Ws)[72] = reinterpret_cast<T(*)[72]>(smem);                                   // [64 NSUB][72]
  T(*As)[72] = reinterpret_cast<T(*)[72]>(smem + sizeof(T) * 64 * NSUB * 72);      // [128][72]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  for (int idx = tid; idx < 64 * NSUB * 8; idx += 256) store8(&Ws[idx >> 3][(idx & 7) * 8], load8(Wp + (long long)(idx >> 3) * 64 + (idx & 7) * 8));
  const long long ntiles = (M + 127) / 128;
  const long long t0 = (long long)blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
  Frag8<T> ra[4];
  auto fetch = [&](long long t) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;
      const long long m = t * 128 + (idx >> 3);
      ra[it] = frag_zero<T>();
      if (m < M) ra[it] = load8(X + (x_p64 ? p64(M, m, (idx & 7) * 8) : m * 64 + (idx & 7) * 8));
    }
  };
  if (t0 < t1) fetch(t0);
  const int rr = r * r;
  for (long long t = t0; t < t1; ++t) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;
      store8(&As[idx >> 3][(idx & 7) * 8], ra[it]);
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
    Frag8<T> xf[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) xf[mt][kc] = load8(&As[32 * wv + 16 * mt + lr][32 * kc + 8 * g]);
    // destination pixels of this lane's two rows
    long long pixbase[2];
    bool ok[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const long long m = t * 128 + 32 * wv + 16 * mt + lr;
      ok[mt] = m < M;
      const int w = (int)(m % Wd);
      const long long q = m / Wd;
      const int h = (int)(q % H);
      const long long b = q / H;
      pixbase[mt] = (b * H * r + (long long)h * r) * ((long long)Wd * r) + (long long)w * r;
    }
#pragma unroll 1
    for (int sub = 0; sub < NSUB; ++sub) {
      f32x4 acc[2][4];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const int nl = 64 * sub + 16 * (lr >> 2) + 4 * nt + (lr & 3);
          const Frag8<T> wf = load8(&Ws[nl][32 * kc + 8 * g]);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) mma16(acc[mt][nt], wf, xf[mt][kc]);
        }
      const int i = sub / r, j = sub - i * r;
      float bv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) bv[e] = bias[(16 * g + e) * rr + sub];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (!ok[mt]) continue;
        float v[16], dv[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          float b4[4], a4[4], d4[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) b4[q4] = bv[4 * nt + q4];
          gelu_tail_both4<T>(acc[mt][nt], b4, a4, d4);
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) { v[4 * nt + q4] = a4[q4]; dv[4 * nt + q4] = d4[q4]; }
        }
        const long long pix = pixbase[mt] + (long long)i * Wd * r + j;
        store16f(Y + pix * 64 + 16 * g, v);
        store16f(Yd + pix * 64 + 16 * g, dv);
      }
    }
  }
}
// ---------------------------------------------------------------------------------------
// fp32 x2 expansion on v_mfma_f32_32x32x2_f32 (round 5; r = 2, image rows of whole 128-pixel tiles).  tail_expand_kernel<float> runs one
// 4-wave workgroup per CU whose waves alternate between chains of dependent 16x16x4 products and the erf epilogue (792 us for the
// second expansion at batch 16: 34 GFLOP and 2.4 GB).  Here 8 waves share the CU: the whole 256 x 64 weight matrix and a double-buffered
// 128-row tile in LDS ([row][64 + 4]), wave (wm, wn) = 32 pixels x the two sub-pixels 2 wn, 2 wn + 1 (four 32 x 32 accumulators), the
// pixels as the A operand so that a lane ends with ONE channel of 16 pixels: the stores are 128-byte row pieces, the bias is one value
// per lane and accumulator, and one wave's epilogue runs under the other wave's MFMAs.  GELU and its derivative by the same
// gelu_erf_both as the kernel above (same values for the same pre-activation; the contraction order differs).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
tail_expand_f32_kernel(const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias, float* __restrict__ Y,
                       float* __restrict__ Yd, int M, int H, int Wd, int tiles_per_block, int x_p64) {
  constexpr int LD = 68;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float(*Ws)[LD] = reinterpret_cast<float(*)[LD]>(smem);                                          // [256][68]
  float(*As)[128][LD] = reinterpret_cast<float(*)[128][LD]>(smem + sizeof(float) * 256 * LD);     // [2][128][68]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, wm = wv & 3, wn = wv >> 2;
  for (int idx = tid; idx < 256 * 16; idx += 512)
    *reinterpret_cast<f32x4*>(&Ws[idx >> 4][(idx & 15) * 4]) = *reinterpret_cast<const f32x4*>(Wp + (long long)(idx >> 4) * 64 + (idx & 15) * 4);
  const int ntiles = M >> 7;
  const int t0 = blockIdx.x * tiles_per_block, t1 = min(ntiles, t0 + tiles_per_block);
  f32x4 ra[4];
  auto fetch = [&](int t) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 512;
      const long long m = (long long)t * 128 + (idx >> 4);
      const int k = (idx & 15) * 4;
      ra[it] = *reinterpret_cast<const f32x4*>(X + (x_p64 ? p64(M, m, k) : m * 64 + k));
    }
  };
  if (t0 < t1) fetch(t0);
  // bias of this lane's channel in each of the wave's four accumulators: block (s, ch) = sub-pixel 2 wn + s, channels 32 ch + li
  float bv[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) bv[s][ch] = bias[(32 * ch + li) * 4 + 2 * wn + s];
  int buf = 0;
  for (int t = t0; t < t1; ++t, buf ^= 1) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 512;
      *reinterpret_cast<f32x4*>(&As[buf][idx >> 4][(idx & 15) * 4]) = ra[it];
    }
    __syncthreads();      // (tile staged; the buffer written next iteration was last read before this barrier; Ws staged on the first pass)
    if (t + 1 < t1) fetch(t + 1);
    f32x16 acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][ch][r] = 0.f;
    const float* sx = &As[buf][32 * wm + li][4 * lh];
    const float* sw = &Ws[128 * wn + li][4 * lh];
    f32x4 xf[2], wf[2][4];
    xf[0] = *reinterpret_cast<const f32x4*>(sx);
#pragma unroll
    for (int b = 0; b < 4; ++b) wf[0][b] = *reinterpret_cast<const f32x4*>(sw + 32 * b * LD);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (q + 1 < 8) {
        xf[(q + 1) & 1] = *reinterpret_cast<const f32x4*>(sx + 8 * (q + 1));
#pragma unroll
        for (int b = 0; b < 4; ++b) wf[(q + 1) & 1][b] = *reinterpret_cast<const f32x4*>(sw + 32 * b * LD + 8 * (q + 1));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int b = 0; b < 4; ++b) mfma32f(acc[b >> 1][b & 1], xf[q & 1][e], wf[q & 1][b][e]);
    }
    // tile = 128 consecutive pixels of ONE image row (Wd % 128 == 0): (b, h, w0) once per tile, wave-uniform
    const int m0 = t << 7;
    const int w0 = m0 % Wd, qh = m0 / Wd, h = qh % H, bi = qh / H;
    const long long rowbase = ((long long)bi * H * 2 + 2 * h) * (2LL * Wd);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int sub = 2 * wn + s, si = sub >> 1, sj = sub & 1;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        const float b4[4] = {bv[s][ch], bv[s][ch], bv[s][ch], bv[s][ch]};
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          float a4[4], d4[4];
          const f32x4 v = (f32x4){acc[s][ch][4 * rq], acc[s][ch][4 * rq + 1], acc[s][ch][4 * rq + 2], acc[s][ch][4 * rq + 3]};
          gelu_tail_both4<float>(v, b4, a4, d4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int w = w0 + 32 * wm + 8 * rq + 4 * lh + e;           // D row = pixel 8 (r >> 2) + 4 h + (r & 3), col = channel li
            const long long off = (rowbase + (long long)si * 2 * Wd + 2 * w + sj) * 64 + 32 * ch + li;
            Y[off] = a4[e];
            Yd[off] = d4[e];
          }
        }
      }
    }
  }
}

template <typename T>
static int launch_tail_expand_t(const T* X, const T* Wp, const float* bias, T* Y, T* Yd, long long M, int H, int Wd, int r,
                                bool x_p64, hipStream_t st) {
  const long long ntiles = (M + 127) / 128;
  int nblk = (int)std::min<long long>(ntiles, 1024);
  const int tpb = (int)ceil_divll(ntiles, nblk);
  nblk = (int)ceil_divll(ntiles, tpb);
#define GO(NS_)                                                                                                        \
  {                                                                                                                    \
    const size_t sh = sizeof(T) * (64 * NS_ + 128) * 72;                                                               \
    (void)hipFuncSetAttribute((const void*)tail_expand_kernel<T, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    hipLaunchKernelGGL((tail_expand_kernel<T, NS_>), dim3(nblk), dim3(256), sh, st, X, Wp, bias, Y, Yd, M, H, Wd, r, tpb, x_p64 ? 1 : 0); \
  }
  if (r == 2) GO(4)
  else if (r == 3) GO(9)
  else return m2t_set_error(-2, "tail_expand: r must be 2 or 3");
#undef GO
  M2T_LAUNCH_CHECK();
  return 0;
}
int launch_tail_expand(int dt, const void* X, const void* Wp, const float* bias, void* Y, void* Yd, long long M, int H, int Wd,
                       int r, bool x_p64, hipStream_t st) {
  if (dt == M2T_F32 && r == 3) {
    // fp32 x3: the 576 x 64 fp32 weight matrix does not fit LDS beside the tile -> generic tiled GEMM
    m2t_gemm_args ga{};
    ga.A = X; ga.lda = x_p64 ? M2T_LD_P64 : 64; ga.W = Wp; ga.Y = Y; ga.ldy = 64; ga.bias = bias; ga.M = M; ga.N = 64 * r * r; ga.K = 64;
    ga.H = H; ga.Wd = Wd; ga.r = r; ga.C = 64;
    ga.Y2 = Yd;
    return launch_gemm_nt(dt, M2T_A_PLAIN, M2T_E_BIAS_SHUF, ga, st);
  }
  if (dt == M2T_F32 && g_m2t_f32_fast && r == 2 && Wd % 128 == 0 && M % 128 == 0 && M * 256 < (1LL << 31)) {
    const int ntiles = (int)(M / 128);
    int nblk = std::min(ntiles, 256);
    const int tpb = ceil_div(ntiles, nblk);
    nblk = ceil_div(ntiles, tpb);
    const size_t sh = sizeof(float) * (256 + 2 * 128) * 68;
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)tail_expand_f32_kernel, (int)sh)) return rc__;
    hipLaunchKernelGGL(tail_expand_f32_kernel, dim3(nblk), dim3(512), sh, st, (const float*)X, (const float*)Wp, bias, (float*)Y, (float*)Yd, (int)M, H, Wd,
                       tpb, x_p64 ? 1 : 0);
    M2T_LAUNCH_CHECK();
    return 0;
  }
  if (dt == M2T_F32) return launch_tail_expand_t<float>((const float*)X, (const float*)Wp, bias, (float*)Y, (float*)Yd, M, H, Wd, r, x_p64, st);
  return launch_tail_expand_t<bf16_t>((const bf16_t*)X, (const bf16_t*)Wp, bias, (bf16_t*)Y, (bf16_t*)Yd, M, H, Wd, r, x_p64, st);
}

// =======================================================================================
// wgrad_tn: dW[n][k] = sum_m G[m][n] X[m][k]   (+ optional db[n] = sum_m G[m][n])
// grid (N/64, K/64, nslab).  A workgroup sweeps its M range 128 rows at a time: both operand
// tiles are copied ROW-major into LDS with 16-byte vectors (next tile's global loads are in
// flight while the current one is multiplied) and the contraction-major fragments come from
// the transposing LDS read (ds_read_b64_tr_b16; scalar gather in fp32 parity mode).
// Wave w owns output rows n = 16 w .. 16 w + 15 of the 64 x 64 tile.  The bias gradient rides
// along as one extra tile product against an all-ones operand (no second pass over G).
// =======================================================================================
#define WG_BM 128
#define WG_LD 72

template <typename T, int GMODE, int XMODE>
__global__ void __launch_bounds__(256)
wgrad_tn_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ X, int ldx, float* __restrict__ slabs,
                float* __restrict__ bias_slabs, long long M, int N, int K, long long rows_per_slab, ShufGeom sg) {
  __shared__ __attribute__((aligned(16))) T Gs[WG_BM][WG_LD];
  __shared__ __attribute__((aligned(16))) T Xs[WG_BM][WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
  const long long mb = (long long)blockIdx.z * rows_per_slab;
  const long long me = min(M, mb + rows_per_slab);
  const int nkt = min(4, (K - k0 + 15) / 16);          // live 16-column tiles of this k block
  const bool wave_live = (n0 + 16 * wv) < N;
  const bool do_bias = (bias_slabs != nullptr) && (blockIdx.y == 0);
  f32x4 acc[4], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  Frag8<T> ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.set(e, 1.0f);

  // each thread stages 4 + 4 vectors per step: row = idx >> 3, vec = idx & 7
  Frag8<T> rg[4], rx[4];
  auto fetch = [&](long long ms) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;
      const int row = idx >> 3, cv = idx & 7;
      const long long m = ms + row;
      rg[it] = frag_zero<T>();
      rx[it] = frag_zero<T>();
      if (m < me) {
        if (n0 + cv * 8 < N) rg[it] = gemm_load_a<T, GMODE>(G, ldg, m, n0 + cv * 8, sg);
        if (k0 + cv * 8 < K) rx[it] = gemm_load_a<T, XMODE>(X, ldx, m, k0 + cv * 8, sg);
      }
    }
  };
  fetch(mb);
  for (long long ms = mb; ms < me; ms += WG_BM) {
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 256;
      const int row = idx >> 3, cv = idx & 7;
      store8(&Gs[row][cv * 8], rg[it]);
      store8(&Xs[row][cv * 8], rx[it]);
    }
    __syncthreads();
    if (ms + WG_BM < me) fetch(ms + WG_BM);
    if (wave_live) {
#pragma unroll
      for (int ch = 0; ch < WG_BM / 32; ++ch) {
        const Frag8<T> gf = load8_tr(&Gs[32 * ch + 8 * g][16 * wv], &Gs[32 * ch + 8 * g + 4][16 * wv], WG_LD, lane);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if (kt < nkt) {
            const Frag8<T> xf = load8_tr(&Xs[32 * ch + 8 * g][16 * kt], &Xs[32 * ch + 8 * g + 4][16 * kt], WG_LD, lane);
            mma16(acc[kt], gf, xf);
          }
        }
        if (do_bias) mma16(accb, gf, ones);
      }
    }
  }
  if (!wave_live) return;
  float* out = slabs + (long long)blockIdx.z * N * K;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt) {
    const int k = k0 + 16 * kt + lr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * wv + 4 * g + r;
      if (n < N && k < K) out[(long long)n * K + k] = acc[kt][r];
    }
  }
  if (do_bias && lr == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * wv + 4 * g + r;
      if (n < N) bias_slabs[(long long)blockIdx.z * N + n] = accb[r];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fast path of wgrad_tn for the plain / plain case with full tiles (N % 64 == 0, K % 64 == 0, rows per slab and M
// multiples of 128): no operand modes, no bounds checks, addresses advanced by pointer increments, TWO register
// sets of prefetch -- the generic kernel spends half of its issue slots on index arithmetic and mode branches
// (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = 0.48 for 16 MFMAs per wave and stage).
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
wgrad_tn_fast_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ X, int ldx, float* __restrict__ slabs, long long M,
                     int N, int K, long long rows_per_slab) {
  __shared__ __attribute__((aligned(16))) T Gs[WG_BM][WG_LD];
  __shared__ __attribute__((aligned(16))) T Xs[WG_BM][WG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  // 1-D grid, XCD-aware: the tn x tk tiles of one slab read the same rows of G and X -- keep them behind one L2
  const int tn = N / 64, tk = K / 64;
  const int L = xcd_block_index();
  const int slab = L / (tn * tk), rem = L - slab * (tn * tk);
  const int n0 = (rem % tn) * 64, k0 = (rem / tn) * 64;
  const long long mb = (long long)slab * rows_per_slab;
  const int nst = (int)((min(M, mb + rows_per_slab) - mb) / WG_BM);       // M and rows_per_slab are multiples of 128
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // thread's four (row, vector) slots of a stage: row = (tid + 256 it) >> 3, vector = tid & 7
  const T* pg = G + (mb + (tid >> 3)) * ldg + n0 + (tid & 7) * 8;
  const T* px = X + (mb + (tid >> 3)) * ldx + k0 + (tid & 7) * 8;
  const long long sg32 = 32LL * ldg, sx32 = 32LL * ldx;           // 256 threads cover 32 rows per slot
  Frag8<T> rg[2][4], rx[2][4];
  auto fetch = [&](int set, int st) {
    const T* qg = pg + (long long)st * WG_BM * ldg;
    const T* qx = px + (long long)st * WG_BM * ldx;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rg[set][it] = load8(qg + it * sg32);
      rx[set][it] = load8(qx + it * sx32);
    }
  };
  fetch(0, 0);
  if (nst > 1) fetch(1, 1);
  for (int sb = 0; sb < nst; sb += 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int st = sb + j;
      if (st < nst) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          store8(&Gs[(tid >> 3) + 32 * it][(tid & 7) * 8], rg[j][it]);
          store8(&Xs[(tid >> 3) + 32 * it][(tid & 7) * 8], rx[j][it]);
        }
        __syncthreads();
        if (st + 2 < nst) fetch(j, st + 2);
#pragma unroll
        for (int ch = 0; ch < WG_BM / 32; ++ch) {
          const Frag8<T> gf = load8_tr(&Gs[32 * ch + 8 * g][16 * wv], &Gs[32 * ch + 8 * g + 4][16 * wv], WG_LD, lane);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            mma16(acc[kt], gf, load8_tr(&Xs[32 * ch + 8 * g][16 * kt], &Xs[32 * ch + 8 * g + 4][16 * kt], WG_LD, lane));
        }
      }
    }
  }
  float* out = slabs + (long long)slab * N * K;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(long long)(n0 + 16 * wv + 4 * g + r) * K + k0 + 16 * kt + lr] = acc[kt][r];
}

// ---------------------------------------------------------------------------------------
// 128 x 128 output tiles, 8 waves (wave (wn, wk) = 32 rows n x 64 columns k): the 64 x 64 kernel above re-reads G K/64 times
// and X N/64 times from L2 (C = 256 qkv: 200 MB of L2 traffic per launch for 33.5 MB of operands, 2.5 transposing LDS
// reads per MFMA) and is bound by exactly that; this one halves both and needs 1.5 LDS reads per MFMA.  Same stage
// structure (128 rows, two register sets of prefetch); N % 128 == 0, K % 128 == 0.
// ---------------------------------------------------------------------------------------
#define WGB_LD 136
template <typename T>
__global__ void __launch_bounds__(512)
wgrad_tn_big_kernel(const T* __restrict__ G, int ldg, const T* __restrict__ X, int ldx, float* __restrict__ slabs, long long M,
                    int N, int K, long long rows_per_slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T(*Gs)[WGB_LD] = reinterpret_cast<T(*)[WGB_LD]>(smem);
  T(*Xs)[WGB_LD] = reinterpret_cast<T(*)[WGB_LD]>(smem + sizeof(T) * WG_BM * WGB_LD);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const int wn = wv & 3, wk = wv >> 2;
  const int tn = N / 128, tk = K / 128;
  const int L = xcd_block_index();
  const int slab = L / (tn * tk), rem = L - slab * (tn * tk);
  const int n0 = (rem % tn) * 128, k0 = (rem / tn) * 128;
  const long long mb = (long long)slab * rows_per_slab;
  const int nst = (int)((min(M, mb + rows_per_slab) - mb) / WG_BM);
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // thread's four (row, vector) slots of a stage: row = (tid >> 4) + 32 it, vector = tid & 15 (16 x 8 = 128 columns)
  const T* pg = G + (mb + (tid >> 4)) * ldg + n0 + (tid & 15) * 8;
  const T* px = X + (mb + (tid >> 4)) * ldx + k0 + (tid & 15) * 8;
  const long long sg32 = 32LL * ldg, sx32 = 32LL * ldx;
  Frag8<T> rg[2][4], rx[2][4];
  auto fetch = [&](int set, int st) {
    const T* qg = pg + (long long)st * WG_BM * ldg;
    const T* qx = px + (long long)st * WG_BM * ldx;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rg[set][it] = load8(qg + it * sg32);
      rx[set][it] = load8(qx + it * sx32);
    }
  };
  fetch(0, 0);
  if (nst > 1) fetch(1, 1);
  for (int sb = 0; sb < nst; sb += 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int st = sb + j;
      if (st < nst) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          store8(&Gs[(tid >> 4) + 32 * it][(tid & 15) * 8], rg[j][it]);
          store8(&Xs[(tid >> 4) + 32 * it][(tid & 15) * 8], rx[j][it]);
        }
        __syncthreads();
        if (st + 2 < nst) fetch(j, st + 2);
#pragma unroll
        for (int ch = 0; ch < WG_BM / 32; ++ch) {
          Frag8<T> gf[2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
            gf[i] = load8_tr(&Gs[32 * ch + 8 * g][32 * wn + 16 * i], &Gs[32 * ch + 8 * g + 4][32 * wn + 16 * i], WGB_LD, lane);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            const Frag8<T> xf = load8_tr(&Xs[32 * ch + 8 * g][64 * wk + 16 * kt], &Xs[32 * ch + 8 * g + 4][64 * wk + 16 * kt], WGB_LD, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) mma16(acc[i][kt], gf[i], xf);
          }
        }
      }
    }
  }
  float* out = slabs + (long long)slab * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(long long)(n0 + 32 * wn + 16 * i + 4 * g + r) * K + k0 + 64 * wk + 16 * kt + lr] = acc[i][kt][r];
}
static bool wgrad_big_ok(long long M, int N, int K) { return N % 128 == 0 && K % 128 == 0 && M % WG_BM == 0 && N * (long long)K >= 128 * 256; }
static int wgrad_big_slabs(long long M, int N, int K, int target = 256) {
  // ~256 workgroups of 512 threads (two fit on a CU): enough to keep the MFMAs of the used CUs busy, few enough that the
  // slab partials (slabs x N x K fp32, written here and read once more by the deferred reduction) stay below the operands
  const long long want = std::max<long long>(1, target / ((N / 128) * (K / 128)));
  return (int)std::min<long long>(M2T_MAX_SLABS, std::min<long long>(want, ceil_divll(M, WG_BM)));
}

// ---------------------------------------------------------------------------------------
// fp32 weight gradient on v_mfma_f32_32x32x2_f32 (see gemm_nt_f32_kernel): 64 x 64 output tile, 2 x 2 waves of one 32 x 32 block,
// 32-row stages of [G 64 | X 64] rows (row stride 160 floats = 32 banks mod 64: the two rows a step reads cover all 64 banks),
// double-buffered, one barrier per stage.  The contraction runs over the ROW index, so both operands are ds_read_b32 down a column:
// step s of a stage takes rows 2 s + h.  Same slab contract as wgrad_tn_kernel (slabs of [N][K], rows per slab a multiple of 32).
// ---------------------------------------------------------------------------------------
// UNSHUF: G is the pixel-shuffled tensor [B][H r][W r][64], row m = (b, h, w), columns n = sub-pixel * 64 + c (a 64-column tile is one
// sub-pixel; the tail expansions' weight gradients, gemm_load_a's mapping); XP64: X is a P64 feature map; BIAS: column sums of G
// (the bias gradient) summed by the staging threads -- a thread always stages the same eight columns -- written by the k0 = 0 tiles.
template <bool UNSHUF, bool XP64, bool BIAS>
__global__ void __launch_bounds__(256, 2)
wgrad_tn_f32_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx, float* __restrict__ slabs,
                    float* __restrict__ bias_slabs, long long M, int N, int K, long long rows_per_slab, ShufGeom sg) {
  constexpr int BR = 32, LD = 160;
  __shared__ __attribute__((aligned(16))) float S[2][BR][LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5, wn = wv & 1, wk = wv >> 1;
  const int tn = N / 64, tk = K / 64;
  const int L = xcd_block_index();
  const int slab = L / (tn * tk), rem = L - slab * (tn * tk);
  const int n0 = (rem % tn) * 64, k0 = (rem / tn) * 64;
  const long long mb = (long long)slab * rows_per_slab, me = min(M, mb + rows_per_slab);
  const int nst = (int)((me - mb + BR - 1) / BR);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // staging slot: row tid >> 3 of the stage, float4 columns 4 v and 4 v + 32 of G and of X
  const int srow = tid >> 3, sv = (tid & 7) * 4;
  const float* pg = G + (mb + srow) * ldg + n0 + sv;
  const float* px = X + (mb + srow) * ldx + k0 + sv;
  const int sub = n0 >> 6, si = UNSHUF ? sub / sg.r : 0, sj = UNSHUF ? sub - si * sg.r : 0;
  f32x4 rg[4], bsum[2];
  bsum[0] = bsum[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int st) {
    const long long m = mb + (long long)st * BR + srow;
#pragma unroll
    for (int j = 0; j < 4; ++j) rg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m < me) {
      const float* qg = pg + (long long)st * BR * ldg;
      if constexpr (UNSHUF) {
        const int mi = (int)m;                              // (the launcher takes this path for M < 2^31 only: 64-bit division is a software loop)
        const int w = mi % sg.W, q = mi / sg.W;
        const int h = q % sg.H, b = q / sg.H;
        const long long pix = ((long long)b * sg.H * sg.r + (h * sg.r + si)) * ((long long)sg.W * sg.r) + (w * sg.r + sj);
        qg = G + pix * 64 + sv;
      }
      rg[0] = *reinterpret_cast<const f32x4*>(qg);
      rg[1] = *reinterpret_cast<const f32x4*>(qg + 32);
      if constexpr (XP64) {
        rg[2] = *reinterpret_cast<const f32x4*>(X + p64(sg.npix, m, k0 + sv));
        rg[3] = *reinterpret_cast<const f32x4*>(X + p64(sg.npix, m, k0 + sv + 32));
      } else {
        const float* qx = px + (long long)st * BR * ldx;
        rg[2] = *reinterpret_cast<const f32x4*>(qx);
        rg[3] = *reinterpret_cast<const f32x4*>(qx + 32);
      }
    }
  };
  if (nst > 0) fetch(0);
  int buf = 0;
  for (int st = 0; st < nst; ++st, buf ^= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(&S[buf][srow][sv + 32 * j]) = rg[j];
    if constexpr (BIAS) { bsum[0] += rg[0]; bsum[1] += rg[1]; }
    __syncthreads();
    if (st + 1 < nst) fetch(st + 1);
    const float* lg = &S[buf][lh][32 * wn + li];
    const float* sx = &S[buf][lh][64 + 32 * wk + li];
    float gf[2][4], xf[2][4];
#pragma unroll
    for (int s = 0; s < 4; ++s) { gf[0][s] = lg[2 * s * LD]; xf[0][s] = sx[2 * s * LD]; }
#pragma unroll
    for (int q = 0; q < BR / 8; ++q) {
      if (q + 1 < BR / 8) {
#pragma unroll
        for (int s = 0; s < 4; ++s) { gf[(q + 1) & 1][s] = lg[(8 * (q + 1) + 2 * s) * LD]; xf[(q + 1) & 1][s] = sx[(8 * (q + 1) + 2 * s) * LD]; }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) mfma32f(acc, gf[q & 1][s], xf[q & 1][s]);
    }
  }
  // D[row n = 8 (r >> 2) + 4 h + (r & 3)][col k = li]: half a wave writes 128 contiguous bytes of one row of the slab
  float* out = slabs + (long long)slab * N * K + (long long)(n0 + 32 * wn + 4 * lh) * K + k0 + 32 * wk + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) out[(long long)(8 * (r >> 2) + (r & 3)) * K] = acc[r];
  if constexpr (BIAS) {
    if (bias_slabs != nullptr && k0 == 0) {          // (uniform per workgroup)
      __syncthreads();
      float(*Rb)[64] = reinterpret_cast<float(*)[64]>(&S[0][0][0]);          // [32 staging rows][64 columns]
      *reinterpret_cast<f32x4*>(&Rb[srow][sv]) = bsum[0];
      *reinterpret_cast<f32x4*>(&Rb[srow][sv + 32]) = bsum[1];
      __syncthreads();
      if (tid < 64) {
        float sum = 0.f;
        for (int r = 0; r < 32; ++r) sum += Rb[r][tid];
        bias_slabs[(long long)slab * N + n0 + tid] = sum;
      }
    }
  }
}

int wgrad_slab_count(long long M, int N, int K) {
  const int tn = ceil_div(N, 64), tk = ceil_div(K, 64);
  const long long want = std::max<long long>(1, 512 / (tn * tk));
  int n = (int)std::min<long long>(M2T_MAX_SLABS, std::min<long long>(want, ceil_divll(M, WG_BM)));
  if (wgrad_big_ok(M, N, K)) n = std::max(n, wgrad_big_slabs(M, N, K, 512));     // upper bound over the selectable targets
  return n;
}

template <typename T>
static int launch_wgrad_tn_t(const m2t_wgrad_args& a, int* nslab_out, hipStream_t st) {
  if (a.K % 8 || a.N % 8) return m2t_set_error(-2, "wgrad_tn: N,K must be multiples of 8");
  const int tn = ceil_div(a.N, 64), tk = ceil_div(a.K, 64);
  // enough slabs for ~2 workgroups per CU (more slabs only inflate the deferred reduction); rows per slab a multiple of the 128-row step
  if constexpr (sizeof(T) == 2) {
  if (a.big_tiles && a.gmode == M2T_A_PLAIN && a.xmode == M2T_A_PLAIN && !a.bias_slabs && a.ldg > 0 && a.ldx > 0 &&
      wgrad_big_ok(a.M, a.N, a.K)) {
    int nb = wgrad_big_slabs(a.M, a.N, a.K, std::min(std::max(a.big_tiles, 64), 512));
    const long long rpb = ceil_divll(ceil_divll(a.M, nb), WG_BM) * WG_BM;
    nb = (int)ceil_divll(a.M, rpb);
    const size_t sh = sizeof(T) * 2 * WG_BM * WGB_LD;
    if (int rc__ = m2t_ensure_dynamic_lds((const void*)wgrad_tn_big_kernel<T>, (int)sh)) return rc__;
    hipLaunchKernelGGL((wgrad_tn_big_kernel<T>), dim3((a.N / 128) * (a.K / 128) * nb), dim3(512), sh, st, (const T*)a.G, a.ldg, (const T*)a.X,
                       a.ldx, a.slabs, a.M, a.N, a.K, rpb);
    M2T_LAUNCH_CHECK();
    *nslab_out = nb;
    return 0;
  }
  }
  const long long want64 = std::max<long long>(1, 512 / (tn * tk));
  int nslab = (int)std::min<long long>(M2T_MAX_SLABS, std::min<long long>(want64, ceil_divll(a.M, WG_BM)));
  if constexpr (sizeof(T) == 4) {
    const bool plain = a.gmode == M2T_A_PLAIN && !a.bias_slabs && a.ldg > 0 && a.ldg % 4 == 0 && a.ldx > 0 && a.ldx % 4 == 0;
    const bool unshuf = a.gmode == M2T_A_UNSHUF && a.C == 64 && a.N == 64 * a.r * a.r && a.bias_slabs && a.M < (1LL << 31) &&
                        ((a.ldx > 0 && a.ldx % 4 == 0) || a.ldx == M2T_LD_P64);
    if (g_m2t_f32_fast && a.xmode == M2T_A_PLAIN && (plain || unshuf) && a.N % 64 == 0 && a.K % 64 == 0) {
      // same slab bound as below (wgrad_slab_count); rows per slab a multiple of the 32-row stage so the slabs balance
      const long long rps32 = ceil_divll(ceil_divll(a.M, nslab), 32) * 32;
      const int ns32 = (int)ceil_divll(a.M, rps32);
      ShufGeom sg{a.H, a.Wd, a.r, a.C, a.halo_win, a.M};
#define GO32(U_, P_, B_)                                                                                                          \
  hipLaunchKernelGGL((wgrad_tn_f32_kernel<U_, P_, B_>), dim3(tn * tk * ns32), dim3(256), 0, st, (const float*)a.G, a.ldg, (const float*)a.X, \
                     a.ldx, a.slabs, a.bias_slabs, a.M, a.N, a.K, rps32, sg)
      if (plain) GO32(false, false, false);
      else if (a.ldx == M2T_LD_P64) GO32(true, true, true);
      else GO32(true, false, true);
#undef GO32
      M2T_LAUNCH_CHECK();
      *nslab_out = ns32;
      return 0;
    }
  }
  // whole slabs per XCD: the tn x tk tiles of a slab share its rows of G and X through ONE L2 (xcd_block_index gives every XCD a
  // contiguous eighth of the grid); 10 slabs of 48 tiles straddled the XCD runs and re-fetched 1.5x the operand bytes from HBM
  if (nslab > 8) nslab = nslab / 8 * 8;
  long long rps = ceil_divll(ceil_divll(a.M, nslab), WG_BM) * WG_BM;
  nslab = (int)ceil_divll(a.M, rps);
  if (sizeof(T) == 2 && a.gmode == M2T_A_PLAIN && a.xmode == M2T_A_PLAIN && !a.bias_slabs && a.ldg > 0 && a.ldx > 0 &&
      a.N % 64 == 0 && a.K % 64 == 0 && a.M % WG_BM == 0) {
    hipLaunchKernelGGL((wgrad_tn_fast_kernel<T>), dim3(tn * tk * nslab), dim3(256), 0, st, (const T*)a.G, a.ldg, (const T*)a.X, a.ldx,
                       a.slabs, a.M, a.N, a.K, rps);
    M2T_LAUNCH_CHECK();
    *nslab_out = nslab;
    return 0;
  }
  ShufGeom sg{a.H, a.Wd, a.r, a.C, a.halo_win, a.M};
  dim3 grid(tn, tk, nslab);
#define GO(GM, XM)                                                                                               \
  hipLaunchKernelGGL((wgrad_tn_kernel<T, GM, XM>), grid, dim3(256), 0, st, (const T*)a.G, a.ldg, (const T*)a.X, \
                     a.ldx, a.slabs, a.bias_slabs, a.M, a.N, a.K, rps, sg)
  if (a.gmode == M2T_A_PLAIN && a.xmode == M2T_A_PLAIN) GO(M2T_A_PLAIN, M2T_A_PLAIN);
  else if (a.gmode == M2T_A_UNSHUF && a.xmode == M2T_A_PLAIN) GO(M2T_A_UNSHUF, M2T_A_PLAIN);
  else return m2t_set_error(-2, "wgrad_tn: unsupported operand modes");
#undef GO
  M2T_LAUNCH_CHECK();
  *nslab_out = nslab;
  return 0;
}
int launch_wgrad_tn(int dt, const m2t_wgrad_args& a, int* nslab_out, hipStream_t st) {
  if (dt == M2T_F32) return launch_wgrad_tn_t<float>(a, nslab_out, st);
  return launch_wgrad_tn_t<bf16_t>(a, nslab_out, st);
}
