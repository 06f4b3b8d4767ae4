// m2t_kernels.h -- internal launcher prototypes shared by the .hip files and the C-ABI layer.
#pragma once
#include <algorithm>
#include "m2t_common.h"
#include <hip/hip_ext.h>

#define M2T_NORM_SPLIT 32     // pixel splits per image in the InstanceNorm reductions
#define M2T_LOSS_BLOCKS 1024  // partial sums of the L1 loss
#define M2T_MAX_SLABS 1024    // split-M slabs of a weight-gradient GEMM
#define M2T_UNSUPPORTED (-1000)
#define M2T_PACK_CHUNK 2048   // output elements one workgroup of the weight-packing kernel converts

enum m2t_pack_kind {
  M2T_PACK_COPY = 0, M2T_PACK_TRANSPOSE = 1, M2T_PACK_CONV3 = 2, M2T_PACK_CONV3_T = 3,
  M2T_PACK_SHUF_ROWS = 4, M2T_PACK_SHUF_ROWS_T = 5,
  M2T_PACK_FRAG16 = 6,    // src [N = d0][K = d1] -> MFMA A-operand fragment order [N/16][K/32][64 lanes][8]: element j of lane l of
                          // fragment (tile, ks) is src[16 tile + (l & 15)][32 ks + 8 (l >> 4) + j]; one wave load = 1 KB contiguous
  M2T_PACK_FRAG16_T = 7,  // the same fragments of the TRANSPOSE: src is [K = d1][N = d0] (element = src[32 ks + 8 (l >> 4) + j][16 tile + (l & 15)])
  M2T_PACK_CONV3_ROWS = 8,    // 64 -> 64 3x3 conv weight as the register-resident A-fragments of conv3x3_c64_rows_kernel:
  M2T_PACK_CONV3_ROWS_T = 9   // [tap][channel half][k-chunk][tile][64 lanes][8], one contiguous 1 KB per wave load (_T: data gradient)
};
struct m2t_red_desc {      // one deferred slab reduction: grads[dst_off + perm(e)] = sum_s arena[src_off + s*n + e]
  long long src_off, dst_off, n;
  int ns, perm, p0, p1, p2, pad_;
};
struct m2t_pack_desc {
  long long src_off, dst_off, n;
  int kind, d0, d1, d2;
};

// ---- in-library kernel timing (HIP events on the launch stream; off by default) -----------
enum m2t_prof_cat {
  M2T_PROF_ATTN_FWD_16 = 0, M2T_PROF_ATTN_FWD_64, M2T_PROF_ATTN_FWD_256,
  M2T_PROF_ATTN_BWD_16, M2T_PROF_ATTN_BWD_64, M2T_PROF_ATTN_BWD_256,
  M2T_PROF_CONV3_FWD, M2T_PROF_CONV3_DGRAD, M2T_PROF_CONV3_WGRAD,
  M2T_PROF_GEMM_QKV, M2T_PROF_GEMM_QKV_DGRAD, M2T_PROF_WGRAD_QKV,
  M2T_PROF_TAIL_GEMM, M2T_PROF_TAIL_WGRAD, M2T_PROF_FINAL_FWD, M2T_PROF_FINAL_DGRAD, M2T_PROF_FINAL_WGRAD,
  M2T_PROF_ATTN_FUSED_64, M2T_PROF_ATTN_FUSED_256,      // fused qkv projection + window attention forward (k_attn_fused.hip)
  M2T_PROF_TAIL_FWD_FUSED,                              // tail.3 expansion + PixelShuffle + GELU + tail conv (k_tail_fwd.hip)
  M2T_PROF_ATTN_FUSED_16,                               // InstanceNorm apply + qkv projection + window attention, C = 16 (k_attn_c16.hip)
  M2T_PROF_CONV3_BWD,                                   // conv 64 -> 64 data + weight gradient in one pass (k_conv.hip)
  M2T_PROF_NCAT
};
void m2t_prof_begin(int cat, hipStream_t st);
void m2t_prof_end(int cat, hipStream_t st);
// Single-kernel categories (attention, conv3x3, final conv): the scope's two events ride on the kernel dispatch itself
// (hipExtLaunchKernelGGL start / stop events = the dispatch's own begin / end timestamps, the quantity rocprofv3
// reports) instead of bracketing it with marker packets, which add the ~5 us launch gap to every sample.
#define M2T_PROF_DISPATCH_CATS ((1ull << M2T_PROF_GEMM_QKV) - 1ull | (1ull << M2T_PROF_FINAL_FWD) | (1ull << M2T_PROF_FINAL_DGRAD) | (1ull << M2T_PROF_FINAL_WGRAD) | \
                                (1ull << M2T_PROF_ATTN_FUSED_64) | (1ull << M2T_PROF_ATTN_FUSED_256) | (1ull << M2T_PROF_TAIL_FWD_FUSED) | (1ull << M2T_PROF_ATTN_FUSED_16) | \
                                (1ull << M2T_PROF_CONV3_BWD))
bool m2t_prof_take(hipEvent_t* a, hipEvent_t* b);
// an armed fork event (m2t_backward, option "fork_on_kernel") rides on the dispatch as its stop event instead of being recorded by
// a marker packet behind it; nullptr when none is armed or a timing pair already took the dispatch
hipEvent_t m2t_fork_take();
#define M2T_LAUNCH_TIMED(kernel, grid, block, sh, st, ...)                                                       \
  do {                                                                                                           \
    hipEvent_t ea__, eb__;                                                                                       \
    if (m2t_prof_take(&ea__, &eb__)) hipExtLaunchKernelGGL(kernel, grid, block, sh, st, ea__, eb__, 0, __VA_ARGS__); \
    else if (hipEvent_t ef__ = m2t_fork_take()) hipExtLaunchKernelGGL(kernel, grid, block, sh, st, nullptr, ef__, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, sh, st, __VA_ARGS__);                                           \
  } while (0)
struct M2TProfScope {
  int cat; hipStream_t st;
  M2TProfScope(int c, hipStream_t s) : cat(c), st(s) { m2t_prof_begin(cat, st); }
  ~M2TProfScope() { m2t_prof_end(cat, st); }
};

// ---- k_pointwise.hip --------------------------------------------------------------------
int launch_dwt(int dt, int L, const void* src, int lds_, int c0, void* dst, int ldd, int d0, int B, int H, int W,
               int C, bool inverse, hipStream_t st);
int launch_pixel_shuffle_nchw(const float* in, float* out, int B, int C, int H, int W, int r, int inverse, hipStream_t st);
int launch_instnorm_finalize(const float* part, float* mean, float* rstd, int B, int nsplit, hipStream_t st);
int launch_instnorm_stats(int dt, const void* x, float* mean, float* rstd, float* part, int B, int P, hipStream_t st);
int launch_branch_prep(int dt, int L, const void* x, const float* mean, const float* rstd, const void* xc, int k,
                       void* xin, void* d, int B, int H, int W, hipStream_t st);
int launch_branch_post(int dt, int L, const void* a, const void* xin, void* xc, int k, int B, int H, int W, hipStream_t st);
int launch_branch_post_bwd(int dt, int L, const void* gxc, int k, void* ga, int B, int H, int W, hipStream_t st);
// gdwin != nullptr: ring rows [window][36][16 * 4^L] of the fused projection data gradient, added to the border pixels on load
int launch_branch_prep_bwd(int dt, int L, const void* gd, void* gxc, void* gn, int k, int B, int H, int W, hipStream_t st,
                           const void* gdwin = nullptr);
// part0 != nullptr (bf16): the first reduction stage already ran inside launch_c16_dgrad_prep (its norm arguments): skip it
int launch_instnorm_bwd(int dt, const void* gn, const void* x, const float* mean, const float* rstd, const void* gres,
                        void* gx, float* part, float* s, int B, int P, hipStream_t st, const float* part0 = nullptr, int tiles0 = 0,
                        const void* gres2 = nullptr);       // gres2: a second residual gradient added in the same pass (block 0: g(Y) of `res + x`)
int launch_add(int dt, const void* a, const void* b, void* o, long long n, hipStream_t st);
int launch_colsum(int dt, const void* a, int lda, long long M, int N, float* part, int max_part_blocks, float* out,
                  int accumulate, hipStream_t st, int unshuf = 0, int gH = 0, int gW = 0, int gr = 1, int gC = 64,
                  int* nblk_out = nullptr);   // nblk_out != null: write the partials only and report their count
int launch_multi_reduce(const float* arena, float* grads, const m2t_red_desc* descs, int ndesc, hipStream_t st);
int launch_clamp_l1(const float* pre, const float* hr, float* sr, float* gpre, float* part, float* loss, int B, int Hp,
                    int Wp, int Hs, int Ws, float R, float loss_scale, float gscale, hipStream_t st);
// loss = loss_scale * sum(part[0 .. n)) in a fixed order (n <= M2T_LOSS_BLOCKS)
int launch_loss_finish(const float* part, int n, float loss_scale, float* loss, hipStream_t st);
int launch_adam(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                int step, float gscale, hipStream_t st);
// blocks: device table int2[nblocks] = (descriptor index, chunk of M2T_PACK_CHUNK output elements)
int launch_pack(int dt, const float* master, void* packed, const m2t_pack_desc* descs, const void* blocks, int nblocks, hipStream_t st);
int launch_layout(int dt, const float* nchw, void* nhwc, float* nchw_out, int B, int C, int HW, int inverse, hipStream_t st);

// ---- k_gemm.hip -------------------------------------------------------------------------
// Y[M][N] = A[M][K] * W[N][K]^T  with A-side and epilogue variants
// leading dimension sentinel: the operand / output is a P64 feature map ([4][M][16], m2t_common.h) with M rows
#define M2T_LD_P64 (-64)
enum m2t_gemm_a { M2T_A_PLAIN = 0, M2T_A_GELU = 1, M2T_A_UNSHUF = 2 };
enum m2t_gemm_epi { M2T_E_PLAIN = 0, M2T_E_BIAS = 1, M2T_E_BIAS_SHUF = 2, M2T_E_GELU_GRAD = 3, M2T_E_BIAS_GELU = 4, M2T_E_BIAS_RESID = 5,
                    M2T_E_BIAS_RELU = 6 };   // (+ bias, ReLU: Mlp of util/rlutrans.py:22-23)
struct m2t_gemm_args {
  const void* A; int lda;       // A rows (or, UNSHUF: the [B][H*r][W*r][C] tensor)
  const void* W;                // [N][K] element type T
  void* Y; int ldy;             // output (or, SHUF: the [B][H*r][W*r][C] tensor)
  const float* bias;            // [N] fp32 (E_BIAS*)
  const void* aux; int ldaux;   // E_GELU_GRAD: stored derivative gelu'(t), same shape as Y; E_BIAS_RESID: residual
  void* Y2 = nullptr;           // E_BIAS_SHUF: second output, the derivative gelu'(t) (Y receives gelu(t))
  long long M; int N, K;
  int H, Wd, r, C;              // shuffle geometry: rows m = (b, h, w) over [B][H][Wd]; C channels after shuffle
  const void* halo_win = nullptr;   // M2T_A_HALO: per-window dK|dV scratch (H, Wd = branch grid, C = branch channels)
};
int launch_gemm_nt(int dt, int amode, int emode, const m2t_gemm_args& a, hipStream_t st);
// fp32 (parity mode): 1 = the v_mfma_f32_32x32x2_f32 kernels of round 5 for the plain GEMMs, the qkv weight gradients and the 3x3 conv,
// 0 = the 16x16x4 kernels of rounds 1-4 (option fp32_fast of the plan being run; set by m2t_api.hip on entry)
extern thread_local int g_m2t_f32_fast;
// upsampler 1x1 conv + bias + pixel-shuffle scatter + GELU, K = 64, N = 64 r^2; X rows over [B][H][Wd];
// Y = gelu(t), Yd = gelu'(t) (both [B][H r][Wd r][64])
int launch_tail_expand(int dt, const void* X, const void* Wp, const float* bias, void* Y, void* Yd, long long M, int H, int Wd,
                       int r, bool x_p64, hipStream_t st);   // x_p64: X is a P64 feature map (else rows of 64)
// dW[N][K] (fp32 slabs) = sum_m G[m][N]^T X[m][K];  G/X side variants as above
struct m2t_wgrad_args {
  const void* G; int ldg; int gmode;   // M2T_A_PLAIN or M2T_A_UNSHUF
  const void* X; int ldx; int xmode;   // M2T_A_PLAIN or M2T_A_GELU
  float* slabs;                        // [nslab][N][K]
  float* bias_slabs;                   // optional [nslab][N]: column sums of G (bias gradient)
  long long M; int N, K;
  int H, Wd, r, C;
  const void* halo_win = nullptr;
  int big_tiles = 0;                   // > 0: bf16 plain / plain, N % 128 == 0, K % 128 == 0: 128 x 128 output tiles (wgrad_tn_big_kernel),
                                       // value = target number of workgroups (tiles x slabs), 64 .. 512
};
int launch_wgrad_tn(int dt, const m2t_wgrad_args& a, int* nslab_out, hipStream_t st);
int wgrad_slab_count(long long M, int N, int K);   // upper bound of the slabs launch_wgrad_tn will write

// ---- k_conv.hip -------------------------------------------------------------------------
int launch_head_conv_fwd(int dt, const float* x, const float* w, const float* b, void* out, int B, int H0, int W0,
                         int H, int W, hipStream_t st);
int launch_head_im2col(int dt, const float* x, void* cols, int B, int H0, int W0, int H, int W, hipStream_t st);   // cols [B*H*W][32] (T)
// 64->64 3x3, zero padding.  wp: packed [9][64 out][64 in] (T). y = conv(x) + bias + res1 + res2 (each optional)
int launch_conv3x3_c64(int dt, const void* x, const void* wp, const float* bias, const void* res1, const void* res2,
                       void* y, int B, int H, int W, hipStream_t st,
                       const void* wrows = nullptr,       // bf16: the same weight in M2T_PACK_CONV3_ROWS(_T) order and ...
                       const void* zero_page = nullptr,   // ... >= 64 zero bytes in device memory -> the row-streaming kernel
                       int variant = 0,                   // 0: row-streaming, DMA depth 2 (default); 1: the tile kernel
                                                          // (conv3x3_c64_pipe_kernel); 3: depth 3; 4: depth 2 + pipelined epilogue
                       float* stat_part = nullptr);       // one residual, row-streaming: InstanceNorm partials of y, [B][n][64][3] with
                                                          // n = conv3x3_c64_stat_partials(...) (0: this shape / variant cannot)
int conv3x3_c64_stat_partials(int dt, int B, int H, int W, int variant);
int launch_conv3x3_c64_wgrad(int dt, const void* x, const void* gy, float* slabs, float* bias_slabs, int* nslab, int B, int H,
                             int W, hipStream_t st);   // bias_slabs [nslab][64]: column sums of gy
// bf16: data gradient + weight / bias gradient partials in one pass over gy (conv3x3_c64_bwd_rows_kernel); M2T_UNSUPPORTED for
// shapes without a strip decomposition.  wrows_t: M2T_PACK_CONV3_ROWS_T; slabs [<= 256][9][64][64], bias_slabs [<= 256][64]
bool conv3x3_c64_bwd_fusable(int B, int H, int W);
int launch_conv3x3_c64_bwd_fused(const void* gy, const void* x, const void* wrows_t, void* gx, float* slabs, float* bias_slabs, int* nslab,
                                 const void* zero_page, int B, int H, int W, hipStream_t st);
// tail conv 64->3, reflect padding, input = the stored activation gelu(t); output NCHW fp32 [B][3][H][W]
int launch_final_conv_fwd(int dt, const void* tpre, const float* w, float* out, int B, int H, int W, hipStream_t st);
int launch_final_conv_dgrad(int dt, const float* gout, const float* w, const void* tpre, void* gtpre, int B, int H, int W,
                            hipStream_t st);
int launch_final_conv_wgrad(int dt, const float* gout, const void* tpre, float* slabs, int* nslab, int B, int H, int W,
                            hipStream_t st);

// ---- k_branch.hip -----------------------------------------------------------------------

// ---- k_tail_fwd.hip ----------------------------------------------------------------------
// x4 tail, bf16: tail.3 expansion + PixelShuffle + GELU + tail conv in one pass; gelu(t2) never reaches HBM.
int launch_tail_fwd_fused(const void* a1, const void* w3p, const float* b3, const float* wf, float* out, int B, int H, int W,
                          hipStream_t st);

// ---- k_tail_stream.hip -------------------------------------------------------------------
// bf16: expansion 1x1 (64 -> 64 R^2) + PixelShuffle(R) + GELU + tail conv as ONE row-streaming kernel; R = 2 (x4's tail.3 stage on the
// NHWC mid-resolution map, or x2's tail.0 stage on the P64 body output) or 3 (x3).  seg_rows <= 2: default segment length.
int launch_tail_fwd_stream(const void* a, int a_is_p64, const void* wp, const float* bias, const float* wf, float* out, int B, int Hi,
                           int Wi, int R, int seg_rows, hipStream_t st);

// ---- k_tail_bwd_stream.hip ---------------------------------------------------------------
// bf16: the recomputing backward of the tail's last stage as a row-streaming kernel (round 4).  tail0 = 1 (R = 2 / 3): x2's / x3's whole
// tail on the P64 body output; tail0 = 0 (R = 2): x4's tail.3 stage (kept for A/B: slower than the tile kernel).  Hi, Wi: size of the
// expansion's INPUT map.  Slabs: tail_bwd_stream_blocks() of each kind.
int tail_bwd_stream_blocks(int B, int Hi, int Wi, int R);
int launch_tail_bwd_stream(const float* gout, const float* wf, const void* a, const void* d1, const void* w3t, const float* b3, void* ga,
                           float* slab_wf, float* slab_w3, float* slab_b3, int* nslab_out, int B, int Hi, int Wi, int R, int tail0,
                           hipStream_t st);

// ---- k_tail_bwd.hip ----------------------------------------------------------------------
// x4 tail, bf16: tail conv data + weight gradient, GELU backward, tail.3 data + weight + bias gradient in one pass
// over the stored activation / derivative tensors (g(t2) never reaches HBM).  Slabs: wf [n][32][64], w3 [n][256][64],
// b3 [n][256]; n = *nslab_out <= tail_bwd_fused_blocks().
int tail_bwd_fused_blocks(int B, int H, int W);
// act == der == nullptr: gelu(t2) / gelu'(t2) are recomputed per tile from a1, w3t and the tail.3 bias b3 (torch order)
// l1_pre != nullptr (recomputing variant only): the clamp + L1 seed of launch_clamp_l1 is taken inside the kernel from the pre-clamp
// output l1_pre [B][3][H][W] and the target l1_hr [B][3][Hs][Ws] (gout is then unused); l1_part [tail_bwd_fused_blocks] receives the
// partial sums of |clamp(pre) - hr| for launch_loss_finish
int launch_tail_bwd_fused(const float* gout, const float* wf, const void* act, const void* der, const void* a1, const void* d1,
                          const void* w3t, const float* b3, void* gt1, float* slab_wf, float* slab_w3, float* slab_b3,
                          int* nslab_out, int B, int H, int W, hipStream_t st, const float* l1_pre = nullptr, const float* l1_hr = nullptr,
                          float* l1_part = nullptr, int Hs = 0, int Ws = 0, float R = 0.f, float gscale = 0.f, int variant = 32);
// variant (recomputing form): 32 = the round-6 kernel on v_mfma_f32_32x32x16_bf16 (conflict-free LDS operand reads, g(t2) formed in
// registers; results agree with the older kernel to fp32 summation order), 16 = the 16x16x32 kernel of rounds 2-5 (what the stored
// form always runs)

// ---- k_attn.hip -------------------------------------------------------------------------
// qkv [B][h][w][3C] (q | k | v), rel_h/rel_w fp32 [10][C/2];  out rows at ldo (+ optional residual rows at ldr)
// post_levels = 1, 2: out/res are the FULL-RES xc (ld ldo, channel offset oc0) / xin (ld ldr) tensors and the
// epilogue writes xc = IWT^levels(attention) + xin
int launch_window_attn_fwd(int dt, const void* qkv, const float* rel_h, const float* rel_w, void* out, int ldo, int oc0,
                           const void* res, int ldr, int B, int h, int w, int C, hipStream_t st, int post_levels = 0);
// gout [B][h][w][ldg] (channels gc0..gc0+C) -> gqkv [B][h][w][3C]; `win` [windows][36][2C] (T): the dK|dV rows of each
// window's 36 ring keys (its own 64 pixels are written straight to gqkv; halo_gather adds the ring rows to the
// border pixels of the neighbouring windows);
// rel-pos gradient slabs
int launch_window_attn_bwd(int dt, const void* qkv, const float* rel_h, const float* rel_w, const void* gout, int ldg,
                           int gc0, void* gqkv, void* win, float* relw, int B, int h, int w, int C, hipStream_t st,
                           int dwt_levels = 0,    // 1, 2: gout is the FULL-RES g_xc tensor; DWT^levels applied on load
                           bool gather = true,    // false: skip the halo gather (border pixels then lack their neighbours' ring rows)
                           bool resident = true); // bf16, C = 64 / 256: whole-window-resident kernel (k_attn_res.hip)
// C = 16, bf16, no fused DWT: one wave per window (k_attn_c16.hip)
int launch_window_attn_fwd_c16(const void* qkv, const float* rel_h, const float* rel_w, void* out, int ldo, int oc0, const void* res,
                               int ldr, int B, int h, int w, hipStream_t st);
// d != nullptr (with wqkv = the packed [48][16] weight, M2T_PACK_COPY): qkv was not saved (may be nullptr); q | k | v are
// recomputed from the branch input d [pixel][16] with the forward kernel's own products (identical bits)
int launch_window_attn_bwd_c16(const void* qkv, const float* rel_h, const float* rel_w, const void* gout, int ldg, int gc0,
                               void* gqkv, void* win, float* relw, int B, int h, int w, hipStream_t st, const void* d = nullptr,
                               const void* wqkv = nullptr);
// k_attn_fwd2.hip (round 5): the same fused C = 256 forward branch as launch_window_attn_fused_prep_fwd, built for two workgroups per
// CU (4 waves, 80 KB of LDS, channel-chunked projection, scores in registers); vring [B (h/8)(w/8)][36][256]: scratch for the ring
// keys' v rows (window_attn_fwd2_vring_elems elements)
size_t window_attn_fwd2_vring_elems(int B, int h, int w);
int launch_window_attn_fused_prep_fwd2(const void* xn, const void* xprev, const float* mean, const float* rstd, int k, void* xin, void* d,
                                       const void* wfrag, const float* rel_h, const float* rel_w, void* qkv, void* out, void* vring,
                                       int B, int h, int w, int windows_per_wg, hipStream_t st, int stagger = 0);
// bf16 C = 16 branch: halo gather + projection data gradient + branch_prep_bwd (k = 0) in one kernel (k_attn_c16.hip)
// nx != nullptr (round 5): the first stage of the InstanceNorm backward rides in the same launch -- extra workgroups sum planes 1 .. 3
// of g_n (complete by then) into npart [B][M2T_NORM_SPLIT][64][2], and every tile leaves the plane-0 sums of its 16 pixels in
// npart0 [tile][2][16]; gn / nx are then the FULL P64 tensors (plane 0 first), mean / rstd [B][64]
int launch_c16_dgrad_prep(void* gqkv, const void* win, const void* wT, const void* gxc, void* gn, int B, int h, int w, hipStream_t st,
                          const void* nx = nullptr, const float* mean = nullptr, const float* rstd = nullptr, float* npart = nullptr,
                          float* npart0 = nullptr);
int launch_window_attn_fwd_resident(const void* qkv, const float* rel_h, const float* rel_w, void* out, int ldo, int oc0,
                                    const void* res, int ldr, int B, int h, int w, int C, int post_levels, hipStream_t st);
// the resident kernel alone (bf16); M2T_UNSUPPORTED when (C, dwt_levels) has no instantiation
// wdfrag != nullptr ((C, dwt_levels) = (256, 2) or (64, 1)): the data gradient of the qkv projection is taken in the same
// launch -- wdfrag = Wqkv^T as MFMA fragments (M2T_PACK_FRAG16_T); own pixels' rows -> gd [pixel][C], the 36 ring keys'
// rows -> gdwin [window][36][C] (added to the border pixels by launch_halo_gather(gdwin, gd, .., C, C, 0))
int launch_window_attn_bwd_resident(const void* qkv, const float* rel_h, const float* rel_w, const void* gout, int ldg, int gc0,
                                    void* gqkv, void* win, float* relw, int B, int h, int w, int C, int dwt_levels, hipStream_t st,
                                    const void* wdfrag, void* gd, void* gdwin,
                                    // (C, dwt_levels) = (64, 1) with the fused data gradient: xsrc [pixel][C] = the branch input and wfrag = Wqkv as
                                    // M2T_PACK_FRAG16: qkv was not saved (may be nullptr), q | k | v are recomputed (identical bits)
                                    const void* xsrc = nullptr, const void* wfrag = nullptr,
                                    // (256, 2) with pb_gd != nullptr: branch_prep_bwd of the NEXT branch k = i + 1 inside (its own-window g_d rows, their
                                    // ring rows, plane k of g_xc, plane k of g_n (written)); gout = plane i of g_xc is then updated in place
                                    const void* pb_gd = nullptr, const void* pb_gdwin = nullptr, const void* pb_gxk = nullptr, void* pb_gnk = nullptr);
int launch_halo_gather(int dt, const void* win, void* dst, int B, int h, int w, int rw, int ld, int coff, hipStream_t st);
// k_attn_c16.hip: the whole C = 16 branch forward (InstanceNorm apply of chunk 0 + qkv projection + attention + residual), bf16.
// x = chunk-0 plane of the block input; wqkv [48][16] (M2T_PACK_COPY); d [B*h*w][16] and qkv [B*h*w][48] are WRITTEN
// (qkv == nullptr: not saved -- launch_window_attn_bwd_c16 then recomputes it from d)
int launch_window_attn_fused_c16_fwd(const void* x, const float* mean, const float* rstd, const void* wqkv, const float* rel_h,
                                     const float* rel_w, void* d, void* qkv, void* out, int ldo, int oc0, int B, int h, int w,
                                     hipStream_t st);
// k_attn_fused.hip: qkv projection + window attention + IWT / residual epilogue in one kernel (bf16, C = 64 / 256).
// x [B][h][w][C]; wfrag = the [3C][C] qkv weight in M2T_PACK_FRAG16 order; qkv [B][h][w][3C] is WRITTEN (saved for the backward)
int launch_window_attn_fused_fwd(const void* x, const void* wfrag, const float* rel_h, const float* rel_w, void* qkv, void* out,
                                 int ldo, int oc0, const void* res, int ldr, int B, int h, int w, int C, int post_levels,
                                 hipStream_t st);
int launch_window_attn_fused_prep_fwd(const void* xn, const void* xprev, const float* mean, const float* rstd, int k, void* xin, void* d,
                                      const void* wfrag, const float* rel_h, const float* rel_w, void* qkv, void* out, int B, int h, int w,
                                      int C, int post_levels, hipStream_t st);
// relw [nwin][10][C] per-window partials -> grel_h / grel_w (torch layouts)
int launch_rel_reduce(const float* relw, float* rel_part, float* grel_h, float* grel_w, int nwin, int C, hipStream_t st);
int launch_rel_reduce1(const float* relw, float* rel_part, int nwin, int C, int* nsplit_out, hipStream_t st);

// ---- k_swin.hip (MedCLIP image tower = Swin-T forward, losses.py:68-69) --------------------
int launch_swin_patchify(int dt, const float* src, const float* src_b, int n_a, int Hs, int Ws, const int* crops, int n, void* out,
                         hipStream_t st);
int launch_layernorm(int dt, const void* x, const float* gamma, const float* beta, void* y, long long M, int C, hipStream_t st,
                     float eps = 1e-5f);     // (Swin / TransBlock: 1e-5; BERT: 1e-12)
// fused Swin MLP (bf16, C = 96 / 192): X <- X + fc2(gelu(fc1(LayerNorm(X)) + b1)) + b2, weights in FRAG16 order
int launch_swin_mlp_fused(void* X, const float* gamma, const float* beta, const void* w1f, const float* b1, const void* w2f,
                          const float* b2, long long M, int C, hipStream_t st);
int launch_frag16_pack(int dt, const float* src, void* dst, int N, int K, hipStream_t st);
int launch_swin_attn(int dt, const void* qkv, const float* bias_table, void* out, int nimg, int H, int W, int C, int heads,
                     int shift, hipStream_t st);
int launch_swin_merge_gather(int dt, const void* x, void* y, int nimg, int H, int W, int C, hipStream_t st);
int launch_swin_head(int dt, const void* x, const float* proj, float* emb, int nimg, hipStream_t st);
int launch_semantic_loss(const float* emb, const float* text, int B, int n_patches, float* per_sample, float* total, hipStream_t st);
int launch_bicubic_resize(const float* src, float* dst, int NC, int Hin, int Win, int Hout, int Wout, hipStream_t st);
int launch_convert(int dt, const float* src, void* dst, long long n, hipStream_t st);
