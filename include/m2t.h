/* m2t.h -- C ABI of libm2t.so: the MI355X (gfx950) implementation of the M2Trans
 * data-parallel training-step hot path.
 *
 * The reference (eezkni/M2Trans) has no FFI: its hot path sits behind a Python plugin hook,
 *     utils.import_module('models.M2Trans_network').create_model(args)      (train.py:70)
 *     sr = model(lr)                                                         (train.py:183)
 *     loss = L1Loss()(sr, hr) * lambda_l1 ; loss.backward() ; optimizer.step() (train.py:199-210)
 * This library is what a drop-in `models/M2Trans_network.py` binds with ctypes
 * (m2trans_amd/M2Trans_network.py; the binding a maintainer adds is shown in INTEGRATION.md).
 * Every entry point takes plain device pointers, sizes and a hipStream_t; no torch types.
 *
 * Conventions
 *   - return 0 on success, < 0 for argument errors (m2t_status), > 0 = hipError_t of a failed
 *     launch; m2t_last_error_string() describes the last failure on the calling thread.
 *     (The reference raises Python exceptions: `assert` M2Trans_network.py:305, ValueError
 *     train.py:72, RuntimeError/KeyError M2Trans_network.py:100-112.)
 *   - kernels never allocate, free or synchronise; the caller owns every buffer, including the
 *     workspace whose size the plan reports.  All launches go to the given stream, so the whole
 *     step can be captured in a HIP graph.
 *   - tensors at the boundary use the reference's layout: NCHW float32, values in [0, rgb_range].
 *     Internal activations are NHWC in `dtype` (0 = float32, exact-fp32 MFMA; 1 = bfloat16 MFMA
 *     with fp32 accumulation / statistics / softmax / master weights).
 *   - parameters live in ONE flat float32 buffer, the trainable tensors of the reference's
 *     state_dict in registration order (head.weight, head.bias, body.0.attn1.rel_h, ...,
 *     tail.*; the 4 frozen MeanShift tensors are not part of it -- they are never used by
 *     forward, M2Trans_network.py:30-31,58-76).  Gradients use the same layout, which is what
 *     makes the data-parallel exchange a single RCCL all-reduce.
 */
#ifndef M2T_H
#define M2T_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#pragma GCC visibility push(default)

typedef struct m2t_plan m2t_plan;

enum m2t_status { M2T_OK = 0, M2T_ERR_ARG = -2, M2T_ERR_STATE = -3 };

int m2t_version(void);
const char* m2t_last_error_string(void);

/* ---- plan: shapes, parameter offsets, workspace layout -------------------------------- */
/* replaces M2Trans.__init__ (models/M2Trans_network.py:17-56) for fixed n_feats = 64, colors = 3.
 * (B, H0, W0) = LR batch shape; the image is reflect-padded to multiples of 32 internally
 * (check_image_size, :78-86). scale in {2,3,4}. */
int m2t_plan_create(m2t_plan** out, int B, int H0, int W0, int scale, int n_blocks, int dtype);
void m2t_plan_destroy(m2t_plan* p);
/* named integer queries: "workspace_bytes", "num_params", "padded_h", "padded_w",
 * "param:<state_dict name>" (offset in floats), "numel:<state_dict name>",
 * "ws:<tensor>" (byte offset of a workspace tensor, e.g. "ws:b0.qkv3"), "wsn:<tensor>" (elements).
 * Returns -1 for an unknown key. */
long long m2t_plan_query(const m2t_plan* p, const char* key);
/* Scheduling / kernel-selection options: each selects between a fused gfx950 kernel and the plain kernels it replaces
 * (which fp32 parity mode always uses), or places the parameter-gradient work; results are the same up to bf16 rounding and
 * every pair is A/B-tested in tests/test_gpu_model.py.  Defaults in brackets; m2t_plan_query("opt:<key>") reads the value in
 * force (gate_branch / wgrad_big_tiles are reported + 1000).  Variants that were measured and lost live under scratch/ with
 * their numbers in profiles/README.md, not behind options.
 *   "side_stream"       [1] parameter-gradient kernels of m2t_backward on a plan-owned second stream
 *   "gate_branch"       [-1] -1: a branch's side work is released right behind its attention launch; 0..3: a block's side work waits for the
 *                           attention launch of that branch (2 = behind the two LDS-filling C = 256 launches; the default of rounds 1-2,
 *                           when the main chain still waited for the side stream once per branch)
 *   "fork_on_kernel"    [1] the event that releases side-stream work rides on the main-stream dispatch it follows (hipExtLaunchKernelGGL
 *                           stop event) instead of a marker packet behind it: -0.7 % step time; 0 = hipEventRecord.  Ignored (0) while the
 *                           stream is being captured into a HIP graph
 *   "fused_prep_fwd"    [1] bf16, C = 64 / 256 branches: branch_prep (norm apply + branch mixing + DWT^L) inside the fused forward attention kernel
 *                           (needs "fused_attn_fwd" >= 1; bit-identical); 0 = branch_prep launches in front of it
 *   "fused_prep_bwd"    [1] bf16: the backward of branch 4's branch_prep inside branch 3's attention backward (same level and window grid; needs
 *                           "attn_bwd" >= 2; bit-identical); 0 = a branch_prep_bwd launch between the two
 *   "fused_l1"          [1] bf16 x4 with the recomputing fused tail backward ("fused_tail" = 2 / 3): a loss requested through m2t_l1_loss_deferred
 *                           is taken inside that kernel (clamp, |sr - hr| partial sums, sign seed on the staged halo); 0 = m2t_backward runs the
 *                           clamp + L1 kernel first (bit-identical gradients either way)
 *   "tail_bwd_mfma32"   [1] bf16 x4 with the recomputing fused tail backward ("fused_tail" = 2 / 3): the round-6 form of that kernel on
 *                           v_mfma_f32_32x32x16_bf16 (k_tail_bwd.hip: LDS pixel order by sub-pixel position, conflict-free operand reads with half
 *                           the bytes per FLOP, gelu'(t2) and g(t2) formed in registers, reflect-border gather without divergent loops, three
 *                           barriers per tile): 346 us against 426 us stand-alone at batch 16.  g(t1), dW3 bit-identical to the 16x16x32 kernel
 *                           (0), dWf / db3 the same products in another fp32 order
 *   "fp32_fast"         [1] fp32: the round-5 kernels of the parity mode -- plain GEMMs (qkv projections and their data gradients, N % 32 == 0,
 *                           K % 32 == 0), qkv weight gradients (N % 64 == 0, K % 64 == 0), the x2 expansions 64 -> 256 (image rows of whole
 *                           128-pixel tiles) and their weight / bias gradients on v_mfma_f32_32x32x2_f32 (k_gemm.hip), the 64 -> 3 tail conv
 *                           weight and data gradient on the VALU (k_conv.hip): exact fp32 products, fp32 accumulation, other summation orders;
 *                           0 = the 16x16x4 kernels of rounds 1-4
 *   "fused_attn_fwd2"   [0] bf16, C = 256 branches with "fused_prep_fwd": the forward kernels that put TWO windows on a CU (k_attn_fwd2.hip: projection
 *                           in four output-channel chunks, scores / softmax / P in registers, v re-read from L2 for P V, IWT^2 straight from the
 *                           accumulators): 1 = 4-wave workgroups of one window (80 KB of LDS, two per CU), 2 = 8-wave workgroups of two
 *                           neighbouring windows (1 for an odd number of windows per image), -1 = variant 1 when the branch has more windows than
 *                           the chip has CUs (batch >= 17 at 128 x 128), 0 = never.  q | k | v bit-identical to the one-window-per-CU kernel, the
 *                           output to fp32 summation order.  Measured at batch 32: 59.3 us against 61.7 us stand-alone, a tie inside the step (both
 *                           windows of a CU start together and stay in the same phase: there is nothing for co-residency to overlap): off by default
 *   "fused_norm_red"    [0] bf16 with "attn_bwd" = 3: the first reduction stage of the InstanceNorm backward (sums of g_n and g_n * xhat per image
 *                           and channel) rides in the C = 16 prep launch -- extra workgroups for the planes of branches 2 .. 4, per-tile sums of
 *                           plane 0 from the tiles that produce it; 0 = its own launch behind that kernel (same sums, another addition order).
 *                           Measured 2.2 % SLOWER on the step (round 5: both roles are memory-heavy and do not overlap): off by default
 *   "wgrad_big_tiles"   [-1] qkv weight gradient of the C = 256 branches with 128 x 128 output tiles: value = target number of
 *                           workgroups (64..512), 0 = off, -1 = auto (256 from 24 576 branch pixels on, i.e. batch >= 24)
 *   "fused_tail"        [3] bf16 x4: 1 = one fused kernel for the high-resolution half of the tail backward (k_tail_bwd.hip); 2 = the same
 *                           and tail.3 expansion + PixelShuffle + GELU + tail conv of the FORWARD in one kernel (k_tail_fwd.hip): gelu(t2)
 *                           and gelu'(t2) are then never stored, the fused backward recomputes them per tile
 *                           (m2t_plan_query("stores_t2") tells whether ws:t2act / ws:t2der are written); 3 = 2 with the forward as the
 *                           row-streaming kernel of round 4 (k_tail_stream.hip, same bits); 4 = 3 with the backward as a row-streaming
 *                           kernel too (k_tail_bwd_stream.hip: same data gradient bits, slower -- kept for A/B); 0 = the plain kernels.
 *                           bf16 x2 / x3: >= 1 = the whole tail as ONE row-streaming forward and ONE recomputing backward kernel
 *                           (m2t_plan_query("stores_t1") = 0: gelu(t) / gelu'(t) are never stored), 0 = the plain kernels
 *   "attn_bwd"          [3] bf16 attention backward: 0 = chunked kernels + halo gather + data-gradient GEMM, 1 = whole-window-
 *                           resident / wave-per-window kernels, 2 = 1 + the data gradient of the qkv projection inside the
 *                           C = 64 / 256 kernels (k_attn_res.hip), 3 = 2 + one kernel for the C = 16 branch's overlap-add,
 *                           projection data gradient and branch_prep_bwd (k_attn_c16.hip; bit-identical to 2)
 *   "conv_rows"         [1] bf16 conv3x3 64 -> 64 (forward and data gradient): row-streaming kernel fed by LDS-DMA with the weights in
 *                           registers (k_conv.hip); 0 = the 8 x 16 tile kernel it replaced (also the fallback for widths the strip
 *                           geometry does not divide).  Bit-identical
 *   "fused_conv_bwd"    [1] bf16 conv3x3 64 -> 64 backward: data gradient and weight / bias gradient in ONE row-streaming pass over the
 *                           output gradient on the main stream (k_conv.hip; data gradient bit-identical, weight gradient the same
 *                           products summed in a different order); 0 = data gradient kernel + weight-gradient kernel on the side stream
 *   "fused_attn_fwd"    [2] bf16, C = 64 / 256: 1 = qkv projection + window attention + IWT / residual in one kernel per window; 2 = the
 *                           same and q | k | v of the C = 64 branch are NOT stored: the resident backward recomputes them from the
 *                           branch input (identical bits; needs "attn_bwd" = 2; m2t_plan_query("stores_qkv2")); 0 = GEMM + attention
 *   "fused_c16_fwd"     [2] bf16, C = 16: 1 = InstanceNorm apply + qkv projection + window attention + residual in one kernel, one wave per
 *                           window; 2 = the same and q | k | v of that branch are NOT stored: the backward kernel recomputes them from
 *                           the branch input (identical bits; needs "attn_bwd" >= 1; m2t_plan_query("stores_qkv1") tells whether
 *                           ws:b*.qkv1 is written); 0 = three kernels
 *   "debug_skip_side"   [0] timing experiments only: skips every parameter-gradient kernel (results are WRONG) */
int m2t_set_option(m2t_plan* p, const char* key, long long value);
/* Gradient buckets for communication overlap (replaces the reduce-to-GPU-0 of nn.DataParallel, train.py:73):
 * m2t_backward completes the flat gradient buffer in "grad_buckets" contiguous ranges, in this order: tail, block
 * pairs from the last to the first, head.  m2t_plan_query("grad_bucket_lo:<i>" / "grad_bucket_hi:<i>") give bucket
 * i's float range; after m2t_backward has been ENQUEUED, m2t_stream_wait_bucket makes `stream` wait until bucket i
 * is final, so an all-reduce of that range can run under the rest of the backward pass. */
int m2t_stream_wait_bucket(m2t_plan* p, int bucket, void* stream);
/* one-time initialisation of the workspace (uploads the weight-packing table). */
int m2t_plan_init_workspace(m2t_plan* p, void* workspace, void* stream);

/* ---- model ----------------------------------------------------------------------------- */
/* M2Trans.forward (models/M2Trans_network.py:58-76): x [B,3,H0,W0] -> sr [B,3,H0*s,W0*s].
 * keep_activations != 0 saves what m2t_backward needs. */
int m2t_forward(m2t_plan* p, const float* params, const float* x, float* sr, float rgb_range,
                int keep_activations, void* workspace, void* stream);
/* L1Loss()(sr, hr) * lambda_l1 (train.py:76,199) on the forward's output + the backward seed.
 * loss_out: device float[1].  divisor = number of elements the mean runs over (pass the GLOBAL
 * count B_global*3*Hs*Ws under data parallelism so that SUM-all-reduced gradients equal the
 * full-batch gradient). */
int m2t_l1_loss(m2t_plan* p, const float* hr, float lambda_l1, double divisor, float rgb_range,
                float* loss_out, void* workspace, void* stream);
/* The same loss and seed, produced INSIDE the next m2t_backward (round 5): loss_out is valid in stream order behind that call, and
 * hr must stay valid until it has been issued.  On the bf16 x4 path the fused tail backward takes the clamp + L1 seed while it
 * stages its g(sr) halo (option "fused_l1"): the seed tensor is never written, one 150 MB pass and two launches leave the step;
 * on every other path m2t_backward runs m2t_l1_loss's kernel first (identical results to m2t_l1_loss).  The loss value differs
 * from m2t_l1_loss's by the order of an fp32 sum.  A later m2t_l1_loss / m2t_set_output_grad / m2t_forward cancels it. */
int m2t_l1_loss_deferred(m2t_plan* p, const float* hr, float lambda_l1, double divisor, float rgb_range,
                         float* loss_out, void* workspace, void* stream);
/* alternative seed: an arbitrary upstream gradient g_sr [B,3,H0*s,W0*s] (torch autograd). */
int m2t_set_output_grad(m2t_plan* p, const float* g_sr, float rgb_range, void* workspace, void* stream);
/* loss.backward() (train.py:209) restricted to the model: fills grads (flat, same layout as
 * params; every element is written). */
int m2t_backward(m2t_plan* p, const float* params, const float* x, float* grads, void* workspace,
                 void* stream);
/* torch.optim.Adam(lr, betas, eps, weight_decay=0).step() (train.py:81,210), fused over the flat
 * buffers; step = 1-based count; grad_scale multiplies g first (1/world_size after a SUM all-reduce). */
int m2t_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* ---- measurement: per-kernel-category timing with HIP events recorded on the launch stream.
 * category ids are listed in m2trans_amd/profile.py; mask bit i enables category i; 0 = off. */
int m2t_profile_enable(unsigned long long category_mask);
int m2t_profile_read(int category, double* total_ms, long long* launches);
/* single-kernel categories (attention, 3x3 conv, tail kernels) are timed by events that ride on the dispatch itself; such a
 * dispatch costs ~10 us of launch path, so a step with 16 timed launches runs 2 % slower than an untimed one.  n > 1 times a
 * uniform sample -- every n-th launch of each category -- instead (process-wide, default 1 = every launch); m2t_profile_read
 * then returns the total and the count of the SAMPLED launches. */
int m2t_profile_sample_every(int n);

/* ---- operators (NHWC tensors in `dtype` unless stated) ---------------------------------- */
/* DWT.forward / IWT.forward (models/M2Trans_network.py:198-237), `levels` in {1,2} applied
 * back-to-back; bit-exact in float32.  src [B,H,W,C] <-> dst [B,H/2^l,W/2^l,C*4^l]. */
int m2t_dwt(int dtype, int levels, const void* src, void* dst, int B, int H, int W, int C, void* stream);
int m2t_iwt(int dtype, int levels, const void* src, void* dst, int B, int H, int W, int C, void* stream);
/* nn.PixelShuffle(r) / its inverse on NCHW float32 (models/M2Trans_network.py:43,46,53): bit-exact.
 * in [B,C*r*r,H,W] -> out [B,C,H*r,W*r]. */
int m2t_pixel_shuffle(const float* in, float* out, int B, int C, int H, int W, int r, void* stream);
int m2t_pixel_unshuffle(const float* in, float* out, int B, int C, int H, int W, int r, void* stream);
/* NCHW float32 -> NHWC dtype and back. */
int m2t_to_nhwc(int dtype, const float* nchw, void* nhwc, int B, int C, int HW, void* stream);
int m2t_to_nchw(int dtype, const void* nhwc, float* nchw, int B, int C, int HW, void* stream);
/* TBlock.forward after the qkv projection (models/M2Trans_network.py:310-332):
 * qkv [B,h,w,3C] (q|k|v), rel_h/rel_w float32 [10*C/2] -> out [B,h,w,C]; C in {16,64,256}. */
int m2t_window_attention_fwd(int dtype, const void* qkv, const float* rel_h, const float* rel_w, void* out,
                             int B, int h, int w, int C, void* stream);
/* its autograd: gout [B,h,w,C] -> gqkv [B,h,w,3C], grel_h/grel_w float32 [10*C/2].
 * scratch: m2t_window_attention_bwd_scratch_bytes(dtype,B,h,w,C) bytes. */
size_t m2t_window_attention_bwd_scratch_bytes(int dtype, int B, int h, int w, int C);
int m2t_window_attention_bwd(int dtype, const void* qkv, const float* rel_h, const float* rel_w, const void* gout,
                             void* gqkv, float* grel_h, float* grel_w, void* scratch, int B, int h, int w, int C,
                             void* stream);

/* ---- SemanticLoss (losses.py:18-81): MedCLIP image tower = Swin-T 224 forward + the loss value ----
 * Forward only: the reference evaluates it under torch.no_grad() (losses.py:63), so it contributes a
 * constant to the logged loss and no gradient.  Weights: ONE flat float32 buffer in the order reported
 * by m2t_swin_param_name() (HF swin-tiny checkpoint names of transformers 4.24 + "projection_head.weight"
 * [512,768], the MedCLIP vision projection). */
typedef struct m2t_swin m2t_swin;
int m2t_swin_create(m2t_swin** out, int max_images, int dtype);
void m2t_swin_destroy(m2t_swin* p);
/* "workspace_bytes", "num_params", "num_param_tensors", "param:<name>", "numel:<name>" */
long long m2t_swin_query(const m2t_swin* p, const char* key);
const char* m2t_swin_param_name(const m2t_swin* p, int index);
/* once per weight set: converts / fuses the frozen weights into the workspace */
int m2t_swin_load_weights(m2t_swin* p, const float* weights, void* workspace, void* stream);
/* medmodel.encode_image on n 224x224 crops (createNRandompatches, losses.py:29-40):
 * src [n_src,3,Hs,Ws] float32 NCHW on the device, crops_host int[n][3] = (source index, row0, col0)
 * -> emb [n,512] float32, unit L2 norm. */
int m2t_swin_encode(m2t_swin* p, const float* src, int n_src, int Hs, int Ws, const int* crops_host, int n,
                    float* emb, void* workspace, void* stream);
/* the same with the source images in two tensors (source index < n_a: src_a, else src_b[index - n_a]): the SR and HR
 * batches of one training step (train.py:203-205) are encoded together without a torch.cat of 2 x B x 3 x Hs x Ws floats */
int m2t_swin_encode_pair(m2t_swin* p, const float* src_a, int n_a, const float* src_b, int n_b, int Hs, int Ws,
                         const int* crops_host, int n, float* emb, void* workspace, void* stream);
/* losses.py:71-79 for a batch: emb [2B,512] (SR embeddings then HR embeddings), text [B,512] (any norm)
 * -> per_sample[B] = |sr.t - hr.t| / n_patches, total[1] = their sum. */
int m2t_semantic_loss(const float* emb, const float* text, int B, int n_patches, float* per_sample, float* total,
                      void* stream);
/* F.interpolate(mode='bicubic', align_corners=True) (losses.py:53-54): src [NC,Hin,Win] -> dst [NC,Hout,Wout]. */
int m2t_bicubic_resize(const float* src, float* dst, int NC, int Hin, int Win, int Hout, int Wout, void* stream);

/* ---- SemanticLoss text tower (losses.py:22-25,64-65,74): medmodel.encode_text ---------------------------------
 * BERT-base forward (Bio_ClinicalBERT geometry: 12 layers x 768, 12 heads, 3072 intermediate, vocab 28 996, LayerNorm eps
 * 1e-12, erf GELU) -> mean over the tokens of hidden states 1, 2 and -1 -> Linear(768, 512, no bias) -> unit L2 norm, as the
 * un-vendored `medclip` package defines MedCLIPTextModel.forward / MedCLIPModel.encode_text (PARITY UNPINNED: package and
 * checkpoint absent from the reference tree).  Forward only (torch.no_grad() in the reference, losses.py:63).
 * Weights: ONE flat float32 buffer in the order of m2t_text_param_name(): HF BertModel checkpoint names of transformers
 * 4.24 without `pooler.*`, then "projection_head.weight" [512,768]. */
typedef struct m2t_text m2t_text;
int m2t_text_create(m2t_text** out, int max_seqs, int max_len /* <= 128 */, int dtype);
void m2t_text_destroy(m2t_text* p);
/* "workspace_bytes", "num_params", "num_param_tensors", "max_seqs", "max_len", "param:<name>", "numel:<name>" */
long long m2t_text_query(const m2t_text* p, const char* key);
const char* m2t_text_param_name(const m2t_text* p, int index);
int m2t_text_load_weights(m2t_text* p, const float* weights, void* workspace, void* stream);
/* ids_host, mask_host: int[n][len] in HOST memory (tokenizer output; position ids are 0..len-1 and token types 0, BertModel's
 * defaults).  The reference passes outputs['token_type_ids'] in the input_ids slot (losses.py:65): a caller that wants the
 * reference's value passes those zeros here.  -> emb [n][512] float32 on the device, unit norm. */
int m2t_text_encode(m2t_text* p, const int* ids_host, const int* mask_host, int n, int len, float* emb, void* workspace,
                    void* stream);

/* ---- util/rlutrans.py TransBlock (dead code in the reference: nothing imports it) --------------------
 * TransBlock.forward (util/rlutrans.py:82-87) for dim = 64, 8 heads: x + EffAttention(LayerNorm(x)) (:30-67: reduce,
 * qkv, softmax attention inside token chunks of length N // 16, proj), then x + Mlp(LayerNorm(x)) (:11-27: 64 -> 16,
 * ReLU, 16 -> 64).  Forward only.  params: the 22 928 float32 values of TransBlock(n_feat=64, dim=64).state_dict() in
 * state_dict order (atten.reduce.weight, atten.qkv.weight, atten.proj.weight, atten.proj.bias, norm1.weight,
 * norm1.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias, norm2.weight, norm2.bias); x, y [B,N,64]
 * float32 on the device (y may alias x); dtype = compute element type (0 fp32, 1 bf16); N >= 16. */
size_t m2t_transblock_workspace_bytes(int B, int N, int dtype);
int m2t_transblock_forward(const float* params, const float* x, float* y, int B, int N, int dtype, void* workspace,
                           void* stream);

/* ---- evaluation metrics of the reference's test loop ------------------------------------------
 * test.py:101-113 / train.py:299-312: Y channel of utils.rgb_to_ycbcr (utils.py:121-146), `crop` (= scale) border
 * pixels removed, x255 when rgb_range == 1; then utils.calc_psnr (utils.py:179-184) and utils.calc_ssim
 * (utils.py:232-234 -> pytorch_msssim.ssim defaults).  sr, hr: float32 NCHW [B,3,H,W] on the device.
 * out: double[B][2] on the device = { mean(((Ysr-Yhr)/255)^2), mean SSIM } per image (PSNR = -10 log10 out[b][0]).
 * Y is bit-identical to the reference's fp32 Y; the reductions and the SSIM filtering are fp64.
 * window_host: the 11 fp32 taps of the SSIM window in HOST memory (the dependency computes them with torch.exp in
 * fp32, whose last bit a caller may want to reproduce), or NULL for the built-in exp(-(i-5)^2/4.5) normalised in fp32. */
size_t m2t_eval_metrics_scratch_bytes(int B, int H, int W, int crop);
int m2t_eval_metrics(const float* sr, const float* hr, int B, int H, int W, int crop, float rgb_range,
                     const float* window_host, void* scratch, double* out, void* stream);

/* piq.gmsd(hr, sr, data_range=1., reduction='none') of the same loop (test.py:98): gradient magnitude similarity deviation of
 * the 2x2-pooled luminance (0.299 R + 0.587 G + 0.114 B), Prewitt gradients, t = 170 / 255^2, population standard deviation of
 * the similarity map.  x, y: float32 NCHW [B,3,H,W] on the device; out: double[B] on the device.  The dependency (`piq`) is
 * absent from the reference tree: PARITY UNPINNED, checked against an fp64 restatement of the published algorithm. */
size_t m2t_eval_gmsd_scratch_bytes(int B);
int m2t_eval_gmsd(const float* x, const float* y, int B, int H, int W, float data_range, void* scratch, double* out, void* stream);

/* piq.fsim(hr, sr, data_range=1., reduction='none') of the same loop (test.py:95-96; piq 0.8.0 per environment.yml:118): FSIMc --
 * k x k average pooling (k = max(1, round(min(H, W) / 256))), YIQ, phase congruency from a 4-orientation x 4-scale log-Gabor bank
 * (explicit 2-D DFTs, any image size), Scharr gradient magnitude, chroma similarity, sum(S_L S_C^0.03 PC_m) / sum(PC_m).  fp64 on the
 * device (k_fsim.hip).  x, y: float32 NCHW [B,3,H,W] on the device; out: double[B] on the device; scratch:
 * m2t_eval_fsim_scratch_bytes(H, W) bytes (independent of B: image pairs are processed one after the other).
 * The dependency is absent from the reference tree: PARITY UNPINNED, checked against oracle/fsim_oracle.py. */
size_t m2t_eval_fsim_scratch_bytes(int H, int W);
int m2t_eval_fsim(const float* x, const float* y, int B, int H, int W, float data_range, void* scratch, double* out, void* stream);

/* ---- training input pipeline --------------------------------------------------------------------
 * datas/us1k.py:16-36 crop_patch + utils.py ndarray2tensor + the /255 of datas/us1k.py:169, for n samples at once,
 * cut out of uint8 HWC images that stay resident in device memory (lr_pool / hr_pool: the npy cache, back to back).
 * desc_host: long long[n][8] in HOST memory = { byte offset of the LR image in lr_pool, of the HR image in hr_pool,
 * LR row length in pixels, HR row length in pixels, lx, ly (LR corner; HR corner = scale x), flags (bit0 [:, ::-1],
 * bit1 [::-1, :], bit2 transpose(1,0,2), applied in that order like the reference), LR image height }.
 * lr_out [n,channels,patch/scale,patch/scale], hr_out [n,channels,patch,patch]: float32 NCHW, bit-identical to the
 * reference's tensors.  The random draws themselves stay with the caller (reference order: lx, ly, hflip, vflip, rot). */
int m2t_crop_patches(const unsigned char* lr_pool, const unsigned char* hr_pool, const long long* desc_host, int n,
                     int channels, int patch_size, int scale, float* lr_out, float* hr_out, void* stream);

/* datas/benchmark.py:62-72 (`Benchmark.__getitem__`): top-left h x w crop of a uint8 HWC image [img_h,img_w,channels]
 * resident in device memory -> float32 [channels,h,w] / 255, bit-identical to ndarray2tensor(...)/255. */
int m2t_image_to_tensor(const unsigned char* img, int img_h, int img_w, int channels, int h, int w, float* out, void* stream);
/* train.py:177-181: utils.cutmix (utils.py:16-71) and utils.cut_out (utils.py:74-108) on a device batch.  The random
 * draws stay with the caller (m2trans_amd/augment.py mirrors the reference's order); table_dev: int[B][1 + 5 * max_boxes]
 * on the device = { n, then n x (x1, y1, x2, y2, source sample) } in the order the reference applies the boxes.
 * mode 0 = cutmix: a pixel takes the value of the LAST covering box's source sample in the ORIGINAL tensor;
 * mode 1 = cut_out: a covered pixel is multiplied by 0.  mult scales the boxes (1 for LR, `scale` for HR, utils.py:49).
 * src, dst: float32 [B,C,H,W], distinct buffers.  Bit-identical to the reference's tensors. */
int m2t_box_mix(const float* src, float* dst, int B, int C, int H, int W, const int* table_dev, int max_boxes, int mode,
                int mult, void* stream);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
