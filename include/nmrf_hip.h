/* libnmrf_hip.so -- C ABI of the MI355X (gfx950) NMRF-Stereo inference hot path.
 *
 * Every entry point is `extern "C"`, takes raw DEVICE pointers + sizes + a
 * hipStream_t (passed as void*), enqueues on that stream, never allocates,
 * never synchronises, never throws; returns 0 on success or a negative
 * NMRF_E* code (nmrf_strerror() gives text).  All tensors are dense
 * row-major fp32 unless stated; "token-major" = [tokens, channels] with token
 * index ((b*H + y)*W + x)*N + n  (pixel-major, label-minor -- the reference's
 * '(b h w) n c' order, nmrf/models/NMP.py:191,347).
 *
 * Each declaration cites the reference interface it replaces
 * (file:line relative to the reference repo root).
 */
#ifndef NMRF_HIP_H
#define NMRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMRF_OK 0
#define NMRF_EINVAL -1      /* bad size / unsupported configuration */
#define NMRF_ELAUNCH -2     /* hipGetLastError() after the launch was not hipSuccess */
#define NMRF_ENULL -3       /* null pointer */

const char *nmrf_strerror(int code);

/* fp16 range of the split-operand kernels (csrc/split_mfma.h).  Every contraction of the path multiplies fp32 operands as fp16
 * hi/lo pairs, so an ACTIVATION with |x| >= 65520 (or a NaN) cannot be represented; the reference's plain fp32 has no such limit
 * (nmrf/models/NMP.py:54-66).  The limit is guarded: the entry points that split activations take `int *range_flag` -- a device
 * int32 owned by the caller (may be NULL: no reporting) into which the kernel atomically ORs 1 when any activation it converted was
 * out of range.  The flag is sticky; the caller reads and clears it at a point where it synchronises anyway
 * (nmrf_amd.kernels.check_range -> NmrfHipError).  Weights are rescaled by a power of two when they are packed and have no limit.
 * Below the range: absolute error <= 2^-25 per operand element (values below 2^-24 are flushed) -- one fp32 rounding of an O(1)
 * accumulator.
 * The two window-attention kernels run at their VGPR cap and carry no guard of their own: nmrf_window_attn_f32 scans its qkv operand
 * in a separate pass when range_flag != NULL; pass NULL when qkv was produced by nmrf_nmp_block16_f32, which range-checks its q_out. */
int nmrf_range_scan_f32(const float *x, int64_t n, int *range_flag, void *stream);   /* n % 4 == 0, x 16-byte aligned */
/* ABI version of this header; bumps on any signature change. */
#define NMRF_ABI_VERSION 27
#define NMRF_ABI_STR "27"
int nmrf_abi_version(void);   /* = NMRF_ABI_VERSION */
/* "abi<version>-<16 hex digits>": the digits are a hash of every source and header the library was built from (python -m
 * nmrf_amd.build).  The tools library (libnmrf_hip_debug.so) of the same build returns the same string; the binding refuses one that
 * does not.  No counterpart in the reference (its extension is built by setup.py into site-packages, ops/setup.py:66-71). */
const char *nmrf_build_stamp(void);

/* A2  group-wise correlation volume.
 * replaces build_correlation_volume + the permute of DPN.forward
 * (nmrf/models/submodule.py:4-23, nmrf/models/DPN.py:117).
 * f1,f2 [B,C,H,W] (NCHW) -> vol [B*H*W, G, D]; vol[p,g,d] = mean_{c in group g} f1[c,y,x]*f2[c,y,x-d], 0 for x<d.
 * C % G == 0, D <= 64. */
int nmrf_cost_volume_f32(const float *f1, const float *f2, int B, int C, int H, int W, int D, int G,
                         float *vol, void *stream);

/* A3  Conv1d(G->8,k5,p2)-ReLU-Conv1d(8->16)-ReLU-Conv1d(16->1) along D, softmax over D.
 * replaces DPN.mlp + softmax (nmrf/models/DPN.py:32-38,118-119).
 * vol [P,G,D]; w0 [8,G,5] b0[8] w1 [16,8,5] b1[16] w2 [1,16,5] b2[1] -> prob [P,D].  G<=4, D<=64. */
int nmrf_dpn_filter_softmax_f32(const float *vol, const float *w0, const float *b0, const float *w1,
                                const float *b1, const float *w2, const float *b2, int64_t P, int G, int D,
                                float *prob, void *stream);

/* A4  label-seed NMS + top-k with ATen-CPU tie order.
 * replaces max_pool1d / masked assign / torch.topk (nmrf/models/DPN.py:120-125).
 * prob [P,D] -> seeds [P,K] int64, sorted by suppressed prob desc; ties resolved exactly as
 * libstdc++ nth_element+sort on (value,index) pairs (ATen TopKImpl.h), bit-exact.
 * do_nms=0 skips the suppression (plain top-k).  D<=64, K<=8, K*64>D. */
int nmrf_nms_topk_f32(const float *prob, int64_t P, int D, int K, float eps, int do_nms,
                      int64_t *seeds, void *stream);

/* A6  seed features: 9-tap x G-group cost gather + Fourier(31) of the seed.
 * replaces Propagation.sample_cost and fourier_coord_embed (nmrf/models/NMP.py:619-634,35-51,646-647).
 * vol [P,G,D], seeds [P,N] int64 -> cost [P*N, G*9] (group-major, tap-minor, taps clamped to [0,D-1]),
 * enc rows of enc_ld >= 31 floats = [sin(c*2^i) i<15 | cos(c*2^i) | c | 0 ...], c = seed*normalizer.  enc may be NULL. */
int nmrf_seed_features_f32(const float *vol, const int64_t *seeds, int64_t P, int G, int D, int N,
                           float normalizer, float *cost, float *enc, int enc_ld, void *stream);

/* A4 + A6 in one launch: NMS + top-k (semantics and tie order of nmrf_nms_topk_f32, bit-exact) on one wave per pixel with the
 * row in registers, followed at once by the seed features of nmrf_seed_features_f32 and the float copy of the seeds that
 * Propagation.forward hands on (nmrf/models/DPN.py:120-125; NMP.py:619-634,35-51,646-647).
 * prob [P,D], vol [P,G,D] (may be NULL when cost is) -> seeds [P,K] int64, seeds_f [P,K] float (may be NULL),
 * cost [P*K, G*9] (may be NULL), enc rows of enc_ld >= 31 floats (may be NULL).  D<=64, K<=8, K*64>D. */
int nmrf_seed_select_f32(const float *prob, const float *vol, int64_t P, int G, int D, int K, float eps, int do_nms,
                         float normalizer, int64_t *seeds, float *seeds_f, float *cost, float *enc, int enc_ld, void *stream);

/* Fourier(31) of arbitrary fp32 coordinates (labels / refined disparity).
 * replaces fourier_coord_embed call sites nmrf/models/NMP.py:743,846.
 * coord [T] -> enc rows of 31 floats (+ zeros up to ld) written at enc + row*ld, row = out_map ? out_map[t] : t (negative:
 * skipped; the zero-padded token grids of NMP.py:745-762 are filled in place). */
int nmrf_fourier_embed_f32(const float *coord, int64_t T, float normalizer, float *enc, int ld, const int *out_map,
                           void *stream);

/* LayerNorm(x) concatenated with a per-token (extra_div=1) or per-pixel (extra_div=N) side vector;
 * builds the q/k(/v) GEMM operand of every message-passing layer in one pass.
 * replaces norm1 + torch.cat (nmrf/models/NMP.py:92-93,344-346,545-548).
 * x [T,C] (C==128), gamma,beta [C], extra [T/extra_div, E] (or NULL with E=0)
 * -> out[t, 0:C] = LN(x[t]) (eps), out[t, C:C+E] = extra[t/extra_div], out[t, C+E:ld] = 0.  ld>=C+E, ld%4==0. */
int nmrf_ln_concat_f32(const float *x, const float *gamma, const float *beta, float eps, const float *extra,
                       int E, int extra_div, int64_t T, int C, float *out, int ld, void *stream);

/* Same with the residual add of the preceding block fused in: x_out = x + y, out = [LN(x_out) | extra | 0].
 * replaces `shortcut + proj(msg)` / `x + mlp(...)` followed by the next norm (nmrf/models/NMP.py:106,361-362,572-573).
 * x_out may alias x. */
int nmrf_add_ln_concat_f32(const float *x, const float *y, float *x_out, const float *gamma, const float *beta,
                           float eps, const float *extra, int E, int extra_div, int64_t T, int C, float *out, int ld,
                           void *stream);

/* A7  cross-stripe attention with LePE, both stripe directions in one call (SPLIT_SIZE==1).
 * replaces CSWinAttention.forward x2 + cat (nmrf/models/NMP.py:429-505,568-570).
 * qkv [T,3C] token-major (q|k|v, each C=128 = 2 halves x 2 heads x 32); half 0 -> vertical stripes
 * (one per image column), half 1 -> horizontal stripes (one per image row).  q is scaled by 32^-0.5.
 * lepe_v, lepe_h: depthwise 3x3 kernels [C/2,1,3,3] of attns.0 / attns.1.  -> out [T,C].
 * axes: bit 0 = run the vertical stripes (writes out[:, 0:C/2]), bit 1 = the horizontal ones (out[:, C/2:C]);
 * 3 = both (two kernel launches on `stream`). */
int nmrf_stripe_attn_f32(const float *qkv, const float *lepe_v, const float *lepe_h, int B, int H, int W, int N,
                         int C, int axes, int kv16, float *out, int *range_flag, void *stream);
/* (kv16 != 0: the k | v thirds of qkv hold split fp16 operand pairs, see nmrf_nmp_block16_f32; N == 4.  Same results, bit for bit,
 *  as on the fp32 rows they were split from; the range of k and v is then the producer's to check.) */

/* A9  warp right maps at x-label, group correlation, concat.
 * replaces Inference.sample_fmap x2 + corr + cat (nmrf/models/NMP.py:683-741, 839-844).
 * labels [B*H*W*N]; f1,f2 [B,Cf,H,W]; g1,g2 [B,Cg,H,W] (NCHW);
 * -> out[t, 0:Cf]=f1, [Cf:2Cf]=warp(f2), [2Cf:2Cf+groups]=mean over Cg/groups channels of g1*warp(g2); row stride ld.
 * Sampling reproduces F.grid_sample(bilinear, zeros, align_corners=True) incl. its normalise/unnormalise round trip.
 * token_major != 0: the four maps are [B,HW,C] (nmrf_conv1x1_in_relu_f32 with token_major = 1; Cf = 64, Cg = 256, groups = 32
 * only): every tap is then one contiguous row read with 16-byte loads.  Same results, bit for bit. */
int nmrf_warp_corr_concat_f32(const float *labels, const float *f1, const float *f2, const float *g1,
                              const float *g2, int B, int H, int W, int N, int Cf, int Cg, int groups,
                              float *out, int ld, int token_major, void *stream);
/* The same with the Fourier embedding of the labels (nmrf_fourier_embed_f32: FourierEmbedding of the stage's label_seed,
 * nmrf/models/NMP.py:675, 683-741) written in the same launch: enc[row(t), 0:31] (+ zero pad columns up to enc_ld), row(t) =
 * enc_map ? enc_map[t] : t, rows with a negative map entry are skipped.  token_major maps only (NMRF_EINVAL otherwise); enc == NULL:
 * exactly nmrf_warp_corr_concat_f32.  Same bits as the two separate calls. */
int nmrf_warp_corr_concat_fourier_f32(const float *labels, const float *f1, const float *f2, const float *g1,
                                      const float *g2, int B, int H, int W, int N, int Cf, int Cg, int groups,
                                      float *out, int ld, int token_major, float normalizer, float *enc, int enc_ld,
                                      const int *enc_map, void *stream);

/* A10(i)  per-pixel self-edge attention over the N sibling labels.
 * replaces the attention core of BasicAttention.forward_pre (nmrf/models/NMP.py:97-103).
 * qkv [T,3C] (q|k|v), heads*32==C, N<=8 -> out [T,C]. */
int nmrf_self_attn_f32(const float *qkv, int64_t T, int N, int C, int heads, float *out, void *stream);

/* A10(ii)/A13  (shifted-)window attention with relative-position q/k/v embeddings.
 * replaces WindowAttention.forward incl. roll, partition, masks (nmrf/models/NMP.py:185-289, 803-826).
 * qkv [B,Hp,Wp,N,3C]; table [(2*win-1)^2, 3C] (= relative_position_enc_table; per head 96 cols eq|ek|ev);
 * Hp%win==0, Wp%win==0; shift in [0,win); sibling_mask!=0 forbids attention between different labels of one pixel.
 * -> out [B,Hp,Wp,N,C] (already un-rolled). heads*32==C. */
int nmrf_window_attn_f32(const float *qkv, const float *table, int B, int Hp, int Wp, int N, int C, int heads,
                         int win, int shift, int sibling_mask, int kv16, float *out, int *range_flag, void *stream);
/* (kv16 != 0: the k | v thirds of qkv hold split fp16 operand pairs, see nmrf_nmp_block16_f32; win == 6, N == 4, range_flag == NULL:
 *  the range of such rows was checked by their producer.  In this form the q . E_k[rel] / k . E_q[rel] terms (NMP.py:255-268) run on the
 *  fp16 matrix pipe with the table staged x 2^10 as split fp16 pairs: |table| must stay below 32 -- a parameter, checked by the CALLER.
 *  kv16 == 2: the same rows, the same kernel, but those relative-position terms on the VALU in fp32 -- no bound on the table; the
 *  form a caller selects for a checkpoint whose table exceeds 32 (nmrf_amd.kernels.window_attn does, once per parameter version).
 *  Rows in a configuration the pre-split kernel does not cover return NMRF_EINVAL instead of being read as floats.) */

/* A10(ii), the shipped inference configuration as a persistent kernel (csrc/window_attn6.hip): 6 x 6 windows, four labels per pixel,
 * C == 128 (four heads), kv16 rows (see nmrf_nmp_block16_f32).  Same result as nmrf_window_attn_f32(win 6, N 4, kv16 1) up to
 * summation order (WindowAttention.forward, nmrf/models/NMP.py:185-289, 803-826).  The relative-position table arrives PACKED:
 *   nmrf_window_table_pack_f32(table [121, 3C], C, heads) -> packed, heads x 49152 bytes: per head ek and eq as split fp16 pairs
 *   x 2^10 (eq also x s log2 e) in [hi | lo][8 channel chunks][121 rows][4 halves] order, then ev in fp32 [8][121][4], zero-padded;
 *   the layout a block copies into LDS as is.  Pack once per parameter version; |table| < 32 is the CALLER's check as above.
 * B * Hp * Wp * 4 * 384 must stay below 2^30 (32-bit byte offsets), else NMRF_EINVAL: use nmrf_window_attn_f32. */
int nmrf_window_table_pack_f32(const float *table, int C, int heads, void *packed, void *stream);
int nmrf_window_attn6_f32(const float *qkv, const void *packed, int B, int Hp, int Wp, int C, int heads, int shift,
                          int sibling_mask, float *out, void *stream);

/* A8/A11/A14  narrow prediction-head layers: out[T,N] = act(x[T,K] w[N,K]^T + bias), N <= 64, K % 4 == 0, K <= 512
 * (LDS: 32*(K+4) + 8*npt*K floats <= 64 KiB), act 0 = identity, 1 = ReLU; bias may be NULL.
 * replaces the last nn.Linear of prop_head / infer_head / refine_head and infer_score_head
 * (nmrf/models/NMP.py:54-66, nmrf/models/NMRF.py:82-83,105,218-220,238; nmrf/models/DPN.py:65,131). */
int nmrf_linear_smalln_f32(const float *x, const float *w, const float *bias, int64_t T, int K, int N, int act,
                           float *out, void *stream);

/* A16  superpixel-guided disparity downsample (evaluation only).  PARITY UNPINNED: the reference announces this operator
 * (README.md:48) but ships neither its source nor frame_utils.downsample_disp; semantics reconstructed from the call site
 * nmrf/utils/evaluation.py:361-378 and fixed in oracle/superpixel_oracle.py: per 8x8 cell, one mode per superpixel
 * segment = mean of its valid (> 0) disparities; modes ordered by pixel count (desc) then label (asc); first K kept, 0 =
 * empty slot.  disp [B,H,W] f32 (0 = invalid), labels [B,H,W] int32 -> out [B, H/8, W/8, K] (H, W truncated to x8). */
int nmrf_superpixel_downsample_f32(const float *disp, const int *labels, int B, int H, int W, int K, float *out,
                                   void *stream);

/* A11/A12  coarse heads epilogue: relu(label+delta), winner-take-all over N by score (first max),
 * x2, 4x4 lower median.  replaces NMRF.forward (nmrf/models/NMRF.py:219-232).
 * delta, score [T,64] (token-major, 64 = 8x8 sub-pixels hs-major); labels [T] -> disp_curr [B, 2H, 2W] (1/4-px units). */
int nmrf_wta_median_f32(const float *delta, const float *score, const float *labels, int B, int H, int W, int N,
                        float *disp_curr, void *stream);

/* A11 + A12 in one launch: infer_head (MLP 128-128-128-64, ReLU) and infer_score_head (Linear 128-64) on tgt [T = B*H*W*4, 128], then
 * what nmrf_wta_median_f32 does with their rows -- which never leave the CU.  replaces NMRF.forward (nmrf/models/NMRF.py:218-232); the
 * 0.25 factor of the score (NMRF.py:221) does not change the arg-max and is not applied.  N must be 4.
 * stream_w: nmrf_pack_split_weight_f32 pairs (Kp = 128) of infer_head.layers[0].weight, infer_score_head.weight, layers[1].weight,
 * layers[2].weight, in this order (96 pairs: total_stages = 12); b1 / b2 / b3: the head's biases, bs: the score head's (may be NULL);
 * inv_scales: HOST array of 4 floats (1 / pack scale of W1, W2, W3, Ws).  Same bits as the three separate launches. */
int nmrf_heads_wta_f32(const float *tgt, int B, int H, int W, int N, const void *stream_w, int total_stages, const float *b1,
                       const float *b2, const float *b3, const float *bs, const float *inv_scales, const float *labels,
                       float *disp_curr, int *range_flag, void *stream);

/* A14  refinement epilogue: relu(disp_curr+delta), 4x4 pixel shuffle, x4, crop.
 * replaces NMRF.forward (nmrf/models/NMRF.py:240-251) + InputPadder.unpad (nmrf/utils/frame_utils.py:277-281).
 * delta [B*H4*W4,16], disp_curr [B,H4,W4] -> disp_pred [B,4H4,4W4] (1/4-px units), disp [B,outH,outW] = 4*pred cropped. */
int nmrf_refine_epilogue_f32(const float *delta, const float *disp_curr, int B, int H4, int W4, int outH, int outW,
                             float *disp_pred, float *disp, void *stream);
/* The same with its head in one launch: tgt [T = B*H4*W4, 128] -> refine_head (MLP 128-128-128-16, ReLU; nmrf/models/NMRF.py:105, 238)
 * -> the epilogue above, the head's [T,16] rows staying in registers.  stream_w / total_stages / b1..b3 / inv_scales (HOST, 3 floats):
 * exactly those of nmrf_mlp_chain_f32 kind 2 with n_out = 16.  Same bits as the two launches. */
int nmrf_refine_head_epilogue_f32(const float *tgt, int B, int H4, int W4, const void *stream_w, int total_stages,
                                  const float *b1, const float *b2, const float *b3, const float *inv_scales,
                                  const float *disp_curr, int outH, int outW, float *disp_pred, float *disp, int *range_flag,
                                  void *stream);

/* N2 (stock conv band)  InstanceNorm2d without affine, fused with what follows it:
 *   y = [relu_mid] IN(x) ; y = [relu_out] (y + residual)        residual may be NULL
 * replaces nn.InstanceNorm2d + ReLU (+ residual add + ReLU) of the reference's conv heads and backbone blocks
 * (nmrf/models/NMRF.py:56-65, nmrf/models/DPN.py:45-49, nmrf/models/backbone.py:38-46,87-88).
 * x, residual, y: [planes, HW] (planes = B*C of an NCHW tensor); ws: workspace of 2*planes*ceil(HW/8192) floats. */
int nmrf_instance_norm_f32(const float *x, const float *residual, int64_t planes, int64_t HW, float eps,
                           int relu_mid, int relu_out, float *ws, float *y, void *stream);

/* A15  multi-scale deformable attention.
 * replaces ms_deform_attn_forward / ms_deform_attn_backward of the reference extension
 * (ops/src/vision.cpp:13-16, ops/src/cuda/ms_deform_attn_cuda.cu:20-153, ops/src/cuda/ms_deform_im2col_cuda.cuh).
 * value [B,S,M,D]; shapes [L,2] int64 (H,W) ON DEVICE; lvl_start [L] int64 ON DEVICE;
 * loc [B,Lq,M,L,P,2] (x,y in [0,1]); w [B,Lq,M,L,P] -> out [B,Lq,M*D].
 * backward: grad_value must be zero-filled by the caller (accumulated with atomics);
 * grad_loc / grad_w are fully written. */
int nmrf_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start, const float *loc,
                          const float *w, int B, int S, int M, int D, int L, int Lq, int P, float *out, void *stream);
int nmrf_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start, const double *loc,
                          const double *w, int B, int S, int M, int D, int L, int Lq, int P, double *out, void *stream);
int nmrf_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start, const float *loc,
                           const float *w, const float *grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                           float *grad_value, float *grad_loc, float *grad_w, void *stream);
int nmrf_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start, const double *loc,
                           const double *w, const double *grad_out, int B, int S, int M, int D, int L, int Lq, int P,
                           double *grad_value, double *grad_loc, double *grad_w, void *stream);

/* N3 / A7, A10, A13: one message-passing block's per-token linear algebra fused, on split-operand fp16 MFMA
 * (fp32-grade products, csrc/split_mfma.h).  Replaces, per block of the reference,
 *   x = x + proj(attn_out)                       (nmrf/models/NMP.py:100-106, 356-361, 566-571)
 *   x = x + mlp(norm2(x))      [has_mlp]         (NMP.py:337, 361-362, 537, 572-573; timm Mlp: fc1-GELU(erf)-fc2)
 *   qkv_next = [norm(x) | extra] . Wq^T + bq     (the NEXT block's q|k|v Linear on cat(norm1(x), side), NMP.py:90-96, 343-350, 544-556)
 *   ln_out   = norm(x)                           (the stage's final LayerNorm, NMP.py:658-659, 789-790, 891-892)
 * x, msg [T,128]; x_out [T,128]; q_out [T,NQ] (NQ % 128 == 0); ln_out [T,128]; any of the three outputs may be NULL (not all).
 * msg NULL: no projection stage (x1 = x).  KQ = operand width of the q stage: 0 (none), 128, 160 (LayerNorm | 32 side columns:
 * Fourier31 + one zero), 192 (LayerNorm | 64 context columns); side rows extra[t / extra_div, 0..KQ-128) with row stride
 * extra_ld (multiple of 4, rows 16-byte aligned).
 * 16 tokens per wave on v_mfma_f32_16x16x32_f16, two waves per SIMD (csrc/nmp_block16.hip).
 * stream_w: the block's weights as split-fp16 MFMA fragments in consumption order -- nmrf_pack_split_weight16_f32 pairs (16-row
 * strips x 32-deep chunks), 8 pairs = one 16 KB stage: proj (4 stages) | W1g[0] | W1g[1], W2g[0] | ... | W2g[15] (32 stages) | q
 * (NQ/128 * KQ/32 stages), where proj, every W1g[h] (strips 2h, 2h+1 of W1 [512,128]) and q list their strips two at a time,
 * interleaved chunk by chunk -- (s, c0) (s+1, c0) (s, c1) (s+1, c1) ... -- and W2g[h] = pairs (strip 0..7, chunk h) of W2 [128,512]:
 * the kernel feeds two adjacent pairs, which share their activation operand, to two accumulators with alternating MFMAs.
 * total_stages must equal the sum.  Biases / LayerNorm parameters are plain fp32 vectors.  inv_scales: HOST array of 4 floats,
 * 1 / scale the proj, fc1, fc2 and q weights were packed with (unused entries ignored).
 * ln_out_map (optional, device int32 [T]): row of ln_out that token t is written to, negative = dropped (the crop of the padded
 * token grid, NMP.py:786-788, 888-890).
 * attn_qkv (instead of msg; proj-only blocks, attn_n == 4, T % 4 == 0): [T, 384] q | k | v of the self-edge attention among the
 * attn_n sibling labels of a pixel (4 heads of 32, BasicAttention, NMP.py:90-108); the message softmax(q k^T / sqrt(32)) v is
 * computed on the way in (same arithmetic as nmrf_self_attn_f32). */
int nmrf_nmp_block16_f32(const float *x, const float *msg, const float *attn_qkv, int attn_n, const void *stream_w,
                         int total_stages, const float *bp,
                         const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                         const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld,
                         int extra_div, const float *bq, int has_mlp, int KQ, int NQ, int64_t T, const float *inv_scales,
                         float *x_out, float *q_out, float *ln_out, const int *ln_out_map, int kv16, int *range_flag, void *stream);

/* N3: a full block and the self-edge block that follows it in the layer sequence, in ONE launch (the pair of launches
 * nmrf_nmp_block16_f32(has_mlp, KQ 160, NQ 384) -> nmrf_nmp_block16_f32(attn_qkv = its q_out, KQ 160, NQ2); NMP.py:90-108, 337-364):
 * the first block's q | k | v never leave the registers, the 4 x 4 sibling attention runs on them, then the second parameter set
 * (bp2, lnq2_*, extra2, bq2) with the stages that follow in the same weight stream.  stream_w = the two launches' streams back to
 * back (51 + 4 + 5 NQ2 / 128 stages); inv_scales: HOST array of 6 floats (proj, fc1, fc2, q of the first block; proj, q of the second).
 * T a multiple of 4.  Outputs as the second launch's (x_out2, q_out2 [kv16_2], ln_out2 through ln_out2_map).  Same bits.
 * msg == NULL (with bp, ln2_*, b1, b2 NULL): the first block is a q stage alone -- the launch that opens the inference stage -- and
 * stream_w holds its 15 stages in front of the second block's. */
int nmrf_nmp_block16_pair_f32(const float *x, const float *msg, const void *stream_w, int total_stages, const float *bp,
                              const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                              const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld, int extra_div,
                              const float *bq, const float *bp2, const float *lnq2_g, const float *lnq2_b, float epsq2,
                              const float *extra2, int extra2_ld, int extra2_div, const float *bq2, int NQ2, int64_t T,
                              const float *inv_scales, float *x_out2, float *q_out2, float *ln_out2, const int *ln_out2_map,
                              int kv16_2, int *range_flag, void *stream);
/* kv16 != 0 (NQ == 384, q_out = q | k | v for nmrf_stripe_attn_f32 / nmrf_window_attn_f32 called with kv16 != 0): the k and v
 * thirds of a q_out row carry the split fp16 operand pairs the attention kernels contract (csrc/split_mfma.h: hi = rn_f16(x),
 * lo = rn_f16(x - hi)) instead of the floats -- the same 4 bytes per value, split once by the producer instead of by every query
 * tile that reads the row:
 *   floats 128 .. 255 (k): per 32-channel head h, bytes [128 h, 128 h + 64) = hi of channels 0 .. 31 (fp16), the next 64 = lo;
 *   floats 256 .. 383 (v): value c as the 32-bit word  hi | lo << 16.
 * q (floats 0 .. 127) stays fp32. */

/* Weight packing for nmrf_nmp_block16_f32: w [N,K] -> N/16 x Kp/32 pairs of 2 KB in [strip][chunk] order; lane (i = l & 15,
 * g = l >> 4) slot jj holds scale * w[16*strip + i][32*chunk + (jj&3) + 16*(jj>>2) + 4*g], zero beyond K; hi fragment then lo
 * fragment.  N % 16 == 0, Kp % 32 == 0. */
int nmrf_pack_split_weight16_f32(const float *w, int N, int K, int Kp, float scale, void *out, void *stream);

/* Self-test of the 16x16x32 split form: out[16*16] = A[16,K] . B[K,16] on one wave (K % 32 == 0). */
int nmrf_selftest_mfma16x16_f16split(const float *A, const float *Bm, int K, float *out, void *stream);

/* Per-token MLP chains of the hot path, one launch each (csrc/mlp_chain.hip), split-operand fp16 MFMA:
 *   kind 0  Inference.ffn / Refinement.ffn: timm Mlp(160,128,128), GELU             (NMP.py:675, 735-741, 839-844)
 *   kind 1  Propagation.cost_encoder + proj: Linear(36,128)-GELU-Linear(128,128); [. | Fourier31+0] -> Linear(159,128)   (NMP.py:607-612, 643-649)
 *   kind 2  prop_head / infer_head / refine_head: MLP(128,128,n_out <= 64, 3 layers, ReLU)   (DPN.py:65; NMRF.py:82,105; NMP.py:54-66)
 *   kind 3  infer_score_head: Linear(128, n_out <= 64)                              (NMRF.py:83)
 * in [T, in_ld] rows with K1 live columns (K1, in_ld multiples of 4); extra [T, extra_ld] (kind 1: the 32-float Fourier rows);
 * out rows of out_ld floats, n_out stored, at row out_map ? out_map[t] : t (negative: dropped).
 * stream_w: nmrf_pack_split_weight_f32 pairs of layer 1 [strip][chunk] (Kp = 16*ceil(K1/16); last layer rows zero-padded to a
 * multiple of 32), layer 2, layer 3, then zero pairs up to a multiple of 8 pairs; total_stages = pairs / 8.
 * b1/b2/b3 may be NULL.  inv_scales: HOST array of 3 floats. */
int nmrf_mlp_chain_f32(int kind, const float *in, int in_ld, int K1, const void *stream_w, int total_stages,
                       const float *b1, const float *b2, const float *b3, const float *extra, int extra_ld,
                       const float *inv_scales, int64_t T, float *out, int out_ld, int n_out, const int *out_map,
                       const float *row_add, int row_add_ld, int relu_out, int *range_flag, void *stream);
/* (row_add [T, row_add_ld >= n_out], optional: added to the stored columns; relu_out != 0: ReLU after that -- the label update
 *  labels = relu(prop_head(memory) + seeds) of nmrf/models/DPN.py:131-132 leaves with the head's rows.) */

/* Weight packing of the 32-row split-fp16 fragment streams (mlp_chain, conv kernels): w [N,K] row-major fp32 (an nn.Linear weight) -> N/32 x Kp/16 pairs of 2 KB in
 * [strip][chunk] order; a pair = [64 lanes][8 fp16] hi parts then the same for the lo parts (lo = fp16(w - hi), csrc/split_mfma.h);
 * lane (i = l & 31, h = l >> 5) slot jj holds scale * w[32*strip + i][16*chunk + (jj&3) + 8*(jj>>2) + 4*h], zero beyond K.
 * scale: a power of two that brings the largest |w| into [2^13, 2^14) (exact; the kernel multiplies the contraction by 1/scale):
 * the low parts of every weight are then normal fp16 numbers.
 * out: N * Kp * 4 bytes.  N % 32 == 0, Kp % 16 == 0, Kp >= K. */
int nmrf_pack_split_weight_f32(const float *w, int N, int K, int Kp, float scale, void *out, void *stream);

/* N2: statistics pass of InstanceNorm2d alone: ws [planes][ceil(HW/8192)][2] = per-chunk (mean, M2) of x [planes, HW]. */
int nmrf_instance_stats_f32(const float *x, int64_t planes, int64_t HW, float *ws, void *stream);
/* Apply pass alone on given statistics: y = [relu_out]( [relu_mid] IN(x; ws) + f(residual) ), f = identity or, with res_ws (the
 * statistics workspace of `residual`), IN(residual; res_ws) [+ ReLU if res_relu]: the residual operand may be a RAW convolution
 * output whose own InstanceNorm (+ ReLU) is still pending -- the stem output entering layer1.0, the 1x1 shortcut of a downsampling
 * block (nmrf/models/backbone.py:40-46, 70-72, 85) -- so that its own apply pass never runs.  x, residual, y: same 16-byte phase. */
int nmrf_instance_apply_f32(const float *x, const float *ws, const float *residual, const float *res_ws, int res_relu,
                            int64_t planes, int64_t HW, float eps, int relu_mid, int relu_out, float *y, void *stream);

/* N2: 1x1 convolution of a conv head with the InstanceNorm + ReLU in front of it folded into its operand load
 * (nmrf/models/NMRF.py:56-65 `concatconv` / `gw`, DPN.py:45-49 `proj`: Conv3x3 - IN - ReLU - Conv1x1):
 *   out[b,co,p] = sum_ci W[co,ci] * relu((x[b,c0+ci,p] - mean[b,c0+ci]) * rstd[b,c0+ci]) (+ bias[co])      (stats == NULL: plain x)
 * x [B,Cx,HW] NCHW (the 3x3 conv output; several heads may share it through c0), stats = nmrf_instance_stats_f32 workspace of x,
 * K in {64, 128} input channels, N % 64 == 0 output channels, out [B,N,HW].  stream_w = nmrf_pack_split_weight_f32(W[N,K], Kp = K)
 * pairs (strip-major), total_stages = N/32 * K/16 / 8, inv_scale = 1 / its scale.  Split-operand fp16 MFMA.
 * token_major != 0: out is written [B,HW,N] (a pixel's N channels contiguous) -- the layout nmrf_warp_corr_concat_f32 reads with
 * 16-byte loads and the layout of the DPN context rows (DPN.py:120: `context.permute(0, 2, 3, 1)`). */
int nmrf_conv1x1_in_relu_f32(const float *x, int B, int Cx, int64_t HW, int c0, int K, const float *stats, int chunks, float eps,
                             const void *stream_w, int total_stages, float inv_scale, const float *bias, int N, float *out,
                             int token_major, int *range_flag, void *stream);
/* The same kernel as a plain strided 1x1 convolution over [B,Cx,H,W] (the down-sampling shortcuts of the encoder,
 * nmrf/models/backbone.py:33-35: Conv2d(64, 96, 1, stride 2), Conv2d(96, 128, 1)): out[b,co,y,x] = sum_ci W[co,ci] x[b,c0+ci,y*s,x*s]
 * (+ bias) -> [B,N,Ho,Wo], Ho = (H-1)/s + 1.  K in {16..128} (multiple of 16), any N: stream_w = nmrf_pack_split_weight_f32 of W
 * zero-padded to [64*ceil(N/64), Kp] with Kp = 64 (K <= 64) or 128; total_stages = 2*ceil(N/64) * Kp/16 / 8.  stats (stride 1 only)
 * as in nmrf_conv1x1_in_relu_f32, which is this entry point with H*W = HW, stride 1. */
int nmrf_conv1x1_f32(const float *x, int B, int Cx, int H, int W, int stride, int c0, int K, const float *stats, int chunks,
                     float eps, const void *stream_w, int total_stages, float inv_scale, const float *bias, int N, float *out,
                     int token_major, int *range_flag, void *stream);

/* N2: 3x3 / stride 1 / pad 1 / no-bias convolution as a direct implicit GEMM on the split-operand fp16 MFMA, optionally with the
 * InstanceNorm + ReLU of its INPUT folded into the operand load (conv1 / conv2 of ResidualBlock, nmrf/models/backbone.py:38-46;
 * first conv of concatconv / gw, nmrf/models/NMRF.py:56-65):
 *   out[b,co,y,x] = sum_{ci,dy,dx} W[co,ci,dy,dx] * f(x[b,ci,y+dy-1,x+dx-1]),  f(v) = relu((v - mean[b,ci]) * rstd[b,ci]), or v when
 *   stats == NULL (stats = nmrf_instance_stats_f32 workspace of x; zero padding applies after f).
 * x [B,Ci,H,W], out [B,Co,H,W] NCHW fp32; Ci % 16 == 0 (<= 256 with stats); Co = groups * strips * 32, strips in {2,3,4}.
 * stream_w: nmrf_pack_split_weight_f32 of the matrix Wm[Co][9*Ci], Wm[co][((ci/16 * 3 + dy) * 3 + dx) * 16 + ci%16] = W[co,ci,dy,dx],
 * with its pairs reordered [group][chunk = 9*Ci/16][strip][512 x int32]; inv_scale = 1 / its scale. */
int nmrf_conv3x3_split_f32(const float *x, int B, int Ci, int H, int W, const float *stats, int chunks, float eps,
                           const void *stream_w, int strips, int groups, float inv_scale, int Co, float *out, int *range_flag, void *stream);

/* N2: the same kernel for other tap counts / strides of the backbone (nmrf/models/backbone.py:70,74):
 *   kt = 3, stride = 2, pad = 1: the 3x3 / stride-2 convolution of layer2.0 (strips in {2,3});
 *   kt = 4, stride = 1, pad = 2: a 4x4 convolution with padding 2 before / 1 after -- the 7x7 / stride-2 / pad-3 stem over the 2x2
 *                                space-to-depth image of nmrf_prep_images_s2d_f32 (Ci = 16, strips = 2).
 * x [B,Ci,H,W]; out [B,Co,Ho,Wo] with Ho = (H - 1) / stride + 1 (total padding kt - 1).  stream_w: as nmrf_conv3x3_split_f32 with
 * Wm[co][((ci/16 * kt + dy) * kt + dx) * 16 + ci%16] = W[co,ci,dy,dx] and 9 replaced by kt * kt. */
int nmrf_conv_split_f32(const float *x, int B, int Ci, int H, int W, const float *stats, int chunks, float eps, const void *stream_w,
                        int kt, int stride, int pad, int strips, int groups, float inv_scale, int Co, float *out, int *range_flag, void *stream);

/* Encoder input staging as nmrf_prep_images_f32, written as the 2x2 space-to-depth image: out [2B, 16, Hp/2, Wp/2], channel
 * c*4 + p*2 + q = normalised padded pixel (2Y+p, 2X+q) of colour c (3 colours), channels 12..15 zero.  Hp, Wp even. */
int nmrf_prep_images_s2d_f32(const float *img1, const float *img2, int B, int H, int W, int Hp, int Wp, float *out, void *stream);
/* The same staging from uint8 images (decoded PNGs, inference.py:66-70 read_gen -> np.uint8): the batched driver (N1) moves bytes
 * over PCIe and converts here; (float)px is exact, so the result is bit-identical to the _f32 entry points on img.float(). */
int nmrf_prep_images_s2d_u8(const uint8_t *img1, const uint8_t *img2, int B, int H, int W, int Hp, int Wp, float *out, void *stream);
int nmrf_prep_images_u8(const uint8_t *img1, const uint8_t *img2, int B, int C, int H, int W, int Hp, int Wp, float *out,
                        void *stream);

/* N1 host helper (no device work): memcpy into a pinned staging buffer with non-temporal stores, so that the H2D DMA that follows
 * does not have to snoop freshly written lines out of the CPU cache.  Any alignment; returns after an sfence. */
int nmrf_host_copy_nt(void *dst, const void *src, size_t bytes);
/* ... and out of one: memcpy(dst, src) followed by a cache-line flush of [src, src + bytes), so that the next D2H DMA into that
 * pinned buffer finds none of its lines in the CPU cache. */
int nmrf_host_read_evict(void *dst, const void *src, size_t bytes);

/* A1 + encoder input staging: replicate-pad both views right/bottom to (Hp, Wp) (InputPadder mode 'proposal',
 * nmrf/utils/frame_utils.py:268-275), stack them along the batch (NMRF.py:173) and normalise 2*(x/255)-1 (backbone.py:86).
 * img1, img2 [B,C,H,W] -> out [2B,C,Hp,Wp] (left views first). */
int nmrf_prep_images_f32(const float *img1, const float *img2, int B, int C, int H, int W, int Hp, int Wp, float *out,
                         void *stream);

/* Encoder tail (N2): y [planes = 2B*C, H, W] = the last 1x1 convolution WITHOUT its bias -> x = y + bias[c] and the 2x2 average
 * of x (backbone.py:96-98) in one pass.  H, W even.  bias may be NULL; x may be NULL: only the average is written (for a producer
 * that added its bias itself: y is then the finished 1/4-resolution map). */
int nmrf_bias_avgpool2_f32(const float *y, const float *bias, int64_t planes, int C, int H, int W, float *x, float *pooled,
                           void *stream);

/* Self-test: fills out[32*32] with the 32x32 product A*B computed by one wave of
 * v_mfma_f32_32x32x2_f32 (A [32,K], B [K,32] row-major, K even <= 64); pins the operand/accumulator
 * lane layout every attention kernel relies on. */
int nmrf_selftest_mfma_f32(const float *A, const float *Bm, int K, float *out, void *stream);

/* Self-test of the split-operand fp16 MFMA (csrc/split_mfma.h): out[32*32] = A*B from one wave of
 * v_mfma_f32_32x32x16_f16, A [32,K], B [K,32] row-major fp32, K % 16 == 0.  mode 0: full split product (hi/lo pairs,
 * three MFMAs per chunk, two accumulators); 1: hi parts only; 2: full split with the k slots in C/D order (the order
 * chained GEMMs use).  Pins the lane layout and the precision claim of every split-MFMA kernel. */
int nmrf_selftest_mfma_f16split(const float *A, const float *Bm, int K, int mode, float *out, void *stream);

/* Self-test of the LDS-DMA path (global_load_lds_dwordx4) the weight streams use: dst[t] = src[t ^ 65] within each
 * 256-float4 block, routed through LDS.  n_float4 % 256 == 0. */
int nmrf_selftest_lds_dma(const float *src, float *dst, int n_float4, void *stream);

/* Measurement hook (bench.py's `sustained_clock_ghz`): while buf != NULL, every block of the nmrf_nmp_block16_f32 /
 * nmrf_nmp_block16_pair_f32 launches of this process writes {shader-clock counter at entry, at exit, the chip's 100 MHz counter at
 * entry, at exit} to buf[4 * block ..] (device memory, uint64; launches of more than capacity_blocks blocks are not recorded; the grid
 * is min(tiles, CUs)).  sum(exit - entry of the first) / sum(of the second) x 100 MHz = the clock the CUs ran at under that kernel --
 * the peak figures of the roofline assume the 2.4 GHz boost clock.  buf == NULL switches the hook off (the default; a kernel argument
 * tested once per block).  No counterpart in the reference: its timing is wall-clock (nmrf/utils/evaluation.py:222-226). */
int nmrf_nmp_block16_clock_records(unsigned long long *buf, int capacity_blocks);

/* ---- N4, first slice: the pieces of a backward pass through the token-linear chains (csrc/backward.hip) -------------
 * The reference differentiates its whole forward with autograd (nmrf/models/NMRF.py:387-429 losses, main.py:413-430 step).  Here the
 * FORWARD of a chain / block is the product's fused launch and the backward is composed from these entry points by
 * nmrf_amd/models/autograd_ops.py (torch.autograd.Function): dgrad / wgrad of an nn.Linear as one strided GEMM on the split-operand
 * fp16 MFMA (fp32-grade products, fp32 accumulate), bias gradients as column sums, the activation derivatives, LayerNorm backward.
 * Every reduction over tokens is "per-part partial sums + nmrf_sum_partials_f32 in fixed order": deterministic.
 *
 * C[M,N] = op(A)[M,K] . op(B)[K,N]: element (i,k) of op(A) at A[i*sa_i + k*sa_k], element (k,j) of op(B) at B[k*sb_k + j*sb_j].
 *   dgrad   dx = dy . W       : A = dy [T,N] (sa_i = N, sa_k = 1), B = W [N,K] (sb_k = K, sb_j = 1)            (torch: grad_input of F.linear)
 *   wgrad   dW = dy^T . x     : A = dy (sa_i = 1, sa_k = N), B = x [T,K] (sb_k = K, sb_j = 1), reduction over the T tokens
 *   forward y  = x . W^T      : A = x [T,K] (sa_i = K, sa_k = 1), B = W (sb_k = 1, sb_j = K)                   (recomputation of saved-for-backward values)
 * splits > 1: the K range is cut into `splits` parts, part s writes its product to C + s*split_stride (>= M*ldc); the caller sums them.
 * a_amax: NULL, or a DEVICE float holding max |A|: A is then multiplied by the power of two that puts that maximum at [2^13, 2^14) before
 *   its fp16 split and the product scaled back -- for a GRADIENT operand (dy of a mean loss is ~1 / (B H W): unscaled, entries of 1e-6 ...
 *   1e-7 would lose most of their bits in the split); B (weights, activations) is used as it is.
 * range_flag: the fp16 range guard of the split operands (see the top of this header), may be NULL. */
int nmrf_gemm_split_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, int M, int N, int K,
                        float *C, int ldc, int splits, int64_t split_stride, const float *a_amax, int *range_flag, void *stream);
/* qkv16 [T,384] rows whose k | v thirds are split fp16 pairs (kv16 of nmrf_nmp_block16_f32) -> qkv [T,384] fp32 rows with k = hi + lo,
 * v = hi + lo (2^-22 relative of what the producer split): the operand of the attention backward kernels.  Both 16-byte aligned. */
int nmrf_from_kv16_f32(const float *qkv16, int64_t T, float *qkv, void *stream);
/* *out = max(*out, max_i |x[i]|) for a caller-ZEROED device float (the a_amax operand of nmrf_gemm_split_f32): one pass, one atomic per
 * block on the value's bit pattern -- order-independent, deterministic; NaNs are dropped (the consumer's range guard reports them).
 * x 16-byte aligned. */
int nmrf_absmax_f32(const float *x, int64_t n, float *out, void *stream);
/* out[i] = sum_{s < S} parts[s*stride + i] (i < n), added in a fixed order (four interleaved partial sums): deterministic. */
int nmrf_sum_partials_f32(const float *parts, int S, int64_t n, int64_t stride, float *out, void *stream);
/* The same in groups: out[g*n + i] = sum over the parts s in [g*group, min(S, (g+1)*group)) -- a caller with many parts reduces in rounds
 * (a fixed tree: deterministic) instead of one long serial chain per element. */
int nmrf_sum_partials_grouped_f32(const float *parts, int S, int64_t n, int64_t stride, int group, float *out, void *stream);
/* The same sum for MANY parts of a NARROW row in one launch: per column, 32 lanes add the parts p, p + 32, ... in ascending order and
 * the 32 lane sums are added in ascending order -- a fixed tree for a given S (deterministic; not the bits of the serial sum above). */
int nmrf_sum_partials_tree_f32(const float *parts, int S, int64_t n, int64_t stride, float *out, void *stream);
/* parts[b][n] = sum of x[t][n] over the rows_per_block rows of block b (b < ceil(T / rows_per_block)): the bias gradient's first pass. */
int nmrf_colsum_partials_f32(const float *x, int64_t T, int N, int rows_per_block, float *parts, void *stream);
/* pre_out = pre_in + bias (bias, pre_out may be NULL; pre_out may alias pre_in); act_out (may be NULL) = act(pre_in + bias):
 * act 0 identity, 1 ReLU, 2 GELU(erf) with the forward kernels' own gelu_fast (nn.ReLU / nn.GELU of NMP.py:54-66, timm Mlp). */
int nmrf_bias_act_f32(const float *pre_in, const float *bias, int64_t T, int N, int act, float *pre_out, float *act_out, void *stream);
/* dx = dy * act'(pre): act 1 ReLU (pre > 0), 2 GELU(erf): Phi(pre) + pre * phi(pre). */
int nmrf_act_bwd_f32(const float *pre, const float *dy, int64_t n, int act, float *dx, void *stream);
/* nn.LayerNorm over the last dimension, C in {64, 128, 256, 512} (NMP.py: every norm of the path has C = 128). */
int nmrf_layernorm_f32(const float *x, const float *g, const float *b, int64_t T, int C, float eps, float *y, void *stream);
/* its backward: dx [T,C]; part_dg / part_db [4*blocks][C]: per-wave partial sums of dy*xhat and dy (then nmrf_sum_partials_f32). */
int nmrf_layernorm_bwd_f32(const float *x, const float *g, const float *dy, int64_t T, int C, float eps, int blocks, float *dx,
                           float *part_dg, float *part_db, void *stream);

/* Backward of nmrf_window_attn_f32 on fp32 q | k | v rows (WindowAttention.forward, nmrf/models/NMP.py:185-289 -- what the reference's
 * autograd differentiates): dout [B,Hp,Wp,N,C] -> dqkv [B,Hp,Wp,N,3C] (every element written) and the gradient of the relative-position
 * table as one part per (image, window), dtab_parts [B * (Hp/win)*(Wp/win)][(2 win - 1)^2][3C] (sum with nmrf_sum_partials_f32: fixed
 * order).  scratch: 2 * B * heads * windows * (win*win*N)^2 floats (P and dS of every window).  win*win*N <= 256; same masks as the forward. */
int nmrf_window_attn_bwd_f32(const float *qkv, const float *table, const float *dout, int B, int Hp, int Wp, int N, int C, int heads,
                             int win, int shift, int sibling_mask, float *dqkv, float *dtab_parts, float *scratch, void *stream);

/* Backward of nmrf_self_attn_f32 (the per-pixel self-edge attention over the N sibling labels, BasicAttention.forward_pre, NMP.py:97-103):
 * qkv [T,3C] fp32, dout [T,C] -> dqkv [T,3C].  T % N == 0, N in {1, 2, 4}, heads*32 == C. */
int nmrf_self_attn_bwd_f32(const float *qkv, const float *dout, int64_t T, int N, int C, int heads, float *dqkv, void *stream);

/* Backward of nmrf_stripe_attn_f32 on fp32 q | k | v rows (CSWinAttention.forward + get_rpe, nmrf/models/NMP.py:429-505): dout
 * [B,H,W,N,128] -> dqkv [B,H,W,N,384] (every element written) and, per stripe, the gradients of the (previous, centre, next) LePE taps of
 * that axis' 64 channels: dtap_v_parts [B*W][64][3] (centre COLUMN of attns.0.get_v.weight), dtap_h_parts [B*H][64][3] (centre ROW of
 * attns.1.get_v.weight) -- sum the parts with nmrf_sum_partials_f32; the other six taps of either kernel have gradient 0 (they only see the
 * zero padding of a width-1 stripe).  scratch: 2 * B * 2 * max(W * (H N)^2, H * (W N)^2) floats.  The two axes run one after the other. */
int nmrf_stripe_attn_bwd_f32(const float *qkv, const float *lepe_v, const float *lepe_h, const float *dout, int B, int H, int W, int N,
                             float *dqkv, float *dtap_v_parts, float *dtap_h_parts, float *scratch, void *stream);

/* Backward pieces of the seed filter (DPN.mlp: three Conv1d(kernel 5, padding 2) along the disparity axis + softmax, DPN.py:32-38,117-119):
 * a Conv1d over D is an nn.Linear on 5-tap columns, col[(p, d)][c*5 + t] = A[(p, d + t - 2)][c] (zero outside the row) with the weight
 * [O][C][5] flattened, so dgrad / wgrad are nmrf_gemm_split_f32 on `col`.  nmrf_unfold5_f32 builds col [P*D, 5C] from rows (p, d) x C
 * (src_pcd == 0) or from the cost volume's [P][C][D] (src_pcd != 0); nmrf_fold5_f32 is its adjoint, dA[(p, e)][c] = sum_t dcol[(p, e - t +
 * 2)][c*5 + t]; nmrf_softmax_bwd_f32: dz = p (dp - sum_d p dp) over rows of D <= 64. */
int nmrf_unfold5_f32(const float *src, int64_t P, int C, int D, int src_pcd, float *col, void *stream);
int nmrf_fold5_f32(const float *dcol, int64_t P, int C, int D, float *dA, void *stream);
int nmrf_softmax_bwd_f32(const float *prob, const float *dprob, int64_t P, int D, float *dz, void *stream);

/* ---- N4, towards the feature maps: the backward of the three kernels that read the convolutional maps (the convolutions themselves --
 * encoder, matching heads, DPN context -- run on stock PyTorch-ROCm autograd in training mode: north_star keeps them there).
 * nmrf_cost_volume_bwd_f32: backward of nmrf_cost_volume_f32 (build_correlation_volume, nmrf/models/submodule.py:4-23): dcv [B*H*W, G, D]
 *   -> df1, df2 [B,C,H,W], gather form (one thread per map element, fixed summation order over d).
 * nmrf_seed_taps_bwd_f32: backward of the 9 cost taps of nmrf_seed_features_f32 (Propagation.sample_cost, NMP.py:619-634): dcost rows
 *   [(p, n)][ldc >= 9 G] -> dcv [P, G, D]; clamped taps that alias the same bin add up (what index_select's backward does).
 * nmrf_warp_corr_concat_bwd_f32: backward of nmrf_warp_corr_concat_f32 (Inference.sample_fmap / corr, NMP.py:683-741) for NCHW maps:
 *   drow [T, 2 Cf + Gr] -> df1, df2 [B,Cf,H,W], dg1, dg2 [B,Cg,H,W].  The labels are constants (NMP.py:694 builds the grid under no_grad;
 *   NMRF.py:215,232 detach them); the bilinear weights along x are the forward's, the ~1e-7 share of the neighbouring row (H6) is dropped. */
int nmrf_cost_volume_bwd_f32(const float *f1, const float *f2, const float *dcv, int B, int C, int H, int W, int D, int G, float *df1,
                             float *df2, void *stream);
int nmrf_seed_taps_bwd_f32(const float *dcost, const int64_t *seeds, int64_t P, int N, int G, int D, int ldc, float *dcv, void *stream);
int nmrf_warp_corr_concat_bwd_f32(const float *labels, const float *drow, const float *f1, const float *f2, const float *g1,
                                  const float *g2, int B, int H, int W, int N, int Cf, int Cg, int Gr, float *df1, float *df2, float *dg1,
                                  float *dg2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NMRF_HIP_H */
