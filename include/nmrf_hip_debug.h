/* libnmrf_hip_debug.so -- entry points that exist ONLY in the tools / test build of the library (python -m nmrf_amd.build --debug):
 * the round-1 / round-2 kernels that the product no longer launches, kept as reference paths for A/B parity runs
 * (NMRF_LINEAR=fp32: the per-token linears on the fp32 MFMA) and micro-benchmarks (tools/kernel_bench.py).  The debug library
 * exports everything include/nmrf_hip.h declares as well; the product library exports none of the symbols below. */
#ifndef NMRF_HIP_DEBUG_H
#define NMRF_HIP_DEBUG_H
#include "nmrf_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* N3  token linear with fused prologue / epilogue on fp32 MFMA:
 *     out[T,N] = act( P(x) . W^T + bias ) + residual
 *     P(x)[t]  = [ LayerNorm_128(x[t] + y[t]) | extra[t / extra_div][0..E) ]   when ln_gamma != NULL  (Cx == 128, K == 128 + E)
 *              = x[t][0..K)                                                     otherwise              (Cx == K, K % 4 == 0)
 * replaces, per message-passing block, nn.LayerNorm + torch.cat + nn.Linear (+ nn.GELU) (+ the residual add):
 * BasicAttention.forward_pre (nmrf/models/NMP.py:90-108), SwinNMP.forward_pre/get_qkv_input (:343-364),
 * CSWinNMP.forward_pre/get_qkv (:544-574), the timm Mlp fc1-GELU-fc2 (:337,537,675) and the proj layers.
 * y (optional, with x_out): pending residual; x + y is what is normalised and x_out receives it.
 * w_packed: the [N,K] weight in MFMA fragment order from nmrf_pack_linear_weight_f32 (N % 32 == 0; K zero-padded to
 * 32*ceil(K/32), supported ceil(K/32): 4,5,6 with LayerNorm; 1,2,4,5,16 without).
 * act: 0 identity, 1 ReLU, 2 GELU(erf).  bias [N] / residual [T,N] may be NULL. */
int nmrf_token_linear_f32(const float *x, const float *y, float *x_out, const float *ln_gamma, const float *ln_beta,
                          float eps, const float *extra, int E, int extra_div, const float *w_packed, const float *bias,
                          const float *residual, int act, int64_t T, int Cx, int K, int N, float *out, void *stream);
/* w [N,K] row-major -> packed [N/32][ceil(K/32)][4][64][4] floats (one contiguous 1 KiB line per wave load) followed by
 * N/32 int32: per 32-column strip the number of leading 32-wide k chunks holding a non-zero weight (the kernels skip the
 * rest: the v rows of a fused q|k|v weight are zero on the side-input columns).  Size: N*32*ceil(K/32) + N/32 words. */
int nmrf_pack_linear_weight_f32(const float *w, int N, int K, float *packed, void *stream);

/* N2  3x3 / stride 1 / pad 1 / no bias convolution, NCHW fp32, as fused Winograd F(2x2,3x3) on fp32 MFMA.
 * replaces nn.Conv2d(Ci, Co, 3, 1, 1, bias=False) of the stock conv band: backbone residual blocks
 * (nmrf/models/backbone.py:38-46), concatconv / gw (nmrf/models/NMRF.py:56-65), dpn.proj (nmrf/models/DPN.py:45-49).
 * x [B,Ci,H,W]; u_packed from nmrf_wino_pack_filter_f32 (16*Ci*Co floats); Ci % 16 == 0, Co % 32 == 0 -> y [B,Co,H,W]. */
int nmrf_conv3x3_wino_f32(const float *x, const float *u_packed, int B, int Ci, int H, int W, int Co, float *y,
                          void *stream);
/* w [Co,Ci,3,3] -> U = G w G^T in MFMA fragment order [Ci/16][Co/32][4][4][2][64][4]. */
int nmrf_wino_pack_filter_f32(const float *w, int Co, int Ci, float *packed, void *stream);

/* The 32-tokens-per-wave form of nmrf_nmp_block16_f32 (csrc/nmp_block.hip, v_mfma_f32_32x32x16_f16, one wave per SIMD): same
 * operator and arguments; stream_w from nmrf_pack_split_weight_f32 pairs (32-row strips x 16-deep chunks):
 *   proj: Wp [128,128] pairs (strip 0..3, chunk 0..7), 4 stages | mlp: W1[0] | W1[1], W2s[0] | ... | W2s[15] (W1[h] = pairs (h, 0..7),
 *   W2s[h] = pairs (strip n, chunk 2h + c), n = 0..3, c = 0..1), 32 stages | q: Wq [NQ,KQ] pairs strip-major, NQ/128 * KQ/32 stages. */
int nmrf_nmp_block_f32(const float *x, const float *msg, const void *stream_w, int total_stages, const float *bp,
                       const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                       const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld,
                       int extra_div, const float *bq, int has_mlp, int KQ, int NQ, int64_t T, const float *inv_scales,
                       float *x_out, float *q_out, float *ln_out, const int *ln_out_map, int *range_flag /* unused */, void *stream);

#ifdef __cplusplus
}
#endif
#endif
