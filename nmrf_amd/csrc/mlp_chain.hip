// Per-token MLP chains of the hot path in one kernel each, on split-operand fp16 MFMA (split_mfma.h, split_stream.h):
//
//     h1  = act1( in[t, 0:K1] . W1^T + b1 )                          K1 <= 160, N1 = 32 * N1S <= 128
//     h2  = act2( h1 . W2^T + b2 )                [L2]               128 -> 128
//     out = [h2 | extra[t, 0:32]] . W3^T + b3     [K3C > 0]          N3 = 32 * N3S <= 128
//
//   Inference.ffn / Refinement.ffn   timm Mlp(160 -> 128 -> 128, GELU)                              NMP.py:675, 735-741, 839-844
//   Propagation.cost_encoder + proj  Linear(36,128)-GELU-Linear(128,128); cat Fourier31; Linear(159,128)   NMP.py:607-612, 643-649
//   prop_head / infer_head / refine_head   MLP(128 -> 128 -> 128 -> 1 | 64 | 16, ReLU)               DPN.py:65, NMRF.py:82, 105; NMP.py:54-66
//   infer_score_head                 Linear(128, 64)                                                 NMRF.py:83
//
//   WTA (kind 4): infer_head + infer_score_head + the winner-take-all over a pixel's four labels + x2 + the 4 x 4 lower medians
//   (NMRF.py:218-232) in ONE launch: the score layer runs on the layer-1 operand right behind layer 1, and the 16 C/D registers of
//   (output strip s, lane half hi) of the disparity head ARE the 4 x 4 block (row half s, column half hi) of the token's 8 x 8
//   sub-pixel cell, so a lane selects per register across its quad (the four labels of its pixel) and takes the median of its own
//   registers -- no row of delta or score ever leaves the CU.
//
// replacing chains of token_linear (fp32 MFMA) / hipBLASLt / linear_smalln launches with their intermediates in HBM.  Same
// formulation as nmp_block.hip: transposed GEMMs (weights = A operand from the shared LDS stream, activations = B operand with
// the token on the lane), a layer's C/D registers are the next layer's B operand.  Rows leave through a wave-private LDS tile;
// `out_map` (optional) sends token t to output row out_map[t] (< 0: dropped): the zero-padded token grids of the window stages
// (NMP.py:745-762, 848-865) are written in place instead of F.pad'ing a dense result.
#include "split_stream.h"

#define MC_TOK 128
#define MC_OLD 132
#define MC_PF 4

struct ChainArgs {
    const float *in;
    int in_ld, K1;
    const void *stream;
    int total_stages;
    const float *b1, *b2, *b3;
    const float *extra;          // [T, extra_ld] side columns of layer 3 (first 32 used) or NULL
    int extra_ld;
    float *out;
    int out_ld, n_out;           // row stride and number of columns actually stored (<= 32 * last layer's strips)
    const int *out_map;
    int64_t T;
    int n_tiles;
    float inv1, inv2, inv3;
    int *range_flag;             // sticky fp16-range flag of the split operands (split_mfma.h), may be NULL
    const float *row_add;        // [T, row_add_ld] added to the stored columns (NULL: nothing) ...
    int row_add_ld, relu_out;    // ... followed by a ReLU if relu_out: labels = relu(prop_head(x) + seeds), DPN.py:131-132
    // WTA form only:
    const float *bs;             // bias of the score layer (64) or NULL
    float invs;                  // 1 / scale of its packed weights
    const float *labels;         // [T] disparity label of every token (tokens = pixels x 4 labels, pixel-major)
    int H, W;                    // the 1/8 grid: T = B * H * W * 4
    float *disp_curr;            // [B, 2H, 2W]
    // EPI form only (labels = disp_curr [T], H / W = the 1/4 grid, disp_curr = disp_pred [B, 4H, 4W]):
    int outH, outW;              // un-padded image size
    float *disp;                 // [B, outH, outW]
};

template <int ACT>
__device__ __forceinline__ float mc_act(float v) {
    if constexpr (ACT == 2) return gelu_fast(v);
    if constexpr (ACT == 1) return fmaxf(v, 0.f);
    return v;
}

// K1C: k chunks of layer 1; N1S: its output strips; L2: 128 -> 128 layer present; K3C: k chunks of layer 3 (0 = absent, 8 =
// h2 only, 10 = h2 | 32 side columns); N3S: its output strips.
// value of quad lane N (lanes 4p .. 4p+3 = the four labels of one pixel) in every lane of the quad
template <int N>
__device__ __forceinline__ float mc_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), N * 0x55, 0xf, 0xf, false));
}

// EPI (refine_head): the 16 outputs of a token are the 4 x 4 patch of its 1/4-resolution pixel -- relu(disp_curr + delta), pixel
// shuffle, x4 and crop (refine_epilogue_kernel of token.hip, NMRF.py:238-251) leave straight from the C/D registers: lane half hi
// holds patch rows hi and 2 + hi as registers 0-3 and 4-7.
template <int K1C, int N1S, int ACT1, bool L2, int ACT2, int K3C, int N3S, bool WTA = false, bool EPI = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mlp_chain_kernel(ChainArgs a) {
    static_assert(!EPI || (!WTA && K3C == 8 && N3S == 1), "EPI form: a head with one output strip");
    static_assert(!L2 || N1S == 4, "layer 2 consumes a 128-wide layer 1");
    static_assert(!WTA || (K1C == 8 && L2 && K3C == 8 && N3S == 2), "WTA form: 128 -> 128 -> 128 -> 64 head beside a 128 -> 64 score layer");
    static_assert(K3C == 0 || (L2 ? true : N1S == 4), "layer 3 consumes a 128-wide activation");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    float *Ot = reinterpret_cast<float *>(smem + SS_RING_BYTES) + wv * 32 * MC_OLD;          // wave-private [32][132]
    float *Par = reinterpret_cast<float *>(smem + SS_RING_BYTES + 4 * 32 * MC_OLD * 4);      // b1 | b2 | b3 (| bs), 128 floats each
    constexpr bool LAST_IS_L1 = !L2 && K3C == 0;                       // then b1 has n_out entries, not 32 * N1S
    for (int i = tid; i < 128; i += 256) {
        Par[i] = (a.b1 && i < (LAST_IS_L1 ? a.n_out : 32 * N1S)) ? a.b1[i] : 0.f;
        Par[128 + i] = (L2 && a.b2) ? a.b2[i] : 0.f;
        Par[256 + i] = (K3C > 0 && a.b3 && i < a.n_out) ? a.b3[i] : 0.f;
        if constexpr (WTA) Par[384 + i] = (a.bs && i < 64) ? a.bs[i] : 0.f;
    }
    auto par4 = [&](int off) { return *reinterpret_cast<const f32x4 *>(Par + off); };
    SplitStream<MC_PF> ss;
    ss.init(a.stream, smem, a.total_stages, tid);
    float guard = 0.f;                                                 // fp16 range guard of the activation splits

    constexpr int PS = WTA ? 16 : 0;                                   // the score layer's pairs, right behind layer 1's
    constexpr int P1 = K1C * N1S + PS, P2 = L2 ? 32 : 0, P3 = K3C * N3S;
    constexpr int PAIRS = P1 + P2 + P3, PADDED = (PAIRS + 7) / 8 * 8;
    constexpr int NLAST = K3C > 0 ? N3S : (L2 ? 4 : N1S);

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t t0 = (int64_t)tile * MC_TOK + wv * 32;
        const int64_t tq = t0 + j;
        const int64_t tc = tq < a.T ? tq : a.T - 1;
        // ---- layer 1: B operand straight from the input rows ---------------------------------------------------------------
        h16x8 bh[K1C > 10 ? K1C : 10], bl[K1C > 10 ? K1C : 10];
#pragma unroll
        for (int c = 0; c < K1C; ++c) {
            const int k0 = 16 * c + 4 * hi, k1 = k0 + 8;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v0 = k0 < a.K1 ? ldg4(a.in + tc * a.in_ld + k0) : z, v1 = k1 < a.K1 ? ldg4(a.in + tc * a.in_ld + k1) : z;
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            split8u_g(v, bh[c], bl[c], guard);
        }
        float h[4][16];
        ss_static_for<N1S>([&](auto ss_) {
            constexpr int st = decltype(ss_)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            ss_static_for<K1C>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                ss_pair<st * K1C + c>(ss, bh[c], bl[c], acc);
            });
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = par4(st * 32 + 8 * q + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[st][4 * q + e] = mc_act<ACT1>(fmaf(acc[4 * q + e], a.inv1, b4[e]));
            }
        });
        // ---- score layer (WTA form): the same operand, 128 -> 64 -----------------------------------------------------------------------
        float sc[WTA ? 2 : 1][16];
        if constexpr (WTA) {
            ss_static_for<2>([&](auto ss_) {
                constexpr int st = decltype(ss_)::value;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ss_static_for<8>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    ss_pair<K1C * N1S + st * 8 + c>(ss, bh[c], bl[c], acc);
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(384 + st * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sc[st][4 * q + e] = fmaf(acc[4 * q + e], a.invs, b4[e]);
                }
            });
        }
        // ---- layer 2 ---------------------------------------------------------------------------------------------------------------
        if constexpr (L2) {
#pragma unroll
            for (int c = 0; c < 8; ++c) split8u_g(&h[c >> 1][8 * (c & 1)], bh[c], bl[c], guard);
            ss_static_for<4>([&](auto ss_) {
                constexpr int st = decltype(ss_)::value;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ss_static_for<8>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    ss_pair<P1 + st * 8 + c>(ss, bh[c], bl[c], acc);
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(128 + st * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[st][4 * q + e] = mc_act<ACT2>(fmaf(acc[4 * q + e], a.inv2, b4[e]));
                }
            });
        }
        // ---- layer 3 ---------------------------------------------------------------------------------------------------------------
        if constexpr (K3C > 0) {
#pragma unroll
            for (int c = 0; c < 8; ++c) split8u_g(&h[c >> 1][8 * (c & 1)], bh[c], bl[c], guard);
            if constexpr (K3C > 8) {
#pragma unroll
                for (int c = 8; c < K3C; ++c) {
                    const float *e = a.extra + tc * a.extra_ld + 16 * (c - 8) + 4 * hi;
                    const float4 v0 = ldg4(e), v1 = ldg4(e + 8);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    split8u_g(v, bh[c], bl[c], guard);
                }
            }
            ss_static_for<N3S>([&](auto ss_) {
                constexpr int st = decltype(ss_)::value;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ss_static_for<K3C>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    ss_pair<P1 + P2 + st * K3C + c>(ss, bh[c], bl[c], acc);
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(256 + st * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[st][4 * q + e] = fmaf(acc[4 * q + e], a.inv3, b4[e]);
                }
            });
        }
        // the stream is padded with zero pairs to a whole stage: walk them so that the ring and the fragment queue stay in step
        if constexpr (PADDED > PAIRS) {
            f32x16 dump;
#pragma unroll
            for (int r = 0; r < 16; ++r) dump[r] = 0.f;
            ss_static_for<PADDED - PAIRS>([&](auto pp) { ss_pair<PAIRS + decltype(pp)::value>(ss, bh[0], bl[0], dump); });
        }
        if constexpr (WTA) {
            // ---- winner-take-all over the pixel's four labels, x2, 4 x 4 lower medians (wta_median_kernel of token.hip, same steps) ---------
            // register r of strip st, half hi  <->  sub-pixel (row 4 st + (r >> 2), column 4 hi + (r & 3)) of the token's 8 x 8 cell
            const float lab = a.labels[tc];
            float med[2];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                float val[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mine = fmaxf(lab + h[st][r], 0.f) * 2.0f;
                    float best_s = mc_quad<0>(sc[st][r]), v = mc_quad<0>(mine);
                    // torch.max returns the first maximal index; a NaN score also wins (ATen max semantics)
                    { const float s1 = mc_quad<1>(sc[st][r]), v1 = mc_quad<1>(mine);
                      const bool take = (s1 > best_s) || (isnan(s1) && !isnan(best_s)); best_s = take ? s1 : best_s; v = take ? v1 : v; }
                    { const float s2 = mc_quad<2>(sc[st][r]), v2 = mc_quad<2>(mine);
                      const bool take = (s2 > best_s) || (isnan(s2) && !isnan(best_s)); best_s = take ? s2 : best_s; v = take ? v2 : v; }
                    { const float s3 = mc_quad<3>(sc[st][r]), v3 = mc_quad<3>(mine);
                      const bool take = (s3 > best_s) || (isnan(s3) && !isnan(best_s)); best_s = take ? s3 : best_s; v = take ? v3 : v; }
                    val[r] = v;
                }
                // lower median of 16 = element of rank 7 (0-based) under a stable order; the block's sub-pixels in row-major order
                // are exactly r = 0 .. 15
                float m = val[0];
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    int rank = 0;
#pragma unroll
                    for (int c = 0; c < 16; ++c) rank += (val[c] < val[x]) || (val[c] == val[x] && c < x);
                    if (rank == 7) m = val[x];
                }
                med[st] = m;
            }
            if ((j & 3) == 0 && tq < a.T) {
                const int64_t pix = tq >> 2;                                   // (b * H + y) * W + x
                const int x = (int)(pix % a.W);
                const int64_t by = pix / a.W;                                  // b * H + y
                float *o = a.disp_curr + (by * 2) * (int64_t)(2 * a.W) + 2 * x + hi;
                o[0] = med[0];
                o[2 * a.W] = med[1];
            }
        } else if constexpr (EPI) {
            if (tq < a.T) {
                const float base = a.labels[tq];
                const int xq = (int)(tq % a.W);
                const int64_t byq = tq / a.W;                                  // b * H4 + yq
                const int bimg = (int)(byq / a.H), yq = (int)(byq - (int64_t)bimg * a.H);
                const int Wf = 4 * a.W;
#pragma unroll
                for (int half = 0; half < 2; ++half) {                         // registers 4 half .. 4 half + 3 = patch row hi + 2 half
                    const int Y = 4 * yq + hi + 2 * half, X = 4 * xq;
                    const f32x4 p = {fmaxf(base + h[0][4 * half], 0.f), fmaxf(base + h[0][4 * half + 1], 0.f),
                                     fmaxf(base + h[0][4 * half + 2], 0.f), fmaxf(base + h[0][4 * half + 3], 0.f)};
                    *reinterpret_cast<f32x4 *>(a.disp_curr + ((size_t)bimg * 4 * a.H + Y) * Wf + X) = p;
                    if (Y < a.outH) {
                        float *o = a.disp + ((size_t)bimg * a.outH + Y) * a.outW;
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (X + c < a.outW) o[X + c] = p[c] * 4.0f;
                    }
                }
            }
        } else {
        // ---- rows out ------------------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int st = 0; st < NLAST; ++st)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4 *>(Ot + j * MC_OLD + st * 32 + 8 * q + 4 * hi) =
                    f32x4{h[st][4 * q], h[st][4 * q + 1], h[st][4 * q + 2], h[st][4 * q + 3]};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int n4 = (a.n_out + 3) >> 2;                                     // float4 per output row
        for (int idx = lane; idx < 32 * n4; idx += 64) {
            const int row = idx / n4, c4 = idx - row * n4;
            const int64_t t = t0 + row;
            if (t >= a.T) continue;
            int64_t orow = t;
            if (a.out_map) { orow = a.out_map[t]; if (orow < 0) continue; }
            f32x4 v = *reinterpret_cast<const f32x4 *>(Ot + row * MC_OLD + 4 * c4);
            if (a.row_add) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * c4 + e < a.n_out) v[e] += a.row_add[(size_t)t * a.row_add_ld + 4 * c4 + e];
            }
            if (a.relu_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            float *dst = a.out + (size_t)orow * a.out_ld + 4 * c4;
            if (4 * c4 + 4 <= a.n_out && !(a.out_ld & 3)) *reinterpret_cast<f32x4 *>(dst) = v;
            else
                for (int e = 0; e < 4 && 4 * c4 + e < a.n_out; ++e) dst[e] = v[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }
    }
    split_guard_commit(guard, a.range_flag);
}

template <int K1C, int N1S, int ACT1, bool L2, int ACT2, int K3C, int N3S, bool WTA = false, bool EPI = false>
static int launch_chain(const ChainArgs &a, hipStream_t st) {
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)SS_RING_BYTES + 4 * 32 * MC_OLD * 4 + 512 * 4;
    auto kern = mlp_chain_kernel<K1C, N1S, ACT1, L2, ACT2, K3C, N3S, WTA, EPI>;
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMRF_ELAUNCH;
        n_cu_dev[dev] = prop.multiProcessorCount;
    }
    constexpr int PAIRS = K1C * N1S + (WTA ? 16 : 0) + (L2 ? 32 : 0) + K3C * N3S;
    if (a.total_stages != (PAIRS + 7) / 8) return NMRF_EINVAL;
    const int grid = a.n_tiles < n_cu_dev[dev] ? a.n_tiles : n_cu_dev[dev];
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

// kind: 0 ffn (K1 160 -> 128 GELU -> 128) | 1 seed embed (K1 36 -> 128 GELU -> 128, [h | extra32] -> 128) |
//       2 head (128 -> 128 ReLU -> 128 ReLU -> n_out <= 64) | 3 single Linear(128 -> n_out <= 64)
extern "C" int nmrf_mlp_chain_f32(int kind, const float *in, int in_ld, int K1, const void *stream_w, int total_stages,
                                  const float *b1, const float *b2, const float *b3, const float *extra, int extra_ld,
                                  const float *inv_scales, int64_t T, float *out, int out_ld, int n_out, const int *out_map,
                                  const float *row_add, int row_add_ld, int relu_out, int *range_flag, void *stream) {
    if (!in || !stream_w || !out || !inv_scales) return NMRF_ENULL;
    if (row_add && row_add_ld < n_out) return NMRF_EINVAL;
    if (T < 1 || ceil_div64(T, MC_TOK) > 0x7fffffff || in_ld < K1 || (in_ld & 3) || (K1 & 3) || n_out < 1 || out_ld < n_out)
        return NMRF_EINVAL;
    ChainArgs a{in, in_ld, K1, stream_w, total_stages, b1, b2, b3, extra, extra_ld, out, out_ld, n_out, out_map, T,
                (int)ceil_div64(T, MC_TOK), inv_scales[0], inv_scales[1], inv_scales[2], range_flag, row_add, row_add_ld, relu_out,
                nullptr, 0.f, nullptr, 0, 0, nullptr, 0, 0, nullptr};
    hipStream_t st = (hipStream_t)stream;
    switch (kind) {
        case 0:
            if (K1 > 160 || n_out != 128) return NMRF_EINVAL;
            return launch_chain<10, 4, 2, true, 0, 0, 0>(a, st);
        case 1:
            if (K1 > 48 || n_out != 128 || !extra || extra_ld < 32 || (extra_ld & 3)) return NMRF_EINVAL;
            return launch_chain<3, 4, 2, true, 0, 10, 4>(a, st);
        case 2:
            if (K1 != 128 || n_out > 64) return NMRF_EINVAL;
            return n_out > 32 ? launch_chain<8, 4, 1, true, 1, 8, 2>(a, st) : launch_chain<8, 4, 1, true, 1, 8, 1>(a, st);
        case 3:
            if (K1 != 128 || n_out > 64) return NMRF_EINVAL;
            return n_out > 32 ? launch_chain<8, 2, 0, false, 0, 0, 0>(a, st) : launch_chain<8, 1, 0, false, 0, 0, 0>(a, st);
        default: return NMRF_EINVAL;
    }
}

// A11 + A12 in one launch (the WTA form above): tgt [T,128] -> disp_curr [B, 2H, 2W].  stream_w: the pairs of W1 [128,128],
// Ws [64,128], W2 [128,128], W3 [64,128] in this order (12 stages); inv_scales: 1 / scale of W1, W2, W3, Ws.
extern "C" int nmrf_heads_wta_f32(const float *tgt, int B, int H, int W, int N, const void *stream_w, int total_stages, const float *b1,
                                  const float *b2, const float *b3, const float *bs, const float *inv_scales, const float *labels,
                                  float *disp_curr, int *range_flag, void *stream) {
    if (!tgt || !stream_w || !inv_scales || !labels || !disp_curr) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N != 4) return NMRF_EINVAL;           // (four labels per pixel: the quad of a wave)
    const int64_t T = (int64_t)B * H * W * N;
    if (ceil_div64(T, MC_TOK) > 0x7fffffff) return NMRF_EINVAL;
    ChainArgs a{tgt, 128, 128, stream_w, total_stages, b1, b2, b3, nullptr, 0, nullptr, 0, 64, nullptr, T, (int)ceil_div64(T, MC_TOK),
                inv_scales[0], inv_scales[1], inv_scales[2], range_flag, nullptr, 0, 0, bs, inv_scales[3], labels, H, W, disp_curr,
                0, 0, nullptr};
    return launch_chain<8, 4, 1, true, 1, 8, 2, true>(a, (hipStream_t)stream);
}

// A14 with its head (the EPI form above): tgt [T = B*H4*W4, 128] -> refine_head (MLP 128-128-128-16, ReLU) -> what
// nmrf_refine_epilogue_f32 does with its rows.  stream_w / biases / inv_scales: those of nmrf_mlp_chain_f32 kind 2 with n_out = 16.
extern "C" int nmrf_refine_head_epilogue_f32(const float *tgt, int B, int H4, int W4, const void *stream_w, int total_stages,
                                             const float *b1, const float *b2, const float *b3, const float *inv_scales,
                                             const float *disp_curr, int outH, int outW, float *disp_pred, float *disp, int *range_flag,
                                             void *stream) {
    if (!tgt || !stream_w || !inv_scales || !disp_curr || !disp_pred || !disp) return NMRF_ENULL;
    if (B < 1 || H4 < 1 || W4 < 1 || outH < 1 || outW < 1 || outH > 4 * H4 || outW > 4 * W4) return NMRF_EINVAL;
    const int64_t T = (int64_t)B * H4 * W4;
    if (ceil_div64(T, MC_TOK) > 0x7fffffff) return NMRF_EINVAL;
    ChainArgs a{tgt, 128, 128, stream_w, total_stages, b1, b2, b3, nullptr, 0, nullptr, 0, 16, nullptr, T, (int)ceil_div64(T, MC_TOK),
                inv_scales[0], inv_scales[1], inv_scales[2], range_flag, nullptr, 0, 0, nullptr, 0.f, disp_curr, H4, W4, disp_pred,
                outH, outW, disp};
    return launch_chain<8, 4, 1, true, 1, 8, 1, false, true>(a, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------------------
// weight packing: W [N,K] row-major fp32 -> N/32 x Kp/16 pairs of 2 KB in [strip][chunk] order; pair = [64 lanes][8 fp16] hi
// then the same for lo'; lane (i = l&31, h = l>>5) slot jj holds W[32*strip + i][16*chunk + split_kslot(jj, h)], 0 beyond K.
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_split_weight_kernel(const float *__restrict__ w, int N, int K, int KC, float scale,
                                                               uint4 *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;           // one lane of one pair
    const int64_t total = (int64_t)(N / 32) * KC * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t pair = idx >> 6;
    const int c = (int)(pair % KC), s = (int)(pair / KC);
    const int n = s * 32 + (lane & 31), h = lane >> 5;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int k = 16 * c + split_kslot(jj, h);
        v[jj] = k < K ? w[(int64_t)n * K + k] * scale : 0.f;
    }
    h16x8 vh, vl;
    split8u(v, vh, vl);
    out[pair * 128 + lane] = *reinterpret_cast<const uint4 *>(&vh);
    out[pair * 128 + 64 + lane] = *reinterpret_cast<const uint4 *>(&vl);
}

extern "C" int nmrf_pack_split_weight_f32(const float *w, int N, int K, int Kp, float scale, void *out, void *stream) {
    if (!w || !out) return NMRF_ENULL;
    if (N < 32 || (N & 31) || K < 1 || Kp < K || (Kp & 15) || !(scale > 0.f)) return NMRF_EINVAL;
    const int KC = Kp / 16;
    const int64_t total = (int64_t)(N / 32) * KC * 64;
    hipLaunchKernelGGL(pack_split_weight_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, N, K,
                       KC, scale, reinterpret_cast<uint4 *>(out));
    return nmrf_launch_status();
}

