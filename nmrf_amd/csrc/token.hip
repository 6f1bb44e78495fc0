// Per-token / per-pixel HBM-bound kernels (SURVEY section 8 rows A7/A10 prologue, A9, A10(i), A12, A14).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm(x) | extra  -> GEMM operand.  32 lanes per token (C == 128: one float4 per lane),
// two tokens per wave, statistics by half-wave shuffles.  Two-pass variance like ATen.
// ------------------------------------------------------------------------------------------------
// Optional fused residual: v = x + y is normalised and also written back to x_out (every residual add of
// the message-passing blocks is immediately followed by a LayerNorm of its result).
__global__ __launch_bounds__(256) void ln_concat_kernel(const float *__restrict__ x, const float *__restrict__ y,
        float *__restrict__ x_out, const float *__restrict__ gamma,
        const float *__restrict__ beta, float eps, const float *__restrict__ extra, int E, int extra_div, int64_t T,
        float *__restrict__ out, int ld) {
    const int sub = threadIdx.x & 31;
    const int64_t t = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    if (t >= T) return;                                  // whole 32-lane group exits together
    float4 v = ldg4(x + t * 128 + sub * 4);
    if (y) {
        const float4 r = ldg4(y + t * 128 + sub * 4);
        v = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
        stg4(x_out + t * 128 + sub * 4, v);
    }
    float s = (v.x + v.y) + (v.z + v.w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.0f / 128.0f);
    float4 d = make_float4(v.x - mean, v.y - mean, v.z - mean, v.w - mean);
    float q = (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 128.0f) + eps);
    float4 g = ldg4(gamma + sub * 4), bt = ldg4(beta + sub * 4);
    float4 r = make_float4(d.x * rstd * g.x + bt.x, d.y * rstd * g.y + bt.y, d.z * rstd * g.z + bt.z,
                           d.w * rstd * g.w + bt.w);
    float *o = out + t * ld;
    stg4(o + sub * 4, r);
    const float *e = extra ? extra + (t / extra_div) * E : nullptr;
    for (int j = 128 + sub; j < ld; j += 32) o[j] = (j - 128 < E) ? e[j - 128] : 0.f;
}

extern "C" int nmrf_ln_concat_f32(const float *x, const float *gamma, const float *beta, float eps, const float *extra,
                                  int E, int extra_div, int64_t T, int C, float *out, int ld, void *stream) {
    if (!x || !gamma || !beta || !out || (E > 0 && !extra)) return NMRF_ENULL;
    if (C != 128 || T < 1 || E < 0 || extra_div < 1 || ld < C + E || (ld & 3)) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64(T * 32, 256));
    hipLaunchKernelGGL(ln_concat_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (const float *)nullptr,
                       (float *)nullptr, gamma, beta, eps, extra, E, extra_div, T, out, ld);
    return nmrf_launch_status();
}

extern "C" int nmrf_add_ln_concat_f32(const float *x, const float *y, float *x_out, const float *gamma, const float *beta,
                                      float eps, const float *extra, int E, int extra_div, int64_t T, int C, float *out,
                                      int ld, void *stream) {
    if (!x || !y || !x_out || !gamma || !beta || !out || (E > 0 && !extra)) return NMRF_ENULL;
    if (C != 128 || T < 1 || E < 0 || extra_div < 1 || ld < C + E || (ld & 3)) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64(T * 32, 256));
    hipLaunchKernelGGL(ln_concat_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, x_out, gamma, beta, eps, extra, E,
                       extra_div, T, out, ld);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A10(i): self-edge attention among the N sibling labels of one pixel.
// Eight lanes = one (token, head): lane c holds channels 4c..4c+3 of the 32-wide head slice, so every load instruction of a
// wave reads 8 full 128-byte lines (8 (token, head) slices) -- the first version (one thread per (token, head), 72 float4 loads
// each touching 64 different lines for 16 bytes) was bound by cache-line look-ups at 2.5 TB/s.  q.k = 4 FMAs per lane + a
// three-step butterfly over the 8 lanes; the softmax is replicated; out = sum_j p_j v_j on the lane's own 4 channels.
// ------------------------------------------------------------------------------------------------
#define SA_MAXN 8
__global__ __launch_bounds__(256) void self_attn_kernel(const float *__restrict__ qkv, int64_t T, int N, int C, int heads,
                                                       float scale, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(i & 7);
    const int64_t grp = i >> 3;
    const bool ok = grp < T * heads;
    const int64_t gc = ok ? grp : T * heads - 1;                   // (idle lanes of the last wave shadow the last group: shuffles stay defined)
    const int64_t t = gc / heads;
    const int h = (int)(gc - t * heads);
    const int64_t t0 = (t / N) * N;
    const size_t ld = (size_t)3 * C;
    const float4 q = ldg4(qkv + t * ld + h * 32 + c * 4);
    float logit[SA_MAXN];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SA_MAXN; ++j) {
        if (j < N) {
            const float4 k = ldg4(qkv + (t0 + j) * ld + C + h * 32 + c * 4);
            float s = fmaf(q.w, k.w, fmaf(q.z, k.z, fmaf(q.y, k.y, q.x * k.x)));
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            logit[j] = s * scale;
            m = fmaxf(m, logit[j]);
        } else {
            logit[j] = -INFINITY;
        }
    }
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < SA_MAXN; ++j) { logit[j] = (j < N) ? expf(logit[j] - m) : 0.f; z += logit[j]; }
    const float rz = 1.0f / z;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < SA_MAXN; ++j) {
        if (j < N) {
            const float pj = logit[j] * rz;
            const float4 v = ldg4(qkv + (t0 + j) * ld + 2 * C + h * 32 + c * 4);
            o.x = fmaf(pj, v.x, o.x); o.y = fmaf(pj, v.y, o.y); o.z = fmaf(pj, v.z, o.z); o.w = fmaf(pj, v.w, o.w);
        }
    }
    if (ok) stg4(out + t * C + h * 32 + c * 4, o);
}

extern "C" int nmrf_self_attn_f32(const float *qkv, int64_t T, int N, int C, int heads, float *out, void *stream) {
    if (!qkv || !out) return NMRF_ENULL;
    if (T < 1 || N < 1 || N > SA_MAXN || T % N || heads < 1 || heads * 32 != C) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64(T * heads * 8, 256));
    hipLaunchKernelGGL(self_attn_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, T, N, C, heads,
                       1.0f / sqrtf(32.0f), out);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A9: warp right feature maps at x - label, group correlation, concat -> [T, 2Cf+groups].
// One block = one (b, y, 64-wide x tile); wave = label n (loops if N > 4), lane = x, so every
// per-channel NCHW read is coalesced for f1/g1 and near-coalesced for the warped f2/g2 taps.
// The sampling position reproduces grid_sample(align_corners=True)'s float round trip (H6).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float f4_get(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

struct WarpTaps {
    int off[4];     // y*W + x of the four taps (clamped in range)
    float w[4];     // weight, already zeroed for out-of-range taps
};

__device__ __forceinline__ WarpTaps make_taps(float label, int x, int y, int H, int W) {
    // every operation below is rounded on its own, as in ATen (left to -ffp-contract=fast the compiler fused `ix - floor(ix)` with
    // the product that makes ix, and (1 - wx1) * wy with its subtraction, differently from one kernel to the next)
#pragma clang fp contract(off)
    // reference grid (NMP.py:696-704): gx = 2*(x + (-d))/(W-1) - 1 ; gy = 2*(y + 0)/(H-1) - 1
    float gx = 2.0f * ((float)x + (-label)) / (float)(W - 1) - 1.0f;
    float gy = 2.0f * ((float)y + 0.0f) / (float)(H - 1) - 1.0f;
    // ATen vectorised CPU grid_sampler, align_corners=True: (g + 1) * ((size-1)/2)
    float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
    float iy = (gy + 1.0f) * ((float)(H - 1) / 2.0f);
    float x0f = floorf(ix), y0f = floorf(iy);
    float wx1 = ix - x0f, wy1 = iy - y0f;
    float wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    // floorf of a value up to ~1e9 would overflow int; labels are O(W) so clamp defensively
    x0f = fminf(fmaxf(x0f, -2.0f), (float)W);
    y0f = fminf(fmaxf(y0f, -2.0f), (float)H);
    int x0 = (int)x0f, y0 = (int)y0f;
    WarpTaps t;
    const float wy[2] = {wy0, wy1}, wx[2] = {wx0, wx1};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            int xx = x0 + dx, yy = y0 + dy;
            bool ok = xx >= 0 && xx < W && yy >= 0 && yy < H;
            int xc = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            int yc = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            t.off[dy * 2 + dx] = yc * W + xc;
            t.w[dy * 2 + dx] = ok ? wy[dy] * wx[dx] : 0.f;
        }
    return t;
}

// nw*v + ne*v + sw*v + se*v in ATen's order.  All four taps are ALWAYS loaded (their offsets are clamped into the map; a
// zero-weight tap contributes 0 * finite = exactly 0): unconditional loads let the compiler issue a whole channel batch as one
// clause -- with the former `if (w != 0) load` every tap was its own branch + wait and the kernel sat at 88 % SQ_WAIT_ANY.
// Products and sums are rounded separately, like ATen's vectorised CPU kernel (no FMA contraction: left to the compiler, each
// instantiation got its own mix of fused and unfused terms, and the NCHW and token-major kernels differed in the last bit).
__device__ __forceinline__ float warp_combine(const float (&v)[4], const WarpTaps &t) {
#pragma clang fp contract(off)
    float r = v[0] * t.w[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) r = r + v[k] * t.w[k];
    return r;
}

// Work split: lane = x (64 consecutive pixels of a row), and the channels of one token are divided over
// WC_CHUNKS threads (8 feature channels + 4 correlation groups each at the default 64/256/32 sizes), so a KITTI
// pair launches ~3.7k waves instead of ~460 and every thread issues ~100 independent loads.
#define WC_CHUNKS 8
__global__ __launch_bounds__(256, 5) void warp_corr_concat_kernel(const float *__restrict__ labels,
        const float *__restrict__ f1, const float *__restrict__ f2, const float *__restrict__ g1,
        const float *__restrict__ g2, int H, int W, int N, int Cf, int Cg, int groups, float *__restrict__ out, int ld) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware block order (workgroups go round-robin over the 8 XCDs): the N labels of a pixel read the same rows of all
    // four maps, so logical items are numbered label-fastest and every XCD gets a contiguous run of them -- with the
    // natural (x, y, b*N+n) order the labels of a row were spread over different L2s and each fetched the maps again.
    const int gx = (W + 63) / 64, gy = H * (WC_CHUNKS / 4);
    const int per_xcd = gridDim.x >> 3;                             // the launch pads the grid to a multiple of 8
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= gx * gy * (int)gridDim.y * N) return;
    const int n = item % N;
    const int bx = (item / N) % gx, by = (item / (N * gx)) % gy;
    const int b = blockIdx.y;
    const int x = bx * 64 + lane;
    const int y = by / (WC_CHUNKS / 4);
    const int chunk = (by % (WC_CHUNKS / 4)) * 4 + wv;
    if (x >= W) return;
    const size_t plane = (size_t)H * W;
    const int pix = y * W + x;
    const int cpg = Cg / groups;
    const int64_t t = (((int64_t)b * H + y) * W + x) * N + n;
    const WarpTaps tp = make_taps(labels[t], x, y, H, W);
    float *o = out + t * ld;
    const float *pf1 = f1 + (size_t)b * Cf * plane, *pf2 = f2 + (size_t)b * Cf * plane;
    const float *pg1 = g1 + (size_t)b * Cg * plane, *pg2 = g2 + (size_t)b * Cg * plane;
    const int fc = Cf / WC_CHUNKS;                          // feature channels per chunk (multiple of 4, host-checked)
    const int gc = groups / WC_CHUNKS;                      // correlation groups per chunk (multiple of 4)
    const float inv = 1.0f / (float)cpg;
    // The sampling row is the pixel's own row: gy -> iy reproduces y EXACTLY for ~4 of 5 rows (35 of 47 at KITTI 1/8 resolution),
    // and then the two taps of row y0 + 1 carry the weight wy1 * wx = 0 * wx = exactly 0 -- for every lane of the wave, which
    // shares y.  Such waves load two taps instead of four (0 * finite = 0 and r + 0 = r: bit-identical); the other rows, where
    // the float round trip lands an ulp beside y, keep all four (H6).  A wave-uniform choice, so both bodies stay straight-line.
    const bool two_taps = __builtin_amdgcn_ballot_w64(tp.w[2] != 0.f || tp.w[3] != 0.f) == 0;
    auto body = [&](auto two_c) {
        constexpr int NT = decltype(two_c)::value ? 2 : 4;
        // Loads first, arithmetic after, in batches of 4 channels = 4 + 4 NT independent loads in flight per batch and thread; a
        // channel row is `plane` floats from the next, the lane part of every address is one of five 32-bit offsets.
        for (int c = chunk * fc; c < (chunk + 1) * fc; c += 4) {
            float a[4], v[4][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float *p1 = pf1 + (size_t)(c + k) * plane, *p2 = pf2 + (size_t)(c + k) * plane;      // uniform
                a[k] = p1[pix];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[k][q] = q < NT ? p2[tp.off[q]] : 0.f;
            }
            float w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = warp_combine(v[k], tp);
            stg4(o + c, make_float4(a[0], a[1], a[2], a[3]));
            stg4(o + Cf + c, make_float4(w[0], w[1], w[2], w[3]));
        }
        for (int g = chunk * gc; g < (chunk + 1) * gc; g += 4) {
            float r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float s = 0.f;
                if (cpg == 8) {                             // every shipped config: 256 channels in 32 groups -- 8 + 8 NT loads per group
                    float a[8], v[8][4];
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const size_t ch = (size_t)((g + k) * 8 + c) * plane;
                        a[c] = pg1[ch + pix];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[c][q] = q < NT ? (pg2 + ch)[tp.off[q]] : 0.f;
                    }
#pragma unroll
                    for (int c = 0; c < 8; ++c) s = fmaf(a[c], warp_combine(v[c], tp), s);
                } else {
                    for (int c = 0; c < cpg; ++c) {
                        const size_t ch = (size_t)((g + k) * cpg + c) * plane;
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = q < NT ? (pg2 + ch)[tp.off[q]] : 0.f;
                        s = fmaf(pg1[ch + pix], warp_combine(v, tp), s);
                    }
                }
                r[k] = s * inv;
            }
            stg4(o + 2 * Cf + g, make_float4(r[0], r[1], r[2], r[3]));
        }
    };
    if (two_taps) body(std::true_type{});
    else body(std::false_type{});
}

// The same operation on TOKEN-MAJOR maps ([B, HW, C]: what nmrf_conv1x1_in_relu_f32 writes with token_major = 1), for the shipped
// 64 feature channels / 256 correlation channels in 32 groups.  32 lanes = one token: lane l owns correlation group l (its 8
// channels are two 16-byte pieces of a pixel's row) and, for l < 16, feature channels 4l .. 4l+3; every tap is one contiguous row,
// so a token costs 9 (two taps) or 15 16-byte loads per lane, all in flight at once, instead of 120-216 dword gathers from 320
// channel planes.  Same taps, same products, same order of additions as the NCHW kernel: identical bits.
__global__ __launch_bounds__(256) void warp_corr_concat_tok_kernel(const float *__restrict__ labels,
        const float *__restrict__ f1, const float *__restrict__ f2, const float *__restrict__ g1,
        const float *__restrict__ g2, int H, int W, int N, unsigned T, float *__restrict__ out, int ld, float normalizer,
        float *__restrict__ enc, int enc_ld, const int *__restrict__ enc_map) {
    constexpr int Cf = 64, Cg = 256;
    const int sub = threadIdx.x & 31;
    const unsigned t_raw = blockIdx.x * 8u + (threadIdx.x >> 5);
    const bool live = t_raw < T;
    const unsigned t = live ? t_raw : T - 1;
    const unsigned hw = (unsigned)(H * W);
    const unsigned pixn = t / (unsigned)N;                                 // b * HW + pix
    const unsigned b = pixn / hw, pix = pixn - b * hw;
    const int y = (int)(pix / (unsigned)W), x = (int)(pix - (unsigned)y * W);
    const float label = labels[t];
    const WarpTaps tp = make_taps(label, x, y, H, W);
    // enc != NULL: the Fourier embedding of the label (the side input of the stage's blocks) leaves with the token -- lanes 0..15 are
    // the 16 work items of fourier_embed_kernel (seed.hip), same arithmetic, same row map; one launch less per stage
    if (enc && live && sub < 16) {
        const int64_t row = enc_map ? (int64_t)enc_map[t] : (int64_t)t;
        if (row >= 0) {
            fourier_write(label, normalizer, sub, enc + row * enc_ld);
            fourier_pad(sub, enc + row * enc_ld, enc_ld);
        }
    }
    const float *pf1 = f1 + (size_t)pixn * Cf + 4 * sub, *pf2 = f2 + (size_t)b * hw * Cf + 4 * sub;
    const float *pg1 = g1 + (size_t)pixn * Cg + 8 * sub, *pg2 = g2 + (size_t)b * hw * Cg + 8 * sub;
    const bool feat = sub < 16;
    const bool two_taps = __builtin_amdgcn_ballot_w64(tp.w[2] != 0.f || tp.w[3] != 0.f) == 0;   // (see the NCHW kernel)
    auto body = [&](auto two_c) {
        constexpr int NT = decltype(two_c)::value ? 2 : 4;
        float4 ga[2], gv[NT][2], fa = make_float4(0.f, 0.f, 0.f, 0.f), fv[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) ga[i] = ldg4(pg1 + 4 * i);
#pragma unroll
        for (int q = 0; q < NT; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i) gv[q][i] = ldg4(pg2 + (size_t)tp.off[q] * Cg + 4 * i);
            fv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (feat) {
            fa = ldg4(pf1);
#pragma unroll
            for (int q = 0; q < NT; ++q) fv[q] = ldg4(pf2 + (size_t)tp.off[q] * Cf);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = q < NT ? f4_get(gv[q < NT ? q : 0][c >> 2], c & 3) : 0.f;
            s = fmaf(f4_get(ga[c >> 2], c & 3), warp_combine(v, tp), s);
        }
        float w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = q < NT ? f4_get(fv[q < NT ? q : 0], k) : 0.f;
            w[k] = warp_combine(v, tp);
        }
        if (live) {
            float *o = out + (size_t)t * ld;
            if (feat) {
                stg4(o + 4 * sub, fa);
                stg4(o + Cf + 4 * sub, make_float4(w[0], w[1], w[2], w[3]));
            }
            o[2 * Cf + sub] = s * (1.0f / 8.0f);
        }
    };
    if (two_taps) body(std::true_type{});
    else body(std::false_type{});
}

extern "C" int nmrf_warp_corr_concat_fourier_f32(const float *labels, const float *f1, const float *f2, const float *g1,
                                                 const float *g2, int B, int H, int W, int N, int Cf, int Cg, int groups,
                                                 float *out, int ld, int token_major, float normalizer, float *enc, int enc_ld,
                                                 const int *enc_map, void *stream) {
    if (!labels || !f1 || !f2 || !g1 || !g2 || !out) return NMRF_ENULL;
    if (enc && (!token_major || enc_ld < 31)) return NMRF_EINVAL;          // the embedding rides with the token-major kernel only
    if (token_major) {
        const int64_t T = (int64_t)B * H * W * N;
        if (B < 1 || H < 2 || W < 2 || N < 1 || Cf != 64 || Cg != 256 || groups != 32 || ld < 2 * Cf + groups || (ld & 3) ||
            T >= ((int64_t)1 << 31))
            return NMRF_EINVAL;
        hipLaunchKernelGGL(warp_corr_concat_tok_kernel, dim3((unsigned)ceil_div64(T, 8)), dim3(256), 0, (hipStream_t)stream, labels,
                           f1, f2, g1, g2, H, W, N, (unsigned)T, out, ld, normalizer, enc, enc_ld, enc_map);
        return nmrf_launch_status();
    }
    if (B < 1 || H < 2 || W < 2 || N < 1 || Cf < 4 * WC_CHUNKS || Cf % (4 * WC_CHUNKS) || groups < 4 * WC_CHUNKS ||
        groups % (4 * WC_CHUNKS) || Cg % groups || ld < 2 * Cf + groups || (ld & 3) || B > 65535)
        return NMRF_EINVAL;
    const int64_t items = (int64_t)((W + 63) / 64) * H * (WC_CHUNKS / 4) * N;      // per image
    if (items > 0x7ffffff0) return NMRF_EINVAL;
    dim3 grid((unsigned)((items + 7) / 8 * 8), B);
    hipLaunchKernelGGL(warp_corr_concat_kernel, grid, dim3(256), 0, (hipStream_t)stream, labels, f1, f2, g1, g2, H, W, N,
                       Cf, Cg, groups, out, ld);
    return nmrf_launch_status();
}

extern "C" int nmrf_warp_corr_concat_f32(const float *labels, const float *f1, const float *f2, const float *g1,
                                         const float *g2, int B, int H, int W, int N, int Cf, int Cg, int groups,
                                         float *out, int ld, int token_major, void *stream) {
    return nmrf_warp_corr_concat_fourier_f32(labels, f1, f2, g1, g2, B, H, W, N, Cf, Cg, groups, out, ld, token_major, 0.f, nullptr, 0,
                                             nullptr, stream);
}

// ------------------------------------------------------------------------------------------------
// A11/A12: relu(label+delta) -> WTA over N by score (first max wins) -> x2 -> 4x4 lower median.
// One thread = one 1/4-res output pixel = one 4x4 block of full-res sub-pixels, all inside a single
// 1/8 cell, so it reads 4 float4 rows of delta and of score per label.
// ------------------------------------------------------------------------------------------------
#define WTA_MAXN 8
// NS > 0: N == NS known at compile time -- all 9 NS loads of the thread are requested before the first comparison (the runtime loop
// waited for every label's rows in turn: 4 dependent round trips on a launch of 115 blocks that is pure latency)
template <int NS>
__global__ __launch_bounds__(256) void wta_median_kernel(const float *__restrict__ delta, const float *__restrict__ score,
        const float *__restrict__ labels, int B, int H, int W, int N, float *__restrict__ disp_curr) {
    const int H4 = 2 * H, W4 = 2 * W;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H4 * W4) return;
    const int xq = (int)(i % W4), yq = (int)((i / W4) % H4), b = (int)(i / ((int64_t)W4 * H4));
    const int y = yq >> 1, x = xq >> 1;
    const int64_t t0 = (((int64_t)b * H + y) * W + x) * N;
    const int j0 = (4 * (yq & 1)) * 8 + 4 * (xq & 1);     // sub-pixel index hs*8+ws of the block's corner
    float best_s[16], val[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { best_s[k] = -INFINITY; val[k] = 0.f; }
    auto take_label = [&](int n, float lab, const float4 (&d4)[4], const float4 (&s4)[4]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dd[4] = {d4[r].x, d4[r].y, d4[r].z, d4[r].w}, ss[4] = {s4[r].x, s4[r].y, s4[r].z, s4[r].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = r * 4 + c;
                // torch.max returns the first maximal index; a NaN score also wins (ATen max semantics)
                bool take = (n == 0) || (ss[c] > best_s[k]) || (isnan(ss[c]) && !isnan(best_s[k]));
                if (take) { best_s[k] = ss[c]; val[k] = fmaxf(lab + dd[c], 0.f) * 2.0f; }
            }
        }
    };
    if constexpr (NS > 0) {
        float lab[NS];
        float4 d4[NS][4], s4[NS][4];
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            lab[n] = labels[t0 + n];
            const float *dp = delta + (t0 + n) * 64 + j0, *sp = score + (t0 + n) * 64 + j0;
#pragma unroll
            for (int r = 0; r < 4; ++r) { d4[n][r] = ldg4(dp + r * 8); s4[n][r] = ldg4(sp + r * 8); }
        }
#pragma unroll
        for (int n = 0; n < NS; ++n) take_label(n, lab[n], d4[n], s4[n]);
    } else {
        for (int n = 0; n < N; ++n) {
            const float lab = labels[t0 + n];
            const float *dp = delta + (t0 + n) * 64 + j0, *sp = score + (t0 + n) * 64 + j0;
            float4 d4[4], s4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { d4[r] = ldg4(dp + r * 8); s4[r] = ldg4(sp + r * 8); }
            take_label(n, lab, d4, s4);
        }
    }
    // lower median of 16 = element of rank 7 (0-based) under a stable order
    float med = val[0];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        int rank = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) rank += (val[c] < val[a]) || (val[c] == val[a] && c < a);
        if (rank == 7) med = val[a];
    }
    disp_curr[i] = med;
}

extern "C" int nmrf_wta_median_f32(const float *delta, const float *score, const float *labels, int B, int H, int W,
                                   int N, float *disp_curr, void *stream) {
    if (!delta || !score || !labels || !disp_curr) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N < 1 || N > WTA_MAXN) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64((int64_t)B * 4 * H * W, 256));
    if (N == 4)
        hipLaunchKernelGGL(wta_median_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, delta, score, labels, B, H, W, N, disp_curr);
    else
        hipLaunchKernelGGL(wta_median_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, delta, score, labels, B, H, W, N, disp_curr);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A14: relu(disp_curr + delta) -> 4x4 pixel shuffle -> disp_pred ; x4 + crop -> disp.
// One thread = one 1/4-res cell: reads 16 deltas (4 float4), writes a 4x4 patch (row-wise float4).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void refine_epilogue_kernel(const float *__restrict__ delta,
        const float *__restrict__ disp_curr, int B, int H4, int W4, int outH, int outW, float *__restrict__ disp_pred,
        float *__restrict__ disp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H4 * W4) return;
    const int xq = (int)(i % W4), yq = (int)((i / W4) % H4), b = (int)(i / ((int64_t)W4 * H4));
    const float base = disp_curr[i];
    const int Hf = 4 * H4, Wf = 4 * W4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float4 d4 = ldg4(delta + i * 16 + r * 4);
        float4 p = make_float4(fmaxf(base + d4.x, 0.f), fmaxf(base + d4.y, 0.f), fmaxf(base + d4.z, 0.f),
                               fmaxf(base + d4.w, 0.f));
        const int Y = 4 * yq + r, X = 4 * xq;
        stg4(disp_pred + ((size_t)b * Hf + Y) * Wf + X, p);
        if (Y < outH) {
            const float pv[4] = {p.x, p.y, p.z, p.w};
            float *o = disp + ((size_t)b * outH + Y) * outW;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (X + c < outW) o[X + c] = pv[c] * 4.0f;
        }
    }
}

extern "C" int nmrf_refine_epilogue_f32(const float *delta, const float *disp_curr, int B, int H4, int W4, int outH,
                                        int outW, float *disp_pred, float *disp, void *stream) {
    if (!delta || !disp_curr || !disp_pred || !disp) return NMRF_ENULL;
    if (B < 1 || H4 < 1 || W4 < 1 || outH < 1 || outW < 1 || outH > 4 * H4 || outW > 4 * W4) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64((int64_t)B * H4 * W4, 256));
    hipLaunchKernelGGL(refine_epilogue_kernel, grid, dim3(256), 0, (hipStream_t)stream, delta, disp_curr, B, H4, W4, outH,
                       outW, disp_pred, disp);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Narrow-output linear layers (prediction heads: N = 1, 16, 64 outputs from K = 128 features).
// hipBLASLt picks 64x64 tiles for these and lands at ~70 us (profiles/r01e); they are HBM-bound
// ([T,128] in, [T,N] out).  One block = 32 tokens x 8 output groups: the x tile and W are staged in LDS (x rows padded
// to K+4 floats -> conflict-free b128 reads), thread (token, output group) accumulates NPT outputs.
// ------------------------------------------------------------------------------------------------
#define SL_TOK 32
template <int NPT>      // outputs per thread; the block covers 8*NPT outputs
__global__ __launch_bounds__(256) void linear_smalln_kernel(const float *__restrict__ x, const float *__restrict__ w,
        const float *__restrict__ bias, int64_t T, int K, int N, int act, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ldx = K + 4;
    float *sx = sm;                          // [SL_TOK][K+4]
    float *sw = sm + SL_TOK * ldx;           // [8*NPT][K]   (rows >= N are zero)
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * SL_TOK;
    const int k4n = K >> 2;
    for (int i = tid; i < SL_TOK * k4n; i += 256) {
        const int r = i / k4n, c = i - r * k4n;
        const int64_t t = t0 + r;
        float4 v = t < T ? ldg4(x + t * K + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(sx + r * ldx + 4 * c) = v;
    }
    for (int i = tid; i < 8 * NPT * k4n; i += 256) {
        const int r = i / k4n, c = i - r * k4n;
        float4 v = r < N ? ldg4(w + (size_t)r * K + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4 *>(sw + r * K + 4 * c) = v;
    }
    __syncthreads();
    const int tok = tid & 31, og = tid >> 5;                 // half-wave = output group (W addresses broadcast)
    float acc[NPT];
#pragma unroll
    for (int o = 0; o < NPT; ++o) acc[o] = 0.f;
    const float *xr = sx + tok * ldx;
    const float *wr = sw + (size_t)og * NPT * K;
    for (int c = 0; c < k4n; ++c) {
        const float4 xv = *reinterpret_cast<const float4 *>(xr + 4 * c);
#pragma unroll
        for (int o = 0; o < NPT; ++o) {
            const float4 wv = *reinterpret_cast<const float4 *>(wr + o * K + 4 * c);
            acc[o] = fmaf(xv.x, wv.x, acc[o]); acc[o] = fmaf(xv.y, wv.y, acc[o]);
            acc[o] = fmaf(xv.z, wv.z, acc[o]); acc[o] = fmaf(xv.w, wv.w, acc[o]);
        }
    }
    const int64_t t = t0 + tok;
    if (t >= T) return;
#pragma unroll
    for (int o = 0; o < NPT; ++o) {
        const int n = og * NPT + o;
        if (n < N) {
            float v = acc[o] + (bias ? bias[n] : 0.f);
            out[t * N + n] = act == 1 ? fmaxf(v, 0.f) : v;
        }
    }
}

extern "C" int nmrf_linear_smalln_f32(const float *x, const float *w, const float *bias, int64_t T, int K, int N, int act,
                                      float *out, void *stream) {
    if (!x || !w || !out) return NMRF_ENULL;
    if (T < 1 || K < 4 || (K & 3) || K > 512 || N < 1 || N > 64 || act < 0 || act > 1) return NMRF_EINVAL;
    const int npt = N <= 8 ? 1 : (N <= 16 ? 2 : 8);
    const size_t smem = ((size_t)SL_TOK * (K + 4) + (size_t)8 * npt * K) * sizeof(float);
    dim3 grid((unsigned)ceil_div64(T, SL_TOK));
    hipStream_t st = (hipStream_t)stream;
    if (smem > 64 * 1024) return NMRF_EINVAL;
    if (npt == 1) hipLaunchKernelGGL(linear_smalln_kernel<1>, grid, dim3(256), smem, st, x, w, bias, T, K, N, act, out);
    else if (npt == 2) hipLaunchKernelGGL(linear_smalln_kernel<2>, grid, dim3(256), smem, st, x, w, bias, T, K, N, act, out);
    else hipLaunchKernelGGL(linear_smalln_kernel<8>, grid, dim3(256), smem, st, x, w, bias, T, K, N, act, out);
    return nmrf_launch_status();
}
