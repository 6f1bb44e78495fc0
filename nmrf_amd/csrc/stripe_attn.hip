// A7: cross-stripe (CSWin, split_size 1) attention with LePE on fp32 matrix cores.
//
// One wave = one 32-query tile of one (image b, stripe, head) and one of KSPLIT key ranges.  K and V
// operand fragments are read straight from the token-major qkv tensor (each token's 32-wide head slice
// is one 128-byte line, shared through L1/L2 by the waves of the block, which walk the same stripe).
// Flash-style streaming softmax, so the [T,T] matrix never exists (H9).
//
// Matrix-core formulation (v_mfma_f32_32x32x2_f32, exact fp32, 16 k-steps per 32-deep contraction):
//   S^T[key][q]  = sum_c K[key][c] * Q[q][c]          A = K fragment, B = Q fragment
//   O^T[d][q]   += sum_key V[key][d] * P[q][key]      A = V fragment, B = P (the S^T accumulator itself)
// With the transposed forms the softmax statistics of query q live in lane q (both half-waves),
// and the S^T accumulator registers ARE the B operand of the second product: MFMA k-slot (step s,
// half h) <-> key mfma_row(s,h), which is exactly where S^T register s of half h sits.
// Channel <-> k-slot map for the first product is (step s, half h) <-> c = 16h + s, so each lane
// loads 16 contiguous floats of its row.
//
// Parallelism (round-1 PMC profile: 1880 waves of 20 serial key tiles each at KITTI batch 1, every wave
// alone on its SIMD for 184k cycles): the key range of a query tile is split over KSPLIT waves of the
// same block (flash-decoding style) and the partial (m, l, O) triples are merged through LDS; the
// host picks KSPLIT so that the launch has a few thousand waves.  Masks are evaluated only where they
// can fire (sibling mask on the diagonal tile, tail mask on the last tile); exp2 with log2(e) folded
// into the q scale; the next K/V fragments are fetched while the current tile computes.
#include "common.h"

#define SA_TILE 32
#define SA_LOG2E 1.4426950408889634f

struct StripeGeom {
    int H, W, N, C;        // grid, labels per pixel, embed dim (128)
    int L;                 // pixels per stripe
    int Ts;                // tokens per stripe = L*N
    int64_t pix_stride;    // token-pixel stride between consecutive stripe positions (W or 1)
};

// NSHIFT >= 0: N == 1<<NSHIFT at compile time (N=4 in every shipped config); NSHIFT < 0: runtime N
template <int NSHIFT>
__device__ __forceinline__ int div_n(const StripeGeom &g, int s) { return NSHIFT >= 0 ? (s >> NSHIFT) : (s / g.N); }

template <int NSHIFT>
__device__ __forceinline__ int64_t stripe_row(const StripeGeom &g, int64_t base_pix, int s) {
    const int nn = NSHIFT >= 0 ? (1 << NSHIFT) : g.N;
    int l = div_n<NSHIFT>(g, s), n = s - l * nn;
    return (base_pix + (int64_t)l * g.pix_stride) * nn + n;
}

// CENSUS: debug instantiation recording [smid, realtime start, realtime end] of every block (nmrf_debug_stripe_census)
template <int AXIS, int NSHIFT, int KSPLIT, bool CENSUS = false>
__global__ __launch_bounds__(256) void stripe_attn_kernel(const float *__restrict__ qkv, const float *__restrict__ lepe,
                                                         StripeGeom g, float scale, float *__restrict__ out,
                                                         unsigned long long *__restrict__ census = nullptr) {
    constexpr int QPB = 4 / KSPLIT;                           // query tiles per block
    struct Scope {
        unsigned long long *p;
        __device__ Scope(unsigned long long *c) : p(c) {
            if (CENSUS && threadIdx.x == 0) {
                p = c + 3 * ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
                p[0] = __smid(); p[1] = wall_clock64();
            }
        }
        __device__ ~Scope() { if (CENSUS && threadIdx.x == 0) p[2] = wall_clock64(); }
    } scope(census);
    __shared__ float s_o[KSPLIT > 1 ? 4 : 1][16][64];          // partial O^T of the non-leading key ranges
    __shared__ float s_ml[KSPLIT > 1 ? 4 : 1][2][64];          // their (m, l)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const int qslot = wv / KSPLIT, ks = wv % KSPLIT;           // which query tile of the block, which key range
    const int qt = blockIdx.x * QPB + qslot;
    const int q0 = qt * SA_TILE;
    const bool wave_on = q0 < g.Ts;                            // wave-uniform
    const int stripe = blockIdx.y >> 1, head = blockIdx.y & 1;
    const int b = blockIdx.z;
    const int64_t base_pix = (AXIS == 0) ? ((int64_t)b * g.H * g.W + stripe)          // column x = stripe
                                         : ((int64_t)b * g.H * g.W + (int64_t)stripe * g.W);  // row y = stripe
    const size_t ld = (size_t)3 * g.C;
    const int coff = AXIS * (g.C / 2) + head * 32;            // channel offset of this (half, head) inside q / k / v
    const int nlab = NSHIFT >= 0 ? (1 << NSHIFT) : g.N;

    const int qs = q0 + qi;
    const bool q_ok = qs < g.Ts;
    const int qsc = q_ok ? qs : g.Ts - 1;
    const int64_t qrow = stripe_row<NSHIFT>(g, base_pix, wave_on ? qsc : 0);
    const int q_pix = div_n<NSHIFT>(g, qsc);

    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if (wave_on) {
        // ---- Q fragment (B operand): lane (qi,hi) holds Q[q0+qi][16*hi + s], pre-scaled by s*log2(e) ------
        float qf[16];
        {
            const float sc2 = scale * SA_LOG2E;
            const float *p = qkv + qrow * ld + coff + 16 * hi;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + 4 * c);
                qf[4 * c + 0] = v.x * sc2; qf[4 * c + 1] = v.y * sc2; qf[4 * c + 2] = v.z * sc2; qf[4 * c + 3] = v.w * sc2;
            }
        }
        const int n_kt = (g.Ts + SA_TILE - 1) / SA_TILE;
        const int per = (n_kt + KSPLIT - 1) / KSPLIT;
        const int kt_begin = ks * per, kt_end = (kt_begin + per < n_kt) ? kt_begin + per : n_kt;
        const float *kbase = qkv + g.C + coff + 16 * hi;
        const float *vbase = qkv + 2 * g.C + coff + qi;
        // K fragment (A operand): lane (ki=qi, hi) holds K[k0+ki][16*hi + s]
        auto load_k = [&](int kt, float *kd) {
            int kk = kt * SA_TILE + qi;
            kk = kk < g.Ts ? kk : g.Ts - 1;
            const float *p = kbase + stripe_row<NSHIFT>(g, base_pix, kk) * ld;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + 4 * c);
                kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
            }
        };
        // V fragment (A operand of the 2nd product): lane (d=qi, hi), step s: V[k0+mfma_row(s,hi)][d]
        auto load_v = [&](int kt, float *vd) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                int kv = kt * SA_TILE + mfma_row(s, hi);
                kv = kv < g.Ts ? kv : g.Ts - 1;
                vd[s] = vbase[stripe_row<NSHIFT>(g, base_pix, kv) * ld];
            }
        };
        float kf[16], vf[16];
        if (kt_begin < kt_end) { load_k(kt_begin, kf); load_v(kt_begin, vf); }
#pragma unroll 1
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int k0 = kt * SA_TILE;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) st = mfma32(kf[s], qf[s], st);
            if (kt + 1 < kt_end) load_k(kt + 1, kf);          // K fragment is dead: refill now
            if (kt == n_kt - 1) {                              // keys beyond the stripe (last tile only)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (k0 + mfma_row(r, hi) >= g.Ts) st[r] = -INFINITY;
            }
            if ((NSHIFT < 0 || kt == qt) && nlab > 1) {        // sibling labels of the query's own pixel (diagonal tile
                                                               // when N divides the tile; every tile for a generic N)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kk = k0 + mfma_row(r, hi);
                    if (div_n<NSHIFT>(g, kk) == q_pix && kk != qsc) st[r] = -INFINITY;
                }
            }
            float m_tile = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
            m_tile = half_max(m_tile);
            const float m_new = fmaxf(m_run, m_tile);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);          // m_run=-inf -> 0
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_use);                  // masked (-inf) -> 0
                psum += st[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], st[s], acc_o);
            if (kt + 1 < kt_end) load_v(kt + 1, vf);          // V fragment likewise
        }
        l_run = half_sum(l_run);                        // both halves now hold the range's full (m, l)
    }

    // ---- merge the KSPLIT key ranges of each query tile through LDS ----------------------------------------
    if (KSPLIT > 1) {
        if (ks != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_o[wv][r][lane] = acc_o[r];
            s_ml[wv][0][lane] = m_run;
            s_ml[wv][1][lane] = l_run;
        }
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int j = 1; j < KSPLIT; ++j) {
            const float m2 = s_ml[wv + j][0][lane], l2 = s_ml[wv + j][1][lane];
            const float m_new = fmaxf(m_run, m2);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float a1 = __builtin_amdgcn_exp2f(m_run - m_use), a2 = __builtin_amdgcn_exp2f(m2 - m_use);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_o[r] = acc_o[r] * a1 + s_o[wv + j][r][lane] * a2;
            l_run = l_run * a1 + l2 * a2;
            m_run = m_new;
        }
    }
    if (!wave_on || !q_ok) return;
    const float inv_l = 1.0f / l_run;

    // ---- epilogue: normalise, add LePE, store.  Lane (q=qi, hi) owns channels d = mfma_row(r,hi) --------
    // LePE for width-1 stripes (NMP.py:433-449 / SURVEY H3):
    //   rpe_j(p)[c] = w_c v_j(p)[c] + sum_k ( w_- v_k(p-1)[c] + w_+ v_k(p+1)[c] )
    // taps = centre column (AXIS 0: kernel[:,1]) or centre row (AXIS 1: kernel[1,:]) of the 3x3 kernel.
    const int tap_m = (AXIS == 0) ? 1 : 3, tap_c = 4, tap_p = (AXIS == 0) ? 7 : 5;
    const bool has_prev = q_pix > 0, has_next = q_pix < g.L - 1;
    const int64_t prev_row = stripe_row<NSHIFT>(g, base_pix, (q_pix - 1) * nlab);
    const int64_t next_row = stripe_row<NSHIFT>(g, base_pix, (q_pix + 1) * nlab);
    float *op = out + qrow * g.C + coff;
    const float *vb = qkv + 2 * g.C + coff;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int d0 = mfma_row(4 * rb, hi);                   // 4 consecutive channels d0..d0+3
        float4 vq = ldg4(vb + qrow * ld + d0);
        float4 sp = make_float4(0.f, 0.f, 0.f, 0.f), sn = sp;
        if (has_prev)
            for (int n = 0; n < nlab; ++n) {
                float4 t = ldg4(vb + (prev_row + n) * ld + d0);
                sp.x += t.x; sp.y += t.y; sp.z += t.z; sp.w += t.w;
            }
        if (has_next)
            for (int n = 0; n < nlab; ++n) {
                float4 t = ldg4(vb + (next_row + n) * ld + d0);
                sn.x += t.x; sn.y += t.y; sn.z += t.z; sn.w += t.w;
            }
        const float vqa[4] = {vq.x, vq.y, vq.z, vq.w}, spa[4] = {sp.x, sp.y, sp.z, sp.w}, sna[4] = {sn.x, sn.y, sn.z, sn.w};
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float *wk = lepe + (size_t)(head * 32 + d0 + e) * 9;
            const float rpe = wk[tap_c] * vqa[e] + wk[tap_m] * spa[e] + wk[tap_p] * sna[e];
            res[e] = acc_o[4 * rb + e] * inv_l + rpe;
        }
        stg4(op + d0, make_float4(res[0], res[1], res[2], res[3]));
    }
}

template <int AXIS, int NSHIFT>
static void launch_stripe(const float *qkv, const float *lepe, const StripeGeom &g, int stripes, int B, float scale,
                          float *out, hipStream_t st) {
    // key split: aim for a few thousand waves per launch (256 CUs x 4 SIMDs x ~3), keep >= 2 key tiles per wave
    const int n_qt = (g.Ts + SA_TILE - 1) / SA_TILE;
    // (decided per image, NOT per batch, so that results do not depend on the batch size)
    const long waves1 = (long)n_qt * stripes * 2;
    int ksplit = 1;
    if (waves1 * 2 <= 8192 && n_qt >= 4) ksplit = 2;
    if (waves1 * 4 <= 8192 && n_qt >= 8) ksplit = 4;
    const int qpb = 4 / ksplit;
    dim3 grid((n_qt + qpb - 1) / qpb, stripes * 2, B);
    if (ksplit == 1) hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 1>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
    else if (ksplit == 2) hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 2>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
    else hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 4>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
}

// Debug: census run of the horizontal N=4 KSPLIT=4 kernel (KITTI batch-1 configuration); census[blocks*3] on device.
extern "C" int nmrf_debug_stripe_census(const float *qkv, const float *lepe_h, int B, int H, int W, float *out,
                                        unsigned long long *census, int *grid_out, void *stream) {
    StripeGeom g{H, W, 4, 128, W, W * 4, (int64_t)1};
    const int n_qt = (g.Ts + SA_TILE - 1) / SA_TILE;
    dim3 grid(n_qt, H * 2, B);                                // KSPLIT = 4 -> one query tile per block
    grid_out[0] = grid.x; grid_out[1] = grid.y; grid_out[2] = grid.z;
    hipLaunchKernelGGL((stripe_attn_kernel<1, 2, 4, true>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_h, g,
                       1.0f / sqrtf(32.0f), out, census);
    return nmrf_launch_status();
}

extern "C" int nmrf_stripe_attn_f32(const float *qkv, const float *lepe_v, const float *lepe_h, int B, int H, int W,
                                    int N, int C, int axes, float *out, void *stream) {
    if (!qkv || !lepe_v || !lepe_h || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N < 1 || C != 128 || axes < 1 || axes > 3) return NMRF_EINVAL;
    if (W * 2 > 65535 || H * 2 > 65535 || B > 65535) return NMRF_EINVAL;
    const float scale = 1.0f / sqrtf(32.0f);
    hipStream_t st = (hipStream_t)stream;
    if (axes & 1) {   // vertical stripes: one per column, H*N tokens each, channel half 0
        StripeGeom g{H, W, N, C, H, H * N, (int64_t)W};
        if (N == 4) launch_stripe<0, 2>(qkv, lepe_v, g, W, B, scale, out, st);
        else launch_stripe<0, -1>(qkv, lepe_v, g, W, B, scale, out, st);
    }
    if (axes & 2) {   // horizontal stripes: one per row, W*N tokens each, channel half 1
        StripeGeom g{H, W, N, C, W, W * N, (int64_t)1};
        if (N == 4) launch_stripe<1, 2>(qkv, lepe_h, g, H, B, scale, out, st);
        else launch_stripe<1, -1>(qkv, lepe_h, g, H, B, scale, out, st);
    }
    return nmrf_launch_status();
}
