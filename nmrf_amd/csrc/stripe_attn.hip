// A7: cross-stripe (CSWin, split_size 1) attention with LePE on fp32 matrix cores.
//
// One wave = one 32-query tile of one (image b, stripe, head).  No LDS, no barriers: K and V
// operand fragments are read straight from the token-major qkv tensor (each token's 32-wide head
// slice is one 128-byte line, shared through L1/L2 by the 4 waves of the block, which walk the
// same stripe).  Flash-style streaming softmax, so the [T,T] matrix never exists (H9).
//
// Matrix-core formulation (v_mfma_f32_32x32x2_f32, exact fp32, 16 k-steps per 32-deep contraction):
//   S^T[key][q]  = sum_c K[key][c] * Q[q][c]          A = K fragment, B = Q fragment
//   O^T[d][q]   += sum_key V[key][d] * P[q][key]      A = V fragment, B = P (the S^T accumulator itself)
// With the transposed forms the softmax statistics of query q live in lane q (both half-waves),
// and the S^T accumulator registers ARE the B operand of the second product: MFMA k-slot (step s,
// half h) <-> key mfma_row(s,h), which is exactly where S^T register s of half h sits.
// Channel <-> k-slot map for the first product is (step s, half h) <-> c = 16h + s, so each lane
// loads 16 contiguous floats of its row.
#include "common.h"

#define SA_TILE 32

struct StripeGeom {
    int H, W, N, C;        // grid, labels per pixel, embed dim (128)
    int nshift;            // log2(N) when N is a power of two, else -1 (avoids integer division)
    int L;                 // pixels per stripe
    int Ts;                // tokens per stripe = L*N
    int64_t pix_stride;    // token-pixel stride between consecutive stripe positions (W or 1)
};

// token row (in units of tokens) of in-stripe token s
// NSHIFT >= 0: N == 1<<NSHIFT at compile time (N=4 in every shipped config); NSHIFT < 0: runtime N
template <int NSHIFT>
__device__ __forceinline__ int div_n(const StripeGeom &g, int s) { return NSHIFT >= 0 ? (s >> NSHIFT) : (s / g.N); }

template <int NSHIFT>
__device__ __forceinline__ int64_t stripe_row(const StripeGeom &g, int64_t base_pix, int s) {
    const int nn = NSHIFT >= 0 ? (1 << NSHIFT) : g.N;
    int l = div_n<NSHIFT>(g, s), n = s - l * nn;
    return (base_pix + (int64_t)l * g.pix_stride) * nn + n;
}

template <int AXIS, int NSHIFT>
__global__ __launch_bounds__(256) void stripe_attn_kernel(const float *__restrict__ qkv, const float *__restrict__ lepe,
                                                         StripeGeom g, float scale, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const int qt = blockIdx.x * 4 + wv;
    const int q0 = qt * SA_TILE;
    if (q0 >= g.Ts) return;                                   // wave-uniform
    const int stripe = blockIdx.y >> 1, head = blockIdx.y & 1;
    const int b = blockIdx.z;
    const int64_t base_pix = (AXIS == 0) ? ((int64_t)b * g.H * g.W + stripe)          // column x = stripe
                                         : ((int64_t)b * g.H * g.W + (int64_t)stripe * g.W);  // row y = stripe
    const size_t ld = (size_t)3 * g.C;
    const int coff = AXIS * (g.C / 2) + head * 32;            // channel offset of this (half, head) inside q / k / v

    // ---- Q fragment (B operand): lane (qi,hi) holds Q[q0+qi][16*hi + s], pre-scaled ------------------
    const int qs = q0 + qi;
    const bool q_ok = qs < g.Ts;
    const int64_t qrow = stripe_row<NSHIFT>(g, base_pix, q_ok ? qs : g.Ts - 1);
    float qf[16];
    {
        const float *p = qkv + qrow * ld + coff + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            qf[4 * c + 0] = v.x * scale; qf[4 * c + 1] = v.y * scale;
            qf[4 * c + 2] = v.z * scale; qf[4 * c + 3] = v.w * scale;
        }
    }
    const int q_pix = div_n<NSHIFT>(g, qs);

    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int n_kt = (g.Ts + SA_TILE - 1) / SA_TILE;
    // K fragment (A operand): lane (ki=qi, hi) holds K[k0+ki][16*hi + s]
    // V fragment (A operand of the 2nd product): lane (d=qi, hi), step s: V[k0+mfma_row(s,hi)][d]
    auto load_k = [&](int kt, float *kd) {
        const int k0 = kt * SA_TILE;
        int ks = k0 + qi;
        ks = ks < g.Ts ? ks : g.Ts - 1;
        const float *p = qkv + stripe_row<NSHIFT>(g, base_pix, ks) * ld + g.C + coff + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
        }
    };
    auto load_v = [&](int kt, float *vd) {
        const int k0 = kt * SA_TILE;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            int kv = k0 + mfma_row(s, hi);
            kv = kv < g.Ts ? kv : g.Ts - 1;
            vd[s] = qkv[stripe_row<NSHIFT>(g, base_pix, kv) * ld + 2 * g.C + coff + qi];
        }
    };
    float kf[16], vf[16];
    load_k(0, kf);
    load_v(0, vf);
#pragma unroll 1
    for (int kt = 0; kt < n_kt; ++kt) {
        const int k0 = kt * SA_TILE;
        // ---- S^T = K Q^T -----------------------------------------------------------------------------
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) st = mfma32(kf[s], qf[s], st);
        if (kt + 1 < n_kt) load_k(kt + 1, kf);     // K fragment is dead: refill now, in flight during softmax + P.V
        // ---- mask: out-of-stripe keys, and sibling labels of the query's own pixel ---------------------
        float m_tile = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ks = k0 + mfma_row(r, hi);
            const bool dead = (ks >= g.Ts) || ((div_n<NSHIFT>(g, ks) == q_pix) && (ks != qs));
            st[r] = dead ? -INFINITY : st[r];
            m_tile = fmaxf(m_tile, st[r]);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_use);              // m_run=-inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - m_use);                      // masked (-inf) -> 0
            psum += st[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
        // ---- O^T += V^T P^T --------------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], st[s], acc_o);
        if (kt + 1 < n_kt) load_v(kt + 1, vf);     // V fragment likewise: in flight during the next S^T and softmax
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;

    if (!q_ok) return;
    // ---- epilogue: normalise, add LePE, store.  Lane (q=qi, hi) owns channels d = mfma_row(r,hi) --------
    // LePE for width-1 stripes (NMP.py:433-449 / SURVEY H3):
    //   rpe_j(p)[c] = w_c v_j(p)[c] + sum_k ( w_- v_k(p-1)[c] + w_+ v_k(p+1)[c] )
    // taps = centre column (AXIS 0: kernel[:,1]) or centre row (AXIS 1: kernel[1,:]) of the 3x3 kernel.
    const int tap_m = (AXIS == 0) ? 1 : 3, tap_c = 4, tap_p = (AXIS == 0) ? 7 : 5;
    const bool has_prev = q_pix > 0, has_next = q_pix < g.L - 1;
    const int nlab = NSHIFT >= 0 ? (1 << NSHIFT) : g.N;
    const int64_t prev_row = stripe_row<NSHIFT>(g, base_pix, (q_pix - 1) * nlab);
    const int64_t next_row = stripe_row<NSHIFT>(g, base_pix, (q_pix + 1) * nlab);
    float *op = out + qrow * g.C + coff;
    const float *vbase = qkv + 2 * g.C + coff;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int d0 = mfma_row(4 * rb, hi);                   // 4 consecutive channels d0..d0+3
        float4 vq = ldg4(vbase + qrow * ld + d0);
        float4 sp = make_float4(0.f, 0.f, 0.f, 0.f), sn = sp;
        if (has_prev)
            for (int n = 0; n < nlab; ++n) {
                float4 t = ldg4(vbase + (prev_row + n) * ld + d0);
                sp.x += t.x; sp.y += t.y; sp.z += t.z; sp.w += t.w;
            }
        if (has_next)
            for (int n = 0; n < nlab; ++n) {
                float4 t = ldg4(vbase + (next_row + n) * ld + d0);
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

extern "C" int nmrf_stripe_attn_f32(const float *qkv, const float *lepe_v, const float *lepe_h, int B, int H, int W,
                                    int N, int C, int axes, float *out, void *stream) {
    if (!qkv || !lepe_v || !lepe_h || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N < 1 || C != 128 || axes < 1 || axes > 3) return NMRF_EINVAL;
    const float scale = 1.0f / sqrtf(32.0f);
    int nshift = -1;
    for (int k = 0; k < 5; ++k) if ((1 << k) == N) nshift = k;
    if (axes & 1) {   // vertical stripes: one per column, H*N tokens each, channel half 0
        StripeGeom g{H, W, N, C, nshift, H, H * N, (int64_t)W};
        dim3 grid((g.Ts + 4 * SA_TILE - 1) / (4 * SA_TILE), W * 2, B);
        if (N == 4) hipLaunchKernelGGL((stripe_attn_kernel<0, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_v, g, scale, out);
        else hipLaunchKernelGGL((stripe_attn_kernel<0, -1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_v, g, scale, out);
    }
    if (axes & 2) {   // horizontal stripes: one per row, W*N tokens each, channel half 1
        StripeGeom g{H, W, N, C, nshift, W, W * N, (int64_t)1};
        dim3 grid((g.Ts + 4 * SA_TILE - 1) / (4 * SA_TILE), H * 2, B);
        if (N == 4) hipLaunchKernelGGL((stripe_attn_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_h, g, scale, out);
        else hipLaunchKernelGGL((stripe_attn_kernel<1, -1>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_h, g, scale, out);
    }
    return nmrf_launch_status();
}
