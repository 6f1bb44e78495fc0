// A10(ii) / A13: (shifted-)window attention with relative-position q/k/v embeddings on fp32 matrix cores.
//
// One block = one (image, window, head); wave w of the block owns query tile w (32 window tokens).
// Window tokens are ordered (a, b, n) = pixel-major, label-minor, on the ROLLED grid; the roll by
// -shift and the inverse roll of the output are folded into the row map (H4/H5 of SURVEY).
//
// logits_ij = s q_i.k_j + s q_i.ek[rel(pi,pj)] + s k_j.eq[rel(pi,pj)]     rel from the (2w-1)^2 table
// out_i     = sum_j softmax_j(logits) (v_j + ev[rel(pi,pj)])
//
// The q.k and p.v contractions run on v_mfma_f32_32x32x2_f32 in the transposed form described in
// stripe_attn.hip.  The three relative-position terms only depend on the key PIXEL (w^2 values, not
// w^2 N), so they are done on the vector ALU at 1/N of the reference's cost:
//   phase 0: QRt[pj][i] = s q_i.ek[rel(pi,pj)],  KR[j][pi] = s k_j.eq[rel(pi,pj)]  -> LDS
//            (lanes 0-31 of wave w do QR, lanes 32-63 do KR, for window token 32w + lane%32)
//   phase 1: per key tile: S^T on MFMA + LDS look-ups + masks, streaming softmax, P.V on MFMA,
//            sum_pj (sum_n p) ev[rel] on the VALU;  then cross-half exchange and store.
#include "common.h"

struct WinGeom {
    int Hp, Wp, N, C, heads, win, shift, sibling;
    int nshift;          // log2 N or -1
    int Tw;              // tokens per window = win*win*N
    int R;               // (2 win - 1)^2
};

__device__ __forceinline__ int wdiv_n(const WinGeom &g, int s) { return g.nshift >= 0 ? (s >> g.nshift) : (s / g.N); }

template <int NKT>
__global__ __launch_bounds__(64 * NKT) void window_attn_kernel(const float *__restrict__ qkv,
        const float *__restrict__ table, WinGeom g, float scale, float *__restrict__ out) {
    constexpr int TP = NKT * 32;                   // padded tokens per window
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int W2 = g.win * g.win;
    float *tab_a = smem;                            // ek [R][32]   (later: ev)
    float *tab_b = tab_a + g.R * 32;                // eq*s [R][32]
    float *qrt = tab_b + g.R * 32;                  // [W2][TP]
    float *kr = qrt + W2 * TP;                      // [TP][W2]
    int *rowmap = reinterpret_cast<int *>(kr + TP * W2);   // [TP] token row (units of tokens) of window token i

    const int tid = threadIdx.x, nthr = 64 * NKT;
    const int lane = tid & 63, wv = tid >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const int nwx = g.Wp / g.win;
    const int wj = blockIdx.x % nwx, wi = blockIdx.x / nwx;
    const int head = blockIdx.y, bimg = blockIdx.z;
    const size_t ld = (size_t)3 * g.C;
    const int tab_ld = 3 * g.C;
    const int tcol = head * 96;                     // per head: [eq(32) | ek(32) | ev(32)]  (NMP.py:257-260)

    // ---- stage ek / eq*s, build the row map --------------------------------------------------------
    for (int i = tid; i < g.R * 8; i += nthr) {
        int r = i >> 3, c4 = (i & 7) * 4;
        float4 ek = ldg4(table + (size_t)r * tab_ld + tcol + 32 + c4);
        float4 eq = ldg4(table + (size_t)r * tab_ld + tcol + c4);
        stg4(tab_a + r * 32 + c4, ek);
        stg4(tab_b + r * 32 + c4, make_float4(eq.x * scale, eq.y * scale, eq.z * scale, eq.w * scale));
    }
    for (int i = tid; i < TP; i += nthr) {
        int ii = i < g.Tw ? i : g.Tw - 1;          // padded tokens alias the last real one (always masked)
        int pt = wdiv_n(g, ii), n = ii - pt * g.N;
        int a = pt / g.win, b = pt - a * g.win;
        int Y = wi * g.win + a + g.shift, X = wj * g.win + b + g.shift;
        Y = Y >= g.Hp ? Y - g.Hp : Y;
        X = X >= g.Wp ? X - g.Wp : X;
        rowmap[i] = (((bimg * g.Hp + Y) * g.Wp) + X) * g.N + n;
    }
    __syncthreads();

    // ---- phase 0: relative-position logit terms -----------------------------------------------------
    {
        const int tok = 32 * wv + qi;
        if (tok < g.Tw) {
            const float *src = qkv + (size_t)rowmap[tok] * ld + (hi ? g.C : 0) + head * 32;
            float vec[32];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float4 v = ldg4(src + 4 * c);
                const float sc = hi ? 1.0f : scale;
                vec[4 * c + 0] = v.x * sc; vec[4 * c + 1] = v.y * sc; vec[4 * c + 2] = v.z * sc; vec[4 * c + 3] = v.w * sc;
            }
            const int pt = wdiv_n(g, tok);
            const int at = pt / g.win, bt = pt - at * g.win;
            const float *tab = hi ? tab_b : tab_a;
            const int span = 2 * g.win - 1;
            for (int pj = 0; pj < W2; ++pj) {
                const int aj = pj / g.win, bj = pj - aj * g.win;
                // hi==0: this token is the query i, pj the key pixel:  rel(i,j) = (a_i-a_j, b_i-b_j)
                // hi==1: this token is the key j,  pj the query pixel: rel(i,j) = (a_pj-a_t, b_pj-b_t)
                const int da = hi ? (aj - at) : (at - aj), db = hi ? (bj - bt) : (bt - bj);
                const float *e = tab + ((da + g.win - 1) * span + (db + g.win - 1)) * 32;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float4 t = *reinterpret_cast<const float4 *>(e + 4 * c);
                    s = fmaf(vec[4 * c + 0], t.x, s); s = fmaf(vec[4 * c + 1], t.y, s);
                    s = fmaf(vec[4 * c + 2], t.z, s); s = fmaf(vec[4 * c + 3], t.w, s);
                }
                if (hi) kr[tok * W2 + pj] = s; else qrt[pj * TP + tok] = s;
            }
        }
    }
    __syncthreads();
    // ek is dead: overwrite it with ev for phase 2
    for (int i = tid; i < g.R * 8; i += nthr) {
        int r = i >> 3, c4 = (i & 7) * 4;
        stg4(tab_a + r * 32 + c4, ldg4(table + (size_t)r * tab_ld + tcol + 64 + c4));
    }
    __syncthreads();

    // ---- phase 1: S^T tiles ---------------------------------------------------------------------------
    const int q0 = 32 * wv;
    const int qs = q0 + qi;
    const bool q_ok = qs < g.Tw;
    const int qsc = q_ok ? qs : g.Tw - 1;
    const int64_t qrow = rowmap[qsc];
    const int q_pix = wdiv_n(g, qsc);
    const int qa = q_pix / g.win, qb = q_pix - qa * g.win;
    float qf[16];
    {
        const float *p = qkv + qrow * ld + head * 32 + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            qf[4 * c + 0] = v.x * scale; qf[4 * c + 1] = v.y * scale; qf[4 * c + 2] = v.z * scale; qf[4 * c + 3] = v.w * scale;
        }
    }
    // Swin shift regions on the rolled grid (NMP.py:211-239): id = f(Y')*3 + f(X')
    auto region = [&](int a, int b) -> int {
        int Yr = wi * g.win + a, Xr = wj * g.win + b;
        int fy = Yr < g.Hp - g.win ? 0 : (Yr < g.Hp - g.shift ? 1 : 2);
        int fx = Xr < g.Wp - g.win ? 0 : (Xr < g.Wp - g.shift ? 1 : 2);
        return fy * 3 + fx;
    };
    const int q_reg = g.shift ? region(qa, qb) : 0;

    // ---- phase 1+2: streaming softmax over the key tiles (flash style; S never leaves registers) -----
    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float oe[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) oe[d] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int span = 2 * g.win - 1;
    const bool group4 = (g.N & 3) == 0;                    // the 4 keys of one register quad share a pixel
    const float *qrt_q = qrt + qsc;
    const float *vcol = qkv + 2 * g.C + head * 32 + qi;
#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
        const int k0 = 32 * kt;
        float kf[16];
        {
            const float *p = qkv + (size_t)rowmap[k0 + qi] * ld + g.C + head * 32 + 16 * hi;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + 4 * c);
                kf[4 * c + 0] = v.x; kf[4 * c + 1] = v.y; kf[4 * c + 2] = v.z; kf[4 * c + 3] = v.w;
            }
        }
        float vf[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) vf[s] = vcol[(size_t)rowmap[k0 + mfma_row(s, hi)] * ld];
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) st = mfma32(kf[s], qf[s], st);
        float m_tile = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + mfma_row(r, hi);
            const int keyc = key < g.Tw ? key : g.Tw - 1;
            const int pk = wdiv_n(g, keyc);
            float v = st[r] + qrt_q[pk * TP] + kr[keyc * W2 + q_pix];
            bool dead = key >= g.Tw;
            if (g.sibling) dead = dead || (pk == q_pix && keyc != qsc);
            if (g.shift) {
                const int ka = pk / g.win, kb = pk - ka * g.win;
                dead = dead || (region(ka, kb) != q_reg);
            }
            v = dead ? -INFINITY : v;
            st[r] = v;
            m_tile = fmaxf(m_tile, v);
        }
        m_tile = fmaxf(m_tile, __shfl_xor(m_tile, 32));
        const float m_new = fmaxf(m_run, m_tile);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = expf(st[r] - m_use);
            psum += st[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
#pragma unroll
        for (int d = 0; d < 32; ++d) oe[d] *= alpha;
        // P.V on MFMA
#pragma unroll
        for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], st[s], acc_o);
        // value-embedding term on the VALU: sum over key PIXELS of (sum_n p) * ev[rel(pq,pk)]
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int nsub = group4 ? 1 : 4;
            for (int e4 = 0; e4 < nsub; ++e4) {
                int key = k0 + mfma_row(4 * rq, hi) + e4;
                key = key < g.Tw ? key : g.Tw - 1;
                const int pk = wdiv_n(g, key);
                const int ka = pk / g.win, kb = pk - ka * g.win;
                float pp;
                if (group4) pp = (st[4 * rq] + st[4 * rq + 1]) + (st[4 * rq + 2] + st[4 * rq + 3]);
                else pp = e4 == 0 ? st[4 * rq] : (e4 == 1 ? st[4 * rq + 1] : (e4 == 2 ? st[4 * rq + 2] : st[4 * rq + 3]));
                const float *e = tab_a + ((qa - ka + g.win - 1) * span + (qb - kb + g.win - 1)) * 32;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float4 t = *reinterpret_cast<const float4 *>(e + 4 * c);
                    oe[4 * c + 0] = fmaf(pp, t.x, oe[4 * c + 0]); oe[4 * c + 1] = fmaf(pp, t.y, oe[4 * c + 1]);
                    oe[4 * c + 2] = fmaf(pp, t.z, oe[4 * c + 2]); oe[4 * c + 3] = fmaf(pp, t.w, oe[4 * c + 3]);
                }
            }
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv_l = 1.0f / l_tot;
    // lane (q,hi) owns output channels d = mfma_row(r,hi); it needs its partner's oe at those channels and
    // owes the partner its own oe at mfma_row(r,1-hi).  Indices are compile-time on both sides of the select.
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float mine = hi ? oe[mfma_row(r, 1)] : oe[mfma_row(r, 0)];
        const float send = hi ? oe[mfma_row(r, 0)] : oe[mfma_row(r, 1)];
        const float recv = __shfl_xor(send, 32);
        res[r] = (acc_o[r] + (mine + recv)) * inv_l;
    }
    if (!q_ok) return;
    float *op = out + qrow * g.C + head * 32;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        stg4(op + mfma_row(4 * rb, hi), make_float4(res[4 * rb], res[4 * rb + 1], res[4 * rb + 2], res[4 * rb + 3]));
}

template <int NKT>
static int launch_window(const float *qkv, const float *table, const WinGeom &g, int B, float *out, hipStream_t st) {
    const int TP = NKT * 32, W2 = g.win * g.win;
    size_t smem = (size_t)(2 * g.R * 32 + 2 * W2 * TP) * sizeof(float) + (size_t)TP * sizeof(int);
    if (smem > 160 * 1024) return NMRF_EINVAL;
    static bool attr_done = false;      // set once, outside any stream capture
    if (smem > 64 * 1024 && !attr_done) {
        attr_done = true;
        if (hipFuncSetAttribute((const void *)window_attn_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return NMRF_ELAUNCH;
    }
    dim3 grid((g.Hp / g.win) * (g.Wp / g.win), g.heads, B);
    hipLaunchKernelGGL(window_attn_kernel<NKT>, grid, dim3(64 * NKT), smem, st, qkv, table, g, 1.0f / sqrtf(32.0f), out);
    return nmrf_launch_status();
}

extern "C" int nmrf_window_attn_f32(const float *qkv, const float *table, int B, int Hp, int Wp, int N, int C, int heads,
                                    int win, int shift, int sibling_mask, float *out, void *stream) {
    if (!qkv || !table || !out) return NMRF_ENULL;
    if (B < 1 || N < 1 || win < 1 || Hp % win || Wp % win || shift < 0 || shift >= win || heads * 32 != C || (C & 3))
        return NMRF_EINVAL;
    if ((int64_t)B * Hp * Wp * N >= (int64_t)1 << 31) return NMRF_EINVAL;
    WinGeom g{Hp, Wp, N, C, heads, win, shift, sibling_mask ? 1 : 0, -1, win * win * N, (2 * win - 1) * (2 * win - 1)};
    for (int k = 0; k < 5; ++k) if ((1 << k) == N) g.nshift = k;
    const int nkt = (g.Tw + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    switch (nkt) {
        case 1: return launch_window<1>(qkv, table, g, B, out, st);
        case 2: return launch_window<2>(qkv, table, g, B, out, st);
        case 3: return launch_window<3>(qkv, table, g, B, out, st);
        case 4: return launch_window<4>(qkv, table, g, B, out, st);
        case 5: return launch_window<5>(qkv, table, g, B, out, st);
        case 6: return launch_window<6>(qkv, table, g, B, out, st);
        default: return NMRF_EINVAL;
    }
}
