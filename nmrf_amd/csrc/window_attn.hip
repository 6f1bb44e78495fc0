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
//
// The kernel is latency- not throughput-limited at batch 1 (a few hundred blocks of 1-5 waves), so
// every global read is issued as early as its address is known: the phase-0 operand before the tables
// are staged, the ev table into registers before phase 0 (stored to LDS after the barrier that retires
// ek), Q and the first K/V fragments before that barrier, the K/V fragments of tile t+1 before the
// MFMAs of tile t.
// This is the generic kernel (runtime window size / labels per pixel) used by non-shipped configurations; the
// shipped ones (6x6x4 inference windows, 4x4x1 refinement windows) run window_attn_fast_kernel below.
#include "common.h"
#include "split_mfma.h"

struct WinGeom {
    int Hp, Wp, N, C, heads, win, shift, sibling;
    int Tw;              // tokens per window = win*win*N
    int R;               // (2 win - 1)^2
};

template <int NKT>
__global__ __launch_bounds__(64 * NKT) void window_attn_kernel(const float *__restrict__ qkv,
        const float *__restrict__ table, WinGeom g, float scale, float *__restrict__ out) {
    constexpr int TP = NKT * 32;                   // padded tokens per window
    constexpr int NTHR = 64 * NKT;
    const int win = g.win;
    const int nl = g.N;
    const int W2 = win * win;
    const int span = 2 * win - 1;
    const int R = span * span;
    const int Tw = W2 * nl;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tab_a = smem;                            // ek [R][32]   (later: ev)
    float *tab_b = tab_a + R * 32;                  // eq*s [R][32]
    float *qrt = tab_b + R * 32;                    // [W2][TP]
    float *kr = qrt + W2 * TP;                      // [TP][W2]
    int *rowmap = reinterpret_cast<int *>(kr + TP * W2);   // [TP] token row (units of tokens) of window token i

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int qi = lane & 31, hi = lane >> 5;
    const int nwx = g.Wp / win;
    const int wj = blockIdx.x % nwx, wi = blockIdx.x / nwx;
    const int head = blockIdx.y, bimg = blockIdx.z;
    const size_t ld = (size_t)3 * g.C;
    const int tab_ld = 3 * g.C;
    const int tcol = head * 96;                     // per head: [eq(32) | ek(32) | ev(32)]  (NMP.py:257-260)

    auto token_row = [&](int i) -> int {            // window token -> row of the token-major tensors
        int ii = i < Tw ? i : Tw - 1;               // padded tokens alias the last real one (always masked)
        int pt = ii / nl, n = ii - pt * nl;
        int a = pt / win, b = pt - a * win;
        int Y = wi * win + a + g.shift, X = wj * win + b + g.shift;
        Y = Y >= g.Hp ? Y - g.Hp : Y;
        X = X >= g.Wp ? X - g.Wp : X;
        return (((bimg * g.Hp + Y) * g.Wp) + X) * nl + n;
    };

    // ---- earliest loads: phase-0 operand (q or k of token 32w+qi) and this wave's Q fragment -----------
    const int tok = 32 * wv + qi;                   // == query token of lane (qi, *) in phase 1
    const bool tok_ok = tok < Tw;
    const int tokc = tok_ok ? tok : Tw - 1;
    const int64_t trow = token_row(tokc);
    float vec[32];
    {
        const float *src = qkv + trow * ld + (hi ? g.C : 0) + head * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 v = ldg4(src + 4 * c);
            vec[4 * c + 0] = v.x; vec[4 * c + 1] = v.y; vec[4 * c + 2] = v.z; vec[4 * c + 3] = v.w;
        }
    }

    // ---- stage ek / eq*s (all loads first, then the LDS stores), build the row map -------------------
    for (int i = tid; i < R * 8; i += NTHR) {
        const int r = i >> 3, c4 = (i & 7) * 4;
        float4 ek = ldg4(table + (size_t)r * tab_ld + tcol + 32 + c4);
        float4 eq = ldg4(table + (size_t)r * tab_ld + tcol + c4);
        stg4(tab_a + r * 32 + c4, ek);
        stg4(tab_b + r * 32 + c4, make_float4(eq.x * scale, eq.y * scale, eq.z * scale, eq.w * scale));
    }
    for (int i = tid; i < TP; i += NTHR) rowmap[i] = token_row(i);
    __syncthreads();

    float kf[16], vf[16];
    auto load_k = [&](int kt, float *kd) {
        const float *p = qkv + (size_t)rowmap[32 * kt + qi] * ld + g.C + head * 32 + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
        }
    };
    auto load_v = [&](int kt, float *vd) {
        const float *vcol = qkv + 2 * g.C + head * 32 + qi;
#pragma unroll
        for (int s = 0; s < 16; ++s) vd[s] = vcol[(size_t)rowmap[32 * kt + mfma_row(s, hi)] * ld];
    };

    // ---- phase 0: relative-position logit terms -----------------------------------------------------
    if (tok_ok) {
        const float sc = hi ? 1.0f : scale;
#pragma unroll
        for (int c = 0; c < 32; ++c) vec[c] *= sc;
        const int pt = tok / nl;
        const int at = pt / win, bt = pt - at * win;
        const float *tab = hi ? tab_b : tab_a;
#pragma unroll 1
        for (int aj = 0; aj < win; ++aj)
#pragma unroll 2
            for (int bj = 0; bj < win; ++bj) {
                const int pj = aj * win + bj;
                // hi==0: this token is the query i, pj the key pixel:  rel(i,j) = (a_i-a_j, b_i-b_j)
                // hi==1: this token is the key j,  pj the query pixel: rel(i,j) = (a_pj-a_t, b_pj-b_t)
                const int da = hi ? (aj - at) : (at - aj), db = hi ? (bj - bt) : (bt - bj);
                const float *e = tab + ((da + win - 1) * span + (db + win - 1)) * 32;
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
    // Q fragment and the first K / V fragments: their latency overlaps the barrier and the ev stores
    float qf[16];
    {
        const float *p = qkv + trow * ld + head * 32 + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            qf[4 * c + 0] = v.x; qf[4 * c + 1] = v.y; qf[4 * c + 2] = v.z; qf[4 * c + 3] = v.w;
        }
    }
    load_k(0, kf);
    load_v(0, vf);
    __syncthreads();
    // ek is dead: overwrite it with ev
    for (int i = tid; i < R * 8; i += NTHR)
        stg4(tab_a + (i >> 3) * 32 + (i & 7) * 4, ldg4(table + (size_t)(i >> 3) * tab_ld + tcol + 64 + (i & 7) * 4));
    __syncthreads();

    // ---- phase 1: streaming softmax over the key tiles (flash style; S never leaves registers) ---------
    const int qs = tok;
    const bool q_ok = tok_ok;
    const int qsc = tokc;
    const int64_t qrow = trow;
    const int q_pix = qsc / nl;
    const int qa = q_pix / win, qb = q_pix - qa * win;
#pragma unroll
    for (int c = 0; c < 16; ++c) qf[c] *= scale;
    // Swin shift regions on the rolled grid (NMP.py:211-239): id = f(Y')*3 + f(X')
    auto region = [&](int a, int b) -> int {
        int Yr = wi * win + a, Xr = wj * win + b;
        int fy = Yr < g.Hp - win ? 0 : (Yr < g.Hp - g.shift ? 1 : 2);
        int fx = Xr < g.Wp - win ? 0 : (Xr < g.Wp - g.shift ? 1 : 2);
        return fy * 3 + fx;
    };
    // only windows in the last window row / column contain more than one region
    const bool need_shift = g.shift && (wi == g.Hp / win - 1 || wj == nwx - 1);
    const int q_reg = need_shift ? region(qa, qb) : 0;

    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float oe[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) oe[d] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const bool group4 = (nl & 3) == 0;                     // the 4 keys of one register quad share a pixel
    const float *qrt_q = qrt + qsc;
    const float *kr_q = kr + q_pix;
#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
        const int k0 = 32 * kt;
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        split_dot16(kf, qf, st);                            // S^T = K . Q^T on split-fp16 MFMA (split_mfma.h)
        if (kt + 1 < NKT) load_k(kt + 1, kf);               // K fragment is dead: refill it for the next tile now,
        //                                                     in flight during this tile's softmax and P.V
        float m_tile = -INFINITY;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            // keys of register quad rq: k0 + mfma_row(4rq,hi) + {0,1,2,3}
            const int kq = k0 + mfma_row(4 * rq, hi);
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const int r = 4 * rq + e4;
                const int key = kq + e4;
                const int keyc = key < Tw ? key : Tw - 1;
                const int pk = keyc / nl;
                float v = st[r] + qrt_q[pk * TP] + kr_q[keyc * W2];
                bool dead = key >= Tw;
                if (g.sibling) dead = dead || (pk == q_pix && keyc != qsc);
                if (need_shift) {
                    const int ka = pk / win, kb = pk - ka * win;
                    dead = dead || (region(ka, kb) != q_reg);
                }
                v = dead ? -INFINITY : v;
                st[r] = v;
                m_tile = fmaxf(m_tile, v);
            }
        }
        m_tile = half_max(m_tile);
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
        split_dot16(vf, st, acc_o);                         // O^T += V^T . P^T on split-fp16 MFMA
        if (kt + 1 < NKT) load_v(kt + 1, vf);               // same for V: in flight during the ev term and the next S^T
        // value-embedding term on the VALU: sum over key PIXELS of (sum_n p) * ev[rel(pq,pk)]
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int nsub = group4 ? 1 : 4;
            for (int e4 = 0; e4 < nsub; ++e4) {
                int key = k0 + mfma_row(4 * rq, hi) + e4;
                key = key < Tw ? key : Tw - 1;
                const int pk = key / nl;
                const int ka = pk / win, kb = pk - ka * win;
                float pp;
                if (group4) pp = (st[4 * rq] + st[4 * rq + 1]) + (st[4 * rq + 2] + st[4 * rq + 3]);
                else pp = e4 == 0 ? st[4 * rq] : (e4 == 1 ? st[4 * rq + 1] : (e4 == 2 ? st[4 * rq + 2] : st[4 * rq + 3]));
                const float *e = tab_a + ((qa - ka + win - 1) * span + (qb - kb + win - 1)) * 32;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float4 t = *reinterpret_cast<const float4 *>(e + 4 * c);
                    oe[4 * c + 0] = fmaf(pp, t.x, oe[4 * c + 0]); oe[4 * c + 1] = fmaf(pp, t.y, oe[4 * c + 1]);
                    oe[4 * c + 2] = fmaf(pp, t.z, oe[4 * c + 2]); oe[4 * c + 3] = fmaf(pp, t.w, oe[4 * c + 3]);
                }
            }
        }
    }
    const float l_tot = half_sum(l_run);
    const float inv_l = 1.0f / l_tot;

    // lane (q,hi) owns output channels d = mfma_row(r,hi); it needs its partner's oe at those channels and
    // owes the partner its own oe at mfma_row(r,1-hi).  Indices are compile-time on both sides of the select.
    // lane (q,hi) owns channels d = mfma_row(r,hi) = d0 + 4*hi; one permlane swap of (oe[d0], oe[d0+4]) leaves
    // {own, partner's} value of exactly that channel in the two registers of every lane
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float a = oe[mfma_row(r, 0)], b = oe[mfma_row(r, 1)];
        half_swap(a, b);
        res[r] = (acc_o[r] + (a + b)) * inv_l;
    }
    if (!q_ok) return;
    (void)qs;
    float *op = out + qrow * g.C + head * 32;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        stg4(op + mfma_row(4 * rb, hi), make_float4(res[4 * rb], res[4 * rb + 1], res[4 * rb + 2], res[4 * rb + 3]));
}


// =================================================================================================
// Fast path for the shipped configurations (compile-time window size WIN and labels per pixel NL in
// {1,2,4}).  Same algorithm as the generic kernel, restructured after the round-1 profiles
// (profiles/r01_pmc_attention_kernels_*.txt, tools/kernel_bench.py --which timing):
//   * WPB windows of the same head per block, NKT waves each, sharing the staged tables.  The census showed
//     that a 5-wave block never shares a CU with a second one (two waves land on one SIMD), so two windows
//     are put into ONE 10-wave block at <=168 VGPRs (6x6x4), and eight one-wave windows into one 8-wave
//     block (4x4x1, tables staged once instead of eight times).
//   * masks only where they can fire (sibling mask on the diagonal tile, out-of-window keys on the last
//     tile, shift regions on border windows), behind wave-uniform branches;
//   * KR stored transposed (one ds_read_b128 per key quad); table rows XOR-swizzled in 16-byte chunks
//     (conflict-free b128 reads, no padding); exp2 with log2(e) folded into the q / eq scales;
//   * half-wave exchanges by v_permlane32_swap (no ds_bpermute); the value-embedding accumulator is kept
//     in the O^T register layout (16 channels per lane, both pixels of a quad pair after one swap of the
//     pixel probabilities), so no cross-half reduction is left at the end;
//   * the scaled Q fragment lives in LDS when NKT > 1 (16 VGPRs less across the tile loop).
// =================================================================================================
#define WA_TROW 32                         // floats per staged table row
#define WA_LOG2E 1.4426950408889634f

// QAL (tools-only instantiation: one window per five-wave block, kv16 + MFMA phase 0): the parked Q fragments (20 KB) take over the ek
// table's region -- widened to their size -- once phase 0 is done with it, and ev goes where eq was: 77.6 KB per block instead of
// 93.7, meant to let TWO independent blocks share a CU.  Measured (profiles/r04e): the runtime reports 2 resident blocks, the hardware
// never runs two at once (a five-wave block has two waves on one SIMD; two blocks would need 4 x 160 VGPRs there), 67 vs 51 us --
// the product stays at two windows per ten-wave block.
template <int NKT, int WIN, int NL, int WPB, int PK = 1, bool QAL = false>
struct WinFastLds {
    static constexpr int W2 = WIN * WIN;
    static constexpr int R = (2 * WIN - 1) * (2 * WIN - 1);
    static constexpr int Tw = W2 * NL;
    // token stride of the QR^T / KR^T rows (PK windows side by side); NL == 1: every lane of a b128 phase reads a different KR^T row,
    // so the stride is padded to 4 (mod 16) floats -- 16 different 16-byte bank slots (a stride of 32 floats hit two of them)
    static constexpr int TS = (Tw * PK + 3) / 4 * 4 + (NL == 1 ? 4 : 0);
    // staged embedding tables: chunk-major [8 chunks][POSN row slots][4 floats]; row (ra, rb) of the (2 WIN - 1)^2 table sits in
    // slot tab_pos(ra, rb).  WIN == 4: slot mod 16 = (ra % 4) * 4 + rb % 4 -- the 16 pixels of a window (4 consecutive ra x 4
    // consecutive rb for any key pixel) read 16 different bank slots; otherwise slot = row: a b128 phase holds 4 consecutive
    // pixels (NL == 4), whose rows differ by 1..3 or wrap by SPAN - WIN + {1, 2}: distinct mod 16 as well.  (The first layout,
    // row-major with the chunks XOR-swizzled by row & 7, measured a conflict ratio of 0.40 / 0.47: rows 8 apart collided.)
    static constexpr int GB = (2 * WIN - 1 + 3) / 4;
    // (+ 1: consecutive chunks of one row then sit 16 bytes apart modulo the 256-byte bank row, so the eight lanes that stage a
    //  row write eight different slots; the reads at a fixed chunk keep their bijection)
    static constexpr int POSN = WIN == 4 ? GB * GB * 16 + 1 : R;
    static constexpr int TP = NKT * 32;
    static constexpr bool QLDS = NKT > 1;
    static constexpr int QSC = QLDS ? NKT * 16 * 64 : 0;                           // floats of the parked Q fragments of a window
    static constexpr int SLOT = 2 * W2 * TS + 32 + (QAL ? 0 : QSC) + TP;           // floats per window slot
    static constexpr int TABA = QAL && QSC > POSN * WA_TROW ? QSC : POSN * WA_TROW;   // floats of the first table region
    static constexpr size_t BYTES = (size_t)(TABA + POSN * WA_TROW + WPB * SLOT) * 4;
    static_assert(!QAL || WPB == 1, "aliased Q park: one window per block");
};

// TIMED: debug instantiation writing s_memtime stamps per wave and a per-block census (nmrf_debug_window_timing).
#define WA_STAMP(k) do { if (TIMED && lane == 0 && blockIdx.y == 0 && blockIdx.x < 64) \
        stamps[((size_t)blockIdx.x * NKT * WPB + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WA_CENSUS(k, v) do { if (TIMED && tid == 0) \
        stamps[(size_t)64 * NKT * WPB * 16 + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 3 + (k)] = (v); } while (0)

// PK = 2 (refinement windows: 16 tokens): TWO windows share a wave's 32-token tile -- tokens 0..15 of window 2k, 16..31 of window
// 2k+1; the off-diagonal quarters of S^T are masked.  Every per-lane phase (relative-position dots, softmax, value-embedding
// term) then works on 32 real tokens instead of 16, and one MFMA pair serves two windows.
// KV16: k | v arrive as split fp16 pairs (written so by the producing block kernel, include/nmrf_hip.h): the K fragment loads ARE the
// MFMA operands, V takes 16 v_perm_b32, and the scaled Q fragment is parked in LDS already split -- 104 VALU instructions less per
// key tile; the phase-0 operand of the key lanes is rebuilt as hi + lo once (2^-22 relative, the precision of the products).
// P0M (NL == 4): phase 0 -- the relative-position logit terms QR[i, pj] = s q_i . ek[rel(pi, pj)] and KR[j, pi] = s k_j . eq[rel(pi, pj)] -- on
// v_mfma_f32_4x4x4_16b_f16: 16 independent 4x4x4 products per wave (layout pinned by tools/ab/mfma4x4.hip on the MI355X: block b =
// lanes 4b..4b+3; A: lane 4b+i holds A_b[i][0..3]; B: lane 4b+j holds B_b[0..3][j]; D: lane 4b+j, register i = D_b[i][j]).  The four
// lanes of a block are the four labels of ONE pixel -- they need the same 36 table rows -- so block b multiplies (4 key pixels x 4
// channels of the table rows rel(P_b, .)) by (4 channels x the pixel's 4 tokens): lane j ends up with its own token's dot products
// for 4 key pixels, accumulated over the 8 channel chunks.  Split fp16 operands (3 products, table scaled by 2^10 so that its low
// parts are normal fp16 numbers).  Per lane and phase: 144 ds_read_b64 + 216 of these 8-cycle MFMAs instead of 288 ds_read_b128 + 576
// v_pk_fma_f32 -- the VALU form was bound by the LDS (11.5k array cycles per block of ten waves for the table rows, of a 12.4k phase).
template <int NKT, int WIN, int NL, int WPB, int OCC, bool TIMED = false, int PK = 1, bool KV16 = false, bool P0M = false, bool QAL = false>
__global__ __launch_bounds__(64 * NKT * WPB, OCC) void window_attn_fast_kernel(const float *__restrict__ qkv,
        const float *__restrict__ table, WinGeom g, float scale, float *__restrict__ out,
        unsigned long long *__restrict__ stamps = nullptr) {
    using L = WinFastLds<NKT, WIN, NL, WPB, PK, QAL>;
    static_assert(!QAL || (KV16 && P0M && L::QLDS), "aliased Q park: the kv16 / MFMA-phase-0 form");
    constexpr int TP = L::TP, W2 = L::W2, R = L::R, Tw = L::Tw, TS = L::TS;
    constexpr int TwE = Tw * PK;                    // tokens of a wave's tile
    static_assert(PK == 1 || (PK == 2 && NKT == 1 && NL == 1 && 2 * Tw <= 32), "two windows per tile: one key tile, one label");
    constexpr int NTHR = 64 * NKT * WPB;
    constexpr int SPAN = 2 * WIN - 1;
    constexpr int TAB_IT = (R * 8 + NTHR - 1) / NTHR;
    constexpr int PPQ = 4 / NL;                    // pixels per register quad (4 consecutive keys)
    constexpr bool QLDS = L::QLDS;
    static_assert(NL == 1 || NL == 2 || NL == 4, "fast path: labels per pixel must divide 4");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tab_a = smem;                            // ek (later: ev; QAL: later the parked Q fragments), chunk-major [8][POSN][4]
    float *tab_b = tab_a + L::TABA;                 // eq*s*log2e, same layout (QAL: later ev)
    float *tab_ev = QAL ? tab_b : tab_a;
    constexpr int POSN = L::POSN;
    auto tab_pos = [](int ra, int rb) -> int {      // row slot of table row (ra, rb), see WinFastLds
        return WIN == 4 ? ((ra >> 2) * L::GB + (rb >> 2)) * 16 + (ra & 3) * 4 + (rb & 3) : ra * SPAN + rb;
    };
    auto tab_off = [&](int r, int c) -> int {       // float offset of chunk c of table row r (row-major index)
        return (c * POSN + (WIN == 4 ? tab_pos(r / SPAN, r % SPAN) : r)) * 4;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int slot = wv / NKT, qt = wv - slot * NKT;   // window slot of the block, query tile inside the window
    const int qi = lane & 31, hi = lane >> 5;
    float *slot_mem = tab_b + POSN * WA_TROW + slot * L::SLOT;
    float *qrt = slot_mem;                          // [W2][TS]   QR^T : [key pixel][query token]
    float *krt = qrt + W2 * TS;                     // [W2][TS]+32 KR^T : [query pixel][key token]; the last tile's quad
    //                                                 reads may run past Tw (those keys are masked)
    float *qsc = QAL ? tab_a : krt + W2 * TS + 32;  // [NKT][4][64] float4: scaled Q fragments (QLDS only)
    unsigned *rowoff = reinterpret_cast<unsigned *>(krt + W2 * TS + 32 + (QAL ? 0 : L::QSC));   // [TP]

    const int nwx = g.Wp / WIN, nwin = nwx * (g.Hp / WIN);
    const int win_raw = (blockIdx.x * WPB + slot) * PK;      // first window of the wave; idle ones keep running (barriers) on window 0
    auto window_of = [&](int i) -> int {            // window of tile token i
        const int w = win_raw + (PK == 2 && i >= Tw ? 1 : 0);
        return w < nwin ? w : 0;
    };
    const int head = blockIdx.y, bimg = blockIdx.z;
    const unsigned ld = 3u * g.C;
    const int tab_ld = 3 * g.C;
    const int tcol = head * 96;                     // per head: [eq(32) | ek(32) | ev(32)]  (NMP.py:257-260)
    const float sc2 = scale * WA_LOG2E;

    auto token_row = [&](int i) -> unsigned {
        int ii = i < TwE ? i : TwE - 1;             // padded tokens alias the last real one (always masked)
        const int widx = window_of(ii);
        const int wj = widx % nwx, wi = widx / nwx;
        if (PK == 2 && ii >= Tw) ii -= Tw;
        int pt = ii / NL, n = ii - pt * NL;
        int a = pt / WIN, b = pt - a * WIN;
        int Y = wi * WIN + a + g.shift, X = wj * WIN + b + g.shift;
        Y = Y >= g.Hp ? Y - g.Hp : Y;
        X = X >= g.Wp ? X - g.Wp : X;
        return (unsigned)((((bimg * g.Hp + Y) * g.Wp) + X) * NL + n);
    };

    WA_STAMP(0);
    WA_CENSUS(0, (unsigned long long)__smid());
    WA_CENSUS(1, wall_clock64());
    // ---- earliest load: phase-0 operand (q or k of token 32*qt+qi) --------------------------------------
    const int tok = 32 * qt + qi;
    const bool tok_ok = tok < TwE;
    const int tokc = tok_ok ? tok : TwE - 1;
    const int tokl = PK == 2 && tokc >= Tw ? tokc - Tw : tokc;           // token inside its own window
    const bool win_ok = win_raw + (PK == 2 && tokc >= Tw ? 1 : 0) < nwin;  // (wave-uniform for PK == 1)
    const int my_w = window_of(tokc);
    const int wj = my_w % nwx, wi = my_w / nwx;     // window of this lane's token
    const unsigned trow = token_row(tokc);
    float vec[32];
    {
        const float *src = qkv + (size_t)trow * ld + (hi ? g.C : 0) + head * 32;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 v = ldg4(src + 4 * c);
            vec[4 * c + 0] = v.x; vec[4 * c + 1] = v.y; vec[4 * c + 2] = v.z; vec[4 * c + 3] = v.w;
        }
        if constexpr (KV16) {
            // key lanes hold the head's [32 hi halves | 32 lo halves]: channel c = half c % 2 of word c / 2 (hi), of word 16 + c / 2 (lo)
            auto dec = [&](int c) {                 // (float) hi + (float) lo of channel c
                const unsigned wh = __builtin_bit_cast(unsigned, vec[c >> 1]), wl = __builtin_bit_cast(unsigned, vec[16 + (c >> 1)]);
                const unsigned short hh = (unsigned short)((c & 1) ? (wh >> 16) : (wh & 0xffffu)), ll = (unsigned short)((c & 1) ? (wl >> 16) : (wl & 0xffffu));
                return (float)__builtin_bit_cast(_Float16, hh) + (float)__builtin_bit_cast(_Float16, ll);
            };
            // in place with 16 temporaries: channels 16 .. 31 first (words 8 .. 15 and 24 .. 31), then 15 .. 0 downwards -- channel c
            // lands in slot c, whose word (c < 16: hi word c, needed by channels 2c, 2c + 1 > c) has been consumed by then
            float up[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) up[c] = dec(16 + c);
#pragma unroll
            for (int c = 15; c >= 0; --c) {
                const float v = dec(c);
                vec[c] = hi ? v : vec[c];
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) vec[16 + c] = hi ? up[c] : vec[16 + c];
        }
    }
    // ---- stage ek / eq (all loads first, then the LDS stores), build the row map ----------------------
    static_assert(!P0M || (NL == 4 && WIN == 6 && PK == 1), "MFMA phase 0: four labels per pixel = one 4-lane block");
    constexpr float P0_SCALE = 1024.0f;            // tables are staged x 2^10 as split fp16 (entries up to 32 in magnitude; checked by the caller)
    // P0M layout of a table in its 15.5 KB: [hi | lo][8 chunks][R rows][4 halves]
    typedef _Float16 wa_h4 __attribute__((ext_vector_type(4)));
    {
        float4 te[TAB_IT], tq[TAB_IT];
#pragma unroll
        for (int it = 0; it < TAB_IT; ++it) {
            const int i = tid + it * NTHR;
            if (i < R * 8) {
                const int r = i >> 3, c4 = (i & 7) * 4;
                te[it] = ldg4(table + (size_t)r * tab_ld + tcol + 32 + c4);
                tq[it] = ldg4(table + (size_t)r * tab_ld + tcol + c4);
            }
        }
#pragma unroll
        for (int it = 0; it < TAB_IT; ++it) {
            const int i = tid + it * NTHR;
            if (i < R * 8) {
                if constexpr (P0M) {
                    auto put = [&](float *tab, float4 v, float mul) {
                        h16x2 h01, l01, h23, l23;
                        split2u(f32x2{v.x * mul, v.y * mul}, h01, l01);
                        split2u(f32x2{v.z * mul, v.w * mul}, h23, l23);
                        wa_h4 *th = reinterpret_cast<wa_h4 *>(tab) + (i & 7) * R + (i >> 3);
                        th[0] = wa_h4{h01[0], h01[1], h23[0], h23[1]};
                        th[8 * R] = wa_h4{l01[0], l01[1], l23[0], l23[1]};
                    };
                    put(tab_a, te[it], P0_SCALE);
                    put(tab_b, tq[it], P0_SCALE * sc2);
                } else {
                    const int sw = tab_off(i >> 3, i & 7);
                    stg4(tab_a + sw, te[it]);
                    stg4(tab_b + sw, make_float4(tq[it].x * sc2, tq[it].y * sc2, tq[it].z * sc2, tq[it].w * sc2));
                }
            }
        }
    }
    for (int i = tid - slot * NKT * 64; i < TP; i += NKT * 64) rowoff[i] = token_row(i) * ld;   // by the slot's own waves
    WA_STAMP(1);
    __syncthreads();
    WA_STAMP(2);

    // ev goes to registers now (hidden behind phase 0), to LDS once ek is dead
    float4 tv[TAB_IT];
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = tid + it * NTHR;
        if (i < R * 8) tv[it] = ldg4(table + (size_t)(i >> 3) * tab_ld + tcol + 64 + (i & 7) * 4);
    }
    // scaled Q fragment of this wave: lane (qi,hi) holds Q[tok][16*hi + s] * s * log2(e)
    float qf[16];
    {
        const float *p = qkv + (size_t)trow * ld + head * 32 + 16 * hi;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + 4 * c);
            qf[4 * c + 0] = v.x * sc2; qf[4 * c + 1] = v.y * sc2; qf[4 * c + 2] = v.z * sc2; qf[4 * c + 3] = v.w * sc2;
            if (QLDS && !KV16) stg4(qsc + ((qt * 4 + c) * 64 + lane) * 4, make_float4(qf[4 * c], qf[4 * c + 1], qf[4 * c + 2], qf[4 * c + 3]));
        }
        if constexpr (QLDS && KV16 && !QAL) {                          // parked as operand chunks: hi 0, hi 1, lo 0, lo 1
            h16x8 sh[2], sl[2];
            split8u(qf, sh[0], sl[0]);
            split8u(qf + 8, sh[1], sl[1]);
            *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 0) * 64 + lane) * 4) = sh[0];
            *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 1) * 64 + lane) * 4) = sh[1];
            *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 2) * 64 + lane) * 4) = sl[0];
            *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 3) * 64 + lane) * 4) = sl[1];
        }
    }

    // ---- phase 0: relative-position logit terms (log2 domain) -------------------------------------------
    if constexpr (P0M) {
        // every lane takes part (the matrix instruction ignores EXEC): lanes of a padded token compute on the clamped token and store nothing
        const float sc = hi ? 1.0f : sc2;
        wa_h4 vh[8], vl[8];                            // this token's vector as the B operand of the 8 channel chunks
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            h16x2 h01, l01, h23, l23;
            split2u(f32x2{vec[4 * c] * sc, vec[4 * c + 1] * sc}, h01, l01);
            split2u(f32x2{vec[4 * c + 2] * sc, vec[4 * c + 3] * sc}, h23, l23);
            vh[c] = wa_h4{h01[0], h01[1], h23[0], h23[1]};
            vl[c] = wa_h4{l01[0], l01[1], l23[0], l23[1]};
        }
        const int pt = tokl / NL;
        const int at = pt / WIN, bt = pt - at * WIN;
        const wa_h4 *tab = reinterpret_cast<const wa_h4 *>(hi ? tab_b : tab_a);
        float *dst = (hi ? krt : qrt) + tok;
        const int i4 = lane & 3;
#pragma unroll 3
        for (int gk = 0; gk < W2 / 4; ++gk) {              // key (hi == 0) / query (hi == 1) pixels 4 gk .. 4 gk + 3; this lane's A row: 4 gk + i4
            const int p = 4 * gk + i4;
            const int aj = p / WIN, bj = p - aj * WIN;
            const int da = hi ? (aj - at) : (at - aj), db = hi ? (bj - bt) : (bt - bj);
            const wa_h4 *row = tab + (da + WIN - 1) * SPAN + (db + WIN - 1);
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ah = {0.f, 0.f, 0.f, 0.f};      // cross terms / hi x hi: two chains
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const wa_h4 th = row[c * R], tl = row[(8 + c) * R];
                ax = __builtin_amdgcn_mfma_f32_4x4x4f16(tl, vh[c], ax, 0, 0, 0);
                ah = __builtin_amdgcn_mfma_f32_4x4x4f16(th, vh[c], ah, 0, 0, 0);
                ax = __builtin_amdgcn_mfma_f32_4x4x4f16(th, vl[c], ax, 0, 0, 0);
            }
            if (tok_ok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[(4 * gk + r) * TS] = (ah[r] + ax[r]) * (1.0f / P0_SCALE);
            }
        }
    } else
    if (tok_ok) {
        const float sc = hi ? 1.0f : sc2;
#pragma unroll
        for (int c = 0; c < 32; ++c) vec[c] *= sc;
        const int pt = tokl / NL;
        const int at = pt / WIN, bt = pt - at * WIN;
        const float *tab = hi ? tab_b : tab_a;
        float *dst = (hi ? krt : qrt) + tok;
#pragma unroll 1
        for (int aj = 0; aj < WIN; ++aj) {
            const int da = hi ? (aj - at) : (at - aj);
            const int rrow = (da + WIN - 1) * SPAN + (WIN - 1);
#pragma unroll 2
            for (int bj = 0; bj < WIN; ++bj) {
                const int db = hi ? (bj - bt) : (bt - bj);
                const float *e = tab + (WIN == 4 ? tab_pos(da + WIN - 1, db + WIN - 1) : rrow + db) * 4;
                f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};           // two packed accumulators: 16 v_pk_fma_f32 per dot
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 t = *reinterpret_cast<const float4 *>(e + c * (POSN * 4));
                    s0 = pk_fma(f32x2{vec[4 * c + 0], vec[4 * c + 1]}, f32x2{t.x, t.y}, s0);
                    s1 = pk_fma(f32x2{vec[4 * c + 2], vec[4 * c + 3]}, f32x2{t.z, t.w}, s1);
                }
                dst[(aj * WIN + bj) * TS] = (s0.x + s1.x) + (s0.y + s1.y);
            }
        }
    }
    WA_STAMP(3);
    // first K fragment: its latency overlaps the barrier and the ev stores
    const float *kbase = qkv + g.C + head * 32 + (KV16 ? 8 : 16) * hi;       // KV16: hi halves at 8 hi floats, lo halves 16 floats on
    const float *vbase = qkv + 2 * g.C + head * 32 + qi;
    float kf[16], vf[16];
    auto load_k = [&](int kt, float *kd) {
        const float *p = kbase + rowoff[32 * kt + qi];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float4 v = ldg4(p + (KV16 ? (c < 2 ? 4 * c : 8 + 4 * c) : 4 * c));
            kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
        }
    };
    auto load_v = [&](int kt, float *vd) {
#pragma unroll
        for (int s = 0; s < 16; ++s) vd[s] = vbase[rowoff[32 * kt + mfma_row(s, hi)]];
    };
    load_k(0, kf);
    __syncthreads();
    WA_STAMP(4);
#pragma unroll
    for (int it = 0; it < TAB_IT; ++it) {
        const int i = tid + it * NTHR;
        if (i < R * 8) stg4(tab_ev + tab_off(i >> 3, i & 7), tv[it]);
    }
    if constexpr (QAL) {                                               // ek is dead: its region takes the Q fragments (held in registers across phase 0)
        h16x8 sh[2], sl[2];
        split8u(qf, sh[0], sl[0]);
        split8u(qf + 8, sh[1], sl[1]);
        *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 0) * 64 + lane) * 4) = sh[0];
        *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 1) * 64 + lane) * 4) = sh[1];
        *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 2) * 64 + lane) * 4) = sl[0];
        *reinterpret_cast<h16x8 *>(qsc + ((qt * 4 + 3) * 64 + lane) * 4) = sl[1];
    }
    __syncthreads();
    WA_STAMP(5);

    // ---- phase 1 ---------------------------------------------------------------------------------------
    const int q_pix = tokl / NL;
    const int qa = q_pix / WIN, qb = q_pix - qa * WIN;
    auto local_key = [&](int key) -> int { return PK == 2 && key >= Tw ? key - Tw : key; };    // key inside its own window
    auto region = [&](int a, int b) -> int {       // Swin shift regions on the rolled grid (NMP.py:211-239)
        int Yr = wi * WIN + a, Xr = wj * WIN + b;
        int fy = Yr < g.Hp - WIN ? 0 : (Yr < g.Hp - g.shift ? 1 : 2);
        int fx = Xr < g.Wp - WIN ? 0 : (Xr < g.Wp - g.shift ? 1 : 2);
        return fy * 3 + fx;
    };
    const bool on_border = g.shift && (wi == g.Hp / WIN - 1 || wj == nwx - 1);
    const bool need_shift = PK == 1 ? on_border : (__builtin_amdgcn_ballot_w64(on_border) != 0);     // wave-uniform
    const int q_reg = need_shift ? region(qa, qb) : 0;
    const bool sib = g.sibling && NL > 1;

    f32x16 acc_o;                                   // O^T[d = mfma_row(r,hi)][q] from the MFMAs
    f32x2 oe[8];                                    // value-embedding term in the SAME layout (register pairs)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) oe[r] = f32x2{0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float *qrt_q = qrt + tokc;
    const float *krt_q = krt + q_pix * TS;
    // rel(pq, pixel 0), plus this half's chunk offset (its chunks are 2 gq + hi) in row-slot units
    const int ev_r0 = (qa + WIN - 1) * SPAN + (qb + WIN - 1) + hi * POSN;

#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
        const int k0 = 32 * kt;
        load_v(kt, vf);                             // needed only after S^T + softmax (~1.5k cycles from here); the K
        //                                             fragment of the NEXT tile is fetched right after this tile's S^T, so
        //                                             only one of the two 16-register fragments is live during the ev term
        if (QLDS && !KV16) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = *reinterpret_cast<const float4 *>(qsc + ((qt * 4 + c) * 64 + lane) * 4);
                qf[4 * c + 0] = v.x; qf[4 * c + 1] = v.y; qf[4 * c + 2] = v.z; qf[4 * c + 3] = v.w;
            }
        }
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        if constexpr (KV16) {                               // every operand chunk arrives split: no arithmetic before the MFMAs
#pragma unroll
            for (int c = 0; c < 2; ++c) {                   // (one Q chunk in registers at a time: the kernel sits at its register cap)
                const h16x8 kh = __builtin_bit_cast(h16x8, f32x4{kf[4 * c], kf[4 * c + 1], kf[4 * c + 2], kf[4 * c + 3]});
                const h16x8 kl = __builtin_bit_cast(h16x8, f32x4{kf[8 + 4 * c], kf[9 + 4 * c], kf[10 + 4 * c], kf[11 + 4 * c]});
                h16x8 qh2, ql2;
                if constexpr (QLDS) {
                    qh2 = *reinterpret_cast<const h16x8 *>(qsc + ((qt * 4 + c) * 64 + lane) * 4);
                    ql2 = *reinterpret_cast<const h16x8 *>(qsc + ((qt * 4 + 2 + c) * 64 + lane) * 4);
                } else {
                    split8u(qf + 8 * c, qh2, ql2);
                }
                split_mma1(kh, kl, qh2, ql2, st);
                if (c == 0) __builtin_amdgcn_sched_barrier(0);
            }
        } else
        split_dot16(kf, qf, st);                            // S^T = K . Q^T on split-fp16 MFMA (split_mfma.h)
        if (kt + 1 < NKT) load_k(kt + 1, kf);
        // relative-position terms: one b128 of KR^T per key quad, QR^T per key pixel
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int keyq = k0 + 8 * rq + 4 * hi;                           // first key of the quad (multiple of 4)
            const float4 kr4 = *reinterpret_cast<const float4 *>(krt_q + keyq);
            const float krv[4] = {kr4.x, kr4.y, kr4.z, kr4.w};
#pragma unroll
            for (int pp = 0; pp < PPQ; ++pp) {
                int pk = local_key(keyq) / NL + pp;
                pk = pk < W2 ? pk : W2 - 1;
                const float qv = qrt_q[pk * TS];
#pragma unroll
                for (int e = 0; e < NL; ++e) st[4 * rq + pp * NL + e] += qv + krv[pp * NL + e];
            }
        }
        if (kt == NKT - 1 && TwE < TP) {                                     // keys beyond the window (last tile only)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + mfma_row(r, hi) >= TwE) st[r] = -INFINITY;
        }
        if constexpr (PK == 2) {                                              // keys of the other window of the tile
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((mfma_row(r, hi) >= Tw) != (tokc >= Tw)) st[r] = -INFINITY;
        }
        if (sib && kt == qt) {                                                // sibling labels of the query's own pixel
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + mfma_row(r, hi);
                if (key / NL == q_pix && key != tokc) st[r] = -INFINITY;
            }
        }
        if (need_shift) {                                                     // Swin regions (border windows only)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq)
#pragma unroll
                for (int pp = 0; pp < PPQ; ++pp) {
                    int pk = local_key(k0 + 8 * rq + 4 * hi) / NL + pp;
                    pk = pk < W2 ? pk : W2 - 1;
                    const bool other = region(pk / WIN, pk % WIN) != q_reg;
#pragma unroll
                    for (int e = 0; e < NL; ++e)
                        if (other) st[4 * rq + pp * NL + e] = -INFINITY;
                }
        }
        float m_tile = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
        m_tile = half_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = __builtin_amdgcn_exp2f(st[r] - m_use);
            psum += st[r];
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
#pragma unroll
        for (int r = 0; r < 8; ++r) oe[r] *= alpha;
        if constexpr (KV16) {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = st[r];
            h16x8 ph[2], pl[2], vh[2], vl[2];
            split8u(pv, ph[0], pl[0]);
            split8u(pv + 8, ph[1], pl[1]);
            kv16_chunks(vf, vh, vl);
            split_mma1(vh[0], vl[0], ph[0], pl[0], acc_o);
            split_mma1(vh[1], vl[1], ph[1], pl[1], acc_o);
        } else
        split_dot16(vf, st, acc_o);                         // O^T += V^T . P^T on split-fp16 MFMA
        // value-embedding term: sum over key PIXELS of (sum_n p) * ev[rel(pq,pk)].  The two half-lanes of a query
        // swap the probabilities of their pixels, then each accumulates BOTH pixels for its own 16 channels.
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            if (kt == NKT - 1 && 32 * (NKT - 1) + 8 * rq >= TwE) continue;    // quad pair entirely beyond the window
#pragma unroll
            for (int pp = 0; pp < PPQ; ++pp) {
                float p0 = st[4 * rq + pp * NL];
#pragma unroll
                for (int e = 1; e < NL; ++e) p0 += st[4 * rq + pp * NL + e];
                float p1 = p0;
                half_swap(p0, p1);                 // p0 = probability mass of the hi=0 lane's pixel, p1 = of the hi=1 lane's
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    int pk = local_key(k0 + 8 * rq + 4 * h2) / NL + pp;
                    pk = pk < W2 ? pk : W2 - 1;
                    const int ka = pk / WIN, kb = pk - ka * WIN;
                    const float *e = tab_ev + (WIN == 4 ? tab_pos(qa - ka + WIN - 1, qb - kb + WIN - 1) + hi * POSN : ev_r0 - (ka * SPAN + kb)) * 4;
                    const float psv = h2 ? p1 : p0;
                    const f32x2 ps = {psv, psv};
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {               // channels 8*gq + 4*hi .. +3  = O^T registers 4*gq .. 4*gq+3
                        const float4 t = *reinterpret_cast<const float4 *>(e + 2 * gq * (POSN * 4));
                        oe[2 * gq + 0] = pk_fma(ps, f32x2{t.x, t.y}, oe[2 * gq + 0]);
                        oe[2 * gq + 1] = pk_fma(ps, f32x2{t.z, t.w}, oe[2 * gq + 1]);
                    }
                }
            }
            // (the table reads carry immediate offsets now: without a fence the scheduler hoists all 32 of a tile and spills)
            if (NKT > 1) __builtin_amdgcn_sched_barrier(0);
        }
        WA_STAMP(6 + kt);
    }
    const float inv_l = 1.0f / half_sum(l_run);
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = (acc_o[r] + ((r & 1) ? oe[r >> 1].y : oe[r >> 1].x)) * inv_l;
    if (TIMED) asm volatile("" :: "v"(res[0]), "v"(res[15]));
    WA_STAMP(6 + NKT);
    if (!tok_ok || !win_ok) return;
    float *op = out + (size_t)trow * g.C + head * 32;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
        stg4(op + mfma_row(4 * rb, hi), make_float4(res[4 * rb], res[4 * rb + 1], res[4 * rb + 2], res[4 * rb + 3]));
    WA_STAMP(7 + NKT);
    WA_CENSUS(2, wall_clock64());
}

template <int NKT, int WIN, int NL, int WPB, int OCC, int PK = 1, bool KV16 = false, bool P0M = false, bool QAL = false>
static int launch_window_fast(const float *qkv, const float *table, const WinGeom &g, int B, float *out, hipStream_t st) {
    using L = WinFastLds<NKT, WIN, NL, WPB, PK, QAL>;
    static_assert(L::BYTES <= 160 * 1024, "LDS budget of one CU");
    static bool attr_set_dev[NMRF_MAX_DEV] = {};      // set once per instantiation and device, outside any stream capture
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    if (L::BYTES > 64 * 1024 && !attr_set_dev[dev]) {
        if (hipFuncSetAttribute((const void *)window_attn_fast_kernel<NKT, WIN, NL, WPB, OCC, false, PK, KV16, P0M, QAL>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    const int nwin = (g.Hp / WIN) * (g.Wp / WIN);
    dim3 grid((nwin + WPB * PK - 1) / (WPB * PK), g.heads, B);
    hipLaunchKernelGGL((window_attn_fast_kernel<NKT, WIN, NL, WPB, OCC, false, PK, KV16, P0M, QAL>), grid, dim3(64 * NKT * WPB), L::BYTES, st, qkv,
                       table, g, 1.0f / sqrtf(32.0f), out, (unsigned long long *)nullptr);
    return nmrf_launch_status();
}

template <int NKT>
static int launch_window(const float *qkv, const float *table, const WinGeom &g, int B, float *out, hipStream_t st) {
    const int TP = NKT * 32, W2 = g.win * g.win;
    size_t smem = (size_t)(2 * g.R * 32 + 2 * W2 * TP) * sizeof(float) + (size_t)TP * sizeof(int);
    if (smem > 160 * 1024) return NMRF_EINVAL;
    static size_t attr_smem_dev[NMRF_MAX_DEV] = {};   // largest request so far per device (the size depends on the geometry)
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    if (smem > 64 * 1024 && smem > attr_smem_dev[dev]) {
        if (hipFuncSetAttribute((const void *)window_attn_kernel<NKT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_smem_dev[dev] = smem;
    }
    dim3 grid((g.Hp / g.win) * (g.Wp / g.win), g.heads, B);
    hipLaunchKernelGGL((window_attn_kernel<NKT>), grid, dim3(64 * NKT), smem, st, qkv, table, g,
                       1.0f / sqrtf(32.0f), out);
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES   // tools-only library libnmrf_hip_debug.so (python -m nmrf_amd.build --debug)
// Debug helper (not part of the public header): what the HIP runtime thinks the residency of the two fast
// instantiations is (the census of nmrf_debug_window_timing shows what the hardware actually does).
extern "C" int nmrf_debug_window_occupancy(int *blocks_infer, int *blocks_refine) {
    using L5 = WinFastLds<5, 6, 4, 2>;
    using L1 = WinFastLds<1, 4, 1, 8>;
    hipFuncSetAttribute((const void *)window_attn_fast_kernel<5, 6, 4, 2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L5::BYTES);
    hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_infer, window_attn_fast_kernel<5, 6, 4, 2, 3>, 640, L5::BYTES);
    hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_refine, window_attn_fast_kernel<1, 4, 1, 8, 2>, 512, L1::BYTES);
    return (e1 == hipSuccess && e2 == hipSuccess) ? 0 : -2;
}

// Debug: instrumented run of the 6x6x4 kernel; stamps [64 blocks][10 waves][16] of s_memtime values followed by the
// census [blocks][3] (device buffer).
extern "C" int nmrf_debug_window_timing(const float *qkv, const float *table, int B, int Hp, int Wp, int shift, float *out,
                                        unsigned long long *stamps, void *stream) {
    // (qkv: kv16 rows -- the product instantiation, phase 0 on the 4x4x4 MFMA; shift >= 100: phase 0 on the VALU, shift - 100)
    using L = WinFastLds<5, 6, 4, 2>;
    const bool valu = shift >= 100 && shift < 200, one = shift >= 200;
    WinGeom g{Hp, Wp, 4, 128, 4, 6, shift % 100, 1, 144, 121};
    const int nwin = (Hp / 6) * (Wp / 6);
    dim3 grid((nwin + 1) / 2, 4, B);
    if (one) {                                       // one window per five-wave block, Q park aliased (77.6 KB of LDS)
        using L1 = WinFastLds<5, 6, 4, 1, 1, true>;
        hipFuncSetAttribute((const void *)window_attn_fast_kernel<5, 6, 4, 1, 3, true, 1, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L1::BYTES);
        int occ = -1;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, window_attn_fast_kernel<5, 6, 4, 1, 3, true, 1, true, true, true>, 320, L1::BYTES);
        printf("[window timing] one-window form: %zu bytes of LDS, runtime occupancy %d blocks per CU\n", (size_t)L1::BYTES, occ);
        hipLaunchKernelGGL((window_attn_fast_kernel<5, 6, 4, 1, 3, true, 1, true, true, true>), dim3(nwin, 4, B), dim3(320), L1::BYTES, (hipStream_t)stream, qkv,
                           table, g, 1.0f / sqrtf(32.0f), out, stamps);
        return nmrf_launch_status();
    }
    if (valu) {
        hipFuncSetAttribute((const void *)window_attn_fast_kernel<5, 6, 4, 2, 3, true, 1, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
        hipLaunchKernelGGL((window_attn_fast_kernel<5, 6, 4, 2, 3, true, 1, true, false>), grid, dim3(640), L::BYTES, (hipStream_t)stream, qkv, table, g,
                           1.0f / sqrtf(32.0f), out, stamps);
    } else {
        hipFuncSetAttribute((const void *)window_attn_fast_kernel<5, 6, 4, 2, 3, true, 1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
        hipLaunchKernelGGL((window_attn_fast_kernel<5, 6, 4, 2, 3, true, 1, true, true>), grid, dim3(640), L::BYTES, (hipStream_t)stream, qkv, table, g,
                           1.0f / sqrtf(32.0f), out, stamps);
    }
    return nmrf_launch_status();
}
#endif  // NMRF_DEBUG_PROBES

#ifdef NMRF_DEBUG_PROBES
static int g_window_pack1 = 0;      // tools: run the refinement windows one per tile (A/B of the packing)
extern "C" int nmrf_debug_window_pack1(int v) { g_window_pack1 = v; return NMRF_OK; }
#endif

// fp16 range of the q | k | v operand (include/nmrf_hip.h).  The two fast kernels sit at their VGPR cap (164-167 of 168: ONE more
// live value spills 258 registers and the refinement kernel runs 6x slower -- measured), so they carry no guard accumulator.  The
// product's qkv comes from nmrf_nmp_block16_f32, which range-checks its q_out; for any other producer the entry point scans the
// operand in a pass of its own (one extra read of qkv) when a flag is given.
__global__ __launch_bounds__(256) void range_scan_kernel(const float4 *__restrict__ v, int64_t n4, int *__restrict__ flag) {
    float m = 0.f, z = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 t = v[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
        z += (t.x + t.y + t.z + t.w) * 0.f;                          // NaN / inf anywhere -> NaN (fmax drops NaN operands)
    }
    if (!(m < 65520.0f) || z != 0.f) atomicOr(flag, 1);
}

extern "C" int nmrf_range_scan_f32(const float *x, int64_t n, int *range_flag, void *stream) {
    if (!x || !range_flag) return NMRF_ENULL;
    if (n < 0 || (n & 3) || (reinterpret_cast<uintptr_t>(x) & 15)) return NMRF_EINVAL;
    if (n == 0) return NMRF_OK;
    int64_t blocks = ceil_div64(n / 4, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(range_scan_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(x), n / 4, range_flag);
    return nmrf_launch_status();
}

extern "C" int nmrf_window_attn_f32(const float *qkv, const float *table, int B, int Hp, int Wp, int N, int C, int heads,
                                    int win, int shift, int sibling_mask, int kv16, float *out, int *range_flag, void *stream) {
    if (!qkv || !table || !out) return NMRF_ENULL;
    if (kv16 && (win != 6 || N != 4 || range_flag)) return NMRF_EINVAL;     // the pre-split form: 6 x 6 x 4 inference windows, range checked by the producer
    if (B < 1 || N < 1 || win < 1 || Hp % win || Wp % win || shift < 0 || shift >= win || heads * 32 != C || (C & 3))
        return NMRF_EINVAL;
    if ((int64_t)B * Hp * Wp * N >= (int64_t)1 << 31) return NMRF_EINVAL;
    WinGeom g{Hp, Wp, N, C, heads, win, shift, sibling_mask ? 1 : 0, win * win * N, (2 * win - 1) * (2 * win - 1)};
    const int nkt = (g.Tw + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
    if (range_flag) {
        const int rc = nmrf_range_scan_f32(qkv, (int64_t)B * Hp * Wp * N * 3 * C, range_flag, stream);
        if (rc != NMRF_OK) return rc;
    }
    if ((int64_t)B * Hp * Wp * N * 3 * C < ((int64_t)1 << 32)) {                            // 32-bit element offsets
#ifdef NMRF_DEBUG_PROBES
        if (win == 6 && N == 4 && kv16 && g_window_pack1 == 10) return launch_window_fast<5, 6, 4, 2, 3, 1, true>(qkv, table, g, B, out, st);   // A/B: phase 0 on the VALU
#endif
#ifdef NMRF_DEBUG_PROBES
        if (win == 6 && N == 4 && kv16 && g_window_pack1 == 11) return launch_window_fast<5, 6, 4, 1, 3, 1, true, true, true>(qkv, table, g, B, out, st);   // A/B: ONE window per block
#endif
        // kv16 == 2: phase 0 (the relative-position dot products) on the VALU in fp32 -- for a table with an entry >= 32 in magnitude,
        // which the x 2^10 fp16 staging of the matrix-pipe form (P0M) cannot hold; same key tiles, ~4 us slower per launch
        if (win == 6 && N == 4 && kv16 == 2) return launch_window_fast<5, 6, 4, 2, 3, 1, true>(qkv, table, g, B, out, st);
        if (win == 6 && N == 4 && kv16) return launch_window_fast<5, 6, 4, 2, 3, 1, true, true>(qkv, table, g, B, out, st);
        if (win == 6 && N == 4) return launch_window_fast<5, 6, 4, 2, 3>(qkv, table, g, B, out, st);   // inference windows
#ifdef NMRF_DEBUG_PROBES
        if (win == 4 && N == 1 && g_window_pack1 == 1) return launch_window_fast<1, 4, 1, 8, 2, 1>(qkv, table, g, B, out, st);
        if (win == 4 && N == 1 && g_window_pack1 == 2) return launch_window_fast<1, 4, 1, 8, 2, 2>(qkv, table, g, B, out, st);
        if (win == 4 && N == 1 && g_window_pack1 == 3) return launch_window_fast<1, 4, 1, 2, 3, 2>(qkv, table, g, B, out, st);
        if (win == 4 && N == 1 && g_window_pack1 == 4) return launch_window_fast<1, 4, 1, 8, 1, 2>(qkv, table, g, B, out, st);
#endif
        // refinement windows, two per tile, four tiles per block (three blocks per CU: 24.0 us; eight tiles per block: 26.6 us)
        if (win == 4 && N == 1) return launch_window_fast<1, 4, 1, 4, 3, 2>(qkv, table, g, B, out, st);
    }
    if (kv16) return NMRF_EINVAL;     // pre-split k | v rows are only understood by the fast kernel above (32-bit offsets): the generic
                                      // kernel would read the fp16 pairs as floats
    switch (nkt) {                                                                         // any other configuration
        case 1: return launch_window<1>(qkv, table, g, B, out, st);
        case 2: return launch_window<2>(qkv, table, g, B, out, st);
        case 3: return launch_window<3>(qkv, table, g, B, out, st);
        case 4: return launch_window<4>(qkv, table, g, B, out, st);
        case 5: return launch_window<5>(qkv, table, g, B, out, st);
        case 6: return launch_window<6>(qkv, table, g, B, out, st);
        default: return NMRF_EINVAL;
    }
}
