// A10(ii), the shipped inference configuration: 6 x 6 windows of four labels per pixel (144 tokens), heads of 32 channels, k | v rows
// pre-split by their producer (kv16, include/nmrf_hip.h).  Persistent form of window_attn.hip's kernel (round 6):
//
//   logits_ij = s q_i.k_j + s q_i.ek[rel(pi,pj)] + s k_j.eq[rel(pi,pj)]          (WindowAttention.forward, nmrf/models/NMP.py:185-289)
//   out_i     = sum_j softmax_j(logits) (v_j + ev[rel(pi,pj)])
//
// What the profiles of round 5 said about the two-windows-per-block kernel (profiles/HISTORY.md 10.2): of a block's 46.8k cycles, 8.8k
// stage the three embedding tables, 4.6k wait at barriers around the LDS round trip of the relative-position logit terms, and 416 blocks
// of 156 KB make 1.6 rounds on 256 CUs.  Here
//   * a block belongs to ONE head and keeps that head's tables in LDS for its whole life: they arrive pre-packed (split fp16 pairs of
//     ek / eq x 2^10, ev in fp32, in exactly the LDS layout -- nmrf_window_table_pack_f32, once per parameter version), so staging
//     is one linear copy of 46 KB per block instead of a gather + conversion per pair of windows;
//   * the work item is one QUERY TILE of a window (32 tokens = 8 pixels), one wave each, and waves never talk to each other: the
//     relative-position terms a wave needs are formed just in time, per key tile, on v_mfma_f32_4x4x4_16b_f16 (16 independent 4x4x4
//     products: the four lanes of a block are the four labels of one pixel, which share their table rows) in two passes --
//       pass Q (every lane a query of the tile):  QR[i][p] = s q_i . ek[rel(pi, p)]   for key pixels  p = 4 half + 0..3 of the key tile,
//       pass K (every lane a key of the tile):    KR[j][p] = s k_j . eq[rel(p, pj)]   for query pixels p = 4 half + 0..3 of the wave's tile
//     (half = lane / 32), and land in the registers of exactly the lane that owns them as an MFMA operand.  They enter S^T = K Q^T as
//     ONE more 16-deep chunk of the contraction with one-hot partners: k slots 8 half + 0..3 carry [key's pixel == p] x QR[i][p],
//     slots 8 half + 4..7 carry KR[j][p] x [query's pixel == p]: two v_mfma_f32_32x32x16_f16 (hi and lo parts of QR / KR; the one-hot
//     side is exact) replace the LDS round trip, its barrier, and 2 VALU adds + 2 LDS reads per 4 logits;
//   * the operands are held ONCE: a lane keeps all 32 channels of its query (split, scaled) and of its key slot, the half-1 lanes with
//     the two 16-channel halves exchanged (they read the table chunks in the matching order), so that the first 16 channels of either
//     array ARE the lane's fragment of K Q^T -- no separate Q / K fragments, no parked copies;
//   * every global load is issued a whole key tile ahead into registers that have just become free (k rows behind K Q^T, v behind
//     P V): nothing the loop waits for was requested less than ~2k cycles earlier, no branch stands between a load and its wait;
//   * no barrier after the table copy; the streaming softmax rescales its accumulators only when some row's maximum moved by more than
//     2^8 (wave-uniform branch).
// Masks (sibling labels, Swin shift regions, keys beyond the window) and the value-embedding term are the predecessor's.
#include "common.h"
#include "split_mfma.h"

typedef _Float16 w6_h4 __attribute__((ext_vector_type(4)));

#define W6_R 121                                   // (2 * 6 - 1)^2 table rows
#define W6_PART (8 * W6_R * 8)                     // bytes of one [8 chunks][121 rows][4 halves] part = 7744
#define W6_HEAD_USED (4 * W6_PART + 8 * W6_R * 16)    // ek [8][121]{hi, lo} | eq [8][121]{hi, lo} | ev fp32 [8][121][4] = 46464
#define W6_HEAD_BYTES 49152                        // ... padded to a whole number of 16-byte copies per thread of an 8- or 12-wave block
#define W6_LOG2E 1.4426950408889634f
#define W6_P0_SCALE 1024.0f                        // tables are packed x 2^10 (their low parts are normal fp16 numbers; |table| < 32, checked by the caller)
#define W6_LAZY 8.0f                               // log2 units a row maximum may grow before the accumulators are rescaled (p <= 2^8)

// ---- table packing: table [121, 3C] (per head 96 columns eq | ek | ev) -> heads x W6_HEAD_BYTES --------------------------------------
__global__ __launch_bounds__(256) void window_table_pack_kernel(const float *__restrict__ table, int tab_ld, int heads, float sc2,
                                                                 char *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (row r, chunk c) of head blockIdx.y
    if (i >= W6_R * 8) return;
    const int r = i >> 3, c = i & 7, head = blockIdx.y;
    const float *src = table + (size_t)r * tab_ld + head * 96 + 4 * c;
    const float4 tq = ldg4(src), te = ldg4(src + 32), tv = ldg4(src + 64);
    char *base = out + (size_t)head * W6_HEAD_BYTES;
    auto put = [&](char *part, float4 v, float mul) {          // [chunk][row]{hi x 4, lo x 4}: one 16-byte read per (chunk, row)
        h16x2 h01, l01, h23, l23;
        split2u(f32x2{v.x * mul, v.y * mul}, h01, l01);
        split2u(f32x2{v.z * mul, v.w * mul}, h23, l23);
        w6_h4 *th = reinterpret_cast<w6_h4 *>(part) + 2 * (c * W6_R + r);
        th[0] = w6_h4{h01[0], h01[1], h23[0], h23[1]};
        th[1] = w6_h4{l01[0], l01[1], l23[0], l23[1]};
    };
    put(base, te, W6_P0_SCALE);
    put(base + 2 * W6_PART, tq, W6_P0_SCALE * sc2);
    reinterpret_cast<float4 *>(base + 4 * W6_PART)[c * W6_R + r] = tv;
}

extern "C" int nmrf_window_table_pack_f32(const float *table, int C, int heads, void *packed, void *stream) {
    if (!table || !packed) return NMRF_ENULL;
    if (heads < 1 || heads * 32 != C) return NMRF_EINVAL;
    hipLaunchKernelGGL(window_table_pack_kernel, dim3((W6_R * 8 + 255) / 256, heads), dim3(256), 0, (hipStream_t)stream, table, 3 * C, heads,
                       (1.0f / sqrtf(32.0f)) * W6_LOG2E, (char *)packed);
    return nmrf_launch_status();
}

struct Win6Args {
    const float *qkv;
    const char *tabp;
    float *out;
    int B, Hp, Wp, heads, shift, sibling;
    int nwx, nwin;           // windows per row / per image
    int nblk_head;           // blocks per head
    unsigned long long *stamps;   // tools build: [block][wave][64] s_memtime values of the wave's first item
};
#define W6_STAMP(k) do { if (TIMED && lane == 0 && q == wv) g.stamps[((size_t)blockIdx.x * NW + wv) * 64 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)

__device__ __forceinline__ int w6_x11(int p) { return p + 5 * ((p * 43) >> 8); }       // pixel p = 6 a + b of a window -> 11 a + b

// ABL (tools build only): timing ablations -- bit 0: no value-embedding term, 1: no 4x4x4 passes, 2: no P V, 3: no K Q^T (+ one-hot chunk),
// 4: no exponentials, 5: K / V rows loaded once per item.  Results are wrong by construction; nmrf_debug_window6_variant.
// TIMED (tools build): s_memtime stamps per wave and phase into g.stamps.
template <int NW, int OCC, int ABL = 0, bool TIMED = false>
__global__ __launch_bounds__(64 * NW, OCC) void window_attn6_kernel(Win6Args g) {
    constexpr int C = 128, LD = 3 * C;                       // four heads of 32 channels (the launcher checks)
    constexpr int R = W6_R;
    __shared__ __attribute__((aligned(16))) char smem[W6_HEAD_BYTES + NW * 40 * 8];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int qi = lane & 31, half = lane >> 5;
    const int head = blockIdx.x % g.heads, bih = blockIdx.x / g.heads;

    {   // the head's packed tables: one linear copy
        const uint4 *src = reinterpret_cast<const uint4 *>(g.tabp + (size_t)head * W6_HEAD_BYTES);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        static_assert(W6_HEAD_BYTES % (64 * NW * 16) == 0 && W6_HEAD_USED <= W6_HEAD_BYTES, "whole copies per thread");
        constexpr int IT = W6_HEAD_BYTES / (64 * NW * 16);
        uint4 t[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) t[it] = src[tid + it * 64 * NW];
#pragma unroll
        for (int it = 0; it < IT; ++it) dst[tid + it * 64 * NW] = t[it];
    }
    // register chunk c of a lane holds channels 4 ((c + 4 half) % 8) .. + 3: the table chunk it meets is c + 4 half for c < 4, c - 4 half beyond
    const uint4 *tab_ek = reinterpret_cast<const uint4 *>(smem), *tab_eq = reinterpret_cast<const uint4 *>(smem + 2 * W6_PART);
    const int rotA = half * 4 * R, rotB = -half * 4 * R;
    const float *tab_ev = reinterpret_cast<const float *>(smem + 4 * W6_PART);
    uint2 *pix = reinterpret_cast<uint2 *>(smem + W6_HEAD_BYTES) + wv * 40;   // per wave: {byte offset of the qkv row of label 0, shift region} of the window's pixels
    __syncthreads();

    const float sc2 = (1.0f / sqrtf(32.0f)) * W6_LOG2E;
    const int i4 = lane & 3, pb = qi >> 2;                       // lane of its 4-lane block; pixel of the lane's token inside its 32-token tile
    // one-hot partner of the relative-position chunk: element e = [pixel-in-tile of this lane's token == 4 half + e]  (two registers)
    const h16x2 oh01 = {(_Float16)(pb == 4 * half + 0 ? 1.0f : 0.0f), (_Float16)(pb == 4 * half + 1 ? 1.0f : 0.0f)};
    const h16x2 oh23 = {(_Float16)(pb == 4 * half + 2 ? 1.0f : 0.0f), (_Float16)(pb == 4 * half + 3 ? 1.0f : 0.0f)};
    const unsigned ohw0 = __builtin_bit_cast(unsigned, oh01), ohw1 = __builtin_bit_cast(unsigned, oh23);
    const char *qkv_b = reinterpret_cast<const char *>(g.qkv);
    const int G = g.B * g.nwin;
    const unsigned lab_off = (unsigned)(qi & 3) * (LD * 4);          // label of this lane's token (query or key slot qi) inside its pixel

#pragma unroll 1
    for (int q = wv;; q += NW) {
        const int p_loc = q / 5, qt = q - 5 * p_loc;
        const int gidx = bih + p_loc * g.nblk_head;
        if (gidx >= G) break;
        const int bimg = gidx / g.nwin, w_in = gidx - bimg * g.nwin;
        const int wi = w_in / g.nwx, wj = w_in - wi * g.nwx;
        const bool on_border = g.shift && (wi == g.Hp / 6 - 1 || wj == g.nwx - 1);      // wave-uniform
        // ---- the window's pixel table (rolled grid: the roll by -shift and its inverse are folded into the rows, NMP.py:803-826) ----
        if (lane < 40) {
            const int p = lane < 36 ? lane : 35;
            const int a = (p * 43) >> 8, b = p - 6 * a;
            int Y = wi * 6 + a + g.shift, X = wj * 6 + b + g.shift;
            Y = Y >= g.Hp ? Y - g.Hp : Y;
            X = X >= g.Wp ? X - g.Wp : X;
            const int Yr = wi * 6 + a, Xr = wj * 6 + b;              // Swin shift regions on the rolled grid (NMP.py:211-239)
            const int fy = Yr < g.Hp - 6 ? 0 : (Yr < g.Hp - g.shift ? 1 : 2);
            const int fx = Xr < g.Wp - 6 ? 0 : (Xr < g.Wp - g.shift ? 1 : 2);
            pix[lane] = make_uint2((unsigned)(((bimg * g.Hp + Y) * g.Wp + X) * 4) * (unsigned)(LD * 4), (unsigned)(fy * 3 + fx));
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        W6_STAMP(0);
        const int tok = 32 * qt + qi;
        const bool tok_ok = tok < 144;
        const int tokc = tok_ok ? tok : 143;                        // padded queries (last tile) compute on the last token and store nothing
        const int q_pix = tokc >> 2;
        const uint2 qp = pix[q_pix];
        const int q_reg = (int)qp.y;
        const unsigned q_boff = qp.x + (unsigned)(tokc & 3) * (LD * 4) + head * 128;          // byte offset of this token's q, this head

        // ---- operands.  QV / KV: words 0..15 = hi halves of the lane's 32 channels (in its rotated order), words 16..31 = lo halves -------
        unsigned QV[32], KV[32];
        float vf[16];
        // k row of key slot qi of tile kt (kv16 rows: 64 bytes of hi halves, 64 of lo halves per head), rotated by 32 bytes for half 1
        auto issue_kv = [&](unsigned kpix) {
            const char *kb = qkv_b + (kpix + lab_off + (C + head * 32) * 4);
            const uint4 *pa = reinterpret_cast<const uint4 *>(kb + 32 * half), *pbk = reinterpret_cast<const uint4 *>(kb - 32 * half);
#pragma unroll
            for (int part = 0; part < 2; ++part) {                  // hi halves, lo halves
                const uint4 v0 = pa[4 * part + 0], v1 = pa[4 * part + 1], v2 = pbk[4 * part + 2], v3 = pbk[4 * part + 3];
                unsigned *d = KV + 16 * part;
                d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
                d[8] = v2.x; d[9] = v2.y; d[10] = v2.z; d[11] = v2.w; d[12] = v3.x; d[13] = v3.y; d[14] = v3.z; d[15] = v3.w;
            }
        };
        // V^T fragment: lane (channel qi, half) holds v[key mfma_row(s, half)][qi], s = 0..15: the four labels of pixel 2 rq + half are four
        // consecutive rows
        auto issue_v = [&](const unsigned (&vpix)[4]) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const float *p = reinterpret_cast<const float *>(qkv_b + (vpix[rq] + (2 * C + head * 32 + qi) * 4 + LD * 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[4 * rq + e] = p[(e - 1) * LD];
            }
        };
        unsigned kpix_n = pix[pb].x, vpix_n[4];
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) vpix_n[rq] = pix[2 * rq + half].x;
        {
            float raw[32];
            const float *src = reinterpret_cast<const float *>(qkv_b + q_boff);
            const float *sa = src + 16 * half, *sb = src - 16 * half;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = ldg4((c < 4 ? sa : sb) + 4 * c);
                raw[4 * c + 0] = v.x; raw[4 * c + 1] = v.y; raw[4 * c + 2] = v.z; raw[4 * c + 3] = v.w;
            }
            issue_kv(kpix_n);
            issue_v(vpix_n);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                h16x2 h, l;
                split2u(f32x2{raw[2 * c] * sc2, raw[2 * c + 1] * sc2}, h, l);
                QV[c] = __builtin_bit_cast(unsigned, h);
                QV[16 + c] = __builtin_bit_cast(unsigned, l);
            }
        }
        W6_STAMP(1);

        f32x16 acc_o;                                   // O^T[d = mfma_row(r, half)][q] from the MFMAs
        f32x2 oe[8];                                    // value-embedding term in the SAME layout (register pairs)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) oe[r] = f32x2{0.f, 0.f};
        float m_ref = -INFINITY, l_run = 0.f;
        const int q11 = w6_x11(q_pix);
        const int ev_r0 = q11 + 60 + half * R;          // rel(pq, pixel 0) + this half's chunk offset (its chunks are 2 gq + half), in row units
        const bool sib = g.sibling != 0;
        // pass K: the partner pixel (a query pixel of this wave's tile) is the same for every key tile
        int tq_pix = 8 * qt + 4 * half + i4;
        tq_pix = tq_pix < 36 ? tq_pix : 35;
        const int tq11 = w6_x11(tq_pix);

#pragma unroll 1
        for (int kt = 0; kt < 5; ++kt) {
            // addresses of the NEXT tile's rows (the last tile re-requests itself: no branch around loads)
            {
                const int kn = kt < 4 ? kt + 1 : 4;
                kpix_n = pix[8 * kn + pb].x;
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) vpix_n[rq] = pix[8 * kn + 2 * rq + half].x;
            }
            // ---- relative-position logit terms of this (query tile, key tile) pair on the 4x4x4 MFMA ---------------------------------
            // pass Q / pass K: lo x hi, hi x hi, hi x lo -- six independent chains of 8 (a dependent 4x4x4 MFMA waits ~60 cycles for its
            // accumulator: two chains of 16 set the pace of the whole pass, profiles/r06c)
            f32x4 qx = {0.f, 0.f, 0.f, 0.f}, qh = qx, kx = qx, kh = qx, qy = qx, ky = qx;
            {
                int tk_pix = 8 * kt + 4 * half + i4;                             // pass Q: own = the query's pixel, partner = a key pixel of the tile
                tk_pix = tk_pix < 36 ? tk_pix : 35;
                int ok_pix = 8 * kt + pb;                                        // pass K: own = the key slot's pixel, partner = tq_pix
                ok_pix = ok_pix < 36 ? ok_pix : 35;
                const int rq_row = q11 - w6_x11(tk_pix) + 60;                    // rel(query pixel, key pixel) = 11 (aq - ak + 5) + (bq - bk + 5)
                const int rk_row = tq11 - w6_x11(ok_pix) + 60;
                const uint4 *qa = tab_ek + rq_row + rotA, *qb = tab_ek + rq_row + rotB;
                const uint4 *ka = tab_eq + rk_row + rotA, *kb = tab_eq + rk_row + rotB;
#pragma unroll
                for (int c = 0; c < ((ABL & 2) ? 1 : 8); ++c) {
                    const uint4 tq = (c < 4 ? qa : qb)[c * R], tk = (c < 4 ? ka : kb)[c * R];
                    const w6_h4 tqh = __builtin_bit_cast(w6_h4, uint2{tq.x, tq.y}), tql = __builtin_bit_cast(w6_h4, uint2{tq.z, tq.w});
                    const w6_h4 tkh = __builtin_bit_cast(w6_h4, uint2{tk.x, tk.y}), tkl = __builtin_bit_cast(w6_h4, uint2{tk.z, tk.w});
                    const w6_h4 qvh = __builtin_bit_cast(w6_h4, uint2{QV[2 * c], QV[2 * c + 1]});
                    const w6_h4 qvl = __builtin_bit_cast(w6_h4, uint2{QV[16 + 2 * c], QV[17 + 2 * c]});
                    const w6_h4 kvh = __builtin_bit_cast(w6_h4, uint2{KV[2 * c], KV[2 * c + 1]});
                    const w6_h4 kvl = __builtin_bit_cast(w6_h4, uint2{KV[16 + 2 * c], KV[17 + 2 * c]});
                    qx = __builtin_amdgcn_mfma_f32_4x4x4f16(tql, qvh, qx, 0, 0, 0);
                    kx = __builtin_amdgcn_mfma_f32_4x4x4f16(tkl, kvh, kx, 0, 0, 0);
                    qh = __builtin_amdgcn_mfma_f32_4x4x4f16(tqh, qvh, qh, 0, 0, 0);
                    kh = __builtin_amdgcn_mfma_f32_4x4x4f16(tkh, kvh, kh, 0, 0, 0);
                    qy = __builtin_amdgcn_mfma_f32_4x4x4f16(tqh, qvl, qy, 0, 0, 0);
                    ky = __builtin_amdgcn_mfma_f32_4x4x4f16(tkh, kvl, ky, 0, 0, 0);
                }
            }
            W6_STAMP(2 + 6 * kt);
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            {
                // xq[e] = QR[this query][key pixel 4 half + e of the tile], xk[e] = KR[this key slot][query pixel 4 half + e of the wave's tile]
                h16x2 qh01, ql01, qh23, ql23, kh01, kl01, kh23, kl23;
                const float s = 1.0f / W6_P0_SCALE;
                split2u(f32x2{(qh[0] + (qx[0] + qy[0])) * s, (qh[1] + (qx[1] + qy[1])) * s}, qh01, ql01);
                split2u(f32x2{(qh[2] + (qx[2] + qy[2])) * s, (qh[3] + (qx[3] + qy[3])) * s}, qh23, ql23);
                split2u(f32x2{(kh[0] + (kx[0] + ky[0])) * s, (kh[1] + (kx[1] + ky[1])) * s}, kh01, kl01);
                split2u(f32x2{(kh[2] + (kx[2] + ky[2])) * s, (kh[3] + (kx[3] + ky[3])) * s}, kh23, kl23);
                auto w = [](h16x2 v) { return __builtin_bit_cast(unsigned, v); };
                // k slots 8 half + 0..3: A = [key's pixel == 4 half + e], B = QR;  slots 8 half + 4..7: A = KR, B = [query's pixel == 4 half + e]
                const h16x8 a_lo = __builtin_bit_cast(h16x8, make_uint4(ohw0, ohw1, w(kl01), w(kl23)));
                const h16x8 b_lo = __builtin_bit_cast(h16x8, make_uint4(w(ql01), w(ql23), ohw0, ohw1));
                const h16x8 a_hi = __builtin_bit_cast(h16x8, make_uint4(ohw0, ohw1, w(kh01), w(kh23)));
                const h16x8 b_hi = __builtin_bit_cast(h16x8, make_uint4(w(qh01), w(qh23), ohw0, ohw1));
                st = mfma16h(a_lo, b_lo, st);
                if (!(ABL & 8)) st = mfma16h(a_hi, b_hi, st);
            }
#pragma unroll
            for (int c = 0; c < ((ABL & 8) ? 0 : 2); ++c) {          // the first 16 channels of a lane's (rotated) arrays are its fragment of K Q^T
                const h16x8 kfh = __builtin_bit_cast(h16x8, make_uint4(KV[4 * c], KV[4 * c + 1], KV[4 * c + 2], KV[4 * c + 3]));
                const h16x8 kfl = __builtin_bit_cast(h16x8, make_uint4(KV[16 + 4 * c], KV[17 + 4 * c], KV[18 + 4 * c], KV[19 + 4 * c]));
                const h16x8 qfh = __builtin_bit_cast(h16x8, make_uint4(QV[4 * c], QV[4 * c + 1], QV[4 * c + 2], QV[4 * c + 3]));
                const h16x8 qfl = __builtin_bit_cast(h16x8, make_uint4(QV[16 + 4 * c], QV[17 + 4 * c], QV[18 + 4 * c], QV[19 + 4 * c]));
                split_mma1(kfh, kfl, qfh, qfl, st);
            }
            if (!(ABL & 32)) issue_kv(kpix_n);                       // the k rows of the next tile: in flight during softmax, P V and the ev term
            W6_STAMP(3 + 6 * kt);
            // ---- masks (wave-uniform branches: they fire on one tile of five / on border windows only) -----------------------------
            if (kt == 4) {                                                       // keys beyond the window: slots 16..31 of the last tile
#pragma unroll
                for (int r = 8; r < 16; ++r) st[r] = -INFINITY;
            }
            if (sib && kt == qt) {                                               // sibling labels of the query's own pixel
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ks = mfma_row(r, half);
                    if ((ks >> 2) == pb && ks != qi) st[r] = -INFINITY;
                }
            }
            if (on_border) {                                                     // Swin regions
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const bool other = (int)pix[8 * kt + 2 * rq + half].y != q_reg;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (other) st[4 * rq + e] = -INFINITY;
                }
            }
            // ---- streaming softmax (log2 domain), lazy rescale ----------------------------------------------------------------------
            float m_tile = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
            m_tile = half_max(m_tile);
            if (__builtin_amdgcn_ballot_w64(m_tile > m_ref + W6_LAZY) != 0) {    // (first tile: m_ref = -inf)
                const float m_new = fmaxf(m_ref, m_tile);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_ref - m_use);       // exp2(-inf) = 0 on the first tile (the accumulators are 0)
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
#pragma unroll
                for (int r = 0; r < 8; ++r) oe[r] *= alpha;
                m_ref = m_new;
            }
            const float m_use = (m_ref == -INFINITY) ? 0.f : m_ref;
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = (ABL & 16) ? (st[r] - m_use) * 0.01f : __builtin_amdgcn_exp2f(st[r] - m_use);
                psum += st[r];
            }
            l_run += psum;
            W6_STAMP(4 + 6 * kt);
            {   // O^T += V^T P^T
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = st[r];
                h16x8 ph[2], pl[2], vh[2], vl[2];
                split8u(pv, ph[0], pl[0]);
                split8u(pv + 8, ph[1], pl[1]);
                kv16_chunks(vf, vh, vl);
                if (!(ABL & 4)) {
                    split_mma1(vh[0], vl[0], ph[0], pl[0], acc_o);
                    split_mma1(vh[1], vl[1], ph[1], pl[1], acc_o);
                } else {
                    acc_o[0] += (float)ph[0][0] + (float)pl[1][3] + (float)vh[0][1] + (float)vl[1][2];
                }
            }
            if (!(ABL & 32)) issue_v(vpix_n);                        // v of the next tile: in flight until its P V
            W6_STAMP(5 + 6 * kt);
            // ---- value-embedding term: sum over key PIXELS of (sum_n p) ev[rel(pq, pk)]; the two half-lanes of a query swap the
            //      probability masses of their pixels, then each accumulates BOTH pixels for its own 16 channels
            float pm[8];                                                           // pm[2 rq + h2] = mass of key pixel 2 rq + h2 of the tile
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float p0 = (st[4 * rq] + st[4 * rq + 1]) + (st[4 * rq + 2] + st[4 * rq + 3]);
                float p1 = p0;
                half_swap(p0, p1);
                pm[2 * rq] = p0;
                pm[2 * rq + 1] = p1;
            }
#pragma unroll
            for (int pk8 = 0; pk8 < ((ABL & 1) ? 0 : 8); ++pk8) {
                if (kt == 4 && pk8 >= 4) continue;                                // pixels beyond the window
                const int pk = 8 * kt + pk8;                                      // (wave-uniform)
                const float *e = tab_ev + (ev_r0 - w6_x11(pk)) * 4;
                const f32x2 ps = {pm[pk8], pm[pk8]};
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {                                  // channels 8 gq + 4 half .. + 3 = O^T registers 4 gq .. 4 gq + 3
                    const float4 t = *reinterpret_cast<const float4 *>(e + 2 * gq * (R * 4));
                    oe[2 * gq + 0] = pk_fma(ps, f32x2{t.x, t.y}, oe[2 * gq + 0]);
                    oe[2 * gq + 1] = pk_fma(ps, f32x2{t.z, t.w}, oe[2 * gq + 1]);
                }
                // two pixels' rows in flight at a time: unfenced, the scheduler hoists all 32 reads and spills; a hand-pipelined form
                // (rows two steps ahead, three rotating buffers) spilled 16 registers around the softmax and ran 2.3x slower (profiles/r06c)
                if (pk8 & 1) __builtin_amdgcn_sched_barrier(0);
            }
            W6_STAMP(7 + 6 * kt);
        }
        W6_STAMP(32);
        const float inv_l = 1.0f / half_sum(l_run);
        if (tok_ok) {
            float *op = g.out + (size_t)(qp.x / (unsigned)(LD * 4) + (tokc & 3)) * C + head * 32;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                stg4(op + mfma_row(4 * rb, half), make_float4((acc_o[4 * rb] + oe[2 * rb].x) * inv_l, (acc_o[4 * rb + 1] + oe[2 * rb].y) * inv_l,
                                                              (acc_o[4 * rb + 2] + oe[2 * rb + 1].x) * inv_l, (acc_o[4 * rb + 3] + oe[2 * rb + 1].y) * inv_l));
        }
        W6_STAMP(33);
        __builtin_amdgcn_wave_barrier();                  // the pixel table is rewritten by the next item
    }
}

static int g_w6_cus[NMRF_MAX_DEV] = {};

template <int NW, int OCC, int ABL = 0, bool TIMED = false>
static int launch_window6(const float *qkv, const void *packed, int B, int Hp, int Wp, int heads, int shift, int sibling_mask, float *out,
                          hipStream_t st, unsigned long long *stamps = nullptr) {
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    if (!g_w6_cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) return NMRF_ELAUNCH;
        g_w6_cus[dev] = n;
    }
    Win6Args a;
    a.qkv = qkv; a.tabp = (const char *)packed; a.out = out;
    a.B = B; a.Hp = Hp; a.Wp = Wp; a.heads = heads; a.shift = shift; a.sibling = sibling_mask ? 1 : 0;
    a.nwx = Wp / 6; a.nwin = (Hp / 6) * (Wp / 6);
    const int G = B * a.nwin;
    int nblk = g_w6_cus[dev] / heads;                      // one block per CU, a head's blocks strided over the XCDs
    if (nblk < 1) nblk = 1;
    if (nblk > G) nblk = G;
    a.nblk_head = nblk;
    a.stamps = stamps;
    hipLaunchKernelGGL((window_attn6_kernel<NW, OCC, ABL, TIMED>), dim3(nblk * heads), dim3(64 * NW), 0, st, a);
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES
static int g_w6_variant = 0;        // tools: 0 product (8 waves); 1: 12 waves; 3: 4 waves; 10-12 timed; 100 + mask: timing ablation `mask`
static unsigned long long *g_w6_stamps = nullptr;
extern "C" int nmrf_debug_window6_variant(int v) { g_w6_variant = v; return NMRF_OK; }
extern "C" int nmrf_debug_window6_stamps(unsigned long long *p) { g_w6_stamps = p; return NMRF_OK; }   // device buffer [blocks * 12][64]
#endif

// qkv [B, Hp, Wp, 4, 384] with kv16 rows, packed = nmrf_window_table_pack_f32 of the layer's table -> out [B, Hp, Wp, 4, 128]
extern "C" int nmrf_window_attn6_f32(const float *qkv, const void *packed, int B, int Hp, int Wp, int C, int heads, int shift,
                                     int sibling_mask, float *out, void *stream) {
    if (!qkv || !packed || !out) return NMRF_ENULL;
    if (B < 1 || Hp < 6 || Wp < 6 || Hp % 6 || Wp % 6 || shift < 0 || shift >= 6 || C != 128 || heads != 4) return NMRF_EINVAL;
    if ((int64_t)B * Hp * Wp * 4 * 3 * C >= ((int64_t)1 << 30)) return NMRF_EINVAL;        // 32-bit BYTE offsets into qkv
    hipStream_t st = (hipStream_t)stream;
#ifdef NMRF_DEBUG_PROBES
#define W6_ARGS qkv, packed, B, Hp, Wp, heads, shift, sibling_mask, out, st
    switch (g_w6_variant) {
        case 1: return launch_window6<12, 3>(W6_ARGS);
        case 3: return launch_window6<4, 1>(W6_ARGS);
        case 10: return launch_window6<8, 2, 0, true>(W6_ARGS, g_w6_stamps);
        case 11: return launch_window6<12, 3, 0, true>(W6_ARGS, g_w6_stamps);
        case 12: return launch_window6<4, 1, 0, true>(W6_ARGS, g_w6_stamps);
        case 101: return launch_window6<8, 2, 1>(W6_ARGS);
        case 102: return launch_window6<8, 2, 2>(W6_ARGS);
        case 104: return launch_window6<8, 2, 4>(W6_ARGS);
        case 108: return launch_window6<8, 2, 8>(W6_ARGS);
        case 116: return launch_window6<8, 2, 16>(W6_ARGS);
        case 132: return launch_window6<8, 2, 32>(W6_ARGS);
        case 163: return launch_window6<8, 2, 63>(W6_ARGS);
        case 131: return launch_window6<8, 2, 31>(W6_ARGS);
        default: break;
    }
#endif
    return launch_window6<8, 2>(qkv, packed, B, Hp, Wp, heads, shift, sibling_mask, out, st);
}
