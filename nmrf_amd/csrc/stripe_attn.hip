// A7: cross-stripe (CSWin, split_size 1) attention with LePE on fp32 matrix cores.
//
// One wave = one 32-query tile of one (image b, stripe, head) and one of KSPLIT key ranges.  K and V
// operand fragments are read straight from the token-major qkv tensor (each token's 32-wide head slice
// is one 128-byte line, shared through L1/L2 by the waves of the block, which walk the same stripe).
// Flash-style streaming softmax, so the [T,T] matrix never exists (H9).
//
// Matrix-core formulation (round 1: v_mfma_f32_32x32x2_f32, 16 k-steps per 32-deep contraction; round 2: the same maps on
// v_mfma_f32_32x32x16_f16 with split fp16 operand pairs, 2 chunks x 3 MFMAs of 32 cycles instead of 16 of 64 -- split_mfma.h):
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
#include "split_mfma.h"
#include <type_traits>
#include <stdlib.h>

#define SA_TILE 32
#define SA_LOG2E 1.4426950408889634f

struct StripeGeom {
    int H, W, N, C;        // grid, labels per pixel, embed dim (128)
    int L;                 // pixels per stripe
    int Ts;                // tokens per stripe = L*N
    int64_t pix_stride;    // token-pixel stride between consecutive stripe positions (W or 1)
    int gx, gy, gz;        // logical grid: query-tile groups x (stripe, head) x image
    int *range_flag;       // sticky fp16-range flag of the split operands (split_mfma.h), may be NULL
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
//
// __launch_bounds__(256, 3): with a register budget <= 256 the compiler keeps the MFMA accumulators in VGPRs; without the
// occupancy hint it parks them in AGPRs and every `acc *= alpha` / softmax pass pays v_accvgpr_read/write round trips
// (112 of them per key tile in the first version of this kernel).
// KV16: the k and v thirds of a qkv row hold split fp16 operand pairs (the producing block kernel wrote them so: kv16 of
// nmrf_nmp_block16_f32, include/nmrf_hip.h) -- the K fragment is four 16-byte loads that ARE the MFMA operands, the V fragment
// 16 words and 16 v_perm_b32; without it each key tile pays 2 x 48 VALU instructions to split them (of ~250 in a tile).
// The body of one block: work item `item` of the logical grid g.gx x g.gy x g.gz (the kernels below map blockIdx to items).
// LDS pool of the SHARED form (four-label stripes with pre-split k | v, one key range), declared by the KERNEL and handed to the body:
// the merged kernel runs either body in a block, never both, and two private pools cost it its fourth block per CU.
#define SA_SROW 36
#define SA_LP_ROWS 144                                         // LePE window: 4 query tiles + 4 tokens before + 12 after (4 used, 8 pad a chunk)
#define SA_POOL_FLOATS (SA_LP_ROWS * SA_SROW + 96)             // >= the K and V tile rings of the key loop (4 x 32 x SA_SROW)
template <int NSHIFT, int KSPLIT, bool KV16>
constexpr int stripe_pool_floats() { return (KV16 && KSPLIT == 1 && NSHIFT == 2) ? SA_POOL_FLOATS : 4; }

template <int AXIS, int NSHIFT, int KSPLIT, bool CENSUS = false, bool KV16 = false>
__device__ __forceinline__ void stripe_attn_body(const float *__restrict__ qkv, const float *__restrict__ lepe, const StripeGeom &g,
                                                 float scale, float *__restrict__ out, unsigned long long *__restrict__ census,
                                                 const int item, float *s_pool) {
    constexpr int QPB = 4 / KSPLIT;                           // query tiles per block
    float guard = 0.f;                                        // fp16 range guard of the q / k / v splits (split_mfma.h)
    const int total = g.gx * g.gy * g.gz;
    const int bx = item % g.gx, by = (item / g.gx) % g.gy, bz = item / (g.gx * g.gy);
    struct Scope {
        unsigned long long *p;
        __device__ Scope(unsigned long long *c, int item) : p(c) {
            if (CENSUS && threadIdx.x == 0) {
                p = c + 3 * (size_t)item;
                p[0] = __smid(); p[1] = wall_clock64();
            }
        }
        __device__ ~Scope() { if (CENSUS && threadIdx.x == 0) p[2] = wall_clock64(); }
    } scope(census, item);
    // CENSUS also records s_memtime stamps [64 blocks][4 waves][16] behind the census triples (tools/kernel_bench.py)
#define SA_STAMP(k) do { if (CENSUS && lane == 0 && by == 0 && bz == 0 && bx < 64) \
        census[3 * (size_t)total + ((size_t)bx * 4 + wv) * 16 + (k)] = \
            __builtin_amdgcn_s_memtime(); } while (0)
    __shared__ float s_o[KSPLIT > 1 ? 4 : 1][16][64];          // partial O^T of the non-leading key ranges
    __shared__ float s_ml[KSPLIT > 1 ? 4 : 1][2][64];          // their (m, l)
    constexpr int RB_PER = 4 / KSPLIT;                        // LePE channel blocks computed by each key-range wave
    __shared__ float s_rpe[KSPLIT > 1 ? 4 : 1][4 * RB_PER][64];
    // SHARED (four-label stripes with pre-split k | v, one key range): the four query tiles of a block walk the same key tiles, so
    // the block stages each K and V tile ONCE in LDS -- every thread one 16-byte load of a row's 128 contiguous bytes -- instead of
    // every wave fetching its own fragments, where a K load instruction has each lane on a different row: 64 cache lines per
    // instruction, four such per tile and wave, and the L1 address path (not the matrix pipe, not the VALU: cutting a third of a
    // tile's VALU work bought 8 %) set the pace.  Rows are padded to 36 floats: the b128 fragment reads are conflict-free.
    constexpr bool SHARED = KV16 && KSPLIT == 1 && NSHIFT == 2;
    constexpr int SROW = SA_SROW;
    // one pool (s_pool, the kernel's): the K and V tile rings of the key loop ([2][32 * SROW] each), then -- every wave past the loop's
    // last barrier -- the LePE window of the block (LP_ROWS token rows of v, SROW floats apart) and the three tap vectors of the head
    constexpr int LP_ROWS = SA_LP_ROWS;
    static_assert(SA_POOL_FLOATS >= 4 * 32 * SA_SROW, "the pool holds the two tile rings");
    float (*s_k)[SHARED ? 32 * SROW : 1] = reinterpret_cast<float (*)[SHARED ? 32 * SROW : 1]>(s_pool);
    float (*s_v)[SHARED ? 32 * SROW : 1] = reinterpret_cast<float (*)[SHARED ? 32 * SROW : 1]>(s_pool + (SHARED ? 2 * 32 * SROW : 0));
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: key loop + addresses on the SALU
    const int qi = lane & 31, hi = lane >> 5;
    const int qslot = wv / KSPLIT, ks = wv % KSPLIT;           // which query tile of the block, which key range
    const int qt = bx * QPB + qslot;
    const int q0 = qt * SA_TILE;
    const bool wave_on = q0 < g.Ts;                            // wave-uniform
    const int stripe = by >> 1, head = by & 1;
    const int b = bz;
    const int64_t base_pix = (AXIS == 0) ? ((int64_t)b * g.H * g.W + stripe)          // column x = stripe
                                         : ((int64_t)b * g.H * g.W + (int64_t)stripe * g.W);  // row y = stripe
    constexpr size_t ld = 384;                                // 3*C; the entry point only admits C == 128
    const int coff = AXIS * (g.C / 2) + head * 32;            // channel offset of this (half, head) inside q / k / v
    const int nlab = NSHIFT >= 0 ? (1 << NSHIFT) : g.N;

    SA_STAMP(0);
    const int qs = q0 + qi;
    const bool q_ok = qs < g.Ts;
    const int qsc = q_ok ? qs : g.Ts - 1;
    const int64_t qrow = stripe_row<NSHIFT>(g, base_pix, wave_on ? qsc : 0);
    const int q_pix = div_n<NSHIFT>(g, qsc);

    f32x16 acc_o;                               // (two accumulator chains per contraction, summed afterwards: measured 37.4 vs 37.1 us -- not kept)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    if constexpr (SHARED) {
        // ---- Q fragment (B operand), as below ---------------------------------------------------------------
        h16x8 qh[2], ql[2];
        {
            float qf[16];
            const float sc2 = scale * SA_LOG2E;
            const float *p = qkv + qrow * ld + coff + 16 * hi;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + 4 * c);
                qf[4 * c + 0] = v.x * sc2; qf[4 * c + 1] = v.y * sc2; qf[4 * c + 2] = v.z * sc2; qf[4 * c + 3] = v.w * sc2;
            }
            split8u_g(qf, qh[0], ql[0], guard);
            split8u_g(qf + 8, qh[1], ql[1], guard);
        }
        const int n_kt = (g.Ts + SA_TILE - 1) / SA_TILE;
        // staging role of this thread: row (key) tid / 8 of the tile, 16-byte piece tid % 8 of its 128-byte head block
        const int srow = threadIdx.x >> 3, spc = threadIdx.x & 7;
        const int64_t ps = (AXIS == 1) ? (int64_t)1 : g.pix_stride;
        auto stage_ptr = [&](int kt) -> const float * {
            int kk = kt * SA_TILE + srow;
            kk = kk < g.Ts ? kk : g.Ts - 1;                       // rows beyond the stripe shadow its last token (masked below)
            return qkv + stripe_row<NSHIFT>(g, base_pix, kk) * ld + coff + 4 * spc;
        };
        float4 rk, rv;
        auto fetch_tile = [&](int kt) { const float *p = stage_ptr(kt); rk = ldg4(p + g.C); rv = ldg4(p + 2 * g.C); };
        auto store_tile = [&](int buf) {
            stg4(&s_k[buf][srow * SROW + 4 * spc], rk);
            stg4(&s_v[buf][srow * SROW + 4 * spc], rv);
        };
        fetch_tile(0);
        store_tile(0);
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < n_kt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < n_kt) fetch_tile(kt + 1);                // in flight under this tile's arithmetic
            if (wave_on) {
                const int k0 = kt * SA_TILE;
                h16x8 kh[2], kl[2], vh[2], vl[2];
                {
                    const float *kr = &s_k[buf][qi * SROW + 8 * hi];            // hi halves of channels 16 hi .. +15, lo 16 floats on
                    kh[0] = *reinterpret_cast<const h16x8 *>(kr);      kh[1] = *reinterpret_cast<const h16x8 *>(kr + 4);
                    kl[0] = *reinterpret_cast<const h16x8 *>(kr + 16); kl[1] = *reinterpret_cast<const h16x8 *>(kr + 20);
                    float vw[16];
#pragma unroll
                    for (int s2 = 0; s2 < 16; ++s2) vw[s2] = s_v[buf][mfma_row(s2, hi) * SROW + qi];
                    kv16_chunks(vw, vh, vl);
                }
                f32x16 st;
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 0.f;
                split_mma1(kh[0], kl[0], qh[0], ql[0], st);
                split_mma1(kh[1], kl[1], qh[1], ql[1], st);
                if (kt == n_kt - 1) {                             // keys beyond the stripe (last tile only)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (k0 + mfma_row(r, hi) >= g.Ts) st[r] = -INFINITY;
                }
                if (kt == qt && nlab > 1) {                       // sibling labels of the query's own pixel (diagonal tile)
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
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = st[r];
                h16x8 ph[2], pl[2];
                split8u(pv, ph[0], pl[0]);
                split8u(pv + 8, ph[1], pl[1]);
                split_mma1(vh[0], vl[0], ph[0], pl[0], acc_o);
                split_mma1(vh[1], vl[1], ph[1], pl[1], acc_o);
            }
            if (kt + 1 < n_kt) store_tile(buf ^ 1);               // (last read during tile kt - 1: every wave is past that barrier)
            __syncthreads();
        }
        l_run = half_sum(l_run);
    }
    if (!SHARED && wave_on) {
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
        // both contractions run on the fp16 matrix pipe with split operands (split_mfma.h, single-accumulator form): chunk c,
        // slot jj of half hi <-> MFMA k-slot (step s = 8c + jj, half hi) of the fp32 formulation above, so every map stays
        h16x8 qh[2], ql[2];
        split8u_g(qf, qh[0], ql[0], guard);
        split8u_g(qf + 8, qh[1], ql[1], guard);
        const int n_kt = (g.Ts + SA_TILE - 1) / SA_TILE;
        const int per = (n_kt + KSPLIT - 1) / KSPLIT;
        const int kt_begin = ks * per, kt_end = (kt_begin + per < n_kt) ? kt_begin + per : n_kt;
        // K fragment of lane (key, hi): 16 channels of the head -- 16 floats, or (KV16) 16 hi halves at float offset 8 hi of the head's
        // 32-float block and the 16 lo halves 16 floats further; kd[0..7] then hold the hi chunks, kd[8..15] the lo chunks
        const float *kbase = qkv + g.C + coff + (KV16 ? 8 : 16) * hi;
        const float *vbase = qkv + 2 * g.C + coff + qi;
        constexpr int KO2 = KV16 ? 16 : 8, KO3 = KV16 ? 20 : 12;     // float offsets of the 3rd / 4th 16-byte piece
        // N == 4: a key tile is 8 whole pixels, so the rows of a FULL tile sit at fixed offsets from one per-lane
        // pointer that advances by `tile_step` per tile (the generic row arithmetic cost 68 v_mul_lo_u32 + 34
        // v_mad_u64_u32 per tile -- more VALU issue than the softmax).  Only a ragged last tile takes the clamped path.
        const int64_t ps = (AXIS == 1) ? (int64_t)1 : g.pix_stride;
        const int64_t tile_step = 32 * ps * (int64_t)ld;        // floats per key tile: 8 pixels x 4 labels
        const int64_t vj_step = 8 * ps * (int64_t)ld;           // two pixels: key k0 + 4*hi + 8*j + i
        const float *kfast = kbase + ((base_pix + (int64_t)(qi >> 2) * ps) * 4 + (qi & 3)) * (int64_t)ld;
        const float *vfast = vbase + ((base_pix + (int64_t)hi * ps) * 4) * (int64_t)ld;
        auto tile_full = [&](int kt) { return NSHIFT == 2 && (kt + 1) * SA_TILE <= g.Ts; };   // wave-uniform
        // K fragment (A operand): lane (ki=qi, hi) holds K[k0+ki][16*hi + s]
        auto load_k_fast = [&](int kt, float *kd) {
            const float *p = kfast + (int64_t)kt * tile_step;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + (c < 2 ? 4 * c : (c == 2 ? KO2 : KO3)));
                kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
            }
        };
        auto load_v_fast = [&](int kt, float *vd) {
            const float *p = vfast + (int64_t)kt * tile_step;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) vd[4 * j + i] = p[j * vj_step + i * (int64_t)ld];
        };
        auto load_k = [&](int kt, float *kd) {
            const float *p;
            if (tile_full(kt)) {
                p = kfast + (int64_t)kt * tile_step;
            } else {
                int kk = kt * SA_TILE + qi;
                kk = kk < g.Ts ? kk : g.Ts - 1;
                p = kbase + stripe_row<NSHIFT>(g, base_pix, kk) * ld;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float4 v = ldg4(p + (c < 2 ? 4 * c : (c == 2 ? KO2 : KO3)));
                kd[4 * c + 0] = v.x; kd[4 * c + 1] = v.y; kd[4 * c + 2] = v.z; kd[4 * c + 3] = v.w;
            }
        };
        // V fragment (A operand of the 2nd product): lane (d=qi, hi), step s: V[k0+mfma_row(s,hi)][d]
        auto load_v = [&](int kt, float *vd) {
            if (tile_full(kt)) {
                load_v_fast(kt, vd);
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    int kv = kt * SA_TILE + mfma_row(s, hi);
                    kv = kv < g.Ts ? kv : g.Ts - 1;
                    vd[s] = vbase[stripe_row<NSHIFT>(g, base_pix, kv) * ld];
                }
            }
        };
        // Two K/V register sets, each refilled in place for the tile two steps ahead: a fragment is requested ~1.6 tile
        // times before its MFMAs (one tile time is ~1.3 us for a lone wave, less than a MALL/HBM round trip, so the
        // single in-place buffer of the first version left every tile waiting on memory).
        float kf0[16], vf0[16], kf1[16], vf1[16];
        if (kt_begin < kt_end) { load_k(kt_begin, kf0); load_v(kt_begin, vf0); }
        if (kt_begin + 1 < kt_end) { load_k(kt_begin + 1, kf1); load_v(kt_begin + 1, vf1); }
        SA_STAMP(1);
        // One key tile on register set (kf, vf).  STEADY: the tile two steps ahead is a full tile inside the range, so
        // the refill is straight-line code (the compiler then emits exact vmcnt waits; with the full/ragged branch in
        // the loop it fell back to vmcnt(0) and every tile stalled on the fragment it had just requested) and the tail
        // mask is compiled out.
        auto tile = [&](int kt, float *kf, float *vf, auto steady_tag) {
            constexpr bool STEADY = decltype(steady_tag)::value;
            const int k0 = kt * SA_TILE;
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            {
                h16x8 kh[2], kl[2];
                if constexpr (KV16) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f32x4 hv = {kf[4 * c], kf[4 * c + 1], kf[4 * c + 2], kf[4 * c + 3]};
                        const f32x4 lv = {kf[8 + 4 * c], kf[8 + 4 * c + 1], kf[8 + 4 * c + 2], kf[8 + 4 * c + 3]};
                        kh[c] = __builtin_bit_cast(h16x8, hv);
                        kl[c] = __builtin_bit_cast(h16x8, lv);
                    }
                } else {
                    split8u_g(kf, kh[0], kl[0], guard);
                    split8u_g(kf + 8, kh[1], kl[1], guard);
                }
                split_mma1(kh[0], kl[0], qh[0], ql[0], st);
                split_mma1(kh[1], kl[1], qh[1], ql[1], st);
            }
            if (STEADY) load_k_fast(kt + 2, kf);                // K fragment is dead: refill now
            else if (kt + 2 < kt_end) load_k(kt + 2, kf);
            if (!STEADY && kt == n_kt - 1) {                   // keys beyond the stripe (last tile only)
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
            {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pv[r] = st[r];
                h16x8 ph[2], pl[2], vh[2], vl[2];
                split8u(pv, ph[0], pl[0]);
                split8u(pv + 8, ph[1], pl[1]);
                if constexpr (KV16) {
                    kv16_chunks(vf, vh, vl);
                } else {
                    split8u_g(vf, vh[0], vl[0], guard);
                    split8u_g(vf + 8, vh[1], vl[1], guard);
                }
                split_mma1(vh[0], vl[0], ph[0], pl[0], acc_o);
                split_mma1(vh[1], vl[1], ph[1], pl[1], acc_o);
            }
            if (STEADY) load_v_fast(kt + 2, vf);                // V fragment likewise
            else if (kt + 2 < kt_end) load_v(kt + 2, vf);
            if (CENSUS) { asm volatile("" :: "v"(acc_o[0])); if (kt - kt_begin < 6) SA_STAMP(2 + kt - kt_begin); }
        };
        int kt = kt_begin;
        if (NSHIFT == 2) {
            const int n_full = g.Ts / SA_TILE;
            const int steady_end = (kt_end < n_full ? kt_end : n_full) - 3;    // kt + 3 is still a full tile of the range
#pragma unroll 1
            for (; kt < steady_end; kt += 2) {
                tile(kt, kf0, vf0, std::true_type{});
                tile(kt + 1, kf1, vf1, std::true_type{});
            }
        }
#pragma unroll 1
        for (; kt < kt_end; kt += 2) {
            tile(kt, kf0, vf0, std::false_type{});
            if (kt + 1 < kt_end) tile(kt + 1, kf1, vf1, std::false_type{});
        }
        SA_STAMP(7);
        l_run = half_sum(l_run);                        // both halves now hold the range's full (m, l)
    }

    // ---- LePE for width-1 stripes (NMP.py:433-449 / SURVEY H3), independent of the attention itself:
    //   rpe_j(p)[c] = w_c v_j(p)[c] + sum_k ( w_- v_k(p-1)[c] + w_+ v_k(p+1)[c] )
    // taps = centre column (AXIS 0: kernel[:,1]) or centre row (AXIS 1: kernel[1,:]) of the 3x3 kernel.
    // Lane (q=qi, hi) owns output channels d = mfma_row(r, hi): four blocks rb of 4 consecutive channels.  The KSPLIT
    // waves of a query tile share the work (wave ks takes blocks ks*RB_PER ..) right after their key loop -- computing it
    // before the loop keeps 20-36 more VGPRs live and costs the 4th wave per SIMD; the leading wave collects the other
    // blocks from LDS after the merge barrier (it used to do all 36 float4 + 48 weight loads alone while 3 waves idled).
    float rpe[4 * RB_PER];
    if constexpr (SHARED) {
        // The same three sums as products on the matrix pipe (round 6; the VALU form below took 8-11k of a block's ~55k cycles: 36
        // row-per-lane 16-byte loads and 48 tap loads per lane behind the key loop, profiles/r06m_stripe_census.txt):
        //   R_tap^T[d][q] = sum_key V[key][d] * pat_tap[q][key],   pat_c = [key == q], pat_-/+ = [pixel(key) == pixel(q) -/+ 1]
        // over the 40 tokens around the wave's query tile -- V^T is the A operand exactly as in P V, the 0 / 1 patterns are exact in
        // fp16 (two MFMAs per 16-key chunk: V's hi and lo halves), and rpe = w_c R_c + w_- R_- + w_+ R_+ per channel afterwards.
        // The block stages the v rows of its 128 queries + 4 before + 12 after ONCE, coalesced (every thread 16-byte pieces of whole
        // rows), into the pool the key loop has left; rows outside the stripe shadow its end tokens and meet a zero pattern.
        const int tap_m = (AXIS == 0) ? 1 : 3, tap_c = 4, tap_p = (AXIS == 0) ? 7 : 5;
        float *s_lp = s_pool, *s_w = s_pool + LP_ROWS * SROW;
        const int Q0 = bx * QPB * SA_TILE;
        for (int i = threadIdx.x; i < LP_ROWS * 8; i += 256) {
            const int row = i >> 3, pc = i & 7;
            int sk = Q0 - 4 + row;
            sk = sk < 0 ? 0 : (sk < g.Ts ? sk : g.Ts - 1);
            stg4(&s_lp[row * SROW + 4 * pc], ldg4(qkv + stripe_row<NSHIFT>(g, base_pix, sk) * ld + 2 * g.C + coff + 4 * pc));
        }
        if (threadIdx.x < 96) {
            const int tp = threadIdx.x >> 5, d = threadIdx.x & 31;
            s_w[threadIdx.x] = lepe[(size_t)(head * 32 + d) * 9 + (tp == 0 ? tap_c : (tp == 1 ? tap_m : tap_p))];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) rpe[r] = 0.f;
        if (wave_on) {
            // k slot (chunk c, jj, half hi) <-> token Q0 - 4 + 32 qslot + 16 c + 8 hi + jj of the stripe, on both operands
            h16x8 lvh[3], lvl[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float *vr = s_lp + (32 * qslot + 16 * c + 8 * hi) * SROW + qi;
                unsigned h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned e0 = __builtin_bit_cast(unsigned, vr[(2 * k) * SROW]), e1 = __builtin_bit_cast(unsigned, vr[(2 * k + 1) * SROW]);
                    h[k] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);
                    l[k] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);
                }
                lvh[c] = __builtin_bit_cast(h16x8, make_uint4(h[0], h[1], h[2], h[3]));
                lvl[c] = __builtin_bit_cast(h16x8, make_uint4(l[0], l[1], l[2], l[3]));
            }
            const int qb = (qi >> 2) + 8;                       // pixel of the query, in pixels from the one before the tile's first, + 7
#pragma unroll
            for (int tp = 0; tp < 3; ++tp) {
                f32x16 R;
#pragma unroll
                for (int r = 0; r < 16; ++r) R[r] = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    unsigned pw[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        unsigned wd = 0;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int kk = 16 * c + 8 * hi + 2 * k + e;            // window slot of this key; its token: q0 - 4 + kk
                            const int sk = q0 - 4 + kk;
                            const int ka = (kk + 28) >> 2;                         // its pixel on the scale of qb
                            const bool on = sk >= 0 && sk < g.Ts && (tp == 0 ? kk - 4 == qi : (tp == 1 ? ka == qb - 1 : ka == qb + 1));
                            wd |= on ? (e ? 0x3c000000u : 0x00003c00u) : 0u;       // fp16 1.0
                        }
                        pw[k] = wd;
                    }
                    const h16x8 pat = __builtin_bit_cast(h16x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
                    R = mfma16h(lvl[c], pat, R);
                    R = mfma16h(lvh[c], pat, R);
                }
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const f32x4 wv = *reinterpret_cast<const f32x4 *>(s_w + 32 * tp + 8 * rb + 4 * hi);    // channels mfma_row(4 rb, hi) ..
#pragma unroll
                    for (int e = 0; e < 4; ++e) rpe[4 * rb + e] = fmaf(wv[e], R[4 * rb + e], rpe[4 * rb + e]);
                }
            }
        }
    } else {
        const int tap_m = (AXIS == 0) ? 1 : 3, tap_c = 4, tap_p = (AXIS == 0) ? 7 : 5;
        const bool has_prev = q_pix > 0, has_next = q_pix < g.L - 1;
        const int64_t prev_row = stripe_row<NSHIFT>(g, base_pix, (has_prev ? q_pix - 1 : q_pix) * nlab);
        const int64_t next_row = stripe_row<NSHIFT>(g, base_pix, (has_next ? q_pix + 1 : q_pix) * nlab);
        const float *vb = qkv + 2 * g.C + coff;
#pragma unroll
        for (int j = 0; j < RB_PER; ++j) {
            const int rb = ks * RB_PER + j;
            const int d0 = 8 * rb + 4 * hi;                        // == mfma_row(4*rb, hi)
            auto vrow4 = [&](int64_t row) {                       // four consecutive channels of a token's v
                float4 t = ldg4(vb + row * ld + d0);
                if constexpr (KV16) t = make_float4(kv16_value(t.x), kv16_value(t.y), kv16_value(t.z), kv16_value(t.w));
                return t;
            };
            float4 vq = vrow4(qrow);
            float4 sp = make_float4(0.f, 0.f, 0.f, 0.f), sn = sp;
            if (wave_on)
                for (int n = 0; n < nlab; ++n) {
                    float4 t = vrow4(prev_row + n);
                    sp.x += t.x; sp.y += t.y; sp.z += t.z; sp.w += t.w;
                    float4 u = vrow4(next_row + n);
                    sn.x += u.x; sn.y += u.y; sn.z += u.z; sn.w += u.w;
                }
            const float fp = has_prev ? 1.f : 0.f, fn = has_next ? 1.f : 0.f;
            const float vqa[4] = {vq.x, vq.y, vq.z, vq.w}, spa[4] = {sp.x, sp.y, sp.z, sp.w}, sna[4] = {sn.x, sn.y, sn.z, sn.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float *wk = lepe + (size_t)(head * 32 + d0 + e) * 9;
                rpe[4 * j + e] = wk[tap_c] * vqa[e] + (wk[tap_m] * fp) * spa[e] + (wk[tap_p] * fn) * sna[e];
            }
        }
    }


    if (CENSUS) asm volatile("" :: "v"(rpe[0]));
    SA_STAMP(8);
    // ---- merge the KSPLIT key ranges of each query tile through LDS ----------------------------------------
    if (KSPLIT > 1) {
        if (ks != 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_o[wv][r][lane] = acc_o[r];
            s_ml[wv][0][lane] = m_run;
            s_ml[wv][1][lane] = l_run;
#pragma unroll
            for (int r = 0; r < 4 * RB_PER; ++r) s_rpe[wv][r][lane] = rpe[r];
        }
        SA_STAMP(9);
        __syncthreads();
        SA_STAMP(10);
        split_guard_commit(guard, g.range_flag);
        guard = 0.f;
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
    split_guard_commit(guard, g.range_flag);
    if (!wave_on || !q_ok) return;
    const float inv_l = 1.0f / l_run;

    // ---- epilogue: normalise, add LePE, store.  Lane (q=qi, hi) owns channels d = mfma_row(r,hi) --------
    float *op = out + qrow * g.C + coff;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int d0 = mfma_row(4 * rb, hi);                   // 4 consecutive channels d0..d0+3
        float res[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r;
            if (KSPLIT == 1) r = rpe[4 * rb + e];
            else if (rb / RB_PER == 0) r = rpe[4 * (rb % RB_PER) + e];
            else r = s_rpe[wv + rb / RB_PER][4 * (rb % RB_PER) + e][lane];
            res[e] = acc_o[4 * rb + e] * inv_l + r;
        }
        stg4(op + d0, make_float4(res[0], res[1], res[2], res[3]));
    }
    SA_STAMP(11);
}

template <int AXIS, int NSHIFT, int KSPLIT, bool CENSUS = false, bool KV16 = false>
__global__ __launch_bounds__(256, 2) void stripe_attn_kernel(const float *__restrict__ qkv, const float *__restrict__ lepe,
                                                         StripeGeom g, float scale, float *__restrict__ out,
                                                         unsigned long long *__restrict__ census = nullptr) {
    // XCD-aware block order.  Workgroups go round-robin over the 8 XCDs (linear id % 8), each with a private 4 MB L2.
    // All query tiles of one (stripe, head) read the same K/V rows, so the logical work items are handed out in
    // contiguous runs per XCD: with the natural order every XCD streamed the whole K/V set (15 MB at KITTI) through
    // its L2 and the kernel re-fetched it from MALL/HBM (137 MB of traffic for 60 MB of tensors).
    const int total = g.gx * g.gy * g.gz;
    const int chunk = gridDim.x >> 3;                          // the launch pads the grid to a multiple of 8
    const int item = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
    if (item >= total) return;
    __shared__ __attribute__((aligned(16))) float s_pool[stripe_pool_floats<NSHIFT, KSPLIT, KV16>()];
    stripe_attn_body<AXIS, NSHIFT, KSPLIT, CENSUS, KV16>(qkv, lepe, g, scale, out, census, item, s_pool);
}

// Both axes of a propagation layer in ONE launch (four labels per pixel, pre-split k | v rows): the two kernels are independent --
// vertical stripes attend on channel half 0, horizontal ones on half 1 (NMP.py:429-600) -- and each alone leaves the chip partly
// idle at batch 1 (470 and 624 blocks of 4 waves for 768 resident slots).  The horizontal items (20 key tiles each) are handed out
// first, the vertical ones (6 key tiles) fill in behind them; per axis the same per-XCD runs as above.
template <int NSHIFT, bool KV16>
__global__ __launch_bounds__(256, 2) void stripe_attn_both_kernel(const float *__restrict__ qkv, const float *__restrict__ lepe_v,
                                                              const float *__restrict__ lepe_h, StripeGeom gv, StripeGeom gh,
                                                              float scale, float *__restrict__ out, int chunk_h, int chunk_v) {
    __shared__ __attribute__((aligned(16))) float s_pool[stripe_pool_floats<NSHIFT, 1, KV16>()];
    const int xcd = (int)(blockIdx.x & 7), k = (int)(blockIdx.x >> 3);
    if (k < chunk_h) {
        const int item = xcd * chunk_h + k;
        if (item < gh.gx * gh.gy * gh.gz) stripe_attn_body<1, NSHIFT, 1, false, KV16>(qkv, lepe_h, gh, scale, out, nullptr, item, s_pool);
    } else {
        const int item = xcd * chunk_v + (k - chunk_h);
        if (item < gv.gx * gv.gy * gv.gz) stripe_attn_body<0, NSHIFT, 1, false, KV16>(qkv, lepe_v, gv, scale, out, nullptr, item, s_pool);
    }
}

template <int AXIS, int NSHIFT, bool KV16 = false>
static void launch_stripe(const float *qkv, const float *lepe, const StripeGeom &g_in, int stripes, int B, float scale,
                          float *out, hipStream_t st) {
    // key split, decided per image and NOT per batch so that results do not depend on the batch size.  Measured on MI355X
    // (tools/kernel_bench.py --which stripe, NMRF_STRIPE_KSPLIT sweep, batch 1 / 4 / 8):
    //   horizontal KITTI stripes (20 key tiles): 1 -> 64 / 244 / 425 us,  2 -> 72 / 246 / 465,  4 -> 76 / 257 / 475
    //   vertical   KITTI stripes ( 6 key tiles): 1 -> 37 / 109 / 211 us,  2 -> 33 / 100 / 209,  4 -> 37 / 112 / 220
    // a wave needs ~3 key tiles to amortise its Q load, LePE and merge; long stripes are best left whole.
    const int n_qt = (g_in.Ts + SA_TILE - 1) / SA_TILE;
    int ksplit = 1;
    if (n_qt >= 4 && n_qt < 16) ksplit = 2;
    // pre-split rows: the block stages each key tile once for its four query tiles (SHARED), which needs them on one key range --
    // vertical KITTI stripes 20.4 us against 22.2 for two key ranges with per-wave fragments (and 27.3 for fp32 rows unsplit)
    if (KV16 && NSHIFT == 2) ksplit = 1;
#ifdef NMRF_DEBUG_PROBES
    static const char *force = getenv("NMRF_STRIPE_KSPLIT");   // tuning override (tools/kernel_bench.py), debug library only
    if (force && (force[0] == '1' || force[0] == '2' || force[0] == '4')) ksplit = force[0] - '0';
#endif
    const int qpb = 4 / ksplit;
    StripeGeom g = g_in;
    g.gx = (n_qt + qpb - 1) / qpb; g.gy = stripes * 2; g.gz = B;
    dim3 grid((unsigned)(((int64_t)g.gx * g.gy * g.gz + 7) / 8 * 8));
    if (ksplit == 1) hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 1, false, KV16>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
    else if (ksplit == 2) hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 2, false, KV16>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
    else hipLaunchKernelGGL((stripe_attn_kernel<AXIS, NSHIFT, 4, false, KV16>), grid, dim3(256), 0, st, qkv, lepe, g, scale, out,
                                          (unsigned long long *)nullptr);
}

#ifdef NMRF_DEBUG_PROBES   // tools-only library libnmrf_hip_debug.so (python -m nmrf_amd.build --debug)
// Debug: census run of the horizontal N=4 KSPLIT=1 kernel (KITTI configuration); census[blocks*3] + stamps on device.
extern "C" int nmrf_debug_stripe_census(const float *qkv, const float *lepe_h, int B, int H, int W, float *out,
                                        unsigned long long *census, int *grid_out, void *stream) {
    StripeGeom g{H, W, 4, 128, W, W * 4, (int64_t)1, 0, 0, 0, nullptr};
    const int n_qt = (g.Ts + SA_TILE - 1) / SA_TILE;
    g.gx = (n_qt + 3) / 4; g.gy = H * 2; g.gz = B;            // KSPLIT = 1 -> four query tiles per block
    dim3 grid((unsigned)(((int64_t)g.gx * g.gy * g.gz + 7) / 8 * 8));
    grid_out[0] = g.gx; grid_out[1] = g.gy; grid_out[2] = g.gz;
    hipLaunchKernelGGL((stripe_attn_kernel<1, 2, 1, true>), grid, dim3(256), 0, (hipStream_t)stream, qkv, lepe_h, g,
                       1.0f / sqrtf(32.0f), out, census);
    return nmrf_launch_status();
}
#endif  // NMRF_DEBUG_PROBES

extern "C" int nmrf_stripe_attn_f32(const float *qkv, const float *lepe_v, const float *lepe_h, int B, int H, int W,
                                    int N, int C, int axes, int kv16, float *out, int *range_flag, void *stream) {
    if (!qkv || !lepe_v || !lepe_h || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N < 1 || C != 128 || axes < 1 || axes > 3) return NMRF_EINVAL;
    if (kv16 && N != 4) return NMRF_EINVAL;                     // (the pre-split form is built for the shipped four labels per pixel)
    if (W * 2 > 65535 || H * 2 > 65535 || B > 65535) return NMRF_EINVAL;
    const float scale = 1.0f / sqrtf(32.0f);
    hipStream_t st = (hipStream_t)stream;
    if (axes == 3 && N == 4 && kv16) {                      // (a caller that wants the two kernels one after the other passes 1, then 2)
        StripeGeom gv{H, W, N, C, H, H * N, (int64_t)W, 0, 0, 0, range_flag};
        StripeGeom gh{H, W, N, C, W, W * N, (int64_t)1, 0, 0, 0, range_flag};
        gv.gx = ((gv.Ts + SA_TILE - 1) / SA_TILE + 3) / 4; gv.gy = W * 2; gv.gz = B;
        gh.gx = ((gh.Ts + SA_TILE - 1) / SA_TILE + 3) / 4; gh.gy = H * 2; gh.gz = B;
        const int64_t tv = (int64_t)gv.gx * gv.gy * gv.gz, th = (int64_t)gh.gx * gh.gy * gh.gz;
        if (tv + th <= 0x7ffffff0) {                        // (larger grids: the two launches below)
            const int chunk_v = (int)((tv + 7) / 8), chunk_h = (int)((th + 7) / 8);
            hipLaunchKernelGGL((stripe_attn_both_kernel<2, true>), dim3((unsigned)(8 * (chunk_v + chunk_h))), dim3(256), 0, st, qkv,
                               lepe_v, lepe_h, gv, gh, scale, out, chunk_h, chunk_v);
            return nmrf_launch_status();
        }
    }
    if (axes & 1) {   // vertical stripes: one per column, H*N tokens each, channel half 0
        StripeGeom g{H, W, N, C, H, H * N, (int64_t)W, 0, 0, 0, range_flag};
        if (N == 4 && kv16) launch_stripe<0, 2, true>(qkv, lepe_v, g, W, B, scale, out, st);
        else if (N == 4) launch_stripe<0, 2>(qkv, lepe_v, g, W, B, scale, out, st);
        else launch_stripe<0, -1>(qkv, lepe_v, g, W, B, scale, out, st);
    }
    if (axes & 2) {   // horizontal stripes: one per row, W*N tokens each, channel half 1
        StripeGeom g{H, W, N, C, W, W * N, (int64_t)1, 0, 0, 0, range_flag};
        if (N == 4 && kv16) launch_stripe<1, 2, true>(qkv, lepe_h, g, H, B, scale, out, st);
        else if (N == 4) launch_stripe<1, 2>(qkv, lepe_h, g, H, B, scale, out, st);
        else launch_stripe<1, -1>(qkv, lepe_h, g, H, B, scale, out, st);
    }
    return nmrf_launch_status();
}
