// nmp_block (see nmp_block.hip for the operator and the formulation) with 16 tokens per wave on v_mfma_f32_16x16x32_f16:
// block = 8 waves x 16 tokens = the same 128 tokens and the same weight stream, but TWO waves per SIMD, so that one wave's
// serial phases (row loads, LayerNorm, GELU, staging, row stores, barrier waits) run under the other wave's MFMAs.  The
// 32-token kernel has one wave per SIMD at KITTI batch 1 (234 tiles for 256 CUs) and its census shows MFMAs (31 us) and everything
// else (41 us) back to back with no overlap (profiles/r02e_block_experiments.txt).
//
// v_mfma_f32_16x16x32_f16 lane layout (g = lane >> 4, j = lane & 15; pinned by nmrf_selftest_mfma_f16split mode 4):
//   A operand: lane holds A[i = j][k = 8g + 0..7]       B operand: lane holds B[k = 8g + 0..7][col = j]
//   C/D      : reg r (0..3) of the lane is D[row = 4g + r][col = j]
// Activations are the B operand with the token on j; a 16-row output strip s of a layer puts channel 16s + 4g + r into reg r,
// and two consecutive strips ARE one 32-deep k chunk of the next layer with the slot order
//   slot jj of k-group g  <->  k = 32c + (jj & 3) + 16 (jj >> 2) + 4g           (split_kslot16)
// which the weight packing (nmrf_pack_split_weight16_f32) uses for every A operand.  A "pair" is still 2 KB (hi + lo fragment,
// 64 lanes x 16 B) = one (16-row strip, 32-deep chunk); 8 pairs per 16 KB stage; same ring protocol as nmp_block.hip.  Stream
// order: within proj / fc1 / q two adjacent strips are interleaved chunk by chunk (s0c0 s1c0 s0c1 s1c1 ...), so that the two
// pairs of a `consume2` share their B operand and alternate between two accumulators.
#include "common.h"
#include "split_mfma.h"
#include <type_traits>
#include <utility>

typedef unsigned int b16_u32x4 __attribute__((ext_vector_type(4)));

#define B16_TOK 128                // tokens per block (8 waves x 16)
#define B16_THR 512
#define B16_STAGE_U4 1024
#define B16_RING 3
#define B16_OLD 132
#define B16_PF 4
#define B16_PAR_OFF (B16_RING * B16_STAGE_U4 * 16 + 8 * 16 * B16_OLD * 4)
#define B16P_BP 0
#define B16P_G2 128
#define B16P_B2N 256
#define B16P_B1 384
#define B16P_B2 896
#define B16P_GQ 1024
#define B16P_BQN 1152
#define B16P_BQ 1280
#define B16P_FLOATS (1280 + 512)
// FUSE: the parameters of the second block of the launch (bp | lnq gamma | lnq beta | bq) behind the first set
#define B16P2_BP 1792
#define B16P2_GQ 1920
#define B16P2_BQN 2048
#define B16P2_BQ 2176
#define B16P2_FLOATS (2176 + 512)

__host__ __device__ __forceinline__ int split_kslot16(int jj, int g) { return (jj & 3) + 16 * (jj >> 2) + 4 * g; }

__device__ __forceinline__ f32x4 mfma16x16h(h16x8 a, h16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split_mma16(h16x8 ah, h16x8 al, h16x8 bh, h16x8 bl, f32x4 &acc) {
    acc = mfma16x16h(al, bh, acc);
    acc = mfma16x16h(ah, bl, acc);
    acc = mfma16x16h(ah, bh, acc);
}

template <class F, int... I>
__device__ __forceinline__ void b16_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void b16_static_for(F &&f) {
    b16_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// value of quad lane N (lanes 4p .. 4p+3 = the sibling tokens of one pixel) in every lane of the quad
template <int N>
__device__ __forceinline__ float b16_quad(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), N * 0x55, 0xf, 0xf, false));
}

struct NmpBlock16Args {
    const float *x, *msg;
    const float *attn_qkv;       // MLP == false: [T, 384] q | k | v of the self-edge attention (4 sibling tokens per pixel, 4 heads of 32);
                                 // the message is then computed here instead of being read from `msg`
    const b16_u32x4 *stream;
    int total_stages;
    const float *bp, *ln2_g, *ln2_b, *b1, *b2, *lnq_g, *lnq_b, *extra;
    int extra_ld, extra_div;
    const float *bq;
    float *x_out, *q_out, *ln_out;
    const int *ln_out_map;
    int64_t T;
    int n_tiles;
    float eps2, epsq;
    int NQ;
    float inv_p, inv_1, inv_2, inv_q;
    unsigned long long *stamps;  // debug build: s_memtime stamps of the first 64 blocks (DBG & 32)
    int *range_flag;             // sticky fp16-range flag of the split operands (split_mfma.h), may be NULL
    int kv16;                    // q_out is q | k | v for an attention kernel (NQ == 384): write k and v as the split fp16 operand
                                 // pairs those kernels would otherwise make of them on every key tile (format: include/nmrf_hip.h)
    // FUSE: the self-edge block that consumes this block's q | k | v (which then stay in registers) runs in the same launch:
    const float *bp2, *lnq2_g, *lnq2_b, *extra2, *bq2;
    int extra2_ld, extra2_div;
    float *x_out2, *q_out2, *ln_out2;
    const int *ln_out2_map;
    float epsq2, inv_p2, inv_q2;
    int NQ2, kv16_2;
    unsigned long long *clk;     // measurement hook (nmrf_nmp_block16_clock_records), NULL in every product call: per block
                                 // {shader clock at entry, at exit, 100 MHz counter at entry, at exit}
};

// MLP: run fc1-GELU-fc2.  KQC: 32-deep k chunks of the q stage's operand [LNq(x2) | extra]: 0 none, 4 = LayerNorm only,
// 5 = + 32 side columns (Fourier31 + 0), 6 = + 64 context columns.
// FUSE (KQC == 5, NQ == 384, four sibling labels; with or without proj / MLP in front): the launch continues with the self-edge block that follows in the layer
// sequence (NMP.py:90-108, 337-364) -- this block's q | k | v never leave the registers, the 4 x 4 sibling attention runs on them,
// then proj + residual -> LayerNorm | enc -> the window attention's q | k | v with the second parameter set and the stages that
// follow in the same weight stream.  One launch, one x round trip and one [T,384] round trip less per layer.
template <bool MLP, int KQC, int DBG = 0, bool FUSE = false>          // DBG: timing experiments of the debug build (wrong results)
__global__ __launch_bounds__(B16_THR, 2) void nmp_block16_kernel(NmpBlock16Args a) {
    static_assert(!FUSE || KQC == 5, "FUSE: a block whose q stage (LayerNorm | 32 side columns) feeds a self-edge block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    b16_u32x4 *ring = reinterpret_cast<b16_u32x4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    float *Ot = reinterpret_cast<float *>(smem + B16_RING * B16_STAGE_U4 * 16) + wv * 16 * B16_OLD;     // wave-private [16][132]
    float *Par = reinterpret_cast<float *>(smem + B16_PAR_OFF);
#define B16_STAMP(k) do { if constexpr ((DBG & 32) != 0) { if (lane == 0 && blockIdx.x < 64) a.stamps[(blockIdx.x * 8 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
    B16_STAMP(0);
#ifndef NMRF_NO_CLK_HOOK                     // (A/B build without the hook: tools/build_ab_flag.sh noclk -DNMRF_NO_CLK_HOOK nmp_block16)
    if (a.clk && tid == 0) {                // (uniform branch on a kernel argument: nothing when the hook is off)
        a.clk[(size_t)blockIdx.x * 4] = __builtin_amdgcn_s_memtime();
        a.clk[(size_t)blockIdx.x * 4 + 2] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    // DBG & 32: every block also leaves (realtime at entry, realtime at exit, XCC << 32 | HW_ID) behind the 64 stamped blocks' records
    // (s_memrealtime: the 100 MHz counter that all CUs share -- s_memtime is per CU)
    if constexpr ((DBG & 32) != 0) {
        if (tid == 0) a.stamps[64 * 8 * 16 + (size_t)blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime();
    }
    float guard = 0.f;                                                  // fp16 range guard of the activation splits (split_mfma.h)
    float qmax = 0.f;                       // ... and of q_out: the attention kernels split it without a guard of their own (window_attn.hip)
    // The x / msg rows of the block's FIRST tile are requested before the prologue (parameter table, first weight stages, barrier):
    // their latency then runs under it instead of behind it (census: 5.5k cycles of a wave's life were this phase, and at batch 1
    // the first tile is the only one).
    constexpr bool HOIST = !(DBG & 256);
    float4 xpre[8], mpre[8];
    if constexpr (HOIST) {
        const int64_t tq0 = (int64_t)blockIdx.x * B16_TOK + wv * 16 + j;
        const int64_t tc0 = tq0 < a.T ? tq0 : a.T - 1;
#pragma unroll
        for (int st = 0; st < 8; ++st) xpre[st] = ldg4(a.x + tc0 * 128 + 16 * st + 4 * g);
        if (a.msg) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                mpre[2 * c] = ldg4(a.msg + tc0 * 128 + 32 * c + 4 * g);
                mpre[2 * c + 1] = ldg4(a.msg + tc0 * 128 + 32 * c + 16 + 4 * g);
            }
        }
    }
    {
        // The eight parameter vectors are requested TOGETHER: written as `Par[i] = src ? src[i] : 0` one vector after the other, every
        // vector was its own uniform branch + load + full wait -- eight serial L2 round trips at the head of every launch
        // (tools/isa_scan.py).  A NULL vector reads the head of the weight stream instead (always >= 2 KB) and is zeroed afterwards.
        // (one `tid < ns[k]` pass per vector: the longest fixed vector, b1, has 512 entries; only bq has a strided tail loop)
        static_assert(B16_THR >= 512, "the parameter prologue loads b1 (512 floats) and the NULL-vector dummy reads in ONE pass of B16_THR threads");
        const float *dummy = reinterpret_cast<const float *>(a.stream);
        const float *srcs[8] = {a.bp, a.ln2_g, a.ln2_b, a.b1, a.b2, a.lnq_g, a.lnq_b, a.bq};
        const int offs[8] = {B16P_BP, B16P_G2, B16P_B2N, B16P_B1, B16P_B2, B16P_GQ, B16P_BQN, B16P_BQ};
        const int ns[8] = {128, 128, 128, 512, 128, 128, 128, a.bq ? a.NQ : 512};
        float pv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = (srcs[k] ? srcs[k] : dummy)[tid < ns[k] ? tid : 0];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (tid < ns[k]) Par[offs[k] + tid] = srcs[k] ? pv[k] : 0.f;
        for (int i = tid + B16_THR; i < ns[7]; i += B16_THR) Par[B16P_BQ + i] = a.bq ? a.bq[i] : 0.f;    // (NQ > 512: not a shipped shape)
    }
    if constexpr (FUSE) {
        const float *dummy = reinterpret_cast<const float *>(a.stream);
        const float *srcs[4] = {a.bp2, a.lnq2_g, a.lnq2_b, a.bq2};
        const int offs[4] = {B16P2_BP, B16P2_GQ, B16P2_BQN, B16P2_BQ};
        const int ns[4] = {128, 128, 128, a.bq2 ? a.NQ2 : 512};
        float pv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pv[k] = (srcs[k] ? srcs[k] : dummy)[tid < ns[k] ? tid : 0];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid < ns[k]) Par[offs[k] + tid] = srcs[k] ? pv[k] : 0.f;
    }
    auto par4 = [&](int off) { return *reinterpret_cast<const f32x4 *>(Par + off); };

    // ---- weight stream (protocol of nmp_block.hip; 512 threads move 2 x 16 B each per stage) -----------------------------------
    b16_u32x4 R[2];
    int src_stage = 0, wr_slot = 0, rd_slot = 0;
    auto fetch = [&]() {
        const b16_u32x4 *p = a.stream + (size_t)src_stage * B16_STAGE_U4 + tid;
        R[0] = p[0]; R[1] = p[B16_THR];
        src_stage = (src_stage + 1 == a.total_stages) ? 0 : src_stage + 1;
    };
    auto commit = [&]() {
        b16_u32x4 *d = ring + wr_slot * B16_STAGE_U4 + tid;
        d[0] = R[0]; d[B16_THR] = R[1];
        wr_slot = (wr_slot == B16_RING - 1) ? 0 : wr_slot + 1;
    };
    const b16_u32x4 *cur = ring, *nxt = ring + B16_STAGE_U4;
    h16x8 fqh[B16_PF], fql[B16_PF];
    auto read_pair = [&](const b16_u32x4 *base, int p, h16x8 &h, h16x8 &l) {
        h = *reinterpret_cast<const h16x8 *>(base + p * 128 + lane);
        l = *reinterpret_cast<const h16x8 *>(base + p * 128 + 64 + lane);
    };
    bool have_barrier = true;
    unsigned long long tk_bar = 0, tk_use = 0, tk_commit = 0, tk_t = 0;     // DBG & 32: ticks inside barriers / consume / commit+fetch
#define B16_NOW() (((DBG & 32) != 0) ? __builtin_amdgcn_s_memtime() : 0ull)
    auto stage_top = [&]() {
        const unsigned long long t0 = B16_NOW();
        if (!have_barrier && !(DBG & 1)) __syncthreads();
        have_barrier = false;
        tk_t = B16_NOW();
        tk_bar += tk_t - t0;
    };
    auto stage_end = [&]() {
        unsigned long long t1 = 0;
        if constexpr ((DBG & 32) != 0) {
            asm volatile("s_nop 0" ::: "memory");
            t1 = B16_NOW();
            tk_use += t1 - tk_t;
        }
        if constexpr (!(DBG & 2)) {
            commit();
            fetch();
        }
        if constexpr ((DBG & 32) != 0) tk_commit += B16_NOW() - t1;
        rd_slot = (rd_slot == B16_RING - 1) ? 0 : rd_slot + 1;
        cur = nxt;
        nxt = ring + ((rd_slot == B16_RING - 1) ? 0 : rd_slot + 1) * B16_STAGE_U4;
    };
    auto consume = [&](auto pc, const h16x8 &bh, const h16x8 &bl, f32x4 &acc) {
        constexpr int P = decltype(pc)::value;
        const h16x8 ah = fqh[P % B16_PF], al = fql[P % B16_PF];
        if constexpr (!(DBG & 4)) {
            if constexpr (P + B16_PF < 8) read_pair(cur, P + B16_PF, fqh[P % B16_PF], fql[P % B16_PF]);
            else read_pair(nxt, P + B16_PF - 8, fqh[P % B16_PF], fql[P % B16_PF]);
        }
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);       // pin the read-ahead (see nmp_block.hip)
        if constexpr (!(DBG & 8)) split_mma16(ah, al, bh, bl, acc);
        else acc[P & 3] += (float)ah[0] + (float)bl[1];
    };
    // Two consecutive pairs (P0 even, P0 + 1) against ONE B operand into two accumulators, the six MFMAs alternating between
    // them: a dependent MFMA then issues 2 slots after its predecessor instead of back to back (three MFMAs chained on one
    // accumulator stall on each other).  The stream interleaves the two strips that share a chunk for this (block_stream16).
    auto consume2 = [&](auto pc, const h16x8 &bh, const h16x8 &bl, f32x4 &acc0, f32x4 &acc1) {
        constexpr int P0 = decltype(pc)::value, P1 = P0 + 1;
        static_assert(P0 % 2 == 0 && B16_PF % 2 == 0, "pairs of one consume2 share a stage and use different read-ahead slots");
        const h16x8 ah0 = fqh[P0 % B16_PF], al0 = fql[P0 % B16_PF], ah1 = fqh[P1 % B16_PF], al1 = fql[P1 % B16_PF];
        if constexpr (!(DBG & 4)) {
            if constexpr (P0 + B16_PF < 8) {
                read_pair(cur, P0 + B16_PF, fqh[P0 % B16_PF], fql[P0 % B16_PF]);
                read_pair(cur, P1 + B16_PF, fqh[P1 % B16_PF], fql[P1 % B16_PF]);
            } else {
                read_pair(nxt, P0 + B16_PF - 8, fqh[P0 % B16_PF], fql[P0 % B16_PF]);
                read_pair(nxt, P1 + B16_PF - 8, fqh[P1 % B16_PF], fql[P1 % B16_PF]);
            }
        }
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);       // pin the read-ahead (see nmp_block.hip)
        if constexpr (!(DBG & 8)) {
            acc0 = mfma16x16h(al0, bh, acc0);
            acc1 = mfma16x16h(al1, bh, acc1);
            acc0 = mfma16x16h(ah0, bl, acc0);
            acc1 = mfma16x16h(ah1, bl, acc1);
            acc0 = mfma16x16h(ah0, bh, acc0);
            acc1 = mfma16x16h(ah1, bh, acc1);
        } else {
            acc0[P0 & 3] += (float)ah0[0] + (float)bl[1];
            acc1[P0 & 3] += (float)ah1[0] + (float)bl[1];
        }
    };
    {
        // the first three stages are requested TOGETHER (two extra register pairs, free here): one L2 round trip instead of three
        // serial fetch -> commit pairs (the prologue was ~5k cycles of a wave's life, 27 launches per forward)
        b16_u32x4 Ra[2], Rb[2];
        auto fetch_into = [&](b16_u32x4 (&Rx)[2]) {
            const b16_u32x4 *p = a.stream + (size_t)src_stage * B16_STAGE_U4 + tid;
            Rx[0] = p[0]; Rx[1] = p[B16_THR];
            src_stage = (src_stage + 1 == a.total_stages) ? 0 : src_stage + 1;
        };
        auto commit_from = [&](const b16_u32x4 (&Rx)[2]) {
            b16_u32x4 *d = ring + wr_slot * B16_STAGE_U4 + tid;
            d[0] = Rx[0]; d[B16_THR] = Rx[1];
            wr_slot = (wr_slot == B16_RING - 1) ? 0 : wr_slot + 1;
        };
        fetch_into(Ra); fetch_into(Rb); fetch();
        commit_from(Ra); commit_from(Rb);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < B16_PF; ++p) read_pair(cur, p, fqh[p], fql[p]);
    B16_STAMP(1);

    // the 4 C/D registers of a 16-channel strip <-> columns [col0, col0 + 16) of the wave's tile (lane-private addresses)
    auto stage_strip = [&](const float *v, int col0) {
        *reinterpret_cast<f32x4 *>(Ot + j * B16_OLD + col0 + 4 * g) = f32x4{v[0], v[1], v[2], v[3]};
    };
    auto unstage_strip = [&](float *v, int col0) {
        const f32x4 t = *reinterpret_cast<const f32x4 *>(Ot + j * B16_OLD + col0 + 4 * g);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    };
    // whole rows of the tile -> dst: lanes 0-31 one row, lanes 32-63 the next (512 B each)
    auto flush_rows = [&](float *dst, int ld, int col0, int64_t t0, const int *map = nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 2 * i + (lane >> 5);
            const float4 v = *reinterpret_cast<const float4 *>(Ot + row * B16_OLD + 4 * (lane & 31));
            if (t0 + row < a.T) {
                int64_t orow = t0 + row;
                if (map) orow = map[orow];
                if (orow >= 0) stg4(dst + (size_t)orow * ld + col0 + 4 * (lane & 31), v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // LayerNorm over the 128 channels of a token: a lane holds 32 of them (8 strips x 4), lanes j, j+16, j+32, j+48 the rest
    auto group_sum = [&](float v) {
        v += __shfl_xor(v, 16);
        return half_sum(v);
    };
    auto layer_norm = [&](const float (&v)[32], int g_off, int b_off, float eps, float (&o)[32]) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += v[i];
        const float mean = group_sum(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(group_sum(q) * (1.0f / 128.0f) + eps);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const f32x4 gv = par4(g_off + 16 * st + 4 * g), bv = par4(b_off + 16 * st + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * st + e] = (v[4 * st + e] - mean) * rstd * gv[e] + bv[e];
        }
    };

    // one head of the 4 x 4 sibling attention: this lane's 8 channels of q, k, v of its token -> its 8 channels of the message
    auto sib_weights = [&](const float (&q)[8], const float (&k)[8], float (&w)[4]) {       // softmax_n(q . k_n / sqrt(32))
        const float sc = 0.17677669529663687f;                               // 32^-0.5
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s[0] = fmaf(q[e], b16_quad<0>(k[e]), s[0]);
            s[1] = fmaf(q[e], b16_quad<1>(k[e]), s[1]);
            s[2] = fmaf(q[e], b16_quad<2>(k[e]), s[2]);
            s[3] = fmaf(q[e], b16_quad<3>(k[e]), s[3]);
        }
        float m = -INFINITY;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            s[n] += __shfl_xor(s[n], 16);
            s[n] += __shfl_xor(s[n], 32);
            s[n] *= sc;
            m = fmaxf(m, s[n]);
        }
        float z = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) { s[n] = expf(s[n] - m); z += s[n]; }
        const float rz = 1.0f / z;
#pragma unroll
        for (int n = 0; n < 4; ++n) w[n] = s[n] * rz;
    };
    auto sib_apply = [&](const float (&w)[4], const float (&vv)[8], float (&o)[8]) {         // sum_n w_n v_n
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float acc_o = w[0] * b16_quad<0>(vv[e]);
            acc_o = fmaf(w[1], b16_quad<1>(vv[e]), acc_o);
            acc_o = fmaf(w[2], b16_quad<2>(vv[e]), acc_o);
            acc_o = fmaf(w[3], b16_quad<3>(vv[e]), acc_o);
            o[e] = acc_o;
        }
    };
    auto sib_attn = [&](const float (&q)[8], const float (&k)[8], const float (&vv)[8], float (&o)[8]) {
        float w[4];
        sib_weights(q, k, w);
        sib_apply(w, vv, o);
    };

    // the body of one tile; PRE: its x / msg rows are the ones requested before the prologue (first tile of the block)
    auto tile_body = [&](const int tile, auto pre_c) {
        constexpr bool first = decltype(pre_c)::value;
        const int64_t t0 = (int64_t)tile * B16_TOK + wv * 16;
        const int64_t tq = t0 + j;
        const int64_t tc = tq < a.T ? tq : a.T - 1;
        float x1[32];                                                          // channel 16*(i >> 2) + 4g + (i & 3)
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            float4 v;
            if constexpr (first) v = xpre[st];
            else v = ldg4(a.x + tc * 128 + 16 * st + 4 * g);
            x1[4 * st] = v.x; x1[4 * st + 1] = v.y; x1[4 * st + 2] = v.z; x1[4 * st + 3] = v.w;
        }
        h16x8 bmh[4], bml[4];
        f32x4 acc[8];
        // x1 += proj(message operand) + bias: four stages, strips 2k, 2k+1 = one stage
        auto proj_stage = [&](const h16x8 (&mh)[4], const h16x8 (&ml)[4], int bp_off, float inv_p) {
            b16_static_for<4>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[2 * k][r] = acc[2 * k + 1][r] = 0.f;
                stage_top();
                b16_static_for<4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    consume2(std::integral_constant<int, 2 * c>{}, mh[c], ml[c], acc[2 * k], acc[2 * k + 1]);
                });
                stage_end();
            });
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const f32x4 b4 = par4(bp_off + 16 * st + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) x1[4 * st + e] += fmaf(acc[st][e], inv_p, b4[e]);
            }
        };
        // ---- stage P -----------------------------------------------------------------------------------------------------------
        bool have_msg = a.msg != nullptr;
        if constexpr (!MLP) have_msg = have_msg || a.attn_qkv != nullptr;
        if (have_msg) {
            bool from_attn = false;
            if constexpr (!MLP) from_attn = a.attn_qkv != nullptr;
            if (from_attn) {
                // Self-edge attention of BasicAttention (NMP.py:90-108) on the way in: the 16 tokens of a wave are 4 pixels x 4 sibling
                // labels, the siblings of a token sit in its lane quad, head c of a token is k chunk c of the message operand, of
                // which this lane holds 8 channels (the other 24 in lanes j + 16, j + 32, j + 48).  softmax_j(q_i . k_j / sqrt(32)) v_j
                // with the arithmetic of self_attn_kernel (token.hip); replaces that launch and the [T,128] round trip of its output.
                const float *qp = a.attn_qkv + tc * 384;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float q[8], k[8], vv[8];
                    {
                        const float4 q0 = ldg4(qp + 32 * c + 4 * g), q1 = ldg4(qp + 32 * c + 16 + 4 * g);
                        const float4 k0 = ldg4(qp + 128 + 32 * c + 4 * g), k1 = ldg4(qp + 128 + 32 * c + 16 + 4 * g);
                        const float4 v0 = ldg4(qp + 256 + 32 * c + 4 * g), v1 = ldg4(qp + 256 + 32 * c + 16 + 4 * g);
                        q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
                        k[0] = k0.x; k[1] = k0.y; k[2] = k0.z; k[3] = k0.w; k[4] = k1.x; k[5] = k1.y; k[6] = k1.z; k[7] = k1.w;
                        vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
                    }
                    float o[8];
                    sib_attn(q, k, vv, o);
                    split8u_g(o, bmh[c], bml[c], guard);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float4 v0, v1;
                    if constexpr (first) { v0 = mpre[2 * c]; v1 = mpre[2 * c + 1]; }
                    else { v0 = ldg4(a.msg + tc * 128 + 32 * c + 4 * g); v1 = ldg4(a.msg + tc * 128 + 32 * c + 16 + 4 * g); }
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    split8u_g(v, bmh[c], bml[c], guard);
                }
            }
            B16_STAMP(2);
            proj_stage(bmh, bml, B16P_BP, a.inv_p);
        }
        B16_STAMP(3);
        // ---- stage M -----------------------------------------------------------------------------------------------------------
        if constexpr (MLP) {
            h16x8 bnh[4], bnl[4];
            {
                float ln[32];
                layer_norm(x1, B16P_G2, B16P_B2N, a.eps2, ln);
#pragma unroll
                for (int c = 0; c < 4; ++c) split8u_g(&ln[8 * c], bnh[c], bnl[c], guard);
            }
#pragma unroll
            for (int st = 0; st < 8; ++st) stage_strip(&x1[4 * st], 16 * st);     // x1 waits in LDS while the hidden layer runs
#pragma unroll
            for (int st = 0; st < 8; ++st)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[st][r] = 0.f;
            // hidden group hg = 32 hidden channels = two 16-row strips of fc1 = one k chunk of fc2
            auto fc1 = [&](f32x4 &f0, f32x4 &f1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) f0[r] = f1[r] = 0.f;
                stage_top();
                b16_static_for<4>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    consume2(std::integral_constant<int, 2 * c>{}, bnh[c], bnl[c], f0, f1);
                });
                stage_end();
            };
            auto act_fc2 = [&](int hg, const f32x4 &f0, const f32x4 &f1) {
                float hv[8];
                const f32x4 ba = par4(B16P_B1 + 32 * hg + 4 * g), bb = par4(B16P_B1 + 32 * hg + 16 + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if constexpr (DBG & 16) {                                  // (timing experiment: no GELU)
                        hv[e] = fmaf(f0[e], a.inv_1, ba[e]);
                        hv[4 + e] = fmaf(f1[e], a.inv_1, bb[e]);
                    } else {
                        hv[e] = gelu_fast(fmaf(f0[e], a.inv_1, ba[e]));
                        hv[4 + e] = gelu_fast(fmaf(f1[e], a.inv_1, bb[e]));
                    }
                }
                h16x8 hh, hl;
                split8u_g(hv, hh, hl, guard);
                stage_top();
                b16_static_for<4>([&](auto cc) {
                    constexpr int p = 2 * decltype(cc)::value;
                    consume2(std::integral_constant<int, p>{}, hh, hl, acc[p], acc[p + 1]);
                });
                stage_end();
            };
            B16_STAMP(4);
            f32x4 fa0, fa1, fb0, fb1;
            fc1(fa0, fa1);
#pragma unroll 1
            for (int hg = 0; hg < 16; hg += 2) {
                fc1(fb0, fb1);
                act_fc2(hg, fa0, fa1);
                if (hg + 2 < 16) fc1(fa0, fa1);
                act_fc2(hg + 1, fb0, fb1);
            }
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                unstage_strip(&x1[4 * st], 16 * st);
                const f32x4 b4 = par4(B16P_B2 + 16 * st + 4 * g);
#pragma unroll
                for (int e = 0; e < 4; ++e) x1[4 * st + e] += fmaf(acc[st][e], a.inv_2, b4[e]);
            }
        }
        B16_STAMP(5);
        if (a.x_out) {
#pragma unroll
            for (int st = 0; st < 8; ++st) stage_strip(&x1[4 * st], 16 * st);
            flush_rows(a.x_out, 128, 0, t0);
        }
        B16_STAMP(6);
        // ---- stage Q -----------------------------------------------------------------------------------------------------------
        if constexpr (KQC > 0) {
            h16x8 bqh[KQC], bql[KQC];
            {
                float ln[32];
                layer_norm(x1, B16P_GQ, B16P_BQN, a.epsq, ln);
                if (a.ln_out) {
#pragma unroll
                    for (int st = 0; st < 8; ++st) stage_strip(&ln[4 * st], 16 * st);
                    flush_rows(a.ln_out, 128, 0, t0, a.ln_out_map);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) split8u_g(&ln[8 * c], bqh[c], bql[c], guard);
            }
            if constexpr (KQC > 4) {
                const float *e = a.extra + (tc / a.extra_div) * a.extra_ld;
#pragma unroll
                for (int c = 4; c < KQC; ++c) {
                    const float4 v0 = ldg4(e + 32 * (c - 4) + 4 * g), v1 = ldg4(e + 32 * (c - 4) + 16 + 4 * g);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    split8u_g(v, bqh[c], bql[c], guard);
                }
            }
            B16_STAMP(7);
            // the q stage proper: NQ / 128 groups of 128 outputs (5 stages each at KQC = 5), every group flushed as rows of q_out
            auto emit_q = [&](const h16x8 (&bh)[KQC], const h16x8 (&bl)[KQC], int bq_off, float inv_q, int kv16, float *q_out, int NQ,
                              bool stamps) {
                const int n_groups = NQ >> 7;
#pragma unroll 1
                for (int gq = 0; gq < n_groups; ++gq) {
                    b16_static_for<4>([&](auto ss) {                          // strips 2sp, 2sp+1 of the group, chunk-interleaved
                        constexpr int sp = decltype(ss)::value;
                        f32x4 qh0, qh1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) qh0[r] = qh1[r] = 0.f;
                        b16_static_for<KQC>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            constexpr int pg = sp * 2 * KQC + 2 * c;
                            if constexpr (pg % 8 == 0) stage_top();
                            consume2(std::integral_constant<int, pg % 8>{}, bh[c], bl[c], qh0, qh1);
                            if constexpr (pg % 8 == 6) stage_end();
                        });
                        const f32x4 ba = par4(bq_off + gq * 128 + 32 * sp + 4 * g), bb = par4(bq_off + gq * 128 + 32 * sp + 16 + 4 * g);
                        // kv16: the k group leaves as [hi fp16 x 32 | lo fp16 x 32] per 32-channel head, the v group as (hi, lo)
                        // half pairs in place of the floats -- the same 4 bytes per value, split ONCE here (split2u, bit for bit what
                        // the attention kernels' split8u would produce) instead of by every query tile that reads the row
                        const int kvmode = (kv16 && n_groups == 3) ? gq : 0;          // 0: floats, 1: k, 2: v
                        auto put_strip = [&](const float (&v)[4], int col0) {
                            if (kvmode == 0) { stage_strip(v, col0); return; }
                            h16x2 h01, l01, h23, l23;
                            split2u(f32x2{v[0], v[1]}, h01, l01);
                            split2u(f32x2{v[2], v[3]}, h23, l23);
                            const unsigned uh01 = __builtin_bit_cast(unsigned, h01), ul01 = __builtin_bit_cast(unsigned, l01);
                            const unsigned uh23 = __builtin_bit_cast(unsigned, h23), ul23 = __builtin_bit_cast(unsigned, l23);
                            float *row = Ot + j * B16_OLD;
                            if (kvmode == 1) {                            // channels c0 .. c0+3 of head col0 / 32, c0 = col0 % 32 + 4g
                                const int hb = col0 & ~31, c0 = (col0 & 31) + 4 * g;
                                *reinterpret_cast<uint2 *>(row + hb + (c0 >> 1)) = make_uint2(uh01, uh23);
                                *reinterpret_cast<uint2 *>(row + hb + 16 + (c0 >> 1)) = make_uint2(ul01, ul23);
                            } else {                                      // element e -> hi | lo << 16
                                const uint4 pk = make_uint4(__builtin_amdgcn_perm(ul01, uh01, 0x05040100u), __builtin_amdgcn_perm(ul01, uh01, 0x07060302u),
                                                            __builtin_amdgcn_perm(ul23, uh23, 0x05040100u), __builtin_amdgcn_perm(ul23, uh23, 0x07060302u));
                                *reinterpret_cast<uint4 *>(row + col0 + 4 * g) = pk;
                            }
                        };
                        float ov[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = fmaf(qh0[e], inv_q, ba[e]);
                        qmax = fmaxf(fmaxf(qmax, fmaxf(fabsf(ov[0]), fabsf(ov[1]))), fmaxf(fabsf(ov[2]), fabsf(ov[3])));
                        put_strip(ov, 32 * sp);
#pragma unroll
                        for (int e = 0; e < 4; ++e) ov[e] = fmaf(qh1[e], inv_q, bb[e]);
                        qmax = fmaxf(fmaxf(qmax, fmaxf(fabsf(ov[0]), fabsf(ov[1]))), fmaxf(fabsf(ov[2]), fabsf(ov[3])));
                        put_strip(ov, 32 * sp + 16);
                    });
                    flush_rows(q_out, NQ, gq * 128, t0);
                    if (stamps && gq < 3) B16_STAMP(8 + gq);
                }
            };
            if constexpr (FUSE) {
                // ---- this block's q | k | v (three groups of 128) stay in registers: a group = [8 * head + 0..7] = the lane's 8 channels of
                // a head.  q and k live until the softmax weights are known (16 registers), then v; x2 waits in the wave's LDS tile.
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 8; ++st) stage_strip(&x1[4 * st], 16 * st);
                __builtin_amdgcn_sched_barrier(0);
                auto q_group = [&](auto gg, float (&dst)[32]) {
                    constexpr int gq = decltype(gg)::value;
                    b16_static_for<4>([&](auto ss) {
                        constexpr int sp = decltype(ss)::value;
                        f32x4 qh0, qh1;
#pragma unroll
                        for (int r = 0; r < 4; ++r) qh0[r] = qh1[r] = 0.f;
                        b16_static_for<KQC>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            constexpr int pg = sp * 2 * KQC + 2 * c;
                            if constexpr (pg % 8 == 0) stage_top();
                            consume2(std::integral_constant<int, pg % 8>{}, bqh[c], bql[c], qh0, qh1);
                            if constexpr (pg % 8 == 6) stage_end();
                        });
                        const f32x4 ba = par4(B16P_BQ + gq * 128 + 32 * sp + 4 * g), bb = par4(B16P_BQ + gq * 128 + 32 * sp + 16 + 4 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dst[8 * sp + e] = fmaf(qh0[e], a.inv_q, ba[e]);
                            dst[8 * sp + 4 + e] = fmaf(qh1[e], a.inv_q, bb[e]);
                        }
                    });
                };
                float wgt[4][4];
                {
                    float qv[32], kv[32];
                    q_group(std::integral_constant<int, 0>{}, qv);
                    q_group(std::integral_constant<int, 1>{}, kv);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float q[8], k[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { q[e] = qv[8 * c + e]; k[e] = kv[8 * c + e]; }
                        sib_weights(q, k, wgt[c]);
                    }
                }
                B16_STAMP(8);
                __builtin_amdgcn_sched_barrier(0);
                // ---- the self-edge block (second parameter set, the stages that follow in the stream): sibling attention on the way in
                h16x8 bm2h[4], bm2l[4];
                {
                    float vvv[32];
                    q_group(std::integral_constant<int, 2>{}, vvv);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float vv[8], o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) vv[e] = vvv[8 * c + e];
                        sib_apply(wgt[c], vv, o);
                        split8u_g(o, bm2h[c], bm2l[c], guard);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 8; ++st) unstage_strip(&x1[4 * st], 16 * st);
                proj_stage(bm2h, bm2l, B16P2_BP, a.inv_p2);
                __builtin_amdgcn_sched_barrier(0);
                B16_STAMP(9);
                if (a.x_out2) {
#pragma unroll
                    for (int st = 0; st < 8; ++st) stage_strip(&x1[4 * st], 16 * st);
                    flush_rows(a.x_out2, 128, 0, t0);
                }
                h16x8 b2h[KQC], b2l[KQC];
                {
                    float ln[32];
                    layer_norm(x1, B16P2_GQ, B16P2_BQN, a.epsq2, ln);
                    if (a.ln_out2) {
#pragma unroll
                        for (int st = 0; st < 8; ++st) stage_strip(&ln[4 * st], 16 * st);
                        flush_rows(a.ln_out2, 128, 0, t0, a.ln_out2_map);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) split8u_g(&ln[8 * c], b2h[c], b2l[c], guard);
                }
                {
                    const float *e = a.extra2 + (tc / a.extra2_div) * a.extra2_ld;
                    const float4 v0 = ldg4(e + 4 * g), v1 = ldg4(e + 16 + 4 * g);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    split8u_g(v, b2h[4], b2l[4], guard);
                }
                if (a.q_out2) emit_q(b2h, b2l, B16P2_BQ, a.inv_q2, a.kv16_2, a.q_out2, a.NQ2, false);
                B16_STAMP(10);
            } else {
                if (a.q_out) emit_q(bqh, bql, B16P_BQ, a.inv_q, a.kv16, a.q_out, a.NQ, true);
            }
        }
        if constexpr ((DBG & 32) != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            B16_STAMP(11);
            if (lane == 0 && blockIdx.x < 64) {
                unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 8 + wv) * 16;
                o[12] = tk_bar; o[13] = tk_use; o[14] = tk_commit;
            }
        }
    };
    int tile = blockIdx.x;
    if constexpr (HOIST) {
        tile_body(tile, std::true_type{});
        tile += gridDim.x;
    }
#pragma unroll 1
    for (; tile < a.n_tiles; tile += gridDim.x) tile_body(tile, std::false_type{});
    if constexpr ((DBG & 32) != 0) {
        __syncthreads();
        if (tid == 0) {
            a.stamps[64 * 8 * 16 + (size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            a.stamps[64 * 8 * 16 + (size_t)blockIdx.x * 4 + 2] =
                ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));
        }
    }
#ifndef NMRF_NO_CLK_HOOK
    if (a.clk && tid == 0) {
        a.clk[(size_t)blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        a.clk[(size_t)blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    split_guard_commit(guard, a.range_flag);
    if (a.range_flag && !(qmax < 65520.0f)) atomicOr(a.range_flag, 1);   // (a NaN in q_out has a NaN operand upstream: caught by `guard`)
}

// w [N,K] fp32 -> N/16 x Kp/32 pairs in [strip][chunk] order: lane (i = l & 15, g = l >> 4) slot jj holds
// scale * w[16*strip + i][32*chunk + split_kslot16(jj, g)], zero beyond K; hi fragment (1 KB) then lo fragment.
__global__ __launch_bounds__(256) void pack_split_weight16_kernel(const float *__restrict__ w, int N, int K, int KC, float scale,
                                                                 uint4 *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = (int64_t)(N / 16) * KC * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int64_t pair = idx >> 6;
    const int c = (int)(pair % KC), s = (int)(pair / KC);
    const int n = s * 16 + (lane & 15), g = lane >> 4;
    float v[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int k = 32 * c + split_kslot16(jj, g);
        v[jj] = k < K ? w[(int64_t)n * K + k] * scale : 0.f;
    }
    h16x8 vh, vl;
    split8u(v, vh, vl);
    out[pair * 128 + lane] = *reinterpret_cast<const uint4 *>(&vh);
    out[pair * 128 + 64 + lane] = *reinterpret_cast<const uint4 *>(&vl);
}

extern "C" int nmrf_pack_split_weight16_f32(const float *w, int N, int K, int Kp, float scale, void *out, void *stream) {
    if (!w || !out) return NMRF_ENULL;
    if (N < 16 || (N & 15) || K < 1 || Kp < K || (Kp & 31) || !(scale > 0.f)) return NMRF_EINVAL;
    const int KC = Kp / 32;
    const int64_t total = (int64_t)(N / 16) * KC * 64;
    hipLaunchKernelGGL(pack_split_weight16_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, N, K,
                       KC, scale, reinterpret_cast<uint4 *>(out));
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES
static int g_b16_variant = 0;
static unsigned long long *g_b16_stamps = nullptr;
extern "C" int nmrf_debug_nmp_block16_variant(int v) { g_b16_variant = v; return NMRF_OK; }
// stamps: device buffer of 64 blocks x 8 waves x 16 words + 4 words per block of the grid, or NULL to switch the timing build off
extern "C" int nmrf_debug_nmp_block16_timing(void *stamps) { g_b16_stamps = (unsigned long long *)stamps; return NMRF_OK; }
// one thread that writes s_memrealtime: a mark on the stream between two launches
__global__ void b16_realtime_mark_kernel(unsigned long long *dst) { *dst = __builtin_amdgcn_s_memrealtime(); }
extern "C" int nmrf_debug_realtime_mark(void *dst, void *stream) {
    hipLaunchKernelGGL(b16_realtime_mark_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)dst);
    return nmrf_launch_status();
}
#endif

// Measurement hook (bench.py's sustained_clock_ghz): while buf != NULL every block of the block-kernel launches of THIS process leaves
// its entry / exit readings of the CU's shader-clock counter and of the chip's 100 MHz counter in buf[4 * blockIdx.x ..] (launches
// with more than capacity_blocks blocks are not recorded).  The clock the matrix pipe ran at = sum(t1 - t0) / sum(r1 - r0) x 100 MHz.
static unsigned long long *g_b16_clk = nullptr;
static int g_b16_clk_cap = 0;
extern "C" int nmrf_nmp_block16_clock_records(unsigned long long *buf, int capacity_blocks) {
    if (buf && capacity_blocks < 1) return NMRF_EINVAL;
    g_b16_clk = buf;
    g_b16_clk_cap = buf ? capacity_blocks : 0;
    return NMRF_OK;
}

template <bool MLP, int KQC, int DBG = 0, bool FUSE = false>
static int launch_nmp_block16(const NmpBlock16Args &a_in, hipStream_t st) {
    NmpBlock16Args a = a_in;
#ifdef NMRF_DEBUG_PROBES
    if constexpr (DBG == 0 && !FUSE) {
        if (g_b16_stamps) {
            NmpBlock16Args b = a;
            b.stamps = g_b16_stamps;
            if (g_b16_variant == 256) return launch_nmp_block16<MLP, KQC, 288>(b, st);
            return launch_nmp_block16<MLP, KQC, 32>(b, st);
        }
        switch (g_b16_variant) {
            case 1: return launch_nmp_block16<MLP, KQC, 1>(a, st);
            case 2: return launch_nmp_block16<MLP, KQC, 2>(a, st);
            case 4: return launch_nmp_block16<MLP, KQC, 4>(a, st);
            case 8: return launch_nmp_block16<MLP, KQC, 8>(a, st);
            case 7: return launch_nmp_block16<MLP, KQC, 7>(a, st);
            case 16: return launch_nmp_block16<MLP, KQC, 16>(a, st);
            case 24: return launch_nmp_block16<MLP, KQC, 24>(a, st);
            case 256: return launch_nmp_block16<MLP, KQC, 256>(a, st);
            default: break;
        }
    }
#endif
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)B16_PAR_OFF + (FUSE ? B16P2_FLOATS : B16P_FLOATS) * sizeof(float);
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(nmp_block16_kernel<MLP, KQC, DBG, FUSE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMRF_ELAUNCH;
        n_cu_dev[dev] = prop.multiProcessorCount;
    }
    const int grid = a.n_tiles < n_cu_dev[dev] ? a.n_tiles : n_cu_dev[dev];
    a.clk = (g_b16_clk && grid <= g_b16_clk_cap) ? g_b16_clk : nullptr;
    hipLaunchKernelGGL((nmp_block16_kernel<MLP, KQC, DBG, FUSE>), dim3(grid), dim3(B16_THR), lds, st, a);
    return nmrf_launch_status();
}

extern "C" int nmrf_nmp_block16_f32(const float *x, const float *msg, const float *attn_qkv, int attn_n, const void *stream_w,
                                    int total_stages, const float *bp,
                                    const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                                    const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld,
                                    int extra_div, const float *bq, int has_mlp, int KQ, int NQ, int64_t T, const float *inv_scales,
                                    float *x_out, float *q_out, float *ln_out, const int *ln_out_map, int kv16, int *range_flag,
                                    void *stream) {
    if (!x || !stream_w || !inv_scales) return NMRF_ENULL;
    if (kv16 && (!q_out || NQ != 384)) return NMRF_EINVAL;
    if (T < 1 || ceil_div64(T, B16_TOK) > 0x7fffffff) return NMRF_EINVAL;
    if (has_mlp && (!ln2_g || !ln2_b || !b1 || !b2)) return NMRF_ENULL;
    if (attn_qkv && (msg || has_mlp || attn_n != 4 || (T & 3))) return NMRF_EINVAL;     // self-edge attention: 4 siblings, proj-only blocks
    if (KQ != 0 && KQ != 128 && KQ != 160 && KQ != 192) return NMRF_EINVAL;
    if (KQ && (!lnq_g || !lnq_b)) return NMRF_ENULL;
    if (KQ > 128 && (!extra || extra_ld < KQ - 128 || (extra_ld & 3) || extra_div < 1)) return NMRF_EINVAL;
    if (q_out && (KQ == 0 || NQ < 128 || (NQ & 127) || NQ > 512)) return NMRF_EINVAL;
    if (ln_out && KQ == 0) return NMRF_EINVAL;
    if (!q_out && !ln_out && !x_out) return NMRF_ENULL;
    const int want = ((msg || attn_qkv) ? 4 : 0) + (has_mlp ? 32 : 0) + (q_out ? (NQ / 128) * (KQ / 32) : 0);
    if (total_stages != want || total_stages < 1) return NMRF_EINVAL;
    NmpBlock16Args a{x, msg, attn_qkv, reinterpret_cast<const b16_u32x4 *>(stream_w), total_stages, bp, ln2_g, ln2_b, b1, b2, lnq_g, lnq_b, extra,
                     extra_ld, extra_div, bq, x_out, q_out, ln_out, ln_out_map, T, (int)ceil_div64(T, B16_TOK), eps2, epsq, NQ,
                     inv_scales[0], inv_scales[1], inv_scales[2], inv_scales[3], nullptr, range_flag, kv16 ? 1 : 0,
                     nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    const int kqc = KQ / 32;
    if (has_mlp) {
        switch (kqc) {
            case 0: return launch_nmp_block16<true, 0>(a, st);
            case 4: return launch_nmp_block16<true, 4>(a, st);
            case 5: return launch_nmp_block16<true, 5>(a, st);
            case 6: return launch_nmp_block16<true, 6>(a, st);
        }
    } else {
        switch (kqc) {
            case 0: return launch_nmp_block16<false, 0>(a, st);
            case 4: return launch_nmp_block16<false, 4>(a, st);
            case 5: return launch_nmp_block16<false, 5>(a, st);
            case 6: return launch_nmp_block16<false, 6>(a, st);
        }
    }
    return NMRF_EINVAL;
}

// A full block and the self-edge block behind it in one launch (the FUSE form of the kernel): x, msg [T,128] (the window attention's
// output) -> proj + residual + MLP -> LayerNorm | extra -> q | k | v of the self-edge attention (kept in registers) -> the 4 x 4 sibling
// attention -> proj2 + residual -> LayerNorm2 | extra2 -> q_out2 [T, NQ2] (kv16_2: k | v as split fp16 pairs), x_out2 [T,128].
// (msg == NULL, no bp / ln2 / b1 / b2: the first block is its q stage alone -- the entry of the inference stage -- 15 stages.)
// stream_w: the first block's stream (4 + 32 + 15 stages, nmrf_nmp_block16_f32 with has_mlp, KQ = 160, NQ = 384) followed by the second's
// (4 + NQ2 / 128 * 5 stages: KQ = 160, no MLP); inv_scales: HOST array of 6 floats -- proj, fc1, fc2, q of the first block, proj, q of
// the second.  T a multiple of 4 (four sibling labels per pixel).  Same bits as the two launches.
extern "C" int nmrf_nmp_block16_pair_f32(const float *x, const float *msg, const void *stream_w, int total_stages, const float *bp,
                                         const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                                         const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld, int extra_div,
                                         const float *bq, const float *bp2, const float *lnq2_g, const float *lnq2_b, float epsq2,
                                         const float *extra2, int extra2_ld, int extra2_div, const float *bq2, int NQ2, int64_t T,
                                         const float *inv_scales, float *x_out2, float *q_out2, float *ln_out2, const int *ln_out2_map,
                                         int kv16_2, int *range_flag, void *stream) {
    if (!x || !stream_w || !inv_scales || !lnq_g || !lnq_b || !lnq2_g || !lnq2_b || !extra || !extra2) return NMRF_ENULL;
    const bool full = msg != nullptr;                 // msg == NULL: the first block is a q stage alone (no proj, no MLP: the stage's entry)
    if (full && (!ln2_g || !ln2_b || !b1 || !b2)) return NMRF_ENULL;
    if (!full && (ln2_g || ln2_b || b1 || b2 || bp)) return NMRF_EINVAL;
    if (T < 4 || (T & 3) || ceil_div64(T, B16_TOK) > 0x7fffffff) return NMRF_EINVAL;
    if (extra_ld < 32 || (extra_ld & 3) || extra_div < 1 || extra2_ld < 32 || (extra2_ld & 3) || extra2_div < 1) return NMRF_EINVAL;
    if (q_out2 && (NQ2 < 128 || (NQ2 & 127) || NQ2 > 512)) return NMRF_EINVAL;
    if (kv16_2 && (!q_out2 || NQ2 != 384)) return NMRF_EINVAL;
    if (!q_out2 && !ln_out2 && !x_out2) return NMRF_ENULL;
    const int want = (full ? 4 + 32 : 0) + 15 + 4 + (q_out2 ? (NQ2 / 128) * 5 : 0);
    if (total_stages != want) return NMRF_EINVAL;
    NmpBlock16Args a{x, msg, nullptr, reinterpret_cast<const b16_u32x4 *>(stream_w), total_stages, bp, ln2_g, ln2_b, b1, b2, lnq_g, lnq_b, extra,
                     extra_ld, extra_div, bq, nullptr, nullptr, nullptr, nullptr, T, (int)ceil_div64(T, B16_TOK), eps2, epsq, 384,
                     inv_scales[0], inv_scales[1], inv_scales[2], inv_scales[3], nullptr, range_flag, 0,
                     bp2, lnq2_g, lnq2_b, extra2, bq2, extra2_ld, extra2_div, x_out2, q_out2, ln_out2, ln_out2_map, epsq2, inv_scales[4],
                     inv_scales[5], NQ2, kv16_2 ? 1 : 0};
    return full ? launch_nmp_block16<true, 5, 0, true>(a, (hipStream_t)stream) : launch_nmp_block16<false, 5, 0, true>(a, (hipStream_t)stream);
}

#ifdef NMRF_DEBUG_PROBES
// attainable v_mfma_f32_16x16x32_f16 rate: CHAINS independent accumulators per wave, iters x 24 MFMAs each, constant operands;
// THREADS = 256 (one wave per SIMD) or 512 (two)
template <int CHAINS, int THREADS>
__global__ __launch_bounds__(THREADS) void mfma16x16_peak_kernel(int iters, float *__restrict__ out) {
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    h16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (threadIdx.x ^ k)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mfma16x16h(a, b, acc[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) sum += acc[c][0] + acc[c][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
extern "C" int nmrf_debug_mfma16x16_peak(int chains, int threads, int iters, int blocks, float *out, void *stream) {
    if (!out) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
#define PK(C, T) if (chains == C && threads == T) { hipLaunchKernelGGL((mfma16x16_peak_kernel<C, T>), dim3(blocks), dim3(T), 0, st, iters, out); return nmrf_launch_status(); }
    PK(1, 256) PK(2, 256) PK(4, 256) PK(8, 256) PK(1, 512) PK(2, 512) PK(4, 512) PK(8, 512)
#undef PK
    return NMRF_EINVAL;
}
#endif

// self-test of the 16x16x32 form: out[16x16] = A[16,K] . B[K,16] (row-major fp32, K % 32 == 0) on one wave, split operands with
// the k slots in split_kslot16 order on both sides
__global__ __launch_bounds__(64) void selftest_mfma16x16_kernel(const float *__restrict__ A, const float *__restrict__ Bm, int K,
                                                               float *__restrict__ out) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        float av[8], bv[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = k0 + split_kslot16(jj, g);
            av[jj] = A[j * K + k];
            bv[jj] = Bm[k * 16 + j];
        }
        h16x8 ah, al, bh, bl;
        split8u(av, ah, al);
        split8u(bv, bh, bl);
        split_mma16(ah, al, bh, bl, acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(4 * g + r) * 16 + j] = acc[r];
}

extern "C" int nmrf_selftest_mfma16x16_f16split(const float *A, const float *Bm, int K, float *out, void *stream) {
    if (!A || !Bm || !out) return NMRF_ENULL;
    if (K < 32 || K > 1024 || (K & 31)) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_mfma16x16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, K, out);
    return nmrf_launch_status();
}
