// One message-passing block's per-token linear algebra in ONE kernel:
//
//     x1  = x + msg . Wp^T + bp                                  attention output projection + residual      (NMP.py:100-106, 356-361, 566-571)
//     x2  = x1 + GELU(LN2(x1) . W1^T + b1) . W2^T + b2           timm Mlp, pre-norm                           (NMP.py:337, 361-362, 537, 572-573)
//     qkv = [LNq(x2) | extra] . Wq^T + bq                        the NEXT block's fused q|k|v operand         (NMP.py:90-96, 343-350, 544-556)
//     or    ln_out = LNq(x2)                                     the stage's final norm                       (NMP.py:658-659, 789-790, 891-892)
//
// replacing, per block, token_linear(proj) + token_linear_pipe(LN+fc1+GELU) + hipBLASLt fc2 + token_linear_pipe(LN+q|k|v) of
// round 1: the 512-wide hidden tile, x1 and both LayerNorm operands never leave the CU (HBM traffic per token 3.2 KB instead of
// 10.3 KB) and the four contractions run on the fp16 matrix pipe with split operands (split_mfma.h): 3 x 32 cycles per
// 16-deep k chunk instead of 8 x 64 on the fp32 MFMA.
//
// Formulation: everything is computed TRANSPOSED,  out^T[n, t] = W[n, :] . act^T[:, t]:  weights are the MFMA A operand,
// activations the B operand with the token on the lane (j = lane & 31).  A wave owns 32 tokens for the whole chain and needs
// no other wave's data: LayerNorm statistics are sums over a lane's own registers plus one half-wave swap, and the C/D
// registers of one contraction ARE the B operand of the next (k slot order split_kslot(), same trick as S^T -> P^T in the
// attention kernels), so activations never move between lanes.  Block = 4 waves = 128 tokens; the only thing the waves share
// is the WEIGHT STREAM: all A fragments of the kernel in consumption order, 2 KB (hi + lo', 64 lanes x 16 B each) per
// (32-row strip, 16-deep k chunk) "pair", read once per block from L2 in 16 KB stages through a 3-deep LDS ring
// (global -> registers one stage ahead, registers -> LDS, one barrier per stage).  Per 128 tokens the stream is 816 KB
// (K_q = 160): 85 B/clk/CU of LDS reads next to a fully busy matrix pipe.
//
// Outputs leave through a wave-private LDS tile as whole 512-byte rows (token-major fp32, same layouts as round 1).
#include "common.h"
#include "../../include/nmrf_hip_debug.h"      // tools / test build only (not in libnmrf_hip.so)
#include "split_mfma.h"
#include <type_traits>
#include <utility>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));    // (HIP's uint4 is a struct: arrays of it are left in scratch)

#define NB_TOK 128                 // tokens per block (4 waves x 32)
#define NB_STAGE_U4 1024           // uint4 per stage (16 KB = 8 pairs)
#define NB_RING 3
#define NB_FD 1                    // default pipeline depths of the product build (see the kernel template)
#define NB_PF 4
#define NB_TOUCH_AHEAD 4           // stages between the L2 touch and the real fetch of a stage
#define NB_DUMP_OFF (NB_RING * NB_STAGE_U4 * 16 + 4 * 32 * NB_OLD * 4)     // 1 KB of LDS the touch loads land in (never read)
#define NB_PAR_OFF (NB_DUMP_OFF + 1024)                                    // bias / LayerNorm vectors, staged once per block
// float offsets inside the parameter area
#define NBP_BP 0
#define NBP_G2 128
#define NBP_B2N 256
#define NBP_B1 384
#define NBP_B2 896
#define NBP_GQ 1024
#define NBP_BQN 1152
#define NBP_BQ 1280
#define NBP_FLOATS (1280 + 512)
#define NB_OLD 132                 // floats per row of the output staging tile (128 + 4: conflict-free b128 writes)

template <class F, int... I>
__device__ __forceinline__ void nb_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void nb_static_for(F &&f) {
    nb_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

struct NmpBlockArgs {
    const float *x;            // [T,128] residual stream
    const float *msg;          // [T,128] attention output before its projection, or NULL (x1 = x, no proj stage in the stream)
    const u32x4 *stream;       // weight fragment stream (nmrf_pack_split_weight_f32 pairs in consumption order)
    int total_stages;          // 16 KB stages in the stream
    const float *bp;           // [128] or NULL
    const float *ln2_g, *ln2_b, *b1, *b2;          // MLP stage (MLP = true)
    const float *lnq_g, *lnq_b;                    // LayerNorm of the q stage / final norm
    const float *extra;        // [ceil(T / extra_div), extra_ld] side input rows (Fourier31 padded to 32, or context64)
    int extra_ld, extra_div;
    const float *bq;           // [NQ] or NULL
    float *x_out;              // [T,128] x2 (x1 when MLP = false), or NULL
    float *q_out;              // [T,NQ] or NULL
    float *ln_out;             // [T,128] LNq(x2) or NULL
    const int *ln_out_map;     // row of ln_out per token (negative: dropped) or NULL
    int64_t T;
    int n_tiles;
    float eps2, epsq;
    int NQ;
    float inv_p, inv_1, inv_2, inv_q;   // 1 / (power-of-two scale the proj / fc1 / fc2 / q weights were packed with)
    unsigned long long *stamps;   // debug build (nmrf_debug_nmp_block_timing): s_memtime per wave and phase of the first 64 blocks
};

// MLP: run the fc1-GELU-fc2 stage.  KQC: k chunks (of 16) of the q stage's operand [LNq(x2) | extra]: 0 = no q stage,
// 8 = LayerNorm columns only, 10 = + 32 side columns (Fourier31 + 0), 12 = + 64 side columns (context).
// FD: stages of latency budget of the global fetch (1 or 2 register sets); PF: pairs read ahead from LDS; TOUCH: L2 warming.
template <bool MLP, int KQC, int FD, int PF, bool TOUCH, int DBG = 0>      // DBG: timing experiments of the debug build (wrong results)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void nmp_block_kernel(NmpBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *ring = reinterpret_cast<u32x4 *>(smem);                                        // [3][1024] x 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    float *Ot = reinterpret_cast<float *>(smem + NB_RING * NB_STAGE_U4 * 16) + wv * 32 * NB_OLD;   // wave-private [32][132]
    // Biases and LayerNorm vectors live in LDS: read from global inside the stage loop they would queue behind the weight-stream
    // fetches on the in-order vmcnt counter, and every bias wait would expose a full L2 round trip (measured: the stage loop
    // ran at 60 cycles per MFMA with the loads in it).
    float *Par = reinterpret_cast<float *>(smem + NB_PAR_OFF);
    {
        auto put = [&](int off, const float *src, int n) {
            for (int i = tid; i < n; i += 256) Par[off + i] = src ? src[i] : 0.f;
        };
        put(NBP_BP, a.bp, 128); put(NBP_G2, a.ln2_g, 128); put(NBP_B2N, a.ln2_b, 128); put(NBP_B1, a.b1, 512);
        put(NBP_B2, a.b2, 128); put(NBP_GQ, a.lnq_g, 128); put(NBP_BQN, a.lnq_b, 128); put(NBP_BQ, a.bq, a.bq ? a.NQ : 512);
    }
    auto par4 = [&](int off) { return *reinterpret_cast<const f32x4 *>(Par + off); };
#ifdef NMRF_DEBUG_PROBES
#define NB_STAMP(k) do { if (a.stamps && lane == 0 && blockIdx.x < 64) \
        a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NB_STAMP(k) do { } while (0)
#endif
    NB_STAMP(0);

    // ---- weight stream ----------------------------------------------------------------------------------------------------
    // Stage s (16 KB = 8 pairs) lives in ring slot s % 3.  Timeline of a wave at stage g:  barrier g | consume stage g (its
    // first PF pairs were read from LDS during stage g-1) while reading the first PF pairs of stage g+1 ahead | commit stage g+2
    // (in registers since the end of stage g-FD) to slot (g+2) % 3 = the slot stage g-1 vacated | fetch stage g+2+FD.
    // Barrier g orders: every wave has finished reading stage g-1, and stage g+1 (committed during stage g-1) is visible.
    // TOUCH (experiment, off in the product build: no measurable effect, 72.3 vs 72.2 us): wave 0 of each block pulls 1 KB of the
    // stage TOUCH_AHEAD stages further on into L2 with an LDS-DMA load nobody reads; blocks on the same XCD (blockIdx % 8) touch
    // different sixteenths.
    u32x4 R[FD][4];
    int src_stage = 0;                          // next stage to fetch from global (wraps: persistent blocks re-read the stream)
    int wr_slot = 0, rd_slot = 0;               // ring slots of the next commit / of the stage being consumed
    int par = 0;                                // register set of the next commit (FD == 2)
    auto fetch = [&](u32x4 (&r)[4]) {
        const u32x4 *p = a.stream + (size_t)src_stage * NB_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = p[256 * i];
        src_stage = (src_stage + 1 == a.total_stages) ? 0 : src_stage + 1;
    };
    auto commit = [&](const u32x4 (&r)[4]) {
        u32x4 *d = ring + wr_slot * NB_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[256 * i] = r[i];
        wr_slot = (wr_slot == NB_RING - 1) ? 0 : wr_slot + 1;
    };
    const u32x4 *cur = ring, *nxt = ring + NB_STAGE_U4;      // stage being consumed / the one after it
    h16x8 fqh[PF], fql[PF];                                  // fragment queue: pairs p .. p+PF-1 of the stream position
    auto read_pair = [&](const u32x4 *base, int p, h16x8 &h, h16x8 &l) {
        h = *reinterpret_cast<const h16x8 *>(base + p * 128 + lane);
        l = *reinterpret_cast<const h16x8 *>(base + p * 128 + 64 + lane);
    };
    const int touch_slice = (blockIdx.x >> 3) & 15;
    auto touch = [&]() {
        if constexpr (TOUCH) {
            if (wv == 0) {
                int ts = src_stage + NB_TOUCH_AHEAD;
                if (ts >= a.total_stages) ts -= a.total_stages;
                if (ts >= a.total_stages) ts -= a.total_stages;
                const u32x4 *p = a.stream + (size_t)ts * NB_STAGE_U4 + touch_slice * 64 + lane;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(NB_DUMP_OFF) : "memory", "m0");
            }
        }
    };
    bool have_barrier = true;                   // the barrier of the very first stage is the one in the prologue
#ifdef NMRF_DEBUG_PROBES
    unsigned long long tk_bar = 0, tk_use = 0, tk_commit = 0, tk0 = 0, tk1 = 0;     // per-wave totals: barrier wait / consume / commit+fetch
#define NB_TICK(v) do { if (a.stamps) v = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NB_TICK(v) do { } while (0)
#endif
    auto stage_top = [&]() {
#ifdef NMRF_DEBUG_PROBES
        NB_TICK(tk0);
#endif
        if (!have_barrier && !(DBG & 1)) __syncthreads();
        have_barrier = false;
        touch();
#ifdef NMRF_DEBUG_PROBES
        NB_TICK(tk1);
        tk_bar += tk1 - tk0;
#endif
    };
    auto stage_end = [&]() {
#ifdef NMRF_DEBUG_PROBES
        NB_TICK(tk0);
        tk_use += tk0 - tk1;
#endif
        if constexpr (!(DBG & 2)) {
            if constexpr (FD == 2) {
                if (par) { commit(R[1]); fetch(R[1]); } else { commit(R[0]); fetch(R[0]); }
                par ^= 1;
            } else {
                commit(R[0]);
                fetch(R[0]);
            }
        }
#ifdef NMRF_DEBUG_PROBES
        NB_TICK(tk1);
        tk_commit += tk1 - tk0;
#endif
        rd_slot = (rd_slot == NB_RING - 1) ? 0 : rd_slot + 1;
        cur = nxt;
        nxt = ring + ((rd_slot == NB_RING - 1) ? 0 : rd_slot + 1) * NB_STAGE_U4;
    };
    // pair P (compile-time, 0..7) of the current stage: take it from the queue, refill the queue entry with pair P + PF (of this
    // stage or the next), then run the three MFMAs
    auto consume = [&](auto pc, const h16x8 &bh, const h16x8 &bl, f32x16 &acc) {
        constexpr int P = decltype(pc)::value;
        const h16x8 ah = fqh[P % PF], al = fql[P % PF];
        if constexpr (!(DBG & 4)) {
            if constexpr (P + PF < 8) read_pair(cur, P + PF, fqh[P % PF], fql[P % PF]);
            else read_pair(nxt, P + PF - 8, fqh[P % PF], fql[P % PF]);
        }
        // Pin the read-ahead: LDS reads may not sink below this point and MFMAs may not rise above it (VALU / SALU / VMEM /
        // transcendental instructions may still cross, so the compiler keeps interleaving the GELU with the MFMAs).  Without it
        // the scheduler moved every fragment read down to just before its MFMAs and each pair paid the LDS latency:
        // ~2.5k cycles per 24-MFMA stage whatever else the stage did (s_memtime census, profiles/r02d_block_phases.txt).
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
        if constexpr (!(DBG & 8)) split_mma1(ah, al, bh, bl, acc);
        else acc[P] += (float)ah[0] + (float)bl[1];
    };

    fetch(R[0]); commit(R[0]);                  // stage 0 -> slot 0
    fetch(R[0]); commit(R[0]);                  // stage 1 -> slot 1
    fetch(R[0]);                                // stage 2, committed at the end of stage 0
    if constexpr (FD == 2) fetch(R[1]);         // stage 3, committed at the end of stage 1
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PF; ++p) read_pair(cur, p, fqh[p], fql[p]);

    // output staging: the 16 C/D registers of a 32-channel strip <-> columns [col0, col0+32) of the wave's tile (lane-private
    // addresses: a lane reads back exactly what it wrote)
    auto stage_strip = [&](const float *v, int col0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4 *>(Ot + j * NB_OLD + col0 + 8 * q + 4 * hi) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    };
    auto unstage_strip = [&](float *v, int col0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(Ot + j * NB_OLD + col0 + 8 * q + 4 * hi);
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    };
    // whole rows of the tile -> dst[t, col0 .. col0+128): lanes 0-31 one row, lanes 32-63 the next (512 B each)
    auto flush_rows = [&](float *dst, int ld, int col0, int64_t t0, const int *map = nullptr) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * i + hi;
            const float4 v = *reinterpret_cast<const float4 *>(Ot + row * NB_OLD + 4 * j);
            if (t0 + row < a.T) {
                int64_t orow = t0 + row;
                if (map) orow = map[orow];
                if (orow >= 0) stg4(dst + (size_t)orow * ld + col0 + 4 * j, v);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // LayerNorm over the 128 channels of each token: a lane holds 64 of them (C/D layout, 4 strips x 16), lane ^ 32 the rest
    auto layer_norm = [&](const float (&v)[4][16], int g_off, int b_off, float eps, float (&o)[4][16]) {
        float s = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += v[st][r];
        const float mean = half_sum(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = v[st][r] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(half_sum(q) * (1.0f / 128.0f) + eps);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 gv = par4(g_off + st * 32 + 8 * qd + 4 * hi), bv = par4(b_off + st * 32 + 8 * qd + 4 * hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[st][4 * qd + e] = (v[st][4 * qd + e] - mean) * rstd * gv[e] + bv[e];
            }
    };

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t t0 = (int64_t)tile * NB_TOK + wv * 32;                 // first token of this wave
        const int64_t tq = t0 + j;
        const int64_t tc = tq < a.T ? tq : a.T - 1;                            // clamped: straight-line loads, masked stores
        float x1[4][16];                                                       // residual stream, C/D layout: channel = 32*st + mfma_row(r, hi)
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = ldg4(a.x + tc * 128 + st * 32 + 8 * q + 4 * hi);
                x1[st][4 * q] = v.x; x1[st][4 * q + 1] = v.y; x1[st][4 * q + 2] = v.z; x1[st][4 * q + 3] = v.w;
            }

        NB_STAMP(1);
        f32x16 acc[4];
        // ---- stage P: x1 = x + msg . Wp^T + bp ------------------------------------------------------------------------------
        if (a.msg) {
            h16x8 bmh[8], bml[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v0 = ldg4(a.msg + tc * 128 + 16 * c + 4 * hi), v1 = ldg4(a.msg + tc * 128 + 16 * c + 8 + 4 * hi);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                split8u(v, bmh[c], bml[c]);
            }
            nb_static_for<4>([&](auto ss) {
                constexpr int st = decltype(ss)::value;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[st][r] = 0.f;
                stage_top();
                nb_static_for<8>([&](auto cc) { consume(cc, bmh[decltype(cc)::value], bml[decltype(cc)::value], acc[st]); });
                stage_end();
            });
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(NBP_BP + st * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x1[st][4 * q + e] += fmaf(acc[st][4 * q + e], a.inv_p, b4[e]);
                }
        }

        NB_STAMP(2);
        // ---- stage M: x2 = x1 + fc2(GELU(fc1(LN2(x1)))) ------------------------------------------------------------------------
        if constexpr (MLP) {
            h16x8 bnh[8], bnl[8];                                              // LN2(x1) as the B operand of fc1 (k = channel)
            {
                float ln[4][16];
                layer_norm(x1, NBP_G2, NBP_B2N, a.eps2, ln);
#pragma unroll
                for (int c = 0; c < 8; ++c) split8u(&ln[c >> 1][8 * (c & 1)], bnh[c], bnl[c]);
            }
            // x1 waits in the wave's LDS tile while the 512-wide hidden layer occupies the registers
#pragma unroll
            for (int st = 0; st < 4; ++st) stage_strip(x1[st], st * 32);
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[st][r] = 0.f;
            // hidden strip hs (32 of the 512 hidden channels): fc1 -> fh/fx, GELU, then its 2 k chunks of fc2 into all 4 output
            // strips.  Stream order: W1[0] | W1[1], W2s[0] | W1[2], W2s[1] | ... | W1[15], W2s[14] | W2s[15]: the MFMAs of
            // fc1(hs+1) are issued before GELU(hs) so that the activation's VALU work runs under them.
            auto fc1 = [&](f32x16 &fh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) fh[r] = 0.f;
                stage_top();
                nb_static_for<8>([&](auto cc) { consume(cc, bnh[decltype(cc)::value], bnl[decltype(cc)::value], fh); });
                stage_end();
            };
            auto act_fc2 = [&](int hs, const f32x16 &fh) {
                float hv[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(NBP_B1 + hs * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) hv[4 * q + e] = gelu_fast(fmaf(fh[4 * q + e], a.inv_1, b4[e]));
                }
                h16x8 hh[2], hl[2];
                split8u(hv, hh[0], hl[0]);
                split8u(hv + 8, hh[1], hl[1]);
                stage_top();
                nb_static_for<8>([&](auto cc) {
                    constexpr int p = decltype(cc)::value;                     // pair p = (output strip p / 2, k chunk p % 2)
                    consume(cc, hh[p & 1], hl[p & 1], acc[p >> 1]);
                });
                stage_end();
            };
            f32x16 fa, fb;
            NB_STAMP(3);
            fc1(fa);
#pragma unroll 1
            for (int hs = 0; hs < 16; hs += 2) {
                fc1(fb);
                act_fc2(hs, fa);
                if (hs + 2 < 16) fc1(fa);
                act_fc2(hs + 1, fb);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                unstage_strip(x1[st], st * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 b4 = par4(NBP_B2 + st * 32 + 8 * q + 4 * hi);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x1[st][4 * q + e] += fmaf(acc[st][4 * q + e], a.inv_2, b4[e]);
                }
            }
        }
        NB_STAMP(4);
        if (a.x_out) {
#pragma unroll
            for (int st = 0; st < 4; ++st) stage_strip(x1[st], st * 32);
            flush_rows(a.x_out, 128, 0, t0);
        }
        NB_STAMP(5);

        // ---- stage Q: next block's q|k|v (or the final norm) ---------------------------------------------------------------------
        if constexpr (KQC > 0) {
            h16x8 bqh[KQC], bql[KQC];
            {
                float ln[4][16];
                layer_norm(x1, NBP_GQ, NBP_BQN, a.epsq, ln);
                if (a.ln_out) {
#pragma unroll
                    for (int st = 0; st < 4; ++st) stage_strip(ln[st], st * 32);
                    flush_rows(a.ln_out, 128, 0, t0, a.ln_out_map);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) split8u(&ln[c >> 1][8 * (c & 1)], bqh[c], bql[c]);
            }
            if constexpr (KQC > 8) {
                const float *e = a.extra + (tc / a.extra_div) * a.extra_ld;
#pragma unroll
                for (int c = 8; c < KQC; ++c) {
                    const float4 v0 = ldg4(e + 16 * (c - 8) + 4 * hi), v1 = ldg4(e + 16 * (c - 8) + 8 + 4 * hi);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    split8u(v, bqh[c], bql[c]);
                }
            }
            NB_STAMP(6);
            if (a.q_out) {
                const int n_groups = a.NQ >> 7;                                // 128 output columns = 4 strips per group
#pragma unroll 1
                for (int g = 0; g < n_groups; ++g) {
                    nb_static_for<4>([&](auto ss) {
                        constexpr int sl = decltype(ss)::value;
                        f32x16 qh;
#pragma unroll
                        for (int r = 0; r < 16; ++r) qh[r] = 0.f;
                        nb_static_for<KQC>([&](auto cc) {
                            constexpr int c = decltype(cc)::value;
                            constexpr int pg = sl * KQC + c;                   // pair index within the group: 8 pairs per stage
                            if constexpr (pg % 8 == 0) stage_top();
                            consume(std::integral_constant<int, pg % 8>{}, bqh[c], bql[c], qh);
                            if constexpr (pg % 8 == 7) stage_end();
                        });
                        float ov[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 b4 = par4(NBP_BQ + g * 128 + sl * 32 + 8 * q + 4 * hi);
#pragma unroll
                            for (int e2 = 0; e2 < 4; ++e2) ov[4 * q + e2] = fmaf(qh[4 * q + e2], a.inv_q, b4[e2]);
                        }
                        stage_strip(ov, sl * 32);
                    });
                    flush_rows(a.q_out, a.NQ, g * 128, t0);
                    NB_STAMP(7 + g);
                }
            }
        }
    }
    NB_STAMP(15);
#ifdef NMRF_DEBUG_PROBES
    if (a.stamps && lane == 0 && blockIdx.x < 64) {
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 4 + wv) * 16;
        o[10] = tk_bar; o[11] = tk_use; o[12] = tk_commit;
    }
#endif
}

#ifdef NMRF_DEBUG_PROBES
// attainable v_mfma_f32_32x32x16_f16 rate: CHAINS independent accumulators per wave, iters x 24 MFMAs each, operands constant
template <int CHAINS>
__global__ __launch_bounds__(256) void mfma16_peak_kernel(int iters, float *__restrict__ out) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    h16x8 a, b;
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (threadIdx.x ^ k)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24 / CHAINS; ++k)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mfma16h(a, b, acc[c]);
    }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) sum += acc[c][0] + acc[c][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
extern "C" int nmrf_debug_mfma16_peak(int chains, int iters, int blocks, float *out, void *stream) {
    if (!out) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (chains == 1) hipLaunchKernelGGL((mfma16_peak_kernel<1>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 2) hipLaunchKernelGGL((mfma16_peak_kernel<2>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 3) hipLaunchKernelGGL((mfma16_peak_kernel<3>), dim3(blocks), dim3(256), 0, st, iters, out);
    else return NMRF_EINVAL;
    return nmrf_launch_status();
}

static int g_nb_variant = 0;      // pipeline-depth variants for tools/kernel_bench.py --which block
extern "C" int nmrf_debug_nmp_block_variant(int v) { g_nb_variant = v; return NMRF_OK; }
static unsigned long long *g_nb_stamps = nullptr;
extern "C" int nmrf_debug_nmp_block_timing(unsigned long long *stamps) { g_nb_stamps = stamps; return NMRF_OK; }
#else
static unsigned long long *const g_nb_stamps = nullptr;
#endif

template <bool MLP, int KQC, int FD, int PF, bool TOUCH, int DBG = 0>
static int launch_nmp_block_v(const NmpBlockArgs &a, hipStream_t st) {
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)NB_PAR_OFF + NBP_FLOATS * sizeof(float);
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(nmp_block_kernel<MLP, KQC, FD, PF, TOUCH, DBG>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMRF_ELAUNCH;
        n_cu_dev[dev] = prop.multiProcessorCount;
    }
    const int grid = a.n_tiles < n_cu_dev[dev] ? a.n_tiles : n_cu_dev[dev];
    hipLaunchKernelGGL((nmp_block_kernel<MLP, KQC, FD, PF, TOUCH, DBG>), dim3(grid), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

template <bool MLP, int KQC>
static int launch_nmp_block(const NmpBlockArgs &a, hipStream_t st) {
#ifdef NMRF_DEBUG_PROBES
    switch (g_nb_variant) {
        case 1: return launch_nmp_block_v<MLP, KQC, 1, 2, false>(a, st);
        case 2: return launch_nmp_block_v<MLP, KQC, 2, 2, true>(a, st);
        case 3: return launch_nmp_block_v<MLP, KQC, 2, 2, false>(a, st);
        case 4: return launch_nmp_block_v<MLP, KQC, 1, 4, true>(a, st);
        case 10: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 1>(a, st);
        case 11: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 2>(a, st);
        case 12: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 4>(a, st);
        case 13: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 8>(a, st);
        case 14: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 3>(a, st);
        case 15: return launch_nmp_block_v<MLP, KQC, 1, 2, true, 7>(a, st);
        default: break;
    }
#endif
    return launch_nmp_block_v<MLP, KQC, NB_FD, NB_PF, false>(a, st);
}

extern "C" int nmrf_nmp_block_f32(const float *x, const float *msg, const void *stream_w, int total_stages, const float *bp,
                                  const float *ln2_g, const float *ln2_b, float eps2, const float *b1, const float *b2,
                                  const float *lnq_g, const float *lnq_b, float epsq, const float *extra, int extra_ld,
                                  int extra_div, const float *bq, int has_mlp, int KQ, int NQ, int64_t T, const float *inv_scales,
                                  float *x_out, float *q_out, float *ln_out, const int *ln_out_map, int *range_flag, void *stream) {
    if (!x || !stream_w || !inv_scales) return NMRF_ENULL;
    if (T < 1 || ceil_div64(T, NB_TOK) > 0x7fffffff) return NMRF_EINVAL;
    if (has_mlp && (!ln2_g || !ln2_b || !b1 || !b2)) return NMRF_ENULL;
    if (KQ != 0 && KQ != 128 && KQ != 160 && KQ != 192) return NMRF_EINVAL;
    if (KQ && (!lnq_g || !lnq_b)) return NMRF_ENULL;
    if (KQ > 128 && (!extra || extra_ld < KQ - 128 || (extra_ld & 3) || extra_div < 1)) return NMRF_EINVAL;
    if (q_out && (KQ == 0 || NQ < 128 || (NQ & 127) || NQ > 512)) return NMRF_EINVAL;
    if (ln_out && KQ == 0) return NMRF_EINVAL;
    if (!q_out && !ln_out && !x_out) return NMRF_ENULL;
    // the stream must hold exactly the stages this configuration consumes
    const int want = (msg ? 4 : 0) + (has_mlp ? 32 : 0) + (q_out ? (NQ / 128) * (KQ / 16) / 2 : 0);
    if (total_stages != want || total_stages < 1) return NMRF_EINVAL;
    NmpBlockArgs a{x, msg, reinterpret_cast<const u32x4 *>(stream_w), total_stages, bp, ln2_g, ln2_b, b1, b2, lnq_g, lnq_b, extra,
                   extra_ld, extra_div, bq, x_out, q_out, ln_out, ln_out_map, T, (int)ceil_div64(T, NB_TOK), eps2, epsq, NQ, inv_scales[0], inv_scales[1], inv_scales[2], inv_scales[3], g_nb_stamps};
    hipStream_t st = (hipStream_t)stream;
    const int kqc = KQ / 16;
    if (has_mlp) {
        switch (kqc) {
            case 0: return launch_nmp_block<true, 0>(a, st);
            case 8: return launch_nmp_block<true, 8>(a, st);
            case 10: return launch_nmp_block<true, 10>(a, st);
            case 12: return launch_nmp_block<true, 12>(a, st);
        }
    } else {
        switch (kqc) {
            case 0: return launch_nmp_block<false, 0>(a, st);
            case 8: return launch_nmp_block<false, 8>(a, st);
            case 10: return launch_nmp_block<false, 10>(a, st);
            case 12: return launch_nmp_block<false, 12>(a, st);
        }
    }
    return NMRF_EINVAL;
}
