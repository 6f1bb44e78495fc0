// a DIFFERENT decomposition of the block kernel's MLP phase, as a stand-alone micro-kernel.
//
//   out[t] = x[t] + fc2(gelu(fc1(LayerNorm(x[t])) + b1)) + b2          (NMP.py:337, timm Mlp; = nmrf_nmp_block16_f32 with msg = NULL,
//                                                                        has_mlp = 1, KQ = 0: "the product's MLP-only launch")
//
// Product form (csrc/nmp_block16.hip): 8 waves x 16 tokens, two waves per SIMD, every weight fragment read from LDS by all 8 waves
// and used for ONE 16-token B tile.  This form ("b32"): 4 waves x 32 tokens, ONE wave per SIMD, every fragment read by 4 waves and
// used for TWO token tiles (the verdict's (b)) -- half the LDS fragment traffic per token, half the waves at every stage barrier --
// and, because no second wave covers a wave's serial phases any more, the GELU of a hidden group is placed INSIDE the neighbouring
// MFMA stages of the same wave (PIPE >= 1): G(hg) first half beside fc2(hg-1), second half beside fc1(hg+1).
// Same weight stream (nmrf_amd.kernels.block_stream16(w1=, w2=): 32 stages of 16 KB), same ring protocol, same per-accumulator
// MFMA order, same LayerNorm / GELU / split code => the SAME BITS as the product's MLP-only launch (checked by the driver script).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/_ab/mlp_b32.so tools/ab/mlp_b32.hip
#include "../../nmrf_amd/csrc/common.h"
#include "../../nmrf_amd/csrc/split_mfma.h"
#include <type_traits>
#include <utility>

typedef unsigned int mv_u32x4 __attribute__((ext_vector_type(4)));

#define MV_STAGE_U4 1024
#define MV_RING 3
#define MV_OLD 132
#define MV_PF 4
#define MV_TOK 128
#define MV_THR 256
#define MVP_G2 0
#define MVP_B2N 128
#define MVP_B1 256
#define MVP_B2 768
#define MVP_FLOATS 896
#define MV_OT_OFF (MV_RING * MV_STAGE_U4 * 16)
#define MV_PAR_OFF (MV_OT_OFF + 4 * 32 * MV_OLD * 4)
#define MV_LDS (MV_PAR_OFF + MVP_FLOATS * 4)

struct MlpvArgs {
    const float *x;
    const mv_u32x4 *stream;
    int total_stages;
    const float *g2, *b2n, *b1, *b2;
    float eps, inv1, inv2;
    float *out;
    int64_t T;
    int n_tiles;
    int *range_flag;
    int copies;                 // the stream exists `copies` times, copy_stride_u4 16-byte words apart; block b reads copy b % copies
    int64_t copy_stride_u4;     // (probe: do 256 CUs streaming the SAME 512 KB hot-spot a few L2 channels?)
};

__device__ __forceinline__ f32x4 mv_mfma(h16x8 a, h16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

template <class F, int... I>
__device__ __forceinline__ void mv_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void mv_static_for(F &&f) {
    mv_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

// PIPE 0: GELU of a group in front of its fc2 stage (the product's order; the compiler may move it)
// PIPE 1: GELU halves placed inside the neighbouring MFMA stages
// PIPE 2: ... and interleaved with them by sched_group_barrier (1 MFMA, VPM VALU) x 12 per sub-step
// DIST: how many stages a weight fetch is issued ahead of its commit to the ring (product: 1 -- the global load of stage g + 3 is issued
// at the end of stage g and its data is needed at the end of stage g + 1; DIST staging register sets of 16 VGPRs)
template <int PIPE, int DIST = 1>
__global__ __launch_bounds__(MV_THR, 1) void mlp_b32_kernel(MlpvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    mv_u32x4 *ring = reinterpret_cast<mv_u32x4 *>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    float *Ot = reinterpret_cast<float *>(smem + MV_OT_OFF) + wv * 32 * MV_OLD;         // wave-private [32][132]
    float *Par = reinterpret_cast<float *>(smem + MV_PAR_OFF);
    float guard = 0.f;
    for (int i = tid; i < 128; i += MV_THR) {
        Par[MVP_G2 + i] = a.g2[i];
        Par[MVP_B2N + i] = a.b2n[i];
        Par[MVP_B2 + i] = a.b2[i];
    }
    for (int i = tid; i < 512; i += MV_THR) Par[MVP_B1 + i] = a.b1[i];
    auto par4 = [&](int off) { return *reinterpret_cast<const f32x4 *>(Par + off); };

    // ---- weight ring (protocol of csrc/split_stream.h; 256 threads move 4 x 16 B per stage) ----------------------------------------
    mv_u32x4 R[DIST][4];
    int src_stage = 0, wr_slot = 0, rd_slot = 0;
    const mv_u32x4 *my_stream = a.stream + (size_t)(blockIdx.x % a.copies) * a.copy_stride_u4;
    auto fetch_into = [&](mv_u32x4 (&Rx)[4]) {
        const mv_u32x4 *p = my_stream + (size_t)src_stage * MV_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) Rx[i] = p[MV_THR * i];
        src_stage = (src_stage + 1 == a.total_stages) ? 0 : src_stage + 1;
    };
    auto commit_from = [&](const mv_u32x4 (&Rx)[4]) {
        mv_u32x4 *d = ring + wr_slot * MV_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[MV_THR * i] = Rx[i];
        wr_slot = (wr_slot == MV_RING - 1) ? 0 : wr_slot + 1;
    };
    const mv_u32x4 *cur = ring, *nxt = ring + MV_STAGE_U4;
    h16x8 fqh[MV_PF], fql[MV_PF];
    auto read_pair = [&](const mv_u32x4 *base, int p, h16x8 &h, h16x8 &l) {
        h = *reinterpret_cast<const h16x8 *>(base + p * 128 + lane);
        l = *reinterpret_cast<const h16x8 *>(base + p * 128 + 64 + lane);
    };
    bool have_barrier = true;
    auto stage_top = [&]() {
        if (!have_barrier) __syncthreads();
        have_barrier = false;
    };
    auto stage_end = [&]() {
        commit_from(R[0]);                                   // the oldest request (issued DIST stages ago)
#pragma unroll
        for (int d = 0; d + 1 < DIST; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) R[d][i] = R[d + 1][i];   // (register renaming after unrolling: no moves survive when DIST == 1)
        fetch_into(R[DIST - 1]);
        rd_slot = (rd_slot == MV_RING - 1) ? 0 : rd_slot + 1;
        cur = nxt;
        nxt = ring + ((rd_slot == MV_RING - 1) ? 0 : rd_slot + 1) * MV_STAGE_U4;
    };
    // pairs P0, P0 + 1 (two 16-row strips sharing a k chunk -- fc1 -- or two strips of one fc2 chunk) against the B operands of BOTH
    // token tiles: 12 MFMAs on four accumulators, each accumulator's own order (lo.hi, hi.lo, hi.hi) as in the product's consume2
    auto consume2x2 = [&](auto pc, const h16x8 &bh0, const h16x8 &bl0, const h16x8 &bh1, const h16x8 &bl1, f32x4 &a00, f32x4 &a01,
                          f32x4 &a10, f32x4 &a11) {
        constexpr int P0 = decltype(pc)::value, P1 = P0 + 1;
        static_assert(P0 % 2 == 0 && MV_PF % 2 == 0, "");
        const h16x8 ah0 = fqh[P0 % MV_PF], al0 = fql[P0 % MV_PF], ah1 = fqh[P1 % MV_PF], al1 = fql[P1 % MV_PF];
        if constexpr (P0 + MV_PF < 8) {
            read_pair(cur, P0 + MV_PF, fqh[P0 % MV_PF], fql[P0 % MV_PF]);
            read_pair(cur, P1 + MV_PF, fqh[P1 % MV_PF], fql[P1 % MV_PF]);
        } else {
            read_pair(nxt, P0 + MV_PF - 8, fqh[P0 % MV_PF], fql[P0 % MV_PF]);
            read_pair(nxt, P1 + MV_PF - 8, fqh[P1 % MV_PF], fql[P1 % MV_PF]);
        }
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);       // LDS reads stay in front, MFMAs behind; VALU / SALU / VMEM / TRANS may cross
        a00 = mv_mfma(al0, bh0, a00);
        a01 = mv_mfma(al0, bh1, a01);
        a10 = mv_mfma(al1, bh0, a10);
        a11 = mv_mfma(al1, bh1, a11);
        a00 = mv_mfma(ah0, bl0, a00);
        a01 = mv_mfma(ah0, bl1, a01);
        a10 = mv_mfma(ah1, bl0, a10);
        a11 = mv_mfma(ah1, bl1, a11);
        a00 = mv_mfma(ah0, bh0, a00);
        a01 = mv_mfma(ah0, bh1, a01);
        a10 = mv_mfma(ah1, bh0, a10);
        a11 = mv_mfma(ah1, bh1, a11);
    };
    auto interleave = [&]() {                       // one sub-step: 12 MFMAs, the VALU work placed beside them spread between
        if constexpr (PIPE == 2) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002 | 0x400, 4, 0);
            }
        }
    };
    {
        mv_u32x4 Ra[4], Rb[4];
        fetch_into(Ra); fetch_into(Rb);
#pragma unroll
        for (int d = 0; d < DIST; ++d) fetch_into(R[d]);
        commit_from(Ra); commit_from(Rb);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < MV_PF; ++p) read_pair(cur, p, fqh[p], fql[p]);

    auto group_sum = [&](float v) {
        v += __shfl_xor(v, 16);
        return half_sum(v);
    };
    auto layer_norm = [&](const float (&v)[32], float (&o)[32]) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += v[i];
        const float mean = group_sum(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) { const float d = v[i] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(group_sum(q) * (1.0f / 128.0f) + a.eps);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const f32x4 gv = par4(MVP_G2 + 16 * st + 4 * g), bv = par4(MVP_B2N + 16 * st + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * st + e] = (v[4 * st + e] - mean) * rstd * gv[e] + bv[e];
        }
    };

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int64_t t0 = (int64_t)tile * MV_TOK + wv * 32;
        h16x8 bnh[2][4], bnl[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t tq = t0 + 16 * t + j;
            const int64_t tc = tq < a.T ? tq : a.T - 1;
            float x1[32], ln[32];
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const float4 v = ldg4(a.x + tc * 128 + 16 * st + 4 * g);
                x1[4 * st] = v.x; x1[4 * st + 1] = v.y; x1[4 * st + 2] = v.z; x1[4 * st + 3] = v.w;
            }
            layer_norm(x1, ln);
#pragma unroll
            for (int c = 0; c < 4; ++c) split8u_g(&ln[8 * c], bnh[t][c], bnl[t][c], guard);
#pragma unroll
            for (int st = 0; st < 8; ++st)       // x1 waits in the wave's LDS tile while the hidden layer runs
                *reinterpret_cast<f32x4 *>(Ot + (16 * t + j) * MV_OLD + 16 * st + 4 * g) = f32x4{x1[4 * st], x1[4 * st + 1], x1[4 * st + 2], x1[4 * st + 3]};
        }
        f32x4 acc[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- the pieces -----------------------------------------------------------------------------------------------------------
        // GELU + split of token tile t of hidden group hg: f[strip][t] -> (hh[t], hl[t]); split over 4 calls (sub = 0..3: two values each)
        auto gelu_part = [&](int hg, const f32x4 (&f)[2][2], auto tc, auto subc, float (&hv)[8]) {
            const f32x4 ba = par4(MVP_B1 + 32 * hg + 4 * g), bb = par4(MVP_B1 + 32 * hg + 16 + 4 * g);
            constexpr int t = decltype(tc)::value, e = decltype(subc)::value;      // values (strip 0, reg e) and (strip 1, reg e)
            hv[e] = gelu_fast(fmaf(f[0][t][e], a.inv1, ba[e]));
            hv[4 + e] = gelu_fast(fmaf(f[1][t][e], a.inv1, bb[e]));
        };
        auto fc1_stage = [&](f32x4 (&f)[2][2], auto &&work) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) f[s][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            stage_top();
            mv_static_for<4>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                consume2x2(std::integral_constant<int, 2 * c>{}, bnh[0][c], bnl[0][c], bnh[1][c], bnl[1][c], f[0][0], f[0][1], f[1][0], f[1][1]);
                work(cc);
                interleave();
            });
            stage_end();
        };
        auto fc2_stage = [&](const h16x8 (&hh)[2], const h16x8 (&hl)[2], auto &&work) {
            stage_top();
            mv_static_for<4>([&](auto cc) {
                constexpr int p = 2 * decltype(cc)::value;
                consume2x2(std::integral_constant<int, p>{}, hh[0], hl[0], hh[1], hl[1], acc[p][0], acc[p][1], acc[p + 1][0], acc[p + 1][1]);
                work(cc);
                interleave();
            });
            stage_end();
        };
        auto nothing = [](auto) {};
        using T0 = std::integral_constant<int, 0>;
        using T1 = std::integral_constant<int, 1>;

        if constexpr (PIPE == 0) {
            // the product's order: fc1(next) ; GELU(cur) + split ; fc2(cur)
            f32x4 fa[2][2], fb[2][2];
            auto act_fc2 = [&](int hg, const f32x4 (&f)[2][2]) {
                h16x8 hh[2], hl[2];
                mv_static_for<2>([&](auto tc) {
                    float hv[8];
                    mv_static_for<4>([&](auto sc) { gelu_part(hg, f, tc, sc, hv); });
                    split8u_g(hv, hh[decltype(tc)::value], hl[decltype(tc)::value], guard);
                });
                fc2_stage(hh, hl, nothing);
            };
            fc1_stage(fa, nothing);
#pragma unroll 1
            for (int hg = 0; hg < 16; hg += 2) {
                fc1_stage(fb, nothing);
                act_fc2(hg, fa);
                if (hg + 2 < 16) fc1_stage(fa, nothing);
                act_fc2(hg + 1, fb);
            }
        } else {
            // stage list (stream order F0 F1 S0 F2 S1 ... F15 S14 S15):
            //   F0 | F1 + G(0) | then for hg = 1 .. 15:  S(hg-1) + G(hg) tile 0 | F(hg+1) + G(hg) tile 1   (hg = 15: no F16: G tile 1 serial) | S15
            f32x4 f0[2][2], f1[2][2];
            h16x8 h0h[2], h0l[2], h1h[2], h1l[2];
            float hva[8], hvb[8];
            fc1_stage(f0, nothing);                                               // F0
            fc1_stage(f1, [&](auto sc) {                                          // F1 + G(0) both tiles
                gelu_part(0, f0, T0{}, sc, hva);
                gelu_part(0, f0, T1{}, sc, hvb);
            });
            split8u_g(hva, h0h[0], h0l[0], guard);
            split8u_g(hvb, h0h[1], h0l[1], guard);
            // one step: hidden group hg (its fc1 output in fcur), previous group's operands in (ph, pl), this group's go to (ch, cl)
            auto step = [&](int hg, const f32x4 (&fcur)[2][2], f32x4 (&fnext)[2][2], const h16x8 (&ph)[2], const h16x8 (&pl)[2],
                            h16x8 (&ch)[2], h16x8 (&cl)[2]) {
                fc2_stage(ph, pl, [&](auto sc) { gelu_part(hg, fcur, T0{}, sc, hva); });      // S(hg-1) + G(hg) tile 0
                split8u_g(hva, ch[0], cl[0], guard);
                if (hg < 15) {
                    fc1_stage(fnext, [&](auto sc) { gelu_part(hg, fcur, T1{}, sc, hvb); });   // F(hg+1) + G(hg) tile 1
                } else {
                    mv_static_for<4>([&](auto sc) { gelu_part(hg, fcur, T1{}, sc, hvb); });
                }
                split8u_g(hvb, ch[1], cl[1], guard);
            };
#pragma unroll 1
            for (int hg = 1; hg < 15; hg += 2) {
                step(hg, f1, f0, h0h, h0l, h1h, h1l);
                step(hg + 1, f0, f1, h1h, h1l, h0h, h0l);
            }
            step(15, f1, f0, h0h, h0l, h1h, h1l);
            fc2_stage(h1h, h1l, nothing);                                         // S15
        }

        // ---- residual + bias, rows out ------------------------------------------------------------------------------------------------
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                float *p = Ot + (16 * t + j) * MV_OLD + 16 * st + 4 * g;
                const f32x4 xv = *reinterpret_cast<const f32x4 *>(p);
                const f32x4 b4 = par4(MVP_B2 + 16 * st + 4 * g);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = xv[e] + fmaf(acc[st][t][e], a.inv2, b4[e]);
                *reinterpret_cast<f32x4 *>(p) = o;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * i + (lane >> 5);
            const float4 v = *reinterpret_cast<const float4 *>(Ot + row * MV_OLD + 4 * (lane & 31));
            if (t0 + row < a.T) stg4(a.out + (size_t)(t0 + row) * 128 + 4 * (lane & 31), v);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    split_guard_commit(guard, a.range_flag);
}

template <int PIPE, int DIST = 1>
static int launch(const MlpvArgs &a, hipStream_t st) {
    static bool attr_set = false;
    static int n_cu = 0;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(mlp_b32_kernel<PIPE, DIST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return -1;
        attr_set = true;
    }
    if (!n_cu) {
        hipDeviceProp_t prop;
        int dev = 0;
        hipGetDevice(&dev);
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu = prop.multiProcessorCount;
    }
    const int grid = a.n_tiles < n_cu ? a.n_tiles : n_cu;
    hipLaunchKernelGGL((mlp_b32_kernel<PIPE, DIST>), dim3(grid), dim3(MV_THR), MV_LDS, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// stream_w: the 32 MLP stages of nmrf_amd.kernels.block_stream16(w1=W1, w2=W2); inv1 / inv2: its 1 / scale of fc1 / fc2
extern "C" int mlp_b32_f32(int pipe, const float *x, const void *stream_w, int total_stages, const float *ln_g, const float *ln_b, float eps,
                           const float *b1, const float *b2, float inv1, float inv2, int64_t T, float *out, int *range_flag, void *stream,
                           int copies, int64_t copy_stride_bytes) {
    if (!x || !stream_w || !ln_g || !ln_b || !b1 || !b2 || !out) return -3;
    if (total_stages != 32 || T < 1) return -4;
    MlpvArgs a{x, reinterpret_cast<const mv_u32x4 *>(stream_w), total_stages, ln_g, ln_b, b1, b2, eps, inv1, inv2, out, T,
               (int)((T + MV_TOK - 1) / MV_TOK), range_flag, copies < 1 ? 1 : copies, copy_stride_bytes / 16};
    if (copy_stride_bytes & 15) return -4;
    hipStream_t st = (hipStream_t)stream;
    switch (pipe) {
        case 0: return launch<0>(a, st);
        case 1: return launch<1>(a, st);
        case 2: return launch<2>(a, st);
        case 10: return launch<0, 2>(a, st);                 // pipe 0 with the fetch 2 / 3 / 4 stages ahead
        case 20: return launch<0, 3>(a, st);
        case 30: return launch<0, 4>(a, st);
        case 22: return launch<2, 3>(a, st);
    }
    return -4;
}
