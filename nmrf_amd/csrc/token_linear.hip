// N3: the token linears of the message-passing blocks with their prologue and epilogue fused.
//
//   out[t, :] = act( P(x)[t, :] . W^T + bias ) + residual[t, :]
//   P(x)      = [ LayerNorm(x[t] + y[t]) | extra[t / extra_div] | 0-pad ]      (q|k|v and fc1 operands, NMP.py:90-96,
//               or x[t] itself                                                    343-350, 544-556, 903-929)
//
// replaces  ln_concat kernel -> hipBLASLt GEMM -> (GELU kernel) -> (add kernel)  of the first version.  hipBLASLt runs
// these skinny shapes (T = 29 952 tokens, K = 128..192, N = 128..512) at 50-105 TFLOP/s (tools/gemm_probe.py) and the
// elementwise passes around them move 4-5x the tensor bytes; here the normalised operand never leaves the CU.
//
// Block = 4 waves x 32 tokens.  The operand tile P(x) [32][Kp] is built once in LDS (row stride Kp + 4 floats:
// conflict-free ds_read_b128 for the MFMA A fragments).  Each wave owns 32-column output strips (wv, wv+4, ... of the
// current column group) and walks their 32-deep k chunks: B fragments come straight from L2 in a host-packed fragment
// order (one contiguous 1 KB line per wave load) through a ring of register buffers filled a whole strip ahead, A
// fragments from LDS, 16 v_mfma_f32_32x32x2_f32 per chunk.
//   A operand  lane (i = l&31, h = l>>5), step s : P(x)[m0 + i][32c + 16h + s]
//   B operand  lane (j = l&31, h),        step s : W[32*strip + j][32c + 16h + s]
//   D          reg r of lane l                   : out[m0 + mfma_row(r, h)][32*strip + (l&31)]
// Results (bias and activation applied) are staged in an LDS tile [32][<=384 columns] and leave the CU as whole rows,
// float4 per lane, 1 KB per wave instruction.  Written straight from the accumulators (128-byte pieces with a 1.5 KB
// stride, every piece of a DRAM page arriving at a different time) the same tensor reached HBM at 1.5-3 TB/s and the
// store phases took as long as the MFMAs; a plain fill of that tensor runs at 5-7 TB/s (tools/hbm_probe.py).
#include "common.h"
#include "../../include/nmrf_hip_debug.h"      // tools / test build only (not in libnmrf_hip.so)
#include <type_traits>
#include <utility>
#include <stdlib.h>

#define TL_PAD 4

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}).  The register arrays of
// the main loop are indexed only through these constants: with ordinary unrolled loops the drained accumulator copy was
// left in scratch memory (a private array with "dynamic" indices), and every reload carried an s_waitcnt vmcnt(0).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}


// 16-lane all-reduce on the DPP network (no LDS crossbar): quad xor 1, quad xor 2, row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));
    return v;
}

struct TokenLinearArgs {
    const float *x, *y;
    float *x_out;
    const float *gamma, *beta;
    float eps;
    const float *extra;
    int E, extra_div;
    const float *w_packed, *bias, *residual;
    int act;
    int64_t T;
    int Cx, N;
    float *out;
    unsigned long long *stamps;     // debug (nmrf_debug_token_linear_timing): s_memtime per wave and phase, or NULL
};

// gw: columns per output group (N if N <= 384, else 256); alias_out: the staging tile reuses the operand tile (K = 512)
template <int KC, bool LN, bool GELU>
__global__ __launch_bounds__(256, 2) void token_linear_kernel(TokenLinearArgs a, int gw, int alias_out) {
    constexpr int MT = 32, KP = KC * 32, LDA = KP + TL_PAD;
    extern __shared__ __attribute__((aligned(16))) float At[];       // [32][LDA] operand tile, then [32][gw + 4] staging tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * MT;
#define TL_STAMP(k) do { if (a.stamps && lane == 0 && blockIdx.x < 64) \
        a.stamps[((size_t)blockIdx.x * 4 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    TL_STAMP(0);

    // ---- prologue: operand tile -> LDS ------------------------------------------------------------------
    if (LN) {
        // 16 lanes per token (two float4 each), 16 tokens per pass; two-pass variance like ATen
        const int sub = tid & 15, rr = tid >> 4;
        const float4 g0 = ldg4(a.gamma + 4 * sub), g1 = ldg4(a.gamma + 64 + 4 * sub);
        const float4 b0 = ldg4(a.beta + 4 * sub), b1 = ldg4(a.beta + 64 + 4 * sub);
        float4 v0[MT / 16], v1[MT / 16];
        constexpr int EX = (KP - 128) / 16;                         // extra columns per lane (2 for Fourier31, 4 for context64)
        float ex[MT / 16][EX > 0 ? EX : 1];
#pragma unroll
        for (int p = 0; p < MT / 16; ++p) {
            const int64_t t = m0 + p * 16 + rr;
            if (EX > 0) {                                            // requested together with x so that one latency covers all
                const float *e = (t < a.T && a.E > 0) ? a.extra + (t / a.extra_div) * a.E : nullptr;
#pragma unroll
                for (int j = 0; j < EX; ++j) ex[p][j] = (e && sub + 16 * j < a.E) ? e[sub + 16 * j] : 0.f;
            }
            if (t < a.T) {
                v0[p] = ldg4(a.x + t * 128 + 4 * sub);
                v1[p] = ldg4(a.x + t * 128 + 64 + 4 * sub);
                if (a.y) {
                    const float4 r0 = ldg4(a.y + t * 128 + 4 * sub), r1 = ldg4(a.y + t * 128 + 64 + 4 * sub);
                    v0[p] = make_float4(v0[p].x + r0.x, v0[p].y + r0.y, v0[p].z + r0.z, v0[p].w + r0.w);
                    v1[p] = make_float4(v1[p].x + r1.x, v1[p].y + r1.y, v1[p].z + r1.z, v1[p].w + r1.w);
                    stg4(a.x_out + t * 128 + 4 * sub, v0[p]);
                    stg4(a.x_out + t * 128 + 64 + 4 * sub, v1[p]);
                }
            } else {
                v0[p] = v1[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int p = 0; p < MT / 16; ++p) {
            const int row = p * 16 + rr;
            const int64_t t = m0 + row;
            const float s = ((v0[p].x + v0[p].y) + (v0[p].z + v0[p].w)) + ((v1[p].x + v1[p].y) + (v1[p].z + v1[p].w));
            const float mean = row16_sum(s) * (1.0f / 128.0f);
            const float4 d0 = make_float4(v0[p].x - mean, v0[p].y - mean, v0[p].z - mean, v0[p].w - mean);
            const float4 d1 = make_float4(v1[p].x - mean, v1[p].y - mean, v1[p].z - mean, v1[p].w - mean);
            const float q = ((d0.x * d0.x + d0.y * d0.y) + (d0.z * d0.z + d0.w * d0.w)) +
                            ((d1.x * d1.x + d1.y * d1.y) + (d1.z * d1.z + d1.w * d1.w));
            const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / 128.0f) + a.eps);
            float *ar = At + row * LDA;
            const bool ok = t < a.T;
            *reinterpret_cast<float4 *>(ar + 4 * sub) = ok ? make_float4(d0.x * rstd * g0.x + b0.x, d0.y * rstd * g0.y + b0.y,
                                                                         d0.z * rstd * g0.z + b0.z, d0.w * rstd * g0.w + b0.w)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(ar + 64 + 4 * sub) = ok ? make_float4(d1.x * rstd * g1.x + b1.x, d1.y * rstd * g1.y + b1.y,
                                                                              d1.z * rstd * g1.z + b1.z, d1.w * rstd * g1.w + b1.w)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < EX; ++j) ar[128 + sub + 16 * j] = ex[p][j];
        }
    } else {
        constexpr int V4 = KP / 4;                                   // float4 per operand row
        constexpr int PER = MT * V4 / 256;                           // float4 per thread (8 for K=128, 16 for K=512/MT=32)
        constexpr int STG = PER % 8 == 0 ? 8 : (PER % 5 == 0 ? 5 : (PER % 4 == 0 ? 4 : (PER % 2 == 0 ? 2 : 1)));   // loads in flight
        static_assert(MT * V4 % 256 == 0 && PER % STG == 0, "operand tile must split evenly over the block");
#pragma unroll
        for (int base = 0; base < PER; base += STG) {
            float4 v[STG];
#pragma unroll
            for (int u = 0; u < STG; ++u) {
                const int idx = tid + 256 * (base + u);
                const int row = idx / V4, c4 = idx - row * V4;
                const int64_t t = m0 + row;
                v[u] = (t < a.T && 4 * c4 < a.Cx) ? ldg4(a.x + t * a.Cx + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < STG; ++u) {
                const int idx = tid + 256 * (base + u);
                const int row = idx / V4, c4 = idx - row * V4;
                *reinterpret_cast<float4 *>(At + row * LDA + 4 * c4) = v[u];
            }
        }
    }
    TL_STAMP(1);
    __syncthreads();
    TL_STAMP(2);

    // ---- main loop: column groups -> strips of 32 output columns -> k chunks ---------------------------------------
    const int i32 = lane & 31, hi = lane >> 5;
    const int ldo = gw + TL_PAD;
    float *Ot = alias_out ? At : At + MT * LDA;
    const int n_groups = a.N / gw;                                 // the launcher guarantees gw | N
    const int gstrips = gw >> 5;
    const int cnt = (gstrips - wv + 3) >> 2;                       // strips of this wave per group: wv, wv+4, ...
    const int total = cnt * n_groups;
    auto strip_of = [&](int q) { const int g = q / cnt; return g * gstrips + wv + 4 * (q - g * cnt); };
    const float4 *wp = reinterpret_cast<const float4 *>(a.w_packed) + lane;
    // per-strip count of leading non-zero k chunks, appended to the packed weight by nmrf_pack_linear_weight_f32 (the v rows
    // of a fused q|k|v weight are zero on the side-input columns: 7-11 % of that GEMM's MFMAs)
    const int *strip_chunks = reinterpret_cast<const int *>(a.w_packed + (size_t)a.N * KC * 32);
    auto load_w = [&](int strip, int c, float *wd) {
        const float4 *p = wp + (size_t)(strip * KC + c) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = p[j * 64];
            wd[4 * j + 0] = v.x; wd[4 * j + 1] = v.y; wd[4 * j + 2] = v.z; wd[4 * j + 3] = v.w;
        }
    };
    const float *a_lane = At + i32 * LDA + 16 * hi;
    // B fragments: a ring of NB buffers refilled NB chunks ahead (buffer c % NB, NB | KC so the mapping survives the strip
    // boundary): an L2 round trip is longer than one chunk of 16 MFMAs.
    constexpr int NB = KC <= 5 ? KC : (KC == 6 ? 3 : 4);
    static_assert(KC % NB == 0, "ring size must divide the chunk count");
    float wbuf[NB][16];
    if (total > 0) {
        const int s0 = strip_of(0);
        static_for<NB>([&](auto cc) { load_w(s0, decltype(cc)::value, wbuf[decltype(cc)::value]); });
    }
    auto act_fn = [&](float v) {
        if (GELU) return gelu_fast(v);
        return a.act == 1 ? fmaxf(v, 0.f) : v;
    };
    int q = 0;
#pragma unroll 1
    for (int g = 0; g < n_groups; ++g) {
#pragma unroll 1
        for (int j = 0; j < cnt; ++j, ++q) {
            const int strip = g * gstrips + wv + 4 * j;
            const int next_strip = (q + 1 < total) ? strip_of(q + 1) : -1;
            const float bias_s = a.bias ? a.bias[strip * 32 + i32] : 0.f;   // requested now, needed after the strip
            const int kc_eff = strip_chunks[strip];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            static_for<KC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                float af[16];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 v = *reinterpret_cast<const float4 *>(a_lane + c * 32 + 4 * jj);
                    af[4 * jj + 0] = v.x; af[4 * jj + 1] = v.y; af[4 * jj + 2] = v.z; af[4 * jj + 3] = v.w;
                }
                if (c < kc_eff) {                                   // trailing all-zero chunks of this strip are skipped
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = mfma32(af[k], wbuf[c % NB][k], acc);
                }
                // this buffer's MFMAs are issued: refill it with the fragment NB chunks ahead (possibly of the next strip)
                if constexpr (c + NB < KC) load_w(strip, c + NB, wbuf[c % NB]);
                else if (next_strip >= 0) load_w(next_strip, c + NB - KC, wbuf[c % NB]);
                __builtin_amdgcn_sched_barrier(0);  // keeps the refill here and the next chunk's LDS reads below these MFMAs
            });
            if (alias_out) __syncthreads();                         // every wave is done reading the operand tile
            float *o = Ot + (4 * hi) * ldo + (strip - g * gstrips) * 32 + i32;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * ldo] = act_fn(acc[r] + bias_s);
        }
        if (g == 0) TL_STAMP(3);
        __syncthreads();
        if (g == 0) TL_STAMP(4);
        // whole rows of the group: 8 rows per wave, float4 per lane
        const int gw4 = gw >> 2;
#pragma unroll 1
        for (int rr = 0; rr < 8; ++rr) {
            const int row = wv * 8 + rr;
            const int64_t t = m0 + row;
            if (t < a.T) {
                for (int c4 = lane; c4 < gw4; c4 += 64) {
                    float4 v = *reinterpret_cast<const float4 *>(Ot + row * ldo + 4 * c4);
                    const size_t off = (size_t)t * a.N + (size_t)g * gw + 4 * c4;
                    if (a.residual) {
                        const float4 rv = ldg4(a.residual + off);
                        v = make_float4(v.x + rv.x, v.y + rv.y, v.z + rv.z, v.w + rv.w);
                    }
                    stg4(a.out + off, v);
                }
            }
        }
        if (g == 0) TL_STAMP(5);
        if (g + 1 < n_groups) __syncthreads();                      // the staging tile is reused by the next group
    }
    TL_STAMP(11);
}

// ------------------------------------------------------------------------------------------------------------------
// Persistent, software-pipelined form of the LayerNorm variants (the q|k|v and fc1 operands: 35 of the 62 linears of a
// forward).  A block walks tiles blockIdx.x, +gridDim.x, ...: while the MFMAs of tile i run, the rows of tile i+1 are
// already in flight into registers; they are normalised into the second operand buffer after the last column group,
// and the row stores of every 128-column group (staged through one of two LDS buffers) are issued right behind its
// barrier and retire under the next group's MFMAs.  In the one-tile-per-block kernel above every block runs
// load -> LayerNorm -> MFMA -> store back to back, in step with all its neighbours: 0.9 of the MFMA pipe inside the
// MFMA phase, 0.4 over the kernel.
// ------------------------------------------------------------------------------------------------------------------
template <int KC, bool GELU, int OBUF>
__global__ __launch_bounds__(256, 2) void token_linear_pipe_kernel(TokenLinearArgs a, int n_tiles) {
    constexpr int MT = 32, KP = KC * 32, LDA = KP + TL_PAD, GW = 128, LDO = GW + TL_PAD;
    constexpr int EX = (KP - 128) / 16;                             // extra columns per lane (2 for Fourier31, 4 for context64)
    extern __shared__ __attribute__((aligned(16))) float sm[];       // A[2][32][LDA] | Ot[OBUF][32][LDO]
    float *Abuf = sm, *Obuf = sm + 2 * MT * LDA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, hi = lane >> 5;
    const int sub = tid & 15, rr = tid >> 4;                        // LayerNorm: 16 lanes per token, 16 tokens per pass
    const float4 g0 = ldg4(a.gamma + 4 * sub), g1 = ldg4(a.gamma + 64 + 4 * sub);
    const float4 b0 = ldg4(a.beta + 4 * sub), b1 = ldg4(a.beta + 64 + 4 * sub);

    float4 v0[2], v1[2], w0[2], w1[2];                              // x and y rows of the tile being fetched
    float ex[2][EX > 0 ? EX : 1];
    auto fetch = [&](int tile) {                                    // issue the loads, do not wait
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int64_t t = (int64_t)tile * MT + p * 16 + rr;
            const bool ok = t < a.T;
            const int64_t tc = ok ? t : a.T - 1;                    // clamped: straight-line loads, masked at use
            if (EX > 0) {
                const float *e = a.extra + (tc / a.extra_div) * a.E;
#pragma unroll
                for (int j = 0; j < EX; ++j) ex[p][j] = (a.E > 0 && sub + 16 * j < a.E) ? e[sub + 16 * j] : 0.f;
            }
            v0[p] = ldg4(a.x + tc * 128 + 4 * sub);
            v1[p] = ldg4(a.x + tc * 128 + 64 + 4 * sub);
            if (a.y) {
                w0[p] = ldg4(a.y + tc * 128 + 4 * sub);
                w1[p] = ldg4(a.y + tc * 128 + 64 + 4 * sub);
            }
        }
    };
    auto normalise = [&](int tile, float *At) {                     // (x + y) -> x_out, LayerNorm | extra -> At
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = p * 16 + rr;
            const int64_t t = (int64_t)tile * MT + row;
            const bool ok = t < a.T;
            float4 s0 = v0[p], s1 = v1[p];
            if (a.y) {
                s0 = make_float4(s0.x + w0[p].x, s0.y + w0[p].y, s0.z + w0[p].z, s0.w + w0[p].w);
                s1 = make_float4(s1.x + w1[p].x, s1.y + w1[p].y, s1.z + w1[p].z, s1.w + w1[p].w);
                if (ok) {
                    stg4(a.x_out + t * 128 + 4 * sub, s0);
                    stg4(a.x_out + t * 128 + 64 + 4 * sub, s1);
                }
            }
            const float s = ((s0.x + s0.y) + (s0.z + s0.w)) + ((s1.x + s1.y) + (s1.z + s1.w));
            const float mean = row16_sum(s) * (1.0f / 128.0f);
            const float4 d0 = make_float4(s0.x - mean, s0.y - mean, s0.z - mean, s0.w - mean);
            const float4 d1 = make_float4(s1.x - mean, s1.y - mean, s1.z - mean, s1.w - mean);
            const float q = ((d0.x * d0.x + d0.y * d0.y) + (d0.z * d0.z + d0.w * d0.w)) +
                            ((d1.x * d1.x + d1.y * d1.y) + (d1.z * d1.z + d1.w * d1.w));
            const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / 128.0f) + a.eps);
            float *ar = At + row * LDA;
            *reinterpret_cast<float4 *>(ar + 4 * sub) = ok ? make_float4(d0.x * rstd * g0.x + b0.x, d0.y * rstd * g0.y + b0.y,
                                                                         d0.z * rstd * g0.z + b0.z, d0.w * rstd * g0.w + b0.w)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(ar + 64 + 4 * sub) = ok ? make_float4(d1.x * rstd * g1.x + b1.x, d1.y * rstd * g1.y + b1.y,
                                                                              d1.z * rstd * g1.z + b1.z, d1.w * rstd * g1.w + b1.w)
                                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < EX; ++j) ar[128 + sub + 16 * j] = ok ? ex[p][j] : 0.f;
        }
    };

    const int n_groups = a.N / GW;                                  // one 32-column strip per wave and group
    const float4 *wp = reinterpret_cast<const float4 *>(a.w_packed) + lane;
    // per-strip count of leading non-zero k chunks, appended to the packed weight by nmrf_pack_linear_weight_f32 (the v rows
    // of a fused q|k|v weight are zero on the side-input columns: 7-11 % of that GEMM's MFMAs)
    const int *strip_chunks = reinterpret_cast<const int *>(a.w_packed + (size_t)a.N * KC * 32);
    auto load_w = [&](int strip, int c, float *wd) {
        const float4 *p = wp + (size_t)(strip * KC + c) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = p[j * 64];
            wd[4 * j + 0] = v.x; wd[4 * j + 1] = v.y; wd[4 * j + 2] = v.z; wd[4 * j + 3] = v.w;
        }
    };
    constexpr int NB = KC <= 5 ? KC : (KC == 6 ? 3 : 4);
    static_assert(KC % NB == 0, "ring size must divide the chunk count");
    float wbuf[NB][16];
    auto act_fn = [&](float v) {
        if (GELU) return gelu_fast(v);
        return a.act == 1 ? fmaxf(v, 0.f) : v;
    };

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    fetch(tile);
    static_for<NB>([&](auto cc) { load_w(wv, decltype(cc)::value, wbuf[decltype(cc)::value]); });
    normalise(tile, Abuf);
    __syncthreads();
    int cur = 0, ob = 0;
#pragma unroll 1
    for (;;) {
        const int next = tile + gridDim.x;
        const bool has_next = next < n_tiles;
        if (has_next) fetch(next);
        const float *a_lane = Abuf + cur * MT * LDA + i32 * LDA + 16 * hi;
        const int64_t m0 = (int64_t)tile * MT;
#pragma unroll 1
        for (int g = 0; g < n_groups; ++g) {
            const int strip = g * 4 + wv;
            // the strip after this one: next group, or the first strip of the next tile (same weights)
            const int next_strip = (g + 1 < n_groups) ? strip + 4 : (has_next ? wv : -1);
            const float bias_s = a.bias ? a.bias[strip * 32 + i32] : 0.f;
            const int kc_eff = strip_chunks[strip];
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            static_for<KC>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                float af[16];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float4 v = *reinterpret_cast<const float4 *>(a_lane + c * 32 + 4 * jj);
                    af[4 * jj + 0] = v.x; af[4 * jj + 1] = v.y; af[4 * jj + 2] = v.z; af[4 * jj + 3] = v.w;
                }
                if (c < kc_eff) {                                   // trailing all-zero chunks of this strip are skipped
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc = mfma32(af[k], wbuf[c % NB][k], acc);
                }
                if constexpr (c + NB < KC) load_w(strip, c + NB, wbuf[c % NB]);
                else if (next_strip >= 0) load_w(next_strip, c + NB - KC, wbuf[c % NB]);
                __builtin_amdgcn_sched_barrier(0);
            });
            float *Ot = Obuf + ob * MT * LDO;
            float *o = Ot + (4 * hi) * LDO + wv * 32 + i32;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2)) * LDO] = act_fn(acc[r] + bias_s);
            __syncthreads();
            // 8 rows per wave, 512 B per row: lanes 0-31 one row, lanes 32-63 the next
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                const int row = wv * 8 + 2 * rp + hi;
                const int64_t t = m0 + row;
                if (t < a.T) stg4(a.out + (size_t)t * a.N + (size_t)g * GW + 4 * i32,
                                  *reinterpret_cast<const float4 *>(Ot + row * LDO + 4 * i32));
            }
            if (OBUF == 2) ob ^= 1;                                 // the barrier of the next group protects the reuse
            else __syncthreads();
        }
        if (!has_next) break;
        normalise(next, Abuf + (cur ^ 1) * MT * LDA);
        __syncthreads();
        cur ^= 1;
        tile = next;
    }
}

// weights [N, K] row-major -> fragment order [N/32][Kp/32][4][64 lanes][4], zero-padded in K, followed by N/32 ints: the
// number of leading k chunks of each strip that hold a non-zero weight (strip_chunks_kernel)
__global__ __launch_bounds__(256) void pack_linear_weight_kernel(const float *__restrict__ w, int N, int K, int KC,
                                                                float *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one float4 of the packed tensor
    const int64_t total = (int64_t)(N / 32) * KC * 4 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), j = (int)((idx >> 6) & 3);
    const int64_t sc = idx >> 8;
    const int c = (int)(sc % KC), s = (int)(sc / KC);
    const int n = s * 32 + (lane & 31), k0 = c * 32 + 16 * (lane >> 5) + 4 * j;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (k0 + e < K) ? w[(int64_t)n * K + k0 + e] : 0.f;
    stg4(packed + idx * 4, make_float4(v[0], v[1], v[2], v[3]));
}

// one wave per 32-column strip: number of leading k chunks after which every weight of the strip is exactly zero
__global__ __launch_bounds__(64) void strip_chunks_kernel(const float *__restrict__ w, int K, int KC, int *__restrict__ out) {
    const int strip = blockIdx.x, lane = threadIdx.x;
    int last = 0;                                                   // 1 + index of the last chunk with a non-zero weight
    for (int n = 0; n < 32; ++n)
        for (int k = lane; k < K; k += 64)
            if (w[(size_t)(strip * 32 + n) * K + k] != 0.f) last = max(last, k / 32 + 1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o));
    if (lane == 0) out[strip] = last < 1 ? 1 : last;
}

extern "C" int nmrf_pack_linear_weight_f32(const float *w, int N, int K, float *packed, void *stream) {
    if (!w || !packed) return NMRF_ENULL;
    if (N < 32 || (N & 31) || K < 1) return NMRF_EINVAL;
    const int KC = (K + 31) / 32;
    const int64_t total = (int64_t)(N / 32) * KC * 4 * 64;
    hipLaunchKernelGGL(pack_linear_weight_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       N, K, KC, packed);
    hipLaunchKernelGGL(strip_chunks_kernel, dim3(N / 32), dim3(64), 0, (hipStream_t)stream, w, K, KC,
                       reinterpret_cast<int *>(packed + (size_t)N * KC * 32));
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES
static unsigned long long *g_tl_stamps = nullptr;   // set by nmrf_debug_token_linear_timing for the next launches
extern "C" int nmrf_debug_token_linear_timing(unsigned long long *stamps) { g_tl_stamps = stamps; return NMRF_OK; }
static bool tl_nopipe() { return g_tl_stamps || getenv("NMRF_TL_NOPIPE"); }
#else
static unsigned long long *const g_tl_stamps = nullptr;
static bool tl_nopipe() { return false; }
#endif

template <int KC, bool LN, bool GELU>
static int launch_token_linear_g(const TokenLinearArgs &a, hipStream_t st) {
    const int gw = a.N <= 384 ? a.N : 256;                         // output columns staged per group
    if (a.N % gw || gw > 384) return NMRF_EINVAL;
    const int alias_out = (KC == 16);                               // K = 512: the operand tile alone is 66 KB
    if (alias_out && gw != 128) return NMRF_EINVAL;                 // (exactly one strip per wave, so the tile is dead by then)
    const size_t lds_a = (size_t)32 * (KC * 32 + TL_PAD) * sizeof(float), lds_o = (size_t)32 * (gw + TL_PAD) * sizeof(float);
    const size_t lds = alias_out ? (lds_a > lds_o ? lds_a : lds_o) : lds_a + lds_o;
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int cur_dev = nmrf_cur_device();
    if (cur_dev < 0) return NMRF_ELAUNCH;
    bool &attr_set = attr_set_dev[cur_dev];                                    // > 64 KB of dynamic LDS needs the opt-in once
    if (lds > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(token_linear_kernel<KC, LN, GELU>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set = true;
    }
    hipLaunchKernelGGL((token_linear_kernel<KC, LN, GELU>), dim3((unsigned)ceil_div64(a.T, 32)), dim3(256), lds, st, a, gw,
                       alias_out);
    return nmrf_launch_status();
}

template <int KC, bool LN>
static int launch_token_linear(const TokenLinearArgs &a, hipStream_t st) {
    return a.act == 2 ? launch_token_linear_g<KC, LN, true>(a, st) : launch_token_linear_g<KC, LN, false>(a, st);
}

template <int KC, bool GELU>
static int launch_token_linear_pipe_g(const TokenLinearArgs &a, hipStream_t st) {
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int cu_dev = nmrf_cur_device();
    if (cu_dev < 0) return NMRF_ELAUNCH;
    int &n_cu = n_cu_dev[cu_dev];
    if (!n_cu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cu_dev) != hipSuccess) return NMRF_ELAUNCH;
        n_cu = prop.multiProcessorCount;
    }
    constexpr size_t lds_a = (size_t)2 * 32 * (KC * 32 + TL_PAD) * sizeof(float), lds_o1 = (size_t)32 * (128 + TL_PAD) * sizeof(float);
    constexpr int OBUF = (lds_a + 2 * lds_o1 <= 80 * 1024) ? 2 : 1;          // two blocks per CU must fit in 160 KB
    constexpr size_t lds = lds_a + OBUF * lds_o1;
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int cur_dev = nmrf_cur_device();
    if (cur_dev < 0) return NMRF_ELAUNCH;
    bool &attr_set = attr_set_dev[cur_dev];
    if (lds > 65536 && !attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(token_linear_pipe_kernel<KC, GELU, OBUF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set = true;
    }
    const int n_tiles = (int)ceil_div64(a.T, 32);
    const int grid = n_tiles < 2 * n_cu ? n_tiles : 2 * n_cu;
    hipLaunchKernelGGL((token_linear_pipe_kernel<KC, GELU, OBUF>), dim3(grid), dim3(256), lds, st, a, n_tiles);
    return nmrf_launch_status();
}

template <int KC>
static int launch_token_linear_pipe(const TokenLinearArgs &a, hipStream_t st) {
    return a.act == 2 ? launch_token_linear_pipe_g<KC, true>(a, st) : launch_token_linear_pipe_g<KC, false>(a, st);
}

extern "C" int nmrf_token_linear_f32(const float *x, const float *y, float *x_out, const float *ln_gamma,
                                     const float *ln_beta, float eps, const float *extra, int E, int extra_div,
                                     const float *w_packed, const float *bias, const float *residual, int act, int64_t T,
                                     int Cx, int K, int N, float *out, void *stream) {
    if (!x || !w_packed || !out || (y && !x_out) || (E > 0 && !extra)) return NMRF_ENULL;
    const bool ln = ln_gamma != nullptr;
    if (ln && !ln_beta) return NMRF_ENULL;
    if (T < 1 || N < 32 || (N & 31) || act < 0 || act > 2 || E < 0 || extra_div < 1) return NMRF_EINVAL;
    if (ln ? (Cx != 128 || K != 128 + E) : (K != Cx || (Cx & 3) || E != 0 || y)) return NMRF_EINVAL;
    if (ceil_div64(T, 32) > 0x7fffffff) return NMRF_EINVAL;
    TokenLinearArgs a{x, y, x_out, ln_gamma, ln_beta, eps, extra, E, extra_div, w_packed, bias, residual, act, T, Cx, N, out, g_tl_stamps};
    hipStream_t st = (hipStream_t)stream;
    const int KC = (K + 31) / 32;
    if (ln && !residual && N % 128 == 0 && !tl_nopipe()) {
        switch (KC) {
            case 4: return launch_token_linear_pipe<4>(a, st);
            case 5: return launch_token_linear_pipe<5>(a, st);
            case 6: return launch_token_linear_pipe<6>(a, st);
            default: break;
        }
    }
    if (ln) {
        switch (KC) {
            case 4: return launch_token_linear<4, true>(a, st);
            case 5: return launch_token_linear<5, true>(a, st);
            case 6: return launch_token_linear<6, true>(a, st);
            default: return NMRF_EINVAL;
        }
    }
    switch (KC) {
        case 1: return launch_token_linear<1, false>(a, st);
        case 2: return launch_token_linear<2, false>(a, st);
        case 4: return launch_token_linear<4, false>(a, st);
        case 5: return launch_token_linear<5, false>(a, st);
        case 16: return launch_token_linear<16, false>(a, st);
        default: return NMRF_EINVAL;
    }
}
