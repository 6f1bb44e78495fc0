// N4, first slice: the pieces a backward pass of the token-linear chains needs -- dgrad / wgrad of a
// Linear as ONE strided GEMM on the split-operand fp16 MFMA (csrc/split_mfma.h: fp32-grade products, fp32 accumulate), the column sums
// of the bias gradient, the activation derivatives and the LayerNorm backward.  nmrf_amd/models/autograd_ops.py composes them into
// torch.autograd.Functions whose FORWARD is the product's fused launch (mlp_chain / nmp_block16) and whose backward recomputes the
// intermediates from the saved inputs (nmrf/models/NMRF.py:387-429 is what the reference differentiates; main.py:413-430 the step).
// Correctness first: operands are read straight from global memory in MFMA operand layout (no LDS staging); a K split with
// per-split partial products and a fixed-order sum keeps wgrad (K = all tokens) parallel AND deterministic.
#include "common.h"
#include "split_mfma.h"

// C[M,N] = op(A)[M,K] . op(B)[K,N]; element (i,k) of op(A) at A[i*sa_i + k*sa_k], element (k,j) of op(B) at B[k*sb_k + j*sb_j].
// One wave = one 32x32 tile of C, 16-deep k chunks, three v_mfma_f32_32x32x16_f16 per chunk (lo.hi, hi.lo, hi.hi into one fp32
// accumulator).  gridDim.y = K splits: split s covers chunks [s*cps, (s+1)*cps) and writes to C + s*split_stride.
struct GemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K;
    int64_t sa_i, sa_k, sb_k, sb_j;
    int ldc, tiles_n, n_tiles, cps;
    int64_t split_stride;
    int *range_flag;
    const float *a_amax;      // device: max |A| (or NULL): A is multiplied by a power of two that puts this at [2^13, 2^14) before its split
};

// AK1 / BK1: the operand's k axis is contiguous, K % 8 == 0, rows 16-byte aligned: a lane's 8-run is two dwordx4 loads (with scalar loads
// every lane of such an operand touches its own cache line 8 times per chunk and the address path, not the matrix pipe, sets the pace).
template <bool AK1, bool BK1>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv;
    if (tile >= a.n_tiles) return;                         // (no barrier in this kernel)
    const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
    const int i = tm * 32 + (lane & 31), j = tn * 32 + (lane & 31), kh = lane >> 5;
    const bool iok = i < a.M, jok = j < a.N;
    const float *pa = a.A + (int64_t)(iok ? i : 0) * a.sa_i;
    const float *pb = a.B + (int64_t)(jok ? j : 0) * a.sb_j;
    const int nchunks = (a.K + 15) / 16;
    const int c0 = blockIdx.y * a.cps, c1 = min(nchunks, c0 + a.cps);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float guard = 0.f;
    // The unscaled split (split_mfma.h) keeps 22 bits down to |v| = 2^-3 and an ABSOLUTE error of 2^-25 below: fine for O(1) activations,
    // not for a GRADIENT operand -- dy of a mean loss is ~1 / (B H W), 1e-6 ... 1e-7 at training sizes, where entries would lose most of
    // their bits or flush to zero.  With a_amax the A operand is rescaled by an exact power of two first (as packed weights
    // are) and the product is scaled back in the epilogue: errors relative to the tensor's largest entry, ~2^-38.
    float a_mul = 1.f, c_mul = 1.f;
    if (a.a_amax) {
        const float m = *a.a_amax;
        if (m > 0.f && m < INFINITY) {
            const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;         // floor(log2 m) (a subnormal maximum: e = -127, clamped below)
            const int sh = 13 - (e < -100 ? -100 : e);
            a_mul = ldexpf(1.0f, sh < 126 ? sh : 126);
            c_mul = ldexpf(1.0f, -(sh < 126 ? sh : 126));
        }
    }
    // register double buffer: the operands of chunk c + 1 are requested before chunk c is split and multiplied (a lone wave per tile
    // otherwise sits out a full memory round trip per 16-deep chunk)
    auto fetch8 = [&](auto vec_tag, const float *p, int64_t sk, bool ok, int k0, float *v) {
        if constexpr (decltype(vec_tag)::value) {
            float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
            if (ok && k0 < a.K) {
                lo = *reinterpret_cast<const float4 *>(p + k0);
                hi = *reinterpret_cast<const float4 *>(p + k0 + 4);
            }
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (ok && k0 + e < a.K) ? p[(int64_t)(k0 + e) * sk] : 0.f;
        }
    };
    auto fetch = [&](int c, float *av, float *bv) {
        const int k0 = c * 16 + 8 * kh;
        fetch8(std::integral_constant<bool, AK1>{}, pa, a.sa_k, iok, k0, av);
        fetch8(std::integral_constant<bool, BK1>{}, pb, a.sb_k, jok, k0, bv);
    };
    float av[8], bv[8], an[8], bn[8];
    if (c0 < c1) fetch(c0, av, bv);
    for (int c = c0; c < c1; ++c) {
        if (c + 1 < c1) fetch(c + 1, an, bn);
        h16x8 ah, al, bh, bl;
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] *= a_mul;
        split8u_g(av, ah, al, guard);
        split8u_g(bv, bh, bl, guard);
        split_mma1(ah, al, bh, bl, acc);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            av[e] = an[e];
            bv[e] = bn[e];
        }
    }
    float *cp = a.C + (int64_t)blockIdx.y * a.split_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 32 + mfma_row(r, kh);
        if (row < a.M && jok) cp[(int64_t)row * a.ldc + j] = acc[r] * c_mul;
    }
    split_guard_commit(guard, a.range_flag);
}

// Both operands contiguous along their OUTPUT index (sa_i == 1, sb_j == 1: the weight gradient dW = dy^T x, whose contraction runs over
// the tokens): a lane of the kernel above reads its 8-run of k with eight 4-byte loads per operand and chunk -- sixteen load
// instructions per three MFMAs (45.8 us per call, 95 calls per training step).  Here a wave reads each 16 x 32 operand tile as whole
// 128-byte rows (k fixed, 32 outputs: one 16-byte load per lane for 8 k-rows, two per chunk and operand), parks it in a wave-private
// LDS tile [16 k][32 + 1] and picks its MFMA 8-runs up from there.  Same products, same accumulation order: the same bits.
template <bool AIC, bool BIC>       // operand contiguous along its output index (through the LDS tile) -- otherwise along k (two 16-byte loads per 8-run)
__global__ __launch_bounds__(256) void gemm_split_ic_kernel(GemmArgs a) {
    static_assert(AIC || BIC, "both operands k-contiguous: gemm_split_kernel<true, true>");
    __shared__ float s_t[4][2][16][33];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wv;
    if (tile >= a.n_tiles) return;                         // (no block barrier in this kernel)
    const int tm = tile / a.tiles_n, tn = tile % a.tiles_n;
    const int kh = lane >> 5, jl = lane & 31;
    const int i = tm * 32 + jl, j = tn * 32 + jl;
    const bool iok = i < a.M, jok = j < a.N;
    const int nchunks = (a.K + 15) / 16;
    const int c0 = blockIdx.y * a.cps, c1 = min(nchunks, c0 + a.cps);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float guard = 0.f;
    float a_mul = 1.f, c_mul = 1.f;
    if (a.a_amax) {
        const float m = *a.a_amax;
        if (m > 0.f && m < INFINITY) {
            const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
            const int sh = 13 - (e < -100 ? -100 : e);
            a_mul = ldexpf(1.0f, sh < 126 ? sh : 126);
            c_mul = ldexpf(1.0f, -(sh < 126 ? sh : 126));
        }
    }
    // IC operand, loading role of a lane: k-row lane >> 3 (of 8) of a half chunk, outputs 4 (lane & 7) .. + 3 of the tile
    const int lk = lane >> 3, lo4 = 4 * (lane & 7);
    const bool a_in = tm * 32 + lo4 < a.M, b_in = tn * 32 + lo4 < a.N;       // (IC: M / N multiples of 4, checked by the launcher)
    const float *pa = AIC ? a.A + tm * 32 + lo4 : a.A + (int64_t)(iok ? i : 0) * a.sa_i;
    const float *pb = BIC ? a.B + tn * 32 + lo4 : a.B + (int64_t)(jok ? j : 0) * a.sb_j;
    // per chunk and operand: IC -> two float4 (k rows lk and 8 + lk); K1 -> the lane's own 8-run as two float4
    auto fetch1 = [&](auto ic, const float *p, int64_t sk, bool in_tile, bool ok, int c, float4 (&v)[2]) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (decltype(ic)::value) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = c * 16 + 8 * h + lk;
                v[h] = (in_tile && k < a.K) ? ldg4(p + (int64_t)k * sk) : z;
            }
        } else {
            const int k0 = c * 16 + 8 * kh;
            const bool on = ok && k0 < a.K;                 // (K % 8 == 0: checked by the launcher)
            v[0] = on ? ldg4(p + k0) : z;
            v[1] = on ? ldg4(p + k0 + 4) : z;
        }
    };
    auto fetch = [&](int c, float4 (&va)[2], float4 (&vb)[2]) {
        fetch1(std::integral_constant<bool, AIC>{}, pa, a.sa_k, a_in, iok, c, va);
        fetch1(std::integral_constant<bool, BIC>{}, pb, a.sb_k, b_in, jok, c, vb);
    };
    float4 va[2], vb[2], na[2], nb[2];
    if (c0 < c1) fetch(c0, va, vb);
    float (*ta)[33] = s_t[wv][0], (*tb)[33] = s_t[wv][1];
    for (int c = c0; c < c1; ++c) {
        if (c + 1 < c1) fetch(c + 1, na, nb);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (AIC) { float *ra = &ta[8 * h + lk][lo4]; ra[0] = va[h].x; ra[1] = va[h].y; ra[2] = va[h].z; ra[3] = va[h].w; }
            if constexpr (BIC) { float *rb = &tb[8 * h + lk][lo4]; rb[0] = vb[h].x; rb[1] = vb[h].y; rb[2] = vb[h].z; rb[3] = vb[h].w; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float av[8], bv[8];
        if constexpr (AIC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = ta[8 * kh + e][jl] * a_mul;
        } else {
            const float t[8] = {va[0].x, va[0].y, va[0].z, va[0].w, va[1].x, va[1].y, va[1].z, va[1].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = t[e] * a_mul;
        }
        if constexpr (BIC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bv[e] = tb[8 * kh + e][jl];
        } else {
            bv[0] = vb[0].x; bv[1] = vb[0].y; bv[2] = vb[0].z; bv[3] = vb[0].w; bv[4] = vb[1].x; bv[5] = vb[1].y; bv[6] = vb[1].z; bv[7] = vb[1].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        h16x8 ah, al, bh, bl;
        split8u_g(av, ah, al, guard);
        split8u_g(bv, bh, bl, guard);
        split_mma1(ah, al, bh, bl, acc);
#pragma unroll
        for (int h = 0; h < 2; ++h) { va[h] = na[h]; vb[h] = nb[h]; }
    }
    float *cp = a.C + (int64_t)blockIdx.y * a.split_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 32 + mfma_row(r, kh);
        if (row < a.M && jok) cp[(int64_t)row * a.ldc + j] = acc[r] * c_mul;
    }
    split_guard_commit(guard, a.range_flag);
}

extern "C" int nmrf_gemm_split_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, int M, int N,
                                   int K, float *C, int ldc, int splits, int64_t split_stride, const float *a_amax, int *range_flag,
                                   void *stream) {
    if (!A || !B || !C) return NMRF_ENULL;
    if (M < 1 || N < 1 || K < 1 || ldc < N || splits < 1 || splits > 65535 || (splits > 1 && split_stride < (int64_t)M * ldc)) return NMRF_EINVAL;
    const int tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32, nchunks = (K + 15) / 16;
    const int cps = (nchunks + splits - 1) / splits;
    GemmArgs a{A, B, C, M, N, K, sa_i, sa_k, sb_k, sb_j, ldc, tiles_n, tiles_m * tiles_n, cps, split_stride, range_flag, a_amax};
    const dim3 grid((unsigned)((a.n_tiles + 3) / 4), (unsigned)splits);
    hipStream_t st = (hipStream_t)stream;
    const bool k8 = K % 8 == 0;
    const bool ak1 = k8 && sa_k == 1 && sa_i % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const bool bk1 = k8 && sb_k == 1 && sb_j % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    const bool aic = sa_i == 1 && M % 4 == 0 && sa_k % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const bool bic = sb_j == 1 && N % 4 == 0 && sb_k % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
    if (aic && bic) hipLaunchKernelGGL((gemm_split_ic_kernel<true, true>), grid, dim3(256), 0, st, a);          // wgrad
    else if (ak1 && bic) hipLaunchKernelGGL((gemm_split_ic_kernel<false, true>), grid, dim3(256), 0, st, a);     // dgrad: dy rows . W
    else if (aic && bk1) hipLaunchKernelGGL((gemm_split_ic_kernel<true, false>), grid, dim3(256), 0, st, a);
    else if (ak1 && bk1) hipLaunchKernelGGL((gemm_split_kernel<true, true>), grid, dim3(256), 0, st, a);
    else if (ak1) hipLaunchKernelGGL((gemm_split_kernel<true, false>), grid, dim3(256), 0, st, a);
    else if (bk1) hipLaunchKernelGGL((gemm_split_kernel<false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gemm_split_kernel<false, false>), grid, dim3(256), 0, st, a);
    return nmrf_launch_status();
}

// out[g*n + i] = sum of parts[s*stride + i] over the `group` parts s of group g, in a fixed order (the deterministic second pass of every
// split reduction here; a caller with many parts reduces in rounds of `group`: a fixed tree, same bits every run)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ parts, int S, int64_t n, int64_t stride, int group,
                                                           float *__restrict__ out) {
    const int s0 = blockIdx.y * group, s1 = min(S, s0 + group);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        // four interleaved accumulators (four loads in flight), combined in a fixed order: deterministic, not the serial sum's bits
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int k = s0;
        for (; k + 3 < s1; k += 4) {
            a0 += parts[(int64_t)k * stride + i];
            a1 += parts[(int64_t)(k + 1) * stride + i];
            a2 += parts[(int64_t)(k + 2) * stride + i];
            a3 += parts[(int64_t)(k + 3) * stride + i];
        }
        for (; k < s1; ++k) a0 += parts[(int64_t)k * stride + i];
        out[(int64_t)blockIdx.y * n + i] = (a0 + a1) + (a2 + a3);
    }
}
extern "C" int nmrf_sum_partials_grouped_f32(const float *parts, int S, int64_t n, int64_t stride, int group, float *out, void *stream) {
    if (!parts || !out) return NMRF_ENULL;
    if (S < 1 || n < 1 || stride < n || group < 1) return NMRF_EINVAL;
    const int groups = (S + group - 1) / group;
    if (groups > 65535) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)blocks, (unsigned)groups), dim3(256), 0, (hipStream_t)stream, parts, S, n, stride, group, out);
    return nmrf_launch_status();
}
// The whole reduction of MANY parts of a NARROW row in one launch (round 6: the training step spent 504 launches / 4 ms per step in
// rounds of the grouped kernel above): a block owns 8 columns; its 32 part-lanes each add the parts p, p + 32, p + 64, ... in ascending
// order (four interleaved accumulators), then lane 0 adds the 32 lane sums in ascending order through LDS -- a fixed tree for a given S:
// the same bits every run.
__global__ __launch_bounds__(256) void sum_partials_tree_kernel(const float *__restrict__ parts, int S, int64_t n, int64_t stride,
                                                                float *__restrict__ out) {
    __shared__ float sh[32][9];
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int64_t col = (int64_t)blockIdx.x * 8 + c8;
    float s = 0.f;
    if (col < n) {
        // four independent accumulators: four loads in flight instead of a chain of dependent round trips (combined in a fixed order)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = pl;
        for (; k + 96 < S; k += 128) {
            s0 += parts[(int64_t)k * stride + col];
            s1 += parts[(int64_t)(k + 32) * stride + col];
            s2 += parts[(int64_t)(k + 64) * stride + col];
            s3 += parts[(int64_t)(k + 96) * stride + col];
        }
        for (; k < S; k += 32) s0 += parts[(int64_t)k * stride + col];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[pl][c8] = s;
    __syncthreads();
    if (pl == 0 && col < n) {
        float t = sh[0][c8];
#pragma unroll
        for (int p = 1; p < 32; ++p) t += sh[p][c8];
        out[col] = t;
    }
}
extern "C" int nmrf_sum_partials_tree_f32(const float *parts, int S, int64_t n, int64_t stride, float *out, void *stream) {
    if (!parts || !out) return NMRF_ENULL;
    if (S < 1 || n < 1 || stride < n || ceil_div64(n, 8) > 0x7fffffff) return NMRF_EINVAL;
    hipLaunchKernelGGL(sum_partials_tree_kernel, dim3((unsigned)ceil_div64(n, 8)), dim3(256), 0, (hipStream_t)stream, parts, S, n, stride, out);
    return nmrf_launch_status();
}
extern "C" int nmrf_sum_partials_f32(const float *parts, int S, int64_t n, int64_t stride, float *out, void *stream) {
    return nmrf_sum_partials_grouped_f32(parts, S, n, stride, S < 1 ? 1 : S, out, stream);
}

// kv16 rows (q fp32 | k | v as split fp16 pairs, include/nmrf_hip.h: what the block kernels write for the attention kernels) -> fp32 rows
// with k = hi + lo, v = hi + lo: the operand of the attention BACKWARD kernels, which take fp32 rows.  One launch instead of the ~15 torch
// view / shift / mask / stack / add passes per layer of the training-mode tape (round 6).  k third, per head: 16 words of hi halves then
// 16 words of lo halves, element 2 j in the low half of word j; v third: word c = hi | lo << 16 of channel c.
__global__ __launch_bounds__(256) void from_kv16_kernel(const float *__restrict__ in, int64_t T, float *__restrict__ out) {
    const int64_t total = T * 96;                                  // float4 groups per row: 32 q + 32 k + 32 v
    auto h2f = [](unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / 96;
        const int g = (int)(i - t * 96);
        const float *row = in + t * 384;
        float4 o;
        if (g < 32) {
            o = ldg4(row + 4 * g);
        } else if (g < 64) {
            const int hd = (g - 32) >> 3, e0 = ((g - 32) & 7) * 4;
            const unsigned *w = reinterpret_cast<const unsigned *>(row) + 128 + 32 * hd + (e0 >> 1);
            const unsigned h0 = w[0], h1 = w[1], l0 = w[16], l1 = w[17];
            o.x = h2f((unsigned short)(h0 & 0xffffu)) + h2f((unsigned short)(l0 & 0xffffu));
            o.y = h2f((unsigned short)(h0 >> 16)) + h2f((unsigned short)(l0 >> 16));
            o.z = h2f((unsigned short)(h1 & 0xffffu)) + h2f((unsigned short)(l1 & 0xffffu));
            o.w = h2f((unsigned short)(h1 >> 16)) + h2f((unsigned short)(l1 >> 16));
        } else {
            const unsigned *w = reinterpret_cast<const unsigned *>(row) + 256 + 4 * (g - 64);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = h2f((unsigned short)(w[e] & 0xffffu)) + h2f((unsigned short)(w[e] >> 16));
            o = make_float4(v[0], v[1], v[2], v[3]);
        }
        stg4(out + t * 384 + 4 * g, o);
    }
}
extern "C" int nmrf_from_kv16_f32(const float *qkv16, int64_t T, float *qkv, void *stream) {
    if (!qkv16 || !qkv) return NMRF_ENULL;
    if (T < 1 || ((reinterpret_cast<uintptr_t>(qkv16) | reinterpret_cast<uintptr_t>(qkv)) & 15)) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(T * 96, 256 * 2);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(from_kv16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv16, T, qkv);
    return nmrf_launch_status();
}

// max |x| into *out (which the caller has ZEROED): per-thread maxima over 16-byte loads, wave and block reduction, one atomicMax of the
// value's bit pattern per block (non-negative floats order like their bits; the maximum does not depend on the order of the atomics:
// deterministic).  NaNs are dropped by fmaxf -- the consumers' own range guard reports them.  (Round 6: torch's norm(inf) took 14.7 us
// per gradient tensor, 92 per training step.)
__global__ __launch_bounds__(256) void absmax_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ out) {
    __shared__ float sh[4];
    float m = 0.f;
    const int64_t n4 = n >> 2;
    const float4 *v = reinterpret_cast<const float4 *>(x);
    const int64_t step = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto m4 = [](float4 t) { return fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))); };
    for (; i + 3 * step < n4; i += 4 * step) {                     // four loads in flight
        const float4 t0 = v[i], t1 = v[i + step], t2 = v[i + 2 * step], t3 = v[i + 3 * step];
        m = fmaxf(m, fmaxf(fmaxf(m4(t0), m4(t1)), fmaxf(m4(t2), m4(t3))));
    }
    for (; i < n4; i += step) m = fmaxf(m, m4(v[i]));
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
        atomicMax(reinterpret_cast<unsigned *>(out), __float_as_uint(m));
    }
}
extern "C" int nmrf_absmax_f32(const float *x, int64_t n, float *out, void *stream) {
    if (!x || !out) return NMRF_ENULL;
    if (n < 1 || (reinterpret_cast<uintptr_t>(x) & 15)) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(n / 4 > 0 ? n / 4 : 1, 256 * 4);
    if (blocks > 512) blocks = 512;                                // (more blocks: their atomics on one address serialise -- 4096: 21 us)
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, out);
    return nmrf_launch_status();
}

// bias gradient: parts[b][n] = sum over the rows of block b of x[t][n]   (then nmrf_sum_partials_f32 over b)
// All 256 threads work whatever N is: a column is shared by RL = 256 / N' row lanes (N' = N rounded up to a power of two <= 256),
// lane rl adds the rows rl, rl + RL, ... of the block with four independent accumulators (four loads in flight, combined in a fixed
// order), and the RL lane sums are added in ascending order through LDS -- a fixed tree: the same bits every run.  (Round 6: the
// one-thread-per-column walk before it took 21.8 us per call for 8 MB, 92 calls per training step.)
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float *__restrict__ x, int64_t T, int N, int rows_per_block,
                                                              float *__restrict__ parts) {
    __shared__ float sh[256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < T ? r0 + rows_per_block : T;
    int np = 1;
    while (np < N && np < 256) np <<= 1;                          // columns per pass (power of two <= 256)
    const int RL = 256 / np, c0 = threadIdx.x & (np - 1), rl = threadIdx.x / np;
    for (int cb = 0; cb < N; cb += np) {                          // (N > 256: several passes)
        const int c = cb + c0;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (c < N) {
            int64_t r = r0 + rl;
            for (; r + 3 * RL < r1; r += 4 * RL) {
                s0 += x[r * N + c];
                s1 += x[(r + RL) * N + c];
                s2 += x[(r + 2 * RL) * N + c];
                s3 += x[(r + 3 * RL) * N + c];
            }
            for (; r < r1; r += RL) s0 += x[r * N + c];
        }
        sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (rl == 0 && c < N) {
            float t = sh[c0];
            for (int k = 1; k < RL; ++k) t += sh[k * np + c0];
            parts[(int64_t)blockIdx.x * N + c] = t;
        }
        __syncthreads();
    }
}
extern "C" int nmrf_colsum_partials_f32(const float *x, int64_t T, int N, int rows_per_block, float *parts, void *stream) {
    if (!x || !parts) return NMRF_ENULL;
    if (T < 1 || N < 1 || rows_per_block < 1 || ceil_div64(T, rows_per_block) > 0x7fffffff) return NMRF_EINVAL;
    hipLaunchKernelGGL(colsum_partials_kernel, dim3((unsigned)ceil_div64(T, rows_per_block)), dim3(256), 0, (hipStream_t)stream, x, T, N,
                       rows_per_block, parts);
    return nmrf_launch_status();
}

// pre_out = pre_in + bias (bias / pre_out may be NULL; pre_out may alias pre_in); act_out = act(pre_in + bias)  (act 0 identity,
// 1 ReLU, 2 GELU(erf) -- gelu_fast, the forward kernels' own)
__global__ __launch_bounds__(256) void bias_act_kernel(const float *__restrict__ pre_in, const float *__restrict__ bias, int64_t n, int N, int act,
                                                       float *pre_out, float *act_out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = pre_in[i] + (bias ? bias[i % N] : 0.f);
        if (pre_out) pre_out[i] = p;
        if (act_out) act_out[i] = act == 1 ? fmaxf(p, 0.f) : (act == 2 ? gelu_fast(p) : p);
    }
}
extern "C" int nmrf_bias_act_f32(const float *pre_in, const float *bias, int64_t T, int N, int act, float *pre_out, float *act_out, void *stream) {
    if (!pre_in || (!pre_out && !act_out)) return NMRF_ENULL;
    if (T < 1 || N < 1 || act < 0 || act > 2) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(T * N, 256 * 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bias_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pre_in, bias, T * N, N, act, pre_out, act_out);
    return nmrf_launch_status();
}

// dx = dy * act'(pre): ReLU: pre > 0; GELU(erf): Phi(pre) + pre * phi(pre)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float *__restrict__ pre, const float *__restrict__ dy, int64_t n, int act, float *__restrict__ dx) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float p = pre[i];
        float d;
        if (act == 1) d = p > 0.f ? 1.f : 0.f;
        else d = 0.5f * (1.0f + erff(p * 0.70710678118654752440f)) + p * 0.39894228040143267794f * expf(-0.5f * p * p);
        dx[i] = dy[i] * d;
    }
}
extern "C" int nmrf_act_bwd_f32(const float *pre, const float *dy, int64_t n, int act, float *dx, void *stream) {
    if (!pre || !dy || !dx) return NMRF_ENULL;
    if (n < 1 || (act != 1 && act != 2)) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(n, 256 * 4);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, pre, dy, n, act, dx);
    return nmrf_launch_status();
}

// LayerNorm over the last dimension C (C % 64 == 0, C <= 1024), one wave per row.
// fwd: y = (x - mean) * rstd * g + b.   bwd: xhat = (x - mean) * rstd, u = dy * g,
//      dx = rstd * (u - mean_c(u) - xhat * mean_c(u * xhat));  per-wave partial sums of dg = dy * xhat and db = dy over the wave's rows.
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ b, int64_t T,
                                                            float eps, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int C = CPL * 64;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < T; r += (int64_t)gridDim.x * 4) {
        float v[CPL], s = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { v[e] = x[r * C + lane + 64 * e]; s += v[e]; }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { const float d = v[e] - mean; q = fmaf(d, d, q); }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
#pragma unroll
        for (int e = 0; e < CPL; ++e) y[r * C + lane + 64 * e] = (v[e] - mean) * rstd * g[lane + 64 * e] + b[lane + 64 * e];
    }
}
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ dy,
                                                            int64_t T, float eps, float *__restrict__ dx, float *__restrict__ part_dg,
                                                            float *__restrict__ part_db) {
    const int lane = threadIdx.x & 63;
    const int C = CPL * 64;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float sdg[CPL], sdb[CPL], gv[CPL];
#pragma unroll
    for (int e = 0; e < CPL; ++e) { sdg[e] = sdb[e] = 0.f; gv[e] = g[lane + 64 * e]; }
    for (int64_t r = wave; r < T; r += (int64_t)gridDim.x * 4) {
        float v[CPL], d[CPL], s = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { v[e] = x[r * C + lane + 64 * e]; d[e] = dy[r * C + lane + 64 * e]; s += v[e]; }
        const float mean = wave_sum(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) { const float t = v[e] - mean; q = fmaf(t, t, q); }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
        float su = 0.f, sux = 0.f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) {
            v[e] = (v[e] - mean) * rstd;                       // xhat
            const float u = d[e] * gv[e];
            su += u;
            sux = fmaf(u, v[e], sux);
            sdg[e] = fmaf(d[e], v[e], sdg[e]);
            sdb[e] += d[e];
        }
        const float mu = wave_sum(su) * (1.0f / C), mux = wave_sum(sux) * (1.0f / C);
#pragma unroll
        for (int e = 0; e < CPL; ++e) dx[r * C + lane + 64 * e] = rstd * (d[e] * gv[e] - mu - v[e] * mux);
    }
#pragma unroll
    for (int e = 0; e < CPL; ++e) {
        part_dg[wave * C + lane + 64 * e] = sdg[e];
        part_db[wave * C + lane + 64 * e] = sdb[e];
    }
}

extern "C" int nmrf_layernorm_f32(const float *x, const float *g, const float *b, int64_t T, int C, float eps, float *y, void *stream) {
    if (!x || !g || !b || !y) return NMRF_ENULL;
    if (T < 1 || C < 64 || (C & 63) || C > 512) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(T, 4 * 4);
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: hipLaunchKernelGGL(layernorm_fwd_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, T, eps, y); break;
        case 2: hipLaunchKernelGGL(layernorm_fwd_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, T, eps, y); break;
        case 4: hipLaunchKernelGGL(layernorm_fwd_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, T, eps, y); break;
        case 8: hipLaunchKernelGGL(layernorm_fwd_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, x, g, b, T, eps, y); break;
        default: return NMRF_EINVAL;
    }
    return nmrf_launch_status();
}

// blocks workgroups of 4 waves; part_dg / part_db: [4 * blocks][C] (every wave writes its row, zeros if it had no token)
extern "C" int nmrf_layernorm_bwd_f32(const float *x, const float *g, const float *dy, int64_t T, int C, float eps, int blocks, float *dx,
                                      float *part_dg, float *part_db, void *stream) {
    if (!x || !g || !dy || !dx || !part_dg || !part_db) return NMRF_ENULL;
    if (T < 1 || C < 64 || (C & 63) || C > 512 || blocks < 1 || blocks > 65535) return NMRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    switch (C / 64) {
        case 1: hipLaunchKernelGGL(layernorm_bwd_kernel<1>, dim3(blocks), dim3(256), 0, st, x, g, dy, T, eps, dx, part_dg, part_db); break;
        case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, dim3(blocks), dim3(256), 0, st, x, g, dy, T, eps, dx, part_dg, part_db); break;
        case 4: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, dim3(blocks), dim3(256), 0, st, x, g, dy, T, eps, dx, part_dg, part_db); break;
        case 8: hipLaunchKernelGGL(layernorm_bwd_kernel<8>, dim3(blocks), dim3(256), 0, st, x, g, dy, T, eps, dx, part_dg, part_db); break;
        default: return NMRF_EINVAL;
    }
    return nmrf_launch_status();
}

// ---- (shifted-)window attention with relative-position q / k / v embeddings: backward -----------------------------------------------
// Forward (WindowAttention.forward, nmrf/models/NMP.py:185-289; restated in oracle/nmrf_oracle.py:window_attention): per window and head,
// tokens i, j of the window on the rolled grid, r = rel(pixel_i, pixel_j), (eq | ek | ev) = the head's 96 columns of table row r,
//     logit_ij = s (q_i . k_j + q_i . ek_r + k_j . eq_r)   (-inf: sibling labels of the query's pixel; other Swin region when shifted)
//     p = softmax_j(logit),   out_i = sum_j p_ij (v_j + ev_r)
// Backward, given dout: with dp_ij = dout_i . (v_j + ev_r), D_i = sum_j p_ij dp_ij, ds_ij = p_ij (dp_ij - D_i):
//     dq_i = s sum_j ds_ij (k_j + ek_r)      dk_j = s sum_i ds_ij (q_i + eq_r)      dv_j = sum_i p_ij dout_i
//     dek_r += s ds_ij q_i                   deq_r += s ds_ij k_j                    dev_r += p_ij dout_i
// One workgroup per (window, head, image); P and ds of the window live in a global scratch (2 x Tw^2 floats per workgroup), the table
// gradient leaves as one part per (image, window) -- summed in fixed order by nmrf_sum_partials_f32: deterministic.  fp32 VALU arithmetic
// (5 x Tw^2 x 32 MACs per window and head) spread over the whole workgroup: one thread per (query, key) pair for the logits and dp, one wave
// per row for the softmax statistics, one thread per (token, channel) for dq | dk | dv and per (table row, channel) for the table.
struct WinBwdArgs {
    const float *qkv, *table, *dout;
    float *dqkv, *dtab_parts, *scratch;
    int Hp, Wp, N, C, heads, win, shift, sibling;
    float scale;
};

#define WB_LD 33          // LDS row stride of the [Tw][32] tiles
#define WB_NMAX 4         // labels per pixel (register-blocked: the kernel is instantiated for N = 1 .. 4)
#define WB_NT 512         // threads per workgroup: two waves per SIMD (the LDS image allows one workgroup per CU)

// LDS: Q | K | dO [Tw][33], Eq | Ek [R][33], four int arrays [Tw], then a POOL that is, in turn,
//   phase 0-1a: V [Tw][33] | Ev [R][33]                      phase 1b-3: A | Ap | Bq [Tw][W2 + 1] + one (ds, p) row buffer per wave
//   phase 2:    a [16 N][Tw] row tile of ds, then [Tw][33] column tiles of ds and p
__host__ __device__ inline size_t wb_pool_floats(int Tw, int W2, int R, int N) {
    size_t a = (size_t)(Tw + R) * WB_LD, b = (size_t)Tw * (3 * W2 + 1) + (WB_NT / 64) * 2 * (size_t)Tw, c = (size_t)16 * N * Tw, d = (size_t)2 * Tw * WB_LD;
    size_t m = a > b ? a : b;
    m = m > c ? m : c;
    return m > d ? m : d;
}

template <int N>
__global__ __launch_bounds__(WB_NT) void window_attn_bwd_kernel(WinBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wb_smem[];
    const int win = a.win, W2 = win * win, Tw = W2 * N, span = 2 * win - 1, R = span * span, LA = W2, LB = W2 + 1;
    float *Q = wb_smem, *Kt = Q + Tw * WB_LD, *dO = Kt + Tw * WB_LD;
    float *Eq = dO + Tw * WB_LD, *Ek = Eq + R * WB_LD;
    int *rowoff = reinterpret_cast<int *>(Ek + R * WB_LD);                    // [Tw] token index on the (un-rolled) grid
    int *reg = rowoff + Tw;                                                    // [Tw] Swin region of the token's pixel
    int *lin = reg + Tw;                                                       // [Tw] row * span + column of the token's pixel in the window
    float *pool = reinterpret_cast<float *>(lin + Tw);
    float *V = pool, *Ev = V + Tw * WB_LD;
    float *A = pool, *Ap = A + Tw * LA, *Bq = Ap + Tw * LA, *rowbuf = Bq + Tw * LB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwx = a.Wp / win, nwin = nwx * (a.Hp / win);
    const int w = blockIdx.x, head = blockIdx.y, bimg = blockIdx.z;
    const int wi = w / nwx, wj = w % nwx;
    const int ld = 3 * a.C;
    const float s = a.scale;
    const int roff = (win - 1) * (span + 1);                                   // rel(i, j) = lin[i] - lin[j] + roff
    for (int i = tid; i < Tw; i += WB_NT) {
        const int pt = i / N, n = i - pt * N, pa = pt / win, pb = pt - pa * win;
        const int Yr = wi * win + pa, Xr = wj * win + pb;                      // rolled grid
        int Y = Yr + a.shift, X = Xr + a.shift;
        Y = Y >= a.Hp ? Y - a.Hp : Y;
        X = X >= a.Wp ? X - a.Wp : X;
        rowoff[i] = ((bimg * a.Hp + Y) * a.Wp + X) * N + n;
        const int fy = Yr < a.Hp - win ? 0 : (Yr < a.Hp - a.shift ? 1 : 2), fx = Xr < a.Wp - win ? 0 : (Xr < a.Wp - a.shift ? 1 : 2);
        reg[i] = a.shift ? fy * 3 + fx : 0;
        lin[i] = pa * span + pb;
    }
    __syncthreads();
    for (int e = tid; e < Tw * 32; e += WB_NT) {
        const int i = e >> 5, c = e & 31;
        const float *row = a.qkv + (size_t)rowoff[i] * ld + head * 32 + c;
        Q[i * WB_LD + c] = row[0];
        Kt[i * WB_LD + c] = row[a.C];
        V[i * WB_LD + c] = row[2 * a.C];
        dO[i * WB_LD + c] = a.dout[(size_t)rowoff[i] * a.C + head * 32 + c];
    }
    for (int e = tid; e < R * 32; e += WB_NT) {
        const int r = e >> 5, c = e & 31;
        const float *row = a.table + (size_t)r * ld + head * 96 + c;
        Eq[r * WB_LD + c] = row[0];
        Ek[r * WB_LD + c] = row[32];
        Ev[r * WB_LD + c] = row[64];
    }
    __syncthreads();
    float *P = a.scratch + ((size_t)(bimg * a.heads + head) * nwin + w) * 2 * Tw * Tw, *dS = P + (size_t)Tw * Tw;
#ifdef WB_STOP                                                   // (tools/build_ab_winbwd.sh: where does the time go)
#define WB_PHASE_END(k) if (WB_STOP <= (k)) return;
#else
#define WB_PHASE_END(k)
#endif
    WB_PHASE_END(0)
    // ---- phase 1a: one thread per (query PIXEL, key): the N logits and dp = dout_i . (v_j + ev_r) of the pixel's labels share k, v and the
    // three table rows (lanes = consecutive keys: conflict-free rows, coalesced stores) ---------------------------------------------------
    for (int e = tid; e < W2 * Tw; e += WB_NT) {
        const int pi = e / Tw, j = e - pi * Tw, i0 = pi * N;
        const bool region = reg[i0] != reg[j];
        const bool sib = a.sibling && N > 1 && j / N == pi;
        const int r = lin[i0] - lin[j] + roff;
        const float *k = Kt + j * WB_LD, *v = V + j * WB_LD, *ek = Ek + r * WB_LD, *eq = Eq + r * WB_LD, *ev = Ev + r * WB_LD;
        float acc[N], dp[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = dp[n] = 0.f;
        if (!region) {
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                const float kc = k[c], ke = kc + ek[c], ve = v[c] + ev[c], keq = kc * eq[c];
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    acc[n] = fmaf(Q[(i0 + n) * WB_LD + c], ke, acc[n] + keq);
                    dp[n] = fmaf(dO[(i0 + n) * WB_LD + c], ve, dp[n]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const bool off = region || (sib && i0 + n != j);
            P[(size_t)(i0 + n) * Tw + j] = off ? -INFINITY : acc[n] * s;
            dS[(size_t)(i0 + n) * Tw + j] = off ? 0.f : dp[n];
        }
    }
    __threadfence_block();
    __syncthreads();                                             // (V and Ev are dead from here: the pool holds A | Ap | Bq)
    WB_PHASE_END(1)
    // ---- phase 1b: one wave per query pixel, its N rows in turn: softmax, D = sum_j p dp, ds = p (dp - D) (fixed butterfly order:
    // deterministic); and the label sums the table gradient needs: A[i][pj] = sum_n2 ds[i][(pj, n2)], Ap likewise of p,
    // Bq[j][pi] = sum_n ds[(pi, n)][j] ---------------------------------------------------------------------------------------------------
    {
        float *rb = rowbuf + wave * 2 * Tw;
        for (int pi = wave; pi < W2; pi += WB_NT / 64) {
            float bq[4] = {0.f, 0.f, 0.f, 0.f};
            for (int n = 0; n < N; ++n) {
                const int i = pi * N + n;
                float lv[4], dv4[4];
                float m = -INFINITY;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = lane + 64 * u;
                    lv[u] = j < Tw ? P[(size_t)i * Tw + j] : -INFINITY;
                    dv4[u] = j < Tw ? dS[(size_t)i * Tw + j] : 0.f;
                    m = fmaxf(m, lv[u]);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                float Z = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    lv[u] = lv[u] == -INFINITY ? 0.f : expf(lv[u] - m);
                    Z += lv[u];
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) Z += __shfl_xor(Z, o, 64);
                const float rz = 1.0f / Z;
                float D = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    lv[u] *= rz;
                    D = fmaf(lv[u], dv4[u], D);
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) D += __shfl_xor(D, o, 64);
                __builtin_amdgcn_wave_barrier();                 // (the previous row's label sums have read the row buffer)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = lane + 64 * u;
                    if (j < Tw) {
                        const float ds = lv[u] * (dv4[u] - D);
                        P[(size_t)i * Tw + j] = lv[u];
                        dS[(size_t)i * Tw + j] = ds;
                        rb[j] = ds;
                        rb[Tw + j] = lv[u];
                        bq[u] += ds;
                    }
                }
                __builtin_amdgcn_wave_barrier();                 // (LDS operations of one wave complete in order)
                for (int pj = lane; pj < W2; pj += 64) {
                    float sa = 0.f, sp = 0.f;
                    for (int n2 = 0; n2 < N; ++n2) {
                        sa += rb[pj * N + n2];
                        sp += rb[Tw + pj * N + n2];
                    }
                    A[i * LA + pj] = sa;
                    Ap[i * LA + pj] = sp;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = lane + 64 * u;
                if (j < Tw) Bq[j * LB + pi] = bq[u];
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    WB_PHASE_END(2)
    // ---- phase 3: one thread per (table row, channel): every (query pixel, key pixel) pair with that offset ------------------------------
    //     dek_r = s sum_i q_i A[i][pj]      deq_r = s sum_j k_j Bq[j][pi]      dev_r = sum_i dout_i Ap[i][pj]
    float *part = a.dtab_parts + ((size_t)bimg * nwin + w) * R * ld + head * 96;
    for (int e = tid; e < R * 32; e += WB_NT) {
        const int r = e >> 5, c = e & 31;
        const int da = r / span - (win - 1), db = r % span - (win - 1);        // query pixel - key pixel
        float geq = 0.f, gek = 0.f, gev = 0.f;
        for (int pa = 0; pa < win; ++pa) {
            const int ka = pa - da;
            if (ka < 0 || ka >= win) continue;
            for (int pb = 0; pb < win; ++pb) {
                const int kb = pb - db;
                if (kb < 0 || kb >= win) continue;
                const int pi = pa * win + pb, pj = ka * win + kb;
                for (int n = 0; n < N; ++n) {
                    const int i = pi * N + n, j = pj * N + n;
                    gek = fmaf(A[i * LA + pj], Q[i * WB_LD + c], gek);
                    gev = fmaf(Ap[i * LA + pj], dO[i * WB_LD + c], gev);
                    geq = fmaf(Bq[j * LB + pi], Kt[j * WB_LD + c], geq);
                }
            }
        }
        part[(size_t)r * ld + c] = geq * s;
        part[(size_t)r * ld + 32 + c] = gek * s;
        part[(size_t)r * ld + 64 + c] = gev;
    }
    WB_PHASE_END(3)
    // ---- phase 2: tiles of ds / p through the pool; one thread per (pixel of the tile, channel), its N labels in registers ---------------
    const int pp = tid >> 5, c = tid & 31;
    for (int pb0 = 0; pb0 < W2; pb0 += 16) {                      // dq: 16 query pixels (16 N rows of ds, contiguous in the scratch) at a time
        const int npx = W2 - pb0 < 16 ? W2 - pb0 : 16;
        __syncthreads();
        for (int e = tid; e < npx * N * Tw; e += WB_NT) pool[e] = dS[(size_t)pb0 * N * Tw + e];
        __syncthreads();
        if (pp < npx) {
            const int i0 = (pb0 + pp) * N, li = lin[i0] + roff;
            const float *t = pool + pp * N * Tw;
            float acc[N];
#pragma unroll
            for (int n = 0; n < N; ++n) acc[n] = 0.f;
#pragma unroll 4
            for (int j = 0; j < Tw; ++j) {                        // (a masked pair has ds = p = 0: it adds exactly 0)
                const float ke = Kt[j * WB_LD + c] + Ek[(li - lin[j]) * WB_LD + c];
#pragma unroll
                for (int n = 0; n < N; ++n) acc[n] = fmaf(t[n * Tw + j], ke, acc[n]);
            }
#pragma unroll
            for (int n = 0; n < N; ++n) a.dqkv[(size_t)rowoff[i0 + n] * ld + head * 32 + c] = acc[n] * s;
        }
    }
    WB_PHASE_END(4)
    float *tC = pool, *tP = pool + Tw * WB_LD;
    for (int pb0 = 0; pb0 < W2; pb0 += 8) {                       // dk | dv: 8 key pixels (8 N columns of ds and p) at a time
        const int npx = W2 - pb0 < 8 ? W2 - pb0 : 8, ncol = npx * N, j0 = pb0 * N;
        __syncthreads();
        for (int e = tid; e < Tw * 32; e += WB_NT) {
            const int i = e >> 5, jj = e & 31;
            tC[i * WB_LD + jj] = jj < ncol ? dS[(size_t)i * Tw + j0 + jj] : 0.f;
            tP[i * WB_LD + jj] = jj < ncol ? P[(size_t)i * Tw + j0 + jj] : 0.f;
        }
        __syncthreads();
        const int pq = pp & 7;                                    // (waves 0-3: dk of the 8 key pixels, waves 4-7: their dv)
        if (pq < npx) {
            const int jb = (pb0 + pq) * N, lj = roff - lin[jb];
            float g[N];
#pragma unroll
            for (int n = 0; n < N; ++n) g[n] = 0.f;
            if (pp < 8) {
#pragma unroll 4
                for (int i = 0; i < Tw; ++i) {
                    const float qe = Q[i * WB_LD + c] + Eq[(lin[i] + lj) * WB_LD + c];
#pragma unroll
                    for (int n = 0; n < N; ++n) g[n] = fmaf(tC[i * WB_LD + pq * N + n], qe, g[n]);
                }
#pragma unroll
                for (int n = 0; n < N; ++n) a.dqkv[(size_t)rowoff[jb + n] * ld + head * 32 + c + a.C] = g[n] * s;
            } else {
#pragma unroll 4
                for (int i = 0; i < Tw; ++i) {
                    const float go = dO[i * WB_LD + c];
#pragma unroll
                    for (int n = 0; n < N; ++n) g[n] = fmaf(tP[i * WB_LD + pq * N + n], go, g[n]);
                }
#pragma unroll
                for (int n = 0; n < N; ++n) a.dqkv[(size_t)rowoff[jb + n] * ld + head * 32 + c + 2 * a.C] = g[n];
            }
        }
    }
}

// qkv [B,Hp,Wp,N,3C] fp32 rows, table [(2 win - 1)^2, 3C], dout [B,Hp,Wp,N,C] -> dqkv (every element written), dtab_parts
// [B * windows][(2 win - 1)^2][3C] (sum them with nmrf_sum_partials_f32), scratch: 2 * B * heads * windows * Tw^2 floats.  N <= 4.
extern "C" int nmrf_window_attn_bwd_f32(const float *qkv, const float *table, const float *dout, int B, int Hp, int Wp, int N, int C, int heads,
                                        int win, int shift, int sibling_mask, float *dqkv, float *dtab_parts, float *scratch, void *stream) {
    if (!qkv || !table || !dout || !dqkv || !dtab_parts || !scratch) return NMRF_ENULL;
    if (B < 1 || N < 1 || N > WB_NMAX || win < 1 || Hp % win || Wp % win || shift < 0 || shift >= win || heads * 32 != C) return NMRF_EINVAL;
    const int W2 = win * win, Tw = W2 * N, R = (2 * win - 1) * (2 * win - 1);
    if (Tw > 256) return NMRF_EINVAL;
    const size_t lds = ((size_t)(3 * Tw + 2 * R) * WB_LD + 3 * Tw + wb_pool_floats(Tw, W2, R, N)) * sizeof(float);
    if (lds > 160 * 1024) return NMRF_EINVAL;
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    if (!attr_set_dev[dev]) {
        const void *fns[4] = {reinterpret_cast<const void *>(window_attn_bwd_kernel<1>), reinterpret_cast<const void *>(window_attn_bwd_kernel<2>),
                              reinterpret_cast<const void *>(window_attn_bwd_kernel<3>), reinterpret_cast<const void *>(window_attn_bwd_kernel<4>)};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    WinBwdArgs a{qkv, table, dout, dqkv, dtab_parts, scratch, Hp, Wp, N, C, heads, win, shift, sibling_mask ? 1 : 0, 1.0f / sqrtf(32.0f)};
    const dim3 grid((Hp / win) * (Wp / win), heads, B);
    hipStream_t st = (hipStream_t)stream;
    switch (N) {
        case 1: hipLaunchKernelGGL(window_attn_bwd_kernel<1>, grid, dim3(WB_NT), lds, st, a); break;
        case 2: hipLaunchKernelGGL(window_attn_bwd_kernel<2>, grid, dim3(WB_NT), lds, st, a); break;
        case 3: hipLaunchKernelGGL(window_attn_bwd_kernel<3>, grid, dim3(WB_NT), lds, st, a); break;
        default: hipLaunchKernelGGL(window_attn_bwd_kernel<4>, grid, dim3(WB_NT), lds, st, a); break;
    }
    return nmrf_launch_status();
}

// ---- per-pixel self-edge attention over the N sibling labels: backward ------------------------------------------------------------------
// Forward (BasicAttention.forward_pre, nmrf/models/NMP.py:97-103; nmrf_self_attn_f32): per pixel and head, tokens n, m of the pixel,
//     p = softmax_m(s q_n . k_m),  out_n = sum_m p_nm v_m.      One thread per (pixel, head): the N x N matrices in registers, q / k / v /
// dout rows streamed twice from global (L1-resident: 4 x N x 32 floats).  N <= 4.
template <int N>
__global__ __launch_bounds__(256) void self_attn_bwd_kernel(const float *__restrict__ qkv, const float *__restrict__ dout, int64_t pixels, int C,
                                                            int heads, float s, float *__restrict__ dqkv) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pixels * heads) return;
    const int head = (int)(idx % heads);
    const int64_t t0 = (idx / heads) * N;
    const int ld = 3 * C;
    const float *base = qkv + t0 * ld + head * 32;
    const float *gbase = dout + t0 * C + head * 32;
    float l[N][N], dp[N][N];
#pragma unroll
    for (int n = 0; n < N; ++n)
#pragma unroll
        for (int m = 0; m < N; ++m) l[n][m] = dp[n][m] = 0.f;
    for (int c = 0; c < 32; c += 4) {
        float4 q[N], k[N], v[N], g[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            q[n] = ldg4(base + (size_t)n * ld + c);
            k[n] = ldg4(base + (size_t)n * ld + C + c);
            v[n] = ldg4(base + (size_t)n * ld + 2 * C + c);
            g[n] = ldg4(gbase + (size_t)n * C + c);
        }
#pragma unroll
        for (int n = 0; n < N; ++n)
#pragma unroll
            for (int m = 0; m < N; ++m) {
                l[n][m] += q[n].x * k[m].x + q[n].y * k[m].y + q[n].z * k[m].z + q[n].w * k[m].w;
                dp[n][m] += g[n].x * v[m].x + g[n].y * v[m].y + g[n].z * v[m].z + g[n].w * v[m].w;
            }
    }
    float p[N][N], ds[N][N];
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float mx = -INFINITY, z = 0.f, d = 0.f;
#pragma unroll
        for (int m = 0; m < N; ++m) mx = fmaxf(mx, l[n][m] * s);
#pragma unroll
        for (int m = 0; m < N; ++m) { p[n][m] = expf(l[n][m] * s - mx); z += p[n][m]; }
        const float rz = 1.0f / z;
#pragma unroll
        for (int m = 0; m < N; ++m) { p[n][m] *= rz; d = fmaf(p[n][m], dp[n][m], d); }
#pragma unroll
        for (int m = 0; m < N; ++m) ds[n][m] = p[n][m] * (dp[n][m] - d) * s;
    }
    float *obase = dqkv + t0 * ld + head * 32;
    for (int c = 0; c < 32; c += 4) {
        float4 q[N], k[N], g[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            q[n] = ldg4(base + (size_t)n * ld + c);
            k[n] = ldg4(base + (size_t)n * ld + C + c);
            g[n] = ldg4(gbase + (size_t)n * C + c);
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float4 dq = {0.f, 0.f, 0.f, 0.f}, dk = {0.f, 0.f, 0.f, 0.f}, dv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < N; ++m) {
                dq.x = fmaf(ds[n][m], k[m].x, dq.x); dq.y = fmaf(ds[n][m], k[m].y, dq.y); dq.z = fmaf(ds[n][m], k[m].z, dq.z); dq.w = fmaf(ds[n][m], k[m].w, dq.w);
                dk.x = fmaf(ds[m][n], q[m].x, dk.x); dk.y = fmaf(ds[m][n], q[m].y, dk.y); dk.z = fmaf(ds[m][n], q[m].z, dk.z); dk.w = fmaf(ds[m][n], q[m].w, dk.w);
                dv.x = fmaf(p[m][n], g[m].x, dv.x); dv.y = fmaf(p[m][n], g[m].y, dv.y); dv.z = fmaf(p[m][n], g[m].z, dv.z); dv.w = fmaf(p[m][n], g[m].w, dv.w);
            }
            stg4(obase + (size_t)n * ld + c, dq);
            stg4(obase + (size_t)n * ld + C + c, dk);
            stg4(obase + (size_t)n * ld + 2 * C + c, dv);
        }
    }
}

// qkv [T,3C] fp32 (q | k | v), dout [T,C] -> dqkv [T,3C]; T a multiple of N, N in {1, 2, 4}, heads * 32 == C.
extern "C" int nmrf_self_attn_bwd_f32(const float *qkv, const float *dout, int64_t T, int N, int C, int heads, float *dqkv, void *stream) {
    if (!qkv || !dout || !dqkv) return NMRF_ENULL;
    if (T < 1 || (N != 1 && N != 2 && N != 4) || T % N || heads * 32 != C || (C & 3)) return NMRF_EINVAL;
    const int64_t pixels = T / N, items = pixels * heads;
    const float s = 1.0f / sqrtf(32.0f);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)ceil_div64(items, 256));
    switch (N) {
        case 1: hipLaunchKernelGGL(self_attn_bwd_kernel<1>, grid, dim3(256), 0, st, qkv, dout, pixels, C, heads, s, dqkv); break;
        case 2: hipLaunchKernelGGL(self_attn_bwd_kernel<2>, grid, dim3(256), 0, st, qkv, dout, pixels, C, heads, s, dqkv); break;
        case 4: hipLaunchKernelGGL(self_attn_bwd_kernel<4>, grid, dim3(256), 0, st, qkv, dout, pixels, C, heads, s, dqkv); break;
    }
    return nmrf_launch_status();
}

// ---- cross-stripe (CSWin, split size 1) attention with LePE: backward --------------------------------------------------------------------
// Forward (CSWinAttention.forward / get_rpe, nmrf/models/NMP.py:429-505; oracle/nmrf_oracle.py:stripe_attention): channel half `axis` of
// q | k | v (axis 0: channels 0..63, vertical stripes = columns; axis 1: channels 64..127, horizontal stripes = rows), 2 heads of 32;
// per stripe and head, tokens i, j = (pixel along the stripe, label):
//     p = softmax_j(s q_i . k_j)  (-inf between different labels of one pixel),   out_i = sum_j p_ij v_j + rpe_i
//     rpe[p, n] = tc * v[p, n] + tm * sum_k v[p - 1, k] + tp * sum_k v[p + 1, k]     (tm, tc, tp: the centre column / row of the depthwise 3x3)
// Backward: the attention part as in window_attn_bwd_kernel (P and dS in a global scratch, one thread per query row, then per key column;
// q / k / v / dout rows come from global memory -- a stripe has up to W * N tokens); LePE: dv += tc * dout + tm * sum dout[p + 1] + tp *
// sum dout[p - 1], and per (stripe, head, image) partial sums of the three tap gradients.
struct StripeBwdArgs {
    const float *qkv, *lepe, *dout;
    float *dqkv, *dtap_parts, *scratch;
    int H, W, N, axis;
    float scale;
};

__global__ __launch_bounds__(256) void stripe_attn_bwd_kernel(StripeBwdArgs a) {
    const int N = a.N, L = a.axis == 0 ? a.H : a.W, Ts = L * N;
    const int stripe = blockIdx.x, head = blockIdx.y, bimg = blockIdx.z;
    const int n_stripes = gridDim.x;
    const int coff = a.axis * 64 + head * 32;
    const int ld = 384, C = 128;
    const float s = a.scale;
    const int tid = threadIdx.x;
    auto row_of = [&](int t) -> size_t {                       // global token row of stripe token t = (pixel p, label n)
        const int p = t / N, n = t - p * N;
        const int y = a.axis == 0 ? p : stripe, x = a.axis == 0 ? stripe : p;
        return ((size_t)(bimg * a.H + y) * a.W + x) * N + n;
    };
    float *P = a.scratch + ((size_t)(bimg * 2 + head) * n_stripes + stripe) * 2 * Ts * Ts, *dS = P + (size_t)Ts * Ts;
    // ---- phase 1: one thread per query row ----------------------------------------------------------------------------------------------
    for (int i = tid; i < Ts; i += 256) {
        const size_t ri = row_of(i);
        float q[32], go[32];
        for (int c = 0; c < 32; c += 4) {
            const float4 a4 = ldg4(a.qkv + ri * ld + coff + c), g4 = ldg4(a.dout + ri * C + coff + c);
            q[c] = a4.x; q[c + 1] = a4.y; q[c + 2] = a4.z; q[c + 3] = a4.w;
            go[c] = g4.x; go[c + 1] = g4.y; go[c + 2] = g4.z; go[c + 3] = g4.w;
        }
        float m = -INFINITY;
        for (int j = 0; j < Ts; ++j) {
            float l = -INFINITY;
            if (!(i / N == j / N && i != j)) {
                const float *k = a.qkv + row_of(j) * ld + C + coff;
                float acc = 0.f;
                for (int c = 0; c < 32; ++c) acc = fmaf(q[c], k[c], acc);
                l = acc * s;
            }
            P[(size_t)i * Ts + j] = l;
            m = fmaxf(m, l);
        }
        float Z = 0.f;
        for (int j = 0; j < Ts; ++j) {
            const float l = P[(size_t)i * Ts + j];
            const float e = l == -INFINITY ? 0.f : expf(l - m);
            P[(size_t)i * Ts + j] = e;
            Z += e;
        }
        const float rz = 1.0f / Z;
        float D = 0.f;
        for (int j = 0; j < Ts; ++j) {
            const float p = P[(size_t)i * Ts + j] * rz;
            P[(size_t)i * Ts + j] = p;
            float dp = 0.f;
            if (p != 0.f) {
                const float *v = a.qkv + row_of(j) * ld + 2 * C + coff;
                for (int c = 0; c < 32; ++c) dp = fmaf(go[c], v[c], dp);
            }
            dS[(size_t)i * Ts + j] = dp;
            D = fmaf(p, dp, D);
        }
        float dq[32];
        for (int c = 0; c < 32; ++c) dq[c] = 0.f;
        for (int j = 0; j < Ts; ++j) {
            const float ds = P[(size_t)i * Ts + j] * (dS[(size_t)i * Ts + j] - D);
            dS[(size_t)i * Ts + j] = ds;
            if (ds != 0.f) {
                const float *k = a.qkv + row_of(j) * ld + C + coff;
                for (int c = 0; c < 32; ++c) dq[c] = fmaf(ds, k[c], dq[c]);
            }
        }
        float *o = a.dqkv + ri * ld + coff;
        for (int c = 0; c < 32; ++c) o[c] = dq[c] * s;
    }
    __threadfence_block();
    __syncthreads();
    // ---- phase 2: one thread per key column: dk, dv (attention + LePE) ---------------------------------------------------------------------
    const float *taps = a.lepe + (size_t)(head * 32) * 9;      // channel ch of the half: taps[ch * 9 + 3 * dy + dx]
    const int t_m = a.axis == 0 ? 1 : 3, t_c = 4, t_p = a.axis == 0 ? 7 : 5;      // centre column (dy = 0, 1, 2) / centre row (dx = 0, 1, 2)
    for (int j = tid; j < Ts; j += 256) {
        float dk[32], dv[32];
        for (int c = 0; c < 32; ++c) dk[c] = dv[c] = 0.f;
        for (int i = 0; i < Ts; ++i) {
            const float ds = dS[(size_t)i * Ts + j], p = P[(size_t)i * Ts + j];
            if (p == 0.f && ds == 0.f) continue;
            const size_t ri = row_of(i);
            const float *q = a.qkv + ri * ld + coff, *go = a.dout + ri * C + coff;
            for (int c = 0; c < 32; ++c) {
                dk[c] = fmaf(ds, q[c], dk[c]);
                dv[c] = fmaf(p, go[c], dv[c]);
            }
        }
        const int pj = j / N;
        const float *gj = a.dout + row_of(j) * C + coff;
        for (int c = 0; c < 32; ++c) {
            float e = taps[c * 9 + t_c] * gj[c];
            if (pj + 1 < L) {                                  // v[pj] is the "previous" pixel of pj + 1: weight tm there
                float sum = 0.f;
                for (int n = 0; n < N; ++n) sum += a.dout[row_of((pj + 1) * N + n) * C + coff + c];
                e = fmaf(taps[c * 9 + t_m], sum, e);
            }
            if (pj > 0) {                                      // ... and the "next" pixel of pj - 1: weight tp there
                float sum = 0.f;
                for (int n = 0; n < N; ++n) sum += a.dout[row_of((pj - 1) * N + n) * C + coff + c];
                e = fmaf(taps[c * 9 + t_p], sum, e);
            }
            dv[c] += e;
        }
        float *o = a.dqkv + row_of(j) * ld + coff;
        for (int c = 0; c < 32; ++c) {
            o[C + c] = dk[c] * s;
            o[2 * C + c] = dv[c];
        }
    }
    // ---- phase 3: the three tap gradients of this head's 32 channels, summed over the stripe's tokens (threads 0 .. 31) ----------------------
    if (tid < 32) {
        const int c = tid;
        float gm = 0.f, gc = 0.f, gp = 0.f;
        for (int p = 0; p < L; ++p) {
            float vprev = 0.f, vnext = 0.f;
            for (int n = 0; n < N; ++n) {
                if (p > 0) vprev += a.qkv[row_of((p - 1) * N + n) * ld + 2 * C + coff + c];
                if (p + 1 < L) vnext += a.qkv[row_of((p + 1) * N + n) * ld + 2 * C + coff + c];
            }
            for (int n = 0; n < N; ++n) {
                const size_t r = row_of(p * N + n);
                const float g = a.dout[r * C + coff + c];
                gc = fmaf(g, a.qkv[r * ld + 2 * C + coff + c], gc);
                gm = fmaf(g, vprev, gm);
                gp = fmaf(g, vnext, gp);
            }
        }
        float *part = a.dtap_parts + (((size_t)bimg * n_stripes + stripe) * 64 + head * 32 + c) * 3;
        part[0] = gm; part[1] = gc; part[2] = gp;
    }
}

// qkv [B,H,W,N,384] fp32 rows, lepe_v / lepe_h [64,1,3,3] (attns.0 / attns.1 get_v.weight), dout [B,H,W,N,128] -> dqkv (every element
// written); dtap_v_parts [B*W][64][3], dtap_h_parts [B*H][64][3]: per stripe the gradients of the (previous, centre, next) taps of the 64
// channels of that axis (sum over the parts; the other six taps of a 3x3 kernel only ever see the zero padding of a width-1 stripe:
// gradient 0).  scratch: 2 * B * 2 * max(W * (H N)^2, H * (W N)^2) floats.
extern "C" int nmrf_stripe_attn_bwd_f32(const float *qkv, const float *lepe_v, const float *lepe_h, const float *dout, int B, int H, int W, int N,
                                        float *dqkv, float *dtap_v_parts, float *dtap_h_parts, float *scratch, void *stream) {
    if (!qkv || !lepe_v || !lepe_h || !dout || !dqkv || !dtap_v_parts || !dtap_h_parts || !scratch) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || N < 1 || B > 65535) return NMRF_EINVAL;
    const float scale = 1.0f / sqrtf(32.0f);
    StripeBwdArgs av{qkv, lepe_v, dout, dqkv, dtap_v_parts, scratch, H, W, N, 0, scale};
    hipLaunchKernelGGL(stripe_attn_bwd_kernel, dim3(W, 2, B), dim3(256), 0, (hipStream_t)stream, av);
    StripeBwdArgs ah{qkv, lepe_h, dout, dqkv, dtap_h_parts, scratch, H, W, N, 1, scale};
    hipLaunchKernelGGL(stripe_attn_bwd_kernel, dim3(H, 2, B), dim3(256), 0, (hipStream_t)stream, ah);
    return nmrf_launch_status();
}

// ---- the seed filter (DPN.mlp: three Conv1d(k = 5, pad 2) along the disparity axis + softmax, DPN.py:32-38,117-119) as linears ----------
// A Conv1d over D is a Linear on 5-tap columns: col[(p, d)][c * 5 + t] = A[(p, d + t - 2)][c] (zero outside 0 <= d + t - 2 < D), weight
// [O][C][5] flattened -- so its backward is the dgrad / wgrad GEMM above plus the two data movements here and the softmax backward.
// src_pcd != 0: the source is [P][C][D] (the cost volume's layout), else rows (p, d) x C.
__global__ __launch_bounds__(256) void unfold5_kernel(const float *__restrict__ src, int64_t P, int C, int D, int src_pcd, float *__restrict__ col) {
    const int64_t total = P * D * C * 5;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int t = (int)(idx % 5);
        const int c = (int)((idx / 5) % C);
        const int64_t pd = idx / (5 * C);
        const int d = (int)(pd % D);
        const int64_t p = pd / D;
        const int e = d + t - 2;
        float v = 0.f;
        if (e >= 0 && e < D) v = src_pcd ? src[(p * C + c) * D + e] : src[(p * D + e) * C + c];
        col[idx] = v;
    }
}
// dA[(p, e)][c] = sum_t dcol[(p, e - t + 2)][c * 5 + t]   (t ascending: fixed order)
__global__ __launch_bounds__(256) void fold5_kernel(const float *__restrict__ dcol, int64_t P, int C, int D, float *__restrict__ dA) {
    const int64_t total = P * D * C;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)(idx % C);
        const int64_t pe = idx / C;
        const int e = (int)(pe % D);
        const int64_t p = pe / D;
        float s = 0.f;
        for (int t = 0; t < 5; ++t) {
            const int d = e - t + 2;
            if (d >= 0 && d < D) s += dcol[((p * D + d) * C + c) * 5 + t];
        }
        dA[idx] = s;
    }
}
// dz = p * (dp - sum_d p dp), one wave per row of D <= 64 entries
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float *__restrict__ prob, const float *__restrict__ dprob, int64_t P, int D,
                                                          float *__restrict__ dz) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < P; r += (int64_t)gridDim.x * 4) {
        const float p = lane < D ? prob[r * D + lane] : 0.f, g = lane < D ? dprob[r * D + lane] : 0.f;
        const float dot = wave_sum(p * g);
        if (lane < D) dz[r * D + lane] = p * (g - dot);
    }
}
extern "C" int nmrf_unfold5_f32(const float *src, int64_t P, int C, int D, int src_pcd, float *col, void *stream) {
    if (!src || !col) return NMRF_ENULL;
    if (P < 1 || C < 1 || D < 1) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(P * D * C * 5, 256 * 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(unfold5_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, P, C, D, src_pcd, col);
    return nmrf_launch_status();
}
extern "C" int nmrf_fold5_f32(const float *dcol, int64_t P, int C, int D, float *dA, void *stream) {
    if (!dcol || !dA) return NMRF_ENULL;
    if (P < 1 || C < 1 || D < 1) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(P * D * C, 256 * 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(fold5_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dcol, P, C, D, dA);
    return nmrf_launch_status();
}
extern "C" int nmrf_softmax_bwd_f32(const float *prob, const float *dprob, int64_t P, int D, float *dz, void *stream) {
    if (!prob || !dprob || !dz) return NMRF_ENULL;
    if (P < 1 || D < 1 || D > 64) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(P, 4 * 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, prob, dprob, P, D, dz);
    return nmrf_launch_status();
}

// ---- backward towards the feature maps (the stock-PyTorch convolutions take over from there: north_star keeps the backbone on stock ROCm) ---
// (1) group-wise correlation volume (submodule.py:4-23): cv[(b,y,x)][g][d] = mean_c f1[b,g*cpg+c,y,x] * f2[b,g*cpg+c,y,x-d] (x >= d)
//     df1[b,ch,y,x] = (1/cpg) sum_{d <= x} dcv[(b,y,x)][g][d] f2[b,ch,y,x-d];   df2[b,ch,y,x'] = (1/cpg) sum_{x'+d < W} dcv[(b,y,x'+d)][g][d] f1[b,ch,y,x'+d]
__global__ __launch_bounds__(256) void cost_volume_bwd_kernel(const float *__restrict__ f1, const float *__restrict__ f2, const float *__restrict__ dcv,
                                                              int B, int C, int H, int W, int D, int G, float *__restrict__ df1,
                                                              float *__restrict__ df2) {
    const int64_t total = (int64_t)B * C * H * W;
    const int cpg = C / G;
    const float inv = 1.0f / (float)cpg;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int ch = (int)((idx / ((int64_t)W * H)) % C);
        const int b = (int)(idx / ((int64_t)W * H * C));
        const int g = ch / cpg;
        const float *r1 = f1 + (((int64_t)b * C + ch) * H + y) * W, *r2 = f2 + (((int64_t)b * C + ch) * H + y) * W;
        const float *drow = dcv + ((int64_t)(b * H + y) * W) * G * D + (int64_t)g * D;       // + x * G * D + d
        float a1 = 0.f, a2 = 0.f;
        for (int d = 0; d < D; ++d) {
            if (d <= x) a1 = fmaf(drow[(int64_t)x * G * D + d], r2[x - d], a1);
            if (x + d < W) a2 = fmaf(drow[(int64_t)(x + d) * G * D + d], r1[x + d], a2);
        }
        df1[idx] = a1 * inv;
        df2[idx] = a2 * inv;
    }
}
extern "C" int nmrf_cost_volume_bwd_f32(const float *f1, const float *f2, const float *dcv, int B, int C, int H, int W, int D, int G,
                                        float *df1, float *df2, void *stream) {
    if (!f1 || !f2 || !dcv || !df1 || !df2) return NMRF_ENULL;
    if (B < 1 || C < 1 || H < 1 || W < 1 || D < 1 || G < 1 || C % G) return NMRF_EINVAL;
    int64_t blocks = ceil_div64((int64_t)B * C * H * W, 256);
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(cost_volume_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f1, f2, dcv, B, C, H, W, D, G, df1, df2);
    return nmrf_launch_status();
}

// (2) the cost taps of the seed embedding (Propagation.sample_cost, NMP.py:619-634): cost[(p,n)][g*9+t] = cv[p][g][clamp(seed_n - 4 + t, 0, D-1)]
//     dcv[p][g][d] = sum over the labels n and taps t that read bin d  (one thread per (p, g, d): fixed order)
__global__ __launch_bounds__(256) void seed_taps_bwd_kernel(const float *__restrict__ dcost, const int64_t *__restrict__ seeds, int64_t P, int N, int G, int D,
                                                            int ldc, float *__restrict__ dcv) {
    const int64_t total = P * G * D;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int d = (int)(idx % D), g = (int)((idx / D) % G);
        const int64_t p = idx / ((int64_t)D * G);
        float s = 0.f;
        for (int n = 0; n < N; ++n) {
            const int sd = (int)seeds[p * N + n];
            for (int t = 0; t < 9; ++t) {
                int e = sd - 4 + t;
                e = e < 0 ? 0 : (e > D - 1 ? D - 1 : e);
                if (e == d) s += dcost[(p * N + n) * ldc + g * 9 + t];
            }
        }
        dcv[idx] = s;
    }
}
extern "C" int nmrf_seed_taps_bwd_f32(const float *dcost, const int64_t *seeds, int64_t P, int N, int G, int D, int ldc, float *dcv, void *stream) {
    if (!dcost || !seeds || !dcv) return NMRF_ENULL;
    if (P < 1 || N < 1 || G < 1 || D < 1 || ldc < G * 9) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(P * G * D, 256);
    if (blocks > 65535 * 8) blocks = 65535 * 8;
    hipLaunchKernelGGL(seed_taps_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dcost, seeds, P, N, G, D, ldc, dcv);
    return nmrf_launch_status();
}

// (3) warp + correlation + concat (Inference.sample_fmap / corr, NMP.py:683-741): rows [left f1 (Cf) | f2 warped at x - label (Cf) | group corr (Gr)],
//     corr_g = mean_c g1[g*cpg+c] * warp(g2)[g*cpg+c].  Labels are constants (the reference detaches them).  The sampling is bilinear along x
//     (ix of the reference's float round trip; the ~1e-7 share of the adjacent row that round trip leaks -- SURVEY H6 -- is not differentiated).
//     LEFT maps: one thread per (b, ch, y, x): df1 = sum_n drow[t][ch]; dg1 = sum_n dcorr[t][g] * warp(g2)[ch] / cpg.
//     RIGHT maps: one wave per (b, y, xd), lanes = channels: the wave walks the W * N tokens of the row (uniform), and where a token's
//     sample touches xd every lane adds its channel's share -- a gather, deterministic.
__device__ __forceinline__ void wc_tap(float label, int x, int W, int &x0, float &w0, float &w1) {
#pragma clang fp contract(off)
    const float gx = 2.0f * ((float)x + (-label)) / (float)(W - 1) - 1.0f;
    const float ix = (gx + 1.0f) * ((float)(W - 1) / 2.0f);
    float x0f = floorf(ix);
    w1 = ix - x0f;
    w0 = 1.0f - w1;
    x0f = fminf(fmaxf(x0f, -2.0f), (float)W);
    x0 = (int)x0f;
}
struct WcBwdArgs {
    const float *labels, *drow, *f1, *f2, *g1, *g2;     // maps NCHW; drow [T, 2 Cf + Gr]
    float *df1, *df2, *dg1, *dg2;
    int B, H, W, N, Cf, Cg, Gr;
};
__global__ __launch_bounds__(256) void warp_corr_bwd_left_kernel(WcBwdArgs a) {
    const int64_t plane = (int64_t)a.H * a.W, total = (int64_t)a.B * (a.Cf + a.Cg) * plane;
    const int ld = 2 * a.Cf + a.Gr, cpg = a.Cg / a.Gr;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int x = (int)(idx % a.W), y = (int)((idx / a.W) % a.H);
        const int chall = (int)((idx / plane) % (a.Cf + a.Cg));
        const int b = (int)(idx / (plane * (a.Cf + a.Cg)));
        const int64_t t0 = (((int64_t)b * a.H + y) * a.W + x) * a.N;
        float s = 0.f;
        if (chall < a.Cf) {
            for (int n = 0; n < a.N; ++n) s += a.drow[(t0 + n) * ld + chall];
            a.df1[((int64_t)b * a.Cf + chall) * plane + (int64_t)y * a.W + x] = s;
        } else {
            const int ch = chall - a.Cf, g = ch / cpg;
            const float *row2 = a.g2 + ((int64_t)b * a.Cg + ch) * plane + (int64_t)y * a.W;
            for (int n = 0; n < a.N; ++n) {
                int x0; float w0, w1;
                wc_tap(a.labels[t0 + n], x, a.W, x0, w0, w1);
                float v = 0.f;
                if (x0 >= 0 && x0 <= a.W - 1) v = w0 * row2[x0];
                if (x0 + 1 >= 0 && x0 + 1 <= a.W - 1) v = fmaf(w1, row2[x0 + 1], v);
                s = fmaf(a.drow[(t0 + n) * ld + 2 * a.Cf + g], v, s);
            }
            a.dg1[((int64_t)b * a.Cg + ch) * plane + (int64_t)y * a.W + x] = s / (float)cpg;
        }
    }
}
__global__ __launch_bounds__(256) void warp_corr_bwd_right_kernel(WcBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t plane = (int64_t)a.H * a.W, items = (int64_t)a.B * plane;
    const int ld = 2 * a.Cf + a.Gr, cpg = a.Cg / a.Gr;
    const int nchunk = (a.Cf + a.Cg + 63) / 64;            // channel chunks of 64 lanes: the Cf channels of f2 first, then the Cg of g2
    for (int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); it < items; it += (int64_t)gridDim.x * 4) {
        const int xd = (int)(it % a.W), y = (int)((it / a.W) % a.H), b = (int)(it / plane);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        const int64_t trow = (((int64_t)b * a.H + y) * a.W) * a.N;
        for (int x = 0; x < a.W; ++x)
            for (int n = 0; n < a.N; ++n) {
                int x0; float w0, w1;
                wc_tap(a.labels[trow + (int64_t)x * a.N + n], x, a.W, x0, w0, w1);
                float wgt;
                if (x0 == xd) wgt = w0;
                else if (x0 + 1 == xd) wgt = w1;
                else continue;                              // (wave-uniform: the weight depends on (x, n, xd) only)
                const float *dr = a.drow + (trow + (int64_t)x * a.N + n) * ld;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int chall = 64 * k + lane;
                    if (k < nchunk && chall < a.Cf + a.Cg) {
                        float up;
                        if (chall < a.Cf) up = dr[a.Cf + chall];
                        else {
                            const int ch = chall - a.Cf;
                            up = dr[2 * a.Cf + ch / cpg] * a.g1[((int64_t)b * a.Cg + ch) * plane + (int64_t)y * a.W + x] / (float)cpg;
                        }
                        acc[k] = fmaf(wgt, up, acc[k]);
                    }
                }
            }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int chall = 64 * k + lane;
            if (k < nchunk && chall < a.Cf + a.Cg) {
                if (chall < a.Cf) a.df2[((int64_t)b * a.Cf + chall) * plane + (int64_t)y * a.W + xd] = acc[k];
                else a.dg2[((int64_t)b * a.Cg + (chall - a.Cf)) * plane + (int64_t)y * a.W + xd] = acc[k];
            }
        }
    }
}
// labels [B*H*W*N] (constants), drow [T, 2 Cf + Gr] = the gradient of nmrf_warp_corr_concat_f32's rows, maps [B,C,H,W] (NCHW) ->
// df1, df2 [B,Cf,H,W], dg1, dg2 [B,Cg,H,W].  Cf + Cg <= 512, Cg % Gr == 0.
extern "C" int nmrf_warp_corr_concat_bwd_f32(const float *labels, const float *drow, const float *f1, const float *f2, const float *g1,
                                             const float *g2, int B, int H, int W, int N, int Cf, int Cg, int Gr, float *df1, float *df2,
                                             float *dg1, float *dg2, void *stream) {
    if (!labels || !drow || !f1 || !f2 || !g1 || !g2 || !df1 || !df2 || !dg1 || !dg2) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 2 || N < 1 || Cf < 1 || Cg < 1 || Gr < 1 || Cg % Gr || Cf + Cg > 512) return NMRF_EINVAL;
    WcBwdArgs a{labels, drow, f1, f2, g1, g2, df1, df2, dg1, dg2, B, H, W, N, Cf, Cg, Gr};
    hipStream_t st = (hipStream_t)stream;
    int64_t bl = ceil_div64((int64_t)B * (Cf + Cg) * H * W, 256);
    if (bl > 65535 * 8) bl = 65535 * 8;
    hipLaunchKernelGGL(warp_corr_bwd_left_kernel, dim3((unsigned)bl), dim3(256), 0, st, a);
    int64_t br = ceil_div64((int64_t)B * H * W, 4);
    if (br > 65535 * 8) br = 65535 * 8;
    hipLaunchKernelGGL(warp_corr_bwd_right_kernel, dim3((unsigned)br), dim3(256), 0, st, a);
    return nmrf_launch_status();
}
