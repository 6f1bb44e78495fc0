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
};

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
    for (int c = c0; c < c1; ++c) {
        const int k0 = c * 16 + 8 * kh;
        float av[8], bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            const bool kok = k < a.K;
            av[e] = (iok && kok) ? pa[(int64_t)k * a.sa_k] : 0.f;
            bv[e] = (jok && kok) ? pb[(int64_t)k * a.sb_k] : 0.f;
        }
        h16x8 ah, al, bh, bl;
        split8u_g(av, ah, al, guard);
        split8u_g(bv, bh, bl, guard);
        split_mma1(ah, al, bh, bl, acc);
    }
    float *cp = a.C + (int64_t)blockIdx.y * a.split_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tm * 32 + mfma_row(r, kh);
        if (row < a.M && jok) cp[(int64_t)row * a.ldc + j] = acc[r];
    }
    split_guard_commit(guard, a.range_flag);
}

extern "C" int nmrf_gemm_split_f32(const float *A, int64_t sa_i, int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_j, int M, int N,
                                   int K, float *C, int ldc, int splits, int64_t split_stride, int *range_flag, void *stream) {
    if (!A || !B || !C) return NMRF_ENULL;
    if (M < 1 || N < 1 || K < 1 || ldc < N || splits < 1 || splits > 65535 || (splits > 1 && split_stride < (int64_t)M * ldc)) return NMRF_EINVAL;
    const int tiles_m = (M + 31) / 32, tiles_n = (N + 31) / 32, nchunks = (K + 15) / 16;
    const int cps = (nchunks + splits - 1) / splits;
    GemmArgs a{A, B, C, M, N, K, sa_i, sa_k, sb_k, sb_j, ldc, tiles_n, tiles_m * tiles_n, cps, split_stride, range_flag};
    hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)((a.n_tiles + 3) / 4), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, a);
    return nmrf_launch_status();
}

// out[i] = sum_s parts[s*stride + i], s in ascending order (the deterministic second pass of every split reduction here)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float *__restrict__ parts, int S, int64_t n, int64_t stride, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += parts[(int64_t)k * stride + i];
        out[i] = s;
    }
}
extern "C" int nmrf_sum_partials_f32(const float *parts, int S, int64_t n, int64_t stride, float *out, void *stream) {
    if (!parts || !out) return NMRF_ENULL;
    if (S < 1 || n < 1 || stride < n) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, parts, S, n, stride, out);
    return nmrf_launch_status();
}

// bias gradient: parts[b][n] = sum over the rows of block b of x[t][n]   (then nmrf_sum_partials_f32 over b)
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float *__restrict__ x, int64_t T, int N, int rows_per_block,
                                                              float *__restrict__ parts) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < T ? r0 + rows_per_block : T;
    for (int c = threadIdx.x; c < N; c += 256) {
        float s = 0.f;
        for (int64_t r = r0; r < r1; ++r) s += x[r * N + c];
        parts[(int64_t)blockIdx.x * N + c] = s;
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
