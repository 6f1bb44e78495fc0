// A15: multi-scale deformable attention, forward and backward (fp32 / fp64).
//
// value [B,S,M,D] keeps the head's D channels contiguous, so a thread that owns CH consecutive
// channels of one (b, query, head) reads each bilinear tap as one vector.  HBM/L2-bound gather.
// Forward : one thread = (b, q, m, channel chunk); loops levels x points.
// Backward: one thread = one sample (b, q, m, l, p); loops all D channels, so grad_loc / grad_w need no
//           cross-thread reduction (the reference needs seven shared-memory reduction variants for that);
//           grad_value is scattered with hardware float atomics.
// Sampling convention (ms_deform_im2col_cuda.cuh:285-288): h_im = loc_y*H - 0.5, zero outside,
// identical to grid_sample(align_corners=False).
#include "common.h"
#include "split_mfma.h"

template <typename T, int CH>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const T *__restrict__ loc, const T *__restrict__ wgt, int B, int S, int M,
        int D, int L, int Lq, int P, T *__restrict__ out) {
    const int chunks = D / CH;
    const int64_t total = (int64_t)B * Lq * M * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t bqm = i / chunks;                        // (b*Lq + q)*M + m
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / ((int64_t)M * Lq));
        const T *lp = loc + bqm * L * P * 2;
        const T *wp = wgt + bqm * L * P;
        T acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0;
        for (int l = 0; l < L; ++l) {
            const int Hh = (int)shapes[2 * l], Ww = (int)shapes[2 * l + 1];
            const T *vbase = value + (((size_t)b * S + (size_t)lvl_start[l]) * M + m) * D + ck * CH;
            for (int p = 0; p < P; ++p) {
                const T w_im = lp[(l * P + p) * 2] * Ww - (T)0.5;
                const T h_im = lp[(l * P + p) * 2 + 1] * Hh - (T)0.5;
                const T aw = wp[l * P + p];
                if (h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww) {
                    const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
                    const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
                    const T tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    T s[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) s[c] = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int yy = h0 + (k >> 1), xx = w0 + (k & 1);
                        if (yy >= 0 && yy < Hh && xx >= 0 && xx < Ww) {
                            const T *v = vbase + ((size_t)yy * Ww + xx) * M * D;
#pragma unroll
                            for (int c = 0; c < CH; ++c) s[c] += tw[k] * v[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[c] += s[c] * aw;
                }
            }
        }
        T *o = out + bqm * D + ck * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = acc[c];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const T *__restrict__ loc, const T *__restrict__ wgt,
        const T *__restrict__ gout, int B, int S, int M, int D, int L, int Lq, int P, T *__restrict__ gvalue,
        T *__restrict__ gloc, T *__restrict__ gw) {
    const int64_t total = (int64_t)B * Lq * M * L * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l = (int)((i / P) % L);
        const int64_t bqm = i / ((int64_t)L * P);
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / ((int64_t)M * Lq));
        const int Hh = (int)shapes[2 * l], Ww = (int)shapes[2 * l + 1];
        const T w_im = loc[i * 2] * Ww - (T)0.5;
        const T h_im = loc[i * 2 + 1] * Hh - (T)0.5;
        const T aw = wgt[i];
        T g_w = 0, g_x = 0, g_y = 0;
        if (h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww) {
            const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
            const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
            const T tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            // d(weight_k)/d(w_im), d(weight_k)/d(h_im)   (ms_deform_im2col_cuda.cuh:112-156)
            const T dwx[4] = {-hh, hh, -lh, lh};
            const T dwy[4] = {-hw, -lw, hw, lw};
            const size_t voff = (((size_t)b * S + (size_t)lvl_start[l]) * M + m) * D;
            const T *go = gout + bqm * D;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = h0 + (k >> 1), xx = w0 + (k & 1);
                if (yy >= 0 && yy < Hh && xx >= 0 && xx < Ww) {
                    const size_t o = voff + ((size_t)yy * Ww + xx) * M * D;
                    T dot = 0;
                    for (int c = 0; c < D; ++c) {
                        const T g = go[c];
                        dot += g * value[o + c];
                        atomicAdd(gvalue + o + c, tw[k] * aw * g);
                    }
                    g_w += tw[k] * dot;
                    g_x += dwx[k] * dot;
                    g_y += dwy[k] * dot;
                }
            }
        }
        gw[i] = g_w;
        gloc[i * 2] = Ww * g_x * aw;
        gloc[i * 2 + 1] = Hh * g_y * aw;
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = ceil_div64(total, 256);
    return (unsigned)(blocks > 65536 ? 65536 : (blocks < 1 ? 1 : blocks));
}

template <typename T>
static int msda_forward(const T *value, const int64_t *shapes, const int64_t *lvl_start, const T *loc, const T *w, int B,
                        int S, int M, int D, int L, int Lq, int P, T *out, void *stream) {
    if (!value || !shapes || !lvl_start || !loc || !w || !out) return NMRF_ENULL;
    if (B < 1 || S < 1 || M < 1 || D < 1 || L < 1 || Lq < 1 || P < 1) return NMRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (D % 4 == 0) {
        hipLaunchKernelGGL((msda_fwd_kernel<T, 4>), dim3(grid_for((int64_t)B * Lq * M * (D / 4))), dim3(256), 0, st, value,
                           shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out);
    } else {
        hipLaunchKernelGGL((msda_fwd_kernel<T, 1>), dim3(grid_for((int64_t)B * Lq * M * D)), dim3(256), 0, st, value,
                           shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out);
    }
    return nmrf_launch_status();
}

template <typename T>
static int msda_backward(const T *value, const int64_t *shapes, const int64_t *lvl_start, const T *loc, const T *w,
                         const T *gout, int B, int S, int M, int D, int L, int Lq, int P, T *gvalue, T *gloc, T *gw,
                         void *stream) {
    if (!value || !shapes || !lvl_start || !loc || !w || !gout || !gvalue || !gloc || !gw) return NMRF_ENULL;
    if (B < 1 || S < 1 || M < 1 || D < 1 || L < 1 || Lq < 1 || P < 1) return NMRF_EINVAL;
    hipLaunchKernelGGL((msda_bwd_kernel<T>), dim3(grid_for((int64_t)B * Lq * M * L * P)), dim3(256), 0,
                       (hipStream_t)stream, value, shapes, lvl_start, loc, w, gout, B, S, M, D, L, Lq, P, gvalue, gloc, gw);
    return nmrf_launch_status();
}

extern "C" int nmrf_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start, const float *loc,
                                     const float *w, int B, int S, int M, int D, int L, int Lq, int P, float *out,
                                     void *stream) {
    return msda_forward<float>(value, shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out, stream);
}
extern "C" int nmrf_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start,
                                     const double *loc, const double *w, int B, int S, int M, int D, int L, int Lq, int P,
                                     double *out, void *stream) {
    return msda_forward<double>(value, shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out, stream);
}
extern "C" int nmrf_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start,
                                      const float *loc, const float *w, const float *grad_out, int B, int S, int M, int D,
                                      int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_w,
                                      void *stream) {
    return msda_backward<float>(value, shapes, lvl_start, loc, w, grad_out, B, S, M, D, L, Lq, P, grad_value, grad_loc,
                                grad_w, stream);
}
extern "C" int nmrf_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start,
                                      const double *loc, const double *w, const double *grad_out, int B, int S, int M,
                                      int D, int L, int Lq, int P, double *grad_value, double *grad_loc, double *grad_w,
                                      void *stream) {
    return msda_backward<double>(value, shapes, lvl_start, loc, w, grad_out, B, S, M, D, L, Lq, P, grad_value, grad_loc,
                                 grad_w, stream);
}

// ------------------------------------------------------------------------------------------------
// misc: MFMA lane-layout self-test, error strings, ABI version
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void selftest_mfma_kernel(const float *__restrict__ A, const float *__restrict__ Bm, int K,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[i * K + k0 + hi];          // A[i][k]
        const float b = Bm[(k0 + hi) * 32 + i];      // B[k][j]
        acc = mfma32(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[mfma_row(r, hi) * 32 + i] = acc[r];
}

extern "C" int nmrf_selftest_mfma_f32(const float *A, const float *Bm, int K, float *out, void *stream) {
    if (!A || !Bm || !out) return NMRF_ENULL;
    if (K < 2 || K > 64 || (K & 1)) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, K, out);
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES   // tools-only library libnmrf_hip_debug.so (python -m nmrf_amd.build --debug)
// Debug: attainable v_mfma_f32_32x32x2_f32 rate (tools/kernel_bench.py --which mfma_peak).  CHAINS independent accumulator
// chains per wave, `iters` x 16 MFMAs each; out receives one value per thread so nothing is optimised away.
template <int CHAINS>
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(int iters, float *__restrict__ out) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mfma32(a, b, acc[c]);
        a += 1e-6f;
    }
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// Debug: the streaming-softmax tile loop of the attention kernels with the operands held in registers (no memory
// traffic at all) -- the issue-structure ceiling of that loop at a given number of waves per SIMD.
// VARIANT 0: S chain, softmax, PV chain (as in stripe_attn.hip).  VARIANT 1: two query-independent tiles interleaved
// (S of tile B issued before the softmax of tile A) so a lone wave has MFMA work in flight during its VALU phase.
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void attn_core_peak_kernel(int iters, const float *__restrict__ seed, float *__restrict__ out) {
    float qf[16], kf[16], vf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        qf[s] = seed[(threadIdx.x + s) & 255] * 0.1f; kf[s] = seed[(threadIdx.x * 3 + s) & 255]; vf[s] = seed[(threadIdx.x * 7 + s) & 255];
    }
    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    auto softmax = [&](f32x16 &st) {
        float m_tile = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
        m_tile = half_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); psum += st[r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
    };
    if (VARIANT == 0) {
        for (int it = 0; it < iters; ++it) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) st = mfma32(kf[s], qf[s], st);
            softmax(st);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], st[s], acc_o);
            kf[it & 15] += 1e-7f;
        }
    } else {
        f32x16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sa = mfma32(kf[s], qf[s], sa);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sb = mfma32(kf[s], qf[s], sb);       // S(B) in flight ...
            softmax(sa);                                                       // ... during softmax(A)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], sa[s], acc_o);
            kf[it & 15] += 1e-7f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sa = mfma32(kf[s], qf[s], sa);
            softmax(sb);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], sb[s], acc_o);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc_o[0] + acc_o[15] + l_run;
}

extern "C" int nmrf_debug_attn_core_peak(int variant, int iters, int blocks, const float *seed, float *out, void *stream) {
    if (!out || !seed) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) hipLaunchKernelGGL((attn_core_peak_kernel<0>), dim3(blocks), dim3(256), 0, st, iters, seed, out);
    else hipLaunchKernelGGL((attn_core_peak_kernel<1>), dim3(blocks), dim3(256), 0, st, iters, seed, out);
    return nmrf_launch_status();
}

extern "C" int nmrf_debug_mfma_peak(int chains, int iters, int blocks, float *out, void *stream) {
    if (!out) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (chains == 1) hipLaunchKernelGGL((mfma_peak_kernel<1>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 2) hipLaunchKernelGGL((mfma_peak_kernel<2>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 4) hipLaunchKernelGGL((mfma_peak_kernel<4>), dim3(blocks), dim3(256), 0, st, iters, out);
    else return NMRF_EINVAL;
    return nmrf_launch_status();
}
#endif  // NMRF_DEBUG_PROBES

// ---- split-operand fp16 MFMA self-test (split_mfma.h): one wave, out = A*B with A [32,K], B [K,32] row-major fp32 -------------
//   mode 0: the full split product (3 MFMAs per 16-deep chunk, two accumulators) -- accuracy + lane layout
//   mode 1: hi parts only (plain fp16 product) -- what the split buys, and how fp16 subnormal inputs are treated
//   mode 2: like 0 with the k slots in C/D order (split_kslot) on BOTH operands -- the order chained GEMMs use
//   mode 3: single-accumulator form (split8u / split_mma1): unscaled low parts, A multiplied by 2^10 like a packed weight
__global__ __launch_bounds__(64) void selftest_mfma_f16split_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                                   int K, int mode, float *__restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    f32x16 acc_hh, acc_x;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_hh[r] = acc_x[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = k0 + (mode == 2 ? split_kslot(jj, hi) : 8 * hi + jj);
            av[jj] = A[i * K + k];
            bv[jj] = Bm[k * 32 + i];
        }
        h16x8 ah, al, bh, bl;
        if (mode == 3) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) av[jj] *= 1024.0f;
            split8u(av, ah, al);
            split8u(bv, bh, bl);
            split_mma1(ah, al, bh, bl, acc_hh);
            continue;
        }
        split8(av, ah, al);
        split8(bv, bh, bl);
        if (mode == 1) acc_hh = mfma16h(ah, bh, acc_hh);
        else split_mma(ah, al, bh, bl, acc_hh, acc_x);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        out[mfma_row(r, hi) * 32 + i] = mode == 3 ? acc_hh[r] * (1.0f / 1024.0f) : acc_hh[r] + SPLIT_LO_INV * acc_x[r];
}

extern "C" int nmrf_selftest_mfma_f16split(const float *A, const float *Bm, int K, int mode, float *out, void *stream) {
    if (!A || !Bm || !out) return NMRF_ENULL;
    if (K < 16 || K > 1024 || (K & 15) || mode < 0 || mode > 3) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_mfma_f16split_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, K, mode, out);
    return nmrf_launch_status();
}

// ---- LDS-DMA self-test: global_load_lds_dwordx4 places lane l's 16 bytes at (wave-uniform LDS base) + 16*l ----------------
__global__ __launch_bounds__(256) void selftest_lds_dma_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst) {
    __shared__ __attribute__((aligned(16))) float4 buf[256];
    const int t = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + blockIdx.x * 256 + t),
                                     (__attribute__((address_space(3))) void *)(buf + (t & ~63)), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[blockIdx.x * 256 + t] = buf[t ^ 65];          // read through LDS across waves: dst[t] = src[t ^ 65]
}

extern "C" int nmrf_selftest_lds_dma(const float *src, float *dst, int n_float4, void *stream) {
    if (!src || !dst) return NMRF_ENULL;
    if (n_float4 < 256 || (n_float4 & 255)) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_lds_dma_kernel, dim3(n_float4 / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst));
    return nmrf_launch_status();
}

extern "C" const char *nmrf_strerror(int code) {
    switch (code) {
        case NMRF_OK: return "ok";
        case NMRF_EINVAL: return "invalid size or unsupported configuration";
        case NMRF_ELAUNCH: return "HIP kernel launch failed";
        case NMRF_ENULL: return "null pointer argument";
        default: return "unknown error";
    }
}

extern "C" int nmrf_abi_version(void) { return 19; }
