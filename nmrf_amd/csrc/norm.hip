// InstanceNorm2d (no affine, biased variance, eps) fused with the ReLU / residual-add / ReLU that follow it in the
// stock conv band (backbone residual blocks nmrf/models/backbone.py:38-46, conv heads NMRF.py:56-65,
// DPN.py:45-49).  torch runs this as collect_statistics + calc_invstd + transform_input + clamp (+ add + clamp):
// 4-6 passes over the activation; here it is one reduction pass and one apply pass.
//
// A plane (one channel of one sample, HW contiguous floats) is cut into chunks of IN_CHUNK elements so that the
// 1/2-resolution planes (117k elements, only 128 of them) still fill the chip.  Kernel 1 writes per-chunk
// (mean, M2) with a two-pass (centered) reduction; kernel 2 merges the chunks of its plane with Chan's formula
// -- no E[x^2]-mean^2 cancellation -- and applies  y = [relu]((x-mean)*rstd) ; y = [relu](y + residual).
#include "common.h"

#define IN_CHUNK 8192

__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void in_stats_kernel(const float *__restrict__ x, int64_t HW, int chunks,
                                                      float *__restrict__ ws) {
    __shared__ float red[4];
    const int64_t plane = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t beg = (int64_t)chunk * IN_CHUNK;
    const int64_t end = beg + IN_CHUNK < HW ? beg + IN_CHUNK : HW;
    const float *p = x + plane * HW;
    const int n = (int)(end - beg);
    float s = 0.f;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) s += p[i];
    const float mean = block_sum_256(s, red) / (float)n;
    float q = 0.f;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) { const float d = p[i] - mean; q = fmaf(d, d, q); }
    const float m2 = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        float *o = ws + (plane * chunks + chunk) * 2;
        o[0] = mean;
        o[1] = m2;
    }
}

__global__ __launch_bounds__(256) void in_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
        int64_t HW, int chunks, float eps, int relu_mid, int relu_out, const float *__restrict__ ws,
        float *__restrict__ y) {
    const int64_t plane = blockIdx.y;
    const int chunk = blockIdx.x;
    // merge the chunk statistics of this plane (every thread redundantly: chunks <= a few dozen, scalar loads)
    const float *w = ws + plane * chunks * 2;
    float mean = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int64_t nb = (int64_t)c * IN_CHUNK;
        const float nc = (float)((nb + IN_CHUNK < HW ? nb + IN_CHUNK : HW) - nb);
        mean += w[2 * c] * nc;
    }
    mean /= (float)HW;
    float m2 = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int64_t nb = (int64_t)c * IN_CHUNK;
        const float nc = (float)((nb + IN_CHUNK < HW ? nb + IN_CHUNK : HW) - nb);
        const float d = w[2 * c] - mean;
        m2 += w[2 * c + 1] + nc * d * d;
    }
    const float rstd = 1.0f / sqrtf(m2 / (float)HW + eps);
    const int64_t beg = (int64_t)chunk * IN_CHUNK;
    const int64_t end = beg + IN_CHUNK < HW ? beg + IN_CHUNK : HW;
    const float *p = x + plane * HW;
    const float *r = res ? res + plane * HW : nullptr;
    float *o = y + plane * HW;
    for (int64_t i = beg + threadIdx.x; i < end; i += 256) {
        float v = (p[i] - mean) * rstd;
        if (relu_mid) v = fmaxf(v, 0.f);
        if (r) v += r[i];
        if (relu_out) v = fmaxf(v, 0.f);
        o[i] = v;
    }
}

extern "C" int nmrf_instance_norm_f32(const float *x, const float *residual, int64_t planes, int64_t HW, float eps,
                                      int relu_mid, int relu_out, float *ws, float *y, void *stream) {
    if (!x || !ws || !y) return NMRF_ENULL;
    if (planes < 1 || planes > 65535 || HW < 1) return NMRF_EINVAL;
    const int chunks = (int)ceil_div64(HW, IN_CHUNK);
    dim3 grid(chunks, (unsigned)planes);
    hipLaunchKernelGGL(in_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, HW, chunks, ws);
    hipLaunchKernelGGL(in_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, residual, HW, chunks, eps, relu_mid,
                       relu_out, ws, y);
    return nmrf_launch_status();
}
