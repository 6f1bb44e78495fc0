// InstanceNorm folded into a consumer's operand load: merge the per-chunk (mean, M2) statistics that nmrf_instance_stats_f32
// (norm.hip) wrote for one (sample, channel) plane -- Chan's parallel formula, the same merge as in_apply_kernel -- into
// y = x * scale + shift  with  scale = rstd, shift = -mean * rstd.
#pragma once
#include "common.h"

#define IN_AFFINE_CHUNK 8192          // = IN_CHUNK of norm.hip
#define IN_AFFINE_GROUP 8             // chunk records requested together

// The records of a plane are `chunks` consecutive (mean, M2) pairs.  As two loops of "load, wait, accumulate" this function was
// 2 * chunks SERIAL memory round trips at the head of every convolution workgroup (30 at KITTI 1/2 resolution: most of the
// ~19 us the census had found before the first MFMA).  Now the records are requested in groups of eight -- unconditional 8-byte
// loads, the index clamped to the last record -- and summed from registers in the same order as before: identical bits, four
// round trips at 15 chunks, two at 4.
__device__ __forceinline__ void in_affine_of(const float *w, int chunks, int64_t HW, float eps, float &scale, float &shift) {
    const float2 *w2 = reinterpret_cast<const float2 *>(w);
    const int last = chunks - 1;
    const float n_last = (float)(HW - (int64_t)last * IN_AFFINE_CHUNK);      // pixels of the last chunk; every other one is full
    float mean = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += IN_AFFINE_GROUP) {
        float2 r[IN_AFFINE_GROUP];
#pragma unroll
        for (int i = 0; i < IN_AFFINE_GROUP; ++i) r[i] = w2[c0 + i < last ? c0 + i : last];
#pragma unroll
        for (int i = 0; i < IN_AFFINE_GROUP; ++i)
            if (c0 + i <= last) mean += r[i].x * (c0 + i == last ? n_last : (float)IN_AFFINE_CHUNK);
    }
    mean /= (float)HW;
    float m2 = 0.f;
    for (int c0 = 0; c0 < chunks; c0 += IN_AFFINE_GROUP) {
        float2 r[IN_AFFINE_GROUP];
#pragma unroll
        for (int i = 0; i < IN_AFFINE_GROUP; ++i) r[i] = w2[c0 + i < last ? c0 + i : last];
#pragma unroll
        for (int i = 0; i < IN_AFFINE_GROUP; ++i)
            if (c0 + i <= last) {
                const float nc = c0 + i == last ? n_last : (float)IN_AFFINE_CHUNK;
                const float d = r[i].x - mean;
                m2 += r[i].y + d * d * nc;
            }
    }
    scale = 1.0f / sqrtf(m2 / (float)HW + eps);
    shift = -mean * scale;
}
