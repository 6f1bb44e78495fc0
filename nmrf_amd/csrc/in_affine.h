// InstanceNorm folded into a consumer's operand load: merge the per-chunk (mean, M2) statistics that nmrf_instance_stats_f32
// (norm.hip) wrote for one (sample, channel) plane -- Chan's parallel formula, the same merge as in_apply_kernel -- into
// y = x * scale + shift  with  scale = rstd, shift = -mean * rstd.
#pragma once
#include "common.h"

#define IN_AFFINE_CHUNK 8192          // = IN_CHUNK of norm.hip

__device__ __forceinline__ void in_affine_of(const float *w, int chunks, int64_t HW, float eps, float &scale, float &shift) {
    float mean = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int64_t nb = (int64_t)c * IN_AFFINE_CHUNK;
        mean += w[2 * c] * (float)((nb + IN_AFFINE_CHUNK < HW ? nb + IN_AFFINE_CHUNK : HW) - nb);
    }
    mean /= (float)HW;
    float m2 = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int64_t nb = (int64_t)c * IN_AFFINE_CHUNK;
        const float nc = (float)((nb + IN_AFFINE_CHUNK < HW ? nb + IN_AFFINE_CHUNK : HW) - nb);
        const float d = w[2 * c] - mean;
        m2 += w[2 * c + 1] + d * d * nc;
    }
    scale = 1.0f / sqrtf(m2 / (float)HW + eps);
    shift = -mean * scale;
}
