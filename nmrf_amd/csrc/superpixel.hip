// A16: superpixel-guided disparity downsample (eval-only; call site nmrf/utils/evaluation.py:361-378).
// PARITY UNPINNED: the reference source of this operator is absent from the snapshot; semantics reconstructed from the
// call site and fixed in oracle/superpixel_oracle.py (mode = mean of the valid disparities of one superpixel segment
// inside an 8x8 cell, modes ordered by pixel count descending then label ascending, first K kept, 0 = empty slot).
//
// One wave = one 8x8 cell, lane = pixel (row-major).  Segment sums are built by 64 broadcast steps (v_readlane), in pixel
// order, so the fp32 mean is bit-identical to the sequential CPU restatement.
#include "common.h"

__global__ __launch_bounds__(256) void superpixel_downsample_kernel(const float *__restrict__ disp,
        const int *__restrict__ labels, int H, int W, int K, int64_t n_cells, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= n_cells) return;                                    // whole wave
    const int wd = W >> 3, ht = H >> 3;
    const int cx = (int)(cell % wd), cy = (int)((cell / wd) % ht), b = (int)(cell / ((int64_t)wd * ht));
    const size_t pix = ((size_t)b * H + 8 * cy + (lane >> 3)) * W + 8 * cx + (lane & 7);
    const float d = disp[pix];
    const int l = labels[pix];
    const bool valid = d > 0.f;
    // per lane: statistics of ITS segment (sum in pixel order), and whether it is the segment's first valid pixel
    float sum = 0.f;
    int cnt = 0;
    bool leader = valid;
    for (int j = 0; j < 64; ++j) {
        const int lj = __builtin_amdgcn_readlane(l, j);
        const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), j));
        const bool vj = dj > 0.f;
        if (vj && lj == l) {
            sum += dj;
            cnt += 1;
            if (j < lane) leader = false;
        }
    }
    // rank of a leader among leaders: (count desc, label asc)
    int rank = 0;
    for (int j = 0; j < 64; ++j) {
        const int lj = __builtin_amdgcn_readlane(l, j);
        const int cj = __builtin_amdgcn_readlane(cnt, j);
        const bool leadj = __builtin_amdgcn_readlane((int)leader, j) != 0;
        if (leadj && (cj > cnt || (cj == cnt && lj < l))) rank += 1;
    }
    float *o = out + cell * K;
    const int n_groups = __builtin_popcountll(__ballot(leader));
    if (lane < K && lane >= n_groups) o[lane] = 0.f;               // empty slots
    if (leader && rank < K) o[rank] = sum / (float)cnt;
}

extern "C" int nmrf_superpixel_downsample_f32(const float *disp, const int *labels, int B, int H, int W, int K, float *out,
                                              void *stream) {
    if (!disp || !labels || !out) return NMRF_ENULL;
    if (B < 1 || H < 8 || W < 8 || K < 1 || K > 64) return NMRF_EINVAL;
    const int64_t n_cells = (int64_t)B * (H >> 3) * (W >> 3);
    hipLaunchKernelGGL(superpixel_downsample_kernel, dim3((unsigned)ceil_div64(n_cells, 4)), dim3(256), 0, (hipStream_t)stream,
                       disp, labels, H, W, K, n_cells, out);
    return nmrf_launch_status();
}
