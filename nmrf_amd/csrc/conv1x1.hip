// N2: the 1x1 convolutions of the conv heads with their InstanceNorm + ReLU folded in
//
//     out[b, co, p] = sum_ci W[co, ci] * f(x[b, c0 + ci, p]) (+ bias[co]),   f(v) = relu((v - mean[b,c]) * rstd[b,c])  or  v
//
// replacing  in_apply (read + write of the 3x3 conv output) -> channel-slice copy -> MIOpen 1x1 conv  of
// concatconv / gw / dpn.proj (nmrf/models/NMRF.py:56-65, 211-214, 233-236; DPN.py:45-49): the normalised activation exists only
// as an MFMA operand.  NCHW in, NCHW out, no layout change: with the transposed formulation of nmp_block.hip (weights = A operand
// from the shared LDS stream, activations = B operand with the PIXEL on the lane) a channel row of 32 consecutive pixels is one
// 128-byte line on the way in, and a C/D register is one 128-byte line of an output channel on the way out.
// Block = 4 waves x 32 pixels of one image.  Split-operand fp16 MFMA (split_mfma.h).
#include "split_stream.h"

#define C1_PIX 128
#define C1_PF 4

struct Conv1x1Args {
    const float *x;              // [B, Cx, HW]
    int Cx, c0, K;               // channels of x, first input channel of this conv, its input channels (<= 128, multiple of 16)
    const float *stats;          // in_stats workspace of x ([B*Cx][chunks][2] = per-chunk mean, M2) or NULL (no norm, no ReLU)
    int chunks;
    float eps;
    const void *stream;
    int total_stages;
    const float *bias;           // [N] or NULL
    float *out;                  // [B, N, HW]
    int N;
    int64_t HW;
    int tiles_per_image, n_tiles;
    float inv;
};

template <int KC>                // 16-deep k chunks: 4 (K = 64) or 8 (K = 128)
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(Conv1x1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    float *Aff = reinterpret_cast<float *>(smem + SS_RING_BYTES);           // [2][128]: scale, shift of the block's image
    SplitStream<C1_PF> ss;
    ss.init(a.stream, smem, a.total_stages, tid);
    const int n_strips = a.N >> 5;

#pragma unroll 1
    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int b = tile / a.tiles_per_image;
        const int64_t p0 = (int64_t)(tile - b * a.tiles_per_image) * C1_PIX + wv * 32;
        const int64_t p = p0 + j;
        const int64_t pc = p < a.HW ? p : a.HW - 1;
        // per-channel affine of this image: merge the chunk statistics (Chan), scale = rstd, shift = -mean * rstd
        __syncthreads();                                                       // previous tile is done with Aff
        if (tid < a.K) {
            float sc = 1.f, sh = 0.f;
            if (a.stats) {
                const float *w = a.stats + ((size_t)b * a.Cx + a.c0 + tid) * a.chunks * 2;
                float mean = 0.f;
                for (int c = 0; c < a.chunks; ++c) {
                    const int64_t nb = (int64_t)c * 8192;
                    mean += w[2 * c] * (float)((nb + 8192 < a.HW ? nb + 8192 : a.HW) - nb);
                }
                mean /= (float)a.HW;
                float m2 = 0.f;
                for (int c = 0; c < a.chunks; ++c) {
                    const int64_t nb = (int64_t)c * 8192;
                    const float nc = (float)((nb + 8192 < a.HW ? nb + 8192 : a.HW) - nb);
                    const float d = w[2 * c] - mean;
                    m2 += w[2 * c + 1] + d * d * nc;
                }
                sc = 1.0f / sqrtf(m2 / (float)a.HW + a.eps);
                sh = -mean * sc;
            }
            Aff[tid] = sc;
            Aff[128 + tid] = sh;
        }
        __syncthreads();
        // B operand: lane (pixel j, half hi), chunk c, slot jj <-> channel 16c + split_kslot(jj, hi)
        const float *xb = a.x + ((size_t)b * a.Cx + a.c0) * a.HW + pc;
        h16x8 bh[KC], bl[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            float v[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[jj] = xb[(size_t)(16 * c + split_kslot(jj, hi)) * a.HW];
            if (a.stats) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int ch = 16 * c + split_kslot(jj, hi);
                    v[jj] = fmaxf(fmaf(v[jj], Aff[ch], Aff[128 + ch]), 0.f);
                }
            }
            split8u(v, bh[c], bl[c]);
        }
        float *ob = a.out + (size_t)b * a.N * a.HW + p;
#pragma unroll 1
        for (int s = 0; s < n_strips; s += 2) {                              // two strips = 2*KC pairs per iteration (KC even: whole stages)
            ss_static_for<2>([&](auto hh) {
                constexpr int half = decltype(hh)::value;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                ss_static_for<KC>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    ss_pair<half * KC + c>(ss, bh[c], bl[c], acc);
                });
                const int co0 = 32 * (s + half);
                if (p < a.HW && co0 < a.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + mfma_row(r, hi);
                        ob[(size_t)co * a.HW] = fmaf(acc[r], a.inv, a.bias ? a.bias[co] : 0.f);
                    }
                }
            });
        }
    }
}

template <int KC>
static int launch_conv1x1(const Conv1x1Args &a, hipStream_t st) {
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)SS_RING_BYTES + 256 * sizeof(float);
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_kernel<KC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    if (!n_cu_dev[dev]) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMRF_ELAUNCH;
        n_cu_dev[dev] = prop.multiProcessorCount;
    }
    const int grid = a.n_tiles < 2 * n_cu_dev[dev] ? a.n_tiles : 2 * n_cu_dev[dev];
    hipLaunchKernelGGL((conv1x1_kernel<KC>), dim3(grid), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

extern "C" int nmrf_conv1x1_in_relu_f32(const float *x, int B, int Cx, int64_t HW, int c0, int K, const float *stats, int chunks,
                                        float eps, const void *stream_w, int total_stages, float inv_scale, const float *bias, int N,
                                        float *out, void *stream) {
    if (!x || !stream_w || !out) return NMRF_ENULL;
    if (B < 1 || HW < 1 || (K != 64 && K != 128) || c0 < 0 || c0 + K > Cx || N < 64 || (N & 63) || (stats && chunks < 1))
        return NMRF_EINVAL;
    const int kc = K / 16;
    if (total_stages != (N / 32) * kc / 8) return NMRF_EINVAL;
    const int tpi = (int)ceil_div64(HW, C1_PIX);
    if ((int64_t)tpi * B > 0x7fffffff) return NMRF_EINVAL;
    Conv1x1Args a{x, Cx, c0, K, stats, chunks, eps, stream_w, total_stages, bias, out, N, HW, tpi, tpi * B, inv_scale};
    hipStream_t st = (hipStream_t)stream;
    switch (kc) {
        case 4: return launch_conv1x1<4>(a, st);
        case 8: return launch_conv1x1<8>(a, st);
        default: return NMRF_EINVAL;
    }
}

