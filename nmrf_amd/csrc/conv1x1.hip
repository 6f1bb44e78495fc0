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
#include "in_affine.h"

#define C1_PIX 128
#define C1_PF 4
#define C1_OLD 132

struct Conv1x1Args {
    const float *x;              // [B, Cx, HW]
    int Cx, c0, K;               // channels of x, first input channel of this conv, its input channels (<= 16 * KC, multiple of 16;
                                 // the weight stream is zero-padded to 16 * KC columns)
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
    int *range_flag;             // sticky fp16-range flag of the split operands (split_mfma.h), may be NULL
    int stride, Wo, Win;         // stride > 1: output pixel (y, x) of a Wo-wide map reads input pixel (y * stride, x * stride) of a Win-wide one
    int64_t HWin;                // pixels of an input plane (== HW when stride == 1)
    int tok;                     // out is token-major [B, HW, N] (a pixel's channels contiguous: the layout the per-token kernels read)
};

template <int KC>                // 16-deep k chunks: 4 (K = 64) or 8 (K = 128)
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(Conv1x1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    float *Aff = reinterpret_cast<float *>(smem + SS_RING_BYTES);           // [2][128]: scale, shift of the block's image
    float *Ot = Aff + 256;                                                   // [32 co][C1_OLD] output staging (one strip: 66 KB of LDS
    //                                                                          per block, TWO blocks per CU -- with a tile per strip PAIR
    //                                                                          it was 83 KB: one block, one wave per SIMD)
    // blockIdx.y = group of output-channel strip pairs (maps with fewer pixel tiles than CUs split their channels over blocks: every
    // block then re-reads the input tile, which is small, and walks 1 / gridDim.y of the weight stream)
    const int pairs_all = (a.N + 63) >> 6, pairs_blk = pairs_all / (int)gridDim.y;
    const int stages_blk = a.total_stages / (int)gridDim.y;
    SplitStream<C1_PF> ss;
    ss.init(reinterpret_cast<const ss_u32x4 *>(a.stream) + (size_t)blockIdx.y * stages_blk * SS_STAGE_U4, smem, stages_blk, tid);
    const int s_first = 2 * pairs_blk * (int)blockIdx.y, n_strips = s_first + 2 * pairs_blk;    // rows beyond N are zero in the stream, never stored
    float guard = 0.f;                                                       // fp16 range guard of the activation splits

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
            if (a.stats) in_affine_of(a.stats + ((size_t)b * a.Cx + a.c0 + tid) * a.chunks * 2, a.chunks, a.HW, a.eps, sc, sh);
            Aff[tid] = sc;
            Aff[128 + tid] = sh;
        }
        __syncthreads();
        // B operand: lane (pixel j, half hi), chunk c, slot jj <-> channel 16c + split_kslot(jj, hi)
        // Addresses as (uniform channel-row pointer) + (32-bit lane offset): the 64 loads of a lane then share ONE offset register
        // (global_load saddr form) instead of 64 precomputed 64-bit addresses, which had pushed the kernel into scratch.
        const float *xu = a.x + ((size_t)b * a.Cx + a.c0) * a.HWin;           // uniform
        int64_t pin = pc;
        if (a.stride > 1) {
            const int64_t yo = pc / a.Wo;
            pin = yo * a.stride * a.Win + (pc - yo * a.Wo) * a.stride;
        }
        const unsigned loff = (unsigned)(pin + (int64_t)(4 * hi) * a.HWin);   // pixel + the lane half's 4 channels
        const float *affl = Aff + 4 * hi;
        h16x8 bh[KC], bl[KC];
        // The 8 loads of a 16-channel chunk used to be followed by a full wait before its split: KC serial memory round trips per
        // tile (and a uniform branch per load for the zero-padded chunks).  Now the loads of four chunks are in flight together,
        // unconditionally (a chunk beyond K reads chunk 0's rows and is zeroed afterwards): two round trips at K = 128, one at 64.
        constexpr int GC = 4;
        static_assert(KC % GC == 0, "chunks are loaded in groups of four");
#pragma unroll
        for (int c0 = 0; c0 < KC; c0 += GC) {
            float v[GC][8];
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                const int cb = 16 * (c0 + c) < a.K ? 16 * (c0 + c) : 0;                                  // uniform
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[c][jj] = (xu + (size_t)(cb + (jj & 3) + 8 * (jj >> 2)) * a.HWin)[loff];
            }
#pragma unroll
            for (int c = 0; c < GC; ++c) {
                const bool live = 16 * (c0 + c) < a.K;                                                   // uniform
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[c][jj] = live ? v[c][jj] : 0.f;
                if (a.stats) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const f32x4 sc = *reinterpret_cast<const f32x4 *>(affl + 16 * (c0 + c) + 8 * g);
                        const f32x4 sh = *reinterpret_cast<const f32x4 *>(affl + 128 + 16 * (c0 + c) + 8 * g);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[c][4 * g + e] = fmaxf(fmaf(v[c][4 * g + e], sc[e], sh[e]), 0.f);
                    }
                }
                split8u_g(v[c], bh[c0 + c], bl[c0 + c], guard);
            }
        }
        // Output: a C/D register is 32 pixels of one channel = a 128-byte piece; written as such (8 strips x 16 pieces per wave,
        // 117 KB apart) the 1/4-resolution gw head ran at ~1 TB/s.  The four waves of the block stage each 32-channel strip in an
        // LDS tile [32 co][128 px] and the block writes it out as 512-byte rows.
        const int64_t pb = (int64_t)(tile - b * a.tiles_per_image) * C1_PIX;   // first pixel of the block
        float *obase = a.out + (size_t)b * a.N * a.HW;
        const bool vec_ok = (a.HW & 3) == 0;
#pragma unroll 1
        for (int s = s_first; s < n_strips; s += 2) {                        // two strips = 2*KC pairs per iteration (whole stages)
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
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = mfma_row(r, hi);
                    Ot[row * C1_OLD + 32 * wv + j] = fmaf(acc[r], a.inv, a.bias ? a.bias[co0 + row] : 0.f);
                }
                __syncthreads();
                // 32 rows (channels) of 128 pixels: thread = (row = tid / 32 + 8 * pass, 4 pixels)
                if (a.tok) {
                    // token-major: thread = (pixel tid / 8 + 32 pass, 4 channels): 8 threads write the strip's 128 bytes of a pixel
                    float *tbase = a.out + (size_t)b * a.HW * a.N;
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        const int px = (tid >> 3) + 32 * pass, cc = (tid & 7) * 4;
                        const int co = co0 + cc;
                        const int64_t pp = pb + px;
                        if (co < a.N && pp < a.HW) {
                            f32x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = Ot[(cc + e) * C1_OLD + px];
                            float *dst = tbase + (size_t)pp * a.N + co;
                            if (co + 3 < a.N && (a.N & 3) == 0) *reinterpret_cast<f32x4 *>(dst) = v;
                            else
                                for (int e = 0; e < 4 && co + e < a.N; ++e) dst[e] = v[e];
                        }
                    }
                } else
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = (tid >> 5) + 8 * pass, px = (tid & 31) * 4;
                    const int co = co0 + row;
                    const int64_t pp = pb + px;
                    if (co < a.N && pp < a.HW) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(Ot + row * C1_OLD + px);
                        float *dst = obase + (size_t)co * a.HW + pp;
                        if (vec_ok && pp + 3 < a.HW) *reinterpret_cast<f32x4 *>(dst) = v;
                        else
                            for (int e = 0; e < 4 && pp + e < a.HW; ++e) dst[e] = v[e];
                    }
                }
                __syncthreads();                                               // the tile is rewritten by the next strip
            });
        }
    }
    split_guard_commit(guard, a.range_flag);
}

template <int KC>
static int launch_conv1x1(const Conv1x1Args &a, hipStream_t st) {
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    static int n_cu_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)SS_RING_BYTES + (256 + 32 * C1_OLD) * sizeof(float);
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
    const int slots = 2 * n_cu_dev[dev];
    const int grid = a.n_tiles < slots ? a.n_tiles : slots;
    // fewer pixel tiles than block slots: split the output channels (strip pairs; whole stages each) over blockIdx.y
    const int pairs = (a.N + 63) >> 6;
    int split = 1;
    while (split * 2 <= pairs && pairs % (split * 2) == 0 && a.n_tiles * split * 2 <= slots &&
           (a.total_stages % (split * 2)) == 0)
        split *= 2;
    hipLaunchKernelGGL((conv1x1_kernel<KC>), dim3(grid, split), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

extern "C" int nmrf_conv1x1_in_relu_f32(const float *x, int B, int Cx, int64_t HW, int c0, int K, const float *stats, int chunks,
                                        float eps, const void *stream_w, int total_stages, float inv_scale, const float *bias, int N,
                                        float *out, int token_major, int *range_flag, void *stream) {
    return nmrf_conv1x1_f32(x, B, Cx, (int)HW, 1, 1, c0, K, stats, chunks, eps, stream_w, total_stages, inv_scale, bias, N, out,
                            token_major, range_flag, stream);
}

extern "C" int nmrf_conv1x1_f32(const float *x, int B, int Cx, int H, int W, int stride, int c0, int K, const float *stats, int chunks,
                                float eps, const void *stream_w, int total_stages, float inv_scale, const float *bias, int N,
                                float *out, int token_major, int *range_flag, void *stream) {
    if (!x || !stream_w || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || stride < 1 || stride > 4 || K < 16 || K > 128 || (K & 15) || c0 < 0 || c0 + K > Cx || N < 1 ||
        (stats && (chunks < 1 || stride != 1)))
        return NMRF_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t HW = (int64_t)Ho * Wo, HWin = (int64_t)H * W;
    if ((int64_t)Cx * HWin >= ((int64_t)1 << 32)) return NMRF_EINVAL;         // 32-bit lane offsets
    const int kc = K <= 64 ? 4 : 8;                                          // k chunks of the (zero-padded) stream
    const int n_pad = (N + 63) / 64 * 64;
    if (total_stages != (n_pad / 32) * kc / 8) return NMRF_EINVAL;
    const int tpi = (int)ceil_div64(HW, C1_PIX);
    if ((int64_t)tpi * B > 0x7fffffff) return NMRF_EINVAL;
    Conv1x1Args a{x, Cx, c0, K, stats, chunks, eps, stream_w, total_stages, bias, out, N, HW, tpi, tpi * B, inv_scale, range_flag,
                  stride, Wo, W, HWin, token_major != 0};
    hipStream_t st = (hipStream_t)stream;
    switch (kc) {
        case 4: return launch_conv1x1<4>(a, st);
        case 8: return launch_conv1x1<8>(a, st);
        default: return NMRF_EINVAL;
    }
}

