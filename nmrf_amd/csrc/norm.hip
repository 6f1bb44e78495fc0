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

// A chunk is read ONCE: up to 8 float4 per thread stay in registers between the mean pass and the centred
// sum-of-squares pass (the first version re-read the chunk from L2).  Planes whose byte offset is not a multiple of 16
// (odd HW, e.g. the 94x311 quarter-resolution KITTI plane) are handled by peeling up to 3 head / 3 tail elements, so the
// body is always 16-byte aligned.
struct ChunkView {
    const float *p;      // first element of the chunk
    int n, head, nvec;   // elements, scalar head elements, float4 body elements
};
__device__ __forceinline__ ChunkView chunk_view(const float *plane, int64_t HW, int chunk) {
    const int64_t beg = (int64_t)chunk * IN_CHUNK;
    const int64_t end = beg + IN_CHUNK < HW ? beg + IN_CHUNK : HW;
    ChunkView v;
    v.p = plane + beg;
    v.n = (int)(end - beg);
    const int mis = (int)(((uintptr_t)v.p >> 2) & 3);
    v.head = mis ? 4 - mis : 0;
    if (v.head > v.n) v.head = v.n;
    v.nvec = (v.n - v.head) >> 2;
    return v;
}

__global__ __launch_bounds__(256) void in_stats_kernel(const float *__restrict__ x, int64_t HW, int chunks,
                                                      float *__restrict__ ws) {
    __shared__ float red[4];
    const int64_t plane = blockIdx.y;
    const int chunk = blockIdx.x;
    const ChunkView cv = chunk_view(x + plane * HW, HW, chunk);
    const float4 *pv = reinterpret_cast<const float4 *>(cv.p + cv.head);
    float4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = threadIdx.x + 256 * k;
        v[k] = i < cv.nvec ? pv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // scalar edge elements: head (< 4) then tail (< 4), one per thread
    const int tail0 = cv.head + 4 * cv.nvec, n_edge = cv.head + (cv.n - tail0);
    const bool has_e = (int)threadIdx.x < n_edge;
    const float ev = has_e ? cv.p[(int)threadIdx.x < cv.head ? (int)threadIdx.x : tail0 + ((int)threadIdx.x - cv.head)] : 0.f;
    float s = ev;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float mean = block_sum_256(s, red) / (float)cv.n;
    float q = 0.f;
    if (has_e) { const float d = ev - mean; q = d * d; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if ((int)threadIdx.x + 256 * k < cv.nvec) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            q = fmaf(dx, dx, q); q = fmaf(dy, dy, q); q = fmaf(dz, dz, q); q = fmaf(dw, dw, q);
        }
    }
    const float m2 = block_sum_256(q, red);
    if (threadIdx.x == 0) {
        float *o = ws + (plane * chunks + chunk) * 2;
        o[0] = mean;
        o[1] = m2;
    }
}

// merge the chunk statistics of one plane (every thread redundantly: chunks <= a few dozen, scalar loads)
__device__ __forceinline__ void in_merge_chunks(const float *w, int chunks, int64_t HW, float eps, float &mean, float &rstd) {
    mean = 0.f;
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
    rstd = 1.0f / sqrtf(m2 / (float)HW + eps);
}

// res_ws != NULL: the residual operand is itself a RAW convolution output whose InstanceNorm (+ ReLU if res_relu) has not been
// applied yet -- the stem output entering layer1.0, the 1x1 shortcut of layer2.0 / layer3.0 (nmrf/models/backbone.py:40-46, 85):
// it is normalised on the way in, so that its own apply pass (a read and a write of the whole map) never runs.
__global__ __launch_bounds__(256) void in_apply_kernel(const float *__restrict__ x, const float *__restrict__ res,
        int64_t HW, int chunks, float eps, int relu_mid, int relu_out, const float *__restrict__ ws,
        float *__restrict__ y, const float *__restrict__ res_ws = nullptr, int res_relu = 0) {
    const int64_t plane = blockIdx.y;
    const int chunk = blockIdx.x;
    float mean, rstd;
    in_merge_chunks(ws + plane * chunks * 2, chunks, HW, eps, mean, rstd);
    float rmean = 0.f, rrstd = 1.f;
    const bool rnorm = res && res_ws;
    if (rnorm) in_merge_chunks(res_ws + plane * chunks * 2, chunks, HW, eps, rmean, rrstd);
    const ChunkView cv = chunk_view(x + plane * HW, HW, chunk);
    const int64_t off = cv.p - x;                               // same element offset (and alignment) in res / y
    const float *r = res ? res + off : nullptr;
    float *o = y + off;
    auto f = [&](float xv, float rv) {
        float t = (xv - mean) * rstd;
        if (relu_mid) t = fmaxf(t, 0.f);
        if (rnorm) {
            rv = (rv - rmean) * rrstd;
            if (res_relu) rv = fmaxf(rv, 0.f);
        }
        t += rv;
        if (relu_out) t = fmaxf(t, 0.f);
        return t;
    };
    const float4 *pv = reinterpret_cast<const float4 *>(cv.p + cv.head);
    const float4 *rv = reinterpret_cast<const float4 *>(r ? r + cv.head : nullptr);
    float4 *ov = reinterpret_cast<float4 *>(o + cv.head);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = threadIdx.x + 256 * k;
        if (i < cv.nvec) {
            const float4 a = pv[i];
            const float4 b = r ? rv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            ov[i] = make_float4(f(a.x, b.x), f(a.y, b.y), f(a.z, b.z), f(a.w, b.w));
        }
    }
    const int tail0 = cv.head + 4 * cv.nvec, n_edge = cv.head + (cv.n - tail0);
    if ((int)threadIdx.x < n_edge) {
        const int i = (int)threadIdx.x < cv.head ? (int)threadIdx.x : tail0 + ((int)threadIdx.x - cv.head);
        o[i] = f(cv.p[i], r ? r[i] : 0.f);
    }
}

// apply pass alone, statistics given: y = [relu_out]( [relu_mid] IN(x; ws) + f(residual) ),  f = identity, or IN(.; res_ws) [+ ReLU]
extern "C" int nmrf_instance_apply_f32(const float *x, const float *ws, const float *residual, const float *res_ws, int res_relu,
                                       int64_t planes, int64_t HW, float eps, int relu_mid, int relu_out, float *y, void *stream) {
    if (!x || !ws || !y) return NMRF_ENULL;
    if (res_ws && !residual) return NMRF_ENULL;
    if (planes < 1 || planes > 65535 || HW < 1) return NMRF_EINVAL;
    if ((((uintptr_t)x ^ (uintptr_t)y) & 15) || (residual && (((uintptr_t)x ^ (uintptr_t)residual) & 15))) return NMRF_EINVAL;
    const int chunks = (int)ceil_div64(HW, IN_CHUNK);
    hipLaunchKernelGGL(in_apply_kernel, dim3(chunks, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, residual, HW, chunks, eps,
                       relu_mid, relu_out, ws, y, res_ws, res_relu);
    return nmrf_launch_status();
}

// statistics pass alone: ws [planes][ceil(HW / 8192)][2] = per-chunk (mean, M2), merged by the consumer (in_apply_kernel or the
// fused 1x1 convolution of conv1x1.hip)
extern "C" int nmrf_instance_stats_f32(const float *x, int64_t planes, int64_t HW, float *ws, void *stream) {
    if (!x || !ws) return NMRF_ENULL;
    if (planes < 1 || planes > 65535 || HW < 1) return NMRF_EINVAL;
    const int chunks = (int)ceil_div64(HW, IN_CHUNK);
    hipLaunchKernelGGL(in_stats_kernel, dim3(chunks, (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, HW, chunks, ws);
    return nmrf_launch_status();
}

extern "C" int nmrf_instance_norm_f32(const float *x, const float *residual, int64_t planes, int64_t HW, float eps,
                                      int relu_mid, int relu_out, float *ws, float *y, void *stream) {
    if (!x || !ws || !y) return NMRF_ENULL;
    if (planes < 1 || planes > 65535 || HW < 1) return NMRF_EINVAL;
    // x, residual and y must share their 16-byte phase (torch allocations are 256-byte aligned; views may not be)
    if ((((uintptr_t)x ^ (uintptr_t)y) & 15) || (residual && (((uintptr_t)x ^ (uintptr_t)residual) & 15))) return NMRF_EINVAL;
    const int chunks = (int)ceil_div64(HW, IN_CHUNK);
    dim3 grid(chunks, (unsigned)planes);
    hipLaunchKernelGGL(in_stats_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, HW, chunks, ws);
    hipLaunchKernelGGL(in_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, residual, HW, chunks, eps, relu_mid,
                       relu_out, ws, y, (const float *)nullptr, 0);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Input staging of the CNN encoder in one pass: replicate-pad right/bottom to (Hp, Wp) (InputPadder mode 'proposal',
// nmrf/utils/frame_utils.py:268-275), stack the two views along the batch (NMRF.py:173) and normalise
// 2 * (x / 255) - 1 (nmrf/models/backbone.py:86), same fp32 operation order.  Replaces two replication pads, a cat and three
// elementwise kernels.
// ------------------------------------------------------------------------------------------------------------------
// PX = float (the reference's sample['img1'] is float 0..255) or uint8_t (decoded images as they come off the disk: the driver
// moves them over PCIe as bytes -- a quarter of the traffic -- and the conversion (float)px is exact).
template <typename PX>
__global__ __launch_bounds__(256) void prep_images_kernel(const PX *__restrict__ img1, const PX *__restrict__ img2, int B,
                                                         int C, int H, int W, int Hp, int Wp, float *__restrict__ out) {
    const int64_t total = (int64_t)2 * B * C * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % Wp);
        const int y = (int)((i / Wp) % Hp);
        const int64_t pc = i / ((int64_t)Wp * Hp);                 // (view * B + b) * C + c
        const int64_t v = pc / ((int64_t)B * C), bc = pc - v * (int64_t)B * C;
        const PX *src = (v == 0 ? img1 : img2) + bc * (int64_t)H * W;
        const float px = (float)src[(int64_t)(y < H ? y : H - 1) * W + (x < W ? x : W - 1)];
        out[i] = 2.0f * (px / 255.0f) - 1.0f;
    }
}

template <typename PX>
static int launch_prep_images(const PX *img1, const PX *img2, int B, int C, int H, int W, int Hp, int Wp, float *out, void *stream) {
    if (!img1 || !img2 || !out) return NMRF_ENULL;
    if (B < 1 || C < 1 || H < 1 || W < 1 || Hp < H || Wp < W) return NMRF_EINVAL;
    int64_t blocks = ceil_div64((int64_t)2 * B * C * Hp * Wp, 256 * 4);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(prep_images_kernel<PX>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, img1, img2, B, C, H, W, Hp,
                       Wp, out);
    return nmrf_launch_status();
}

extern "C" int nmrf_prep_images_f32(const float *img1, const float *img2, int B, int C, int H, int W, int Hp, int Wp, float *out,
                                    void *stream) {
    return launch_prep_images(img1, img2, B, C, H, W, Hp, Wp, out, stream);
}

extern "C" int nmrf_prep_images_u8(const uint8_t *img1, const uint8_t *img2, int B, int C, int H, int W, int Hp, int Wp, float *out,
                                   void *stream) {
    return launch_prep_images(img1, img2, B, C, H, W, Hp, Wp, out, stream);
}

// Same staging, written as the 2x2 space-to-depth image the stem convolution consumes (conv3x3.hip, KT = 4): out [2B, 16, Hp/2, Wp/2],
// channel c*4 + p*2 + q = padded / normalised pixel (2Y + p, 2X + q) of colour c (C == 3), channels 12..15 zero.
template <typename PX>
__global__ __launch_bounds__(256) void prep_images_s2d_kernel(const PX *__restrict__ img1, const PX *__restrict__ img2, int B,
                                                             int H, int W, int H2, int W2, float *__restrict__ out) {
    // thread = one output cell (Y, X) of image blockIdx.y = view * B + b: its 12 source pixels are requested together (clamped
    // addresses = the replicate padding) and go out as 16 coalesced plane stores.  (As a flat grid-stride loop over output
    // elements -- three 64-bit divisions and one exposed load per element -- this pass took 15 us at KITTI.)
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= H2 * W2) return;
    const int Y = cell / W2, X = cell - Y * W2;
    const int vb = blockIdx.y, v = vb / B, bb = vb - v * B;
    const PX *src = (v == 0 ? img1 : img2) + (size_t)bb * 3 * H * W;
    const int y0 = min(2 * Y, H - 1), y1 = min(2 * Y + 1, H - 1), xa = min(2 * X, W - 1), xb = min(2 * X + 1, W - 1);
    PX px[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const PX *pl = src + (size_t)c * H * W;
        px[4 * c + 0] = pl[(size_t)y0 * W + xa];
        px[4 * c + 1] = pl[(size_t)y0 * W + xb];
        px[4 * c + 2] = pl[(size_t)y1 * W + xa];
        px[4 * c + 3] = pl[(size_t)y1 * W + xb];
    }
    float *dst = out + (size_t)vb * 16 * H2 * W2 + cell;
#pragma unroll
    for (int ch = 0; ch < 12; ++ch) dst[(size_t)ch * H2 * W2] = 2.0f * ((float)px[ch] / 255.0f) - 1.0f;
#pragma unroll
    for (int ch = 12; ch < 16; ++ch) dst[(size_t)ch * H2 * W2] = 0.f;
}

template <typename PX>
static int launch_prep_images_s2d(const PX *img1, const PX *img2, int B, int H, int W, int Hp, int Wp, float *out, void *stream) {
    if (!img1 || !img2 || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || Hp < H || Wp < W || (Hp & 1) || (Wp & 1) || 2 * B > 65535) return NMRF_EINVAL;
    const int64_t cells = (int64_t)(Hp / 2) * (Wp / 2);
    if (cells > (int64_t)1 << 30) return NMRF_EINVAL;
    hipLaunchKernelGGL(prep_images_s2d_kernel<PX>, dim3((unsigned)ceil_div64(cells, 256), 2 * B), dim3(256), 0, (hipStream_t)stream,
                       img1, img2, B, H, W, Hp / 2, Wp / 2, out);
    return nmrf_launch_status();
}

extern "C" int nmrf_prep_images_s2d_f32(const float *img1, const float *img2, int B, int H, int W, int Hp, int Wp, float *out,
                                        void *stream) {
    return launch_prep_images_s2d(img1, img2, B, H, W, Hp, Wp, out, stream);
}

extern "C" int nmrf_prep_images_s2d_u8(const uint8_t *img1, const uint8_t *img2, int B, int H, int W, int Hp, int Wp, float *out,
                                       void *stream) {
    return launch_prep_images_s2d(img1, img2, B, H, W, Hp, Wp, out, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Tail of the CNN encoder: y = conv1x1 output without its bias [BC planes of H x W]; x = y + bias[c] (the 1/4-resolution map)
// and its 2x2 average (the 1/8 map, nmrf/models/backbone.py:96-98) in one pass over y.  H, W even; thread = one 2x2 cell.
// x == NULL: only the average is written (the producer already added its bias: 75 MB instead of 135 MB at KITTI).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_avgpool2_kernel(const float *__restrict__ y, const float *__restrict__ bias, int Cc,
                                                           int H, int W, int64_t cells, float *__restrict__ x,
                                                           float *__restrict__ pooled) {
    const int W2 = W >> 1, H2 = H >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (int64_t)gridDim.x * 256) {
        const int cx = (int)(i % W2), cy = (int)((i / W2) % H2);
        const int64_t plane = i / ((int64_t)W2 * H2);
        const float bv = bias ? bias[plane % Cc] : 0.f;
        const int64_t o = plane * (int64_t)H * W + (int64_t)(2 * cy) * W + 2 * cx;
        const float2 t = *reinterpret_cast<const float2 *>(y + o), u = *reinterpret_cast<const float2 *>(y + o + W);
        const float a = t.x + bv, b = t.y + bv, c = u.x + bv, d = u.y + bv;
        if (x) {
            *reinterpret_cast<float2 *>(x + o) = make_float2(a, b);
            *reinterpret_cast<float2 *>(x + o + W) = make_float2(c, d);
        }
        pooled[i] = (((a + b) + c) + d) * 0.25f;          // ATen's avg_pool2d sums the window row by row, then divides
    }
}

extern "C" int nmrf_bias_avgpool2_f32(const float *y, const float *bias, int64_t planes, int C, int H, int W, float *x, float *pooled,
                                      void *stream) {
    if (!y || !pooled) return NMRF_ENULL;                          // x NULL: only the pooled map is written
    if (planes < 1 || C < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return NMRF_EINVAL;
    const int64_t cells = planes * (int64_t)(H / 2) * (W / 2);
    int64_t blocks = ceil_div64(cells, 256 * 2);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(bias_avgpool2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, C, H, W, cells, x,
                       pooled);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Host helper of the batched driver (N1): copy into a PINNED staging buffer with non-temporal stores.  A plain memcpy leaves the
// last ~1-2 MB it wrote as dirty lines in the writing core's private cache, and the H2D DMA that follows has to snoop every one
// of them out: measured on the MI355X host, ~20 ms per batch regardless of its size (31 pairs/s instead of 270 at KITTI
// batch 1, tools/driver_probe3.py).  Streaming stores go to memory through the write-combining buffers and leave nothing behind.
// ------------------------------------------------------------------------------------------------------------------
// (The MI355X hosts of this build are x86-64; on any other host architecture both helpers are plain memcpy -- correct, minus the
// cache hygiene.)
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
extern "C" int nmrf_host_copy_nt(void *dst, const void *src, size_t bytes) {
    if (!dst || !src) return NMRF_ENULL;
#if !defined(__x86_64__)
    memcpy(dst, src, bytes);
    return NMRF_OK;
#else
    unsigned char *d = static_cast<unsigned char *>(dst);
    const unsigned char *s = static_cast<const unsigned char *>(src);
    size_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    if (head > bytes) head = bytes;
    memcpy(d, s, head);
    d += head; s += head; bytes -= head;
    const size_t n16 = bytes / 16;
    for (size_t i = 0; i < n16; ++i)
        _mm_stream_si128(reinterpret_cast<__m128i *>(d) + i, _mm_loadu_si128(reinterpret_cast<const __m128i *>(s) + i));
    memcpy(d + 16 * n16, s + 16 * n16, bytes - 16 * n16);
    if (head || bytes != 16 * n16) {                         // the few bytes that went through the cache: push them out too
        _mm_clflush(static_cast<unsigned char *>(dst));
        _mm_clflush(d + bytes - 1);
    }
    _mm_sfence();
    return NMRF_OK;
#endif
}

// The other direction: copy a finished result OUT of a pinned buffer and evict the lines the read pulled into the CPU cache, so
// that the next D2H DMA into the same buffer does not have to invalidate them one by one (the same ~20 ms per batch).
extern "C" int nmrf_host_read_evict(void *dst, const void *src, size_t bytes) {
    if (!dst || !src) return NMRF_ENULL;
    memcpy(dst, src, bytes);
#if defined(__x86_64__)
    const unsigned char *s = static_cast<const unsigned char *>(src);
    const uintptr_t first = reinterpret_cast<uintptr_t>(s) & ~(uintptr_t)63, last = reinterpret_cast<uintptr_t>(s) + bytes;
    for (uintptr_t a = first; a < last; a += 64) _mm_clflush(reinterpret_cast<const void *>(a));
    _mm_sfence();
#endif
    return NMRF_OK;
}
