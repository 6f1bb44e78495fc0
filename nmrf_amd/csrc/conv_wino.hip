// N2: 3x3 / stride 1 / pad 1 convolution of the stock conv band (backbone residual blocks
// nmrf/models/backbone.py:38-46, conv heads NMRF.py:56-65, DPN.py:45-49) as a fused Winograd F(2x2,3x3) on fp32 MFMA.
//
//   V = B^T d B   (4x4 input patch d of a 2x2 output tile),   U = G g G^T   (3x3 filter g, packed once on the host side),
//   M_p[tile][co] = sum_ci V_p[tile][ci] U_p[ci][co]   for each of the 16 positions p,      Y = A^T M A   (2x2 outputs)
//
// 16 skinny GEMMs with 2.25x fewer multiplies than the direct form; MIOpen's kernel for these shapes is the gfx9 VALU
// Winograd (88-105 TFLOP/s counted as direct-conv FLOPs, tools/conv_probe.py).
//
// Block = 8 waves = 64 tiles (2 tile rows x 32 tile columns) x 32 output channels.  A wave owns 16 tiles of one tile row
// and 16 output channels for ALL 16 positions (v_mfma_f32_16x16x4_f32, 16 accumulators of 4 registers), so the inverse
// transform needs no exchange.  Input channels are walked in chunks of 8: the raw 6 x 66 input patch of the chunk and its
// 16 x 8 x 32 filter slab sit in one of two LDS buffers (fetched into registers during the previous chunks' MFMAs); each
// lane transforms its own tile for its 2 channels into the A operands on the fly.
//   A operand  lane l : V_p[tile = l%16][ci = 8*chunk + 2*(l/16) + m]           (one MFMA per m = 0, 1)
//   B operand  lane l : U_p[ci (same)][co = 16*strip + l%16]
//   D          lane l, reg r : M_p[tile = 4*(l/16) + r][co = l%16]
#include "common.h"
#include "../../include/nmrf_hip_debug.h"      // tools / test build only (not in libnmrf_hip.so)

#define WN_CK 8                  // input channels per chunk
#define WN_RS 68                 // raw patch row stride (66 columns used)
#define WN_RAW (WN_CK * 6 * WN_RS)
#define WN_U (16 * WN_CK * 32)   // filter slab of a chunk: 16 positions x 8 ci x 32 co
#define WN_BUF (WN_RAW + WN_U)    // one LDS buffer (floats)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// filter [Co, Ci, 3, 3] -> U in fragment order [Ci/8][Co/32][m 2][pg 4][strip 2][lane 64][4 positions]
__global__ __launch_bounds__(256) void wino_pack_filter_kernel(const float *__restrict__ w, int Co, int Ci,
                                                              float *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one float4 = 4 positions of one (co, ci)
    const int64_t total = (int64_t)(Ci / 8) * (Co / 32) * 2 * 4 * 2 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63), cs = (int)((idx >> 6) & 1), pg = (int)((idx >> 7) & 3), m = (int)((idx >> 9) & 1);
    const int64_t slab = idx >> 10;
    const int cb = (int)(slab % (Co / 32)), kc = (int)(slab / (Co / 32));
    const int co = 32 * cb + 16 * cs + (lane & 15), ci = 8 * kc + 2 * (lane >> 4) + m;
    const float *g = w + ((int64_t)co * Ci + ci) * 9;
    // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; row i = pg of U
    float t[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        t[0][c] = g[c];
        t[1][c] = 0.5f * (g[c] + g[3 + c] + g[6 + c]);
        t[2][c] = 0.5f * (g[c] - g[3 + c] + g[6 + c]);
        t[3][c] = g[6 + c];
    }
    const float *r = t[pg];
    stg4(packed + idx * 4, make_float4(r[0], 0.5f * (r[0] + r[1] + r[2]), 0.5f * (r[0] - r[1] + r[2]), r[2]));
}

extern "C" int nmrf_wino_pack_filter_f32(const float *w, int Co, int Ci, float *packed, void *stream) {
    if (!w || !packed) return NMRF_ENULL;
    if (Co < 32 || (Co & 31) || Ci < 8 || (Ci & 7)) return NMRF_EINVAL;
    const int64_t total = (int64_t)(Ci / 8) * (Co / 32) * 2 * 4 * 2 * 64;
    hipLaunchKernelGGL(wino_pack_filter_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, Co,
                       Ci, packed);
    return nmrf_launch_status();
}

__global__ __launch_bounds__(512, 4) void conv3x3_wino_kernel(const float *__restrict__ x, const float *__restrict__ up,
                                                              int Ci, int H, int W, int Co, float *__restrict__ y,
                                                              unsigned long long *__restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // raw[16][6][68] | U slab[8192]  (the output tile reuses it)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tg = wv & 3, cs = wv >> 2;                            // tile segment (row tg>>1, columns 16*(tg&1)..), co strip
    const int n_cb = Co >> 5;
    // XCD-aware block order (workgroups go round-robin over the 8 XCDs, each with its own L2): the Co/32 blocks of one tile
    // block read the same input patches, so logical items are numbered output-channel-block fastest and every XCD gets a
    // contiguous run of them (with channel blocks as the slowest grid dimension the patches came from MALL/HBM each time:
    // 229 MB fetched per launch for ~40 MB of input).
    const int gx = (((W + 1) >> 1) + 31) >> 5, gy = (((H + 1) >> 1) + 1) >> 1;
    const int per_xcd = gridDim.x >> 3;                             // the launch pads the grid to a multiple of 8
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= n_cb * gx * gy) return;
    const int cb = item % n_cb, bxi = (item / n_cb) % gx, byi = item / (n_cb * gx);
    const int b = blockIdx.y;
    const int ty0 = byi * 2, tx0 = bxi * 32;                        // first tile row / column of the block
    const int y0 = 2 * ty0 - 1, x0 = 2 * tx0 - 1;                   // top-left input pixel of the raw patch
    const size_t plane = (size_t)H * W;
    const float *xb = x + (size_t)b * Ci * plane;
    const int n_chunks = Ci / WN_CK;
    const int lin_blk = blockIdx.x + gridDim.x * blockIdx.y;
#define WN_STAMP(k) do { if (stamps && lane == 0 && lin_blk < 64) \
        stamps[((size_t)lin_blk * 8 + wv) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
    WN_STAMP(0);

    // cooperative fetch of one chunk into registers.  Raw patch: thread t < 396 owns pixel (r, col) = (t / 66, t % 66) of
    // the 6 x 66 patch for all 16 channels -- one address increment per load and per LDS store (the first version divided
    // a flat element index per element: ~350 VALU instructions per chunk, as much issue time as the chunk's MFMAs).
    // Filter slab: 4 float4 per thread.
    const int pr = tid / 66, pc = tid - pr * 66;
    const bool p_own = tid < 396;
    const int pyy = y0 + pr, pxx = x0 + pc;
    const bool p_in = p_own && pyy >= 0 && pyy < H && pxx >= 0 && pxx < W;
    const float *p_src = xb + (p_in ? (size_t)pyy * W + pxx : 0);
    const int p_dst = pr * WN_RS + pc;
    float rreg[WN_CK];
    f32x4 ureg[2];                                                  // (ext_vector_type: an array of HIP float4 structs is not promoted to registers)
    auto fetch = [&](int kc) {                                      // chunk kc -> registers: in flight during the MFMAs
        const float *src = p_src + (size_t)kc * WN_CK * plane;
#pragma unroll
        for (int c = 0; c < WN_CK; ++c) {                           // unconditional load of a valid address, masked after
            const float v = src[(size_t)c * plane];
            rreg[c] = p_in ? v : 0.f;
        }
        const f32x4 *p = reinterpret_cast<const f32x4 *>(up) + ((size_t)kc * n_cb + cb) * 1024 + tid;
#pragma unroll
        for (int u = 0; u < 2; ++u) ureg[u] = p[512 * u];
    };
    auto commit = [&](float *buf) {
        if (p_own) {
#pragma unroll
            for (int c = 0; c < WN_CK; ++c) buf[c * 6 * WN_RS + p_dst] = rreg[c];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) reinterpret_cast<f32x4 *>(buf + WN_RAW)[tid + 512 * u] = ureg[u];
    };

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ti = lane & 15, q = lane >> 4;
    const int trow = tg >> 1, tcol = 16 * (tg & 1) + ti;            // this lane's tile inside the block (A operand)
    const int r_off = (2 * trow) * WN_RS + 2 * tcol;                // + (ci*6 + r) * WN_RS
    const int u_off = WN_RAW + (cs * 64 + lane) * 4;                // + (m*4 + pg) * 512 floats

    // 8-channel chunks, two LDS buffers (2 x 29 KB: still two blocks per CU): chunk k+1 is written -- from registers filled
    // during chunk k-1..k -- by whichever wave has finished its MFMAs of chunk k, while the others still compute; one
    // barrier per chunk.  (16-channel chunks in a single buffer: 2.8k cycles of barrier + commit + barrier per 8.0k of MFMAs.)
    fetch(0);
    commit(sm);
    __syncthreads();
    if (n_chunks > 1) fetch(1);
#pragma unroll 1
    for (int kc = 0; kc < n_chunks; ++kc) {
        if (kc < 8) WN_STAMP(2 + 3 * kc);
        const float *buf = sm + (kc & 1) * WN_BUF;
#pragma unroll 1
        for (int m = 0; m < 2; ++m) {                               // (unrolled, both patches are fetched up front: spills)
            const float *rp = buf + r_off + (2 * q + m) * 6 * WN_RS;
            // patch rows as two packed halves (columns 01 | 23); V = B^T d B with v_pk_add_f32
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dl[r] = *reinterpret_cast<const f32x2 *>(rp + r * WN_RS);
                dh[r] = *reinterpret_cast<const f32x2 *>(rp + r * WN_RS + 2);
            }
            const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
            const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
            float v[16];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                           // row r of V from t[r][0..3] = (tl.x, tl.y, th.x, th.y)
                v[4 * r + 0] = tl[r].x - th[r].x;
                v[4 * r + 1] = tl[r].y + th[r].x;
                v[4 * r + 2] = th[r].x - tl[r].y;
                v[4 * r + 3] = tl[r].y - th[r].y;
            }
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) {
                const float4 u = *reinterpret_cast<const float4 *>(buf + u_off + (m * 4 + pg) * 512);
                acc[4 * pg + 0] = mfma16(v[4 * pg + 0], u.x, acc[4 * pg + 0]);
                acc[4 * pg + 1] = mfma16(v[4 * pg + 1], u.y, acc[4 * pg + 1]);
                acc[4 * pg + 2] = mfma16(v[4 * pg + 2], u.z, acc[4 * pg + 2]);
                acc[4 * pg + 3] = mfma16(v[4 * pg + 3], u.w, acc[4 * pg + 3]);
            }
        }
        if (stamps) asm volatile("" :: "v"(acc[0][0]), "v"(acc[15][3]));
        if (kc < 8) WN_STAMP(3 + 3 * kc);
        if (kc + 1 < n_chunks) {
            commit(sm + ((kc + 1) & 1) * WN_BUF);
            if (kc + 2 < n_chunks) fetch(kc + 2);
        }
        __syncthreads();
        if (kc < 7) WN_STAMP(1 + 3 * (kc + 1));
    }
    WN_STAMP(28);

    // ---- inverse transform (wave-local) and output.  Lane (co = 16*cs + l%16, quad q): tiles 4q .. 4q+3 of its segment ----
    // (the loop ended on a barrier) LDS is reused as the output tile [32 co][4 rows][64+4]
    float *ot = sm;
    constexpr int OS = 68;
    const int co_l = 16 * cs + ti;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float mm[4][4];
#pragma unroll
        for (int p = 0; p < 16; ++p) mm[p >> 2][p & 3] = acc[p][r];
        float t0[4], t1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t0[j] = mm[0][j] + mm[1][j] + mm[2][j];
            t1[j] = mm[1][j] - mm[2][j] - mm[3][j];
        }
        const float y00 = t0[0] + t0[1] + t0[2], y01 = t0[1] - t0[2] - t0[3];
        const float y10 = t1[0] + t1[1] + t1[2], y11 = t1[1] - t1[2] - t1[3];
        const int tc = 16 * (tg & 1) + 4 * q + r;                   // tile column inside the block
        float *o = ot + (co_l * 4 + 2 * trow) * OS + 2 * tc;
        *reinterpret_cast<float2 *>(o) = make_float2(y00, y01);
        *reinterpret_cast<float2 *>(o + OS) = make_float2(y10, y11);
    }
    __syncthreads();
    // rows of 64 outputs (256 B): 32 co x 4 rows = 128 rows, 16 per wave, one float per lane
    float *yb = y + ((size_t)b * Co + 32 * cb) * plane;
#pragma unroll 1
    for (int rr = 0; rr < 16; ++rr) {
        const int row = wv * 16 + rr;                               // = co*4 + r
        const int co = row >> 2, r = row & 3;
        const int yy = 2 * ty0 + r, xx = 2 * tx0 + lane;
        if (yy < H && xx < W) yb[(size_t)co * plane + (size_t)yy * W + xx] = ot[row * OS + lane];
    }
    WN_STAMP(29);
}

#ifdef NMRF_DEBUG_PROBES
static unsigned long long *g_wn_stamps = nullptr;   // debug (nmrf_debug_wino_timing): s_memtime stamps of the next launches
extern "C" int nmrf_debug_wino_timing(unsigned long long *stamps) { g_wn_stamps = stamps; return NMRF_OK; }
#else
static unsigned long long *const g_wn_stamps = nullptr;
#endif

extern "C" int nmrf_conv3x3_wino_f32(const float *x, const float *u_packed, int B, int Ci, int H, int W, int Co, float *y,
                                     void *stream) {
    if (!x || !u_packed || !y) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || Ci < 8 || (Ci & 7) || Co < 32 || (Co & 31)) return NMRF_EINVAL;
    const int tw = (W + 1) / 2, th = (H + 1) / 2;
    const long items = (long)((tw + 31) / 32) * ((th + 1) / 2) * (Co / 32);
    if (B > 65535 || items > 0x7ffffff0L) return NMRF_EINVAL;
    const size_t lds = (size_t)2 * WN_BUF * sizeof(float);
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int cur_dev = nmrf_cur_device();
    if (cur_dev < 0) return NMRF_ELAUNCH;
    bool &attr_set = attr_set_dev[cur_dev];
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set = true;
    }
    dim3 grid((unsigned)((items + 7) / 8 * 8), (unsigned)B);
    hipLaunchKernelGGL(conv3x3_wino_kernel, grid, dim3(512), lds, (hipStream_t)stream, x, u_packed, Ci, H, W, Co, y, g_wn_stamps);
    return nmrf_launch_status();
}
