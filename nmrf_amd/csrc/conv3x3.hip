// N2: 3x3 / stride 1 / pad 1 / no-bias convolution, NCHW fp32 in and out, as a DIRECT implicit GEMM on the
// split-operand fp16 MFMA (split_mfma.h), optionally with the InstanceNorm + ReLU of its INPUT folded into the operand load
//
//     out[b, co, y, x] = sum_{ci, dy, dx} W[co, ci, dy, dx] * f(x[b, ci, y + dy - 1, x + dx - 1]),   f = relu(IN(.)) or identity
//
// (conv1 / conv2 of ResidualBlock, nmrf/models/backbone.py:38-46; first conv of concatconv / gw, nmrf/models/NMRF.py:56-65).
// It replaces the Winograd fp32-MFMA kernel (conv_wino.hip: 0.46 of the fp32 matrix pipe, transforms on the VALU): the direct
// form has 2.25x the multiplies but runs them on a pipe that is 5.3x faster per product and needs no transforms at all.
//
// Formulation (same as nmp_block.hip): weights = A operand, activations = B operand with the PIXEL on the MFMA column.
//   block  = 4 waves, output tile 8 rows x 32 columns of one image, STRIPS x 32 output channels (blockIdx.y = channel group);
//   wave w = rows 2w, 2w+1 (two 32-pixel groups), STRIPS x 2 accumulators;
//   K loop = slabs of 16 input channels; per slab the (8+2) x (32+2) input halo is staged in LDS as split fp16:
//            record of a pixel = [half 0: hi 8ch | lo 8ch][half 1: hi 8ch | lo 8ch] + 16 B pad = 80 B -- the B operand of tap
//            (dy, dx) is two ds_read_b128 at pixel (row + dy, col + dx), conflict-free (80 = 5 x 16: the 16 lanes of a b128
//            group hit 16 different 16-byte slots of the 256-byte bank row);
//   weights = stream of "pairs" (2 KB = hi + lo fragment of one 32-channel strip x 16-deep chunk, nmrf_pack_split_weight_f32) in
//            the order slab > dy > dx > strip; a STAGE = one (slab, dy) = 3 x STRIPS pairs, double-buffered in LDS, filled one
//            stage ahead by LDS-DMA, one barrier per stage;
//   the next slab's halo is prefetched into registers during the current slab and written (normalised, split) at the slab change.
// LDS traffic per MFMA trio: 2 KB (STRIPS = 2) = 8 of the 24 LDS cycles a trio lasts on the CU: the matrix pipe is the bound.
#include "split_stream.h"
#include "in_affine.h"

#define C3_TR 8
#define C3_TC 32
#define C3_HR (C3_TR + 2)
#define C3_HC (C3_TC + 2)
#define C3_NPIX (C3_HR * C3_HC)          // 340 halo pixels
#define C3_PSTRIDE 80                     // bytes per pixel record
#define C3_NITEM (2 * C3_NPIX)            // (pixel, 8-channel half) items per slab
#define C3_IPT 3                          // items per thread: ceil(680 / 256)
#define C3_AFF 256                        // channels of the affine table
#define C3_OUTSIDE 0xffffffffu
// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; here vmcnt = n, the others unconstrained
#define C3_VMCNT(n) (((n) & 15) | (((n) >> 4) << 14) | 0x0f70)

struct Conv3Args {
    const float *x;              // [B, Ci, H, W]
    int Ci, H, W;
    const float *stats;          // in_stats workspace of x ([B*Ci][chunks][2]) or NULL
    int chunks;
    float eps;
    const ss_u32x4 *wstream;     // [groups][Ci/16 * 3 stages][3 * STRIPS pairs][128] 16-byte words
    int64_t group_stride;        // 16-byte words between channel groups
    float *out;                  // [B, Co, H, W]
    int Co;
    float inv;
    int tiles_x, tiles_per_image, n_tiles, per_xcd;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads (and a workgroup fence, once LDS-DMA is outstanding) also drains
// the vector-memory counter: every barrier would wait for the halo prefetch that is meant to stay in flight across a slab.  The
// vmcnt waits that the LDS-DMA needs are written out by hand at the call sites.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int STRIPS>
__global__ __launch_bounds__(256, STRIPS == 2 ? 3 : 2) void conv3x3_split_kernel(Conv3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int STAGE_U4 = STRIPS * 3 * 128;              // 16-byte words per weight stage
    ss_u32x4 *ring = reinterpret_cast<ss_u32x4 *>(smem);                       // 2 slots
    unsigned char *tile = smem + 2 * STAGE_U4 * 16;
    float *Aff = reinterpret_cast<float *>(tile + C3_NPIX * C3_PSTRIDE);        // [2][C3_AFF]: scale, shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    // consecutive block ids are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (shared halos in its L2)
    const int t = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (t >= a.n_tiles) return;
    const int b = t / a.tiles_per_image;
    const int rem = t - b * a.tiles_per_image;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * C3_TR, x0 = tx * C3_TC;
    const int64_t HW = (int64_t)a.H * a.W;
    const int grp = blockIdx.y;
    const ss_u32x4 *wst = a.wstream + (size_t)grp * a.group_stride;
    const int n_slabs = a.Ci >> 4, total = n_slabs * 3;

    // ---- weight stages: global -> LDS ring by LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at wave-uniform base +
    // 16 l; pinned by nmrf_selftest_lds_dma), no staging registers.  A stage is 6 * STRIPS one-KB wave-instructions, dealt to
    // the 4 waves; issued at the top of stage g-1 into the slot stage g-2 vacated, waited for (vmcnt) before barrier g.
    auto dma_w = [&](int g, int slot) {
        const ss_u32x4 *src = wst + (size_t)g * STAGE_U4 + lane;
        ss_u32x4 *dst = ring + slot * STAGE_U4;
#pragma unroll
        for (int k = 0; k < (6 * STRIPS + 3) / 4; ++k) {
            const int n = wv + 4 * k;                                            // wave-uniform
            if (n < 6 * STRIPS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * n),
                                                 (__attribute__((address_space(3))) void *)(dst + 64 * n), 16, 0, 0);
        }
    };
    dma_w(0, 0);

    // ---- per-channel affine of this image (InstanceNorm folded into the load) ----------------------------------------------
    if (a.stats)
        for (int c = tid; c < a.Ci; c += 256) {
            float sc, sh;
            in_affine_of(a.stats + ((size_t)b * a.Ci + c) * a.chunks * 2, a.chunks, HW, a.eps, sc, sh);
            Aff[c] = sc;
            Aff[C3_AFF + c] = sh;
        }

    // ---- halo items of this thread: (pixel p of the 10 x 34 halo, channel half h) ------------------------------------------
    const float *xu = a.x + (size_t)b * a.Ci * HW;                             // uniform
    unsigned loff[C3_IPT];          // lane offset into a channel row (pixel + the half's 4 channels); C3_OUTSIDE when outside
    int ldo[C3_IPT];                // byte offset of the record half in the tile; -1: no item
#pragma unroll
    for (int k = 0; k < C3_IPT; ++k) {
        const int i = tid + 256 * k;
        const int h = i >= C3_NPIX ? 1 : 0;
        const int p = i - h * C3_NPIX;
        const int py = p / C3_HC, px = p - py * C3_HC;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool item = i < C3_NITEM;
        const bool inside = item && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        loff[k] = inside ? (unsigned)((int64_t)(4 * h) * HW + (int64_t)gy * a.W + gx) : C3_OUTSIDE;
        ldo[k] = item ? p * C3_PSTRIDE + h * 32 : -1;
    }
    float Rt[C3_IPT][8];
    auto prefetch_tile = [&](int slab) {
        const float *xs = xu + (size_t)(16 * slab) * HW;                        // uniform
#pragma unroll
        for (int k = 0; k < C3_IPT; ++k) {
            const unsigned o = loff[k] == C3_OUTSIDE ? 0u : loff[k];            // (any valid address; the value is discarded)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) Rt[k][jj] = (xs + (size_t)((jj & 3) + 8 * (jj >> 2)) * HW)[o];
        }
    };
    auto write_tile = [&](int slab) {
#pragma unroll
        for (int k = 0; k < C3_IPT; ++k) {
            if (ldo[k] < 0) continue;
            float v[8];
            const int hh = (tid + 256 * k) >= C3_NPIX ? 1 : 0;
            if (a.stats) {
                const float *af = Aff + 16 * slab + 4 * hh;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const f32x4 sc = *reinterpret_cast<const f32x4 *>(af + 8 * g);
                    const f32x4 sh = *reinterpret_cast<const f32x4 *>(af + C3_AFF + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = fmaxf(fmaf(Rt[k][4 * g + e], sc[e], sh[e]), 0.f);
                }
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = Rt[k][jj];
            }
            if (loff[k] == C3_OUTSIDE) {                                                    // zero padding applies AFTER the normalisation
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[jj] = 0.f;
            }
            h16x8 vh, vl;
            split8u(v, vh, vl);
            *reinterpret_cast<h16x8 *>(tile + ldo[k]) = vh;
            *reinterpret_cast<h16x8 *>(tile + ldo[k] + 16) = vl;
        }
    };

    f32x16 acc[STRIPS][2];
#pragma unroll
    for (int s = 0; s < STRIPS; ++s)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][g][r] = 0.f;

    prefetch_tile(0);

    const unsigned char *tb0 = tile + ((2 * wv) * C3_HC + j) * C3_PSTRIDE + hi * 32;
    int dy = 0, slab = 0;
#pragma unroll 1
    for (int g = 0; g < total; ++g) {
        // Top of stage g: this wave's share of stage g has landed (vmcnt), then the barrier: every wave is done with stage g-1 and
        // stage g is visible; then the DMA of stage g+1 into the slot stage g-1 vacated.  vmcnt is in order: the DMA that the
        // dy == 1 stage waits for was issued BEFORE the 24 halo loads of the next slab, which stay in flight; the others after.
        // (s_waitcnt as the builtin and inside each branch: the compiler's own wait-count bookkeeping then knows that nothing is
        //  pending where the halo registers are reused, and adds no vmcnt(0) of its own behind the fresh DMA issue)
        if (dy == 0) {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT(0));
            lds_barrier();
            write_tile(slab);
            lds_barrier();
            dma_w(g + 1, (g + 1) & 1);           // (total = 3 * slabs: a dy == 0 stage is never the last one)
            prefetch_tile(slab + 1 < n_slabs ? slab + 1 : slab);            // (always 24 loads: the vmcnt bookkeeping counts them)
        } else if (dy == 1) {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT(C3_IPT * 8));
            lds_barrier();
            dma_w(g + 1, (g + 1) & 1);
        } else {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT(0));
            lds_barrier();
            if (g + 1 < total) dma_w(g + 1, (g + 1) & 1);
        }
        // ---- stage g: taps (dy, 0..2) of this slab against STRIPS strips --------------------------------------------------------
        const ss_u32x4 *wb = ring + (g & 1) * STAGE_U4 + lane;
        const unsigned char *tb = tb0 + dy * (C3_HC * C3_PSTRIDE);
        h16x8 ah[2], al[2], bh[2][2], bl[2][2];
        auto read_a = [&](int q, int buf) {
            ah[buf] = *reinterpret_cast<const h16x8 *>(wb + q * 128);
            al[buf] = *reinterpret_cast<const h16x8 *>(wb + q * 128 + 64);
        };
        auto read_b = [&](int dx, int buf) {
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const unsigned char *p = tb + (gg * C3_HC + dx) * C3_PSTRIDE;
                bh[buf][gg] = *reinterpret_cast<const h16x8 *>(p);
                bl[buf][gg] = *reinterpret_cast<const h16x8 *>(p + 16);
            }
        };
        read_b(0, 0);
        read_a(0, 0);
        ss_static_for<3 * STRIPS>([&](auto qq) {
            constexpr int q = decltype(qq)::value;
            constexpr int dx = q / STRIPS, s = q % STRIPS;
            if constexpr (q + 1 < 3 * STRIPS) read_a(q + 1, (q + 1) & 1);
            if constexpr (s == STRIPS - 1 && dx < 2) read_b(dx + 1, (dx + 1) & 1);
            // LDS reads may not sink below this point, MFMAs may not rise above it (see nmp_block.hip)
            __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
            const h16x8 fh = ah[q & 1], fl = al[q & 1];
            // small terms first; the two pixel groups alternate so that consecutive MFMAs never share an accumulator
            acc[s][0] = mfma16h(fl, bh[dx & 1][0], acc[s][0]);
            acc[s][1] = mfma16h(fl, bh[dx & 1][1], acc[s][1]);
            acc[s][0] = mfma16h(fh, bl[dx & 1][0], acc[s][0]);
            acc[s][1] = mfma16h(fh, bl[dx & 1][1], acc[s][1]);
            acc[s][0] = mfma16h(fh, bh[dx & 1][0], acc[s][0]);
            acc[s][1] = mfma16h(fh, bh[dx & 1][1], acc[s][1]);
        });
        if (++dy == 3) { dy = 0; ++slab; }
    }

    // ---- epilogue: a C/D register is 32 consecutive pixels of one output channel ---------------------------------------------------
    const int co_base = grp * STRIPS * 32;
    float *ou = a.out + ((size_t)b * a.Co + co_base) * HW;                      // uniform
    const int x = x0 + j;
#pragma unroll
    for (int gg = 0; gg < 2; ++gg) {
        const int y = y0 + 2 * wv + gg;
        if (y >= a.H || x >= a.W) continue;
        const unsigned oo = (unsigned)((int64_t)(4 * hi) * HW + (int64_t)y * a.W + x);
#pragma unroll
        for (int s = 0; s < STRIPS; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                (ou + (size_t)(32 * s + (r & 3) + 8 * (r >> 2)) * HW)[oo] = acc[s][gg][r] * a.inv;
    }
}

template <int STRIPS>
static int launch_conv3(const Conv3Args &a, int groups, hipStream_t st) {
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    const size_t lds = (size_t)2 * STRIPS * 3 * 2048 + C3_NPIX * C3_PSTRIDE + 2 * C3_AFF * sizeof(float);
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_split_kernel<STRIPS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    hipLaunchKernelGGL((conv3x3_split_kernel<STRIPS>), dim3(8 * a.per_xcd, groups), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

extern "C" int nmrf_conv3x3_split_f32(const float *x, int B, int Ci, int H, int W, const float *stats, int chunks, float eps,
                                      const void *stream_w, int strips, int groups, float inv_scale, int Co, float *out,
                                      void *stream) {
    if (!x || !stream_w || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || Ci < 16 || (Ci & 15) || strips < 2 || strips > 4 || groups < 1 || Co != strips * groups * 32 ||
        (stats && (chunks < 1 || Ci > C3_AFF)))
        return NMRF_EINVAL;
    const int64_t HW = (int64_t)H * W;
    if (HW * 8 + HW > 0xffffffffLL) return NMRF_EINVAL;                         // 32-bit lane offsets
    const int tx = (W + C3_TC - 1) / C3_TC, ty = (H + C3_TR - 1) / C3_TR;
    const int64_t n = (int64_t)tx * ty * B;
    if (n > 0x7ffffff) return NMRF_EINVAL;
    Conv3Args a{x, Ci, H, W, stats, chunks, eps, reinterpret_cast<const ss_u32x4 *>(stream_w),
                (int64_t)(Ci / 16) * 3 * strips * 3 * 128, out, Co, inv_scale, tx, tx * ty, (int)n, (int)((n + 7) / 8)};
    hipStream_t st = (hipStream_t)stream;
    switch (strips) {
        case 2: return launch_conv3<2>(a, groups, st);
        case 3: return launch_conv3<3>(a, groups, st);
        case 4: return launch_conv3<4>(a, groups, st);
        default: return NMRF_EINVAL;
    }
}
