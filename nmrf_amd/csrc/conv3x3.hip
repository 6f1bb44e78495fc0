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

#define C3_TC 32                          // tile columns = the MFMA's 32 columns
#define C3_PSTRIDE 80                     // bytes per pixel record
#define C3_AFF 256                        // most channels the affine table takes (its LDS is sized by the call: 2 * Ci floats)
#define C3_OLD 36                         // row pitch (floats) of the output staging tile
#define C3_OUTSIDE 0xffffffffu
// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14; here vmcnt = n, the others unconstrained
#define C3_VMCNT(n) (((n) & 15) | (((n) >> 4) << 14) | 0x0f70)

// Generalisations of the 3x3 / stride-1 kernel (template parameters KT = taps per side, STRIDE):
//   * STRIDE 2 (layer2.0.conv1 of the backbone, nmrf/models/backbone.py:74): a wave owns ONE output row (G = 1 pixel group), the
//     tile is 4 rows x 32 columns, the halo (2*4+1) x (2*32+1); halo columns are stored de-interleaved (even columns, then odd) so
//     that the 32 lanes of a tap still read 32 consecutive 80-byte records.
//   * ROWS 1 at stride 1: the same one-row-per-wave form (4 x 32 tile) for maps too small to fill the chip with 8 x 32 tiles (the
//     1/8-resolution DPN context conv at batch 1: 60 blocks of 24 stages -> 120 blocks of half the work).
//   * KT 4 (the 7x7 / stride-2 stem, backbone.py:70, as a 4x4 / stride-1 convolution over the 2x2 space-to-depth image: 12
//     channels padded to one 16-channel slab, pad 2 before / 1 after): same kernel, four taps per stage.
template <int KT, int STRIDE, int ROWS = (STRIDE == 1 ? 2 : 1)>
struct C3Geom {
    static constexpr int G = ROWS;                             // 32-pixel groups (output rows) per wave
    static constexpr int TR = 4 * G;                           // tile rows
    static constexpr int HR = (TR - 1) * STRIDE + KT;          // halo rows
    static constexpr int HC = (C3_TC - 1) * STRIDE + KT;       // halo columns
    static constexpr int NPIX = HR * HC;
    static constexpr int IPT = (2 * NPIX + 255) / 256;         // (pixel, 8-channel half) items per thread
    static constexpr int EVEN = (HC + 1) / 2;                  // STRIDE 2: records of the even halo columns of a row
    // record of halo pixel (py, px)
    __host__ __device__ static constexpr int rec(int py, int px) {
        return STRIDE == 1 ? py * HC + px : py * HC + (px & 1) * EVEN + (px >> 1);
    }
};

struct Conv3Args {
    const float *x;              // [B, Ci, H, W]
    int Ci, H, W;                // input size
    int Ho, Wo, pad;             // output size; padding before (rows and columns)
    const float *stats;          // in_stats workspace of x ([B*Ci][chunks][2]) or NULL
    int chunks;
    float eps;
    const ss_u32x4 *wstream;     // [groups][Ci/16 * KT stages][KT * STRIPS pairs][128] 16-byte words
    int64_t group_stride;        // 16-byte words between channel groups
    float *out;                  // [B, Co, H, W]
    int Co;
    float inv;
    int tiles_x, tiles_per_image, n_tiles, per_xcd;
    unsigned long long *stamps;  // debug build: s_memtime stamps of the first stamp_tiles tiles (DBG & 64)
    int stamp_tiles;
    int *range_flag;             // sticky fp16-range flag of the split operands (split_mfma.h), may be NULL
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads (and a workgroup fence, once LDS-DMA is outstanding) also drains
// the vector-memory counter: every barrier would wait for the halo prefetch that is meant to stay in flight across a slab.  The
// vmcnt waits that the LDS-DMA needs are written out by hand at the call sites.
template <bool SKIP = false>
__device__ __forceinline__ void lds_barrier() {
    if constexpr (SKIP) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DBG: timing experiments of the tools-only debug build (wrong results): 1 no weight DMA after the first stage, 2 no halo
// restaging after the first slab, 4 no barriers, 8 no MFMAs, 16 no output stores, 32 no LDS fragment reads after a stage's first,
// 64 s_memtime stamps per wave of the first 128 tiles
template <int STRIPS, int KT, int STRIDE, int DBG = 0, int ROWS = (STRIDE == 1 ? 2 : 1)>
__global__ __launch_bounds__(256, STRIDE == 2 ? 1 : (STRIPS == 2 && KT == 3 ? 3 : 2)) void conv3x3_split_kernel(Conv3Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using GM = C3Geom<KT, STRIDE, ROWS>;
    constexpr int G = GM::G, C3_TR = GM::TR, C3_HC = GM::HC, C3_NPIX = GM::NPIX, C3_IPT = GM::IPT, C3_NITEM = 2 * GM::NPIX;
    constexpr int STAGE_U4 = STRIPS * KT * 128;             // 16-byte words per weight stage
    ss_u32x4 *ring = reinterpret_cast<ss_u32x4 *>(smem);                       // 2 slots
    unsigned char *tile = smem + 2 * STAGE_U4 * 16;
    float *Aff = reinterpret_cast<float *>(tile + C3_NPIX * C3_PSTRIDE);        // [2][Ci]: scale, shift

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hi = lane >> 5;
    // consecutive block ids are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (shared halos in its L2)
    const int t = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);
    if (t >= a.n_tiles) return;
    unsigned long long acc_wait = 0, acc_comp = 0, t_prev = 0;
#define C3_STAMP(k) do { if constexpr ((DBG & 64) != 0) { if (lane == 0 && t < a.stamp_tiles && blockIdx.y == 0) a.stamps[(t * 4 + wv) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define C3_NOW() (((DBG & 64) != 0) ? __builtin_amdgcn_s_memtime() : 0ull)
    C3_STAMP(0);
    if constexpr ((DBG & 64) != 0) {             // word 12 / 13: the chip-wide 100 MHz counter at entry / exit (the shader clock = the ratio)
        if (lane == 0 && t < a.stamp_tiles && blockIdx.y == 0) a.stamps[(t * 4 + wv) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
    }
    const int b = t / a.tiles_per_image;
    const int rem = t - b * a.tiles_per_image;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int y0 = ty * C3_TR, x0 = tx * C3_TC;
    const int64_t HW = (int64_t)a.H * a.W, HWo = (int64_t)a.Ho * a.Wo;
    const int grp = blockIdx.y;
    const ss_u32x4 *wst = a.wstream + (size_t)grp * a.group_stride;
    const int n_slabs = a.Ci >> 4, total = n_slabs * KT;

    // ---- weight stages: global -> LDS ring by LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at wave-uniform base +
    // 16 l; pinned by nmrf_selftest_lds_dma), no staging registers.  A stage is 2 * KT * STRIPS one-KB wave-instructions, dealt to
    // the 4 waves; issued at the top of stage g-1 into the slot stage g-2 vacated, waited for (vmcnt) before barrier g.
    auto dma_w = [&](int g, int slot) {
        if ((DBG & 1) && g > 1) return;
        const ss_u32x4 *src = wst + (size_t)g * STAGE_U4 + lane;
        ss_u32x4 *dst = ring + slot * STAGE_U4;
#pragma unroll
        for (int k = 0; k < (2 * KT * STRIPS + 3) / 4; ++k) {
            const int n = wv + 4 * k;                                            // wave-uniform
            if (n < 2 * KT * STRIPS)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 64 * n),
                                                 (__attribute__((address_space(3))) void *)(dst + 64 * n), 16, 0, 0);
        }
    };
    dma_w(0, 0);

    // ---- halo items of this thread: (pixel p of the 10 x 34 halo, channel half h) ------------------------------------------
    const float *xu = a.x + (size_t)b * a.Ci * HW;                             // uniform
    unsigned loff[C3_IPT];          // lane offset into a channel row (pixel + the half's 4 channels); C3_OUTSIDE when outside
    int ldo[C3_IPT];                // byte offset of the record half in the tile; -1: no item
#pragma unroll
    for (int k = 0; k < C3_IPT; ++k) {
        const int i = tid + 256 * k;
        const int h = i >= C3_NPIX ? 1 : 0;
        const int p = i - h * C3_NPIX;                                          // record index
        const int py = p / C3_HC, q = p - py * C3_HC;
        const int px = STRIDE == 1 ? q : (q < GM::EVEN ? 2 * q : 2 * (q - GM::EVEN) + 1);
        const int gy = y0 * STRIDE - a.pad + py, gx = x0 * STRIDE - a.pad + px;
        const bool item = i < C3_NITEM;
        const bool inside = item && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        loff[k] = inside ? (unsigned)((int64_t)(4 * h) * HW + (int64_t)gy * a.W + gx) : C3_OUTSIDE;
        ldo[k] = item ? p * C3_PSTRIDE + h * 32 : -1;
    }
    // One item = the 8 channels of (halo pixel, channel half): 8 scalar loads (a channel row is HW floats away from the next), then
    // normalise / zero the padding / split, then two 16-byte LDS stores.  The three steps are spread over a slab's stages: the
    // loads of slab s+1 are issued BETWEEN the MFMAs of slab s's first stage, converted between the MFMAs of its last stage, and
    // stored at the slab change -- only the stores and two barriers are left outside the matrix stream (census of the first
    // version, which did all of it at the slab change: halo restaging 25 % of a wave's time, 1.9k cycles per slab for the load
    // issue alone, 1.3k for the conversion; profiles/r02o_conv_census.txt).  The compiler's scheduler sinks such side work below
    // the MFMAs of its block, so the order is written out and fenced (sched_barrier(0)) MFMA by MFMA.
    float guard = 0.f;                 // fp16 range guard of the activation splits (split_mfma.h)
    float Rt[C3_IPT][8];               // raw fp32 channels of an item (slot jj <-> channel (jj & 3) + 8 (jj >> 2) + 4 h)
    unsigned Pk[C3_IPT][8];            // the same item converted: packed fp16 pairs, words 0..3 = hi, 4..7 = lo
    // part i (0..3) of an item = its slots 2i, 2i+1 = one packed word of hi and one of lo
    auto load_part = [&](int k, int i, int slab) {
        const float *xs = xu + (size_t)(16 * slab) * HW;                        // uniform
        const unsigned o = loff[k] == C3_OUTSIDE ? 0u : loff[k];                // (any valid address; the value is discarded)
#pragma unroll
        for (int jj = 2 * i; jj < 2 * i + 2; ++jj) Rt[k][jj] = (xs + (size_t)((jj & 3) + 8 * (jj >> 2)) * HW)[o];
    };
    auto convert_part = [&](int k, int i, int slab) {
        f32x2 v = {Rt[k][2 * i], Rt[k][2 * i + 1]};
        if (a.stats) {
            const int hh = (tid + 256 * k) >= C3_NPIX ? 1 : 0;
            const float *af = Aff + 16 * slab + 4 * hh + ((2 * i) & 3) + 8 * ((2 * i) >> 2);
            const f32x2 sc = *reinterpret_cast<const f32x2 *>(af), sh = *reinterpret_cast<const f32x2 *>(af + a.Ci);
            v[0] = fmaxf(fmaf(v[0], sc[0], sh[0]), 0.f);
            v[1] = fmaxf(fmaf(v[1], sc[1], sh[1]), 0.f);
        }
        if (loff[k] == C3_OUTSIDE) v = f32x2{0.f, 0.f};                          // zero padding applies AFTER the normalisation
        h16x2 h2, l2;
        split2u_g(v, h2, l2, guard);
        Pk[k][i] = *reinterpret_cast<const unsigned *>(&h2);
        Pk[k][4 + i] = *reinterpret_cast<const unsigned *>(&l2);
    };
    auto store_item = [&](int k) {
        if (ldo[k] < 0) return;
        *reinterpret_cast<ss_u32x4 *>(tile + ldo[k]) = ss_u32x4{Pk[k][0], Pk[k][1], Pk[k][2], Pk[k][3]};
        *reinterpret_cast<ss_u32x4 *>(tile + ldo[k] + 16) = ss_u32x4{Pk[k][4], Pk[k][5], Pk[k][6], Pk[k][7]};
    };

    auto mma = [](const h16x8 &x, const h16x8 &y, const f32x16 &c) -> f32x16 {
        if constexpr (DBG & 8) {                 // keep the operands alive, no matrix work
            f32x16 r = c;
            r[0] += (float)x[0] + (float)y[1];
            return r;
        } else {
            return mfma16h(x, y, c);
        }
    };
    f32x16 acc[STRIPS][G];
#pragma unroll
    for (int s = 0; s < STRIPS; ++s)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][g][r] = 0.f;

#pragma unroll
    for (int k = 0; k < C3_IPT; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) load_part(k, i, 0);

    // ---- per-channel affine of this image (InstanceNorm folded into the load); after the first halo loads are in flight: one
    // memory round trip for both
    if (a.stats)
        for (int c = tid; c < a.Ci; c += 256) {
            float sc, sh;
            in_affine_of(a.stats + ((size_t)b * a.Ci + c) * a.chunks * 2, a.chunks, HW, a.eps, sc, sh);
            Aff[c] = sc;
            Aff[a.Ci + c] = sh;
        }
    __builtin_amdgcn_s_waitcnt(C3_VMCNT(0));
    C3_STAMP(1);
    lds_barrier<(DBG & 4) != 0>();               // stage 0 of the weights and the affine table are visible
    C3_STAMP(2);
#pragma unroll
    for (int k = 0; k < C3_IPT; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) convert_part(k, i, 0);

    const unsigned char *tb0 = tile + ((G * wv * STRIDE) * C3_HC + j) * C3_PSTRIDE + hi * 32;    // halo row of the wave's first output row, tap 0
    constexpr int NSTEP = G == 2 ? KT * STRIPS : KT;                            // MFMA groups of a stage
    // the side work of MFMA group q of stage dy: item k rides with group k * NSTEP / IPT, its part i behind the group's MFMA i
    auto side = [&](auto dd, auto qq, auto ii, int nslab) {
        constexpr int dy = decltype(dd)::value, q = decltype(qq)::value, i = decltype(ii)::value;
        if constexpr ((DBG & 2) == 0 && i < 4) {
            ss_static_for<C3_IPT>([&](auto kk) {
                constexpr int k = decltype(kk)::value;
                if constexpr (k * NSTEP / C3_IPT == q) {
                    if constexpr (dy == 0) load_part(k, i, nslab);
                    if constexpr (dy == KT - 1) convert_part(k, i, nslab);
                }
            });
        }
        if constexpr (dy == 0 || dy == KT - 1) __builtin_amdgcn_sched_barrier(0);     // keep it here: between MFMA i and MFMA i+1
    };
    auto stage = [&](auto dd, int slab) {
        constexpr int dy = decltype(dd)::value;
        const int g = slab * KT + dy;
        const int nslab = slab + 1 < n_slabs ? slab + 1 : slab;                 // (the last slab reloads itself: same vmcnt bookkeeping)
        // Top of stage g: this wave's share of stage g has landed (vmcnt), then the barrier: every wave is done with stage g-1 and
        // stage g is visible; then the DMA of stage g+1 into the slot stage g-1 vacated.  vmcnt is in order: the DMA that the
        // dy == 1 stage waits for was issued BEFORE the 8 * IPT halo loads riding with stage dy == 0, which stay in flight until the
        // slab's last stage converts them; the others after.  (s_waitcnt as the builtin: the compiler's own wait-count bookkeeping
        // then knows what is pending and adds no vmcnt(0) of its own behind the fresh DMA issue.)
        t_prev = C3_NOW();
        if constexpr (dy == 0) {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT(0));
            lds_barrier<(DBG & 4) != 0>();
            if (!(DBG & 2) || slab == 0) {
#pragma unroll
                for (int k = 0; k < C3_IPT; ++k) store_item(k);
            }
            lds_barrier<(DBG & 4) != 0>();
            dma_w(g + 1, (g + 1) & 1);           // (total = KT * slabs: a dy == 0 stage is never the last one)
        } else if constexpr (dy == 1) {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT((DBG & 2) ? 0 : C3_IPT * 8));
            lds_barrier<(DBG & 4) != 0>();
            if (g + 1 < total) dma_w(g + 1, (g + 1) & 1);
        } else {
            __builtin_amdgcn_s_waitcnt(C3_VMCNT(0));
            lds_barrier<(DBG & 4) != 0>();
            if (g + 1 < total) dma_w(g + 1, (g + 1) & 1);
        }
        { const unsigned long long n = C3_NOW(); acc_wait += n - t_prev; t_prev = n; }
        if (g == 0) C3_STAMP(3);
        // ---- stage g: taps (dy, 0..KT-1) of this slab against STRIPS strips ---------------------------------------------------------
        const ss_u32x4 *wb = ring + (g & 1) * STAGE_U4 + lane;
        const unsigned char *tb = tb0 + dy * (C3_HC * C3_PSTRIDE);
        // record offset of tap dx relative to tap 0 (STRIDE 2: odd taps live in the odd-column half of the row)
        auto tap_rec = [](int dx) { return STRIDE == 1 ? dx : (dx & 1) * GM::EVEN + (dx >> 1); };
        if constexpr (G == 2) {
            h16x8 ah[2], al[2], bh[2][2], bl[2][2];
            auto read_a = [&](int q, int buf) {
                ah[buf] = *reinterpret_cast<const h16x8 *>(wb + q * 128);
                al[buf] = *reinterpret_cast<const h16x8 *>(wb + q * 128 + 64);
            };
            auto read_b = [&](int dx, int buf) {
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const unsigned char *p = tb + (gg * STRIDE * C3_HC + tap_rec(dx)) * C3_PSTRIDE;
                    bh[buf][gg] = *reinterpret_cast<const h16x8 *>(p);
                    bl[buf][gg] = *reinterpret_cast<const h16x8 *>(p + 16);
                }
            };
            read_b(0, 0);
            read_a(0, 0);
            if constexpr (DBG & 32) { read_b(1, 1); read_a(1, 1); }
            ss_static_for<KT * STRIPS>([&](auto qq) {
                constexpr int q = decltype(qq)::value;
                constexpr int dx = q / STRIPS, s = q % STRIPS;
                if constexpr (!(DBG & 32)) {
                    if constexpr (q + 1 < KT * STRIPS) read_a(q + 1, (q + 1) & 1);
                    if constexpr (s == STRIPS - 1 && dx < KT - 1) read_b(dx + 1, (dx + 1) & 1);
                }
                // LDS reads may not sink below this point, MFMAs may not rise above it (ALU / vector memory may cross)
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
                const h16x8 fh = ah[q & 1], fl = al[q & 1];
                // small terms first; the two pixel groups alternate so that consecutive MFMAs never share an accumulator
                acc[s][0] = mma(fl, bh[dx & 1][0], acc[s][0]);
                side(dd, qq, std::integral_constant<int, 0>{}, nslab);
                acc[s][1] = mma(fl, bh[dx & 1][1], acc[s][1]);
                side(dd, qq, std::integral_constant<int, 1>{}, nslab);
                acc[s][0] = mma(fh, bl[dx & 1][0], acc[s][0]);
                side(dd, qq, std::integral_constant<int, 2>{}, nslab);
                acc[s][1] = mma(fh, bl[dx & 1][1], acc[s][1]);
                side(dd, qq, std::integral_constant<int, 3>{}, nslab);
                acc[s][0] = mma(fh, bh[dx & 1][0], acc[s][0]);
                acc[s][1] = mma(fh, bh[dx & 1][1], acc[s][1]);
            });
        } else {
            // one pixel group per wave: a tap's fragments of ALL strips are read together and the MFMAs go term-major over the
            // strips, so that consecutive MFMAs never share an accumulator
            h16x8 ah[2][STRIPS], al[2][STRIPS], bh[2], bl[2];
            auto read_tap = [&](int dx, int buf) {
                const unsigned char *p = tb + tap_rec(dx) * C3_PSTRIDE;
                bh[buf] = *reinterpret_cast<const h16x8 *>(p);
                bl[buf] = *reinterpret_cast<const h16x8 *>(p + 16);
#pragma unroll
                for (int s = 0; s < STRIPS; ++s) {
                    ah[buf][s] = *reinterpret_cast<const h16x8 *>(wb + (dx * STRIPS + s) * 128);
                    al[buf][s] = *reinterpret_cast<const h16x8 *>(wb + (dx * STRIPS + s) * 128 + 64);
                }
            };
            read_tap(0, 0);
            ss_static_for<KT>([&](auto qq) {
                constexpr int dx = decltype(qq)::value;
                if constexpr (dx + 1 < KT) read_tap(dx + 1, (dx + 1) & 1);
                __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
                ss_static_for<3 * STRIPS>([&](auto mm) {                        // term-major over the strips
                    constexpr int m = decltype(mm)::value, term = m / STRIPS, s = m % STRIPS;
                    if constexpr (term == 0) acc[s][0] = mma(al[dx & 1][s], bh[dx & 1], acc[s][0]);
                    if constexpr (term == 1) acc[s][0] = mma(ah[dx & 1][s], bl[dx & 1], acc[s][0]);
                    if constexpr (term == 2) acc[s][0] = mma(ah[dx & 1][s], bh[dx & 1], acc[s][0]);
                    if constexpr (m < 4) side(dd, qq, mm, nslab);
                });
            });
        }
        if constexpr ((DBG & 64) != 0) {
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]));     // (the stamp waits for the stage's last MFMA)
            acc_comp += C3_NOW() - t_prev;
            if (g == 0) C3_STAMP(4);
            if (g == KT - 1) C3_STAMP(5);
        }
    };
#pragma unroll 1
    for (int slab = 0; slab < n_slabs; ++slab) ss_static_for<KT>([&](auto dd) { stage(dd, slab); });
    C3_STAMP(6);
    split_guard_commit(guard, a.range_flag);

    // ---- epilogue: a C/D register is 32 consecutive pixels of one output channel ---------------------------------------------------
    const int co_base = grp * STRIPS * 32;
    float *ou = a.out + ((size_t)b * a.Co + co_base) * HWo;                     // uniform
    const int x = x0 + j;
    // Rows that are 16-byte aligned (Wo % 4 == 0; x0 is a multiple of 32) go through a wave-private LDS tile [32 co][36] and leave
    // as 16-byte stores, 8 lanes per 128-byte run: 16 store instructions per wave and tile instead of 64 (a census of the scalar
    // form: 3.5-5.8k of a wave's 48k cycles went into issuing them).  The tile reuses the weight ring: nothing is in flight any more.
    if (!(DBG & 16) && (a.Wo & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0) {
        lds_barrier<(DBG & 4) != 0>();                           // every wave is done with the ring
        float *ot = reinterpret_cast<float *>(smem) + wv * (32 * C3_OLD);
        const int rco = lane >> 3, rpx = 4 * (lane & 7);
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int y = y0 + G * wv + gg;
#pragma unroll
            for (int s = 0; s < STRIPS; ++s) {
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[((r & 3) + 8 * (r >> 2) + 4 * hi) * C3_OLD + j] = acc[s][gg][r] * a.inv;
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int co = rco + 8 * pass;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(ot + co * C3_OLD + rpx);
                    if (y < a.Ho && x0 + rpx < a.Wo)
                        *reinterpret_cast<f32x4 *>(ou + (size_t)(32 * s + co) * HWo + (size_t)y * a.Wo + x0 + rpx) = v;
                }
            }
        }
    } else {
#pragma unroll
        for (int gg = 0; gg < G; ++gg) {
            const int y = y0 + G * wv + gg;
            if (y >= a.Ho || x >= a.Wo) continue;
            const unsigned oo = (unsigned)((int64_t)(4 * hi) * HWo + (int64_t)y * a.Wo + x);
#pragma unroll
            for (int s = 0; s < STRIPS; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (!(DBG & 16) || acc[s][gg][r] == 123.456f)
                        (ou + (size_t)(32 * s + (r & 3) + 8 * (r >> 2)) * HWo)[oo] = acc[s][gg][r] * a.inv;
        }
    }
    C3_STAMP(7);
    if constexpr ((DBG & 64) != 0) {
        if (lane == 0 && t < a.stamp_tiles && blockIdx.y == 0) {
            a.stamps[(t * 4 + wv) * 16 + 15] = ((unsigned long long)(blockIdx.x & 7) << 32) | __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
            a.stamps[(t * 4 + wv) * 16 + 8] = acc_wait;
            a.stamps[(t * 4 + wv) * 16 + 9] = acc_comp;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a.stamps[(t * 4 + wv) * 16 + 11] = __builtin_amdgcn_s_memtime();
            a.stamps[(t * 4 + wv) * 16 + 13] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

#ifdef NMRF_DEBUG_PROBES
static int g_conv3_variant = 0;
static unsigned long long *g_conv3_stamps = nullptr;
static int g_conv3_stamp_tiles = 128;
extern "C" int nmrf_debug_conv3_variant(int v) { g_conv3_variant = v; return NMRF_OK; }
// stamps: device buffer of 128 tiles x 4 waves x 16 words, or NULL to switch the timing build off
extern "C" int nmrf_debug_conv3_timing(void *stamps) { g_conv3_stamps = (unsigned long long *)stamps; g_conv3_stamp_tiles = 128; return NMRF_OK; }
// the same for the first `tiles` tiles (word 15 of a wave's record: XCD of the block << 32 | HW_ID)
extern "C" int nmrf_debug_conv3_timing_n(void *stamps, int tiles) {
    g_conv3_stamps = (unsigned long long *)stamps;
    g_conv3_stamp_tiles = tiles;
    return NMRF_OK;
}
#endif

template <int STRIPS, int KT, int STRIDE, int DBG = 0, int ROWS = (STRIDE == 1 ? 2 : 1)>
static int launch_conv3(const Conv3Args &a, int groups, hipStream_t st) {
#ifdef NMRF_DEBUG_PROBES
    if constexpr (DBG == 0 && KT == 3 && STRIDE == 1 && STRIPS != 3 && ROWS == 2) {
        if (g_conv3_stamps) {
            Conv3Args b = a;
            b.stamps = g_conv3_stamps;
            b.stamp_tiles = g_conv3_stamp_tiles;
            return launch_conv3<STRIPS, KT, STRIDE, 64>(b, groups, st);
        }
        switch (g_conv3_variant) {
            case 1: return launch_conv3<STRIPS, KT, STRIDE, 1>(a, groups, st);
            case 2: return launch_conv3<STRIPS, KT, STRIDE, 2>(a, groups, st);
            case 3: return launch_conv3<STRIPS, KT, STRIDE, 3>(a, groups, st);
            case 4: return launch_conv3<STRIPS, KT, STRIDE, 4>(a, groups, st);
            case 7: return launch_conv3<STRIPS, KT, STRIDE, 7>(a, groups, st);
            case 8: return launch_conv3<STRIPS, KT, STRIDE, 8>(a, groups, st);
            case 16: return launch_conv3<STRIPS, KT, STRIDE, 16>(a, groups, st);
            case 23: return launch_conv3<STRIPS, KT, STRIDE, 23>(a, groups, st);
            case 32: return launch_conv3<STRIPS, KT, STRIDE, 32>(a, groups, st);
            case 55: return launch_conv3<STRIPS, KT, STRIDE, 55>(a, groups, st);
            case 63: return launch_conv3<STRIPS, KT, STRIDE, 63>(a, groups, st);
            default: break;
        }
    }
#endif
    static bool attr_set_dev[NMRF_MAX_DEV] = {};
    const int dev = nmrf_cur_device();
    if (dev < 0) return NMRF_ELAUNCH;
    // The CU hands out its 160 KB of LDS in 1 280-byte granules (tools/ab/occupancy_probe.hip: at 53 824 B -- this kernel with a fixed
    // 256-channel table -- a CU holds TWO blocks, at 53 760 B three, whatever the occupancy API computes): the table is sized by Ci.
    size_t lds = (size_t)2 * STRIPS * KT * 2048 + C3Geom<KT, STRIDE, ROWS>::NPIX * C3_PSTRIDE + 2 * (size_t)a.Ci * sizeof(float);
#ifdef NMRF_DEBUG_PROBES
    if (g_conv3_variant == 300) lds += 2 * (size_t)(C3_AFF - a.Ci) * sizeof(float);       // A/B: the fixed 256-channel table of rounds 2-4
#endif
    if (!attr_set_dev[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_split_kernel<STRIPS, KT, STRIDE, DBG, ROWS>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return NMRF_ELAUNCH;
        attr_set_dev[dev] = true;
    }
    hipLaunchKernelGGL((conv3x3_split_kernel<STRIPS, KT, STRIDE, DBG, ROWS>), dim3(8 * a.per_xcd, groups), dim3(256), lds, st, a);
    return nmrf_launch_status();
}

extern "C" int nmrf_conv_split_f32(const float *x, int B, int Ci, int H, int W, const float *stats, int chunks, float eps,
                                   const void *stream_w, int kt, int stride, int pad, int strips, int groups, float inv_scale,
                                   int Co, float *out, int *range_flag, void *stream) {
    if (!x || !stream_w || !out) return NMRF_ENULL;
    if (B < 1 || H < 1 || W < 1 || Ci < 16 || (Ci & 15) || strips < 2 || strips > 4 || groups < 1 || Co != strips * groups * 32 ||
        (stats && (chunks < 1 || Ci > C3_AFF)) || pad < 0 || pad >= kt)
        return NMRF_EINVAL;
    const int Ho = (H + kt - 1 - kt) / stride + 1, Wo = (W + kt - 1 - kt) / stride + 1;     // total padding kt - 1 (pad before, rest after)
    const int64_t HW = (int64_t)H * W;
    if (HW * 8 + HW > 0xffffffffLL) return NMRF_EINVAL;                         // 32-bit lane offsets
    // small maps: one output row per wave (4-row tiles) when the 8-row tiling leaves most of the chip idle
    const int tx = (Wo + C3_TC - 1) / C3_TC;
    bool small = stride == 1 && kt == 3 && strips == 2 && (int64_t)tx * ((Ho + 7) / 8) * B * groups <= 256;
#ifdef NMRF_DEBUG_PROBES
    if (g_conv3_variant == 100 && stride == 1 && kt == 3 && strips == 2) small = true;       // A/B: 4-row tiles regardless of size
    if (g_conv3_variant == 101) small = false;
#endif
    const int tr = stride == 1 && !small ? 8 : 4;
    const int ty = (Ho + tr - 1) / tr;
    const int64_t n = (int64_t)tx * ty * B;
    if (n > 0x7ffffff) return NMRF_EINVAL;
    Conv3Args a{x, Ci, H, W, Ho, Wo, pad, stats, chunks, eps, reinterpret_cast<const ss_u32x4 *>(stream_w),
                (int64_t)(Ci / 16) * kt * strips * kt * 128, out, Co, inv_scale, tx, tx * ty, (int)n, (int)((n + 7) / 8), nullptr, 0, range_flag};
    hipStream_t st = (hipStream_t)stream;
    const int key = kt * 100 + stride * 10 + strips;
    switch (key) {
        case 312: return small ? launch_conv3<2, 3, 1, 0, 1>(a, groups, st) : launch_conv3<2, 3, 1>(a, groups, st);
        case 313: return launch_conv3<3, 3, 1>(a, groups, st);
        case 314: return launch_conv3<4, 3, 1>(a, groups, st);
        case 322: return launch_conv3<2, 3, 2>(a, groups, st);
        case 323: return launch_conv3<3, 3, 2>(a, groups, st);
        case 412: return launch_conv3<2, 4, 1>(a, groups, st);
        default: return NMRF_EINVAL;
    }
}

extern "C" int nmrf_conv3x3_split_f32(const float *x, int B, int Ci, int H, int W, const float *stats, int chunks, float eps,
                                      const void *stream_w, int strips, int groups, float inv_scale, int Co, float *out,
                                      int *range_flag, void *stream) {
    return nmrf_conv_split_f32(x, B, Ci, H, W, stats, chunks, eps, stream_w, 3, 1, 1, strips, groups, inv_scale, Co, out, range_flag,
                               stream);
}

#ifdef NMRF_DEBUG_PROBES
// resident blocks per CU of the three stride-1 3x3 variants, as the runtime sees them
extern "C" int nmrf_debug_conv3_occupancy(int *out3) {
    const size_t tile = C3Geom<3, 1>::NPIX * C3_PSTRIDE + 2 * C3_AFF * sizeof(float);
    hipFuncSetAttribute((const void *)conv3x3_split_kernel<2, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)conv3x3_split_kernel<3, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void *)conv3x3_split_kernel<4, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(out3 + 0, conv3x3_split_kernel<2, 3, 1>, 256, 2 * 2 * 3 * 2048 + tile);
    hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(out3 + 1, conv3x3_split_kernel<3, 3, 1>, 256, 2 * 3 * 3 * 2048 + tile);
    hipError_t e3 = hipOccupancyMaxActiveBlocksPerMultiprocessor(out3 + 2, conv3x3_split_kernel<4, 3, 1>, 256, 2 * 4 * 3 * 2048 + tile);
    return (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) ? 0 : -2;
}
#endif
