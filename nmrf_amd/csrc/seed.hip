// Label-seed extraction kernels:
//   cost volume -> 1-D conv filter + softmax -> NMS + tie-exact top-k -> seed features.
// All HBM-bound / latency-bound; no matrix cores (tiny FLOP counts).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// A2: group-wise correlation volume.  One block = one (b, g, y, 64-wide x tile).
// The channels of the group go through LDS in passes of CK (= all 64 of them at the shipped sizes: with 16 per pass the block
// paid four exposed load latencies and eight barriers -- 50 us for 564 blocks): f1 tile [CK][64],
// f2 tile [CK][64+D-1] (left halo, zero for x<0).  Wave w of the block owns disparities
// {w, w+4, w+8, ...}; lane = x, so the f2 reads of one wave are consecutive LDS addresses.
// ------------------------------------------------------------------------------------------------
#define CV_CK 64                           // channels staged per pass: a whole group at the shipped 256 / 4 (one load phase, one barrier pair)
#define CV_TX 64
#define CV_MAXD 64

// DT: the number of disparities at compile time (0 = run-time D): the inner loop is then 1 LDS read with an immediate offset + 1
// FMA per (channel, disparity) instead of ~5 instructions of predicates and address arithmetic (the kernel was bound by them).
template <int DT>
__global__ __launch_bounds__(256) void cost_volume_kernel(const float *__restrict__ f1, const float *__restrict__ f2,
                                                          int C, int H, int W, int D_rt, int G, float *__restrict__ vol) {
    const int D = DT ? DT : D_rt;
    constexpr int NJ = DT ? (DT + 3) / 4 : CV_MAXD / 4;
    __shared__ float s1[CV_CK][CV_TX];
    __shared__ float s2[CV_CK][CV_TX + CV_MAXD];
    const int x0 = blockIdx.x * CV_TX;
    const int y = blockIdx.y;
    const int b = blockIdx.z / G, g = blockIdx.z % G;
    const int cpg = C / G;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int halo = D - 1;
    const size_t plane = (size_t)H * W;
    const float *p1 = f1 + ((size_t)b * C + (size_t)g * cpg) * plane + (size_t)y * W;
    const float *p2 = f2 + ((size_t)b * C + (size_t)g * cpg) * plane + (size_t)y * W;

    float acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0.f;

    for (int c0 = 0; c0 < cpg; c0 += CV_CK) {
        // Staging: wave w takes channels w, w + 4, ...; lane = x.  All 48 loads of a thread are issued before the first LDS store
        // (clamped addresses, values selected afterwards): as a loop of "load, wait, store" the block paid 42 serial memory round
        // trips -- most of the kernel's 29 us.
        float v1[CV_CK / 4], va[CV_CK / 4], vb[CV_CK / 4];
        const int xa = x0 - halo + lane, xb = xa + CV_TX;
#pragma unroll
        for (int i = 0; i < CV_CK / 4; ++i) {
            const int c = c0 + wv + 4 * i;
            const size_t row = (size_t)(c < cpg ? c : cpg - 1) * plane;
            v1[i] = p1[row + min(x0 + lane, W - 1)];
            va[i] = p2[row + min(max(xa, 0), W - 1)];
            vb[i] = p2[row + min(max(xb, 0), W - 1)];
        }
        __syncthreads();                                                    // the previous pass is done with the tiles
#pragma unroll
        for (int i = 0; i < CV_CK / 4; ++i) {
            const int c = wv + 4 * i;
            const bool okc = c0 + c < cpg;
            s1[c][lane] = (okc && x0 + lane < W) ? v1[i] : 0.f;
            s2[c][lane] = (okc && xa >= 0 && xa < W) ? va[i] : 0.f;
            if (lane < halo) s2[c][CV_TX + lane] = (okc && xb >= 0 && xb < W) ? vb[i] : 0.f;
        }
        __syncthreads();
        const float *r2 = &s2[0][0] + lane + halo - wv;                     // disparity d = wv + 4 j sits 4 j floats below
#pragma unroll 4
        for (int c = 0; c < CV_CK; ++c) {
            float a = s1[c][lane];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (DT % 4 == 0 && DT) acc[j] = fmaf(a, r2[c * (CV_TX + CV_MAXD) - 4 * j], acc[j]);
                else if (wv + 4 * j < D) acc[j] = fmaf(a, r2[c * (CV_TX + CV_MAXD) - 4 * j], acc[j]);
            }
        }
    }
    // Output through LDS: a pixel's D values of this group are one contiguous run of vol; written straight from the registers
    // (lane = x, one disparity per store) every store instruction hits 64 different lines for 4 bytes each.  The tile is
    // transposed in LDS ([x][D + 1]: odd stride, conflict-free; it reuses s2) and written as runs of consecutive floats.
    __syncthreads();                                                        // all waves are done reading s2
    float *so = &s2[0][0];                                                  // 64 x (D + 1) <= CV_CK x (CV_TX + CV_MAXD) floats
    const int sld = D + 1;
    {
        const float inv = 1.0f / (float)cpg;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int d = wv + 4 * j;
            if (d < D) so[lane * sld + d] = acc[j] * inv;
        }
    }
    __syncthreads();
    const int nx = W - x0 < CV_TX ? W - x0 : CV_TX;                         // pixels of this tile inside the row
    for (int i = threadIdx.x; i < nx * D; i += 256) {
        const int px = i / D, d = i - px * D;
        vol[((((size_t)b * H + y) * W + x0 + px) * G + g) * D + d] = so[px * sld + d];
    }
}

extern "C" int nmrf_cost_volume_f32(const float *f1, const float *f2, int B, int C, int H, int W, int D, int G,
                                    float *vol, void *stream) {
    if (!f1 || !f2 || !vol) return NMRF_ENULL;
    if (B < 1 || C < 1 || H < 1 || W < 1 || G < 1 || C % G || D < 1 || D > CV_MAXD) return NMRF_EINVAL;
    dim3 grid((W + CV_TX - 1) / CV_TX, H, B * G);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {                        // D_max / 8 of the shipped configs: 320, 256, 192
        case 40: hipLaunchKernelGGL(cost_volume_kernel<40>, grid, dim3(256), 0, st, f1, f2, C, H, W, D, G, vol); break;
        case 32: hipLaunchKernelGGL(cost_volume_kernel<32>, grid, dim3(256), 0, st, f1, f2, C, H, W, D, G, vol); break;
        case 24: hipLaunchKernelGGL(cost_volume_kernel<24>, grid, dim3(256), 0, st, f1, f2, C, H, W, D, G, vol); break;
        default: hipLaunchKernelGGL(cost_volume_kernel<0>, grid, dim3(256), 0, st, f1, f2, C, H, W, D, G, vol); break;
    }
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A3: Conv1d(G->8)-ReLU-Conv1d(8->16)-ReLU-Conv1d(16->1) along D (k=5, zero pad 2) + softmax.
// One wave = TWO pixels, lane = disparity bin: the two pixels are the two halves of a packed fp32 FMA
// (v_pk_fma_f32: each half an IEEE fma, same (channel, tap) order as the scalar form -> same bits), the
// weight is a wave-uniform SGPR broadcast to both halves, so the weights are read in their reference
// layout and in memory order (s_load_dwordx16), once per two pixels.  Activations go through a
// wave-private LDS strip [ch][2+64+2] of pixel pairs (zero guard cells realise the padding; one 8-byte
// read per tap); softmax max/sum are wave shuffles.  Nothing is shared between waves: every ordering
// point is a wave-level one.
// (One pixel per wave with scalar v_fmac and SGPR weights: 5 400 cycles per pixel -- a wave64 v_fma_f32 with an SGPR operand
// issues every 5.3 cycles on this chip, against 2.9 with VGPR operands and 4.9 for v_pk_fma_f32 with either; measured,
// profiles/r03s_filter_ab.txt.  This form: 7 400 cycles per PAIR of pixels, unroll / block shape / reduction style make no
// difference.)
// ------------------------------------------------------------------------------------------------
#define FS_WPB 4   // waves per block
#define FS_LD 68
typedef float fs_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ fs_f2 fs_fma(float w, fs_f2 v, fs_f2 acc) { return __builtin_elementwise_fma(fs_f2{w, w}, v, acc); }

template <int GT>   // cost groups held in registers (4), or 0: any G <= 16, taps re-read from LDS
__global__ __launch_bounds__(64 * FS_WPB) void dpn_filter_softmax_kernel(const float *__restrict__ vol,
        const float *__restrict__ w0, const float *__restrict__ b0, const float *__restrict__ w1,
        const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
        int64_t P, int G, int D, float *__restrict__ prob) {
    __shared__ fs_f2 sa[FS_WPB][16][FS_LD];   // input (G ch), then layer-1 output (8 ch), then layer-2 output (16 ch): each layer's taps
                                              // are in registers before its outputs overwrite them (34 KB: four blocks per CU)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t p = ((int64_t)blockIdx.x * FS_WPB + wv) * 2;
    if (p >= P) return;                         // wave uniform; no block barrier below
    const bool two = p + 1 < P;
    const bool in = lane < D;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const fs_f2 zero{0.f, 0.f};
    {   // guard cells of this wave's strip (never written again)
        const int ch = lane >> 2, k = lane & 3;
        sa[wv][ch][k < 2 ? k : 64 + k] = zero;
    }
    if (GT) G = GT;
    for (int g = 0; g < G; ++g) {
        const float *src = vol + ((size_t)p * G + g) * D + lane;
        sa[wv][g][2 + lane] = fs_f2{in ? src[0] : 0.f, (in && two) ? src[(size_t)G * D] : 0.f};
    }
    wave_sync();
    // ---- layer 1
    fs_f2 h1[8];
    if (GT) {
        fs_f2 x[GT ? GT * 5 : 1];
#pragma unroll
        for (int t = 0; t < GT * 5; ++t) x[t] = sa[wv][t / 5][lane + t % 5];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float bo = b0[o];
            fs_f2 s{bo, bo};
#pragma unroll
            for (int t = 0; t < GT * 5; ++t) s = fs_fma(w0[o * GT * 5 + t], x[t], s);
            h1[o] = in ? __builtin_elementwise_max(s, zero) : zero;
        }
    } else {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float bo = b0[o];
            fs_f2 s{bo, bo};
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < 5; ++k) s = fs_fma(w0[(o * G + g) * 5 + k], sa[wv][g][lane + k], s);
            h1[o] = in ? __builtin_elementwise_max(s, zero) : zero;
        }
    }
    wave_sync();                                // every lane has read its input taps
#pragma unroll
    for (int o = 0; o < 8; ++o) sa[wv][o][2 + lane] = h1[o];
    wave_sync();
    // ---- layer 2: the 40 taps of the pair in registers
    {
        fs_f2 x[40];
#pragma unroll
        for (int t = 0; t < 40; ++t) x[t] = sa[wv][t / 5][lane + t % 5];
        wave_sync();                            // ... before the outputs overwrite the strip
#pragma unroll 2
        for (int o = 0; o < 16; ++o) {
            const float bo = b1[o];
            fs_f2 s{bo, bo};
#pragma unroll
            for (int t = 0; t < 40; ++t) s = fs_fma(w1[o * 40 + t], x[t], s);
            sa[wv][o][2 + lane] = in ? __builtin_elementwise_max(s, zero) : zero;
        }
    }
    wave_sync();
    // ---- layer 3 + softmax of both pixels
    const float b2s = b2[0];
    fs_f2 s{b2s, b2s};
#pragma unroll
    for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) s = fs_fma(w2[c * 5 + k], sa[wv][c][lane + k], s);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float logit = in ? s[h] : -INFINITY;
        const float m = wave_max_nolds(logit);
        const float e = in ? expf(logit - m) : 0.f;
        const float z = wave_sum_nolds(e);                    // (same addition tree as the butterfly: common.h)
        if (in && (h == 0 || two)) prob[(size_t)(p + h) * D + lane] = e / z;
    }
}

extern "C" int nmrf_dpn_filter_softmax_f32(const float *vol, const float *w0, const float *b0, const float *w1,
                                           const float *b1, const float *w2, const float *b2, int64_t P, int G, int D,
                                           float *prob, void *stream) {
    if (!vol || !w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !prob) return NMRF_ENULL;
    if (P < 1 || G < 1 || G > 16 || D < 1 || D > 64) return NMRF_EINVAL;
    dim3 grid((unsigned)ceil_div64(P, 2 * FS_WPB));
    if (G == 4)
        hipLaunchKernelGGL(dpn_filter_softmax_kernel<4>, grid, dim3(64 * FS_WPB), 0, (hipStream_t)stream, vol, w0, b0, w1, b1, w2, b2,
                           P, G, D, prob);
    else
        hipLaunchKernelGGL(dpn_filter_softmax_kernel<0>, grid, dim3(64 * FS_WPB), 0, (hipStream_t)stream, vol, w0, b0, w1, b1, w2, b2,
                           P, G, D, prob);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A4: NMS + top-k with the tie order of ATen's CPU top-k (libstdc++ nth_element + sort).
// One LANE = one pixel; its D (value,index) pairs live in an LDS column (element j of lane l at
// [j][l], so lanes that are at the same step hit different banks).  The steps are the ones of
// oracle/topk_ref.c, same comparator, same swaps -> bit-identical indices.
// ------------------------------------------------------------------------------------------------
#define TK_MAXD 64
struct TkQ {
    float *v;       // [TK_MAXD][rows per block]
    int *i;
    int lane, ld;
    __device__ __forceinline__ float &V(int j) { return v[j * ld + lane]; }
    __device__ __forceinline__ int &I(int j) { return i[j * ld + lane]; }
};
__device__ __forceinline__ bool tk_gt(float xv, float yv) { return (isnan(xv) && !isnan(yv)) || (xv > yv); }

__device__ void tk_insertion_sort(TkQ q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i < last; ++i) {
        float vv = q.V(i);
        int vi = q.I(i);
        if (tk_gt(vv, q.V(first))) {
            for (int j = i; j > first; --j) { q.V(j) = q.V(j - 1); q.I(j) = q.I(j - 1); }
            q.V(first) = vv; q.I(first) = vi;
        } else {
            int j = i;
            while (tk_gt(vv, q.V(j - 1))) { q.V(j) = q.V(j - 1); q.I(j) = q.I(j - 1); --j; }
            q.V(j) = vv; q.I(j) = vi;
        }
    }
}

__device__ void tk_adjust_heap(TkQ q, int start, int hole, int len, float vv, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (tk_gt(q.V(start + child), q.V(start + child - 1))) child--;
        q.V(start + hole) = q.V(start + child); q.I(start + hole) = q.I(start + child);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q.V(start + hole) = q.V(start + child - 1); q.I(start + hole) = q.I(start + child - 1);
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && tk_gt(q.V(start + parent), vv)) {
        q.V(start + hole) = q.V(start + parent); q.I(start + hole) = q.I(start + parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q.V(start + hole) = vv; q.I(start + hole) = vi;
}

__device__ void tk_heap_select(TkQ q, int first, int middle, int last) {
    int len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {
            tk_adjust_heap(q, first, parent, len, q.V(first + parent), q.I(first + parent));
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; ++i)
        if (tk_gt(q.V(i), q.V(first))) {
            float vv = q.V(i); int vi = q.I(i);
            q.V(i) = q.V(first); q.I(i) = q.I(first);
            tk_adjust_heap(q, first, 0, len, vv, vi);
        }
}

__device__ void tk_nth_element(TkQ q, int n, int nth) {
    int first = 0, last = n;
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz(n));
    while (last - first > 3) {
        if (depth == 0) {
            tk_heap_select(q, first, nth + 1, last);
            float tv = q.V(first); int ti = q.I(first);
            q.V(first) = q.V(nth); q.I(first) = q.I(nth);
            q.V(nth) = tv; q.I(nth) = ti;
            return;
        }
        --depth;
        int a = first + 1, b = first + (last - first) / 2, c = last - 1, m;
        float va = q.V(a), vb = q.V(b), vc = q.V(c);
        if (tk_gt(va, vb)) m = tk_gt(vb, vc) ? b : (tk_gt(va, vc) ? c : a);
        else               m = tk_gt(va, vc) ? a : (tk_gt(vb, vc) ? c : b);
        { float tv = q.V(first); int ti = q.I(first);
          q.V(first) = q.V(m); q.I(first) = q.I(m); q.V(m) = tv; q.I(m) = ti; }
        const float pv = q.V(first);
        int lo = first + 1, hi = last;
        for (;;) {
            while (tk_gt(q.V(lo), pv)) ++lo;
            --hi;
            while (tk_gt(pv, q.V(hi))) --hi;
            if (!(lo < hi)) break;
            float tv = q.V(lo); int ti = q.I(lo);
            q.V(lo) = q.V(hi); q.I(lo) = q.I(hi); q.V(hi) = tv; q.I(hi) = ti;
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    tk_insertion_sort(q, first, last);
}

// RPB rows (pixels) per one-wave block.  A row's introselect is serial and data-dependent, so the 64 rows of a wave diverge and
// the wave pays for the union of their paths (49 us for the 7 332 pixels of a KITTI pair = 115 waves on a chip with 8 192 wave
// slots).  With fewer rows per wave -- down to one, on lane 0 -- the paths diverge less and the idle slots are used instead.
template <int RPB>
__global__ __launch_bounds__(64) void nms_topk_kernel(const float *__restrict__ prob, int64_t P, int D, int K, float eps,
                                                     int do_nms, int64_t *__restrict__ seeds) {
    __shared__ float sv[TK_MAXD * RPB];
    __shared__ int si[TK_MAXD * RPB];
    const int lane = threadIdx.x;
    const int64_t p0 = (int64_t)blockIdx.x * RPB;
    const int rows = (int)((P - p0) < RPB ? (P - p0) : RPB);
    // coalesced staging: element (row r, bin j) of the tile is at linear index r*D + j
    for (int i = lane; i < rows * D; i += 64) {
        int r = i / D, j = i - r * D;
        sv[j * RPB + r] = prob[(size_t)p0 * D + i];
    }
    __syncthreads();
    if (lane >= rows) return;
    TkQ q{sv, si, lane, RPB};
    if (do_nms) {
        float prev = -INFINITY, cur = q.V(0);
        for (int j = 0; j < D; ++j) {
            float nxt = (j + 1 < D) ? q.V(j + 1) : -INFINITY;
            bool any_nan = isnan(prev) || isnan(cur) || isnan(nxt);
            float m = any_nan ? NAN : fmaxf(fmaxf(prev, nxt), cur);     // max_pool1d(k=3,p=1), NaN-propagating
            float out = (cur != m && cur > eps) ? eps : cur;            // DPN.py:121-124
            q.V(j) = out;
            prev = cur; cur = nxt;
        }
    }
    for (int j = 0; j < D; ++j) q.I(j) = j;
    tk_nth_element(q, D, K - 1);
    tk_insertion_sort(q, 0, K - 1);
    for (int j = 0; j < K; ++j) seeds[(size_t)(p0 + lane) * K + j] = (int64_t)q.I(j);
}

template <int RPB>
static void launch_nms_topk(const float *prob, int64_t P, int D, int K, float eps, int do_nms, int64_t *seeds, hipStream_t st) {
    hipLaunchKernelGGL(nms_topk_kernel<RPB>, dim3((unsigned)ceil_div64(P, RPB)), dim3(64), 0, st, prob, P, D, K, eps, do_nms, seeds);
}

extern "C" int nmrf_nms_topk_f32(const float *prob, int64_t P, int D, int K, float eps, int do_nms, int64_t *seeds,
                                 void *stream) {
    if (!prob || !seeds) return NMRF_ENULL;
    if (P < 1 || D < 1 || D > TK_MAXD || K < 1 || K > 8 || K > D || K * 64 <= D) return NMRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    // as few rows per wave as keeps the launch within one generation of ~8k resident waves
    if (P <= 8192) launch_nms_topk<1>(prob, P, D, K, eps, do_nms, seeds, st);
    else if (P <= 4 * 8192) launch_nms_topk<4>(prob, P, D, K, eps, do_nms, seeds, st);
    else if (P <= 16 * 8192) launch_nms_topk<16>(prob, P, D, K, eps, do_nms, seeds, st);
    else launch_nms_topk<64>(prob, P, D, K, eps, do_nms, seeds, st);
    return nmrf_launch_status();
}

// ------------------------------------------------------------------------------------------------
// A6: 9-tap x G-group cost gather and Fourier(31) of the integer seed.
// Fourier op order (H2): c = coord*normalizer ; f = c * 2^i (exact) ; full-range sinf/cosf.
// ------------------------------------------------------------------------------------------------
// (fourier_write / fourier_pad: common.h -- shared with the warp kernel of token.hip)

// ------------------------------------------------------------------------------------------------
// A4 + A6 in one launch, one WAVE per pixel, the row in REGISTERS: lane j holds bin j as an order-preserving integer key and its
// index; the NMS is three shuffles, and the introselect + insertion sort above (same steps, same comparator, same swaps, hence
// the same tie order as ATen's CPU top-k) runs on the SCALAR unit: every index is wave-uniform, so element j is read with
// v_readlane / written with v_writelane and compared with s_cmp -- no LDS round trip per step (the LDS form above spends ~30 us
// on 7 332 rows of 40 bins: a serial chain of dependent LDS accesses on one lane per wave).  The K winners are then used at once
// by all 64 lanes for the seed features (A6: 9 taps x G groups of the cost volume, Fourier(31) of the seed) and the float copy of
// the seeds that the propagation consumes (NMP.py:619-649) -- the separate gather launch and the int64 -> float pass disappear.
//   key(x): tk_gt(x, y) <=> key(x) > key(y) as unsigned integers: NaN -> 0xffffffff (all NaNs tie, above +inf), -0 -> +0,
//           sign-magnitude -> biased two's complement.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned tk_key(float x) {
    if (isnan(x)) return 0xffffffffu;
    const unsigned b = __float_as_uint(x + 0.0f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

struct TkR {                                   // the row: lane j <-> element j (uniform j only)
    unsigned key;
    int idx;
    __device__ __forceinline__ unsigned K(int j) const { return (unsigned)__builtin_amdgcn_readlane((int)key, j); }
    __device__ __forceinline__ int I(int j) const { return __builtin_amdgcn_readlane(idx, j); }
    __device__ __forceinline__ void put(int j, unsigned k, int i) {       // (this clang has no writelane builtin: a select on the lane id)
        const bool me = (int)(threadIdx.x & 63) == j;
        key = me ? k : key;
        idx = me ? i : idx;
    }
    __device__ __forceinline__ void move(int dst, int src) { put(dst, K(src), I(src)); }
    __device__ __forceinline__ void swap(int a, int b) {
        const unsigned ka = K(a), kb = K(b);
        const int ia = I(a), ib = I(b);
        put(a, kb, ib);
        put(b, ka, ia);
    }
};

__device__ __forceinline__ void tkr_insertion_sort(TkR &q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i < last; ++i) {
        const unsigned vk = q.K(i);
        const int vi = q.I(i);
        if (vk > q.K(first)) {
            for (int j = i; j > first; --j) q.move(j, j - 1);
            q.put(first, vk, vi);
        } else {
            int j = i;
            while (vk > q.K(j - 1)) { q.move(j, j - 1); --j; }
            q.put(j, vk, vi);
        }
    }
}

__device__ __forceinline__ void tkr_adjust_heap(TkR &q, int start, int hole, int len, unsigned vk, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (q.K(start + child) > q.K(start + child - 1)) child--;
        q.move(start + hole, start + child);
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q.move(start + hole, start + child - 1);
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && q.K(start + parent) > vk) {
        q.move(start + hole, start + parent);
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q.put(start + hole, vk, vi);
}

__device__ __forceinline__ void tkr_heap_select(TkR &q, int first, int middle, int last) {
    const int len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {
            tkr_adjust_heap(q, first, parent, len, q.K(first + parent), q.I(first + parent));
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; ++i)
        if (q.K(i) > q.K(first)) {
            const unsigned vk = q.K(i);
            const int vi = q.I(i);
            q.move(i, first);
            tkr_adjust_heap(q, first, 0, len, vk, vi);
        }
}

__device__ __forceinline__ void tkr_nth_element(TkR &q, int n, int nth) {
    int first = 0, last = n;
    if (first == last || nth == last) return;
    int depth = 2 * (31 - __clz(n));
    while (last - first > 3) {
        if (depth == 0) {
            tkr_heap_select(q, first, nth + 1, last);
            q.swap(first, nth);
            return;
        }
        --depth;
        const int a = first + 1, b = first + (last - first) / 2, c = last - 1;
        int m;
        const unsigned va = q.K(a), vb = q.K(b), vc = q.K(c);
        if (va > vb) m = vb > vc ? b : (va > vc ? c : a);
        else         m = va > vc ? a : (vb > vc ? c : b);
        q.swap(first, m);
        const unsigned pv = q.K(first);
        int lo = first + 1, hi = last;
        for (;;) {
            while (q.K(lo) > pv) ++lo;
            --hi;
            while (pv > q.K(hi)) --hi;
            if (!(lo < hi)) break;
            q.swap(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    tkr_insertion_sort(q, first, last);
}

#define SS_ROWS 4                                // rows (waves) per block
__global__ __launch_bounds__(64 * SS_ROWS) void seed_select_kernel(const float *__restrict__ prob, const float *__restrict__ vol,
        int64_t P, int G, int D, int K, float eps, int do_nms, float normalizer, int64_t *__restrict__ seeds,
        float *__restrict__ seeds_f, float *__restrict__ cost, float *__restrict__ enc, int enc_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * SS_ROWS + (threadIdx.x >> 6);
    if (p >= P) return;                                                     // (wave-uniform; no barriers below)
    const bool in = lane < D;
    float cur = in ? prob[(size_t)p * D + lane] : -INFINITY;
    if (do_nms) {
        float prev = __shfl_up(cur, 1), nxt = __shfl_down(cur, 1);
        if (lane == 0) prev = -INFINITY;
        if (lane >= D - 1) nxt = -INFINITY;
        const bool any_nan = isnan(prev) || isnan(cur) || isnan(nxt);
        const float m = any_nan ? NAN : fmaxf(fmaxf(prev, nxt), cur);       // max_pool1d(k=3,p=1), NaN-propagating
        cur = (cur != m && cur > eps) ? eps : cur;                          // DPN.py:121-124
    }
    TkR q{tk_key(cur), lane};
    tkr_nth_element(q, D, K - 1);
    tkr_insertion_sort(q, 0, K - 1);
    // lanes 0 .. K-1 now hold the winners in top-k order
    if (lane < K) {
        seeds[(size_t)p * K + lane] = (int64_t)q.idx;
        if (seeds_f) seeds_f[(size_t)p * K + lane] = (float)q.idx;
    }
    if (cost) {
        const int per = G * 9;
        for (int i = lane; i < K * per; i += 64) {
            const int k = i / per, j = i - k * per;
            const int g = j / 9, tap = j - g * 9;
            int d = __shfl(q.idx, k) + tap - 4;
            d = d < 0 ? 0 : (d > D - 1 ? D - 1 : d);
            cost[((size_t)p * K + k) * per + j] = vol[((size_t)p * G + g) * D + d];
        }
    }
    if (enc) {
        for (int i = lane; i < K * 16; i += 64) {
            const int k = i >> 4;
            float *row = enc + ((size_t)p * K + k) * enc_ld;
            fourier_write((float)__shfl(q.idx, k), normalizer, i & 15, row);
            fourier_pad(i & 15, row, enc_ld);
        }
    }
}

extern "C" int nmrf_seed_select_f32(const float *prob, const float *vol, int64_t P, int G, int D, int K, float eps, int do_nms,
                                    float normalizer, int64_t *seeds, float *seeds_f, float *cost, float *enc, int enc_ld,
                                    void *stream) {
    if (!prob || !seeds) return NMRF_ENULL;
    if (cost && !vol) return NMRF_ENULL;
    if (P < 1 || D < 1 || D > TK_MAXD || K < 1 || K > 8 || K > D || K * 64 <= D || (cost && G < 1) || (enc && enc_ld < 31))
        return NMRF_EINVAL;
    if (ceil_div64(P, SS_ROWS) > 0x7fffffff) return NMRF_EINVAL;
    hipLaunchKernelGGL(seed_select_kernel, dim3((unsigned)ceil_div64(P, SS_ROWS)), dim3(64 * SS_ROWS), 0, (hipStream_t)stream, prob,
                       vol, P, G, D, K, eps, do_nms, normalizer, seeds, seeds_f, cost, enc, enc_ld);
    return nmrf_launch_status();
}

__global__ __launch_bounds__(256) void seed_features_kernel(const float *__restrict__ vol, const int64_t *__restrict__ seeds,
        int64_t P, int G, int D, int N, float normalizer, float *__restrict__ cost, float *__restrict__ enc, int enc_ld) {
    const int per = G * 9;
    const int slots = per + 16;                       // cost entries + 16 Fourier work items per token
    const int64_t total = P * N * slots;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i / slots;
        int j = (int)(i - t * slots);
        int64_t p = t / N;
        int seed = (int)seeds[t];
        if (j < per) {
            int g = j / 9, tap = j - g * 9;
            int d = seed + tap - 4;
            d = d < 0 ? 0 : (d > D - 1 ? D - 1 : d);
            cost[t * per + j] = vol[((size_t)p * G + g) * D + d];
        } else if (enc) {
            fourier_write((float)seed, normalizer, j - per, enc + t * enc_ld);
            fourier_pad(j - per, enc + t * enc_ld, enc_ld);
        }
    }
}

extern "C" int nmrf_seed_features_f32(const float *vol, const int64_t *seeds, int64_t P, int G, int D, int N,
                                      float normalizer, float *cost, float *enc, int enc_ld, void *stream) {
    if (!vol || !seeds || !cost) return NMRF_ENULL;
    if (P < 1 || G < 1 || D < 1 || N < 1 || (enc && enc_ld < 31)) return NMRF_EINVAL;
    int64_t total = P * N * (G * 9 + 16);
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(seed_features_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vol, seeds, P, G,
                       D, N, normalizer, cost, enc, enc_ld);
    return nmrf_launch_status();
}

__global__ __launch_bounds__(256) void fourier_embed_kernel(const float *__restrict__ coord, int64_t T, float normalizer,
                                                           float *__restrict__ enc, int ld, const int *__restrict__ out_map) {
    const int64_t total = T * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i >> 4;
        int64_t row = t;
        if (out_map) { row = out_map[t]; if (row < 0) continue; }
        fourier_write(coord[t], normalizer, (int)(i & 15), enc + row * ld);
        fourier_pad((int)(i & 15), enc + row * ld, ld);
    }
}

extern "C" int nmrf_fourier_embed_f32(const float *coord, int64_t T, float normalizer, float *enc, int ld, const int *out_map,
                                      void *stream) {
    if (!coord || !enc) return NMRF_ENULL;
    if (T < 1 || ld < 31) return NMRF_EINVAL;
    int64_t blocks = ceil_div64(T * 16, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(fourier_embed_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, coord, T,
                       normalizer, enc, ld, out_map);
    return nmrf_launch_status();
}
