// A15: multi-scale deformable attention, forward and backward (fp32 / fp64).
//
// value [B,S,M,D] keeps the head's D channels contiguous, so a thread that owns CH consecutive
// channels of one (b, query, head) reads each bilinear tap as one vector.  HBM/L2-bound gather.
// Forward : one thread = (b, q, m, channel chunk); loops levels x points.
// Backward: one thread = one sample (b, q, m, l, p); loops all D channels, so grad_loc / grad_w need no
//           cross-thread reduction (the reference needs seven shared-memory reduction variants for that);
//           grad_value is scattered with hardware float atomics.
// Sampling convention (ms_deform_im2col_cuda.cuh:285-288): h_im = loc_y*H - 0.5, zero outside,
// identical to grid_sample(align_corners=False).
#include "common.h"
#include "split_mfma.h"
#include <type_traits>

template <typename T, int CH>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const T *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const T *__restrict__ loc, const T *__restrict__ wgt, int B, int S, int M,
        int D, int L, int Lq, int P, T *__restrict__ out) {
    const int chunks = D / CH;
    const int64_t total = (int64_t)B * Lq * M * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ck = (int)(i % chunks);
        const int64_t bqm = i / chunks;                        // (b*Lq + q)*M + m
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / ((int64_t)M * Lq));
        const T *lp = loc + bqm * L * P * 2;
        const T *wp = wgt + bqm * L * P;
        T acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0;
        for (int l = 0; l < L; ++l) {
            const int Hh = (int)shapes[2 * l], Ww = (int)shapes[2 * l + 1];
            const T *vbase = value + (((size_t)b * S + (size_t)lvl_start[l]) * M + m) * D + ck * CH;
            for (int p = 0; p < P; ++p) {
                const T w_im = lp[(l * P + p) * 2] * Ww - (T)0.5;
                const T h_im = lp[(l * P + p) * 2 + 1] * Hh - (T)0.5;
                const T aw = wp[l * P + p];
                if (h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww) {
                    const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
                    const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
                    const T tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
                    T s[CH];
#pragma unroll
                    for (int c = 0; c < CH; ++c) s[c] = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int yy = h0 + (k >> 1), xx = w0 + (k & 1);
                        if (yy >= 0 && yy < Hh && xx >= 0 && xx < Ww) {
                            const T *v = vbase + ((size_t)yy * Ww + xx) * M * D;
#pragma unroll
                            for (int c = 0; c < CH; ++c) s[c] += tw[k] * v[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc[c] += s[c] * aw;
                }
            }
        }
        T *o = out + bqm * D + ck * CH;
#pragma unroll
        for (int c = 0; c < CH; ++c) o[c] = acc[c];
    }
}


// Fast path of the Swin-T neck's shape family (fp32, D = 8 channels per head, L * P = NPT = 4 sampling points): one thread =
// one (b, query, head) = ALL 8 channels of the head, so
//   * the sampling locations (NPT x 2 floats) and weights (NPT floats) of a thread are three 16-byte loads,
//   * a bilinear tap is the head's whole 32-byte row = two 16-byte loads, and the taps of PB points (PB x 4 x 2 loads) are ALL in
//     flight before the first is consumed (the generic kernel walks point by point, tap by tap behind range branches: a chain of
//     dependent L2 round trips per thread; 0.19 of HBM at [2, 96 256, 8, 8]),
//   * taps outside the map load a clamped (valid) address and get weight 0 -- no divergence,
//   * consecutive lanes are consecutive heads: 8 lanes write one full 256-byte output row.
// Same arithmetic as the generic kernel: s = sum_k tw[k] * v_k in tap order, acc += s * aw in point order.
// PB: points whose taps are requested together.  Measured at [2, 96 256, 8, 8] over the four levels of the neck (r04d): PB = 2 with
// four blocks per CU (126 VGPRs) 109 / 84 / 72 / 66 us, PB = 4 with three (168 VGPRs, 4 spilled) 101 / 83 / 77 / 73, one item per thread
// without the prefetch 124 / 87 / 77 / 69, the generic kernel 155 / 101 / 89 / 73: PB = 2 is the product form.
template <int NPT, int PB = 2, bool PRE = true>
__global__ __launch_bounds__(256, PB == 2 ? 4 : 3) void msda_fwd_d8_kernel(const float *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const float *__restrict__ loc, const float *__restrict__ wgt, int B, int S, int M,
        int L, int Lq, int P, float *__restrict__ out) {
    static_assert(NPT == 4, "three 16-byte operand loads per thread");
    // Persistent: a thread walks items gid, gid + stride, ... and requests the NEXT item's three operand vectors before it touches
    // the current one (PRE; pinned by the sched_barrier), so their HBM round trip runs under the current item's 32 tap loads and
    // FMAs instead of in front of the next item's -- with one item per thread a wave's life is two dependent memory round trips
    // back to back and ~12 waves per CU do not cover them.
    const int64_t total = (int64_t)B * Lq * M, stride = (int64_t)gridDim.x * 256;
    int64_t bqm = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (bqm >= total) return;
    float4 l0 = ldg4(loc + bqm * (NPT * 2)), l1 = ldg4(loc + bqm * (NPT * 2) + 4), w4 = ldg4(wgt + bqm * NPT);
    for (;;) {
        const int64_t nxt = bqm + stride;
        const int64_t nc = PRE && nxt < total ? nxt : bqm;                  // (the last item re-reads itself: no divergent load)
        float4 n0, n1, nw;
        if constexpr (PRE) {
            n0 = ldg4(loc + nc * (NPT * 2)); n1 = ldg4(loc + nc * (NPT * 2) + 4); nw = ldg4(wgt + nc * NPT);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / ((int64_t)M * Lq));
        const float lx[NPT] = {l0.x, l0.z, l1.x, l1.z}, ly[NPT] = {l0.y, l0.w, l1.y, l1.w}, aw[NPT] = {w4.x, w4.y, w4.z, w4.w};
        // tap addresses as 32-bit element offsets from `value` (the launcher checks that the tensor is below 2^32 bytes): one integer
        // multiply-add per tap; the bilinear weights as products of per-axis weights that are zeroed where the axis index leaves the
        // map (the same products as the generic kernel's hh * hw ... for the taps it takes, 0 for those it skips)
        unsigned tap[NPT][4];
        // geometry of point pt: floor / validity / (with WEIGHTS) the four bilinear weights.  Evaluated twice -- for the addresses
        // before the loads, for the weights behind them -- so that only the 8 location floats stay live across the 32 loads
        // (16 weight registers less: 170 -> under the 168 of three waves per SIMD)
        auto point = [&](int pt, unsigned *offs, float *tw4) {
            const int l = pt / P;                                       // (uniform)
            const int Hh = (int)shapes[2 * l], Ww = (int)shapes[2 * l + 1];
            const float w_im = lx[pt] * Ww - 0.5f, h_im = ly[pt] * Hh - 0.5f;
            const bool inside = h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww;
            const float hf = floorf(h_im), wf = floorf(w_im);
            const int h0 = inside ? (int)hf : 0, w0 = inside ? (int)wf : 0;
            if (offs) {
                const unsigned base = (unsigned)((((int64_t)b * S + lvl_start[l]) * M + m) * 8);
                const int yc[2] = {h0 < 0 ? 0 : h0, h0 + 1 < Hh ? h0 + 1 : Hh - 1}, xc[2] = {w0 < 0 ? 0 : w0, w0 + 1 < Ww ? w0 + 1 : Ww - 1};
#pragma unroll
                for (int k = 0; k < 4; ++k) offs[k] = base + (unsigned)(yc[k >> 1] * Ww + xc[k & 1]) * (unsigned)(M * 8);
            }
            if (tw4) {
                // (!inside includes non-finite locations -- the comparisons above are false for NaN: the fractions are then zeroed as
                // well, so that such a point contributes exactly 0 like in msda_fwd_kernel instead of 0 * NaN)
                const float lh = inside ? h_im - hf : 0.f, lw = inside ? w_im - wf : 0.f;
                const float wy[2] = {inside && h0 >= 0 ? 1 - lh : 0.f, inside && h0 + 1 < Hh ? lh : 0.f};
                const float wx[2] = {w0 >= 0 ? 1 - lw : 0.f, w0 + 1 < Ww ? lw : 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) tw4[k] = wy[k >> 1] * wx[k & 1];
            }
        };
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) point(pt, tap[pt], nullptr);
        // a batch of PB points: all its loads first, then one wait -- the empty asm "uses" the eight vectors of a point, so none of the
        // FMAs below can be scheduled (by the DAG or the machine scheduler) between the loads; left alone they are fed five at a time
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p0 = 0; p0 < NPT; p0 += PB) {
            f32x4 va[PB][4], vb[PB][4];
#pragma unroll
            for (int pp = 0; pp < PB; ++pp)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    va[pp][k] = *reinterpret_cast<const f32x4 *>(value + tap[p0 + pp][k]);
                    vb[pp][k] = *reinterpret_cast<const f32x4 *>(value + tap[p0 + pp][k] + 4);
                }
#pragma unroll
            for (int pp = 0; pp < PB; ++pp)
                asm volatile("" : "+v"(va[pp][0]), "+v"(va[pp][1]), "+v"(va[pp][2]), "+v"(va[pp][3]), "+v"(vb[pp][0]), "+v"(vb[pp][1]),
                             "+v"(vb[pp][2]), "+v"(vb[pp][3]));
#pragma unroll
            for (int pp = 0; pp < PB; ++pp) {
                float sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                float tw4[4];
                point(p0 + pp, nullptr, tw4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t = tw4[k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        sx[c] += t * va[pp][k][c];
                        sx[4 + c] += t * vb[pp][k][c];
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] += sx[c] * aw[p0 + pp];
            }
        }
        stg4(out + bqm * 8, make_float4(acc[0], acc[1], acc[2], acc[3]));
        stg4(out + bqm * 8 + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
        if (nxt >= total) break;
        bqm = nxt;
        if constexpr (PRE) { l0 = n0; l1 = n1; w4 = nw; }
        else { l0 = ldg4(loc + bqm * (NPT * 2)); l1 = ldg4(loc + bqm * (NPT * 2) + 4); w4 = ldg4(wgt + bqm * NPT); }
    }
}

// Tiled form of the fast path (round 6).  The kernel above runs at the rate of L2 -> L1 line fills, not of HBM: a tap is the 32-byte row of
// one head of one pixel, a pixel's eight heads are two 128-byte lines, so the 64 lanes of a load touch 64 lines and use a quarter of each
// (~2 KB of fills per (query, head); 68-101 us per call whether the level holds 49 MB or 0.8 MB).  Here a block owns an 8 x 8 tile of
// the QUERY grid (all eight heads: 512 threads, lanes = consecutive heads as above, so loc / weights / out stay fully coalesced),
// reduces the bounding box of its taps on the sampled level, stages that box ONCE with coalesced 256-byte pixel rows into LDS
// (pixel pitch 72 floats: adjacent pixels start 8 banks apart) and takes every tap from there.  The query grid is not an argument of
// the operator: the launcher GUESSES it (an integer multiple k of the level's H x W with k^2 H W = Lq -- the neck's queries are the
// 1/4-resolution grid, its levels 1/4 ... 1/32); a wrong guess costs locality, never correctness: the box comes from the data, and a
// block whose box exceeds the LDS budget (large learned offsets) takes its taps from global memory as before.
// Arithmetic: the point / tap / accumulation order of msda_fwd_d8_kernel; points outside the map are skipped (what the generic
// kernel does; the kernel above multiplies their clamped taps by a zero weight: the same value for finite data).
#define MSDA_T_PITCH 68
// TY: rows of the query tile (8 x TY queries x 8 heads = 64 TY threads).  The product form is TY = 8 (512 threads, up to 504 staged
// pixels, one block per CU); TY = 4 (256 threads, 252 pixels, two blocks per CU) measured slower at three of the four levels.  The next
// tile's three operand vectors are requested before the current tile's taps.
template <int NPT, int TY>
__global__ __launch_bounds__(64 * TY) void msda_fwd_d8_tiled_kernel(const float *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const float *__restrict__ loc, const float *__restrict__ wgt, int B, int S, int Lq, int kq,
        int cap_px, float *__restrict__ out) {
    static_assert(NPT == 4, "three 16-byte operand loads per thread");
    constexpr int NTHR = 64 * TY;
    extern __shared__ __attribute__((aligned(16))) float s_px[];                 // [cap_px][MSDA_T_PITCH]
    __shared__ int s_box[TY][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = tid & 7, ql = tid >> 3, ty = ql >> 3, tx = ql & 7;
    const int Hh = (int)shapes[0], Ww = (int)shapes[1];
    // the guessed query grid: kq x the level's (the launcher checked kq^2 S = Lq); a level whose H W is not S falls back to one row
    const bool grid_ok = (int64_t)Hh * Ww == S;
    const int qh = grid_ok ? Hh * kq : 1, qw = grid_ok ? Ww * kq : Lq;
    const int tiles_x = (qw + 7) / 8, tiles_y = (qh + TY - 1) / TY;
    // persistent, XCD-aware: the blocks of one XCD walk a contiguous run of tiles (neighbouring boxes share their margins in that L2)
    const int per_img = tiles_x * tiles_y, total = per_img * B;
    const int chunk = (total + 7) / 8;
    const int jstep = (int)(gridDim.x >> 3), j0 = (int)(blockIdx.x >> 3), xcd = (int)(blockIdx.x & 7);
    auto item_of = [&](int jj) { return (jj < chunk && xcd * chunk + jj < total) ? xcd * chunk + jj : -1; };
    struct Q { int64_t bqm; bool valid; int b; };
    auto query_of = [&](int item) {
        const int b = item / per_img, tile = item - b * per_img;
        const int qy = (tile / tiles_x) * TY + ty, qx = (tile % tiles_x) * 8 + tx;
        const bool valid = qy < qh && qx < qw;
        const int64_t q = (int64_t)(valid ? qy : qh - 1) * qw + (valid ? qx : qw - 1);
        return Q{((int64_t)b * Lq + q) * 8 + m, valid, b};
    };
    int item = item_of(j0);
    if (item < 0) return;                                                         // (whole block)
    Q cur = query_of(item);
    float4 l0 = ldg4(loc + cur.bqm * (NPT * 2)), l1 = ldg4(loc + cur.bqm * (NPT * 2) + 4), w4 = ldg4(wgt + cur.bqm * NPT);
    for (int jj = j0;; jj += jstep) {
    const int nitem = item_of(jj + jstep);
    const Q nxt = nitem >= 0 ? query_of(nitem) : cur;
    const bool valid = cur.valid;
    const int64_t bqm = cur.bqm;
    const int b = cur.b;
    const float lx[NPT] = {l0.x, l0.z, l1.x, l1.z}, ly[NPT] = {l0.y, l0.w, l1.y, l1.w}, aw[NPT] = {w4.x, w4.y, w4.z, w4.w};
    // geometry of the four points: clamped corner coordinates, bilinear weights (zero where an axis index leaves the map)
    int x0[NPT], x1[NPT], y0[NPT], y1[NPT];
    float tw[NPT][4];
    bool in[NPT];
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const float w_im = lx[pt] * Ww - 0.5f, h_im = ly[pt] * Hh - 0.5f;
        const bool inside = h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww;
        const float hf = floorf(h_im), wf = floorf(w_im);
        const int h0 = inside ? (int)hf : 0, w0 = inside ? (int)wf : 0;
        y0[pt] = h0 < 0 ? 0 : h0; y1[pt] = h0 + 1 < Hh ? h0 + 1 : Hh - 1;
        x0[pt] = w0 < 0 ? 0 : w0; x1[pt] = w0 + 1 < Ww ? w0 + 1 : Ww - 1;
        const float lh = inside ? h_im - hf : 0.f, lw = inside ? w_im - wf : 0.f;
        const float wy[2] = {inside && h0 >= 0 ? 1 - lh : 0.f, inside && h0 + 1 < Hh ? lh : 0.f};
        const float wx[2] = {w0 >= 0 ? 1 - lw : 0.f, w0 + 1 < Ww ? lw : 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) tw[pt][k] = wy[k >> 1] * wx[k & 1];
        in[pt] = inside && valid;
        if (in[pt]) {
            bx0 = min(bx0, x0[pt]); bx1 = max(bx1, x1[pt]);
            by0 = min(by0, y0[pt]); by1 = max(by1, y1[pt]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, o)); by0 = min(by0, __shfl_xor(by0, o));
        bx1 = max(bx1, __shfl_xor(bx1, o)); by1 = max(by1, __shfl_xor(by1, o));
    }
    if (lane == 0) { s_box[wv][0] = bx0; s_box[wv][1] = by0; s_box[wv][2] = bx1; s_box[wv][3] = by1; }
    __syncthreads();                                                              // (also: the previous tile's taps are done -- s_px is free)
#pragma unroll
    for (int w = 0; w < TY; ++w) {
        bx0 = min(bx0, s_box[w][0]); by0 = min(by0, s_box[w][1]);
        bx1 = max(bx1, s_box[w][2]); by1 = max(by1, s_box[w][3]);
    }
    const int wb = bx1 - bx0 + 1, hb = by1 - by0 + 1;
    const bool staged = bx1 >= bx0 && by1 >= by0 && (int64_t)wb * hb <= cap_px;      // (block-uniform)
    const float *vimg = value + ((int64_t)b * S + lvl_start[0]) * 64;             // pixel 0, head 0 of this image's level
    if (staged) {
        const int pieces = wb * hb * 16;                                          // 16-byte pieces: 16 per 256-byte pixel
        for (int i = tid; i < pieces; i += NTHR) {
            const int pix = i >> 4, pc = i & 15, py = pix / wb, px = pix - py * wb;
            stg4(s_px + pix * MSDA_T_PITCH + 4 * pc, ldg4(vimg + ((int64_t)(by0 + py) * Ww + bx0 + px) * 64 + 4 * pc));
        }
    }
    // the next tile's operands: their round trip runs under this tile's taps
    const float4 n0 = ldg4(loc + nxt.bqm * (NPT * 2)), n1 = ldg4(loc + nxt.bqm * (NPT * 2) + 4), nw = ldg4(wgt + nxt.bqm * NPT);
    __syncthreads();                                                              // staged pixels visible; s_box read by every wave
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        if (!in[pt]) continue;
        const int yy[2] = {y0[pt], y1[pt]}, xx[2] = {x0[pt], x1[pt]};
        f32x4 va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float *src = staged ? s_px + ((yy[k >> 1] - by0) * wb + (xx[k & 1] - bx0)) * MSDA_T_PITCH + 8 * m
                                      : vimg + ((int64_t)yy[k >> 1] * Ww + xx[k & 1]) * 64 + 8 * m;
            va[k] = *reinterpret_cast<const f32x4 *>(src);
            vb[k] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
        float sx[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float t = tw[pt][k];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                sx[c] += t * va[k][c];
                sx[4 + c] += t * vb[k][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += sx[c] * aw[pt];
    }
    if (valid) {
        stg4(out + bqm * 8, make_float4(acc[0], acc[1], acc[2], acc[3]));
        stg4(out + bqm * 8 + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
    }
    if (nitem < 0) break;
    item = nitem; cur = nxt; l0 = n0; l1 = n1; w4 = nw;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T *__restrict__ value, const int64_t *__restrict__ shapes,
        const int64_t *__restrict__ lvl_start, const T *__restrict__ loc, const T *__restrict__ wgt,
        const T *__restrict__ gout, int B, int S, int M, int D, int L, int Lq, int P, T *__restrict__ gvalue,
        T *__restrict__ gloc, T *__restrict__ gw) {
    const int64_t total = (int64_t)B * Lq * M * L * P;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int l = (int)((i / P) % L);
        const int64_t bqm = i / ((int64_t)L * P);
        const int m = (int)(bqm % M);
        const int b = (int)(bqm / ((int64_t)M * Lq));
        const int Hh = (int)shapes[2 * l], Ww = (int)shapes[2 * l + 1];
        const T w_im = loc[i * 2] * Ww - (T)0.5;
        const T h_im = loc[i * 2 + 1] * Hh - (T)0.5;
        const T aw = wgt[i];
        T g_w = 0, g_x = 0, g_y = 0;
        if (h_im > -1 && w_im > -1 && h_im < Hh && w_im < Ww) {
            const int h0 = (int)floor(h_im), w0 = (int)floor(w_im);
            const T lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
            const T tw[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
            // d(weight_k)/d(w_im), d(weight_k)/d(h_im)   (ms_deform_im2col_cuda.cuh:112-156)
            const T dwx[4] = {-hh, hh, -lh, lh};
            const T dwy[4] = {-hw, -lw, hw, lw};
            const size_t voff = (((size_t)b * S + (size_t)lvl_start[l]) * M + m) * D;
            const T *go = gout + bqm * D;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = h0 + (k >> 1), xx = w0 + (k & 1);
                if (yy >= 0 && yy < Hh && xx >= 0 && xx < Ww) {
                    const size_t o = voff + ((size_t)yy * Ww + xx) * M * D;
                    T dot = 0;
                    for (int c = 0; c < D; ++c) {
                        const T g = go[c];
                        dot += g * value[o + c];
                        atomicAdd(gvalue + o + c, tw[k] * aw * g);
                    }
                    g_w += tw[k] * dot;
                    g_x += dwx[k] * dot;
                    g_y += dwy[k] * dot;
                }
            }
        }
        gw[i] = g_w;
        gloc[i * 2] = Ww * g_x * aw;
        gloc[i * 2 + 1] = Hh * g_y * aw;
    }
}

#ifdef NMRF_DEBUG_PROBES
static int g_msda_variant = 0;       // tools: 1 = generic kernel, 2 = four points per load batch, 3 = one item per thread without prefetch, 4 = untiled d8 form, 5 = tiled with 8 x 4 tiles
extern "C" int nmrf_debug_msda_variant(int v) { g_msda_variant = v; return NMRF_OK; }
#endif

static inline unsigned grid_for(int64_t total) {
    int64_t blocks = ceil_div64(total, 256);
    return (unsigned)(blocks > 65536 ? 65536 : (blocks < 1 ? 1 : blocks));
}

template <typename T>
static int msda_forward(const T *value, const int64_t *shapes, const int64_t *lvl_start, const T *loc, const T *w, int B,
                        int S, int M, int D, int L, int Lq, int P, T *out, void *stream) {
    if (!value || !shapes || !lvl_start || !loc || !w || !out) return NMRF_ENULL;
    if (B < 1 || S < 1 || M < 1 || D < 1 || L < 1 || Lq < 1 || P < 1) return NMRF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if constexpr (std::is_same<T, float>::value) {
        const auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        if (D == 8 && L * P == 4 && (int64_t)B * S * M * 8 < ((int64_t)1 << 30) && al16(value) && al16(loc) && al16(w) && al16(out)) {   // the Swin-T neck's shapes
            static int n_cu_dev[NMRF_MAX_DEV] = {};
            const int dev = nmrf_cur_device();
            if (dev < 0) return NMRF_ELAUNCH;
            if (!n_cu_dev[dev]) {
                hipDeviceProp_t prop;
                if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NMRF_ELAUNCH;
                n_cu_dev[dev] = prop.multiProcessorCount;
            }
            const int64_t items = (int64_t)B * Lq * M;
            const unsigned one_each = grid_for(items), resident = (unsigned)n_cu_dev[dev] * 4;      // four 256-thread blocks per CU (126 VGPRs)
            const unsigned persistent = one_each < resident ? one_each : resident;
#ifdef NMRF_DEBUG_PROBES
            if (g_msda_variant == 2) {                                       // all 32 tap loads of an item together, three blocks per CU
                hipLaunchKernelGGL((msda_fwd_d8_kernel<4, 4>), dim3(n_cu_dev[dev] * 3 < (int)one_each ? n_cu_dev[dev] * 3 : one_each), dim3(256), 0, st,
                                   value, shapes, lvl_start, loc, w, B, S, M, L, Lq, P, out);
                return nmrf_launch_status();
            }
            if (g_msda_variant == 3) {                                       // one item per thread, no prefetch (the first form of this kernel)
                hipLaunchKernelGGL((msda_fwd_d8_kernel<4, 4, false>), dim3(one_each), dim3(256), 0, st, value, shapes, lvl_start, loc, w, B, S, M,
                                   L, Lq, P, out);
                return nmrf_launch_status();
            }
            if (g_msda_variant != 1)
#endif
            {
                // tiled form (see msda_fwd_d8_tiled_kernel): one level, eight heads, Lq = kq^2 S for an integer kq
                int kq = 0;
                if (L == 1 && M == 8 && Lq % S == 0) {
                    const int64_t r = Lq / S;
                    const int k = (int)(sqrt((double)r) + 0.5);
                    if ((int64_t)k * k == r && k >= 1) kq = k;
                }
#ifdef NMRF_DEBUG_PROBES
                if (g_msda_variant == 4) kq = 0;                               // tools: the untiled product form of rounds 4-5
#endif
                if (kq) {
                    // 8 x 8 query tiles, 512 threads, one block per CU with up to 504 staged pixels (137 KB): measured against 8 x 4 tiles /
                    // 256 threads / 252 pixels / two blocks per CU at the four levels of the neck (profiles/r06s_msda_tiled.txt)
                    static bool attr_set_dev[NMRF_MAX_DEV] = {};
                    bool small = false;
#ifdef NMRF_DEBUG_PROBES
                    small = g_msda_variant == 5;                                   // tools: the 8 x 4-tile form
#endif
                    const int cap_small = 252, cap_big = 504;
                    if (!attr_set_dev[dev]) {
                        // (the kernels also hold a few bytes of static LDS: the attribute is the dynamic part only)
                        if (hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_d8_tiled_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                cap_small * MSDA_T_PITCH * 4) != hipSuccess ||
                            hipFuncSetAttribute(reinterpret_cast<const void *>(msda_fwd_d8_tiled_kernel<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                cap_big * MSDA_T_PITCH * 4) != hipSuccess)
                            return NMRF_ELAUNCH;
                        attr_set_dev[dev] = true;
                    }
                    if (small)
                        hipLaunchKernelGGL((msda_fwd_d8_tiled_kernel<4, 4>), dim3((unsigned)((2 * n_cu_dev[dev] + 7) / 8 * 8)), dim3(256),
                                           (size_t)cap_small * MSDA_T_PITCH * 4, st, value, shapes, lvl_start, loc, w, B, S, Lq, kq, cap_small, out);
                    else
                        hipLaunchKernelGGL((msda_fwd_d8_tiled_kernel<4, 8>), dim3((unsigned)((n_cu_dev[dev] + 7) / 8 * 8)), dim3(512),
                                           (size_t)cap_big * MSDA_T_PITCH * 4, st, value, shapes, lvl_start, loc, w, B, S, Lq, kq, cap_big, out);
                    return nmrf_launch_status();
                }
                hipLaunchKernelGGL((msda_fwd_d8_kernel<4>), dim3(persistent), dim3(256), 0, st, value, shapes, lvl_start, loc, w, B, S, M, L, Lq,
                                   P, out);
                return nmrf_launch_status();
            }
        }
    }
    if (D % 4 == 0) {
        hipLaunchKernelGGL((msda_fwd_kernel<T, 4>), dim3(grid_for((int64_t)B * Lq * M * (D / 4))), dim3(256), 0, st, value,
                           shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out);
    } else {
        hipLaunchKernelGGL((msda_fwd_kernel<T, 1>), dim3(grid_for((int64_t)B * Lq * M * D)), dim3(256), 0, st, value,
                           shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out);
    }
    return nmrf_launch_status();
}

template <typename T>
static int msda_backward(const T *value, const int64_t *shapes, const int64_t *lvl_start, const T *loc, const T *w,
                         const T *gout, int B, int S, int M, int D, int L, int Lq, int P, T *gvalue, T *gloc, T *gw,
                         void *stream) {
    if (!value || !shapes || !lvl_start || !loc || !w || !gout || !gvalue || !gloc || !gw) return NMRF_ENULL;
    if (B < 1 || S < 1 || M < 1 || D < 1 || L < 1 || Lq < 1 || P < 1) return NMRF_EINVAL;
    hipLaunchKernelGGL((msda_bwd_kernel<T>), dim3(grid_for((int64_t)B * Lq * M * L * P)), dim3(256), 0,
                       (hipStream_t)stream, value, shapes, lvl_start, loc, w, gout, B, S, M, D, L, Lq, P, gvalue, gloc, gw);
    return nmrf_launch_status();
}

extern "C" int nmrf_msda_forward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start, const float *loc,
                                     const float *w, int B, int S, int M, int D, int L, int Lq, int P, float *out,
                                     void *stream) {
    return msda_forward<float>(value, shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out, stream);
}
extern "C" int nmrf_msda_forward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start,
                                     const double *loc, const double *w, int B, int S, int M, int D, int L, int Lq, int P,
                                     double *out, void *stream) {
    return msda_forward<double>(value, shapes, lvl_start, loc, w, B, S, M, D, L, Lq, P, out, stream);
}
extern "C" int nmrf_msda_backward_f32(const float *value, const int64_t *shapes, const int64_t *lvl_start,
                                      const float *loc, const float *w, const float *grad_out, int B, int S, int M, int D,
                                      int L, int Lq, int P, float *grad_value, float *grad_loc, float *grad_w,
                                      void *stream) {
    return msda_backward<float>(value, shapes, lvl_start, loc, w, grad_out, B, S, M, D, L, Lq, P, grad_value, grad_loc,
                                grad_w, stream);
}
extern "C" int nmrf_msda_backward_f64(const double *value, const int64_t *shapes, const int64_t *lvl_start,
                                      const double *loc, const double *w, const double *grad_out, int B, int S, int M,
                                      int D, int L, int Lq, int P, double *grad_value, double *grad_loc, double *grad_w,
                                      void *stream) {
    return msda_backward<double>(value, shapes, lvl_start, loc, w, grad_out, B, S, M, D, L, Lq, P, grad_value, grad_loc,
                                 grad_w, stream);
}

// ------------------------------------------------------------------------------------------------
// misc: MFMA lane-layout self-test, error strings, ABI version
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void selftest_mfma_kernel(const float *__restrict__ A, const float *__restrict__ Bm, int K,
                                                          float *__restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = A[i * K + k0 + hi];          // A[i][k]
        const float b = Bm[(k0 + hi) * 32 + i];      // B[k][j]
        acc = mfma32(a, b, acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[mfma_row(r, hi) * 32 + i] = acc[r];
}

extern "C" int nmrf_selftest_mfma_f32(const float *A, const float *Bm, int K, float *out, void *stream) {
    if (!A || !Bm || !out) return NMRF_ENULL;
    if (K < 2 || K > 64 || (K & 1)) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, K, out);
    return nmrf_launch_status();
}

#ifdef NMRF_DEBUG_PROBES   // tools-only library libnmrf_hip_debug.so (python -m nmrf_amd.build --debug)
// Debug: attainable v_mfma_f32_32x32x2_f32 rate (tools/kernel_bench.py --which mfma_peak).  CHAINS independent accumulator
// chains per wave, `iters` x 16 MFMAs each; out receives one value per thread so nothing is optimised away.
template <int CHAINS>
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(int iters, float *__restrict__ out) {
    f32x16 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-4f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = mfma32(a, b, acc[c]);
        a += 1e-6f;
    }
    float r = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) r += acc[c][0] + acc[c][15];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// Debug: the streaming-softmax tile loop of the attention kernels with the operands held in registers (no memory
// traffic at all) -- the issue-structure ceiling of that loop at a given number of waves per SIMD.
// VARIANT 0: S chain, softmax, PV chain (as in stripe_attn.hip).  VARIANT 1: two query-independent tiles interleaved
// (S of tile B issued before the softmax of tile A) so a lone wave has MFMA work in flight during its VALU phase.
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void attn_core_peak_kernel(int iters, const float *__restrict__ seed, float *__restrict__ out) {
    float qf[16], kf[16], vf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        qf[s] = seed[(threadIdx.x + s) & 255] * 0.1f; kf[s] = seed[(threadIdx.x * 3 + s) & 255]; vf[s] = seed[(threadIdx.x * 7 + s) & 255];
    }
    f32x16 acc_o;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_o[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    auto softmax = [&](f32x16 &st) {
        float m_tile = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
        m_tile = half_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __builtin_amdgcn_exp2f(st[r] - m_new); psum += st[r]; }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_o[r] *= alpha;
    };
    if (VARIANT == 0) {
        for (int it = 0; it < iters; ++it) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) st = mfma32(kf[s], qf[s], st);
            softmax(st);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], st[s], acc_o);
            kf[it & 15] += 1e-7f;
        }
    } else {
        f32x16 sa, sb;
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) sa = mfma32(kf[s], qf[s], sa);
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sb = mfma32(kf[s], qf[s], sb);       // S(B) in flight ...
            softmax(sa);                                                       // ... during softmax(A)
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], sa[s], acc_o);
            kf[it & 15] += 1e-7f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) sa = mfma32(kf[s], qf[s], sa);
            softmax(sb);
#pragma unroll
            for (int s = 0; s < 16; ++s) acc_o = mfma32(vf[s], sb[s], acc_o);
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc_o[0] + acc_o[15] + l_run;
}

extern "C" int nmrf_debug_attn_core_peak(int variant, int iters, int blocks, const float *seed, float *out, void *stream) {
    if (!out || !seed) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 0) hipLaunchKernelGGL((attn_core_peak_kernel<0>), dim3(blocks), dim3(256), 0, st, iters, seed, out);
    else hipLaunchKernelGGL((attn_core_peak_kernel<1>), dim3(blocks), dim3(256), 0, st, iters, seed, out);
    return nmrf_launch_status();
}

extern "C" int nmrf_debug_mfma_peak(int chains, int iters, int blocks, float *out, void *stream) {
    if (!out) return NMRF_ENULL;
    hipStream_t st = (hipStream_t)stream;
    if (chains == 1) hipLaunchKernelGGL((mfma_peak_kernel<1>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 2) hipLaunchKernelGGL((mfma_peak_kernel<2>), dim3(blocks), dim3(256), 0, st, iters, out);
    else if (chains == 4) hipLaunchKernelGGL((mfma_peak_kernel<4>), dim3(blocks), dim3(256), 0, st, iters, out);
    else return NMRF_EINVAL;
    return nmrf_launch_status();
}
#endif  // NMRF_DEBUG_PROBES

// ---- split-operand fp16 MFMA self-test (split_mfma.h): one wave, out = A*B with A [32,K], B [K,32] row-major fp32 -------------
//   mode 0: the full split product (3 MFMAs per 16-deep chunk, two accumulators) -- accuracy + lane layout
//   mode 1: hi parts only (plain fp16 product) -- what the split buys, and how fp16 subnormal inputs are treated
//   mode 2: like 0 with the k slots in C/D order (split_kslot) on BOTH operands -- the order chained GEMMs use
//   mode 3: single-accumulator form (split8u / split_mma1): unscaled low parts, A multiplied by 2^10 like a packed weight
__global__ __launch_bounds__(64) void selftest_mfma_f16split_kernel(const float *__restrict__ A, const float *__restrict__ Bm,
                                                                   int K, int mode, float *__restrict__ out) {
    const int lane = threadIdx.x, i = lane & 31, hi = lane >> 5;
    f32x16 acc_hh, acc_x;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc_hh[r] = acc_x[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = k0 + (mode == 2 ? split_kslot(jj, hi) : 8 * hi + jj);
            av[jj] = A[i * K + k];
            bv[jj] = Bm[k * 32 + i];
        }
        h16x8 ah, al, bh, bl;
        if (mode == 3) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) av[jj] *= 1024.0f;
            split8u(av, ah, al);
            split8u(bv, bh, bl);
            split_mma1(ah, al, bh, bl, acc_hh);
            continue;
        }
        split8(av, ah, al);
        split8(bv, bh, bl);
        if (mode == 1) acc_hh = mfma16h(ah, bh, acc_hh);
        else split_mma(ah, al, bh, bl, acc_hh, acc_x);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        out[mfma_row(r, hi) * 32 + i] = mode == 3 ? acc_hh[r] * (1.0f / 1024.0f) : acc_hh[r] + SPLIT_LO_INV * acc_x[r];
}

extern "C" int nmrf_selftest_mfma_f16split(const float *A, const float *Bm, int K, int mode, float *out, void *stream) {
    if (!A || !Bm || !out) return NMRF_ENULL;
    if (K < 16 || K > 1024 || (K & 15) || mode < 0 || mode > 3) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_mfma_f16split_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, K, mode, out);
    return nmrf_launch_status();
}

// ---- LDS-DMA self-test: global_load_lds_dwordx4 places lane l's 16 bytes at (wave-uniform LDS base) + 16*l ----------------
__global__ __launch_bounds__(256) void selftest_lds_dma_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst) {
    __shared__ __attribute__((aligned(16))) float4 buf[256];
    const int t = threadIdx.x;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + blockIdx.x * 256 + t),
                                     (__attribute__((address_space(3))) void *)(buf + (t & ~63)), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    dst[blockIdx.x * 256 + t] = buf[t ^ 65];          // read through LDS across waves: dst[t] = src[t ^ 65]
}

extern "C" int nmrf_selftest_lds_dma(const float *src, float *dst, int n_float4, void *stream) {
    if (!src || !dst) return NMRF_ENULL;
    if (n_float4 < 256 || (n_float4 & 255)) return NMRF_EINVAL;
    hipLaunchKernelGGL(selftest_lds_dma_kernel, dim3(n_float4 / 256), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst));
    return nmrf_launch_status();
}

extern "C" const char *nmrf_strerror(int code) {
    switch (code) {
        case NMRF_OK: return "ok";
        case NMRF_EINVAL: return "invalid size or unsupported configuration";
        case NMRF_ELAUNCH: return "HIP kernel launch failed";
        case NMRF_ENULL: return "null pointer argument";
        default: return "unknown error";
    }
}

extern "C" int nmrf_abi_version(void) { return NMRF_ABI_VERSION; }
