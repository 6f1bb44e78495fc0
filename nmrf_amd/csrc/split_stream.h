// Weight-fragment stream shared by the waves of a block (used by mlp_chain.hip; nmp_block.hip carries the same pipeline inline).
//
// A kernel's weights are packed (nmrf_pack_split_weight_f32) as "pairs": 2 KB = hi fragment + lo fragment of one
// (32-row strip, 16-deep k chunk), 64 lanes x 16 B each, concatenated in the order the kernel consumes them.  8 pairs = one
// 16 KB stage.  Stage s lives in LDS ring slot s % 3.  Timeline of a wave at stage g:
//     barrier g | consume the 8 pairs of stage g (each pair: A fragments from LDS, 3 MFMAs against a B operand held in registers),
//     reading fragments PF pairs ahead -- across the stage boundary: stage g+1 is visible since barrier g | commit stage g+2
//     (fetched into registers at the end of stage g-1) to the slot stage g-1 vacated | fetch stage g+3.
// Barrier g orders: every wave has finished reading stage g-1, and stage g+1 (committed during stage g-1) is visible.
#pragma once
#include "common.h"
#include "split_mfma.h"
#include <type_traits>
#include <utility>

typedef unsigned int ss_u32x4 __attribute__((ext_vector_type(4)));    // (HIP's uint4 is a struct: arrays of it are left in scratch)

#define SS_STAGE_U4 1024           // 16-byte words per stage (16 KB = 8 pairs)
#define SS_RING 3
#define SS_RING_BYTES (SS_RING * SS_STAGE_U4 * 16)

template <class F, int... I>
__device__ __forceinline__ void ss_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ss_static_for(F &&f) {
    ss_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

template <int PF>
struct SplitStream {
    const ss_u32x4 *stream;       // global
    ss_u32x4 *ring;               // LDS
    int total_stages, tid, lane;
    int src_stage, wr_slot, rd_slot;
    const ss_u32x4 *cur, *nxt;
    ss_u32x4 R[4];
    h16x8 fqh[PF], fql[PF];
    bool have_barrier;

    __device__ __forceinline__ void fetch() {
        const ss_u32x4 *p = stream + (size_t)src_stage * SS_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) R[i] = p[256 * i];
        src_stage = (src_stage + 1 == total_stages) ? 0 : src_stage + 1;
    }
    __device__ __forceinline__ void fetch_into(ss_u32x4 (&Rx)[4]) {
        const ss_u32x4 *p = stream + (size_t)src_stage * SS_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) Rx[i] = p[256 * i];
        src_stage = (src_stage + 1 == total_stages) ? 0 : src_stage + 1;
    }
    __device__ __forceinline__ void commit_from(const ss_u32x4 (&Rx)[4]) {
        ss_u32x4 *d = ring + wr_slot * SS_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[256 * i] = Rx[i];
        wr_slot = (wr_slot == SS_RING - 1) ? 0 : wr_slot + 1;
    }
    __device__ __forceinline__ void commit() {
        ss_u32x4 *d = ring + wr_slot * SS_STAGE_U4 + tid;
#pragma unroll
        for (int i = 0; i < 4; ++i) d[256 * i] = R[i];
        wr_slot = (wr_slot == SS_RING - 1) ? 0 : wr_slot + 1;
    }
    __device__ __forceinline__ void read_pair(const ss_u32x4 *base, int p, h16x8 &h, h16x8 &l) {
        h = *reinterpret_cast<const h16x8 *>(base + p * 128 + lane);
        l = *reinterpret_cast<const h16x8 *>(base + p * 128 + 64 + lane);
    }
    // call once, by all 256 threads, before any other LDS traffic that the first barrier should cover
    __device__ __forceinline__ void init(const void *stream_, void *ring_, int total_stages_, int tid_) {
        stream = reinterpret_cast<const ss_u32x4 *>(stream_);
        ring = reinterpret_cast<ss_u32x4 *>(ring_);
        total_stages = total_stages_; tid = tid_; lane = tid_ & 63;
        src_stage = 0; wr_slot = 0; rd_slot = 0;
        cur = ring; nxt = ring + SS_STAGE_U4;
        // stages 0, 1 -> slots 0, 1 and stage 2 (committed at the end of stage 0) requested TOGETHER: one L2 round trip instead of
        // three serial fetch -> commit pairs (streams of one stage wrap onto themselves)
        ss_u32x4 Ra[4], Rb[4];
        fetch_into(Ra); fetch_into(Rb); fetch();
        commit_from(Ra); commit_from(Rb);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < PF; ++p) read_pair(cur, p, fqh[p], fql[p]);
        have_barrier = true;
    }
    __device__ __forceinline__ void stage_top() {
        if (!have_barrier) __syncthreads();
        have_barrier = false;
    }
    __device__ __forceinline__ void stage_end() {
        commit();
        fetch();
        rd_slot = (rd_slot == SS_RING - 1) ? 0 : rd_slot + 1;
        cur = nxt;
        nxt = ring + ((rd_slot == SS_RING - 1) ? 0 : rd_slot + 1) * SS_STAGE_U4;
    }
    // pair P (compile-time, 0..7) of the current stage against the B operand (bh, bl): acc += A . B
    template <int P>
    __device__ __forceinline__ void consume(const h16x8 &bh, const h16x8 &bl, f32x16 &acc) {
        const h16x8 ah = fqh[P % PF], al = fql[P % PF];
        if constexpr (P + PF < 8) read_pair(cur, P + PF, fqh[P % PF], fql[P % PF]);
        else read_pair(nxt, P + PF - 8, fqh[P % PF], fql[P % PF]);
        // LDS reads may not sink below this point, MFMAs may not rise above it (ALU / VMEM may cross): see nmp_block.hip
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
        split_mma1(ah, al, bh, bl, acc);
    }
};

// running pair counter -> stage_top / consume / stage_end at the right places: PAIR is the index of this pair in the stream
template <int PAIR, int PF>
__device__ __forceinline__ void ss_pair(SplitStream<PF> &s, const h16x8 &bh, const h16x8 &bl, f32x16 &acc) {
    if constexpr (PAIR % 8 == 0) s.stage_top();
    s.template consume<PAIR % 8>(bh, bl, acc);
    if constexpr (PAIR % 8 == 7) s.stage_end();
}
