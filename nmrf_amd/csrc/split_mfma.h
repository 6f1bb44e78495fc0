// Split-operand fp16 MFMA: fp32-grade products on the 2.5 PFLOP/s fp16 matrix pipe of gfx950.
//
// A fp32 value a is carried as two fp16 numbers
//     hi  = rn_f16(a)                      (11 significant bits)
//     lo' = rn_f16((a - hi) * 2^11)        (the next 11 bits, rescaled so that lo' lives in the same exponent range as hi:
//                                           no fp16 subnormals are needed, whatever the hardware does with them)
// and a product of two such pairs as   a*b ~= hi_a*hi_b + 2^-11 * (lo'_a*hi_b + hi_a*lo'_b)   (the dropped lo*lo term is
// 2^-22 relative).  Per 16-deep k chunk: three v_mfma_f32_32x32x16_f16 (32 cycles each) into TWO fp32 accumulators
// (acc_hh, acc_x) that are combined once at the end of the K loop: acc_hh + 2^-11 * acc_x.  The same contraction on the
// fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles per 2-deep step) costs 512 cycles: 5.3x more matrix-pipe time.
// Error per product ~3 * 2^-24 * |a*b| (fp32 FMA: 2^-24): sums of 128...512 terms stay well inside the 2e-5 kernel tolerances
// against fp64 that the fp32-MFMA kernels are held to (tests/test_hip_kernels.py).
// Range: |a| must stay below 65520 (fp16 max 65504); activations of this network are O(1..100), weights are rescaled by a power of
// two at pack time.  The range is GUARDED, not assumed: every activation split feeds a per-lane guard accumulator (split2u_g,
// one v_dot2c_f32_f16 per pair) that turns -inf / NaN forever once a value overflowed (or was NaN), and the kernel ORs a sticky
// device flag at its end (split_guard_commit) -- nmrf_amd.kernels.check_range raises NmrfHipError from it.  Below the range the
// error model is the one of the single-accumulator form further down: absolute 2^-25 per element.
//
// v_mfma_f32_32x32x16_f16 lane layout (gfx950; pinned by nmrf_selftest_mfma_f16split):
//   A operand: lane l holds A[i = l&31][k = 8*(l>>5) + 0..7]    (8 fp16 = 4 VGPRs)
//   B operand: lane l holds B[k = 8*(l>>5) + 0..7][j = l&31]
//   C/D      : reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]        (same as the fp32 form)
#pragma once
#include "common.h"

typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

#define SPLIT_LO_SCALE 2048.0f
#define SPLIT_LO_INV (1.0f / 2048.0f)

__device__ __forceinline__ f32x16 mfma16h(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// two fp32 -> packed (hi, lo') pairs: v_cvt_pk_f16_f32, 2 v_cvt_f32_f16, v_pk_add_f32, v_pk_mul_f32, v_cvt_pk_f16_f32
__device__ __forceinline__ void split2(f32x2 a, h16x2 &hi, h16x2 &lo) {
    hi = __builtin_convertvector(a, h16x2);
    const f32x2 back = __builtin_convertvector(hi, f32x2);
    lo = __builtin_convertvector((a - back) * SPLIT_LO_SCALE, h16x2);
}

// eight fp32 (v[0..7]) -> one MFMA operand pair
__device__ __forceinline__ void split8(const float *v, h16x8 &hi, h16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h16x2 h, l;
        split2(f32x2{v[2 * j], v[2 * j + 1]}, h, l);
        hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
        lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
    }
}

// one k chunk of the split product: acc_hh += Ah*Bh ; acc_x += Al*Bh + Ah*Bl
__device__ __forceinline__ void split_mma(h16x8 ah, h16x8 al, h16x8 bh, h16x8 bl, f32x16 &acc_hh, f32x16 &acc_x) {
    acc_hh = mfma16h(ah, bh, acc_hh);
    acc_x = mfma16h(al, bh, acc_x);
    acc_x = mfma16h(ah, bl, acc_x);
}

// ---- single-accumulator form ------------------------------------------------------------------------------------------------
// gfx950's fp16 MFMA honours subnormal inputs (probed by nmrf_selftest_mfma_f16split mode 1: 2^-20 * 2^10 = 2^-10 exactly), so
// the low part may stay unscaled, lo = rn_f16(a - hi), and all three products of a chunk go into ONE accumulator.  lo is a
// normal fp16 number while |a| >= 2^-3 (22 significant bits in hi + lo); below that it is subnormal with spacing 2^-24, i.e. an
// ABSOLUTE error <= 2^-25 = 3e-8 per element -- the size of one fp32 rounding of an O(1) accumulator.  Weights are multiplied
// by a power of two at pack time (largest entry in [2^13, 2^14), exact, undone in the epilogue), so their low parts are normal
// whatever their magnitude; activations are used as they are.  Half the accumulator registers, no combine step.
__device__ __forceinline__ void split2u(f32x2 a, h16x2 &hi, h16x2 &lo) {
    hi = __builtin_convertvector(a, h16x2);
#if defined(NMRF_MIX_SPLIT)
    // A/B form (tools/build_ab_flag.sh mix -DNMRF_MIX_SPLIT), NOT the product's: lo = rn_f16(a - hi) in one instruction per value --
    // v_fma_mixlo/hi_f16 reads hi as an fp16 source, forms fma(hi, -1, a) in fp32 (exact: a - hi is representable) and rounds once to
    // fp16: the bits of the five-instruction form below in 3 instructions per pair.  Measured in round 6: +0.35 % on the whole step
    // (profiles/r06g_mix_split_ab.txt) -- and WRONG RESULTS in one kernel (mlp_chain kind 1) once the surrounding schedule changed:
    // the compiler never forms the mixed instruction itself (it folds fma(x, -1, a) to a subtraction), so it has to be inline asm, and
    // the hazard recognizer does not see inside inline asm: a partial-register write (mixlo / mixhi write one half of the VGPR) needs a
    // wait state before the next VALU read of that register on this chip.  With `s_nop 0` behind each pair the results are right
    // again (profiles/r06j_mix_split_hazard.txt) and the gain is gone.  Kept for the record; not compiled into the product.
    {
        unsigned l;
        const unsigned h = __builtin_bit_cast(unsigned, hi);
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\ts_nop 0" : "=v"(l) : "v"(h), "v"(a[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0" : "+v"(l) : "v"(h), "v"(a[1]));
        lo = __builtin_bit_cast(h16x2, l);
    }
#elif defined(NMRF_SCALAR_SPLIT)
    // A/B build (tools/build_ab_nopk.sh): the same two subtractions as plain v_sub_f32 instead of one v_pk_add_f32 -- a packed fp32
    // instruction beside MFMAs costs more than its issue slot (MI355X_MICROARCH.md, "price of one filler beside MFMAs")
    const float b0 = (float)hi[0], b1 = (float)hi[1];
    const float l0 = a[0] - b0, l1 = a[1] - b1;
    lo = h16x2{(_Float16)l0, (_Float16)l1};
#else
    lo = __builtin_convertvector(a - __builtin_convertvector(hi, f32x2), h16x2);
#endif
}
__device__ __forceinline__ void split8u(const float *v, h16x8 &hi, h16x8 &lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h16x2 h, l;
        split2u(f32x2{v[2 * j], v[2 * j + 1]}, h, l);
        hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
        lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
    }
}
// ---- range guard --------------------------------------------------------------------------------------------------------------
// g += hi . lo (v_dot2c_f32_f16).  While |a| < 65520: a finite number of no meaning (|hi * lo| <= 2^-11 * 65504^2, sums of 1e5 terms
// stay below 1e12).  Once a value rounds to hi = +-inf, lo = a - hi = -+inf (or NaN): the product is -inf or NaN and g never
// becomes finite again; a NaN activation does the same.  One VALU instruction per pair on top of the five of the split.
__device__ __forceinline__ void split2u_g(f32x2 a, h16x2 &hi, h16x2 &lo, float &g) {
    split2u(a, hi, lo);
    g = __builtin_amdgcn_fdot2(hi, lo, g, false);
}
__device__ __forceinline__ void split8u_g(const float *v, h16x8 &hi, h16x8 &lo, float &g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h16x2 h, l;
        split2u_g(f32x2{v[2 * j], v[2 * j + 1]}, h, l, g);
        hi[2 * j] = h[0]; hi[2 * j + 1] = h[1];
        lo[2 * j] = l[0]; lo[2 * j + 1] = l[1];
    }
}
// end of kernel: OR bit `bit` into the caller's sticky flag word if this lane saw an out-of-range / NaN activation
__device__ __forceinline__ void split_guard_commit(float g, int *flag, int bit = 1) {
    if (flag && !(__builtin_fabsf(g) <= 3.0e38f)) atomicOr(flag, bit);
}

// acc += Al*Bh + Ah*Bl + Ah*Bh (small terms first)
__device__ __forceinline__ void split_mma1(h16x8 ah, h16x8 al, h16x8 bh, h16x8 bl, f32x16 &acc) {
    acc = mfma16h(al, bh, acc);
    acc = mfma16h(ah, bl, acc);
    acc = mfma16h(ah, bh, acc);
}

// Drop-in replacements of the two 16-step fp32 MFMA loops of the attention kernels (st = mfma32(kf[s], qf[s], st) and
// acc = mfma32(vf[s], p[s], acc), s = 0..15): chunk c, slot jj of half hi takes the place of MFMA k-slot (step s = 8c + jj,
// half hi), so every operand map of those kernels stays as it is; 2 x 3 MFMAs of 32 cycles instead of 16 of 64.
__device__ __forceinline__ void split_dot16(const float *a, const float *b, f32x16 &acc) {
    h16x8 ah[2], al[2], bh[2], bl[2];
    split8u(a, ah[0], al[0]);
    split8u(a + 8, ah[1], al[1]);
    split8u(b, bh[0], bl[0]);
    split8u(b + 8, bh[1], bl[1]);
    split_mma1(ah[0], al[0], bh[0], bl[0], acc);
    split_mma1(ah[1], al[1], bh[1], bl[1], acc);
}
__device__ __forceinline__ void split_dot16(const float *a, const f32x16 &b, f32x16 &acc) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = b[r];
    split_dot16(a, bv, acc);
}
// the same with the range guard on a (GB: on b as well -- b = softmax probabilities need none)
template <bool GB>
__device__ __forceinline__ void split_dot16_g(const float *a, const float *b, f32x16 &acc, float &g) {
    h16x8 ah[2], al[2], bh[2], bl[2];
    split8u_g(a, ah[0], al[0], g);
    split8u_g(a + 8, ah[1], al[1], g);
    if constexpr (GB) {
        split8u_g(b, bh[0], bl[0], g);
        split8u_g(b + 8, bh[1], bl[1], g);
    } else {
        split8u(b, bh[0], bl[0]);
        split8u(b + 8, bh[1], bl[1]);
    }
    split_mma1(ah[0], al[0], bh[0], bl[0], acc);
    split_mma1(ah[1], al[1], bh[1], bl[1], acc);
}
template <bool GB>
__device__ __forceinline__ void split_dot16_g(const float *a, const f32x16 &b, f32x16 &acc, float &g) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = b[r];
    split_dot16_g<GB>(a, bv, acc, g);
}

// ---- kv16: k | v of an attention operand stored as split pairs by their producer (include/nmrf_hip.h, nmrf_nmp_block16_f32) ----------
// value c of a kv16 row's v third: hi | lo << 16 (include/nmrf_hip.h) -> the float it was split from (up to 2^-22 relative)
__device__ __forceinline__ float kv16_value(float w) {
    const unsigned u = __builtin_bit_cast(unsigned, w);
    return (float)__builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu)) + (float)__builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
}
// 16 such words (MFMA k-slot order s = 8c + jj) -> the hi / lo operand chunks: two v_perm_b32 per pair of values, no arithmetic
__device__ __forceinline__ void kv16_chunks(const float *w, h16x8 (&vh)[2], h16x8 (&vl)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        unsigned h[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned e0 = __builtin_bit_cast(unsigned, w[8 * c + 2 * k]), e1 = __builtin_bit_cast(unsigned, w[8 * c + 2 * k + 1]);
            h[k] = __builtin_amdgcn_perm(e1, e0, 0x05040100u);
            l[k] = __builtin_amdgcn_perm(e1, e0, 0x07060302u);
        }
        const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]), lv = make_uint4(l[0], l[1], l[2], l[3]);
        vh[c] = __builtin_bit_cast(h16x8, hv);
        vl[c] = __builtin_bit_cast(h16x8, lv);
    }
}

// k slot order of a B operand that is taken straight from a C/D result (and of every A operand contracted with it):
// slot jj (0..7) of half hi of k chunk c  <->  k = 16*c + (jj&3) + 8*(jj>>2) + 4*hi.  With this order the 16 registers of
// a 32-row D strip ARE two consecutive k chunks of the next contraction (regs 0-7 -> chunk 0, regs 8-15 -> chunk 1): no
// cross-lane movement between chained GEMMs.  Host-side weight packing uses the same map (pack_split_weight_kernel).
__host__ __device__ __forceinline__ int split_kslot(int jj, int hi) { return (jj & 3) + 8 * (jj >> 2) + 4 * hi; }
