// Shared device/host helpers for the gfx950 kernels of libnmrf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/nmrf_hip.h"

#define NMRF_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// packed FMA (v_pk_fma_f32): a plain v_fma_f32 costs the same issue slot for half the work
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

static inline int nmrf_launch_status() {
    return hipGetLastError() == hipSuccess ? NMRF_OK : NMRF_ELAUNCH;
}

// per-device one-time state of the launchers (hipFuncSetAttribute opt-ins, CU counts): a process may drive several GPUs
#define NMRF_MAX_DEV 64
static inline int nmrf_cur_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NMRF_MAX_DEV) return -1;
    return dev;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- v_mfma_f32_32x32x2_f32 lane layout (MI355X guide, cdna_hip_programming.md section 3) -------------
//   A operand: lane l holds A[i = l&31][k = l>>5]          (one f32)
//   B operand: lane l holds B[k = l>>5][j = l&31]          (one f32)
//   C/D      : reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// mfma_row(r, hi) is that row map; the attention kernels use it for the "key index" of S^T and
// for the "channel index" of O^T.
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Cross-half exchange without touching the LDS crossbar (ds_bpermute queues behind the b128 traffic of the other
// waves: ~600 cycles per shuffle measured in the window kernel).  v_permlane32_swap exchanges lanes 32-63 of its
// first operand with lanes 0-31 of its second:  a' = [a.lo, b.lo]  b' = [a.hi, b.hi].
__device__ __forceinline__ void half_swap(float &a, float &b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
// combine lane l with lane l^32; both end up with the same value
__device__ __forceinline__ float half_max(float x) { float a = x, b = x; half_swap(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float x) { float a = x, b = x; half_swap(a, b); return a + b; }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The same two reductions without the LDS crossbar (six dependent ds_bpermute round trips each): v_permlane32_swap / v_permlane16_swap
// for the strides 32 and 16, row rotations (DPP) for 8, 4, 2, 1.  A rotation by o inside a row of 16 pairs lane l with lane l^o once
// the values have period 2o inside the row, which is what the previous level leaves behind; so the additions form the SAME tree as
// wave_sum's butterfly (fp add is commutative): identical bits, every lane holds the result.
#define NMRF_ROW_ROR(v, n) __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x120 + (n), 0xf, 0xf, false))
__device__ __forceinline__ float wave_sum_nolds(float v) {
    v = half_sum(v);
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    v += NMRF_ROW_ROR(v, 8);
    v += NMRF_ROW_ROR(v, 4);
    v += NMRF_ROW_ROR(v, 2);
    v += NMRF_ROW_ROR(v, 1);
    return v;
}
__device__ __forceinline__ float wave_max_nolds(float v) {
    v = half_max(v);
    {
        auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    v = fmaxf(v, NMRF_ROW_ROR(v, 8));
    v = fmaxf(v, NMRF_ROW_ROR(v, 4));
    v = fmaxf(v, NMRF_ROW_ROR(v, 2));
    v = fmaxf(v, NMRF_ROW_ROR(v, 1));
    return v;
}

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void stg4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// GELU(erf) = v Phi(v) with ONE transcendental: erfc(t / sqrt 2) = 2^(-t P(t)) for t = |v| in [0, 7], P the degree-5 minimax
// polynomial of -log2(erfc(t / sqrt 2)) / t weighted by the error it causes in v Phi(v) (5.3e-8 in exact arithmetic; tools/gelu_fit.py),
// and v Phi(v) = max(v, 0) - |v| erfc(|v| / sqrt 2) / 2 for either sign.  11 VALU instructions (v_min, 5 v_fma, v_mul, v_exp, v_max,
// v_mul, v_fma) against the 16 of the Abramowitz-Stegun 7.1.26 form it replaces (which also paid a v_rcp: transcendentals issue at a
// quarter of the rate; an explicitly packed two-value form measured the same, profiles/r06h_gelu_ab.txt) -- on this chip a SIMD's vector and matrix instructions issue one after the other (tools/ab/pipe_overlap.hip), so
// every instruction saved per hidden value is matrix-pipe time returned.  Over [-12, 12] in fp32: max |gelu_fast - gelu_fp64| = 3.0e-7
// (half an ulp at 4; the 7.1.26 form: 4.7e-7; torch's own fp32 GELU: 1.2e-6).  |v| > 7: erfc is held at its value at 7 (2.6e-12).
__device__ __forceinline__ float gelu_fast(float v) {
#ifndef NMRF_GELU_AS
    const float a = fabsf(v), t = fminf(a, 7.0f);
    float p = fmaf(-3.0177725420799106e-05f, t, 0.0007414184510707855f);
    p = fmaf(p, t, -0.007980725727975368f);
    p = fmaf(p, t, 0.053240980952978134f);
    p = fmaf(p, t, 0.45891475677490234f);
    p = fmaf(p, t, 1.151147484779358f);
    const float e = __builtin_amdgcn_exp2f(-t * p);
    return fmaf(e * t, -0.5f, fmaxf(v, 0.f));
#else       // A/B build (tools/build_ab_flag.sh gelu_as -DNMRF_GELU_AS): the rounds 2-5 form
    const float z = v * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(az * az * -1.4426950408889634f);
    const float er = copysignf(fmaf(-poly, e, 1.0f), z);
    return 0.5f * v * (1.0f + er);
#endif
}

// Fourier(31) of a disparity label (NMP.py: the label embedding of every stage): c = coord * normalizer; bands c * 2^f (exact), full-range
// sincosf.  16 work items per token (f = 0..15).
__device__ __forceinline__ void fourier_write(float coord, float normalizer, int f, float *row) {
    // f in [0,16): f<15 -> sin/cos of band f ; f==15 -> the scaled coordinate itself
    float c = coord * normalizer;
    if (f < 15) {
        float arg = c * (float)(1 << f);
        float s, co;
        sincosf(arg, &s, &co);
        row[f] = s;
        row[15 + f] = co;
    } else {
        row[30] = c;
    }
}

// rows wider than 31 floats (ld = 32: 16-byte aligned rows for the fused block kernel's side input): the pad columns are zero
__device__ __forceinline__ void fourier_pad(int f, float *row, int ld) {
    if (f == 15) for (int k = 31; k < ld; ++k) row[k] = 0.f;
}
