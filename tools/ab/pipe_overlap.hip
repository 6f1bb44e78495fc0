// Do the matrix pipe and the VALU of ONE SIMD run concurrently when they are fed by two DIFFERENT waves (a matrix-only wave beside a
// VALU-only wave), or does the SIMD issue one of them at a time?  (VERDICT r05 next #4b: the premise of wave-specialised block kernels.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/_ab/pipe_overlap.so tools/ab/pipe_overlap.hip      (the .so suffix keeps the binary out of git)
// One 512-thread block per CU: waves w and w + 4 share a SIMD (checked: HW_ID is printed).  Modes:
//   0  waves 0-3 matrix-only (16x16x32 f16, four independent accumulators), waves 4-7 exit
//   1  waves 0-3 VALU-only (v_fma_f32, eight independent chains), waves 4-7 exit
//   2  waves 0-3 matrix-only, waves 4-7 VALU-only                 <- the question
//   3  all eight waves matrix-only      4  all eight waves VALU-only
//   5  every wave alternates 1 MFMA + 4 VALU (what the product kernels look like), four waves     6  the same, eight waves
//   7  mode 2 with v_pk_fma_f32 as the VALU work      8  mode 1 with v_pk_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 2000

__device__ __forceinline__ void mfma_loop(float *sink, int n) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int it = 0; it < n; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma32_loop(float *sink, int n) {            // 2 x v_mfma_f32_32x32x16_f16 per iteration (32 cycles each)
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f16v c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
    for (int it = 0; it < n; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    }
    sink[threadIdx.x] = c0[0] + c1[5];
}
__device__ __forceinline__ void mfma4_loop(float *sink, int n) {             // 8 x v_mfma_f32_4x4x4_16b_f16 per iteration
    h4 a, b;
    for (int i = 0; i < 4; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < n; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][i & 3];
    sink[threadIdx.x] = s;
}
template <bool PK>
__device__ __forceinline__ void valu_loop(float *sink, int n) {
    float x = 1.0f + 1e-6f * threadIdx.x, y = 0.999f;
    if (PK) {
        f2 r[8];
        for (int i = 0; i < 8; ++i) r[i] = f2{0.1f * i, 0.2f * i};
        for (int it = 0; it < n; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = __builtin_elementwise_fma(r[i], f2{y, y}, f2{x, x});
        float s = 0;
        for (int i = 0; i < 8; ++i) s += r[i][0] + r[i][1];
        sink[threadIdx.x] = s;
    } else {
        float r[16];
        for (int i = 0; i < 16; ++i) r[i] = 0.1f * i;
        for (int it = 0; it < n; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i] = fmaf(r[i], y, x);
        float s = 0;
        for (int i = 0; i < 16; ++i) s += r[i];
        sink[threadIdx.x] = s;
    }
}
__device__ __forceinline__ void mixed_loop(float *sink, int n) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float r[16], x = 1.0f + 1e-6f * threadIdx.x, y = 0.999f;
    for (int i = 0; i < 16; ++i) r[i] = 0.1f * i;
    for (int it = 0; it < n; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = fmaf(r[i], y, x);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int i = 4; i < 8; ++i) r[i] = fmaf(r[i], y, x);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
#pragma unroll
        for (int i = 8; i < 12; ++i) r[i] = fmaf(r[i], y, x);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int i = 12; i < 16; ++i) r[i] = fmaf(r[i], y, x);
    }
    float s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int i = 0; i < 16; ++i) s += r[i];
    sink[threadIdx.x] = s;
}

__global__ __launch_bounds__(512) void k(int mode, float *sink, unsigned long long *t, unsigned *hwid) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *s = sink + (size_t)blockIdx.x * 512;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool first = wv < 4;
    switch (mode) {
        case 0: if (first) mfma_loop(s, ITER); break;
        case 1: if (first) valu_loop<false>(s, ITER); break;
        case 2: if (first) mfma_loop(s, ITER); else valu_loop<false>(s, ITER); break;
        case 3: mfma_loop(s, ITER); break;
        case 4: valu_loop<false>(s, ITER); break;
        case 5: if (first) mixed_loop(s, ITER); break;
        case 6: mixed_loop(s, ITER); break;
        case 7: if (first) mfma_loop(s, ITER); else valu_loop<true>(s, ITER); break;
        case 8: if (first) valu_loop<true>(s, ITER); break;
        case 9: if (first) mfma_loop(s, ITER); else { __builtin_amdgcn_s_setprio(3); valu_loop<false>(s, ITER); } break;
        case 10: if (first) valu_loop<false>(s, ITER); else mfma_loop(s, ITER); break;
        case 11: if (first) mfma32_loop(s, ITER); break;
        case 12: if (first) mfma32_loop(s, ITER); else valu_loop<false>(s, ITER); break;
        case 13: if (first) mfma4_loop(s, ITER); break;
        case 14: if (first) mfma4_loop(s, ITER); else valu_loop<false>(s, ITER); break;
        case 15: if (first) { __builtin_amdgcn_s_setprio(3); mfma_loop(s, ITER); } else valu_loop<false>(s, ITER); break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        t[(size_t)blockIdx.x * 8 + wv] = t1 - t0;
        if (blockIdx.x == 0) hwid[wv] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
    }
}

int main() {
    float *sink; unsigned long long *t; unsigned *hw;
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&t, 256 * 8 * 8); hipMalloc(&hw, 8 * 4);
    const char *names[16] = {"4 waves matrix-only (one per SIMD)", "4 waves VALU-only (v_fma_f32)", "4 matrix-only + 4 VALU-only, paired on the SIMDs",
                            "8 waves matrix-only", "8 waves VALU-only", "4 waves of 1 MFMA + 4 VALU interleaved", "8 waves of 1 MFMA + 4 VALU interleaved",
                            "4 matrix-only + 4 v_pk_fma-only", "4 waves v_pk_fma-only",
                            "mode 2 with the VALU waves at s_setprio 3", "4 VALU-only (older) + 4 matrix-only (younger)",
                            "4 waves 32x32x16 only (2 per iteration)", "4 x 32x32x16-only + 4 VALU-only", "4 waves 4x4x4 only (8 per iteration)",
                            "4 x 4x4x4-only + 4 VALU-only", "mode 2 with the MATRIX waves at s_setprio 3"};
    for (int mode = 0; mode < 16; ++mode) {
        hipMemset(t, 0, 256 * 8 * 8);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, sink, t, hw);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 8); unsigned hh[8];
        hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hh, hw, 32, hipMemcpyDeviceToHost);
        double a = 0, b = 0; int na = 0, nb = 0;
        for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) { if (w < 4) { a += h[i * 8 + w]; ++na; } else { b += h[i * 8 + w]; ++nb; } }
        if (mode == 0) { printf("SIMD of waves 0..7 of block 0:"); for (int w = 0; w < 8; ++w) printf(" %u", (hh[w] >> 4) & 3); printf("\n"); }
        // per iteration: 4 MFMAs (16 cycles each at one per SIMD) / 16 v_fma (or 8 v_pk_fma)
        printf("mode %d  %-52s waves 0-3: %7.1f cycles / iteration   waves 4-7: %7.1f\n", mode, names[mode], a / na / ITER, b / nb / ITER);
    }
    return 0;
}
