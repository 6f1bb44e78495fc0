// VALU issue-rate probe (v_fma_f32 / v_pk_fma_f32, VGPR vs SGPR operands).  hipcc --offload-arch=gfx950 -O3 -o tools/_ab/pkbench.so tools/ab/pkbench.hip
// (the odd suffix keeps the binary out of git: *.so is ignored but travels with gpurun).  Output: profiles/r03s_filter_ab.txt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void k(float *out, const float *wsrc, int iters) {
    f2 a[8]; float s[8];
    for (int i = 0; i < 8; ++i) { a[i] = f2{(float)threadIdx.x, 1.f + i}; s[i] = threadIdx.x + i; }
    f2 x{1.0001f, 0.9999f};
    float w0 = wsrc[0], w1 = wsrc[1];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "v"(x[0]), "v"(x[1])); }
                if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(x)); }
                if (MODE == 2) { f2 ws{w0, w1}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "s"(ws), "v"(x)); }
                if (MODE == 3) { f2 ws{w0, w1}; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "s"(ws), "v"(x)); }
                if (MODE == 4) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(s[i]) : "s"(w0), "v"(x[1])); }
                if (MODE == 5) { asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x)); }
            }
        }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i][0] + a[i][1] + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, float *out, float *w, int lanes_fma) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 4;   // 4 blocks x 4 waves per CU = 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, w, iters); hipEventRecord(e1);
    hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = 4.0 * iters * 32;            // waves per SIMD x instructions per wave
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-44s %8.3f ms  %.2f cycles per wave-instruction at 2.4 GHz  (%.1f TFLOP/s)\n", name, ms, cyc / instr_per_simd,
           2.0 * lanes_fma * 64 * 32.0 * iters * blocks * 4 / (ms * 1e-3) / 1e12);
}
int main() {
    float *out, *w; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&w, 64); float hw[2] = {1.0001f, 0.9999f}; hipMemcpy(w, hw, 8, hipMemcpyHostToDevice);
    run<0>("v_fma_f32 vgpr", out, w, 1);
    run<4>("v_fma_f32 sgpr src0", out, w, 1);
    run<1>("v_pk_fma_f32 vgpr", out, w, 2);
    run<2>("v_pk_fma_f32 sgpr-pair src0", out, w, 2);
    run<3>("v_pk_fma_f32 sgpr src0 broadcast (op_sel_hi 0)", out, w, 2);
    run<5>("v_pk_mul_f32 vgpr", out, w, 2);
    return 0;
}
