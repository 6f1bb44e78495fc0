// Fixed cost of a kernel launch as a function of what a workgroup reserves: duration (HIP events around 200 back-to-back launches)
// of a kernel that does nothing, for the launch shapes of this library's big kernels at KITTI batch 1.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_ab/launch_floor.so tools/ab/launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int VGPR> __global__ void k(float *out, int n) {
    extern __shared__ float lds[];
    if (VGPR >= 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else if (VGPR >= 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    if (n < 0) { lds[threadIdx.x] = 1.f; out[blockIdx.x] = lds[0]; }
}
template <int VGPR> void run(const char *name, int grid, int block, size_t lds, float *out) {
    hipFuncSetAttribute((const void *)k<VGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<VGPR>, dim3(grid), dim3(block), lds, 0, out, 0);
    hipDeviceSynchronize();
    const int N = 200;
    hipEventRecord(e0);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<VGPR>, dim3(grid), dim3(block), lds, 0, out, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-62s grid %5d x %4d thr, %6zu B LDS: %6.2f us per launch\n", name, grid, block, lds, ms * 1e3 / N);
}
int main() {
    float *out; hipMalloc(&out, 1 << 20);
    run<0>("empty kernel, 1 small block per CU", 256, 64, 0, out);
    run<0>("empty, block-kernel grid, no LDS, few VGPRs", 234, 512, 0, out);
    run<240>("empty, block-kernel shape (234 x 512, 146 KB LDS, 240 VGPRs)", 234, 512, 146 * 1024, out);
    run<0>("empty, same but few VGPRs", 234, 512, 146 * 1024, out);
    run<240>("empty, same but no LDS", 234, 512, 0, out);
    run<160>("empty, window-kernel shape (416 x 640, 156 KB LDS, 160 VGPRs)", 416, 640, 156 * 1024, out);
    run<160>("empty, conv shape (960 x 256, 53 KB LDS, 160 VGPRs)", 960, 256, 53 * 1024, out);
    run<0>("empty, 7332 x 256 (seed-select shape)", 1833, 256, 0, out);
    run<0>("empty, 30000 x 256", 30000, 256, 0, out);
    return 0;
}
