// How many workgroups of a given shape a CU REALLY holds at once (the occupancy API's answer is a calculation, not an observation):
// every block records s_memrealtime at its start and end plus its CU (XCC_ID, HW_ID) and spins ~20 us; the host counts the
// largest number of overlapping blocks per CU.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_ab/occupancy_probe.so tools/ab/occupancy_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
template <int VGPR> __global__ void k(unsigned long long *rec, int spin_ticks) {
    extern __shared__ float lds[];
    if (VGPR >= 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else if (VGPR >= 200) asm volatile("v_mov_b32 v207, 0" ::: "v207");
    else if (VGPR >= 168) asm volatile("v_mov_b32 v167, 0" ::: "v167");
    else if (VGPR >= 160) asm volatile("v_mov_b32 v159, 0" ::: "v159");
    else if (VGPR >= 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (spin_ticks < 0) lds[threadIdx.x] = 1.f;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));       // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // XCC_ID
        rec[blockIdx.x * 3 + 0] = t0;
        rec[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
        rec[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | (hw & 0xff00);      // se_id | sh_id | cu_id
        if (blockIdx.x == 0) rec[4096 * 3] = __builtin_amdgcn_s_memtime() - c0;          // shader cycles of block 0's spin
    }
}
template <int VGPR> void run(const char *name, int grid, int block, size_t lds, unsigned long long *rec) {
    hipFuncSetAttribute((const void *)k<VGPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int api = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<VGPR>, block, lds);
    hipMemset(rec, 0, grid * 24);
    hipLaunchKernelGGL(k<VGPR>, dim3(grid), dim3(block), lds, 0, rec, 2000);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 3);
    hipMemcpy(h.data(), rec, grid * 24, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
    for (int b = 0; b < grid; ++b) {
        ev[h[b * 3 + 2]].push_back({h[b * 3 + 0], +1});
        ev[h[b * 3 + 2]].push_back({h[b * 3 + 1], -1});
    }
    int worst = 0, best = 1 << 30;
    for (auto &kv : ev) {
        std::sort(kv.second.begin(), kv.second.end());
        int cur = 0, mx = 0;
        for (auto &e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
        worst = std::max(worst, mx); best = std::min(best, mx);
    }
    unsigned long long cyc = 0;
    hipMemcpy(&cyc, rec + 4096 * 3, 8, hipMemcpyDeviceToHost);
    printf("%-44s %4d thr %6zu B LDS %3d VGPRs: API says %d / CU; observed max concurrent per CU %d (min over %zu CUs %d); shader clock while "
           "spinning %.0f MHz\n", name, block, lds, VGPR, api, worst, ev.size(), best, (double)cyc / ((double)(h[1] - h[0]) / 100.0));
}
int main() {
    unsigned long long *rec; hipMalloc(&rec, 4096 * 24 + 8);
    run<168>("conv3x3<2,3,1> shape", 2048, 256, 53824, rec);
    run<168>("same, 53760 B", 2048, 256, 53760, rec);
    run<168>("same, 53248 B (52 KB)", 2048, 256, 53248, rec);
    run<168>("same, 52224 B (51 KB)", 2048, 256, 52224, rec);
    run<168>("same, 49152 B (48 KB)", 2048, 256, 49152, rec);
    run<168>("same, 40960 B", 2048, 256, 40960, rec);
    run<160>("160 VGPRs, 53824 B", 2048, 256, 53824, rec);
    run<160>("160 VGPRs, 49152 B", 2048, 256, 49152, rec);
    run<128>("128 VGPRs, 53824 B", 2048, 256, 53824, rec);
    run<128>("128 VGPRs, 49152 B", 2048, 256, 49152, rec);
    run<128>("128 VGPRs, 40960 B", 2048, 256, 40960, rec);
    run<128>("128 VGPRs, 32768 B", 2048, 256, 32768, rec);
    run<0>("few VGPRs, 53824 B", 2048, 256, 53824, rec);
    run<0>("few VGPRs, 40960 B", 2048, 256, 40960, rec);
    run<0>("few VGPRs, 32768 B", 2048, 256, 32768, rec);
    run<0>("few VGPRs, 16384 B", 4096, 256, 16384, rec);
    run<200>("conv3x3<3,3,1> / <4,3,1> shape (2 / CU)", 2048, 256, 2 * 4 * 3 * 2048 + 27200 + 2048, rec);
    run<240>("block kernel shape", 1024, 512, 146 * 1024, rec);
    run<160>("window kernel shape (640 thr)", 1024, 640, 156 * 1024, rec);
    run<160>("window one-window shape (320 thr, 77.6 KB)", 2048, 320, 79462, rec);
    run<128>("8 waves 128 VGPRs 64 KB", 2048, 512, 65536, rec);
    run<128>("8 waves 128 VGPRs 40 KB", 2048, 512, 40960, rec);
    return 0;
}
