// Lane layout probe of v_mfma_f32_4x4x4_16b_f16 (16 independent 4x4x4 products per wave) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o tools/_ab/mfma4x4.so tools/ab/mfma4x4.hip   (the .so suffix keeps the binary out of git)
// Hypothesis H0: block b = lane >> 2; A: lane 4b+i holds A_b[i][0..3]; B: lane 4b+j holds B_b[0..3][j]; D: lane 4b+j, register i = D_b[i][j].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const h4 *a, const h4 *b, f4 *c) {
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    c[threadIdx.x] = acc;
}
int main() {
    _Float16 ha[64][4], hb[64][4];
    float fa[64][4], fb[64][4];
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
        fa[l][q] = (float)((l * 7 + q * 3) % 11 - 5); fb[l][q] = (float)((l * 5 + q * 2) % 13 - 6);
        ha[l][q] = (_Float16)fa[l][q]; hb[l][q] = (_Float16)fb[l][q];
    }
    h4 *da, *db; f4 *dc; float out[64][4];
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dc, sizeof(out));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dc);
    hipMemcpy(out, dc, sizeof(out), hipMemcpyDeviceToHost);
    // H0 and its transposes
    const char *names[4] = {"H0: D lane 4b+j reg i = sum_k A[4b+i][k] B[4b+j][k]", "H1: D lane 4b+i reg j", "H2: A/B roles swapped, lane 4b+j reg i", "H3: swapped, lane 4b+i reg j"};
    for (int h = 0; h < 4; ++h) {
        double err = 0;
        for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            float s = 0;
            for (int kk = 0; kk < 4; ++kk) s += (h < 2 ? fa[4 * b + i][kk] * fb[4 * b + j][kk] : fb[4 * b + i][kk] * fa[4 * b + j][kk]);
            const float got = (h % 2 == 0) ? out[4 * b + j][i] : out[4 * b + i][j];
            err = fmax(err, fabs(got - s));
        }
        printf("%s : max err %.3g\n", names[h], err);
    }
    printf("lane 5 regs: %g %g %g %g\n", out[5][0], out[5][1], out[5][2], out[5][3]);
    return 0;
}
