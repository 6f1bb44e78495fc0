// Host check for the batched in_affine_of (branch next/in-affine-batched): old.h = main:nmrf_amd/csrc/in_affine.h with the function renamed
// in_affine_old, new.h = the branch version, both without their #include / #pragma lines; g++ -O2 t.cpp && ./a.out -> "20000 cases, 0 differ".
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <vector>
#define __device__
#define __forceinline__ inline
struct float2 { float x, y; };
#include "old.h"
#undef IN_AFFINE_CHUNK
#include "new.h"
int main() {
    srand(1);
    long bad = 0, n = 0;
    for (int chunks = 1; chunks <= 50; ++chunks)
        for (int rep = 0; rep < 400; ++rep) {
            int64_t HW = (int64_t)(chunks - 1) * 8192 + 1 + rand() % 8192;
            std::vector<float> w(2 * chunks + 64, NAN);          // NaN beyond the records: must never be read into the sums
            for (int c = 0; c < chunks; ++c) { w[2 * c] = (rand() / (float)RAND_MAX - 0.5f) * 8; w[2 * c + 1] = rand() / (float)RAND_MAX * 9000; }
            float s0, h0, s1, h1;
            in_affine_old(w.data(), chunks, HW, 1e-5f, s0, h0);
            in_affine_of(w.data(), chunks, HW, 1e-5f, s1, h1);
            ++n;
            if (memcmp(&s0, &s1, 4) || memcmp(&h0, &h1, 4)) { if (bad < 5) printf("chunks %d HW %ld: %a %a vs %a %a\n", chunks, (long)HW, s0, h0, s1, h1); ++bad; }
        }
    printf("%ld cases, %ld differ\n", n, bad);
    return bad != 0;
}
