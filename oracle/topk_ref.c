/* ORACLE (test infrastructure only): C restatement of the top-k tie order the
 * reference inherits from ATen's CPU kernel (aten/src/ATen/native/cpu/TopKImpl.h,
 * called at nmrf/models/DPN.py:125 with k=4, largest, sorted):
 *
 *   queue[j] = (value_j, j)
 *   if (k*64 <= n)  partial_sort(...)            -- never taken for n <= 255
 *   else { nth_element(q, q+k-1, q+n, gt); sort(q, q+k-1, gt); }
 *   gt(x,y) = (isnan(x) && !isnan(y)) || x > y
 *
 * std::nth_element / std::sort are libstdc++'s (bits/stl_algo.h): introselect
 * with median-of-3 pivot moved to `first`, unguarded Hoare partition, insertion
 * sort once the range is <= 3, heap-select fallback after 2*floor(log2 n)
 * partitions; std::sort of <= 16 elements is a plain insertion sort.
 * This file restates that published algorithm in C so that (a) it can be
 * pinned against torch.topk goldens (tests/golden/nms_cases.npz) and (b) the
 * HIP kernel (nmrf_amd/csrc/seed.hip) can follow the same steps one lane per
 * pixel.  The NMS of DPN.py:120-124 is included so the pair is one call.
 */
#include <math.h>
#include <stdint.h>

typedef struct { float v; int32_t i; } elem_t;

static inline int gt(elem_t x, elem_t y) {
    return (isnan(x.v) && !isnan(y.v)) || (x.v > y.v);
}

static void insertion_sort(elem_t *q, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i < last; ++i) {
        elem_t val = q[i];
        if (gt(val, q[first])) {
            for (int j = i; j > first; --j) q[j] = q[j - 1];
            q[first] = val;
        } else {
            int j = i;
            while (gt(val, q[j - 1])) { q[j] = q[j - 1]; --j; }
            q[j] = val;
        }
    }
}

static void adjust_heap(elem_t *q, int start, int hole, int len, elem_t val) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (gt(q[start + child], q[start + child - 1])) child--;
        q[start + hole] = q[start + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        q[start + hole] = q[start + child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && gt(q[start + parent], val)) {
        q[start + hole] = q[start + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    q[start + hole] = val;
}

static void heap_select(elem_t *q, int first, int middle, int last) {
    int len = middle - first;
    if (len >= 2) {
        for (int parent = (len - 2) / 2;; --parent) {
            adjust_heap(q, first, parent, len, q[first + parent]);
            if (parent == 0) break;
        }
    }
    for (int i = middle; i < last; ++i)
        if (gt(q[i], q[first])) {
            elem_t val = q[i];
            q[i] = q[first];
            adjust_heap(q, first, 0, len, val);
        }
}

static void nth_element(elem_t *q, int n, int nth) {
    int first = 0, last = n;
    if (first == last || nth == last) return;
    int depth = 0;
    for (int t = n; t > 1; t >>= 1) depth++;
    depth *= 2;
    while (last - first > 3) {
        if (depth == 0) {
            heap_select(q, first, nth + 1, last);
            elem_t t = q[first]; q[first] = q[nth]; q[nth] = t;
            return;
        }
        --depth;
        int a = first + 1, b = first + (last - first) / 2, c = last - 1, m;
        if (gt(q[a], q[b])) m = gt(q[b], q[c]) ? b : (gt(q[a], q[c]) ? c : a);
        else                m = gt(q[a], q[c]) ? a : (gt(q[b], q[c]) ? c : b);
        { elem_t t = q[first]; q[first] = q[m]; q[m] = t; }
        int lo = first + 1, hi = last;
        for (;;) {
            while (gt(q[lo], q[first])) ++lo;
            --hi;
            while (gt(q[first], q[hi])) --hi;
            if (!(lo < hi)) break;
            elem_t t = q[lo]; q[lo] = q[hi]; q[hi] = t;
            ++lo;
        }
        if (lo <= nth) first = lo; else last = lo;
    }
    insertion_sort(q, first, last);
}

/* prob [rows, n] -> seeds [rows, k] (int64).  n <= 256, k*64 > n. Returns 0 or -1. */
int oracle_nms_topk_f32(const float *prob, int64_t rows, int n, int k, float eps,
                        int do_nms, int64_t *seeds) {
    if (n < 1 || n > 256 || k < 1 || k > n || k * 64 <= n) return -1;
    elem_t q[256];
    for (int64_t r = 0; r < rows; ++r) {
        const float *p = prob + r * n;
        for (int j = 0; j < n; ++j) {
            float v = p[j];
            if (do_nms) {
                float l = j > 0 ? p[j - 1] : -INFINITY, rr = j + 1 < n ? p[j + 1] : -INFINITY;
                float m = fmaxf(fmaxf(l, rr), v);          /* max_pool1d(k=3,pad=1) */
                if (v != m && v > eps) v = eps;            /* DPN.py:121-124 */
            }
            q[j].v = v; q[j].i = j;
        }
        nth_element(q, n, k - 1);
        insertion_sort(q, 0, k - 1);                       /* std::sort(begin, begin+k-1) */
        for (int j = 0; j < k; ++j) seeds[r * k + j] = q[j].i;
    }
    return 0;
}
