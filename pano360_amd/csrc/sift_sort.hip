// KeyPointsFilter::removeDuplicatedSorted + the first-octave adjustment of SIFT's
// detectAndCompute, on the device.
//
// Reference: features.py:192-201 gets keypoints from OpenCV already in this order; what is
// restated is OpenCV's (PARITY UNPINNED: OpenCV is not in the reference repo): sort by x, y, size
// (descending), angle, response (descending), octave (descending); of keypoints that share
// x, y, size and angle keep the first; then, for firstOctave = -1, halve positions and sizes
// and shift the octave byte.
//
// On the host this was np.lexsort over six keys: 54 of the 69 ms of a 4K frame's
// detectAndCompute (135 k keypoints).  Here: ONE stable radix sort of (64-bit key, index)
// pairs on the two leading keys, x and y - rocPRIM's device radix sort, called directly (the
// ROCm library primitive for a plain key-value sort; rounds 3 - 5 went through the hipCUB
// compatibility layer) - and a kernel that orders the short runs of keypoints sharing
// a position by the other four keys (a stable insertion sort per run: the same order as six
// stable least-significant-key-first passes, which is what round 2 ran - a hundred launches
// per frame, 0.6 ms of launch latency); hand-written kernels around them build the
// order-preserving integer keys, flag the duplicates and compact the survivors.
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "common.h"

namespace {

// float -> uint32 whose unsigned order is the float's order (-0 counted as +0, as the
// comparisons of a lexsort count it)
__device__ __forceinline__ uint32_t ordered(float v) {
    const uint32_t b = __float_as_uint(v + 0.0f);
    return (b & 0x80000000u) ? ~b : b | 0x80000000u;
}

// the leading keys x, y as one 64-bit key, and the identity permutation
__global__ __launch_bounds__(256) void sift_keys_kernel(const pano_sift_keypoint *__restrict__ kp,
                                                        int n, const int *__restrict__ n_dev,
                                                        uint64_t *__restrict__ keys,
                                                        uint32_t *__restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    idx[i] = (uint32_t)i;
    // slots past the device-side count hold nothing: the largest key keeps them at the end
    if (n_dev && i >= min(*n_dev, n)) {
        keys[i] = ~0ull;
        return;
    }
    const pano_sift_keypoint k = kp[i];
    keys[i] = (uint64_t)ordered(k.x) << 32 | ordered(k.y);
}

// a < b in the order of the four trailing keys: size descending, angle, response descending,
// octave descending (equal: keep the arrival order - the sort is stable)
__device__ __forceinline__ bool sift_tail_less(const pano_sift_keypoint &a, const pano_sift_keypoint &b) {
    const uint32_t as = ~ordered(a.size), bs = ~ordered(b.size);
    if (as != bs) return as < bs;
    const uint32_t aa = ordered(a.angle), ba = ordered(b.angle);
    if (aa != ba) return aa < ba;
    const uint32_t ar = ~ordered(a.response), br = ~ordered(b.response);
    if (ar != br) return ar < br;
    const uint32_t ao = ~((uint32_t)a.octave ^ 0x80000000u), bo = ~((uint32_t)b.octave ^ 0x80000000u);
    return ao < bo;
}

// The thread at the head of a run of equal (x, y) keys orders the run by the trailing keys.
__global__ __launch_bounds__(256) void sift_runs_kernel(const pano_sift_keypoint *__restrict__ kp,
                                                        const uint64_t *__restrict__ keys, int n,
                                                        uint32_t *__restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    if (key == ~0ull || (i > 0 && keys[i - 1] == key) || i + 1 >= n || keys[i + 1] != key) return;
    int end = i + 2;
    while (end < n && keys[end] == key) ++end;
    for (int a = i + 1; a < end; ++a) {                  // stable insertion sort of idx[i .. end)
        const uint32_t cur = idx[a];
        const pano_sift_keypoint kc = kp[cur];
        int b = a;
        while (b > i && sift_tail_less(kc, kp[idx[b - 1]])) {
            idx[b] = idx[b - 1];
            --b;
        }
        idx[b] = cur;
    }
}

// 1 = the first of its (x, y, size, angle) group in sorted order
__global__ __launch_bounds__(256) void sift_flags_kernel(const pano_sift_keypoint *__restrict__ kp,
                                                         const uint32_t *__restrict__ idx, int n,
                                                         const int *__restrict__ n_dev,
                                                         int *__restrict__ flags) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (n_dev && i >= min(*n_dev, n)) {                 // sorted to the end: not a keypoint
        flags[i] = 0;
        return;
    }
    int keep = 1;
    if (i > 0) {
        const pano_sift_keypoint a = kp[idx[i - 1]], b = kp[idx[i]];
        keep = !(a.x == b.x && a.y == b.y && a.size == b.size && a.angle == b.angle);
    }
    flags[i] = keep;
}

// survivors to their places, with detectAndCompute's first-octave adjustment
__global__ __launch_bounds__(256) void sift_compact_kernel(
    const pano_sift_keypoint *__restrict__ kp, const uint32_t *__restrict__ idx,
    const int *__restrict__ flags, const int *__restrict__ pos, int n, int first_octave,
    float scale, pano_sift_keypoint *__restrict__ out, int *__restrict__ n_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *n_out = pos[i] + flags[i];         // flags past the count are 0
    if (!flags[i]) return;
    pano_sift_keypoint k = kp[idx[i]];
    k.octave = (k.octave & ~255) | ((k.octave + first_octave) & 255);
    k.x *= scale;
    k.y *= scale;
    k.size *= scale;
    out[pos[i]] = k;
}

struct SortLayout {
    size_t temp_bytes, keys_a, keys_b, idx_a, idx_b, flags, pos, total;
};

SortLayout sort_layout(int n) {
    SortLayout L = {};
    size_t sort_bytes = 0, scan_bytes = 0;
    uint64_t *k64 = nullptr;
    uint32_t *ku = nullptr;
    int *iu = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, k64, k64, ku, ku, (size_t)n, 0u, 64u);   // size queries
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, iu, iu, 0, (size_t)n, rocprim::plus<int>());
    L.temp_bytes = ((sort_bytes > scan_bytes ? sort_bytes : scan_bytes) + 255) & ~(size_t)255;
    const size_t arr = ((size_t)n * 4 + 255) & ~(size_t)255;
    L.keys_a = L.temp_bytes;                            // 64-bit keys: two arrays' worth each
    L.keys_b = L.keys_a + 2 * arr;
    L.idx_a = L.keys_b + 2 * arr;
    L.idx_b = L.idx_a + arr;
    L.flags = L.idx_b + arr;
    L.pos = L.flags + arr;
    L.total = L.pos + arr;
    return L;
}

}  // namespace

extern "C" size_t pano_sift_sort_work_bytes(int n) { return sort_layout(n > 0 ? n : 1).total; }

extern "C" int pano_sift_sort_unique(pano_ctx *ctx, const pano_sift_keypoint *kpts, int n,
                                     const int *n_dev, int first_octave, void *work,
                                     pano_sift_keypoint *out, int *n_out) {
    PANO_ENTER(ctx, "pano_sift_sort_unique");
    PANO_REQUIRE(n >= 0 && n_out, "pano_sift_sort_unique: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        PANO_HIP(hipMemsetAsync(n_out, 0, sizeof(int), s));
        return PANO_OK;
    }
    PANO_REQUIRE(kpts && work && out && kpts != out, "pano_sift_sort_unique: null pointer");
    PANO_REQUIRE(first_octave >= -8 && first_octave <= 8, "pano_sift_sort_unique: first octave %d",
                 first_octave);
    const SortLayout L = sort_layout(n);
    unsigned char *base = (unsigned char *)work;
    uint64_t *keys_a = (uint64_t *)(base + L.keys_a), *keys_b = (uint64_t *)(base + L.keys_b);
    uint32_t *idx_b = (uint32_t *)(base + L.idx_a), *idx_a = (uint32_t *)(base + L.idx_b);
    int *flags = (int *)(base + L.flags), *pos = (int *)(base + L.pos);
    const dim3 grid(ceil_div(n, 256)), block(256);
    hipLaunchKernelGGL(sift_keys_kernel, grid, block, 0, s, kpts, n, n_dev, keys_a, idx_b);
    PANO_LAUNCH_CHECK("sift_keys_kernel");
    {
        size_t temp = L.temp_bytes;
        PANO_HIP(rocprim::radix_sort_pairs(base, temp, keys_a, keys_b, idx_b, idx_a, (size_t)n, 0u,
                                           64u, s));
    }
    hipLaunchKernelGGL(sift_runs_kernel, grid, block, 0, s, kpts, keys_b, n, idx_a);
    PANO_LAUNCH_CHECK("sift_runs_kernel");
    hipLaunchKernelGGL(sift_flags_kernel, grid, block, 0, s, kpts, idx_a, n, n_dev, flags);
    PANO_LAUNCH_CHECK("sift_flags_kernel");
    size_t temp = L.temp_bytes;
    PANO_HIP(rocprim::exclusive_scan(base, temp, flags, pos, 0, (size_t)n, rocprim::plus<int>(), s));
    const float scale = first_octave < 0 ? 1.0f / (float)(1 << -first_octave)
                                         : (float)(1 << first_octave);
    hipLaunchKernelGGL(sift_compact_kernel, grid, block, 0, s, kpts, idx_a, flags, pos, n,
                       first_octave, scale, out, n_out);
    PANO_LAUNCH_CHECK("sift_compact_kernel");
    return PANO_OK;
}
