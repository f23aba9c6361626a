// Largest all-valid rectangle with the reference's exact tie-breaking.
//
// Reference arithmetic replaced: stitcher.py:340-369 (crop_mosaic).  For every
// row i (as bottom row) and column j the reference's candidate is the widest
// run around j whose column heights are >= heights[j]; it keeps the FIRST
// candidate in (row, column) scan order whose area is strictly larger than
// everything before.  Quirk kept: its right-extent loop never touches column 0
// (range(width-1, 0, -1), :359), so column 0's candidate is one pixel wide.
//
// Integer work, bit-exact target.  Three kernels:
//   1. column scan  : 16 threads per column walk a segment of the rows each -> heights[H][W]
//   2. row search   : one block per row; 64-column chunk minima in LDS let a
//                     thread skip whole chunks while looking for the nearest
//                     strictly smaller height on each side; candidates are
//                     packed as (area << 32 | ~(i*W + j)) and max-reduced, so
//                     the maximum key is the reference's winner
//   3. finalise     : one thread re-derives the winner's extents.
#include "common.h"

// heights[y][x] = length of the run of valid pixels that ends at (y, x) going up (:354).
// A block = 64 columns x CROP_HSEG waves; wave s walks rows [s Hs, (s + 1) Hs): once to find
// what its segment hands on (the run at its last row and whether the segment is valid
// throughout), then - the carries of the segments above combined through LDS - again to
// write.  (One thread per column walking all rows: 1.9 ms for config 5's 4948 rows.)
#define CROP_HSEG 16
__global__ __launch_bounds__(64 * CROP_HSEG) void crop_heights_kernel(
    const uint8_t *__restrict__ valid, int H, int W, int32_t *__restrict__ heights) {
    __shared__ int s_run[CROP_HSEG][64];
    __shared__ unsigned char s_all[CROP_HSEG][64];
    const int lane = threadIdx.x, seg = threadIdx.y;
    const int x = blockIdx.x * 64 + lane;
    const int hs = (H + CROP_HSEG - 1) / CROP_HSEG;
    const int y0 = seg * hs, y1 = min(y0 + hs, H);
    if (x < W) {
        int run = 0;
        bool all = true;
        for (int y = y0; y < y1; ++y) {
            const bool v = valid[(size_t)y * W + x] != 0;
            run = v ? run + 1 : 0;
            all &= v;
        }
        s_run[seg][lane] = run;
        s_all[seg][lane] = all ? 1 : 0;
    }
    __syncthreads();
    if (x >= W) return;
    int run = 0;                                        // the run that reaches this segment's first row
    for (int k = 0; k < seg; ++k) run = s_all[k][lane] ? run + s_run[k][lane] : s_run[k][lane];
    for (int y = y0; y < y1; ++y) {
        run = valid[(size_t)y * W + x] ? run + 1 : 0;                              // :354
        heights[(size_t)y * W + x] = run;
    }
}

#define CROP_CHUNK 64
#define CROP_MAX_CHUNKS 1024       // mosaic width up to 65536
#define CROP_SEG 2048              // columns of a row per workgroup

// nearest index left of j with height < h, or -1.  cmin: minima of 64-column chunks, smin:
// minima of 64-chunk super-chunks (a row of equal heights - the bottom rows of a closed sweep's
// mask - made the walk over chunk minima 720 steps long per column)
__device__ __forceinline__ int smaller_left(const int32_t *row, const int32_t *cmin,
                                            const int32_t *smin, int j, int h) {
    int idx = j - 1;
    const int cstart = (j / CROP_CHUNK) * CROP_CHUNK;
    while (idx >= cstart && row[idx] >= h) --idx;
    if (idx >= cstart) return idx;
    int c = j / CROP_CHUNK - 1;
    const int sstart = (c >= 0 ? c / CROP_CHUNK : 0) * CROP_CHUNK;    // first chunk of c's super-chunk
    while (c >= sstart && cmin[c] >= h) --c;
    if (c < sstart) {                                       // nothing in this super-chunk
        int s = sstart / CROP_CHUNK - 1;
        while (s >= 0 && smin[s] >= h) --s;
        if (s < 0) return -1;
        c = s * CROP_CHUNK + CROP_CHUNK - 1;
        while (cmin[c] >= h) --c;                           // this super-chunk holds a smaller one
    }
    idx = c * CROP_CHUNK + CROP_CHUNK - 1;
    while (row[idx] >= h) --idx;      // this chunk holds a smaller height
    return idx;
}

// nearest index right of j with height < h, or W
__device__ __forceinline__ int smaller_right(const int32_t *row, const int32_t *cmin,
                                             const int32_t *smin, int j, int h, int W, int nchunks) {
    int idx = j + 1;
    int cend = (j / CROP_CHUNK + 1) * CROP_CHUNK;
    if (cend > W) cend = W;
    while (idx < cend && row[idx] >= h) ++idx;
    if (idx < cend) return idx;
    int c = j / CROP_CHUNK + 1;
    int send = (j / CROP_CHUNK / CROP_CHUNK + 1) * CROP_CHUNK;       // one past the super-chunk's chunks
    if (send > nchunks) send = nchunks;
    if (c > send) send = c;
    while (c < send && cmin[c] >= h) ++c;
    if (c >= send) {
        if (send >= nchunks) return W;
        int s = send / CROP_CHUNK;
        const int nsuper = (nchunks + CROP_CHUNK - 1) / CROP_CHUNK;
        while (s < nsuper && smin[s] >= h) ++s;
        if (s >= nsuper) return W;
        c = s * CROP_CHUNK;
        while (cmin[c] >= h) ++c;
    }
    idx = c * CROP_CHUNK;
    while (row[idx] >= h) ++idx;
    return idx;
}

__global__ __launch_bounds__(256) void crop_rows_kernel(
    const int32_t *__restrict__ heights, int H, int W,
    unsigned long long *__restrict__ best) {
    __shared__ int32_t s_cmin[CROP_MAX_CHUNKS];
    __shared__ int32_t s_smin[CROP_MAX_CHUNKS / CROP_CHUNK];
    __shared__ unsigned long long s_red[4];
    // Bottom rows first: they hold the tallest columns and settle the best area early; a
    // candidate of height h is at most h W large, so everything that cannot reach the best
    // area found so far is skipped - whole rows (h <= i + 1) at the top of the mosaic.  Only
    // strictly smaller bounds are skipped: an equal area earlier in scan order must still win.
    // (Every (row, column) evaluated: 79 ms for the 4948 x 46 079 mask of config 5.)
    // grid = (segments of CROP_SEG columns, rows): the rows that survive the bound are few
    // (the bottom ~160 of config 3's 2474) and one workgroup per row left most CUs idle
    const int i = H - 1 - (int)blockIdx.y;
    const int j_begin = (int)blockIdx.x * CROP_SEG, j_end = min(j_begin + CROP_SEG, W);
    // The bound rises while other rows finish, so two waves of this block could read different
    // values: ONE thread reads it, and the whole block leaves or stays on that value (a wave
    // that left on its own would skip the barriers below and its share of the chunk minima).
    __shared__ unsigned long long s_floor;
    if (threadIdx.x == 0)
        s_floor = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32;
    __syncthreads();
    unsigned long long floor_area = s_floor;
    if ((unsigned long long)(i + 1) * (unsigned)W < floor_area) return;       // block-uniform
    const int32_t *row = heights + (size_t)i * W;
    const int nchunks = (W + CROP_CHUNK - 1) / CROP_CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < nchunks; c += 4) {
        const int j = c * CROP_CHUNK + lane;
        int v = j < W ? row[j] : 0x7fffffff;
        for (int off = 32; off > 0; off >>= 1) {
            int o = __shfl_xor(v, off);
            v = o < v ? o : v;
        }
        if (lane == 0) s_cmin[c] = v;
    }
    __syncthreads();
    for (int s = wave; s * CROP_CHUNK < nchunks; s += 4) {
        const int c = s * CROP_CHUNK + lane;
        int v = c < nchunks ? s_cmin[c] : 0x7fffffff;
        for (int off = 32; off > 0; off >>= 1) {
            int o = __shfl_xor(v, off);
            v = o < v ? o : v;
        }
        if (lane == 0) s_smin[s] = v;
    }
    __syncthreads();

    unsigned long long key = 0;
    for (int j = j_begin + (int)threadIdx.x; j < j_end; j += 256) {
        const int h = row[j];
        // the bound rises while other rows finish: re-read it (an L2 hit) before every search
        floor_area = __hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32;
        if (h == 0 || (unsigned long long)h * (unsigned)W < floor_area) continue;
        const int l = smaller_left(row, s_cmin, s_smin, j, h) + 1;
        const int r = j == 0 ? 0 : smaller_right(row, s_cmin, s_smin, j, h, W, nchunks) - 1;
        const unsigned long long area = (unsigned long long)(r - l + 1) * (unsigned)h;
        const unsigned long long pos = (unsigned long long)i * W + j;
        const unsigned long long k = (area << 32) | (0xffffffffull - pos);
        if (k > key) {
            key = k;
            if (area > floor_area) atomicMax(best, k);    // published at once: it prunes the others
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_xor(key, off);
        key = o > key ? o : key;
    }
    if (lane == 0) s_red[wave] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) key = s_red[k] > key ? s_red[k] : key;
        if (key) atomicMax(best, key);
    }
}

// One wave re-derives the winner's extents, 64 columns per step (one thread walking a
// 46 079-column row took 2.8 ms).
__global__ __launch_bounds__(64) void crop_finalize_kernel(const int32_t *__restrict__ heights, int H,
                                                          int W,
                                                          const unsigned long long *__restrict__ best,
                                                          int64_t *__restrict__ result) {
    const int lane = threadIdx.x;
    const unsigned long long key = *best;
    if (key == 0) {
        if (lane < 6) result[lane] = 0;
        return;
    }
    const unsigned long long pos = 0xffffffffull - (key & 0xffffffffull);
    const int i = (int)(pos / (unsigned)W), j = (int)(pos % (unsigned)W);
    const int32_t *row = heights + (size_t)i * W;
    const int h = row[j];
    int l = j, r = j;
    for (;;) {                                  // leftwards: first column below h stops the run
        const int c = l - 1 - lane;
        const unsigned long long stop = __ballot(c < 0 || row[c] < h);
        const int run = stop ? __ffsll((long long)stop) - 1 : 64;   // columns taken this step
        l -= run;
        if (run < 64) break;
    }
    if (j != 0)
        for (;;) {
            const int c = r + 1 + lane;
            const unsigned long long stop = __ballot(c >= W || row[c] < h);
            const int run = stop ? __ffsll((long long)stop) - 1 : 64;
            r += run;
            if (run < 64) break;
        }
    if (lane == 0) {
        result[0] = 1;
        result[1] = i - h + 1;                                                     // :369
        result[2] = l;
        result[3] = h;
        result[4] = r - l + 1;
        result[5] = (int64_t)(key >> 32);
    }
}

extern "C" int pano_crop_rect(pano_ctx *ctx, const uint8_t *valid, int H, int W, int32_t *heights,
                              int64_t *result) {
    PANO_ENTER_READONLY(ctx, "pano_crop_rect");
    PANO_REQUIRE(valid && heights && result, "pano_crop_rect: null pointer");
    PANO_REQUIRE(H > 0 && W > 0, "pano_crop_rect: bad shape %dx%d", H, W);
    PANO_REQUIRE(W <= CROP_CHUNK * CROP_MAX_CHUNKS, "pano_crop_rect: width %d above %d", W,
                 CROP_CHUNK * CROP_MAX_CHUNKS);
    PANO_REQUIRE((unsigned long long)H * W < 0xffffffffull,
                 "pano_crop_rect: %dx%d does not fit the 32-bit position key", H, W);
    hipStream_t s = (hipStream_t)stream;
    // result[5] doubles as the 64-bit max-reduction cell until finalise rewrites it
    unsigned long long *best = (unsigned long long *)(result + 5);
    PANO_HIP(hipMemsetAsync(best, 0, sizeof(unsigned long long), s));
    PANO_TIMED(PK_CROP_HEIGHTS, s, hipLaunchKernelGGL(crop_heights_kernel, dim3(ceil_div(W, 64)), dim3(64, CROP_HSEG), 0, s, valid,
                       H, W, heights));
    PANO_LAUNCH_CHECK("crop_heights_kernel");
    PANO_TIMED(PK_CROP_ROWS, s, hipLaunchKernelGGL(crop_rows_kernel, dim3(ceil_div(W, CROP_SEG), H), dim3(256), 0, s, heights, H, W, best));
    PANO_LAUNCH_CHECK("crop_rows_kernel");
    hipLaunchKernelGGL(crop_finalize_kernel, dim3(1), dim3(64), 0, s, heights, H, W, best,
                       result);
    PANO_LAUNCH_CHECK("crop_finalize_kernel");
    return PANO_OK;
}
