// Keypoints and descriptors of OpenCV's SIFT on the scale space of pyramid.hip
// (features.py:192-201: cv2.xfeatures2d.SIFT_create().detectAndCompute).
// The arithmetic is inside OpenCV, not in the reference repository: OpenCV 3.4 / 4.x
// xfeatures2d/src/sift.cpp is restated with the SIFT_create() defaults - parity
// unpinned, checked against the SIFT oracle under oracle/ (an independent NumPy restatement).
//
//   sift_scan_kernel +    findScaleSpaceExtrema + adjustLocalExtrema: the 26-neighbour test as a
//   sift_refine_kernel    stream over the DoG rows (a load per lane and layer, the triples from
//                         the neighbouring lanes' registers), the extrema listed per wave in LDS;
//                         the Newton refinement, contrast and edge tests on the listed ones
//                         (sift_extrema_kernel: the same per thread, for other layer counts).
//   sift_orient_kernel    calcOrientationHist: one wave per candidate, the lanes walk the
//                         (2r+1)^2 window and add into a 36-bin LDS histogram; smoothing
//                         and peak picking emit one keypoint per dominant orientation.
//   sift_describe_kernel  calcSIFTDescriptor: one wave per keypoint, trilinear votes into
//                         the 6 x 6 x 10 LDS histogram, then clip / scale / saturate.
//
// The orientation and descriptor kernels run four such waves per workgroup, each with its
// own keypoint and LDS arrays and no barrier between them (wave_sync below).
// Lists are filled with atomics, so their order varies from run to run; sift_sort.hip sorts
// the keypoints the way KeyPointsFilter::removeDuplicatedSorted does before they are described.
// The histograms (orientation, descriptor) are summed in 64-bit fixed point with integer LDS
// atomics: independent of the order of the additions, and several times faster than LDS float
// atomics on gfx950 (see sift_describe_kernel).  The window samples' arithmetic uses the
// hardware's exp / sqrt / reciprocal and fused multiply-adds (tolerance-tested against the
// oracle, as everything here).
#include <type_traits>

#include "common.h"

#define SIFT_BORDER 5
#define SIFT_MAX_STEPS 5
#define SIFT_ORI_BINS 36
#define SIFT_D 4
#define SIFT_N 8
#define SIFT_VOTE_SCALE 16777216.0f  // histogram votes are summed in 64-bit fixed point, 2^-24 units

// cv::fastAtan2: degrees in [0, 360), 0.3 degree accuracy
__device__ __forceinline__ float fast_atan2(float y, float x) {
    const float p1 = (float)(0.9997878412794807 * 57.29577951308232);
    const float p3 = (float)(-0.3258083974640975 * 57.29577951308232);
    const float p5 = (float)(0.1555786518463281 * 57.29577951308232);
    const float p7 = (float)(-0.04432655554792128 * 57.29577951308232);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, ax + 2.220446049250313e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = __fdiv_rn(ax, ay + 2.220446049250313e-16f);
        c2 = c * c;
        a = 90.0f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

// The same polynomial for the window samples of the orientation and descriptor kernels, with
// the quotient min / max formed once, by the hardware reciprocal (1 ulp; the polynomial itself
// is good to 0.3 degrees, and a sample's votes are continuous in its angle), instead of one
// correctly rounded division per branch: 25 vector instructions fewer per sample.
__device__ __forceinline__ float fast_atan2_sample(float y, float x) {
#pragma clang fp contract(fast)
    const float p1 = (float)(0.9997878412794807 * 57.29577951308232);
    const float p3 = (float)(-0.3258083974640975 * 57.29577951308232);
    const float p5 = (float)(0.1555786518463281 * 57.29577951308232);
    const float p7 = (float)(-0.04432655554792128 * 57.29577951308232);
    const float ax = fabsf(x), ay = fabsf(y);
    const float lo = fminf(ax, ay), hi = fmaxf(ax, ay);
    const float c = lo * __builtin_amdgcn_rcpf(hi + 2.220446049250313e-16f), c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (ax < ay) a = 90.0f - a;
    if (x < 0.0f) a = 180.0f - a;
    if (y < 0.0f) a = 360.0f - a;
    return a;
}

// exp and sqrt of a sample's weight and gradient magnitude: the hardware's (v_exp_f32 on
// x log2 e, v_sqrt_f32: 1 ulp) instead of the library's range-checked forms - the arguments
// are bounded (weights of a window, gradients of an image in [0, 255]).
__device__ __forceinline__ float exp_sample(float x) {
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
__device__ __forceinline__ float sqrt_sample(float x) { return __builtin_amdgcn_sqrtf(x); }

// The four neighbours of octave pixel (y, x), read at 32-bit offsets from the layer's base.
typedef const __attribute__((address_space(1))) float *layer_ptr;
// pixel at BYTE offset `off` of a layer (a layer is below 2^32 bytes; a 32-bit byte offset from a
// scalar base is one address operand, a 64-bit index is an add with carry per load)
__device__ __forceinline__ float layer_at(layer_ptr img, uint32_t off) {
    return *(layer_ptr)((const __attribute__((address_space(1))) char *)img + off);
}

// The orientation and descriptor kernels give every WAVE its own keypoint and its own LDS
// arrays; waves of a workgroup never talk to each other, so nothing needs s_barrier - a wave's
// LDS instructions execute in issue order, and this keeps the compiler from reordering them
// across the points where lanes read what other lanes wrote.  (As one-wave workgroups that
// called __syncthreads() the kernels ran at two waves per SIMD - SQ_WAVE_CYCLES over the busy
// time - and spent four fifths of their cycles waiting for their gathers.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define SIFT_WAVES 4                     // waves (keypoints in flight) per workgroup

// Matx33f::solve(b, DECOMP_LU): float Gaussian elimination with partial pivoting
__device__ __forceinline__ bool solve3(float a[3][3], float b[3], float x[3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int k = i;
#pragma unroll
        for (int j = i + 1; j < 3; ++j)
            if (fabsf(a[j][i]) > fabsf(a[k][i])) k = j;
        if (fabsf(a[k][i]) < 1.1920929e-06f) return false;          // FLT_EPSILON * 10
        if (k != i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float t = a[i][j];
                a[i][j] = a[k][j];
                a[k][j] = t;
            }
            const float t = b[i];
            b[i] = b[k];
            b[k] = t;
        }
        const float d = __fdiv_rn(-1.0f, a[i][i]);
#pragma unroll
        for (int j = i + 1; j < 3; ++j) {
            const float alpha = a[j][i] * d;
#pragma unroll
            for (int c = i + 1; c < 3; ++c) a[j][c] = a[j][c] + alpha * a[i][c];
            b[j] = b[j] + alpha * b[i];
        }
    }
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        float s = b[i];
#pragma unroll
        for (int k = i + 1; k < 3; ++k) s = s - a[i][k] * x[k];
        x[i] = __fdiv_rn(s, a[i][i]);
    }
    return true;
}

// adjustLocalExtrema of one scale-space extremum (layer0, r0, c0), the contrast and edge
// tests, and the keypoint record appended to cands.
__device__ __forceinline__ void sift_refine(
    const float *__restrict__ dog, int rows, int cols, int octv, int n_layers,
    float contrast_thr, float edge_thr, float sigma, int layer0, int r0, int c0,
    pano_sift_keypoint *__restrict__ cands, int *__restrict__ count, int max_cands) {
    const size_t plane = (size_t)rows * cols;
#define DOG(l, r, c) dog[(size_t)(l) * plane + (size_t)(r) * cols + (c)]
    // adjustLocalExtrema
    const float img_scale = 1.0f / 255.0f, deriv_scale = img_scale * 0.5f;
    const float second_scale = img_scale, cross_scale = img_scale * 0.25f;
    int r = r0, c = c0, layer = layer0;
    float xi = 0.0f, xr = 0.0f, xc = 0.0f;
    int step = 0;
    for (; step < SIFT_MAX_STEPS; ++step) {
        float dd[3] = {(DOG(layer, r, c + 1) - DOG(layer, r, c - 1)) * deriv_scale,
                       (DOG(layer, r + 1, c) - DOG(layer, r - 1, c)) * deriv_scale,
                       (DOG(layer + 1, r, c) - DOG(layer - 1, r, c)) * deriv_scale};
        const float v2 = DOG(layer, r, c) * 2.0f;
        const float dxx = (DOG(layer, r, c + 1) + DOG(layer, r, c - 1) - v2) * second_scale;
        const float dyy = (DOG(layer, r + 1, c) + DOG(layer, r - 1, c) - v2) * second_scale;
        const float dss = (DOG(layer + 1, r, c) + DOG(layer - 1, r, c) - v2) * second_scale;
        const float dxy = (DOG(layer, r + 1, c + 1) - DOG(layer, r + 1, c - 1) -
                           DOG(layer, r - 1, c + 1) + DOG(layer, r - 1, c - 1)) * cross_scale;
        const float dxs = (DOG(layer + 1, r, c + 1) - DOG(layer + 1, r, c - 1) -
                           DOG(layer - 1, r, c + 1) + DOG(layer - 1, r, c - 1)) * cross_scale;
        const float dys = (DOG(layer + 1, r + 1, c) - DOG(layer + 1, r - 1, c) -
                           DOG(layer - 1, r + 1, c) + DOG(layer - 1, r - 1, c)) * cross_scale;
        float h[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float x[3] = {0.0f, 0.0f, 0.0f};
        if (!solve3(h, dd, x)) x[0] = x[1] = x[2] = 0.0f;
        xi = -x[2];
        xr = -x[1];
        xc = -x[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        const float big = (float)(2147483647 / 3);
        if (fabsf(xi) > big || fabsf(xr) > big || fabsf(xc) > big) return;
        c += (int)rintf(xc);
        r += (int)rintf(xr);
        layer += (int)rintf(xi);
        if (layer < 1 || layer > n_layers || c < SIFT_BORDER || c >= cols - SIFT_BORDER ||
            r < SIFT_BORDER || r >= rows - SIFT_BORDER)
            return;
    }
    if (step >= SIFT_MAX_STEPS) return;
    {
        const float d0 = (DOG(layer, r, c + 1) - DOG(layer, r, c - 1)) * deriv_scale;
        const float d1 = (DOG(layer, r + 1, c) - DOG(layer, r - 1, c)) * deriv_scale;
        const float d2 = (DOG(layer + 1, r, c) - DOG(layer - 1, r, c)) * deriv_scale;
        const float t = d0 * xc + d1 * xr + d2 * xi;
        const float contr = DOG(layer, r, c) * img_scale + t * 0.5f;
        if (fabsf(contr) * n_layers < contrast_thr) return;
        const float v2 = DOG(layer, r, c) * 2.0f;
        const float dxx = (DOG(layer, r, c + 1) + DOG(layer, r, c - 1) - v2) * second_scale;
        const float dyy = (DOG(layer, r + 1, c) + DOG(layer, r - 1, c) - v2) * second_scale;
        const float dxy = (DOG(layer, r + 1, c + 1) - DOG(layer, r + 1, c - 1) -
                           DOG(layer, r - 1, c + 1) + DOG(layer, r - 1, c - 1)) * cross_scale;
        const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
        if (det <= 0.0f || tr * tr * edge_thr >= (edge_thr + 1.0f) * (edge_thr + 1.0f) * det) return;
        const int slot = atomicAdd(count, 1);
        if (slot >= max_cands) return;
        const float scale = (float)(1 << octv);
        pano_sift_keypoint k;
        k.x = ((float)c + xc) * scale;
        k.y = ((float)r + xr) * scale;
        k.size = sigma * powf(2.0f, __fdiv_rn((float)layer + xi, (float)n_layers)) * scale * 2.0f;
        k.angle = 0.0f;
        k.response = fabsf(contr);
        k.octave = octv + (layer << 8) + ((int)rintf((xi + 0.5f) * 255.0f) << 16);
        k.r = r;
        k.c = c;
        cands[slot] = k;
    }
#undef DOG
}

// findScaleSpaceExtrema, one thread per DoG pixel and layer: the general form (any number of
// layers per octave); the refinement in the same thread.
__global__ __launch_bounds__(256) void sift_extrema_kernel(
    const float *__restrict__ dog, int rows, int cols, int octv, int n_layers, int threshold,
    float contrast_thr, float edge_thr, float sigma, pano_sift_keypoint *__restrict__ cands,
    int *__restrict__ count, int max_cands) {
    const int c0 = blockIdx.x * 64 + threadIdx.x + SIFT_BORDER;
    const int r0 = blockIdx.y * 4 + threadIdx.y + SIFT_BORDER;
    const int layer0 = blockIdx.z + 1;
    if (c0 >= cols - SIFT_BORDER || r0 >= rows - SIFT_BORDER) return;
    const size_t plane = (size_t)rows * cols;
#define DOG(l, r, c) dog[(size_t)(l) * plane + (size_t)(r) * cols + (c)]
    const float val = DOG(layer0, r0, c0);
    if (!(fabsf(val) > (float)threshold)) return;
    bool is_max = val > 0.0f, is_min = val < 0.0f;
    for (int dl = -1; dl <= 1; ++dl)
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const float v = DOG(layer0 + dl, r0 + dy, c0 + dx);
                is_max &= val >= v;
                is_min &= val <= v;
            }
#undef DOG
    if (!(is_max || is_min)) return;
    sift_refine(dog, rows, cols, octv, n_layers, contrast_thr, edge_thr, sigma, layer0, r0, c0,
                cands, count, max_cands);
}

// The same search as a stream (three layers per octave, SIFT_create()'s default): a wave owns
// 62 columns (+ one halo column either side) and walks a segment of rows; per row it loads ONE
// value per lane and DoG layer, takes the maxima / minima of the row's triples from its
// neighbours' registers, and keeps the last three rows of those per layer: a sample is an
// extremum iff it equals the maximum (minimum) of the 27 values around it.  5/3 loads per
// sample instead of up to 27, no divergence: extrema only go on a list (packed layer, row,
// column), which sift_refine_kernel then works through, one thread per entry.  (One thread
// per sample with the refinement inside it took 1.58 ms for the DoG planes of a 4K frame, more
// than building the scale space.)
#ifndef SIFT_SCAN_WAVES
#define SIFT_SCAN_WAVES 4096        // a launch's rows are cut until it has this many waves (if it can)
#endif
template <int NL>                   // DoG layers of the octave = layers per octave + 2
__global__ __launch_bounds__(256) void sift_scan_kernel(const float *__restrict__ dog, int rows,
                                                        int cols, float threshold, int seg_rows,
                                                        uint32_t *__restrict__ raw,
                                                        int *__restrict__ raw_count, int cap) {
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int c = SIFT_BORDER + 62 * (int)blockIdx.x + lane - 1, cl = min(c, cols - 1);
    const int seg = (int)blockIdx.y * 4 + wave;
    // A wave walks `seg_rows` rows of its 62 columns, a chain of dependent row loads: 96 rows a
    // wave made every octave's launch last a hundred load latencies, however small the octave
    // (0.85 ms for the eleven octaves of a 4K frame); the host cuts the rows until the launch
    // has a few thousand waves.
    const int r_begin = SIFT_BORDER + seg * seg_rows;
    const int r_end = min(r_begin + seg_rows, rows - SIFT_BORDER);
    if (r_begin >= r_end) return;                        // wave-uniform
    const bool mine = lane >= 1 && lane <= 62 && c < cols - SIFT_BORDER;
    const size_t plane = (size_t)rows * cols;
    // the wave's extrema are collected in LDS and appended to the list 192 or more at a time:
    // one returning atomic per (row, layer) with an extremum - seven in ten of them - was a
    // million atomics on one word per 4K frame and set the kernel's duration (2.0 ms)
    __shared__ uint32_t s_found[4][256];
    uint32_t *found = s_found[wave];
    int n_found = 0;                                     // wave-uniform
    auto flush = [&]() {
        if (n_found == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(raw_count, n_found);
        base = __shfl(base, 0, 64);
        for (int k = lane; k < n_found; k += 64)
            if (base + k < cap) raw[base + k] = found[k];
        n_found = 0;
    };
    float hmx[NL][3], hmn[NL][3], ctr[NL][3];
    auto load_row = [&](const int y, auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        const float *row = dog + (size_t)min(y, rows - 1) * cols + cl;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const float v = row[l * plane];
            const float a = __shfl_up(v, 1, 64), b = __shfl_down(v, 1, 64);
            hmx[l][S] = fmaxf(v, fmaxf(a, b));
            hmn[l][S] = fminf(v, fminf(a, b));
            ctr[l][S] = v;
        }
    };
    auto eval_row = [&](const int y, auto slot_c) {     // the row whose centre values are in slot S
        constexpr int S = decltype(slot_c)::value;
        if (y >= r_end) return;                          // wave-uniform
        float vmx[NL], vmn[NL];
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            vmx[l] = fmaxf(hmx[l][0], fmaxf(hmx[l][1], hmx[l][2]));
            vmn[l] = fminf(hmn[l][0], fminf(hmn[l][1], hmn[l][2]));
        }
#pragma unroll
        for (int l = 1; l < NL - 1; ++l) {
            const float val = ctr[l][S];
            const float M = fmaxf(vmx[l - 1], fmaxf(vmx[l], vmx[l + 1]));
            const float m = fminf(vmn[l - 1], fminf(vmn[l], vmn[l + 1]));
            const bool ext = mine && fabsf(val) > threshold &&
                             ((val > 0.0f && val >= M) || (val < 0.0f && val <= m));
            const unsigned long long bal = __ballot(ext);
            if (bal) {                                   // wave-uniform
                if (ext)
                    found[n_found + __popcll(bal & ((1ull << lane) - 1ull))] =
                        (uint32_t)l << 28 | (uint32_t)y << 14 | (uint32_t)c;
                n_found += __popcll(bal);
                if (n_found > 192) flush();              // room for one more ballot of 64
            }
        }
    };
    using s0 = std::integral_constant<int, 0>;
    using s1 = std::integral_constant<int, 1>;
    using s2 = std::integral_constant<int, 2>;
    load_row(r_begin - 1, s0{});
    load_row(r_begin, s1{});
    for (int y = r_begin; y < r_end; y += 3) {
        load_row(y + 1, s2{});
        eval_row(y, s1{});
        load_row(y + 2, s0{});
        eval_row(y + 1, s2{});
        load_row(y + 3, s1{});
        eval_row(y + 2, s0{});
    }
    flush();
}

__global__ __launch_bounds__(256) void sift_refine_kernel(
    const float *__restrict__ dog, int rows, int cols, int octv, int n_layers, float contrast_thr,
    float edge_thr, float sigma, const uint32_t *__restrict__ raw, int *__restrict__ raw_count,
    int cap, pano_sift_keypoint *__restrict__ cands, int *__restrict__ count, int max_cands) {
    const int total = *raw_count;
    if (total > cap) {               // the list overflowed: make the caller's capacity check fail
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicMax(count, 0x7fffffff);
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const uint32_t w = raw[i];
        sift_refine(dog, rows, cols, octv, n_layers, contrast_thr, edge_thr, sigma, (int)(w >> 28),
                    (int)((w >> 14) & 0x3fffu), (int)(w & 0x3fffu), cands, count, max_cands);
    }
}

// One wave per candidate (blockDim 64).  A wave keeps the keypoints it finds (candidate index
// and angle) in LDS and reserves their slots in the output list a batch at a time: one
// returning atomic per candidate on the list's single counter - 100 k of them queuing on one
// address - kept every wave waiting (the kernel issued vector instructions in a quarter of its
// cycles; profiles/r03/pmc_cfg4_detect_summary.txt).
#define SIFT_ORI_VOTE_SCALE 1048576.0f   // 2^-20 units: a vote (< 361) fits 32 bits
#define SIFT_ORI_BATCH 96
__global__ __launch_bounds__(64 * SIFT_WAVES) void sift_orient_kernel(
    const float *const *__restrict__ gauss, const int *__restrict__ dims, int n_layers,
    const pano_sift_keypoint *__restrict__ cands, const int *__restrict__ n_cands, int max_cands,
    pano_sift_keypoint *__restrict__ kpts, int *__restrict__ count, int max_kpts) {
    __shared__ unsigned long long s_votes[SIFT_WAVES][SIFT_ORI_BINS];   // fixed point, as in the descriptor
    __shared__ float s_temp[SIFT_WAVES][SIFT_ORI_BINS + 4];
    __shared__ float s_hist[SIFT_WAVES][SIFT_ORI_BINS];
    __shared__ int s_out_idx[SIFT_WAVES][SIFT_ORI_BATCH + SIFT_ORI_BINS];
    __shared__ float s_out_angle[SIFT_WAVES][SIFT_ORI_BATCH + SIFT_ORI_BINS];
    __shared__ int s_out_base[SIFT_WAVES];
    const int total = min(*n_cands, max_cands);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long *votes = s_votes[wv];
    float *temp = s_temp[wv], *hist = s_hist[wv], *out_angle = s_out_angle[wv];
    int *out_idx = s_out_idx[wv];
    int n_out = 0;                                           // wave-uniform
    auto flush = [&]() {
        if (n_out == 0) return;
        if (lane == 0) s_out_base[wv] = atomicAdd(count, n_out);
        wave_sync();
        const int base = s_out_base[wv];
        for (int i = lane; i < n_out; i += 64) {
            if (base + i < max_kpts) {
                pano_sift_keypoint out = cands[out_idx[i]];
                out.angle = out_angle[i];
                kpts[base + i] = out;
            }
        }
        wave_sync();
        n_out = 0;
    };
    for (int idx = blockIdx.x * SIFT_WAVES + wv; idx < total; idx += gridDim.x * SIFT_WAVES) {
        const pano_sift_keypoint k = cands[idx];
        const int packed = __builtin_amdgcn_readfirstlane(k.octave);     // the wave's keypoint: scalars
        const int octv = packed & 255, layer = (packed >> 8) & 255;
        const int rows = dims[2 * octv], cols = dims[2 * octv + 1];
        const layer_ptr img = (layer_ptr)(gauss[octv] + (size_t)layer * rows * cols);
        const float scl = __fdiv_rn(k.size * 0.5f, (float)(1 << octv));
        const int radius = (int)rintf(4.5f * scl);
        const float sig = 1.5f * scl, expf_scale = __fdiv_rn(-1.0f, 2.0f * sig * sig);
        if (lane < SIFT_ORI_BINS) votes[lane] = 0;
        wave_sync();
        const int side = 2 * radius + 1;
        // t / side without an integer divide per sample: exact for t < 2^22 (side < 2048)
        const float inv_side = __fdiv_rn(1.0f, (float)side);
        const bool small = side < 2048;
        // a lane's samples are 64 apart in row-major order: (row, col) advance by 64 = q side + rem
        int row = lane / side, col = lane - row * side;
        const int q64 = 64 / side, rem64 = 64 - q64 * side;
        (void)inv_side;
        (void)small;
        for (int t = lane; t < side * side; t += 64, row += q64, col += rem64) {
#pragma clang fp contract(fast)              // the window samples' arithmetic may fuse (tolerance-tested)
            if (col >= side) { ++row; col -= side; }
            const int i = row - radius, j = col - radius;
            const int y = k.r + i, x = k.c + j;
            if (y <= 0 || y >= rows - 1 || x <= 0 || x >= cols - 1) continue;
            const uint32_t o = ((uint32_t)y * (uint32_t)cols + (uint32_t)x) << 2, pitch = (uint32_t)cols << 2;
            const float dx = layer_at(img, o + 4u) - layer_at(img, o - 4u);
            const float dy = layer_at(img, o - pitch) - layer_at(img, o + pitch);
            const float w = exp_sample((float)(i * i + j * j) * expf_scale);
            const float ori = fast_atan2_sample(dy, dx), mag = sqrt_sample(dx * dx + dy * dy);
            int bin = (int)rintf((SIFT_ORI_BINS / 360.0f) * ori);
            if (bin >= SIFT_ORI_BINS) bin -= SIFT_ORI_BINS;
            if (bin < 0) bin += SIFT_ORI_BINS;
            const unsigned vote = (unsigned)fmaxf(__builtin_fmaf(w * mag, SIFT_ORI_VOTE_SCALE, 0.5f), 0.0f);
            atomicAdd(&votes[bin], (unsigned long long)vote);
        }
        wave_sync();
        if (lane < SIFT_ORI_BINS + 4) {                  // circular padding of two bins either side
            const int b = (lane + SIFT_ORI_BINS - 2) % SIFT_ORI_BINS;
            temp[lane] = (float)(long long)votes[b] * (1.0f / SIFT_ORI_VOTE_SCALE);
        }
        wave_sync();
        float h = 0.0f;
        if (lane < SIFT_ORI_BINS) {
            const int t = lane + 2;
            h = (temp[t - 2] + temp[t + 2]) * (1.0f / 16.0f) +
                (temp[t - 1] + temp[t + 1]) * (4.0f / 16.0f) + temp[t] * (6.0f / 16.0f);
            hist[lane] = h;
        }
        float omax = h;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) omax = fmaxf(omax, __shfl_xor(omax, off, 64));
        wave_sync();
        bool peak = false;
        float angle = 0.0f;
        if (lane < SIFT_ORI_BINS) {
            const int l = lane > 0 ? lane - 1 : SIFT_ORI_BINS - 1;
            const int r2 = lane < SIFT_ORI_BINS - 1 ? lane + 1 : 0;
            const float hl = hist[l], hr = hist[r2];
            if (h > hl && h > hr && h >= omax * 0.8f) {
                float bin = (float)lane + __fdiv_rn(0.5f * (hl - hr), hl - 2.0f * h + hr);
                bin = bin < 0.0f ? SIFT_ORI_BINS + bin : (bin >= SIFT_ORI_BINS ? bin - SIFT_ORI_BINS : bin);
                angle = 360.0f - (360.0f / SIFT_ORI_BINS) * bin;
                if (fabsf(angle - 360.0f) < 1.1920929e-07f) angle = 0.0f;
                peak = true;
            }
        }
        const unsigned long long found = __ballot(peak);
        if (peak) {
            const int at = n_out + __popcll(found & ((1ull << lane) - 1ull));
            out_idx[at] = idx;
            out_angle[at] = angle;
        }
        n_out += __popcll(found);
        wave_sync();
        if (n_out >= SIFT_ORI_BATCH) flush();            // (at most SIFT_ORI_BINS more next time)
    }
    flush();
}

// One wave per keypoint.  kpts hold full-resolution coordinates (after the halving for
// first octave -1) and the adjusted packed octave, as SIFT::detectAndCompute returns them.
__global__ __launch_bounds__(64 * SIFT_WAVES) void sift_describe_kernel(
    const float *const *__restrict__ gauss, const int *__restrict__ dims, int first_octave,
    const pano_sift_keypoint *__restrict__ kpts, int n_cap, const int *__restrict__ n_dev,
    float *__restrict__ desc) {
    constexpr int d = SIFT_D, nb = SIFT_N;
    // The votes are summed in 64-bit fixed point (2^-24 units) with integer LDS atomics.  LDS
    // FLOAT atomics run far below the integer rate on gfx950: with ds_add_f32 the eight votes
    // of a sample were 84 % of the kernel (11.3 ms for the 135 k keypoints of a 4K frame, 1.8 ms
    // with the votes dropped, 2.9 ms with ds_add_u64; eight interleaved copies of a float
    // histogram against same-address conflicts: 9.4 ms).  The sums no longer depend on the
    // order of the additions; a vote is rounded to 6e-8, far below the final 8-bit rounding.
    // A sample's eight votes go to two neighbouring orientation bins of four (row, column)
    // cells: the two bins of a cell are summed by ONE 64-bit atomic, as two 32-bit fixed-point
    // numbers side by side.  For that a cell keeps its ten bins twice, as pairs: (0,1) (2,3) ..
    // (8,9) and (1,2) (3,4) .. (7,8); a vote pair (o, o + 1) goes to the first set when o is
    // even, to the second when it is odd, and a bin's sum is the sum of its two homes.  The
    // unit 2^-k is chosen per keypoint so that no 32-bit half can overflow: a bin collects at
    // most the samples of 2 x 2 cells, 36 scl^2 of them, each at most 361 (255 sqrt 2).
    // (Eight 64-bit atomics per sample: 2.85 ms for the 135 k keypoints of a 4K frame.)
    constexpr int PAIRS = 9, CELLS = (d + 2) * (d + 2);
    // SIFT_HCOPIES copies of a wave's histogram, one per group of 64 / SIFT_HCOPIES lanes: the 64
    // samples of a step are neighbours in the window and mostly vote into the same two or three
    // cells - same-address atomics, which the LDS serialises
#ifndef SIFT_HCOPIES
#define SIFT_HCOPIES 1                   // (2 or 4 copies against same-address atomics: no faster)
#endif
    constexpr int HSIZE = CELLS * PAIRS;
    __shared__ unsigned long long s_hist[SIFT_WAVES][SIFT_HCOPIES * HSIZE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long *hist = s_hist[wv];
    unsigned long long *mine = hist + (lane / (64 / SIFT_HCOPIES)) * HSIZE;
    const int n = n_dev ? min(*n_dev, n_cap) : n_cap;   // the count may still be on the device
    for (int idx = blockIdx.x * SIFT_WAVES + wv; idx < n; idx += gridDim.x * SIFT_WAVES) {
        const pano_sift_keypoint k = kpts[idx];
        // (the keypoint is the wave's: its octave, layer and with them the layer's base address
        // and size are scalars - the compiler cannot see that through threadIdx)
        const int packed = __builtin_amdgcn_readfirstlane(k.octave);
        int octave = packed & 255;
        const int layer = (packed >> 8) & 255;
        octave = octave < 128 ? octave : (-128 | octave);
        const float scale = octave >= 0 ? __fdiv_rn(1.0f, (float)(1 << octave)) : (float)(1 << -octave);
        const int o = octave - first_octave;
        const int rows = dims[2 * o], cols = dims[2 * o + 1];
        const layer_ptr img = (layer_ptr)(gauss[o] + (size_t)layer * rows * cols);
        float ori = 360.0f - k.angle;
        if (fabsf(ori - 360.0f) < 1.1920929e-07f) ori = 0.0f;
        const float scl = k.size * scale * 0.5f;
        const int px = (int)rintf(k.x * scale), py = (int)rintf(k.y * scale);
        float cos_t = cosf(ori * (float)(3.141592653589793 / 180.0));
        float sin_t = sinf(ori * (float)(3.141592653589793 / 180.0));
        const float bins_per_rad = nb / 360.0f, exp_scale = -1.0f / (d * d * 0.5f);
        const float hist_width = 3.0f * scl;
        int radius = (int)rintf(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
        radius = min(radius, (int)sqrt((double)cols * cols + (double)rows * rows));
        cos_t = __fdiv_rn(cos_t, hist_width);
        sin_t = __fdiv_rn(sin_t, hist_width);
        // 2^kbits units: 36 scl^2 samples x 361 x 2^kbits < 2^31
        int kbits = 31 - (int)ceilf(log2f(36.0f * fmaxf(scl * scl, 1.0f) * 361.0f));
        kbits = kbits > 24 ? 24 : (kbits < 0 ? 0 : kbits);
        const float to_fixed = exp2f((float)kbits), from_fixed = exp2f(-(float)kbits);
        for (int t = lane; t < SIFT_HCOPIES * HSIZE; t += 64) hist[t] = 0;
        wave_sync();
        const int side = 2 * radius + 1;
        // t / side without an integer divide per sample: exact for t < 2^22 (side < 2048)
        const float inv_side = __fdiv_rn(1.0f, (float)side);
        const bool small = side < 2048;
        // a lane's samples are 64 apart in row-major order: (row, col) advance by 64 = q side + rem
        int row = lane / side, col = lane - row * side;
        const int q64 = 64 / side, rem64 = 64 - q64 * side;
        (void)inv_side;
        (void)small;
        for (int t = lane; t < side * side; t += 64, row += q64, col += rem64) {
#pragma clang fp contract(fast)              // the window samples' arithmetic may fuse (tolerance-tested)
            if (col >= side) { ++row; col -= side; }
            const int i = row - radius, j = col - radius;
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r = py + i, c = px + j;
            if (!(rbin > -1.0f && rbin < d && cbin > -1.0f && cbin < d && r > 0 && r < rows - 1 &&
                  c > 0 && c < cols - 1))
                continue;
            const uint32_t at = ((uint32_t)r * (uint32_t)cols + (uint32_t)c) << 2, pitch = (uint32_t)cols << 2;
            const float dx = layer_at(img, at + 4u) - layer_at(img, at - 4u);
            const float dy = layer_at(img, at - pitch) - layer_at(img, at + pitch);
            const float w = exp_sample((c_rot * c_rot + r_rot * r_rot) * exp_scale);
            float obin = (fast_atan2_sample(dy, dx) - ori) * bins_per_rad;
            // the magnitude in the histogram's fixed-point unit from here on: the eight votes
            // are then one add and one conversion each
            const float mag = sqrt_sample(dx * dx + dy * dy) * w * to_fixed;
            const int r0 = (int)floorf(rbin), c0 = (int)floorf(cbin);
            int o0 = (int)floorf(obin);
            rbin -= r0;
            cbin -= c0;
            obin -= o0;
            if (o0 < 0) o0 += nb;
            if (o0 >= nb) o0 -= nb;
            const float v_r1 = mag * rbin, v_r0 = mag - v_r1;
            const float v_rc11 = v_r1 * cbin, v_rc10 = v_r1 - v_rc11;
            const float v_rc01 = v_r0 * cbin, v_rc00 = v_r0 - v_rc01;
            const float v111 = v_rc11 * obin, v110 = v_rc11 - v111;
            const float v101 = v_rc10 * obin, v100 = v_rc10 - v101;
            const float v011 = v_rc01 * obin, v010 = v_rc01 - v011;
            const float v001 = v_rc00 * obin, v000 = v_rc00 - v001;
            // pair index of (o0, o0 + 1) inside a cell: even o0 -> 0 .. 3 (.. 4), odd -> 5 .. 8
            const int pair = (o0 & 1) ? 5 + (o0 >> 1) : (o0 >> 1);
            const int cell = (r0 + 1) * (d + 2) + c0 + 1;
            auto vote = [&](const int which, const float lo, const float hi) {
                const unsigned long long a = (unsigned)fmaxf(lo + 0.5f, 0.0f);
                const unsigned long long b = (unsigned)fmaxf(hi + 0.5f, 0.0f);
                atomicAdd(&mine[which * PAIRS + pair], a | b << 32);
            };
            vote(cell, v000, v001);
            vote(cell + 1, v010, v011);
            vote(cell + (d + 2), v100, v101);
            vote(cell + (d + 3), v110, v111);
        }
        wave_sync();
        // circular orientation bins, then the 4 x 4 x 8 vector (two entries per lane)
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = lane + 64 * e, cell_q = q / nb, kk = q % nb;
            const int i = cell_q / d, j = cell_q % d;
            const unsigned long long *h = hist + ((i + 1) * (d + 2) + (j + 1)) * PAIRS;
            // bin b of a cell: low or high half of pair b >> 1 of the even set, and of the odd
            // set (pairs (1,2) (3,4) (5,6) (7,8) at 5 .. 8)
            auto bin = [&](const int b) -> long long {
                long long sum = 0;
#pragma unroll
                for (int cp = 0; cp < SIFT_HCOPIES; ++cp) {
                    const unsigned long long *hc = h + cp * HSIZE;
                    sum += (long long)((hc[b >> 1] >> (32 * (b & 1))) & 0xffffffffull);
                    if (b >= 1 && b <= 8)
                        sum += (long long)((hc[5 + ((b - 1) >> 1)] >> (32 * ((b - 1) & 1))) & 0xffffffffull);
                }
                return sum;
            };
            long long sum = bin(kk);
            if (kk < 2) sum += bin(nb + kk);
            v[e] = (float)sum * from_fixed;
        }
        float nrm2 = v[0] * v[0] + v[1] * v[1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm2 += __shfl_xor(nrm2, off, 64);
        const float thr = sqrtf(nrm2) * 0.2f;
        v[0] = fminf(v[0], thr);
        v[1] = fminf(v[1], thr);
        nrm2 = v[0] * v[0] + v[1] * v[1];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm2 += __shfl_xor(nrm2, off, 64);
        const float s = __fdiv_rn(512.0f, fmaxf(sqrtf(nrm2), 1.1920929e-07f));
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float q = rintf(v[e] * s);
            desc[(size_t)idx * (d * d * nb) + lane + 64 * e] = fminf(fmaxf(q, 0.0f), 255.0f);
        }
        wave_sync();
    }
}

extern "C" int pano_sift_extrema(pano_ctx *ctx, const float *dog, int rows, int cols, int octave,
                                 int n_layers, float contrast_thr, float edge_thr, float sigma,
                                 pano_sift_keypoint *cands, int *count, int max_cands) {
    PANO_ENTER(ctx, "pano_sift_extrema");
    PANO_REQUIRE(dog && cands && count, "pano_sift_extrema: null pointer");
    PANO_REQUIRE(rows > 0 && cols > 0 && n_layers >= 1 && octave >= 0 && octave < 32 && max_cands > 0,
                 "pano_sift_extrema: bad argument");
    if (rows <= 2 * SIFT_BORDER || cols <= 2 * SIFT_BORDER) return PANO_OK;
    const int threshold = (int)floor(0.5 * contrast_thr / n_layers * 255.0);
    if (n_layers == 3 && rows < 16384 && cols < 16384) {
        // the streaming search + the refinement of the listed extrema
        const hipStream_t s = (hipStream_t)stream;
        const size_t cap = (size_t)rows * cols / 4 > (1u << 20) ? (size_t)rows * cols / 4 : (1u << 20);
        if (cap > ctx->sift_raw_cap) {
            if (ctx->sift_raw) {
                PANO_HIP(hipStreamSynchronize(s));       // a queued kernel may still read it
                PANO_HIP(hipFree(ctx->sift_raw));
                ctx->sift_raw = nullptr;
            }
            PANO_HIP(hipMalloc((void **)&ctx->sift_raw, (cap + 1) * sizeof(uint32_t)));
            ctx->sift_raw_cap = cap;
        }
        int *raw_count = (int *)(ctx->sift_raw + ctx->sift_raw_cap);
        if (int rc = pano_zero_i32(s, raw_count, 1)) return rc;   // (a kernel: graph-safe, detect.hip)
        int seg_rows = 96;
        while (seg_rows > 12 && (long)ceil_div(cols - 2 * SIFT_BORDER, 62) *
                                    ceil_div(rows - 2 * SIFT_BORDER, seg_rows) < SIFT_SCAN_WAVES)
            seg_rows /= 2;
        dim3 block(64, 4), grid(ceil_div(cols - 2 * SIFT_BORDER, 62),
                                ceil_div(rows - 2 * SIFT_BORDER, 4 * seg_rows));
        PANO_TIMED(PK_SIFT_EXTREMA, s, {
            hipLaunchKernelGGL(sift_scan_kernel<5>, grid, block, 0, s, dog, rows, cols,
                               (float)threshold, seg_rows, ctx->sift_raw, raw_count,
                               (int)ctx->sift_raw_cap);
            hipLaunchKernelGGL(sift_refine_kernel, dim3(512), dim3(256), 0, s, dog, rows, cols, octave,
                               n_layers, contrast_thr, edge_thr, sigma, ctx->sift_raw, raw_count,
                               (int)ctx->sift_raw_cap, cands, count, max_cands);
        });
        PANO_LAUNCH_CHECK("sift_scan_kernel / sift_refine_kernel");
        return PANO_OK;
    }
    dim3 block(64, 4), grid(ceil_div(cols - 2 * SIFT_BORDER, 64), ceil_div(rows - 2 * SIFT_BORDER, 4),
                            n_layers);
    PANO_TIMED(PK_SIFT_EXTREMA, (hipStream_t)stream,
               hipLaunchKernelGGL(sift_extrema_kernel, grid, block, 0, (hipStream_t)stream, dog, rows,
                                  cols, octave, n_layers, threshold, contrast_thr, edge_thr, sigma,
                                  cands, count, max_cands));
    PANO_LAUNCH_CHECK("sift_extrema_kernel");
    return PANO_OK;
}

extern "C" int pano_sift_orient(pano_ctx *ctx, const float *const *gauss, const int *dims,
                                int n_layers, const pano_sift_keypoint *cands, const int *n_cands,
                                int max_cands, pano_sift_keypoint *kpts, int *count,
                                int max_kpts) {
    PANO_ENTER(ctx, "pano_sift_orient");
    PANO_REQUIRE(gauss && dims && cands && n_cands && kpts && count, "pano_sift_orient: null pointer");
    PANO_REQUIRE(max_cands > 0 && max_kpts > 0 && n_layers >= 1, "pano_sift_orient: bad argument");
    const int blocks = ceil_div(max_cands < 16384 ? max_cands : 16384, SIFT_WAVES);
    PANO_TIMED(PK_SIFT_ORIENT, (hipStream_t)stream,
               hipLaunchKernelGGL(sift_orient_kernel, dim3(blocks), dim3(64 * SIFT_WAVES), 0, (hipStream_t)stream,
                                  gauss, dims, n_layers, cands, n_cands, max_cands, kpts, count,
                                  max_kpts));
    PANO_LAUNCH_CHECK("sift_orient_kernel");
    return PANO_OK;
}

extern "C" int pano_sift_describe(pano_ctx *ctx, const float *const *gauss, const int *dims,
                                  int first_octave, const pano_sift_keypoint *kpts, int n,
                                  const int *n_dev, float *desc) {
    PANO_ENTER(ctx, "pano_sift_describe");
    PANO_REQUIRE(gauss && dims && (n == 0 || (kpts && desc)), "pano_sift_describe: null pointer");
    PANO_REQUIRE(n >= 0, "pano_sift_describe: bad count");
    if (n == 0) return PANO_OK;
    const int blocks = ceil_div(n < 65536 ? n : 65536, SIFT_WAVES);
    PANO_TIMED(PK_SIFT_DESCRIBE, (hipStream_t)stream,
               hipLaunchKernelGGL(sift_describe_kernel, dim3(blocks), dim3(64 * SIFT_WAVES), 0,
                                  (hipStream_t)stream, gauss, dims, first_octave, kpts, n, n_dev, desc));
    PANO_LAUNCH_CHECK("sift_describe_kernel");
    return PANO_OK;
}
