// Overlap statistics of equalize_gains (stitcher.py:36-63) for a batch of camera
// pairs: for every pixel of frame i, frame j is sampled through the pair's
// homography exactly as cv2.warpPerspective(INTER_LINEAR, BORDER_TRANSPARENT)
// would sample its float32 RGBA image (semantics in include/pano360.h); where the
// sampled alpha is non-zero the pixel counts, and the colours of both frames are
// summed.  Frame j's RGBA image is never built: colour = lut255[u8], alpha =
// float32(hat_y * hat_x), as in the warp kernels.
//
// Sums are taken in double, per workgroup in a fixed order, written as partials
// and reduced by a second kernel in a fixed order: deterministic, and closer to
// the exact mean than the float32 pairwise sum of np.mean the reference runs
// (the two differ by ~1e-7 relative; the pixel count is exact).
#include "geom.h"

static constexpr int OV_TW = 64, OV_TH = 16;     // pixels per workgroup (4 rows per thread)

struct PairMap {
    double m[9];
    int bw0;
    // Fixed-point source coordinates of destination pixel (x, y): the numerators
    // use x = block start + offset as WarpPerspectiveInvoker does.
    __device__ __forceinline__ void at(int x, int y, long long &X, long long &Y) const {
        const double xb = (double)((x / bw0) * bw0), x1 = (double)(x % bw0), yd = (double)y;
        const double X0 = __dadd_rn(__dadd_rn(__dmul_rn(m[0], xb), __dmul_rn(m[1], yd)), m[2]);
        const double Y0 = __dadd_rn(__dadd_rn(__dmul_rn(m[3], xb), __dmul_rn(m[4], yd)), m[5]);
        const double W0 = __dadd_rn(__dadd_rn(__dmul_rn(m[6], xb), __dmul_rn(m[7], yd)), m[8]);
        double W = __dadd_rn(W0, __dmul_rn(m[6], x1));
        W = W != 0.0 ? __ddiv_rn(32.0, W) : 0.0;
        const double fX = __dmul_rn(__dadd_rn(X0, __dmul_rn(m[0], x1)), W);
        const double fY = __dmul_rn(__dadd_rn(Y0, __dmul_rn(m[3], x1)), W);
        X = (long long)rint(fmax(-2147483648.0, fmin(2147483647.0, fX)));
        Y = (long long)rint(fmax(-2147483648.0, fmin(2147483647.0, fY)));
    }
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void overlap_stats_kernel(
    const pano_camera *__restrict__ cams, const pano_pair *__restrict__ pairs, int h, int w,
    int bw0, const float *__restrict__ lut255, double *__restrict__ partials) {
    __shared__ float s_lut[256];
    __shared__ double s_red[4][3];
    __shared__ int s_skip;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const pano_pair pr = pairs[blockIdx.z];
    PairMap pm;
#pragma unroll
    for (int k = 0; k < 9; ++k) pm.m[k] = pr.minv[k];
    pm.bw0 = bw0;
    const pano_camera *ci = cams + pr.i, *cj = cams + pr.j;
    const int sw = cj->sw, sh = cj->sh;
    const int tx0 = blockIdx.x * OV_TW, ty0 = blockIdx.y * OV_TH;
    double *out = partials + ((size_t)blockIdx.z * gridDim.x * gridDim.y +
                              (size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
    s_lut[tid] = lut255[tid];
    if (tid == 0) {
        // A tile on which the denominator keeps its sign maps to the convex hull of
        // its mapped corners: when that hull (2 px of slack for the fixed-point
        // rounding) misses the source frame, no pixel of the tile is written.
        const int xe = min(tx0 + OV_TW, w) - 1, ye = min(ty0 + OV_TH, h) - 1;
        const int cx[4] = {tx0, xe, tx0, xe}, cy[4] = {ty0, ty0, ye, ye};
        double lo_x = 1e300, hi_x = -1e300, lo_y = 1e300, hi_y = -1e300;
        int pos = 0, neg = 0;
        for (int k = 0; k < 4; ++k) {
            const double d = pm.m[6] * cx[k] + pm.m[7] * cy[k] + pm.m[8];
            pos += d > 0.0;
            neg += d < 0.0;
            const double u = (pm.m[0] * cx[k] + pm.m[1] * cy[k] + pm.m[2]) / d;
            const double v = (pm.m[3] * cx[k] + pm.m[4] * cy[k] + pm.m[5]) / d;
            lo_x = fmin(lo_x, u); hi_x = fmax(hi_x, u);
            lo_y = fmin(lo_y, v); hi_y = fmax(hi_y, v);
        }
        const bool one_sign = pos == 4 || neg == 4;
        s_skip = one_sign && (hi_x < -2.0 || lo_x > (double)sw + 1.0 || hi_y < -2.0 ||
                              lo_y > (double)sh + 1.0);
    }
    __syncthreads();
    if (s_skip) {
        if (tid == 0) out[0] = out[1] = out[2] = 0.0;
        return;
    }

    const uint8_t *__restrict__ fi = ci->frame, *__restrict__ fj = cj->frame;
    const double *__restrict__ hat_x = cj->hat_x, *__restrict__ hat_y = cj->hat_y;
    const int x = tx0 + threadIdx.x;
    double cnt = 0.0, sum_i = 0.0, sum_j = 0.0;
    if (x < w) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = ty0 + threadIdx.y * 4 + r;
            if (y >= h) break;
            long long X, Y;
            pm.at(x, y, X, Y);
            const int sx = sat16((int)(X >> 5)), sy = sat16((int)(Y >> 5));
            if (sx < 0 || sx >= sw - 1 || sy < 0 || sy >= sh - 1) continue;   // BORDER_TRANSPARENT
            Taps tp;
            tp.x0 = sx; tp.x1 = sx + 1; tp.y0 = sy; tp.y1 = sy + 1;
            const float ax = (float)(int)(X & 31) * (1.0f / 32.0f);
            const float ay = (float)(int)(Y & 31) * (1.0f / 32.0f);
            tp.w00 = (1.0f - ay) * (1.0f - ax);
            tp.w01 = (1.0f - ay) * ax;
            tp.w10 = ay * (1.0f - ax);
            tp.w11 = ay * ax;
            if (alpha_at(hat_x, hat_y, tp) == 0.0f) continue;                  // stitcher.py:58
            const uint8_t *p00 = fj + ((size_t)sy * sw + sx) * 3, *p01 = p00 + 3;
            const uint8_t *p10 = p00 + (size_t)sw * 3, *p11 = p10 + 3;
            const uint8_t *q = fi + ((size_t)y * w + x) * 3;
            cnt += 1.0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sum_j += (double)lerp4(s_lut[p00[c]], s_lut[p01[c]], s_lut[p10[c]],
                                       s_lut[p11[c]], tp);
                sum_i += (double)s_lut[q[c]];
            }
        }
    }
    cnt = wave_sum(cnt);
    sum_i = wave_sum(sum_i);
    sum_j = wave_sum(sum_j);
    if (threadIdx.x == 0) {
        s_red[threadIdx.y][0] = cnt;
        s_red[threadIdx.y][1] = sum_i;
        s_red[threadIdx.y][2] = sum_j;
    }
    __syncthreads();
    if (tid < 3)
        out[tid] = (s_red[0][tid] + s_red[1][tid]) + (s_red[2][tid] + s_red[3][tid]);
}

// stats[pair] = sum of the pair's partials, in a fixed order.
__global__ __launch_bounds__(256) void overlap_reduce_kernel(const double *__restrict__ partials,
                                                             int nblk, double *__restrict__ stats) {
    __shared__ double s_red[4][3];
    const double *src = partials + (size_t)blockIdx.x * nblk * 3;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += src[(size_t)b * 3 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < 3; ++k) s_red[threadIdx.x >> 6][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < 3)
        stats[(size_t)blockIdx.x * 3 + threadIdx.x] =
            (s_red[0][threadIdx.x] + s_red[1][threadIdx.x]) +
            (s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

extern "C" int pano_overlap_blocks(int h, int w) {
    return h > 0 && w > 0 ? ceil_div(w, OV_TW) * ceil_div(h, OV_TH) : 0;
}

extern "C" int pano_overlap_stats(pano_ctx *ctx, const pano_camera *cams, const pano_pair *pairs,
                                  int n_pairs, int h, int w, int bw0, const float *lut255,
                                  double *partials, double *stats) {
    PANO_ENTER(ctx, "pano_overlap_stats");
    PANO_REQUIRE(cams && pairs && lut255 && partials && stats, "pano_overlap_stats: null pointer");
    PANO_REQUIRE(n_pairs >= 0 && h > 1 && w > 1 && bw0 > 0 && h < 32768 && w < 32768,
                 "pano_overlap_stats: bad sizes (%d pairs, %d x %d, block width %d)", n_pairs, h,
                 w, bw0);
    if (n_pairs == 0) return PANO_OK;
    PANO_REQUIRE(n_pairs <= 65535, "pano_overlap_stats: %d pairs exceed one launch", n_pairs);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(ceil_div(w, OV_TW), ceil_div(h, OV_TH), n_pairs);
    PANO_TIMED(PK_OVERLAP, s,
               hipLaunchKernelGGL(overlap_stats_kernel, grid, dim3(64, 4), 0, s, cams, pairs, h,
                                  w, bw0, lut255, partials));
    PANO_LAUNCH_CHECK("overlap_stats_kernel");
    hipLaunchKernelGGL(overlap_reduce_kernel, dim3(n_pairs), dim3(256), 0, s, partials,
                       (int)(grid.x * grid.y), stats);
    PANO_LAUNCH_CHECK("overlap_reduce_kernel");
    return PANO_OK;
}
