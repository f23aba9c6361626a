// Mosaic-side kernels: ownership/validity, and the three blenders, all written
// gather-style - one thread per mosaic pixel walks the patch table in index
// order - so there are no atomics, no H x W x N weight stack (the reference's
// stitcher.py:196 allocation) and the float sums run in the reference's patch
// order.
//
// Reference arithmetic replaced: stitcher.py:160-168 (no_blend), :171-183
// (linear_blend), :196-241 (multiband_blend minus the GaussianBlur calls),
// :266-271 (_valid).
//
// Roofline: HBM.  Per covered (pixel, patch) pair the multiband collapse reads
// 3 + 4(L-1) floats (the warped colour and the L-1 blurred RGBA copies),
// coalesced along x because every plane is planar; it writes 3 B (+12 B when
// the float mosaic is requested) per mosaic pixel.
#include "common.h"

__global__ __launch_bounds__(256) void ownership_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    int16_t *__restrict__ owner, uint8_t *__restrict__ valid) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float best = 0.0f;
    int who = -1;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        const float a = p.planes[3 * (size_t)p.h * p.pitch + (size_t)py * p.pitch + px];
        if (a > best) {          // strict: the first maximum keeps the pixel
            best = a;
            who = i;
        }
        any |= p.mask[(size_t)py * p.w + px] == 0;
    }
    owner[(size_t)y * W + x] = (int16_t)who;
    valid[(size_t)y * W + x] = any ? 1 : 0;
}

// uint8(255 * v) with C truncation; v is in [0, 1] up to rounding.
__device__ __forceinline__ uint8_t quant255(float v) { return (uint8_t)(int)(255.0f * v); }

template <int L>
__global__ __launch_bounds__(256) void multiband_compose_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    const int16_t *__restrict__ owner, const uint8_t *__restrict__ valid,
    uint8_t *__restrict__ mosaic, float *__restrict__ mosaic_f32) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float layer[L][3], wsum[L];
#pragma unroll
    for (int k = 0; k < L; ++k) layer[k][0] = layer[k][1] = layer[k][2] = wsum[k] = 0.0f;

    for (int i = 0; i < n; ++i) {
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        const size_t plane = (size_t)p.h * p.pitch, o = (size_t)py * p.pitch + px;
        float hi[3], ha = 0.0f;                 // the copy that gets the minus
#pragma unroll
        for (int c = 0; c < 3; ++c) hi[c] = p.planes[c * plane + o];
        if (L == 1) ha = owner[(size_t)y * W + x] == i ? 1.0f : 0.0f;   // sharp alpha (:208)
#pragma unroll
        for (int k = 0; k < L; ++k) {
            float rgb[3], a;
            if (k < L - 1) {
                const float *b = p.blurred + (size_t)k * 4 * plane + o;
                a = b[3 * plane];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g = b[c * plane];
                    rgb[c] = hi[c] - g;          // tile.rgb -= blur.rgb   (:227)
                    hi[c] = g;                   // prevs[idx] = blur      (:229)
                }
                ha = a;
            } else {                             // last level: G_{L-2} itself
                a = ha;
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[c] = hi[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) layer[k][c] = layer[k][c] + rgb[c] * a;   // :231
            wsum[k] = wsum[k] + a;                                                 // :232
        }
    }
    const bool ok = valid[(size_t)y * W + x] != 0;
    float out[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const float ws = wsum[k] == 0.0f ? 1.0f : wsum[k];                         // :237
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[c] = out[c] + __fdiv_rn(ok ? layer[k][c] : 0.0f, ws);              // :236,238
    }
    const size_t g = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = out[c];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                               // :240
        if (mosaic_f32) mosaic_f32[g + c] = v;
        mosaic[g + c] = quant255(v);                                               // :241
    }
}

__global__ __launch_bounds__(256) void linear_blend_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    uint8_t *__restrict__ mosaic) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        const size_t plane = (size_t)p.h * p.pitch, o = (size_t)py * p.pitch + px;
        const bool m = p.mask[(size_t)py * p.w + px] != 0;
        const float a = p.planes[3 * plane + o];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = m ? 0.0f : p.planes[c * plane + o];                    // :176
            acc[c] = acc[c] + t * a;                                               // :177
        }
        wsum = wsum + a;                                                           // :178
    }
    const float ws = wsum == 0.0f ? 1.0f : wsum;                                   // :180
    const size_t g = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) mosaic[g + c] = quant255(__fdiv_rn(acc[c], ws));   // :181-183
}

__global__ __launch_bounds__(256) void no_blend_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    uint8_t *__restrict__ mosaic) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    uint8_t out[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {               // later patches overwrite (:164-166)
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        if (p.mask[(size_t)py * p.w + px]) continue;
        const size_t plane = (size_t)p.h * p.pitch, o = (size_t)py * p.pitch + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = quant255(p.planes[c * plane + o]);
    }
    const size_t g = ((size_t)y * W + x) * 3;
    mosaic[g] = out[0];
    mosaic[g + 1] = out[1];
    mosaic[g + 2] = out[2];
}

static int check_table(const pano_patch *patches, int n, int H, int W, const char *who) {
    PANO_REQUIRE(patches, "%s: null patch table", who);
    PANO_REQUIRE(n >= 0 && n <= 32767, "%s: %d patches (int16 owner map holds 32767)", who, n);
    PANO_REQUIRE(H > 0 && W > 0, "%s: bad mosaic shape %dx%d", who, H, W);
    return PANO_OK;
}

#define MOSAIC_GRID dim3 block(64, 4), grid(ceil_div(W, 64), ceil_div(H, 4))

extern "C" int pano_ownership(const pano_patch *patches, int n, int H, int W,
                              int16_t *owner, uint8_t *valid, void *stream) {
    if (int rc = check_table(patches, n, H, W, "pano_ownership")) return rc;
    PANO_REQUIRE(owner && valid, "pano_ownership: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_OWNERSHIP, (hipStream_t)stream, hipLaunchKernelGGL(ownership_kernel, grid, block, 0, (hipStream_t)stream, patches, n,
                       H, W, owner, valid));
    PANO_LAUNCH_CHECK("ownership_kernel");
    return PANO_OK;
}

extern "C" int pano_multiband_compose(const pano_patch *patches, int n, int H, int W,
                                      int n_levels, const int16_t *owner,
                                      const uint8_t *valid, uint8_t *mosaic,
                                      float *mosaic_f32, void *stream) {
    if (int rc = check_table(patches, n, H, W, "pano_multiband_compose")) return rc;
    PANO_REQUIRE(owner && valid && mosaic, "pano_multiband_compose: null pointer");
    PANO_REQUIRE(n_levels >= 1 && n_levels <= PANO_MAX_LEVELS,
                 "pano_multiband_compose: n_levels %d outside [1, %d]", n_levels, PANO_MAX_LEVELS);
    MOSAIC_GRID;
    hipStream_t s = (hipStream_t)stream;
#define COMPOSE(L)                                                                   \
    case L:                                                                          \
        PANO_TIMED(PK_COMPOSE, s,                                                    \
                   hipLaunchKernelGGL(multiband_compose_kernel<L>, grid, block, 0,   \
                                      s, patches, n, H, W, owner, valid, mosaic,     \
                                      mosaic_f32));                                  \
        break;
    switch (n_levels) {
        COMPOSE(1) COMPOSE(2) COMPOSE(3) COMPOSE(4) COMPOSE(5) COMPOSE(6) COMPOSE(7) COMPOSE(8)
    }
#undef COMPOSE
    PANO_LAUNCH_CHECK("multiband_compose_kernel");
    return PANO_OK;
}

extern "C" int pano_linear_blend(const pano_patch *patches, int n, int H, int W,
                                 uint8_t *mosaic, void *stream) {
    if (int rc = check_table(patches, n, H, W, "pano_linear_blend")) return rc;
    PANO_REQUIRE(mosaic, "pano_linear_blend: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_LINEAR, (hipStream_t)stream, hipLaunchKernelGGL(linear_blend_kernel, grid, block, 0, (hipStream_t)stream, patches,
                       n, H, W, mosaic));
    PANO_LAUNCH_CHECK("linear_blend_kernel");
    return PANO_OK;
}

extern "C" int pano_no_blend(const pano_patch *patches, int n, int H, int W,
                             uint8_t *mosaic, void *stream) {
    if (int rc = check_table(patches, n, H, W, "pano_no_blend")) return rc;
    PANO_REQUIRE(mosaic, "pano_no_blend: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_NOBLEND, (hipStream_t)stream, hipLaunchKernelGGL(no_blend_kernel, grid, block, 0, (hipStream_t)stream, patches, n, H,
                       W, mosaic));
    PANO_LAUNCH_CHECK("no_blend_kernel");
    return PANO_OK;
}
