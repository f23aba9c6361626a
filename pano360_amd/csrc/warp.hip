// Spherical warp of one patch (or of a window of it): inverse map, bounds
// mask, fixed-point bilinear REFLECT resampling straight from the uint8 frame.
//
// Reference arithmetic replaced: stitcher.py:300-317 (inverse map, mask,
// cv2.remap, alpha *= ~mask) and stitcher.py:251-263 (_hat/_add_weights: the
// float32 RGBA source image is never materialised here - a tap's colour is
// lut255[u8] and its alpha is float32(hat_y*hat_x), which is exactly what
// _add_weights stores).
//
// Roofline: HBM.  Algorithmic bytes per warped pixel: 12 (fused path, three
// float planes) or 17 (stage path: four planes + mask) written, + 3 bytes per
// SOURCE pixel under the window read once (neighbouring pixels' taps share
// cache lines, so frame reads are served by L1/L2).
#include "geom.h"

struct ProjK {
    double p[9];
};

// FULL: whole-patch stage output (RGBA planes, mask, optional maps).
// !FULL: colour planes of a window only.
template <bool FULL>
__global__ __launch_bounds__(256) void warp_spherical_kernel(
    const uint8_t *__restrict__ frame, int sh, int sw, ProjK K,
    const double *__restrict__ sin_t, const double *__restrict__ cos_t,
    const double *__restrict__ tan_p, const float *__restrict__ lut255,
    const double *__restrict__ hat_x, const double *__restrict__ hat_y,
    int gx0, int gy0, int pw, int ph, int pitch, float *__restrict__ planes,
    uint8_t *__restrict__ mask, float *__restrict__ map_x,
    float *__restrict__ map_y) {
    __shared__ float s_lut[256];
    s_lut[threadIdx.y * 64 + threadIdx.x] = lut255[threadIdx.y * 64 + threadIdx.x];
    __syncthreads();

    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= pw || y >= ph) return;

    float px, py;
    const bool m = map_pixel(K.p, sin_t[gx0 + x], cos_t[gx0 + x], tan_p[gy0 + y], sw, sh,
                             px, py);
    const Taps tp = make_taps(px, py, sw, sh);
    const uint8_t *r0 = frame + (size_t)tp.y0 * sw * 3;
    const uint8_t *r1 = frame + (size_t)tp.y1 * sw * 3;
    const uint8_t *p00 = r0 + tp.x0 * 3, *p01 = r0 + tp.x1 * 3;
    const uint8_t *p10 = r1 + tp.x0 * 3, *p11 = r1 + tp.x1 * 3;
    const size_t plane = (size_t)ph * pitch;
    const size_t o = (size_t)y * pitch + x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        planes[k * plane + o] =
            lerp4(s_lut[p00[k]], s_lut[p01[k]], s_lut[p10[k]], s_lut[p11[k]], tp);
    if (FULL) {
        float a = alpha_at(hat_x, hat_y, tp);
        a = a * (m ? 0.0f : 1.0f);                                  // :317
        planes[3 * plane + o] = a;
        mask[(size_t)y * pw + x] = m ? 1 : 0;
        if (map_x) {
            map_x[(size_t)y * pw + x] = px;
            map_y[(size_t)y * pw + x] = py;
        }
    }
}

// Colour planes of window V of every patch in one launch (blockIdx.z = patch).
// WARP_ROWS rows per thread (rows y and y + 4 of a 64 x 8 block): the kernel is bound by the
// latency of its dependent loads (trig tables -> taps -> LUT), not by their number, and at
// 22 registers the CU already holds all the waves it can; two independent pixels per thread
// put twice as many loads in flight.  The two share their column's sin / cos.
#ifndef WARP_ROWS
#define WARP_ROWS 4
#endif
__global__ __launch_bounds__(256) void warp_windows_kernel(
    const pano_camera *__restrict__ cams, const pano_patch *__restrict__ patches,
    const double *__restrict__ sin_t, const double *__restrict__ cos_t,
    const double *__restrict__ tan_p, const float *__restrict__ lut, int lut_stride,
    const uint8_t *__restrict__ need) {
    __shared__ float s_lut[256];
    const pano_patch p = patches[blockIdx.z];
    constexpr int BH = 4 * WARP_ROWS;
    if ((int)blockIdx.x * 64 >= p.vw || (int)blockIdx.y * BH >= p.vh) return;   // uniform
    if (need) {
        // the 32 x 32 tiles (grid of rectangle A, clamped) under this block's rows and 64
        // columns: nothing will read the block if none of them is needed
        const int ntx = ((p.ax0 + p.aw - 1) >> 5) - (p.ax0 >> 5) + 1;
        const int nty = ((p.ay0 + p.ah - 1) >> 5) - (p.ay0 >> 5) + 1;
        const int px0 = p.vx0 + (int)blockIdx.x * 64, py0 = p.vy0 + (int)blockIdx.y * BH;
        const int ty0 = min(max((py0 >> 5) - (p.ay0 >> 5), 0), nty - 1);
        const int ty1 = min(max(((py0 + BH - 1) >> 5) - (p.ay0 >> 5), 0), nty - 1);
        const int tx0 = min(max((px0 >> 5) - (p.ax0 >> 5), 0), ntx - 1);
        const int tx1 = min(max(((px0 + 63) >> 5) - (p.ax0 >> 5), 0), ntx - 1);
        bool any = false;
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) any |= need[p.tiles_off + ty * ntx + tx] != 0;
        if (!any) return;                                                       // uniform
    }
    s_lut[threadIdx.y * 64 + threadIdx.x] =
        lut[(size_t)p.index * lut_stride + threadIdx.y * 64 + threadIdx.x];
    __syncthreads();

    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y0 = blockIdx.y * BH + threadIdx.y;
    if (x >= p.vw || y0 >= p.vh) return;
    const pano_camera *cam = cams + p.index;
    const int sw = cam->sw, sh = cam->sh;
    const int gx = p.x0 + p.vx0 + x;
    const double s = table_f64(sin_t, gx), c = table_f64(cos_t, gx);
    double t[WARP_ROWS];
#pragma unroll
    for (int j = 0; j < WARP_ROWS; ++j)
        t[j] = table_f64(tan_p, p.y0 + p.vy0 + min(y0 + 4 * j, p.vh - 1));
    Taps tp[WARP_ROWS];
    TapBytes tb[WARP_ROWS];
    bool inner = true;
#pragma unroll
    for (int j = 0; j < WARP_ROWS; ++j) {
        float px, py;
        map_pixel(cam->proj, s, c, t[j], sw, sh, px, py);
        tp[j] = tap_base(px, py);
        inner &= taps_interior(tp[j], sw, sh);
    }
    // Nearly every wave lies inside its frame with all its pixels' taps: then there is no
    // border to reflect at and the taps of a pixel are neighbours (the kernel is bound by its
    // vector instructions - 160 per pixel, 77 % of the issue cycles - not by its loads; the
    // four range tests of the general path and the branches around their modulos were 30 of
    // them).  The decision is per wave, so neither path runs under a partial mask.
    if (__ballot(!inner) == 0ull) {
#pragma unroll
        for (int j = 0; j < WARP_ROWS; ++j) tb[j] = load_taps_interior(cam->frame, sw, tp[j]);
    } else {
#pragma unroll
        for (int j = 0; j < WARP_ROWS; ++j) {
            reflect_taps(tp[j], sw, sh);
            tb[j] = load_taps(cam->frame, sw, tp[j]);
        }
    }
    // offsets inside a plane fit 32 bits (pano_layout_windows checks it); the three bases are
    // wave-uniform
    // (a plane is below 2 GiB - layout_record_fits - so a pixel's BYTE offset fits 32 bits: the
    // stores take the wave-uniform plane base as scalar operand and one 32-bit offset register
    // for all three planes, instead of a 64-bit address computed per store)
    typedef __attribute__((address_space(1))) float *plane_ptr;
    typedef __attribute__((address_space(1))) char *byte_ptr;
    const size_t plane = (size_t)p.vh * p.vpitch;
    const byte_ptr out[3] = {(byte_ptr)p.planes, (byte_ptr)(p.planes + plane),
                             (byte_ptr)(p.planes + 2 * plane)};
#pragma unroll
    for (int j = 0; j < WARP_ROWS; ++j) {
        const int y = y0 + 4 * j;
        if (y >= p.vh) break;
        const uint32_t o = ((uint32_t)y * (uint32_t)p.vpitch + (uint32_t)x) * 4u;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            *(plane_ptr)(out[k] + o) = lerp4(lut_at(s_lut, tb[j].v[0][k]), lut_at(s_lut, tb[j].v[1][k]),
                                             lut_at(s_lut, tb[j].v[2][k]), lut_at(s_lut, tb[j].v[3][k]), tp[j]);
        }
    }
}

__global__ __launch_bounds__(256) void add_weights_kernel(
    const uint8_t *__restrict__ frame, int h, int w,
    const float *__restrict__ lut255, const double *__restrict__ hat_x,
    const double *__restrict__ hat_y, float4 *__restrict__ rgba) {
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    if (x >= w || y >= h) return;
    const uint8_t *s = frame + ((size_t)y * w + x) * 3;
    float4 v;
    v.x = lut255[s[0]];
    v.y = lut255[s[1]];
    v.z = lut255[s[2]];
    v.w = (float)(hat_y[y] * hat_x[x]);
    rgba[(size_t)y * w + x] = v;
}

extern "C" int pano_add_weights(pano_ctx *ctx, const uint8_t *frame, int h, int w,
                                const float *lut255, const double *hat_x, const double *hat_y,
                                float *rgba) {
    PANO_ENTER(ctx, "pano_add_weights");
    PANO_REQUIRE(frame && lut255 && hat_x && hat_y && rgba, "pano_add_weights: null pointer");
    PANO_REQUIRE(h > 0 && w > 0, "pano_add_weights: bad shape %dx%d", h, w);
    dim3 block(64, 4), grid(ceil_div(w, 64), ceil_div(h, 4));
    PANO_TIMED(PK_ADD_WEIGHTS, (hipStream_t)stream,
               hipLaunchKernelGGL(add_weights_kernel, grid, block, 0, (hipStream_t)stream,
                                  frame, h, w, lut255, hat_x, hat_y, (float4 *)rgba));
    PANO_LAUNCH_CHECK("add_weights_kernel");
    return PANO_OK;
}

static int check_frame(const char *who, int sh, int sw, int pw, int ph, int gx0, int gy0) {
    // cv2.remap itself asserts source dimensions < 32767 (int16 coordinates)
    PANO_REQUIRE(sh > 0 && sw > 0 && sh < 32767 && sw < 32767,
                 "%s: frame %dx%d outside (0, 32767)", who, sh, sw);
    PANO_REQUIRE(pw > 0 && ph > 0 && gx0 >= 0 && gy0 >= 0, "%s: bad rectangle", who);
    return PANO_OK;
}

extern "C" int pano_warp_spherical(pano_ctx *ctx, const uint8_t *frame, int sh, int sw,
                                   const double *proj, const double *sin_t, const double *cos_t,
                                   const double *tan_p, const float *lut255, const double *hat_x,
                                   const double *hat_y, int gx0, int gy0, int pw, int ph,
                                   float *planes, uint8_t *mask, float *map_x, float *map_y) {
    PANO_ENTER(ctx, "pano_warp_spherical");
    PANO_REQUIRE(frame && proj && sin_t && cos_t && tan_p && lut255 && hat_x &&
                     hat_y && planes && mask,
                 "pano_warp_spherical: null pointer");
    PANO_REQUIRE((map_x == nullptr) == (map_y == nullptr),
                 "pano_warp_spherical: map_x and map_y must both be given or both be NULL");
    if (int rc = check_frame("pano_warp_spherical", sh, sw, pw, ph, gx0, gy0)) return rc;
    ProjK K;
    for (int i = 0; i < 9; ++i) K.p[i] = proj[i];
    dim3 block(64, 4), grid(ceil_div(pw, 64), ceil_div(ph, 4));
    PANO_TIMED(PK_WARP, (hipStream_t)stream,
               hipLaunchKernelGGL(warp_spherical_kernel<true>, grid, block, 0,
                                  (hipStream_t)stream, frame, sh, sw, K, sin_t, cos_t, tan_p,
                                  lut255, hat_x, hat_y, gx0, gy0, pw, ph, pano_pitch_of(pw),
                                  planes, mask, map_x, map_y));
    PANO_LAUNCH_CHECK("warp_spherical_kernel");
    return PANO_OK;
}

extern "C" int pano_warp_windows(pano_ctx *ctx, const pano_camera *cams, const pano_patch *patches,
                                 int n, int max_vw, int max_vh, const double *sin_t,
                                 const double *cos_t, const double *tan_p, const float *lut,
                                 int lut_stride, const uint8_t *need) {
    PANO_ENTER(ctx, "pano_warp_windows");
    PANO_REQUIRE(cams && patches && sin_t && cos_t && tan_p && lut && lut_stride >= 0,
                 "pano_warp_windows: null pointer");
    PANO_REQUIRE(n >= 0 && n <= 65535 && max_vw >= 0 && max_vh >= 0,
                 "pano_warp_windows: bad argument");
    if (n == 0 || max_vw == 0 || max_vh == 0) return PANO_OK;
    dim3 block(64, 4), grid(ceil_div(max_vw, 64), ceil_div(max_vh, 4 * WARP_ROWS), n);
    PANO_TIMED(PK_WARP_WINDOWS, (hipStream_t)stream,
               hipLaunchKernelGGL(warp_windows_kernel, grid, block, 0, (hipStream_t)stream,
                                  cams, patches, sin_t, cos_t, tan_p, lut, lut_stride, need));
    PANO_LAUNCH_CHECK("warp_windows_kernel");
    return PANO_OK;
}
