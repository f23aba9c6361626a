// Spherical warp of one patch: inverse map, bounds mask, fixed-point bilinear
// REFLECT resampling straight from the uint8 frame.
//
// Reference arithmetic replaced: stitcher.py:300-317 (inverse map, mask,
// cv2.remap, alpha *= ~mask) and stitcher.py:251-263 (_hat/_add_weights: the
// float32 RGBA source image is never materialised here - a tap's colour is
// lut255[u8] and its alpha is float32(hat_y*hat_x), which is exactly what
// _add_weights stores).
//
// Roofline: HBM.  Algorithmic bytes per patch pixel: 17 written (4 float
// planes + mask) + 3 bytes of frame per SOURCE pixel read once (taps of
// neighbouring pixels share cache lines, so frame reads are served by L1/L2).
#include "common.h"

struct ProjK {
    double p[9];
};

struct Taps {
    int x0, x1, y0, y1;
    float w00, w01, w10, w11;
};

// cv2.remap's coordinate handling (INTER_BITS = 5), see include/pano360.h.
__device__ __forceinline__ Taps make_taps(float px, float py, int sw, int sh) {
    int sx = cv_round(px * 32.0f), sy = cv_round(py * 32.0f);
    int fx = sx & 31, fy = sy & 31;
    int ix = sat16(sx >> 5), iy = sat16(sy >> 5);
    Taps t;
    t.x0 = reflect_edge(ix, sw);
    t.x1 = reflect_edge(ix + 1, sw);
    t.y0 = reflect_edge(iy, sh);
    t.y1 = reflect_edge(iy + 1, sh);
    float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
    t.w00 = (1.0f - ay) * (1.0f - ax);
    t.w01 = (1.0f - ay) * ax;
    t.w10 = ay * (1.0f - ax);
    t.w11 = ay * ax;
    return t;
}

// v00*w00 + v01*w01 + v10*w10 + v11*w11, left to right, one rounding per
// operation (the file is built with -ffp-contract=off).
__device__ __forceinline__ float lerp4(float v00, float v01, float v10,
                                       float v11, const Taps &t) {
    float a = v00 * t.w00;
    a = a + v01 * t.w01;
    a = a + v10 * t.w10;
    a = a + v11 * t.w11;
    return a;
}

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

    // ray = (sin theta, tan phi, cos theta); pixel = K R ray, evaluated in
    // double as an FMA chain over k, then rounded to float32 (:303-306).
    const double s = sin_t[gx0 + x], c = cos_t[gx0 + x], t = tan_p[gy0 + y];
    const double vx = fma(K.p[2], c, fma(K.p[1], t, K.p[0] * s));
    const double vy = fma(K.p[5], c, fma(K.p[4], t, K.p[3] * s));
    const double vz = fma(K.p[8], c, fma(K.p[7], t, K.p[6] * s));
    const float fx = (float)vx, fy = (float)vy, fz = (float)vz;
    const float cx = (float)((double)sw / 2.0), cy = (float)((double)sh / 2.0);
    const float px = __fdiv_rn(fx, fz) + cx;
    const float py = __fdiv_rn(fy, fz) + cy;
    bool m = fz < 0.0f;
    m |= (px < 0.0f) | (px > (float)(sw - 1)) | (py < 0.0f) | (py > (float)(sh - 1));

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

    // alpha plane of _add_weights: float32(hat(y) * hat(x)), product in double
    const double hy0 = hat_y[tp.y0], hy1 = hat_y[tp.y1];
    const double hx0 = hat_x[tp.x0], hx1 = hat_x[tp.x1];
    float a = lerp4((float)(hy0 * hx0), (float)(hy0 * hx1), (float)(hy1 * hx0),
                    (float)(hy1 * hx1), tp);
    a = a * (m ? 0.0f : 1.0f);                                  // :317
    planes[3 * plane + o] = a;
    mask[(size_t)y * pw + x] = m ? 1 : 0;
    if (map_x) {
        map_x[(size_t)y * pw + x] = px;
        map_y[(size_t)y * pw + x] = py;
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

extern "C" int pano_add_weights(const uint8_t *frame, int h, int w,
                                const float *lut255, const double *hat_x,
                                const double *hat_y, float *rgba, void *stream) {
    PANO_REQUIRE(frame && lut255 && hat_x && hat_y && rgba, "pano_add_weights: null pointer");
    PANO_REQUIRE(h > 0 && w > 0, "pano_add_weights: bad shape %dx%d", h, w);
    dim3 block(64, 4), grid(ceil_div(w, 64), ceil_div(h, 4));
    PANO_TIMED(PK_ADD_WEIGHTS, (hipStream_t)stream, hipLaunchKernelGGL(add_weights_kernel, grid, block, 0, (hipStream_t)stream,
                       frame, h, w, lut255, hat_x, hat_y, (float4 *)rgba));
    PANO_LAUNCH_CHECK("add_weights_kernel");
    return PANO_OK;
}

extern "C" int pano_warp_spherical(const uint8_t *frame, int sh, int sw,
                                   const double *proj, const double *sin_t,
                                   const double *cos_t, const double *tan_p,
                                   const float *lut255, const double *hat_x,
                                   const double *hat_y, int gx0, int gy0,
                                   int pw, int ph, float *planes, uint8_t *mask,
                                   float *map_x, float *map_y, void *stream) {
    PANO_REQUIRE(frame && proj && sin_t && cos_t && tan_p && lut255 && hat_x &&
                     hat_y && planes && mask,
                 "pano_warp_spherical: null pointer");
    PANO_REQUIRE((map_x == nullptr) == (map_y == nullptr),
                 "pano_warp_spherical: map_x and map_y must both be given or both be NULL");
    // cv2.remap itself asserts source dimensions < 32767 (int16 coordinates)
    PANO_REQUIRE(sh > 0 && sw > 0 && sh < 32767 && sw < 32767,
                 "pano_warp_spherical: frame %dx%d outside (0, 32767)", sh, sw);
    PANO_REQUIRE(pw > 0 && ph > 0 && gx0 >= 0 && gy0 >= 0,
                 "pano_warp_spherical: bad patch rectangle");
    ProjK K;
    for (int i = 0; i < 9; ++i) K.p[i] = proj[i];
    dim3 block(64, 4), grid(ceil_div(pw, 64), ceil_div(ph, 4));
    PANO_TIMED(PK_WARP, (hipStream_t)stream, hipLaunchKernelGGL(warp_spherical_kernel, grid, block, 0, (hipStream_t)stream,
                       frame, sh, sw, K, sin_t, cos_t, tan_p, lut255, hat_x, hat_y,
                       gx0, gy0, pw, ph, pano_pitch_of(pw), planes, mask, map_x, map_y));
    PANO_LAUNCH_CHECK("warp_spherical_kernel");
    return PANO_OK;
}
