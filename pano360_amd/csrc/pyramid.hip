// Building blocks of the Gaussian / difference-of-Gaussian scale space that
// `features.sift_detector` obtains from OpenCV (features.py:192-201 ->
// cv2.xfeatures2d.SIFT_create().detectAndCompute): grey conversion, the 2x
// bilinear up-sampling of the base image, the nearest-neighbour halving between
// octaves and the DoG subtraction.  The Gaussian steps themselves are
// pano_blur_plane (csrc/blur.hip).
//
// None of this arithmetic is in the reference repo (it lives inside OpenCV,
// which the reference does not pin): the semantics restated here - and in
// oracle/sift_pyramid.py - are OpenCV 3.4/4.x's published SIFT
// (createInitialImage / buildGaussianPyramid / buildDoGPyramid), cvtColor
// BGR2GRAY for 8-bit (14-bit fixed point) and resize INTER_LINEAR /
// INTER_NEAREST.  PARITY UNPINNED.
//
// All four kernels are pure streaming (HBM bound): one read and one write per
// output pixel, 16-B lanes where the layout allows.
#include "common.h"

// (B*1868 + G*9617 + R*4899 + 2^13) >> 14, then to float (scale 1).  Four pixels per thread:
// their twelve bytes as three 4-byte loads, sixteen bytes out (three byte loads and a 4-byte
// store per pixel ran at 1.9 TB/s).
__device__ __forceinline__ float grey_of(uint32_t b, uint32_t g, uint32_t r) {
    return (float)(int)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14);
}
__global__ __launch_bounds__(256) void gray_u8_kernel(const uint8_t *__restrict__ bgr,
                                                      size_t npix, float *__restrict__ out) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x, i = 4 * q;
    if (i >= npix) return;
    if (i + 4 <= npix && ((uintptr_t)bgr & 3) == 0 && ((uintptr_t)out & 15) == 0) {
        const uint32_t *p = (const uint32_t *)(bgr + 3 * i);           // 12 q bytes: 4-byte aligned
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        float4 v;
        v.x = grey_of(w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u);
        v.y = grey_of(w0 >> 24, w1 & 255u, (w1 >> 8) & 255u);
        v.z = grey_of((w1 >> 16) & 255u, w1 >> 24, w2 & 255u);
        v.w = grey_of((w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24);
        *(float4 *)(out + i) = v;
        return;
    }
    for (size_t k = i; k < npix && k < i + 4; ++k) {
        const uint8_t *p = bgr + k * 3;
        out[k] = grey_of(p[0], p[1], p[2]);
    }
}

// resize(src, Size(2w, 2h), INTER_LINEAR): source coordinate (d + 0.5)/2 - 0.5,
// taps clamped at the border with weight 1 on the edge pixel; horizontal
// interpolation first, then vertical, in float.
__device__ __forceinline__ void up2_tap(int d, int n, int &i0, int &i1, float &w0, float &w1) {
    float f = ((float)d + 0.5f) * 0.5f - 0.5f;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) {
        s = 0;
        f = 0.0f;
    }
    if (s >= n - 1) {
        s = n - 1;
        f = 0.0f;
    }
    i0 = s;
    i1 = s + 1 < n ? s + 1 : n - 1;
    w0 = 1.0f - f;
    w1 = f;
}

__global__ __launch_bounds__(256) void resize_up2_kernel(const float *__restrict__ src, int h,
                                                         int w, float *__restrict__ dst) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= 2 * w || y >= 2 * h) return;
    int x0, x1, y0, y1;
    float a0, a1, b0, b1;
    up2_tap(x, w, x0, x1, a0, a1);
    up2_tap(y, h, y0, y1, b0, b1);
    const float *r0 = src + (size_t)y0 * w, *r1 = src + (size_t)y1 * w;
    const float top = r0[x0] * a0 + r0[x1] * a1;
    const float bot = r1[x0] * a0 + r1[x1] * a1;
    dst[(size_t)y * 2 * w + x] = top * b0 + bot * b1;
}

// resize(src, Size(w/2, h/2), INTER_NEAREST): src(min(floor(d * w / (w/2)), w-1))
__global__ __launch_bounds__(256) void decimate2_kernel(const float *__restrict__ src, int h,
                                                        int w, float *__restrict__ dst) {
    const int ow = w / 2, oh = h / 2;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ow || y >= oh) return;
    int sx = (int)floor((double)x * ((double)w / (double)ow));
    int sy = (int)floor((double)y * ((double)h / (double)oh));
    sx = sx < w - 1 ? sx : w - 1;
    sy = sy < h - 1 ? sy : h - 1;
    dst[(size_t)y * ow + x] = src[(size_t)sy * w + sx];
}

__global__ __launch_bounds__(256) void subtract_kernel(const float *__restrict__ a,
                                                       const float *__restrict__ b, size_t n,
                                                       float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

extern "C" int pano_gray_u8(pano_ctx *ctx, const uint8_t *bgr, int h, int w, float *out) {
    PANO_ENTER(ctx, "pano_gray_u8");
    PANO_REQUIRE(bgr && out && h > 0 && w > 0, "pano_gray_u8: bad argument");
    const size_t n = (size_t)h * w;
    hipLaunchKernelGGL(gray_u8_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0,
                       (hipStream_t)stream, bgr, n, out);
    PANO_LAUNCH_CHECK("gray_u8_kernel");
    return PANO_OK;
}

extern "C" int pano_resize_up2(pano_ctx *ctx, const float *src, int h, int w, float *dst) {
    PANO_ENTER(ctx, "pano_resize_up2");
    PANO_REQUIRE(src && dst && h > 0 && w > 0, "pano_resize_up2: bad argument");
    dim3 block(64, 4), grid(ceil_div(2 * w, 64), ceil_div(2 * h, 4));
    hipLaunchKernelGGL(resize_up2_kernel, grid, block, 0, (hipStream_t)stream, src, h, w, dst);
    PANO_LAUNCH_CHECK("resize_up2_kernel");
    return PANO_OK;
}

extern "C" int pano_decimate2(pano_ctx *ctx, const float *src, int h, int w, float *dst) {
    PANO_ENTER(ctx, "pano_decimate2");
    PANO_REQUIRE(src && dst && h > 1 && w > 1, "pano_decimate2: bad argument");
    dim3 block(64, 4), grid(ceil_div(w / 2, 64), ceil_div(h / 2, 4));
    hipLaunchKernelGGL(decimate2_kernel, grid, block, 0, (hipStream_t)stream, src, h, w, dst);
    PANO_LAUNCH_CHECK("decimate2_kernel");
    return PANO_OK;
}

extern "C" int pano_subtract(pano_ctx *ctx, const float *a, const float *b, size_t n, float *out) {
    PANO_ENTER(ctx, "pano_subtract");
    PANO_REQUIRE(a && b && out, "pano_subtract: null pointer");
    if (n == 0) return PANO_OK;
    hipLaunchKernelGGL(subtract_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, a, b, n, out);
    PANO_LAUNCH_CHECK("subtract_kernel");
    return PANO_OK;
}
