// Decimated Laplacian-pyramid blending of two images and the 8-bit shrink of the
// CLI's inputs.
//
// Reference arithmetic replaced: blend.laplacian_blending (blend.py:105-140: cv2.pyrDown
// / cv2.pyrUp pyramids of two float32 images and a float64 mask, per-level
// la*gm + lb*(1-gm) in float64, collapse, clip, uint8) and the cv2.resize of
// stitcher.py:419-420.  pyrDown / pyrUp / resize live inside OpenCV, which the reference
// does not pin: their semantics are restated from OpenCV's published algorithm
// (the oracle says which) - PARITY UNPINNED at that boundary; everything the
// reference computes in NumPy itself is reproduced bit for bit (tests/golden/laplacian.npz).
//
// Images are interleaved [h][w][c] exactly as the reference holds them, c <= 4.  Every
// kernel is streaming (HBM bound): one thread per output element, neighbouring threads
// on neighbouring addresses.
#include "common.h"

// 0 plain pyrUp, 1 other - pyrUp (a Laplacian level, blend.py:126), 2 other + pyrUp
// (collapse, blend.py:138)
enum { UP_PLAIN = 0, UP_SUB = 1, UP_ADD = 2 };

template <typename T>
__global__ __launch_bounds__(256) void lap_pyr_down_kernel(const T *__restrict__ src, int h, int w,
                                                           int c, T *__restrict__ dst, int oh,
                                                           int ow) {
    const int xc = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (xc >= ow * c || y >= oh) return;
    const int x = xc / c, ch = xc - x * c;
    int cx[5], cy[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        cx[k] = reflect_101(2 * x - 2 + k, w) * c + ch;
        cy[k] = reflect_101(2 * y - 2 + k, h);
    }
    T rowv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const T *s = src + (size_t)cy[k] * w * c;
        rowv[k] = s[cx[2]] * (T)6 + (s[cx[1]] + s[cx[3]]) * (T)4 + s[cx[0]] + s[cx[4]];
    }
    dst[(size_t)y * ow * c + xc] =
        (rowv[2] * (T)6 + (rowv[1] + rowv[3]) * (T)4 + rowv[0] + rowv[4]) * (T)(1.0 / 256.0);
}

// Horizontal pass of pyrUp at output column dx of source row s (n samples, stride c):
// even dx: s[i-1] + s[i]*6 + s[i+1], odd: (s[i] + s[i+1])*4, with s[-1] := s[1] and
// s[n] := s[n-1] written the way OpenCV's pyrUp_ writes its two edge columns.
template <typename T>
__device__ __forceinline__ T up_row(const T *__restrict__ s, int n, int c, int dx) {
    const int i = dx >> 1;
    if (dx & 1) {
        if (i == n - 1) return s[(size_t)i * c] * (T)8;
        return (s[(size_t)i * c] + s[(size_t)(i + 1) * c]) * (T)4;
    }
    if (i == 0) return s[0] * (T)6 + s[c] * (T)2;
    if (i == n - 1) return s[(size_t)(i - 1) * c] + s[(size_t)i * c] * (T)7;
    return s[(size_t)(i - 1) * c] + s[(size_t)i * c] * (T)6 + s[(size_t)(i + 1) * c];
}

// pyrUp(src)[:oh, :ow] combined with `other` ([oh][ow][c]) according to MODE.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void lap_pyr_up_kernel(const T *__restrict__ src, int sh, int sw,
                                                         int c, const T *__restrict__ other,
                                                         T *__restrict__ dst, int oh, int ow) {
    const int xc = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (xc >= ow * c || y >= oh) return;
    const int x = xc / c, ch = xc - x * c;
    const int sy = y >> 1;
    // source rows sy-1, sy, sy+1 with borderInterpolate(2 r, 2 sh, REFLECT_101) / 2
    const int r0 = reflect_101(2 * (sy - 1), 2 * sh) >> 1;
    const int r2 = reflect_101(2 * (sy + 1), 2 * sh) >> 1;
    const T *base = src + ch;
    const T v1 = up_row(base + (size_t)sy * sw * c, sw, c, x);
    const T v2 = up_row(base + (size_t)r2 * sw * c, sw, c, x);
    T up;
    if (y & 1) {
        up = ((v1 + v2) * (T)4) * (T)(1.0 / 64.0);
    } else {
        const T v0 = up_row(base + (size_t)r0 * sw * c, sw, c, x);
        up = (v0 + v1 * (T)6 + v2) * (T)(1.0 / 64.0);
    }
    const size_t o = (size_t)y * ow * c + xc;
    if (MODE == UP_SUB)
        dst[o] = other[o] - up;
    else if (MODE == UP_ADD)
        dst[o] = other[o] + up;
    else
        dst[o] = up;
}

__global__ __launch_bounds__(256) void lap_from_u8_kernel(const uint8_t *__restrict__ src, size_t n,
                                                          float *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];                              // .astype("float32")
}

// la * gm + lb * (1.0 - gm) (blend.py:136), one rounding per operation, in the mask's
// type: NumPy promotes the float32 Laplacian levels to float64 against a float64 mask (the
// default sigmoid), and keeps everything float32 against a float32 mask.
template <typename T>
__global__ __launch_bounds__(256) void lap_mix_kernel(const float *__restrict__ la,
                                                      const float *__restrict__ lb,
                                                      const T *__restrict__ gm, size_t n,
                                                      T *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const T g = gm[i];
    const T a = (T)la[i] * g;
    const T b = (T)lb[i] * ((T)1.0 - g);
    out[i] = a + b;
}

// np.clip(blended, 0, 255).astype("uint8") (blend.py:140)
template <typename T>
__global__ __launch_bounds__(256) void lap_finish_kernel(const T *__restrict__ src, size_t n,
                                                         uint8_t *__restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T v = src[i];
    v = v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v);
    dst[i] = (uint8_t)(int)v;
}

// ---- 8-bit resize -------------------------------------------------------------------
// Linear: taps and 11-bit coefficients per output column / row come from the host
// (computed with NumPy exactly as the oracle does); the kernel is the integer arithmetic.
__global__ __launch_bounds__(256) void resize_u8_linear_kernel(
    const uint8_t *__restrict__ src, int sw, int c, const int32_t *__restrict__ xtab,
    const int32_t *__restrict__ ytab, uint8_t *__restrict__ dst, int oh, int ow) {
    const int xc = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (xc >= ow * c || y >= oh) return;
    const int x = xc / c, ch = xc - x * c;
    const int x0 = xtab[4 * x], x1 = xtab[4 * x + 1], a0 = xtab[4 * x + 2], a1 = xtab[4 * x + 3];
    const int y0 = ytab[4 * y], y1 = ytab[4 * y + 1], b0 = ytab[4 * y + 2], b1 = ytab[4 * y + 3];
    const uint8_t *r0 = src + (size_t)y0 * sw * c + ch, *r1 = src + (size_t)y1 * sw * c + ch;
    const int top = r0[(size_t)x0 * c] * a0 + r0[(size_t)x1 * c] * a1;
    const int bot = r1[(size_t)x0 * c] * a0 + r1[(size_t)x1 * c] * a1;
    int v = (((b0 * (top >> 4)) >> 16) + ((b1 * (bot >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(size_t)y * ow * c + xc] = (uint8_t)v;
}

// Exact 2:1 reduction: rounded 2 x 2 box means (the area path cv2.resize takes then).
__global__ __launch_bounds__(256) void resize_u8_half_kernel(const uint8_t *__restrict__ src,
                                                             int sw, int c,
                                                             uint8_t *__restrict__ dst, int oh,
                                                             int ow) {
    const int xc = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (xc >= ow * c || y >= oh) return;
    const int x = xc / c, ch = xc - x * c;
    const uint8_t *r0 = src + ((size_t)(2 * y) * sw + 2 * x) * c + ch;
    const uint8_t *r1 = r0 + (size_t)sw * c;
    dst[(size_t)y * ow * c + xc] = (uint8_t)((r0[0] + r0[c] + r1[0] + r1[c] + 2) >> 2);
}

// ---- C ABI -----------------------------------------------------------------------------
static inline dim3 grid_for(int oh, int owc) { return dim3(ceil_div(owc, 64), ceil_div(oh, 4)); }

template <typename T>
static int pyr_down_any(const T *src, int h, int w, int c, T *dst, hipStream_t stream) {
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    hipLaunchKernelGGL(lap_pyr_down_kernel<T>, grid_for(oh, ow * c), dim3(64, 4), 0, stream, src, h,
                       w, c, dst, oh, ow);
    PANO_LAUNCH_CHECK("lap_pyr_down_kernel");
    return PANO_OK;
}

template <typename T>
static int pyr_up_any(const T *src, int sh, int sw, int c, const T *other, int mode, T *dst,
                      int oh, int ow, hipStream_t stream) {
    const dim3 grid = grid_for(oh, ow * c), block(64, 4);
    if (mode == UP_SUB)
        hipLaunchKernelGGL((lap_pyr_up_kernel<T, UP_SUB>), grid, block, 0, stream, src, sh, sw, c,
                           other, dst, oh, ow);
    else if (mode == UP_ADD)
        hipLaunchKernelGGL((lap_pyr_up_kernel<T, UP_ADD>), grid, block, 0, stream, src, sh, sw, c,
                           other, dst, oh, ow);
    else
        hipLaunchKernelGGL((lap_pyr_up_kernel<T, UP_PLAIN>), grid, block, 0, stream, src, sh, sw,
                           c, other, dst, oh, ow);
    PANO_LAUNCH_CHECK("lap_pyr_up_kernel");
    return PANO_OK;
}

extern "C" int pano_pyr_down_image(pano_ctx *ctx, const void *src, int h, int w, int c, int is_f64,
                                   void *dst) {
    PANO_ENTER(ctx, "pano_pyr_down_image");
    PANO_REQUIRE(src && dst && h > 0 && w > 0 && c >= 1 && c <= 4,
                 "pano_pyr_down_image: bad argument");
    return is_f64 ? pyr_down_any((const double *)src, h, w, c, (double *)dst, (hipStream_t)stream)
                  : pyr_down_any((const float *)src, h, w, c, (float *)dst, (hipStream_t)stream);
}

extern "C" int pano_pyr_up_image(pano_ctx *ctx, const void *src, int sh, int sw, int c, int is_f64,
                                 const void *other, int mode, void *dst, int oh, int ow) {
    PANO_ENTER(ctx, "pano_pyr_up_image");
    PANO_REQUIRE(src && dst && c >= 1 && c <= 4, "pano_pyr_up_image: bad argument");
    PANO_REQUIRE(sh >= 2 && sw >= 2, "pano_pyr_up_image: source %dx%d is narrower than 2", sh, sw);
    PANO_REQUIRE(oh > 0 && ow > 0 && oh <= 2 * sh && ow <= 2 * sw,
                 "pano_pyr_up_image: output %dx%d exceeds twice the source", oh, ow);
    PANO_REQUIRE(mode == UP_PLAIN || (other && (mode == UP_SUB || mode == UP_ADD)),
                 "pano_pyr_up_image: bad mode");
    return is_f64 ? pyr_up_any((const double *)src, sh, sw, c, (const double *)other, mode,
                               (double *)dst, oh, ow, (hipStream_t)stream)
                  : pyr_up_any((const float *)src, sh, sw, c, (const float *)other, mode,
                               (float *)dst, oh, ow, (hipStream_t)stream);
}

extern "C" int pano_u8_to_f32(pano_ctx *ctx, const uint8_t *src, size_t n, float *dst) {
    PANO_ENTER(ctx, "pano_u8_to_f32");
    PANO_REQUIRE(src && dst, "pano_u8_to_f32: null pointer");
    if (n == 0) return PANO_OK;
    hipLaunchKernelGGL(lap_from_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, src, n, dst);
    PANO_LAUNCH_CHECK("lap_from_u8_kernel");
    return PANO_OK;
}

extern "C" int pano_laplacian_mix(pano_ctx *ctx, const float *la, const float *lb, const void *gm,
                                  size_t n, int is_f64, void *out) {
    PANO_ENTER(ctx, "pano_laplacian_mix");
    PANO_REQUIRE(la && lb && gm && out, "pano_laplacian_mix: null pointer");
    if (n == 0) return PANO_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (is_f64)
        hipLaunchKernelGGL(lap_mix_kernel<double>, grid, block, 0, (hipStream_t)stream, la, lb,
                           (const double *)gm, n, (double *)out);
    else
        hipLaunchKernelGGL(lap_mix_kernel<float>, grid, block, 0, (hipStream_t)stream, la, lb,
                           (const float *)gm, n, (float *)out);
    PANO_LAUNCH_CHECK("lap_mix_kernel");
    return PANO_OK;
}

extern "C" int pano_clip_u8(pano_ctx *ctx, const void *src, size_t n, int is_f64, uint8_t *dst) {
    PANO_ENTER(ctx, "pano_clip_u8");
    PANO_REQUIRE(src && dst, "pano_clip_u8: null pointer");
    if (n == 0) return PANO_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (is_f64)
        hipLaunchKernelGGL(lap_finish_kernel<double>, grid, block, 0, (hipStream_t)stream,
                           (const double *)src, n, dst);
    else
        hipLaunchKernelGGL(lap_finish_kernel<float>, grid, block, 0, (hipStream_t)stream,
                           (const float *)src, n, dst);
    PANO_LAUNCH_CHECK("lap_finish_kernel");
    return PANO_OK;
}

extern "C" int pano_resize_u8(pano_ctx *ctx, const uint8_t *src, int sh, int sw, int c,
                              const int32_t *xtab, const int32_t *ytab, uint8_t *dst, int oh,
                              int ow) {
    PANO_ENTER(ctx, "pano_resize_u8");
    PANO_REQUIRE(src && dst && sh > 0 && sw > 0 && oh > 0 && ow > 0 && c >= 1 && c <= 4,
                 "pano_resize_u8: bad argument");
    PANO_REQUIRE((xtab == nullptr) == (ytab == nullptr),
                 "pano_resize_u8: xtab and ytab must both be given or both be NULL");
    const dim3 grid = grid_for(oh, ow * c), block(64, 4);
    if (!xtab) {
        PANO_REQUIRE(sw == 2 * ow && sh == 2 * oh,
                     "pano_resize_u8: the box path needs an exact 2:1 reduction");
        hipLaunchKernelGGL(resize_u8_half_kernel, grid, block, 0, (hipStream_t)stream, src, sw, c,
                           dst, oh, ow);
        PANO_LAUNCH_CHECK("resize_u8_half_kernel");
        return PANO_OK;
    }
    hipLaunchKernelGGL(resize_u8_linear_kernel, grid, block, 0, (hipStream_t)stream, src, sw, c,
                       xtab, ytab, dst, oh, ow);
    PANO_LAUNCH_CHECK("resize_u8_linear_kernel");
    return PANO_OK;
}
