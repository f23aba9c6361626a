/*
 * pano_oracle.c - CPU restatement of the pano360 warp/blend/crop hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the *checker*: tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg are the only
 * callers.  The product (pano360_amd/) never links, imports or falls back to
 * it; the product path raises if the HIP library is missing.
 *
 * Parity status: every function that restates the reference's own NumPy code
 * is pinned bit-exactly against golden vectors produced by running the
 * reference in the build container (oracle/gen_golden.py -> tests/golden/).
 * The three OpenCV primitives the reference calls (remap, GaussianBlur,
 * pyrDown) are NOT in /root/reference and OpenCV is not installed, so for
 * those this file restates OpenCV's published algorithm (imgproc: remap with
 * INTER_BITS=5, getGaussianKernel, sepFilter2D row/column engines,
 * borderInterpolate) - PARITY UNPINNED at that boundary; it is cross-checked
 * against the independent NumPy restatement in oracle/cv2_shim.py.
 *
 * All arithmetic is float32/float64 with one rounding per operation
 * (build with -ffp-contract=off, no -ffast-math), matching NumPy's
 * elementwise semantics.  Each function cites the reference lines it follows
 * (paths relative to the reference repo root).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_REFLECT 2     /* cv2.BORDER_REFLECT      ...cba|abc|cba... */
#define ORC_REFLECT101 4  /* cv2.BORDER_REFLECT_101  ...cb|abc|ba...   */

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* OpenCV borderInterpolate: single reflections repeated until in range. */
int orc_border(int p, int len, int mode) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    int delta = (mode == ORC_REFLECT101);
    do {
        if (p < 0) p = -p - 1 + delta;
        else p = len - 1 - (p - len) - delta;
    } while ((unsigned)p >= (unsigned)len);
    return p;
}

/* cvRound(float) as x86 cvtss2si: half-to-even; NaN / overflow -> INT_MIN. */
static inline int32_t cv_round_f32(float v) {
    double r = nearbyint((double)v);   /* default rounding mode: to even */
    if (!(r >= -2147483648.0 && r < 2147483648.0)) return INT32_MIN;
    return (int32_t)r;
}

/* ------------------------------------------------------------------------
 * _hat / _add_weights                                   stitcher.py:251-263
 * rgb = float32(u8) / 255 ; alpha = float32(hat(y) * hat(x)), hat in double:
 * hat(i) = 0.5 - |(i - n/2) / n|.
 * out: [h][w][4] float32.
 * --------------------------------------------------------------------- */
static inline double hat_at(int i, int n) {
    double x = (double)i - (double)n / 2.0;
    return 0.5 - fabs(x / (double)n);
}

void orc_add_weights(const uint8_t *img, int h, int w, float *out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        double hy = hat_at(y, h);
        for (int x = 0; x < w; ++x) {
            const uint8_t *s = img + ((size_t)y * w + x) * 3;
            float *d = out + ((size_t)y * w + x) * 4;
            d[0] = (float)s[0] / 255.0f;
            d[1] = (float)s[1] / 255.0f;
            d[2] = (float)s[2] / 255.0f;
            d[3] = (float)(hy * hat_at(x, w));
        }
    }
}

/* ------------------------------------------------------------------------
 * inverse map + mask (body of stitch)                    stitcher.py:300-312
 * Caller supplies the trig tables indexed by GLOBAL mosaic column / row
 * (sin_t, cos_t over columns; tan_p over rows), evaluated with NumPy exactly
 * as the reference does (angles = index*resolution + min, :301-303).
 * proj = K R row-major (bundle_adj.py:31-33).  The 3x3 product is done in
 * double as a fused-multiply-add chain over k (what the BLAS dgemm kernel
 * NumPy dispatches to does; see DESIGN.md "inverse map"), cast to float32
 * (:306), then the float32 divide, centre shift and the bounds mask.
 * use_fma = 0 selects separate multiply/add instead (for the study in
 * tests/test_oracle_golden.py).
 * --------------------------------------------------------------------- */
void orc_inverse_map(const double *proj, const double *sin_t,
                     const double *cos_t, const double *tan_p, int gx0,
                     int gy0, int pw, int ph, int sw, int sh, int use_fma,
                     float *mapx, float *mapy, uint8_t *mask) {
    const float cx = (float)((double)sw / 2.0), cy = (float)((double)sh / 2.0);
    const float xmax = (float)(sw - 1), ymax = (float)(sh - 1);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ph; ++y) {
        const double t = tan_p[gy0 + y];
        for (int x = 0; x < pw; ++x) {
            const double s = sin_t[gx0 + x], c = cos_t[gx0 + x];
            double v[3];
            for (int r = 0; r < 3; ++r) {
                const double *p = proj + 3 * r;
                if (use_fma)
                    v[r] = fma(p[2], c, fma(p[1], t, p[0] * s));
                else
                    v[r] = (p[0] * s + p[1] * t) + p[2] * c;
            }
            const float fx = (float)v[0], fy = (float)v[1], fz = (float)v[2];
            const float px = fx / fz + cx, py = fy / fz + cy;
            uint8_t m = fz < 0.0f;
            m |= (px < 0.0f) | (px > xmax) | (py < 0.0f) | (py > ymax);
            size_t o = (size_t)y * pw + x;
            mapx[o] = px;
            mapy[o] = py;
            mask[o] = m;
        }
    }
}

/* ------------------------------------------------------------------------
 * cv2.remap(src, mapx, mapy, INTER_LINEAR, BORDER_REFLECT)
 *                                               call site stitcher.py:315-316
 * OpenCV semantics restated (unpinned): s = cvRound(v*32); int part s>>5
 * saturated to int16; frac s&31; table weights (1-fy)(1-fx), (1-fy)fx,
 * fy(1-fx), fy*fx; dst = v00*w00 + v01*w01 + v10*w10 + v11*w11 left to
 * right; every tap through borderInterpolate(REFLECT).
 * src [sh][sw][cn] float32, dst [ph][pw][cn].
 * --------------------------------------------------------------------- */
static inline int sat16(int v) {
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

void orc_remap_linear_reflect(const float *src, int sh, int sw, int cn,
                              const float *mapx, const float *mapy, int ph,
                              int pw, float *dst) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < ph; ++y) {
        for (int x = 0; x < pw; ++x) {
            size_t o = (size_t)y * pw + x;
            int32_t sx = cv_round_f32(mapx[o] * 32.0f);
            int32_t sy = cv_round_f32(mapy[o] * 32.0f);
            int fx = sx & 31, fy = sy & 31;
            int ix = sat16(sx >> 5), iy = sat16(sy >> 5);
            int x0 = orc_border(ix, sw, ORC_REFLECT);
            int x1 = orc_border(ix + 1, sw, ORC_REFLECT);
            int y0 = orc_border(iy, sh, ORC_REFLECT);
            int y1 = orc_border(iy + 1, sh, ORC_REFLECT);
            float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
            float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax;
            float w10 = ay * (1.0f - ax), w11 = ay * ax;
            const float *p00 = src + ((size_t)y0 * sw + x0) * cn;
            const float *p01 = src + ((size_t)y0 * sw + x1) * cn;
            const float *p10 = src + ((size_t)y1 * sw + x0) * cn;
            const float *p11 = src + ((size_t)y1 * sw + x1) * cn;
            float *d = dst + o * cn;
            for (int k = 0; k < cn; ++k) {
                float a = p00[k] * w00;
                a = a + p01[k] * w01;
                a = a + p10[k] * w10;
                a = a + p11[k] * w11;
                d[k] = a;
            }
        }
    }
}

/* warped[..., 3] = warped[..., 3] * (~mask)                stitcher.py:317 */
void orc_mask_alpha(float *warped, const uint8_t *mask, size_t npix) {
    for (size_t i = 0; i < npix; ++i)
        warped[i * 4 + 3] = warped[i * 4 + 3] * (mask[i] ? 0.0f : 1.0f);
}

/* ------------------------------------------------------------------------
 * cv::getGaussianKernel (float32 kernel)       used by stitcher.py:226 and
 * features.py:24 via GaussianBlur.  ksize for (0,0): cvRound(sigma*8+1)|1.
 * --------------------------------------------------------------------- */
int orc_gaussian_ksize(double sigma) {
    return ((int)nearbyint(sigma * 4.0 * 2.0 + 1.0)) | 1;
}

void orc_gaussian_kernel(int n, double sigma, float *out) {
    if (sigma <= 0) sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2x = -0.5 / (sigma * sigma), sum = 0.0;
    for (int i = 0; i < n; ++i) {
        double x = (double)i - (n - 1) * 0.5;
        out[i] = (float)exp(scale2x * x * x);
        sum += (double)out[i];
    }
    double inv = 1.0 / sum;
    for (int i = 0; i < n; ++i) out[i] = (float)((double)out[i] * inv);
}

/* ------------------------------------------------------------------------
 * sepFilter2D with a symmetric kernel, BORDER_REFLECT_101, float32:
 * row pass  s = k0*x0; s += kj*xj (j ascending)
 * col pass  s = kc*yc; s += k(c+j)*(y(c+j) + y(c-j))
 * src/dst [h][w][cn]; tmp [h][w][cn] scratch supplied by the caller.
 * --------------------------------------------------------------------- */
void orc_sep_filter(const float *src, int h, int w, int cn, const float *taps,
                    int n, float *tmp, float *dst) {
    const int r = n / 2;
    int *cidx = (int *)malloc(sizeof(int) * (size_t)(w + 2 * r));
    int *ridx = (int *)malloc(sizeof(int) * (size_t)(h + 2 * r));
    for (int i = 0; i < w + 2 * r; ++i) cidx[i] = orc_border(i - r, w, ORC_REFLECT101);
    for (int i = 0; i < h + 2 * r; ++i) ridx[i] = orc_border(i - r, h, ORC_REFLECT101);
    const size_t wc = (size_t)w * cn;
#pragma omp parallel
    {
        float *line = (float *)malloc(sizeof(float) * (size_t)(w + 2 * r) * cn);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            const float *s = src + (size_t)y * wc;
            for (int i = 0; i < w + 2 * r; ++i)
                for (int k = 0; k < cn; ++k) line[(size_t)i * cn + k] = s[(size_t)cidx[i] * cn + k];
            float *t = tmp + (size_t)y * wc;
            for (size_t i = 0; i < wc; ++i) t[i] = line[i] * taps[0];
            for (int j = 1; j < n; ++j) {
                const float kj = taps[j];
                const float *lj = line + (size_t)j * cn;
                for (size_t i = 0; i < wc; ++i) t[i] = t[i] + lj[i] * kj;
            }
        }
        free(line);
#pragma omp for schedule(static)
        for (int y = 0; y < h; ++y) {
            float *d = dst + (size_t)y * wc;
            const float *c = tmp + (size_t)ridx[y + r] * wc;
            for (size_t i = 0; i < wc; ++i) d[i] = c[i] * taps[r];
            for (int j = 1; j <= r; ++j) {
                const float kj = taps[r + j];
                const float *a = tmp + (size_t)ridx[y + r + j] * wc;
                const float *b = tmp + (size_t)ridx[y + r - j] * wc;
                for (size_t i = 0; i < wc; ++i) d[i] = d[i] + (a[i] + b[i]) * kj;
            }
        }
    }
    free(cidx);
    free(ridx);
}

/* cv2.GaussianBlur(src, (k,k) or (0,0), sigma)  stitcher.py:226, features.py:24 */
void orc_gaussian_blur(const float *src, int h, int w, int cn, int ksize,
                       double sigma, float *tmp, float *dst) {
    if (ksize <= 0) ksize = orc_gaussian_ksize(sigma);
    float *taps = (float *)malloc(sizeof(float) * (size_t)ksize);
    orc_gaussian_kernel(ksize, sigma, taps);
    orc_sep_filter(src, h, w, cn, taps, ksize, tmp, dst);
    free(taps);
}

/* ------------------------------------------------------------------------
 * Patch table shared by the blenders: warped[i] float32 [h][w][4],
 * mask[i] uint8 [h][w], rect[i] = {y0, y1, x0, x1} in mosaic coordinates.
 * --------------------------------------------------------------------- */

/* no_blend                                               stitcher.py:160-168 */
void orc_no_blend(int n, float *const *warped, uint8_t *const *mask,
                  const int *rect, int H, int W, uint8_t *mosaic) {
    memset(mosaic, 0, (size_t)H * W * 3);
    for (int i = 0; i < n; ++i) {
        const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
        const int pw = x1 - x0;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                size_t o = (size_t)(y - y0) * pw + (x - x0);
                if (mask[i][o]) continue;
                uint8_t *d = mosaic + ((size_t)y * W + x) * 3;
                const float *s = warped[i] + o * 4;
                for (int k = 0; k < 3; ++k) d[k] = (uint8_t)(255.0f * s[k]);
            }
    }
}

/* linear_blend                                           stitcher.py:171-183 */
void orc_linear_blend(int n, float *const *warped, uint8_t *const *mask,
                      const int *rect, int H, int W, uint8_t *mosaic) {
    const size_t M = (size_t)H * W;
    float *acc = (float *)calloc(M * 3, sizeof(float));
    float *wsum = (float *)calloc(M, sizeof(float));
    for (int i = 0; i < n; ++i) {
        const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
        const int pw = x1 - x0;
#pragma omp parallel for schedule(static)
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                size_t o = (size_t)(y - y0) * pw + (x - x0);
                const float *s = warped[i] + o * 4;
                size_t g = (size_t)y * W + x;
                const float a = s[3];
                for (int k = 0; k < 3; ++k) {
                    float t = mask[i][o] ? 0.0f : s[k];
                    acc[g * 3 + k] = acc[g * 3 + k] + t * a;
                }
                wsum[g] = wsum[g] + a;
            }
    }
#pragma omp parallel for schedule(static)
    for (size_t g = 0; g < M; ++g) {
        float ws = wsum[g] == 0.0f ? 1.0f : wsum[g];
        for (int k = 0; k < 3; ++k)
            mosaic[g * 3 + k] = (uint8_t)(255.0f * (acc[g * 3 + k] / ws));
    }
    free(acc);
    free(wsum);
}

/* ownership: first-index argmax of alpha, -1 where the sum is 0
 *                                                        stitcher.py:196-204 */
void orc_ownership(int n, float *const *warped, const int *rect, int H, int W,
                   int32_t *owner) {
    const size_t M = (size_t)H * W;
    float *best = (float *)calloc(M, sizeof(float));
    for (size_t g = 0; g < M; ++g) owner[g] = -1;
    for (int i = 0; i < n; ++i) {
        const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
        const int pw = x1 - x0;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                float a = warped[i][((size_t)(y - y0) * pw + (x - x0)) * 4 + 3];
                size_t g = (size_t)y * W + x;
                if (a > best[g]) { best[g] = a; owner[g] = i; }
            }
    }
    free(best);
}

/* multiband_blend                                        stitcher.py:186-241
 * Restated functionally (the reference aliases and mutates in place):
 *   alpha_i := (owner == i)                                        :207-208
 *   G_k,i   := GaussianBlur(warped_i with sharp alpha, 4*sqrt(2k+1)) :218,226
 *   tile_0 = (I - G_0).rgb, a = G_0.a ; tile_k = (G_{k-1} - G_k).rgb,
 *   a = G_k.a ; tile_{L-1} = G_{L-2}                               :224-229
 *   layer_k = sum_i tile.rgb*tile.a ; wsum_k = sum_i tile.a        :231-232
 *   layer_k[~allmask] = 0 ; wsum_k[==0] = 1 ; mosaic += layer/wsum :236-238
 *   clip [0,1]; uint8(255*mosaic)                                  :240-241
 * The caller's warped alpha IS overwritten with the sharp mask, as in the
 * reference.  float_out (optional, [H][W][3]) receives the pre-quantisation
 * mosaic for the 1e-4 relative-error criterion.
 * --------------------------------------------------------------------- */
void orc_multiband_blend(int n, float *const *warped, uint8_t *const *mask,
                         const int *rect, int H, int W, int n_levels,
                         uint8_t *mosaic, float *float_out) {
    const size_t M = (size_t)H * W;
    int32_t *owner = (int32_t *)malloc(M * sizeof(int32_t));
    orc_ownership(n, warped, rect, H, W, owner);
    uint8_t *allmask = (uint8_t *)calloc(M, 1);
    size_t maxpix = 0;
    for (int i = 0; i < n; ++i) {
        const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
        const int pw = x1 - x0;
        size_t np_ = (size_t)(y1 - y0) * pw;
        if (np_ > maxpix) maxpix = np_;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                size_t o = (size_t)(y - y0) * pw + (x - x0), g = (size_t)y * W + x;
                warped[i][o * 4 + 3] = (owner[g] == i) ? 1.0f : 0.0f;
                if (!mask[i][o]) allmask[g] = 1;
            }
    }
    float *acc = (float *)calloc(M * 3, sizeof(float));
    float *layer = (float *)malloc(M * 3 * sizeof(float));
    float *wsum = (float *)malloc(M * sizeof(float));
    float **prev = (float **)calloc((size_t)n, sizeof(float *));
    float *tmp = (float *)malloc(maxpix * 4 * sizeof(float));
    for (int lvl = 0; lvl < n_levels; ++lvl) {
        const double sigma = sqrt(2 * lvl + 1.0) * 4;
        const int last = lvl == n_levels - 1;
        memset(layer, 0, M * 3 * sizeof(float));
        memset(wsum, 0, M * sizeof(float));
        for (int i = 0; i < n; ++i) {
            const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
            const int pw = x1 - x0, ph = y1 - y0;
            const size_t np_ = (size_t)ph * pw;
            float *blur = NULL;
            if (!last) {
                blur = (float *)malloc(np_ * 4 * sizeof(float));
                orc_gaussian_blur(warped[i], ph, pw, 4, 0, sigma, tmp, blur);
            }
            const float *hi = prev[i] ? prev[i] : warped[i]; /* what gets the minus */
#pragma omp parallel for schedule(static)
            for (int y = 0; y < ph; ++y)
                for (int x = 0; x < pw; ++x) {
                    size_t o = (size_t)y * pw + x, g = (size_t)(y + y0) * W + (x + x0);
                    float a, rgb[3];
                    if (!last) {
                        a = blur[o * 4 + 3];
                        for (int k = 0; k < 3; ++k) rgb[k] = hi[o * 4 + k] - blur[o * 4 + k];
                    } else {
                        a = hi[o * 4 + 3];
                        for (int k = 0; k < 3; ++k) rgb[k] = hi[o * 4 + k];
                    }
                    for (int k = 0; k < 3; ++k) layer[g * 3 + k] = layer[g * 3 + k] + rgb[k] * a;
                    wsum[g] = wsum[g] + a;
                }
            if (!last) {
                free(prev[i]);
                prev[i] = blur;
            }
        }
#pragma omp parallel for schedule(static)
        for (size_t g = 0; g < M; ++g) {
            float ws = wsum[g] == 0.0f ? 1.0f : wsum[g];
            for (int k = 0; k < 3; ++k) {
                float l = allmask[g] ? layer[g * 3 + k] : 0.0f;
                acc[g * 3 + k] = acc[g * 3 + k] + l / ws;
            }
        }
    }
    for (size_t g = 0; g < M * 3; ++g) {
        float v = acc[g];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        if (float_out) float_out[g] = v;
        mosaic[g] = (uint8_t)(255.0f * v);
    }
    for (int i = 0; i < n; ++i) free(prev[i]);
    free(prev); free(tmp); free(wsum); free(layer); free(acc); free(allmask); free(owner);
}

/* _valid                                                 stitcher.py:266-271 */
void orc_valid(int n, uint8_t *const *mask, const int *rect, int H, int W,
               uint8_t *valid) {
    memset(valid, 0, (size_t)H * W);
    for (int i = 0; i < n; ++i) {
        const int y0 = rect[4 * i], y1 = rect[4 * i + 1], x0 = rect[4 * i + 2], x1 = rect[4 * i + 3];
        const int pw = x1 - x0;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x)
                if (!mask[i][(size_t)(y - y0) * pw + (x - x0)]) valid[(size_t)y * W + x] = 1;
    }
}

/* ------------------------------------------------------------------------
 * crop_mosaic rectangle                                  stitcher.py:340-369
 * For every row i (as bottom row) and column j the candidate is the widest
 * run around j whose column heights are all >= heights[j]; the winner is the
 * FIRST candidate, scanning rows then columns, with strictly larger area.
 * Reference quirk kept: its right-extent loop starts at width-1 and stops at
 * 1 (:359), so column 0's right extent is always 0 - its candidate is the
 * one-pixel-wide column.  Nearest-smaller extents are found here with a
 * monotonic stack (equivalent to the reference's pointer jumping).
 * rect_out = {y0, x0, h, w}; returns 0 when nothing is valid (the reference
 * raises UnboundLocalError there), 1 otherwise.
 * --------------------------------------------------------------------- */
int orc_crop_rect(const uint8_t *valid, int H, int W, int64_t *rect_out) {
    int32_t *hgt = (int32_t *)calloc((size_t)W, sizeof(int32_t));
    int32_t *lft = (int32_t *)malloc((size_t)W * sizeof(int32_t));
    int32_t *rgt = (int32_t *)malloc((size_t)W * sizeof(int32_t));
    int32_t *stk = (int32_t *)malloc((size_t)W * sizeof(int32_t));
    int64_t best = 0;
    int found = 0;
    for (int i = 0; i < H; ++i) {
        const uint8_t *row = valid + (size_t)i * W;
        for (int j = 0; j < W; ++j) hgt[j] = row[j] ? hgt[j] + 1 : 0;
        int sp = 0;
        for (int j = 0; j < W; ++j) {           /* nearest strictly smaller on the left */
            while (sp > 0 && hgt[stk[sp - 1]] >= hgt[j]) --sp;
            lft[j] = sp ? stk[sp - 1] + 1 : 0;
            stk[sp++] = j;
        }
        sp = 0;
        for (int j = W - 1; j >= 0; --j) {      /* ... and on the right */
            while (sp > 0 && hgt[stk[sp - 1]] >= hgt[j]) --sp;
            rgt[j] = sp ? stk[sp - 1] - 1 : W - 1;
            stk[sp++] = j;
        }
        rgt[0] = 0;                              /* the reference never updates rights[0] */
        for (int j = 0; j < W; ++j) {
            int64_t area = (int64_t)(rgt[j] - lft[j] + 1) * hgt[j];
            if (area > best) {
                best = area;
                rect_out[0] = i - hgt[j] + 1;
                rect_out[1] = lft[j];
                rect_out[2] = hgt[j];
                rect_out[3] = rgt[j] - lft[j] + 1;
                found = 1;
            }
        }
    }
    free(hgt); free(lft); free(rgt); free(stk);
    return found;
}

/* ------------------------------------------------------------------------
 * cv2.pyrDown on one float32 plane           features.py:155 (MSOP pyramid)
 * [1 4 6 4 1] along rows then columns, REFLECT_101, even samples, x1/256.
 * dst is [(h+1)/2][(w+1)/2]; tmp is [h][(w+1)/2].
 * --------------------------------------------------------------------- */
void orc_pyr_down(const float *src, int h, int w, float *tmp, float *dst) {
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; ++y) {
        const float *s = src + (size_t)y * w;
        for (int x = 0; x < ow; ++x) {
            int c = 2 * x;
            float l2 = s[orc_border(c - 2, w, ORC_REFLECT101)], l1 = s[orc_border(c - 1, w, ORC_REFLECT101)];
            float r1 = s[orc_border(c + 1, w, ORC_REFLECT101)], r2 = s[orc_border(c + 2, w, ORC_REFLECT101)];
            tmp[(size_t)y * ow + x] = s[c] * 6.0f + (l1 + r1) * 4.0f + l2 + r2;
        }
    }
#pragma omp parallel for schedule(static)
    for (int y = 0; y < oh; ++y) {
        int c = 2 * y;
        const float *m = tmp + (size_t)c * ow;
        const float *l2 = tmp + (size_t)orc_border(c - 2, h, ORC_REFLECT101) * ow;
        const float *l1 = tmp + (size_t)orc_border(c - 1, h, ORC_REFLECT101) * ow;
        const float *r1 = tmp + (size_t)orc_border(c + 1, h, ORC_REFLECT101) * ow;
        const float *r2 = tmp + (size_t)orc_border(c + 2, h, ORC_REFLECT101) * ow;
        for (int x = 0; x < ow; ++x)
            dst[(size_t)y * ow + x] = (m[x] * 6.0f + (l1[x] + r1[x]) * 4.0f + l2[x] + r2[x]) * (1.0f / 256.0f);
    }
}

/* ------------------------------------------------------------------------
 * Overlap statistics of equalize_gains for one pair (i, j)
 *                                                       stitcher.py:44-63
 * overlap = cv2.warpPerspective(img_j, hom, (w, h), BORDER_TRANSPARENT);
 * mask = overlap[..., 3] != 0; size = sum(mask);
 * mean_i = np.mean(img_i[mask, :3]); mean_j = np.mean(overlap[mask, :3]).
 *
 * OpenCV semantics restated (unpinned, see cv2_shim.warpPerspective): minv =
 * cv::invert(hom) is passed in; per pixel, in double, with x = block start +
 * x1 (block width bw0): W = 32 / (W0 + m6*x1) (0 if the denominator is 0),
 * X = cvRound(clamp((X0 + m0*x1) * W)), tap = X >> 5 saturated to int16,
 * frac = X & 31; a pixel is written only when all four taps are inside
 * (BORDER_TRANSPARENT, cn = 4) and the destination starts as zeros.
 *
 * NumPy semantics restated (checked against NumPy 2.2 in this container):
 * np.mean over float32 = float32 sum / float32(count); the sum runs over the
 * row-major (pixel, channel) sequence in chunks of 8192 values (the ufunc
 * buffer size), chunk results added left to right, each chunk by pairwise
 * summation (8 interleaved partial sums below 128 values, halves above).
 *
 * img_i, img_j: [h][w][4] float32 (RGBA as _add_weights leaves them).
 * out: size (pixel count), mean_i, mean_j.
 * --------------------------------------------------------------------- */
static float np_pairwise_f32(const float *a, size_t n) {
    if (n < 8) {
        float res = 0.0f;
        for (size_t i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        size_t i;
        for (i = 0; i < 8; ++i) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] = r[k] + a[i + k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_f32(a, n2) + np_pairwise_f32(a + n2, n - n2);
}

static float np_sum_f32(const float *a, size_t n) {
    float total = 0.0f;
    for (size_t s = 0; s < n; s += 8192) {
        size_t len = n - s < 8192 ? n - s : 8192;
        float part = np_pairwise_f32(a + s, len);
        total = s == 0 ? part : total + part;
    }
    return total;
}

void orc_overlap_stats(const float *img_i, const float *img_j, int h, int w,
                       const double *m, int bw0, double *size_out,
                       float *mean_i, float *mean_j) {
    float *gi = (float *)malloc((size_t)h * w * 3 * sizeof(float));
    float *gj = (float *)malloc((size_t)h * w * 3 * sizeof(float));
    size_t n = 0;
    for (int y = 0; y < h; ++y) {
        for (int x = 0; x < w; ++x) {
            double xb = (double)((x / bw0) * bw0), x1 = (double)(x % bw0);
            double X0 = m[0] * xb + m[1] * (double)y + m[2];
            double Y0 = m[3] * xb + m[4] * (double)y + m[5];
            double W0 = m[6] * xb + m[7] * (double)y + m[8];
            double W = W0 + m[6] * x1;
            W = W != 0.0 ? 32.0 / W : 0.0;
            double fX = fmax(-2147483648.0, fmin(2147483647.0, (X0 + m[0] * x1) * W));
            double fY = fmax(-2147483648.0, fmin(2147483647.0, (Y0 + m[3] * x1) * W));
            long X = lrint(fX), Y = lrint(fY);
            int sx = sat16((int)(X >> 5)), sy = sat16((int)(Y >> 5));
            if (sx < 0 || sx >= w - 1 || sy < 0 || sy >= h - 1) continue;
            float ax = (float)(X & 31) * (1.0f / 32.0f), ay = (float)(Y & 31) * (1.0f / 32.0f);
            float w00 = (1.0f - ay) * (1.0f - ax), w01 = (1.0f - ay) * ax;
            float w10 = ay * (1.0f - ax), w11 = ay * ax;
            const float *p00 = img_j + ((size_t)sy * w + sx) * 4;
            const float *p01 = p00 + 4, *p10 = p00 + (size_t)w * 4, *p11 = p10 + 4;
            float v[4];
            for (int k = 0; k < 4; ++k) {
                float a = p00[k] * w00;
                a = a + p01[k] * w01;
                a = a + p10[k] * w10;
                a = a + p11[k] * w11;
                v[k] = a;
            }
            if (v[3] == 0.0f) continue;
            const float *pi = img_i + ((size_t)y * w + x) * 4;
            for (int k = 0; k < 3; ++k) {
                gi[n * 3 + k] = pi[k];
                gj[n * 3 + k] = v[k];
            }
            ++n;
        }
    }
    *size_out = (double)n;
    *mean_i = n ? np_sum_f32(gi, n * 3) / (float)(n * 3) : 0.0f;
    *mean_j = n ? np_sum_f32(gj, n * 3) / (float)(n * 3) : 0.0f;
    free(gi);
    free(gj);
}
