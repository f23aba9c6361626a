// One Gaussian step of a SIFT octave, fused: both passes of the separable blur in one
// launch (the row-pass image lives in LDS only) and the difference-of-Gaussian layer
// emitted from the same tile.
//
// Reference: features.py:192-201 hands the whole scale space to OpenCV
// (cv2.xfeatures2d.SIFT_create().detectAndCompute); what is restated is OpenCV's
// buildGaussianPyramid / buildDoGPyramid: layer i = GaussianBlur(layer i-1, sigma_i)
// (REFLECT_101, aperture cvRound(8 sigma + 1) | 1), DoG i-1 = layer i - layer i-1.
// PARITY UNPINNED (OpenCV is not in the reference repo; oracle/sift_pyramid.py).
//
// Why one kernel: the step is pure streaming - 2 x (11..27) FMAs per pixel against 12
// algorithmic bytes (read the previous layer, write the new one and the DoG).  As a row
// launch + a column launch through a scratch plane + a subtract launch it moved 28 bytes
// per pixel; fused it moves 12 + the halo re-reads (L2 hits).
//
// Tile: 64 x 32 outputs per 256-thread workgroup.  The (32 + 2r) x (64 + 2r) input tile
// (r <= 16) is staged in LDS with REFLECT_101 applied on the way in, the row pass writes
// (32 + 2r) x 64 sums back to LDS, the column pass reads them.  Register blocking: a
// thread makes 4 adjacent row-pass outputs from one sliding window (r/2 + 1 LDS reads per
// output instead of 2r + 1) and 8 stacked column-pass outputs of one column.  Sums run in
// ascending tap order, one FMA per tap, like pano_blur_plane (csrc/blur.hip).
#include "common.h"

#define SS_TW 64
#define SS_TH 32
#define SS_RMAX 16
#define SS_IN_W 128                            // >= 64 + 2 r + 7; a multiple of 64 floats keeps the
                                               // row pass's 16-byte reads conflict-free
#define SS_IN_H (SS_TH + 2 * SS_RMAX)          // 64
#define SS_MID_PITCH (SS_TW + 4)               // 16-byte rows; the column pass reads float4
#define SS_NTAP 36                             // 2 r + 1 rounded up to the row pass's trip of 4

struct SsTaps {
    float w[SS_NTAP];                          // zero beyond n
    int n;
};

// dog (optional) = out - in at the same pixel.  NT = the aperture (compile-time for the
// five steps of an octave and the base blur, so that every loop unrolls, the taps sit in
// scalar registers and the LDS reads of a pass are all in flight before its first FMA);
// NT = 0: any odd aperture up to 33, looped.
template <int NT>
__global__ __launch_bounds__(256) void scale_step_kernel(const float *__restrict__ in, int h, int w,
                                                         SsTaps taps, float *__restrict__ out,
                                                         float *__restrict__ dog) {
    __shared__ __attribute__((aligned(16))) float s_in[SS_IN_H * SS_IN_W];
    __shared__ __attribute__((aligned(16))) float s_mid[SS_IN_H * SS_MID_PITCH];
    const int tid = threadIdx.x;
    const int nt = NT ? NT : taps.n;
    const int r = nt >> 1;
    const int x0 = blockIdx.x * SS_TW, y0 = blockIdx.y * SS_TH;
    const int iw = SS_TW + 2 * r, ih = SS_TH + 2 * r;

    // stage the input tile; rows / columns beyond the image arrive reflected.  A wave takes
    // every fourth row (the reflected source row is wave-uniform), a lane the same two columns
    // in every row (reflected once).  The row pass reads whole 16-byte groups: columns up to
    // the next multiple of 4 past the halo (+ 4) are zero-filled, so that the zero taps there
    // multiply finite numbers
    const int iwp = ((iw + 3) & ~3) + 4;
    const int lane = tid & 63, wv = tid >> 6;
    const int ca = lane, cb = lane + 64;
    const int sxa = reflect_101(x0 - r + ca, w), sxb = reflect_101(x0 - r + (cb < iw ? cb : 0), w);
    // all of a wave's loads are issued before the first LDS store: a load waited for on the
    // spot, sixteen times over, is sixteen memory latencies per tile
    const int gx = x0 - r;                                    // image column of tile column 0
    constexpr int CPR = (SS_TW + (NT ? NT : 1) - 1 + 3 + 3) / 4;   // 16-byte chunks per tile row
    const bool inner = NT && gx >= 0 && gx + 4 * CPR <= w && y0 - r >= 0 &&
                       y0 + SS_TH + r <= h && (w & 3) == 0;    // uniform: no reflection, 16-B rows
    if (inner) {
        // 16-byte loads aligned in the image: chunk c of a row covers image columns
        // ga + 4 c .. + 3 with ga = gx rounded down to a multiple of 4
        const int ga = gx & ~3, shift = gx - ga;
        constexpr int TOT = (SS_TH + NT - 1) * CPR;
        constexpr int PER = (TOT + 255) / 256;
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + 256 * i;
            const int ty = c / CPR, cc = c - ty * CPR;
            v[i] = c < TOT ? *(const float4 *)(in + (size_t)(y0 - r + ty) * w + ga + 4 * cc)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = tid + 256 * i;
            if (c < TOT) {
                const int ty = c / CPR, cc = c - ty * CPR;
                float *d = s_in + ty * SS_IN_W + 4 * cc - shift;       // tile column of v[i].x
                const int tc = 4 * cc - shift;
                if (tc >= 0) d[0] = v[i].x;
                if (tc + 1 >= 0) d[1] = v[i].y;
                if (tc + 2 >= 0) d[2] = v[i].z;
                d[3] = v[i].w;
            }
        }
        // the zero columns past the halo (the chunks above end at or past column iw - 1)
        for (int i = tid; i < ih * 8; i += 256) {
            const int ty = i >> 3, tx = CPR * 4 - 3 + (i & 7);
            if (tx >= iw && tx < iwp) s_in[ty * SS_IN_W + tx] = 0.f;
        }
    } else {
        constexpr int ROWS = (SS_IN_H + 3) / 4;               // rows per wave, at most
        float va[ROWS], vb[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int ty = wv + 4 * i;
            const int sy = __builtin_amdgcn_readfirstlane(reflect_101(y0 - r + (ty < ih ? ty : 0), h));
            const float *src = in + (size_t)sy * w;
            va[i] = src[sxa];
            vb[i] = src[sxb];
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int ty = wv + 4 * i;
            if (ty < ih) {
                s_in[ty * SS_IN_W + ca] = va[i];
                if (cb < iwp) s_in[ty * SS_IN_W + cb] = cb < iw ? vb[i] : 0.f;
            }
        }
    }
    __syncthreads();

    // row pass: ih rows x 64 outputs; a thread makes 4 adjacent outputs, four taps per trip
    // from one aligned 16-byte LDS read (window = the previous read + this one)
    constexpr int TRIPS = NT ? (NT + 3) / 4 : 0;
    const int trips = NT ? TRIPS : (nt + 3) >> 2;
    for (int i = tid; i < ih * (SS_TW / 4); i += 256) {
        const int ty = i >> 4, q = (i & 15) * 4;
        const float4 *row = (const float4 *)(s_in + ty * SS_IN_W + q);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        auto trip = [&](const float4 lo, const float4 hi, const int k) {
            const float w0 = taps.w[4 * k], w1 = taps.w[4 * k + 1], w2 = taps.w[4 * k + 2],
                        w3 = taps.w[4 * k + 3];
            // output j meets tap 4 k + t at window entry j + t
            a0 = __builtin_fmaf(w0, lo.x, a0);
            a1 = __builtin_fmaf(w0, lo.y, a1);
            a2 = __builtin_fmaf(w0, lo.z, a2);
            a3 = __builtin_fmaf(w0, lo.w, a3);
            a0 = __builtin_fmaf(w1, lo.y, a0);
            a1 = __builtin_fmaf(w1, lo.z, a1);
            a2 = __builtin_fmaf(w1, lo.w, a2);
            a3 = __builtin_fmaf(w1, hi.x, a3);
            a0 = __builtin_fmaf(w2, lo.z, a0);
            a1 = __builtin_fmaf(w2, lo.w, a1);
            a2 = __builtin_fmaf(w2, hi.x, a2);
            a3 = __builtin_fmaf(w2, hi.y, a3);
            a0 = __builtin_fmaf(w3, lo.w, a0);
            a1 = __builtin_fmaf(w3, hi.x, a1);
            a2 = __builtin_fmaf(w3, hi.y, a2);
            a3 = __builtin_fmaf(w3, hi.z, a3);
        };
        if (NT) {
            float4 win[TRIPS + 1];
#pragma unroll
            for (int k = 0; k <= TRIPS; ++k) win[k] = row[k];
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) trip(win[k], win[k + 1], k);
        } else {
            float4 lo = row[0];
            for (int k = 0; k < trips; ++k) {
                const float4 hi = row[k + 1];
                trip(lo, hi, k);
                lo = hi;
            }
        }
        float *m = s_mid + ty * SS_MID_PITCH + q;
        m[0] = a0;
        m[1] = a1;
        m[2] = a2;
        m[3] = a3;
    }
    __syncthreads();

    // column pass: 64 columns x 32 rows; a thread makes 4 adjacent columns x 2 stacked rows
    // from 16-byte LDS reads (the FMAs pair up into v_pk_fma_f32) and stores 16 bytes per row
    const int cq = (tid & 15) * 4, cy = (tid >> 4) * 2;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    const float *col = s_mid + cy * SS_MID_PITCH + cq;
    auto fma4 = [](float wk, const float4 v, float4 &a) {
        a.x = __builtin_fmaf(wk, v.x, a.x);
        a.y = __builtin_fmaf(wk, v.y, a.y);
        a.z = __builtin_fmaf(wk, v.z, a.z);
        a.w = __builtin_fmaf(wk, v.w, a.w);
    };
    if (NT) {
        float4 v[NT + 1];
#pragma unroll
        for (int k = 0; k < NT + 1; ++k) v[k] = *(const float4 *)(col + k * SS_MID_PITCH);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            fma4(taps.w[k], v[k], acc0);
            fma4(taps.w[k], v[k + 1], acc1);
        }
    } else {
        float4 lo = *(const float4 *)col;
        for (int k = 0; k < nt; ++k) {
            const float4 hi = *(const float4 *)(col + (k + 1) * SS_MID_PITCH);
            fma4(taps.w[k], lo, acc0);
            fma4(taps.w[k], hi, acc1);
            lo = hi;
        }
    }
    const int x = x0 + cq;
    const float4 res[2] = {acc0, acc1};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int y = y0 + cy + o;
        if (y >= h || x >= w) break;
        const size_t at = (size_t)y * w + x;
        const float *c = s_in + (cy + o + r) * SS_IN_W + cq + r;          // the input's centre
        const float4 d = make_float4(res[o].x - c[0], res[o].y - c[1], res[o].z - c[2],
                                     res[o].w - c[3]);
        if (x + 4 <= w && (w & 3) == 0) {
            *(float4 *)(out + at) = res[o];
            if (dog) *(float4 *)(dog + at) = d;
        } else {
            const float rv[4] = {res[o].x, res[o].y, res[o].z, res[o].w};
            const float dv[4] = {d.x, d.y, d.z, d.w};
            for (int j = 0; j < 4 && x + j < w; ++j) {
                out[at + j] = rv[j];
                if (dog) dog[at + j] = dv[j];
            }
        }
    }
}

extern "C" int pano_scale_step(pano_ctx *ctx, const float *src, int h, int w, const float *taps,
                               int ntaps, float *dst, float *dog) {
    PANO_ENTER(ctx, "pano_scale_step");
    PANO_REQUIRE(src && dst && taps, "pano_scale_step: null pointer");
    PANO_REQUIRE(h > 0 && w > 0, "pano_scale_step: bad shape %dx%d", h, w);
    PANO_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= 2 * SS_RMAX + 1 && ntaps <= SS_NTAP,
                 "pano_scale_step: aperture %d must be odd and at most %d", ntaps, 2 * SS_RMAX + 1);
    PANO_REQUIRE(src != dst && src != dog, "pano_scale_step: in place is not supported");
    SsTaps t = {};
    t.n = ntaps;
    for (int k = 0; k < ntaps; ++k) t.w[k] = taps[k];
    dim3 grid(ceil_div(w, SS_TW), ceil_div(h, SS_TH));
#define SS_LAUNCH(N)                                                                          \
    PANO_TIMED(PK_SCALE_STEP, (hipStream_t)stream,                                            \
               hipLaunchKernelGGL(scale_step_kernel<N>, grid, dim3(256), 0, (hipStream_t)stream, \
                                  src, h, w, t, dst, dog))
    switch (ntaps) {              // the apertures of SIFT's sigma 1.6, 3 layers per octave
        case 11: SS_LAUNCH(11); break;
        case 13: SS_LAUNCH(13); break;
        case 17: SS_LAUNCH(17); break;
        case 21: SS_LAUNCH(21); break;
        case 27: SS_LAUNCH(27); break;
        default: SS_LAUNCH(0); break;
    }
#undef SS_LAUNCH
    PANO_LAUNCH_CHECK("scale_step_kernel");
    return PANO_OK;
}

static int scale_step_launch(pano_ctx *ctx, const float *src, int h, int w, const float *taps,
                             int ntaps, float *dst, float *dog) {
    return pano_scale_step(ctx, src, h, w, taps, ntaps, dst, dog);
}

// The whole scale space of one frame in one call: a frame is ~60 launches, and a Python /
// ctypes round trip per launch (10-20 us) would cost more than the kernels of the small
// octaves.
extern "C" int pano_scale_space(pano_ctx *ctx, const uint8_t *frame, int h, int w, int n_octaves,
                                int n_layers, const float *taps, const int *ntaps,
                                float *const *gauss, float *const *dog, float *work) {
    PANO_ENTER(ctx, "pano_scale_space");
    PANO_REQUIRE(frame && taps && ntaps && gauss && dog && work, "pano_scale_space: null pointer");
    PANO_REQUIRE(h > 0 && w > 0 && n_octaves >= 1 && n_layers >= 1 && n_layers <= 8,
                 "pano_scale_space: bad argument");
    const float *kern[16];
    size_t off = 0;
    for (int i = 0; i < n_layers + 3; ++i) {
        kern[i] = taps + off;
        off += (size_t)ntaps[i];
    }
    float *grey = work, *base = work + (size_t)h * w;
    if (int rc = pano_gray_u8(ctx, frame, h, w, grey)) return rc;
    if (int rc = pano_resize_up2(ctx, grey, h, w, base)) return rc;
    int rows = 2 * h, cols = 2 * w;
    for (int o = 0; o < n_octaves; ++o) {
        PANO_REQUIRE(rows >= 1 && cols >= 1 && gauss[o] && dog[o],
                     "pano_scale_space: octave %d is empty", o);
        const size_t plane = (size_t)rows * cols;
        if (o == 0) {
            if (int rc = scale_step_launch(ctx, base, rows, cols, kern[0], ntaps[0], gauss[0], nullptr))
                return rc;
        }
        for (int i = 1; i < n_layers + 3; ++i)
            if (int rc = scale_step_launch(ctx, gauss[o] + (i - 1) * plane, rows, cols, kern[i],
                                           ntaps[i], gauss[o] + i * plane, dog[o] + (i - 1) * plane))
                return rc;
        if (o + 1 < n_octaves) {
            PANO_REQUIRE(rows >= 2 && cols >= 2, "pano_scale_space: octave %d cannot be halved", o);
            if (int rc = pano_decimate2(ctx, gauss[o] + (size_t)n_layers * plane, rows, cols,
                                        gauss[o + 1]))
                return rc;
            rows /= 2;
            cols /= 2;
        }
    }
    return PANO_OK;
}
