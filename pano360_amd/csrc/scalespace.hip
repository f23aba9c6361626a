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
// (r <= 16) is staged in LDS (16-byte loads and stores away from the borders, REFLECT_101
// applied on the way in at the borders), the row pass writes (32 + 2r) x 64 sums back to
// LDS, the column pass reads them.  Register blocking: a thread makes 4 adjacent row-pass
// outputs from one sliding window of 16-byte reads and 4 columns x 2 rows of column-pass
// outputs; every LDS access is 16 bytes wide and laid out conflict-free (the first version
// spent 60 % of its LDS cycles in bank conflicts).  Sums run in ascending tap order, one
// FMA per tap, like pano_blur_plane (csrc/blur.hip).
#include "common.h"

#define SS_TW 64
#define SS_TH 32
#define SS_RMAX 16
#define SS_IN_W 132                 // floats per tile row in LDS: 4 banks past a multiple of 64, so the
                                    // 16 lanes of a 16-byte read group - four rows, four 64-byte
                                    // strides in each - start on 16 different multiples of 4 banks
#define SS_MID_W 68                 // row-pass image: the same skew for the row pass's 16-byte stores;
                                    // the column pass reads 256 contiguous bytes per half wave
#define SS_NTAP 40                  // 3 leading zeros + 33 taps, rounded up to the trip of 4

struct SsTaps {
    float w[SS_NTAP];               // `lead` zeros, the n taps, zeros
    int n, lead;
};

// LDS column L of the tile is image column ga + L with ga = x0 - r - lead a multiple of 4
// (lead = (-r) & 3): 16-byte groups of the image land on 16-byte groups of the tile, and the
// row pass takes the `lead` extra columns in front of its window with zero taps.
//
// dog (optional) = out - in at the same pixel.  NT = the aperture (compile-time for the five
// steps of an octave and the base blur: every loop unrolls, the taps sit in scalar registers,
// the LDS reads of a pass are all in flight before its first FMA, and the LDS arrays are cut
// to the rows that aperture needs); NT = 0: any odd aperture up to 33, looped.
template <int NT>
__global__ __launch_bounds__(256) void scale_step_kernel(const float *__restrict__ in, int h, int w,
                                                         SsTaps taps, float *__restrict__ out,
                                                         float *__restrict__ dog) {
    constexpr int IH = NT ? SS_TH + NT - 1 : SS_TH + 2 * SS_RMAX;
    __shared__ __attribute__((aligned(16))) float s_in[IH * SS_IN_W];
    __shared__ __attribute__((aligned(16))) float s_mid[IH * SS_MID_W];
    const int tid = threadIdx.x;
    const int nt = NT ? NT : taps.n;
    const int r = nt >> 1;
    const int lead = NT ? ((4 - (r & 3)) & 3) : taps.lead;
    const int x0 = blockIdx.x * SS_TW, y0 = blockIdx.y * SS_TH;
    const int ih = SS_TH + 2 * r;
    const int ga = x0 - r - lead;                             // image column of LDS column 0
    const int trips = (nt + lead + 3) >> 2;                   // row pass: 4 taps per trip
    const int cols = SS_TW + 4 * trips + 4;                   // LDS columns the row pass reads
    constexpr int CPR = NT ? (SS_TW + 4 * ((NT + ((4 - ((NT >> 1) & 3)) & 3) + 3) / 4) + 4 + 3) / 4 : 1;

    // ---- stage the input tile -----------------------------------------------------------
    const bool inner = NT && ga >= 0 && ga + 4 * CPR <= w && y0 - r >= 0 && y0 + SS_TH + r <= h &&
                       (w & 3) == 0;                          // uniform: no reflection, 16-B rows
    if (inner) {
        // all loads are issued before the first LDS store: a load waited for on the spot is a
        // memory latency per row
        constexpr int TOT = IH * CPR, PER = (TOT + 255) / 256;
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
                *(float4 *)(s_in + ty * SS_IN_W + 4 * cc) = v[i];
            }
        }
    } else {
        // borders (and the looped form): rows / columns beyond the image arrive reflected
        // (REFLECT_101), the leading columns and those past the halo as zeros
        const int lane = tid & 63, wv = tid >> 6;
        int sx[2];
        bool real[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int L = lane + 64 * j, tc = L - lead;       // tile column
            real[j] = tc >= 0 && tc < SS_TW + 2 * r;
            sx[j] = reflect_101(x0 - r + (real[j] ? tc : 0), w);
        }
        constexpr int ROWS = (IH + 3) / 4;                    // rows per wave, at most
        float va[ROWS], vb[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int ty = wv + 4 * i;
            const int sy = __builtin_amdgcn_readfirstlane(reflect_101(y0 - r + (ty < ih ? ty : 0), h));
            const float *src = in + (size_t)sy * w;
            va[i] = src[sx[0]];
            vb[i] = src[sx[1]];
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int ty = wv + 4 * i;
            if (ty < ih) {
                s_in[ty * SS_IN_W + lane] = real[0] ? va[i] : 0.f;
                if (lane + 64 < cols) s_in[ty * SS_IN_W + lane + 64] = real[1] ? vb[i] : 0.f;
            }
        }
    }
    __syncthreads();

    // ---- row pass: ih rows x 64 outputs; a thread makes 4 adjacent outputs, four taps per trip
    // from one aligned 16-byte LDS read (window = the previous read + this one)
    constexpr int TRIPS = NT ? (NT + ((4 - ((NT >> 1) & 3)) & 3) + 3) / 4 : 0;
    if (NT) {
        // 16 adjacent outputs per thread: TRIPS + 4 reads of 16 bytes feed 16 x 4 TRIPS FMAs
        // (8 outputs from TRIPS + 2 reads: 20 bytes of LDS per output for 27 taps, now 12; the
        // kernel is bound by its LDS traffic)
        for (int i = tid; i < ih * (SS_TW / 16); i += 256) {
            const int ty = i >> 2, q = (i & 3) * 16;
            const float4 *row = (const float4 *)(s_in + ty * SS_IN_W + q);
            float4 win[TRIPS + 4];
#pragma unroll
            for (int k = 0; k < TRIPS + 4; ++k) win[k] = row[k];
            float a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = 0.f;
#pragma unroll
            for (int k = 0; k < TRIPS; ++k) {
                const float e[20] = {win[k].x,     win[k].y,     win[k].z,     win[k].w,
                                     win[k + 1].x, win[k + 1].y, win[k + 1].z, win[k + 1].w,
                                     win[k + 2].x, win[k + 2].y, win[k + 2].z, win[k + 2].w,
                                     win[k + 3].x, win[k + 3].y, win[k + 3].z, win[k + 3].w,
                                     win[k + 4].x, win[k + 4].y, win[k + 4].z, win[k + 4].w};
#pragma unroll
                for (int tt = 0; tt < 4; ++tt) {
                    const float wt = taps.w[4 * k + tt];
#pragma unroll
                    for (int j = 0; j < 16; ++j) a[j] = __builtin_fmaf(wt, e[j + tt], a[j]);
                }
            }
            float4 *m = (float4 *)(s_mid + ty * SS_MID_W + q);
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = make_float4(a[4 * j], a[4 * j + 1], a[4 * j + 2], a[4 * j + 3]);
        }
    } else {
        for (int i = tid; i < ih * (SS_TW / 4); i += 256) {
            const int ty = i >> 4, q = (i & 15) * 4;
            const float4 *row = (const float4 *)(s_in + ty * SS_IN_W + q);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            float4 lo = row[0];
            for (int k = 0; k < trips; ++k) {
                const float4 hi = row[k + 1];
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
                lo = hi;
            }
            *(float4 *)(s_mid + ty * SS_MID_W + q) = make_float4(a0, a1, a2, a3);
        }
    }
    __syncthreads();

    // ---- column pass: 64 columns x 32 rows
    if (NT) {
        // a thread makes 2 adjacent columns x 4 stacked rows from NT + 3 reads of 8 bytes (a
        // half wave = 32 column pairs of one row group = 256 contiguous bytes: conflict-free):
        // 2 (NT + 3) bytes of LDS per output where 4 columns x 2 rows took 2 (NT + 1) x 2 (eight
        // rows of one column per thread, (NT + 7) / 2 bytes, are no faster: 1.11 against 1.10 ms)
        const int cq = (tid & 31) * 2, cy = (tid >> 5) * 4;
        const float *col = s_mid + cy * SS_MID_W + cq;
        float2 v[NT + 3];
#pragma unroll
        for (int k = 0; k < NT + 3; ++k) v[k] = *(const float2 *)(col + k * SS_MID_W);
        float2 acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o] = make_float2(0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const float wk = taps.w[lead + k];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                acc[o].x = __builtin_fmaf(wk, v[k + o].x, acc[o].x);
                acc[o].y = __builtin_fmaf(wk, v[k + o].y, acc[o].y);
            }
        }
        const int x = x0 + cq;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int y = y0 + cy + o;
            if (y >= h || x >= w) break;
            const size_t at = (size_t)y * w + x;
            // the input's centre: r + lead is a multiple of 4, cq even: one aligned 8-byte read
            const float2 c = *(const float2 *)(s_in + (cy + o + r) * SS_IN_W + cq + r + lead);
            const float2 d = make_float2(acc[o].x - c.x, acc[o].y - c.y);
            if (x + 2 <= w && (w & 1) == 0) {
                *(float2 *)(out + at) = acc[o];
                if (dog) *(float2 *)(dog + at) = d;
            } else {
                out[at] = acc[o].x;
                if (dog) dog[at] = d.x;
                if (x + 1 < w) {
                    out[at + 1] = acc[o].y;
                    if (dog) dog[at + 1] = d.y;
                }
            }
        }
        return;
    }
    // (looped form) a thread makes 4 adjacent columns x 2 stacked rows
    // from 16-byte LDS reads and stores 16 bytes per row
    const int cq = (tid & 15) * 4, cy = (tid >> 4) * 2;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    const float *col = s_mid + cy * SS_MID_W + cq;
    auto fma4 = [](float wk, const float4 v, float4 &a) {
        a.x = __builtin_fmaf(wk, v.x, a.x);
        a.y = __builtin_fmaf(wk, v.y, a.y);
        a.z = __builtin_fmaf(wk, v.z, a.z);
        a.w = __builtin_fmaf(wk, v.w, a.w);
    };
    {
        float4 lo = *(const float4 *)col;
        for (int k = 0; k < nt; ++k) {
            const float4 hi = *(const float4 *)(col + (k + 1) * SS_MID_W);
            fma4(taps.w[lead + k], lo, acc0);
            fma4(taps.w[lead + k], hi, acc1);
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
        // the input's centre: r + lead is a multiple of 4, one aligned 16-byte read
        const float4 c = *(const float4 *)(s_in + (cy + o + r) * SS_IN_W + cq + r + lead);
        const float4 d = make_float4(res[o].x - c.x, res[o].y - c.y, res[o].z - c.z,
                                     res[o].w - c.w);
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
    PANO_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= 2 * SS_RMAX + 1,
                 "pano_scale_step: aperture %d must be odd and at most %d", ntaps, 2 * SS_RMAX + 1);
    PANO_REQUIRE(src != dst && src != dog, "pano_scale_step: in place is not supported");
    SsTaps t = {};
    t.n = ntaps;
    t.lead = (4 - ((ntaps >> 1) & 3)) & 3;
    for (int k = 0; k < ntaps; ++k) t.w[t.lead + k] = taps[k];
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
