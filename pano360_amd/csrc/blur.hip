// Separable Gaussian blur of float planes (REFLECT_101), the arithmetic of
// cv2.GaussianBlur at stitcher.py:226 (multiband levels, 33..97 taps) and
// features.py:24 (gaussian_filter, 5/11 taps), plus cv2.pyrDown
// (features.py:155).
//
// This is the one stage that is NOT HBM-bound: 2*taps FMA per pixel and
// channel against 8 bytes of traffic, so both passes are organised around the
// f32 VALU: every LDS read of 16 B feeds 32 FMAs held in registers, the taps
// arrive as SGPR operands through the scalar cache, and nothing is re-read
// from HBM inside a tile.
//
//   row pass : wave = one image row, lane = 8 consecutive outputs; the row
//              segment (512 + 2R px) is staged in LDS with 4 pad floats per 8
//              so the lanes' 32-B-strided b128 reads hit 16 distinct slots.
//   col pass : 16x16 threads, thread = 4 adjacent columns x 8 rows; the
//              (128 + 2R) x 64 tile is staged in LDS, a b128 read gives one
//              input row for 4 columns and feeds the 8 row-outputs above it.
//
// Taps come as a zero-padded table wz: PANO_TAP_LEAD zeros, the taps, zeros;
// the weight of input p for output o (both relative to the thread's first
// input / output) is wz[p - o + 7], which is 0 outside the aperture, so the
// inner loops need no edge cases.  Sums run in ascending tap order with one
// FMA per tap.
//
// Windows (include/pano360.h): the passes only produce rectangle A of a patch.
// The row pass reads the colour planes over window V and writes rows V x
// columns A into scratch; the column pass reads that and writes A.  Border
// reflection always uses the full patch size, so results equal blurring the
// whole patch and cutting A out of it.
#include "common.h"

#define ROW_TW 512            // outputs per wave-row
#define ROW_RMAX 64           // (PANO_MAX_TAPS - 1) / 2
#define ROW_TILE (ROW_TW + 2 * ROW_RMAX + 16)
#define ROW_LDS (ROW_TILE + (ROW_TILE / 8) * 4 + 8)

#define COL_TW 64
#define COL_TH 128

struct RowGeom {
    int w;              // full patch width (reflection period)
    int vx0, vw, vh;    // window V: x origin (patch-local), width, rows
    int vpitch;         // source plane pitch
    int ax0, aw;        // output columns: patch-local origin, count
    int apitch;         // destination pitch
    // sharp-alpha source: owner[(oy + row) * opitch + ox + patch_col] == oindex
    int opitch, oy, ox, oindex;
};

struct RowJobs {
    const float *src[4];    // plane over V; NULL -> sharp alpha from the owner map
    float *dst[4];          // [vh][apitch]
};

__device__ __forceinline__ int row_pos(int i) { return i + ((i >> 3) << 2); }

__global__ __launch_bounds__(256) void blur_rows_kernel(
    RowJobs jobs, RowGeom g, const float *wz_global, int ntaps,
    const int16_t *__restrict__ owner) {
    const kptr_f32 wz = (kptr_f32)(uintptr_t)wz_global;
    __shared__ __attribute__((aligned(16))) float s_row[4][ROW_LDS];
    const float *__restrict__ src = jobs.src[blockIdx.z];
    float *__restrict__ dst = jobs.dst[blockIdx.z];
    const int lane = threadIdx.x, wv = threadIdx.y;
    const int R = ntaps >> 1;
    const int xt = blockIdx.x * ROW_TW;          // first output column, A-relative
    const int y = blockIdx.y * 4 + wv;
    const int yc = y < g.vh ? y : g.vh - 1;
    float *tile = s_row[wv];

    const int steps = (ntaps + 7 + 3) >> 2;      // 4 inputs per step
    const int need = 8 * 63 + 4 * steps;         // tile entries read by lane 63
    for (int i = lane; i < need; i += 64) {
        const int pcol = reflect_101(g.ax0 + xt - R + i, g.w);
        float v;
        if (src) {
            int vc = pcol - g.vx0;               // inside V by construction; clamp anyway
            vc = vc < 0 ? 0 : (vc >= g.vw ? g.vw - 1 : vc);
            v = src[(size_t)yc * g.vpitch + vc];
        } else {
            v = owner[(size_t)(g.oy + yc) * g.opitch + g.ox + pcol] == g.oindex ? 1.0f : 0.0f;
        }
        tile[row_pos(i)] = v;
    }
    __syncthreads();

    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.0f;
    const float *base = tile + 12 * lane;        // row_pos(8*lane)
    for (int s = 0; s < steps; ++s) {
        const int p = 4 * s;
        const float4 v = *(const float4 *)(base + p + ((p >> 3) << 2));
        float wq[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) wq[k] = wz[p + k];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            acc[o] = __builtin_fmaf(wq[7 - o + 0], v.x, acc[o]);
            acc[o] = __builtin_fmaf(wq[7 - o + 1], v.y, acc[o]);
            acc[o] = __builtin_fmaf(wq[7 - o + 2], v.z, acc[o]);
            acc[o] = __builtin_fmaf(wq[7 - o + 3], v.w, acc[o]);
        }
    }
    if (y < g.vh) {
        const int x = xt + 8 * lane;
        float *d = dst + (size_t)y * g.apitch + x;
        if (x + 8 <= g.apitch) {
            *(float4 *)d = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4 *)(d + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (x + o < g.aw) d[o] = acc[o];
        }
    }
}

struct ColGeom {
    int h;              // full patch height (reflection period)
    int vy0, vh;        // rows held by the source: patch-local origin, count
    int ay0, ah;        // output rows: patch-local origin, count
    int aw, apitch;     // columns, pitch of source and destination
};

struct ColJobs {
    const float *src[4];    // [vh][apitch]
    float *dst[4];          // [ah][apitch]
};

__global__ __launch_bounds__(256) void blur_cols_kernel(ColJobs jobs, ColGeom g,
                                                        const float *wz_global,
                                                        int ntaps) {
    const kptr_f32 wz = (kptr_f32)(uintptr_t)wz_global;
    extern __shared__ __attribute__((aligned(16))) float s_col[];   // [rows][64]
    const float *__restrict__ src = jobs.src[blockIdx.z];
    float *__restrict__ dst = jobs.dst[blockIdx.z];
    const int tx = threadIdx.x, ty = threadIdx.y;           // 16 x 16
    const int R = ntaps >> 1;
    const int x0 = blockIdx.x * COL_TW, y0 = blockIdx.y * COL_TH;   // A-relative
    const int steps = ntaps + 7;
    const int rows = 8 * 15 + steps;                        // rows read by ty = 15
    const int x = x0 + 4 * tx;

    for (int r = ty; r < rows; r += 16) {
        int vr = reflect_101(g.ay0 + y0 - R + r, g.h) - g.vy0;
        vr = vr < 0 ? 0 : (vr >= g.vh ? g.vh - 1 : vr);     // inside by construction
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x < g.apitch) v = *(const float4 *)(src + (size_t)vr * g.apitch + x);
        *(float4 *)(s_col + r * COL_TW + 4 * tx) = v;
    }
    __syncthreads();

    float4 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *base = s_col + (8 * ty) * COL_TW + 4 * tx;
    for (int p = 0; p < steps; ++p) {
        const float4 v = *(const float4 *)(base + p * COL_TW);
        float wq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wq[k] = wz[p + k];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float wgt = wq[7 - o];
            acc[o].x = __builtin_fmaf(wgt, v.x, acc[o].x);
            acc[o].y = __builtin_fmaf(wgt, v.y, acc[o].y);
            acc[o].z = __builtin_fmaf(wgt, v.z, acc[o].z);
            acc[o].w = __builtin_fmaf(wgt, v.w, acc[o].w);
        }
    }
    if (x < g.apitch) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int y = y0 + 8 * ty + o;
            if (y < g.ah) *(float4 *)(dst + (size_t)y * g.apitch + x) = acc[o];
        }
    }
}

static int launch_rows(const RowJobs &jobs, int njobs, const RowGeom &g, const float *wz,
                       int ntaps, const int16_t *owner, hipStream_t stream) {
    dim3 block(64, 4), grid(ceil_div(g.aw, ROW_TW), ceil_div(g.vh, 4), njobs);
    PANO_TIMED(PK_BLUR_ROWS, stream,
               hipLaunchKernelGGL(blur_rows_kernel, grid, block, 0, stream, jobs, g, wz,
                                  ntaps, owner));
    PANO_LAUNCH_CHECK("blur_rows_kernel");
    return PANO_OK;
}

static int launch_cols(const ColJobs &jobs, int njobs, const ColGeom &g, const float *wz,
                       int ntaps, hipStream_t stream) {
    const int rows = 8 * 15 + ntaps + 7;
    const size_t lds = (size_t)rows * COL_TW * sizeof(float);
    dim3 block(16, 16), grid(ceil_div(g.aw, COL_TW), ceil_div(g.ah, COL_TH), njobs);
    static bool lds_opt_in = false;   // tiles above 64 KiB need the opt-in
    if (!lds_opt_in) {
        PANO_HIP(hipFuncSetAttribute((const void *)blur_cols_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        lds_opt_in = true;
    }
    PANO_TIMED(PK_BLUR_COLS, stream,
               hipLaunchKernelGGL(blur_cols_kernel, grid, block, lds, stream, jobs, g, wz,
                                  ntaps));
    PANO_LAUNCH_CHECK("blur_cols_kernel");
    return PANO_OK;
}

static int check_taps(int ntaps, const char *who) {
    PANO_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= PANO_MAX_TAPS,
                 "%s: aperture %d must be odd and within [1, %d]", who, ntaps, PANO_MAX_TAPS);
    return PANO_OK;
}

extern "C" int pano_blur_plane(const float *src, float *dst, float *tmp, int h,
                               int w, int pitch, const float *taps, int ntaps,
                               void *stream) {
    PANO_REQUIRE(src && dst && tmp && taps, "pano_blur_plane: null pointer");
    PANO_REQUIRE(h > 0 && w > 0 && pitch >= w && (pitch & 3) == 0,
                 "pano_blur_plane: bad shape %dx%d pitch %d", h, w, pitch);
    if (int rc = check_taps(ntaps, "pano_blur_plane")) return rc;
    RowJobs rj = {};
    rj.src[0] = src;
    rj.dst[0] = tmp;
    RowGeom rg = {};
    rg.w = w; rg.vx0 = 0; rg.vw = w; rg.vh = h; rg.vpitch = pitch;
    rg.ax0 = 0; rg.aw = w; rg.apitch = pitch;
    if (int rc = launch_rows(rj, 1, rg, taps, ntaps, nullptr, (hipStream_t)stream)) return rc;
    ColJobs cj = {};
    cj.src[0] = tmp;
    cj.dst[0] = dst;
    ColGeom cg = {};
    cg.h = h; cg.vy0 = 0; cg.vh = h; cg.ay0 = 0; cg.ah = h; cg.aw = w; cg.apitch = pitch;
    return launch_cols(cj, 1, cg, taps, ntaps, (hipStream_t)stream);
}

// Shared by blur.hip and blend.hip: structural checks of a patch record.
int pano_check_patch(const pano_patch *p, const char *who) {
    PANO_REQUIRE(p->h > 0 && p->w > 0, "%s: empty patch %dx%d", who, p->h, p->w);
    PANO_REQUIRE(p->vy0 >= 0 && p->vx0 >= 0 && p->vh >= 0 && p->vw >= 0 &&
                     p->vy0 + p->vh <= p->h && p->vx0 + p->vw <= p->w,
                 "%s: window V outside the patch", who);
    PANO_REQUIRE(p->ay0 >= p->vy0 && p->ax0 >= p->vx0 && p->ah >= 0 && p->aw >= 0 &&
                     p->ay0 + p->ah <= p->vy0 + p->vh && p->ax0 + p->aw <= p->vx0 + p->vw,
                 "%s: rectangle A outside window V", who);
    PANO_REQUIRE(p->vpitch >= p->vw && (p->vpitch & 3) == 0 && p->apitch >= p->aw &&
                     (p->apitch & 3) == 0,
                 "%s: bad pitch (v %d for %d, a %d for %d)", who, p->vpitch, p->vw, p->apitch,
                 p->aw);
    return PANO_OK;
}

extern "C" int pano_multiband_blur(const pano_patch *patch, int index,
                                   const int16_t *owner, int W, const float *taps,
                                   const int *ntaps, int n_blur, float *scratch,
                                   void *stream) {
    PANO_REQUIRE(patch && owner && taps && ntaps && scratch, "pano_multiband_blur: null pointer");
    PANO_REQUIRE(n_blur >= 0 && n_blur < PANO_MAX_LEVELS, "pano_multiband_blur: %d blur levels", n_blur);
    if (int rc = pano_check_patch(patch, "pano_multiband_blur")) return rc;
    if (patch->ah == 0 || patch->aw == 0 || n_blur == 0) return PANO_OK;   // owns nothing
    PANO_REQUIRE(patch->planes && patch->blurred, "pano_multiband_blur: patch without planes/blurred");
    PANO_REQUIRE(W > 0, "pano_multiband_blur: bad mosaic width %d", W);

    RowGeom rg = {};
    rg.w = patch->w; rg.vx0 = patch->vx0; rg.vw = patch->vw; rg.vh = patch->vh;
    rg.vpitch = patch->vpitch; rg.ax0 = patch->ax0; rg.aw = patch->aw; rg.apitch = patch->apitch;
    rg.opitch = W; rg.oy = patch->y0 + patch->vy0; rg.ox = patch->x0; rg.oindex = index;
    ColGeom cg = {};
    cg.h = patch->h; cg.vy0 = patch->vy0; cg.vh = patch->vh; cg.ay0 = patch->ay0;
    cg.ah = patch->ah; cg.aw = patch->aw; cg.apitch = patch->apitch;

    const size_t vplane = (size_t)patch->vh * patch->vpitch;
    const size_t splane = (size_t)patch->vh * patch->apitch;
    const size_t aplane = (size_t)patch->ah * patch->apitch;
    size_t off = 0;
    for (int k = 0; k < n_blur; ++k) {
        if (int rc = check_taps(ntaps[k], "pano_multiband_blur")) return rc;
        const float *wz = taps + off;
        off += (size_t)ntaps[k] + PANO_TAP_PAD;
        RowJobs rj = {};
        ColJobs cj = {};
        for (int c = 0; c < 4; ++c) {
            rj.src[c] = c < 3 ? patch->planes + c * vplane : nullptr;
            rj.dst[c] = scratch + c * splane;
            cj.src[c] = scratch + c * splane;
            cj.dst[c] = patch->blurred + ((size_t)k * 4 + c) * aplane;
        }
        if (int rc = launch_rows(rj, 4, rg, wz, ntaps[k], owner, (hipStream_t)stream)) return rc;
        if (int rc = launch_cols(cj, 4, cg, wz, ntaps[k], (hipStream_t)stream)) return rc;
    }
    return PANO_OK;
}

// ---- cv2.pyrDown -----------------------------------------------------------
// c*6 + (l1 + r1)*4 + l2 + r2 along rows, the same along columns, then /256.
__global__ __launch_bounds__(256) void pyr_down_kernel(const float *__restrict__ src,
                                                       int h, int w,
                                                       float *__restrict__ dst,
                                                       int oh, int ow) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= ow || y >= oh) return;
    int cx[5], cy[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        cx[k] = reflect_101(2 * x - 2 + k, w);
        cy[k] = reflect_101(2 * y - 2 + k, h);
    }
    float rowv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float *s = src + (size_t)cy[k] * w;
        rowv[k] = s[cx[2]] * 6.0f + (s[cx[1]] + s[cx[3]]) * 4.0f + s[cx[0]] + s[cx[4]];
    }
    dst[(size_t)y * ow + x] =
        (rowv[2] * 6.0f + (rowv[1] + rowv[3]) * 4.0f + rowv[0] + rowv[4]) * (1.0f / 256.0f);
}

extern "C" int pano_pyr_down(const float *src, int h, int w, float *dst, void *stream) {
    PANO_REQUIRE(src && dst, "pano_pyr_down: null pointer");
    PANO_REQUIRE(h > 0 && w > 0, "pano_pyr_down: bad shape %dx%d", h, w);
    const int oh = (h + 1) / 2, ow = (w + 1) / 2;
    dim3 block(64, 4), grid(ceil_div(ow, 64), ceil_div(oh, 4));
    PANO_TIMED(PK_PYR_DOWN, (hipStream_t)stream,
               hipLaunchKernelGGL(pyr_down_kernel, grid, block, 0, (hipStream_t)stream, src,
                                  h, w, dst, oh, ow));
    PANO_LAUNCH_CHECK("pyr_down_kernel");
    return PANO_OK;
}
