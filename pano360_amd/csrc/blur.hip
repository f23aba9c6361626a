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
//              segment (512 + 2R px) is staged in LDS ONCE, with 4 pad floats
//              per 8 so the lanes' 32-B-strided b128 reads hit 16 distinct
//              slots, and every blur level is computed from that one tile.
//   col pass : 16x16 threads, thread = 4 adjacent columns x 8 rows; the
//              (128 + 2r) x 64 tile is staged in LDS, a b128 read gives one
//              input row for 4 columns and feeds the 8 row-outputs above it.
//              One launch per level so the tile is sized to that level.
//
// Every launch covers ALL patches (blockIdx.z = patch x channel): a stitch is
// a dozen launches with tens of thousands of workgroups each, not hundreds of
// small ones.
//
// Taps come as a zero-padded table wz (layout in include/pano360.h): the
// weight of input p for output o (both relative to the thread's first input /
// output) is wz[p - o + 7], which is 0 outside the aperture, so the inner
// loops need no edge cases.  Sums run in ascending tap order, one FMA per tap.
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
// input rows a column-pass thread walks: aperture + its 8 outputs - 1, rounded
// up to the 8-row trip of the inner loop
#define COL_STEPS(ntaps) (((ntaps) + 7 + 7) & ~7)

struct LevelDesc {
    const float *wz;     // row-pass table: 7 + extra leading zeros
    int ntaps;           // true aperture
    int start;           // row pass: first tile entry of lane 0 = (R - r) - extra
    int steps;           // row pass: 4-input steps = ceil((ntaps + extra + 7) / 4)
};

struct Levels {
    LevelDesc lv[PANO_MAX_LEVELS];
    int n;               // number of blur levels
    int rmax;            // largest radius: halo of the staged row tile
    int need;            // tile entries the row pass reads (lane 63, widest level)
};

__device__ __forceinline__ int row_pos(int i) { return i + ((i >> 3) << 2); }

// nch = channels per patch (4: R,G,B + sharp alpha from the owner map; 1: a
// plain plane); alpha_ch = channel whose source is the owner map, or -1.
// Column tile (tx, ty) of a record is "active" when it holds a pixel that is not
// interior (blend.hip: interior map): only those tiles are ever gathered, so only
// they - and the scratch rows they read - are computed.  flags == NULL: all active.
__device__ __forceinline__ bool col_tile_active(const uint8_t *__restrict__ flags,
                                                const pano_patch &p, int tx, int ty) {
    return !flags || flags[p.tiles_off + ty * ((p.aw + COL_TW - 1) / COL_TW) + tx] != 0;
}

template <bool TABLE>
__global__ __launch_bounds__(256) void blur_rows_kernel(
    const pano_patch *__restrict__ table, pano_patch single, int nch, int alpha_ch,
    Levels L, const int16_t *__restrict__ owner, int W, const uint8_t *__restrict__ flags) {
    __shared__ __attribute__((aligned(16))) float s_row[4][ROW_LDS];
    const int pid = blockIdx.z / nch, c = blockIdx.z - pid * nch;
    const pano_patch p = TABLE ? table[pid] : single;
    const int xt = blockIdx.x * ROW_TW;          // first output column, A-relative
    if (xt >= p.aw || (int)blockIdx.y * 4 >= p.vh) return;   // uniform per block
    if (flags && p.h > 2 * L.rmax + 2) {
        // these 4 rows x 512 columns are needed iff an active column tile over the
        // same columns reads them: its own 128 rows grown by the radius (the
        // REFLECT_101 images of rows beyond the patch fall inside that range)
        const int ntx = (p.aw + COL_TW - 1) / COL_TW, nty = (p.ah + COL_TH - 1) / COL_TH;
        const int ry0 = p.vy0 + (int)blockIdx.y * 4 - p.ay0;      // A-relative first row
        const int t = threadIdx.y * 64 + threadIdx.x;             // 24 candidates: 8 x 3
        bool need = false;
        if (t < 24) {
            const int tx = xt / COL_TW + (t & 7);
            int t0 = ry0 / COL_TH;
            t0 = t0 < 0 ? 0 : (t0 >= nty ? nty - 1 : t0);
            const int ty = t0 - 1 + (t >> 3);
            if (tx < ntx && ty >= 0 && ty < nty && ty * COL_TH - L.rmax <= ry0 + 3 &&
                ty * COL_TH + COL_TH + L.rmax > ry0)
                need = flags[p.tiles_off + ty * ntx + tx] != 0;
        }
        if (!__syncthreads_or(need)) return;
    }

    const int lane = threadIdx.x, wv = threadIdx.y;
    const int y = blockIdx.y * 4 + wv;
    const int yc = y < p.vh ? y : p.vh - 1;
    float *tile = s_row[wv];
    const float *__restrict__ src =
        c == alpha_ch ? nullptr : p.planes + (size_t)c * p.vh * p.vpitch;
    const int16_t *__restrict__ orow =
        owner + (size_t)(p.y0 + p.vy0 + yc) * W + p.x0;      // sharp-alpha source row

    // stage the row segment, four loads in flight per lane
    for (int i0 = lane; i0 < L.need; i0 += 256) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 64 * u < L.need ? i0 + 64 * u : L.need - 1;
            const int pcol = reflect_101(p.ax0 + xt - L.rmax + i, p.w);
            if (src) {
                int vc = pcol - p.vx0;           // inside V by construction; clamp anyway
                vc = vc < 0 ? 0 : (vc >= p.vw ? p.vw - 1 : vc);
                v[u] = src[(size_t)yc * p.vpitch + vc];
            } else {
                v[u] = orow[pcol] == p.index ? 1.0f : 0.0f;      // stitcher.py:208
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + 64 * u < L.need) tile[row_pos(i0 + 64 * u)] = v[u];
    }
    __syncthreads();

    const float *base = tile + 12 * lane;        // row_pos(8 * lane)
    const int x = xt + 8 * lane;
    for (int k = 0; k < L.n; ++k) {
        const kptr_f32 wz = (kptr_f32)(uintptr_t)L.lv[k].wz;
        const int start = L.lv[k].start, steps = L.lv[k].steps;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.0f;
        // 8 inputs (two 16-B LDS reads) per trip against 15 consecutive taps
        for (int s = 0; s < steps; s += 2) {
            const int q0 = start + 4 * s, q1 = q0 + 4;
            const float4 va = *(const float4 *)(base + q0 + ((q0 >> 3) << 2));
            const float4 vb = *(const float4 *)(base + q1 + ((q1 >> 3) << 2));
            float wq[15];
#pragma unroll
            for (int j = 0; j < 15; ++j) wq[j] = wz[4 * s + j];
            const float in[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    acc[o] = __builtin_fmaf(wq[i + 7 - o], in[i], acc[o]);
            }
        }
        if (y < p.vh) {
            float *d = p.scratch + ((size_t)(k * nch + c) * p.vh + y) * p.apitch + x;
            if (x + 8 <= p.apitch) {
                *(float4 *)d = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *(float4 *)(d + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            } else {
#pragma unroll
                for (int o = 0; o < 8; ++o)
                    if (x + o < p.aw) d[o] = acc[o];
            }
        }
    }
}

// One trip of the column pass: 8 tile rows and the 15 taps they meet.
struct ColTrip {
    float w[15];
    float4 v[8];
};

__device__ __forceinline__ void col_trip_load(ColTrip &t, const float *base, kptr_f32 wz, int q) {
#pragma unroll
    for (int j = 0; j < 15; ++j) t.w[j] = wz[q + j];
#pragma unroll
    for (int s = 0; s < 8; ++s) t.v[s] = *(const float4 *)(base + (q + s) * COL_TW);
}

__device__ __forceinline__ void col_trip_fma(const ColTrip &t, float4 (&acc)[8]) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const float wgt = t.w[s + 7 - o];
            acc[o].x = __builtin_fmaf(wgt, t.v[s].x, acc[o].x);
            acc[o].y = __builtin_fmaf(wgt, t.v[s].y, acc[o].y);
            acc[o].z = __builtin_fmaf(wgt, t.v[s].z, acc[o].z);
            acc[o].w = __builtin_fmaf(wgt, t.v[s].w, acc[o].w);
        }
    }
}

template <bool TABLE>
__global__ __launch_bounds__(256) void blur_cols_kernel(
    const pano_patch *__restrict__ table, pano_patch single, int nch, int level,
    const float *wz_global, int ntaps, const uint8_t *__restrict__ flags) {
    const kptr_f32 wz = (kptr_f32)(uintptr_t)wz_global;
    extern __shared__ __attribute__((aligned(16))) float s_col[];   // [rows][64]
    const int pid = blockIdx.z / nch, c = blockIdx.z - pid * nch;
    const pano_patch p = TABLE ? table[pid] : single;
    const int x0 = blockIdx.x * COL_TW, y0 = blockIdx.y * COL_TH;   // A-relative
    if (x0 >= p.aw || y0 >= p.ah) return;                           // uniform per block
    if (p.h > 2 * (ntaps >> 1) + 2 && !col_tile_active(flags, p, blockIdx.x, blockIdx.y)) return;

    const float *__restrict__ src = p.scratch + (size_t)(level * nch + c) * p.vh * p.apitch;
    float *__restrict__ dst = p.blurred + (size_t)(level * nch + c) * p.ah * p.apitch;
    const int tx = threadIdx.x, ty = threadIdx.y;           // 16 x 16
    const int R = ntaps >> 1;
    const int steps = COL_STEPS(ntaps);                     // input rows per thread, x8
    const int rows = 8 * 15 + steps;                        // rows read by ty = 15
    const int x = x0 + 4 * tx;

    // Stage the tile with direct-to-LDS loads (global_load_lds_dwordx4): a wave
    // owns 4 consecutive tile rows per trip - 64 lanes x 16 B = 1 KiB contiguous
    // in LDS, which is what the instruction's "wave base + lane * 16" addressing
    // needs - while every lane fetches from its own (reflected) source row.  No
    // VGPRs are held, so the whole tile (14-16 KiB per wave) is in flight at
    // once; the tile load was latency-bound when it went through registers
    // (profiles/r01/notes.md).  Rows past the aperture only meet zero taps but
    // must hold finite numbers: they are staged like any other row.  Columns
    // past the pitch re-read the last valid 16 bytes (never stored).
    const int xl = x < p.apitch ? x : p.apitch - 4;
    const int wave4 = (ty >> 2) << 2;                       // first row of this wave's group
    for (int r0 = 0; r0 < rows; r0 += 16) {
        const int r = r0 + ty;
        int vr = reflect_101(p.ay0 + y0 - R + (r < rows ? r : rows - 1), p.h) - p.vy0;
        vr = vr < 0 ? 0 : (vr >= p.vh ? p.vh - 1 : vr);     // inside by construction
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(src + (size_t)vr * p.apitch + xl),
            (__attribute__((address_space(3))) void *)(s_col + (r0 + wave4) * COL_TW), 16, 0, 0);
    }
    __syncthreads();

    float4 acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *base = s_col + (8 * ty) * COL_TW + 4 * tx;
    // 8 input rows per trip: 15 consecutive taps serve all 8 x 8 (row, output)
    // pairs; a trip is 8 LDS reads + 15 scalar loads feeding 256 FMAs.
    // (issuing trip t + 1's reads before trip t's FMAs was measured slower: 1.92 against 1.78 ms)
    for (int q = 0; q < steps; q += 8) {
        ColTrip a;
        col_trip_load(a, base, wz, q);
        col_trip_fma(a, acc);
    }
    if (x < p.apitch) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int y = y0 + 8 * ty + o;
            if (y < p.ah) *(float4 *)(dst + (size_t)y * p.apitch + x) = acc[o];
        }
    }
}

// One wave per column tile of every record: active = some interior-map block under the
// tile is not interior.
__global__ __launch_bounds__(64) void tile_flags_kernel(const pano_patch *__restrict__ table,
                                                        const uint8_t *__restrict__ interior,
                                                        int W8, uint8_t *__restrict__ flags) {
    const pano_patch p = table[blockIdx.z];
    const int ntx = (p.aw + COL_TW - 1) / COL_TW, nty = (p.ah + COL_TH - 1) / COL_TH;
    const int tx = blockIdx.x, ty = blockIdx.y;
    if (tx >= ntx || ty >= nty) return;
    const int gx0 = p.x0 + p.ax0 + tx * COL_TW, gy0 = p.y0 + p.ay0 + ty * COL_TH;
    int gx1 = gx0 + COL_TW, gy1 = gy0 + COL_TH;
    const int ax1 = p.x0 + p.ax0 + p.aw, ay1 = p.y0 + p.ay0 + p.ah;
    gx1 = gx1 < ax1 ? gx1 : ax1;
    gy1 = gy1 < ay1 ? gy1 : ay1;
    const int bx0 = gx0 / PANO_INTERIOR_BLOCK, bx1 = (gx1 - 1) / PANO_INTERIOR_BLOCK;
    const int by0 = gy0 / PANO_INTERIOR_BLOCK, by1 = (gy1 - 1) / PANO_INTERIOR_BLOCK;
    const int nbx = bx1 - bx0 + 1, total = nbx * (by1 - by0 + 1);
    bool active = false;
    for (int i = threadIdx.x; i < total; i += 64)
        active |= interior[(size_t)(by0 + i / nbx) * W8 + bx0 + i % nbx] == 0;
    if (__ballot(active) && threadIdx.x == 0) flags[p.tiles_off + ty * ntx + tx] = 1;
    if (!__ballot(active) && threadIdx.x == 0) flags[p.tiles_off + ty * ntx + tx] = 0;
}

static int check_taps(int ntaps, const char *who) {
    PANO_REQUIRE(ntaps >= 1 && (ntaps & 1) && ntaps <= PANO_MAX_TAPS,
                 "%s: aperture %d must be odd and within [1, %d]", who, ntaps, PANO_MAX_TAPS);
    return PANO_OK;
}

// Level descriptors from the caller's tap tables (layout: include/pano360.h).
static int make_levels(const float *taps, const int *ntaps, int n_blur, Levels *L,
                       const float **col_wz, const char *who) {
    PANO_REQUIRE(n_blur >= 1 && n_blur < PANO_MAX_LEVELS, "%s: %d blur levels", who, n_blur);
    L->n = n_blur;
    L->rmax = 0;
    for (int k = 0; k < n_blur; ++k) {
        if (int rc = check_taps(ntaps[k], who)) return rc;
        if (ntaps[k] / 2 > L->rmax) L->rmax = ntaps[k] / 2;
    }
    L->need = 0;
    size_t off = 0;
    for (int k = 0; k < n_blur; ++k) {
        const int delta = L->rmax - ntaps[k] / 2, extra = delta & 3;
        L->lv[k].wz = taps + off;
        L->lv[k].ntaps = ntaps[k];
        L->lv[k].start = delta - extra;
        L->lv[k].steps = (((ntaps[k] + extra + 7 + 3) >> 2) + 1) & ~1;   // even: 2 per trip
        col_wz[k] = taps + off + extra;          // 7 leading zeros for the column pass
        const int need = 8 * 63 + L->lv[k].start + 4 * L->lv[k].steps;
        if (need > L->need) L->need = need;
        off += (size_t)ntaps[k] + PANO_TAP_PAD;
    }
    PANO_REQUIRE(L->need <= ROW_TILE, "%s: row tile %d exceeds %d", who, L->need, ROW_TILE);
    return PANO_OK;
}

static int launch_blur(pano_ctx *ctx, const pano_patch *table, const pano_patch &single, int n,
                       int nch, int alpha_ch, int max_aw, int max_vh, int max_ah,
                       const int16_t *owner, int W, const float *host_taps, const int *ntaps,
                       int n_blur, const uint8_t *interior, uint8_t *tile_flags, const char *who) {
    const hipStream_t stream = ctx->stream;
    PanoTapSet *set = nullptr;
    bool fresh = false;
    if (int rc = pano_ctx_tap_set(ctx, host_taps, ntaps, n_blur, 0, &set, &fresh)) return rc;
    Levels L = {};
    const float *col_wz[PANO_MAX_LEVELS];
    if (int rc = make_levels(set->taps, ntaps, n_blur, &L, col_wz, who)) return rc;
    const uint8_t *flags = nullptr;
    if (interior && table) {
        dim3 grid(ceil_div(max_aw, COL_TW), ceil_div(max_ah, COL_TH), n);
        PANO_TIMED(PK_TILE_FLAGS, stream,
                   hipLaunchKernelGGL(tile_flags_kernel, grid, dim3(64), 0, stream, table, interior,
                                      ceil_div(W, PANO_INTERIOR_BLOCK), tile_flags));
        PANO_LAUNCH_CHECK("tile_flags_kernel");
        flags = tile_flags;
    }
    {
        dim3 block(64, 4), grid(ceil_div(max_aw, ROW_TW), ceil_div(max_vh, 4), n * nch);
        if (table)
            PANO_TIMED(PK_BLUR_ROWS, stream,
                       hipLaunchKernelGGL(blur_rows_kernel<true>, grid, block, 0, stream, table,
                                          single, nch, alpha_ch, L, owner, W, flags));
        else
            PANO_TIMED(PK_BLUR_ROWS, stream,
                       hipLaunchKernelGGL(blur_rows_kernel<false>, grid, block, 0, stream, table,
                                          single, nch, alpha_ch, L, owner, W, flags));
        PANO_LAUNCH_CHECK("blur_rows_kernel");
    }
    for (int k = 0; k < n_blur; ++k) {
        const int rows = (8 * 15 + COL_STEPS(ntaps[k]) + 15) & ~15;   // staged 16 at a time
        const size_t lds = (size_t)rows * COL_TW * sizeof(float);
        dim3 block(16, 16), grid(ceil_div(max_aw, COL_TW), ceil_div(max_ah, COL_TH), n * nch);
#define LAUNCH_COLS(T)                                                                 \
    PANO_TIMED(PK_BLUR_COLS, stream,                                                   \
               hipLaunchKernelGGL(blur_cols_kernel<T>, grid, block, lds, stream, table, \
                                  single, nch, k, col_wz[k], ntaps[k], flags))
        if (table) LAUNCH_COLS(true); else LAUNCH_COLS(false);
#undef LAUNCH_COLS
        PANO_LAUNCH_CHECK("blur_cols_kernel");
    }
    return PANO_OK;
}

// column tiles above 64 KiB need the opt-in (once per device; pano_ctx_create)
int pano_blur_valu_opt_in(void) {
    const void *fns[] = {(const void *)blur_cols_kernel<true>,
                         (const void *)blur_cols_kernel<false>};
    for (const void *fn : fns)
        PANO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return PANO_OK;
}

extern "C" int pano_blur_plane(pano_ctx *ctx, const float *src, float *dst, float *tmp, int h,
                               int w, int pitch, const float *taps, int ntaps) {
    PANO_ENTER(ctx, "pano_blur_plane");
    PANO_REQUIRE(src && dst && tmp && taps, "pano_blur_plane: null pointer");
    PANO_REQUIRE(h > 0 && w > 0 && pitch >= w && (pitch & 3) == 0,
                 "pano_blur_plane: bad shape %dx%d pitch %d", h, w, pitch);
    if (int rc = check_taps(ntaps, "pano_blur_plane")) return rc;
    pano_patch p = {};
    p.planes = const_cast<float *>(src);
    p.scratch = tmp;
    p.blurred = dst;
    p.h = p.vh = p.ah = h;
    p.w = p.vw = p.aw = w;
    p.vpitch = p.apitch = pitch;
    return launch_blur(ctx, nullptr, p, 1, 1, -1, w, h, h, nullptr, 0, taps, &ntaps, 1, nullptr,
                       nullptr, "pano_blur_plane");
}

// The context's PANO_OPT_BLUR_KERNEL selects the vector-ALU kernels above or the
// matrix-core kernel of blur_mfma.hip (default; cfg3: 0.95 ms against 2.4 ms for the row +
// column passes) for the multiband levels; single planes always take the kernels above.
extern "C" int pano_blur_tile_grid(const pano_ctx *ctx) {
    return ctx && !pano_blur_uses_mfma(ctx) ? 0 : 32;
}

extern "C" int pano_multiband_blur(pano_ctx *ctx, const pano_patch *patches, int n, int max_aw,
                                   int max_vh, int max_ah, const int16_t *owner, int W,
                                   const float *taps, const int *ntaps, int n_blur,
                                   const uint8_t *interior, uint8_t *tile_flags) {
    PANO_ENTER(ctx, "pano_multiband_blur");
    PANO_REQUIRE(patches && owner && taps && ntaps, "pano_multiband_blur: null pointer");
    PANO_REQUIRE(n >= 0 && n <= 32767 && W > 0, "pano_multiband_blur: bad argument");
    PANO_REQUIRE(max_aw >= 0 && max_vh >= 0 && max_ah >= 0, "pano_multiband_blur: bad extents");
    PANO_REQUIRE(!interior || tile_flags, "pano_multiband_blur: interior map without tile_flags");
    if (n == 0 || n_blur == 0 || max_aw == 0 || max_vh == 0 || max_ah == 0) return PANO_OK;
    PANO_REQUIRE(n_blur >= 1 && n_blur < PANO_MAX_LEVELS, "pano_multiband_blur: %d blur levels", n_blur);
    for (int k = 0; k < n_blur; ++k)
        if (int rc = check_taps(ntaps[k], "pano_multiband_blur")) return rc;
    if (pano_blur_uses_mfma(ctx))
        return pano_launch_blur_mfma(ctx, patches, n, max_aw, max_ah, owner, W, taps, ntaps, n_blur,
                                     interior, tile_flags);
    pano_patch none = {};
    return launch_blur(ctx, patches, none, n, 4, 3, max_aw, max_vh, max_ah, owner, W, taps, ntaps,
                       n_blur, interior, tile_flags, "pano_multiband_blur");
}

extern "C" int pano_blur_tiles(pano_ctx *ctx, const pano_patch *patches, int n, int max_aw,
                               int max_ah, int W, int radius, const uint8_t *interior,
                               uint8_t *tile_flags, uint8_t *warp_need) {
    PANO_ENTER(ctx, "pano_blur_tiles");
    PANO_REQUIRE(patches && interior && tile_flags, "pano_blur_tiles: null pointer");
    PANO_REQUIRE(n >= 0 && n <= 32767 && W > 0 && max_aw >= 0 && max_ah >= 0 && radius >= 0,
                 "pano_blur_tiles: bad argument");
    PANO_REQUIRE(pano_blur_uses_mfma(ctx), "pano_blur_tiles: needs the 32 x 32 tile grid");
    if (n == 0 || max_aw == 0 || max_ah == 0) return PANO_OK;
    return pano_tiles_blur_mfma(ctx, patches, n, max_aw, max_ah, W, radius, interior, tile_flags,
                                warp_need);
}

extern "C" int pano_multiband_blur_prepare(pano_ctx *ctx, const pano_patch *patches, int n,
                                           int max_aw, int max_ah, int W,
                                           const uint8_t *interior, uint8_t *tile_flags) {
    PANO_ENTER(ctx, "pano_multiband_blur_prepare");
    PANO_REQUIRE(patches, "pano_multiband_blur_prepare: null pointer");
    PANO_REQUIRE(n >= 0 && n <= 32767 && W > 0 && max_aw >= 0 && max_ah >= 0,
                 "pano_multiband_blur_prepare: bad argument");
    PANO_REQUIRE(!interior || tile_flags,
                 "pano_multiband_blur_prepare: interior map without tile_flags");
    if (n == 0 || max_aw == 0 || max_ah == 0 || !pano_blur_uses_mfma(ctx)) return PANO_OK;
    return pano_prepare_blur_mfma(ctx, patches, n, max_aw, max_ah, W, interior, tile_flags);
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

extern "C" int pano_pyr_down(pano_ctx *ctx, const float *src, int h, int w, float *dst) {
    PANO_ENTER(ctx, "pano_pyr_down");
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
