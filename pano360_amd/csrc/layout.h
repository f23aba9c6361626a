// One record of the window layout (include/pano360.h, "Windows"): rectangles A and V of a
// (camera, span of owned columns) and the sizes its planes take in the arenas.  Shared by the
// host routine pano_layout_windows and by the device kernel that lays the table out without a
// host round trip (stitch.hip): the same integer arithmetic on both sides.
#pragma once
#include "common.h"

__host__ __device__ static inline void reflect_closed(long lo, long hi, long n, long &a, long &b) {
    // smallest [a, b) inside [0, n) holding reflect_101(p, n) for every p in [lo, hi)
    if (n == 1) {
        a = 0;
        b = 1;
        return;
    }
    if (lo < -(n - 1) || hi > 2 * n - 1) {      // a second reflection could occur
        a = 0;
        b = n;
        return;
    }
    a = lo > 0 ? lo : 0;
    b = hi < n ? hi : n;
    if (lo < 0) {                               // p in [lo, 0) lands on [1, -lo]
        a = a < 1 ? a : 1;
        b = b > 1 - lo ? b : 1 - lo;
    }
    if (hi > n) {                               // p in [n, hi) lands on [2n-1-hi, n-2]
        a = a < 2 * n - 1 - hi ? a : 2 * n - 1 - hi;
        b = b > n - 1 ? b : n - 1;
    }
    a = a > 0 ? a : 0;
    b = b < n ? b : n;
}

struct LayoutSizes {
    long planes, blurred, scratch, tiles, lead;    // floats (tiles: count); lead = floats in front of `blurred`
};

// Record of span `sp` of camera i (regions row `rg` = {ymin, ymax, xmin, xmax, count, xa_0,
// xb_0, ...}, patch rectangle `rect` = {y0, y1, x0, x1}), cut to the mosaic columns [xs0, xs1).
// false: the span owns nothing there.  The record's pointers and tiles_off are left 0.
__host__ __device__ static inline bool layout_record(const int32_t *rg, int sp, const int32_t *rect,
                                                     int i, int radius, int xs0, int xs1,
                                                     int n_blur, bool grid32, pano_patch &r,
                                                     LayoutSizes &sz) {
    const long ymin = rg[0], ymax = rg[1];
    const long y0 = rect[0], y1 = rect[1], x0 = rect[2], x1 = rect[3];
    const long h = y1 - y0, w = x1 - x0;
    const long xmin = rg[5 + 2 * sp], xmax = rg[6 + 2 * sp];
    if (ymax < ymin || xmax < xmin) return false;
    long ay0 = ymin - y0 - radius, ay1 = ymax - y0 + 1 + radius;
    long ax0 = xmin - x0 - radius, ax1 = xmax - x0 + 1 + radius;
    ay0 = ay0 > 0 ? ay0 : 0;
    ay1 = ay1 < h ? ay1 : h;
    ax0 = ax0 > 0 ? ax0 : 0;
    ax1 = ax1 < w ? ax1 : w;
    ax0 = ax0 > xs0 - x0 ? ax0 : xs0 - x0;       // one GPU's share of the mosaic
    ax1 = ax1 < xs1 - x0 ? ax1 : xs1 - x0;
    if (ax1 <= ax0) return false;
    long vy0, vy1, vx0, vx1;
    reflect_closed(ay0 - radius, ay1 + radius, h, vy0, vy1);
    reflect_closed(ax0 - radius, ax1 + radius, w, vx0, vx1);
    vy0 = vy0 < ay0 ? vy0 : ay0;
    vy1 = vy1 > ay1 ? vy1 : ay1;
    vx0 = vx0 < ax0 ? vx0 : ax0;
    vx1 = vx1 > ax1 ? vx1 : ax1;
    // both ends on multiples of 4 patch columns (the far one clipped to the patch): the
    // blur stages its bands in aligned chunks of 4 columns, and a chunk is then inside V
    // or outside it as a whole
    vx0 &= ~3l;
    vx1 = (vx1 + 3) & ~3l;
    vx1 = vx1 < w ? vx1 : w;
    r = pano_patch{};
    r.y0 = (int)y0, r.x0 = (int)x0, r.h = (int)h, r.w = (int)w;
    r.index = i;
    r.vy0 = (int)vy0, r.vx0 = (int)vx0, r.vh = (int)(vy1 - vy0), r.vw = (int)(vx1 - vx0);
    r.ay0 = (int)ay0, r.ax0 = (int)ax0, r.ah = (int)(ay1 - ay0), r.aw = (int)(ax1 - ax0);
    r.vpitch = (r.vw + 3) & ~3;
    sz = LayoutSizes{};
    if (grid32) {
        // 32-column tile rows anchored at multiples of 32 in patch coordinates: 128-byte
        // rows with the anchor column on a 128-byte boundary are one cache line each
        r.apitch = (r.aw + 31) & ~31;
        sz.lead = r.ax0 & 31;
        sz.blurred = (long)n_blur * 4 * r.ah * r.apitch + 32;
        sz.tiles = (long)(((r.ax0 + r.aw - 1) >> 5) - (r.ax0 >> 5) + 1) *
                   (((r.ay0 + r.ah - 1) >> 5) - (r.ay0 >> 5) + 1);
    } else {
        r.apitch = (r.aw + 3) & ~3;
        sz.blurred = (long)n_blur * 4 * r.ah * r.apitch;
        sz.scratch = (long)n_blur * 4 * r.vh * r.apitch;
        sz.tiles = (long)((r.aw + 63) / 64) * ((r.ah + 127) / 128);
    }
    sz.planes = 3l * r.vh * r.vpitch;
    return true;
}

// the blur addresses a plane through a buffer descriptor with 32-bit byte offsets
__host__ __device__ static inline bool layout_record_fits(const pano_patch &r) {
    return (long)r.vh * r.vpitch * 4 < (1l << 31) && (long)r.ah * r.apitch * 4 < (1l << 31);
}

// What the device-side layout (layout_windows_kernel) leaves for the host to check.
struct LayoutSummary {
    int64_t planes_floats, blurred_floats, scratch_floats;
    int32_t n_records, n_tiles;
    int32_t max_vw, max_vh, max_aw, max_ah;
    int32_t missing;
    int32_t ok;                // 1: the table holds the records; 0: it was emptied (see `why`)
    int32_t why;               // bit 0 arenas / tile flags too small, 1 launch bounds exceeded,
                               // 2 more records than slots, 3 a plane beyond 2 GiB, 4 frames missing
    int32_t pad;
};
