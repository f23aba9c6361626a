// Mosaic-side kernels: ownership/validity, and the three blenders, all written
// gather-style - one thread per mosaic pixel walks the patch table in index
// order - so there are no atomics, no H x W x N weight stack (the reference's
// stitcher.py:196 allocation) and the float sums run in the reference's patch
// order.
//
// Reference arithmetic replaced: stitcher.py:160-168 (no_blend), :171-183
// (linear_blend), :196-241 (multiband_blend minus the GaussianBlur calls),
// :266-271 (_valid).
//
// Roofline: HBM.  Per gathered (pixel, patch) pair the multiband collapse reads
// 3 + 4(L-1) floats (the warped colour and the L-1 blurred RGBA copies),
// coalesced along x because every plane is planar; it writes 3 B (+12 B when
// the float mosaic is requested) per mosaic pixel.  The camera-driven ownership
// kernel reads no pixel data at all: 3 B written per mosaic pixel, the rest is
// arithmetic (about 120 instructions per covering camera).
#include <stdlib.h>

#include "geom.h"

// ---- ownership from warped alpha planes (stage-level API) ---------------------
__global__ __launch_bounds__(256) void ownership_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    int16_t *__restrict__ owner, uint8_t *__restrict__ valid) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float best = 0.0f;
    int who = -1;
    bool any = false;
    for (int i = 0; i < n; ++i) {
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        const float a = p.planes[3 * (size_t)p.vh * p.vpitch + (size_t)py * p.vpitch + px];
        if (a > best) {          // strict: the first maximum keeps the pixel
            best = a;
            who = i;
        }
        any |= p.mask[(size_t)py * p.w + px] == 0;
    }
    owner[(size_t)y * W + x] = (int16_t)who;
    valid[(size_t)y * W + x] = any ? 1 : 0;
}

// ---- ownership straight from the cameras (fused path) ---------------------------
// alpha_i(pixel) is a closed form of camera i alone: the inverse map, the
// bounds mask and the bilinear sample of hat_y (x) hat_x, all evaluated with
// the operations pano_warp_spherical uses, so the owner map is identical to
// the one computed from warped alpha planes.
//
// About 110 vector instructions per (pixel, covering camera) - but the walk over
// the camera table is what a naive loop pays for: ~20 scalar instructions and a
// dependent scalar load per camera and wave, for all n cameras, when only the
// ~10 whose patch rectangle meets this 64 x 4 block matter.  The block therefore
// first builds that short list (ordered, so the first-maximum rule survives) in
// LDS with a ballot compaction, and every wave walks only the list.
#define OWN_LIST 256

struct CamList {
    int list[OWN_LIST];
    int wave[4];
    int count;
};

// Ordered list of the cameras whose patch rectangle meets the block's pixels
// [bx0, bx1) x [by0, by1); returns its length, or -1 when it does not fit (the
// caller then walks all n cameras).  Every thread of the 256-thread block calls it.
__device__ __forceinline__ int build_camera_list(CamList &sh, const pano_camera *__restrict__ cams,
                                                 int n, int bx0, int bx1, int by0, int by1) {
    const int tid = threadIdx.y * 64 + threadIdx.x, lane = threadIdx.x, wave = threadIdx.y;
    if (tid == 0) sh.count = 0;
    __syncthreads();
    bool overflow = false;
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        bool hit = false;
        if (i < n) {
            const pano_camera *cam = cams + i;
            hit = cam->x0 < bx1 && cam->x0 + cam->w > bx0 && cam->y0 < by1 && cam->y0 + cam->h > by0;
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) sh.wave[wave] = __popcll(bal);
        __syncthreads();
        int off = sh.count;
        for (int w = 0; w < wave; ++w) off += sh.wave[w];
        const int total = sh.count + sh.wave[0] + sh.wave[1] + sh.wave[2] + sh.wave[3];
        off += __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && off < OWN_LIST) sh.list[off] = i;
        __syncthreads();
        if (tid == 0) sh.count = total;
        if (total > OWN_LIST) overflow = true;                       // uniform across the block
        __syncthreads();
    }
    return overflow ? -1 : sh.count;
}

// Pruning (exact).  Before any pixel is evaluated, one thread per listed camera bounds
// that camera's alpha over the whole 64 x 16 tile: interval arithmetic, in double, on
// the ray components (the ranges of sin / cos over the tile's columns and of tan over
// its rows), through K R and the perspective divide, gives a box of source coordinates
// that holds every pixel's sample position; the box is grown by 2 px - far more than
// the float32 rounding of the exact path, its 1/32 px fixed-point coordinates and the
// one-sample reach of the bilinear taps - and hat_y (x) hat_x, concave and piecewise
// linear, is bounded on it from above (its value nearest the frame centre; bilinear
// interpolation of a concave function never exceeds it) and from below (its smaller
// end value, when the box lies inside the frame and in front of the camera, so that
// no pixel is masked; else 0).  A camera whose upper bound is below the largest lower
// bound L loses every pixel of the tile to the camera that attains L, strictly, so it
// can be skipped: neither the argmax nor - since that winner is unmasked everywhere
// when L > 0 - the valid flag depends on it.  With L = 0 only cameras that are masked
// on the whole tile go.  Far from seams one camera survives, near a seam two; the
// pixels then run the exact evaluation on the survivors only, in index order.
struct AlphaBound {
    float lo, hi;      // hi < 0: masked on the whole tile
};

__device__ __forceinline__ void hat_range(double a, double b, int n, double &lo, double &hi) {
    // hat(x) = 0.5 - |x - n/2| / n on [a, b]
    const double mid = 0.5 * n, peak = fmin(fmax(mid, a), b);
    hi = 0.5 - fabs(peak - mid) / n;
    lo = fmin(0.5 - fabs(a - mid) / n, 0.5 - fabs(b - mid) / n);
}

__device__ __forceinline__ void iv_axpy(double k, double lo, double hi, double &alo, double &ahi) {
    alo += k >= 0.0 ? k * lo : k * hi;
    ahi += k >= 0.0 ? k * hi : k * lo;
}

__device__ __forceinline__ AlphaBound alpha_bound_of(const pano_camera *cam, const double (*v)[2]);

__device__ __forceinline__ AlphaBound alpha_bound(const pano_camera *cam, const double *r) {
    // r = {s_lo, s_hi, t_lo, t_hi, c_lo, c_hi}
    const double *K = cam->proj;
    double v[3][2];
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        double lo = 0.0, hi = 0.0;
        iv_axpy(K[3 * row + 0], r[0], r[1], lo, hi);
        iv_axpy(K[3 * row + 1], r[2], r[3], lo, hi);
        iv_axpy(K[3 * row + 2], r[4], r[5], lo, hi);
        const double slack = 1e-9 * (fabs(lo) + fabs(hi)) + 1e-300;
        v[row][0] = lo - slack;
        v[row][1] = hi + slack;
    }
    return alpha_bound_of(cam, v);
}

// The same bound from the tile's FIRST and LAST column (sa, ca), (sb, cb) instead of the ranges of
// sin and cos.  A row of K R ray is g(theta) + K1 t with g = K0 sin theta + K2 cos theta; intervals
// of sin and cos taken as independent overestimate g's range badly where it matters - the depth
// v_z = cos(theta - theta_0) moves by sin(theta - theta_0) d_theta over a tile, the independent
// intervals say sin(theta + theta_0) d_theta, and through the divide every per cent of v_z is 15
// source pixels at the frame's edge, as much as a 16-pixel quarter is wide.  g is a sinusoid of
// amplitude A <= |K0| + |K2| and theta runs monotonically over the tile's columns, so g stays
// within A d_theta^2 / 8 of the chord between its end values: [min(ga, gb) - e, max(ga, gb) + e].
// d_theta follows from the chord of the unit circle between the two columns, 2 sin(d_theta / 2):
// for chords up to 0.25 (the caller's test; else the ranges above) d_theta <= 1.003 chord.
// t is independent of theta, so adding K1 [t_lo, t_hi] is exact.  Same slack, same tail.
__device__ __forceinline__ AlphaBound alpha_bound_ends(const pano_camera *cam, double sa, double ca,
                                                       double sb, double cb, double t_lo,
                                                       double t_hi) {
    const double *K = cam->proj;
    const double chord2 = (sb - sa) * (sb - sa) + (cb - ca) * (cb - ca);
    double v[3][2];
#pragma unroll
    for (int row = 0; row < 3; ++row) {
        const double a = K[3 * row + 0], b = K[3 * row + 2];
        const double ga = a * sa + b * ca, gb = a * sb + b * cb;
        const double e = (fabs(a) + fabs(b)) * chord2 * 0.1258;
        double lo = fmin(ga, gb) - e, hi = fmax(ga, gb) + e;
        iv_axpy(K[3 * row + 1], t_lo, t_hi, lo, hi);
        const double slack = 1e-9 * (fabs(lo) + fabs(hi) + fabs(ga) + fabs(gb)) + 1e-300;
        v[row][0] = lo - slack;
        v[row][1] = hi + slack;
    }
    return alpha_bound_of(cam, v);
}

__device__ __forceinline__ AlphaBound alpha_bound_of(const pano_camera *cam, const double (*v)[2]) {
    AlphaBound out;
    if (v[2][1] <= 0.0) {                    // behind the camera everywhere: mask (stitcher.py:308)
        out.lo = 0.0f;
        out.hi = -1.0f;
        return out;
    }
    if (v[2][0] <= 1e-6 * (fabs(v[0][0]) + fabs(v[0][1]) + fabs(v[1][0]) + fabs(v[1][1]) + 1.0)) {
        out.lo = 0.0f;                       // the tile touches the camera's horizon: no bound
        out.hi = 1.0f;
        return out;
    }
    const int sw = cam->sw, sh = cam->sh;
    double box[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double q0 = v[a][0] / v[2][0], q1 = v[a][0] / v[2][1];
        const double q2 = v[a][1] / v[2][0], q3 = v[a][1] / v[2][1];
        const double half = 0.5 * (a == 0 ? sw : sh);
        box[a][0] = fmin(fmin(q0, q1), fmin(q2, q3)) + half - 2.0;
        box[a][1] = fmax(fmax(q0, q1), fmax(q2, q3)) + half + 2.0;
    }
    if (box[0][1] < 0.0 || box[0][0] > sw - 1.0 || box[1][1] < 0.0 || box[1][0] > sh - 1.0) {
        out.lo = 0.0f;                       // outside the frame everywhere (stitcher.py:311-312)
        out.hi = -1.0f;
        return out;
    }
    const bool inside = box[0][0] >= 0.0 && box[0][1] <= sw - 1.0 && box[1][0] >= 0.0 &&
                        box[1][1] <= sh - 1.0;
    double xlo, xhi, ylo, yhi;
    hat_range(fmax(box[0][0], 0.0), fmin(box[0][1], sw - 1.0), sw, xlo, xhi);
    hat_range(fmax(box[1][0], 0.0), fmin(box[1][1], sh - 1.0), sh, ylo, yhi);
    out.hi = (float)(fmax(xhi, 0.0) * fmax(yhi, 0.0) * (1.0 + 1e-6)) + 1e-7f;
    out.lo = inside ? fmaxf((float)(fmax(xlo, 0.0) * fmax(ylo, 0.0) * (1.0 - 1e-6)) - 1e-7f, 0.0f)
                    : 0.0f;
    return out;
}

#ifndef OWN_ROWS
#define OWN_ROWS 16
#endif
// A workgroup takes OWN_SUB sub-tiles of 64 x OWN_ROWS pixels, one above the other: the camera
// list is built once for all of them and wave w bounds the listed cameras on sub-tile w, a
// camera per lane, without a barrier of its own (the set-up of a 64 x 16 tile - list, ranges,
// bounds, five barriers and their dependent loads - was 0.10 of the kernel's 0.21 ms on
// config 3, profiles/r03/probes/ownership_noeval_ablation.patch; with one set-up per four
// sub-tiles the waves bound four sub-tiles in the time of one).
#define OWN_SUB 4
#define OWN_TILE_ROWS (OWN_ROWS * OWN_SUB)

// Grows a camera's box in memory.  Thousands of workgroups meet on the 16 bytes of a camera (a
// read-modify-write per value and workgroup made the kernel twice as slow: they queue up at one
// cache line), but nearly all of them bring nothing new: the box is read first - a stale value
// only errs towards a redundant atomic, the fields move one way - and only what grows it is sent.
__device__ __forceinline__ void box_merge(int32_t *g, int ymin, int ymax, int xmin, int xmax) {
    const int c0 = __atomic_load_n(&g[0], __ATOMIC_RELAXED), c1 = __atomic_load_n(&g[1], __ATOMIC_RELAXED);
    const int c2 = __atomic_load_n(&g[2], __ATOMIC_RELAXED), c3 = __atomic_load_n(&g[3], __ATOMIC_RELAXED);
    if (ymin < c0) atomicMin(&g[0], ymin);
    if (ymax > c1) atomicMax(&g[1], ymax);
    if (xmin < c2) atomicMin(&g[2], xmin);
    if (xmax > c3) atomicMax(&g[3], xmax);
}

__device__ __forceinline__ bool tile_inside(const pano_camera *cam, int x0, int x1, int y0, int y1) {
    return x0 >= cam->x0 && x1 <= cam->x0 + cam->w && y0 >= cam->y0 && y1 <= cam->y0 + cam->h;
}

__global__ __launch_bounds__(256) void ownership_cameras_l1_kernel(
    const pano_camera *__restrict__ cams, int n, int H, int W, int xs0, int xs1,
    const double *__restrict__ sin_t, const double *__restrict__ cos_t,
    const double *__restrict__ tan_p, int16_t *__restrict__ owner,
    uint8_t *__restrict__ valid, int prune, int32_t *__restrict__ boxes, int box_stride,
    uint8_t *__restrict__ marks) {
    __shared__ CamList sh;
    __shared__ int s_box[OWN_LIST][4];             // region search: {ymin, ymax, xmin, xmax} per listed camera
    __shared__ short s_pos[OWN_SUB][OWN_LIST];     // a survivor's position in the list
    __shared__ int s_keep[OWN_SUB][OWN_LIST];
    __shared__ float s_hi[4][OWN_LIST];            // [wave] upper bounds, lists of more than 64 cameras only
    __shared__ int s_ncand[OWN_SUB];
    __shared__ float s_low[OWN_SUB];
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int bx0 = xs0 + blockIdx.x * 64, by0 = blockIdx.y * OWN_TILE_ROWS;
    const int bx1 = min(bx0 + 64, xs1), by1 = min(by0 + OWN_TILE_ROWS, H);
    if (marks) {                                               // (the list's barriers publish it)
        int *b = s_box[wave * 64 + lane];
        b[0] = b[2] = 0x7fffffff;
        b[1] = b[3] = -1;
    }
    const int listed = build_camera_list(sh, cams, n, bx0, bx1, by0, by1);
    const bool pruned = listed > 1 && prune;

    if (pruned) {
        // A group of G = 16, 32 or 64 lanes takes one sub-tile, a listed camera per lane: with
        // the dozen cameras of a 5-degree sweep one wave bounds all four sub-tiles in a single
        // pass (the bound is ~250 double-precision instructions whatever the number of live lanes).
        const int G = listed <= 16 ? 16 : (listed <= 32 ? 32 : 64);          // block-uniform
        // (a workgroup's wave w sits on SIMD w: the one or two working waves rotate with the
        // workgroup so that the bounds do not all queue on SIMD 0)
        const int slot = (wave - (int)(blockIdx.x + blockIdx.y)) & 3;
        const int sub = slot * (64 / G) + lane / G, k0 = lane & (G - 1);
        const int sy0 = by0 + OWN_ROWS * sub, sy1 = min(sy0 + OWN_ROWS, H);
        const bool live = sub < OWN_SUB && sy0 < H;
        if (__ballot(live) != 0ull) {                                        // wave-uniform
            // ranges of the ray components over the sub-tile's columns and rows (every lane of
            // the group ends up with all six)
            const int xc = min(bx0 + lane, bx1 - 1);
            double lo_s = sin_t[xc], hi_s = lo_s, lo_c = cos_t[xc], hi_c = lo_c;
            const int yr = live ? min(sy0 + (lane & (OWN_ROWS - 1)), sy1 - 1) : by0;
            double lo_t = tan_p[yr], hi_t = lo_t;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo_s = fmin(lo_s, __shfl_xor(lo_s, off, 64));
                hi_s = fmax(hi_s, __shfl_xor(hi_s, off, 64));
                lo_c = fmin(lo_c, __shfl_xor(lo_c, off, 64));
                hi_c = fmax(hi_c, __shfl_xor(hi_c, off, 64));
            }
#pragma unroll
            for (int off = OWN_ROWS / 2; off > 0; off >>= 1) {
                lo_t = fmin(lo_t, __shfl_xor(lo_t, off, 64));
                hi_t = fmax(hi_t, __shfl_xor(hi_t, off, 64));
            }
            const double rng[6] = {lo_s, hi_s, lo_t, hi_t, lo_c, hi_c};
            float L = 0.0f, hi = -1.0f;
            for (int base = 0; base < listed; base += 64) {                  // one trip unless G = 64
                const int k = base + k0;
                AlphaBound bnd = {0.0f, -1.0f};
                if (live && k < listed) {
                    const pano_camera *cam = cams + sh.list[k];
                    bnd = alpha_bound(cam, rng);
                    // A lower bound beats other cameras on EVERY pixel of the tile only if this camera
                    // is a candidate on every pixel, i.e. the tile lies inside its patch rectangle
                    // (stitcher.py:289-297).  The rectangle of a frame across the +-pi seam stops short
                    // of the mosaic's ends (its range comes from border samples, :107-122), although
                    // the frame itself reaches them: found by the full-size config 5 test, where such
                    // a camera's bound pruned the only candidate of the last 32 columns.
                    if (!tile_inside(cam, bx0, bx1, sy0, sy1)) bnd.lo = 0.0f;
                    if (listed > 64) s_hi[wave][k] = bnd.hi;
                }
                hi = bnd.hi;
                L = fmaxf(L, bnd.lo);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1)
                if (off < G) L = fmaxf(L, __shfl_xor(L, off, 64));
            // the survivors, order preserved: the first maximum wins
            const int shift = (lane / G) * G;                                // G = 64: 0
            const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
            int nc = 0;
            for (int base = 0; base < listed; base += 64) {
                const int k = base + k0;
                if (listed > 64) hi = k < listed ? s_hi[wave][k] : -1.0f;    // this lane's own writes
                const bool keep = live && k < listed && hi >= L;
                const unsigned long long bal = (__ballot(keep) >> shift) & gmask;
                if (keep) {
                    const int at = nc + __popcll(bal & ((1ull << k0) - 1ull));
                    s_keep[sub][at] = sh.list[k];
                    s_pos[sub][at] = (short)k;
                }
                nc += __popcll(bal);
            }
            if (live && k0 == 0) {
                s_ncand[sub] = nc;
                s_low[sub] = L;
            }
        }
        __syncthreads();
    }

    const int x = bx0 + lane;
    const bool in_strip = x < xs1;
    const int xc = in_strip ? x : xs1 - 1;
    const double s = sin_t[xc], c = cos_t[xc];
    // Region search (pano_owned_regions' boxes and column marks) while the owners are in
    // registers: a thread follows the run of rows its column's current owner holds and hands a
    // finished run - at a change of owner, and once at the end - to the workgroup's box of that
    // camera in LDS (a wave-wide exchange per row cost as much as the separate kernel).
    int run_o = -1, run_pos = 0, run_y0 = 0, run_y1 = 0;
    auto flush = [&]() {
        if (run_o < 0 || !in_strip) return;
        marks[(size_t)run_o * W + x] = 1;
        if (listed >= 0) {
            int *b = s_box[run_pos];
            atomicMin(&b[0], run_y0);
            atomicMax(&b[1], run_y1);
            atomicMin(&b[2], x);
            atomicMax(&b[3], x);
        } else {                             // more cameras than the list holds: straight to memory
            box_merge(boxes + (size_t)box_stride * run_o, run_y0, run_y1, x, x);
        }
    };
#pragma unroll 1
    for (int sub = 0; sub < OWN_SUB; ++sub) {
        const int sy0 = by0 + OWN_ROWS * sub, sy1 = min(sy0 + OWN_ROWS, H);
        if (sy0 >= H) break;
        const int *list = sh.list;
        int ncand = listed < 0 ? n : listed;
        if (pruned) {
            list = s_keep[sub];
            ncand = s_ncand[sub];
            // One survivor with a positive lower bound: its alpha is positive on every pixel
            // of the tile and above every other camera's, so it owns the whole tile and no
            // pixel needs evaluating - provided the tile lies inside its patch rectangle
            // (outside it the camera is no candidate, stitcher.py:289-297).
            if (ncand == 1 && s_low[sub] > 0.0f) {
                const int i = __builtin_amdgcn_readfirstlane(list[0]);
                if (tile_inside(cams + i, bx0, bx1, sy0, sy1)) {
                    if (in_strip)
                        for (int y = sy0 + wave; y < sy1; y += 4) {
                            owner[(size_t)y * W + x] = (int16_t)i;
                            valid[(size_t)y * W + x] = 1;
                        }
                    if (marks && sy0 + wave < sy1) {
                        if (run_o != i) {
                            flush();
                            run_o = i;
                            run_pos = s_pos[sub][0];
                            run_y0 = sy0 + wave;
                        }
                        run_y1 = sy0 + wave + ((sy1 - 1 - sy0 - wave) & ~3);     // this wave's last row
                    }
                    continue;
                }
            }
        }
#pragma unroll 1
        for (int y = sy0 + wave; y < sy1; y += 4) {
            const double t = tan_p[y];
            float best = 0.0f;
            int who = -1, who_k = 0;
            bool any = false;
            for (int k = 0; k < ncand; ++k) {
                const int i = listed < 0 ? k : __builtin_amdgcn_readfirstlane(list[k]);
                const pano_camera *cam = cams + i;
                const int px = xc - cam->x0, py = y - cam->y0;
                if ((unsigned)px >= (unsigned)cam->w || (unsigned)py >= (unsigned)cam->h) continue;
                float fx, fy;
                const int sw = cam->sw, sh_ = cam->sh;
                if (map_pixel(cam->proj, s, c, t, sw, sh_, fx, fy)) continue;   // alpha * 0
                any = true;
                const Taps tp = make_taps_unmasked(fx, fy, sw, sh_);
                const float a = alpha_at(cam->hat_x, cam->hat_y, tp);
                if (a > best) {          // strict: the first maximum keeps the pixel
                    best = a;
                    who = i;
                    who_k = k;
                }
            }
            if (in_strip) {
                owner[(size_t)y * W + x] = (int16_t)who;
                valid[(size_t)y * W + x] = any ? 1 : 0;
            }
            if (marks) {
                if (who != run_o) {
                    flush();
                    run_o = who;
                    run_pos = pruned && who >= 0 ? s_pos[sub][who_k] : who_k;
                    run_y0 = y;
                }
                run_y1 = y;
            }
        }
    }
    if (!marks) return;
    flush();
    __syncthreads();
    // the workgroup's boxes into the cameras'
    const int tid = wave * 64 + lane;
    if (tid < listed) {
        const int *b = s_box[tid];
        if (b[1] >= b[0]) box_merge(boxes + (size_t)box_stride * sh.list[tid], b[0], b[1], b[2], b[3]);
    }
}

// ---- ownership with two levels of bounds (the default) ------------------------------------------
// The kernel above prunes per 64 x 16 sub-tile: a sub-tile within ~80 pixels of a seam keeps two
// cameras and evaluates both, exactly, at every one of its pixels - 40 % of config 3's sub-tiles -
// although the seam crosses a row once.  Here the survivors of a sub-tile are bounded again on its
// four 16 x 16 QUARTERS (the same rigorous interval bound, on a quarter's ranges of sin / cos /
// tan: nothing about the bound depends on the size of the tile), so only the quarters the seam
// really passes still evaluate two cameras; every other quarter is filled with its one survivor.
// Evaluation is the same exact code on the same operands, so owner and valid equal the exhaustive
// maps bit for bit (tests: test_ownership_bounds_against_exhaustive_evaluation,
// test_ownership_pruning_is_exact, test_cfg5_full_size_properties).
//
// A workgroup = one 64 x 64 tile, and its life is a CHAIN of latencies, not of instruction issue
// (profiles/r05/notes.md: phase timers): camera rectangles -> list -> the listed cameras' matrices ->
// bounds -> per pixel: matrix, divide, four table loads, compare.  So:
// * the cameras' records (all of them when n <= OW_CAMS, else the listed ones) and the tile's 64
//   sines / cosines / tangents are copied to LDS once - their loads leave together with the
//   rectangles' - and every later step reads them there;
// * a wave evaluates 16 x (4 OW_ILP) pixel groups of ONE quarter, OW_ILP independent pixels per
//   lane (a wave of 64 x 1 pixels would span four quarters and run every camera any of them kept);
// * the LDS a workgroup needs is sized by the host from n (17 KB for 32 cameras): seven workgroups
//   per CU where a static 30 KB allowed five;
// * results go to an LDS tile and leave in whole rows - 128 / 64 contiguous bytes per wave and
//   store - with the region search (boxes, column marks) in the same pass.
#define OW_T 64            // tile width = a wave's lanes
#define OW_Q 16
#define OW_CAMS 32          // camera records staged in LDS
#define OW_QLIST 8          // survivors of a sub-tile that are bounded again per quarter
#ifndef OW_ILP
#define OW_ILP 4            // independent pixels per lane in the evaluation
#endif
#define OW_EVAL (-3)        // qfill: evaluate the quarter's pixels
#define OW_NOBODY (-1)      // qfill / tile: no candidate: owner -1, valid 0
#define OW_UNOWNED (-2)     // tile: a candidate is unmasked but its alpha is 0: owner -1, valid 1

#ifdef OW_STAMP
// phase timers (timing experiments only, tools/probe_own_stamps.py): thread 0 of every
// OW_STAMP_STRIDE-th workgroup writes the cycles between consecutive stamps into a row of its own
// (plain stores: atomics on shared counters, one per stamp and workgroup, made the kernel 4.5 x
// slower and timed mostly themselves); [14] = 1, [15] = evaluated quarters
#define OW_STAMP_ROWS 2048
#define OW_STAMP_STRIDE 4
__device__ unsigned long long g_ow_stamps[OW_STAMP_ROWS][16];
#define OW_STAMP_ROW()                                                                       \
    (((blockIdx.y * gridDim.x + blockIdx.x) % OW_STAMP_STRIDE == 0 &&                        \
      (blockIdx.y * gridDim.x + blockIdx.x) / OW_STAMP_STRIDE < OW_STAMP_ROWS &&             \
      threadIdx.x == 0 && threadIdx.y == 0)                                                  \
         ? (int)((blockIdx.y * gridDim.x + blockIdx.x) / OW_STAMP_STRIDE)                    \
         : -1)
#define OW_STAMP_AT(k)                                                      \
    do {                                                                    \
        const int row_ = OW_STAMP_ROW();                                    \
        if (row_ >= 0) {                                                    \
            const unsigned long long now_ = __builtin_readcyclecounter();   \
            g_ow_stamps[row_][k] = now_ - ow_last;                          \
            ow_last = __builtin_readcyclecounter();                         \
        }                                                                   \
    } while (0)
extern "C" int pano_debug_own_stamps(unsigned long long *out, int reset) {
    // out[16]: the sampled workgroups' rows added up ([14] = how many)
    if (out) {
        static unsigned long long host[OW_STAMP_ROWS][16];
        PANO_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ow_stamps), sizeof(host)));
        for (int k = 0; k < 16; ++k) out[k] = 0;
        for (int r = 0; r < OW_STAMP_ROWS; ++r)
            if (host[r][14])
                for (int k = 0; k < 16; ++k) out[k] += host[r][k];
    }
    if (reset) {
        static unsigned long long zero[OW_STAMP_ROWS][16];
        PANO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ow_stamps), zero, sizeof(zero)));
    }
    return PANO_OK;
}
#else
#define OW_STAMP_AT(k) do { } while (0)
#endif

// The tile is 64 x 128 (eight sub-tiles: one camera list, one staging, one pair of bound passes per
// workgroup - 9 % faster than 64 x 64 on config 3's 4 000 workgroups, 64 x 256 loses 16 %).  The
// kernel's text is a parameter of the tile height (own_tile.inc); a 64 x 64 instance for small grids
// (a world-8 strip is 580 tall workgroups, ONE round on 256 CUs) was measured and gained nothing -
// strips 0.066 - 0.068 ms either way, config 2 0.058 against 0.065 (profiles/r05/own_small_*.txt).
#define OW_SUBS 8
#define OW_FN(name) name
#include "own_tile.inc"
#undef OW_FN
#define OW_TH_TALL (OW_Q * OW_SUBS)

// ---- linear_blend / no_blend straight from the frames (fused path) --------------
// stitcher.py:160-183 without materialising any patch: per mosaic pixel the
// covering cameras are mapped, unmasked ones sampled (same taps, LUT and alpha as
// the warp kernel) and combined in index order, exactly the reference's sums.
// A masked sample contributes tile = 0 and alpha = 0 (stitcher.py:176, 317), i.e.
// nothing, so it is skipped; no_blend keeps the LAST unmasked camera (:164-166).
// Reads 3 bytes of frame per sampled tap (L1/L2 absorb the 4-tap overlap),
// writes 3 B per mosaic pixel: HBM-light, bound by the per-sample arithmetic.
// PERCAM: every camera has its own colour table (equalised exposures); the
// tables then stay in global memory (1 KiB each, L1-resident) instead of LDS.
template <bool LINEAR, bool PERCAM>
__global__ __launch_bounds__(256) void blend_cameras_kernel(
    const pano_camera *__restrict__ cams, int n, int H, int W, int xs0, int xs1,
    const double *__restrict__ sin_t, const double *__restrict__ cos_t,
    const double *__restrict__ tan_p, const float *__restrict__ lut,
    uint8_t *__restrict__ mosaic, uint8_t *__restrict__ valid) {
    __shared__ CamList sh;
    __shared__ float s_lut[256];
    if (!PERCAM) s_lut[threadIdx.y * 64 + threadIdx.x] = lut[threadIdx.y * 64 + threadIdx.x];
    const int bx0 = xs0 + blockIdx.x * 64, by0 = blockIdx.y * 4;
    const int bx1 = min(bx0 + 64, xs1), by1 = min(by0 + 4, H);
    const int listed = build_camera_list(sh, cams, n, bx0, bx1, by0, by1);   // has the barriers s_lut needs
    int ncand = listed < 0 ? n : listed;
    const int *list = sh.list;
    // Cameras masked on the whole block (the interval bound of alpha_bound: behind the camera
    // or outside the frame everywhere) contribute nothing to either blend and leave the list.
    // On a closed sweep every pixel lies in the full-width rectangles of the ~20 frames across
    // the +-pi seam (stitcher.py:107-122 has no wrap handling), nearly all of them masked there:
    // 228 MP x 40 inverse maps made config 5's linear blend slower than its multiband blend.
    __shared__ double s_rng[6];
    __shared__ int s_keep[OWN_LIST];
    __shared__ int s_kept[4];
    if (listed > 16) {           // (a dozen cameras, config 3: the bounds cost more than they save)
        const int lane = threadIdx.x, wave = threadIdx.y;
        if (wave == 0) {
            const int xc = min(bx0 + lane, bx1 - 1);
            double lo_s = sin_t[xc], hi_s = lo_s, lo_c = cos_t[xc], hi_c = lo_c;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                lo_s = fmin(lo_s, __shfl_xor(lo_s, off, 64));
                hi_s = fmax(hi_s, __shfl_xor(hi_s, off, 64));
                lo_c = fmin(lo_c, __shfl_xor(lo_c, off, 64));
                hi_c = fmax(hi_c, __shfl_xor(hi_c, off, 64));
            }
            if (lane == 0) {
                s_rng[0] = lo_s; s_rng[1] = hi_s; s_rng[4] = lo_c; s_rng[5] = hi_c;
            }
        } else if (wave == 1 && lane == 0) {
            double lo_t = tan_p[by0], hi_t = lo_t;
            for (int yy = by0 + 1; yy < by1; ++yy) {
                lo_t = fmin(lo_t, tan_p[yy]);
                hi_t = fmax(hi_t, tan_p[yy]);
            }
            s_rng[2] = lo_t; s_rng[3] = hi_t;
        }
        __syncthreads();
        const int tid = wave * 64 + lane;
        const bool keep = tid < listed && alpha_bound(cams + sh.list[tid], s_rng).hi >= 0.0f;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_kept[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += s_kept[w];
        off += __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) s_keep[off] = sh.list[tid];                 // index order preserved
        ncand = s_kept[0] + s_kept[1] + s_kept[2] + s_kept[3];
        list = s_keep;
        __syncthreads();
    }

    const int x = bx0 + threadIdx.x, y = by0 + threadIdx.y;
    if (x >= xs1 || y >= H) return;
    const double s = sin_t[x], c = cos_t[x], t = tan_p[y];
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
    uint8_t last[3] = {0, 0, 0};
    bool any = false;
    // no_blend keeps the LAST unmasked camera (stitcher.py:164-166): walked from the end, the
    // first unmasked one is the answer
    for (int kk = 0; kk < ncand; ++kk) {
        const int k = LINEAR ? kk : ncand - 1 - kk;
        const int i = listed < 0 ? k : __builtin_amdgcn_readfirstlane(list[k]);
        const pano_camera *cam = cams + i;
        const int px = x - cam->x0, py = y - cam->y0;
        if ((unsigned)px >= (unsigned)cam->w || (unsigned)py >= (unsigned)cam->h) continue;
        float fx, fy;
        const int sw = cam->sw, sh_ = cam->sh;
        if (map_pixel(cam->proj, s, c, t, sw, sh_, fx, fy)) continue;
        any = true;
        const Taps tp = make_taps_unmasked(fx, fy, sw, sh_);
        const TapBytes tb = load_taps(cam->frame, sw, tp);
        float rgb[3];
        const float *__restrict__ gl = lut + (size_t)i * 256;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            rgb[ch] = PERCAM ? lerp4(lut_at(gl, tb.v[0][ch]), lut_at(gl, tb.v[1][ch]), lut_at(gl, tb.v[2][ch]),
                                     lut_at(gl, tb.v[3][ch]), tp)
                             : lerp4(lut_at(s_lut, tb.v[0][ch]), lut_at(s_lut, tb.v[1][ch]), lut_at(s_lut, tb.v[2][ch]),
                                     lut_at(s_lut, tb.v[3][ch]), tp);
        if (LINEAR) {
            const float a = alpha_at(cam->hat_x, cam->hat_y, tp);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) acc[ch] = acc[ch] + rgb[ch] * a;         // :177
            wsum = wsum + a;                                                         // :178
        } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) last[ch] = (uint8_t)(int)(255.0f * rgb[ch]);   // :166
            break;
        }
    }
    const size_t g = ((size_t)y * W + x) * 3;
    if (LINEAR) {
        const float ws = wsum == 0.0f ? 1.0f : wsum;                                 // :180
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            mosaic[g + ch] = (uint8_t)(int)(255.0f * __fdiv_rn(acc[ch], ws));         // :181-183
    } else {
        mosaic[g] = last[0];
        mosaic[g + 1] = last[1];
        mosaic[g + 2] = last[2];
    }
    if (valid) valid[(size_t)y * W + x] = any ? 1 : 0;
}

// Bounding boxes of the owned regions.  Same-address atomics serialise in L2,
// so only "corner" pixels issue them: a pixel whose left AND upper neighbours
// belong to someone else can be the first row / first column of its region, one
// whose left and lower neighbours differ the last row, one whose right and upper
// neighbours differ the last column.  Every extreme of a region is attained at
// such a pixel (the topmost pixel of the leftmost column has a foreign left and
// upper neighbour, and so on), and a region bounded by near-vertical seams has
// only a handful of them.
//
// The same pass marks, per patch, the columns in which it owns anything
// (marks[o][x] = 1 at the top pixel of every vertical run - idempotent plain
// stores); owned_spans_kernel turns the marks into column spans.
typedef short own8 __attribute__((ext_vector_type(8), aligned(2)));     // 8 owners, any address

__device__ __forceinline__ void owned_box_pixel(int o, int x, int y, bool up, bool down, bool left,
                                                bool right, int W, int32_t *__restrict__ boxes,
                                                int stride, uint8_t *__restrict__ marks) {
    if (o < 0) return;
    if (up && marks) marks[(size_t)o * W + x] = 1;
    if (!left && !right) return;
    if (left && up) {
        atomicMin(&boxes[(size_t)stride * o + 0], y);
        atomicMin(&boxes[(size_t)stride * o + 2], x);
    }
    if (left && down) atomicMax(&boxes[(size_t)stride * o + 1], y);
    if (right && up) atomicMax(&boxes[(size_t)stride * o + 3], x);
}

// Eight pixels of a row per thread: the row and its two neighbours as 16-byte loads (one
// thread per pixel made five 2-byte loads each and was bound by their number).
__global__ __launch_bounds__(256) void owned_boxes_kernel(const int16_t *__restrict__ owner,
                                                          int H, int W, int xs0, int xs1,
                                                          int32_t *__restrict__ boxes, int stride,
                                                          uint8_t *__restrict__ marks) {
    const int x8 = xs0 + (blockIdx.x * 64 + threadIdx.x) * 8, y = blockIdx.y * 4 + threadIdx.y;
    if (x8 >= xs1 || y >= H) return;
    const int16_t *row = owner + (size_t)y * W;
    const bool top = y == 0, bottom = y == H - 1;
    if (x8 + 8 <= xs1) {
        const own8 cur = *(const own8 *)(row + x8);
        const own8 upv = top ? cur : *(const own8 *)(row - W + x8);
        const own8 dnv = bottom ? cur : *(const own8 *)(row + W + x8);
        const int before = x8 == xs0 ? -32768 : row[x8 - 1];       // the strip's edge is foreign
        const int after = x8 + 8 == xs1 ? -32768 : row[x8 + 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = cur[j];
            const int l = j == 0 ? before : cur[j - 1], r = j == 7 ? after : cur[j + 1];
            owned_box_pixel(o, x8 + j, y, top || upv[j] != o, bottom || dnv[j] != o, l != o, r != o,
                            W, boxes, stride, marks);
        }
        return;
    }
    for (int x = x8; x < xs1; ++x) {                                // ragged end of the strip
        const int o = row[x];
        owned_box_pixel(o, x, y, top || row[x - W] != o, bottom || row[x + W] != o,
                        x == xs0 || row[x - 1] != o, x == xs1 - 1 || row[x + 1] != o, W, boxes,
                        stride, marks);
    }
}

// uint8(255 * v) with C truncation; v is in [0, 1] up to rounding.
__device__ __forceinline__ uint8_t quant255(float v) { return (uint8_t)(int)(255.0f * v); }

// ---- interior map ---------------------------------------------------------------
// The band-pass stack is a partition of unity: where every pixel within the
// largest Gaussian radius R of p (a (2R+1)^2 window) is owned by the same patch
// i, all of i's blurred alphas equal the full tap sum s_k, every other patch's
// are exact zeros, and  sum_k (band_k * s_k) / s_k  telescopes to the warped
// colour itself (|difference| <= 6e-8 measured against the oracle).  Such
// "interior" pixels need no blur at all; multiband blending only happens within
// R of a seam or of the border of the covered area.  The test runs on IB x IB
// blocks (IB = PANO_INTERIOR_BLOCK = 4; conservative): a block is interior when all blocks
// within ceil((R + IB - 1) / IB) of it are uniformly owned by the same patch.
#define IB PANO_INTERIOR_BLOCK

__global__ __launch_bounds__(256) void block_owner_kernel(const int16_t *__restrict__ owner,
                                                          int H, int W, int xs0, int xs1,
                                                          int H8, int W8, int blo, int bhi,
                                                          int16_t *__restrict__ bown) {
    // (block columns [blo, bhi) only: a strip's share, grown by the interior test's reach)
    const int bx = blo + blockIdx.x * 64 + threadIdx.x, by = blockIdx.y * 4 + threadIdx.y;
    if (bx >= bhi || by >= H8) return;
    const int x0 = bx * IB, y0 = by * IB;
    int o = -2;                                  // -2: mixed, or not inside the strip
    if (x0 >= xs0 && (x0 + IB < W ? x0 + IB : W) <= xs1) {
        o = owner[(size_t)y0 * W + x0];
        const bool whole = x0 + IB <= W && (W & 1) == 0;      // rows 4-byte aligned, IB in range
        if (x0 + IB <= W && y0 + IB <= H) {
            // all rows' owners at once, IB = 4 of them per 8-byte load at any (2-byte) alignment:
            // a loop that stops at the first mixed row is a chain of dependent loads, and with an
            // odd mosaic width (config 5: 46 079) it went pixel by pixel - 0.41 ms for 228 MP
            static_assert(IB == 4, "one 8-byte load per block row");
            typedef uint64_t u64_any __attribute__((aligned(2)));
            const uint64_t o16 = (uint16_t)o;
            const uint64_t all = o16 | o16 << 16 | o16 << 32 | o16 << 48;
            uint64_t diff = 0;
#pragma unroll
            for (int dy = 0; dy < IB; ++dy)
                diff |= *(const u64_any *)(owner + (size_t)(y0 + dy) * W + x0) ^ all;
            if (diff) o = -2;
        } else
        for (int dy = 0; dy < IB && o != -2; ++dy) {
            const int y = y0 + dy;
            if (y >= H) break;
            if (whole) {                         // one row of the block = IB / 2 32-bit words
                const uint32_t *q = (const uint32_t *)(owner + (size_t)y * W + x0);
                const uint32_t both = ((uint32_t)(uint16_t)o << 16) | (uint16_t)o;
                uint32_t diff = 0;
#pragma unroll
                for (int k = 0; k < IB / 2; ++k) diff |= q[k] ^ both;
                if (diff) o = -2;
            } else {
                for (int dx = 0; dx < IB; ++dx) {
                    const int x = x0 + dx;
                    if (x >= W) break;
                    if (owner[(size_t)y * W + x] != o) {
                        o = -2;
                        break;
                    }
                }
            }
        }
    }
    bown[(size_t)by * W8 + bx] = (int16_t)o;
}

// interior = the block's owner o >= 0 fills the (2 reach + 1)^2 neighbourhood of blocks
// (neighbours beyond the mosaic do not count: nothing is there).  Separable - first down the
// columns (col = o if the blocks above and below within reach all belong to o, else -2),
// then along the rows of that - and both passes run out of one LDS tile: 64 x 16 blocks per
// workgroup with a halo of `reach` blocks, so the block owners are read from memory 3-4 times
// instead of 2 (2 reach + 1) times.
// A window holds one owner exactly when every two neighbours in it agree (positions beyond the
// mosaic, -3, agree with anything: they lie outside all valid ones, never between two), so a
// pass is one bit per pair of neighbours - a 64-bit word per tile column, two ballots per tile
// row - and a window test is a shift and a compare instead of 2 reach + 1 LDS reads (61 k reads
// per workgroup at reach 12: 47 us of a config-3 stitch's side chain).
#define IT_W 64
#define IT_H 16
#define IT_REACH_MAX 20
__device__ __forceinline__ bool agree(int a, int b) { return a == b || a == -3 || b == -3; }

// Level classes (round 6): the same test at SEVERAL reaches - one per Gaussian level, ascending,
// the last one `reach` - on the one staged tile: a block's class is the number of leading levels
// whose window holds one owner (`classes`, optional; interior = the class is the number of levels).
// The collapse gathers, for a pixel of class j >= 1, only the copies of the levels j - 1 and up
// (multiband_compose_kernel).
struct LevelReaches {
    int n;
    int reach[PANO_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void interior_tile_kernel(const int16_t *__restrict__ bown,
                                                            int H8, int W8, int reach, int blo,
                                                            int bhi, int ilo, int ihi,
                                                            uint8_t *__restrict__ interior,
                                                            LevelReaches lv,
                                                            uint8_t *__restrict__ classes) {
    __shared__ int16_t s_own[(IT_H + 2 * IT_REACH_MAX) * (IT_W + 2 * IT_REACH_MAX)];
    __shared__ int16_t s_col[IT_H * (IT_W + 2 * IT_REACH_MAX)];
    __shared__ unsigned long long s_vmask[IT_W + 2 * IT_REACH_MAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // block columns [ilo, ihi) are classified (one GPU's strip: the others' blocks are nobody's
    // business - on a world-8 strip the two map kernels used to walk the whole mosaic's width,
    // 25 of a strip's 290 us of kernels); block owners exist for [blo, bhi) = that range grown by
    // the reach, everything else inside the mosaic counts as "not this strip's" (-2)
    const int bx0 = ilo + blockIdx.x * IT_W, by0 = blockIdx.y * IT_H;
    const int tw = IT_W + 2 * reach, th = IT_H + 2 * reach;          // th <= 56: a column's bits fit a word
    // (rows by wave, columns by lane: an index i = ty * tw + tx split by division cost twenty
    // instructions per element, more than the loads it addressed)
    // -3 marks a position beyond the mosaic
    for (int ty = wave; ty < th; ty += 4) {
        const int y = by0 - reach + ty;
        const bool row_in = y >= 0 && y < H8;
        for (int tx = lane; tx < tw; tx += 64) {
            const int x = bx0 - reach + tx;
            s_own[ty * tw + tx] = (row_in && x >= 0 && x < W8)
                                      ? (x >= blo && x < bhi ? bown[(size_t)y * W8 + x] : (int16_t)-2)
                                      : (int16_t)-3;
        }
    }
    __syncthreads();
    // down the columns: bit ty of a column's word = rows ty and ty + 1 agree
    for (int tx = tid; tx < tw; tx += 256) {
        unsigned long long m = 0;
        int prev = s_own[tx];
#pragma unroll 8
        for (int ty = 0; ty + 1 < th; ++ty) {
            const int cur = s_own[(ty + 1) * tw + tx];
            m |= (unsigned long long)agree(prev, cur) << ty;
            prev = cur;
        }
        s_vmask[tx] = m;
    }
    __syncthreads();
    // one pass of both window tests per level, smallest reach first; a level counts while every
    // level before it passed (the reaches ascend, so the tests are nested anyway)
    int cls[IT_H / 4];
#pragma unroll
    for (int r = 0; r < IT_H / 4; ++r) cls[r] = 0;
    const int nlv = classes ? lv.n : 1;
    for (int q = 0; q < nlv; ++q) {
        const int rq = classes ? lv.reach[q] : reach, off = reach - rq;
        const unsigned long long full = (1ull << (2 * rq)) - 1ull;       // 2 reach <= 40 pairs
        for (int ry = wave; ry < IT_H; ry += 4)
            for (int tx = lane; tx < tw; tx += 64) {
                const int o = s_own[(ry + reach) * tw + tx];
                const bool same = ((s_vmask[tx] >> (ry + off)) & full) == full;
                s_col[ry * tw + tx] = (int16_t)(o >= 0 && !same ? -2 : o);  // -3 stays -3
            }
        __syncthreads();
        // along the rows of that: a row's pairs as two ballots, a pixel's window as a shift
#pragma unroll
        for (int r = 0; r < IT_H / 4; ++r) {
            const int ry = wave + 4 * r;
            const int16_t *row = s_col + ry * tw;
            const int a0 = row[lane], a1 = row[min(lane + 1, tw - 1)];
            const int b0 = row[min(lane + 64, tw - 1)], b1 = row[min(lane + 65, tw - 1)];
            const unsigned long long lo = __ballot(lane + 1 < tw && agree(a0, a1));
            const unsigned long long hi = __ballot(lane + 65 < tw && agree(b0, b1));
            const int sh = lane + off;                   // first pair of this lane's window
            const unsigned long long win = sh == 0 ? lo
                                           : sh < 64 ? (lo >> sh) | (hi << (64 - sh))
                                                     : hi >> (sh - 64);
            const int o = row[lane + reach];
            if (o >= 0 && (win & full) == full && cls[r] == q) cls[r] = q + 1;
        }
        __syncthreads();                                 // s_col is the next level's
    }
#pragma unroll
    for (int r = 0; r < IT_H / 4; ++r) {
        const int y = by0 + wave + 4 * r, x = bx0 + lane;
        if (y >= H8 || x >= ihi) continue;
        interior[(size_t)y * W8 + x] = cls[r] == nlv ? 1 : 0;
        if (classes) classes[(size_t)y * W8 + x] = (uint8_t)cls[r];
    }
}

// What the collapse needs to finish an interior pixel on its own: the owner's
// frame is sampled exactly as the warp kernel samples it.
struct InteriorArgs {
    const uint8_t *interior;     // [H8][W8], NULL = no shortcut
    int W8;
    const pano_camera *cams;
    const double *sin_t, *cos_t, *tan_p;
    const float *lut;            // [256], or [n][256] when PERCAM
    int part;                    // 0 every pixel, 2 only the pixels the interior pass left
    const uint8_t *classes;      // [H8][W8] level classes (interior_tile_kernel), NULL = all class 0
};

// Where an interior pixel's colour table is read: staged in LDS, or in global memory (the
// per-camera tables of an equalised stitch).
enum { LUT_LDS = 0, LUT_GLOBAL = 1 };

// One interior pixel of the mosaic: the owner's warped colour, clipped, quantised.
template <int LUT>
__device__ __forceinline__ void shade_interior(const pano_camera *cam, const float *__restrict__ table,
                                               const InteriorArgs &ia, int x, int y, int W,
                                               uint8_t *__restrict__ mosaic,
                                               float *__restrict__ mosaic_f32) {
    // queued before the host has seen the layout's summary (device-side layout): a camera
    // whose frame is not resident is skipped here and reported by the caller
    if (!cam->frame) return;
    const int sw = cam->sw, sh = cam->sh;
    float fx, fy;
    map_pixel(cam->proj, table_f64(ia.sin_t, x), table_f64(ia.cos_t, x), table_f64(ia.tan_p, y), sw, sh, fx, fy);
    // an owned pixel is unmasked; where the whole wave samples away from the frame's last
    // row and column the taps need no border handling at all
    Taps tp = tap_base(fx, fy);
    TapBytes tb;
    if (__ballot(!taps_interior(tp, sw, sh)) == 0ull) {
        tb = load_taps_interior(cam->frame, sw, tp);
    } else {
        tp.x1 = min(tp.x1, sw - 1);
        tp.y1 = min(tp.y1, sh - 1);
        tb = load_taps(cam->frame, sw, tp);
    }
    const size_t g = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = lerp4(lut_at(table, tb.v[0][c]), lut_at(table, tb.v[1][c]),
                        lut_at(table, tb.v[2][c]), lut_at(table, tb.v[3][c]), tp);
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        if (mosaic_f32) mosaic_f32[g + c] = v;
        mosaic[g + c] = (uint8_t)(int)(255.0f * v);
    }
}

// The interior pixels of a wave nearly always have one owner (a pixel is interior when
// everything within the blur radius of it has): then the camera record is read through a
// wave-uniform address - scalar loads of its 9 doubles instead of 64 lanes fetching the same
// 120 bytes each.  `table`: the LDS copy, or the global table(s) at `stride` floats per camera.
template <int LUT>
__device__ __forceinline__ void shade_interior_of(int own, const float *__restrict__ table, int stride,
                                                  const InteriorArgs &ia, int x, int y, int W,
                                                  uint8_t *__restrict__ mosaic,
                                                  float *__restrict__ mosaic_f32) {
    const int own_u = __builtin_amdgcn_readfirstlane(own);
    if (__ballot(own != own_u) == 0)
        shade_interior<LUT>(ia.cams + own_u, table + (size_t)own_u * stride, ia, x, y, W, mosaic, mosaic_f32);
    else
        shade_interior<LUT>(ia.cams + own, table + (size_t)own * stride, ia, x, y, W, mosaic, mosaic_f32);
}

// The interior pixels alone (part 1 of the collapse): they need the owner map and the
// frames, not the blurred planes, so a caller can run them on a second stream.  (Beside the
// blur, with no LDS so that its waves fit on a CU next to the blur's workgroup, it costs the blur
// more than it saves the collapse: profiles/r05/notes.md, section 10.)
template <int LUT>
__global__ __launch_bounds__(256) void compose_interior_kernel(
    int H, int W, int xs0, int xs1, const int16_t *__restrict__ owner,
    uint8_t *__restrict__ mosaic, float *__restrict__ mosaic_f32, InteriorArgs ia, int lut_stride) {
    __shared__ float s_lut[LUT == LUT_LDS ? 256 : 1];
    if (LUT == LUT_LDS) {
        s_lut[threadIdx.y * 64 + threadIdx.x] = ia.lut[threadIdx.y * 64 + threadIdx.x];
        __syncthreads();
    }
    const int x = xs0 + blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= xs1 || y >= H) return;
    if (!ia.interior[(size_t)(y / IB) * ia.W8 + x / IB]) return;
    const int own = owner[(size_t)y * W + x];
    if (LUT == LUT_LDS)
        shade_interior_of<LUT>(own, s_lut, 0, ia, x, y, W, mosaic, mosaic_f32);
    else
        shade_interior_of<LUT>(own, ia.lut, lut_stride, ia, x, y, W, mosaic, mosaic_f32);
}

#define COMPOSE_MASKS 4            // 256 records through the wave-wide test, more: plain scan
// CLS: the gathers follow the pixels' level classes (ia.classes; option PANO_OPT_LEVEL_CLASSES).  A
// template parameter, not a test of the pointer: with the class a run-time value in every pixel's
// band loop the kernel was 12 - 20 % slower even where every class is 0 (0.49 -> 0.55 - 0.61 ms on
// config 3, visits a / e - g of round 6).
template <int L, bool PERCAM, bool CLS>
__global__ __launch_bounds__(256) void multiband_compose_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W, int xs0, int xs1,
    const int16_t *__restrict__ owner, const uint8_t *__restrict__ valid,
    uint8_t *__restrict__ mosaic, float *__restrict__ mosaic_f32, InteriorArgs ia) {
    __shared__ float s_lut[256];
    if (ia.interior && !PERCAM) {
        s_lut[threadIdx.y * 64 + threadIdx.x] = ia.lut[threadIdx.y * 64 + threadIdx.x];
        __syncthreads();
    }
    const int x = xs0 + blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const bool inside = x < xs1 && y < H;
    const bool is_interior = inside && ia.interior && ia.interior[(size_t)(y / IB) * ia.W8 + x / IB];
    // A wave with a seam pixel first finds, with all its lanes (lane l tests record l), the
    // records whose rectangle A meets its 64 pixels of row y: the seam pixels then walk the two
    // or three set bits instead of testing all n records one scalar load at a time.  No
    // barriers, no LDS (a per-block list built with barriers was slower than the plain scan).
    unsigned long long cand[COMPOSE_MASKS] = {};
    const bool masked = n <= 64 * COMPOSE_MASKS && __ballot(inside && !is_interior) != 0;
    if (masked) {
        const int wx0 = xs0 + blockIdx.x * 64, wx1 = min(wx0 + 64, xs1);
#pragma unroll
        for (int m = 0; m < COMPOSE_MASKS; ++m) {
            const int i = 64 * m + (int)threadIdx.x;
            bool hit = false;
            if (i < n) {
                const pano_patch *q = patches + i;
                const int ax = q->x0 + q->ax0, ay = q->y0 + q->ay0;
                hit = ax < wx1 && ax + q->aw > wx0 && ay <= y && y < ay + q->ah;
            }
            cand[m] = __ballot(hit);
        }
    }
    if (!inside) return;
    if (is_interior) {
        // interior pixel: the mosaic is the owner's warped colour, clipped, quantised
        if (ia.part == 2) return;                // compose_interior_kernel wrote it
        const int own = owner[(size_t)y * W + x];
        if (PERCAM)
            shade_interior_of<LUT_GLOBAL>(own, ia.lut, 256, ia, x, y, W, mosaic, mosaic_f32);
        else
            shade_interior_of<LUT_LDS>(own, s_lut, 0, ia, x, y, W, mosaic, mosaic_f32);
        return;
    }
    float layer[L][3], wsum[L];
#pragma unroll
    for (int k = 0; k < L; ++k) layer[k][0] = layer[k][1] = layer[k][2] = wsum[k] = 0.0f;
    // The pixel's level class j (0 .. L - 2 here): within the reach of the levels k < j everything
    // is the owner's, so for those levels the owner's blurred alpha is the full tap sum, every
    // other record's an exact zero, and  sum_{k<j} band_k  telescopes to  I - G_{j-1} I  of the
    // owner alone (the interior shortcut's argument, level by level).  Such a pixel needs, of
    // every record, the colour of copy j - 1 and the copies j .. L - 2 - not the copies below,
    // not alpha j - 1, and the warped planes of the owner's record only.
    const int j = CLS && L > 1 ? (int)ia.classes[(size_t)(y / IB) * ia.W8 + x / IB] : 0;
    const int own_px = CLS && j > 0 ? (int)owner[(size_t)y * W + x] : -1;
    float base[3] = {0.0f, 0.0f, 0.0f};          // I - G_{j-1} I of the owner (class j >= 1)

    // records in index order (the reference's summation order): the set bits, or all of them
    int m = 0;
    unsigned long long bits = masked ? cand[0] : 0ull;
    for (int i = 0;; ++i) {
        if (masked) {
            while (bits == 0ull && ++m < COMPOSE_MASKS) bits = cand[m];
            if (bits == 0ull) break;
            i = 64 * m + __ffsll((long long)bits) - 1;
            bits &= bits - 1ull;
        } else if (i >= n) {
            break;
        }
        const pano_patch p = patches[i];
        // outside A every weight of this patch is an exact 0 (header, "Windows")
        const int ax = x - p.x0 - p.ax0, ay = y - p.y0 - p.ay0;
        if ((unsigned)ax >= (unsigned)p.aw || (unsigned)ay >= (unsigned)p.ah) continue;
        const size_t vplane = (size_t)p.vh * p.vpitch;
        const size_t vo = (size_t)(ay + p.ay0 - p.vy0) * p.vpitch + (ax + p.ax0 - p.vx0);
        const size_t aplane = (size_t)p.ah * p.apitch, ao = (size_t)ay * p.apitch + ax;
        float hi[3], ha = 0.0f;                 // the copy that gets the minus
        if (!CLS || j == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) hi[c] = p.planes[c * vplane + vo];
        } else {
            const float *b = p.blurred + (size_t)(j - 1) * 4 * aplane + ao;
#pragma unroll
            for (int c = 0; c < 3; ++c) hi[c] = b[c * aplane];
            if (p.index == own_px) {
#pragma unroll
                for (int c = 0; c < 3; ++c) base[c] = p.planes[c * vplane + vo] - hi[c];
            }
        }
        if (L == 1) ha = owner[(size_t)y * W + x] == p.index ? 1.0f : 0.0f;   // sharp alpha (:208)
#pragma unroll
        for (int k = 0; k < L; ++k) {
            if (CLS && k < j) continue;          // telescoped into `base`
            float rgb[3], a;
            if (k < L - 1) {
                const float *b = p.blurred + (size_t)k * 4 * aplane + ao;
                a = b[3 * aplane];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float g = b[c * aplane];
                    rgb[c] = hi[c] - g;          // tile.rgb -= blur.rgb   (:227)
                    hi[c] = g;                   // prevs[idx] = blur      (:229)
                }
                ha = a;
            } else {                             // last level: G_{L-2} itself
                a = ha;
#pragma unroll
                for (int c = 0; c < 3; ++c) rgb[c] = hi[c];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) layer[k][c] = layer[k][c] + rgb[c] * a;   // :231
            wsum[k] = wsum[k] + a;                                                 // :232
        }
    }
    const bool ok = valid[(size_t)y * W + x] != 0;
    float out[3] = {CLS && ok ? base[0] : 0.0f, CLS && ok ? base[1] : 0.0f, CLS && ok ? base[2] : 0.0f};
#pragma unroll
    for (int k = 0; k < L; ++k) {
        if (CLS && k < j) continue;
        const float ws = wsum[k] == 0.0f ? 1.0f : wsum[k];                         // :237
#pragma unroll
        for (int c = 0; c < 3; ++c)
            out[c] = out[c] + __fdiv_rn(ok ? layer[k][c] : 0.0f, ws);              // :236,238
    }
    const size_t g = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = out[c];
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);                               // :240
        if (mosaic_f32) mosaic_f32[g + c] = v;
        mosaic[g + c] = quant255(v);                                               // :241
    }
}

__global__ __launch_bounds__(256) void linear_blend_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    uint8_t *__restrict__ mosaic) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float acc[3] = {0.0f, 0.0f, 0.0f}, wsum = 0.0f;
    for (int i = 0; i < n; ++i) {
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        const size_t plane = (size_t)p.vh * p.vpitch, o = (size_t)py * p.vpitch + px;
        const bool m = p.mask[(size_t)py * p.w + px] != 0;
        const float a = p.planes[3 * plane + o];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float t = m ? 0.0f : p.planes[c * plane + o];                    // :176
            acc[c] = acc[c] + t * a;                                               // :177
        }
        wsum = wsum + a;                                                           // :178
    }
    const float ws = wsum == 0.0f ? 1.0f : wsum;                                   // :180
    const size_t g = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) mosaic[g + c] = quant255(__fdiv_rn(acc[c], ws));   // :181-183
}

__global__ __launch_bounds__(256) void no_blend_kernel(
    const pano_patch *__restrict__ patches, int n, int H, int W,
    uint8_t *__restrict__ mosaic) {
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    uint8_t out[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {               // later patches overwrite (:164-166)
        const pano_patch p = patches[i];
        const int px = x - p.x0, py = y - p.y0;
        if ((unsigned)px >= (unsigned)p.w || (unsigned)py >= (unsigned)p.h) continue;
        if (p.mask[(size_t)py * p.w + px]) continue;
        const size_t plane = (size_t)p.vh * p.vpitch, o = (size_t)py * p.vpitch + px;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = quant255(p.planes[c * plane + o]);
    }
    const size_t g = ((size_t)y * W + x) * 3;
    mosaic[g] = out[0];
    mosaic[g + 1] = out[1];
    mosaic[g + 2] = out[2];
}

static int check_table(const void *table, int n, int H, int W, const char *who) {
    PANO_REQUIRE(table, "%s: null table", who);
    PANO_REQUIRE(n >= 0 && n <= 32767, "%s: %d patches (int16 owner map holds 32767)", who, n);
    PANO_REQUIRE(H > 0 && W > 0, "%s: bad mosaic shape %dx%d", who, H, W);
    return PANO_OK;
}

#define MOSAIC_GRID dim3 block(64, 4), grid(ceil_div(W, 64), ceil_div(H, 4))

// The stage-level kernels index whole-patch planes: V must be the patch.
extern "C" int pano_ownership(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                              int16_t *owner, uint8_t *valid) {
    PANO_ENTER(ctx, "pano_ownership");
    if (int rc = check_table(patches, n, H, W, "pano_ownership")) return rc;
    PANO_REQUIRE(owner && valid, "pano_ownership: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_OWNERSHIP, (hipStream_t)stream,
               hipLaunchKernelGGL(ownership_kernel, grid, block, 0, (hipStream_t)stream,
                                  patches, n, H, W, owner, valid));
    PANO_LAUNCH_CHECK("ownership_kernel");
    return PANO_OK;
}

extern "C" int pano_ownership_cameras(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                                      int xs0, int xs1, const double *sin_t, const double *cos_t,
                                      const double *tan_p, int16_t *owner, uint8_t *valid) {
    PANO_ENTER(ctx, "pano_ownership_cameras");
    if (int rc = check_table(cams, n, H, W, "pano_ownership_cameras")) return rc;
    PANO_REQUIRE(sin_t && cos_t && tan_p && owner && valid, "pano_ownership_cameras: null pointer");
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1, "pano_ownership_cameras: bad strip [%d, %d)", xs0, xs1);
    if (xs0 == xs1) return PANO_OK;
    // option PANO_OPT_OWN_PRUNE: bit 0 = prune by bounds (0: every listed camera at every pixel),
    // bit 1 = the round-4 kernel with one level of bounds (A/B, and a second implementation the
    // exactness tests compare with)
    const int prune = ctx->opt[PANO_OPT_OWN_PRUNE];
    dim3 block(64, 4), grid(ceil_div(xs1 - xs0, 64), ceil_div(H, (prune & 2) ? OWN_TILE_ROWS : OW_TH_TALL));
    PANO_TIMED(PK_OWNERSHIP_CAMS, (hipStream_t)stream,
               hipLaunchKernelGGL((prune & 2) ? ownership_cameras_l1_kernel : ownership_cameras_kernel,
                                  grid, block, (prune & 2) ? 0 : own_shared_bytes(n),
                                  (hipStream_t)stream, cams, n, H, W, xs0, xs1, sin_t, cos_t,
                                  tan_p, owner, valid, prune & 1, (int32_t *)nullptr, 0,
                                  (uint8_t *)nullptr));
    PANO_LAUNCH_CHECK("ownership_cameras_kernel");
    return PANO_OK;
}

extern "C" int pano_blend_cameras(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                                  int xs0, int xs1, int linear, const double *sin_t,
                                  const double *cos_t, const double *tan_p, const float *lut,
                                  int lut_stride, uint8_t *mosaic, uint8_t *valid) {
    PANO_ENTER(ctx, "pano_blend_cameras");
    if (int rc = check_table(cams, n, H, W, "pano_blend_cameras")) return rc;
    PANO_REQUIRE(sin_t && cos_t && tan_p && lut && mosaic, "pano_blend_cameras: null pointer");
    PANO_REQUIRE(lut_stride == 0 || lut_stride == 256,
                 "pano_blend_cameras: lut_stride %d (0 = shared table, 256 = per camera)", lut_stride);
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1, "pano_blend_cameras: bad strip [%d, %d)", xs0, xs1);
    if (xs0 == xs1) return PANO_OK;
    dim3 block(64, 4), grid(ceil_div(xs1 - xs0, 64), ceil_div(H, 4));
    hipStream_t s = (hipStream_t)stream;
#define BLEND(LIN, PC)                                                                        \
    PANO_TIMED(PK_BLEND_CAMERAS, s,                                                           \
               hipLaunchKernelGGL((blend_cameras_kernel<LIN, PC>), grid, block, 0, s, cams, n, \
                                  H, W, xs0, xs1, sin_t, cos_t, tan_p, lut, mosaic, valid))
    if (linear && lut_stride) BLEND(true, true);
    else if (linear) BLEND(true, false);
    else if (lut_stride) BLEND(false, true);
    else BLEND(false, false);
#undef BLEND
    PANO_LAUNCH_CHECK("blend_cameras_kernel");
    return PANO_OK;
}

// One workgroup per patch turns its column marks into merged runs: the four waves turn 64
// columns at a time into ballot words in LDS (the loads of 256 columns in flight together),
// then wave 0 peels the runs off the words with bit scans (all its lanes run the same scalar
// bookkeeping, lane 0 writes).  (One wave per patch doing both - a serial chain through every
// load - took 0.2 ms on the 46 079 columns of config 5's seam frames.)
//   regions[i] = {ymin, ymax, xmin, xmax, count, xa_0, xb_0, xa_1, xb_1, ...}
#define SPAN_WORDS 1024
__global__ __launch_bounds__(256) void owned_spans_kernel(const uint8_t *__restrict__ marks,
                                                          int W, int xs0, int xs1, int min_gap,
                                                          int max_spans, int stride,
                                                          int32_t *__restrict__ regions) {
    __shared__ unsigned long long s_bits[SPAN_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t *row = marks + (size_t)blockIdx.x * W;
    int32_t *out = regions + (size_t)blockIdx.x * stride + 5;
    int cnt = 0, last = 0;
    // marks exist only between the box's first and last column (the boxes are complete)
    const int32_t *box = regions + (size_t)blockIdx.x * stride;
    xs0 = max(xs0, box[2]);
    xs1 = min(xs1, box[3] + 1);
    for (int chunk0 = xs0; chunk0 < xs1; chunk0 += 64 * SPAN_WORDS) {
        const int words = min(SPAN_WORDS, (xs1 - chunk0 + 63) >> 6);
        for (int w = wave; w < words; w += 4) {
            const int x = chunk0 + 64 * w + lane;
            const unsigned long long bal = __ballot(x < xs1 && row[x] != 0);
            if (lane == 0) s_bits[w] = bal;
        }
        __syncthreads();
        if (wave == 0) {
            for (int w = 0; w < words; ++w) {
                const int base = chunk0 + 64 * w;
                unsigned long long bal = s_bits[w];
                while (bal) {
                    const int s = __ffsll((long long)bal) - 1;
                    const unsigned long long rest = ~(bal >> s);      // 0 bits = the run
                    const int len = rest ? __ffsll((long long)rest) - 1 : 64 - s;
                    const int xa = base + s, xb = xa + len - 1;
                    if (cnt && (xa - last - 1 < min_gap || cnt == max_spans)) {
                        if (lane == 0) out[2 * (cnt - 1) + 1] = xb;   // extend the current span
                    } else {
                        if (lane == 0) {
                            out[2 * cnt] = xa;
                            out[2 * cnt + 1] = xb;
                        }
                        ++cnt;
                    }
                    last = xb;
                    bal = s + len >= 64 ? 0ull : bal & ~((1ull << (s + len)) - 1ull);
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) regions[(size_t)blockIdx.x * stride + 4] = cnt;
}

// Empty boxes, no spans, and the column marks cleared (one launch instead of a memset and a
// kernel: a stitch of 8 MP is seventeen launches with 6 - 10 us between them).
__global__ __launch_bounds__(256) void init_regions_kernel(int32_t *regions, int n, int stride,
                                                           uint8_t *marks, size_t mark_bytes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)n) {
        int32_t *r = regions + i * stride;
        r[0] = 0x7fffffff;
        r[1] = -1;
        r[2] = 0x7fffffff;
        r[3] = -1;
        r[4] = 0;
    }
    // 16 bytes per thread and trip (hipMalloc'd buffers are 256-byte aligned), the tail by bytes
    const size_t words = mark_bytes / 16, step = (size_t)gridDim.x * 256;
    uint4 *m16 = (uint4 *)marks;
    for (size_t k = i; k < words; k += step) m16[k] = make_uint4(0u, 0u, 0u, 0u);
    if (i < mark_bytes - words * 16) marks[words * 16 + i] = 0;
}

static int launch_init_regions(hipStream_t s, int32_t *regions, int n, int stride, uint8_t *marks,
                               int W) {
    size_t bytes = (size_t)n * W;
    if ((uintptr_t)marks & 15) {                         // (no allocator hands this out)
        PANO_HIP(hipMemsetAsync(marks, 0, bytes, s));
        bytes = 0;
    }
    const size_t want = bytes / 16 / 256 + 1;
    const int blocks = (int)(want < 2048 ? (want > (size_t)ceil_div(n, 256) ? want : ceil_div(n, 256))
                                         : 2048);
    hipLaunchKernelGGL(init_regions_kernel, dim3(blocks), dim3(256), 0, s, regions, n, stride, marks,
                       bytes);
    PANO_LAUNCH_CHECK("init_regions_kernel");
    return PANO_OK;
}

extern "C" int pano_owned_regions(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0,
                                  int xs1, int n, int min_gap, int max_spans, uint8_t *marks,
                                  int32_t *regions) {
    PANO_ENTER(ctx, "pano_owned_regions");
    PANO_REQUIRE(owner && marks && regions, "pano_owned_regions: null pointer");
    PANO_REQUIRE(H > 0 && W > 0 && n >= 0 && n <= 32767 && max_spans >= 1 && min_gap >= 0,
                 "pano_owned_regions: bad argument");
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1, "pano_owned_regions: bad strip [%d, %d)", xs0, xs1);
    if (n == 0) return PANO_OK;
    hipStream_t s = (hipStream_t)stream;
    const int stride = 5 + 2 * max_spans;
    if (int rc = launch_init_regions(s, regions, n, stride, marks, W)) return rc;
    if (xs0 == xs1) return PANO_OK;
    dim3 block(64, 4), grid(ceil_div(xs1 - xs0, 512), ceil_div(H, 4));
    // the box fields are the first four ints of each record: stride-aware view
    PANO_TIMED(PK_OWNED_BOXES, s,
               hipLaunchKernelGGL(owned_boxes_kernel, grid, block, 0, s, owner, H, W, xs0, xs1,
                                  regions, stride, marks));
    PANO_LAUNCH_CHECK("owned_boxes_kernel");
    PANO_TIMED(PK_OWNED_SPANS, s,
               hipLaunchKernelGGL(owned_spans_kernel, dim3(n), dim3(256), 0, s, marks, W, xs0, xs1,
                                  min_gap, max_spans, stride, regions));
    PANO_LAUNCH_CHECK("owned_spans_kernel");
    return PANO_OK;
}

// pano_ownership_cameras + pano_owned_regions in one pass over the mosaic: the ownership
// kernel leaves each camera's bounding box and column marks behind while the owners are
// still in its registers (the separate box kernel re-read the owner map three times and sat,
// with its launch gaps, between the ownership and the one host wait of a stitch).
extern "C" int pano_ownership_regions(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                                      int xs0, int xs1, const double *sin_t, const double *cos_t,
                                      const double *tan_p, int16_t *owner, uint8_t *valid,
                                      int min_gap, int max_spans, uint8_t *marks,
                                      int32_t *regions) {
    PANO_ENTER(ctx, "pano_ownership_regions");
    if (int rc = check_table(cams, n, H, W, "pano_ownership_regions")) return rc;
    PANO_REQUIRE(sin_t && cos_t && tan_p && owner && valid && marks && regions,
                 "pano_ownership_regions: null pointer");
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1, "pano_ownership_regions: bad strip [%d, %d)", xs0, xs1);
    PANO_REQUIRE(max_spans >= 1 && min_gap >= 0, "pano_ownership_regions: bad argument");
    if (n == 0) return PANO_OK;
    hipStream_t s = (hipStream_t)stream;
    const int stride = 5 + 2 * max_spans;
    if (int rc = launch_init_regions(s, regions, n, stride, marks, W)) return rc;
    if (xs0 == xs1) return PANO_OK;
    const int prune = ctx->opt[PANO_OPT_OWN_PRUNE];
    dim3 block(64, 4), grid(ceil_div(xs1 - xs0, 64), ceil_div(H, (prune & 2) ? OWN_TILE_ROWS : OW_TH_TALL));
    PANO_TIMED(PK_OWNERSHIP_CAMS, s,
               hipLaunchKernelGGL((prune & 2) ? ownership_cameras_l1_kernel : ownership_cameras_kernel,
                                  grid, block, (prune & 2) ? 0 : own_shared_bytes(n), s, cams, n, H, W, xs0,
                                  xs1, sin_t, cos_t, tan_p, owner, valid, prune & 1, regions, stride,
                                  marks));
    PANO_LAUNCH_CHECK("ownership_cameras_kernel");
    PANO_TIMED(PK_OWNED_SPANS, s,
               hipLaunchKernelGGL(owned_spans_kernel, dim3(n), dim3(256), 0, s, marks, W, xs0, xs1,
                                  min_gap, max_spans, stride, regions));
    PANO_LAUNCH_CHECK("owned_spans_kernel");
    return PANO_OK;
}

static int interior_map_launch(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0, int xs1,
                               const int *radii, int n_radii, int16_t *block_owner,
                               uint8_t *interior, uint8_t *classes, const char *who) {
    void *const stream = (void *)ctx->stream;
    PANO_REQUIRE(owner && block_owner && interior && radii, "%s: null pointer", who);
    PANO_REQUIRE(H > 0 && W > 0 && n_radii >= 1 && n_radii < PANO_MAX_LEVELS, "%s: bad argument", who);
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1, "%s: bad strip [%d, %d)", who, xs0, xs1);
    LevelReaches lv = {};
    lv.n = n_radii;
    for (int k = 0; k < n_radii; ++k) {
        PANO_REQUIRE(radii[k] >= 0 && (k == 0 || radii[k] >= radii[k - 1]),
                     "%s: radii must ascend (level %d: %d)", who, k, radii[k]);
        lv.reach[k] = ceil_div(radii[k] + IB - 1, IB);
    }
    const int radius = radii[n_radii - 1];
    const int H8 = ceil_div(H, IB), W8 = ceil_div(W, IB), reach = lv.reach[n_radii - 1];
    if (xs0 == xs1) return PANO_OK;
    // the blocks that meet the columns [xs0, xs1) are classified; their owners are needed `reach`
    // blocks further (blocks that do not lie inside [xs0, xs1) are "not this strip's" either way)
    const int ilo = xs0 / IB, ihi = ceil_div(xs1, IB);
    const int blo = ilo - reach > 0 ? ilo - reach : 0, bhi = ihi + reach < W8 ? ihi + reach : W8;
    dim3 block(64, 4), grid(ceil_div(bhi - blo, 64), ceil_div(H8, 4));
    hipStream_t s = (hipStream_t)stream;
    PANO_TIMED(PK_INTERIOR, s,
               hipLaunchKernelGGL(block_owner_kernel, grid, block, 0, s, owner, H, W, xs0, xs1,
                                  H8, W8, blo, bhi, block_owner));
    PANO_LAUNCH_CHECK("block_owner_kernel");
    PANO_REQUIRE(reach <= IT_REACH_MAX, "%s: radius %d reaches %d blocks (at most %d)", who, radius,
                 reach, IT_REACH_MAX);
    hipLaunchKernelGGL(interior_tile_kernel, dim3(ceil_div(ihi - ilo, IT_W), ceil_div(H8, IT_H)),
                       dim3(256), 0, s, block_owner, H8, W8, reach, blo, bhi, ilo, ihi, interior, lv,
                       classes);
    PANO_LAUNCH_CHECK("interior_tile_kernel");
    return PANO_OK;
}

extern "C" int pano_interior_map(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0,
                                 int xs1, int radius, int16_t *block_owner, uint8_t *interior) {
    PANO_ENTER(ctx, "pano_interior_map");
    return interior_map_launch(ctx, owner, H, W, xs0, xs1, &radius, 1, block_owner, interior, nullptr,
                               "pano_interior_map");
}

extern "C" int pano_interior_classes(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0,
                                     int xs1, const int *radii, int n_radii, int16_t *block_owner,
                                     uint8_t *interior, uint8_t *classes) {
    PANO_ENTER(ctx, "pano_interior_classes");
    PANO_REQUIRE(classes, "pano_interior_classes: null pointer");
    return interior_map_launch(ctx, owner, H, W, xs0, xs1, radii, n_radii, block_owner, interior,
                               classes, "pano_interior_classes");
}

extern "C" int pano_multiband_compose(pano_ctx *ctx, const pano_patch *patches, int n, int H,
                                      int W, int xs0, int xs1, int n_levels, const int16_t *owner,
                                      const uint8_t *valid, const uint8_t *interior,
                                      const uint8_t *classes, const pano_camera *cams,
                                      const double *sin_t,
                                      const double *cos_t, const double *tan_p, const float *lut,
                                      int lut_stride, uint8_t *mosaic, float *mosaic_f32,
                                      int part) {
    PANO_ENTER(ctx, "pano_multiband_compose");
    PANO_REQUIRE(part >= 0 && part <= 2, "pano_multiband_compose: part %d outside 0..2", part);
    PANO_REQUIRE(part == 0 || interior, "pano_multiband_compose: parts need the interior map");
    if (part != 1)
        if (int rc = check_table(patches, n, H, W, "pano_multiband_compose")) return rc;
    PANO_REQUIRE(H > 0 && W > 0, "pano_multiband_compose: bad mosaic shape %dx%d", H, W);
    PANO_REQUIRE(owner && mosaic && (valid || part == 1), "pano_multiband_compose: null pointer");
    PANO_REQUIRE(n_levels >= 1 && n_levels <= PANO_MAX_LEVELS,
                 "pano_multiband_compose: n_levels %d outside [1, %d]", n_levels, PANO_MAX_LEVELS);
    PANO_REQUIRE(xs0 >= 0 && xs1 <= W && xs0 <= xs1,
                 "pano_multiband_compose: bad strip [%d, %d)", xs0, xs1);
    PANO_REQUIRE(!interior || (cams && sin_t && cos_t && tan_p && lut),
                 "pano_multiband_compose: the interior map needs cameras, tables and LUT");
    PANO_REQUIRE(lut_stride == 0 || lut_stride == 256,
                 "pano_multiband_compose: lut_stride %d (0 = shared table, 256 = per camera)",
                 lut_stride);
    if (xs0 == xs1) return PANO_OK;
    PANO_REQUIRE(!classes || interior, "pano_multiband_compose: level classes without the interior map");
    InteriorArgs ia = {interior, ceil_div(W, IB), cams, sin_t, cos_t, tan_p, lut, part,
                       ctx->opt[PANO_OPT_LEVEL_CLASSES] ? classes : nullptr};
    const bool percam = interior && lut_stride != 0;
    dim3 block(64, 4), grid(ceil_div(xs1 - xs0, 64), ceil_div(H, 4));
    hipStream_t s = (hipStream_t)stream;
    if (part == 1) {
        if (percam)
            PANO_TIMED(PK_COMPOSE_INTERIOR, s,
                       hipLaunchKernelGGL(compose_interior_kernel<LUT_GLOBAL>, grid, block, 0, s, H,
                                          W, xs0, xs1, owner, mosaic, mosaic_f32, ia, lut_stride));
        else
            PANO_TIMED(PK_COMPOSE_INTERIOR, s,
                       hipLaunchKernelGGL(compose_interior_kernel<LUT_LDS>, grid, block, 0, s, H,
                                          W, xs0, xs1, owner, mosaic, mosaic_f32, ia, lut_stride));
        PANO_LAUNCH_CHECK("compose_interior_kernel");
        return PANO_OK;
    }
#define COMPOSE_AS(L, PC, CL)                                                            \
    PANO_TIMED(PK_COMPOSE, s,                                                           \
               hipLaunchKernelGGL((multiband_compose_kernel<L, PC, CL>), grid, block, 0, s, \
                                  patches, n, H, W, xs0, xs1, owner, valid, mosaic,     \
                                  mosaic_f32, ia))
#define COMPOSE(L)                                                                      \
    case L:                                                                             \
        if (ia.classes && L > 1) {                                                      \
            if (percam) COMPOSE_AS(L, true, true); else COMPOSE_AS(L, false, true);     \
        } else {                                                                        \
            if (percam) COMPOSE_AS(L, true, false); else COMPOSE_AS(L, false, false);   \
        }                                                                               \
        break;
    switch (n_levels) {
        COMPOSE(1) COMPOSE(2) COMPOSE(3) COMPOSE(4) COMPOSE(5) COMPOSE(6) COMPOSE(7) COMPOSE(8)
    }
#undef COMPOSE_AS
#undef COMPOSE
    PANO_LAUNCH_CHECK("multiband_compose_kernel");
    return PANO_OK;
}

extern "C" int pano_linear_blend(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                                 uint8_t *mosaic) {
    PANO_ENTER(ctx, "pano_linear_blend");
    if (int rc = check_table(patches, n, H, W, "pano_linear_blend")) return rc;
    PANO_REQUIRE(mosaic, "pano_linear_blend: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_LINEAR, (hipStream_t)stream,
               hipLaunchKernelGGL(linear_blend_kernel, grid, block, 0, (hipStream_t)stream,
                                  patches, n, H, W, mosaic));
    PANO_LAUNCH_CHECK("linear_blend_kernel");
    return PANO_OK;
}

extern "C" int pano_no_blend(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                             uint8_t *mosaic) {
    PANO_ENTER(ctx, "pano_no_blend");
    if (int rc = check_table(patches, n, H, W, "pano_no_blend")) return rc;
    PANO_REQUIRE(mosaic, "pano_no_blend: null output");
    MOSAIC_GRID;
    PANO_TIMED(PK_NOBLEND, (hipStream_t)stream,
               hipLaunchKernelGGL(no_blend_kernel, grid, block, 0, (hipStream_t)stream,
                                  patches, n, H, W, mosaic));
    PANO_LAUNCH_CHECK("no_blend_kernel");
    return PANO_OK;
}
