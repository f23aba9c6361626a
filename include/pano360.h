/*
 * pano360.h - C ABI of libpano360_hip.so (MI355X / gfx950).
 *
 * The reference (Banus/pano360) is pure Python and has no FFI; its boundary
 * for the warp/blend hot path is the Python function surface of stitcher.py.
 * Each entry point below names the reference function (file:line, relative to
 * the reference repo) whose arithmetic it replaces.  pano360_amd/stitcher.py
 * keeps those Python signatures and calls these symbols through ctypes
 * (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - every kernel-launching entry point takes a context (pano_ctx, below) first:
 *     it names the device and the stream the work is enqueued on, carries the
 *     option switches and owns everything the library remembers between calls
 *     (work-list buffers of the blur, device copies and operand tables of the
 *     tap sets it has seen, the timing registry).  There is no process-wide
 *     state: two contexts - two streams, two devices, two host threads - do not
 *     see each other.  A context is not thread-safe: one per host thread;
 *   - every pointer marked "dev" is a device (HBM) pointer owned by the
 *     caller; "host" pointers are ordinary host memory read before return;
 *   - all work is enqueued asynchronously on the context's stream, nothing
 *     synchronises;
 *   - return value: 0 on success, a negative PANO_E* code otherwise, with a
 *     thread-local message available from pano_last_error();
 *   - no exceptions cross the boundary.
 *
 * Layouts
 *   frame      uint8  [H][W][3]      as cv2.imread hands it to the reference
 *   planes     float  [c][vh][vpitch] planar R,G,B(,A) of a warped patch
 *                                    window, pitch = pano_pitch(width) floats
 *   mask       uint8  [h][w]         1 = outside the source frame
 *   owner      int16  [H][W]         patch index owning the pixel, -1 = none
 *   mosaic     uint8  [H][W][3]
 *
 * Windows.  Multiband blending only ever uses a patch near the pixels it owns:
 * beyond the Gaussian radius of the last owned pixel every blurred alpha is an
 * exact 0 and the patch contributes 0 to every sum (stitcher.py:231-232).  A
 * patch therefore carries two sub-rectangles, in patch-local coordinates:
 *   A = bounding box of its owned pixels grown by the largest radius R
 *       (clipped to the patch): where blurred copies exist and are gathered;
 *   V = A grown by R again (closed under the REFLECT_101 border rule, its
 *       columns rounded outwards to multiples of 4 and clipped to the patch):
 *       where the warped colour is needed as blur input.
 * Both default to the whole patch (the stage-level blender API).
 */
#ifndef PANO360_H
#define PANO360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANO_OK 0
#define PANO_EINVAL (-1)   /* bad argument (shape, null pointer, limits)   */
#define PANO_EHIP (-2)     /* a HIP runtime call failed                    */
#define PANO_ELIMIT (-3)   /* size beyond a compiled-in limit              */

#define PANO_MAX_TAPS 129      /* largest Gaussian aperture (sigma <= 16)  */
#define PANO_MAX_LEVELS 8      /* n_levels of multiband_blend              */
#define PANO_TAP_LEAD 7        /* zero taps in front of a padded tap table */
#define PANO_TAP_PAD 40        /* padded table length = ntaps + this       */
#ifndef PANO_INTERIOR_BLOCK
#define PANO_INTERIOR_BLOCK 4  /* side of the blocks of pano_interior_map  */
#endif

/* One warped patch as the blenders see it (reference: the tuples appended at
 * stitcher.py:318-319).  All pointers dev. */
typedef struct pano_patch {
    float *planes;             /* [3 or 4][vh][vpitch] over window V        */
    uint8_t *mask;             /* [h][w], full patch; NULL in the fused path */
    float *blurred;            /* [n_levels-1][4][ah][apitch] over A; or NULL */
    float *scratch;            /* [n_levels-1][4][vh][apitch] row-pass output */
    int32_t y0, x0, h, w;      /* patch rectangle in mosaic coordinates     */
    int32_t vy0, vx0, vh, vw;  /* window V, patch-local                     */
    int32_t ay0, ax0, ah, aw;  /* rectangle A, patch-local                  */
    int32_t vpitch, apitch;    /* floats per row of planes / blurred        */
    int32_t index;             /* camera / owner-map index this record belongs
                                  to: a patch whose owned pixels fall into
                                  several far-apart column spans (a frame that
                                  straddles the +-pi seam of a 360 degree sweep)
                                  is split into one record per span           */
    int32_t tiles_off;         /* first entry of this record's 64 x 128 column
                                  tiles in the tile_flags array (row-major,
                                  ceil(aw/64) per row)                        */
} pano_patch;

/* What pano_layout_windows reports besides the records. */
typedef struct pano_layout {
    int64_t planes_floats;     /* arena sizes the records' pointers will index     */
    int64_t blurred_floats;
    int64_t scratch_floats;
    int32_t n_records, n_tiles;
    int32_t max_vw, max_vh, max_aw, max_ah;
    int32_t missing;           /* records of cameras flagged absent in `have`      */
} pano_layout;

/* One registered frame (reference: bundle_adj.Image, bundle_adj.py:18-33,
 * plus the patch rectangle stitch() derives for it). */
typedef struct pano_camera {
    double proj[9];            /* K R, row-major (bundle_adj.py:31-33)      */
    const uint8_t *frame;      /* dev uint8 [sh][sw][3]; may be NULL for the
                                  ownership kernel, which reads no pixels   */
    const double *hat_x;       /* dev double[sw] = _hat(sw) (stitcher.py:251) */
    const double *hat_y;       /* dev double[sh]                            */
    int32_t sh, sw;            /* frame size                                */
    int32_t y0, x0, h, w;      /* patch rectangle in mosaic coordinates     */
} pano_camera;

/* One (i, j) iteration of equalize_gains (stitcher.py:44-63). */
typedef struct pano_pair {
    double minv[9];            /* cv::invert of the pixel homography j -> i
                                  (stitcher.py:48), row-major                */
    int32_t i, j;              /* camera indices, frame i is the destination */
} pano_pair;

/* One SIFT keypoint (reference: the cv2.KeyPoint objects of features.py:197). */
typedef struct pano_sift_keypoint {
    float x, y;                /* position                                   */
    float size;                /* diameter of the meaningful neighbourhood    */
    float angle;               /* degrees, [0, 360)                           */
    float response;            /* |contrast| of the refined extremum          */
    int32_t octave;            /* octave | layer << 8 | round((xi+0.5)*255) << 16 */
    int32_t r, c;              /* refined sample in the octave's own grid     */
} pano_sift_keypoint;

/* Colour tables.  The reference turns a frame into float32(u8)/255
 * (stitcher.py:259) and, with equalize=True, into clip(gain * that, 0, 1)
 * (stitcher.py:66): either way a function of the uint8 value alone, so the
 * kernels sample uint8 frames through a 256-entry float table built on the
 * host with NumPy.  Entry points that sample several cameras take
 * (lut, lut_stride): camera k uses lut + k * lut_stride; stride 0 = one table
 * for all (no equalisation), stride 256 = one table per camera. */

const char *pano_version(void);
const char *pano_last_error(void);
int pano_device_count(void);
/* Row pitch (in floats) used for every float plane of width w. */
int pano_pitch(int w);
/* PANO_INTERIOR_BLOCK as this build of the library was compiled with it. */
int pano_interior_block(void);

/* The context (no reference counterpart: the reference is one Python process
 * with OpenCV's global state).  pano_ctx_create binds a device and a stream
 * (a hipStream_t passed as void*, NULL = the default stream); the caller keeps
 * the stream alive for as long as the context uses it.  pano_ctx_set_stream
 * re-targets later calls (work already queued stays where it is; state built
 * on the old stream - tile flags, work lists, tap tables - is ordered by the
 * caller with events, as for any two streams).  pano_ctx_destroy waits for the
 * context's stream and frees what the context owns.
 * Options (pano_ctx_set_option / pano_ctx_get_option):
 *   PANO_OPT_BLUR_KERNEL  which kernels run the multiband Gaussian levels:
 *                         PANO_BLUR_MFMA (default) = the fused matrix-core kernel
 *                         (split float16 operands, float32 accumulate),
 *                         PANO_BLUR_VALU = separate row / column passes in float32
 *                         on the vector ALU (one FMA per tap)
 *   PANO_OPT_OWN_PRUNE    bit 0: 1 (default) = pano_ownership_cameras skips cameras that
 *                         rigorous bounds exclude (per 64 x 16 sub-tile, the survivors again
 *                         per 16 x 16 quarter); 0 = evaluate every camera at every pixel.
 *                         bit 1: 2 = the one-level kernel of round 4 (bounds per 64 x 16
 *                         sub-tile only; A/B and cross-check).  Same maps, bit for bit.
 *   PANO_OPT_BLUR_SEGMENTS  matrix-core blur: 1 (default) = when a launch has too few column
 *                         strips to fill the CUs (one GPU's share of a panorama, small
 *                         scenes) each strip is cut into vertical segments; 0 = never
 *   PANO_OPT_BLUR_LEAN    matrix-core blur: 1 (default) = a kernel with a short, hand-ordered
 *                         instruction stream takes every work item (those whose bands hold
 *                         reflected columns, unaligned windows or low patches load them
 *                         element by element); 0 = the general kernel for everything.
 *                         Same results bit for bit.  With it on, the five
 *                         Gaussian levels of a six-level pyramid (n_levels = 6) run in ONE
 *                         launch (the two lightest levels on one wave pair); off, they split
 *                         into launches of 2 + 2 + 1 levels that each stage the bands.
 *   PANO_OPT_STITCH_STREAMS  pano_stitch_multiband: 1 (default) = the interior map runs beside
 *                         the region search, and the blur's tile flags and work list beside
 *                         the warp, on a second stream the context owns (ordered by events);
 *                         0 = everything on the context's stream.  Same results.
 *   PANO_OPT_STITCH_ASYNC  pano_stitch_multiband: 1 = when the previous stitch of the same
 *                         shape went through, the record table is laid out by a kernel and
 *                         the warp, blur and collapse are queued behind it at once, sized by
 *                         what the previous layout needed (plus slack); the host then waits
 *                         for the layout's summary while the GPU works, and only if this
 *                         layout needed more (or an arena is too small) the stitch is laid
 *                         out again on the host and its tail queued a second time.
 *                         0 (default) = always the host layout: measured, the round trip it
 *                         saves is already covered by the side stream's kernels (DESIGN.md
 *                         section 4.14), so the simpler path is the default.  Same results.
 *                         (2 = as 1 with launch bounds the layout is sure to exceed: the
 *                         tests' way into the fallback.)
 *   PANO_OPT_BLUR_SEG_LEN matrix-core blur, with PANO_OPT_BLUR_SEGMENTS on: 0 (default) = the
 *                         length of the vertical segments a launch with few work items is cut
 *                         into comes from the sort kernel's list-schedule estimate; n >= 4 =
 *                         segments of n bands (32 rows each); -1 = nothing is cut.  Same results
 *                         bit for bit (A/B of the estimate, and the tests' way to other cuts).
 *   PANO_OPT_SIFT_GRAPH   pano_sift_detect: 1 (default) = a frame's launch sequence is captured
 *                         into a HIP graph the second time a set of buffers is used and replayed
 *                         from then on (one hipGraphLaunch per frame instead of ~110 launches);
 *                         0 = always launch by launch.  Same results bit for bit.
 *   PANO_OPT_LEVEL_CLASSES  pano_multiband_compose / pano_stitch_multiband: 1 = the collapse
 *                         uses the level classes (pano_interior_classes; the stitch computes
 *                         them); 0 (default) = it gathers every copy on every pixel that is not
 *                         interior.  The mosaics agree to float32 rounding.  Measured (round 6,
 *                         profiles/r06/ab_level_classes.txt): the classes take 13 % off the
 *                         collapse's HBM traffic and ADD 15 - 19 % to its time - it is bound by its
 *                         memory instructions, which a wave issues while ANY lane needs them,
 *                         not by bytes - so the option is off. */
typedef struct pano_ctx pano_ctx;
#define PANO_OPT_BLUR_KERNEL 0
#define PANO_OPT_OWN_PRUNE 1
#define PANO_OPT_BLUR_SEGMENTS 2
#define PANO_OPT_BLUR_LEAN 3
#define PANO_OPT_STITCH_STREAMS 4
#define PANO_OPT_STITCH_ASYNC 5
#define PANO_OPT_BLUR_SEG_LEN 6
#define PANO_OPT_SIFT_GRAPH 7
#define PANO_OPT_LEVEL_CLASSES 8
#define PANO_OPT_COMPOSE_COMPACT 9
#define PANO_OPT_COUNT 10
#define PANO_BLUR_MFMA 0
#define PANO_BLUR_VALU 1
int pano_ctx_create(int device, void *stream, pano_ctx **out);
int pano_ctx_destroy(pano_ctx *ctx);
int pano_ctx_set_stream(pano_ctx *ctx, void *stream);
int pano_ctx_set_option(pano_ctx *ctx, int option, int value);
int pano_ctx_get_option(const pano_ctx *ctx, int option, int *value);

/* Instrumentation for bench.py (no reference counterpart): when enabled every
 * kernel launch of this context is bracketed by HIP events recorded on its
 * stream.  pano_timing_read waits for kernel class `kid`'s events and returns
 * the summed duration in ms and the number of launches since the last enable. */
int pano_timing_enable(pano_ctx *ctx, int on);
int pano_kernel_count(void);
const char *pano_kernel_name(int kid);
int pano_timing_read(pano_ctx *ctx, int kid, double *total_ms, int *launches);

/* _add_weights                                        stitcher.py:251-263
 * frame uint8 [h][w][3] -> rgba float32 [h][w][4] interleaved (the array the
 * reference leaves in reg.img).  lut255: dev float[256] = float32(i)/255
 * (built on the host with NumPy); hat_x / hat_y: dev double[w] / [h]. */
int pano_add_weights(pano_ctx *ctx, const uint8_t *frame, int h, int w,
                     const float *lut255, const double *hat_x, const double *hat_y,
                     float *rgba);

/* Inverse map + mask + bilinear REFLECT remap of one whole patch
 *                                       stitcher.py:300-317 (+ cv2.remap)
 * proj: host double[9] = K R row-major (bundle_adj.py:31-33).
 * sin_t, cos_t: dev double tables over mosaic columns, tan_p over mosaic rows
 * (angles = index*resolution + min, stitcher.py:301-303, NumPy on the host).
 * (gx0, gy0) = patch origin in the mosaic, pw x ph = patch size.
 * Outputs: planes [4][ph][pitch] (alpha already multiplied by ~mask, :317),
 * mask [ph][pw]; map_x / map_y float [ph][pw] optional (NULL to skip). */
int pano_warp_spherical(pano_ctx *ctx, const uint8_t *frame, int sh, int sw,
                        const double *proj, const double *sin_t, const double *cos_t,
                        const double *tan_p, const float *lut255, const double *hat_x,
                        const double *hat_y, int gx0, int gy0, int pw, int ph,
                        float *planes, uint8_t *mask, float *map_x, float *map_y);

/* Same arithmetic, colour only, for window V of EVERY patch in one launch:
 * cams[i].frame is warped into patches[i].planes ([3][vh][vpitch]).  Alpha and
 * mask are not produced: the fused path gets them from
 * pano_ownership_cameras.  cams, patches: dev arrays of n records;
 * max_vw / max_vh: the largest window among them (sizes the grid).
 * need (optional): the per-tile flags pano_blur_tiles writes (one byte per 32 x 32
 * tile of every record, at patches[i].tiles_off); 64 x 4 pixel blocks none of whose
 * tiles is needed are left unwritten - nothing reads them. */
int pano_warp_windows(pano_ctx *ctx, const pano_camera *cams, const pano_patch *patches,
                      int n, int max_vw, int max_vh, const double *sin_t,
                      const double *cos_t, const double *tan_p, const float *lut,
                      int lut_stride, const uint8_t *need);

/* Ownership + validity from warped patches   stitcher.py:196-204, 266-271
 * owner = first-index argmax of the patches' alpha plane (planes[3]), -1 where
 * all are 0; valid = OR over patches of ~mask.  patches: dev array of n
 * pano_patch with 4 planes and a mask, V = whole patch. */
int pano_ownership(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                   int16_t *owner, uint8_t *valid);

/* The same two maps straight from the cameras (no pixel data is read): for
 * every mosaic pixel in columns [xs0, xs1) and every camera whose patch
 * rectangle holds it, the inverse map, the mask and the bilinear alpha are
 * re-evaluated exactly as pano_warp_spherical does - but only for the cameras
 * that can still win: per 64 x 16 tile, cameras whose alpha is bounded (interval
 * arithmetic on the ray, rigorous) below another camera's lower bound are
 * skipped, which changes neither map (option PANO_OPT_OWN_PRUNE = 0 disables it).
 * cams: dev array. */
int pano_ownership_cameras(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                           int xs0, int xs1, const double *sin_t, const double *cos_t,
                           const double *tan_p, int16_t *owner, uint8_t *valid);

/* Where each patch owns pixels inside the column strip [xs0, xs1), one record
 * of 5 + 2*max_spans int32 per patch:
 *   {ymin, ymax, xmin, xmax, count, xa_0, xb_0, xa_1, xb_1, ...}
 * bounding box (inclusive, mosaic coordinates; ymax < ymin when the patch owns
 * nothing there) and `count` column spans [xa, xb]: runs of columns in which
 * the patch owns at least one pixel, runs closer than min_gap columns merged
 * (so that the spans' rectangles A stay disjoint), at most max_spans (later
 * runs are folded into the last span).  marks: dev uint8 [n][W] workspace. */
int pano_owned_regions(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0, int xs1,
                       int n, int min_gap, int max_spans, uint8_t *marks,
                       int32_t *regions);

/* pano_ownership_cameras and pano_owned_regions in one pass over the mosaic
 * (stitcher.py:196-204, 266-271 and the bookkeeping of the sharp masks of
 * :207-208): the ownership kernel leaves the cameras' boxes and column marks
 * behind while the owners are in its registers.  Same owner, valid and regions
 * as the two calls in sequence. */
int pano_ownership_regions(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                           int xs0, int xs1, const double *sin_t, const double *cos_t,
                           const double *tan_p, int16_t *owner, uint8_t *valid,
                           int min_gap, int max_spans, uint8_t *marks, int32_t *regions);

/* The n_levels-1 Gaussian blurs of every patch  stitcher.py:207-208, 218, 226
 * (cv2.GaussianBlur(warped, (0,0), 4*sqrt(2k+1)) with the alpha channel
 * replaced by the sharp mask owner == index), evaluated on rectangle A of each
 * patch from its colour planes over V, all levels, records and channels in one
 * launch (matrix-core kernel; the vector-ALU alternative runs one row pass for
 * all levels and one column pass per level through patches[i].scratch).
 * patches: dev array of n records with planes and blurred set (scratch only for
 * the vector-ALU kernels); max_aw / max_vh / max_ah: the largest extents.
 * taps: HOST float, n_blur tables laid out back to back; table k has
 * ntaps[k] + PANO_TAP_PAD floats: PANO_TAP_LEAD + ((R - r_k) & 3) zeros, the
 * ntaps[k] taps, zeros, where r_k = ntaps[k] / 2 and R = max r_k (the extra
 * zeros keep the row pass's 16-byte LDS reads aligned for every level).
 * ntaps: host int[n_blur].  Writes patches[i].blurred.  The context keeps a
 * device copy (and the matrix-core operand tables) of every tap set it has
 * seen, keyed on the apertures and the tap VALUES; the first call with a new
 * set uploads it in stream order, later calls only hash ~400 floats.
 * interior (optional, with tile_flags): the map of pano_interior_map; tiles
 * that hold only interior pixels (and the intermediate rows only they would
 * read) are skipped.  tile_flags: dev uint8, one entry per tile of every record,
 * record i's entries starting at patches[i].tiles_off, written here.  The tile
 * grid follows the context's PANO_OPT_BLUR_KERNEL, reported by pano_blur_tile_grid():
 *   32: 32 x 32 tiles anchored at multiples of 32 in patch coordinates, i.e.
 *       ((ax0+aw-1)>>5) - (ax0>>5) + 1 per row, ((ay0+ah-1)>>5) - (ay0>>5) + 1
 *       rows (the matrix-core kernel, default);
 *    0: 64-column x 128-row tiles relative to A, ceil(aw/64) per row,
 *       ceil(ah/128) rows (the vector-ALU kernels, PANO_BLUR_VALU).
 * The matrix-core kernel computes in split float16 (hi + lo, three products)
 * with float32 accumulation and does not use patches[i].scratch.
 * pano_multiband_blur_prepare (optional): the part of the call that depends on the
 * records' geometry and the interior map only (tile flags and the sorted work list of
 * the matrix-core kernel).  It may be queued on another stream while the warp fills the
 * planes; the caller orders it before the pano_multiband_blur call on the same table
 * (same patches pointer and n) with an event.  Without it pano_multiband_blur does
 * the same work itself.
 * pano_blur_tiles (optional, tile grid 32 only, before the two above, same stream order):
 * writes tile_flags from the interior map and warp_need = the tiles of window V whose
 * pixels the blur (its band fetches reach 16 ceil(radius / 16) columns and two tile rows
 * past an active tile) or the collapse will read, for pano_warp_windows; the following
 * prepare / blur call on the same table does not recompute the flags. */
int pano_blur_tile_grid(const pano_ctx *ctx);
int pano_blur_tiles(pano_ctx *ctx, const pano_patch *patches, int n, int max_aw,
                    int max_ah, int W, int radius, const uint8_t *interior,
                    uint8_t *tile_flags, uint8_t *warp_need);
int pano_multiband_blur_prepare(pano_ctx *ctx, const pano_patch *patches, int n,
                                int max_aw, int max_ah, int W, const uint8_t *interior,
                                uint8_t *tile_flags);
int pano_multiband_blur(pano_ctx *ctx, const pano_patch *patches, int n, int max_aw,
                        int max_vh, int max_ah, const int16_t *owner, int W,
                        const float *taps, const int *ntaps, int n_blur,
                        const uint8_t *interior, uint8_t *tile_flags);

/* Interior map (no reference counterpart: an exact-in-real-arithmetic property
 * of stitcher.py:210-241).  Where every pixel within `radius` (the largest
 * Gaussian radius) of p is owned by one patch, the band-pass stack telescopes:
 * all of that patch's blurred alphas equal the full tap sum, every other
 * patch's are exact zeros, and the multiband mosaic at p is the patch's warped
 * colour (to float32 rounding, <= 6e-8 absolute).  interior: dev uint8
 * [ceil(H/B)][ceil(W/B)], B = PANO_INTERIOR_BLOCK, 1 = every pixel of the B x B
 * block is such a pixel (conservative: tested on whole blocks); block_owner: dev
 * int16 workspace, twice that shape.  Only columns [xs0, xs1) of owner are read, and only the
 * blocks that meet those columns are written (one GPU's strip of the mosaic: the rest of
 * `interior` keeps whatever it held). */
int pano_interior_map(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0, int xs1,
                      int radius, int16_t *block_owner, uint8_t *interior);
/* The same test level by level (round 6).  radii: host int [n_radii], the Gaussian radii of the
 * levels, ascending (stitcher.py:218: the apertures grow with the level); classes: dev uint8, the
 * interior map's shape: the number of leading levels k whose (2 radii[k] + 1)^2 window around the
 * block holds one owner (0 .. n_radii; interior = the map at radii[n_radii - 1] = class n_radii).
 * For a pixel of class j >= 1 the levels below j telescope to  I - G_{j-1} I  of the owner alone:
 * pano_multiband_compose then gathers, of every record over the pixel, the colour of copy j - 1
 * and the copies j and up only - not the copies below, not alpha j - 1, and the warped planes of
 * the owner's record only (same result to float32 rounding, as for the interior pixels). */
int pano_interior_classes(pano_ctx *ctx, const int16_t *owner, int H, int W, int xs0, int xs1,
                          const int *radii, int n_radii, int16_t *block_owner, uint8_t *interior,
                          uint8_t *classes);

/* Host side of the fused path: the record table from the owned regions
 * (no reference counterpart; the arithmetic of "Windows" above).  regions: host copy
 * of pano_owned_regions' output, [n][5 + 2 max_spans]; rects: host int32 [n][4] =
 * patch rectangles (y0, y1, x0, x1) in mosaic coordinates; have (optional): [n], 0 =
 * that camera's frame is not resident; radius = the largest Gaussian radius;
 * [xs0, xs1) = the mosaic columns to produce; tile_grid = pano_blur_tile_grid() of the
 * context that will run the blur (32 or 0).  Writes one record per (camera, owned
 * column span) that reaches the strip, in camera order, with rectangles A and V,
 * pitches, tile offsets, and - until pano_layout_place - arena OFFSETS in the pointer
 * fields.  pano_layout_place turns them into addresses inside the three arenas
 * (blurred aligned up to 128 bytes; blurred / scratch may be NULL when unused). */
int pano_layout_windows(int tile_grid, const int32_t *regions, int n, int max_spans,
                        const int32_t *rects, const uint8_t *have, int radius,
                        int xs0, int xs1, int n_blur, pano_patch *records, int cap,
                        pano_layout *out);
int pano_layout_place(pano_patch *records, int n_records, void *planes, void *blurred,
                      void *scratch);

/* Band-pass build + collapse                     stitcher.py:210-241
 * Gathers, per mosaic pixel and in patch order, layer_k / wsum_k of every
 * level over the patches whose A holds the pixel, zeroes outside `valid`,
 * sums the levels, clips, truncates to uint8 - for the mosaic columns
 * [xs0, xs1) (the whole mosaic: 0, W; one GPU's share when the mosaic is split
 * into column strips).  mosaic / mosaic_f32 are full-size [H][W][3] buffers,
 * only the strip is written; mosaic_f32 is optional.
 * interior (optional): pixels of interior blocks are not gathered; the owner's
 * frame (cams[owner].frame) is sampled there exactly as the warp samples it,
 * which needs cams, the trig tables and the colour tables (all NULL
 * otherwise).
 * classes (optional, with interior): the level classes of pano_interior_classes - a pixel
 * of class j >= 1 gathers the copies j - 1 and up only (see there).
 * part: 0 = every pixel of the strip; 1 = the interior pixels only - they depend
 * on the owner map and the frames, not on the patches (patches / valid may be
 * NULL), so this part can be queued on another stream beside the warp and the
 * blur; 2 = the remaining pixels only. */
int pano_multiband_compose(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                           int xs0, int xs1, int n_levels, const int16_t *owner,
                           const uint8_t *valid, const uint8_t *interior,
                           const uint8_t *classes, const pano_camera *cams,
                           const double *sin_t, const double *cos_t, const double *tan_p,
                           const float *lut, int lut_stride, uint8_t *mosaic,
                           float *mosaic_f32, int part);

/* linear_blend (linear != 0) or no_blend (linear == 0) of the mosaic columns
 * [xs0, xs1) straight from the frames         stitcher.py:160-183 + :300-317
 * No patch is materialised: per mosaic pixel the covering cameras (UNPADDED
 * patch rectangles, stitcher.py:289-291) are mapped, sampled and combined in
 * index order.  cams: dev array with frame pointers set.  valid (optional,
 * dev uint8 [H][W]) receives _valid (stitcher.py:266-271) for the strip. */
int pano_blend_cameras(pano_ctx *ctx, const pano_camera *cams, int n, int H, int W,
                       int xs0, int xs1, int linear, const double *sin_t,
                       const double *cos_t, const double *tan_p, const float *lut,
                       int lut_stride, uint8_t *mosaic, uint8_t *valid);

/* Overlap statistics of equalize_gains           stitcher.py:36-63
 * For each pair, every pixel (x, y) of frame i is looked up in frame j as
 * cv2.warpPerspective(img_j, hom, (w, h), INTER_LINEAR, BORDER_TRANSPARENT)
 * does on the float32 RGBA image (OpenCV semantics restated, parity unpinned):
 * in double, with x = xb + x1, xb = (x / bw0) * bw0 the column-block start,
 *   W = 32 / (m6*xb + m7*y + m8 + m6*x1)          (0 if the denominator is 0)
 *   X = cvRound(clamp_int((m0*xb + m1*y + m2 + m0*x1) * W)),  Y likewise,
 * taps (X >> 5, Y >> 5) saturated to int16, fractions X & 31, Y & 31; the pixel
 * is written only if all four taps lie inside frame j; the bilinear sum runs
 * left to right in float32.  A pixel counts where the sampled alpha (analytic,
 * float32(hat_y*hat_x) per tap) is non-zero (stitcher.py:58).
 * All frames must be h x w (the reference sizes everything by regions[0],
 * stitcher.py:41).  bw0 = min(1024 / min(16, h), w) (WarpPerspectiveInvoker's
 * block width).  lut255: dev float[256], float32(u8)/255.
 * partials: dev double [n_pairs][pano_overlap_blocks(h, w)][3] workspace.
 * stats: dev double [n_pairs][3] = {pixel count (stitcher.py:59), sum of
 * frame i's colours over those pixels and the 3 channels, sum of the sampled
 * colours of frame j}; the means of stitcher.py:62-63 are sum / (3 count).
 * Sums are taken in double in a fixed order (the reference's np.mean sums in
 * float32 pairwise; the two agree to ~1e-7 relative). */
int pano_overlap_blocks(int h, int w);
int pano_overlap_stats(pano_ctx *ctx, const pano_camera *cams, const pano_pair *pairs,
                       int n_pairs, int h, int w, int bw0, const float *lut255,
                       double *partials, double *stats);

/* linear_blend on warped patches                        stitcher.py:171-183 */
int pano_linear_blend(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                      uint8_t *mosaic);

/* no_blend                                              stitcher.py:160-168 */
int pano_no_blend(pano_ctx *ctx, const pano_patch *patches, int n, int H, int W,
                  uint8_t *mosaic);

/* crop_mosaic rectangle                                 stitcher.py:340-369
 * valid: dev uint8 [H][W].  heights: dev int32 [H][W] workspace.
 * result: dev int64[6] = {found, y0, x0, h, w, area}. */
int pano_crop_rect(pano_ctx *ctx, const uint8_t *valid, int H, int W, int32_t *heights,
                   int64_t *result);

/* Separable symmetric filter on one float plane, BORDER_REFLECT_101
 *                     cv2.GaussianBlur at features.py:24 / stitcher.py:226
 * taps: HOST padded table of one aperture (layout and caching as for
 * pano_multiband_blur); tmp: dev [h][pitch]. */
int pano_blur_plane(pano_ctx *ctx, const float *src, float *dst, float *tmp, int h, int w,
                    int pitch, const float *taps, int ntaps);

/* cv2.pyrDown on one float plane                        features.py:155
 * src [h][w] (dense) -> dst [(h+1)/2][(w+1)/2] (dense). */
int pano_pyr_down(pano_ctx *ctx, const float *src, int h, int w, float *dst);

/* Scale-space building blocks of the SIFT detector the reference obtains from
 * OpenCV (features.py:192-201); with pano_blur_plane they make the Gaussian and
 * difference-of-Gaussian pyramid (pano360_amd/features.py: sift_pyramid).
 * OpenCV semantics restated, parity unpinned (see csrc/pyramid.hip).
 *   pano_gray_u8     cvtColor(BGR2GRAY) on uint8 [h][w][3] -> float [h][w]
 *   pano_resize_up2  resize(2w x 2h, INTER_LINEAR): float [h][w] -> [2h][2w]
 *   pano_decimate2   resize(w/2 x h/2, INTER_NEAREST): float [h][w] -> [h/2][w/2]
 *   pano_subtract    out = a - b over n floats (one DoG layer) */
int pano_gray_u8(pano_ctx *ctx, const uint8_t *bgr, int h, int w, float *out);
/* One step of buildGaussianPyramid + buildDoGPyramid, fused (csrc/scalespace.hip):
 *   dst = GaussianBlur(src, taps) (REFLECT_101; both passes in one launch, the row-pass
 *   image stays in LDS), dog (optional) = dst - src.
 * src, dst, dog: dev float [h][w] dense, distinct; taps: HOST float[ntaps] (the bare kernel
 * of cv::getGaussianKernel, no padding), ntaps odd and <= 33 (sigma <= 4). */
int pano_scale_step(pano_ctx *ctx, const float *src, int h, int w, const float *taps, int ntaps,
                    float *dst, float *dog);
/* The whole scale space of one frame in ONE call (createInitialImage + buildGaussianPyramid
 * + buildDoGPyramid): grey -> 2x bilinear -> blur to sigma, then per octave n_layers + 2
 * pano_scale_step launches and the nearest-neighbour halving of layer n_layers into the
 * next octave.  frame: dev uint8 [h][w][3]; octave o is rows_o x cols_o with rows_0 = 2h,
 * cols_0 = 2w, halved (floor) from octave to octave, all >= 1; gauss / dog: HOST arrays of
 * n_octaves dev pointers, gauss[o] = float [n_layers+3][rows_o][cols_o], dog[o] likewise
 * with n_layers+2; taps: HOST float, n_layers + 3 bare kernels back to back (kernel 0: the
 * base blur, kernel i: the step into layer i), ntaps: HOST their apertures;
 * work: dev float [5 h w] scratch (grey image, doubled base). */
int pano_scale_space(pano_ctx *ctx, const uint8_t *frame, int h, int w, int n_octaves,
                     int n_layers, const float *taps, const int *ntaps, float *const *gauss,
                     float *const *dog, float *work);
int pano_resize_up2(pano_ctx *ctx, const float *src, int h, int w, float *dst);
int pano_decimate2(pano_ctx *ctx, const float *src, int h, int w, float *dst);
int pano_subtract(pano_ctx *ctx, const float *a, const float *b, size_t n, float *out);

/* Decimated Laplacian-pyramid blending of two images     blend.py:105-140
 * (blend.laplacian_blending: cv2.pyrDown / cv2.pyrUp pyramids of two float32
 * images and a float64 mask, la*gm + lb*(1-gm) per level in float64, collapse,
 * clip, uint8).  Images are interleaved [h][w][c], c <= 4, dense; is_f64 selects
 * double (the mask and everything after the mix) or float (the image pyramids).
 * OpenCV's pyrDown / pyrUp restated, parity unpinned (csrc/laplacian.hip).
 *   pano_pyr_down_image  cv2.pyrDown: [h][w][c] -> [(h+1)/2][(w+1)/2][c]
 *   pano_pyr_up_image    cv2.pyrUp(src)[:oh, :ow] (blend.py:126,138), combined with
 *                        `other` [oh][ow][c]: mode 0 = the up-sampled image,
 *                        1 = other - up (a Laplacian level), 2 = other + up (collapse)
 *   pano_u8_to_f32       img.astype("float32")                        blend.py:132-133
 *   pano_laplacian_mix   out = la*gm + lb*(1.0 - gm) in the mask's type   blend.py:136
 *                        (float64 against the default / a float64 mask, float32 against a
 *                        float32 mask, as NumPy promotes them)
 *   pano_clip_u8         np.clip(x, 0, 255).astype("uint8")             blend.py:140 */
int pano_pyr_down_image(pano_ctx *ctx, const void *src, int h, int w, int c, int is_f64,
                        void *dst);
int pano_pyr_up_image(pano_ctx *ctx, const void *src, int sh, int sw, int c, int is_f64,
                      const void *other, int mode, void *dst, int oh, int ow);
int pano_u8_to_f32(pano_ctx *ctx, const uint8_t *src, size_t n, float *dst);
int pano_laplacian_mix(pano_ctx *ctx, const float *la, const float *lb, const void *gm, size_t n,
                       int is_f64, void *out);
int pano_clip_u8(pano_ctx *ctx, const void *src, size_t n, int is_f64, uint8_t *dst);

/* cv2.resize(im, None, fx=1/shrink, fy=1/shrink) on a uint8 image   stitcher.py:419-420
 * (INTER_LINEAR, OpenCV's 8-bit fixed-point path restated; parity unpinned).
 * xtab / ytab: dev int32 [ow][4] / [oh][4] = (first tap, second tap, coefficient of
 * the first, of the second; 11-bit fixed point), built by the host exactly as the
 * oracle builds them.  Both NULL: the exact 2:1 reduction, which cv2.resize takes by
 * rounded 2 x 2 box means (sh == 2 oh, sw == 2 ow). */
int pano_resize_u8(pano_ctx *ctx, const uint8_t *src, int sh, int sw, int c,
                   const int32_t *xtab, const int32_t *ytab, uint8_t *dst, int oh, int ow);

/* Keypoints and descriptors of SIFT_create().detectAndCompute  features.py:192-198
 * (the arithmetic lives in OpenCV's xfeatures2d/sift.cpp, restated with the
 * SIFT_create() defaults; parity unpinned).  The scale space comes from the
 * building blocks above, one contiguous stack per octave:
 * gauss[o] = dev float [n_layers+3][rows_o][cols_o], dog likewise with n_layers+2;
 * dims = dev int32 {rows_0, cols_0, rows_1, cols_1, ...}.
 *   pano_sift_extrema   findScaleSpaceExtrema + adjustLocalExtrema of one octave:
 *                       refined extrema (angle 0, coordinates of the doubled base
 *                       image, octave index not yet shifted) appended at
 *                       cands[atomicAdd(count)] while below max_cands.
 *   pano_sift_orient    calcOrientationHist: every candidate becomes one keypoint
 *                       per dominant orientation, appended to kpts.  n_cands: dev.
 *   pano_sift_describe  calcSIFTDescriptor of n keypoints as detectAndCompute
 *                       returns them (full-resolution coordinates, octave field
 *                       shifted by first_octave): desc dev float [n][128], values
 *                       0..255.  RootSIFT (features.py:198) is left to the caller.
 * List order is arbitrary (atomics); pano_sift_sort_unique puts it into OpenCV's:
 *   pano_sift_sort_unique  KeyPointsFilter::removeDuplicatedSorted + the first-octave
 *                       adjustment of detectAndCompute: the n keypoints of pano_sift_orient
 *                       sorted by x, y, size (descending), angle, response (descending),
 *                       octave (descending), the first of every (x, y, size, angle) group
 *                       kept, positions and sizes scaled by 2^first_octave and the octave
 *                       byte shifted by first_octave -> out (dev, room for n), *n_out (dev).
 *                       work: dev scratch of pano_sift_sort_work_bytes(n) bytes.
 * n_dev (optional, pano_sift_sort_unique and pano_sift_describe): a device int holding the
 * real count, at most n - the host then passes the capacity as n and never waits for the
 * counters pano_sift_orient / pano_sift_sort_unique leave on the device. */
int pano_sift_extrema(pano_ctx *ctx, const float *dog, int rows, int cols, int octave,
                      int n_layers, float contrast_thr, float edge_thr, float sigma,
                      pano_sift_keypoint *cands, int *count, int max_cands);
int pano_sift_orient(pano_ctx *ctx, const float *const *gauss, const int *dims,
                     int n_layers, const pano_sift_keypoint *cands, const int *n_cands,
                     int max_cands, pano_sift_keypoint *kpts, int *count, int max_kpts);
int pano_sift_describe(pano_ctx *ctx, const float *const *gauss, const int *dims,
                       int first_octave, const pano_sift_keypoint *kpts, int n,
                       const int *n_dev, float *desc);
size_t pano_sift_sort_work_bytes(int n);
int pano_sift_sort_unique(pano_ctx *ctx, const pano_sift_keypoint *kpts, int n, const int *n_dev,
                          int first_octave, void *work, pano_sift_keypoint *out, int *n_out);

/* One frame of the detector's front end per call           features.py:192-201 (_detect)
 * = pano_scale_space, then - with `detect` - pano_sift_extrema for every octave,
 * pano_sift_orient, pano_sift_sort_unique and pano_sift_describe, in that order on the
 * context's stream, nothing waited for: the candidate / keypoint / kept counts stay on the
 * device in counts[0..2] (zeroed here).  Every launch's grid and arguments follow from the
 * frame size and the buffers alone, so with PANO_OPT_SIFT_GRAPH (default) the sequence is
 * captured into a HIP graph the second time a set of buffers comes by and replayed afterwards:
 * `frame_copy` (dev, h * w * 3 bytes, optional) is the frame buffer the graph reads - the
 * caller's frame is copied into it in front of every replay; without it, or while kernels are
 * being timed, the sequence is queued launch by launch.  Results are the same either way.
 * Buffers: as the entry points named above take them; gauss / dog: host arrays of n_octaves
 * device pointers; gauss_dev / dims_dev: the device arrays pano_sift_orient reads; the result:
 * cands (the kept keypoints, OpenCV's order), desc, counts[2]. */
typedef struct pano_sift_args {
    const uint8_t *frame;      /* dev uint8 [h][w][3]                                   */
    uint8_t *frame_copy;       /* dev, same size: the graph's own frame buffer, or NULL  */
    int32_t h, w, n_octaves, n_layers;
    const float *taps;         /* host: the n_layers + 3 kernels of pano_scale_space     */
    const int32_t *ntaps;      /* host [n_layers + 3]                                    */
    float *const *gauss;       /* host [n_octaves] of dev float [n_layers + 3][rows][cols] */
    float *const *dog;         /* host [n_octaves] of dev float [n_layers + 2][rows][cols] */
    float *work;               /* dev, 5 h w floats                                      */
    int32_t detect;            /* 0 = the scale space only                               */
    float contrast_thr, edge_thr, sigma;
    int32_t first_octave, max_keypoints;
    const float *const *gauss_dev;   /* dev [n_octaves]: the pointers of `gauss`         */
    const int32_t *dims_dev;         /* dev [n_octaves][2]: rows, cols                   */
    pano_sift_keypoint *cands, *kpts;   /* dev [max_keypoints] each                      */
    int32_t *counts;           /* dev int32 [3]: candidates, keypoints, kept             */
    void *sort_work;           /* dev, pano_sift_sort_work_bytes(max_keypoints)          */
    float *desc;               /* dev float [max_keypoints][128]                         */
} pano_sift_args;
int pano_sift_detect(pano_ctx *ctx, const pano_sift_args *args);
/* 1 when the most recently used set of buffers of pano_sift_detect is replayed as a graph */
int pano_sift_detect_replaying(const pano_ctx *ctx);

/* The two nearest rows of `train` for every row of `query` (Euclidean), the search behind
 * flann_matching                                                  features.py:222-232
 * (cv2.FlannBasedMatcher().knnMatch(des1, des2, k=2): FLANN's randomised kd-trees give an
 * approximate answer; this search is exhaustive and its answer exact).  The cross terms of
 * |q - t|^2 run on the matrix cores in split float16 and only rank the candidates; the
 * four best per query are re-evaluated in float32 (sum of squared differences) and a bound
 * on the ranking error proves that no other row can be among the best two - a query whose
 * proof fails is rescanned exactly (counted in *rescans, optional dev int).
 * query: dev float [nq][d], train: dev float [nt][d], nt >= 2, d <= 128; scale: a power of
 * two with max |value| * scale <= 2048 (keeps the float16 halves normal); work: dev scratch
 * of pano_knn2_work_bytes(nq, nt, d) bytes.  idx: dev int32 [nq][2], dist: dev float [nq][2],
 * nearest first (equal distances: lower index first). */
size_t pano_knn2_work_bytes(int nq, int nt, int d);
int pano_knn2(pano_ctx *ctx, const float *query, int nq, const float *train, int nt, int d,
              float scale, void *work, int32_t *idx, float *dist, int *rescans);

/* One multiband stitch of the mosaic columns [xs0, xs1), queued by ONE call
 *                                                  stitcher.py:283-327 (equalize and crop aside)
 * = pano_ownership_cameras, pano_owned_regions (+ its copy to the host), pano_interior_map,
 * the one wait of a stitch, pano_layout_windows / pano_layout_place, the upload of the record
 * table, [pano_blur_tiles,] pano_warp_windows, pano_multiband_blur, pano_multiband_compose in
 * that order on the context's stream: the launch sequence of a stitch without a round trip
 * through the caller's language per launch (a dozen ctypes calls cost 0.3 ms per stitch,
 * as much as one GPU's share of the kernels when eight GPUs split a 4K panorama).
 * The caller owns every buffer; `args` says where they are and how large:
 *   cams               dev [n] records (frames NULL where not resident, see have)
 *   rects, have        host int32 [n][4] = (y0, y1, x0, x1) / host uint8 [n] (or NULL)
 *   [own0, own1)       columns to evaluate ownership on: the strip grown by what the interior
 *                      test and the windows look at (the whole mosaic: 0, W)
 *   taps, ntaps        host tap tables of the n_levels - 1 Gaussians (layout: pano_multiband_blur)
 *   shortcut           1 = interior map on (pano_interior_map, the compose's interior pixels)
 *   warp_need          1 = warp only the blocks anything reads (pano_blur_tiles), 0 = all of V,
 *                      -1 = decide by the mean width of the rectangles A (>= 768 columns)
 *   owner, valid       dev int16 / uint8 [H][W]: results
 *   marks, regions     dev workspaces of pano_owned_regions; regions_host: PINNED host copy
 *   block_owner, interior   dev workspaces of pano_interior_map
 *   records_host       PINNED host pano_patch [cap_records], cap_records >= n * max_spans
 *   table              dev pano_patch [cap_records]
 *   planes, blurred, scratch (+ their sizes in floats), tile_flags, need (cap_tiles bytes each)
 *   mosaic (+ optional mosaic_f32)   dev [H][W][3]: only columns [xs0, xs1) are written
 * Returns PANO_OK; PANO_EGROW (positive, not an error) when an arena or the tile arrays are too
 * small for this stitch: args->layout then says what is needed, everything up to the wait has
 * been queued, and the call is repeated with resume = 1 after the caller has grown them; or a
 * negative error (args->layout.missing > 0: a needed frame is not resident). */
#define PANO_EGROW 1
typedef struct pano_stitch_args {
    const pano_camera *cams;
    const int32_t *rects;
    const uint8_t *have;
    const double *sin_t, *cos_t, *tan_p;
    const float *lut;
    const float *taps;
    const int32_t *ntaps;
    int16_t *owner;
    uint8_t *valid;
    uint8_t *marks;
    int32_t *regions;
    int32_t *regions_host;
    int16_t *block_owner;
    uint8_t *interior;
    pano_patch *records_host;
    pano_patch *table;
    float *planes, *blurred, *scratch;
    uint8_t *tile_flags, *need;
    uint8_t *mosaic;
    float *mosaic_f32;
    int64_t planes_floats, blurred_floats, scratch_floats;
    int32_t n, H, W, xs0, xs1, own0, own1;
    int32_t lut_stride, n_levels, radius, shortcut, warp_need, max_spans, min_gap;
    int32_t cap_records, cap_tiles;
    int32_t used_need;         /* out: 1 = the warp ran on the need flags */
    int32_t trust_layout;      /* in: 1 = the caller vouches that cameras (matrices, rectangles),
                                * strip and resident frames are those of this context's previous
                                * stitch (with PANO_OPT_STITCH_ASYNC on): the stitch is queued with
                                * that stitch's verified layout and NOBODY WAITS - the call returns
                                * when everything is queued.  out: 2 = it did, 0 = it took the
                                * waiting path (first stitch of a shape, option off, ...).  Every
                                * kernel still runs; what is trusted is only that the same cameras
                                * give the same layout, which pano_stitch_verify checks.
                                * in: 3 = the same promise, and the geometry may be KEPT: the owner
                                * map, valid mask, interior map, record table, tile flags and the
                                * context's work list are functions of exactly what the caller
                                * vouches for; when the buffers are the previous stitch's (owner,
                                * valid, table, interior, tile_flags, need, the arenas, sin_t - same
                                * pointers) and no other call has entered the context since, only
                                * the warp, the blur and the collapse are queued: out 4.  Otherwise
                                * as 1.  The reference recomputes all of it per stitch
                                * (stitcher.py:276-306, 196-204) - to the same values. */
    uint8_t *classes;          /* dev workspace of pano_interior_classes (the interior map's shape),
                                * or NULL: no level classes */
    pano_layout layout;        /* out */
} pano_stitch_args;
int pano_stitch_multiband(pano_ctx *ctx, pano_stitch_args *args, int resume);
/* Compares the layout summary of the last trusted stitch with the verified layout it was queued
 * with (waits for that stitch's layout kernel); PANO_EINVAL when they differ - the promise of
 * args->trust_layout did not hold and the trusted mosaics since the last check are void.  A no-op
 * without trusted stitches pending; the next untrusted pano_stitch_multiband calls it itself. */
int pano_stitch_verify(pano_ctx *ctx);
/* How many stitches of this context went through on the device-side layout
 * (PANO_OPT_STITCH_ASYNC) and how many of those attempts fell back to the host layout. */
int pano_stitch_counts(const pano_ctx *ctx, int *device_layouts, int *fallbacks);

#ifdef __cplusplus
}
#endif
#endif /* PANO360_H */
