// Shared host/device helpers for libpano360_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/pano360.h"

// ---- error plumbing --------------------------------------------------------
void pano_set_error(const char *fmt, ...);

#define PANO_REQUIRE(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            pano_set_error(__VA_ARGS__); \
            return PANO_EINVAL;          \
        }                                \
    } while (0)

#define PANO_HIP(call)                                                         \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) {                                                \
            pano_set_error("%s failed: %s", #call, hipGetErrorString(e_));     \
            return PANO_EHIP;                                                  \
        }                                                                      \
    } while (0)

#define PANO_LAUNCH_CHECK(name)                                                \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            pano_set_error("launch of %s failed: %s", name,                    \
                           hipGetErrorString(e_));                             \
            return PANO_EHIP;                                                  \
        }                                                                      \
    } while (0)

// ---- per-kernel timing (HIP events on the launch stream; off by default) ----
enum PanoKernelId {
    PK_ADD_WEIGHTS = 0,
    PK_WARP,
    PK_OWNERSHIP,
    PK_BLUR_ROWS,
    PK_BLUR_COLS,
    PK_COMPOSE,
    PK_LINEAR,
    PK_NOBLEND,
    PK_CROP_HEIGHTS,
    PK_CROP_ROWS,
    PK_PYR_DOWN,
    PK_OWNERSHIP_CAMS,
    PK_OWNED_BOXES,
    PK_WARP_WINDOWS,
    PK_BLEND_CAMERAS,
    PK_OWNED_SPANS,
    PK_INTERIOR,
    PK_TILE_FLAGS,
    PK_OVERLAP,
    PK_BLUR_MFMA,
    PK_SIFT_EXTREMA,
    PK_SIFT_ORIENT,
    PK_SIFT_DESCRIBE,
    PK_COMPOSE_INTERIOR,
    PK_SCALE_STEP,
    PK_KNN2,
    PK_BLUR_LEAN,
    PK_BLUR_LEAN5,
    PK_COUNT
};
// ---- the context (include/pano360.h: pano_ctx) ------------------------------------
// Everything the library remembers between calls lives here: device and stream, the
// option switches, the timing registry, the blur's work-list buffers and the operand
// tables of the tap sets it has seen.  One context per host thread; nothing is shared
// between contexts.
struct PanoTapSet {                 // one set of Gaussian apertures (pano_multiband_blur)
    uint64_t key;                   // hash of the apertures and the tap values
    int n, ntaps[PANO_MAX_LEVELS];
    float *taps;                    // dev: the caller's padded tables, back to back
    unsigned char *tables;          // dev: matrix-core operand tables (built on first use)
    uint64_t used;                  // last use (eviction order)
    float *host;                    // the values themselves: a hash hit is confirmed by comparing them
    size_t floats;
    hipEvent_t ready;               // recorded behind the upload and the table build
    hipStream_t built_on;           // the stream those were queued on
};

struct LayoutSummary;               // layout.h
struct PanoSiftGraph {               // detect.hip: one captured launch sequence of pano_sift_detect
    uint64_t key;                   // hash of sizes, switches, buffer addresses, tap values
    hipGraphExec_t exec;            // null: not captured (yet, or the capture failed)
    int seen;                       // 0 new, 1 ran launch by launch once, 2 capture attempted
    uint64_t used;                  // last use (eviction order)
};

#define GEOM_BUFS 11
#define STITCH_SIG 13             // stitch.hip: stitch_signature
struct pano_ctx {
    int device;
    hipStream_t stream;
    int opt[PANO_OPT_COUNT];
    bool timing_on;
    std::vector<hipEvent_t> t_begin[PK_COUNT], t_end[PK_COUNT];
    // matrix-core blur: work list (unsorted, sorted), its counter, and whose list / tile
    // flags the buffers currently hold
    int2 *item_buf;
    int *item_counter;
    int blur_cm;                    // reach (k-steps) of the blur levels the next work list is for; 0 = unknown
    int item_cap;
    const pano_patch *prepared_table, *flags_table;
    int prepared_n, flags_n;
    // whose work list item_buf holds (stays when a blur has consumed `prepared_table`; an option
    // switch or a re-allocation clears it): what a kept-geometry repeat may re-use
    const pano_patch *list_table;
    int list_n;
    std::vector<PanoTapSet> tap_sets;
    uint64_t tick;
    // pano_stitch_multiband: the regions' copy has landed / the record table has left the
    // caller's pinned buffer
    hipEvent_t ev_regions, ev_upload, ev_fork, ev_join, ev_copy;
    hipStream_t side;               // second stream of pano_stitch_multiband
    bool upload_pending;
    // pano_stitch_multiband without its host round trip (stitch.hip): the device-side layout's
    // summary (device / pinned host), device copies of the patch rectangles and resident flags
    // (and the host values they were made from), and what the previous stitch's layout needed:
    // the next one's launch bounds
    LayoutSummary *lay_sum_host;    // pinned: the layout kernel writes it, the host reads it
    int32_t *lay_rects_dev;
    uint8_t *lay_have_dev;
    int lay_cap_n;
    std::vector<int32_t> lay_rects_host;
    std::vector<uint8_t> lay_have_host;
    pano_layout lay_prev;
    int lay_prev_sig[STITCH_SIG];
    bool lay_prev_valid;
    bool lay_prev_verified;         // lay_prev was read back (device summary) or made on the host
    bool trusted_pending;           // a trusted stitch's summary has not been compared yet
    int lay_prev_used_need;
    // kept geometry (args->trust_layout = 3): the previous stitch of this context went through whole
    // with these buffers and no other call has entered the context since (pano_ctx_enter clears it)
    bool geom_valid, in_stitch;
    const void *geom_bufs[GEOM_BUFS];
    int lay_count[2];               // stitches that went through on the device layout / fell back
    // pano_sift_extrema: the list of scale-space extrema between its two kernels (+ its counter)
    uint32_t *sift_raw;
    size_t sift_raw_cap;
    // pano_sift_detect: the captured launch sequences, one per set of buffers (detect.hip)
    std::vector<PanoSiftGraph> sift_graphs;
};

int pano_ctx_enter(pano_ctx *ctx);
void pano_sift_graphs_free(pano_ctx *ctx);   // detect.hip
int pano_zero_i32(hipStream_t s, int *p, int n);   // detect.hip: p[0 .. n) = 0, as a kernel
int pano_ctx_side_stream(pano_ctx *ctx);     // makes ctx->side and the fork / join events
// Device copy of a host tap table set (and, with `tables`, its matrix-core operand tables'
// buffer, `table_bytes` long, `*fresh` = it was just allocated and must be filled).
int pano_ctx_tap_set(pano_ctx *ctx, const float *taps, const int *ntaps, int n, size_t table_bytes,
                     PanoTapSet **out, bool *fresh);
// After the caller has queued the fill of a `fresh` table buffer on ctx->stream: uses from other
// streams wait for it.
int pano_ctx_tap_set_built(pano_ctx *ctx, PanoTapSet *set);

#define PANO_ENTER(ctx, who)                                       \
    PANO_REQUIRE((ctx) != nullptr, "%s: null context", who);       \
    if (int rc_ = pano_ctx_enter(ctx)) return rc_;                 \
    void *const stream = (void *)(ctx)->stream;                    \
    (void)stream

// An entry point that only READS what a stitch left behind (the crop reads the valid mask): the
// kept geometry (stitch.hip) survives it.
#define PANO_ENTER_READONLY(ctx, who)                              \
    const bool geom_keep_ = (ctx) != nullptr && (ctx)->geom_valid; \
    PANO_ENTER(ctx, who);                                          \
    (ctx)->geom_valid = geom_keep_

void pano_timing_edge(pano_ctx *ctx, int kid, hipStream_t stream, bool begin);

// Launch `...` on `stream`; when timing is enabled bracket it with events.
#define PANO_TIMED(kid, stream, ...)                                        \
    do {                                                                    \
        if (ctx->timing_on) pano_timing_edge(ctx, kid, stream, true);       \
        __VA_ARGS__;                                                        \
        if (ctx->timing_on) pano_timing_edge(ctx, kid, stream, false);      \
    } while (0)

static inline int pano_pitch_of(int w) { return (w + 3) & ~3; }

// blur_mfma.hip: all multiband levels of all records on the matrix cores
int pano_launch_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                          const int16_t *owner, int W, const float *taps, const int *ntaps,
                          int n_blur, const uint8_t *interior, uint8_t *tile_flags);
int pano_prepare_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                           int W, const uint8_t *interior, uint8_t *tile_flags);
int pano_tiles_blur_mfma(pano_ctx *ctx, const pano_patch *table, int n, int max_aw, int max_ah,
                         int W, int radius, const uint8_t *interior, uint8_t *tile_flags,
                         uint8_t *warp_need);
int pano_blur_mfma_opt_in(void);
int pano_blur_valu_opt_in(void);
static inline bool pano_blur_uses_mfma(const pano_ctx *ctx) {
    return ctx->opt[PANO_OPT_BLUR_KERNEL] == PANO_BLUR_MFMA;
}
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- device helpers --------------------------------------------------------
// Scalar-cache view of read-only global memory: loads through this pointer are
// selected as s_load (SGPR operands for the FMA chains of the blur kernels).
typedef const __attribute__((address_space(4))) float *kptr_f32;

// cv2.BORDER_REFLECT      ... c b a | a b c ... z | z y x ...
__device__ __forceinline__ int reflect_edge(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    int per = 2 * len;
    int m = p % per;
    if (m < 0) m += per;
    return m < len ? m : per - 1 - m;
}

// cv2.BORDER_REFLECT_101  ... c b | a b c ... z | y x ...   (len 1 -> 0)
__device__ __forceinline__ int reflect_101(int p, int len) {
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    int per = 2 * len - 2;
    int m = p % per;
    if (m < 0) m += per;
    return m < len ? m : per - m;
}

// cvRound(float) the way x86 cvtss2si does it: ties to even, and the
// "integer indefinite" 0x80000000 for NaN or anything outside int32.
__device__ __forceinline__ int cv_round(float v) {
    const float r = rintf(v);
    // (one compare: |r| < 2^31 fails for NaN, for everything outside int32 and for -2^31 itself,
    // whose conversion is that same bit pattern)
    return fabsf(r) < 2147483648.0f ? (int)r : (int)0x80000000;
}

__device__ __forceinline__ int sat16(int v) {
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}
