// Library-level entry points: version, error string, device probe.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "layout.h"

static thread_local char g_error[512] = "";

void pano_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *pano_version(void) { return "pano360_hip 0.1 (gfx950)"; }

extern "C" const char *pano_last_error(void) { return g_error; }

extern "C" int pano_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int pano_pitch(int w) { return pano_pitch_of(w); }

extern "C" int pano_interior_block(void) { return PANO_INTERIOR_BLOCK; }

// ---- the context ---------------------------------------------------------------------
extern "C" int pano_ctx_create(int device, void *stream, pano_ctx **out) {
    PANO_REQUIRE(out, "pano_ctx_create: null output");
    *out = nullptr;
    PANO_REQUIRE(device >= 0 && device < pano_device_count(), "pano_ctx_create: no device %d",
                 device);
    PANO_HIP(hipSetDevice(device));
    // kernels whose tiles need more than the default 64 KiB of LDS (per device, idempotent)
    if (int rc = pano_blur_mfma_opt_in()) return rc;
    if (int rc = pano_blur_valu_opt_in()) return rc;
    pano_ctx *ctx = new pano_ctx();
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    ctx->opt[PANO_OPT_BLUR_KERNEL] = PANO_BLUR_MFMA;
    ctx->opt[PANO_OPT_OWN_PRUNE] = 1;
    ctx->opt[PANO_OPT_BLUR_SEGMENTS] = 1;
    ctx->opt[PANO_OPT_BLUR_LEAN] = 1;
    ctx->opt[PANO_OPT_STITCH_STREAMS] = 1;
    ctx->opt[PANO_OPT_STITCH_ASYNC] = 0;
    ctx->opt[PANO_OPT_BLUR_SEG_LEN] = 0;
    ctx->opt[PANO_OPT_SIFT_GRAPH] = 1;
    ctx->opt[PANO_OPT_LEVEL_CLASSES] = 0;
    *out = ctx;
    return PANO_OK;
}

int pano_ctx_enter(pano_ctx *ctx) {
    // any call but the stitch's own may write the buffers a stitch left its geometry in
    if (!ctx->in_stitch) ctx->geom_valid = false;
    PANO_HIP(hipSetDevice(ctx->device));
    return PANO_OK;
}

static void timing_clear(pano_ctx *ctx) {
    for (int k = 0; k < PK_COUNT; ++k) {
        for (hipEvent_t e : ctx->t_begin[k]) (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->t_end[k]) (void)hipEventDestroy(e);
        ctx->t_begin[k].clear();
        ctx->t_end[k].clear();
    }
}

static void tap_set_free(PanoTapSet &ts) {
    if (ts.taps) (void)hipFree(ts.taps);
    if (ts.tables) (void)hipFree(ts.tables);
    if (ts.ready) (void)hipEventDestroy(ts.ready);
    free(ts.host);
    ts.taps = nullptr;
    ts.tables = nullptr;
    ts.ready = nullptr;
    ts.host = nullptr;
}

extern "C" int pano_ctx_destroy(pano_ctx *ctx) {
    if (!ctx) return PANO_OK;
    PANO_HIP(hipSetDevice(ctx->device));
    // queued kernels may still read the context's tables; the stream may be gone already
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();
    }
    timing_clear(ctx);
    for (PanoTapSet &ts : ctx->tap_sets) tap_set_free(ts);
    if (ctx->item_buf) (void)hipFree(ctx->item_buf);
    if (ctx->item_counter) (void)hipFree(ctx->item_counter);
    if (ctx->sift_raw) (void)hipFree(ctx->sift_raw);
    pano_sift_graphs_free(ctx);
    if (ctx->lay_sum_host) (void)hipHostFree(ctx->lay_sum_host);
    if (ctx->lay_rects_dev) (void)hipFree(ctx->lay_rects_dev);
    if (ctx->lay_have_dev) (void)hipFree(ctx->lay_have_dev);
    if (ctx->side) {
        (void)hipStreamSynchronize(ctx->side);
        (void)hipStreamDestroy(ctx->side);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_regions) (void)hipEventDestroy(ctx->ev_regions);
    if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    delete ctx;
    return PANO_OK;
}

// The context's second stream and the events that fork work onto it and join it (made on
// first use: pano_stitch_multiband, and the fifth / sixth level of pano_multiband_blur).
int pano_ctx_side_stream(pano_ctx *ctx) {
    if (!ctx->ev_regions) PANO_HIP(hipEventCreateWithFlags(&ctx->ev_regions, hipEventDisableTiming));
    if (!ctx->ev_upload) PANO_HIP(hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    if (!ctx->ev_fork) PANO_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->ev_join) PANO_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    if (!ctx->ev_copy) PANO_HIP(hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming));
    if (!ctx->side) PANO_HIP(hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
    return PANO_OK;
}

extern "C" int pano_ctx_set_stream(pano_ctx *ctx, void *stream) {
    PANO_REQUIRE(ctx, "pano_ctx_set_stream: null context");
    // what the previous stitch left behind was produced in the OTHER stream's order: a kept-geometry
    // repeat (stitch.hip) would read its owner map, table and flags with nothing ordering the two
    if ((hipStream_t)stream != ctx->stream) ctx->geom_valid = false;
    ctx->stream = (hipStream_t)stream;
    return PANO_OK;
}

extern "C" int pano_ctx_set_option(pano_ctx *ctx, int option, int value) {
    PANO_REQUIRE(ctx, "pano_ctx_set_option: null context");
    PANO_REQUIRE(option >= 0 && option < PANO_OPT_COUNT, "pano_ctx_set_option: option %d", option);
    if (option == PANO_OPT_BLUR_KERNEL)
        PANO_REQUIRE(value == PANO_BLUR_MFMA || value == PANO_BLUR_VALU,
                     "pano_ctx_set_option: blur kernel %d", value);
    else if (option == PANO_OPT_STITCH_ASYNC)
        PANO_REQUIRE(value >= 0 && value <= 2, "pano_ctx_set_option: option %d takes 0, 1 or 2", option);
    else if (option == PANO_OPT_OWN_PRUNE)
        PANO_REQUIRE(value >= 0 && value <= 3, "pano_ctx_set_option: option %d takes 0 .. 3", option);
    else if (option == PANO_OPT_BLUR_SEG_LEN)
        PANO_REQUIRE(value == -1 || value == 0 || (value >= 4 && value <= 2047),
                     "pano_ctx_set_option: option %d takes -1, 0 or 4 .. 2047", option);
    else
        PANO_REQUIRE(value == 0 || value == 1, "pano_ctx_set_option: option %d takes 0 or 1", option);
    const bool changed = ctx->opt[option] != value;
    ctx->opt[option] = value;
    // tile flags / work lists made for the other tile grid are void
    ctx->prepared_table = ctx->flags_table = ctx->list_table = nullptr;
    // ... and so is everything a stitch left for a kept-geometry repeat (stitch.hip): the record
    // table, the flags and the work list were laid out for the options as they were.  A changed
    // blur kernel changes the tile grid: the previous layout's bounds are another grid's too.
    ctx->geom_valid = false;
    if (changed && option == PANO_OPT_BLUR_KERNEL)
        ctx->lay_prev_valid = ctx->lay_prev_verified = false;
    return PANO_OK;
}

extern "C" int pano_ctx_get_option(const pano_ctx *ctx, int option, int *value) {
    PANO_REQUIRE(ctx && value, "pano_ctx_get_option: null pointer");
    PANO_REQUIRE(option >= 0 && option < PANO_OPT_COUNT, "pano_ctx_get_option: option %d", option);
    *value = ctx->opt[option];
    return PANO_OK;
}

// Tap sets are keyed on their values: a caller may rebuild its host tables, or free and
// reuse their memory, at will.  FNV-1a over the apertures and the float bits.
static uint64_t tap_hash(const float *taps, const int *ntaps, int n, size_t floats) {
    uint64_t h = 1469598103934665603ull;
    auto eat = [&](const void *p, size_t bytes) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    };
    eat(&n, sizeof(n));
    eat(ntaps, n * sizeof(int));
    eat(taps, floats * sizeof(float));
    return h;
}

#define PANO_TAP_SETS_MAX 64
int pano_ctx_tap_set(pano_ctx *ctx, const float *taps, const int *ntaps, int n, size_t table_bytes,
                     PanoTapSet **out, bool *fresh) {
    size_t floats = 0;
    for (int k = 0; k < n; ++k) floats += (size_t)ntaps[k] + PANO_TAP_PAD;
    const uint64_t key = tap_hash(taps, ntaps, n, floats);
    *fresh = false;
    for (PanoTapSet &ts : ctx->tap_sets)
        if (ts.key == key && ts.n == n && !memcmp(ts.ntaps, ntaps, n * sizeof(int)) &&
            ts.floats == floats && !memcmp(ts.host, taps, floats * sizeof(float))) {
            ts.used = ++ctx->tick;
            // the upload (and the table build) were queued on the stream the context targeted
            // then: a use from another stream is ordered behind them
            if (ctx->stream != ts.built_on) PANO_HIP(hipStreamWaitEvent(ctx->stream, ts.ready, 0));
            if (table_bytes && !ts.tables) {
                PANO_HIP(hipMalloc((void **)&ts.tables, table_bytes));
                *fresh = true;
            }
            *out = &ts;
            return PANO_OK;
        }
    if (ctx->tap_sets.size() >= PANO_TAP_SETS_MAX) {        // drop the least recently used one
        size_t old = 0;
        for (size_t i = 1; i < ctx->tap_sets.size(); ++i)
            if (ctx->tap_sets[i].used < ctx->tap_sets[old].used) old = i;
        // kernels on ANY stream this context has targeted may still read its tables
        PANO_HIP(hipDeviceSynchronize());
        tap_set_free(ctx->tap_sets[old]);
        ctx->tap_sets.erase(ctx->tap_sets.begin() + old);
    }
    PanoTapSet ts = {};
    ts.key = key;
    ts.n = n;
    memcpy(ts.ntaps, ntaps, n * sizeof(int));
    ts.floats = floats;
    ts.host = (float *)malloc(floats * sizeof(float));
    PANO_REQUIRE(ts.host, "pano_ctx_tap_set: out of host memory");
    memcpy(ts.host, taps, floats * sizeof(float));
    PANO_HIP(hipMalloc((void **)&ts.taps, floats * sizeof(float)));
    // pageable source: the runtime stages it before returning, the caller's table is free again
    PANO_HIP(hipMemcpyAsync(ts.taps, taps, floats * sizeof(float), hipMemcpyHostToDevice,
                            ctx->stream));
    PANO_HIP(hipEventCreateWithFlags(&ts.ready, hipEventDisableTiming));
    PANO_HIP(hipEventRecord(ts.ready, ctx->stream));
    ts.built_on = ctx->stream;
    if (table_bytes) {
        PANO_HIP(hipMalloc((void **)&ts.tables, table_bytes));
        *fresh = true;
    }
    ts.used = ++ctx->tick;
    ctx->tap_sets.push_back(ts);
    *out = &ctx->tap_sets.back();
    return PANO_OK;
}

int pano_ctx_tap_set_built(pano_ctx *ctx, PanoTapSet *set) {
    // the fill was queued on ctx->stream behind a wait for the previous `ready` (if that was
    // another stream's), so the new record covers the upload too
    PANO_HIP(hipEventRecord(set->ready, ctx->stream));
    set->built_on = ctx->stream;
    return PANO_OK;
}

// ---- per-kernel timing -------------------------------------------------------
// bench.py measures the dominant kernel's launch durations live with HIP events
// recorded on the stream the kernel is launched on.  Instrumentation only: off
// unless pano_timing_enable(ctx, 1) was called.
static const char *const g_kernel_names[PK_COUNT] = {
    "add_weights_kernel", "warp_spherical_kernel", "ownership_kernel", "blur_rows_kernel",
    "blur_cols_kernel",   "multiband_compose_kernel", "linear_blend_kernel",
    "no_blend_kernel",    "crop_heights_kernel", "crop_rows_kernel", "pyr_down_kernel",
    "ownership_cameras_kernel", "owned_boxes_kernel", "warp_windows_kernel",
    "blend_cameras_kernel", "owned_spans_kernel",
    "block_owner_kernel", "tile_flags_kernel", "overlap_stats_kernel",
    "blur_mfma_kernel", "sift_extrema_kernel", "sift_orient_kernel", "sift_describe_kernel",
    "compose_interior_kernel", "scale_step_kernel", "knn2_kernel", "blur_lean_kernel", "blur_lean5_kernel"};

void pano_timing_edge(pano_ctx *ctx, int kid, hipStream_t stream, bool begin) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, stream);
    (begin ? ctx->t_begin : ctx->t_end)[kid].push_back(ev);
}

extern "C" int pano_timing_enable(pano_ctx *ctx, int on) {
    PANO_REQUIRE(ctx, "pano_timing_enable: null context");
    timing_clear(ctx);
    ctx->timing_on = on != 0;
    return PANO_OK;
}

extern "C" int pano_kernel_count(void) { return PK_COUNT; }

extern "C" const char *pano_kernel_name(int kid) {
    return kid >= 0 && kid < PK_COUNT ? g_kernel_names[kid] : "";
}

extern "C" int pano_timing_read(pano_ctx *ctx, int kid, double *total_ms, int *launches) {
    PANO_REQUIRE(ctx && kid >= 0 && kid < PK_COUNT && total_ms && launches,
                 "pano_timing_read: bad argument");
    double sum = 0.0;
    const size_t nb = ctx->t_begin[kid].size(), ne = ctx->t_end[kid].size();
    const size_t n = ne < nb ? ne : nb;
    for (size_t i = 0; i < n; ++i) {
        PANO_HIP(hipEventSynchronize(ctx->t_end[kid][i]));
        float ms = 0.f;
        PANO_HIP(hipEventElapsedTime(&ms, ctx->t_begin[kid][i], ctx->t_end[kid][i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int)n;
    return PANO_OK;
}

// ---- window layout (host) ----------------------------------------------------------
// Between the region search and the warp the host sits on the critical path of a stitch:
// the GPU idles until the record table is uploaded.  This is that table's construction
// (include/pano360.h, "Windows"), one call instead of a hundred small array operations.
extern "C" int pano_layout_windows(int tile_grid, const int32_t *regions, int n, int max_spans,
                                   const int32_t *rects, const uint8_t *have, int radius,
                                   int xs0, int xs1, int n_blur, pano_patch *records, int cap,
                                   pano_layout *out) {
    PANO_REQUIRE(regions && rects && records && out, "pano_layout_windows: null pointer");
    PANO_REQUIRE(tile_grid == 0 || tile_grid == 32, "pano_layout_windows: tile grid %d", tile_grid);
    PANO_REQUIRE(n >= 0 && max_spans >= 1 && radius >= 0 && n_blur >= 0 && cap >= 0,
                 "pano_layout_windows: bad argument");
    const bool grid32 = tile_grid == 32;
    const int stride = 5 + 2 * max_spans;
    pano_layout lay = {};
    lay.missing = 0;
    long planes = 0, blurred = 0, scratch = 0, tiles = 0;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t *rg = regions + (size_t)i * stride;
        const int spans = rg[4] < max_spans ? rg[4] : max_spans;
        for (int sp = 0; sp < spans; ++sp) {
            pano_patch rec;
            LayoutSizes sz;
            if (!layout_record(rg, sp, rects + 4 * i, i, radius, xs0, xs1, n_blur, grid32, rec, sz))
                continue;
            PANO_REQUIRE(k < cap, "pano_layout_windows: more than %d records", cap);
            if (have && !have[i]) ++lay.missing;
            pano_patch &r = records[k++];
            r = rec;
            PANO_REQUIRE(layout_record_fits(r),
                         "pano_layout_windows: a plane of record %d exceeds 2 GiB", k - 1);
            // arena offsets in floats, turned into addresses by pano_layout_place
            r.planes = (float *)(uintptr_t)planes;
            r.blurred = (float *)(uintptr_t)(blurred + sz.lead);
            r.scratch = (float *)(uintptr_t)scratch;
            PANO_REQUIRE(tiles < (1l << 31), "pano_layout_windows: tile count overflow");
            r.tiles_off = (int)tiles;
            planes += sz.planes;
            blurred += sz.blurred;
            scratch += sz.scratch;
            tiles += sz.tiles;
            lay.max_vw = r.vw > lay.max_vw ? r.vw : lay.max_vw;
            lay.max_vh = r.vh > lay.max_vh ? r.vh : lay.max_vh;
            lay.max_aw = r.aw > lay.max_aw ? r.aw : lay.max_aw;
            lay.max_ah = r.ah > lay.max_ah ? r.ah : lay.max_ah;
        }
    }
    lay.n_records = k;
    lay.n_tiles = (int)tiles;
    lay.planes_floats = planes;
    lay.blurred_floats = blurred + 32;
    lay.scratch_floats = scratch;
    *out = lay;
    return PANO_OK;
}

extern "C" int pano_layout_place(pano_patch *records, int n_records, void *planes, void *blurred,
                                 void *scratch) {
    PANO_REQUIRE(records || n_records == 0, "pano_layout_place: null pointer");
    uintptr_t bbase = (uintptr_t)blurred;
    bbase += (uintptr_t)(-(intptr_t)bbase) % 128;
    for (int k = 0; k < n_records; ++k) {
        pano_patch &r = records[k];
        r.planes = (float *)planes + (uintptr_t)r.planes;
        r.blurred = blurred ? (float *)bbase + (uintptr_t)r.blurred : nullptr;
        r.scratch = scratch ? (float *)scratch + (uintptr_t)r.scratch : nullptr;
    }
    return PANO_OK;
}
