// One multiband stitch per native call: the launch sequence of Engine.multiband_fused
// (stitcher.py:283-327 without equalize / crop) queued from C++.  No kernel of its own.
#include "common.h"

// Runs `call` with the context targeted at its side stream.
#define ON_SIDE(ctx, call)                  \
    do {                                    \
        const hipStream_t main_ = (ctx)->stream; \
        (ctx)->stream = (ctx)->side;        \
        const int rc_side_ = (call);        \
        (ctx)->stream = main_;              \
        if (rc_side_) return rc_side_;      \
    } while (0)

extern "C" int pano_stitch_multiband(pano_ctx *ctx, pano_stitch_args *a, int resume) {
    PANO_ENTER(ctx, "pano_stitch_multiband");
    PANO_REQUIRE(a, "pano_stitch_multiband: null arguments");
    PANO_REQUIRE(a->cams && a->rects && a->sin_t && a->cos_t && a->tan_p && a->lut && a->owner &&
                     a->valid && a->marks && a->regions && a->regions_host && a->records_host &&
                     a->table && a->mosaic,
                 "pano_stitch_multiband: null pointer");
    PANO_REQUIRE(a->n > 0 && a->n <= 32767 && a->H > 0 && a->W > 0 && a->max_spans >= 1,
                 "pano_stitch_multiband: bad argument");
    PANO_REQUIRE(a->n_levels >= 1 && a->n_levels <= PANO_MAX_LEVELS,
                 "pano_stitch_multiband: n_levels %d outside [1, %d]", a->n_levels, PANO_MAX_LEVELS);
    PANO_REQUIRE(a->own0 <= a->xs0 && a->xs1 <= a->own1 && a->own0 >= 0 && a->own1 <= a->W,
                 "pano_stitch_multiband: strip [%d, %d) outside the ownership columns [%d, %d)",
                 a->xs0, a->xs1, a->own0, a->own1);
    PANO_REQUIRE(a->cap_records >= a->n * a->max_spans,
                 "pano_stitch_multiband: %d record slots for %d cameras x %d spans", a->cap_records,
                 a->n, a->max_spans);
    const int n_blur = a->n_levels - 1;
    PANO_REQUIRE(n_blur == 0 || (a->taps && a->ntaps), "pano_stitch_multiband: no tap tables");
    const bool interior = a->shortcut && n_blur > 0;
    PANO_REQUIRE(!interior || (a->block_owner && a->interior),
                 "pano_stitch_multiband: interior map without its buffers");
    const hipStream_t s = (hipStream_t)stream;
    if (int rc = pano_ctx_side_stream(ctx)) return rc;
    // the second stream pays on large mosaics only (config 3: 1.975 -> 1.945 ms per stitch;
    // config 2, 7.9 MP: 0.527 -> 0.537: the forks and joins cost more than the overlap gives)
    const bool big = (long long)a->H * (a->own1 - a->own0) >= (1ll << 24);
    const bool two_streams = ctx->opt[PANO_OPT_STITCH_STREAMS] != 0 && big;
    const int tile_grid = pano_blur_tile_grid(ctx);
    const int stride = 5 + 2 * a->max_spans;

    if (!resume) {
        // ownership and, in the same pass, one record per (camera, span of columns it owns):
        // boxes and column marks from the ownership kernel, the spans' search, its copy to the host
        // (in one pass on mosaics of 16 MP and more: config 5 16.12 -> 15.99 ms, config 3 even;
        // on the 8 MP of config 2 the longer ownership kernel costs what the box kernel saved)
        if (big) {
            if (int rc = pano_ownership_regions(ctx, a->cams, a->n, a->H, a->W, a->own0, a->own1,
                                                a->sin_t, a->cos_t, a->tan_p, a->owner, a->valid,
                                                a->min_gap, a->max_spans, a->marks, a->regions))
                return rc;
        } else {
            if (int rc = pano_ownership_cameras(ctx, a->cams, a->n, a->H, a->W, a->own0, a->own1,
                                                a->sin_t, a->cos_t, a->tan_p, a->owner, a->valid))
                return rc;
            if (int rc = pano_owned_regions(ctx, a->owner, a->H, a->W, a->own0, a->own1, a->n,
                                            a->min_gap, a->max_spans, a->marks, a->regions))
                return rc;
        }
        // Two small chains depend on the owner map only - the interior map and the region
        // search's tail - and two more on the record table only - the warp and the blur's tile
        // flags and work list: the context's side stream takes one of each pair (the short
        // kernels of a config-3 stitch were 0.15 ms of a 2.0 ms timeline, plus the gaps between
        // them).
        const bool forked = two_streams;
        if (forked && interior) {
            PANO_HIP(hipEventRecord(ctx->ev_fork, s));
            PANO_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            ON_SIDE(ctx, pano_interior_map(ctx, a->owner, a->H, a->W, a->own0, a->own1, a->radius,
                                           a->block_owner, a->interior));
        }
        PANO_HIP(hipMemcpyAsync(a->regions_host, a->regions, (size_t)a->n * stride * sizeof(int32_t),
                                hipMemcpyDeviceToHost, s));
        PANO_HIP(hipEventRecord(ctx->ev_regions, s));
        // the interior map needs the owner map only: queued before the wait, it keeps the GPU
        // busy while the host lays out the windows
        if (interior && !forked)
            if (int rc = pano_interior_map(ctx, a->owner, a->H, a->W, a->own0, a->own1, a->radius,
                                           a->block_owner, a->interior))
                return rc;
        PANO_HIP(hipEventSynchronize(ctx->ev_regions));          // the one wait of a stitch
        if (ctx->upload_pending) {                               // records_host may be in use
            PANO_HIP(hipEventSynchronize(ctx->ev_upload));
            ctx->upload_pending = false;
        }
        a->layout = pano_layout{};
        if (int rc = pano_layout_windows(tile_grid, a->regions_host, a->n, a->max_spans, a->rects,
                                         a->have, a->radius, a->xs0, a->xs1, n_blur,
                                         a->records_host, a->cap_records, &a->layout))
            return rc;
        if (a->layout.missing) {
            pano_set_error("pano_stitch_multiband: %d records of cameras whose frames are not "
                           "resident reach columns [%d, %d)", a->layout.missing, a->xs0, a->xs1);
            return PANO_EINVAL;
        }
    }
    const pano_layout &lay = a->layout;
    const bool use_blur = n_blur > 0 && lay.n_records > 0;
    if (lay.planes_floats > a->planes_floats || (use_blur && lay.blurred_floats > a->blurred_floats) ||
        (use_blur && lay.scratch_floats > a->scratch_floats) ||
        (interior && lay.n_tiles > a->cap_tiles))
        return PANO_EGROW;
    PANO_REQUIRE(lay.planes_floats == 0 || a->planes, "pano_stitch_multiband: no plane arena");
    if (int rc = pano_layout_place(a->records_host, lay.n_records, a->planes, a->blurred, a->scratch))
        return rc;
    const int nr = lay.n_records;
    if (nr) {
        PANO_HIP(hipMemcpyAsync(a->table, a->records_host, (size_t)nr * sizeof(pano_patch),
                                hipMemcpyHostToDevice, s));
        PANO_HIP(hipEventRecord(ctx->ev_upload, s));
        ctx->upload_pending = true;
    }
    // Warp only what is read: worth it when the rectangles are wide against the blur's reach
    // (8 x 1080p: -4 %; on 32 x 4K nearly every tile is within reach)
    a->used_need = 0;
    if (interior && tile_grid == 32 && nr && a->tile_flags && a->need) {
        bool on = a->warp_need == 1;
        if (a->warp_need < 0) {
            double sum = 0.0;
            for (int k = 0; k < nr; ++k) sum += a->records_host[k].aw;
            on = sum / nr >= 768.0;
        }
        if (on) {
            if (two_streams) {                                   // the interior map is the side stream's
                PANO_HIP(hipEventRecord(ctx->ev_join, ctx->side));
                PANO_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));
            }
            if (int rc = pano_blur_tiles(ctx, a->table, nr, lay.max_aw, lay.max_ah, a->W, a->radius,
                                         a->interior, a->tile_flags, a->need))
                return rc;
            a->used_need = 1;
        }
    }
    const bool fork2 = two_streams && interior && n_blur && nr &&
                       !a->used_need && tile_grid == 32 && a->tile_flags;
    if (fork2) {            // the blur's tile flags and sorted work list beside the warp
        PANO_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_upload, 0));      // the record table
        ON_SIDE(ctx, pano_multiband_blur_prepare(ctx, a->table, nr, lay.max_aw, lay.max_ah, a->W,
                                                 a->interior, a->tile_flags));
        PANO_HIP(hipEventRecord(ctx->ev_join, ctx->side));
    } else if (two_streams && interior) {
        PANO_HIP(hipEventRecord(ctx->ev_join, ctx->side));               // the interior map
    }
    if (int rc = pano_warp_windows(ctx, a->cams, a->table, nr, lay.max_vw, lay.max_vh, a->sin_t,
                                   a->cos_t, a->tan_p, a->lut, a->lut_stride,
                                   a->used_need ? a->need : nullptr))
        return rc;
    if (two_streams && interior) PANO_HIP(hipStreamWaitEvent(s, ctx->ev_join, 0));
    if (n_blur)
        if (int rc = pano_multiband_blur(ctx, a->table, nr, lay.max_aw, lay.max_vh, lay.max_ah,
                                         a->owner, a->W, a->taps, (const int *)a->ntaps, n_blur,
                                         interior ? a->interior : nullptr,
                                         interior ? a->tile_flags : nullptr))
            return rc;
    return pano_multiband_compose(ctx, a->table, nr, a->H, a->W, a->xs0, a->xs1, a->n_levels,
                                  a->owner, a->valid, interior ? a->interior : nullptr,
                                  interior ? a->cams : nullptr, interior ? a->sin_t : nullptr,
                                  interior ? a->cos_t : nullptr, interior ? a->tan_p : nullptr,
                                  a->lut, a->lut_stride, a->mosaic, a->mosaic_f32, 0);
}
