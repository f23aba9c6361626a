// One multiband stitch per native call: the launch sequence of Engine.multiband_fused
// (stitcher.py:283-327 without equalize / crop) queued from C++.  No kernel of its own.
#include <string.h>

#include "common.h"
#include "layout.h"


// ---- the window layout on the device --------------------------------------------------
// pano_layout_windows + pano_layout_place as a kernel (one workgroup): the record table goes
// from the owned regions straight into `table`, and the warp, blur and collapse can be queued
// behind it without the regions travelling to the host and the table back (0.1 ms per stitch
// during which the GPU had nothing to do).  Those launches are sized on the host by what the
// PREVIOUS layout needed (`b_*`, with slack); this kernel checks that the present one fits
// them and the arenas and otherwise empties the table - every consumer skips an empty record -
// and says so in the summary, which the host reads while the GPU works and answers with the
// host layout and a second round of launches.
#define LAY_THREADS 512
__global__ __launch_bounds__(LAY_THREADS) void layout_windows_kernel(
    const int32_t *__restrict__ regions, const int32_t *__restrict__ rects,
    const uint8_t *__restrict__ have, int n, int max_spans, int radius, int xs0, int xs1,
    int n_blur, int grid32, float *planes, float *blurred, float *scratch, long cap_planes,
    long cap_blurred, long cap_scratch, int cap_tiles, int want_tiles, int b_nr, int b_vw,
    int b_vh, int b_aw, int b_ah, pano_patch *__restrict__ table,
    LayoutSummary *__restrict__ summary, LayoutSummary expect, int check_expect,
    int *__restrict__ sticky) {
    __shared__ long s_planes[LAY_THREADS], s_blurred[LAY_THREADS], s_scratch[LAY_THREADS];
    __shared__ int s_tiles[LAY_THREADS], s_slot[LAY_THREADS];
    __shared__ long s_run[4];                       // planes, blurred, scratch, tiles so far
    __shared__ int s_count, s_max[4], s_missing, s_bad;
    const int tid = threadIdx.x, stride = 5 + 2 * max_spans;
    if (tid == 0) {
        s_run[0] = s_run[1] = s_run[2] = s_run[3] = 0;
        s_count = s_missing = s_bad = 0;
        s_max[0] = s_max[1] = s_max[2] = s_max[3] = 0;
    }
    __syncthreads();
    const int total = n * max_spans;
    for (int base = 0; base < total; base += LAY_THREADS) {
        const int c = base + tid, i = c / max_spans, sp = c - i * max_spans;
        pano_patch rec = pano_patch{};
        LayoutSizes sz = LayoutSizes{};
        bool ok = false;
        if (c < total) {
            const int32_t *rg = regions + (size_t)i * stride;
            const int spans = rg[4] < max_spans ? rg[4] : max_spans;
            if (sp < spans)
                ok = layout_record(rg, sp, rects + 4 * i, i, radius, xs0, xs1, n_blur, grid32 != 0,
                                   rec, sz);
        }
        s_slot[tid] = ok ? 1 : 0;
        s_planes[tid] = sz.planes;
        s_blurred[tid] = sz.blurred;
        s_scratch[tid] = sz.scratch;
        s_tiles[tid] = (int)(sz.tiles < (1l << 30) ? sz.tiles : (1l << 30));
        if (ok) {
            atomicMax(&s_max[0], rec.vw);
            atomicMax(&s_max[1], rec.vh);
            atomicMax(&s_max[2], rec.aw);
            atomicMax(&s_max[3], rec.ah);
            if (have && !have[i]) atomicAdd(&s_missing, 1);
            if (!layout_record_fits(rec)) atomicOr(&s_bad, 8);
        }
        __syncthreads();
        if (tid < 64) {
            // the chunk's offsets, in candidate order: wave 0 scans the slots 64 at a time
            // (a single thread walking them made this kernel 30 us long)
            long pl = s_run[0], bl = s_run[1], sc = s_run[2], tl = s_run[3];
            int k = s_count;
            const int cnt = min(LAY_THREADS, total - base);
            for (int t0 = 0; t0 < cnt; t0 += 64) {
                const int t = t0 + tid;
                const int has = s_slot[t];
                long v[4] = {has ? s_planes[t] : 0, has ? s_blurred[t] : 0, has ? s_scratch[t] : 0,
                             has ? (long)s_tiles[t] : 0};
                int one = has;
                const long own[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int o1 = __shfl_up(one, off, 64);
                    long o[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] = __shfl_up(v[q], off, 64);
                    if (tid >= off) {
                        one += o1;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += o[q];
                    }
                }
                s_slot[t] = has ? k + one - 1 : -1;
                s_planes[t] = pl + v[0] - own[0];
                s_blurred[t] = bl + v[1] - own[1];
                s_scratch[t] = sc + v[2] - own[2];
                const long toff = tl + v[3] - own[3];
                s_tiles[t] = (int)(toff < (1l << 30) ? toff : (1l << 30));
                k += __shfl(one, 63, 64);
                pl += __shfl(v[0], 63, 64);
                bl += __shfl(v[1], 63, 64);
                sc += __shfl(v[2], 63, 64);
                tl += __shfl(v[3], 63, 64);
            }
            for (int t = cnt + tid; t < LAY_THREADS; t += 64) s_slot[t] = -1;
            if (tid == 0) {
                s_run[0] = pl, s_run[1] = bl, s_run[2] = sc, s_run[3] = tl;
                s_count = k;
            }
        }
        __syncthreads();
        const int k = s_slot[tid];
        if (k >= 0 && k < b_nr) {
            rec.planes = planes + s_planes[tid];
            rec.blurred = blurred ? blurred + s_blurred[tid] + sz.lead : nullptr;
            rec.scratch = scratch ? scratch + s_scratch[tid] : nullptr;
            rec.tiles_off = s_tiles[tid];
            table[k] = rec;
        }
        __syncthreads();
    }
    const int nr = s_count;
    const long need_planes = s_run[0], need_blurred = s_run[1] + 32, need_scratch = s_run[2];
    int why = s_bad;
    if (need_planes > cap_planes || (n_blur > 0 && nr > 0 && (need_blurred > cap_blurred ||
                                                              need_scratch > cap_scratch)) ||
        (want_tiles && s_run[3] > cap_tiles))
        why |= 1;
    if (s_max[0] > b_vw || s_max[1] > b_vh || s_max[2] > b_aw || s_max[3] > b_ah) why |= 2;
    if (nr > b_nr) why |= 4;
    if (s_missing) why |= 16;
    // records the launches cover beyond the real ones - or all of them - are empty
    for (int k = (why ? 0 : nr) + tid; k < b_nr; k += LAY_THREADS) table[k] = pano_patch{};
    if (tid == 0) {
        LayoutSummary out;
        out.planes_floats = need_planes;
        out.blurred_floats = need_blurred;
        out.scratch_floats = need_scratch;
        out.n_records = nr;
        out.n_tiles = (int)(s_run[3] < (1l << 30) ? s_run[3] : (1l << 30));
        out.max_vw = s_max[0], out.max_vh = s_max[1], out.max_aw = s_max[2], out.max_ah = s_max[3];
        out.missing = s_missing;
        out.ok = why == 0;
        out.why = why;
        out.pad = 0;
        *summary = out;
        // a trusted stitch: ITS layout against the verified one it was queued with, kept in a
        // sticky word - the one summary slot is overwritten by the next stitch's kernel, and a
        // broken promise in the middle of a run of trusted stitches must not be lost with it
        if (check_expect &&
            !(out.ok && out.planes_floats == expect.planes_floats &&
              out.blurred_floats == expect.blurred_floats && out.scratch_floats == expect.scratch_floats &&
              out.n_records == expect.n_records && out.n_tiles == expect.n_tiles &&
              out.max_vw == expect.max_vw && out.max_vh == expect.max_vh && out.max_aw == expect.max_aw &&
              out.max_ah == expect.max_ah && out.missing == 0))
            *(volatile int *)sticky = 1;
    }
}

static int ensure_layout_buffers(pano_ctx *ctx, int n) {
    if (!ctx->lay_sum_host) {
        // (the summary, then the sticky mismatch word of the trusted stitches)
        PANO_HIP(hipHostMalloc((void **)&ctx->lay_sum_host, sizeof(LayoutSummary) + 64, hipHostMallocDefault));
        memset(ctx->lay_sum_host, 0, sizeof(LayoutSummary) + 64);
    }
    if (n > ctx->lay_cap_n) {
        PANO_HIP(hipStreamSynchronize(ctx->stream));            // a queued kernel may still read them
        if (ctx->lay_rects_dev) PANO_HIP(hipFree(ctx->lay_rects_dev));
        if (ctx->lay_have_dev) PANO_HIP(hipFree(ctx->lay_have_dev));
        ctx->lay_rects_dev = nullptr;
        ctx->lay_have_dev = nullptr;
        PANO_HIP(hipMalloc((void **)&ctx->lay_rects_dev, (size_t)n * 4 * sizeof(int32_t)));
        PANO_HIP(hipMalloc((void **)&ctx->lay_have_dev, (size_t)n));
        ctx->lay_cap_n = n;
        ctx->lay_rects_host.clear();
        ctx->lay_have_host.clear();
    }
    return PANO_OK;
}

// The verified layout as the layout kernel compares it, and the sticky word behind the summary.
static LayoutSummary layout_expectation(const pano_layout &v) {
    LayoutSummary e = {};
    e.planes_floats = v.planes_floats, e.blurred_floats = v.blurred_floats;
    e.scratch_floats = v.scratch_floats;
    e.n_records = v.n_records, e.n_tiles = v.n_tiles;
    e.max_vw = v.max_vw, e.max_vh = v.max_vh, e.max_aw = v.max_aw, e.max_ah = v.max_ah;
    return e;
}
static int *layout_sticky(pano_ctx *ctx) {
    return (int *)((unsigned char *)ctx->lay_sum_host + sizeof(LayoutSummary));
}

static void stitch_signature(const pano_stitch_args *a, int tile_grid, int *sig) {
    const int v[STITCH_SIG] = {a->n, a->H, a->W, a->xs0, a->xs1, a->own0, a->own1, a->n_levels,
                               a->radius, a->max_spans, a->shortcut, a->min_gap, tile_grid};
    for (int k = 0; k < STITCH_SIG; ++k) sig[k] = v[k];
}

// The interior map of a stitch - with the level classes beside it when the caller gave a buffer
// for them (radii of the levels: half their apertures, ascending as stitcher.py:218 makes them).
static int stitch_interior_map(pano_ctx *ctx, const pano_stitch_args *a, int n_blur) {
    if (a->classes && n_blur > 0 && ctx->opt[PANO_OPT_LEVEL_CLASSES]) {
        int radii[PANO_MAX_LEVELS];
        bool ascending = true;
        for (int k = 0; k < n_blur; ++k) {
            radii[k] = a->ntaps[k] / 2;
            if (k && radii[k] < radii[k - 1]) ascending = false;
        }
        if (ascending && radii[n_blur - 1] == a->radius)
            return pano_interior_classes(ctx, a->owner, a->H, a->W, a->own0, a->own1, radii, n_blur,
                                         a->block_owner, a->interior, a->classes);
    }
    return pano_interior_map(ctx, a->owner, a->H, a->W, a->own0, a->own1, a->radius, a->block_owner,
                             a->interior);
}

// Do this stitch's classes hold (the buffer was given and stitch_interior_map filled it)?
static bool stitch_has_classes(const pano_ctx *ctx, const pano_stitch_args *a, int n_blur) {
    if (!a->classes || n_blur <= 0 || !ctx->opt[PANO_OPT_LEVEL_CLASSES]) return false;
    for (int k = 0; k < n_blur; ++k)
        if (k && a->ntaps[k] / 2 < a->ntaps[k - 1] / 2) return false;
    return a->ntaps[n_blur - 1] / 2 == a->radius;
}

// Runs `call` with the context targeted at its side stream.
#define ON_SIDE(ctx, call)                  \
    do {                                    \
        const hipStream_t main_ = (ctx)->stream; \
        (ctx)->stream = (ctx)->side;        \
        const int rc_side_ = (call);        \
        (ctx)->stream = main_;              \
        if (rc_side_) return rc_side_;      \
    } while (0)

// Everything behind the record table: its upload (host layout) or nothing (device layout),
// the tile flags / warp-need flags, the warp, the blur's work list, the blur, the collapse.
// `lay`: the layout, or - device layout - the bounds the launches are sized by.
// `kept`: the tile flags, the warp-need flags and the blur's work list are the previous stitch's
// (kept geometry): only the warp, the blur and the collapse are queued.
static int queue_tail(pano_ctx *ctx, pano_stitch_args *a, const pano_layout &lay, bool device_table,
                      int need_choice, bool two_streams, bool interior, int n_blur, int tile_grid,
                      bool kept = false) {
    const hipStream_t s = ctx->stream;
    const bool use_blur = n_blur > 0 && lay.n_records > 0;
    const int nr = lay.n_records;
    if (!device_table) {
        if (lay.planes_floats > a->planes_floats || (use_blur && lay.blurred_floats > a->blurred_floats) ||
            (use_blur && lay.scratch_floats > a->scratch_floats) ||
            (interior && lay.n_tiles > a->cap_tiles))
            return PANO_EGROW;
        PANO_REQUIRE(lay.planes_floats == 0 || a->planes, "pano_stitch_multiband: no plane arena");
        if (int rc = pano_layout_place(a->records_host, lay.n_records, a->planes, a->blurred, a->scratch))
            return rc;
        if (nr) {
            PANO_HIP(hipMemcpyAsync(a->table, a->records_host, (size_t)nr * sizeof(pano_patch),
                                    hipMemcpyHostToDevice, s));
            PANO_HIP(hipEventRecord(ctx->ev_upload, s));
            ctx->upload_pending = true;
        }
    }
    // Warp only what is read: worth it when the rectangles are wide against the blur's reach
    // (8 x 1080p: -4 %; on 32 x 4K nearly every tile is within reach)
    a->used_need = 0;
    if (kept) {
        a->used_need = need_choice == 1;
        two_streams = false;
    } else if (interior && tile_grid == 32 && nr && a->tile_flags && a->need) {
        bool on = need_choice == 1;
        if (need_choice < 0) {
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
        if (device_table) {                                      // behind the layout kernel
            PANO_HIP(hipEventRecord(ctx->ev_upload, s));
        }
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
    if (kept && use_blur) {                      // the work list in the context's buffers is this table's
        ctx->prepared_table = a->table;
        ctx->prepared_n = nr;
    }
    if (n_blur)
        if (int rc = pano_multiband_blur(ctx, a->table, nr, lay.max_aw, lay.max_vh, lay.max_ah,
                                         a->owner, a->W, a->taps, (const int *)a->ntaps, n_blur,
                                         interior ? a->interior : nullptr,
                                         interior ? a->tile_flags : nullptr))
            return rc;
    return pano_multiband_compose(ctx, a->table, nr, a->H, a->W, a->xs0, a->xs1, a->n_levels,
                                  a->owner, a->valid, interior ? a->interior : nullptr,
                                  interior && stitch_has_classes(ctx, a, n_blur) ? a->classes : nullptr,
                                  interior ? a->cams : nullptr, interior ? a->sin_t : nullptr,
                                  interior ? a->cos_t : nullptr, interior ? a->tan_p : nullptr,
                                  a->lut, a->lut_stride, a->mosaic, a->mosaic_f32, 0);
}

// The buffers a stitch leaves its geometry in (and reads it from when it is kept).
static void geometry_buffers(const pano_stitch_args *a, const void *out[GEOM_BUFS]) {
    const void *v[GEOM_BUFS] = {a->owner, a->valid, a->table, a->interior, a->tile_flags, a->need,
                                a->planes, a->blurred, a->scratch, a->sin_t, a->classes};
    for (int k = 0; k < GEOM_BUFS; ++k) out[k] = v[k];
}
static void geometry_left(pano_ctx *ctx, const pano_stitch_args *a) {
    geometry_buffers(a, ctx->geom_bufs);
    ctx->geom_valid = true;
}

namespace {
struct StitchScope {                 // the stitch's own nested calls do not void the kept geometry
    pano_ctx *ctx;
    explicit StitchScope(pano_ctx *c) : ctx(c) { ctx->in_stitch = true; }
    ~StitchScope() { ctx->in_stitch = false; }
};
}  // namespace

extern "C" int pano_stitch_multiband(pano_ctx *ctx, pano_stitch_args *a, int resume) {
    const bool geom_was_valid = ctx && ctx->geom_valid;  // (PANO_ENTER is a foreign call's entry too)
    PANO_ENTER(ctx, "pano_stitch_multiband");
    ctx->geom_valid = false;
    StitchScope scope(ctx);
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
    // (round 5, with trusted layouts: still nothing below - a world-8 strip 0.363 against 0.362 ms,
    // config 2 0.50 against 0.444; profiles/r05/visit_p_*.txt)
    const bool big = (long long)a->H * (a->own1 - a->own0) >= (1ll << 24);
    const bool two_streams = ctx->opt[PANO_OPT_STITCH_STREAMS] != 0 && big;
    const int tile_grid = pano_blur_tile_grid(ctx);
    const int stride = 5 + 2 * a->max_spans;
    int sig[STITCH_SIG];
    stitch_signature(a, tile_grid, sig);

    if (!resume) {
        // The layout on the device (option PANO_OPT_STITCH_ASYNC): for the default form of the
        // stitch - interior shortcut, matrix-core blur - once a stitch of this shape has gone
        // through and left its layout's needs behind.
        bool spec = ctx->opt[PANO_OPT_STITCH_ASYNC] != 0 && interior && tile_grid == 32 &&
                    ctx->lay_prev_valid && memcmp(sig, ctx->lay_prev_sig, sizeof(sig)) == 0 &&
                    ctx->lay_prev.n_records > 0 && a->planes && a->blurred && a->tile_flags;
        pano_layout bound = ctx->lay_prev;
        // a trusted stitch (args->trust_layout = 1, see below) needs the verified layout of the
        // same shape; its summary is checked first if one is still pending
        const bool want_keep = a->trust_layout == 3;
        const bool want_trust = a->trust_layout == 1 || want_keep;
        a->trust_layout = 0;
        if (ctx->trusted_pending && !(want_trust && spec))
            if (int rc = pano_stitch_verify(ctx)) return rc;
        const bool trusted = want_trust && spec && ctx->lay_prev_verified &&
                             ctx->opt[PANO_OPT_STITCH_ASYNC] == 1;
        // (kept: only on the matrix-core grid - `spec` says so - and only while the work list in
        // the context's buffers is the one the previous stitch built for this very table: an
        // option switch voids it, pano_ctx_set_option)
        const bool list_kept = n_blur == 0 || ctx->lay_prev.n_records == 0 ||
                               (ctx->item_buf && ctx->list_table == a->table &&
                                ctx->list_n == ctx->lay_prev.n_records);
        if (want_keep && trusted && geom_was_valid && tile_grid == 32 && list_kept) {
            // Kept geometry: the owner map, the valid mask, the interior map, the record table, the
            // tile flags and the blur's work list are functions of the cameras, rectangles, strip and
            // resident frames alone - what the caller vouches for - and lie untouched where this
            // context's previous stitch left them: the frames' pixels go through the warp, the blur
            // and the collapse, nothing else is queued.
            const void *bufs[GEOM_BUFS];
            geometry_buffers(a, bufs);
            if (memcmp(bufs, ctx->geom_bufs, sizeof(bufs)) == 0) {
                if (int rc = queue_tail(ctx, a, ctx->lay_prev, true, ctx->lay_prev_used_need, false,
                                        interior, n_blur, tile_grid, true))
                    return rc;
                a->layout = ctx->lay_prev;
                a->trust_layout = 4;                             // out: geometry kept, nobody waited
                ctx->geom_valid = true;
                ++ctx->lay_count[0];
                return PANO_OK;
            }
        }
        if (spec) {
            if (int rc = ensure_layout_buffers(ctx, a->n)) return rc;
            // rectangles and resident flags for the layout kernel: uploaded when they changed,
            // in front of the ownership kernel (off the critical path)
            const size_t nrect = (size_t)a->n * 4;
            if (ctx->lay_rects_host.size() != nrect ||
                memcmp(ctx->lay_rects_host.data(), a->rects, nrect * sizeof(int32_t)) != 0) {
                ctx->lay_rects_host.assign(a->rects, a->rects + nrect);
                PANO_HIP(hipMemcpyAsync(ctx->lay_rects_dev, ctx->lay_rects_host.data(),
                                        nrect * sizeof(int32_t), hipMemcpyHostToDevice, s));
            }
            std::vector<uint8_t> have(a->n, 1);
            if (a->have) have.assign(a->have, a->have + a->n);
            if (ctx->lay_have_host != have) {
                ctx->lay_have_host = have;
                PANO_HIP(hipMemcpyAsync(ctx->lay_have_dev, ctx->lay_have_host.data(), (size_t)a->n,
                                        hipMemcpyHostToDevice, s));
            }
            // launch bounds: what the previous layout needed, with slack for cameras that move
            // (a trusted stitch repeats that layout exactly)
            if (!trusted) {
                bound.n_records = bound.n_records + 2 < a->cap_records ? bound.n_records + 2 : a->cap_records;
                bound.max_vw += 64, bound.max_vh += 64, bound.max_aw += 64, bound.max_ah += 64;
            }
            if (ctx->opt[PANO_OPT_STITCH_ASYNC] == 2) {          // tests: bounds this layout exceeds
                bound.n_records = ctx->lay_prev.n_records > 1 ? ctx->lay_prev.n_records - 1 : 1;
                bound.max_vw = ctx->lay_prev.max_vw - 1;
            }
        }
        // ownership and, in the same pass, one record per (camera, span of columns it owns):
        // boxes and column marks out of the ownership kernel, then the spans' search.  (Rounds 3 - 4
        // took the one pass on mosaics of 16 MP and more only; with round 5's ownership kernel it is
        // as fast or faster on every size - config 2 0.449 / 0.451 ms per stitch, a world-8 strip of
        // config 3 0.321 against 0.331: the separate box kernel's 30 us were a quarter of a strip's
        // ownership - profiles/r05/ab_regions_fused_*.txt.)
        if (int rc = pano_ownership_regions(ctx, a->cams, a->n, a->H, a->W, a->own0, a->own1,
                                            a->sin_t, a->cos_t, a->tan_p, a->owner, a->valid,
                                            a->min_gap, a->max_spans, a->marks, a->regions))
            return rc;
        // Two small chains depend on the owner map only - the interior map and the region
        // search's tail - and two more on the record table only - the warp and the blur's tile
        // flags and work list: the context's side stream takes one of each pair (the short
        // kernels of a config-3 stitch were 0.15 ms of a 2.0 ms timeline, plus the gaps between
        // them).
        const bool forked = two_streams;
        if (forked && interior) {
            PANO_HIP(hipEventRecord(ctx->ev_fork, s));
            PANO_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
            ON_SIDE(ctx, stitch_interior_map(ctx, a, n_blur));
        }
        bool interior_queued = forked && interior;
        if (spec) {
            if (ctx->upload_pending) {                           // records_host may be in use
                PANO_HIP(hipEventSynchronize(ctx->ev_upload));
                ctx->upload_pending = false;
            }
            uintptr_t bbase = (uintptr_t)a->blurred;             // as pano_layout_place aligns it
            bbase += (uintptr_t)(-(intptr_t)bbase) % 128;
            hipLaunchKernelGGL(layout_windows_kernel, dim3(1), dim3(LAY_THREADS), 0, s, a->regions,
                               ctx->lay_rects_dev, ctx->lay_have_dev, a->n, a->max_spans, a->radius,
                               a->xs0, a->xs1, n_blur, 1, a->planes, (float *)bbase, a->scratch,
                               (long)a->planes_floats, (long)a->blurred_floats,
                               (long)a->scratch_floats, a->cap_tiles, 1, bound.n_records,
                               bound.max_vw, bound.max_vh, bound.max_aw, bound.max_ah, a->table,
                               ctx->lay_sum_host, layout_expectation(ctx->lay_prev), trusted ? 1 : 0,
                               layout_sticky(ctx));
            PANO_LAUNCH_CHECK("layout_windows_kernel");
            // the summary goes straight into pinned host memory; the caller's copy of the records
            // leaves through the side stream: nothing stands between this kernel and the warp
            PANO_HIP(hipEventRecord(ctx->ev_regions, s));
            if (!trusted) {
                PANO_HIP(hipStreamWaitEvent(ctx->side, ctx->ev_regions, 0));
                PANO_HIP(hipMemcpyAsync(a->records_host, a->table,
                                        (size_t)bound.n_records * sizeof(pano_patch),
                                        hipMemcpyDeviceToHost, ctx->side));
                PANO_HIP(hipEventRecord(ctx->ev_copy, ctx->side));
            }
            if (interior && !interior_queued) {
                if (int rc = stitch_interior_map(ctx, a, n_blur)) return rc;
                interior_queued = true;
            }
            if (int rc = queue_tail(ctx, a, bound, true, ctx->lay_prev_used_need, two_streams, interior,
                                    n_blur, tile_grid))
                return rc;
            if (trusted) {
                // The caller vouches that cameras, rectangles, strip and resident frames are those
                // of this context's previous stitch, whose layout was read and verified: the owner
                // map, the regions and so the layout are functions of exactly those, the kernels
                // were queued with that layout's own bounds, and nobody waits - the summary this
                // stitch's layout kernel writes is compared with the verified one at the next
                // untrusted call or in pano_stitch_verify.
                a->layout = ctx->lay_prev;
                a->used_need = ctx->lay_prev_used_need;
                a->trust_layout = 2;                             // out: went through without a wait
                ctx->trusted_pending = true;
                geometry_left(ctx, a);
                ++ctx->lay_count[0];
                return PANO_OK;
            }
            PANO_HIP(hipEventSynchronize(ctx->ev_regions));      // the GPU is in the warp by now
            PANO_HIP(hipEventSynchronize(ctx->ev_copy));
            const LayoutSummary sum = *ctx->lay_sum_host;
            a->layout.planes_floats = sum.planes_floats;
            a->layout.blurred_floats = sum.blurred_floats;
            a->layout.scratch_floats = sum.scratch_floats;
            a->layout.n_records = sum.n_records, a->layout.n_tiles = sum.n_tiles;
            a->layout.max_vw = sum.max_vw, a->layout.max_vh = sum.max_vh;
            a->layout.max_aw = sum.max_aw, a->layout.max_ah = sum.max_ah;
            a->layout.missing = sum.missing;
            if (sum.ok) {
                ctx->lay_prev = a->layout;
                ctx->lay_prev_used_need = a->used_need;
                ctx->lay_prev_verified = true;
                geometry_left(ctx, a);
                ++ctx->lay_count[0];
                return PANO_OK;
            }
            ++ctx->lay_count[1];
            // this layout needs more than the launches covered (or an arena is too small): the
            // table was emptied, the queued tail did nothing; lay it out on the host and queue
            // the tail again
            ctx->lay_prev_valid = false;
            ctx->lay_prev_verified = false;
        }
        PANO_HIP(hipMemcpyAsync(a->regions_host, a->regions, (size_t)a->n * stride * sizeof(int32_t),
                                hipMemcpyDeviceToHost, s));
        PANO_HIP(hipEventRecord(ctx->ev_regions, s));
        // the interior map needs the owner map only: queued before the wait, it keeps the GPU
        // busy while the host lays out the windows
        if (interior && !interior_queued)
            if (int rc = stitch_interior_map(ctx, a, n_blur)) return rc;
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
    const int rc = queue_tail(ctx, a, a->layout, false, a->warp_need, two_streams, interior, n_blur,
                              tile_grid);
    if (rc == PANO_OK) {                     // what this layout needed: the next stitch's bounds
        ctx->lay_prev = a->layout;
        memcpy(ctx->lay_prev_sig, sig, sizeof(sig));
        ctx->lay_prev_used_need = a->used_need;
        ctx->lay_prev_valid = true;
        ctx->lay_prev_verified = true;       // laid out on the host from this stitch's own regions
        geometry_left(ctx, a);
    }
    return rc;
}

// The summary the LAST trusted stitch's layout kernel wrote against the verified layout those
// stitches were queued with: the same cameras give the same regions and the same layout, so a
// difference means the caller's promise (args->trust_layout) did not hold - the mosaics of the
// trusted stitches since the last verification are then void (PANO_EINVAL).  Waits for that
// stitch's layout kernel (an event long past when mosaics are collected a stitch later).
extern "C" int pano_stitch_verify(pano_ctx *ctx) {
    PANO_REQUIRE(ctx, "pano_stitch_verify: null context");
    if (!ctx->trusted_pending) return PANO_OK;
    PANO_HIP(hipSetDevice(ctx->device));                 // (writes no buffer: kept geometry stays)
    PANO_HIP(hipEventSynchronize(ctx->ev_regions));
    ctx->trusted_pending = false;
    const LayoutSummary sum = *ctx->lay_sum_host;
    const pano_layout &v = ctx->lay_prev;
    // (the last trusted stitch's summary, and the sticky word every trusted stitch before it left)
    const int sticky = __atomic_exchange_n(layout_sticky(ctx), 0, __ATOMIC_SEQ_CST);
    const bool same = !sticky && sum.ok && sum.planes_floats == v.planes_floats &&
                      sum.blurred_floats == v.blurred_floats && sum.scratch_floats == v.scratch_floats &&
                      sum.n_records == v.n_records && sum.n_tiles == v.n_tiles &&
                      sum.max_vw == v.max_vw && sum.max_vh == v.max_vh && sum.max_aw == v.max_aw &&
                      sum.max_ah == v.max_ah && sum.missing == 0;
    if (!same) {
        ctx->lay_prev_valid = ctx->lay_prev_verified = false;
        ctx->geom_valid = false;
        pano_set_error("pano_stitch_verify: a trusted stitch laid out %d records (ok %d, why %d, an "
                       "earlier one differed: %d) where the verified layout has %d: the cameras were "
                       "not those of the verified stitch",
                       sum.n_records, sum.ok, sum.why, sticky, v.n_records);
        return PANO_EINVAL;
    }
    return PANO_OK;
}

extern "C" int pano_stitch_counts(const pano_ctx *ctx, int *device_layouts, int *fallbacks) {
    PANO_REQUIRE(ctx && device_layouts && fallbacks, "pano_stitch_counts: null pointer");
    *device_layouts = ctx->lay_count[0];
    *fallbacks = ctx->lay_count[1];
    return PANO_OK;
}
