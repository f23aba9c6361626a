// One frame of the SIFT front end per native call, replayed as a HIP graph.
//   reference: features.py:192-201 (cv2.xfeatures2d.SIFT_create().detectAndCompute; PARITY
//   UNPINNED - OpenCV is not under /root/reference, see include/pano360.h).
// pano_scale_space, pano_sift_extrema per octave, pano_sift_orient, pano_sift_sort_unique and
// pano_sift_describe in that order on the context's stream - about 110 dependent launches for a
// 4K frame, of which the last eight octaves are a few workgroups each.  Queued launch by launch they cost the HOST more than the GPU (3.0 ms of kernels took
// 4.8 - 6.5 ms per frame on slower hosts, profiles/r06/notes.md); every grid and every argument
// is fixed by the frame size and the buffers alone - the candidate and keypoint counts stay on
// the device - so the sequence up to the orientations (~100 launches) is captured ONCE per set of
// buffers (hipStreamBeginCapture, thread-local mode) and replayed with one hipGraphLaunch per
// frame; the sort and the descriptors follow launch by launch.
#include <string.h>

#include "common.h"

// Counters are zeroed by a kernel, not by hipMemsetAsync: the captured sequence holds nothing but
// kernel nodes of this library, in stream order.  (The first graph of this sequence, with memset
// nodes and rocPRIM's launches inside, ended in a fault of the orientation kernel on a stale
// keypoint list - profiles/r06/notes.md; which of the two was to blame was not pursued.)
__global__ __launch_bounds__(64) void zero_i32_kernel(int *__restrict__ p, int n) {
    for (int i = threadIdx.x; i < n; i += 64) p[i] = 0;
}

int pano_zero_i32(hipStream_t s, int *p, int n) {
    hipLaunchKernelGGL(zero_i32_kernel, dim3(1), dim3(64), 0, s, p, n);
    PANO_LAUNCH_CHECK("zero_i32_kernel");
    return PANO_OK;
}

static uint64_t fnv(uint64_t h, const void *p, size_t bytes) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

// Everything the launch sequence depends on: sizes, switches, every buffer address, the taps' values.
static uint64_t detect_key(const pano_sift_args *a, const uint8_t *frame) {
    uint64_t h = 1469598103934665603ull;
    const int ints[8] = {a->h, a->w, a->n_octaves, a->n_layers, a->first_octave, a->max_keypoints,
                         a->detect, 0};
    h = fnv(h, ints, sizeof(ints));
    const float fl[3] = {a->contrast_thr, a->edge_thr, a->sigma};
    h = fnv(h, fl, sizeof(fl));
    const void *ptrs[10] = {frame, a->work, a->gauss_dev, a->dims_dev, a->cands, a->kpts, a->counts,
                            a->sort_work, a->desc, nullptr};
    h = fnv(h, ptrs, sizeof(ptrs));
    h = fnv(h, a->gauss, (size_t)a->n_octaves * sizeof(float *));
    h = fnv(h, a->dog, (size_t)a->n_octaves * sizeof(float *));
    size_t floats = 0;
    for (int i = 0; i < a->n_layers + 3; ++i) floats += (size_t)a->ntaps[i];
    h = fnv(h, a->ntaps, (size_t)(a->n_layers + 3) * sizeof(int));
    return fnv(h, a->taps, floats * sizeof(float));
}

// The launch sequence of one frame on ctx->stream, in two parts.  The front - scale space, extrema,
// orientations: ~100 launches of this library's own kernels - is what a graph replays; the back -
// the keypoint sort (rocPRIM, which sizes and zeroes its own state) and the descriptors, a dozen
// launches - is always queued launch by launch behind it.
static int queue_front(pano_ctx *ctx, const pano_sift_args *a, const uint8_t *frame) {
    const hipStream_t s = ctx->stream;
    if (int rc = pano_scale_space(ctx, frame, a->h, a->w, a->n_octaves, a->n_layers, a->taps,
                                  a->ntaps, a->gauss, a->dog, a->work))
        return rc;
    if (!a->detect) return PANO_OK;
    if (int rc = pano_zero_i32(s, a->counts, 3)) return rc;
    int rows = 2 * a->h, cols = 2 * a->w;
    for (int o = 0; o < a->n_octaves; ++o) {
        if (int rc = pano_sift_extrema(ctx, a->dog[o], rows, cols, o, a->n_layers, a->contrast_thr,
                                       a->edge_thr, a->sigma, a->cands, a->counts, a->max_keypoints))
            return rc;
        rows /= 2;
        cols /= 2;
    }
    return pano_sift_orient(ctx, a->gauss_dev, a->dims_dev, a->n_layers, a->cands, a->counts,
                            a->max_keypoints, a->kpts, a->counts + 1, a->max_keypoints);
}

static int queue_back(pano_ctx *ctx, const pano_sift_args *a) {
    if (!a->detect) return PANO_OK;
    // OpenCV's order and duplicate removal, the first-octave adjustment; `cands` is free again
    // and takes the result
    if (int rc = pano_sift_sort_unique(ctx, a->kpts, a->max_keypoints, a->counts + 1, a->first_octave,
                                       a->sort_work, a->cands, a->counts + 2))
        return rc;
    return pano_sift_describe(ctx, a->gauss_dev, a->dims_dev, a->first_octave, a->cands,
                              a->max_keypoints, a->counts + 2, a->desc);
}

static int queue_frame(pano_ctx *ctx, const pano_sift_args *a, const uint8_t *frame) {
    if (int rc = queue_front(ctx, a, frame)) return rc;
    return queue_back(ctx, a);
}

void pano_sift_graphs_free(pano_ctx *ctx) {
    for (PanoSiftGraph &g : ctx->sift_graphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    ctx->sift_graphs.clear();
}

#define PANO_SIFT_GRAPHS_MAX 8

extern "C" int pano_sift_detect(pano_ctx *ctx, const pano_sift_args *a) {
    PANO_ENTER(ctx, "pano_sift_detect");
    PANO_REQUIRE(a, "pano_sift_detect: null arguments");
    PANO_REQUIRE(a->frame && a->taps && a->ntaps && a->gauss && a->dog && a->work,
                 "pano_sift_detect: null pointer");
    PANO_REQUIRE(a->h > 0 && a->w > 0 && a->n_octaves >= 1 && a->n_octaves <= 32 && a->n_layers >= 1 &&
                     a->n_layers <= 8,
                 "pano_sift_detect: bad argument");
    if (a->detect) {
        PANO_REQUIRE(a->gauss_dev && a->dims_dev && a->cands && a->kpts && a->counts && a->sort_work &&
                         a->desc && a->max_keypoints > 0,
                     "pano_sift_detect: detection without its buffers");
        PANO_REQUIRE(a->n_layers == 3, "pano_sift_detect: detection takes 3 layers per octave");
    }
    const hipStream_t s = (hipStream_t)stream;
    // replay needs a frame buffer of its own (a graph holds addresses): `frame_copy`, optional
    const bool graphs = ctx->opt[PANO_OPT_SIFT_GRAPH] != 0 && !ctx->timing_on && a->frame_copy;
    if (!graphs) return queue_frame(ctx, a, a->frame);
    const uint64_t key = detect_key(a, a->frame_copy);
    PanoSiftGraph *slot = nullptr;
    for (PanoSiftGraph &g : ctx->sift_graphs)
        if (g.key == key) slot = &g;
    if (!slot) {
        if (ctx->sift_graphs.size() >= PANO_SIFT_GRAPHS_MAX) {       // the least recently used leaves
            size_t old = 0;
            for (size_t i = 1; i < ctx->sift_graphs.size(); ++i)
                if (ctx->sift_graphs[i].used < ctx->sift_graphs[old].used) old = i;
            if (ctx->sift_graphs[old].exec) (void)hipGraphExecDestroy(ctx->sift_graphs[old].exec);
            ctx->sift_graphs.erase(ctx->sift_graphs.begin() + (long)old);
        }
        ctx->sift_graphs.push_back(PanoSiftGraph{key, nullptr, 0, 0});
        slot = &ctx->sift_graphs.back();
    }
    slot->used = ++ctx->tick;
    const size_t frame_bytes = (size_t)a->h * a->w * 3;
    if (slot->seen == 0) {
        // the first frame of these buffers runs launch by launch: whatever the sequence allocates
        // (the extrema's list, rocPRIM's first-use state) is allocated now, outside any capture
        slot->seen = 1;
        return queue_frame(ctx, a, a->frame);
    }
    if (!slot->exec && slot->seen == 1) {
        slot->seen = 2;                                              // one attempt per set of buffers
        // captured on a stream of its own (the caller's may be the legacy default stream, which
        // cannot capture); the graph is launched wherever the context points
        hipGraph_t graph = nullptr;
        hipStream_t cap = nullptr;
        if (hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) == hipSuccess) {
            hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                ctx->stream = cap;
                const int rc = queue_front(ctx, a, a->frame_copy);
                ctx->stream = s;
                e = hipStreamEndCapture(cap, &graph);                // (always: leaves capture mode)
                if (rc != PANO_OK || e != hipSuccess || !graph) {
                    if (graph) (void)hipGraphDestroy(graph);
                    graph = nullptr;
                }
            }
            (void)hipStreamDestroy(cap);
        }
        if (graph) {
            hipGraphExec_t exec = nullptr;
            if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) slot->exec = exec;
            (void)hipGraphDestroy(graph);
        }
        (void)hipGetLastError();                                     // a failed capture is not an error
    }
    if (!slot->exec) return queue_frame(ctx, a, a->frame);           // no graph: launch by launch
    PANO_HIP(hipMemcpyAsync(a->frame_copy, a->frame, frame_bytes, hipMemcpyDeviceToDevice, s));
    PANO_HIP(hipGraphLaunch(slot->exec, s));
    return queue_back(ctx, a);
}

// 0 = no graph yet for the most recently used set of buffers, 1 = replaying
extern "C" int pano_sift_detect_replaying(const pano_ctx *ctx) {
    if (!ctx) return 0;
    const PanoSiftGraph *last = nullptr;
    for (const PanoSiftGraph &g : ctx->sift_graphs)
        if (!last || g.used > last->used) last = &g;
    return last && last->exec ? 1 : 0;
}
