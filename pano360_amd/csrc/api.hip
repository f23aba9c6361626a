// Library-level entry points: version, error string, device probe.
#include <stdarg.h>
#include <stdio.h>

#include <vector>

#include "common.h"

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

// ---- per-kernel timing -------------------------------------------------------
// bench.py measures the dominant kernel's launch durations live with HIP events
// recorded on the stream the kernel is launched on.  Instrumentation only: off
// unless pano_timing_enable(1) was called; single-threaded use.
bool g_pano_timing_on = false;
static std::vector<hipEvent_t> g_begin[PK_COUNT], g_end[PK_COUNT];
static const char *const g_kernel_names[PK_COUNT] = {
    "add_weights_kernel", "warp_spherical_kernel", "ownership_kernel", "blur_rows_kernel",
    "blur_cols_kernel",   "multiband_compose_kernel", "linear_blend_kernel",
    "no_blend_kernel",    "crop_heights_kernel", "crop_rows_kernel", "pyr_down_kernel",
    "ownership_cameras_kernel", "owned_boxes_kernel", "warp_windows_kernel",
    "blend_cameras_kernel", "owned_spans_kernel",
    "block_owner_kernel", "tile_flags_kernel", "overlap_stats_kernel",
    "blur_mfma_kernel", "sift_extrema_kernel", "sift_orient_kernel", "sift_describe_kernel",
    "compose_interior_kernel"};

void pano_timing_edge(int kid, hipStream_t stream, bool begin) {
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, stream);
    (begin ? g_begin : g_end)[kid].push_back(ev);
}

static void timing_clear() {
    for (int k = 0; k < PK_COUNT; ++k) {
        for (hipEvent_t e : g_begin[k]) (void)hipEventDestroy(e);
        for (hipEvent_t e : g_end[k]) (void)hipEventDestroy(e);
        g_begin[k].clear();
        g_end[k].clear();
    }
}

extern "C" int pano_timing_enable(int on) {
    timing_clear();
    g_pano_timing_on = on != 0;
    return PANO_OK;
}

extern "C" int pano_kernel_count(void) { return PK_COUNT; }

extern "C" const char *pano_kernel_name(int kid) {
    return kid >= 0 && kid < PK_COUNT ? g_kernel_names[kid] : "";
}

extern "C" int pano_timing_read(int kid, double *total_ms, int *launches) {
    PANO_REQUIRE(kid >= 0 && kid < PK_COUNT && total_ms && launches, "pano_timing_read: bad argument");
    double sum = 0.0;
    const size_t n = g_end[kid].size() < g_begin[kid].size() ? g_end[kid].size() : g_begin[kid].size();
    for (size_t i = 0; i < n; ++i) {
        PANO_HIP(hipEventSynchronize(g_end[kid][i]));
        float ms = 0.f;
        PANO_HIP(hipEventElapsedTime(&ms, g_begin[kid][i], g_end[kid][i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = (int)n;
    return PANO_OK;
}
