// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this pool (MI355X_MICROARCH.md, "HBM":
// FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads on gfx950, other access
// widths are uncalibrated).  One kernel per access shape the stitch uses, each reading or
// writing a KNOWN number of bytes exactly once over a footprint far beyond the 256 MiB
// Infinity Cache; the program prints the known bytes per kernel name as JSON, and
// tools/fetch_calib.sh divides them by the counters of the same launches:
//
//   read16      16 B per lane, contiguous (the blur's band chunks: buffer_load_dwordx4)
//   read4       4 B per lane, contiguous (the collapse's planar gathers along a row)
//   read4x5     4 B per lane from five planes at the same offset (the collapse's seam pixels)
//   read2       2 B per lane, contiguous (owner map)
//   read1       1 B per lane, contiguous (uint8 frames, interior map)
//   taps6       per lane a 4-byte + a 2-byte load at a byte position that advances 3 B per
//               lane, two rows (the warp's bilinear taps on a uint8 RGB frame, no reuse
//               between waves: every byte of the footprint is touched once or twice, the
//               footprint is the known figure)
//   gather64    64-byte segments in a pseudo-random order, one byte per lane (a scattered
//               byte gather, every segment once)
//   write16 / write4 / tile4 / write3   16 B per lane, 4 B per lane, the 32 x 32 float tile of
//               an MFMA accumulator (sixteen dword stores of two 128-byte rows each), and the
//               mosaic's 3 bytes per pixel
//
// Build: hipcc -O3 --offload-arch=gfx950 -o build/fetch_calib tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// every kernel folds what it read into `sink` under a condition that never holds, so the
// loads stay and nothing is written
__global__ __launch_bounds__(256) void calib_read16(const u4 *src, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u4 v = __builtin_nontemporal_load(src + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_read4(const unsigned *src, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= src[i];
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_read4x5(const unsigned *src, size_t plane, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < plane; i += (size_t)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 5; ++k) acc ^= src[k * plane + i];
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_read2(const uint16_t *src, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= src[i];
    if (acc == 0x1234u) *sink = acc;          // (a value a 16-bit XOR can take: the loads must stay)
}

__global__ __launch_bounds__(256) void calib_read1(const uint8_t *src, size_t n, unsigned *sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= src[i];
    if (acc == 0x55u) *sink = acc;
}

// rows of `pitch` bytes; a wave takes 64 pixels (192 bytes) of rows y and y + 1
__global__ __launch_bounds__(256) void calib_taps6(const uint8_t *src, int pitch, int rows, unsigned *sink) {
    typedef uint32_t u32_any __attribute__((aligned(1)));
    typedef uint16_t u16_any __attribute__((aligned(1)));
    unsigned acc = 0;
    const int per_row = (pitch - 8) / 192;                     // waves per row pair
    const size_t nwave = (size_t)(rows / 2) * per_row;
    const int lane = threadIdx.x & 63;
    for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nwave; w += (size_t)gridDim.x * 4) {
        const size_t y = 2 * (w / per_row), x = (w % per_row) * 192 + 3 * lane;
        const uint8_t *p = src + y * pitch + x;
        acc ^= *(const u32_any *)p ^ *(const u16_any *)(p + 4);
        acc ^= *(const u32_any *)(p + pitch) ^ *(const u16_any *)(p + pitch + 4);
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_gather64(const uint8_t *src, size_t nseg, unsigned *sink) {
    unsigned acc = 0;
    const int lane = threadIdx.x & 63;
    // nseg is a power of two: an odd multiplier permutes the segments
    for (size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); w < nseg; w += (size_t)gridDim.x * 4) {
        const size_t seg = (w * 2654435761ull) & (nseg - 1);
        acc ^= src[seg * 64 + lane];
    }
    if (acc == 0x55u) *sink = acc;
}

__global__ __launch_bounds__(256) void calib_write16(u4 *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u4 v = {(unsigned)i, 1u, 2u, 3u};
        dst[i] = v;
    }
}

__global__ __launch_bounds__(256) void calib_write4(unsigned *dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        dst[i] = (unsigned)i;
}

// 32 x 32 float tiles of a plane of `pitch` floats, written as an MFMA accumulator leaves a wave
__global__ __launch_bounds__(256) void calib_tile4(float *dst, int pitch, int ntx, size_t ntiles) {
    const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
    for (size_t t = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < ntiles; t += (size_t)gridDim.x * 4) {
        float *base = dst + (t / ntx) * 32 * (size_t)pitch + (t % ntx) * 32;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            base[(size_t)((q & 3) + 8 * (q >> 2) + 4 * h) * pitch + n] = (float)q;
    }
}

__global__ __launch_bounds__(256) void calib_write3(uint8_t *dst, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        dst[3 * i] = (uint8_t)i;
        dst[3 * i + 1] = (uint8_t)(i >> 8);
        dst[3 * i + 2] = (uint8_t)(i >> 16);
    }
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) << 20;     // footprint, MiB
    const int reps = argc > 2 ? atoi(argv[2]) : 3;
    uint8_t *buf;
    unsigned *sink;
    CK(hipMalloc((void **)&buf, bytes + 4096));
    CK(hipMalloc((void **)&sink, 4));
    CK(hipMemset(buf, 1, bytes + 4096));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    const int pitch = 11520 + 8;                               // a 4K RGB row + slack
    const int rows = (int)(bytes / pitch) & ~1;
    const int tile_pitch = 4096, ntx = tile_pitch / 32;
    const size_t ntiles = (bytes / 4 / tile_pitch / 32) * ntx;
    size_t nseg = 1;
    while (nseg * 2 * 64 <= bytes) nseg *= 2;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    printf("{\"footprint_bytes\": %zu, \"reps\": %d, \"kernels\": {\n", bytes, reps);
    struct Row { const char *name; double rd, wr; float ms; } rowsout[16];
    int nrow = 0;
#define RUN(NAME, RD, WR, ...)                                                 \
    do {                                                                       \
        float best = 1e30f;                                                    \
        for (int r = 0; r < reps; ++r) {                                       \
            CK(hipEventRecord(a));                                             \
            hipLaunchKernelGGL(NAME, dim3(grid), dim3(256), 0, 0, __VA_ARGS__); \
            CK(hipEventRecord(b));                                             \
            CK(hipEventSynchronize(b));                                        \
            float ms;                                                          \
            CK(hipEventElapsedTime(&ms, a, b));                                \
            best = ms < best ? ms : best;                                      \
        }                                                                      \
        CK(hipGetLastError());                                                 \
        rowsout[nrow++] = Row{#NAME, (double)(RD), (double)(WR), best};        \
    } while (0)
    RUN(calib_read16, bytes, 0, (const u4 *)buf, bytes / 16, sink);
    RUN(calib_read4, bytes, 0, (const unsigned *)buf, bytes / 4, sink);
    RUN(calib_read4x5, bytes / 20 * 20, 0, (const unsigned *)buf, bytes / 20, sink);
    RUN(calib_read2, bytes, 0, (const uint16_t *)buf, bytes / 2, sink);
    RUN(calib_read1, bytes, 0, (const uint8_t *)buf, bytes, sink);
    RUN(calib_taps6, (double)rows * ((pitch - 8) / 192 * 192 + 3), 0, (const uint8_t *)buf, pitch, rows, sink);
    RUN(calib_gather64, nseg * 64, 0, (const uint8_t *)buf, nseg, sink);
    RUN(calib_write16, 0, bytes, (u4 *)buf, bytes / 16);
    RUN(calib_write4, 0, bytes, (unsigned *)buf, bytes / 4);
    RUN(calib_tile4, 0, ntiles * 4096, (float *)buf, tile_pitch, ntx, ntiles);
    RUN(calib_write3, 0, bytes / 3 * 3, buf, bytes / 3);
    for (int i = 0; i < nrow; ++i)
        printf("  \"%s\": {\"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.0f}%s\n",
               rowsout[i].name, rowsout[i].rd, rowsout[i].wr, rowsout[i].ms,
               (rowsout[i].rd + rowsout[i].wr) / rowsout[i].ms * 1e-6, i + 1 < nrow ? "," : "");
    printf("}}\n");
    return 0;
}
