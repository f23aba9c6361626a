// Cost of writing 32 x 32 float tiles from 8 waves per CU with different store shapes.
//   mode 0: 16 dword stores, a wave instruction = 2 rows x 128 B (the MFMA D layout as it is)
//   mode 1: 4 dwordx4 stores, lane = row, 16 B per lane (the transposed-product layout)
//   mode 2: 4 dwordx4 stores, a quad of lanes = 4 rows x 64 B... (quad-transposed layout)
//   mode 3: 8 dwordx2 stores
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float *dst, int pitch, int tiles_per_wave, int ntx) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 31, h = lane >> 5;
    const int wave_id = blockIdx.x * 8 + wv;
    float v[16];
    for (int q = 0; q < 16; ++q) v[q] = (float)(lane + q);
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int tile = wave_id * tiles_per_wave + t;
        const int tx = tile % ntx, ty = tile / ntx;
        float *base = dst + (size_t)ty * 32 * pitch + tx * 32;
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
                base[(size_t)((q & 3) + 8 * (q >> 2) + 4 * h) * pitch + n] = v[q];
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4 val = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                *(f4 *)(base + (size_t)n * pitch + 8 * g + 4 * h) = val;
            }
        } else if (MODE == 2) {
            const int c = lane & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f4 val = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                *(f4 *)(base + (size_t)(8 * g + 4 * h + c) * pitch + (n & ~3)) = val;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f2 val = {v[2 * g], v[2 * g + 1]};
                *(f2 *)(base + (size_t)(4 * g + 2 * h + (lane & 1)) * pitch + (n & ~1)) = val;
            }
        }
        for (int q = 0; q < 16; ++q) v[q] += 1.0f;
    }
}

int main(int argc, char **argv) {
    const int ntx = 128, nty = 600, pitch = ntx * 32;          // 4096 x 19200 floats = 315 MB
    float *d;
    hipMalloc(&d, (size_t)pitch * nty * 32 * 4);
    const int nwg = argc > 1 ? atoi(argv[1]) : 256;            // workgroups = busy CUs
    const int tiles = ntx * nty, waves = nwg * 8, per = tiles / (256 * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(512), 0, 0, d, pitch, per, ntx);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(512), 0, 0, d, pitch, per, ntx);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(512), 0, 0, d, pitch, per, ntx);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(512), 0, 0, d, pitch, per, ntx);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (rep == 2)
                printf("mode %d: %.3f ms, %.0f GB/s, %.0f cycles per tile per CU (2.4 GHz)\n", mode, ms,
                       (double)per * waves * 4096 / ms / 1e6, ms * 1e-3 * 2.4e9 / (per * 8));
            (void)waves;
        }
    }
    return 0;
}
