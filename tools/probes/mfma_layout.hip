// Checks the lane maps of v_mfma_f32_32x32x16_f16 with exact integer data, and
// the "accumulator as the next B operand" k permutation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const float *A, const float *B, float *D, const float *A2, float *D2) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[r * 16 + 8 * h + j];        // A[row r][k]
        b[j] = (_Float16)B[(8 * h + j) * 32 + r];      // B[k][col r]
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = acc[i];
    }
    // second product: Y = A2 (32 x 32) * X, X = acc (32 x 32), two k-steps
    f32x16 y = {0};
    for (int s = 0; s < 2; ++s) {
        half8 xb, a2;
        for (int j = 0; j < 8; ++j) {
            xb[j] = (_Float16)acc[8 * s + j];
            int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);     // row of X this element is
            a2[j] = (_Float16)A2[r * 32 + k];
        }
        y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, xb, y, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) {
        int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D2[row * 32 + r] = y[i];
    }
}

int main() {
    float A[32 * 16], B[16 * 32], A2[32 * 32], D[1024], D2[1024], X[1024], Y[1024];
    srand(1);
    for (int i = 0; i < 512; ++i) { A[i] = rand() % 5 - 2; B[i] = rand() % 5 - 2; }
    for (int i = 0; i < 1024; ++i) A2[i] = rand() % 3 - 1;
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float s = 0; for (int k = 0; k < 16; ++k) s += A[m * 16 + k] * B[k * 32 + n];
        X[m * 32 + n] = s;
    }
    for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
        float s = 0; for (int k = 0; k < 32; ++k) s += A2[m * 32 + k] * X[k * 32 + n];
        Y[m * 32 + n] = s;
    }
    float *dA, *dB, *dD, *dA2, *dD2;
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
    hipMalloc(&dA2, sizeof A2); hipMalloc(&dD2, sizeof D2);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipMemcpy(dA2, A2, sizeof A2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, dA2, dD2);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost); hipMemcpy(D2, dD2, sizeof D2, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0;
    for (int i = 0; i < 1024; ++i) { bad += D[i] != X[i]; bad2 += D2[i] != Y[i]; }
    printf("first product mismatches %d, chained product mismatches %d (of 1024)\n", bad, bad2);
    return bad || bad2;
}
