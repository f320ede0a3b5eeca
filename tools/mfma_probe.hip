// Probe of the v_mfma_f64_16x16x4_f64 operand / accumulator lane maps (exact integer data, asymmetric operands).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4 row-major*/, const double* Bm /*4x16 row-major*/, double* D /*16x16 row-major*/) {
    const int l = threadIdx.x, lg = l >> 4, jj = l & 15;
    d4 c = {0, 0, 0, 0};
    const double a = A[jj * 4 + lg];      // A[i = l&15][k = l>>4]
    const double b = Bm[lg * 16 + jj];    // B[k = l>>4][j = l&15]
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(lg + 4 * r) * 16 + jj] = c[r];   // row = lg + 4*reg, col = jj
}
int main() {
    double A[64], B[64], D[256], *dA, *dB, *dD;
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = 1 + i + 100 * k;
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = 3 + 7 * j + 1000 * k;
    hipMalloc(&dA, sizeof A); hipMalloc(&dB, sizeof B); hipMalloc(&dD, sizeof D);
    hipMemcpy(dA, A, sizeof A, hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof B, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j];
        if (s != D[i * 16 + j]) { if (bad < 5) printf("mismatch (%d,%d): %.0f vs %.0f\n", i, j, D[i * 16 + j], s); ++bad; }
    }
    printf("mfma_f64_16x16x4 layout probe: %d mismatches\n", bad);
    return bad != 0;
}
