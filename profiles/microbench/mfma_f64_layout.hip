// v_mfma_f64_16x16x4_f64 operand / result lane maps on gfx950, checked with exact integer data (asymmetric A and B).
//   A[i][k]: lane l supplies A[l & 15][4 kb + (l >> 4)],  B[k][j]: lane l supplies B[4 kb + (l >> 4)][l & 15]
//   D[i][j]: lane l, register r holds D[(l >> 4) + 4 r][l & 15]
// build: hipcc -O3 --offload-arch=gfx950 mfma_f64_layout.hip -o mfma_f64_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, double* D, long long* cyc) {
    const int l = threadIdx.x;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int kb = 0; kb < 4; ++kb) {
        const double a = A[(l & 15) * 16 + 4 * kb + (l >> 4)];
        const double b = B[(4 * kb + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
    // issue rate: 64 back-to-back MFMAs on 4 independent accumulators
    d4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    const double a = A[l], b = B[l];
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < 16; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    asm volatile("s_nop 0" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0) cyc[0] = t1 - t0;
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) D[0] = 0.0;
}
int main() {
    double A[256], B[256], D[256], R[256];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            A[i * 16 + j] = (double)((i * 7 + j * 3) % 11 - 5);
            B[i * 16 + j] = (double)((i * 5 + j * 13) % 17 - 8);
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = 0;
            for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 16 + j];
            R[i * 16 + j] = s;
        }
    double *dA, *dB, *dD;
    long long* dc;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D)); hipMalloc(&dc, 8);
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice);
    hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dc);
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    long long cyc = 0;
    hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("mfma_f64_16x16x4: %d of 256 elements wrong; 64 MFMAs in %lld cycles (%.1f per MFMA)\n", bad, cyc, cyc / 64.0);
    return bad != 0;
}
