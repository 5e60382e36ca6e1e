// mfma_f64_allsum.hip -- can the sum over the 16 lanes of a row (one matrix of k_e16) be taken on the matrix cores?
// v_mfma_f64_4x4x4f64 works on four independent 4 x 4 blocks.  With B = 1 the first product leaves D1[i][j] = sum_k A[i][k] (row
// sums of the block the lane's value sits in), and with A = 1 the second one D2[i][j] = sum_k D1[k][j] -- IF a block is 16
// consecutive lanes and the accumulator's layout doubles as the B operand's (as it does for 16x16x4, mfma_f64_layout.hip).
// This prints both intermediate layouts, checks D2 against the 16-lane sums, and times a dependent chain of either form.
//
//   hipcc -O3 --offload-arch=gfx950 profiles/microbench/mfma_f64_allsum.hip -o profiles/microbench/mfma_f64_allsum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double allsum_mfma(const double v) {
    const double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    return __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov(const double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double allsum_dpp(double v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    return v;
}
__global__ void k_layout(double* out) {
    const int lane = threadIdx.x;
    const double v = (double)(1 << (lane & 15)) + 65536.0 * (lane >> 4);   // a bit per lane of the row: the sum names its members
    const double d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v, 1.0, 0.0, 0, 0, 0);
    const double d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(1.0, d1, 0.0, 0, 0, 0);
    out[lane] = d1;
    out[64 + lane] = d2;
    out[128 + lane] = allsum_dpp(v);
}
template <int KIND>
__global__ void k_chain(double* out, const int reps, long long* cyc) {
    double v = 1.0 + 1e-9 * threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const double s = KIND == 0 ? allsum_dpp(v) : allsum_mfma(v);
        v = fma(s, 1e-3, 1.0 + 1e-9 * threadIdx.x);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double* d;
    long long* c;
    hipMalloc(&d, 1 << 24);
    hipMalloc(&c, 8);
    double h[192];
    k_layout<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("lane  d1(row sums)  d2(all)  dpp\n");
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (l < 20 || l % 16 == 0) printf("%2d  %10.0f %10.0f %10.0f\n", l, h[l], h[64 + l], h[128 + l]);
        if (h[64 + l] != h[128 + l]) ++bad;
    }
    printf("lanes where the two-product sum differs from the DPP sum: %d\n", bad);
    for (int kind = 0; kind < 2; ++kind)
        for (int waves = 1; waves <= 4; waves += (waves == 1 ? 2 : 1)) {   // 1, 3, 4 wavefronts per SIMD
            const int reps = 20000;
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            auto go = [&]() {
                if (kind == 0) k_chain<0><<<256 * waves, 256>>>(d, reps, c);
                else k_chain<1><<<256 * waves, 256>>>(d, reps, c);
            };
            go();
            hipEventRecord(e0);
            go();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            long long cy;
            hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            printf("%s  %d wavefronts/SIMD: %.1f ns per dependent all-sum (+1 fma) per wavefront, %.1f cycles (counter)\n", kind ? "mfma" : "dpp ", waves,
                   ms * 1e6 / reps, (double)cy / reps);
        }
    return 0;
}
