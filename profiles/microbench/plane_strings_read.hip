// What the link kernels of 5..8 wide bands (k_chain_links_wave / _tile, tbk_berry.hip) could read at best: the same address
// stream without LDS, arithmetic or stores.  Array [planes][npts][16 components] of c128 (band-major, like a wf_array of 16-orbital
// states); a wavefront walks ONE string of `len` consecutive points and reads the 256-byte rows of `nocc` planes of every point.
//   mode 0: one plain coalesced pass over the same bytes (reference)
//   mode 1: G points per step (nocc * G * 256 B in flight per wavefront), G = 1, 2, 4, 8
// hipcc --offload-arch=gfx950 -O3 plane_strings_read.hip -o plane_strings_read && ./plane_strings_read [side]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_plain(const v2d* p, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    v2d acc = {0.0, 0.0};
    for (; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc.x == 1.2345e300) out[0] = acc.y;
}
template <int G>
__global__ __launch_bounds__(256) void k_strings(const v2d* data, long npts, int len, long nstrings, int nocc, double* out) {
    const int lane = threadIdx.x & 63;
    const long s = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= nstrings) return;
    // element e = j * 64 + lane of a point (nocc * 16 elements): plane e / 16, component e % 16
    const int nld = nocc * 16 / 64;                   // 2 for 8 bands
    v2d acc = {0.0, 0.0};
    for (int i = 0; i < len; i += G) {
        v2d x[G][2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long pt = s * len + min(i + g, len - 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = j * 64 + lane, a = e >> 4, c = e & 15;
                x[g][j] = j < nld ? data[((long)a * npts + pt) * 16 + c] : v2d{0.0, 0.0};
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) acc += x[g][0] + x[g][1];
    }
    if (acc.x == 1.2345e300) out[0] = acc.y;
}
int main(int argc, char** argv) {
    const int side = argc > 1 ? atoi(argv[1]) : 129;
    const int nocc = 8, planes = 8;
    const long npts = (long)side * side * side, nstrings = (long)side * side;
    const size_t bytes = (size_t)planes * npts * 256;
    v2d* p; if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc of %zu bytes failed\n", bytes); return 1; }
    hipMemset(p, 0, bytes);
    double* out; hipMalloc(&out, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            const dim3 g((unsigned)((nstrings + 3) / 4));
            if (mode == 0) hipLaunchKernelGGL(k_plain, dim3(65536), dim3(256), 0, 0, p, bytes / 16, out);
            else if (mode == 1) hipLaunchKernelGGL(k_strings<1>, g, dim3(256), 0, 0, p, npts, side, nstrings, nocc, out);
            else if (mode == 2) hipLaunchKernelGGL(k_strings<2>, g, dim3(256), 0, 0, p, npts, side, nstrings, nocc, out);
            else if (mode == 3) hipLaunchKernelGGL(k_strings<4>, g, dim3(256), 0, 0, p, npts, side, nstrings, nocc, out);
            else hipLaunchKernelGGL(k_strings<8>, g, dim3(256), 0, 0, p, npts, side, nstrings, nocc, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("side %d mode %d (%s): %.3f ms  %.2f TB/s\n", side, mode, mode == 0 ? "plain" : mode == 1 ? "1 point / step" : mode == 2 ? "2 points / step" : mode == 3 ? "4 points / step" : "8 points / step",
               best, bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
