// Streaming-read ceiling past the 256 MiB last-level cache: every lane reads 16 B (or 2 x 16 B at a plane stride) per
// step, a workgroup-wide contiguous run per iteration; arg = MiB.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n, double* out) {
    double2 acc{0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = p[i];
        acc.x += v.x; acc.y += v.y;
    }
    if (acc.x == 1.2345e300) out[0] = acc.y;
}
// column-strip pattern of k_flux_rows: a wave owns 64 consecutive 32-B points of a row and walks `rows` rows down
__global__ __launch_bounds__(256) void k_strip(const double4* __restrict__ p, int ncol, int nrow, int ti, double* out) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int strips = (ncol + 63) / 64;
    const int trow = wave / strips, s = wave - trow * strips;
    const int c = min(s * 64 + lane, ncol - 1);
    double4 acc{0, 0, 0, 0};
    for (int r = trow * ti; r < min(nrow, (trow + 1) * ti); ++r) {
        const double4 v = p[(size_t)r * ncol + c];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x == 1.2345e300) out[0] = acc.y + acc.z + acc.w;
}
int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atol(argv[1]) : 512;
    const size_t bytes = mib << 20;
    void* buf; double* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8);
    hipMemset(buf, 0, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 8192, 32768}) {
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const double2*)buf, bytes / 16, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
        }
        printf("grid-stride read  %zu MiB  %5d blocks: %.1f us  %.2f TB/s\n", mib, blocks, best * 1e3, bytes / best / 1e9);
    }
    const int ncol = 4097, nrow = (int)(bytes / 32 / ncol);
    for (int ti : {4, 16, 33, 64, 256}) {
        const int strips = (ncol + 63) / 64, tiles = ((nrow + ti - 1) / ti) * strips;
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(k_strip, dim3((tiles + 3) / 4), dim3(256), 0, 0, (const double4*)buf, ncol, nrow, ti, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (rep && ms < best) best = ms;
        }
        printf("column strips (32 B/point, row of %d points) ti=%3d: %.1f us  %.2f TB/s\n", ncol, ti, best * 1e3, (double)nrow * ncol * 32 / best / 1e9);
    }
    return 0;
}
