#include <cstdlib>
// Variants of the streaming-write ceiling: plain vs nontemporal stores, grid-stride vs
// block-contiguous addressing (268 MB, 16 B per lane).
// hipcc --offload-arch=gfx950 -O3 stream_write_nt.hip -o stream_write_nt && ./stream_write_nt
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int NT, int CONTIG>
__global__ __launch_bounds__(256) void k_write(double* p, size_t n, double v) {
    // n = number of 16-byte elements
    size_t i, stride, end;
    if (CONTIG) {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        i = (size_t)blockIdx.x * per + threadIdx.x;
        end = (size_t)(blockIdx.x + 1) * per < n ? (size_t)(blockIdx.x + 1) * per : n;
        stride = 256;
    } else {
        i = (size_t)blockIdx.x * 256 + threadIdx.x;
        stride = (size_t)gridDim.x * 256;
        end = n;
    }
    for (; i < end; i += stride) {
        const double a = v + (double)i, b = v;
        if (NT) {
            __builtin_nontemporal_store(a, p + 2 * i);
            __builtin_nontemporal_store(b, p + 2 * i + 1);
        } else {
            p[2 * i] = a;
            p[2 * i + 1] = b;
        }
    }
}
template <int NT, int CONTIG>
static void run(double* p, size_t n, size_t bytes, int grid, const char* name) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_write<NT, CONTIG>), dim3(grid), dim3(256), 0, 0, p, n, 1.0 + rep);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep > 1 && ms < best) best = ms;
    }
    printf("%-28s grid %6d: %.1f us  %.2f TB/s\n", name, grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    // optional argument: buffer size in MiB (default 256, the size of the 2049 x 2049 Haldane grid)
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 256) << 20, n = bytes / 16;
    double* p;
    hipMalloc(&p, bytes);
    printf("buffer %zu MiB\n", bytes >> 20);
    for (int grid : {2048, 8192, 65536}) {
        run<0, 0>(p, n, bytes, grid, "plain grid-stride");
        run<1, 0>(p, n, bytes, grid, "nontemporal grid-stride");
        run<0, 1>(p, n, bytes, grid, "plain block-contiguous");
        run<1, 1>(p, n, bytes, grid, "nontemporal block-contiguous");
    }
    // empty-bracket overhead
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 20; ++rep) {
        hipEventRecord(a); hipLaunchKernelGGL((k_write<0, 0>), dim3(1), dim3(256), 0, 0, p, (size_t)256, 1.0);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("tiny kernel bracket: %.1f us\n", best * 1e3);
    return 0;
}
