// Ceiling check: how fast can MI355X stream-WRITE / stream-READ 268 MB with 16 B per lane?
// hipcc --offload-arch=gfx950 -O3 stream_write.hip -o stream_write && ./stream_write
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_write(double2* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n; i += stride) p[i] = double2{v + (double)i, v};
}
__global__ __launch_bounds__(256) void k_read(const double2* p, size_t n, double* out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    double acc = 0.0;
    for (; i < n; i += stride) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}
int main() {
    const size_t bytes = 268435456, n = bytes / 16;
    double2* p; double* o;
    hipMalloc(&p, bytes); hipMalloc(&o, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int grid : {2048, 8192, 65536}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, p, n, 1.0 + rep);
                else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, n, o);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 1 && ms < best) best = ms;
            }
            printf("%s grid %6d: %.1f us  %.2f TB/s\n", mode ? "read " : "write", grid, best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
