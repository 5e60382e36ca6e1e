// Device-to-host copies of a large result: into pageable memory (a fresh numpy array), into pinned memory, and in 8 MB pieces.
// Build: hipcc --offload-arch=gfx950 -O2 -o pcie_d2h pcie_d2h.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    for (size_t mb : {16, 64, 256, 1024}) {
        const size_t bytes = mb << 20;
        void* d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
        void* pageable = malloc(bytes); memset(pageable, 0, bytes);
        void* pinned; hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
        hipDeviceSynchronize();
        double t0 = now_ms(); hipMemcpy(pageable, d, bytes, hipMemcpyDeviceToHost); double t1 = now_ms();
        hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost); double t2 = now_ms();
        hipMemcpy(pinned, d, bytes, hipMemcpyDeviceToHost); double t3 = now_ms();
        void* fresh = malloc(bytes);                       // untouched pages, like np.empty
        double t4 = now_ms(); hipMemcpy(fresh, d, bytes, hipMemcpyDeviceToHost); double t5 = now_ms();
        double t6 = now_ms(); void* p2; hipHostMalloc(&p2, bytes, hipHostMallocDefault); double t7 = now_ms();
        memcpy(pageable, pinned, bytes); double t8 = now_ms();
        printf("%5zu MB: pageable (touched) %.2f GB/s | pinned %.2f / %.2f GB/s | pageable (fresh) %.2f GB/s | hipHostMalloc %.2f ms | host memcpy pinned->pageable %.2f GB/s\n",
               mb, bytes / (t1 - t0) / 1e6, bytes / (t2 - t1) / 1e6, bytes / (t3 - t2) / 1e6, bytes / (t5 - t4) / 1e6, t7 - t6, bytes / (t8 - t7) / 1e6);
        hipFree(d); free(pageable); free(fresh); hipHostFree(pinned); hipHostFree(p2);
    }
    return 0;
}
