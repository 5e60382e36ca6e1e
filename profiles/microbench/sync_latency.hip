// What one "launch a small kernel and wait for it" costs on the host, three ways (round 4, the python two-call step):
//   hipStreamSynchronize as the runtime is configured by default | after hipSetDeviceFlags(hipDeviceScheduleSpin) |
//   the kernel's last store sets a sequence word in mapped host memory that the host polls.
// Build: hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip ; run: ./sync_latency [spin]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <vector>

__global__ void k_work(double* out, int iters, unsigned* flag, unsigned seq) {
    double x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
    const bool spin = argc > 1 && !strcmp(argv[1], "spin");
    if (spin) hipSetDeviceFlags(hipDeviceScheduleSpin);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double* out;
    hipMalloc(&out, 256 * 64 * sizeof(double));
    unsigned* flag;
    hipHostMalloc(&flag, 64, hipHostMallocMapped);
    *flag = 0;
    unsigned* dflag;
    hipHostGetDevicePointer((void**)&dflag, flag, 0);
    for (int iters : {1, 20000}) {          // ~2 us and ~50 us kernels
        for (int mode = 0; mode < 2; ++mode) {
            std::vector<double> t;
            unsigned seq = 0;
            for (int rep = 0; rep < 400; ++rep) {
                ++seq;
                const double t0 = now_us();
                k_work<<<1, 64, 0, s>>>(out, iters, mode ? dflag : nullptr, seq);
                if (mode == 0) hipStreamSynchronize(s);
                else {
                    volatile unsigned* f = flag;
                    while (*f != seq) { }
                }
                t.push_back(now_us() - t0);
            }
            hipStreamSynchronize(s);
            std::sort(t.begin(), t.end());
            printf("%s iters=%d %s: median %.2f us, p10 %.2f, p90 %.2f\n", spin ? "ScheduleSpin" : "default", iters,
                   mode ? "poll mapped flag" : "hipStreamSynchronize", t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10]);
        }
    }
    // two dependent kernels vs one (what a folded reduction saves)
    for (int nk : {1, 2}) {
        std::vector<double> t;
        for (int rep = 0; rep < 400; ++rep) {
            const double t0 = now_us();
            for (int i = 0; i < nk; ++i) k_work<<<1, 64, 0, s>>>(out, 1, nullptr, 0);
            hipStreamSynchronize(s);
            t.push_back(now_us() - t0);
        }
        std::sort(t.begin(), t.end());
        printf("%s %d small kernels + sync: median %.2f us\n", spin ? "ScheduleSpin" : "default", nk, t[t.size() / 2]);
    }
    return 0;
}
