// How long does a software grid barrier (one atomic arrival counter + generation flag in global memory)
// take with 32..256 co-resident workgroups?  Bounded spins: a lost workgroup ends the kernel with an
// error flag instead of hanging the device.
// hipcc --offload-arch=gfx950 -O3 grid_barrier.hip -o grid_barrier && ./grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ bool grid_sync(unsigned* count, volatile unsigned* gen, unsigned nblocks, unsigned& my_gen, int* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned target = my_gen + 1;
        __threadfence();
        const unsigned arrived = atomicAdd(count, 1u) + 1u;
        if (arrived == nblocks * target) {
            __hip_atomic_store((unsigned*)gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            long spins = 0;
            while (__hip_atomic_load((unsigned*)gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > 20000000L) { *err = 1; ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        my_gen = target;
    }
    __syncthreads();
    return ok;
}
__global__ __launch_bounds__(256) void k_bar(unsigned* count, unsigned* gen, int rounds, int* err, double* sink) {
    unsigned my_gen = 0;
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        acc += r * 1e-9;
        if (!grid_sync(count, gen, gridDim.x, my_gen, err)) break;
    }
    if (acc == 1.2345) sink[0] = acc;
}
int main() {
    unsigned *count, *gen; int* err; double* sink;
    hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&err, 4); hipMalloc(&sink, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nb : {16, 32, 64, 128, 256}) {
        const int rounds = 2000;
        float best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(count, 0, 4); hipMemset(gen, 0, 4); hipMemset(err, 0, 4);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_bar, dim3(nb), dim3(256), 0, 0, count, gen, rounds, err, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("%3d workgroups: %.2f us per barrier%s\n", nb, best * 1e3 / rounds, e ? "  (TIMED OUT)" : "");
    }
    return 0;
}
