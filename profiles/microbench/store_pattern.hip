// The store stream of k_grid_rows (n = 2, 2049 x 2049 mesh) without any arithmetic: which feature of the
// pattern costs the 20 % against a plain streaming write?
//   mode 0: plain grid-stride stream (reference)
//   mode 1: the kernel's tiling (wave = 7 consecutive 64-point chunks of one mesh row, two band planes,
//           1 KiB per store instruction), data straight from registers
//   mode 2: mode 1 + the LDS staging round trip
//   mode 3: mode 1 with one plane only (same bytes: points doubled) -- are two streams per wave the problem?
//   mode 4: mode 2 + the three 16-byte table loads per lane and chunk (L2-resident tables)
//   mode 5: mode 4 + a dependent start-up chain per tile (three dependent loads and a barrier)
//   mode 8: mode 1 with every store shifted by one mesh point (32 bytes): 1 KiB stores that start in the middle of a cache line,
//           as tiles overlapping their left neighbour by one column would issue them
//   mode 7: mode 1 without the transposition: a lane writes its OWN point's 32 bytes as two 16-byte stores, i.e. every store
//           instruction covers 2 KiB with a 32-byte lane stride (half of every cache line per instruction, no LDS staging)
// hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
struct cd { double x, y; };
template <int MODE>
__global__ __launch_bounds__(256) void k_pat(cd* data, int nrow, int nlast, int seg, int tpr, long ntiles, double v, const cd* tab, const int* chain) {
    __shared__ cd stage_all[4 * 128];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    cd* stage = stage_all + wib * 128;
    const long tile = (long)blockIdx.x * 4 + wib;
    if (tile >= ntiles) return;
    const int cpr = (nlast + 63) / 64;
    const long row = tile / tpr;
    const int ts = (int)(tile - row * tpr);
    const int jc0 = ts * seg, jc1 = min(jc0 + seg, cpr);
    const long npts = (long)nrow * nlast;
    double extra = 0.0;
    if (MODE == 5) {
        int i0 = chain[lane & 7];
        int i1 = chain[8 + (i0 & 7)];
        int i2 = chain[16 + (i1 & 7)];
        extra = (double)i2 * 1e-300;
        __syncthreads();
    }
    for (int jc = jc0; jc < jc1; ++jc) {
        cd t0{0, 0}, t1{0, 0}, t2{0, 0};
        if (MODE >= 4) {
            const int jj = min(jc * 64 + lane, nlast - 1);
            t0 = tab[jj];
            t1 = tab[4096 + 2 * jj];
            t2 = tab[4096 + 2 * jj + 1];
        }
        const int nvalid = min(64, nlast - jc * 64) * 2;
        const long point0 = row * nlast + (long)jc * 64;
        for (int r = 0; r < 2; ++r) {
            cd* dst = MODE == 3 ? data + (point0 * 2 + (long)r * 64) * 2 : data + ((long)r * npts + point0) * 2;
            if (MODE == 8) dst += 2 * 64 - 2;     // (one point to the left of the next chunk: stays inside the allocation)
            if (MODE == 2 || MODE >= 4) {
                stage[lane * 2 + 0] = cd{v + lane + t0.x + extra, v + t1.y};
                stage[lane * 2 + 1] = cd{v + t2.x, v + lane + r + t0.y};
                __builtin_amdgcn_s_waitcnt(0xc07f);
            }
            if (MODE == 7) {
                if (2 * lane < nvalid) {
                    dst[2 * lane] = cd{v + lane, v + r};
                    dst[2 * lane + 1] = cd{v - lane, v + r};
                }
                continue;
            }
            for (int i = 0; i < 2; ++i) {
                const int e = i * 64 + lane;
                if (e < nvalid) dst[e] = (MODE == 2 || MODE >= 4) ? stage[e] : cd{v + e, v + r};
            }
        }
    }
}
// mode 6: like mode 4, but the tables of chunk jc+1 are loaded before the stores of chunk jc and full
// chunks store unconditionally (so that the compiler can wait with vmcnt(4) and keep the stores in flight)
__global__ __launch_bounds__(256) void k_pre(cd* data, int nrow, int nlast, int seg, int tpr, long ntiles, double v, const cd* tab) {
    __shared__ cd stage_all[4 * 128];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    cd* stage = stage_all + wib * 128;
    const long tile = (long)blockIdx.x * 4 + wib;
    if (tile >= ntiles) return;
    const int cpr = (nlast + 63) / 64;
    const long row = tile / tpr;
    const int ts = (int)(tile - row * tpr);
    const int jc0 = ts * seg, jc1 = min(jc0 + seg, cpr);
    const long npts = (long)nrow * nlast;
    cd n0, n1, n2;
    {
        const int jj = min(jc0 * 64 + lane, nlast - 1);
        n0 = tab[jj]; n1 = tab[4096 + 2 * jj]; n2 = tab[4096 + 2 * jj + 1];
        asm volatile("" ::"v"(n0.x), "v"(n0.y), "v"(n1.x), "v"(n1.y), "v"(n2.x), "v"(n2.y));
    }
    const int jfull = max(jc0, min(jc1, nlast / 64));
    auto body = [&](const int jc, const bool full) {
        const cd t0 = n0, t1 = n1, t2 = n2;
        if (jc + 1 < jc1) {
            const int jj = min((jc + 1) * 64 + lane, nlast - 1);
            n0 = tab[jj]; n1 = tab[4096 + 2 * jj]; n2 = tab[4096 + 2 * jj + 1];
        }
        const int nvalid = min(64, nlast - jc * 64) * 2;
        const long point0 = row * nlast + (long)jc * 64;
        for (int r = 0; r < 2; ++r) {
            cd* dst = data + ((long)r * npts + point0) * 2;
            stage[lane * 2 + 0] = cd{v + lane + t0.x, v + t1.y};
            stage[lane * 2 + 1] = cd{v + t2.x, v + lane + r + t0.y};
            __builtin_amdgcn_s_waitcnt(0xc07f);
            for (int i = 0; i < 2; ++i) {
                const int e = i * 64 + lane;
                if (full) dst[e] = stage[e];
                else if (e < nvalid) dst[e] = stage[e];
            }
        }
    };
    for (int jc = jc0; jc < jfull; ++jc) body(jc, true);
    for (int jc = jfull; jc < jc1; ++jc) body(jc, false);
}
__global__ __launch_bounds__(256) void k_plain(cd* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) p[i] = cd{v + (double)i, v};
}
int main() {
    const int nrow = 2049, nlast = 2049;
    const size_t bytes = (size_t)nrow * nlast * 64;
    cd* p; hipMalloc(&p, bytes + (1 << 20));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    cd* tab; hipMalloc(&tab, 16 * 3 * 4096); hipMemset(tab, 0, 16 * 3 * 4096);
    int* chain; hipMalloc(&chain, 256); hipMemset(chain, 0, 256);
    for (int mode = 0; mode < 9; ++mode) {
        for (int seg : {7, 9}) {
            const int cpr = (nlast + 63) / 64, tpr = (cpr + seg - 1) / seg, s2 = (cpr + tpr - 1) / tpr;
            const long ntiles = (long)nrow * tpr;
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_plain, dim3(65536), dim3(256), 0, 0, p, bytes / 16, 1.0 + rep);
                else if (mode == 1) hipLaunchKernelGGL(k_pat<1>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 2) hipLaunchKernelGGL(k_pat<2>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 3) hipLaunchKernelGGL(k_pat<3>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 4) hipLaunchKernelGGL(k_pat<4>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 8) hipLaunchKernelGGL(k_pat<8>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 7) hipLaunchKernelGGL(k_pat<7>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                else if (mode == 6) hipLaunchKernelGGL(k_pre, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab);
                else hipLaunchKernelGGL(k_pat<5>, dim3((ntiles + 3) / 4), dim3(256), 0, 0, p, nrow, nlast, s2, tpr, ntiles, 1.0 + rep, tab, chain);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep > 1 && ms < best) best = ms;
            }
            printf("mode %d seg %d: %.1f us  %.2f TB/s\n", mode, s2, best * 1e3, bytes / (best * 1e-3) / 1e12);
            if (mode == 0) break;
        }
    }
    return 0;
}
