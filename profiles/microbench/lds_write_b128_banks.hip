// Which lanes does ds_write_b128 / ds_read_b128 service together, i.e. which slot patterns conflict?  Times 4096 rounds of
// one wave-wide 16-byte LDS write (and read) per pattern; a conflict-free pattern is the baseline, an n-way one takes ~n x.
//   hipcc --offload-arch=gfx950 -O3 lds_write_b128_banks.hip -o lds_banks && ./lds_banks
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
// 16 wavefronts per workgroup (4 per SIMD), each with its own 1024-slot region; 8 independent accesses per round so that the
// LDS pipe, not the issue latency of one wave, sets the time
__global__ __launch_bounds__(1024) void k(const int* __restrict__ slot, int rounds, int do_read, double* out) {
    __shared__ v2d buf[16 * 640];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v2d* mine = buf + wave * 640;
    const int s = slot[lane];
    v2d v{(double)lane, 1.0};
    v2d a0{0, 0}, a1{0, 0}, a2{0, 0}, a3{0, 0};
    for (int r = 0; r < rounds; ++r) {
        if (do_read) {
            a0 += mine[s]; a1 += mine[s + 1]; a2 += mine[s + 2]; a3 += mine[s + 3];
            a0 += mine[s + 4]; a1 += mine[s + 5]; a2 += mine[s + 6]; a3 += mine[s + 7];
        } else {
            mine[s] = v; mine[s + 1] = v; mine[s + 2] = v; mine[s + 3] = v;
            mine[s + 4] = v; mine[s + 5] = v; mine[s + 6] = v; mine[s + 7] = v;
            asm volatile("" ::: "memory");
        }
    }
    a0 += a1 + a2 + a3;
    if (a0.x == 1.2345e300 || mine[lane].x == 1.2345e300) out[0] = a0.y;
}
static float run(const int* h, int do_read) {
    int* d; double* o;
    hipMalloc(&d, 64 * 4); hipMalloc(&o, 8);
    hipMemcpy(d, h, 64 * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, d, 1 << 12, do_read, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    hipFree(d); hipFree(o);
    return best;
}
int main() {
    int s[64];
    auto show = [&](const char* name) { printf("%-58s write %.3f ms   read %.3f ms\n", name, run(s, 0), run(s, 1)); };
    for (int l = 0; l < 64; ++l) s[l] = l;                                   show("slot = lane (contiguous)");
    for (int l = 0; l < 64; ++l) s[l] = 2 * l;                               show("slot = 2 lane (k_grid_rows N=2, o=0, unswizzled)");
    for (int l = 0; l < 64; ++l) s[l] = 4 * l;                               show("slot = 4 lane (N=4 unswizzled)");
    for (int l = 0; l < 64; ++l) { int t = 2 * l; s[l] = t ^ ((t >> 3) & 3); } show("slot = swz(2 lane), swz(s) = s ^ ((s>>3)&3)");
    for (int l = 0; l < 64; ++l) { int t = 4 * l; s[l] = t ^ ((t >> 3) & 3); } show("slot = swz(4 lane)");
    for (int l = 0; l < 64; ++l) { int t = 2 * l; s[l] = t ^ ((t >> 4) & 1); } show("slot = 2 lane ^ bit4>>4");
    for (int l = 0; l < 64; ++l) { int t = 2 * l; s[l] = t ^ ((t >> 5) & 1); } show("slot = 2 lane ^ bit5>>5");
    for (int l = 0; l < 64; ++l) { int t = 2 * l; s[l] = t + ((t >> 3) & 1); } show("slot = 2 lane + ((s>>3)&1)");
    for (int l = 0; l < 64; ++l) s[l] = 2 * l + (l >> 2 & 1);                 show("slot = 2 lane + (lane>>2 & 1)");
    for (int l = 0; l < 64; ++l) s[l] = 2 * l + (l >> 3 & 1);                 show("slot = 2 lane + (lane>>3 & 1)");
    for (int l = 0; l < 64; ++l) s[l] = 2 * l + (l >> 4 & 1);                 show("slot = 2 lane + (lane>>4 & 1)");
    for (int l = 0; l < 64; ++l) s[l] = (l & 31) * 2 + (l >> 5);              show("slot = 2 (lane&31) + lane>>5  (interleave halves)");
    for (int l = 0; l < 64; ++l) { int t = l; s[l] = t ^ ((t >> 3) & 3); }     show("slot = swz(lane)       (k_grid_rows staging READ, i = 0)");
    for (int l = 0; l < 64; ++l) { int t = 64 + l; s[l] = (t ^ ((t >> 3) & 3)) - 64; } show("slot = swz(64 + lane) - 64 (staging READ, i = 1)");
    for (int l = 0; l < 64; ++l) { int t = l; s[l] = t ^ ((t >> 4) & 1) * 1; } show("slot = lane ^ (lane>>4 & 1)");
    for (int l = 0; l < 64; ++l) { int t = l; s[l] = t ^ (((t >> 3) & 1) << 0); } show("slot = lane ^ (lane>>3 & 1)");
    for (int l = 0; l < 64; ++l) { int t = l; s[l] = t ^ (((t >> 3) & 2)); } show("slot = lane ^ (lane>>3 & 2)");
    for (int l = 0; l < 64; ++l) s[l] = 17 * l;                              show("slot = 17 lane (odd stride)");
    for (int l = 0; l < 64; ++l) s[l] = 8 * l;                               show("slot = 8 lane (all lanes one bank group)");
    return 0;
}
