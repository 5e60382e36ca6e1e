// Which lane does each DPP control read from?  (row ops act inside each 16-lane row of a wavefront)
// hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe && ./dpp_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    out[lane] = __builtin_amdgcn_update_dpp(-1, lane, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
void show(const char* name, int* d) {
    int h[64];
    hipLaunchKernelGGL(k<CTRL>, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-16s", name);
    for (int i = 0; i < 20; ++i) printf(" %2d", h[i]);
    printf(" ...\n");
}
__global__ void kb(int* out) {
    const int lane = threadIdx.x;
    const int src = (lane & 48) | ((5 - lane) & 15);
    out[lane] = __builtin_amdgcn_ds_bpermute(src * 4, lane);
}
int main() {
    int* d;
    hipMalloc(&d, 64 * sizeof(int));
    show<0x140>("row_mirror", d);
    show<0x141>("row_half_mirror", d);
    show<0x121>("row_ror:1", d);
    show<0x123>("row_ror:3", d);
    show<0x128>("row_ror:8", d);
    show<0x101>("row_shl:1", d);
    show<0x111>("row_shr:1", d);
    show<0x150>("row_newbcast:0", d);
    show<0x155>("row_newbcast:5", d);
    show<0xB1>("quad_perm 1032", d);
    int h[64];
    hipLaunchKernelGGL(kb, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-16s", "bpermute (5-l)");
    for (int i = 0; i < 20; ++i) printf(" %2d", h[i]);
    printf(" ...\n");
    return 0;
}
