// Relative error of the gfx950 hardware estimates v_rsq_f64 / v_rcp_f64 / v_sqrt_f64, raw and after Newton steps
// (decides how many refinement steps the QL rotation recurrence needs).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double t = x[i];
    double y0 = __builtin_amdgcn_rsq(t);
    double h = 0.5 * t;
    double y1 = y0 * fma(-h * y0, y0, 1.5);
    double y2 = y1 * fma(-h * y1, y1, 1.5);
    o[i] = y0; o[n + i] = y1; o[2 * n + i] = y2;
    o[3 * n + i] = __builtin_amdgcn_rcp(t);
    o[4 * n + i] = __builtin_amdgcn_sqrt(t);
    // one-step variant with a residual-form correction: y1' = y0 + y0 * (0.5 - h y0^2)
    double e = fma(-h * y0, y0, 0.5);
    o[5 * n + i] = fma(y0, e, y0);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (s >> 11) * (1.0 / 9007199254740992.0);
        x[i] = std::exp((u - 0.5) * 60.0);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        long double t = x[i], r = 1.0L / sqrtl(t);
        e[0] = fmax(e[0], (double)fabsl((o[i] - r) / r));
        e[1] = fmax(e[1], (double)fabsl((o[n + i] - r) / r));
        e[2] = fmax(e[2], (double)fabsl((o[2 * n + i] - r) / r));
        e[3] = fmax(e[3], (double)fabsl((o[3 * n + i] - 1.0L / t) * t));
        e[4] = fmax(e[4], (double)fabsl((o[4 * n + i] - sqrtl(t)) / sqrtl(t)));
        e[5] = fmax(e[5], (double)fabsl((o[5 * n + i] - r) / r));
    }
    printf("rsq raw %.3e  newton1 %.3e  newton2 %.3e | rcp raw %.3e | sqrt raw %.3e | rsq newton1(residual form) %.3e\n", e[0], e[1], e[2], e[3], e[4], e[5]);
    return 0;
}
