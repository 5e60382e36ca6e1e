// tbk_solve.hip -- H(k) assembly + batched Hermitian eigen-solve for gfx950.
//
// Reproduces tb_model._gen_ham (pythtb.py:874-925), _sol_ham (:927-953) with
// _nicefy_eig (:3765-3775), their loop in solve_all (:1047-1060) and the mesh
// loop + impose_pbc of wf_array.solve_on_grid (:2475-2497, :2729-2747).
//
// Formulation (DESIGN.md "Kernels"): H(k) = D(k)^+ S(k) D(k) with
//   S_ab(k) = sum_t amp_t z_1^R1 ... z_d^Rd,   z_j = exp(2 pi i k_j)
//   D = diag(exp(2 pi i k.tau_a))
// so the eigenvalues are those of S and the eigenvectors are D^+ times those of
// S: d + norb sincospi per k instead of one per hopping, and S is assembled
// from wave-uniform (scalar) table reads.
//
//   nsta <= 4 : one thread per k, S and V in registers, cyclic Jacobi
//               (nsta == 2: the single exact rotation).
//   nsta  > 4 : one 64-lane wavefront per k, S and V in LDS, parallel-ordered
//               (round-robin) Jacobi, nsta/2 disjoint rotations per round.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "tbk_internal.h"

#define TBK_JACOBI_MAX_SWEEPS 30

struct GridArgs {
    WfsView wv;
    double start_k[TBK_MAX_DIM];
    int gmesh[TBK_MAX_DIM];  // global mesh sizes (axis 0 may exceed the slab)
    int64_t off[TBK_MAX_DIM];  // global index of the window's first point along each axis
    const cd* pbc;           // [TBK_MAX_DIM][nsta]
    unsigned long long* gaps;       // [TBK_GAP_SHARDS][nsta] min-reduced by this launch
    unsigned long long* gaps_next;  // the other parity: re-armed (+inf) for the next launch
    // per-axis tables of the regular mesh: z[d][i] = exp(2 pi i k_d(i)),
    // f[d][i*n+o] = exp(-2 pi i k_d(i) tau_o,d) * (pbc phase if i is the periodic image)
    const cd* tz[TBK_MAX_DIM];
    const cd* tf[TBK_MAX_DIM];
    int last;                // index of the last (fastest) mesh axis
    int cpr;                 // 64-point chunks per mesh row (row = all leading axes)
    int64_t nchunks;
    int wnchunk;             // k_solve_wave: aligned chains per mesh row that touch the window
    int64_t wcfirst;         // ... and the global number of the first of them
    int seg;                 // chunks per wave tile (k_grid_rows)
    int tpr;                 // wave tiles per row
    int64_t ntiles;
    int ablate;              // diagnostics only (TBK_ABLATE_GRID): 1 = no stores, 2 = no eigen-solve
};

struct ListArgs {
    const double* k;  // [nk][dim_k]
    const cd* ham;    // [nk][n][n] (eigh of supplied matrices) or null
    double* eval;     // [n][nk]
    cd* evec;         // [n][nk][n] or null
};

__device__ __forceinline__ cd expi2pi(double x) {
    double s, c;
    sincospi(2.0 * x, &s, &c);
    return cd{c, s};
}

// exp(2 pi i k.R) from the per-dimension unit phases; R is wave-uniform, so the
// loops below are scalar-controlled (no divergence, no indexed registers).
__device__ __forceinline__ cd phase_of_R(const cd (&z)[4], const int4 R) {
    cd e{1.0, 0.0};
    const int r[4] = {R.x, R.y, R.z, R.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int m = r[d];
        cd zz = z[d];
        if (m < 0) {
            m = -m;
            zz.y = -zz.y;
        }
        for (int q = 0; q < m; ++q) e = cmul(e, zz);
    }
    return e;
}

__device__ __forceinline__ cd slot_sum(const ModelView& mv, int slot, const cd (&z)[4]) {
    const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
    cd acc{0.0, 0.0};
    for (int t = t0; t < t1; ++t) cfma(acc, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
    return acc;
}

__device__ __forceinline__ double kdot(const double (&kk)[4], const double4 tau) {
    return kk[0] * tau.x + kk[1] * tau.y + kk[2] * tau.z + kk[3] * tau.w;
}

// Decode a row-major mesh index; returns the reduced k of that point and which
// axes are the periodic image (index == N-1 -> solved at index 0).
__device__ __forceinline__ void grid_point(const GridArgs& G, int64_t id, double (&kk)[4],
                                           bool (&wrap)[4]) {
    int ii[4] = {0, 0, 0, 0};
    if (G.wv.npts < (int64_t)0xffffffffu) {
        unsigned rem = (unsigned)id;
#pragma unroll
        for (int d = 3; d >= 1; --d) {
            const unsigned md = (unsigned)G.wv.mesh[d];
            if (md > 1) {
                const unsigned q = rem / md;
                ii[d] = (int)(rem - q * md);
                rem = q;
            }
        }
        ii[0] = (int)rem;
    } else {
        int64_t rem = id;
#pragma unroll
        for (int d = 3; d >= 1; --d) {
            const int64_t md = G.wv.mesh[d];
            if (md > 1) {
                const int64_t q = rem / md;
                ii[d] = (int)(rem - q * md);
                rem = q;
            }
        }
        ii[0] = (int)rem;
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        kk[d] = 0.0;
        wrap[d] = false;
        if (d < G.wv.dim_arr) {
            int64_t g = ii[d] + G.off[d];
            const int nd = G.gmesh[d];
            if (g == nd - 1) {
                g = 0;
                wrap[d] = true;
            }
            // kpt = start_k + float(i)/float(N-1)      (pythtb.py:2477,2490-2491)
            kk[d] = G.start_k[d] + (double)g / (double)(nd - 1);
        }
    }
}

// Same, from the local row (all leading axes, row-major) and the GLOBAL index along the
// last axis (which may lie before the window: chain predecessors that are not stored).
__device__ __forceinline__ void grid_point_rowcol(const GridArgs& G, int64_t row, int64_t g_last,
                                                  double (&kk)[4], bool (&wrap)[4]) {
    int64_t gi[4] = {0, 0, 0, 0};
    int64_t rem = row;
#pragma unroll
    for (int d = 2; d >= 0; --d) {
        if (d < G.last) {
            const int64_t md = G.wv.mesh[d];
            const int64_t q = rem / md;
            gi[d] = rem - q * md + G.off[d];
            rem = q;
        }
    }
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        kk[d] = 0.0;
        wrap[d] = false;
        if (d <= G.last) {
            int64_t g = d == G.last ? g_last : gi[d];
            const int nd = G.gmesh[d];
            if (g == nd - 1) {
                g = 0;
                wrap[d] = true;
            }
            kk[d] = G.start_k[d] + (double)g / (double)(nd - 1);
        }
    }
}
#include "tbk_solve_row16.inl"
#include <stdio.h>
#include <vector>
#include <complex>
template <int R>
__global__ void k_dbg(const cd* in, cd* out, cd* vout) {
    const int lane = threadIdx.x & 63, x = lane & 15;
    cd a[16], v[16];
    for (int c = 0; c < 16; ++c) { a[c] = in[x * 16 + c]; v[c] = cd{c == x ? 1.0 : 0.0, 0.0}; }
    if (R >= 0) row16_round<(R >= 0 ? R : 0), true>(a, v, x, (lane & 48) * 4);
    else for (int sw = 0; sw < -R; ++sw) row16_sweep<0, true>(a, v, x, (lane & 48) * 4);
    if (lane < 16) for (int c = 0; c < 16; ++c) { out[x * 16 + c] = a[c]; vout[x * 16 + c] = v[c]; }
}
int main() {
    const int n = 16;
    std::vector<std::complex<double>> A(n * n), B(n * n), V(n * n);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
    for (int i = 0; i < n; ++i) for (int j = i; j < n; ++j) { std::complex<double> z(rnd(), i == j ? 0.0 : rnd()); A[i * n + j] = z; A[j * n + i] = std::conj(z); }
    cd *din, *dout, *dv;
    hipMalloc(&din, n * n * 16); hipMalloc(&dout, n * n * 16); hipMalloc(&dv, n * n * 16);
    hipMemcpy(din, A.data(), n * n * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_dbg<3>, dim3(1), dim3(64), 0, 0, din, dout, dv);
    hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(V.data(), dv, n * n * 16, hipMemcpyDeviceToHost);
    // host emulation of round R = 3
    const int R = 3;
    int P[8], Q[8];
    P[0] = 15; Q[0] = R;
    for (int l = 1; l < 8; ++l) { P[l] = (R + l) % 15; Q[l] = (R - l + 15) % 15; }
    std::vector<std::complex<double>> C = A, D;
    double cc[8]; std::complex<double> ss[8];
    for (int l = 0; l < 8; ++l) {
        auto g = A[P[l] * n + Q[l]]; double g2 = std::norm(g);
        double h = 0.5 * (A[Q[l] * n + Q[l]].real() - A[P[l] * n + P[l]].real()), ah = fabs(h), r = sqrt(h * h + g2), inv = 1.0 / sqrt(2 * r * (r + ah)), sg = copysign(1.0, h);
        cc[l] = (ah + r) * inv; ss[l] = sg * g * inv;
    }
    for (int l = 0; l < 8; ++l) for (int r = 0; r < n; ++r) { auto x = C[r * n + P[l]], y = C[r * n + Q[l]]; C[r * n + P[l]] = cc[l] * x - std::conj(ss[l]) * y; C[r * n + Q[l]] = ss[l] * x + cc[l] * y; }
    D = C;
    for (int l = 0; l < 8; ++l) for (int c = 0; c < n; ++c) { D[P[l] * n + c] = cc[l] * C[P[l] * n + c] - ss[l] * C[Q[l] * n + c]; D[Q[l] * n + c] = std::conj(ss[l]) * C[P[l] * n + c] + cc[l] * C[Q[l] * n + c]; }
    double worst = 0; int wi = 0;
    for (int i = 0; i < n * n; ++i) { double e = std::abs(D[i] - B[i]); if (e > worst) { worst = e; wi = i; } }
    printf("max |A_gpu - A_host| after one round = %.3e at (%d,%d)\n", worst, wi / n, wi % n);
    for (int r = 0; r < n; ++r) { double e = 0; for (int c = 0; c < n; ++c) e = fmax(e, std::abs(D[r * n + c] - B[r * n + c])); printf("row %2d err %.2e   ", r, e); if (r % 4 == 3) printf("\n"); }
    double offd = 0; for (int l = 0; l < 8; ++l) offd = fmax(offd, std::abs(B[P[l] * n + Q[l]]));
    printf("max |A_gpu[p][q]| over the rotated pairs = %.3e\n", offd);
    auto offnorm = [&](std::vector<std::complex<double>>& M) { double o = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (i != j) o += std::norm(M[i * n + j]); return sqrt(o); };
    printf("off-norm start %.3e\n", offnorm(A));
    hipLaunchKernelGGL(k_dbg<-1>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 1 sweep  %.3e\n", offnorm(B));
    hipLaunchKernelGGL(k_dbg<-2>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 2 sweeps %.3e\n", offnorm(B));
    hipLaunchKernelGGL(k_dbg<-4>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 4 sweeps %.3e\n", offnorm(B));
    hipLaunchKernelGGL(k_dbg<-6>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 6 sweeps %.3e\n", offnorm(B));
    hipLaunchKernelGGL(k_dbg<-8>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 8 sweeps %.3e\n", offnorm(B));
    hipLaunchKernelGGL(k_dbg<-12>, dim3(1), dim3(64), 0, 0, din, dout, dv); hipMemcpy(B.data(), dout, n * n * 16, hipMemcpyDeviceToHost); printf("after 12 sweeps %.3e\n", offnorm(B));
    { double herm = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) herm = fmax(herm, std::abs(B[i * n + j] - std::conj(B[j * n + i]))); printf("hermiticity defect %.3e\n", herm); }
    { // host emulation, 4 sweeps
      std::vector<std::complex<double>> M = A;
      for (int sweep = 0; sweep < 6; ++sweep) { for (int R2 = 0; R2 < 15; ++R2) {
        int P2[8], Q2[8]; P2[0] = 15; Q2[0] = R2; for (int l = 1; l < 8; ++l) { P2[l] = (R2 + l) % 15; Q2[l] = (R2 - l + 15) % 15; }
        double c2[8]; std::complex<double> s2[8];
        for (int l = 0; l < 8; ++l) { auto g = M[P2[l] * n + Q2[l]]; double g2 = std::norm(g); c2[l] = 1; s2[l] = 0; if (g2 > 0) { double h = 0.5 * (M[Q2[l] * n + Q2[l]].real() - M[P2[l] * n + P2[l]].real()), ah = fabs(h), r = sqrt(h * h + g2), inv = 1.0 / sqrt(2 * r * (r + ah)), sg = copysign(1.0, h); c2[l] = (ah + r) * inv; s2[l] = sg * g * inv; } }
        auto C2 = M;
        for (int l = 0; l < 8; ++l) for (int r = 0; r < n; ++r) { auto x = C2[r * n + P2[l]], y = C2[r * n + Q2[l]]; C2[r * n + P2[l]] = c2[l] * x - std::conj(s2[l]) * y; C2[r * n + Q2[l]] = s2[l] * x + c2[l] * y; }
        auto D2 = C2;
        for (int l = 0; l < 8; ++l) for (int c = 0; c < n; ++c) { D2[P2[l] * n + c] = c2[l] * C2[P2[l] * n + c] - s2[l] * C2[Q2[l] * n + c]; D2[Q2[l] * n + c] = std::conj(s2[l]) * C2[P2[l] * n + c] + c2[l] * C2[Q2[l] * n + c]; }
        M = D2; }
        printf("host sweep %d off %.3e\n", sweep + 1, offnorm(M)); } }
    return 0;
}
