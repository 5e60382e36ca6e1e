// e16_bench.hip -- the fused n = 9..16 eigen-solver (k_e16, tbk_solve_e16.inl) beside round 3's three kernels
// (k_tw16_tridiag | k_tw16_eigvals | k_tw16_vectors) on the same supplied matrices (MODE 2): time per batch, how many matrices
// each leaves to the QL-replay fallback, and the quality of what it writes (eigenvalues against the three-kernel path and a
// Jacobi reference on the host for a sample; residual |H v - lambda v|, orthonormality |V^+ V - I|).
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -disable-machine-licm -I pythtb_amd/csrc -I include \
//         profiles/microbench/e16_bench.hip -o profiles/microbench/e16_bench
//   ./e16_bench [nk = 137312] [n = 16] [kind = 0 random | 1 clustered (two groups of 8) | 2 special structures | 3 H_0 x 1_2 (every level twice)] [reps = 5]
#include <hip/hip_runtime.h>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "tbk_solve_dev.h"
#define TBK_TW16_KERNELS_ONLY
#include "tbk_solve_tw16.inl"
#include "tbk_solve_e16.inl"

void tbk_set_error(const char*, ...) {}

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ double u01(unsigned long long s) {
    s ^= s >> 33;
    s *= 0xff51afd7ed558ccdULL;
    s ^= s >> 33;
    s *= 0xc4ceb9fe1a85ec53ULL;
    s ^= s >> 33;
    return (double)(s >> 11) * (1.0 / 9007199254740992.0);
}
// kind 0: random Hermitian, entries uniform in [-1, 1].  kind 1: like cubic16 (onsite -2 / +2 +- 0.2, couplings ~0.1 x 4)
__global__ void k_make(cd* h, const int64_t nk, const int n, const int kind) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nk) return;
    cd* m = h + id * n * n;
    for (int r = 0; r < n; ++r)
        for (int c = r; c < n; ++c) {
            const unsigned long long s = ((unsigned long long)id * 256 + r * 16 + c) * 2654435761ULL + 12345;
            double re = 2.0 * u01(s) - 1.0, im = 2.0 * u01(s ^ 0x9e3779b97f4a7c15ULL) - 1.0;
            if (kind == 1) {
                re *= 0.35;
                im *= 0.35;
                if (r == c) re = (r < n / 2 ? -2.0 : 2.0) + 0.4 * (2.0 * u01(s + 7) - 1.0);
            }
            if (r == c) im = 0.0;
            m[r * n + c] = cd{re, im};
            m[c * n + r] = cd{re, -im};
        }
    if (kind == 3) {   // H_0 x 1_2: every level twice (a spinful model without spin-orbit coupling), components (orbital, spin)
        const int h2 = n / 2;
        cd t[8][8];
        for (int r = 0; r < h2; ++r)
            for (int c = 0; c < h2; ++c) t[r][c] = m[r * n + c];
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) m[r * n + c] = (r & 1) == (c & 1) && r / 2 < h2 && c / 2 < h2 ? t[r / 2][c / 2] : cd{0.0, 0.0};
    }
}

typedef std::complex<double> cplx;
// cyclic Jacobi on the host (reference eigenvalues of a sample)
static void jacobi_eigvals(std::vector<cplx> a, int n, std::vector<double>& ev) {
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) off += std::norm(a[p * n + q]);
        double dia = 0;
        for (int p = 0; p < n; ++p) dia += std::norm(a[p * n + p]);
        if (off <= 1e-34 * dia || off == 0) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const cplx apq = a[p * n + q];
                const double g = std::abs(apq);
                if (g == 0) continue;
                const cplx ph = apq / g;
                const double app = a[p * n + p].real(), aqq = a[q * n + q].real();
                const double tau = (aqq - app) / (2 * g);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                const double c = 1 / std::sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < n; ++k) {   // columns
                    const cplx akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * std::conj(ph) * akq;
                    a[k * n + q] = s * ph * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {   // rows
                    const cplx apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * ph * aqk;
                    a[q * n + k] = s * std::conj(ph) * apk + c * aqk;
                }
            }
    }
    ev.resize(n);
    for (int i = 0; i < n; ++i) ev[i] = a[i * n + i].real();
    std::sort(ev.begin(), ev.end());
}

struct Quality {
    double eerr = 0, res = 0, orth = 0;
    int unsorted = 0;
};
static Quality check(const std::vector<cd>& H, const std::vector<double>& ev, const std::vector<cd>& vec, int64_t nk, int n,
                     const std::vector<int64_t>& sample, const std::vector<char>* skip) {
    Quality q;
    for (int64_t id : sample) {
        if (skip && (*skip)[id]) continue;
        std::vector<cplx> a(n * n);
        for (int i = 0; i < n * n; ++i) a[i] = cplx(H[id * n * n + i].x, H[id * n * n + i].y);
        std::vector<double> ref;
        jacobi_eigvals(a, n, ref);
        double nrm = 1e-300;
        for (double r : ref) nrm = std::max(nrm, std::fabs(r));
        for (int b = 0; b < n; ++b) {
            q.eerr = std::max(q.eerr, std::fabs(ev[(int64_t)b * nk + id] - ref[b]) / nrm);
            if (b > 0 && ev[(int64_t)b * nk + id] < ev[(int64_t)(b - 1) * nk + id]) q.unsorted++;
        }
        for (int b = 0; b < n; ++b) {
            const cd* v = &vec[((int64_t)b * nk + id) * n];
            for (int r = 0; r < n; ++r) {
                cplx acc = 0;
                for (int c = 0; c < n; ++c) acc += a[r * n + c] * cplx(v[c].x, v[c].y);
                acc -= ev[(int64_t)b * nk + id] * cplx(v[r].x, v[r].y);
                q.res = std::max(q.res, std::abs(acc) / nrm);
            }
            for (int b2 = 0; b2 < n; ++b2) {
                const cd* w = &vec[((int64_t)b2 * nk + id) * n];
                cplx acc = 0;
                for (int c = 0; c < n; ++c) acc += std::conj(cplx(v[c].x, v[c].y)) * cplx(w[c].x, w[c].y);
                q.orth = std::max(q.orth, std::abs(acc - (b == b2 ? 1.0 : 0.0)));
            }
        }
    }
    return q;
}

// ---- MODE 1 (mesh): a synthetic cubic16-like model (R-grouped table of 7 lattice vectors x 136 slots, random orbital positions)
// on a side^3 mesh, per-axis phase tables computed here: times k_e16<1> (and k_tw16_*<1>), no quality check (the GPU tests do that)
static void mesh_leg(int side, int reps) {
    const int n = 16, nslot = 136, nR = 7;
    const int64_t npts = (int64_t)side * side * side;
    std::vector<int4> rvec = {{0, 0, 0, 0}, {1, 0, 0, 0}, {-1, 0, 0, 0}, {0, 1, 0, 0}, {0, -1, 0, 0}, {0, 0, 1, 0}, {0, 0, -1, 0}};
    std::vector<cd> rblock(nR * nslot);
    std::vector<double4> orb(n);
    srand(1);
    auto rnd = []() { return 2.0 * rand() / RAND_MAX - 1.0; };
    for (int r = 0; r < nR; ++r)
        for (int sidx = 0, a = 0; a < n; ++a)
            for (int b = a; b < n; ++b, ++sidx) {
                cd v{0.1 * rnd(), 0.1 * rnd()};
                if (r == 0 && a == b) v = cd{(a < 8 ? -2.0 : 2.0) + 0.2 * rnd(), 0.0};
                rblock[r * nslot + sidx] = v;
            }
    // (hermiticity of S(k) = sum_R U_R e^{ikR} on the diagonal slots needs U_{-R} = conj U_R there)
    for (int r = 1; r < nR; r += 2)
        for (int sidx = 0, a = 0; a < n; ++a)
            for (int b = a; b < n; ++b, ++sidx)
                if (a == b) rblock[(r + 1) * nslot + sidx] = cd{rblock[r * nslot + sidx].x, -rblock[r * nslot + sidx].y};
    for (int o = 0; o < n; ++o) orb[o] = double4{0.5 * (rnd() + 1), 0.5 * (rnd() + 1), 0.5 * (rnd() + 1), 0.0};
    std::vector<cd> tz(3 * side), tf(3 * side * n), pbc(4 * n, cd{1, 0});
    for (int d = 0; d < 3; ++d)
        for (int i = 0; i < side; ++i) {
            const int g = i == side - 1 ? 0 : i;
            const double kd = (double)g / (side - 1);
            tz[d * side + i] = cd{cos(2 * M_PI * kd), sin(2 * M_PI * kd)};
            for (int o = 0; o < n; ++o) {
                const double td = d == 0 ? orb[o].x : d == 1 ? orb[o].y : orb[o].z;
                tf[(d * side + i) * n + o] = cd{cos(2 * M_PI * kd * td), -sin(2 * M_PI * kd * td)};
            }
        }
    int4* d_rvec;
    cd *d_rblock, *d_tz, *d_tf, *d_pbc, *d_wf, *refl;
    double4* d_orb;
    unsigned long long* d_gaps;
    int *list, *cnt, *flags;
    double2* de;
    double* lam;
    uint4* meta;
    CK(hipMalloc(&d_rvec, sizeof(int4) * nR));
    CK(hipMalloc(&d_rblock, sizeof(cd) * rblock.size()));
    CK(hipMalloc(&d_orb, sizeof(double4) * n));
    CK(hipMalloc(&d_tz, sizeof(cd) * tz.size()));
    CK(hipMalloc(&d_tf, sizeof(cd) * tf.size()));
    CK(hipMalloc(&d_pbc, sizeof(cd) * pbc.size()));
    CK(hipMalloc(&d_wf, sizeof(cd) * npts * n * n));
    CK(hipMalloc(&d_gaps, 8 * 2 * TBK_GAP_SHARDS * n));
    CK(hipMalloc(&list, npts * sizeof(int)));
    CK(hipMalloc(&cnt, 256));
    CK(hipMalloc(&flags, 256));
    CK(hipMalloc(&de, npts * 16 * sizeof(double2)));
    CK(hipMalloc(&refl, npts * TW16_REC * sizeof(cd)));
    CK(hipMalloc(&lam, npts * 16 * sizeof(double)));
    CK(hipMalloc(&meta, npts * sizeof(uint4)));
    CK(hipMemset(d_gaps, 0x7f, 8 * 2 * TBK_GAP_SHARDS * n));
    CK(hipMemset(flags, 0, 256));
    CK(hipMemcpy(d_rvec, rvec.data(), sizeof(int4) * nR, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rblock, rblock.data(), sizeof(cd) * rblock.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_orb, orb.data(), sizeof(double4) * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tz, tz.data(), sizeof(cd) * tz.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tf, tf.data(), sizeof(cd) * tf.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pbc, pbc.data(), sizeof(cd) * pbc.size(), hipMemcpyHostToDevice));
    ModelView mv{};
    mv.dim_k = 3;
    mv.nsta = n;
    mv.nspin = 1;
    mv.nslot = nslot;
    mv.orb = d_orb;
    mv.nR = nR;
    mv.pmax = 1;            // (largest |R_last|: the row-cell assembly walks p = -pmax .. pmax)
    mv.rvec = d_rvec;
    mv.rblock = d_rblock;
    GridArgs G{};
    G.wv.dim_arr = 3;
    G.wv.nsta = n;
    G.wv.ncomp = n;
    G.wv.npts = npts;
    G.wv.data = d_wf;
    for (int d = 0; d < 3; ++d) {
        G.wv.mesh[d] = side;
        G.gmesh[d] = side;
        G.off[d] = 0;
        G.start_k[d] = 0.0;
        G.tz[d] = d_tz + d * side;
        G.tf[d] = d_tf + (size_t)d * side * n;
    }
    G.wv.mesh[3] = 1;
    G.wv.stride[0] = (int64_t)side * side;
    G.wv.stride[1] = side;
    G.wv.stride[2] = 1;
    G.pbc = d_pbc;
    G.gaps = d_gaps;
    G.gaps_next = d_gaps + TBK_GAP_SHARDS * n;
    G.last = 2;
    G.flags = flags;
    ListArgs L{};
    L.flags = flags;
    const int64_t nc = npts;
    const unsigned b16 = (unsigned)((nc * 16 + 255) / 256), b1 = (unsigned)((nc + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& fn, const char* name) {
        fn();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-34s %9.1f us   (%.3g matrices/s, %.1f us per 137312)\n", name, best * 1e3, nc / (best * 1e-3), best * 1e3 * 137312.0 / nc);
        return best;
    };
    printf("mesh %d^3 = %lld points, synthetic 16-orbital model, 7 lattice vectors\n", side, (long long)npts);
    time_it([&]() {
        CK(hipMemsetAsync(cnt, 0, 4, 0));
        hipLaunchKernelGGL((k_tw16_tridiag<1>), dim3(b16), dim3(256), 0, 0, mv, nc, L, G, de, refl, (int64_t)0, nc);
        hipLaunchKernelGGL((k_tw16_eigvals<1>), dim3(b1), dim3(256), 0, 0, n, nc, (int64_t)0, nc, (const double2*)de, (double*)nullptr, G, lam, meta, list, cnt, flags, 1e-5);
        hipLaunchKernelGGL((k_tw16_vectors<1>), dim3(b16), dim3(256), 0, 0, n, nc, (int64_t)0, nc, mv, L, G, (const double2*)de, (const double*)lam,
                           (const uint4*)meta, (const cd*)refl, list, cnt);
    }, "three kernels <1> back to back");
    time_it([&]() {
        CK(hipMemsetAsync(cnt, 0, 4, 0));
        hipLaunchKernelGGL((k_e16<1>), dim3(b16), dim3(256), 0, 0, mv, nc, L, G, (int64_t)0, nc, list, cnt, 1e-5, getenv("E16_NS_FULL") ? 1 : 0);
    }, "k_e16<1>");
    int c0;
    CK(hipMemcpy(&c0, cnt, 4, hipMemcpyDeviceToHost));
    printf("listed by k_e16<1>: %d\n", c0);
#ifdef E16_ITER_HIST
    {
        unsigned h[32];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(e16_iter_hist), sizeof h));
        double tl = 0, tw = 0, nl = 0, nw = 0;
        for (int i = 0; i < 16; ++i) { tl += (double)i * h[i]; nl += h[i]; tw += (double)i * h[16 + i]; nw += h[16 + i]; }
        printf("Newton steps (all launches of this run): per lane mean %.2f, per wavefront (the slowest lane) mean %.2f\n  lanes     :", tl / nl, tw / nw);
        for (int i = 0; i < 16; ++i) printf(" %.3f", h[i] / nl);
        printf("\n  wavefronts:");
        for (int i = 0; i < 16; ++i) printf(" %.3f", h[16 + i] / nw);
        printf("\n");
    }
#endif
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "mesh") {
        mesh_leg(argc > 2 ? atoi(argv[2]) : 65, argc > 3 ? atoi(argv[3]) : 5);
        return 0;
    }
    const int64_t nk = argc > 1 ? atoll(argv[1]) : 137312;
    const int n = argc > 2 ? atoi(argv[2]) : 16;
    const int kind = argc > 3 ? atoi(argv[3]) : 0;
    const int reps = argc > 4 ? atoi(argv[4]) : 5;
    const double gaptol = 1e-5;
    cd *h, *vec_a, *vec_b, *refl;
    double *ev_a, *ev_b, *lam;
    double2* de;
    uint4* meta;
    int *list_a, *list_b, *cnt, *flags;
    CK(hipMalloc(&h, nk * n * n * sizeof(cd)));
    CK(hipMalloc(&vec_a, nk * n * n * sizeof(cd)));
    CK(hipMalloc(&vec_b, nk * n * n * sizeof(cd)));
    CK(hipMalloc(&ev_a, nk * n * sizeof(double)));
    CK(hipMalloc(&ev_b, nk * n * sizeof(double)));
    CK(hipMalloc(&de, nk * 16 * sizeof(double2)));
    CK(hipMalloc(&refl, nk * TW16_REC * sizeof(cd)));
    CK(hipMalloc(&lam, nk * 16 * sizeof(double)));
    CK(hipMalloc(&meta, nk * sizeof(uint4)));
    CK(hipMalloc(&list_a, nk * sizeof(int)));
    CK(hipMalloc(&list_b, nk * sizeof(int)));
    CK(hipMalloc(&cnt, 256));
    CK(hipMalloc(&flags, 256));
    CK(hipMemset(flags, 0, 256));
    CK(hipMemset(vec_a, 0, nk * n * n * sizeof(cd)));
    CK(hipMemset(vec_b, 0, nk * n * n * sizeof(cd)));
    hipLaunchKernelGGL(k_make, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, 0, h, nk, n, kind == 2 ? 0 : kind);
    CK(hipDeviceSynchronize());
    if (kind == 2) {   // the structures of tests/test_tw16_path.py::special_matrices at the head of the batch
        std::vector<cd> sp(12 * n * n, cd{0, 0});
        auto M = [&](int m, int r, int c) -> cd& { return sp[(m * n + r) * n + c]; };
        for (int i = 0; i < n; ++i) M(1, i, i) = cd{(double)i, 0};
        for (int i = 0; i < n; ++i) M(2, i, i) = cd{i < n / 2 ? 1.0 : 2.0, 0};
        std::vector<cd> a(n * n);
        CK(hipMemcpy(a.data(), h + 100 * n * n, n * n * sizeof(cd), hipMemcpyDeviceToHost));
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                const bool cross = (r < n / 2) != (c < n / 2);
                M(3, r, c) = cross ? cd{0, 0} : a[r * n + c];
                M(4, r, c) = cross ? cd{a[r * n + c].x * 1e-9, a[r * n + c].y * 1e-9} : a[r * n + c];
                M(9, r, c) = cd{a[r * n + c].x * 1e-150, a[r * n + c].y * 1e-150};
                M(10, r, c) = cd{a[r * n + c].x * 1e120, a[r * n + c].y * 1e120};
            }
        for (int i = 0; i + 1 < n; ++i) {
            M(7, i, i + 1) = M(7, i + 1, i) = cd{1, 0};
            M(8, i, i + 1) = M(8, i + 1, i) = cd{1, 0};
        }
        for (int i = 0; i < n; ++i) M(8, i, i) = cd{std::fabs(i - (n - 1) / 2.0), 0};
        // two identical blocks (exact pairs: every level twice, T splits or nearly so)
        for (int r = 0; r < n / 2; ++r)
            for (int c = 0; c < n / 2; ++c) {
                M(5, r, c) = a[r * n + c];
                M(5, r + n / 2, c + n / 2) = a[r * n + c];
            }
        // identity-like and a rank-one matrix
        for (int i = 0; i < n; ++i) M(6, i, i) = cd{3.0, 0};
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) M(11, r, c) = cd{1.0, 0};
        CK(hipMemcpy(h, sp.data(), sp.size() * sizeof(cd), hipMemcpyHostToDevice));
    }
    ModelView mv{};
    mv.nsta = n;
    mv.nspin = 1;
    mv.nslot = n * (n + 1) / 2;
    GridArgs G{};
    ListArgs La{nullptr, h, ev_a, vec_a, flags, 0}, Lb{nullptr, h, ev_b, vec_b, flags, 0};
    const unsigned b16 = (unsigned)((nk * 16 + 255) / 256), b1 = (unsigned)((nk + 255) / 256);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto&& fn, const char* name) {
        fn();
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < reps; ++r) {
            CK(hipEventRecord(e0, 0));
            fn();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        printf("%-34s %9.1f us   (%.3g matrices/s)\n", name, best * 1e3, nk / (best * 1e-3));
        return best;
    };
    auto tw_tri = [&]() { hipLaunchKernelGGL((k_tw16_tridiag<2>), dim3(b16), dim3(256), 0, 0, mv, nk, La, G, de, refl, (int64_t)0, nk); };
    auto tw_eig = [&]() {
        CK(hipMemsetAsync(cnt, 0, 4, 0));
        hipLaunchKernelGGL((k_tw16_eigvals<2>), dim3(b1), dim3(256), 0, 0, n, nk, (int64_t)0, nk, (const double2*)de, ev_a, G, lam, meta, list_a, cnt, flags, gaptol);
    };
    auto tw_vec = [&]() {
        hipLaunchKernelGGL((k_tw16_vectors<2>), dim3(b16), dim3(256), 0, 0, n, nk, (int64_t)0, nk, mv, La, G, (const double2*)de, (const double*)lam,
                           (const uint4*)meta, (const cd*)refl, list_a, cnt);
    };
    auto e16 = [&]() {
        CK(hipMemsetAsync(cnt + 16, 0, 4, 0));
        hipLaunchKernelGGL((k_e16<2>), dim3(b16), dim3(256), 0, 0, mv, nk, Lb, G, (int64_t)0, nk, list_b, cnt + 16, gaptol, getenv("E16_NS_FULL") ? 1 : 0);
    };
    printf("nk = %lld, n = %d, kind = %d\n", (long long)nk, n, kind);
    const float t1 = time_it(tw_tri, "k_tw16_tridiag<2>");
    const float t2 = time_it(tw_eig, "k_tw16_eigvals<2>");
    const float t3 = time_it(tw_vec, "k_tw16_vectors<2>");
    printf("%-34s %9.1f us\n", "three kernels, sum", (t1 + t2 + t3) * 1e3);
    auto all3 = [&]() {
        tw_tri();
        tw_eig();
        tw_vec();
    };
    time_it(all3, "three kernels back to back");
    time_it(e16, "k_e16<2>");
    // one clean pass of each for the lists and the results
    all3();
    e16();
    CK(hipDeviceSynchronize());
    int counts[32];
    CK(hipMemcpy(counts, cnt, sizeof(counts), hipMemcpyDeviceToHost));
    printf("listed for the fallback: three kernels %d, k_e16 %d of %lld\n", counts[0], counts[16], (long long)nk);
    int fl[4];
    CK(hipMemcpy(fl, flags, sizeof(fl), hipMemcpyDeviceToHost));
    printf("flags: %d %d %d %d\n", fl[0], fl[1], fl[2], fl[3]);
    std::vector<char> skip_a(nk, 0), skip_b(nk, 0);
    {
        std::vector<int> la(std::max(counts[0], 1)), lb(std::max(counts[16], 1));
        CK(hipMemcpy(la.data(), list_a, counts[0] * sizeof(int), hipMemcpyDeviceToHost));
        CK(hipMemcpy(lb.data(), list_b, counts[16] * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < counts[0]; ++i) skip_a[la[i]] = 1;
        for (int i = 0; i < counts[16]; ++i) skip_b[lb[i]] = 1;
        if (kind == 2) {
            printf("special matrices listed by k_e16:");
            for (int i = 0; i < 12; ++i) printf(" %d", (int)skip_b[i]);
            printf("\nspecial matrices listed by tw16 :");
            for (int i = 0; i < 12; ++i) printf(" %d", (int)skip_a[i]);
            printf("\n");
        }
    }
    std::vector<cd> H(nk * n * n), va(nk * n * n), vb(nk * n * n);
    std::vector<double> ea(nk * n), eb(nk * n);
    CK(hipMemcpy(H.data(), h, H.size() * sizeof(cd), hipMemcpyDeviceToHost));
    CK(hipMemcpy(va.data(), vec_a, va.size() * sizeof(cd), hipMemcpyDeviceToHost));
    CK(hipMemcpy(vb.data(), vec_b, vb.size() * sizeof(cd), hipMemcpyDeviceToHost));
    CK(hipMemcpy(ea.data(), ev_a, ea.size() * sizeof(double), hipMemcpyDeviceToHost));
    CK(hipMemcpy(eb.data(), ev_b, eb.size() * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<int64_t> sample;
    for (int64_t i = 0; i < std::min<int64_t>(nk, 24); ++i) sample.push_back(i);
    for (int64_t i = 24; i < nk; i += std::max<int64_t>(1, nk / 400)) sample.push_back(i);
    const Quality qa = check(H, ea, va, nk, n, sample, &skip_a), qb = check(H, eb, vb, nk, n, sample, &skip_b);
    printf("quality on %zu sampled matrices (listed ones skipped)   eigenvalue   residual   orthonormality   unsorted\n", sample.size());
    printf("  three kernels                                         %9.2e  %9.2e  %9.2e   %d\n", qa.eerr, qa.res, qa.orth, qa.unsorted);
    printf("  k_e16                                                 %9.2e  %9.2e  %9.2e   %d\n", qb.eerr, qb.res, qb.orth, qb.unsorted);
    // the whole batch: eigenvalues of the two paths against each other (unlisted by both)
    double dmax = 0;
    int64_t worst = -1;
    for (int64_t id = 0; id < nk; ++id) {
        if (skip_a[id] || skip_b[id]) continue;
        double nrm = 1e-300;
        for (int b = 0; b < n; ++b) nrm = std::max(nrm, std::fabs(ea[(int64_t)b * nk + id]));
        for (int b = 0; b < n; ++b) {
            const double dd = std::fabs(ea[(int64_t)b * nk + id] - eb[(int64_t)b * nk + id]) / nrm;
            if (!(dd <= dmax)) {
                dmax = dd;
                worst = id;
            }
        }
    }
    printf("max |lambda(e16) - lambda(tw16)| / |T| over the whole batch: %.3e (matrix %lld)\n", dmax, (long long)worst);
#ifdef E16_DEBUG
    {
        double dbg[64 * 16];
        CK(hipMemcpyFromSymbol(dbg, HIP_SYMBOL(e16_dbg), sizeof(dbg)));
        printf("debug of matrix %d:  j | lo hi clo chi kb blbh | x it conv inside p dp chg | dlam bad scale | tw16 eigenvalue * scale\n", (int)E16_DEBUG);
        for (int j = 0; j < n; ++j) {
            printf("%2d |", j);
            for (int k = 0; k < 16; ++k) printf(" %.10g", dbg[j * 16 + k]);
            printf(" | %.17g\n", ea[(int64_t)j * nk + E16_DEBUG] * dbg[j * 16 + 15]);
        }
    }
#endif
    if (kind == 2) {
        for (int i = 0; i < 12; ++i) {
            std::vector<int64_t> one{i};
            const Quality q = check(H, eb, vb, nk, n, one, nullptr);
            printf("  special %2d (listed %d): eigenvalue %9.2e residual %9.2e orth %9.2e\n", i, (int)skip_b[i], q.eerr, q.res, q.orth);
        }
    }
    return 0;
}
