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
#include <string.h>
#include <algorithm>
#include "tbk_internal.h"

#define TBK_JACOBI_MAX_SWEEPS 30

struct GridArgs {
    WfsView wv;
    double start_k[TBK_MAX_DIM];
    int gmesh[TBK_MAX_DIM];  // global mesh sizes (axis 0 may exceed the slab)
    int64_t row0;            // first global row of the slab (axis 0)
    const cd* pbc;           // [TBK_MAX_DIM][TBK_MAX_NSTA]
    unsigned long long* gaps;
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
            int64_t g = ii[d] + (d == 0 ? G.row0 : 0);
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

// ---------------------------------------------------------------------------
// nsta <= 4: one thread per matrix, everything in registers.
// ---------------------------------------------------------------------------
template <int N>
struct SmallMat {
    double dg[N];       // diagonal (real)
    cd up[N][N];        // strict upper triangle (p < q) used
    cd v[N][N];         // v[o][b]: component o of eigenvector b
};

// One complex Jacobi rotation on (P,Q), P < Q, zeroing up[P][Q].
//   J_PP = c, J_QP = -s conj(w), J_PQ = s w, J_QQ = c, w = a_PQ/|a_PQ|
template <int N, int P, int Q, bool VEC>
__device__ __forceinline__ void rotate(SmallMat<N>& M) {
    const cd g = M.up[P][Q];
    const double g2 = cabs2(g);
    if (g2 > 0.0) {
        const double ga = sqrt(g2);
        const double inv = 1.0 / ga;
        const cd w{g.x * inv, g.y * inv};
        const double tau = (M.dg[Q] - M.dg[P]) * (0.5 * inv);
        const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t);
        const double s = t * c;
        const cd sw{s * w.x, s * w.y};
        M.dg[P] -= t * ga;
        M.dg[Q] += t * ga;
        M.up[P][Q] = cd{0.0, 0.0};
#pragma unroll
        for (int r = 0; r < N; ++r) {
            if (r == P || r == Q) continue;
            // x = a_rP, y = a_rQ read through the hermitian upper storage
            cd x = r < P ? M.up[r][P] : cconj(M.up[P][r]);
            cd y = r < Q ? M.up[r][Q] : cconj(M.up[Q][r]);
            // a'_rP = c x - conj(sw) y ;  a'_rQ = sw x + c y
            cd xn{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
            cd yn{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
            if (r < P) M.up[r][P] = xn; else M.up[P][r] = cconj(xn);
            if (r < Q) M.up[r][Q] = yn; else M.up[Q][r] = cconj(yn);
        }
        if (VEC) {
#pragma unroll
            for (int r = 0; r < N; ++r) {
                const cd x = M.v[r][P], y = M.v[r][Q];
                M.v[r][P] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                M.v[r][Q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
            }
        }
    }
}

template <int N, int P, int Q, bool VEC>
struct Sweep {
    __device__ static __forceinline__ void run(SmallMat<N>& M) {
        rotate<N, P, Q, VEC>(M);
        if constexpr (Q + 1 < N)
            Sweep<N, P, Q + 1, VEC>::run(M);
        else if constexpr (P + 2 < N)
            Sweep<N, P + 1, P + 2, VEC>::run(M);
    }
};

template <int N, bool VEC>
__device__ __forceinline__ void jacobi_small(SmallMat<N>& M) {
    if constexpr (N == 1) {
        return;
    } else if constexpr (N == 2) {
        rotate<2, 0, 1, VEC>(M);  // exact for 2x2
    } else {
        for (int sweep = 0; sweep < TBK_JACOBI_MAX_SWEEPS; ++sweep) {
            double off = 0.0, dia = 0.0;
#pragma unroll
            for (int p = 0; p < N; ++p) {
                dia += M.dg[p] * M.dg[p];
#pragma unroll
                for (int q = p + 1; q < N; ++q) off += cabs2(M.up[p][q]);
            }
            if (off <= 1.0e-32 * (dia + off)) break;
            Sweep<N, 0, 1, VEC>::run(M);
        }
    }
}

// rank of each eigenvalue in ascending order (stable: ties keep index order)
template <int N>
__device__ __forceinline__ void ranks_small(const double (&ev)[N], int (&rk)[N], double (&sorted)[N]) {
#pragma unroll
    for (int b = 0; b < N; ++b) {
        int r = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) r += (ev[j] < ev[b]) || (ev[j] == ev[b] && j < b);
        rk[b] = r;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < N; ++b) s = rk[b] == r ? ev[b] : s;
        sorted[r] = s;
    }
}

// MODE 0: k list -> eval/evec (band-major).  MODE 1: wf mesh -> _wfs + min gaps.
// MODE 2: supplied matrices -> eval/evec.
template <int N, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_solve_small(const ModelView mv, const int64_t nk,
                                                     const ListArgs L, const GridArgs G) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool active = idx < nk;
    const int64_t id = active ? idx : nk - 1;  // idle lanes redo the last point (no stores)

    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
    } else if constexpr (MODE == 1) {
        grid_point(G, id, kk, wrap);
    }

    SmallMat<N> M;
    if constexpr (MODE == 2) {
        const cd* h = L.ham + id * (int64_t)(N * N);
#pragma unroll
        for (int a = 0; a < N; ++a) {
            M.dg[a] = h[a * N + a].x;
#pragma unroll
            for (int b = a + 1; b < N; ++b) M.up[a][b] = h[a * N + b];
        }
    } else {
        cd z[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        int slot = 0;
#pragma unroll
        for (int a = 0; a < N; ++a) {
#pragma unroll
            for (int b = a; b < N; ++b, ++slot) {
                const cd s = slot_sum(mv, slot, z);
                if (b == a) M.dg[a] = s.x; else M.up[a][b] = s;
            }
        }
    }
    if (VEC) {
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) M.v[a][b] = cd{a == b ? 1.0 : 0.0, 0.0};
    }

    jacobi_small<N, VEC>(M);

    int rk[N];
    double sorted[N];
    ranks_small<N>(M.dg, rk, sorted);

    // eigenvector component o of S  ->  conj(e_o) * (pbc phases) * v   (D^+ v)
    cd fo[N];
    if (VEC) {
#pragma unroll
        for (int o = 0; o < N; ++o) {
            if constexpr (MODE == 2) {
                fo[o] = cd{1.0, 0.0};
            } else {
                if (mv.nspin == 2 && (o & 1)) {
                    fo[o] = fo[o - (o > 0)];
                } else {
                    fo[o] = cconj(expi2pi(kdot(kk, mv.orb[o])));
                }
            }
        }
        if constexpr (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (wrap[d]) {
#pragma unroll
                    for (int o = 0; o < N; ++o) fo[o] = cmul(fo[o], G.pbc[d * TBK_MAX_NSTA + o]);
                }
        }
    }

    if constexpr (MODE == 1) {
        // min direct gaps over the mesh (all_gaps.min, pythtb.py:2495,2530)
        if constexpr (N > 1) {
            __shared__ double red[4][N];
#pragma unroll
            for (int b = 0; b + 1 < N; ++b) {
                double g = sorted[b + 1] - sorted[b];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) g = fmin(g, __shfl_xor(g, off));
                if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][b] = g;
            }
            __syncthreads();
            if (threadIdx.x < N - 1) {
                const double g = fmin(fmin(red[0][threadIdx.x], red[1][threadIdx.x]),
                                      fmin(red[2][threadIdx.x], red[3][threadIdx.x]));
                // gaps are >= 0, so their bit patterns order like the values
                atomicMin(G.gaps + threadIdx.x, (unsigned long long)__double_as_longlong(fmax(g, 0.0)));
            }
        }
        if (active) {
            cd* out = G.wv.data + id * (int64_t)(N * N);
#pragma unroll
            for (int b = 0; b < N; ++b)
#pragma unroll
                for (int o = 0; o < N; ++o) out[rk[b] * N + o] = cmul(M.v[o][b], fo[o]);
        }
    } else {
        if (active) {
#pragma unroll
            for (int b = 0; b < N; ++b) L.eval[(int64_t)b * nk + id] = sorted[b];
            if (VEC) {
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    cd* out = L.evec + ((int64_t)rk[b] * nk + id) * N;
#pragma unroll
                    for (int o = 0; o < N; ++o) out[o] = cmul(M.v[o][b], fo[o]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// nsta > 4: one wavefront per matrix, A and V^T in LDS.
// Row stride n+1 (one c128 of padding) keeps both row and column walks of the
// 16-byte elements on distinct LDS banks.
// ---------------------------------------------------------------------------
struct WaveLds {
    cd* A;      // [n][ld]
    cd* Vt;     // [n][ld]  Vt[b][o] = component o of eigenvector b
    cd* rot;    // [n/2+1]  (c, -) and sw packed: rot[2i] = {c, 0}, rot[2i+1] = sw
    int* pq;    // [n/2+1]  p | q<<16 (p<q), -1 for the bye
    double* ev; // [n]
    int* perm;  // [n]  perm[rank] = column
    cd* eo;     // [n]  conj(e_o) * pbc phases
};

template <int MODE, bool VEC>
__global__ __launch_bounds__(64) void k_solve_wave(const ModelView mv, const int64_t nk,
                                                   const ListArgs L, const GridArgs G,
                                                   int* noconv_flag) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = mv.nsta;
    const int ld = n + 1;
    const int lane = threadIdx.x;
    const int m = (n + 1) & ~1;  // players in the round-robin (bye if n odd)
    const int half = m >> 1;
    WaveLds S;
    S.A = (cd*)lds_raw;
    S.Vt = S.A + n * ld;
    S.rot = S.Vt + n * ld;
    S.eo = S.rot + 2 * half;
    S.ev = (double*)(S.eo + n);
    S.pq = (int*)(S.ev + n);
    S.perm = S.pq + half;

    for (int64_t id = blockIdx.x; id < nk; id += gridDim.x) {
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
        bool wrap[4] = {false, false, false, false};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
        } else if constexpr (MODE == 1) {
            grid_point(G, id, kk, wrap);
        }
        __syncthreads();  // previous matrix fully written out before LDS is reused
        // ---- assemble S(k) (or load the supplied matrix) into A, V^T = I
        if constexpr (MODE == 2) {
            const cd* h = L.ham + id * (int64_t)n * n;
            for (int e = lane; e < n * n; e += 64) {
                const int a = e / n, b = e - a * n;
                // use the upper triangle, mirror it (the reference's eigh reads one triangle)
                cd v = a <= b ? h[a * n + b] : cconj(h[b * n + a]);
                if (a == b) v.y = 0.0;
                S.A[a * ld + b] = v;
            }
        } else {
            cd z[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
            for (int slot = lane; slot < mv.nslot; slot += 64) {
                const int ab = mv.slot_ab[slot];
                const int a = ab & 0xffff, b = ab >> 16;
                const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
                cd acc{0.0, 0.0};
                for (int t = t0; t < t1; ++t) cfma(acc, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
                if (a == b) {
                    S.A[a * ld + a] = cd{acc.x, 0.0};
                } else {
                    S.A[a * ld + b] = acc;
                    S.A[b * ld + a] = cconj(acc);
                }
            }
        }
        if (VEC) {
            for (int e = lane; e < n * n; e += 64) {
                const int a = e / n, b = e - a * n;
                S.Vt[a * ld + b] = cd{a == b ? 1.0 : 0.0, 0.0};
            }
            if (lane < n) {
                cd f{1.0, 0.0};
                if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[lane])));
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (wrap[d]) f = cmul(f, G.pbc[d * TBK_MAX_NSTA + lane]);
                }
                S.eo[lane] = f;
            }
        }
        __syncthreads();

        // ---- parallel-ordered Jacobi sweeps
        bool converged = false;
        for (int sweep = 0; sweep < TBK_JACOBI_MAX_SWEEPS; ++sweep) {
            double off = 0.0, dia = 0.0;
            for (int e = lane; e < n * n; e += 64) {
                const int a = e / n, b = e - a * n;
                const double v2 = cabs2(S.A[a * ld + b]);
                if (a == b) dia += v2; else off += v2;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                off += __shfl_xor(off, o);
                dia += __shfl_xor(dia, o);
            }
            if (off <= 2.0e-32 * (dia + off)) {
                converged = true;
                break;
            }
            for (int round = 0; round < m - 1; ++round) {
                // pairing of round `round`: player m-1 fixed, the others rotate
                if (lane < half) {
                    int p, q;
                    if (lane == 0) {
                        p = m - 1;
                        q = round;
                    } else {
                        p = (round + lane) % (m - 1);
                        q = (round - lane + (m - 1)) % (m - 1);
                    }
                    if (p > q) {
                        const int t = p;
                        p = q;
                        q = t;
                    }
                    double c = 1.0;
                    cd sw{0.0, 0.0};
                    int code = -1;
                    if (q < n) {  // not the bye
                        const cd g = S.A[p * ld + q];
                        const double g2 = cabs2(g);
                        if (g2 > 0.0) {
                            const double ga = sqrt(g2), inv = 1.0 / ga;
                            const double tau = (S.A[q * ld + q].x - S.A[p * ld + p].x) * (0.5 * inv);
                            const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                            c = 1.0 / sqrt(1.0 + t * t);
                            const double s = t * c;
                            sw = cd{s * g.x * inv, s * g.y * inv};
                            code = p | (q << 16);
                        }
                    }
                    S.pq[lane] = code;
                    S.rot[2 * lane] = cd{c, 0.0};
                    S.rot[2 * lane + 1] = sw;
                }
                __syncthreads();
                // columns: A <- A J   (rows r fastest across lanes)
                for (int e = lane; e < half * n; e += 64) {
                    const int i = e / n, r = e - i * n;
                    const int code = S.pq[i];
                    if (code < 0) continue;
                    const int p = code & 0xffff, q = code >> 16;
                    const double c = S.rot[2 * i].x;
                    const cd sw = S.rot[2 * i + 1];
                    const cd x = S.A[r * ld + p], y = S.A[r * ld + q];
                    S.A[r * ld + p] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                    S.A[r * ld + q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
                }
                __syncthreads();
                // rows: A <- J^+ A, and V^T rows p,q (V <- V J)
                for (int e = lane; e < half * n; e += 64) {
                    const int i = e / n, cidx = e - i * n;
                    const int code = S.pq[i];
                    if (code < 0) continue;
                    const int p = code & 0xffff, q = code >> 16;
                    const double c = S.rot[2 * i].x;
                    const cd sw = S.rot[2 * i + 1];
                    {   // a'_pc = c a_pc - sw a_qc ; a'_qc = conj(sw) a_pc + c a_qc
                        const cd x = S.A[p * ld + cidx], y = S.A[q * ld + cidx];
                        S.A[p * ld + cidx] = cd{c * x.x - (sw.x * y.x - sw.y * y.y), c * x.y - (sw.x * y.y + sw.y * y.x)};
                        S.A[q * ld + cidx] = cd{(sw.x * x.x + sw.y * x.y) + c * y.x, (sw.x * x.y - sw.y * x.x) + c * y.y};
                    }
                    if (VEC) {  // v'_rp = c v_rp - conj(sw) v_rq ; v'_rq = sw v_rp + c v_rq
                        const cd x = S.Vt[p * ld + cidx], y = S.Vt[q * ld + cidx];
                        S.Vt[p * ld + cidx] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                        S.Vt[q * ld + cidx] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
                    }
                }
                __syncthreads();
                // pin the rotated pair exactly: zero off-diagonal, real diagonal
                if (lane < half) {
                    const int code = S.pq[lane];
                    if (code >= 0) {
                        const int p = code & 0xffff, q = code >> 16;
                        S.A[p * ld + q] = cd{0.0, 0.0};
                        S.A[q * ld + p] = cd{0.0, 0.0};
                        S.A[p * ld + p].y = 0.0;
                        S.A[q * ld + q].y = 0.0;
                    }
                }
                __syncthreads();
            }
        }
        if (!converged && lane == 0) atomicExch(noconv_flag, 1);

        // ---- order eigenvalues (stable ascending), write out
        if (lane < n) S.ev[lane] = S.A[lane * ld + lane].x;
        __syncthreads();
        if (lane < n) {
            const double mine = S.ev[lane];
            int r = 0;
            for (int j = 0; j < n; ++j) {
                const double o = S.ev[j];
                r += (o < mine) || (o == mine && j < lane);
            }
            S.perm[r] = lane;
        }
        __syncthreads();
        if constexpr (MODE == 1) {
            if (lane + 1 < n) {
                const double g = S.ev[S.perm[lane + 1]] - S.ev[S.perm[lane]];
                atomicMin(G.gaps + lane, (unsigned long long)__double_as_longlong(fmax(g, 0.0)));
            }
            cd* out = G.wv.data + id * (int64_t)n * n;
            for (int e = lane; e < n * n; e += 64) {
                const int rb = e / n, o = e - rb * n;
                out[e] = cmul(S.Vt[S.perm[rb] * ld + o], S.eo[o]);
            }
        } else {
            if (lane < n) L.eval[(int64_t)lane * nk + id] = S.ev[S.perm[lane]];
            if (VEC) {
                for (int e = lane; e < n * n; e += 64) {
                    const int rb = e / n, o = e - rb * n;
                    L.evec[((int64_t)rb * nk + id) * n + o] = cmul(S.Vt[S.perm[rb] * ld + o], S.eo[o]);
                }
            }
        }
    }
}

static size_t wave_lds_bytes(int n) {
    const int ld = n + 1, half = (n + 1) / 2;
    size_t b = (size_t)2 * n * ld * sizeof(cd);  // A, Vt
    b += (size_t)2 * half * sizeof(cd);          // rot
    b += (size_t)n * sizeof(cd);                 // eo
    b += (size_t)n * sizeof(double);             // ev
    b += (size_t)(half + n) * sizeof(int);       // pq, perm
    return (b + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------
// _gen_ham parity hook: H_ab(k) = conj(e_a) e_b S_ab(k), one thread per (k,slot)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_ham(const ModelView mv, const int64_t nk,
                                                 const double* __restrict__ k, cd* __restrict__ ham) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = nk * mv.nslot;
    if (idx >= total) return;
    const int64_t ik = idx / mv.nslot;
    const int slot = (int)(idx - ik * mv.nslot);
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    cd z[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d < mv.dim_k) kk[d] = k[ik * mv.dim_k + d];
        z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
    }
    const int ab = mv.slot_ab[slot];
    const int a = ab & 0xffff, b = ab >> 16;
    const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
    cd s{0.0, 0.0};
    for (int t = t0; t < t1; ++t) cfma(s, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
    const int n = mv.nsta;
    cd* h = ham + ik * (int64_t)n * n;
    if (a == b) {
        h[a * n + a] = cd{s.x, 0.0};
    } else {
        const cd ea = expi2pi(kdot(kk, mv.orb[a])), eb = expi2pi(kdot(kk, mv.orb[b]));
        const cd v = cmul(cmulc(ea, eb), s);
        h[a * n + b] = v;
        h[b * n + a] = cconj(v);
    }
}

__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
template <int MODE, bool VEC>
static int launch_small(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L,
                        const GridArgs& G) {
    const unsigned blocks = (unsigned)((nk + 255) / 256);
    switch (n) {
        case 1: hipLaunchKernelGGL((k_solve_small<1, MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
        case 2: hipLaunchKernelGGL((k_solve_small<2, MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
        case 3: hipLaunchKernelGGL((k_solve_small<3, MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
        case 4: hipLaunchKernelGGL((k_solve_small<4, MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
        default: tbk_set_error("launch_small: n=%d", n); return TBK_EINVAL;
    }
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

template <int MODE, bool VEC>
static int launch_wave(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L,
                       const GridArgs& G) {
    const size_t lds = wave_lds_bytes(n);
    TBK_REQUIRE(lds <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d needs %zu bytes of LDS per wavefront", n, lds);
    static bool attr_set[2][3] = {{false, false, false}, {false, false, false}};
    if (lds > 64 * 1024 && !attr_set[VEC][MODE]) {
        TBK_HIP(hipFuncSetAttribute((const void*)k_solve_wave<MODE, VEC>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[VEC][MODE] = true;
    }
    int* flag = ctx->flags_dev;  // sticky until read by check_noconv
    // enough resident wavefronts to fill the chip; each strides over the k list
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds));
    const int64_t want = (int64_t)ctx->cus * per_cu * 2;
    const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nk, want));
    hipLaunchKernelGGL((k_solve_wave<MODE, VEC>), dim3(blocks), dim3(64), lds, ctx->stream, mv, nk, L, G, flag);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

template <int MODE>
static int launch_solve(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, bool vec,
                        const ListArgs& L, const GridArgs& G, const char* name) {
    if (nk <= 0) return TBK_OK;
    ProfScope ps(ctx, name);
    if (n <= 4) return vec ? launch_small<MODE, true>(ctx, mv, n, nk, L, G)
                           : launch_small<MODE, false>(ctx, mv, n, nk, L, G);
    return vec ? launch_wave<MODE, true>(ctx, mv, n, nk, L, G)
               : launch_wave<MODE, false>(ctx, mv, n, nk, L, G);
}

static int check_noconv(tbk_ctx* ctx, int n) {
    if (n <= 4) return TBK_OK;
    int flag = 0;
    TBK_HIP(hipMemcpyAsync(&flag, ctx->flags_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    if (flag) TBK_HIP(hipMemsetAsync(ctx->flags_dev, 0, sizeof(int), ctx->stream));
    TBK_REQUIRE(flag == 0, TBK_ENOCONV, "Jacobi eigen-solver did not converge in %d sweeps", TBK_JACOBI_MAX_SWEEPS);
    return TBK_OK;
}

extern "C" int tbk_solve_list_dev(tbk_model* m, const double* k_dev, int64_t nk, double* eval_dev,
                                  double* evec_dev) {
    TBK_REQUIRE(m && eval_dev && nk >= 0, TBK_EINVAL, "tbk_solve_list_dev: bad argument");
    TBK_REQUIRE(m->dim_k == 0 || k_dev || nk == 0, TBK_EINVAL, "tbk_solve_list_dev: null k");
    ListArgs L{k_dev, nullptr, eval_dev, (cd*)evec_dev};
    GridArgs G{};
    return launch_solve<0>(m->ctx, m->view, m->nsta, nk, evec_dev != nullptr, L, G,
                           evec_dev ? "solve_list_vec" : "solve_list_val");
}

extern "C" int tbk_solve_list(tbk_model* m, const double* k, int64_t nk, double* eval, double* evec) {
    TBK_REQUIRE(m && eval && nk >= 0, TBK_EINVAL, "tbk_solve_list: bad argument");
    if (nk == 0) return TBK_OK;
    tbk_ctx* ctx = m->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const int n = m->nsta;
    const size_t kb = (size_t)nk * std::max(m->dim_k, 1) * sizeof(double);
    const size_t eb = (size_t)nk * n * sizeof(double);
    const size_t vb = evec ? (size_t)nk * n * n * sizeof(cd) : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + al(kb) + al(eb) + al(vb), &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;  // first 256 bytes: flags
    double* k_dev = (double*)p;
    double* e_dev = (double*)(p + al(kb));
    double* v_dev = evec ? (double*)(p + al(kb) + al(eb)) : nullptr;
    if (m->dim_k > 0) {
        TBK_REQUIRE(k, TBK_EINVAL, "tbk_solve_list: null k");
        TBK_HIP(hipMemcpyAsync(k_dev, k, kb, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = tbk_solve_list_dev(m, k_dev, nk, e_dev, v_dev);
    if (rc) return rc;
    TBK_HIP(hipMemcpyAsync(eval, e_dev, eb, hipMemcpyDeviceToHost, ctx->stream));
    if (evec) TBK_HIP(hipMemcpyAsync(evec, v_dev, vb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return check_noconv(ctx, n);
}

extern "C" int tbk_eigh_batch(tbk_ctx* ctx, int n, const double* ham, int64_t nk, double* eval,
                              double* evec) {
    TBK_REQUIRE(ctx && ham && eval && nk >= 0, TBK_EINVAL, "tbk_eigh_batch: bad argument");
    TBK_REQUIRE(n >= 1 && n <= TBK_MAX_NSTA, TBK_EUNSUPPORTED, "tbk_eigh_batch: n=%d (limit %d)", n, TBK_MAX_NSTA);
    if (nk == 0) return TBK_OK;
    TBK_HIP(hipSetDevice(ctx->device));
    const size_t hb = (size_t)nk * n * n * sizeof(cd);
    const size_t eb = (size_t)nk * n * sizeof(double);
    const size_t vb = evec ? hb : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + al(hb) + al(eb) + al(vb), &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;
    cd* h_dev = (cd*)p;
    double* e_dev = (double*)(p + al(hb));
    cd* v_dev = evec ? (cd*)(p + al(hb) + al(eb)) : nullptr;
    TBK_HIP(hipMemcpyAsync(h_dev, ham, hb, hipMemcpyHostToDevice, ctx->stream));
    ModelView mv{};
    mv.nsta = n;
    mv.nspin = 1;
    mv.nslot = n * (n + 1) / 2;
    ListArgs L{nullptr, h_dev, e_dev, v_dev};
    GridArgs G{};
    rc = launch_solve<2>(ctx, mv, n, nk, evec != nullptr, L, G, "eigh_batch");
    if (rc) return rc;
    TBK_HIP(hipMemcpyAsync(eval, e_dev, eb, hipMemcpyDeviceToHost, ctx->stream));
    if (evec) TBK_HIP(hipMemcpyAsync(evec, v_dev, vb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return check_noconv(ctx, n);
}

extern "C" int tbk_gen_ham(tbk_model* m, const double* k, int64_t nk, double* ham_out) {
    TBK_REQUIRE(m && ham_out && nk >= 0, TBK_EINVAL, "tbk_gen_ham: bad argument");
    if (nk == 0) return TBK_OK;
    tbk_ctx* ctx = m->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const int n = m->nsta;
    const size_t kb = (size_t)nk * std::max(m->dim_k, 1) * sizeof(double);
    const size_t hb = (size_t)nk * n * n * sizeof(cd);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + al(kb) + al(hb), &base);
    if (rc) return rc;
    double* k_dev = (double*)((unsigned char*)base + 256);
    cd* h_dev = (cd*)((unsigned char*)base + 256 + al(kb));
    if (m->dim_k > 0) {
        TBK_REQUIRE(k, TBK_EINVAL, "tbk_gen_ham: null k");
        TBK_HIP(hipMemcpyAsync(k_dev, k, kb, hipMemcpyHostToDevice, ctx->stream));
    }
    TBK_HIP(hipMemsetAsync(h_dev, 0, hb, ctx->stream));
    {
        ProfScope ps(ctx, "gen_ham");
        const int64_t total = nk * m->nslot;
        hipLaunchKernelGGL(k_gen_ham, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           m->view, nk, k_dev, h_dev);
        TBK_HIP(hipGetLastError());
    }
    TBK_HIP(hipMemcpyAsync(ham_out, h_dev, hb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

// ---------------------------------------------------------------------------
// solve_on_grid
// ---------------------------------------------------------------------------
extern "C" int tbk_wfs_solve_grid_async(tbk_wfs* w, tbk_model* m, const double* start_k,
                                        const double* pbc_phase, int64_t row0, int64_t global_n0) {
    TBK_REQUIRE(w && m && start_k && pbc_phase, TBK_EINVAL, "tbk_wfs_solve_grid: null argument");
    TBK_REQUIRE(w->ctx == m->ctx, TBK_EINVAL, "tbk_wfs_solve_grid: model and wfs live on different contexts");
    const WfsView& v = w->view;
    // pythtb.py:2448-2459
    TBK_REQUIRE(v.dim_arr == m->dim_k, TBK_EINVAL,
                "tbk_wfs_solve_grid: dimension of wf_array (%d) must equal dim_k (%d)", v.dim_arr, m->dim_k);
    TBK_REQUIRE(v.nsta == m->nsta && v.ncomp == m->nsta, TBK_EINVAL,
                "tbk_wfs_solve_grid: array holds %d states of %d components, model has %d", v.nsta, v.ncomp, m->nsta);
    TBK_REQUIRE(global_n0 >= v.mesh[0] && row0 >= 0 && row0 + v.mesh[0] <= global_n0, TBK_EINVAL,
                "tbk_wfs_solve_grid: slab rows [%lld,%lld) outside global axis of %lld", (long long)row0,
                (long long)(row0 + v.mesh[0]), (long long)global_n0);
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const int n = m->nsta;
    // pbc phases: re-upload only when they changed
    const size_t np = (size_t)v.dim_arr * n * 2;
    bool same = w->pbc_host.size() == np;
    if (same) same = memcmp(w->pbc_host.data(), pbc_phase, np * sizeof(double)) == 0;
    if (!same) {
        w->pbc_host.assign(pbc_phase, pbc_phase + np);
        std::vector<cd> st((size_t)TBK_MAX_DIM * TBK_MAX_NSTA, cd{0.0, 0.0});
        for (int d = 0; d < v.dim_arr; ++d)
            for (int o = 0; o < n; ++o)
                st[d * TBK_MAX_NSTA + o] = cd{pbc_phase[2 * (d * n + o)], pbc_phase[2 * (d * n + o) + 1]};
        TBK_HIP(hipMemcpyAsync(w->pbc_dev, st.data(), st.size() * sizeof(cd), hipMemcpyHostToDevice,
                               ctx->stream));
        TBK_HIP(hipStreamSynchronize(ctx->stream));  // st is a local: finish before it dies
    }
    GridArgs G{};
    G.wv = v;
    for (int d = 0; d < TBK_MAX_DIM; ++d) {
        G.start_k[d] = d < v.dim_arr ? start_k[d] : 0.0;
        G.gmesh[d] = v.mesh[d];
    }
    G.gmesh[0] = (int)global_n0;
    G.row0 = row0;
    G.pbc = w->pbc_dev;
    G.gaps = w->gaps_dev;
    w->gaps_n = n - 1;
    hipLaunchKernelGGL(k_fill_u64, dim3(1), dim3(TBK_MAX_NSTA), 0, ctx->stream, w->gaps_dev, TBK_MAX_NSTA,
                       0x7ff0000000000000ull);
    ListArgs L{};
    return launch_solve<1>(ctx, m->view, n, v.npts, true, L, G, "solve_grid");
}

extern "C" int tbk_wfs_solve_grid_result(tbk_wfs* w, double* min_gaps) {
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_wfs_solve_grid_result: null wfs");
    tbk_ctx* ctx = w->ctx;
    unsigned long long bits[TBK_MAX_NSTA];
    if (w->gaps_n > 0 && min_gaps) {
        TBK_HIP(hipMemcpyAsync(bits, w->gaps_dev, w->gaps_n * sizeof(unsigned long long),
                               hipMemcpyDeviceToHost, ctx->stream));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    if (w->gaps_n > 0 && min_gaps) memcpy(min_gaps, bits, w->gaps_n * sizeof(double));
    return check_noconv(ctx, w->view.nsta);
}

extern "C" int tbk_wfs_solve_grid(tbk_wfs* w, tbk_model* m, const double* start_k,
                                  const double* pbc_phase, int64_t row0, int64_t global_n0,
                                  double* min_gaps) {
    int rc = tbk_wfs_solve_grid_async(w, m, start_k, pbc_phase, row0, global_n0);
    if (rc) return rc;
    return tbk_wfs_solve_grid_result(w, min_gaps);
}

// impose_pbc / impose_loop: last slice along mesh_dir = first slice * phase[comp]
__global__ __launch_bounds__(256) void k_impose(const WfsView v, const int dir, const cd* __restrict__ phase,
                                                const int64_t nface) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int per = v.nsta * v.ncomp;
    if (idx >= nface * per) return;
    const int64_t f = idx / per;
    const int e = (int)(idx - f * per);
    const int o = e % v.ncomp;
    // face index f enumerates all mesh points with index 0 along dir
    const int64_t inner = v.stride[dir];           // points per unit step of dir
    const int64_t outer = f / inner, in = f - outer * inner;
    const int64_t p0 = outer * inner * v.mesh[dir] + in;
    const int64_t p1 = p0 + (int64_t)(v.mesh[dir] - 1) * inner;
    cd val = v.data[p0 * per + e];
    if (phase) val = cmul(val, phase[o]);
    v.data[p1 * per + e] = val;
}

extern "C" int tbk_wfs_impose(tbk_wfs* w, int mesh_dir, const double* phase) {
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_wfs_impose: null wfs");
    const WfsView& v = w->view;
    TBK_REQUIRE(mesh_dir >= 0 && mesh_dir < v.dim_arr, TBK_EINVAL, "tbk_wfs_impose: mesh_dir=%d", mesh_dir);
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    cd* ph_dev = nullptr;
    if (phase) {
        void* base = nullptr;
        int rc = tbk_ctx_scratch(ctx, 256 + v.ncomp * sizeof(cd), &base);
        if (rc) return rc;
        ph_dev = (cd*)((unsigned char*)base + 256);
        TBK_HIP(hipMemcpyAsync(ph_dev, phase, v.ncomp * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
    }
    const int64_t nface = v.npts / v.mesh[mesh_dir];
    const int64_t total = nface * v.nsta * v.ncomp;
    {
        ProfScope ps(ctx, "impose_pbc");
        hipLaunchKernelGGL(k_impose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, v, mesh_dir,
                           (const cd*)ph_dev, nface);
        TBK_HIP(hipGetLastError());
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}
