// tbk_berry.hip -- Berry flux (plaquettes) and Berry phases (strings) on a
// device-resident wf mesh, gfx950.
//
// Reproduces _wf_dpr (pythtb.py:3793-3796), _one_berry_loop (:3798-3838),
// _one_flux_plane (:3840-3865) and the slicing of wf_array.berry_flux
// (:3135-3202) / berry_phase (:2978-3029).
//
// Identities used (checked against the reference):
//   det(M1 M2 ...) = det M1 * det M2 * ...      -> link determinants are scalars
//   -angle(det prod) for a string / plaquette   -> one complex product, one atan2
//   polar(M) = U Vh of svd(M)                   -> closed form (nocc<=2); from 3 bands on the
//                                                  workgroup-level pipeline of tbk_berry_big.inl
//                                                  (Newton-Schulz, product tree, Cayley + eigh)
// Ordered nocc x nocc products are split into per-thread segments and combined
// in order (matrix products are associative, not commutative).  From 9 bands on the link
// determinants come from one workgroup per link (LU, tbk_berry_big.inl).  The per-thread
// one-sided-Jacobi / QR routines for 3..16 bands stay reachable through TBK_WILSON_BIG_FROM /
// TBK_DET_BIG_FROM for A/B runs.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "tbk_internal.h"

struct AxisSet {           // the axes that enumerate slices / strings
    int n;                 // how many (<= 3)
    int size[3];
    int64_t stride[3];     // in mesh points
};

__device__ __forceinline__ int64_t axis_offset(const AxisSet& ax, int64_t s) {
    int64_t off = 0;
#pragma unroll
    for (int a = 2; a >= 0; --a) {
        if (a < ax.n) {
            const int64_t q = s / ax.size[a];
            off += (s - q * ax.size[a]) * ax.stride[a];
            s = q;
        }
    }
    return off;
}

// ---------------------------------------------------------------- overlaps
// M_ab = <u_a(P) | u_b(Q)> over the ncomp components (pythtb.py:3815-3817)
// P, Q point at the components of band 0 at the two mesh points; band b of the
// same point sits `plane` elements further (band-major device layout).
template <int NOCC>
__device__ __forceinline__ void link_matrix(const cd* __restrict__ P, const cd* __restrict__ Q,
                                            const int* occ, int ncomp, int64_t plane, cd (&M)[NOCC][NOCC]) {
#pragma unroll
    for (int a = 0; a < NOCC; ++a)
#pragma unroll
        for (int b = 0; b < NOCC; ++b) M[a][b] = cd{0.0, 0.0};
    for (int o = 0; o < ncomp; ++o) {
        cd pa[NOCC], qb[NOCC];
#pragma unroll
        for (int a = 0; a < NOCC; ++a) {
            pa[a] = P[occ[a] * plane + o];
            qb[a] = Q[occ[a] * plane + o];
        }
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int b = 0; b < NOCC; ++b) cfmac(M[a][b], pa[a], qb[b]);
    }
}

__device__ __forceinline__ cd det2(cd a, cd b, cd c, cd d) { return csub(cmul(a, d), cmul(b, c)); }

template <int NOCC>
__device__ __forceinline__ cd det_small(const cd (&M)[NOCC][NOCC]) {
    if constexpr (NOCC == 1) {
        return M[0][0];
    } else if constexpr (NOCC == 2) {
        return det2(M[0][0], M[0][1], M[1][0], M[1][1]);
    } else if constexpr (NOCC == 3) {
        cd d = cmul(M[0][0], det2(M[1][1], M[1][2], M[2][1], M[2][2]));
        d = csub(d, cmul(M[0][1], det2(M[1][0], M[1][2], M[2][0], M[2][2])));
        d = cadd(d, cmul(M[0][2], det2(M[1][0], M[1][1], M[2][0], M[2][1])));
        return d;
    } else if constexpr (NOCC == 4) {
        // 4x4 through complementary 2x2 minors of rows (0,1) and (2,3)
        auto top = [&](int i, int j) { return det2(M[0][i], M[0][j], M[1][i], M[1][j]); };
        auto bot = [&](int i, int j) { return det2(M[2][i], M[2][j], M[3][i], M[3][j]); };
        cd d = cmul(top(0, 1), bot(2, 3));
        d = csub(d, cmul(top(0, 2), bot(1, 3)));
        d = cadd(d, cmul(top(0, 3), bot(1, 2)));
        d = cadd(d, cmul(top(1, 2), bot(0, 3)));
        d = csub(d, cmul(top(1, 3), bot(0, 2)));
        d = cadd(d, cmul(top(2, 3), bot(0, 1)));
        return d;
    } else {
        // 5..8: LU with partial pivoting, fully unrolled so the matrix stays in registers;
        // the data-dependent row swap is a chain of register selects
        cd A[NOCC][NOCC];
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int b = 0; b < NOCC; ++b) A[a][b] = M[a][b];
        cd det{1.0, 0.0};
#pragma unroll
        for (int c = 0; c < NOCC; ++c) {
            int piv = c;
            double best = cabs2(A[c][c]);
#pragma unroll
            for (int r = c + 1; r < NOCC; ++r) {
                const double v = cabs2(A[r][c]);
                const bool g = v > best;
                best = g ? v : best;
                piv = g ? r : piv;
            }
#pragma unroll
            for (int r = c + 1; r < NOCC; ++r) {
                const bool sw = piv == r;
#pragma unroll
                for (int j = c; j < NOCC; ++j) {
                    const cd x = A[c][j], y = A[r][j];
                    A[c][j] = cd{sw ? y.x : x.x, sw ? y.y : x.y};
                    A[r][j] = cd{sw ? x.x : y.x, sw ? x.y : y.y};
                }
            }
            if (piv != c) det = cd{-det.x, -det.y};
            const cd p = A[c][c];
            det = cmul(det, p);
            const double ip = best > 0.0 ? 1.0 / best : 0.0;     // singular overlap: det becomes 0
            const cd pinv{p.x * ip, -p.y * ip};
#pragma unroll
            for (int r = c + 1; r < NOCC; ++r) {
                const cd f = cmul(A[r][c], pinv);
#pragma unroll
                for (int j = c + 1; j < NOCC; ++j) A[r][j] = csub(A[r][j], cmul(f, A[c][j]));
            }
        }
        return det;
    }
}

// dynamic sizes: matrices live in per-thread local memory, row-major, leading dim n
__device__ inline void link_matrix_dyn(const cd* __restrict__ P, const cd* __restrict__ Q, const int* occ,
                                       int nocc, int ncomp, int64_t plane, cd* M) {
    for (int a = 0; a < nocc; ++a)
        for (int b = 0; b < nocc; ++b) {
            cd acc{0.0, 0.0};
            const cd* pa = P + occ[a] * plane;
            const cd* qb = Q + occ[b] * plane;
            for (int o = 0; o < ncomp; ++o) cfmac(acc, pa[o], qb[o]);
            M[a * nocc + b] = acc;
        }
}

// determinant by LU with partial pivoting (destroys M)
__device__ inline cd det_dyn(int n, cd* M) {
    cd det{1.0, 0.0};
    for (int c = 0; c < n; ++c) {
        int piv = c;
        double best = cabs2(M[c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            const double v = cabs2(M[r * n + c]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
        if (best == 0.0) return cd{0.0, 0.0};
        if (piv != c) {
            for (int j = c; j < n; ++j) {
                const cd t = M[c * n + j];
                M[c * n + j] = M[piv * n + j];
                M[piv * n + j] = t;
            }
            det = cd{-det.x, -det.y};
        }
        const cd p = M[c * n + c];
        det = cmul(det, p);
        const double ip = 1.0 / cabs2(p);
        const cd pinv{p.x * ip, -p.y * ip};
        for (int r = c + 1; r < n; ++r) {
            const cd f = cmul(M[r * n + c], pinv);
            for (int j = c + 1; j < n; ++j) {
                const cd s = cmul(f, M[c * n + j]);
                M[r * n + j] = csub(M[r * n + j], s);
            }
        }
    }
    return det;
}

// unitary polar factor U Vh of svd(M) (pythtb.py:3825-3826) by one-sided
// (Hestenes) Jacobi: G = M V with orthogonal columns, polar = sum_j g_j/|g_j| v_j^H.
// G: in M, out W.  V, T: work.  All n x n row-major.
__device__ inline void polar_dyn(int n, cd* G, cd* V, cd* T) {
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) V[a * n + b] = cd{a == b ? 1.0 : 0.0, 0.0};
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0.0, beta = 0.0;
                cd gam{0.0, 0.0};
                for (int r = 0; r < n; ++r) {
                    const cd x = G[r * n + p], y = G[r * n + q];
                    alpha += cabs2(x);
                    beta += cabs2(y);
                    cfmac(gam, x, y);
                }
                const double g2 = cabs2(gam);
                if (g2 > 0.0 && g2 > 1.0e-31 * alpha * beta) {
                    rotated = true;
                    const double ga = sqrt(g2), inv = 1.0 / ga;
                    const double tau = (beta - alpha) * (0.5 * inv);
                    const double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
                    const cd sw{s * gam.x * inv, s * gam.y * inv};
                    for (int r = 0; r < n; ++r) {
                        cd x = G[r * n + p], y = G[r * n + q];
                        G[r * n + p] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                        G[r * n + q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
                        x = V[r * n + p];
                        y = V[r * n + q];
                        V[r * n + p] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                        V[r * n + q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
                    }
                }
            }
        if (!rotated) break;
    }
    for (int j = 0; j < n; ++j) {
        double nr = 0.0;
        for (int r = 0; r < n; ++r) nr += cabs2(G[r * n + j]);
        const double inv = nr > 0.0 ? 1.0 / sqrt(nr) : 0.0;
        for (int r = 0; r < n; ++r) G[r * n + j] = cscale(G[r * n + j], inv);
    }
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) {
            cd acc{0.0, 0.0};
            for (int j = 0; j < n; ++j) cfma(acc, G[a * n + j], cconj(V[b * n + j]));
            T[a * n + b] = acc;
        }
    for (int e = 0; e < n * n; ++e) G[e] = T[e];
}

// closed-form 2x2 polar factor: W = M (M^+M)^(-1/2),
//   (P)^(-1/2) = adj(P + s I) / (s t),  s = |det M|,  t = sqrt(tr P + 2 s)
__device__ __forceinline__ void polar2(cd (&M)[2][2]) {
    const cd dm = det2(M[0][0], M[0][1], M[1][0], M[1][1]);
    const double s = sqrt(cabs2(dm));
    const double a = cabs2(M[0][0]) + cabs2(M[1][0]);
    const double d = cabs2(M[0][1]) + cabs2(M[1][1]);
    cd b{0.0, 0.0};
    cfmac(b, M[0][0], M[0][1]);
    cfmac(b, M[1][0], M[1][1]);
    const double t = sqrt(a + d + 2.0 * s);
    if (!(s * t > 1.0e-280)) {  // singular overlap: fall back to the Jacobi SVD path
        cd G[4] = {M[0][0], M[0][1], M[1][0], M[1][1]}, V[4], T[4];
        polar_dyn(2, G, V, T);
        M[0][0] = G[0]; M[0][1] = G[1]; M[1][0] = G[2]; M[1][1] = G[3];
        return;
    }
    const double inv = 1.0 / (s * t);
    const cd cb = cconj(b);
    const cd w00 = csub(cscale(M[0][0], d + s), cmul(M[0][1], cb));
    const cd w01 = csub(cscale(M[0][1], a + s), cmul(M[0][0], b));
    const cd w10 = csub(cscale(M[1][0], d + s), cmul(M[1][1], cb));
    const cd w11 = csub(cscale(M[1][1], a + s), cmul(M[1][0], b));
    M[0][0] = cscale(w00, inv);
    M[0][1] = cscale(w01, inv);
    M[1][0] = cscale(w10, inv);
    M[1][1] = cscale(w11, inv);
}

// eigenvalues of a general complex n x n matrix (destroys H): Householder
// Hessenberg reduction, then explicit single-shift QR (all unitary similarities).
// Returns false if an eigenvalue needs more than 60 iterations.
__device__ inline bool eigvals_dyn(int n, cd* H, cd* ev, cd* rc, cd* rs) {
    auto cdiv = [](cd a, cd b) {
        const double ib = 1.0 / cabs2(b);
        return cd{(a.x * b.x + a.y * b.y) * ib, (a.y * b.x - a.x * b.y) * ib};
    };
    auto csqrt_ = [](cd z) {
        const double r = sqrt(cabs2(z));
        if (r == 0.0) return cd{0.0, 0.0};
        double re = sqrt(0.5 * (r + fabs(z.x)));
        double im = 0.5 * z.y / re;
        if (z.x < 0.0) {
            const double t = re;
            re = fabs(im);
            im = copysign(t, z.y);
        }
        return cd{re, im};
    };
    // Householder reduction to Hessenberg form: unitary similarity, so a normal
    // (here: unitary) matrix stays normal and repeated eigenvalues stay well
    // conditioned.  rc[] doubles as the reflector until the QR phase starts.
    for (int m = 0; m < n - 2; ++m) {
        double nr2 = 0.0;
        for (int i = m + 1; i < n; ++i) nr2 += cabs2(H[i * n + m]);
        const cd x0 = H[(m + 1) * n + m];
        if (nr2 - cabs2(x0) == 0.0) continue;
        const double nr = sqrt(nr2), ax = sqrt(cabs2(x0));
        const cd ph = ax > 0.0 ? cscale(x0, 1.0 / ax) : cd{1.0, 0.0};
        for (int i = m + 1; i < n; ++i) rc[i] = H[i * n + m];
        rc[m + 1] = cadd(x0, cscale(ph, nr));
        double vn2 = 0.0;
        for (int i = m + 1; i < n; ++i) vn2 += cabs2(rc[i]);
        const double beta = 2.0 / vn2;
        for (int j = 0; j < n; ++j) {  // H <- (I - beta v v^H) H
            cd dot{0.0, 0.0};
            for (int i = m + 1; i < n; ++i) cfmac(dot, rc[i], H[i * n + j]);
            dot = cscale(dot, beta);
            for (int i = m + 1; i < n; ++i) H[i * n + j] = csub(H[i * n + j], cmul(rc[i], dot));
        }
        for (int r = 0; r < n; ++r) {  // H <- H (I - beta v v^H)
            cd dot{0.0, 0.0};
            for (int i = m + 1; i < n; ++i) cfma(dot, H[r * n + i], rc[i]);
            dot = cscale(dot, beta);
            for (int i = m + 1; i < n; ++i) H[r * n + i] = csub(H[r * n + i], cmul(dot, cconj(rc[i])));
        }
    }
    cd shift{0.0, 0.0};
    int en = n - 1, its = 0;
    bool ok = true;
    while (en >= 0) {
        int l = en;
        for (; l > 0; --l) {
            const double sub = fabs(H[l * n + l - 1].x) + fabs(H[l * n + l - 1].y);
            // diagonal entries carry the accumulated explicit shift
            const cd d0 = cadd(H[(l - 1) * n + l - 1], shift), d1 = cadd(H[l * n + l], shift);
            const double dsum = fabs(d0.x) + fabs(d0.y) + fabs(d1.x) + fabs(d1.y);
            if (sub <= 2.3e-16 * dsum || sub == 0.0) break;
        }
        if (l == en) {
            ev[en] = cadd(H[en * n + en], shift);
            --en;
            its = 0;
            continue;
        }
        if (its >= 60) {
            ok = false;
            ev[en] = cadd(H[en * n + en], shift);
            --en;
            its = 0;
            continue;
        }
        cd s;
        if (its == 10 || its == 20 || its == 40) {
            s = cd{fabs(H[en * n + en - 1].x) + (en >= 2 ? fabs(H[(en - 1) * n + en - 2].x) : 0.0), 0.0};
        } else {  // Wilkinson shift from the trailing 2x2
            s = H[en * n + en];
            const cd x = cmul(H[(en - 1) * n + en], H[en * n + en - 1]);
            if (x.x != 0.0 || x.y != 0.0) {
                const cd y = cscale(csub(H[(en - 1) * n + en - 1], s), 0.5);
                cd z = csqrt_(cadd(cmul(y, y), x));
                if (y.x * z.x + y.y * z.y < 0.0) z = cd{-z.x, -z.y};
                s = csub(s, cdiv(x, cadd(y, z)));
            }
        }
        for (int i = 0; i <= en; ++i) H[i * n + i] = csub(H[i * n + i], s);
        shift = cadd(shift, s);
        ++its;
        // QR sweep on the active block [l,en]: R = G..G (H - sI), H' = R G^H..G^H
        for (int i = l + 1; i <= en; ++i) {
            const cd a = H[(i - 1) * n + i - 1], b = H[i * n + i - 1];
            const double nr = sqrt(cabs2(a) + cabs2(b));
            cd c{1.0, 0.0}, sn{0.0, 0.0};
            if (nr > 0.0) {
                c = cscale(a, 1.0 / nr);
                sn = cscale(b, 1.0 / nr);
            }
            rc[i] = c;
            rs[i] = sn;
            for (int j = i - 1; j <= en; ++j) {
                const cd u = H[(i - 1) * n + j], v = H[i * n + j];
                H[(i - 1) * n + j] = cadd(cmulc(c, u), cmulc(sn, v));   // conj(c) u + conj(s) v
                H[i * n + j] = csub(cmul(c, v), cmul(sn, u));           // -s u + c v
            }
        }
        for (int i = l + 1; i <= en; ++i) {
            const cd c = rc[i], sn = rs[i];
            const int top = i + 1 <= en ? i + 1 : en;
            for (int r = l; r <= top; ++r) {
                const cd u = H[r * n + i - 1], v = H[r * n + i];
                H[r * n + i - 1] = cadd(cmul(u, c), cmul(v, sn));       // u c + v s
                H[r * n + i] = csub(cmul(v, cconj(c)), cmul(u, cconj(sn)));  // -conj(s) u + conj(c) v
            }
        }
    }
    return ok;
}

// ---------------------------------------------------------------- flux
struct FluxArgs {
    WfsView v;
    int nocc;
    int occ[TBK_MAX_NOCC];
    int n0, n1;        // plaquettes along dir0, dir1
    int64_t s0, s1;    // point strides of dir0, dir1
    AxisSet other;     // slice axes
    int bps;           // partial sums per slice (blocks, or wave tiles of the row kernel)
    double* plaq;      // nullable [nslices][n0][n1]
    double* partial;   // [nslices][bps]
    // row-streaming kernel: canonical orientation a = slower-stride axis, b = faster
    int na, nb;        // plaquettes along a, b
    int64_t sa, sb;    // point strides
    int swap;          // dirs[0] is the fast axis: phases negate, output transposes
    int ti;            // plaquette rows per wave tile
    int ncolw;         // wave tiles per row of tiles (63 plaquette columns each)
    int64_t nwaves;    // nslices * bps
    int bpb;           // blocks per slice of the row kernel (4 wave tiles each)
    int fused;         // row kernel: last-arriving block finishes the sum (else k_flux_reduce does)
    int ascending;     // row kernel: tiles dealt oldest rows first (else newest first)
    unsigned* counters;  // [nslices][16] arrival tickets (8 shards + top), zero between launches
    double* totals;    // [nslices]
#ifdef TBK_DIAG
    int ablate;        // diagnostic build only (TBK_ABLATE_FLUX): 1 = no atan2, 2 = no prefetch loads
#endif
};

template <int NOCC, int MAXN>
__device__ __forceinline__ cd one_link_det(const cd* P, const cd* Q, const int* occ, int nocc, int ncomp,
                                           int64_t plane) {
    if constexpr (NOCC > 0) {
        cd M[NOCC][NOCC];
        link_matrix<NOCC>(P, Q, occ, ncomp, plane, M);
        return det_small<NOCC>(M);
    } else {
        cd M[MAXN * MAXN];
        link_matrix_dyn(P, Q, occ, nocc, ncomp, plane, M);
        return det_dyn(nocc, M);
    }
}

// (arg_small_first: tbk_internal.h -- shared with the fused solve + flux kernel of tbk_solve_fused.inl)

// ---- row-streaming flux kernel (ncomp <= 4): a wavefront owns 63 plaquette
// columns x `ti` plaquette rows; lane = mesh column, rows are walked in order.
// Each mesh point's occupied vectors are fetched once per tile (plus the right
// neighbour, an L1 hit), each link determinant is computed once and shared:
//   dV(ia,jb) = det<u(ia,jb)|u(ia+1,jb)>   own columns, neighbour's by shuffle
//   dH(ia,jb) = det<u(ia,jb)|u(ia,jb+1)>   carried from the previous row
//   F(ia,jb)  = -arg[ dV(ia,jb) dH(ia+1,jb) conj dV(ia,jb+1) conj dH(ia,jb) ]
template <int NOCC, int NCOMP>
__device__ __forceinline__ void load_vectors(const cd* __restrict__ p, const int* occ, int64_t plane,
                                             cd (&u)[NOCC][NCOMP]) {
#pragma unroll
    for (int a = 0; a < NOCC; ++a)
#pragma unroll
        for (int o = 0; o < NCOMP; ++o) u[a][o] = p[occ[a] * plane + o];
}

template <int NOCC, int NCOMP>
__device__ __forceinline__ cd det_overlap(const cd (&p)[NOCC][NCOMP], const cd (&q)[NOCC][NCOMP]) {
    cd M[NOCC][NOCC];
#pragma unroll
    for (int a = 0; a < NOCC; ++a)
#pragma unroll
        for (int b = 0; b < NOCC; ++b) {
            cd acc = cmulc(p[a][0], q[b][0]);      // (not 0 + ...: the add of a literal zero is not free)
#pragma unroll
            for (int o = 1; o < NCOMP; ++o) cfmac(acc, p[a][o], q[b][o]);
            M[a][b] = acc;
        }
    return det_small<NOCC>(M);
}

// value of lane + 1; lane 63 keeps its own (DPP wave_shl:1 leaves the destination of a lane without a source untouched)
__device__ __forceinline__ double flux_shl1(const double v) {
    const long long b = __double_as_longlong(v);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi2 << 32) | (unsigned)lo2);
}

template <int NOCC, int NCOMP>
__global__ __launch_bounds__(256) void k_flux_rows(const FluxArgs A) {
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int slice = blockIdx.x / A.bpb;            // blocks never straddle slices
    const int blk = blockIdx.x - slice * A.bpb;
    const bool live = blk * 4 + wib < A.bps;         // idle waves shadow the last tile, contribute 0
    // Tiles are dealt in DESCENDING row order: the array was usually written a moment ago by solve_on_grid, rows
    // ascending, so the rows still sitting in the last-level cache (and not yet written back) are the last ones --
    // reading newest-first takes them from the cache while the older rows' write-back drains.
    const int t = live ? (A.ascending ? blk * 4 + wib : A.bps - 1 - (blk * 4 + wib)) : A.bps - 1;
    const int trow = t / A.ncolw, colw = t - trow * A.ncolw;
    const int ia0 = trow * A.ti;
    const int ia1 = min(ia0 + A.ti, A.na);
    const int jb = colw * 63 + lane;                 // mesh column held by this lane
    const int jbc = min(jb, A.nb);                   // clamp: columns run 0..nb
    const bool has_plaq = live && lane < 63 && jb < A.nb;
    const int64_t plane = A.v.npts * A.v.ncomp;
    const int64_t rstep = A.sa * A.v.ncomp;
    const cd* col = A.v.data + (axis_offset(A.other, slice) + (int64_t)jbc * A.sb) * A.v.ncomp + (int64_t)ia0 * rstep;
    const int64_t per = (int64_t)A.na * A.nb;
    // Each mesh point is fetched ONCE per tile, by the lane of its column; the right neighbour's vectors come from lane + 1 by a
    // one-lane wavefront shift (DPP wave_shl:1 -- lane 63 keeps its own, it owns no plaquette).  Round 4 loaded them: an L1 hit,
    // but twice the load instructions of a kernel that waits on its loads, and 1.137 x the algorithmic bytes at the L2.
    auto from_right = [](const cd (&u)[NOCC][NCOMP], cd (&r)[NOCC][NCOMP]) {
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int o = 0; o < NCOMP; ++o) r[a][o] = cd{flux_shl1(u[a][o].x), flux_shl1(u[a][o].y)};
    };
    cd cur[NOCC][NCOMP], nxt[NOCC][NCOMP], rgt[NOCC][NCOMP], pn[NOCC][NCOMP];
    load_vectors<NOCC, NCOMP>(col, A.occ, plane, cur);
    from_right(cur, rgt);
    cd dHc = det_overlap<NOCC, NCOMP>(cur, rgt);
    col += rstep;
    load_vectors<NOCC, NCOMP>(col, A.occ, plane, nxt);
    double sum = 0.0;
    for (int ia = ia0; ia < ia1; ++ia) {
        // prefetch mesh row ia+2 while row ia+1 is consumed (the tile's last row re-reads itself)
        if (ia + 1 < ia1 && TBK_ABLATE(A.ablate) != 2) col += rstep;
        load_vectors<NOCC, NCOMP>(col, A.occ, plane, pn);
        from_right(nxt, rgt);
        const cd dV = det_overlap<NOCC, NCOMP>(cur, nxt);
        const cd dHn = det_overlap<NOCC, NCOMP>(nxt, rgt);
        const cd dVr{flux_shl1(dV.x), flux_shl1(dV.y)};
        const cd z = cmul(cmul(dV, dHn), cconj(cmul(dVr, dHc)));
        double pha = 0.0;
        if (has_plaq) {
            pha = TBK_ABLATE(A.ablate) == 1 ? -z.y : -arg_small_first(z.y, z.x);
            if (A.swap) pha = -pha;
            if (A.plaq)
                A.plaq[slice * per + (A.swap ? (int64_t)jb * A.na + ia : (int64_t)ia * A.nb + jb)] = pha;
        }
        sum += pha;
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int o = 0; o < NCOMP; ++o) {
                cur[a][o] = nxt[a][o];
                nxt[a][o] = pn[a][o];
            }
        dHc = dHn;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    // ---- deterministic total without a second launch: every block publishes one
    // partial (fixed shape), the block that arrives last (two-level ticket: 8 shard
    // counters, one top counter) sums the slice's partials in a fixed order.
    __shared__ double wsum[4];
    __shared__ double red[256];
    __shared__ int last_flag;
    if (lane == 0) wsum[wib] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        // hand-off without cache-wide fences (a release fence per block costs microseconds):
        // the partial is one 8-byte agent-scope (write-through) store, drained before the
        // ticket; the last block reads the partials with agent-scope loads (L1 bypass).
        const double mine = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        if (!A.fused) {   // a separate k_flux_reduce launch finishes the sum
            A.partial[(int64_t)slice * A.bpb + blk] = mine;
            last_flag = 0;
        } else {
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(A.partial) + (int64_t)slice * A.bpb + blk,
                           (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned* cnt = A.counters + (int64_t)slice * 16;
        const int shard = blk & 7;
        const unsigned shard_size = (unsigned)(A.bpb - shard + 7) / 8u;
        int last = 0;
        if (atomicAdd(cnt + shard, 1u) == shard_size - 1u) {
            const unsigned nshards = (unsigned)min(A.bpb, 8);
            if (atomicAdd(cnt + 8, 1u) == nshards - 1u) last = 1;
        }
        last_flag = last;
        }
    }
    __syncthreads();
    if (last_flag) {
        const unsigned long long* pb = reinterpret_cast<const unsigned long long*>(A.partial) + (int64_t)slice * A.bpb;
        auto ld = [&](int idx) {
            return __longlong_as_double((long long)__hip_atomic_load(pb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        };
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int i = threadIdx.x;
        for (; i + 3 * 256 < A.bpb; i += 4 * 256) {
            s0 += ld(i);
            s1 += ld(i + 256);
            s2 += ld(i + 512);
            s3 += ld(i + 768);
        }
        for (; i < A.bpb; i += 256) s0 += ld(i);
        red[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
#pragma unroll
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) A.totals[slice] = red[0];
        if (threadIdx.x < 9) A.counters[(int64_t)slice * 16 + threadIdx.x] = 0u;   // re-arm for the next launch
    }
}

// ---- planes that do NOT contain the fastest mesh axis (berry_flux of a 3-D array with its default dirs = [0, 1]: one plane per
// index of axis 2).  Along both plane axes neighbouring points are a whole row of the fastest axis apart, so in the row kernel
// above every lane fetches a 32..256-byte vector out of a cache line of its own (2.5 x the time per plaquette of a plane that
// contains the fastest axis, profiles/berry_dirs_probe.py).  Here lane = SLICE: 64 neighbouring planes walk the same plaquette
// row together, their vectors contiguous in memory; a wavefront owns (64 slices) x (row ia) x (a run of columns) and carries the
// two vertical link determinants along.  Partial sums per (slice, ia, run); k_flux_reduce adds them up per slice.
struct FluxSliceArgs {
    int nfast;         // length of the fastest axis (the last entry of A.other)
    int ngrp;          // 64-slice groups along it
    int nrun, run_len; // column runs per row and their length
    int bps;           // partials per slice = n0 * nrun
};
template <int NOCC, int NCOMP>
__global__ __launch_bounds__(256) void k_flux_slices(const FluxArgs A, const FluxSliceArgs S, const int64_t nitems) {
    // one lane per (slow slice index, row ia, column run, fast slice index j), j fastest: a wavefront is 64 consecutive items --
    // neighbouring slices of one run, and at the end of the fastest axis the first slices of the next run (a slice count of
    // 64 q + 1 no longer costs a wavefront for its last slice)
    const int64_t item0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool act = item0 < nitems;
    const int64_t item = act ? item0 : nitems - 1;
    const int64_t t1 = item / S.nfast;
    const int j = (int)(item - t1 * S.nfast);
    const int64_t t2 = t1 / S.nrun;
    const int run = (int)(t1 - t2 * S.nrun);
    const int64_t s_hi = t2 / A.n0;
    const int ia = (int)(t2 - s_hi * A.n0);
    const int64_t slice = s_hi * S.nfast + j;
    const int64_t plane = A.v.npts * NCOMP;
    const int ib0 = run * S.run_len, ib1 = min(ib0 + S.run_len, A.n1);
    int64_t p = axis_offset(A.other, slice) + (int64_t)ia * A.s0 + (int64_t)ib0 * A.s1;
    cd u0[NOCC][NCOMP], u1[NOCC][NCOMP];
    load_vectors<NOCC, NCOMP>(A.v.data + p * NCOMP, A.occ, plane, u0);
    load_vectors<NOCC, NCOMP>(A.v.data + (p + A.s0) * NCOMP, A.occ, plane, u1);
    cd dV = det_overlap<NOCC, NCOMP>(u0, u1);
    double sum = 0.0;
    for (int ib = ib0; __builtin_amdgcn_ballot_w64(ib < ib1) != 0; ++ib) {     // (the last run of a row is shorter)
        const bool on = ib < ib1;
        if (on) p += A.s1;
        cd v0[NOCC][NCOMP], v1[NOCC][NCOMP];
        load_vectors<NOCC, NCOMP>(A.v.data + p * NCOMP, A.occ, plane, v0);
        load_vectors<NOCC, NCOMP>(A.v.data + (p + A.s0) * NCOMP, A.occ, plane, v1);
        const cd dH0 = det_overlap<NOCC, NCOMP>(u0, v0), dH1 = det_overlap<NOCC, NCOMP>(u1, v1), dVn = det_overlap<NOCC, NCOMP>(v0, v1);
        // det<00|10> det<10|11> det<11|01> det<01|00>  (pythtb.py:3852-3863)
        cd d = cmul(dV, dH1);
        d = cmul(d, cconj(dVn));
        d = cmul(d, cconj(dH0));
        if (on) sum += -atan2(d.y, d.x);
        dV = dVn;
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int o = 0; o < NCOMP; ++o) {
                u0[a][o] = v0[a][o];
                u1[a][o] = v1[a][o];
            }
    }
    if (act) A.partial[slice * S.bps + (int64_t)ia * S.nrun + run] = sum;
}

// F(i,j) = -arg[ det<00|10> det<10|11> det<11|01> det<01|00> ]  (pythtb.py:3852-3863)
template <int NOCC, int MAXN>
__global__ __launch_bounds__(256) void k_flux(const FluxArgs A) {
    const int slice = blockIdx.x / A.bps;
    const int blk = blockIdx.x - slice * A.bps;
    const unsigned p = (unsigned)blk * 256u + threadIdx.x;
    const unsigned per = (unsigned)A.n0 * (unsigned)A.n1;
    double pha = 0.0;
    if (p < per) {
        const unsigned i = p / (unsigned)A.n1, j = p - i * (unsigned)A.n1;
        const int nc = A.v.ncomp;
        const int64_t plane = A.v.npts * nc;
        const int64_t base = axis_offset(A.other, slice) + (int64_t)i * A.s0 + (int64_t)j * A.s1;
        const cd* u00 = A.v.data + base * nc;
        const cd* u10 = u00 + A.s0 * nc;
        const cd* u01 = u00 + A.s1 * nc;
        const cd* u11 = u10 + A.s1 * nc;
        cd d = one_link_det<NOCC, MAXN>(u00, u10, A.occ, A.nocc, nc, plane);
        d = cmul(d, one_link_det<NOCC, MAXN>(u10, u11, A.occ, A.nocc, nc, plane));
        d = cmul(d, one_link_det<NOCC, MAXN>(u11, u01, A.occ, A.nocc, nc, plane));
        d = cmul(d, one_link_det<NOCC, MAXN>(u01, u00, A.occ, A.nocc, nc, plane));
        pha = -atan2(d.y, d.x);
        if (A.plaq) A.plaq[(int64_t)slice * per + p] = pha;
    }
    // fixed-shape block sum (bit-reproducible: no atomics)
    __shared__ double red[4];
    double s = pha;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) A.partial[(int64_t)slice * A.bps + blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one block per slice: fixed-shape sum (4 interleaved accumulators per thread so the
// loads overlap, then an LDS tree) -- the same order every run, no atomics
__global__ __launch_bounds__(1024) void k_flux_reduce(const double* __restrict__ partial, int bps,
                                                      double* __restrict__ totals, const DoneArgs done) {
    const double* p = partial + (int64_t)blockIdx.x * bps;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    // (four loads in flight per thread, the tail included: the 2112 partials of a 2048^2 mesh are one round of loads -- this kernel
    // is pure latency in front of every berry_flux return; the shape of the sum depends on bps alone)
    for (int i = threadIdx.x; i < bps; i += 4 * 1024) {
        const double v0 = p[i];
        const double v1 = i + 1024 < bps ? p[i + 1024] : 0.0;
        const double v2 = i + 2048 < bps ? p[i + 2048] : 0.0;
        const double v3 = i + 3072 < bps ? p[i + 3072] : 0.0;
        s0 += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    // fixed-shape tree: xor-butterfly inside each wavefront (the same bits in every lane), then the 16 wavefront sums in
    // index order -- two barriers instead of the ten of an LDS tree over 1024 entries
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) t += red[i];
        red[0] = t;
    }
    if (threadIdx.x == 0) {
        totals[blockIdx.x] = red[0];
        tbk_signal_done(done);      // (the call's last kernel: tbk_berry_flux_result may be polling the completion word)
    }
}

static int check_occ(const tbk_wfs* w, const int32_t* occ, int nocc) {
    TBK_REQUIRE(occ && nocc >= 1, TBK_EINVAL, "occ must list at least one state");
    for (int i = 0; i < nocc; ++i)
        TBK_REQUIRE(occ[i] >= 0 && occ[i] < w->view.nsta, TBK_EINVAL, "occ[%d]=%d outside 0..%d", i, occ[i],
                    w->view.nsta - 1);
    return TBK_OK;
}

static int fill_occ(const tbk_wfs* w, const int32_t* occ, int nocc, int* dst) {
    int rc = check_occ(w, occ, nocc);
    if (rc) return rc;
    TBK_REQUIRE(nocc <= TBK_MAX_NOCC, TBK_EUNSUPPORTED, "nocc=%d exceeds this build's limit of %d", nocc, TBK_MAX_NOCC);
    for (int i = 0; i < nocc; ++i) dst[i] = occ[i];
    return TBK_OK;
}

#include "tbk_berry_big.inl"   // nocc > TBK_MAX_NOCC: link determinants by LU, one workgroup per link
#include "tbk_berry_lanes.inl" // Wilson loops of 3 and 4 bands: a lane per string / per link, vectors through LDS

// scratch of the large-nocc paths: [occ | dets of ndirs directions | LU workspace]
static int big_scratch(tbk_wfs* w, const int32_t* occ, int nocc, int ndirs, size_t extra, int** occ_dev, cd** dets,
                       void** work, size_t* work_bytes, void** extra_dev) {
    tbk_ctx* ctx = w->ctx;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t ob = al((size_t)nocc * sizeof(int));
    const size_t db = al((size_t)w->view.npts * sizeof(cd));
    const size_t per = (size_t)nocc * nocc * sizeof(cd);
    const bool in_lds = (size_t)nocc * (nocc + 1) * sizeof(cd) <= 96 * 1024;
    const size_t wb = in_lds ? 0 : al(per * (size_t)std::min<int64_t>(w->view.npts, (int64_t)ctx->cus * 4));
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + ob + ndirs * db + wb + al(extra), &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;
    *occ_dev = (int*)p;
    *dets = (cd*)(p + ob);
    *work = wb ? (void*)(p + ob + ndirs * db) : nullptr;
    *work_bytes = wb;
    if (extra_dev) *extra_dev = p + ob + ndirs * db + wb;
    TBK_HIP(hipMemcpyAsync(*occ_dev, occ, (size_t)nocc * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return TBK_OK;
}

static void other_axes(const WfsView& v, int skip0, int skip1, AxisSet* ax, int64_t* count) {
    ax->n = 0;
    *count = 1;
    for (int a = 0; a < 3; ++a) {
        ax->size[a] = 1;
        ax->stride[a] = 0;
    }
    for (int d = 0; d < v.dim_arr; ++d) {
        if (d == skip0 || d == skip1) continue;
        ax->size[ax->n] = v.mesh[d];
        ax->stride[ax->n] = v.stride[d];
        ax->n++;
        *count *= v.mesh[d];
    }
}

template <int NOCC, int MAXN>
static void launch_flux(tbk_ctx* ctx, const FluxArgs& A, int64_t nslices) {
    hipLaunchKernelGGL((k_flux<NOCC, MAXN>), dim3((unsigned)(nslices * A.bps)), dim3(256), 0, ctx->stream, A);
}

static int chain_wave_link_dets(tbk_wfs* w, const int32_t* occ, int nocc, int dir, cd* dets_out);   // defined with its kernels below
// Largest L-form tile (above 64 KB the launch sets the kernel's dynamic-LDS attribute).  Products of determinants gain up to 128 KB
// (16 components: 4 bands 152 -> 61 us, 6 bands 222 -> 160 per 513 x 257 array); the two per-link passes of berry_flux only up to ~72 KB
// (4 bands of 16 components 144 -> 121 us; 5 bands, 87 KB: 172 -> 249) -- profiles/berry_cliff_sweep.py
#define TBK_LANES_DET_LDS_MAX ((size_t)128 * 1024)
#define TBK_LANES_FLUX_LDS_MAX ((size_t)72 * 1024)
static bool lanes_dets_applies(const WfsView& v, int nocc, size_t lds_max);
static int lanes_link_dets(tbk_wfs* w, const int32_t* occ, int nocc, int dir, cd* dets_out);          // (tbk_berry_lanes.inl kernels, OUT = 2)
static bool chain_wave_applies(const WfsView& v, int nocc);

extern "C" int tbk_berry_flux_async(tbk_wfs* w, const int32_t* occ, int nocc, int dir0, int dir1,
                                    int want_plaq) {
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_berry_flux: null wfs");
    const WfsView& v = w->view;
    // pythtb.py:3126-3130
    TBK_REQUIRE(dir0 != dir1, TBK_EINVAL, "Need to specify two different directions for Berry flux calculation.");
    TBK_REQUIRE(dir0 >= 0 && dir1 >= 0 && dir0 < v.dim_arr && dir1 < v.dim_arr, TBK_EINVAL,
                "Direction for Berry flux calculation out of bounds.");
    FluxArgs A{};
    int det_from = 9;      // up to 8 bands the register LU per thread wins; from 9 the workgroup-per-link LU is 4-100x faster (profiles/det_big_probe.py)
    if (tbk_knobs().det_big_from >= 0) det_from = std::max(2, tbk_knobs().det_big_from);
    const bool big = nocc >= det_from;        // link determinants by LU (tbk_berry_big.inl)
    int rc = big ? check_occ(w, occ, nocc) : fill_occ(w, occ, nocc, A.occ);
    if (rc) return rc;
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    A.v = v;
    A.nocc = nocc;
    A.n0 = v.mesh[dir0] - 1;
    A.n1 = v.mesh[dir1] - 1;
    A.s0 = v.stride[dir0];
    A.s1 = v.stride[dir1];
    int64_t nslices = 1;
    other_axes(v, dir0, dir1, &A.other, &nslices);
    const int64_t per = (int64_t)A.n0 * A.n1;
    TBK_REQUIRE(per < (int64_t)0xffffffffu, TBK_EUNSUPPORTED, "plane of %lld plaquettes is too large", (long long)per);
    const bool rows = !big && v.ncomp <= 4 && nocc <= v.ncomp;   // register-resident row-streaming kernel
    if (rows) {
        A.swap = A.s0 < A.s1;                            // lanes run along the smaller stride
        A.na = A.swap ? A.n1 : A.n0;
        A.nb = A.swap ? A.n0 : A.n1;
        A.sa = A.swap ? A.s1 : A.s0;
        A.sb = A.swap ? A.s0 : A.s1;
        A.ncolw = (A.nb + 62) / 63;
        // enough wave tiles to fill the chip, rows per tile in [4, 24].  A tile re-reads one halo row, so its rows set the traffic:
        // 8-row tiles (32 tiles per CU, rounds 1-4) moved 1.13 x the algorithmic bytes; the time is flat from 8 to 24 rows
        // (2048^2: 28.8 / 26.2 / 26.4 / 24.1 us at 8 / 12 / 16 / 24 rows, 4096^2: 100 / 100 / 100 / 104) and rises beyond
        // (32: 29.1 / 106, 64: 36.8 / 108) -- so 12 tiles per CU, at most 24 rows: 1.04 x
        const int64_t want = (int64_t)ctx->cus * 12;
        int64_t ti = ((int64_t)A.na * A.ncolw * nslices + want - 1) / want;
        A.ti = (int)std::max<int64_t>(4, std::min<int64_t>(24, ti));
        if (tbk_knobs().flux_ti >= 0) A.ti = std::max(1, tbk_knobs().flux_ti);
        A.bps = ((A.na + A.ti - 1) / A.ti) * A.ncolw;
    } else {
        A.bps = (int)((per + 255) / 256);
    }
    // planes without the fastest mesh axis: lanes along the slices (k_flux_slices; TBK_FLUX_SLICES=0: the row kernel)
    FluxSliceArgs SL{};
    const int Dm = v.dim_arr;
    const bool slices_k = rows && !want_plaq && Dm >= 3 && dir0 != Dm - 1 && dir1 != Dm - 1 && v.mesh[Dm - 1] >= 16 &&
                          tbk_knobs().flux_slices != 0;
    if (slices_k) {
        SL.nfast = v.mesh[Dm - 1];
        SL.ngrp = (SL.nfast + 63) / 64;
        const int64_t rows_w = (nslices / SL.nfast) * SL.ngrp * A.n0;      // wavefronts with one run per row
        const int64_t want = (int64_t)ctx->cus * 8;
        SL.nrun = (int)std::max<int64_t>(1, std::min<int64_t>((A.n1 + 7) / 8, (want + rows_w - 1) / rows_w));
        SL.run_len = (A.n1 + SL.nrun - 1) / SL.nrun;
        SL.nrun = (A.n1 + SL.run_len - 1) / SL.run_len;
        SL.bps = A.n0 * SL.nrun;
        A.bps = SL.bps;
    }
    A.nwaves = nslices * A.bps;
    A.bpb = (A.bps + 3) / 4;
    A.fused = tbk_knobs().flux_fused;
    A.ascending = tbk_knobs().flux_order == 1 ? 1 : 0;
#ifdef TBK_DIAG
    A.ablate = tbk_knobs().ablate_flux;
#endif
    TBK_REQUIRE(nslices * A.bps < (int64_t)0x7fffffff, TBK_EUNSUPPORTED, "too many plaquette blocks");
    if (w->flux_nslices_cap < nslices) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (w->flux_cnt_dev) TBK_HIP(hipFree(w->flux_cnt_dev));
        w->flux_cnt_dev = nullptr;
        w->flux_nslices_cap = 0;
        {
            const int rct = tbk_wfs_totals_alloc(w, nslices);
            if (rct) return rct;
        }
        TBK_HIP(hipMalloc((void**)&w->flux_cnt_dev, nslices * 16 * sizeof(unsigned)));
        TBK_HIP(hipMemsetAsync(w->flux_cnt_dev, 0, nslices * 16 * sizeof(unsigned), ctx->stream));
        w->flux_nslices_cap = nslices;
    }
    w->flux_nslices = nslices;
    A.counters = w->flux_cnt_dev;
    A.totals = w->flux_totals_dev;
    if (w->flux_partial_cap < nslices * A.bps) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (w->flux_partial_dev) TBK_HIP(hipFree(w->flux_partial_dev));
        w->flux_partial_dev = nullptr;
        TBK_HIP(hipMalloc((void**)&w->flux_partial_dev, nslices * A.bps * sizeof(double)));
        w->flux_partial_cap = nslices * A.bps;
    }
    w->flux_plaq_n = 0;
    if (want_plaq) {
        if (w->flux_plaq_cap < nslices * per) {
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (w->flux_plaq_dev) TBK_HIP(hipFree(w->flux_plaq_dev));
            w->flux_plaq_dev = nullptr;
            hipError_t e = hipMalloc((void**)&w->flux_plaq_dev, nslices * per * sizeof(double));
            if (e != hipSuccess) {
                w->flux_plaq_cap = 0;
                tbk_set_error("tbk_berry_flux: plaquette buffer of %lld doubles: %s", (long long)(nslices * per),
                              hipGetErrorString(e));
                return TBK_ENOMEM;
            }
            w->flux_plaq_cap = nslices * per;
        }
        w->flux_plaq_n = nslices * per;
        A.plaq = w->flux_plaq_dev;
    }
    A.partial = w->flux_partial_dev;
    const bool lanes_flux = !big && !rows && lanes_dets_applies(v, nocc, TBK_LANES_FLUX_LDS_MAX) && tbk_knobs().chain_wave != 2;
    if (!big && (chain_wave_applies(v, nocc) || lanes_flux)) {
        // 5..8 bands of wide states: every lane of the plaquette kernel would walk its own 256-byte rows; instead the link
        // determinants along both directions come from the wave-per-string kernels (coalesced), then the same combine
        // kernel as for large band sets
        void* base = nullptr;
        const size_t db = ((size_t)v.npts * sizeof(cd) + 255) & ~(size_t)255;
        rc = tbk_ctx_scratch(ctx, 256 + 2 * db, &base);
        if (rc) return rc;
        cd* dets = (cd*)((unsigned char*)base + 256);
        // (1..4 bands of states with fewer than 8 components, round 6: the LDS-tile kernels of tbk_berry_lanes.inl)
        rc = lanes_flux ? lanes_link_dets(w, occ, nocc, dir0, dets) : chain_wave_link_dets(w, occ, nocc, dir0, dets);
        if (rc) return rc;
        rc = lanes_flux ? lanes_link_dets(w, occ, nocc, dir1, (cd*)((unsigned char*)dets + db))
                        : chain_wave_link_dets(w, occ, nocc, dir1, (cd*)((unsigned char*)dets + db));
        if (rc) return rc;
        PlaqDetArgs P{};
        P.d0 = dets;
        P.d1 = (cd*)((unsigned char*)dets + db);
        P.n0 = A.n0;
        P.n1 = A.n1;
        P.s0 = A.s0;
        P.s1 = A.s1;
        P.other = A.other;
        P.bps = A.bps;
        P.plaq = A.plaq;
        P.partial = A.partial;
        ProfScope ps(ctx, "berry_flux");
        hipLaunchKernelGGL(k_flux_from_dets, dim3((unsigned)(nslices * A.bps)), dim3(256), 0, ctx->stream, P);
        TBK_HIP(hipGetLastError());
    } else if (big) {
        int* occ_dev = nullptr;
        cd* dets = nullptr;
        void* work = nullptr;
        size_t work_bytes = 0;
        rc = big_scratch(w, occ, nocc, 2, 0, &occ_dev, &dets, &work, &work_bytes, nullptr);
        if (rc) return rc;
        rc = launch_link_dets(w, occ_dev, nocc, dir0, dets, work, work_bytes);
        if (rc) return rc;
        rc = launch_link_dets(w, occ_dev, nocc, dir1, dets + v.npts, work, work_bytes);
        if (rc) return rc;
        PlaqDetArgs P{};
        P.d0 = dets;
        P.d1 = dets + v.npts;
        P.n0 = A.n0;
        P.n1 = A.n1;
        P.s0 = A.s0;
        P.s1 = A.s1;
        P.other = A.other;
        P.bps = A.bps;
        P.plaq = A.plaq;
        P.partial = A.partial;
        ProfScope ps(ctx, "berry_flux");
        hipLaunchKernelGGL(k_flux_from_dets, dim3((unsigned)(nslices * A.bps)), dim3(256), 0, ctx->stream, P);
        TBK_HIP(hipGetLastError());
    } else {
        ProfScope ps(ctx, "berry_flux");
        if (slices_k) {
            const int64_t nitems = nslices * A.n0 * SL.nrun;          // (slices x rows x runs)
            const dim3 grid((unsigned)((nitems + 255) / 256)), blk(256);
#define TBK_SLC(NO, NC) hipLaunchKernelGGL((k_flux_slices<NO, NC>), grid, blk, 0, ctx->stream, A, SL, nitems)
            switch (v.ncomp * 8 + nocc) {
                case 1 * 8 + 1: TBK_SLC(1, 1); break;
                case 2 * 8 + 1: TBK_SLC(1, 2); break;
                case 2 * 8 + 2: TBK_SLC(2, 2); break;
                case 3 * 8 + 1: TBK_SLC(1, 3); break;
                case 3 * 8 + 2: TBK_SLC(2, 3); break;
                case 3 * 8 + 3: TBK_SLC(3, 3); break;
                case 4 * 8 + 1: TBK_SLC(1, 4); break;
                case 4 * 8 + 2: TBK_SLC(2, 4); break;
                case 4 * 8 + 3: TBK_SLC(3, 4); break;
                default: TBK_SLC(4, 4); break;
            }
#undef TBK_SLC
        } else if (rows) {
            const dim3 grid((unsigned)(nslices * A.bpb)), blk(256);
#define TBK_ROWS(NO, NC) hipLaunchKernelGGL((k_flux_rows<NO, NC>), grid, blk, 0, ctx->stream, A)
            switch (v.ncomp * 8 + nocc) {
                case 1 * 8 + 1: TBK_ROWS(1, 1); break;
                case 2 * 8 + 1: TBK_ROWS(1, 2); break;
                case 2 * 8 + 2: TBK_ROWS(2, 2); break;
                case 3 * 8 + 1: TBK_ROWS(1, 3); break;
                case 3 * 8 + 2: TBK_ROWS(2, 3); break;
                case 3 * 8 + 3: TBK_ROWS(3, 3); break;
                case 4 * 8 + 1: TBK_ROWS(1, 4); break;
                case 4 * 8 + 2: TBK_ROWS(2, 4); break;
                case 4 * 8 + 3: TBK_ROWS(3, 4); break;
                default: TBK_ROWS(4, 4); break;
            }
#undef TBK_ROWS
        } else {
            switch (nocc) {
                case 1: launch_flux<1, 1>(ctx, A, nslices); break;
                case 2: launch_flux<2, 1>(ctx, A, nslices); break;
                case 3: launch_flux<3, 1>(ctx, A, nslices); break;
                case 4: launch_flux<4, 1>(ctx, A, nslices); break;
                case 5: launch_flux<5, 1>(ctx, A, nslices); break;
                case 6: launch_flux<6, 1>(ctx, A, nslices); break;
                case 7: launch_flux<7, 1>(ctx, A, nslices); break;
                case 8: launch_flux<8, 1>(ctx, A, nslices); break;
                default: launch_flux<0, TBK_MAX_NOCC>(ctx, A, nslices);
            }
        }
        TBK_HIP(hipGetLastError());
    }
    if (!rows || !A.fused || slices_k) {
        ProfScope ps(ctx, "flux_reduce");
        // totals in mapped host memory and no per-plaquette output: this is the call's last kernel, the result call polls its
        // completion word instead of synchronising the stream
        w->flux_done = (w->flux_totals_host && !want_plaq) ? tbk_done_arm(ctx, false) : DoneArgs{nullptr, nullptr, nullptr, 0u};
        hipLaunchKernelGGL(k_flux_reduce, dim3((unsigned)nslices), dim3(1024), 0, ctx->stream,
                           (const double*)w->flux_partial_dev, (rows && !slices_k) ? A.bpb : A.bps, w->flux_totals_dev, w->flux_done);
        TBK_HIP(hipGetLastError());
    } else {
        w->flux_done = DoneArgs{nullptr, nullptr, nullptr, 0u};
    }
    return TBK_OK;
}

extern "C" int tbk_berry_flux_result(tbk_wfs* w, double* totals, double* plaq) {
    TBK_REQUIRE(w && totals, TBK_EINVAL, "tbk_berry_flux_result: null argument");
    TBK_REQUIRE(w->flux_nslices > 0, TBK_EINVAL, "tbk_berry_flux_result: no flux launch pending");
    tbk_ctx* ctx = w->ctx;
    if (plaq) {
        TBK_REQUIRE(w->flux_plaq_n > 0, TBK_EINVAL, "tbk_berry_flux_result: plaquettes were not requested");
        TBK_HIP(hipMemcpyAsync(plaq, w->flux_plaq_dev, w->flux_plaq_n * sizeof(double), hipMemcpyDeviceToHost,
                               ctx->stream));
    }
    if (w->flux_totals_host) {                       // mapped host memory: the kernel's stores were the transfer
        const DoneArgs done = plaq ? DoneArgs{nullptr, nullptr, nullptr, 0u} : w->flux_done;
        w->flux_done = DoneArgs{nullptr, nullptr, nullptr, 0u};
        const int rc = tbk_done_wait(ctx, done);
        if (rc) return rc;
        memcpy(totals, w->flux_totals_host, w->flux_nslices * sizeof(double));
        return TBK_OK;
    }
    return tbk_small_d2h(ctx, totals, w->flux_totals_dev, w->flux_nslices * sizeof(double));
}

extern "C" int tbk_berry_flux(tbk_wfs* w, const int32_t* occ, int nocc, int dir0, int dir1, double* totals,
                              double* plaq) {
    int rc = tbk_berry_flux_async(w, occ, nocc, dir0, dir1, plaq != nullptr);
    if (rc) return rc;
    return tbk_berry_flux_result(w, totals, plaq);
}

// ---------------------------------------------------------------- strings
struct ChainArgs {
    WfsView v;
    int nocc;
    int occ[TBK_MAX_NOCC];
    int nlinks;        // mesh[dir]-1
    int64_t sdir;      // point stride along dir
    AxisSet other;     // string axes (original order)
    int64_t nstrings;
    int seg_len, nseg;
    int final_lanes;   // k_chain_final_wave: lanes taking part in the ordered tree
    cd* partial;      // det: [nseg][nstrings]; evals: [nseg][nstrings][nocc*nocc]
    double* out;       // det: [nstrings]; evals: [nstrings][nocc]
    int* flags;
};

// ---- det-type Berry phase and flux, 1..8 occupied bands of WIDE states (ncomp >= 8; BASELINE configs[4]: 8 of 16
// bands, 16 components): two kernels, each a WAVEFRONT per (string, segment).
// With a thread per string every lane walks its own 256-byte rows and a load instruction touches 64 different
// cache lines: strings along the last mesh axis over-fetched 8x and ran at 0.4 TB/s (82 ms for the 257^3 array,
// profiles/r02a).
//   k_chain_links_wave  streams the points of ONE string: a point's occupied vectors are NOCC rows of ncomp
//     contiguous c128, loaded by the whole wave (each point once, two points ahead in flight), staged in LDS; the link
//     matrix M[a][b] = <u_a(i) | u_b(i+1)> is one entry per lane (lane = a NOCC + b) and leaves as one contiguous
//     NOCC^2 x 16 B store.  Few registers, 4 KB of LDS per wave: full occupancy.
//   k_chain_lu_wave     reads the link matrices back 32 at a time (coalesced), turns them through LDS so that a lane
//     holds one whole matrix, runs the register-resident pivoted LU (det_small) -- one LU shared by 64 lanes through
//     shuffles would cost ~600 instructions per link, this ~25 -- and multiplies the determinants with a fixed-shape
//     tree, passes in order: bit-reproducible.
// The workspace holds NOCC^2 c128 per link; strings go in batches that bound it (tbk_berry_phase).
// LDS hand-off between the lanes of ONE wavefront: wait for this wave's LDS operations only.  (A release fence
// would also wait for vmcnt(0) -- every global prefetch and every store in flight: measured 15 us per link.)
__device__ __forceinline__ void lds_sync_wave() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");   // LDS address space only
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}

template <int NOCC, int NLD>   // NLD = ceil(NOCC * ncomp / 64): 16-byte loads per lane and point
__global__ __launch_bounds__(256, (NLD <= 2 ? 5 : 4)) void k_chain_links_wave(const ChainArgs A, const int64_t s0, const int64_t ns, cd* __restrict__ ws) {
    extern __shared__ __align__(16) unsigned char chainw_lds[];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + wib;
    if (t >= ns * A.nseg) return;                    // (no workgroup barrier below: waves are independent)
    const int64_t seg = t / ns, sl = t - seg * ns, s = s0 + sl;
    const int ncomp = A.v.ncomp;
    const int ldp = ncomp + 1;                       // row stride of a staged point: rows of different bands on distinct banks
    const int pbuf = NOCC * ldp + 1;                 // (+1: the slot where lanes past the point's last element park)
    cd* bufP = reinterpret_cast<cd*>(chainw_lds) + (size_t)wib * 2 * pbuf;
    cd* bufN = bufP + pbuf;
    const int64_t plane = A.v.npts * ncomp;
    const int i0 = (int)seg * A.seg_len;
    const int i1 = min(i0 + A.seg_len, A.nlinks);
    const cd* P = A.v.data + (axis_offset(A.other, s) + (int64_t)i0 * A.sdir) * ncomp;
    const int64_t step = A.sdir * ncomp;
    const int nel = NOCC * ncomp;                    // c128 per point
    const int a_of = lane / NOCC, b_of = lane - a_of * NOCC;
    const bool pair = lane < NOCC * NOCC;
    // element e = j * 64 + lane of a point: band slot e / ncomp, component e % ncomp.  Loads and LDS writes are
    // UNCONDITIONAL (a lane past the last element re-reads element 0 and parks it in the spare slot): a fixed number of
    // memory operations per step keeps the values in registers and the waits counted.
    int64_t goff[NLD];
    int dst[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = j * 64 + lane;
        const bool ok = e < nel;
        const int a = ok ? e / ncomp : 0, c = ok ? e - a * ncomp : 0;
        goff[j] = (int64_t)A.occ[a] * plane + c;
        dst[j] = ok ? a * ldp + c : NOCC * ldp;
    }
    // (named scalars, not arrays: as arrays the two register sets of the pipeline ended up in scratch memory, with a
    // full vmcnt wait behind every prefetch)
    typedef double v2d __attribute__((ext_vector_type(2)));   // (a first-class vector value: the struct cd stayed an alloca here)
    v2d a0 = {0.0, 0.0}, a1 = a0, a2 = a0, a3 = a0, b0 = a0, b1 = a0, b2 = a0, b3 = a0;   // two points in flight
#define TBK_LD4(x0, x1, x2, x3, ptr)                                                  \
    {                                                                                 \
        x0 = *reinterpret_cast<const v2d*>((ptr) + goff[0]);                          \
        if constexpr (NLD > 1) x1 = *reinterpret_cast<const v2d*>((ptr) + goff[1]);   \
        if constexpr (NLD > 2) x2 = *reinterpret_cast<const v2d*>((ptr) + goff[2]);   \
        if constexpr (NLD > 3) x3 = *reinterpret_cast<const v2d*>((ptr) + goff[3]);   \
    }
#define TBK_PARK4(buf, x0, x1, x2, x3)                                                \
    {                                                                                 \
        *reinterpret_cast<v2d*>((buf) + dst[0]) = x0;                                 \
        if constexpr (NLD > 1) *reinterpret_cast<v2d*>((buf) + dst[1]) = x1;          \
        if constexpr (NLD > 2) *reinterpret_cast<v2d*>((buf) + dst[2]) = x2;          \
        if constexpr (NLD > 3) *reinterpret_cast<v2d*>((buf) + dst[3]) = x3;          \
    }
    TBK_LD4(a0, a1, a2, a3, P)
    TBK_PARK4(bufP, a0, a1, a2, a3)
    {
        const cd* p1 = P + step;                                      // point i0 + 1
        const cd* p2 = P + (int64_t)min(2, i1 - i0) * step;           // point i0 + 2 (clamped: the segment's last point again)
        TBK_LD4(a0, a1, a2, a3, p1)
        TBK_LD4(b0, b1, b2, b3, p2)
    }
    cd* out = ws + ((int64_t)sl * A.nlinks + i0) * (NOCC * NOCC);
    // One link: the register set that holds point i + 1 is parked in LDS and immediately re-used for the load of point
    // i + 3, while the OTHER set (point i + 2) stays in flight.  The loop is unrolled by two with the sets swapping roles --
    // rotating them with register copies made every iteration wait for the younger load (a v_mov of a value still in flight),
    // i.e. one point of prefetch instead of two: 1.0 ms per 2.1 M links whatever the band count, 0.58 ms like this.  (Three
    // sets, unrolled by three, fell back to 1.0 ms: the waits at the merged loop head became vmcnt(0) again.)
#define TBK_LINK_STEP(i, x0, x1, x2, x3)                                                                  \
    {                                                                                                     \
        TBK_PARK4(bufN, x0, x1, x2, x3)                               /* point i + 1 has arrived */        \
        const cd* p3 = P + (int64_t)min((i) + 3 - i0, i1 - i0) * step; /* point i + 3 goes in flight */    \
        TBK_LD4(x0, x1, x2, x3, p3)                                                                       \
        lds_sync_wave();                                                                                  \
        if (pair) {                                                                                       \
            const cd* ua = bufP + a_of * ldp;                                                             \
            const cd* ub = bufN + b_of * ldp;                                                             \
            /* two interleaved accumulators per part: the dependent chain is ncomp/2 FMAs long */         \
            cd m0 = cmulc(ua[0], ub[0]), m1 = cmulc(ua[1], ub[1]);                                        \
            int c = 2;                                                                                    \
            for (; c + 1 < ncomp; c += 2) {                                                               \
                cfmac(m0, ua[c], ub[c]);                                                                  \
                cfmac(m1, ua[c + 1], ub[c + 1]);                                                          \
            }                                                                                             \
            if (c < ncomp) cfmac(m0, ua[c], ub[c]);                                                       \
            out[(int64_t)((i) - i0) * (NOCC * NOCC) + lane] = cadd(m0, m1);                               \
        }                                                                                                 \
        cd* tmp = bufP;                                                                                   \
        bufP = bufN;                                                                                      \
        bufN = tmp;                                                                                       \
        lds_sync_wave();                                                                                  \
    }
    int i = i0;
    for (; i + 1 < i1; i += 2) {
        TBK_LINK_STEP(i, a0, a1, a2, a3)
        TBK_LINK_STEP(i + 1, b0, b1, b2, b3)
    }
    if (i < i1) TBK_LINK_STEP(i, a0, a1, a2, a3)
#undef TBK_LINK_STEP
#undef TBK_LD4
#undef TBK_PARK4
}

// ---- the same link matrices for 5..8 bands, FOUR links per step and a 2 x 2 block of M per lane (lane = link g, block (ta, tb)).
// k_chain_links_wave above spends a whole wavefront step on one point: 33 LDS instructions and 135 vector instructions (most of
// them 64-bit index arithmetic) for 64 FMAs per lane, two barriers, and 2 x 2 KB in flight per wavefront -- it moves 3 TB/s with
// the LDS 33 % and the VALU 32 % busy (profiles/r03jcfg): latency.  Here a step loads four points at once (8 KB in flight per
// wavefront), a lane reads 2 + 2 staged rows for 4 entries of M (half the LDS reads per entry), and the index arithmetic, the
// barriers and the waits are paid once per four links.  The accumulation order of an entry is the one of the kernel above
// (even and odd components in two accumulators).  The workspace layout is unchanged: k_chain_lu_wave reads it.
#define TBK_CHAINT_G 8   // links per wavefront step (a multiple of 4: sixteen lanes per link, four links per pass)
template <int NOCC, int NLD>
__global__ __launch_bounds__(256) void k_chain_links_tile(const ChainArgs A, const int64_t s0, const int64_t ns, cd* __restrict__ ws) {
    static_assert(NOCC >= 5 && NOCC <= 8, "k_chain_links_tile: 5..8 bands");
    extern __shared__ __align__(16) unsigned char chainw_lds[];
    constexpr int G = TBK_CHAINT_G, NT = (NOCC + 1) / 2, NN = NOCC * NOCC;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;   // (2 or 4 wavefronts per block: whatever fits 64 KB of LDS)
    if (t >= ns * A.nseg) return;                    // (no workgroup barrier below: waves are independent)
    const int64_t seg = t / ns, sl = t - seg * ns, s = s0 + sl;
    const int ncomp = A.v.ncomp;
    const int ldp = ncomp + 1;
    const int pbuf = NOCC * ldp + 1;
    cd* const buf = reinterpret_cast<cd*>(chainw_lds) + (size_t)wib * (G + 1) * pbuf;   // slots 0 .. G: points i .. i + G
    const int64_t plane = A.v.npts * ncomp;
    const int i0 = (int)seg * A.seg_len;
    const int i1 = min(i0 + A.seg_len, A.nlinks);
    const int np = i1 - i0;                          // links of this segment; its points are 0 .. np
    const cd* P = A.v.data + (axis_offset(A.other, s) + (int64_t)i0 * A.sdir) * ncomp;
    const int64_t step = A.sdir * ncomp;
    const int nel = NOCC * ncomp;
    int64_t goff[NLD];
    int dst[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = j * 64 + lane;
        const bool ok = e < nel;
        const int a = ok ? e / ncomp : 0, c = ok ? e - a * ncomp : 0;
        goff[j] = (int64_t)A.occ[a] * plane + c;
        dst[j] = ok ? a * ldp + c : NOCC * ldp;
    }
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d x[G][NLD], xl[NLD];
    auto load_group = [&](const int first) {         // points first .. first + G - 1 of the segment (clamped to its last point)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const cd* p = P + (int64_t)min(first + g, np) * step;
#pragma unroll
            for (int j = 0; j < NLD; ++j) x[g][j] = *reinterpret_cast<const v2d*>(p + goff[j]);
        }
    };
#pragma unroll
    for (int j = 0; j < NLD; ++j) xl[j] = *reinterpret_cast<const v2d*>(P + goff[j]);   // point 0
    load_group(1);
    // this lane's block of M
    const int g4 = lane >> 4, tl = lane & 15;
    const int ta = tl / NT, tb = tl - ta * NT;
    const bool active = tl < NT * NT;
    const int a0 = min(2 * ta, NOCC - 1), a1 = min(2 * ta + 1, NOCC - 1), b0 = min(2 * tb, NOCC - 1), b1 = min(2 * tb + 1, NOCC - 1);
    const bool va1 = 2 * ta + 1 < NOCC, vb1 = 2 * tb + 1 < NOCC;
    cd* const out = ws + ((int64_t)sl * A.nlinks + i0) * NN;
    for (int i = 0; i < np; i += G) {
        // points i (kept from the previous step) and i + 1 .. i + G (arriving) go to LDS; the registers take the next group
#pragma unroll
        for (int j = 0; j < NLD; ++j) *reinterpret_cast<v2d*>(buf + dst[j]) = xl[j];
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int j = 0; j < NLD; ++j) *reinterpret_cast<v2d*>(buf + (gg + 1) * pbuf + dst[j]) = x[gg][j];
#pragma unroll
        for (int j = 0; j < NLD; ++j) xl[j] = x[G - 1][j];
        load_group(i + G + 1);
        lds_sync_wave();
#pragma unroll
        for (int gp = 0; gp < G; gp += 4) {
        const int g = g4 + gp;
        if (active && i + g < np) {
            const cd* pa = buf + g * pbuf;
            const cd* pb = pa + pbuf;
            const cd *ua0 = pa + a0 * ldp, *ua1 = pa + a1 * ldp, *ub0 = pb + b0 * ldp, *ub1 = pb + b1 * ldp;
            cd m00 = cmulc(ua0[0], ub0[0]), m01 = cmulc(ua0[0], ub1[0]), m10 = cmulc(ua1[0], ub0[0]), m11 = cmulc(ua1[0], ub1[0]);
            cd n00 = cmulc(ua0[1], ub0[1]), n01 = cmulc(ua0[1], ub1[1]), n10 = cmulc(ua1[1], ub0[1]), n11 = cmulc(ua1[1], ub1[1]);
            int c = 2;
            for (; c + 1 < ncomp; c += 2) {
                const cd p0 = ua0[c], p1 = ua1[c], q0 = ub0[c], q1 = ub1[c];
                cfmac(m00, p0, q0);
                cfmac(m01, p0, q1);
                cfmac(m10, p1, q0);
                cfmac(m11, p1, q1);
                const cd r0 = ua0[c + 1], r1 = ua1[c + 1], t0 = ub0[c + 1], t1 = ub1[c + 1];
                cfmac(n00, r0, t0);
                cfmac(n01, r0, t1);
                cfmac(n10, r1, t0);
                cfmac(n11, r1, t1);
            }
            if (c < ncomp) {
                const cd p0 = ua0[c], p1 = ua1[c], q0 = ub0[c], q1 = ub1[c];
                cfmac(m00, p0, q0);
                cfmac(m01, p0, q1);
                cfmac(m10, p1, q0);
                cfmac(m11, p1, q1);
            }
            cd* o = out + (int64_t)(i + g) * NN;
            o[a0 * NOCC + b0] = cadd(m00, n00);
            if (vb1) o[a0 * NOCC + b1] = cadd(m01, n01);
            if (va1) o[a1 * NOCC + b0] = cadd(m10, n10);
            if (va1 && vb1) o[a1 * NOCC + b1] = cadd(m11, n11);
        }
        }
        lds_sync_wave();
    }
}

#define TBK_CHAINW_PASS 32
template <int NOCC>
__global__ __launch_bounds__(64, 1) void k_chain_lu_wave(const ChainArgs A, const int64_t s0, const int64_t ns, const cd* __restrict__ ws,
                                                         cd* __restrict__ dets_out) {
    extern __shared__ __align__(16) unsigned char chainw_lds[];
    constexpr int NN = NOCC * NOCC, LDM = NN + 1;    // padded row: 32 readers on distinct banks
    cd* Mbuf = reinterpret_cast<cd*>(chainw_lds);
    const int lane = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int64_t seg = t / ns, sl = t - seg * ns, s = s0 + sl;
    const int i0 = (int)seg * A.seg_len;
    const int i1 = min(i0 + A.seg_len, A.nlinks);
    const cd* in = ws + ((int64_t)sl * A.nlinks + i0) * NN;
    cd acc{1.0, 0.0};
    for (int ib = i0; ib < i1; ib += TBK_CHAINW_PASS) {
        const int nb = min(TBK_CHAINW_PASS, i1 - ib);
        const int total = nb * NN;                   // contiguous c128 of this pass
        // eight loads in flight per lane (one at a time, each waited for, was 20 us per pass: the whole kernel)
        typedef double v2d __attribute__((ext_vector_type(2)));
        const v2d* src = reinterpret_cast<const v2d*>(in + (int64_t)(ib - i0) * NN);
        constexpr int PER = (TBK_CHAINW_PASS * NN + 63) / 64;
#pragma unroll
        for (int g = 0; g < PER; g += 8) {
            v2d r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (g + u) * 64 + lane;
                r[u] = g + u < PER && e < total ? src[e] : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = (g + u) * 64 + lane;
                if (g + u < PER && e < total) {
                    const int j = e / NN, k = e - j * NN;
                    *reinterpret_cast<v2d*>(Mbuf + j * LDM + k) = r[u];
                }
            }
        }
        lds_sync_wave();
        cd d{1.0, 0.0};
        if (lane < nb) {
            cd M[NOCC][NOCC];
#pragma unroll
            for (int a = 0; a < NOCC; ++a)
#pragma unroll
                for (int b = 0; b < NOCC; ++b) M[a][b] = Mbuf[lane * LDM + a * NOCC + b];
            d = det_small<NOCC>(M);
            // berry_flux wants the determinant of every link on its own, filed under the link's first point
            if (dets_out) dets_out[axis_offset(A.other, s) + (int64_t)(ib + lane) * A.sdir] = d;
        }
        // ordered fixed-shape product over the pass: lane l takes (l, l+16), then (l, l+8), ...
#pragma unroll
        for (int o = TBK_CHAINW_PASS / 2; o > 0; o >>= 1) {
            const cd other{__shfl_down(d.x, o), __shfl_down(d.y, o)};
            d = cmul(d, other);
        }
        acc = cmul(acc, d);                          // lane 0 holds the pass product
        lds_sync_wave();
    }
    if (lane == 0 && !dets_out) A.partial[seg * A.nstrings + s] = acc;
}

#include "tbk_berry_prod.inl"   // 5..8 bands, berry_phase: the ordered product of a string's link matrices on the matrix cores, no workspace

// 5..8 bands of wide states: link matrices of a batch of strings -> workspace (nocc^2 c128 per link, at most ~1 GiB per
// batch), then their determinants -- multiplied per (string, segment) into A.partial (dets_out == null: berry_phase) or
// filed one by one under the link's first point (berry_flux).  A.seg_len / A.nseg must be set.
static int launch_chain_wave(tbk_ctx* ctx, const WfsView& v, const ChainArgs& A, int nocc, cd* dets_out) {
    const int nn = nocc * nocc;
    // berry_phase of 5..8 bands (no per-link determinants wanted): the product form -- one nocc x nocc matrix per (string, segment)
    // instead of one per link, no link-matrix workspace (TBK_CHAIN_PROD=0: the two kernels below)
    {
        const size_t lds_pts = (size_t)(TBK_CHAINP_G + 1) * (nocc * (v.ncomp + 1) + 1) * sizeof(cd), lds_img = (size_t)TBK_CHAINP_G * 256 * sizeof(double);
        const size_t lds_p1 = nocc == 8 ? std::max(lds_pts, lds_img) : lds_pts + lds_img;   // (8 bands: the images lie over the points)
        if (!dets_out && nocc >= 5 && nocc <= 8 && tbk_knobs().chain_prod != 0 && 2 * lds_p1 <= 64 * 1024) {
            const int64_t nw = A.nstrings * A.nseg;
            const size_t wbytes = (size_t)nw * 64 * sizeof(cd);
            if (wbytes > ctx->work_bytes) {
                TBK_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->work) TBK_HIP(hipFree(ctx->work));
                ctx->work = nullptr;
                ctx->work_bytes = 0;
                hipError_t e = hipMalloc(&ctx->work, wbytes);
                TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "string-product workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
                ctx->work_bytes = wbytes;
            }
            cd* pw = (cd*)ctx->work;
            const int nld = (nocc * v.ncomp + 63) / 64;
            const dim3 gp((unsigned)((nw + 1) / 2)), gd((unsigned)((nw + 63) / 64));
#define TBK_CHAINP(NN, LL)                                                                                                       \
    {                                                                                                                            \
        { ProfScope p1(ctx, "chain_prod"); hipLaunchKernelGGL((k_chain_prod_tile<NN, LL>), gp, dim3(128), 2 * lds_p1, ctx->stream, A, (int64_t)0, A.nstrings, pw); } \
        { ProfScope p2(ctx, "chain_prod_det"); hipLaunchKernelGGL((k_chain_prod_det<NN>), gd, dim3(64), 0, ctx->stream, A, (int64_t)0, A.nstrings, (const cd*)pw); } \
    }
#define TBK_CHAINP_N(NN)                                  \
    switch (nld) {                                        \
        case 1: TBK_CHAINP(NN, 1) break;                  \
        case 2: TBK_CHAINP(NN, 2) break;                  \
        case 3: TBK_CHAINP(NN, 3) break;                  \
        default: TBK_CHAINP(NN, 4) break;                 \
    }
            switch (nocc) {
                case 5: TBK_CHAINP_N(5) break;
                case 6: TBK_CHAINP_N(6) break;
                case 7: TBK_CHAINP_N(7) break;
                default: TBK_CHAINP_N(8) break;
            }
#undef TBK_CHAINP_N
#undef TBK_CHAINP
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
    }
    const size_t per_string = (size_t)A.nlinks * nn * sizeof(cd);
    const size_t ws_cap = (size_t)std::max(1, tbk_knobs().chain_ws_mb) << 20;
    const int64_t nsb = std::max<int64_t>(1, std::min<int64_t>(A.nstrings, (int64_t)(ws_cap / per_string)));
    const size_t wbytes = (size_t)nsb * per_string;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "link-matrix workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    cd* ws = (cd*)ctx->work;
    const size_t lds_a = (size_t)4 * 2 * (nocc * (v.ncomp + 1) + 1) * sizeof(cd);
    const size_t lds_b = (size_t)TBK_CHAINW_PASS * (nn + 1) * sizeof(cd);
    // 5..8 bands: four links per wavefront step (k_chain_links_tile) while its five staged points fit 64 KB per block
    const size_t lds_t1 = (size_t)(TBK_CHAINT_G + 1) * (nocc * (v.ncomp + 1) + 1) * sizeof(cd);   // per wavefront
    const int wpb_t = 4 * lds_t1 <= 64 * 1024 ? 4 : 2;
    const size_t lds_t = wpb_t * lds_t1;
    const bool use_tile = nocc >= 5 && tbk_knobs().chain_tile != 0 && lds_t <= 64 * 1024;
    const int nld = (nocc * v.ncomp + 63) / 64;
    const dim3 blk(256);
    for (int64_t s0 = 0; s0 < A.nstrings; s0 += nsb) {
        const int64_t ns = std::min<int64_t>(nsb, A.nstrings - s0);
        const int64_t nw = ns * A.nseg;
        const dim3 ga((unsigned)((nw + 3) / 4)), gb((unsigned)nw);
#define TBK_CHAINW(NN, LL)                                                                                                          \
    {                                                                                                                               \
        {                                                                                                                           \
            ProfScope p1(ctx, "chain_links");                                                                                       \
            if constexpr (NN >= 5) {                                                                                                \
                if (use_tile) hipLaunchKernelGGL((k_chain_links_tile<NN, LL>), dim3((unsigned)((nw + wpb_t - 1) / wpb_t)), dim3(64 * wpb_t), lds_t, ctx->stream, A, s0, ns, ws); \
                else hipLaunchKernelGGL((k_chain_links_wave<NN, LL>), ga, blk, lds_a, ctx->stream, A, s0, ns, ws);                  \
            } else {                                                                                                                \
                hipLaunchKernelGGL((k_chain_links_wave<NN, LL>), ga, blk, lds_a, ctx->stream, A, s0, ns, ws);                       \
            }                                                                                                                       \
        }                                                                                                                           \
        { ProfScope p2(ctx, "chain_lu"); hipLaunchKernelGGL((k_chain_lu_wave<NN>), gb, dim3(64), lds_b, ctx->stream, A, s0, ns, (const cd*)ws, dets_out); } \
    }
#define TBK_CHAINW_N(NN)                                  \
    switch (nld) {                                        \
        case 1: TBK_CHAINW(NN, 1) break;                  \
        case 2: TBK_CHAINW(NN, 2) break;                  \
        case 3: TBK_CHAINW(NN, 3) break;                  \
        default: TBK_CHAINW(NN, 4) break;                 \
    }
        switch (nocc) {
            case 1: TBK_CHAINW_N(1) break;
            case 2: TBK_CHAINW_N(2) break;
            case 3: TBK_CHAINW_N(3) break;
            case 4: TBK_CHAINW_N(4) break;
            case 5: TBK_CHAINW_N(5) break;
            case 6: TBK_CHAINW_N(6) break;
            case 7: TBK_CHAINW_N(7) break;
            default: TBK_CHAINW_N(8) break;
        }
#undef TBK_CHAINW_N
#undef TBK_CHAINW
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}

static bool chain_wave_applies(const WfsView& v, int nocc) {
    const int from = tbk_knobs().chain_wave_from >= 1 ? tbk_knobs().chain_wave_from : 1;   // (measured faster from one band on, profiles/chain_wave_small_probe.py)
    return nocc >= from && nocc <= 8 && v.ncomp >= 8 && nocc * v.ncomp <= 256 && tbk_knobs().chain_wave != 0;
}

// segments of the strings along `dir`: enough wavefronts to fill the chip, no shorter than one pass of the LU kernel
static void chain_wave_segments(tbk_ctx* ctx, ChainArgs& A) {
    const int64_t target = (int64_t)ctx->cus * 64;
    const int64_t nseg = std::max<int64_t>(1, std::min<int64_t>((A.nlinks + TBK_CHAINW_PASS - 1) / TBK_CHAINW_PASS,
                                                                 target / std::max<int64_t>(A.nstrings, 1)));
    A.seg_len = (int)((A.nlinks + nseg - 1) / nseg);
    A.nseg = (A.nlinks + A.seg_len - 1) / A.seg_len;
}

// berry_flux of 5..8 wide bands: determinants of all links along `dir`, dets_out[point] (see tbk_berry_flux_async)
static int chain_wave_link_dets(tbk_wfs* w, const int32_t* occ, int nocc, int dir, cd* dets_out) {
    const WfsView& v = w->view;
    ChainArgs A{};
    A.v = v;
    A.nocc = nocc;
    for (int i = 0; i < nocc; ++i) A.occ[i] = occ[i];
    A.nlinks = v.mesh[dir] - 1;
    A.sdir = v.stride[dir];
    other_axes(v, dir, -1, &A.other, &A.nstrings);
    chain_wave_segments(w->ctx, A);
    return launch_chain_wave(w->ctx, v, A, nocc, dets_out);
}

// ordered product over one segment of one string
template <int NOCC, int MAXN, bool EVALS>
__global__ __launch_bounds__(256) void k_chain_partial(const ChainArgs A) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= A.nstrings * A.nseg) return;
    const int64_t seg = t / A.nstrings, s = t - seg * A.nstrings;
    const int nocc = A.nocc, ncomp = A.v.ncomp;
    const int64_t plane = A.v.npts * ncomp;       // band-major: band b of a point is b*plane further
    const int i0 = (int)seg * A.seg_len;
    const int i1 = min(i0 + A.seg_len, A.nlinks);
    const cd* P = A.v.data + (axis_offset(A.other, s) + (int64_t)i0 * A.sdir) * ncomp;
    const int64_t step = A.sdir * ncomp;
    if constexpr (!EVALS) {
        cd acc{1.0, 0.0};
        if constexpr (NOCC == 1 || NOCC == 2) {
            // states of at most four components (the one band of a 2-band model, the two spinor bands of Kane-Mele): the right end
            // of a link is the left end of the next one -- kept in registers, every point is loaded once (the eigenphase branch
            // below has done so since round 2; the determinant form read every point twice: berry_phase([0, 1], 2) of a 129^3
            // array of 4 components 0.225 ns per link against 0.088 for the Wilson loop, profiles/berry_dirs_probe.py)
            if (ncomp <= 4) {
                cd prev[NOCC][4], cur[NOCC][4];
                auto load_pt = [&](const cd* pt, cd (&u)[NOCC][4]) {
#pragma unroll
                    for (int a = 0; a < NOCC; ++a)
#pragma unroll
                        for (int o = 0; o < 4; ++o) u[a][o] = o < ncomp ? pt[A.occ[a] * plane + o] : cd{0.0, 0.0};
                };
                load_pt(P, prev);
                for (int i = i0; i < i1; ++i, P += step) {
                    load_pt(P + step, cur);
                    cd M[NOCC][NOCC];
#pragma unroll
                    for (int a = 0; a < NOCC; ++a)
#pragma unroll
                        for (int b = 0; b < NOCC; ++b) {
                            cd m = cmulc(prev[a][0], cur[b][0]);
#pragma unroll
                            for (int o = 1; o < 4; ++o) cfmac(m, prev[a][o], cur[b][o]);
                            M[a][b] = m;
                        }
                    acc = cmul(acc, det_small<NOCC>(M));
#pragma unroll
                    for (int a = 0; a < NOCC; ++a)
#pragma unroll
                        for (int o = 0; o < 4; ++o) prev[a][o] = cur[a][o];
                }
                A.partial[seg * A.nstrings + s] = acc;
                return;
            }
        }
        for (int i = i0; i < i1; ++i, P += step)
            acc = cmul(acc, one_link_det<NOCC, MAXN>(P, P + step, A.occ, nocc, ncomp, plane));
        A.partial[seg * A.nstrings + s] = acc;
    } else if constexpr (NOCC == 1) {
        cd acc{1.0, 0.0};
        for (int i = i0; i < i1; ++i, P += step) {
            cd M[1][1];
            link_matrix<1>(P, P + step, A.occ, ncomp, plane, M);
            const double r = sqrt(cabs2(M[0][0]));
            acc = cmul(acc, r > 0.0 ? cscale(M[0][0], 1.0 / r) : cd{1.0, 0.0});
        }
        A.partial[seg * A.nstrings + s] = acc;
    } else if constexpr (NOCC == 2) {
        cd R[2][2] = {{cd{1.0, 0.0}, cd{0.0, 0.0}}, {cd{0.0, 0.0}, cd{1.0, 0.0}}};
        // states of at most four components (BASELINE configs[3]: two spinor bands of Kane-Mele): the point at the right end of
        // a link is the left end of the next one -- keep it in registers instead of loading every point twice (the PMC run
        // of round 1 showed 2x the algorithmic bytes fetched, profiles/r02a)
        const bool narrow = ncomp <= 4;
        cd prev[2][4], cur[2][4];
        auto load_pt = [&](const cd* pt, cd (&u)[2][4]) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int o = 0; o < 4; ++o) u[a][o] = o < ncomp ? pt[A.occ[a] * plane + o] : cd{0.0, 0.0};
        };
        if (narrow) load_pt(P, prev);
        for (int i = i0; i < i1; ++i, P += step) {
            cd M[2][2];
            if (narrow) {
                load_pt(P + step, cur);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        cd m = cmulc(prev[a][0], cur[b][0]);
#pragma unroll
                        for (int o = 1; o < 4; ++o) cfmac(m, prev[a][o], cur[b][o]);
                        M[a][b] = m;
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int o = 0; o < 4; ++o) prev[a][o] = cur[a][o];
            } else {
                link_matrix<2>(P, P + step, A.occ, ncomp, plane, M);
            }
            polar2(M);
            cd T[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) T[a][b] = cadd(cmul(R[a][0], M[0][b]), cmul(R[a][1], M[1][b]));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) R[a][b] = T[a][b];
        }
        cd* o = A.partial + (seg * A.nstrings + s) * 4;
        o[0] = R[0][0]; o[1] = R[0][1]; o[2] = R[1][0]; o[3] = R[1][1];
    } else {
        cd R[MAXN * MAXN], M[MAXN * MAXN], V[MAXN * MAXN], T[MAXN * MAXN];
        for (int a = 0; a < nocc; ++a)
            for (int b = 0; b < nocc; ++b) R[a * nocc + b] = cd{a == b ? 1.0 : 0.0, 0.0};
        for (int i = i0; i < i1; ++i, P += step) {
            link_matrix_dyn(P, P + step, A.occ, nocc, ncomp, plane, M);
            polar_dyn(nocc, M, V, T);
            for (int a = 0; a < nocc; ++a)
                for (int b = 0; b < nocc; ++b) {
                    cd acc{0.0, 0.0};
                    for (int j = 0; j < nocc; ++j) cfma(acc, R[a * nocc + j], M[j * nocc + b]);
                    T[a * nocc + b] = acc;
                }
            for (int e = 0; e < nocc * nocc; ++e) R[e] = T[e];
        }
        cd* o = A.partial + (seg * A.nstrings + s) * (int64_t)(nocc * nocc);
        for (int e = 0; e < nocc * nocc; ++e) o[e] = R[e];
    }
}

// -angle of the two eigenvalues of a (numerically) unitary 2 x 2 matrix U = [[u00, u01], [u10, u11]] (pythtb.py:3834-3838 takes
// them from numpy.linalg.eigvals).  U = h V with h^2 = det U / |det U| and V in SU(2), V = [[a, b], [-conj b, conj a]]: the
// eigenvalues are h (c +- i s), c = Re a, s = sqrt(Im(a)^2 + |b|^2) -- no cancellation at the Kramers degeneracies of a
// time-reversal symmetric model, where lambda = (tr +- sqrt(tr^2 - 4 det)) / 2 would lose half of the digits.
__device__ __forceinline__ void unit_eigenphases2(const cd u00, const cd u01, const cd u10, const cd u11, double& p0, double& p1) {
    const cd d = det2(u00, u01, u10, u11);
    const double dn = sqrt(cabs2(d));
    cd dh = dn > 0.0 ? cd{d.x / dn, d.y / dn} : cd{1.0, 0.0};
    // principal square root of the unit number dh
    cd h;
    {
        const double re = sqrt(0.5 * (1.0 + fabs(dh.x)));
        const double im = 0.5 * dh.y / re;
        h = dh.x >= 0.0 ? cd{re, im} : cd{fabs(im), copysign(re, dh.y)};
    }
    const cd hc = cconj(h);
    const cd v00 = cmul(u00, hc), v11 = cmul(u11, hc), v01 = cmul(u01, hc), v10 = cmul(u10, hc);
    const cd a{0.5 * (v00.x + v11.x), 0.5 * (v00.y - v11.y)};          // (a + conj a') / 2
    const cd b{0.5 * (v01.x - v10.x), 0.5 * (v01.y + v10.y)};          // (b - conj(-conj b')) / 2 = (v01 - conj v10) / 2
    const double c = a.x, sn = sqrt(a.y * a.y + cabs2(b));
    const cd l0 = cmul(h, cd{c, sn}), l1 = cmul(h, cd{c, -sn});
    p0 = -atan2(l0.y, l0.x);
    p1 = -atan2(l1.y, l1.x);
}

// combine the segments of each string in order and finish
template <int MAXN, bool EVALS>
__global__ __launch_bounds__(64) void k_chain_final(const ChainArgs A) {
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= A.nstrings) return;
    const int nocc = A.nocc;
    if constexpr (!EVALS) {
        cd acc{1.0, 0.0};
        for (int g = 0; g < A.nseg; ++g) acc = cmul(acc, A.partial[(int64_t)g * A.nstrings + s]);
        A.out[s] = -atan2(acc.y, acc.x);   // -angle(det(prd))   (pythtb.py:3829-3831)
    } else {
        if (nocc == 1) {
            cd acc{1.0, 0.0};
            for (int g = 0; g < A.nseg; ++g) acc = cmul(acc, A.partial[(int64_t)g * A.nstrings + s]);
            A.out[s] = -atan2(acc.y, acc.x);
            return;
        }
        if (nocc == 2) {       // two bands: registers and the closed form (the general path below indexes local arrays dynamically)
            cd r0 = A.partial[s * 4], r1 = A.partial[s * 4 + 1], r2 = A.partial[s * 4 + 2], r3 = A.partial[s * 4 + 3];
            for (int g = 1; g < A.nseg; ++g) {
                const cd* M = A.partial + ((int64_t)g * A.nstrings + s) * 4;
                const cd m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3];
                cd t0{0.0, 0.0}, t1{0.0, 0.0}, t2{0.0, 0.0}, t3{0.0, 0.0};
                cfma(t0, r0, m0); cfma(t0, r1, m2);
                cfma(t1, r0, m1); cfma(t1, r1, m3);
                cfma(t2, r2, m0); cfma(t2, r3, m2);
                cfma(t3, r2, m1); cfma(t3, r3, m3);
                r0 = t0; r1 = t1; r2 = t2; r3 = t3;
            }
            double p0, p1;
            unit_eigenphases2(r0, r1, r2, r3, p0, p1);
            A.out[s * 2] = fmin(p0, p1);
            A.out[s * 2 + 1] = fmax(p0, p1);
            return;
        }
        cd R[MAXN * MAXN], T[MAXN * MAXN], ev[MAXN], rc[MAXN], rs[MAXN];
        const int nn = nocc * nocc;
        for (int e = 0; e < nn; ++e) R[e] = A.partial[s * nn + e];
        for (int g = 1; g < A.nseg; ++g) {
            const cd* M = A.partial + ((int64_t)g * A.nstrings + s) * nn;
            for (int a = 0; a < nocc; ++a)
                for (int b = 0; b < nocc; ++b) {
                    cd acc{0.0, 0.0};
                    for (int j = 0; j < nocc; ++j) cfma(acc, R[a * nocc + j], M[j * nocc + b]);
                    T[a * nocc + b] = acc;
                }
            for (int e = 0; e < nn; ++e) R[e] = T[e];
        }
        if (!eigvals_dyn(nocc, R, ev, rc, rs)) atomicExch(A.flags + 1, 1);
        // sort(-angle(eigvals))   (pythtb.py:3834-3838)
        double* o = A.out + s * nocc;
        for (int j = 0; j < nocc; ++j) {
            const double ph = -atan2(ev[j].y, ev[j].x);
            int pos = j;
            while (pos > 0 && o[pos - 1] > ph) {
                o[pos] = o[pos - 1];
                --pos;
            }
            o[pos] = ph;
        }
    }
}

// Same finish for long strings (many segments): one wavefront per string.  Lane l
// multiplies its contiguous run of segments in order, then an ordered pairwise
// tree over the lanes through LDS (associativity keeps the order), lane 0 finishes.
// NC > 0: the band count as a compile-time constant -- every loop over bands unrolls and R, T, M live in registers (with the
// count read from A.nocc they are indexed dynamically, i.e. scratch memory: 52 us per launch for two bands, whatever the number
// of strings, all of it latency).  NC = 2 also takes its eigenphases in closed form (unit_eigenphases2).
template <int MAXN, bool EVALS, int NC = 0>
__global__ __launch_bounds__(64) void k_chain_final_wave(const ChainArgs A) {
    extern __shared__ __align__(16) unsigned char lds_chain[];
    cd* ex = reinterpret_cast<cd*>(lds_chain);
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int nocc = NC > 0 ? NC : A.nocc;
    const int nn = EVALS ? nocc * nocc : 1;
    const int P = A.final_lanes;                           // participating lanes (power of two <= 64)
    const int run = (A.nseg + P - 1) / P;
    const int g0 = lane * run, g1 = min(g0 + run, A.nseg);
    cd R[EVALS ? MAXN * MAXN : 1], T[EVALS ? MAXN * MAXN : 1];
    auto load_seg = [&](int g, cd* dst) {
        const cd* M = A.partial + ((int64_t)g * A.nstrings + s) * nn;
#pragma unroll
        for (int e = 0; e < nn; ++e) dst[e] = M[e];
    };
    auto mul_into_R = [&](const cd* M) {                    // R <- R * M
        if constexpr (!EVALS) {
            R[0] = cmul(R[0], M[0]);
        } else {
#pragma unroll
            for (int a = 0; a < nocc; ++a)
#pragma unroll
                for (int b = 0; b < nocc; ++b) {
                    cd acc{0.0, 0.0};
#pragma unroll
                    for (int j = 0; j < nocc; ++j) cfma(acc, R[a * nocc + j], M[j * nocc + b]);
                    T[a * nocc + b] = acc;
                }
#pragma unroll
            for (int e = 0; e < nn; ++e) R[e] = T[e];
        }
    };
    const bool has = lane < P && g0 < g1;
    if (has) {
        load_seg(g0, R);
        for (int g = g0 + 1; g < g1; ++g) {
            cd M[EVALS ? MAXN * MAXN : 1];
            load_seg(g, M);
            mul_into_R(M);
        }
    } else {                                               // identity: neutral in the ordered product
#pragma unroll
        for (int e = 0; e < nn; ++e) R[e] = cd{0.0, 0.0};
        if constexpr (!EVALS) R[0] = cd{1.0, 0.0};
        else {
#pragma unroll
            for (int a = 0; a < nocc; ++a) R[a * nocc + a] = cd{1.0, 0.0};
        }
    }
    if (lane < P) {
#pragma unroll
        for (int e = 0; e < nn; ++e) ex[lane * nn + e] = R[e];
    }
    __syncthreads();
    for (int off = 1; off < P; off <<= 1) {
        const bool act = lane < P && (lane % (2 * off)) == 0 && lane + off < P;
        if (act) mul_into_R(ex + (lane + off) * nn);
        __syncthreads();
        if (act) {
#pragma unroll
            for (int e = 0; e < nn; ++e) ex[lane * nn + e] = R[e];
        }
        __syncthreads();
    }
    if (lane != 0) return;
    if constexpr (!EVALS) {
        A.out[s] = -atan2(R[0].y, R[0].x);
    } else {
        if (nocc == 1) {
            A.out[s] = -atan2(R[0].y, R[0].x);
            return;
        }
        cd ev[MAXN], rc[MAXN], rs[MAXN];
        if (!eigvals_dyn(nocc, R, ev, rc, rs)) atomicExch(A.flags + 1, 1);
        double* o = A.out + s * nocc;
        for (int j = 0; j < nocc; ++j) {
            const double ph = -atan2(ev[j].y, ev[j].x);
            int pos = j;
            while (pos > 0 && o[pos - 1] > ph) {
                o[pos] = o[pos - 1];
                --pos;
            }
            o[pos] = ph;
        }
    }
}

// ---- the LDS-tile kernels (tbk_berry_lanes.inl) for determinants: which arrays they take, how a call is cut into tiles
struct LanesPlan {
    bool ok, l;          // applies; L form (lane = link) instead of S (lane = string)
    size_t lds;
    int seg_len, nseg;   // S: links per lane segment, segments per string; L: 64, tiles per string
    int64_t ntile;       // S: tiles of 64 strings
};
static bool lanes_dets_applies(const WfsView& v, int nocc, const size_t lds_max) {
    // 1..7 bands (a lane holds the link matrix in registers); not the up-to-two bands of up-to-four components the register kernels
    // serve; the tile must fit 64 KB of LDS.  Wide states too (round 6, profiles/berry_cliff_sweep.py: the wave-per-string kernels
    // have a floor of ~100 us per 132 k points whatever the band count -- 1 band of 8 components 99 us against 15 us for 7).
    return nocc >= 1 && nocc <= 7 && !(nocc <= 2 && v.ncomp <= 4) && tbk_knobs().wilson_reg == 3 && v.npts < (int64_t)0x7fffffff &&
           v.ncomp <= 21 && (size_t)nocc * ((65 * v.ncomp + 63) & ~63) * sizeof(cd) <= lds_max;
}
static LanesPlan lanes_plan(tbk_ctx* ctx, const WfsView& v, int nocc, int64_t sdir, int L, int64_t nstrings) {
    LanesPlan P{};
    const size_t lds_l = (size_t)nocc * ((65 * v.ncomp + 63) & ~63) * sizeof(cd), lds_s = (size_t)2 * nocc * 64 * v.ncomp * sizeof(cd);
    P.l = (sdir == 1 && L >= 32) || (nstrings < 32 && L >= 32) || (lds_s > 40 * 1024 && lds_l <= 40 * 1024 && L >= 32) || lds_s > 64 * 1024;
    if (tbk_knobs().wilson_form >= 0 && !(lds_s > 64 * 1024)) P.l = tbk_knobs().wilson_form != 0;
    P.lds = P.l ? lds_l : lds_s;
    P.ok = P.lds <= (P.l ? TBK_LANES_DET_LDS_MAX : (size_t)64 * 1024);
    P.ntile = 1;
    if (P.l) {
        P.seg_len = 64;
        P.nseg = (L + 63) / 64;
    } else {
        const int res = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(P.lds, 1)));
        P.ntile = (nstrings + 63) / 64;
        const int64_t R = (int64_t)ctx->cus * res;
        int64_t ns_ = P.ntile * 2 <= R ? R / P.ntile : (4 * R + P.ntile - 1) / P.ntile;
        ns_ = std::max<int64_t>(1, std::min<int64_t>(ns_, std::max(1, L / 2)));
        P.seg_len = (int)((L + ns_ - 1) / ns_);
        if (tbk_knobs().wilson_seg > 0) P.seg_len = std::min(L, tbk_knobs().wilson_seg);
        P.nseg = (L + P.seg_len - 1) / P.seg_len;
    }
    return P;
}
static void lanes_fill(WilsonLanesArgs& S, tbk_ctx* ctx, const WfsView& v, const int* occ, int nocc, int L, int64_t sdir, const AxisSet& other,
                       int64_t nstrings, const LanesPlan& P) {
    S = WilsonLanesArgs{};
    S.W.v = v;
    S.W.occ = nullptr;
    S.W.nocc = nocc;
    S.W.nlinks = L;
    S.W.sdir = sdir;
    S.W.other = other;
    S.W.s0 = 0;
    S.W.ns = nstrings;
    S.W.flags = ctx->flags_dev;
    for (int a = 0; a < 8; ++a) S.occ_inl[a] = a < nocc ? occ[a] : 0;
    S.seg_len = P.seg_len;
    S.nseg = P.nseg;
    S.ntile = P.ntile;
    S.magic = (unsigned)((65536 + v.ncomp - 1) / v.ncomp);
    S.swz = v.ncomp == 4 ? 2 : v.ncomp == 8 ? 1 : v.ncomp == 16 ? 0 : -1;
    if (tbk_knobs().wilson_swz == 0) S.swz = -1;
}
#define TBK_LANES_BIG(M_, OUT_, P_) if ((P_).lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_wilson_lanes_l<M_, OUT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define TBK_LANES_LAUNCH(OUT_, S_, P_, nstr_)                                                                                \
    {                                                                                                                        \
        const dim3 g_((unsigned)((P_).l ? (nstr_) * (P_).nseg : (P_).ntile * (P_).nseg)), b_(64);                            \
        switch ((S_).W.nocc) {                                                                                               \
            case 1: if ((P_).l) { TBK_LANES_BIG(1, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<1, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<1, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            case 2: if ((P_).l) { TBK_LANES_BIG(2, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<2, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<2, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            case 3: if ((P_).l) { TBK_LANES_BIG(3, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<3, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<3, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            case 4: if ((P_).l) { TBK_LANES_BIG(4, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<4, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<4, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            case 5: if ((P_).l) { TBK_LANES_BIG(5, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<5, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<5, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            case 6: if ((P_).l) { TBK_LANES_BIG(6, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<6, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                    else hipLaunchKernelGGL((k_wilson_lanes_s<6, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;         \
            default: if ((P_).l) { TBK_LANES_BIG(7, OUT_, P_) hipLaunchKernelGGL((k_wilson_lanes_l<7, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); } \
                     else hipLaunchKernelGGL((k_wilson_lanes_s<7, OUT_>), g_, b_, (P_).lds, ctx->stream, S_); break;        \
        }                                                                                                                    \
    }
// dets_out[p] = det of the link matrix from mesh point p to its neighbour along `dir`, for every string along that axis
static int lanes_link_dets(tbk_wfs* w, const int32_t* occ, int nocc, int dir, cd* dets_out) {
    tbk_ctx* ctx = w->ctx;
    const WfsView& v = w->view;
    AxisSet other{};
    int64_t nstrings = 1;
    other_axes(v, dir, -1, &other, &nstrings);
    const int L = v.mesh[dir] - 1;
    const LanesPlan P = lanes_plan(ctx, v, nocc, v.stride[dir], L, nstrings);
    TBK_REQUIRE(P.ok, TBK_EUNSUPPORTED, "lanes_link_dets: %d bands of %d components do not fit the LDS tile", nocc, v.ncomp);
    int occ4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < nocc; ++a) occ4[a] = occ[a];
    WilsonLanesArgs S;
    lanes_fill(S, ctx, v, occ4, nocc, L, v.stride[dir], other, nstrings, P);
    S.dets = dets_out;
    ProfScope ps(ctx, "lanes_link_dets");
    TBK_LANES_LAUNCH(2, S, P, nstrings)
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

extern "C" int tbk_berry_phase(tbk_wfs* w, const int32_t* occ, int nocc, int dir, int berry_evals, double* out) {
    TBK_REQUIRE(w && out, TBK_EINVAL, "tbk_berry_phase: null argument");
    const WfsView& v = w->view;
    TBK_REQUIRE(dir >= 0 && dir < v.dim_arr, TBK_EINVAL, "Wrong direction for Berry phase calculation!");
    ChainArgs A{};
    int det_from = 9;      // up to 8 bands the register LU per thread wins; from 9 the workgroup-per-link LU is 4-100x faster (profiles/det_big_probe.py)
    if (tbk_knobs().det_big_from >= 0) det_from = std::max(2, tbk_knobs().det_big_from);
    const bool big = nocc >= det_from && !berry_evals;      // det of the string = product of link dets (LU per link)
    // Wilson-loop eigenphases: closed forms up to two bands; from three on the workgroup-level pipeline, which
    // measured 3x (3 bands) to 160x (16 bands) faster than the per-thread polar/QR kernels at every string count
    // and length tried (profiles/wilson_small_probe.py)
    int ev_from = 3;
    if (tbk_knobs().wilson_big_from >= 0) ev_from = std::max(2, tbk_knobs().wilson_big_from);
    const bool big_ev = nocc >= ev_from && berry_evals;  // workgroup-level polar factors, product tree, Cayley + eigh
    int rc = (big || big_ev) ? check_occ(w, occ, nocc) : fill_occ(w, occ, nocc, A.occ);
    if (rc) return rc;
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    A.v = v;
    A.nocc = nocc;
    A.nlinks = v.mesh[dir] - 1;
    A.sdir = v.stride[dir];
    other_axes(v, dir, -1, &A.other, &A.nstrings);
    if (big) {
        int* occ_dev = nullptr;
        cd* dets = nullptr;
        void* work = nullptr;
        void* out_dev = nullptr;
        size_t work_bytes = 0;
        rc = big_scratch(w, occ, nocc, 1, (size_t)A.nstrings * sizeof(double), &occ_dev, &dets, &work, &work_bytes, &out_dev);
        if (rc) return rc;
        rc = launch_link_dets(w, occ_dev, nocc, dir, dets, work, work_bytes);
        if (rc) return rc;
        StringDetArgs S{};
        S.dets = dets;
        S.nlinks = A.nlinks;
        S.sdir = A.sdir;
        S.other = A.other;
        S.nstrings = A.nstrings;
        S.out = (double*)out_dev;
        {
            ProfScope ps(ctx, "string_from_dets");
            hipLaunchKernelGGL(k_string_from_dets, dim3((unsigned)((A.nstrings + 255) / 256)), dim3(256), 0, ctx->stream, S);
            TBK_HIP(hipGetLastError());
        }
        return tbk_small_result(ctx, out, out_dev, (size_t)A.nstrings * sizeof(double), nullptr);
    }
    if (big_ev) {
        // workgroup-level pipeline (tbk_berry_big.inl): link polar factors, pairwise product tree, Cayley
        // transform + Hermitian eigen-solve.  Strings go in batches whose two ping-pong link arrays fit 1 GiB.
        const size_t nn = (size_t)nocc * nocc;
        const int L = A.nlinks;
        // 5..8 wide bands: link matrices, their polar factors and the ordered product of a (string, segment) in ONE wavefront
        // kernel on the matrix cores (k_chain_prod_tile<.., POLAR>, tbk_berry_prod.inl); the tree then runs over the segments.
        // (TBK_WILSON_MFMA=0: the workgroup-per-link kernels)
        bool mfma_route = false;
        size_t lds_p1 = 0;
        // (3 and 4 bands of WIDE states too: a thread per segment walks 256-byte rows of its own -- TBK_WILSON_MFMA=2 keeps them on k_wilson_seg_reg)
        const int mfma_from = tbk_knobs().wilson_mfma == 2 ? 5 : 3;
        if (nocc >= mfma_from && nocc <= 8 && tbk_knobs().wilson_mfma != 0 && chain_wave_applies(v, nocc)) {
            const size_t lds_pts = (size_t)(TBK_CHAINP_G + 1) * (nocc * (v.ncomp + 1) + 1) * sizeof(cd);
            const size_t lds_img = (size_t)TBK_CHAINP_G * 16 * TBK_CHAINP_LD(true) * sizeof(double);
            lds_p1 = nocc == 8 ? std::max(lds_pts, lds_img) : lds_pts + lds_img;
            if (2 * lds_p1 <= 64 * 1024 && fill_occ(w, occ, nocc, A.occ) == TBK_OK) {
                mfma_route = true;
                chain_wave_segments(ctx, A);
                A.flags = ctx->flags_dev;
            }
        }
        // 3 or 4 bands whose occupied vectors fit the LDS tile: a lane per string or per link, vectors through LDS (tbk_berry_lanes.inl;
        // TBK_WILSON_REG=3, the default; 1: round 4's thread per segment, 0: the workgroup kernels)
        const int wreg = tbk_knobs().wilson_reg;
        bool lanes_route = false, lanes_l = false;
        size_t lanes_lds = 0;
        int lanes_res = 1;                                // wavefronts of the S form resident per CU (LDS-bound)
        // (5 and 6 bands too, round 6: 256 VGPRs + accumulation registers, one wavefront per SIMD -- still 3-20 x the workgroup
        // pipeline on narrow states and 2-3 x the matrix-core tile kernel on wide ones, profiles/berry_cliff_sweep.py)
        if (nocc >= 3 && nocc <= 6 && wreg == 3 && v.npts < (int64_t)0x7fffffff && v.ncomp <= 21) {
            // L: the string runs along the fastest axis, or there are too few strings to give every lane one
            const size_t lds_l = (size_t)nocc * ((65 * v.ncomp + 63) & ~63) * sizeof(cd), lds_s = (size_t)2 * nocc * 64 * v.ncomp * sizeof(cd);
            lanes_l = (A.sdir == 1 && L >= 32) || (A.nstrings < 32 && L >= 32);
            // ... or the two row buffers of the S form leave fewer than four wavefronts on a compute unit (a SIMD without one):
            // 4 bands of 8 components, 1025 x 257 along axis 0: S 118 us (2 per CU), L with gathered points 104 (profiles/r06w)
            if (lds_s > 40 * 1024 && lds_l <= 40 * 1024 && L >= 32) lanes_l = true;
            if (lds_s > 64 * 1024 && lds_l <= 64 * 1024) lanes_l = true;                  // (the only form that fits: 3 bands of 16 components)
            if (tbk_knobs().wilson_form >= 0 && lds_s <= 64 * 1024) lanes_l = tbk_knobs().wilson_form != 0;
            lanes_lds = lanes_l ? lds_l : lds_s;
            lanes_route = lanes_lds <= 64 * 1024;
            lanes_res = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lanes_lds, 1)));
        }
        if (lanes_route) mfma_route = false;
        // else 3 or 4 bands: a thread per segment (k_wilson_seg_reg); segments sized on the whole call
        const bool seg_route = !lanes_route && !mfma_route && nocc >= 3 && nocc <= 4 && wreg != 0 && A.nstrings * L < (int64_t)0x7fffffff * 128;
        int seg_len_r = L, nseg_r = 1;
        int64_t lanes_ntile = 1;
        if (lanes_route && lanes_l) {
            seg_len_r = 64;
            nseg_r = ((L + 63) / 64) * (64 / LANES_L_SPAN);        // (a tile leaves the products of its LANES_L_SPAN-link groups)
        } else if (lanes_route) {
            // S: one round of resident wavefronts when the strings allow it (a longer segment pays its first row once), else about four
            const int64_t nsb_guess = A.nstrings;          // (tiles are counted per batch below; the segment length is set once, on the call)
            lanes_ntile = (nsb_guess + 63) / 64;
            const int64_t R = (int64_t)ctx->cus * lanes_res;
            int64_t nseg = lanes_ntile * 2 <= R ? R / lanes_ntile : (4 * R + lanes_ntile - 1) / lanes_ntile;
            nseg = std::max<int64_t>(1, std::min<int64_t>(nseg, std::max(1, L / 2)));
            seg_len_r = (int)((L + nseg - 1) / nseg);
            if (tbk_knobs().wilson_seg > 0) seg_len_r = std::min(L, tbk_knobs().wilson_seg);
            nseg_r = (L + seg_len_r - 1) / seg_len_r;
        }
        if (seg_route) {
            const int64_t want = (int64_t)ctx->cus * 512;
            const int64_t nseg = std::max<int64_t>(1, std::min<int64_t>((L + 3) / 4, (want + A.nstrings - 1) / A.nstrings));
            seg_len_r = (int)((L + nseg - 1) / nseg);
            nseg_r = (L + seg_len_r - 1) / seg_len_r;
        }
        // matrices per string that the buffers hold: every link (the workgroup kernels), or one per segment
        const int Lb = mfma_route ? A.nseg : (seg_route || lanes_route ? nseg_r : L);
        size_t batch_bytes = (size_t)1 << 30;
        if (tbk_knobs().wilson_batch_bytes >= 0) batch_bytes = (size_t)std::max(1ll, tbk_knobs().wilson_batch_bytes);   // test hook
        const int64_t cap = std::max<int64_t>(1, (int64_t)(batch_bytes / (2 * (size_t)Lb * nn * sizeof(cd))));
        const int64_t nsb = std::min<int64_t>(A.nstrings, cap);
        const unsigned nblk = (unsigned)std::min<int64_t>(nsb * L, (int64_t)ctx->cus * 4);
        auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const size_t ob = al((size_t)nocc * sizeof(int));
        const size_t bb = al((size_t)nsb * Lb * nn * sizeof(cd));
        const size_t yb = al((size_t)nblk * nn * sizeof(cd));
        const size_t abb = al((size_t)nsb * 2 * nn * sizeof(cd));
        const size_t hb = al((size_t)nsb * nn * sizeof(cd));
        const size_t eb = al((size_t)nsb * nocc * sizeof(double));
        const size_t mb = al((size_t)nsb * sizeof(double));
        void* base = nullptr;
        rc = tbk_ctx_scratch(ctx, 256 + ob + 2 * bb + yb + abb + hb + 2 * eb + mb, &base);
        if (rc) return rc;
        unsigned char* p = (unsigned char*)base + 256;
        int* occ_dev = (int*)p;
        p += ob;
        cd* buf0 = (cd*)p;
        p += bb;
        cd* buf1 = (cd*)p;
        p += bb;
        cd* ywork = (cd*)p;
        p += yb;
        cd* ab = (cd*)p;
        p += abb;
        cd* herm = (cd*)p;
        p += hb;
        double* ev_dev = (double*)p;
        p += eb;
        double* out_dev = (double*)p;
        p += eb;
        double* hmax_dev = (double*)p;
        TBK_HIP(hipMemcpyAsync(occ_dev, occ, (size_t)nocc * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        std::vector<double> out_h((size_t)nsb * nocc), hmax_h((size_t)nsb), best((size_t)nsb);
        double alphas[4] = {0.7390851332151607, 2.3, 3.9, 5.5};
        if (tbk_knobs().wilson_alpha_set) alphas[0] = tbk_knobs().wilson_alpha;   // test hook: put the pole on an eigenphase
        for (int64_t s0 = 0; s0 < A.nstrings; s0 += nsb) {
            const int64_t ns = std::min<int64_t>(nsb, A.nstrings - s0);
            WilsonBigArgs W{};
            W.v = v;
            W.occ = occ_dev;
            W.nocc = nocc;
            W.nlinks = L;
            W.sdir = A.sdir;
            W.other = A.other;
            W.s0 = s0;
            W.ns = ns;
            W.buf0 = buf0;
            W.buf1 = buf1;
            W.ywork = ywork;
            W.flags = ctx->flags_dev;
            cd *cur = buf0, *nxt = buf1;
            int Lt = L;                            // matrices per string that the tree multiplies
            if (mfma_route) {
                const int64_t nw = ns * A.nseg;
                const int nld = (nocc * v.ncomp + 63) / 64;
                const dim3 gp((unsigned)((nw + 1) / 2));
                ProfScope ps(ctx, "wilson_prod_tile");
#define TBK_WILP(NN, LL) hipLaunchKernelGGL((k_chain_prod_tile<NN, LL, true>), gp, dim3(128), 2 * lds_p1, ctx->stream, A, s0, ns, buf0)
#define TBK_WILP_N(NN)                                    \
    switch (nld) {                                        \
        case 1: TBK_WILP(NN, 1); break;                   \
        case 2: TBK_WILP(NN, 2); break;                   \
        case 3: TBK_WILP(NN, 3); break;                   \
        default: TBK_WILP(NN, 4); break;                  \
    }
                switch (nocc) {
                    case 3: TBK_WILP_N(3) break;
                    case 4: TBK_WILP_N(4) break;
                    case 5: TBK_WILP_N(5) break;
                    case 6: TBK_WILP_N(6) break;
                    case 7: TBK_WILP_N(7) break;
                    default: TBK_WILP_N(8) break;
                }
#undef TBK_WILP_N
#undef TBK_WILP
                TBK_HIP(hipGetLastError());
                Lt = A.nseg;
            } else if (lanes_route) {
                WilsonLanesArgs S{};
                S.W = W;
                S.seg_len = seg_len_r;
                S.nseg = nseg_r;
                S.ntile = (ns + 63) / 64;
                S.magic = (unsigned)((65536 + v.ncomp - 1) / v.ncomp);
                S.swz = v.ncomp == 4 ? 2 : v.ncomp == 8 ? 1 : v.ncomp == 16 ? 0 : -1;
                if (tbk_knobs().wilson_swz == 0) S.swz = -1;
                S.segs = buf1;                     // [ns][nseg][nn]
                S.prod = buf0;                     // string s at buf0 + s nseg nn, where the tree leaves a string's product
                S.pstride = (size_t)Lb * nn;
                S.herm = herm;                     // (the Cayley transform for the first angle comes out of the combine)
                S.ca = cos(alphas[0]);
                S.sa = sin(alphas[0]);
                {
                    ProfScope ps(ctx, lanes_l ? "wilson_lanes_l" : "wilson_lanes_s");
                    const dim3 g((unsigned)(lanes_l ? ns * (S.nseg / (64 / LANES_L_SPAN)) : S.ntile * S.nseg)), b(64);
#define TBK_WL(MM)                                                                                          \
    if (lanes_l) hipLaunchKernelGGL((k_wilson_lanes_l<MM>), g, b, lanes_lds, ctx->stream, S);              \
    else hipLaunchKernelGGL((k_wilson_lanes_s<MM>), g, b, lanes_lds, ctx->stream, S);
                    switch (nocc) {
                        case 3: TBK_WL(3) break;
                        case 4: TBK_WL(4) break;
                        case 5: TBK_WL(5) break;
                        default: TBK_WL(6) break;
                    }
#undef TBK_WL
                    TBK_HIP(hipGetLastError());
                }
                {
                    ProfScope ps(ctx, "wilson_lanes_combine");
                    switch (nocc) {
                        case 3: hipLaunchKernelGGL((k_wilson_lanes_combine<3>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, S); break;
                        case 4: hipLaunchKernelGGL((k_wilson_lanes_combine<4>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, S); break;
                        case 5: hipLaunchKernelGGL((k_wilson_lanes_combine<5>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, S); break;
                        default: hipLaunchKernelGGL((k_wilson_lanes_combine<6>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, S); break;
                    }
                    TBK_HIP(hipGetLastError());
                }
            } else
            if (nocc >= 3 && nocc <= 4 && wreg != 0 && !mfma_route && A.nstrings * L < (int64_t)0x7fffffff * 128) {
                // 3 or 4 bands in registers (tbk_berry_big.inl): a thread per SEGMENT of a string forms its links, their polar
                // factors and their ordered product; the segments of a string are multiplied by k_wilson_lanes_combine.  (TBK_WILSON_REG=0:
                // the workgroup kernels.)
                {
                    WilsonSegArgs S{};
                    S.W = W;
                    S.seg_len = seg_len_r;         // (segments: enough threads to fill the chip, none shorter than 4 links)
                    S.nseg = nseg_r;
                    S.segs = buf1;                 // [ns][nseg][nn]
                    S.prod = buf0;                 // string s at buf0 + s nseg nn, where the tree leaves a string's product
                    S.pstride = (size_t)Lb * nn;
                    {
                        ProfScope ps(ctx, "wilson_seg_reg");
                        const dim3 g((unsigned)((ns * S.nseg + 255) / 256)), b(256);
                        if (nocc == 3) hipLaunchKernelGGL((k_wilson_seg_reg<3>), g, b, 0, ctx->stream, S);
                        else hipLaunchKernelGGL((k_wilson_seg_reg<4>), g, b, 0, ctx->stream, S);
                        TBK_HIP(hipGetLastError());
                    }
                    {
                        // (a wavefront per string and the ordered tree of tbk_berry_lanes.inl: round 4's one-thread-per-string
                        // combine walked 256 matrices through dependent loads -- 167 of a call's 281 us)
                        WilsonLanesArgs C2{};
                        C2.nseg = S.nseg;
                        C2.segs = S.segs;
                        C2.prod = S.prod;
                        C2.pstride = S.pstride;
                        ProfScope ps(ctx, "wilson_lanes_combine");
                        if (nocc == 3) hipLaunchKernelGGL((k_wilson_lanes_combine<3>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, C2);
                        else hipLaunchKernelGGL((k_wilson_lanes_combine<4>), dim3((unsigned)ns), dim3(64), 0, ctx->stream, C2);
                        TBK_HIP(hipGetLastError());
                    }
                }
            } else {
                ProfScope ps(ctx, "link_polar_big");
                hipLaunchKernelGGL(k_link_polar_big, dim3((unsigned)std::min<int64_t>(ns * L, nblk)), dim3(256), 0, ctx->stream, W);
                TBK_HIP(hipGetLastError());
            }
            const bool tree = !seg_route && !lanes_route;
            for (int st = 1; tree && st < Lt; st *= 2) {
                WilsonTreeArgs T{cur, nxt, nocc, Lt, st, ns};
                const int64_t items = ns * ((Lt + 2 * st - 1) / (2 * st));
                ProfScope ps(ctx, "wilson_tree");
                hipLaunchKernelGGL(k_wilson_tree, dim3((unsigned)std::min<int64_t>(items, (int64_t)ctx->cus * 4)), dim3(256), 0,
                                   ctx->stream, T);
                TBK_HIP(hipGetLastError());
                std::swap(cur, nxt);
            }
            std::fill(best.begin(), best.begin() + ns, 1e300);
            for (int attempt = 0; attempt < 4; ++attempt) {
                const double alpha = alphas[attempt];
                CayleyArgs C{cur, (size_t)Lb * nn, ab, herm, nocc, cos(alpha), sin(alpha)};
                if (!(lanes_route && attempt == 0)) {
                    ProfScope ps(ctx, "wilson_cayley");
                    hipLaunchKernelGGL(k_wilson_cayley, dim3((unsigned)ns), dim3(256), 0, ctx->stream, C);
                    TBK_HIP(hipGetLastError());
                }
                rc = tbk_eigh_dev(ctx, nocc, herm, ns, ev_dev, nullptr, "wilson_eigh");
                if (rc) return rc;
                {
                    ProfScope ps(ctx, "wilson_phases");
                    hipLaunchKernelGGL(k_wilson_phases, dim3((unsigned)((ns + 63) / 64)), dim3(64), 0, ctx->stream,
                                       (const double*)ev_dev, ns, nocc, alpha, out_dev, hmax_dev);
                    TBK_HIP(hipGetLastError());
                }
                TBK_HIP(hipMemcpyAsync(out_h.data(), out_dev, (size_t)ns * nocc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                TBK_HIP(hipMemcpyAsync(hmax_h.data(), hmax_dev, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                int polar_flag = 0;
                TBK_HIP(hipMemcpyAsync(&polar_flag, ctx->flags_dev + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                TBK_HIP(hipStreamSynchronize(ctx->stream));
                if (polar_flag) {
                    TBK_HIP(hipMemsetAsync(ctx->flags_dev + 1, 0, sizeof(int), ctx->stream));
                    tbk_set_error("tbk_berry_phase: a link overlap matrix is singular (no polar factor U Vh): the occupied "
                                  "subspaces of two neighbouring points are orthogonal in some direction");
                    return TBK_ENOCONV;
                }
                rc = tbk_eigh_check(ctx, nocc);
                if (rc) return rc;
                bool all_ok = true;
                for (int64_t s = 0; s < ns; ++s) {
                    if (hmax_h[s] < best[s]) {      // also false for NaN
                        best[s] = hmax_h[s];
                        memcpy(out + (size_t)(s0 + s) * nocc, out_h.data() + (size_t)s * nocc, (size_t)nocc * sizeof(double));
                    }
                    if (!(best[s] <= 1e4)) all_ok = false;
                }
                if (all_ok) break;
            }
        }
        return TBK_OK;
    }
    // 5..8 bands of wide states, determinant form: a wavefront per (string, segment) with coalesced loads
    // (k_chain_links_wave + k_chain_lu_wave); TBK_CHAIN_WAVE=0 keeps the thread-per-string kernel (A/B runs)
    const bool wave_chain = !berry_evals && chain_wave_applies(v, nocc);
    // segment length: enough threads to fill the chip, segments no shorter than 8 links
    const int64_t target = (int64_t)ctx->cus * 1024;
    int64_t nseg = std::max<int64_t>(1, std::min<int64_t>((A.nlinks + 7) / 8, target / std::max<int64_t>(A.nstrings, 1)));
    A.seg_len = (int)((A.nlinks + nseg - 1) / nseg);
    A.nseg = (A.nlinks + A.seg_len - 1) / A.seg_len;
    if (wave_chain) chain_wave_segments(ctx, A);
    const bool ev = berry_evals != 0;
    // determinant form of 1..4 bands of NARROW states (fewer than 8 components, where the wave-per-string kernels above do not
    // apply): the LDS-tile kernels of tbk_berry_lanes.inl without the polar iteration (round 6).  Not for up to two bands of up
    // to four components: k_chain_partial keeps a link's shared point in registers there (13 us for Kane-Mele-sized states).
    // TBK_WILSON_REG != 3 keeps k_chain_partial.
    LanesPlan LP{};
    const bool lanes_det = !ev && !big && lanes_dets_applies(v, nocc, TBK_LANES_DET_LDS_MAX) && tbk_knobs().chain_wave != 2;   // (TBK_CHAIN_WAVE=2: wide states stay on the wave-per-string kernels)
    if (lanes_det) {
        LP = lanes_plan(ctx, v, nocc, A.sdir, A.nlinks, A.nstrings);
        A.seg_len = LP.seg_len;
        A.nseg = LP.nseg;
    }
    const int64_t per = ev ? (int64_t)nocc * nocc : 1;
    const int64_t nout = A.nstrings * (ev ? nocc : 1);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t pb = al((size_t)A.nseg * A.nstrings * per * sizeof(cd));
    const size_t ob = al((size_t)nout * sizeof(double));
    void* base = nullptr;
    rc = tbk_ctx_scratch(ctx, pb + ob, &base);
    if (rc) return rc;
    A.partial = (cd*)base;
    A.out = (double*)((unsigned char*)base + pb);
    A.flags = ctx->flags_dev;
    const int64_t nthreads = A.nstrings * A.nseg;
    const dim3 grid((unsigned)((nthreads + 255) / 256)), blk(256);
    {
        ProfScope ps(ctx, ev ? "chain_partial_evals" : "chain_partial_det");
        if (wave_chain && !lanes_det) {
            rc = launch_chain_wave(ctx, v, A, nocc, nullptr);
            if (rc) return rc;
        } else if (lanes_det) {
            WilsonLanesArgs S;
            lanes_fill(S, ctx, v, A.occ, nocc, A.nlinks, A.sdir, A.other, A.nstrings, LP);
            S.dets = A.partial;
            S.det_stride = A.nstrings;
            TBK_LANES_LAUNCH(1, S, LP, A.nstrings)
        } else if (!ev) {
            switch (nocc) {
                case 1: hipLaunchKernelGGL((k_chain_partial<1, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 2: hipLaunchKernelGGL((k_chain_partial<2, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 3: hipLaunchKernelGGL((k_chain_partial<3, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 4: hipLaunchKernelGGL((k_chain_partial<4, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 5: hipLaunchKernelGGL((k_chain_partial<5, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 6: hipLaunchKernelGGL((k_chain_partial<6, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 7: hipLaunchKernelGGL((k_chain_partial<7, 1, false>), grid, blk, 0, ctx->stream, A); break;
                case 8: hipLaunchKernelGGL((k_chain_partial<8, 1, false>), grid, blk, 0, ctx->stream, A); break;
                default: hipLaunchKernelGGL((k_chain_partial<0, TBK_MAX_NOCC, false>), grid, blk, 0, ctx->stream, A);
            }
        } else {
            if (nocc == 1) hipLaunchKernelGGL((k_chain_partial<1, 1, true>), grid, blk, 0, ctx->stream, A);
            else if (nocc == 2) hipLaunchKernelGGL((k_chain_partial<2, 1, true>), grid, blk, 0, ctx->stream, A);
            else if (nocc <= 4) hipLaunchKernelGGL((k_chain_partial<0, 4, true>), grid, blk, 0, ctx->stream, A);
            else if (nocc <= 8) hipLaunchKernelGGL((k_chain_partial<0, 8, true>), grid, blk, 0, ctx->stream, A);
            else hipLaunchKernelGGL((k_chain_partial<0, TBK_MAX_NOCC, true>), grid, blk, 0, ctx->stream, A);
        }
        TBK_HIP(hipGetLastError());
    }
    {
        ProfScope ps(ctx, "chain_final");
        if (A.nseg >= 16 && A.nstrings < (int64_t)0x7fffffff) {   // long strings: a wavefront per string
            const int nn = ev ? nocc * nocc : 1;
            int P = 64;
            while (P > 1 && (P / 2 >= A.nseg || (size_t)P * nn * sizeof(cd) > 64 * 1024)) P /= 2;
            A.final_lanes = P;
            const size_t lds = (size_t)P * nn * sizeof(cd);
            const dim3 g2((unsigned)A.nstrings), b2(64);
            if (!ev) hipLaunchKernelGGL((k_chain_final_wave<1, false>), g2, b2, lds, ctx->stream, A);
            else if (nocc == 2) hipLaunchKernelGGL((k_chain_final_wave<2, true, 2>), g2, b2, lds, ctx->stream, A);
            else if (nocc <= 4) hipLaunchKernelGGL((k_chain_final_wave<4, true>), g2, b2, lds, ctx->stream, A);
            else if (nocc <= 8) hipLaunchKernelGGL((k_chain_final_wave<8, true>), g2, b2, lds, ctx->stream, A);
            else hipLaunchKernelGGL((k_chain_final_wave<TBK_MAX_NOCC, true>), g2, b2, lds, ctx->stream, A);
        } else {
            const dim3 g2((unsigned)((A.nstrings + 63) / 64)), b2(64);
            if (!ev) hipLaunchKernelGGL((k_chain_final<1, false>), g2, b2, 0, ctx->stream, A);
            else if (nocc <= 4) hipLaunchKernelGGL((k_chain_final<4, true>), g2, b2, 0, ctx->stream, A);
            else if (nocc <= 8) hipLaunchKernelGGL((k_chain_final<8, true>), g2, b2, 0, ctx->stream, A);
            else hipLaunchKernelGGL((k_chain_final<TBK_MAX_NOCC, true>), g2, b2, 0, ctx->stream, A);
        }
        TBK_HIP(hipGetLastError());
    }
    int flag = 0;
    const size_t outb = (size_t)nout * sizeof(double);
    {
        // phases and status words in one round trip (small results: one copy kernel that also stores the completion word the
        // host polls -- a berry_phase call on a 31 x 31 array was 34 us, two hipMemcpyAsync and a synchronisation 17 of them)
        int fl[4] = {0, 0, 0, 0};
        const int rcr = tbk_small_result(ctx, out, A.out, outb, fl);
        if (rcr) return rcr;
        flag = fl[1];
    }
    if (flag) {
        TBK_HIP(hipMemsetAsync(ctx->flags_dev + 1, 0, sizeof(int), ctx->stream));
        tbk_set_error("tbk_berry_phase: QR iteration for Wilson-loop eigenvalues did not converge");
        return TBK_ENOCONV;
    }
    return TBK_OK;
}
