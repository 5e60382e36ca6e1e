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
//               (nsta == 2: the single exact rotation); meshes: k_grid_rows.
//   5..8      : one thread per k, registers (tbk_solve_reg.inl).
//   9..64     : one 64-lane wavefront (large batches up to 21) or one workgroup per k,
//               S and V in LDS, parallel-ordered (round-robin) Jacobi, nsta/2 disjoint
//               rotations per round; 13..16 on lists: one DPP row per k (tbk_solve_row16.inl).
//   65..2048  : workgroup per k on an L2 workspace, whole-chip rounds (tbk_solve_big.inl) or,
//               for batches, block Jacobi (tbk_solve_blk.inl) -- see launch_wave().
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include "tbk_internal.h"
#include <cstring>

#include "tbk_solve_dev.h"   // GridArgs / ListArgs, exp(2 pi i x), mesh-point decoding, DPP row helpers, the lane-per-matrix QL step

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
        // division-free parameters (one sqrt, one rsqrt): with a = (d_Q - d_P)/2,
        // r = sqrt(a^2 + |g|^2):  c = (|a|+r) / sqrt(2r(r+|a|)),
        // s w = sgn(a) g / sqrt(2r(r+|a|)),  new diagonal = mid -+ sgn(a) r
        const double a = 0.5 * (M.dg[Q] - M.dg[P]), aa = fabs(a);
        const double r = sqrt(a * a + g2);
        const double inv = rsqrt(2.0 * r * (r + aa));
        const double c = (aa + r) * inv;
        const double sg = copysign(1.0, a);
        const cd sw{sg * g.x * inv, sg * g.y * inv};
        const double mid = 0.5 * (M.dg[P] + M.dg[Q]);
        M.dg[P] = mid - sg * r;
        M.dg[Q] = mid + sg * r;
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
__device__ __forceinline__ bool ql_small(SmallMat<N>& M);   // n = 3, 4: direct solver, defined below

template <int N, bool VEC>
// (returns false when the iteration ran into its cap -- NaN input, or a matrix LAPACK would give up on too: the reference's
// np.linalg.eigh raises "Eigenvalues did not converge" there, pythtb.py:939,944; the kernels raise the sticky flag)
__device__ __forceinline__ bool jacobi_small(SmallMat<N>& M) {
    if constexpr (N == 1) {
        return true;
    } else if constexpr (N == 2) {
        // closed form, already ascending: lambda = m -+ r, one sqrt and one rsqrt.
        //   delta >= 0: v- ~ (delta + r, -conj g)    delta < 0: v- ~ (g, delta - r)
        // (the cancellation-free choice), |v-|^2 = 2 r (r + |delta|), v+ = (-conj v-_1, conj v-_0)
        const double a = M.dg[0], d = M.dg[1];
        const cd g = M.up[0][1];
        const double delta = 0.5 * (d - a), mid = 0.5 * (a + d);
        const double r = sqrt(delta * delta + cabs2(g));
        M.dg[0] = mid - r;
        M.dg[1] = mid + r;
        if (VEC) {
            const double ad = fabs(delta);
            const double nrm2 = 2.0 * r * (r + ad);
            cd v0{1.0, 0.0}, v1{0.0, 0.0};
            if (nrm2 > 0.0) {
                const double inv = rsqrt(nrm2);
                if (delta >= 0.0) {
                    v0 = cd{(ad + r) * inv, 0.0};
                    v1 = cd{-g.x * inv, g.y * inv};
                } else {
                    v0 = cd{g.x * inv, g.y * inv};
                    v1 = cd{-(ad + r) * inv, 0.0};
                }
            }
            M.v[0][0] = v0;
            M.v[1][0] = v1;
            M.v[0][1] = cd{-v1.x, v1.y};
            M.v[1][1] = cd{v0.x, -v0.y};
        }
    } else {
#ifndef TBK_SMALL_JACOBI
        return ql_small<N, VEC>(M);      // n = 3, 4: Householder + implicit QL (-DTBK_SMALL_JACOBI keeps the cyclic Jacobi for A/B runs)
#endif
        int sweep = 0;
        for (; sweep < TBK_JACOBI_MAX_SWEEPS; ++sweep) {
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
        return sweep < TBK_JACOBI_MAX_SWEEPS;
    }
    return true;
}

// ---- n = 3, 4: the DIRECT solver, one matrix per thread, everything in registers with static indices:
// Householder tridiagonalisation (N - 2 reflections) and implicit-shift QL with accumulated eigenvectors -- the
// recurrences of LAPACK's zhetrd + zsteqr, which numpy.linalg.eigh runs for the reference (pythtb.py:939-947).
// Cyclic Jacobi needs 4-5 sweeps of 6 rotations on a 4 x 4 matrix, each rotation touching two rows and columns
// of A and all of V (~4.6 k wave-instructions per 64 points measured on Kane-Mele, profiles/r02a: the kernel is
// VALU-bound at 78 % of the fp64 issue rate); tridiagonalisation costs a few hundred instructions once and QL
// about two shifts of at most N - 1 plane rotations per eigenvalue.
// Every thread runs the same full-range sweep i = N-2 .. 0; a position takes part only inside the thread's
// active block [l, m) (EXEC-masked), so register indices stay static while l and m are data.
template <int I, int E, class Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
// The solver in FACTORED form: eigenvalues d, the REAL eigenvector matrix Q of the tridiagonal T, and what turns a column of Q
// into an eigenvector of the Hermitian input -- the diagonal unitary D (dph) that made T's couplings real and the N - 2
// normalised reflectors (us).  Column b of Z = H_0 .. H_{N-3} D Q is formed on demand (small_vector): the mesh kernels sort
// (d, Q) -- half the selects of sorting complex columns -- and then produce, stage and store ONE band at a time, so the N x N
// complex eigenvector matrix (64 VGPRs at N = 4) never exists in registers.
template <int N>
struct SmallFact {
    double d[N];
    double Q[N][N];                             // Q[r][b]: component r of eigenvector b of T
    cd us[N > 2 ? N - 2 : 1][N];                // reflector K, normalised (H_K = 1 - w w^+): entries r > K (the others are never
                                                // read); all zero = no reflection
    cd dph[N];
};

// Householder tridiagonalisation on the UPPER triangle (real diagonal dg, up[r][c] for r < c: the rank-2 update touches
// N (N + 1) / 2 entries instead of N^2 and nothing below the diagonal is kept -- 64 -> 28 registers at N = 4), then implicit
// QL with Wilkinson shifts on (d, e), the rotations accumulated in the real Q (VEC).
// `put_u(K, r, w_r)`: where the normalised reflector K goes, entry r > K (all zero when there was nothing to reflect).  The n <= 4
// kernels keep it in F.us (registers); k_solve_regd (n = 5..8 with eigenvectors) sends it to LDS.
template <int N, bool VEC, class PutU>
__device__ __forceinline__ void tridiag_small_to(double (&dg)[N], cd (&up)[N][N], SmallFact<N>& F, double (&e)[N], PutU&& put_u) {
    static_assert(N >= 3 && N <= 8, "ql_small: n = 3..8");
    double (&d)[N] = F.d;
    cd delta{1.0, 0.0};                        // D_{K+1} = D_K t_K / |t_K| makes the subdiagonal real
    F.dph[0] = cd{1.0, 0.0};
#pragma unroll
    for (int K = 0; K + 1 < N; ++K) {
        cd tK = cconj(up[K][K + 1]);           // a_{K+1,K}
        if (K + 2 < N) {                       // a reflection annihilates rows K+2.. of column K
            // (the decision is taken on the part to be annihilated ALONE, like LAPACK's zlarfg: compared through the sum
            // with |a_{K+1,K}|^2, entries below ~1e-8 of it would be dropped -- an eigenvalue error of up to their size)
            double rest = 0.0;
#pragma unroll
            for (int r = K + 2; r < N; ++r) rest += cabs2(up[K][r]);
            const double absa2 = cabs2(tK);
            const double sigma = rest + absa2;
            cd u[N];
#pragma unroll
            for (int r = 0; r < N; ++r) u[r] = cd{0.0, 0.0};
            if (rest > 0.0) {
                const double inv_n = rsqrt_full(sigma), nrm = sigma * inv_n;
                double absa = 0.0;
                cd ph{1.0, 0.0};
                if (absa2 > 0.0) {
                    const double inv_a = rsqrt_full(absa2);
                    absa = absa2 * inv_a;
                    ph = cd{tK.x * inv_a, tK.y * inv_a};
                }
#pragma unroll
                for (int r = K + 2; r < N; ++r) u[r] = cconj(up[K][r]);
                // the reflector is kept NORMALISED, w = u sqrt(beta) with beta = 2 / (u^+ u) = 1 / (nrm (nrm + |a|)), so that
                // H = 1 - w w^+: no beta to carry (or to divide for), here or in the back-transformation
                const double sb = rsqrt_full(nrm * (nrm + absa));
#pragma unroll
                for (int r = K + 2; r < N; ++r) u[r] = cd{u[r].x * sb, u[r].y * sb};
                const double u1 = (absa + nrm) * sb;
                u[K + 1] = cd{ph.x * u1, ph.y * u1};
                tK = cd{-ph.x * nrm, -ph.y * nrm};
                cd p[N];
                double upr = 0.0;
#pragma unroll
                // (every accumulation below is a chain of fused multiply-adds spelled out: "acc += a b + c d" as written costs a
                // multiply, a fused multiply-add and an add; the chain costs two)
                for (int r = K + 1; r < N; ++r) {           // p = A w on the trailing block, A read through the upper triangle
                    cd acc{dg[r] * u[r].x, dg[r] * u[r].y};
#pragma unroll
                    for (int c = K + 1; c < N; ++c) {
                        if (c > r) cfma_x(acc, up[r][c], u[c]);
                        else if (c < r) {                   // conj(up[c][r]) u_c
                            acc.x = fma(up[c][r].y, u[c].y, fma(up[c][r].x, u[c].x, acc.x));
                            acc.y = fma(-up[c][r].y, u[c].x, fma(up[c][r].x, u[c].y, acc.y));
                        }
                    }
                    p[r] = acc;
                    upr = fma(u[r].y, p[r].y, fma(u[r].x, p[r].x, upr));
                }
                const double kappa = 0.5 * upr;
                cd q[N];
#pragma unroll
                for (int r = K + 1; r < N; ++r) q[r] = cd{fma(-kappa, u[r].x, p[r].x), fma(-kappa, u[r].y, p[r].y)};
#pragma unroll
                for (int r = K + 1; r < N; ++r) {           // A -= u q^+ + q u^+
                    dg[r] = fma(-2.0 * u[r].y, q[r].y, fma(-2.0 * u[r].x, q[r].x, dg[r]));
#pragma unroll
                    for (int c = r + 1; c < N; ++c) {
                        up[r][c].x = fma(-q[r].y, u[c].y, fma(-q[r].x, u[c].x, fma(-u[r].y, q[c].y, fma(-u[r].x, q[c].x, up[r][c].x))));
                        up[r][c].y = fma(q[r].x, u[c].y, fma(-q[r].y, u[c].x, fma(u[r].x, q[c].y, fma(-u[r].y, q[c].x, up[r][c].y))));
                    }
                }
            }
            if (VEC) {
#pragma unroll
                for (int r = K + 1; r < N; ++r) put_u(K, r, u[r]);
            }
        }
        d[K] = dg[K];
        const double t2 = cabs2(tK);
        double mag = 0.0;
        if (t2 > 0.0) {
            const double inv = rsqrt_full(t2);
            mag = t2 * inv;
            delta = cmul(delta, cd{tK.x * inv, tK.y * inv});
        }
        e[K] = mag;
        F.dph[K + 1] = delta;
    }
    d[N - 1] = dg[N - 1];
    e[N - 1] = 0.0;
}

template <int N, bool VEC>
__device__ __forceinline__ void tridiag_small(double (&dg)[N], cd (&up)[N][N], SmallFact<N>& F, double (&e)[N]) {
    static_assert(N <= 4 || !VEC, "tridiag_small keeps the reflectors in registers: n = 3, 4 with eigenvectors, 3..8 without");
    tridiag_small_to<N, VEC>(dg, up, F, e, [&](const int K, const int r, const cd w) __attribute__((always_inline)) { F.us[K][r] = w; });
}

// implicit QL on (F.d, e), rotations accumulated in F.Q (VEC).  Returns false when the iteration ran into LAPACK's limit
// (30 shifts per eigenvalue).
// tql2's own structure: for l = 0 .. N-2 sweep the block [l, m) -- m the first negligible coupling after l -- until e_l is
// negligible.  One loop PER l, so that d_l, d_{l+1}, e_l and the positions of a sweep are static registers: finding "the
// first coupling at or after l that is not negligible" and picking its d's and e by selects cost 80 of the ~200 instructions
// of a sweep when l was data.
// Every loop leaves on a WAVE-UNIFORM condition (no lane has work left): a lane that is done idles behind its EXEC bit.  With a
// per-lane `break` the compiler keeps a second copy of every value that is live after a divergent loop -- d, e and Q once as
// the running values and once as "the values of the lanes that have left": 48 more registers at N = 4, the difference between
// two and four wavefronts per SIMD for the mesh kernels.
// `first_shift` (with `guided`): the shift of this index's FIRST sweep when the block still reaches the end of T -- an eigenvalue
// of T known to ~1e-6 beforehand (ql_lower_roots4) instead of Wilkinson's guess from the leading 2 x 2.  A shift that good leaves
// e_L ~ 1e-6 |T| after one sweep and the cubically convergent second sweep (Wilkinson's shift again) finishes: two sweeps for
// EVERY lane, where the plain iteration takes 2-4 and a wavefront waits for its slowest lane (Kane-Mele rows: 3.98 + 2.45 + 1.02
// sweeps per wavefront for the three indices, 2 + 2 + 1 with the guesses).  A poor guess costs sweeps, never accuracy.
template <int N, bool VEC, int L>
__device__ __forceinline__ bool ql_deflate_small(SmallFact<N>& F, double (&e)[N], const bool guided = false, const double first_shift = 0.0) {
    double (&d)[N] = F.d;
    double (&Q)[N][N] = F.Q;
    auto negligible = [&](const int j) { return fabs(e[j]) <= 2.220446049250313e-16 * (fabs(d[j]) + fabs(d[j + 1])); };
    bool work = true;
    bool first = guided;                       // (a per-lane flag, not "iter == 0": the compiler peels a loop on the latter and keeps two copies of everything)
    for (int iter = 0; iter < 31; ++iter) {
        work = !negligible(L);
        if (__builtin_amdgcn_ballot_w64(work) == 0) break;
        if (work) {
            int m = N - 1;                     // end of the block: the first negligible coupling after L
#pragma unroll
            for (int j = N - 2; j > L; --j) m = negligible(j) ? j : m;
            double dm = d[N - 1];
#pragma unroll
            for (int j = L + 1; j + 1 < N; ++j) dm = m == j ? d[j] : dm;
            // Wilkinson shift = the eigenvalue of the block's leading 2 x 2 nearer to d_L, to full precision: for a block of two it
            // is exact and the block deflates in ONE sweep.  With delta = (d_{L+1} - d_L) / 2 and h = sqrt(delta^2 + e_L^2) it is
            // d_L - e_L^2 / (delta + sgn(delta) h): tql2's g = delta / e_L, e_L / (g + sgn(g) sqrt(g^2 + 1)) multiplied through
            // by e_L -- one reciprocal instead of two.
            auto recip = [](const double x) {
                double y = __builtin_amdgcn_rcp(x);
                y = y * fma(-x, y, 2.0);
                return y * fma(-x, y, 2.0);
            };
            const double dlt = 0.5 * (d[L + 1] - d[L]), el2 = e[L] * e[L];
            const double t1 = fma(dlt, dlt, el2);
            // (dlt^2 + e_L^2 underflows to 0 for a block of denormal scale whose e_L is still not negligible: 0 * rsqrt(0) would
            // be NaN and the lane would spin its 30 sweeps; any h of the right magnitude does there -- ADVICE r5)
            const double hh = t1 > 0.0 ? t1 * rsqrt_full(t1) : fmax(fabs(dlt), fabs(e[L]));
            double g = dm - d[L] + el2 * recip(dlt + copysign(hh, dlt));
            if (first && m == N - 1) g = dm - first_shift;
            first = false;
            double sn = 1.0, cs = 1.0, pp = 0.0;
            bool alive = true;
#pragma unroll
            for (int i = N - 2; i >= L; --i) {
                if (alive && i < m) {
                    const double f = sn * e[i], b = cs * e[i];
                    const double t = f * f + g * g;
                    if (t > 0.0) {
                        const double inv = rsqrt_full(t), r = t * inv;
                        e[i + 1] = r;
                        sn = f * inv;
                        cs = g * inv;
                        g = d[i + 1] - pp;
                        const double r2 = (d[i] - g) * sn + 2.0 * cs * b;
                        pp = sn * r2;
                        d[i + 1] = g + pp;
                        g = cs * r2 - b;
                        if (VEC) {
#pragma unroll
                            for (int r_ = 0; r_ < N; ++r_) {
                                const double zi = Q[r_][i], zj = Q[r_][i + 1];
                                Q[r_][i + 1] = sn * zi + cs * zj;
                                Q[r_][i] = cs * zi - sn * zj;
                            }
                        }
                    } else {                   // r == 0 (underflow): tql2's recovery
                        d[i + 1] -= pp;
                        alive = false;
                    }
                }
            }
            if (alive) {
                d[L] -= pp;
                e[L] = g;
            }
#pragma unroll
            for (int j = L + 1; j < N; ++j)
                if (j == m) e[j] = 0.0;
        }
    }
    // `work` is the test made BEFORE a sweep: with 31 passes of the loop it is the state after the 30th sweep (LAPACK's limit; the
    // 31st sweep of a lane that still has work is harmless and its outcome is not reported).  Re-evaluating negligible(L) here
    // instead keeps d, e live across the divergent loop in a second copy: k_grid_rows<4,1> 128 -> 192 VGPRs, four -> two wavefronts
    // per SIMD, 152 -> 215 us per 4097 x 513 mesh (profiles/r06cfg caught it)
    return !work;
}

// The two LOWEST eigenvalues of the 4 x 4 symmetric tridiagonal (d, e) in closed form, to ~1e-6 |T|: first-sweep shifts for
// ql_deflate_small<.., 0> and <.., 1>.  det(x - T') = x^4 + p x^2 + q x + r for T' = (T - tr T / 4) / |T| (sums of principal
// minors; the cubic and quartic power sums by Newton's identities since tr T' = 0), factored as (x^2 + a x + b)(x^2 - a x + c) with
// a^2 the LARGEST root of the resolvent z^3 + 2 p z^2 + (p^2 - 4 r) z - q^2 -- three real roots, so the trigonometric form, its
// arccos and cos in single precision (a polynomial and v_cos_f32: the roots are shifts, not results).  The largest a pairs the two
// lowest roots in the first factor: x = (-a -+ sqrt(a^2 - 4 b)) / 2, ascending.  Returns false when T is (numerically) a multiple
// of the identity or the resolvent degenerates: the caller then sweeps with Wilkinson's shift as before.
__device__ __forceinline__ bool ql_lower_roots4(const double (&d)[4], const double (&e)[4], double& x1, double& x2) {
    const double nrm = fmax(fmax(fmax(fabs(d[0]), fabs(d[1])), fmax(fabs(d[2]), fabs(d[3]))), fmax(fmax(fabs(e[0]), fabs(e[1])), fabs(e[2])));
    const double s = __builtin_amdgcn_rcp(nrm);                  // (a scale, not a result: the hardware estimate is plenty)
    const double mu = 0.25 * ((d[0] + d[1]) + (d[2] + d[3]));
    const double a0 = (d[0] - mu) * s, a1 = (d[1] - mu) * s, a2 = (d[2] - mu) * s, a3 = (d[3] - mu) * s;
    const double f0 = e[0] * s, f1 = e[1] * s, f2 = e[2] * s;
    const double b0 = f0 * f0, b1 = f1 * f1, b2 = f2 * f2;
    const double p = fma(-0.5, fma(a0, a0, fma(a1, a1, fma(a2, a2, a3 * a3))), -(b0 + b1 + b2));
    const double cub = fma(a0 * a0, a0, fma(a1 * a1, a1, fma(a2 * a2, a2, a3 * a3 * a3)));
    const double q = fma(b0, a2 + a3, fma(b1, a0 + a3, b2 * (a0 + a1))) - cub * (1.0 / 3.0);
    const double r = fma(a0 * a1, a2 * a3, fma(-b0 * a2, a3, fma(-b1 * a0, a3, fma(-b2 * a0, a1, b0 * b2))));
    const double P = fma(p * p, -1.0 / 3.0, -4.0 * r);
    const double Q = fma(p * p * p, -2.0 / 27.0, fma(p * r, 8.0 / 3.0, -q * q));
    const double m2 = P * (-1.0 / 3.0);
    if (!(m2 > 1e-30) || !(nrm > 0.0)) return false;
    const double im = __builtin_amdgcn_rsq(m2), m = m2 * im;
    const float arg = fminf(1.0f, fmaxf(-1.0f, (float)(-0.5 * Q * (im * im * im))));
    // arccos on [-1, 1]: sqrt(1 - |x|) * (a degree-7 polynomial in |x|), mirrored for x < 0 (Abramowitz & Stegun 4.4.46, 2e-8)
    const float ax = fabsf(arg);
    float pl = -0.0012624911f;
    pl = fmaf(pl, ax, 0.0066700901f);
    pl = fmaf(pl, ax, -0.0170881256f);
    pl = fmaf(pl, ax, 0.0308918810f);
    pl = fmaf(pl, ax, -0.0501743046f);
    pl = fmaf(pl, ax, 0.0889789874f);
    pl = fmaf(pl, ax, -0.2145988016f);
    pl = fmaf(pl, ax, 1.5707963050f);
    float th = __builtin_sqrtf(1.0f - ax) * pl;
    th = arg < 0.0f ? 3.14159265f - th : th;
    const float c3 = __builtin_amdgcn_cosf(th * (1.0f / (3.0f * 6.28318531f)));      // v_cos_f32 takes revolutions
    const double z1 = fma(2.0 * m, (double)c3, p * (-2.0 / 3.0));
    if (!(z1 > 1e-30)) return false;
    const double ia = __builtin_amdgcn_rsq(z1), al = z1 * ia;
    const double beta = 0.5 * (p + z1 - q * ia);
    const double disc = fmax(fma(-4.0, beta, z1), 0.0);
    const double sd = disc > 0.0 ? disc * __builtin_amdgcn_rsq(disc) : 0.0;
    x1 = fma(0.5 * (-al - sd), nrm, mu);
    x2 = fma(0.5 * (-al + sd), nrm, mu);
    return true;
}

// The same for 3 x 3: the LOWEST root of x^3 + p x + q (T centred and scaled), x = 2 m cos((theta + 2 pi) / 3) with m = sqrt(-p / 3),
// cos theta = -q / (2 m^3) -- the first-sweep shift of index 0 (index 1 is a block of two: Wilkinson's shift is exact there).
__device__ __forceinline__ bool ql_lowest_root3(const double (&d)[3], const double (&e)[3], double& x1) {
    const double nrm = fmax(fmax(fmax(fabs(d[0]), fabs(d[1])), fabs(d[2])), fmax(fabs(e[0]), fabs(e[1])));
    const double s = __builtin_amdgcn_rcp(nrm);
    const double mu = (d[0] + d[1] + d[2]) * (1.0 / 3.0);
    const double a0 = (d[0] - mu) * s, a1 = (d[1] - mu) * s, a2 = (d[2] - mu) * s;
    const double f0 = e[0] * s, f1 = e[1] * s;
    const double b0 = f0 * f0, b1 = f1 * f1;
    // det(x - T') = x^3 + p x + q: p = sum of the principal 2 x 2 minors, q = -det T'
    const double p = fma(a0, a1, fma(a0, a2, a1 * a2)) - (b0 + b1);
    const double q = -(fma(a0 * a1, a2, fma(-b0, a2, -b1 * a0)));
    const double m2 = p * (-1.0 / 3.0);
    if (!(m2 > 1e-30) || !(nrm > 0.0)) return false;
    const double im = __builtin_amdgcn_rsq(m2), m = m2 * im;
    const float arg = fminf(1.0f, fmaxf(-1.0f, (float)(-0.5 * q * (im * im * im))));
    const float ax = fabsf(arg);
    float pl = -0.0012624911f;
    pl = fmaf(pl, ax, 0.0066700901f);
    pl = fmaf(pl, ax, -0.0170881256f);
    pl = fmaf(pl, ax, 0.0308918810f);
    pl = fmaf(pl, ax, -0.0501743046f);
    pl = fmaf(pl, ax, 0.0889789874f);
    pl = fmaf(pl, ax, -0.2145988016f);
    pl = fmaf(pl, ax, 1.5707963050f);
    float th = __builtin_sqrtf(1.0f - ax) * pl;
    th = arg < 0.0f ? 3.14159265f - th : th;
    // the three roots are 2 m cos((theta - 2 pi k) / 3); the lowest is k = 1 ... in revolutions for v_cos_f32
    const float c = __builtin_amdgcn_cosf((th + 6.28318531f) * (1.0f / (3.0f * 6.28318531f)));
    x1 = fma(2.0 * m * (double)c, nrm, mu);
    return true;
}

template <int N, bool VEC>
__device__ __forceinline__ bool ql_iterate_small(SmallFact<N>& F, double (&e)[N]) {
    bool guided = false;
    double g0 = 0.0, g1 = 0.0;
    if constexpr (N == 4) guided = ql_lower_roots4(F.d, e, g0, g1);
    if constexpr (N == 3) guided = ql_lowest_root3(F.d, e, g0);
    if (VEC) {
#pragma unroll
        for (int r = 0; r < N; ++r)
#pragma unroll
            for (int c = 0; c < N; ++c) F.Q[r][c] = r == c ? 1.0 : 0.0;
    }
    bool ok = true;
    static_for<0, N - 1>([&](auto lt) __attribute__((always_inline)) {
        constexpr int L = decltype(lt)::value;
        if constexpr ((N == 4 && L < 2) || (N == 3 && L == 0)) ok = ql_deflate_small<N, VEC, L>(F, e, guided, L == 0 ? g0 : g1) && ok;
        else ok = ql_deflate_small<N, VEC, L>(F, e) && ok;
    });
    return ok;
}

template <int N, bool VEC>
__device__ __forceinline__ bool ql_small_core(double (&dg)[N], cd (&up)[N][N], SmallFact<N>& F) {
    double e[N];
    tridiag_small<N, VEC>(dg, up, F, e);
    return ql_iterate_small<N, VEC>(F, e);
}

// staging slots (16 B) per wavefront of the mesh-row kernels: 64 N for a band's transposition; at N = 4 five more rows of 64
// hold, per lane, the first reflector (3 complex numbers) and the running minima of the three gaps: 9.7 KB of LDS per
// wavefront with the coefficient cells, which still lets 16 wavefronts share a compute unit's 160 KB
__host__ __device__ constexpr int rows_stage_slots(const int n) { return 64 * (n == 4 ? 9 : n); }

// column B of Z = H_0 .. H_{N-3} D Q: the eigenvector of the Hermitian input that belongs to d[B]
template <int N, int B>
__device__ __forceinline__ void small_vector(const SmallFact<N>& F, cd (&z)[N]) {
#pragma unroll
    for (int r = 0; r < N; ++r) z[r] = cd{F.dph[r].x * F.Q[r][B], F.dph[r].y * F.Q[r][B]};
#pragma unroll
    for (int K = N - 3; K >= 0; --K) {
        // (no test for "no reflection": then w = 0 and the update below is the identity)
        cd w{0.0, 0.0};
#pragma unroll
        for (int r = K + 1; r < N; ++r) {                                  // w^+ z
            w.x = fma(F.us[K][r].y, z[r].y, fma(F.us[K][r].x, z[r].x, w.x));
            w.y = fma(-F.us[K][r].y, z[r].x, fma(F.us[K][r].x, z[r].y, w.y));
        }
#pragma unroll
        for (int r = K + 1; r < N; ++r) {
            z[r].x = fma(F.us[K][r].y, w.y, fma(-F.us[K][r].x, w.x, z[r].x));
            z[r].y = fma(-F.us[K][r].y, w.x, fma(-F.us[K][r].x, w.y, z[r].y));
        }
    }
}

template <int N>
__device__ __forceinline__ void small_vectors_all(const SmallFact<N>& F, SmallMat<N>& M) {
    static_for<0, N>([&](auto bt) __attribute__((always_inline)) {
        constexpr int b = decltype(bt)::value;
        cd z[N];
        small_vector<N, b>(F, z);
#pragma unroll
        for (int r = 0; r < N; ++r) M.v[r][b] = z[r];
    });
}

template <int N, bool VEC>
__device__ __forceinline__ bool ql_small(SmallMat<N>& M) {
    SmallFact<N> F;
    const bool ok = ql_small_core<N, VEC>(M.dg, M.up, F);
#pragma unroll
    for (int j = 0; j < N; ++j) M.dg[j] = F.d[j];
    if (VEC) small_vectors_all<N>(F, M);
    return ok;
}

// ascending order of (d, Q columns) by the same networks as sort_small below -- on REAL columns
template <int N, int I, int J>
__device__ __forceinline__ void cmpxchg_fact(SmallFact<N>& F) {
    const bool sw = F.d[I] > F.d[J];
    const double lo = sw ? F.d[J] : F.d[I], hi = sw ? F.d[I] : F.d[J];
    F.d[I] = lo;
    F.d[J] = hi;
#pragma unroll
    for (int o = 0; o < N; ++o) {
        const double a = F.Q[o][I], b = F.Q[o][J];
        F.Q[o][I] = sw ? b : a;
        F.Q[o][J] = sw ? a : b;
    }
}
template <int N>
__device__ __forceinline__ void sort_fact(SmallFact<N>& F) {
    if constexpr (N == 3) {
        cmpxchg_fact<N, 0, 1>(F);
        cmpxchg_fact<N, 1, 2>(F);
        cmpxchg_fact<N, 0, 1>(F);
    } else if constexpr (N == 4) {
        // With the guided first sweeps (ql_lower_roots4) index 0 deflates to the lowest eigenvalue and index 1 to the second, so
        // only the last pair can be out of order -- unless a guess missed or T split.  One wave-uniform test decides between
        // one compare-exchange and the full network (5): most wavefronts take the short way.
        const bool nearly = F.d[0] <= F.d[1] && F.d[1] <= fmin(F.d[2], F.d[3]);
        if (__builtin_amdgcn_ballot_w64(!nearly) == 0) {
            cmpxchg_fact<N, 2, 3>(F);
        } else {
            cmpxchg_fact<N, 0, 1>(F);
            cmpxchg_fact<N, 2, 3>(F);
            cmpxchg_fact<N, 0, 2>(F);
            cmpxchg_fact<N, 1, 3>(F);
            cmpxchg_fact<N, 1, 2>(F);
        }
    }
}

// n = 3, 4 on meshes: bring the eigenpairs into ascending order in place with a sorting network (3 / 5 compare-exchanges
// of an eigenvalue and its eigenvector column).  k_grid_rows stores band planes with static register indices, so picking
// "the band of rank r" used to be a 4-way select of freshly multiplied candidates for every (rank, component): 640 of the
// kernel's ~3400 instructions per 64 points; the network is ~200.  (Exactly equal eigenvalues -- Kramers pairs -- may end
// up in either order within the pair: a gauge choice, but a fixed function of the matrix.)
template <int N, int I, int J>
__device__ __forceinline__ void cmpxchg_pair(SmallMat<N>& M) {
    const bool sw = M.dg[I] > M.dg[J];
    const double lo = sw ? M.dg[J] : M.dg[I], hi = sw ? M.dg[I] : M.dg[J];
    M.dg[I] = lo;
    M.dg[J] = hi;
#pragma unroll
    for (int o = 0; o < N; ++o) {
        const cd a = M.v[o][I], b = M.v[o][J];
        M.v[o][I] = cd{sw ? b.x : a.x, sw ? b.y : a.y};
        M.v[o][J] = cd{sw ? a.x : b.x, sw ? a.y : b.y};
    }
}
template <int N>
__device__ __forceinline__ void sort_small(SmallMat<N>& M) {
    if constexpr (N == 3) {
        cmpxchg_pair<N, 0, 1>(M);
        cmpxchg_pair<N, 1, 2>(M);
        cmpxchg_pair<N, 0, 1>(M);
    } else if constexpr (N == 4) {
        cmpxchg_pair<N, 0, 1>(M);
        cmpxchg_pair<N, 2, 3>(M);
        cmpxchg_pair<N, 0, 2>(M);
        cmpxchg_pair<N, 1, 3>(M);
        cmpxchg_pair<N, 1, 2>(M);
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

// assemble S(k) into registers from the slot table
template <int N>
__device__ __forceinline__ void assemble_small(const ModelView& mv, const cd (&z)[4], SmallMat<N>& M) {
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

template <int N, bool VEC>
__device__ __forceinline__ void init_vectors(SmallMat<N>& M) {
    if (VEC) {
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = 0; b < N; ++b) M.v[a][b] = cd{a == b ? 1.0 : 0.0, 0.0};
    }
}

// min over the mesh of E[b+1]-E[b] (all_gaps.min, pythtb.py:2495,2530): wave
// reduction, then one conditional atomicMin per wave into a shard.  Gaps are
// >= 0, so their bit patterns order like the values; the plain pre-read may be
// stale but is never smaller than the truth, so skipping on it is safe.
__device__ __forceinline__ void gap_min_wave(unsigned long long* shard_row, int b, double g) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g = fmin(g, __shfl_xor(g, off));
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(g, 0.0));
        if (bits < __hip_atomic_load(shard_row + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMin(shard_row + b, bits);
    }
}

// a call's last kernel did not take the completion word itself: one more launch that only stores it
__global__ void k_signal_only(const DoneArgs done) {
    if (threadIdx.x == 0) tbk_signal_done(done);
}

// ---- k lists, n <= 4: S(k) = sum_R e^{2 pi i k.R} U_R from an LDS image of the model's R-grouped table (tbk_model_upload: the
// distinct lattice vectors and, per vector, the dense block of slot coefficients) -- BASELINE north_star's "coalesced HBM reads of
// the hopping table and an LDS-staged orbital tile" for the list kernels.  Rounds 1-4 walked the term table per slot with scalar
// loads: a chain of 43 dependent loads per wavefront (Haldane), 5.6 k of a wavefront's 15 k cycles.  Here the workgroup copies the
// table once (16-byte coalesced loads), every lane then reads it as LDS broadcasts, and a lattice vector's phase is formed once
// per point instead of once per term.  The same operations in the same order for one or two points per lane (fused
// multiply-adds spelled out): the two kernels stay bit-identical (tests/test_regimes.py::test_two_points_per_lane_...).
template <int N>
__device__ __forceinline__ void rtable_stage(const ModelView& mv, unsigned char* lds) {
    constexpr int NSLOT = N * (N + 1) / 2;
    int4* rv = reinterpret_cast<int4*>(lds);
    cd* blk = reinterpret_cast<cd*>(rv + mv.nR);
    for (int i = threadIdx.x; i < mv.nR; i += 256) rv[i] = mv.rvec[i];
    for (int i = threadIdx.x; i < mv.nR * NSLOT; i += 256) blk[i] = mv.rblock[i];
    __syncthreads();
}
template <int N, int KPT>
__device__ __forceinline__ void assemble_small_rlds(const unsigned char* lds, const int nR, const cd (&z)[KPT][4], SmallMat<N> (&M)[KPT]) {
    constexpr int NSLOT = N * (N + 1) / 2;
    const int4* rv = reinterpret_cast<const int4*>(lds);
    const cd* blk = reinterpret_cast<const cd*>(rv + nR);
    cd acc[KPT][NSLOT];
#pragma unroll
    for (int j = 0; j < KPT; ++j)
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) acc[j][s] = cd{0.0, 0.0};
    auto one_vector = [&](const int r) __attribute__((always_inline)) {
        const int4 Rv = rv[r];
        // (the same on every lane: into scalar registers, so that the loops over |R_d| below are scalar-controlled)
        const int Rr[4] = {__builtin_amdgcn_readfirstlane(Rv.x), __builtin_amdgcn_readfirstlane(Rv.y),
                           __builtin_amdgcn_readfirstlane(Rv.z), __builtin_amdgcn_readfirstlane(Rv.w)};
        cd e[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) e[j] = cd{1.0, 0.0};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int m = Rr[d] < 0 ? -Rr[d] : Rr[d];
            const double sg = Rr[d] < 0 ? -1.0 : 1.0;
            for (int q = 0; q < m; ++q)
#pragma unroll
                for (int j = 0; j < KPT; ++j) e[j] = cmul_x(e[j], cd{z[j][d].x, sg * z[j][d].y});
        }
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const cd c = blk[r * NSLOT + s];
#pragma unroll
            for (int j = 0; j < KPT; ++j) cfma_x(acc[j][s], c, e[j]);
        }
    };
    // up to 8 lattice vectors (nearest and second neighbours of a 2-D lattice: 7): unrolled under wave-uniform guards, so that
    // the LDS reads of all of them are in flight before the first phase is formed; longer tables run the plain loop
    if (nR <= 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < nR) one_vector(r);
    } else {
        for (int r = 0; r < nR; ++r) one_vector(r);
    }
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        int s = 0;
#pragma unroll
        for (int a = 0; a < N; ++a)
#pragma unroll
            for (int b = a; b < N; ++b, ++s) {
                if (b == a) M[j].dg[a] = acc[j][s].x; else M[j].up[a][b] = acc[j][s];
            }
    }
}

// ---- k list (MODE 0) or supplied matrices (MODE 2) -> eval[b][k], evec[b][k][o]
template <int N, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_solve_small(const ModelView mv, const int64_t nk, const ListArgs L) {
    extern __shared__ __align__(16) unsigned char lds_rtab[];
    if constexpr (MODE == 0) {
        if (mv.nR > 0) rtable_stage<N>(mv, lds_rtab);
    }
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    // (the point's work stays INLINE under this test: moved into a function of its own the compiler fused other products and
    // the eigenvectors differed from k_solve_small_multi's by an ulp -- tests/test_regimes.py::test_two_points_per_lane_...)
    if (idx < nk) {
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    SmallMat<N> M;
    if constexpr (MODE == 2) {
        const cd* h = L.ham + idx * (int64_t)(N * N);
#pragma unroll
        for (int a = 0; a < N; ++a) {
            M.dg[a] = h[a * N + a].x;
#pragma unroll
            for (int b = a + 1; b < N; ++b) M.up[a][b] = h[a * N + b];
        }
    } else {
        cd z[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (d < mv.dim_k) kk[d] = L.k[idx * mv.dim_k + d];
            z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        }
        if constexpr (MODE == 0) {
            if (mv.nR > 0) {
                cd z1[1][4];
#pragma unroll
                for (int d = 0; d < 4; ++d) z1[0][d] = z[d];
                SmallMat<N> M1[1];
                assemble_small_rlds<N, 1>(lds_rtab, mv.nR, z1, M1);
#pragma unroll
                for (int a = 0; a < N; ++a) {
                    M.dg[a] = M1[0].dg[a];
#pragma unroll
                    for (int b = a + 1; b < N; ++b) M.up[a][b] = M1[0].up[a][b];
                }
            } else {
                assemble_small<N>(mv, z, M);
            }
        } else {
            assemble_small<N>(mv, z, M);
        }
    }
    init_vectors<N, VEC>(M);
    if constexpr (N > 2) {
        if (!jacobi_small<N, VEC>(M) && L.flags) L.flags[0] = 1;
    } else {
        jacobi_small<N, VEC>(M);
    }
    int rk[N];
    double sorted[N];
    ranks_small<N>(M.dg, rk, sorted);
#pragma unroll
    for (int b = 0; b < N; ++b) L.eval[(int64_t)b * nk + idx] = sorted[b];
    if (VEC) {
        // eigenvector of H = D^+ (eigenvector of S): component o times conj(e_o)
        cd fo[N];
#pragma unroll
        for (int o = 0; o < N; ++o) {
            if constexpr (MODE == 2) fo[o] = cd{1.0, 0.0};
            else if (mv.nspin == 2 && (o & 1)) fo[o] = fo[o - (o > 0)];
            else fo[o] = cconj(expi2pi(kdot(kk, mv.orb[o])));
        }
#pragma unroll
        for (int b = 0; b < N; ++b) {
            cd* out = L.evec + ((int64_t)rk[b] * nk + idx) * N;
#pragma unroll
            for (int o = 0; o < N; ++o) out[o] = cmul(M.v[o][b], fo[o]);
        }
    }
    }
    if (L.done.word) {               // (kernel-uniform; small calls only: every lane's results reach the host before the ticket)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) tbk_signal_done(L.done);
    }
}

// ---- the same for KPT k-points per thread (eigenvalue-only k lists of 2..4 states from 2^19 points).
// The list kernel above is bound by neither arithmetic nor bytes: a wavefront lives 15 k cycles of which 5.6 k wait on its 43
// dependent scalar table loads and 1.5 k issue scalar control flow (the loops over a slot's terms and over |R_d| factors) --
// replacing sincospi by the 32-instruction expi2pi (-25 % VALU) changed nothing: 21.9 -> 22.3 us per 2^20 points.  With KPT
// points per lane the same scalar stream serves KPT times the vector work.  Point j of a thread is 256 j further on: coalesced.
template <int N, int KPT>
__device__ __forceinline__ void assemble_small_multi(const ModelView& mv, const cd (&z)[KPT][4], SmallMat<N> (&M)[KPT]) {
    int slot = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = a; b < N; ++b, ++slot) {
            const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
            cd acc[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) acc[j] = cd{0.0, 0.0};
            for (int t = t0; t < t1; ++t) {
                const cd amp = mv.term_amp[t];
                const int4 R = mv.term_R[t];
                // (phase_of_R's loops once per term, all points inside: the same products in the same order as slot_sum)
                cd e[KPT];
#pragma unroll
                for (int j = 0; j < KPT; ++j) e[j] = cd{1.0, 0.0};
                const int r[4] = {R.x, R.y, R.z, R.w};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int m = r[d] < 0 ? -r[d] : r[d];
                    const double sg = r[d] < 0 ? -1.0 : 1.0;
                    for (int q = 0; q < m; ++q)
#pragma unroll
                        for (int j = 0; j < KPT; ++j) e[j] = cmul(e[j], cd{z[j][d].x, sg * z[j][d].y});
                }
#pragma unroll
                for (int j = 0; j < KPT; ++j) cfma(acc[j], amp, e[j]);
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                if (b == a) M[j].dg[a] = acc[j].x; else M[j].up[a][b] = acc[j];
            }
        }
    }
}

template <int N, bool VEC, int KPT>
__global__ __launch_bounds__(256) void k_solve_small_multi(const ModelView mv, const int64_t nk, const ListArgs L) {
    extern __shared__ __align__(16) unsigned char lds_rtab[];
    if (mv.nR > 0) rtable_stage<N>(mv, lds_rtab);
    const int64_t idx0 = (int64_t)blockIdx.x * (256 * KPT) + threadIdx.x;
    if (idx0 >= nk) return;
    double kk[KPT][4];
    cd z[KPT][4];
    int64_t idx[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        idx[j] = idx0 + 256 * j;
        const int64_t src = idx[j] < nk ? idx[j] : nk - 1;      // (a point past the end shadows the last one and is not stored)
#pragma unroll
        for (int d = 0; d < 4; ++d) kk[j][d] = d < mv.dim_k ? L.k[src * mv.dim_k + d] : 0.0;
    }
    // exp(2 pi i k_d): a list from k_uniform_mesh runs along its last axis, so the points of a lane (256 apart) mostly share their
    // leading coordinates bit for bit -- the same function of the same argument is then taken once (wave-uniform test; the bits of
    // every result are what they were)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        if (d < mv.dim_k) {
            z[0][d] = expi2pi(kk[0][d]);
#pragma unroll
            for (int j = 1; j < KPT; ++j) {
                if (__builtin_amdgcn_ballot_w64(kk[j][d] != kk[j - 1][d]) == 0) z[j][d] = z[j - 1][d];
                else z[j][d] = expi2pi(kk[j][d]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) z[j][d] = cd{1.0, 0.0};
        }
    }
    SmallMat<N> M[KPT];
    if (mv.nR > 0) assemble_small_rlds<N, KPT>(lds_rtab, mv.nR, z, M);
    else assemble_small_multi<N, KPT>(mv, z, M);
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
        init_vectors<N, VEC>(M[j]);
        if constexpr (N > 2) {
            if (!jacobi_small<N, VEC>(M[j]) && L.flags) L.flags[0] = 1;
        } else {
            jacobi_small<N, VEC>(M[j]);
        }
        int rk[N];
        double sorted[N];
        ranks_small<N>(M[j].dg, rk, sorted);
        if (idx[j] < nk) {
#pragma unroll
            for (int b = 0; b < N; ++b) L.eval[(int64_t)b * nk + idx[j]] = sorted[b];
            if (VEC) {
                cd fo[N];
#pragma unroll
                for (int o = 0; o < N; ++o) {
                    if (mv.nspin == 2 && (o & 1)) fo[o] = fo[o - (o > 0)];
                    else fo[o] = cconj(expi2pi(kdot(kk[j], mv.orb[o])));
                }
#pragma unroll
                for (int b = 0; b < N; ++b) {
                    cd* out = L.evec + ((int64_t)rk[b] * nk + idx[j]) * N;
#pragma unroll
                    for (int o = 0; o < N; ++o) out[o] = cmul(M[j].v[o][b], fo[o]);
                }
            }
        }
    }
}

// ---- per-axis tables of a regular mesh: one thread per (axis, index)
__global__ __launch_bounds__(256) void k_grid_tables(const ModelView mv, const GridArgs G, cd* tz0, cd* tf0, cd* tf_first = nullptr) {
    int t = blockIdx.x * 256 + threadIdx.x;
    int d = 0;
    int64_t zoff = 0, foff = 0;
    for (; d < G.wv.dim_arr; ++d) {
        if (t < G.wv.mesh[d]) break;
        t -= G.wv.mesh[d];
        zoff += G.wv.mesh[d];
        foff += (int64_t)G.wv.mesh[d] * mv.nsta;
    }
    if (d >= G.wv.dim_arr) return;
    int64_t g = t + G.off[d];
    const int nd = G.gmesh[d];
    const bool wrap = g == nd - 1;       // periodic image of index 0 (impose_pbc, pythtb.py:2729-2747)
    if (wrap) g = 0;
    const double kd = G.start_k[d] + (double)g / (double)(nd - 1);   // pythtb.py:2477,2490-2491
    tz0[zoff + t] = expi2pi(kd);
    for (int o = 0; o < mv.nsta; ++o) {
        const double4 tau = mv.orb[o];
        const double td = d == 0 ? tau.x : d == 1 ? tau.y : d == 2 ? tau.z : tau.w;
        cd f = cconj(expi2pi(kd * td));
        if (wrap) f = cmul(f, G.pbc[d * mv.nsta + o]);
        tf0[foff + (int64_t)t * mv.nsta + o] = f;
    }
    // the last axis' phases at GLOBAL index 0 (k_d = start_k[d] + 0 / (nd - 1): the expression above at g = 0, the same bits a
    // window that holds column 0 has in tf[last][0..n)), for a window whose last column is the image but that starts later
    if (tf_first && t == 0 && d == G.wv.dim_arr - 1) {
        const double k0 = G.start_k[d] + (double)(int64_t)0 / (double)(nd - 1);
        for (int o = 0; o < mv.nsta; ++o) {
            const double4 tau = mv.orb[o];
            const double td = d == 0 ? tau.x : d == 1 ? tau.y : d == 2 ? tau.z : tau.w;
            tf_first[o] = cconj(expi2pi(k0 * td));
        }
    }
}

// ---- regular mesh (solve_on_grid): one wavefront per 64-point chunk of a mesh
// row, so every leading-axis quantity is wave-uniform (scalar loads) and the
// per-lane phases come from coalesced reads of the last axis' table.
template <int N>
__global__ __launch_bounds__(256) void k_grid_small(const ModelView mv, const GridArgs G) {
    const int64_t chunk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (chunk >= G.nchunks) return;
    if (blockIdx.x == 0 && threadIdx.x < (N > 1 ? N - 1 : 0)) {   // re-arm the other parity
        for (int s = 0; s < TBK_GAP_SHARDS; ++s) G.gaps_next[s * N + threadIdx.x] = 0x7ff0000000000000ull;
    }
    const int lane = threadIdx.x & 63;
    const int last = G.last;
    const int nlast = G.wv.mesh[last];
    unsigned row = (unsigned)(chunk / G.cpr);
    const int jc = (int)(chunk - (int64_t)row * G.cpr);
    const int jl = jc * 64 + lane;
    const bool active = jl < nlast;
    const int jj = active ? jl : nlast - 1;           // idle lanes shadow the row's last point (no stores)
    const int64_t point = (int64_t)row * nlast + jj;

    cd z[4] = {cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}};
    cd fo[N];
#pragma unroll
    for (int o = 0; o < N; ++o) fo[o] = G.tf[last][(int64_t)jj * N + o];
    // leading axes: wave-uniform indices, last-to-first
#pragma unroll
    for (int d = 2; d >= 0; --d) {
        if (d < last) {
            const unsigned md = (unsigned)G.wv.mesh[d];
            const unsigned q = row / md;
            const unsigned id = row - q * md;
            row = q;
            z[d] = G.tz[d][id];
#pragma unroll
            for (int o = 0; o < N; ++o) fo[o] = cmul(fo[o], G.tf[d][(int64_t)id * N + o]);
        }
    }
    // the last axis' unit phase sits at z[last]; keep the array statically indexed
    const cd zl = G.tz[last][jj];
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d == last) z[d] = zl;

    SmallMat<N> M;
    init_vectors<N, true>(M);
    if (TBK_ABLATE(G.ablate) != 2) {
        assemble_small<N>(mv, z, M);
        if constexpr (N > 2) {
            if (!jacobi_small<N, true>(M) && G.flags) G.flags[0] = 1;
        } else {
            jacobi_small<N, true>(M);
        }
    } else {
#pragma unroll
        for (int a = 0; a < N; ++a) M.dg[a] = zl.x + a;
    }
    int rk[N];
    double sorted[N];
    ranks_small<N>(M.dg, rk, sorted);
    if constexpr (N > 1) {
        unsigned long long* shard = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * N;
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) gap_min_wave(shard, b, sorted[b + 1] - sorted[b]);
    }
    bool do_store = active;
#ifdef TBK_DIAG
    if (G.ablate == 1) do_store = do_store && sorted[0] == 1.2345e300;   // "no stores" without letting the compiler drop the solve
#endif
    if (do_store) {
#pragma unroll
        for (int b = 0; b < N; ++b) {
            cd* out = wf_at(G.wv, rk[b], point);
#pragma unroll
            for (int o = 0; o < N; ++o) out[o] = cmul(M.v[o][b], fo[o]);
        }
    }
}

// ---- n = 3, 4 on mesh rows: S(k) from the row's coefficient cells, every product spelled out in fused operations --
// k_grid_rows and k_grid_rows_flux must assemble the same bits (their minimal gaps are compared bit for bit), and the
// periodic images rely on equal table entries giving equal matrices, hence equal eigenvectors.  (Folding the orbital phases
// into the matrix -- H(k) = F S F^+ as the reference builds it -- would save the N^2 products on the way out, but the image
// of a point would then be solved from a DIFFERENT matrix and come out in another gauge: a closed string's Berry phase
// would pick up the difference.)
// (cfma_x / cfmac_x: tbk_internal.h)
__device__ __forceinline__ cd cmulc_x(const cd a, const cd b) {                    // a conj(b)
    return cd{fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y))};
}
template <int N, int PM>
__device__ __forceinline__ void rows_assemble(const cd* __restrict__ C, const int npow, const int pmax, const cd zl,
                                              double (&dg)[N], cd (&up)[N][N]) {
    int slot = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
        for (int b = a; b < N; ++b, ++slot) {
            const cd* Cs = C + slot * npow + pmax;
            cd acc = Cs[0];
            cd zp = zl;
            if constexpr (PM >= 0) {
#pragma unroll
                for (int p = 1; p <= PM; ++p) {
                    cfma_x(acc, Cs[p], zp);
                    cfmac_x(acc, Cs[-p], zp);
                    if (p < PM) zp = cmul_x(zp, zl);
                }
            } else {
                for (int p = 1; p <= pmax; ++p) {
                    cfma_x(acc, Cs[p], zp);
                    cfmac_x(acc, Cs[-p], zp);
                    zp = cmul_x(zp, zl);
                }
            }
            if (b == a) dg[a] = acc.x;
            else up[a][b] = acc;
            // (left alone the scheduler hoists all N (N + 1) / 2 (2 PM + 1) broadcast reads of the cells to the top -- 120 registers
            // of coefficients in flight at N = 4, the peak of the whole kernel; two slots' worth at a time is plenty of cover)
            if ((slot & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
    }
}
// a wave-uniform value the compiler cannot see to be uniform (it depends on threadIdx.x >> 6): into scalar registers
__device__ __forceinline__ double uniform_d(const double v) {
    const I2 i = __builtin_bit_cast(I2, v);
    return __builtin_bit_cast(double, I2{__builtin_amdgcn_readfirstlane(i.lo), __builtin_amdgcn_readfirstlane(i.hi)});
}

// ---- regular mesh, polynomial form.  On a mesh row the leading-axis phases are
// fixed, so S_ab(k) = sum_p C_ab,p z_last^p with row coefficients
//   C_ab,p = sum_{t in cell(ab,p)} amp_t prod_{d<last} z_d^{R_d}
// A wavefront owns `seg` consecutive 64-point chunks of one row: its lanes build
// the row's coefficient cells once (in parallel, one cell per lane) into LDS and
// every point then costs (2 pmax + 1) complex FMAs per slot read as LDS broadcasts
// -- no per-point table walk, no scalar-load latency chain, no sincospi.
// (TBK_ROWS_OCC: empty in every build; profiles/dev_tu_rows4.sh defines it as __attribute__((amdgpu_waves_per_eu(N, 8))) to measure, by
// the spill bytes it forces, how far the kernel is from the next occupancy step)
#ifndef TBK_ROWS_OCC
#define TBK_ROWS_OCC
#endif
template <int N, int PM>
__global__ __launch_bounds__(256) TBK_ROWS_OCC void k_grid_rows(const ModelView mv, const GridArgs G) {
    extern __shared__ __align__(16) unsigned char lds_rows[];
    constexpr int NSLOT = N * (N + 1) / 2;
    // (the wavefront's number in scalar registers: the tile, its row, chunk range, LDS regions and output base are then scalar too)
    const int wib = N > 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int pmax = PM >= 0 ? PM : mv.pmax;      // PM: compile-time range of the last lattice component
    const int npow = 2 * pmax + 1;
    const int ncell = NSLOT * npow;
    cd* C = reinterpret_cast<cd*>(lds_rows) + wib * ncell;
    cd* stage = reinterpret_cast<cd*>(lds_rows) + 4 * ncell + wib * rows_stage_slots(N);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wib;
    const bool live = tile < G.ntiles;
    const int last = G.last;
    const int nlast = G.wv.mesh[last];
    // The last point of a closed mesh row is the periodic image of the first: the same S(k), the same eigenvectors, another
    // orbital phase (pythtb.py:2729-2747).  When the whole last axis lies in the window the chunks cover nlast - 1 columns and
    // the lane of column 0 stores the image as well -- a row of 64 q + 1 points (513, 257, 65: N + 1 with N a power of two is what
    // everybody types) paid a whole wavefront-chunk for that one point: 11 % of configs[3]'s solve, half of a 65-point row.
    const int ncol = G.img_last ? nlast - 1 : nlast;
    unsigned row = 0;
    int jc0 = 0, jc1 = 0;
    cd frow[N];
#pragma unroll
    for (int o = 0; o < N; ++o) frow[o] = cd{1.0, 0.0};
    if (live) {
        row = (unsigned)(tile / G.tpr);
        if constexpr (N > 2) row = (unsigned)__builtin_amdgcn_readfirstlane((int)row);   // (the division runs on the vector unit)
        const int ts = (int)(tile - (int64_t)row * G.tpr);
        jc0 = ts * G.seg;
        jc1 = min(jc0 + G.seg, G.cpr);
        cd z[4] = {cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}};
        unsigned rem = row;
#pragma unroll
        for (int d = 2; d >= 0; --d) {
            if (d < last) {
                const unsigned md = (unsigned)G.wv.mesh[d];
                const unsigned q = rem / md;
                const unsigned id = rem - q * md;
                rem = q;
                z[d] = G.tz[d][id];
#pragma unroll
                for (int o = 0; o < N; ++o) frow[o] = cmul(frow[o], G.tf[d][(int64_t)id * N + o]);
            }
        }
        for (int cell = lane; cell < ncell; cell += 64) {
            const int t0 = mv.cell_ptr[cell], t1 = mv.cell_ptr[cell + 1];
            cd acc{0.0, 0.0};
            for (int t = t0; t < t1; ++t) {
                int4 R = mv.term_R[t];
                if (last == 0) R.x = 0; else if (last == 1) R.y = 0; else if (last == 2) R.z = 0; else R.w = 0;
                cfma(acc, mv.term_amp[t], phase_of_R(z, R));
            }
            C[cell] = acc;
        }
    }
    __syncthreads();
    if (!live) return;
    cd ips[N];
#pragma unroll
    for (int o = 0; o < N; ++o) ips[o] = cd{1.0, 0.0};
    if constexpr (N > 2) {             // (the row's phases are the same on every lane: 4 N scalar registers instead of vector ones)
#pragma unroll
        for (int o = 0; o < N; ++o) frow[o] = cd{uniform_d(frow[o].x), uniform_d(frow[o].y)};
        if (G.img_last && jc0 == 0) {
#pragma unroll
            for (int o = 0; o < N; ++o) {
                const cd t = cmulc_x(G.tf[last][(int64_t)(nlast - 1) * N + o], G.tf[last][(int64_t)o]);
                ips[o] = cd{uniform_d(t.x), uniform_d(t.y)};
            }
        } else if (G.img_win && jc1 == G.cpr) {       // (this tile solves the image column itself: the same factor, column 0's phases from tf0)
#pragma unroll
            for (int o = 0; o < N; ++o) {
                const cd t = cmulc_x(G.tf[last][(int64_t)(nlast - 1) * N + o], G.tf0[o]);
                ips[o] = cd{uniform_d(t.x), uniform_d(t.y)};
            }
        }
    }
    if (TBK_ABLATE(G.ablate) == 4) {   // diagnostics: tile set-up only
        if (C[0].x == 1.2345e300) G.wv.data[0] = C[1];
        return;
    }

    double gmin[N > 1 ? N - 1 : 1];
#pragma unroll
    for (int b = 0; b + 1 < N; ++b) gmin[b] = __longlong_as_double(0x7ff0000000000000ll);
    // Staging-buffer slots (16 B each).  A lane writes slot lane*N + o and reads slot i*64 + lane; ds_write_b128 serves
    // 8 consecutive lanes per LDS cycle, and for even N those lanes' slots repeat mod 8 (2-way conflicts for N = 2, 4-way
    // for N = 4: 3.5 conflict cycles per LDS instruction measured, profiles/r01g).  XOR-ing bits 0-1 of the slot with
    // bits 3-4 spreads every 8-lane group over all 32 banks and only permutes slots inside aligned groups of four, so
    // the contiguous reads stay conflict-free.
    if constexpr (N == 4) {
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        stage[7 * 64 + lane] = cd{inf, inf};
        stage[8 * 64 + lane] = cd{inf, inf};
    }
    auto slot = [](const int s) { return (N & 1) ? s : (s ^ ((s >> 3) & 3)); };
    int wslot[N], rslot[N];
#pragma unroll
    for (int o = 0; o < N; ++o) {
        wslot[o] = slot(lane * N + o);
        rslot[o] = slot(o * 64 + lane);
    }
    // The per-point table entries of chunk jc+1 are fetched while chunk jc is being solved: vmcnt counts
    // loads and stores in issue order, so a load issued AFTER a chunk's stores could only be waited for
    // together with them -- every wavefront would sit out the full write latency once per chunk.
    cd zl_next{1.0, 0.0}, tf_next[N];
    {
        const int j0 = min(jc0 * 64 + lane, ncol - 1);
        zl_next = G.tz[last][j0];
#pragma unroll
        for (int o = 0; o < N; ++o) tf_next[o] = N > 2 ? cd{0.0, 0.0} : G.tf[last][(int64_t)j0 * N + o];   // (n > 2: fetched after the solve)
        // consume them here: loads still pending at the loop entry would be merged into the loop-head
        // state of the waitcnt pass and cost a vmcnt(0) on every iteration
        asm volatile("" ::"v"(zl_next.x), "v"(zl_next.y));
#pragma unroll
        for (int o = 0; o < N; ++o) asm volatile("" ::"v"(tf_next[o].x), "v"(tf_next[o].y));
    }
    // FULL chunks (64 valid points) store unconditionally: with a fixed number of stores per iteration the
    // compiler can wait for the prefetched loads with vmcnt(2N) and leave the stores in flight; any
    // store under a lane condition makes that count unknown and forces vmcnt(0) at the loop head.
    auto chunk = [&](const int jc, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        const cd zl = zl_next;
        cd tfl[N];
#pragma unroll
        for (int o = 0; o < N; ++o) tfl[o] = tf_next[o];
        if (jc + 1 < jc1 && TBK_ABLATE(G.ablate) != 5) {   // (ablate 5: diagnostics, no per-chunk table loads)
            const int jn = min((jc + 1) * 64 + lane, ncol - 1);
            zl_next = G.tz[last][jn];
#pragma unroll
            for (int o = 0; o < N; ++o) tf_next[o] = G.tf[last][(int64_t)jn * N + o];
        }
        SmallMat<N> M;
        if (TBK_ABLATE(G.ablate) == 2) {   // diagnostics: no assembly, no eigen-solve (the store stream alone)
#pragma unroll
            for (int a = 0; a < N; ++a) {
                M.dg[a] = a + zl.x;
#pragma unroll
                for (int b = 0; b < N; ++b) M.v[a][b] = cd{zl.x + a, zl.y + b};
            }
        } else {
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    const cd* Cs = C + slot * npow + pmax;
                    cd acc = Cs[0];
                    cd zp = zl;
                    if constexpr (PM >= 0) {
#pragma unroll
                        for (int p = 1; p <= PM; ++p) {
                            cfma(acc, Cs[p], zp);
                            cfma(acc, Cs[-p], cconj(zp));
                            if (p < PM) zp = cmul(zp, zl);
                        }
                    } else {
                        for (int p = 1; p <= pmax; ++p) {
                            cfma(acc, Cs[p], zp);
                            cfma(acc, Cs[-p], cconj(zp));
                            zp = cmul(zp, zl);
                        }
                    }
                    if (b == a) M.dg[a] = acc.x; else M.up[a][b] = acc;
                }
            }
            init_vectors<N, true>(M);
            if constexpr (N > 2) {
                if (!jacobi_small<N, true>(M) && G.flags) G.flags[0] = 1;
            } else {
                jacobi_small<N, true>(M);
            }
        }
        double sorted[N];
        if constexpr (N > 2) sort_small<N>(M);    // (N <= 2: the closed forms come out ascending)
#pragma unroll
        for (int b = 0; b < N; ++b) sorted[b] = M.dg[b];
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) gmin[b] = fmin(gmin[b], sorted[b + 1] - sorted[b]);
        // eigenvectors of H: D^+ v, periodic-image phases folded into fo
        cd fo[N];
#pragma unroll
        for (int o = 0; o < N; ++o) fo[o] = cmul_x(frow[o], tfl[o]);     // (cmul_x: the image column below repeats these products)
        // LDS-staged store: per band plane the wave owns one contiguous run of
        // 64*N elements, so lanes trade elements through LDS and every store
        // instruction writes 1 KiB of consecutive bytes.
        const int nvalid = TBK_ABLATE(G.ablate) == 1 ? 0 : min(64, ncol - jc * 64) * N;   // (ablate 1: no stores, diagnostics)
        const int64_t point0 = (int64_t)row * nlast + (int64_t)jc * 64;
#pragma unroll
        for (int r = 0; r < N; ++r) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int o = 0; o < N; ++o) {
                const cd val = cmul_x(M.v[o][r], fo[o]);
                stage[wslot[o]] = val;
            }
            asm volatile("" ::: "memory");
            cd* dst = G.wv.data + ((int64_t)r * G.wv.npts + point0) * N;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int e = i * 64 + lane;
                if constexpr (FULL) dst[e] = stage[rslot[i]];
                else if (e < nvalid) dst[e] = stage[rslot[i]];
            }
        }
        if constexpr (!FULL) {
            // the periodic image of column 0 (the row's first chunk takes this variant when there is one to store)
            if (G.img_last && jc == 0 && lane == 0 && TBK_ABLATE(G.ablate) != 1) {
                cd fi[N];
#pragma unroll
                for (int o = 0; o < N; ++o) fi[o] = cmul_x(frow[o], G.tf[last][(int64_t)(nlast - 1) * N + o]);
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    cd* dst = G.wv.data + ((int64_t)r * G.wv.npts + (int64_t)row * nlast + (nlast - 1)) * N;
#pragma unroll
                    for (int o = 0; o < N; ++o) dst[o] = cmul_x(M.v[o][r], fi[o]);
                }
            }
        }
    };
    // ---- n = 3, 4: the same chunk around the FACTORED solver (ql_small_core): (d, Q) sorted, then one band at a time formed,
    // given its orbital phases, staged and stored.  Nothing but the factors is live across the QL iteration: the row's phases sit
    // in scalar registers and the column's table entries are fetched after the solve (the previous chunk's stores were issued
    // a whole eigen-solve ago, so waiting for a load behind them costs nothing by then); only z_last(j), which the assembly
    // needs first thing, keeps the prefetch described above.
    auto chunk_fact = [&](const int jc, auto) __attribute__((always_inline)) {   // (generic: instantiated for N > 2 only)
        SmallFact<N> F;
        {
            const cd zl = zl_next;
            double dg[N];
            cd up[N][N];
            rows_assemble<N, PM>(C, npow, pmax, zl, dg, up);
            double e[N];
            tridiag_small<N, true>(dg, up, F, e);
            // The first reflector (6 doubles at N = 4) is not needed before the bands are formed, and then only for a moment per
            // band: it lives in this wavefront's staging space, [quantity][lane] so that every access is a conflict-free 1 KB row
            if constexpr (N == 4) {
                stage[4 * 64 + lane] = F.us[0][1];
                stage[5 * 64 + lane] = F.us[0][2];
                stage[6 * 64 + lane] = F.us[0][3];
                asm volatile("" ::: "memory");
            }
            if (!ql_iterate_small<N, true>(F, e) && G.flags) G.flags[0] = 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        sort_fact<N>(F);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (N == 4) {                // (the running minima of the three gaps wait in LDS too: 6 registers)
            const cd g01 = stage[7 * 64 + lane], g2x = stage[8 * 64 + lane];
            stage[7 * 64 + lane] = cd{fmin(g01.x, F.d[1] - F.d[0]), fmin(g01.y, F.d[2] - F.d[1])};
            stage[8 * 64 + lane] = cd{fmin(g2x.x, F.d[3] - F.d[2]), 0.0};
        } else {
#pragma unroll
            for (int b = 0; b + 1 < N; ++b) gmin[b] = fmin(gmin[b], F.d[b + 1] - F.d[b]);
        }
        {
            // the orbital phases F = diag(f_o) go INTO the factors: F H_0 H_1 D Q = H_0' H_1' (F D) Q with u' = F u (F is unitary
            // and diagonal) -- 9 products once instead of 16 per chunk on the way out, and no phase is live while the bands go out
            const int jj = min(jc * 64 + lane, ncol - 1);
            cd fo[N];
#pragma unroll
            for (int o = 0; o < N; ++o) fo[o] = G.tf[last][(int64_t)jj * N + o];
            if (G.img_win && jc == G.cpr - 1 && jj == nlast - 1) {   // the image column solved on its own: column 0's phases first
#pragma unroll
                for (int o = 0; o < N; ++o) fo[o] = G.tf0[o];
            }
            if (jc + 1 < jc1) zl_next = G.tz[last][min((jc + 1) * 64 + lane, ncol - 1)];
#pragma unroll
            for (int o = 0; o < N; ++o) fo[o] = cmul_x(frow[o], fo[o]);
#pragma unroll
            for (int K = 0; K + 2 < N; ++K)
#pragma unroll
                for (int r = K + 1; r < N; ++r) {
                    if (N == 4 && K == 0) stage[(3 + r) * 64 + lane] = cmul_x(fo[r], stage[(3 + r) * 64 + lane]);
                    else F.us[K][r] = cmul_x(fo[r], F.us[K][r]);
                }
            F.dph[0] = fo[0];
#pragma unroll
            for (int r = 1; r < N; ++r) F.dph[r] = cmul_x(fo[r], F.dph[r]);
        }
        // ---- ONE form of the store phase for every chunk, each store instruction unconditional (a store under a lane condition
        // makes the number of outstanding stores unknown to the compiler and costs a vmcnt(0) at the loop head, see above):
        //  * a partial last chunk: a lane whose element lies beyond the valid points fetches and stores the LAST valid point's
        //    element instead -- the same bytes to the same address a second time;
        //  * the periodic image of column 0 (the row's first chunk when the whole last axis is in the window): after a band's
        //    transposition lanes 0 .. N-1 of the first 1 KB row hold point 0's components; one more store instruction per band in
        //    which they write them, under the image's column phase, to the image's place while every other lane repeats its
        //    first store.  The branch around it is wave-uniform.
        __builtin_amdgcn_sched_barrier(0);
        const int npv = min(64, ncol - jc * 64);                    // valid points of this chunk
        const int64_t point0 = (int64_t)row * nlast + (int64_t)jc * 64;
        int lane_here = lane;                                        // (formed here, not carried through the eigen-solve)
        asm volatile("" : "+v"(lane_here));
        cd* const stage_w = stage + lane_here * N;
        const int swz = (N & 1) ? 0 : (lane_here >> 1) & 3;          // = ((lane N + o) >> 3) & 3 at N = 4 (see `slot` above)
        cd* const stage_r = stage + ((N & 1) ? lane_here : (lane_here ^ ((lane_here >> 3) & 3)));   // slot(i 64 + lane) - i 64
        // the element row i of a band's transposition hands this lane (re-formed at every use: a handful of integer operations
        // against four registers held through the bands)
        auto eclamp = [&](const int i) {
            const int e = i * 64 + lane_here;
            const int pnt = min(e / N, npv - 1);
            return pnt * N + (e - (e / N) * N);
        };
        const bool img_chunk = G.img_last && jc == 0;               // wave-uniform
        static_for<0, N>([&](auto rt) __attribute__((always_inline)) {
            constexpr int r = decltype(rt)::value;
            __builtin_amdgcn_sched_barrier(0);         // (one band at a time: the next band's vector is not formed early)
            cd z[N];
            if constexpr (N == 4) {
                // column r of H_0' H_1' D' Q with the first reflector read from LDS where it is used -- once for w^+ z, once more
                // for the update: four registers in flight instead of twelve
                asm volatile("" ::: "memory");
#pragma unroll
                for (int c = 0; c < N; ++c) z[c] = cd{F.dph[c].x * F.Q[c][r], F.dph[c].y * F.Q[c][r]};
                {
                    cd w{0.0, 0.0};
#pragma unroll
                    for (int c = 2; c < N; ++c) {
                        w.x = fma(F.us[1][c].y, z[c].y, fma(F.us[1][c].x, z[c].x, w.x));
                        w.y = fma(-F.us[1][c].y, z[c].x, fma(F.us[1][c].x, z[c].y, w.y));
                    }
#pragma unroll
                    for (int c = 2; c < N; ++c) {
                        z[c].x = fma(F.us[1][c].y, w.y, fma(-F.us[1][c].x, w.x, z[c].x));
                        z[c].y = fma(-F.us[1][c].y, w.x, fma(-F.us[1][c].x, w.y, z[c].y));
                    }
                }
                {
                    cd w{0.0, 0.0};
#pragma unroll
                    for (int c = 1; c < N; ++c) {
                        const cd u = stage[(3 + c) * 64 + lane];
                        w.x = fma(u.y, z[c].y, fma(u.x, z[c].x, w.x));
                        w.y = fma(-u.y, z[c].x, fma(u.x, z[c].y, w.y));
                    }
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int c = 1; c < N; ++c) {
                        const cd u = stage[(3 + c) * 64 + lane];
                        z[c].x = fma(u.y, w.y, fma(-u.x, w.x, z[c].x));
                        z[c].y = fma(-u.y, w.x, fma(-u.x, w.y, z[c].y));
                    }
                }
            } else {
                small_vector<N, r>(F, z);
            }
            if (G.img_win && jc == G.cpr - 1) {        // (wave-uniform) ... then x (tf_image conj tf_0), as the lane of column 0 does
                const bool wl = jc * 64 + lane_here >= nlast - 1;
#pragma unroll
                for (int o = 0; o < N; ++o) {
                    const cd zi = cmul_x(z[o], ips[o]);
                    z[o] = cd{wl ? zi.x : z[o].x, wl ? zi.y : z[o].y};
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int o = 0; o < N; ++o) stage_w[o ^ swz] = z[o];
            asm volatile("" ::: "memory");
            cd* dst = G.wv.data + ((int64_t)r * G.wv.npts + point0) * N;
            if (npv == 64) {                           // (wave-uniform; the common case: constant offsets, no index arithmetic)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    dst[i * 64 + lane_here] = stage_r[i * 64];
                    if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);     // (two 1 KB rows in flight between LDS and the store)
                }
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const int ec = eclamp(i);
                    dst[ec] = stage[slot(ec)];
                    if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (img_chunk) {
                const cd first = stage[slot(eclamp(0))];
                // z carries column 0's phase, the image wants its own: tf(image) conj tf(column 0) of component `lane` (< N), the
                // same for every row -- formed once per tile into scalar registers (ips) and picked by lane here
                const bool mine = lane_here < N;
                // (picked with 0 / 1 weights: a chain of selects on lane == o is turned into an indexed read of a scratch copy)
                cd ip{0.0, 0.0};
#pragma unroll
                for (int o = 0; o < N; ++o) {
                    const double wo = lane_here == o ? 1.0 : 0.0;
                    ip.x = fma(wo, ips[o].x, ip.x);
                    ip.y = fma(wo, ips[o].y, ip.y);
                }
                const cd vi = cmul_x(first, ip);
                cd* di = mine ? G.wv.data + ((int64_t)r * G.wv.npts + (int64_t)row * nlast + (nlast - 1)) * N + lane_here : dst + eclamp(0);
                *di = cd{mine ? vi.x : first.x, mine ? vi.y : first.y};
            }
        });
    };
    auto run_chunk = [&](const int jc, auto full_tag) __attribute__((always_inline)) {
        if constexpr (N > 2) {
            if (TBK_ABLATE(G.ablate) == 0) {
                chunk_fact(jc, 0);
                return;
            }
        }
        chunk(jc, full_tag);
    };
    if constexpr (N > 2) {
        if (TBK_ABLATE(G.ablate) == 0) {
            for (int jc = jc0; jc < jc1; ++jc) chunk_fact(jc, 0);
            jc0 = jc1;
        }
    }
    const int jfull = TBK_ABLATE(G.ablate) == 1 ? jc0 : max(jc0, min(jc1, ncol / 64));   // chunks [jc0, jfull) are complete
    int jc = jc0;
    if (G.img_last && jc0 == 0 && jc < jc1) {      // (the chunk that also stores the image: the variant with conditional stores)
        run_chunk(jc, std::false_type{});
        ++jc;
    }
    for (; jc < jfull; ++jc) run_chunk(jc, std::true_type{});
    for (; jc < jc1; ++jc) run_chunk(jc, std::false_type{});
    if constexpr (N > 1) {
        // min gaps: one plain store per tile and band pair, reduced when the result is asked for.  (A guarded
        // atomicMin here needs the guard's value: loaded at the end it keeps the wavefront from retiring for
        // a memory latency -- 3.8 us of the 2048^2 solve; loaded at set-up it is still +inf for every early
        // wavefront and 10^4 atomics pile up on 64 addresses -- 71 us.)
        if constexpr (N == 4) {
            if (TBK_ABLATE(G.ablate) == 0) {
                const cd g01 = stage[7 * 64 + lane], g2x = stage[8 * 64 + lane];
                gmin[0] = g01.x;
                gmin[1] = g01.y;
                gmin[2] = g2x.x;
            }
        }
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) {
            double g = gmin[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) g = fmin(g, __shfl_xor(g, off));
            if (lane == 0) G.gap_part[tile * (N - 1) + b] = g;
        }
    }
}

// ---- eigenvalues of the model on k_uniform_mesh(mesh) (solve_all on the mesh the model itself generated, configs[1]): the
// tile scheme and the row-polynomial assembly of k_grid_rows with nothing but the eigenvalues going out -- eval[band][point],
// 8 bytes per lane and band, coalesced.  No k list is read (the list kernel read 8 d bytes per point of a list the device had
// just generated) and no exp(2 pi i x) is evaluated per point (the per-axis tables).  k_d = i / N_d comes out of the tables as
// the window [0, N_d) of a global mesh of N_d + 1 points anchored at 0: (double)g / (double)N_d, the generator's own expression.
template <int N, int PM>
__global__ __launch_bounds__(256) void k_mesh_evals(const ModelView mv, const GridArgs G, double* __restrict__ eval) {
    extern __shared__ __align__(16) unsigned char lds_rows[];
    constexpr int NSLOT = N * (N + 1) / 2;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pmax = PM >= 0 ? PM : mv.pmax;
    const int npow = 2 * pmax + 1;
    const int ncell = NSLOT * npow;
    cd* C = reinterpret_cast<cd*>(lds_rows) + wib * ncell;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wib;
    const bool live = tile < G.ntiles;
    const int last = G.last;
    const int nlast = G.wv.mesh[last];
    unsigned row = 0;
    int jc0 = 0, jc1 = 0;
    if (live) {
        row = (unsigned)(tile / G.tpr);
        const int ts = (int)(tile - (int64_t)row * G.tpr);
        jc0 = ts * G.seg;
        jc1 = min(jc0 + G.seg, G.cpr);
        cd z[4] = {cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}};
        unsigned rem = row;
#pragma unroll
        for (int d = 2; d >= 0; --d) {
            if (d < last) {
                const unsigned md = (unsigned)G.wv.mesh[d];
                const unsigned q = rem / md;
                z[d] = G.tz[d][rem - q * md];
                rem = q;
            }
        }
        for (int cell = lane; cell < ncell; cell += 64) {
            const int t0 = mv.cell_ptr[cell], t1 = mv.cell_ptr[cell + 1];
            cd acc{0.0, 0.0};
            for (int t = t0; t < t1; ++t) {
                int4 R = mv.term_R[t];
                if (last == 0) R.x = 0; else if (last == 1) R.y = 0; else if (last == 2) R.z = 0; else R.w = 0;
                cfma(acc, mv.term_amp[t], phase_of_R(z, R));
            }
            C[cell] = acc;
        }
    }
    __syncthreads();
    if (!live) return;
    for (int jc = jc0; jc < jc1; ++jc) {
        const int j = jc * 64 + lane;
        const cd zl = G.tz[last][min(j, nlast - 1)];
        double evs[N];
        if constexpr (N > 2) {
            // n = 3, 4: the factored solver without its vectors -- (d, e) only -- on the cells assembled two slots at a time
            // (rows_assemble; the old form held all the cells' broadcast reads in flight and sorted an eigenvector matrix it never
            // computed: 192 registers, two wavefronts per SIMD, for a kernel that stores 8 bytes per band)
            double dg[N];
            cd up[N][N];
            rows_assemble<N, PM>(C, npow, pmax, zl, dg, up);
            SmallFact<N> F;
            if (!ql_small_core<N, false>(dg, up, F) && G.flags) G.flags[0] = 1;
#pragma unroll
            for (int b = 0; b < N; ++b) evs[b] = F.d[b];
            auto cx = [&](const int i, const int k) {
                const double lo = fmin(evs[i], evs[k]), hi = fmax(evs[i], evs[k]);
                evs[i] = lo;
                evs[k] = hi;
            };
            if constexpr (N == 3) {
                cx(0, 1);
                cx(1, 2);
                cx(0, 1);
            } else {
                cx(0, 1);
                cx(2, 3);
                cx(0, 2);
                cx(1, 3);
                cx(1, 2);
            }
        } else {
            SmallMat<N> M;
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    const cd* Cs = C + slot * npow + pmax;
                    cd acc = Cs[0];
                    cd zp = zl;
                    if constexpr (PM >= 0) {
#pragma unroll
                        for (int p = 1; p <= PM; ++p) {
                            cfma(acc, Cs[p], zp);
                            cfma(acc, Cs[-p], cconj(zp));
                            if (p < PM) zp = cmul(zp, zl);
                        }
                    } else {
                        for (int p = 1; p <= pmax; ++p) {
                            cfma(acc, Cs[p], zp);
                            cfma(acc, Cs[-p], cconj(zp));
                            zp = cmul(zp, zl);
                        }
                    }
                    if (b == a) M.dg[a] = acc.x; else M.up[a][b] = acc;
                }
            }
            jacobi_small<N, false>(M);
#pragma unroll
            for (int b = 0; b < N; ++b) evs[b] = M.dg[b];
        }
        if (j < nlast) {
            const int64_t point = (int64_t)row * nlast + j;
#pragma unroll
            for (int b = 0; b < N; ++b) eval[(int64_t)b * G.wv.npts + point] = evs[b];
        }
    }
}

// eigenvalues of `m` on k_uniform_mesh(mesh) into e_dev[nsta][nk] by k_mesh_evals; *done = false (and nothing launched) where it
// does not apply (more than 4 states, a cell table beyond 48 KB of LDS, TBK_MESH_ROWS=0): the caller solves the generated list
int tbk_mesh_evals_rows(tbk_model* m, const int32_t* mesh, double* e_dev, bool* done) {
    *done = false;
    tbk_ctx* ctx = m->ctx;
    const int n = m->nsta, D = m->dim_k;
    if (n < 1 || n > 4 || D < 1 || D > 3 || tbk_knobs().mesh_rows == 0) return TBK_OK;
    const size_t lds = (size_t)4 * (n * (n + 1) / 2) * (2 * m->view.pmax + 1) * sizeof(cd);
    if (lds > 48 * 1024) return TBK_OK;
    GridArgs G{};
    WfsView& v = G.wv;
    v.dim_arr = D;
    v.nsta = n;
    v.ncomp = n;
    v.npts = 1;
    int64_t ntab = 0;
    for (int d = 0; d < D; ++d) {
        v.mesh[d] = mesh[d];
        G.gmesh[d] = mesh[d] + 1;              // k_d = g / N_d: the first N_d points of a global mesh of N_d + 1 anchored at 0
        G.off[d] = 0;
        G.start_k[d] = 0.0;
        v.npts *= mesh[d];
        ntab += mesh[d];
    }
    for (int d = D; d < TBK_MAX_DIM; ++d) v.mesh[d] = 1;
    if (v.npts / mesh[D - 1] >= (int64_t)0xffffffffu) return TBK_OK;
    // per-axis tables in the context's work area (tz | tf); tf (orbital phases) is written by k_grid_tables and not used here
    const size_t tbytes = (size_t)ntab * (1 + n) * sizeof(cd) + (size_t)TBK_MAX_DIM * n * sizeof(cd);
    if (tbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, std::max(tbytes, (size_t)1 << 20));
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "mesh tables of %zu bytes: %s", tbytes, hipGetErrorString(e));
        ctx->work_bytes = std::max(tbytes, (size_t)1 << 20);
    }
    cd* tab = (cd*)ctx->work;
    cd* pbc = tab + ntab * (1 + n);
    TBK_HIP(hipMemsetAsync(pbc, 0, (size_t)TBK_MAX_DIM * n * sizeof(cd), ctx->stream));
    G.pbc = pbc;
    {
        int64_t zo = 0, fo = ntab;
        for (int d = 0; d < D; ++d) {
            G.tz[d] = tab + zo;
            G.tf[d] = tab + fo;
            zo += mesh[d];
            fo += (int64_t)mesh[d] * n;
        }
    }
    G.flags = ctx->flags_dev;
    G.last = D - 1;
    G.cpr = (mesh[D - 1] + 63) / 64;
    const int64_t nrows = v.npts / mesh[D - 1];
    G.nchunks = nrows * G.cpr;
    const int64_t want = (int64_t)ctx->cus * 32;
    G.seg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, G.cpr), G.nchunks / want));
    G.tpr = (G.cpr + G.seg - 1) / G.seg;
    G.seg = (G.cpr + G.tpr - 1) / G.tpr;
    G.ntiles = nrows * G.tpr;
    {
        ProfScope ps(ctx, "grid_tables");
        hipLaunchKernelGGL(k_grid_tables, dim3((unsigned)((ntab + 255) / 256)), dim3(256), 0, ctx->stream, m->view, G, tab, tab + ntab);
    }
    const unsigned blocks = (unsigned)((G.ntiles + 3) / 4);
    const int pm = m->view.pmax;
    {
        ProfScope ps(ctx, "mesh_evals");
#define TBK_MEV(NN, PP) hipLaunchKernelGGL((k_mesh_evals<NN, PP>), dim3(blocks), dim3(256), lds, ctx->stream, m->view, G, e_dev)
        switch (n) {
            case 1: if (pm == 0) TBK_MEV(1, 0); else if (pm == 1) TBK_MEV(1, 1); else TBK_MEV(1, -1); break;
            case 2: if (pm == 0) TBK_MEV(2, 0); else if (pm == 1) TBK_MEV(2, 1); else TBK_MEV(2, -1); break;
            case 3: if (pm == 1) TBK_MEV(3, 1); else TBK_MEV(3, -1); break;
            default: if (pm == 1) TBK_MEV(4, 1); else TBK_MEV(4, -1); break;
        }
#undef TBK_MEV
    }
    TBK_HIP(hipGetLastError());
    *done = true;
    return TBK_OK;
}

// ---------------------------------------------------------------------------
// S(k) of one point (or the supplied matrix, MODE 2) into the LDS matrix A[n][ld], by the NT threads of a
// workgroup; `ph` is scratch for max(nR, 1) phases.  The last writes are NOT followed by a barrier.
// ---------------------------------------------------------------------------
// TRI: only the upper triangle, packed by columns -- entry (a, b), a <= b, at b (b + 1) / 2 + a (k_hh32, whose LDS sets its occupancy).
template <int MODE, int NT, bool TRI = false>
__device__ __forceinline__ void assemble_lds(const ModelView& mv, const ListArgs& L, const int64_t id, const double (&kk)[4],
                                             cd* __restrict__ A, const int ld, cd* __restrict__ ph, const int lane) {
    const int n = mv.nsta;
    if constexpr (MODE == 2) {
        const cd* h = L.ham + id * (int64_t)n * n;
        for (int e = lane; e < n * n; e += NT) {
            const int a = e / n, b = e - a * n;
            // use the upper triangle, mirror it (the reference's eigh reads one triangle)
            cd v = a <= b ? h[a * n + b] : cconj(h[b * n + a]);
            if (a == b) v.y = 0.0;
            if constexpr (TRI) {
                if (a <= b) A[b * (b + 1) / 2 + a] = v;
            } else {
                A[a * ld + b] = v;
            }
        }
    } else {
        cd z[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        if (mv.nR > 0) {
            // dense model: S_slot = sum_R U_R[slot] e^{2 pi i k.R}.  One lane per phase, then every
            // slot is nR independent, coalesced coefficient loads (the table stays in L2)
            for (int r = lane; r < mv.nR; r += NT) ph[r] = phase_of_R(z, mv.rvec[r]);
            __syncthreads();
            for (int slot = lane; slot < mv.nslot; slot += NT) {
                const int ab = mv.slot_ab[slot];
                const int a = ab & 0xffff, b = ab >> 16;
                const cd* u = mv.rblock + slot;
                cd acc{0.0, 0.0};
#pragma unroll 4
                for (int r = 0; r < mv.nR; ++r) cfma(acc, u[(size_t)r * mv.nslot], ph[r]);
                if constexpr (TRI) {
                    if (a == b) A[a * (a + 1) / 2 + a] = cd{acc.x, 0.0};
                    else if (a < b) A[b * (b + 1) / 2 + a] = acc;
                    else A[a * (a + 1) / 2 + b] = cconj(acc);
                } else if (a == b) {
                    A[a * ld + a] = cd{acc.x, 0.0};
                } else {
                    A[a * ld + b] = acc;
                    A[b * ld + a] = cconj(acc);
                }
            }
            __syncthreads();   // phases consumed before the caller reuses `ph`
        } else {
            // sparse model (ribbons, slabs): clear A, then walk the non-empty slots only
            for (int e = lane; e < (TRI ? n * (n + 1) / 2 : n * ld); e += NT) A[e] = cd{0.0, 0.0};
            __syncthreads();
            for (int i = lane; i < mv.nnz; i += NT) {
                const int4 s = mv.nz[i];
                const int a = s.x & 0xffff, b = s.x >> 16;
                cd acc{0.0, 0.0};
                for (int t = s.y; t < s.z; ++t) cfma(acc, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
                if constexpr (TRI) {
                    if (a == b) A[a * (a + 1) / 2 + a] = cd{acc.x, 0.0};
                    else if (a < b) A[b * (b + 1) / 2 + a] = acc;
                    else A[a * (a + 1) / 2 + b] = cconj(acc);
                } else if (a == b) {
                    A[a * ld + a] = cd{acc.x, 0.0};
                } else {
                    A[a * ld + b] = acc;
                    A[b * ld + a] = cconj(acc);
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
    cd* T;      // [n][ld]  scratch of the warm-start similarity transform
    cd* rot;   // [n/2+1]  (c, -) and sw packed: rot[2i] = {c, 0}, rot[2i+1] = sw
    int* pq;    // [n/2+1]  p | q<<16 (p<q), -1 for the bye
    double* ev; // [n]
    int* perm;  // [n]  perm[rank] = column
    cd* eo;     // [n]  conj(e_o) * pbc phases
    double* red; // [8]  cross-wavefront reduction scratch (workgroup-per-matrix form)
};

// NT = 64: one wavefront per matrix, matrices in LDS.  NT = 256 (n = 65..256): one
// workgroup per matrix, matrices in a global workspace (L2-resident), always cold-started.
template <int MODE, bool VEC, int NT, bool LDSWS = (NT == 64)>
__global__ __launch_bounds__(NT) void k_solve_wave(const ModelView mv, const int64_t nk,
                                                   const ListArgs L, const GridArgs G,
                                                   int* noconv_flag, const int run, cd* work, const int warm) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = mv.nsta;
    const int ld = n + 1;
    const int lane = threadIdx.x;
    const int m = (n + 1) & ~1;  // players in the round-robin (bye if n odd)
    const int half = m >> 1;
    // fixed lane -> (fast index c0, slow start i0) map for the n x (n or n/2) element
    // walks of the sweeps: no divisions inside the rounds
    const int c0 = lane % n, i0 = lane / n, istep = NT / n > 0 ? NT / n : 1;
    const bool walker = lane < n * istep;
    WaveLds S;
    if constexpr (NT == 64) {
        S.A = (cd*)lds_raw;
        S.Vt = S.A + n * ld;
        S.T = S.Vt + n * ld;                 // present only when runs are longer than one point
        S.rot = run > 1 ? S.T + n * ld : S.T;
    } else if constexpr (LDSWS) {           // a whole workgroup on one LDS-resident matrix (few matrices: latency)
        S.A = (cd*)lds_raw;
        S.Vt = S.A + n * ld;
        S.T = S.Vt + n * ld;                 // present only when the workgroup warm-starts along its run
        S.rot = warm ? S.T + n * ld : S.T;
    } else {
        S.A = work + (size_t)blockIdx.x * 2 * n * ld;
        S.Vt = S.A + n * ld;
        S.T = S.Vt;                          // unused: workgroup-per-matrix solves start cold
        S.rot = (cd*)lds_raw;
    }
    S.eo = S.rot + 2 * half;                 // doubles as the per-R phase table during assembly
    S.ev = (double*)(S.eo + (n > mv.nR ? n : mv.nR));
    S.red = S.ev + n;
    S.pq = (int*)(S.red + 32);              // red: two doubles per wavefront of the workgroup (<= 16)
    S.perm = S.pq + half;

    // A wavefront walks a short chain of consecutive points.  After the first one, Jacobi is
    // WARM STARTED: the new matrix is first rotated into the previous point's eigenbasis
    // (A <- V^+ A V, two n^3 products through LDS), where it is already nearly diagonal for
    // neighbouring k -- 2-3 sweeps instead of 7-9.  V keeps accumulating the rotations; chains
    // are short so its orthonormality drift stays at the 1e-15 level.
    //
    // The eigenvector gauge depends on the chain, so for a mesh the chain of a point must be
    // a function of the point alone: chains are the aligned blocks [c*run, (c+1)*run) of the
    // GLOBAL index along the last axis, the periodic image (global index N-1) always starts
    // cold (exactly like index 0, whose k it shares), and a window that begins inside a block
    // recomputes the block's earlier points without storing them.  Periodic images, halo rows
    // and shard windows therefore reproduce the unsharded array bit for bit.
    int64_t it_begin = 0, it_end = 0, grow = 0;
    const int nlast = MODE == 1 ? G.wv.mesh[G.last] : 1;
    const int64_t off_last = MODE == 1 ? G.off[G.last] : 0;
    const int gnl = MODE == 1 ? G.gmesh[G.last] : 1;
    if constexpr (MODE == 1) {
        grow = blockIdx.x / G.wnchunk;
        const int64_t c = G.wcfirst + (blockIdx.x - grow * G.wnchunk);
        it_begin = c * run;                                  // global last-axis indices
        it_end = it_begin + run < gnl ? it_begin + run : gnl;
        if (it_end > off_last + nlast) it_end = off_last + nlast;
    } else {
        it_begin = (int64_t)blockIdx.x * run;                // positions in the k list
        it_end = it_begin + run < nk ? it_begin + run : nk;
    }
    for (int64_t it = it_begin; it < it_end; ++it) {
        bool cold = it == it_begin || !warm;
        bool store = true;
        int64_t id = it;
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
        bool wrap[4] = {false, false, false, false};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
        } else if constexpr (MODE == 1) {
            cold = cold || it == gnl - 1;
            store = it >= off_last;
            id = grow * nlast + (it - off_last);             // local row-major point (valid when store)
            grid_point_rowcol(G, grow, it, kk, wrap);
        }
        __syncthreads();  // previous matrix fully written out before LDS is reused
        // ---- assemble S(k) (or load the supplied matrix) into A, V^T = I
        assemble_lds<MODE, NT>(mv, L, id, kk, S.A, ld, S.eo, lane);
        if (cold) {
            if (walker)
                for (int a = i0; a < n; a += istep) S.Vt[a * ld + c0] = cd{a == c0 ? 1.0 : 0.0, 0.0};
        } else {
            __syncthreads();
            // T[o][c] = sum_p A[o][p] V[p][c]       (Vt[c][p] = V[p][c])
            if (walker)
                for (int o = i0; o < n; o += istep) {
                    cd acc{0.0, 0.0};
                    for (int p = 0; p < n; ++p) cfma(acc, S.A[o * ld + p], S.Vt[c0 * ld + p]);
                    S.T[o * ld + c0] = acc;
                }
            __syncthreads();
            // A'[b][c] = sum_o conj(V[o][b]) T[o][c]
            if (walker)
                for (int b = i0; b < n; b += istep) {
                    cd acc{0.0, 0.0};
                    for (int o = 0; o < n; ++o) cfmac(acc, S.Vt[b * ld + o], S.T[o * ld + c0]);
                    if (b == c0) acc.y = 0.0;
                    S.A[b * ld + c0] = acc;
                }
        }
        if (VEC) {
            if (lane < n) {
                cd f{1.0, 0.0};
                if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[lane])));
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (wrap[d]) f = cmul(f, G.pbc[d * n + lane]);
                }
                S.eo[lane] = f;
            }
        }
        __syncthreads();

        // ---- parallel-ordered Jacobi sweeps
        bool converged = false;
        const int sweep_cap = TBK_ABLATE(G.ablate) >= 10 ? TBK_ABLATE(G.ablate) - 10 : TBK_JACOBI_MAX_SWEEPS;   // diagnostics: TBK_ABLATE_GRID=10+k caps the sweeps at k
        for (int sweep = 0; sweep < sweep_cap; ++sweep) {
            double off = 0.0, dia = 0.0;
            if (walker)
                for (int a = i0; a < n; a += istep) {
                    const double v2 = cabs2(S.A[a * ld + c0]);
                    if (a == c0) dia += v2; else off += v2;
                }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                off += __shfl_xor(off, o);
                dia += __shfl_xor(dia, o);
            }
            if constexpr (NT > 64) {   // combine the wavefronts of the workgroup (fixed order)
                __syncthreads();
                if ((lane & 63) == 0) {
                    S.red[2 * (lane >> 6)] = off;
                    S.red[2 * (lane >> 6) + 1] = dia;
                }
                __syncthreads();
                off = 0.0;
                dia = 0.0;
                for (int wv = 0; wv < NT / 64; ++wv) {
                    off += S.red[2 * wv];
                    dia += S.red[2 * wv + 1];
                }
            }
            if (off <= 2.0e-32 * (dia + off)) {
                converged = true;
                break;
            }
            for (int round = 0; round < m - 1; ++round) {
                // pairing of round `round`: player m-1 fixed (the bye when n is odd), the others rotate
                if (lane < half) {
                    int p, q;
                    if (lane == 0) {
                        p = m - 1;
                        q = round;
                    } else {
                        p = (round + lane) % (m - 1);
                        q = (round - lane + (m - 1)) % (m - 1);
                    }
                    double c = 1.0;
                    cd sw{0.0, 0.0};
                    if (p < n) {  // not the bye
                        const cd g = S.A[p * ld + q];
                        const double g2 = cabs2(g);
                        if (g2 > 0.0) {   // same division-free parameters as rotate<>
                            const double a = 0.5 * (S.A[q * ld + q].x - S.A[p * ld + p].x), aa = fabs(a);
                            const double r = sqrt(a * a + g2);
                            const double inv = rsqrt(2.0 * r * (r + aa));
                            const double sg = copysign(1.0, a);
                            c = (aa + r) * inv;
                            sw = cd{sg * g.x * inv, sg * g.y * inv};
                        }
                    }
                    S.pq[lane] = p | (q << 16);
                    S.rot[2 * lane] = cd{c, 0.0};
                    S.rot[2 * lane + 1] = sw;
                }
                __syncthreads();
                // A <- J^+ A J, one 2x2 block {p_i,q_i} x {p_j,q_j} at a time: the blocks of a round are
                // disjoint, so every element is read and written once, in place
                for (int blk = lane; blk < half * half; blk += NT) {
                    const int i = blk / half, j = blk - i * half;
                    const int ci_ = S.pq[i], cj_ = S.pq[j];
                    const int pi = ci_ & 0xffff, qi = ci_ >> 16, pj = cj_ & 0xffff, qj = cj_ >> 16;
                    const bool vi = pi < n, vj = pj < n;   // false: the bye "player"
                    const double ci = S.rot[2 * i].x, cj = S.rot[2 * j].x;
                    const cd si = S.rot[2 * i + 1], sj = S.rot[2 * j + 1];
                    const cd zero{0.0, 0.0};
                    cd x00 = vi && vj ? S.A[pi * ld + pj] : zero;
                    cd x01 = vi ? S.A[pi * ld + qj] : zero;
                    cd x10 = vj ? S.A[qi * ld + pj] : zero;
                    cd x11 = S.A[qi * ld + qj];
                    {   // columns:  a'_rp = c a_rp - conj(s) a_rq ;  a'_rq = s a_rp + c a_rq
                        const cd a = x00, b = x01;
                        x00 = cd{cj * a.x - (sj.x * b.x + sj.y * b.y), cj * a.y - (sj.x * b.y - sj.y * b.x)};
                        x01 = cd{(sj.x * a.x - sj.y * a.y) + cj * b.x, (sj.x * a.y + sj.y * a.x) + cj * b.y};
                    }
                    {
                        const cd a = x10, b = x11;
                        x10 = cd{cj * a.x - (sj.x * b.x + sj.y * b.y), cj * a.y - (sj.x * b.y - sj.y * b.x)};
                        x11 = cd{(sj.x * a.x - sj.y * a.y) + cj * b.x, (sj.x * a.y + sj.y * a.x) + cj * b.y};
                    }
                    {   // rows:  a'_pc = c a_pc - s a_qc ;  a'_qc = conj(s) a_pc + c a_qc
                        const cd a = x00, b = x10;
                        x00 = cd{ci * a.x - (si.x * b.x - si.y * b.y), ci * a.y - (si.x * b.y + si.y * b.x)};
                        x10 = cd{(si.x * a.x + si.y * a.y) + ci * b.x, (si.x * a.y - si.y * a.x) + ci * b.y};
                    }
                    {
                        const cd a = x01, b = x11;
                        x01 = cd{ci * a.x - (si.x * b.x - si.y * b.y), ci * a.y - (si.x * b.y + si.y * b.x)};
                        x11 = cd{(si.x * a.x + si.y * a.y) + ci * b.x, (si.x * a.y - si.y * a.x) + ci * b.y};
                    }
                    if (i == j) {   // the rotated pair itself: exactly diagonal, real
                        x01 = zero;
                        x10 = zero;
                        x00.y = 0.0;
                        x11.y = 0.0;
                    }
                    if (vi && vj) S.A[pi * ld + pj] = x00;
                    if (vi) S.A[pi * ld + qj] = x01;
                    if (vj) S.A[qi * ld + pj] = x10;
                    S.A[qi * ld + qj] = x11;
                }
                // V <- V J (rows p, q of V^T); V is kept in every mode (it is the next point's basis)
                for (int i = walker ? i0 : half; i < half; i += istep) {
                    const int cidx = c0;
                    const int code = S.pq[i];
                    const int p = code & 0xffff, q = code >> 16;
                    if (p >= n) continue;
                    const double c = S.rot[2 * i].x;
                    const cd sw = S.rot[2 * i + 1];
                    const cd x = S.Vt[p * ld + cidx], y = S.Vt[q * ld + cidx];
                    S.Vt[p * ld + cidx] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                    S.Vt[q * ld + cidx] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
                }
                __syncthreads();
            }
        }
        if (!converged && lane == 0 && TBK_ABLATE(G.ablate) < 10) atomicExch(noconv_flag, 1);

        // ---- order eigenvalues (stable ascending), write out
        if (lane < n) S.ev[lane] = S.A[lane * ld + lane].x;
        __syncthreads();
        if (lane < n) {
            const double mine = S.ev[lane];
            int r = 0;
            for (int j = 0; j < n; ++j) r += tbk_before(S.ev[j], mine, j, lane) ? 1 : 0;
            S.perm[r] = lane;
        }
        __syncthreads();
        if (!store) continue;   // a chain predecessor outside the window: only its basis was needed
        if constexpr (MODE == 1) {
            if (lane + 1 < n) {
                const double g = S.ev[S.perm[lane + 1]] - S.ev[S.perm[lane]];
                unsigned long long* slot = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + lane;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(g, 0.0));
                if (bits < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, bits);
            }
            for (int e = lane; e < n * n; e += NT) {
                const int rb = e / n, o = e - rb * n;
                wf_at(G.wv, rb, id)[o] = cmul(S.Vt[S.perm[rb] * ld + o], S.eo[o]);
            }
        } else {
            if (lane < n) L.eval[(int64_t)lane * nk + id] = S.ev[S.perm[lane]];
            if (VEC) {
                for (int e = lane; e < n * n; e += NT) {
                    const int rb = e / n, o = e - rb * n;
                    L.evec[((int64_t)rb * nk + id) * n + o] = cmul(S.Vt[S.perm[rb] * ld + o], S.eo[o]);
                }
            }
        }
    }
}

static size_t wave_lds_bytes(int n, bool with_t, int nR = 0) {
    const int ld = n + 1, half = (n + 1) / 2;
    size_t b = (size_t)(with_t ? 3 : 2) * n * ld * sizeof(cd);  // A, Vt (, T)
    b += (size_t)2 * half * sizeof(cd);          // rot
    b += (size_t)std::max(n, nR) * sizeof(cd);   // eo (also the per-R phases during assembly)
    b += (size_t)(n + 32) * sizeof(double);      // ev, red
    b += (size_t)(half + n) * sizeof(int);       // pq, perm
    return (b + 15) & ~(size_t)15;
}

// ---------------------------------------------------------------------------
// _gen_ham parity hook: H_ab(k) = conj(e_a) e_b S_ab(k), one thread per (k,slot)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gen_ham_entry(const ModelView& mv, const int64_t nk, const double* __restrict__ k,
                                              cd* __restrict__ ham, const int64_t idx) {
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
__global__ __launch_bounds__(256) void k_gen_ham(const ModelView mv, const int64_t nk,
                                                 const double* __restrict__ k, cd* __restrict__ ham, const DoneArgs done) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < nk * mv.nslot) gen_ham_entry(mv, nk, k, ham, idx);
    if (done.word) {                 // (small calls: H(k) lies in mapped host memory and the host polls the completion word)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) tbk_signal_done(done);
    }
}

// min over the per-tile partial gaps of k_grid_rows: one 1024-thread workgroup per band pair
__global__ __launch_bounds__(1024) void k_gap_part_reduce(const double* __restrict__ part, const int64_t ntiles, const int ng,
                                                          double* __restrict__ out, const DoneArgs done) {
    const int b = blockIdx.x;
    // (twelve loads in flight per thread: the 10 245 tiles of a 2048^2 mesh are ONE round of loads, not three dependent ones --
    // this kernel is pure latency and sits in front of every solve_on_grid return)
    double g0 = INFINITY, g1 = INFINITY, g2 = INFINITY, g3 = INFINITY;
    int64_t t = threadIdx.x;
    for (; t < ntiles; t += 12 * 1024) {
        double v[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int64_t idx = t + (int64_t)i * 1024;
            v[i] = idx < ntiles ? part[idx * ng + b] : INFINITY;
        }
#pragma unroll
        for (int i = 0; i < 12; i += 4) {
            g0 = fmin(g0, v[i]);
            g1 = fmin(g1, v[i + 1]);
            g2 = fmin(g2, v[i + 2]);
            g3 = fmin(g3, v[i + 3]);
        }
    }
    double g = fmin(fmin(g0, g1), fmin(g2, g3));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g = fmin(g, __shfl_xor(g, off));
    __shared__ double red[16];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = red[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = fmin(m, red[i]);
        out[b] = fmax(m, 0.0);
        tbk_signal_done(done);      // (the call's last kernel: the host may be polling the completion word)
    }
}

__global__ void k_arm_gaps(unsigned long long* p, const int n) {
    for (int i = threadIdx.x; i < TBK_GAP_SHARDS * n; i += blockDim.x) p[i] = 0x7ff0000000000000ull;
}

#include "tbk_solve_big.inl"   // n > 256: one kernel launch per Jacobi round, whole chip per batch
#include "tbk_solve_reg.inl"   // n = 5..8: register-resident cyclic Jacobi, 1/2/4 lanes per matrix
#include "tbk_solve_row16.inl" // n = 15, 16 on lists: one DPP row of 16 lanes per matrix, rows of A in registers
#include "tbk_solve_ql16.inl"  // n = 9..16: Householder + implicit QL in registers, one DPP row of 16 lanes per matrix
#include "tbk_solve_tw16.inl"  // n = 9..16 with eigenvectors, large batches: tridiagonalise | eigenvalues | twisted-factorisation vectors (MFMA Newton-Schulz) + back-transformation
#include "tbk_solve_fused.inl" // 2-D meshes, n = 2 / 4: solve_on_grid and berry_flux in one pass (the plaquette phases from registers)
#include "tbk_solve_trig.inl"  // eigenvalues only, n = 65..1024: Householder with A in L2, then one thread per eigenvalue (bisection)
#include "tbk_solve_trigv.inl" // eigenvectors for n = 65..1024 by the direct method: tridiagonalise (reflectors kept) | bisection | twisted-factorisation vectors | Newton-Schulz | back-transformation
#include "tbk_solve_qlw.inl"   // n = 17..64, large batches: Householder in LDS, lane-per-matrix QL, rotation replay
#include "tbk_solve_blk.inl"   // batches of wide matrices: block Jacobi, 16x16 subproblems through k_solve_row16

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
template <int MODE, bool VEC>
static int launch_small(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L) {
    const unsigned blocks = (unsigned)((nk + 255) / 256);
    // LDS image of the R-grouped table (k lists of n <= 4 states: rtable_stage)
    const size_t rlds = MODE == 0 && mv.nR > 0 ? (size_t)mv.nR * (sizeof(int4) + (size_t)(n * (n + 1) / 2) * sizeof(cd)) : 0;
    // eigenvalue-only k lists that fill the chip more than once: two points per lane (k_solve_small_multi; 2^20 Haldane points
    // 21.8 -> 19.9 us; with eigenvectors the second point's registers cost more than the shared scalar stream saves: 30.1 ->
    // 31.2 us, so those keep one point per lane).  TBK_SMALL_KPT=1 / 2 forces either form where both exist.
    if constexpr (MODE == 0) {
        const int kpt = tbk_knobs().small_kpt;
        const bool many = kpt == 2 || (kpt < 0 && !VEC && nk >= ((int64_t)1 << 19));
        if (many && (n == 2 || (!VEC && (n == 3 || n == 4)))) {
            const unsigned b2 = (unsigned)((nk + 511) / 512);
            if (n == 2) hipLaunchKernelGGL((k_solve_small_multi<2, VEC, 2>), dim3(b2), dim3(256), rlds, ctx->stream, mv, nk, L);
            else if (n == 3) hipLaunchKernelGGL((k_solve_small_multi<3, false, 2>), dim3(b2), dim3(256), rlds, ctx->stream, mv, nk, L);
            else hipLaunchKernelGGL((k_solve_small_multi<4, false, 2>), dim3(b2), dim3(256), rlds, ctx->stream, mv, nk, L);
            if (L.done.word) hipLaunchKernelGGL(k_signal_only, dim3(1), dim3(64), 0, ctx->stream, L.done);   // (TBK_SMALL_KPT=2 on a small call)
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
    }
    switch (n) {
        case 1: hipLaunchKernelGGL((k_solve_small<1, MODE, VEC>), dim3(blocks), dim3(256), rlds, ctx->stream, mv, nk, L); break;
        case 2: hipLaunchKernelGGL((k_solve_small<2, MODE, VEC>), dim3(blocks), dim3(256), rlds, ctx->stream, mv, nk, L); break;
        case 3: hipLaunchKernelGGL((k_solve_small<3, MODE, VEC>), dim3(blocks), dim3(256), rlds, ctx->stream, mv, nk, L); break;
        case 4: hipLaunchKernelGGL((k_solve_small<4, MODE, VEC>), dim3(blocks), dim3(256), rlds, ctx->stream, mv, nk, L); break;
        default: tbk_set_error("launch_small: n=%d", n); return TBK_EINVAL;
    }
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// ---------------------------------------------------------------------------
// Which solver takes a batch: ONE table.  A row is (states, eigenvectors?, input form, batch window) -> regime; the first
// row that matches and whose regime is enabled wins; the boundaries are the measured crossovers (numbers in the notes,
// sources in DESIGN.md section 4 and profiles/).  The batch is counted on the GLOBAL mesh for mesh solves, so every window
// and shard of an array takes the same route (bit-identical halo rows and images).  tests/test_regimes.py forces every
// regime on the same matrices against LAPACK.
enum class Regime { Trig, TrigV, Blocked, Big, Reg, Ql16, Qlw, Row16, WgLds, WgGlobal, Wave };
enum BatchUnit { kMatrices, kPerCU, kWork };   // the batch window counts matrices | matrices per CU | matrices * n^2
enum { kList = 1, kMesh = 2, kSupplied = 4, kAnyForm = 7, kNoMesh = 5 };
struct RegimeRule {
    Regime regime;
    int n_lo, n_hi;
    int vec;            // 1: with eigenvectors, 0: eigenvalues only, -1: either
    int forms;          // kList | kMesh | kSupplied
    BatchUnit unit;
    double lo, hi;      // batch window: lo < batch (lo < 0: none) and batch <= hi (hi < 0: none)
    const char* note;
};
static const RegimeRule kRegimeRules[] = {
    // eigenvalues only, 65..1024 states: Householder with the matrix in L2 + one bisection per eigenvalue.  One CU per matrix:
    // 16 / 27 / 50 / 125 / 240 ms at n = 400 / 512 / 600 / 800 / 1024 whatever the batch, so few very large ones stay on Jacobi
    {Regime::Trig, 65, 512, 0, kNoMesh, kMatrices, -1, -1, "101 x n=300: 184 -> 12.6 ms; one n=512: 104 -> 27 ms"},
    {Regime::Trig, 513, 800, 0, kNoMesh, kMatrices, 1, -1, "8 x n=800: 257 -> 125 ms; a single one stays on the whole-chip rounds"},
    {Regime::Trig, 801, 1024, 0, kNoMesh, kMatrices, 5, -1, "a matrix costs 240 ms at n=1024: from 6 matrices on"},
    // with eigenvectors, 65..1024 states: the same reduction with the reflectors kept, twisted-factorisation vectors, Newton-Schulz,
    // back-transformation on LDS column strips (tbk_solve_trigv.inl).  One CU per matrix in the reduction, like Trig.
    {Regime::TrigV, 65, 512, 1, kAnyForm, kMatrices, -1, -1, "101 x n=300 with vectors: 143 (block Jacobi) -> see DESIGN.md"},
    {Regime::TrigV, 513, 1024, 1, kAnyForm, kMatrices, 5, -1, "few very large ones stay on the whole-chip rounds"},
    // batches of wide matrices: block Jacobi (n/8 - 1 passes per sweep instead of n - 1; three launches per round)
    {Regime::Blocked, 96, TBK_MAX_NSTA, -1, kAnyForm, kWork, 1.4e6 - 1, -1, "512 x n=128: 87/105 -> 35/55 ms; 64 x n=128: 11.8 -> 18.7 (stays out)"},
    // whole-chip Jacobi rounds: everything above 256 states, and 65..256 for small batches, eigenvalues only or n > 224
    {Regime::Big, 257, TBK_MAX_NSTA, -1, kAnyForm, kMatrices, -1, -1, "one 800-state flake: 70 / 86 ms"},
    {Regime::Big, 65, 256, -1, kAnyForm, kMatrices, -1, 160, "n=128: 32 k-points 21.6 -> 10.5 ms against a workgroup each"},
    {Regime::Big, 65, 256, 0, kAnyForm, kMatrices, -1, -1, "512 x n=128 eigenvalues: 97 (workgroup) | 81 (whole chip) ms"},
    {Regime::Big, 225, 256, 1, kAnyForm, kMatrices, -1, -1, "256 x n=256 with vectors: 670 | 648 ms"},
    // registers, one thread per matrix
    {Regime::Reg, 5, 8, -1, kAnyForm, kMatrices, -1, -1, "silicon (8 Wannier functions), 65^3: 3.7e7 -> 4.3e8 k/s"},
    // 9..16: direct solver on a DPP row.  With eigenvectors only for chip-filling batches (small ones keep Jacobi's relative
    // accuracy on nearly degenerate pairs: tests/test_reference_suite.py); eigenvalues of lists at any count
    // (the any-count rows hold while TBK_QL16_MIN / TBK_QLW_MIN are unset; an explicit minimum applies to every form)
    {Regime::Ql16, 9, 16, 0, kNoMesh, kMatrices, -1, -1, "262144 x n=16 eigenvalues: 3.5 -> 1.4 ms"},
    {Regime::Ql16, 9, 16, -1, kAnyForm, kPerCU, 8, -1, "cubic16 64^3 with vectors: 12.4 (LDS Jacobi) -> 1.9 ms"},
    // 17..64: tridiagonalise in LDS | lane-per-matrix QL or bisection | replay; the same batch rule
    {Regime::Qlw, 17, 64, 0, kNoMesh, kMatrices, -1, -1, "one 64 x 64: 1.26 -> 0.44 ms; 16384 x n=32: 11.8 -> 1.1 ms"},
    // (round 6: k_hh32 + k_ql32_lanes + k_tw32_vectors overtake the workgroup Jacobi kernels from ~3 matrices per CU with eigenvectors --
    // profiles/qlw_min_probe.py, ms Jacobi | direct: n=17 x 512 0.18 | 0.20, x 1024 0.20 | 0.20, x 2048 0.39 | 0.21; n=24 x 512 0.27 | 0.28,
    // x 1024 0.31 | 0.28, x 2048 0.65 | 0.30; n=32 x 512 0.43 | 0.46, x 1024 0.58 | 0.47, x 2048 1.63 | 0.51)
    {Regime::Qlw, 17, 32, 1, kAnyForm, kPerCU, 3, -1, "2048 x n=32 with vectors: 1.63 (workgroup Jacobi) -> 0.51 ms; 33^3 x n=32: 4.26 (round 5) -> 1.70"},
    {Regime::Qlw, 17, 64, -1, kAnyForm, kPerCU, 8, -1, "16384 x n=32 with vectors: 11.7 -> 2.2 ms (round 2); 33..64 states"},
    // 40..64 with eigenvectors BELOW that batch: the workgroup-scale direct method of 65..1024 states instead of workgroup Jacobi
    // (ms per call, Jacobi | direct, profiles/trigv_small_batches.py: n=64 x 16 1.18 | 0.87, x 1024 5.35 | 1.89; n=48 x 16 0.67 |
    // 0.63, x 1024 2.13 | 1.17; n=40 x 256 0.50 | 0.57, x 1024 1.69 | 0.92; n=32 x 1024 0.61 | 0.68 stays) -- and 3-6 x smaller errors
    {Regime::TrigV, 48, 64, 1, kAnyForm, kPerCU, -1, 8, "n=64: 1024 matrices 5.35 (workgroup Jacobi) -> 1.89 ms; 16: 1.18 -> 0.87"},
    {Regime::TrigV, 40, 47, 1, kAnyForm, kPerCU, 2, 8, "n=40: 1024 matrices 1.69 -> 0.92 ms; 256: 0.50 | 0.57 stays on Jacobi"},
    // 13..16 on lists and supplied matrices when the direct solver is off or the batch is small: Jacobi on a DPP row
    {Regime::Row16, 15, 16, 1, kNoMesh, kMatrices, -1, -1, "262144 x n=16 with vectors: 12.1 against 12.5 / 17.5 ms (LDS kernel)"},
    {Regime::Row16, 13, 16, 0, kNoMesh, kMatrices, -1, -1, "eigenvalues: 6.8-7.1 against 13.1 / 19.0 ms"},
    // a workgroup per LDS-resident matrix: every batch from 22 states, below that only batches that cannot fill the chip
    {Regime::WgLds, 22, 64, -1, kAnyForm, kMatrices, -1, -1, "16384 x n=32: 11.0 -> 5.5 ms; n=22 9.4 -> 8.3 is the crossover"},
    {Regime::WgLds, 5, 21, -1, kAnyForm, kPerCU, -1, 8, "one 30 x 30 matrix: 0.77 -> 0.31 ms"},
    // a 1024-thread workgroup per L2-resident matrix (65..224 with eigenvectors, more than 160 matrices)
    {Regime::WgGlobal, 65, 224, 1, kAnyForm, kMatrices, 160, -1, "512 x n=128 with vectors: 109 ms, the whole chip 145"},
    // one wavefront per LDS-resident matrix, warm-started along runs
    {Regime::Wave, 5, 21, -1, kAnyForm, kMatrices, -1, -1, ""},
};

struct RegimeQuery {
    int n, vec, form;          // form: kList | kMesh | kSupplied
    int64_t nk, batch;         // matrices of this launch | of the global mesh
    int cus;
    bool has_rblocks, qlw_off;
};
static Regime choose_regime(const RegimeQuery& q, const TbkKnobs& K, const char** note = nullptr) {
    for (const RegimeRule& r : kRegimeRules) {
        if (q.n > r.n_hi || (r.vec >= 0 && r.vec != q.vec) || !(r.forms & q.form)) continue;
        double lo = r.lo, hi = r.hi;
        int n_lo = r.n_lo;
        // ---- knobs: switch a regime off, or move its boundary
        switch (r.regime) {
            case Regime::Trig:
                if (K.use_trig == 0) continue;
                if (K.use_trig == 2) lo = -1;
                break;
            case Regime::TrigV:
                if (K.use_trigv == 0 || q.qlw_off) continue;   // (qlw_off: the call is being repeated on the Jacobi kernels)
                if (K.trigv_from > 0 && r.n_lo == 65) n_lo = std::max(17, K.trigv_from);
                break;
            case Regime::Blocked:
                if (K.blocked == 0) continue;
                if (K.blocked == 1) { lo = -1; n_lo = 65; }
                break;
            case Regime::Big:
                if (K.big_from >= 0 && r.n_lo == 257) n_lo = std::max(65, K.big_from);
                break;
            case Regime::Reg:
                if (K.use_reg == 0) continue;
                break;
            case Regime::Ql16:
                if (K.use_ql16 == 0 || !(q.form == kSupplied || q.has_rblocks)) continue;
                if (r.unit == kPerCU && K.ql16_min >= 0) lo = (double)K.ql16_min / q.cus;
                if (r.unit == kMatrices && K.ql16_min >= 0) continue;   // (an explicit minimum applies to eigenvalue-only lists too)
                break;
            case Regime::Qlw:
                if (K.use_qlw == 0 || q.qlw_off) continue;
                if (r.unit == kPerCU && K.qlw_min >= 0) lo = (double)K.qlw_min / q.cus;
                if (r.unit == kMatrices && K.qlw_min >= 0) continue;
                break;
            case Regime::Row16:
                if (K.use_row16 == 0 || !(q.form == kSupplied || q.has_rblocks)) continue;
                break;
            case Regime::WgLds:
                if (r.unit == kPerCU && K.few_max >= 0) hi = (double)K.few_max / q.cus;
                break;
            default: break;
        }
        if (q.n < n_lo) continue;
        const double batch = r.unit == kMatrices ? (double)q.batch : r.unit == kPerCU ? (double)q.batch / q.cus : (double)q.batch * q.n * q.n;
        // (the eigenvalue-only Trig rows count the matrices of THIS call: nothing to keep consistent across windows)
        const double b2 = r.regime == Regime::Trig ? (double)q.nk : batch;   // (TrigV: the global batch, so that every window agrees)
        if (lo >= 0 && !(b2 > lo)) continue;
        if (hi >= 0 && !(b2 <= hi)) continue;
        if (note) *note = r.note;
        return r.regime;
    }
    return q.n > 64 ? Regime::Big : Regime::Wave;
}

// Host-only view of the table (no device needed): which regime a batch would take, and the measurement behind the boundary.
extern "C" const char* tbk_solver_regime(int n, int with_vectors, int form, int64_t nk, int64_t batch, int compute_units,
                                         int has_rblocks, const char** note_out) {
    static const char* names[] = {"trig", "trigv", "blocked", "big", "reg", "ql16", "qlw", "row16", "wg_lds", "wg_global", "wave"};
    if (n <= 4) return n <= 2 ? "closed_form" : "ql_small";
    const RegimeQuery q{n, with_vectors ? 1 : 0, form == 1 ? kMesh : (form == 2 ? kSupplied : kList), nk, batch,
                        compute_units > 0 ? compute_units : 256, has_rblocks != 0, false};
    const char* note = "";
    const Regime r = choose_regime(q, tbk_knobs(), &note);
    if (note_out) *note_out = note;
    return names[(int)r];
}

template <int MODE, bool VEC>
static int launch_wave(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L,
                       const GridArgs& G) {
    int* flag = ctx->flags_dev;  // sticky until read by check_noconv
    GridArgs G2 = G;
    const TbkKnobs& K = tbk_knobs();
    int64_t nk_eff = nk;
    if (MODE == 1) {
        nk_eff = 1;
        for (int d = 0; d < G.wv.dim_arr; ++d) nk_eff *= G.gmesh[d];
    }
    const RegimeQuery q{n, VEC ? 1 : 0, MODE == 0 ? kList : (MODE == 1 ? kMesh : kSupplied), nk, nk_eff, ctx->cus, mv.nR > 0, ctx->qlw_off};
    const Regime regime = choose_regime(q, K);
    switch (regime) {
        case Regime::Trig:
            if constexpr (MODE != 1 && !VEC) return launch_trig<MODE>(ctx, mv, n, nk, L);
            break;
        case Regime::TrigV:
            if constexpr (VEC) return launch_trigv<MODE>(ctx, mv, n, nk, L, G);
            break;
        case Regime::Blocked: return launch_blocked<MODE, VEC>(ctx, mv, n, nk, L, G);
        case Regime::Big: return launch_big<MODE, VEC>(ctx, mv, n, nk, L, G);
        case Regime::Reg: return launch_reg<MODE, VEC>(ctx, mv, n, nk, L, G);
        case Regime::Ql16: return launch_ql16<MODE, VEC>(ctx, mv, nk, L, G, nk_eff);
        case Regime::Qlw: return launch_qlw<MODE, VEC>(ctx, mv, n, nk, L, G);
        case Regime::Row16:
            if constexpr (MODE != 1) return launch_row16<MODE, VEC>(ctx, mv, nk, L, G);
            break;
        default: break;
    }
    // A 256-thread workgroup per LDS-resident matrix instead of one wavefront:
    //  * n >= 22, any batch: a wavefront's matrix with its warm-start buffer takes 3 n(n+1) 16 B of LDS (n = 48:
    //    113 KB, ONE wavefront per CU); several wavefronts sharing one matrix keep 8-16 wavefronts per CU busy, and
    //    large batches warm-start along their runs as well where that still leaves two matrices per CU (n <= 40).
    //    Eigenvalues, mesh-ordered k (ms, wavefront kernel -> this one): 65536 x n=24 13.6 -> 10.6, 16384 x n=32
    //    11.0 -> 5.5, n=40 (4.7 at 1024 k) -> 11.7 (1.8), n=48 92 -> 33, 8192 x n=64 209 -> 39; n=22 9.4 -> 8.3 is
    //    the crossover (n=20: 6.7 vs 7.5).
    //  * n < 22: only while the batch cannot fill the chip with wavefronts (one 30x30 matrix: 0.77 -> 0.31 ms).
    const bool few = regime == Regime::WgLds;
    if (regime == Regime::WgGlobal || few) {
        // ---- workgroup per matrix: 256 threads, cold start; n = 65..256: A and V^T in a global workspace
        // (ribbon / slab models: few, large matrices), n <= 64: in LDS.
        // a large batch walks runs of consecutive points per workgroup anyway: warm-start along them when the
        // extra buffer still leaves two matrices per CU (TBK_FEW_WARM=0 disables)
        const bool warm_knob = K.few_warm != 0;
        const int64_t cap = std::max<int64_t>(64, (int64_t)ctx->cus * 4);
        const int warm_few = few && warm_knob && nk_eff >= 2 * cap && wave_lds_bytes(n, true, mv.nR) <= 80 * 1024 ? 1 : 0;
        const size_t lds = few ? wave_lds_bytes(n, warm_few != 0, mv.nR)
                               : wave_lds_bytes(n, false) - (size_t)2 * n * (n + 1) * sizeof(cd);   // small arrays only
        int64_t run, nblocks;
        if (MODE == 1) {
            const int last = G.last;
            const int64_t off = G.off[last], nl = G.wv.mesh[last];
            // (warm runs must not depend on the window: the run length comes from the global mesh size)
            run = std::max<int64_t>(1, ((warm_few ? nk_eff : G.wv.npts) + cap - 1) / cap);
            if (warm_few) run = std::min<int64_t>(run, 16);
            G2.wcfirst = off / run;
            G2.wnchunk = (int)((off + nl - 1) / run - G2.wcfirst + 1);
            nblocks = (G.wv.npts / nl) * G2.wnchunk;
        } else {
            run = std::max<int64_t>(1, (nk + cap - 1) / cap);
            if (warm_few) run = std::min<int64_t>(run, 64);   // short chains: V's orthonormality drift stays ~1e-15
            nblocks = (nk + run - 1) / run;
        }
        TBK_REQUIRE(nblocks < (int64_t)0x7fffffff, TBK_EUNSUPPORTED, "too many k-points for one launch");
        const size_t wbytes = few ? 0 : (size_t)nblocks * 2 * n * (n + 1) * sizeof(cd);
        if (few) {
            // threads per matrix, measured (eigenvalues, ms; 256 / 512 / 1024 threads): 16384 matrices n=32 9.1 / 12.3 /
            // 19.8, n=48 41 / 33 / 42, 8192 of n=64 66 / 46 / 39; a single matrix n=50 1.13 / 0.84 / 0.70, n=64 1.96 / 1.41 / 1.18
            int nt;
            if (nk_eff > (int64_t)ctx->cus) nt = n <= 40 ? 256 : (n <= 56 ? 512 : 1024);
            else nt = n <= 24 ? 256 : (n <= 36 ? 512 : 1024);
            if (K.few_nt >= 0) nt = K.few_nt >= 1024 ? 1024 : (K.few_nt >= 512 ? 512 : 256);
            static bool attr_few[2][3][3] = {};
            const int ti = nt == 256 ? 0 : (nt == 512 ? 1 : 2);
            if (lds > 64 * 1024 && !attr_few[VEC][MODE][ti]) {
                const void* fn = nt == 256 ? (const void*)k_solve_wave<MODE, VEC, 256, true>
                                 : nt == 512 ? (const void*)k_solve_wave<MODE, VEC, 512, true> : (const void*)k_solve_wave<MODE, VEC, 1024, true>;
                TBK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_few[VEC][MODE][ti] = true;
            }
            if (nt == 256)
                hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 256, true>), dim3((unsigned)nblocks), dim3(256), lds, ctx->stream, mv, nk, L,
                                   G2, flag, (int)run, (cd*)nullptr, warm_few);
            else if (nt == 512)
                hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 512, true>), dim3((unsigned)nblocks), dim3(512), lds, ctx->stream, mv, nk, L,
                                   G2, flag, (int)run, (cd*)nullptr, warm_few);
            else
                hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 1024, true>), dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, mv, nk,
                                   L, G2, flag, (int)run, (cd*)nullptr, warm_few);
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
        if (wbytes > ctx->work_bytes) {
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->work) TBK_HIP(hipFree(ctx->work));
            ctx->work = nullptr;
            ctx->work_bytes = 0;
            hipError_t e = hipMalloc(&ctx->work, wbytes);
            TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "eigen-solver workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
            ctx->work_bytes = wbytes;
        }
        const int wg_nt = K.wg_nt;
        if (wg_nt >= 1024)
            hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 1024, false>), dim3((unsigned)nblocks), dim3(1024), lds, ctx->stream, mv, nk,
                               L, G2, flag, (int)run, (cd*)ctx->work, 0);
        else
            hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 256>), dim3((unsigned)nblocks), dim3(256), lds, ctx->stream, mv, nk, L,
                               G2, flag, (int)run, (cd*)ctx->work, 0);
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
    const bool with_t = wave_lds_bytes(n, true, mv.nR) <= 160 * 1024;   // else every point starts cold
    const size_t lds = wave_lds_bytes(n, with_t, mv.nR);
    TBK_REQUIRE(lds <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d needs %zu bytes of LDS per wavefront", n, lds);
    static bool attr_set[2][3] = {{false, false, false}, {false, false, false}};
    if (lds > 64 * 1024 && !attr_set[VEC][MODE]) {
        TBK_HIP(hipFuncSetAttribute((const void*)k_solve_wave<MODE, VEC, 64>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[VEC][MODE] = true;
    }
    // enough resident wavefronts to fill the chip
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / lds));
    const int64_t want = (int64_t)ctx->cus * per_cu * 2;
    // chains of consecutive points per wavefront (warm-started Jacobi)
    int64_t run, nblocks;
    if (MODE == 1) {
        // mesh: fixed chain length, aligned to the global last-axis index (see the kernel)
        run = 16;
        if (K.wave_run >= 0) run = std::max(1, K.wave_run);   // (1 = always cold)
        if (!with_t) run = 1;
        const int last = G.last;
        const int64_t off = G.off[last], nl = G.wv.mesh[last];
        G2.wcfirst = off / run;
        G2.wnchunk = (int)((off + nl - 1) / run - G2.wcfirst + 1);
        nblocks = (G.wv.npts / nl) * G2.wnchunk;
    } else {
        // k list: contiguous runs by list position, at most 64 long
        run = std::max<int64_t>(1, std::min<int64_t>(64, (nk + want - 1) / want));
        if (K.wave_run >= 0) run = std::max(1, K.wave_run);
        if (!with_t) run = 1;
        nblocks = (nk + run - 1) / run;
    }
    TBK_REQUIRE(nblocks < (int64_t)0x7fffffff, TBK_EUNSUPPORTED, "too many k-points for one launch");
    const unsigned blocks = (unsigned)nblocks;
    hipLaunchKernelGGL((k_solve_wave<MODE, VEC, 64>), dim3(blocks), dim3(64), lds, ctx->stream, mv, nk, L, G2, flag, (int)run,
                       (cd*)nullptr, 1);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// MODE 0 / 2: list kernels
template <int MODE>
static int launch_solve(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, bool vec,
                        const ListArgs& L, const char* name) {
    if (nk <= 0) return TBK_OK;
    ProfScope ps(ctx, name);
    GridArgs G{};
    G.flags = ctx->flags_dev;
    ListArgs Lf = L;
    Lf.flags = ctx->flags_dev;
    if (n <= 4) return vec ? launch_small<MODE, true>(ctx, mv, n, nk, Lf) : launch_small<MODE, false>(ctx, mv, n, nk, Lf);
    return vec ? launch_wave<MODE, true>(ctx, mv, n, nk, Lf, G) : launch_wave<MODE, false>(ctx, mv, n, nk, Lf, G);
}

// TBK_ERETRY_JACOBI (internal): the QL rotation record of the n = 17..64 path overflowed; callers that still have their
// inputs repeat the solve with ctx->qlw_off set, the others report TBK_ENOCONV.
#define TBK_ERETRY_JACOBI (-1000)
static int check_noconv(tbk_ctx* ctx, int n, bool can_retry = false) {
    if (n <= 2) return TBK_OK;   // (closed forms: nothing iterates)
    int flag[4] = {0, 0, 0, 0};   // 0: no convergence, 1: (Berry path) singular link, 2: the QL rotation record overflowed
    TBK_HIP(hipMemcpyAsync(flag, ctx->flags_dev, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    if (flag[0]) TBK_HIP(hipMemsetAsync(ctx->flags_dev, 0, sizeof(int), ctx->stream));
    if (flag[2]) TBK_HIP(hipMemsetAsync(ctx->flags_dev + 2, 0, sizeof(int), ctx->stream));
    TBK_REQUIRE(flag[0] == 0, TBK_ENOCONV, "eigen-solver did not converge (Jacobi: %d sweeps; QL: 30 shifts per eigenvalue)",
                TBK_JACOBI_MAX_SWEEPS);
    if (flag[2] && can_retry && !ctx->qlw_off) return TBK_ERETRY_JACOBI;
    TBK_REQUIRE(flag[2] == 0, TBK_ENOCONV,
                "eigen-solver: a matrix needed more than 3 n^2 QL rotations (4 x the usual); eigenvectors were not "
                "completed -- rerun with TBK_QLW=0 (Jacobi kernels)");
    return TBK_OK;
}

extern "C" int tbk_solve_list_dev(tbk_model* m, const double* k_dev, int64_t nk, double* eval_dev,
                                  double* evec_dev) {
    TBK_REQUIRE(m && eval_dev && nk >= 0, TBK_EINVAL, "tbk_solve_list_dev: bad argument");
    TBK_REQUIRE(m->dim_k == 0 || k_dev || nk == 0, TBK_EINVAL, "tbk_solve_list_dev: null k");
    ListArgs L{k_dev, nullptr, eval_dev, (cd*)evec_dev};
    return launch_solve<0>(m->ctx, m->view, m->nsta, nk, evec_dev != nullptr, L,
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
    // small calls (a band-structure path, solve_one): the k list and the results go through host memory that the device reads
    // and writes itself -- one synchronisation instead of a host-to-device and a device-to-host copy around the launch
    // (solve_all on a 121-point path 30 -> see profiles/call_latency.py)
    const size_t zc_need = al(kb) + al(eb) + al(vb);
    if (tbk_knobs().zero_copy_kb > 0 && zc_need <= (size_t)tbk_knobs().zero_copy_kb << 10) {
        void* zh = nullptr;
        void* zd = nullptr;
        int rcz = tbk_ctx_zero_copy(ctx, zc_need, &zh, &zd);
        if (rcz) return rcz;
        if (zh) {
            unsigned char* hp_ = (unsigned char*)zh;
            unsigned char* dp_ = (unsigned char*)zd;
            if (m->dim_k > 0) {
                TBK_REQUIRE(k, TBK_EINVAL, "tbk_solve_list: null k");
                memcpy(hp_, k, kb);
            }
            double* const e_zd = (double*)(dp_ + al(kb));
            double* const v_zd = evec ? (double*)(dp_ + al(kb) + al(eb)) : nullptr;
            // up to 4 states: ONE kernel (k_solve_small), which stores the completion word after its results -- the host polls that
            // instead of synchronising the stream (tbk_done_wait), and the status words come back beside it
            const DoneArgs done = n <= 4 ? tbk_done_arm(ctx, n > 2) : DoneArgs{nullptr, nullptr, nullptr, 0u};
            if (done.word) {
                ListArgs L{(const double*)dp_, nullptr, e_zd, (cd*)v_zd};
                L.done = done;
                rcz = launch_solve<0>(ctx, m->view, n, nk, evec != nullptr, L, evec ? "solve_list_vec" : "solve_list_val");
                if (rcz) return rcz;
                rcz = tbk_done_wait(ctx, done);
                if (rcz) return rcz;
                if (done.flags_src && ctx->done_host[4] != 0u) {     // an eigen-solver ran into its iteration cap: the usual report
                    rcz = check_noconv(ctx, n);
                    if (rcz) return rcz;
                }
            } else {
                rcz = tbk_solve_list_dev_checked(m, (const double*)dp_, nk, e_zd, v_zd);
                if (rcz) return rcz;
                TBK_HIP(hipStreamSynchronize(ctx->stream));
            }
            memcpy(eval, hp_ + al(kb), eb);
            if (evec) memcpy(evec, hp_ + al(kb) + al(eb), vb);
            return TBK_OK;
        }
    }
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
    rc = tbk_solve_list_dev_checked(m, k_dev, nk, e_dev, v_dev);   // (the inputs stay on the device: a record overflow is repeated on Jacobi)
    if (rc) return rc;
    TBK_HIP(hipMemcpyAsync(eval, e_dev, eb, hipMemcpyDeviceToHost, ctx->stream));
    if (evec) TBK_HIP(hipMemcpyAsync(evec, v_dev, vb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
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
    rc = tbk_eigh_dev_checked(ctx, n, h_dev, nk, e_dev, v_dev, "eigh_batch");
    if (rc) return rc;
    TBK_HIP(hipMemcpyAsync(eval, e_dev, eb, hipMemcpyDeviceToHost, ctx->stream));
    if (evec) TBK_HIP(hipMemcpyAsync(evec, v_dev, vb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

// device-to-device form used by other translation units (position operator path)
int tbk_eigh_dev(tbk_ctx* ctx, int n, const cd* ham_dev, int64_t nk, double* eval_dev, cd* evec_dev,
                 const char* name) {
    ModelView mv{};
    mv.nsta = n;
    mv.nspin = 1;
    mv.nslot = n * (n + 1) / 2;
    ListArgs L{nullptr, ham_dev, eval_dev, evec_dev};
    return launch_solve<2>(ctx, mv, n, nk, evec_dev != nullptr, L, name);
}

int tbk_eigh_check(tbk_ctx* ctx, int n) { return check_noconv(ctx, n); }

// The same two solves WITH their check (one host synchronisation), repeated once on the Jacobi kernels when a direct solver's
// rotation record overflowed -- for every caller whose inputs are still on the device at that point (mesh solves on generated
// k lists, the position operator's small eigenproblems): the one attempt loop instead of a hard TBK_ENOCONV.
int tbk_eigh_dev_checked(tbk_ctx* ctx, int n, const cd* ham_dev, int64_t nk, double* eval_dev, cd* evec_dev, const char* name) {
    int rc = TBK_OK;
    for (int attempt = 0;; ++attempt) {
        rc = tbk_eigh_dev(ctx, n, ham_dev, nk, eval_dev, evec_dev, name);
        if (rc == TBK_OK) rc = check_noconv(ctx, n, attempt == 0);
        ctx->qlw_off = rc == TBK_ERETRY_JACOBI;
        if (rc != TBK_ERETRY_JACOBI) break;
    }
    return rc;
}
extern "C" int tbk_solve_list_dev_checked(tbk_model* m, const double* k_dev, int64_t nk, double* eval_dev, double* evec_dev) {
    TBK_REQUIRE(m && eval_dev && nk >= 0, TBK_EINVAL, "tbk_solve_list_dev_checked: bad argument");
    int rc = TBK_OK;
    for (int attempt = 0;; ++attempt) {
        rc = tbk_solve_list_dev(m, k_dev, nk, eval_dev, evec_dev);
        if (rc == TBK_OK) rc = check_noconv(m->ctx, m->nsta, attempt == 0);
        m->ctx->qlw_off = rc == TBK_ERETRY_JACOBI;
        if (rc != TBK_ERETRY_JACOBI) break;
    }
    return rc;
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
    // (small calls through mapped host memory, like tbk_solve_list)
    void* zh = nullptr;
    void* zd = nullptr;
    if (tbk_knobs().zero_copy_kb > 0 && al(kb) + al(hb) <= (size_t)tbk_knobs().zero_copy_kb << 10) {
        int rcz = tbk_ctx_zero_copy(ctx, al(kb) + al(hb), &zh, &zd);
        if (rcz) return rcz;
    }
    void* base = nullptr;
    if (!zh) {
        int rc = tbk_ctx_scratch(ctx, 256 + al(kb) + al(hb), &base);
        if (rc) return rc;
    }
    double* k_dev = zh ? (double*)zd : (double*)((unsigned char*)base + 256);
    cd* h_dev = zh ? (cd*)((unsigned char*)zd + al(kb)) : (cd*)((unsigned char*)base + 256 + al(kb));
    if (m->dim_k > 0) {
        TBK_REQUIRE(k, TBK_EINVAL, "tbk_gen_ham: null k");
        if (zh) memcpy(zh, k, kb);
        else TBK_HIP(hipMemcpyAsync(k_dev, k, kb, hipMemcpyHostToDevice, ctx->stream));
    }
    if (zh) memset((unsigned char*)zh + al(kb), 0, hb);
    else TBK_HIP(hipMemsetAsync(h_dev, 0, hb, ctx->stream));
    DoneArgs done{nullptr, nullptr, nullptr, 0u};
    {
        ProfScope ps(ctx, "gen_ham");
        const int64_t total = nk * m->nslot;
        done = zh ? tbk_done_arm(ctx, false) : DoneArgs{nullptr, nullptr, nullptr, 0u};
        hipLaunchKernelGGL(k_gen_ham, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           m->view, nk, k_dev, h_dev, done);
        TBK_HIP(hipGetLastError());
    }
    if (zh) {
        const int rcw = tbk_done_wait(ctx, done);
        if (rcw) return rcw;
        memcpy(ham_out, (unsigned char*)zh + al(kb), hb);
        return TBK_OK;
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
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_wfs_solve_grid: null argument");
    int64_t off[TBK_MAX_DIM] = {row0, 0, 0, 0}, gm[TBK_MAX_DIM];
    for (int d = 0; d < TBK_MAX_DIM; ++d) gm[d] = w->view.mesh[d];
    gm[0] = global_n0;
    return tbk_wfs_solve_window_async(w, m, start_k, pbc_phase, off, gm);
}

struct FusedFluxReq {
    int occ[2];
    int nocc;
};
static int solve_window_impl(tbk_wfs* w, tbk_model* m, const double* start_k, const double* pbc_phase, const int64_t* offset,
                             const int64_t* global_mesh, const FusedFluxReq* ff);

extern "C" int tbk_wfs_solve_window_async(tbk_wfs* w, tbk_model* m, const double* start_k,
                                          const double* pbc_phase, const int64_t* offset,
                                          const int64_t* global_mesh) {
    return solve_window_impl(w, m, start_k, pbc_phase, offset, global_mesh, nullptr);
}

// solve_on_grid + berry_flux(occ, dirs = [0, 1]) of a 2-D array in one pass (tbk_solve_fused.inl); results through
// tbk_wfs_solve_grid_result (min gaps) and tbk_berry_flux_result (the total).  TBK_EUNSUPPORTED where the fused kernel does
// not apply (the caller then issues the two calls).
extern "C" int tbk_wfs_solve_grid_flux_async(tbk_wfs* w, tbk_model* m, const double* start_k, const double* pbc_phase,
                                             int64_t row0, int64_t global_n0, const int32_t* occ, int nocc) {
    TBK_REQUIRE(w && occ, TBK_EINVAL, "tbk_wfs_solve_grid_flux: null argument");
    TBK_REQUIRE(nocc >= 1 && nocc <= 2, TBK_EUNSUPPORTED, "tbk_wfs_solve_grid_flux: %d occupied bands (1 or 2)", nocc);
    FusedFluxReq ff{{occ[0], nocc > 1 ? occ[1] : occ[0]}, nocc};
    for (int a = 0; a < nocc; ++a)
        TBK_REQUIRE(occ[a] >= 0 && occ[a] < w->view.nsta, TBK_EINVAL, "tbk_wfs_solve_grid_flux: band %d out of range", occ[a]);
    TBK_REQUIRE(nocc == 1 || occ[0] != occ[1], TBK_EINVAL, "tbk_wfs_solve_grid_flux: repeated band");
    int64_t off[TBK_MAX_DIM] = {row0, 0, 0, 0}, gm[TBK_MAX_DIM];
    for (int d = 0; d < TBK_MAX_DIM; ++d) gm[d] = w->view.mesh[d];
    gm[0] = global_n0;
    return solve_window_impl(w, m, start_k, pbc_phase, off, gm, &ff);
}

static int solve_window_impl(tbk_wfs* w, tbk_model* m, const double* start_k, const double* pbc_phase, const int64_t* offset,
                             const int64_t* global_mesh, const FusedFluxReq* ff) {
    TBK_REQUIRE(w && m && start_k && pbc_phase && offset && global_mesh, TBK_EINVAL, "tbk_wfs_solve_grid: null argument");
    TBK_REQUIRE(w->ctx == m->ctx, TBK_EINVAL, "tbk_wfs_solve_grid: model and wfs live on different contexts");
    const WfsView& v = w->view;
    // pythtb.py:2448-2459
    TBK_REQUIRE(v.dim_arr == m->dim_k, TBK_EINVAL,
                "tbk_wfs_solve_grid: dimension of wf_array (%d) must equal dim_k (%d)", v.dim_arr, m->dim_k);
    TBK_REQUIRE(v.nsta == m->nsta && v.ncomp == m->nsta, TBK_EINVAL,
                "tbk_wfs_solve_grid: array holds %d states of %d components, model has %d", v.nsta, v.ncomp, m->nsta);
    for (int d = 0; d < v.dim_arr; ++d)
        TBK_REQUIRE(global_mesh[d] >= v.mesh[d] && global_mesh[d] < (int64_t)0x7fffffff && offset[d] >= 0 &&
                        offset[d] + v.mesh[d] <= global_mesh[d],
                    TBK_EINVAL, "tbk_wfs_solve_grid: window [%lld,%lld) outside global axis %d of %lld",
                    (long long)offset[d], (long long)(offset[d] + v.mesh[d]), d, (long long)global_mesh[d]);
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const int n = m->nsta;
    const int D = v.dim_arr;
    if (!ctx->qlw_off) {   // (kept for tbk_wfs_solve_grid_result, which may have to repeat this launch)
        w->last_model = m;
        for (int d = 0; d < TBK_MAX_DIM; ++d) {
            w->last_start[d] = d < D ? start_k[d] : 0.0;
            w->last_off[d] = d < D ? offset[d] : 0;
            w->last_gmesh[d] = d < D ? global_mesh[d] : 1;
        }
        w->last_pbc.assign(pbc_phase, pbc_phase + (size_t)D * n * 2);
    }
    GridArgs G{};
    G.wv = v;
    for (int d = 0; d < TBK_MAX_DIM; ++d) {
        G.start_k[d] = d < D ? start_k[d] : 0.0;
        G.gmesh[d] = d < D ? (int)global_mesh[d] : 1;
        G.off[d] = d < D ? offset[d] : 0;
    }
    G.pbc = w->pbc_dev;
    G.flags = ctx->flags_dev;
    // everything the per-axis tables depend on; rebuild them only when it changes
    std::vector<double> key;
    key.push_back((double)m->upload_id);
    for (int d = 0; d < D; ++d) {
        key.push_back(start_k[d]);
        key.push_back((double)offset[d]);
        key.push_back((double)global_mesh[d]);
    }
    key.insert(key.end(), pbc_phase, pbc_phase + (size_t)D * n * 2);
    int64_t ntab = 0;
    for (int d = 0; d < D; ++d) ntab += v.mesh[d];
    const int64_t need = ntab * (1 + n) + n;           // (+ n: the last axis' phases at global index 0, GridArgs::tf0)
    if (w->tab_cap < need) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (w->tab_dev) TBK_HIP(hipFree(w->tab_dev));
        w->tab_dev = nullptr;
        w->tab_cap = 0;
        TBK_HIP(hipMalloc((void**)&w->tab_dev, need * sizeof(cd)));
        w->tab_cap = need;
        w->tab_key.clear();
    }
    {
        int64_t zo = 0, fo = ntab;
        for (int d = 0; d < D; ++d) {
            G.tz[d] = w->tab_dev + zo;
            G.tf[d] = w->tab_dev + fo;
            zo += v.mesh[d];
            fo += (int64_t)v.mesh[d] * n;
        }
        G.tf0 = w->tab_dev + ntab * (1 + n);
    }
    if (key != w->tab_key) {
        std::vector<cd> st((size_t)TBK_MAX_DIM * n, cd{0.0, 0.0});
        for (int d = 0; d < D; ++d)
            for (int o = 0; o < n; ++o)
                st[d * n + o] = cd{pbc_phase[2 * (d * n + o)], pbc_phase[2 * (d * n + o) + 1]};
        TBK_HIP(hipMemcpyAsync(w->pbc_dev, st.data(), st.size() * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
        TBK_HIP(hipStreamSynchronize(ctx->stream));  // st is a local: finish before it dies
        ProfScope ps(ctx, "grid_tables");
        hipLaunchKernelGGL(k_grid_tables, dim3((unsigned)((ntab + 255) / 256)), dim3(256), 0, ctx->stream, m->view, G,
                           w->tab_dev, w->tab_dev + ntab, w->tab_dev + ntab * (1 + n));
        TBK_HIP(hipGetLastError());
        w->tab_key = key;
    }
    // min gaps: this launch reduces into one parity and re-arms the other
    const size_t half = (size_t)TBK_GAP_SHARDS * w->view.ncomp;
    G.gaps = w->gaps_dev + (size_t)w->gaps_parity * half;
    G.gaps_next = w->gaps_dev + (size_t)(1 - w->gaps_parity) * half;
    w->gaps_parity = 1 - w->gaps_parity;   // *_result reads 1 - gaps_parity
    w->gaps_n = n - 1;
    w->gap_part_n = 0;
    G.last = D - 1;
    G.cpr = (v.mesh[D - 1] + 63) / 64;
    G.nchunks = (v.npts / v.mesh[D - 1]) * G.cpr;
    G.reg_cells = tbk_knobs().reg_cells != 0 && G.gmesh[D - 1] >= 32 ? 1 : 0;   // (decided on the GLOBAL mesh: a window of it must take the same form)
#ifdef TBK_DIAG
    G.ablate = tbk_knobs().ablate_grid;
#endif
    if (ff) {
        // ---- the fused kernel: 2-D arrays of 2 or 4 states, short-ranged along the last axis
        const int pm = m->view.pmax;
        TBK_REQUIRE(D == 2 && (n == 2 || n == 4) && ff->nocc <= n && pm >= 0 && pm <= 2 && (n == 2 || pm == 1) &&
                        tbk_knobs().grid_kernel != 1 && v.npts < (int64_t)0x7fffffff,
                    TBK_EUNSUPPORTED, "tbk_wfs_solve_grid_flux: a %d-D array of %d states (last-axis range %d): the fused kernel "
                    "serves 2-D arrays of 2 states (range <= 2) or 4 states (range 1)", D, n, pm);
        const TbkKnobs& K = tbk_knobs();
        FusedArgs F{};
        // tile shape (measured, profiles/fused_sweep.py, us per step, rows R x chunks seg).  2048^2 (the array fits the 256 MiB
        // last-level cache): R = 3 73-86, 4 67-79, 5 67-75, 6 65-74, 8 63-87, 12 63-66, best around 6 x 2.  4096^2 (beyond it): tall
        // and narrow wins -- seg = 1: R = 4 303, 6 266, 8 258, 10 251, 12 257; seg >= 2: 275-318.  After the register diet (five
        // wavefronts per SIMD, no column tables in LDS at seg = 1): 2048^2 flat over R = 5-12 (60-63), 4096^2 R = 8 257, 10 265
        const bool in_llc = w->bytes <= ((int64_t)256 << 20) + (1 << 20);
        F.R = K.fused_rows > 0 ? std::min(K.fused_rows, 16) : (in_llc ? 6 : 8);
        F.nrg = (v.mesh[0] + F.R - 1) / F.R;
        F.occ[0] = ff->occ[0];
        F.occ[1] = ff->occ[1];
        const int64_t want = (int64_t)ctx->cus * 32;     // wave tiles that fill the chip
        const int R1 = F.R + 1, ncell = (n * (n + 1) / 2) * (2 * pm + 1), ncar = ff->nocc * n + 1;
        // LDS per wavefront: cells, leading-axis phases and carries of R + 1 rows, the staging tile, the per-column tables of
        // `seg` chunks; at most 16 KB per wavefront
        const size_t fixed_cd = (size_t)R1 * (ncell + n + ncar) + (size_t)64 * n;   // (k_grid_rows_flux: NB = 1 band staged at once)
        const int seg_cap = (int)std::max<size_t>(1, (1024 - std::min<size_t>(fixed_cd, 960)) / ((size_t)64 * (1 + n)));
        G.seg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(seg_cap, G.cpr), (int64_t)G.cpr * F.nrg / want));
        if (!in_llc) G.seg = 1;
        if (K.grid_seg >= 0) G.seg = std::max(1, std::min(std::min(K.grid_seg, seg_cap), G.cpr));
        G.tpr = (G.cpr + G.seg - 1) / G.seg;
        G.seg = (G.cpr + G.tpr - 1) / G.tpr;
        G.ntiles = (int64_t)F.nrg * G.tpr;
        const int nseam = G.tpr - 1;
        const int64_t seam_threads = (int64_t)(v.mesh[0] - 1) * nseam;
        const int64_t nsb = (seam_threads + 255) / 256;
        // buffers: per-tile gap minima; flux partials (tiles, then seam blocks) and the total
        const int64_t npart = G.ntiles * std::max(n - 1, 1);
        if (w->gap_part_cap < npart) {
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (w->gap_part_dev) TBK_HIP(hipFree(w->gap_part_dev));
            w->gap_part_dev = nullptr;
            w->gap_part_cap = 0;
            TBK_HIP(hipMalloc((void**)&w->gap_part_dev, (size_t)npart * sizeof(double)));
            w->gap_part_cap = npart;
        }
        G.gap_part = w->gap_part_dev;
        w->gap_part_n = G.ntiles;
        if (w->flux_nslices_cap < 1) {
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (w->flux_cnt_dev) TBK_HIP(hipFree(w->flux_cnt_dev));
            w->flux_cnt_dev = nullptr;
            {
                const int rct = tbk_wfs_totals_alloc(w, 1);
                if (rct) return rct;
            }
            TBK_HIP(hipMalloc((void**)&w->flux_cnt_dev, 16 * sizeof(unsigned)));
            TBK_HIP(hipMemsetAsync(w->flux_cnt_dev, 0, 16 * sizeof(unsigned), ctx->stream));
            w->flux_nslices_cap = 1;
        }
        if (w->flux_partial_cap < G.ntiles + nsb + 2) {
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (w->flux_partial_dev) TBK_HIP(hipFree(w->flux_partial_dev));
            w->flux_partial_dev = nullptr;
            TBK_HIP(hipMalloc((void**)&w->flux_partial_dev, (size_t)(G.ntiles + nsb + 2) * sizeof(double)));
            w->flux_partial_cap = G.ntiles + nsb + 2;
        }
        w->flux_nslices = 1;
        w->flux_plaq_n = 0;
        F.partial = w->flux_partial_dev;
        size_t lds = (size_t)4 * (fixed_cd + (G.seg > 1 ? (size_t)G.seg * 64 * (1 + n) : 0)) * sizeof(cd);   // (seg = 1: no column tables in LDS)
        // beyond the last-level cache FEWER resident wavefronts write faster (us per step at 4096^2 for 2 / 3 / 4 / 5 per SIMD:
        // 284 / 245 / 259 / 263; at 2048^2: 72.0 / 61.7 / 59.5 / 58.7): the cap is set through the LDS request (160 KB per CU /
        // blocks per CU), the only launch-time handle on occupancy
        const int occ_cap = K.fused_occ > 0 ? K.fused_occ : (in_llc ? 0 : 3);
        // (a knob that cannot be honoured is an error of its own kind: TBK_EUNSUPPORTED would send solve_on_grid_flux to the
        // two-call path and an A/B run would time the wrong kernel -- ADVICE r3)
        TBK_REQUIRE(K.fused_occ != 1, TBK_EINVAL,
                    "TBK_FUSED_OCC=1 asks for %d bytes of LDS per workgroup (limit 65536 without a function attribute): use 2..7",
                    160 * 1024 / 2 + 1024);
        if (occ_cap > 0 && occ_cap < 8) lds = std::max(lds, (size_t)(160 * 1024) / (size_t)(occ_cap + 1) + 1024);
        TBK_REQUIRE(lds <= 64 * 1024, TBK_EUNSUPPORTED, "tbk_wfs_solve_grid_flux: %zu bytes of LDS per block", lds);
        const unsigned blocks = (unsigned)((G.ntiles + 3) / 4);
        {
        ProfScope ps(ctx, "solve_grid_flux");   // (the fused kernel alone)
#define TBK_FUSED(NN, PP, OO) hipLaunchKernelGGL((k_grid_rows_flux<NN, PP, OO>), dim3(blocks), dim3(256), lds, ctx->stream, m->view, G, F)
        if (n == 2 && ff->nocc == 1) {
            if (pm == 0) TBK_FUSED(2, 0, 1); else if (pm == 1) TBK_FUSED(2, 1, 1); else TBK_FUSED(2, 2, 1);
        } else if (n == 2) {
            if (pm == 0) TBK_FUSED(2, 0, 2); else if (pm == 1) TBK_FUSED(2, 1, 2); else TBK_FUSED(2, 2, 2);
        } else if (ff->nocc == 1) {
            TBK_FUSED(4, 1, 1);
        } else {
            TBK_FUSED(4, 1, 2);
        }
#undef TBK_FUSED
        }
        // ---- seams + total in one launch (k_flux_seams_sum); TBK_FUSED_SUM=0: the total by a kernel of its own
        ProfScope ps2(ctx, "flux_seams_sum");
        const bool own_sum = K.fused_sum == 0;
        const unsigned sblocks = (unsigned)std::max<int64_t>(nsb, 1);
        double* const blk_partial = F.partial + G.ntiles;
#define TBK_SEAMS(NN, OO)                                                                                                              \
    hipLaunchKernelGGL((k_flux_seams_sum<NN, OO>), dim3(sblocks), dim3(256), 0, ctx->stream, v, F.occ[0], F.occ[1], G.seg, nseam,        \
                       (const double*)F.partial, own_sum ? (int64_t)0 : G.ntiles, blk_partial, w->flux_cnt_dev,                            \
                       own_sum ? blk_partial + sblocks : w->flux_totals_dev)
        if (n == 2 && ff->nocc == 1) TBK_SEAMS(2, 1);
        else if (n == 2) TBK_SEAMS(2, 2);
        else if (ff->nocc == 1) TBK_SEAMS(4, 1);
        else TBK_SEAMS(4, 2);
#undef TBK_SEAMS
        if (own_sum)   // (the seam kernel's own total goes to a spare slot; the blocks' sums stand behind the tiles' partials)
            hipLaunchKernelGGL(k_sum_fixed, dim3(1), dim3(1024), 0, ctx->stream, (const double*)F.partial, (int64_t)(G.ntiles + sblocks), w->flux_totals_dev);
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
    ProfScope ps(ctx, "solve_grid");
    if (n <= 4) {
        TBK_REQUIRE(v.npts / v.mesh[D - 1] < (int64_t)0xffffffffu && G.nchunks < (int64_t)0x7fffffff * 4, TBK_EUNSUPPORTED,
                    "tbk_wfs_solve_grid: mesh too large for 32-bit row indices");
        const int64_t nrows = v.npts / v.mesh[D - 1];
        // wave tiles that fill the chip (n = 3, 4: 16 wavefronts per compute unit are resident since round 5, and a tile is a long
        // chain of eigen-solves -- four rounds of shorter tiles measured 5 % faster than two rounds of tiles twice as long)
        const int64_t want = (int64_t)ctx->cus * (n > 2 ? 64 : 32);
        // the last column as the periodic image of the first (k_grid_rows): the whole last axis inside the window
        // (TBK_GRID_IMG=0: every column solved on its own, the A/B and the reference for the bit-identity test)
        const size_t lds_rows_need = ((size_t)4 * (n * (n + 1) / 2) * (2 * m->view.pmax + 1) + (size_t)4 * rows_stage_slots(n)) * sizeof(cd);
        const bool rows_kernel = lds_rows_need <= 48 * 1024 && tbk_knobs().grid_kernel != 1;     // (else k_grid_small: every column)
        G.img_last = rows_kernel && tbk_knobs().grid_img != 0 && v.mesh[D - 1] >= 2 && G.off[D - 1] == 0 && v.mesh[D - 1] == G.gmesh[D - 1] ? 1 : 0;
        if (G.img_last) {
            G.cpr = (v.mesh[D - 1] - 1 + 63) / 64;
            G.nchunks = nrows * G.cpr;
        }
        G.img_win = rows_kernel && n > 2 && !G.img_last && G.off[D - 1] + v.mesh[D - 1] == G.gmesh[D - 1] ? 1 : 0;
        G.seg = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, G.cpr), G.nchunks / want));
        if (tbk_knobs().grid_seg >= 0) G.seg = std::max(1, std::min(tbk_knobs().grid_seg, G.cpr));
        G.tpr = (G.cpr + G.seg - 1) / G.seg;
        G.seg = (G.cpr + G.tpr - 1) / G.tpr;             // balance the tiles of a row (33 chunks -> 7,7,7,7,5)
        G.ntiles = nrows * G.tpr;
        size_t lds = lds_rows_need;
        if (lds <= 48 * 1024 && tbk_knobs().grid_kernel != 1) {
            // (TBK_GRID_OCC: cap on the resident wavefronts per SIMD through the LDS request, as in the fused pass)
            // A dynamic LDS request above 64 KB needs a function attribute this launch does not set: occ = 1 (82 944 B) cannot
            // be honoured and is an explicit error, not an opaque launch failure or a silently different kernel
            TBK_REQUIRE(tbk_knobs().grid_occ != 1, TBK_EINVAL,
                        "TBK_GRID_OCC=1 asks for %d bytes of LDS per workgroup (limit 65536 without a function attribute): use 2..7",
                        160 * 1024 / 2 + 1024);
            if (tbk_knobs().grid_occ > 0 && tbk_knobs().grid_occ < 8)
                lds = std::max(lds, (size_t)(160 * 1024) / (size_t)(tbk_knobs().grid_occ + 1) + 1024);
            const unsigned blocks = (unsigned)((G.ntiles + 3) / 4);
            const int64_t npart = G.ntiles * std::max(n - 1, 1);
            if (w->gap_part_cap < npart) {
                TBK_HIP(hipStreamSynchronize(ctx->stream));
                if (w->gap_part_dev) TBK_HIP(hipFree(w->gap_part_dev));
                w->gap_part_dev = nullptr;
                w->gap_part_cap = 0;
                TBK_HIP(hipMalloc((void**)&w->gap_part_dev, (size_t)npart * sizeof(double)));
                w->gap_part_cap = npart;
            }
            G.gap_part = w->gap_part_dev;
            w->gap_part_n = G.ntiles;       // tbk_wfs_solve_grid_result reduces these instead of the shards
#define TBK_ROWS(NN, PP) hipLaunchKernelGGL((k_grid_rows<NN, PP>), dim3(blocks), dim3(256), lds, ctx->stream, m->view, G)
            const int pm = m->view.pmax;
            // (ranges 0..4 along the last axis are compiled in; the generic-range instance costs 1.7 x at 4 states:
            // profiles/hop_range_probe.py)
            switch (n * 8 + (pm <= 4 ? pm : 5)) {
                case 8 + 0: TBK_ROWS(1, 0); break;
                case 8 + 1: TBK_ROWS(1, 1); break;
                case 8 + 2: TBK_ROWS(1, 2); break;
                case 8 + 3: case 8 + 4: case 8 + 5: TBK_ROWS(1, -1); break;
                case 16 + 0: TBK_ROWS(2, 0); break;
                case 16 + 1: TBK_ROWS(2, 1); break;
                case 16 + 2: TBK_ROWS(2, 2); break;
                case 16 + 3: TBK_ROWS(2, 3); break;
                case 16 + 4: TBK_ROWS(2, 4); break;
                case 16 + 5: TBK_ROWS(2, -1); break;
                case 24 + 0: TBK_ROWS(3, 0); break;
                case 24 + 1: TBK_ROWS(3, 1); break;
                case 24 + 2: TBK_ROWS(3, 2); break;
                case 24 + 3: TBK_ROWS(3, 3); break;
                case 24 + 4: TBK_ROWS(3, 4); break;
                case 24 + 5: TBK_ROWS(3, -1); break;
                case 32 + 0: TBK_ROWS(4, 0); break;
                case 32 + 1: TBK_ROWS(4, 1); break;
                case 32 + 2: TBK_ROWS(4, 2); break;
                case 32 + 3: TBK_ROWS(4, 3); break;
                case 32 + 4: TBK_ROWS(4, 4); break;
                default: TBK_ROWS(4, -1); break;
            }
#undef TBK_ROWS
        } else {   // very long-ranged models: walk the term table per point
            const unsigned blocks = (unsigned)((G.nchunks + 3) / 4);
            switch (n) {
                case 1: hipLaunchKernelGGL((k_grid_small<1>), dim3(blocks), dim3(256), 0, ctx->stream, m->view, G); break;
                case 2: hipLaunchKernelGGL((k_grid_small<2>), dim3(blocks), dim3(256), 0, ctx->stream, m->view, G); break;
                case 3: hipLaunchKernelGGL((k_grid_small<3>), dim3(blocks), dim3(256), 0, ctx->stream, m->view, G); break;
                default: hipLaunchKernelGGL((k_grid_small<4>), dim3(blocks), dim3(256), 0, ctx->stream, m->view, G); break;
            }
        }
        TBK_HIP(hipGetLastError());
        return TBK_OK;
    }
    {   // wavefront-per-matrix path re-arms the other parity with a tiny memset-like kernel
        hipLaunchKernelGGL(k_arm_gaps, dim3(1), dim3(256), 0, ctx->stream, G.gaps_next, n);
    }
    ListArgs L{};
    return launch_wave<1, true>(ctx, m->view, n, v.npts, L, G);
}

static int solve_grid_result_once(tbk_wfs* w, double* min_gaps, bool can_retry) {
    tbk_ctx* ctx = w->ctx;
    if (w->gap_part_n > 0 && w->gaps_n > 0 && min_gaps) {   // the row kernel left per-tile minima
        // the reduced gaps go straight into the context's mapped host buffer when there is one: the kernel's stores are the
        // transfer, one synchronisation, no copy operation
        const size_t gb = (size_t)w->gaps_n * sizeof(double);
        void* zh = nullptr;
        void* zd = nullptr;
        int rc = tbk_knobs().zero_copy_kb > 0 ? tbk_ctx_zero_copy(ctx, gb, &zh, &zd) : TBK_OK;
        if (rc) return rc;
        double* out_dev = (double*)zd;
        if (!zh) {
            void* base = nullptr;
            rc = tbk_ctx_scratch(ctx, 256 + gb, &base);
            if (rc) return rc;
            out_dev = (double*)((unsigned char*)base + 256);
        }
        // with the gaps in mapped memory the reduction is the call's last kernel: it also copies the status words and stores the
        // completion word the host polls (tbk_done_wait) -- no hipStreamSynchronize, no second round trip for the status
        const DoneArgs done = zh ? tbk_done_arm(ctx, w->view.nsta > 2) : DoneArgs{nullptr, nullptr, nullptr, 0u};
        hipLaunchKernelGGL(k_gap_part_reduce, dim3(w->gaps_n), dim3(1024), 0, ctx->stream, w->gap_part_dev, w->gap_part_n,
                           w->gaps_n, out_dev, done);
        TBK_HIP(hipGetLastError());
        if (zh) {
            rc = tbk_done_wait(ctx, done);
            if (rc) return rc;
            memcpy(min_gaps, zh, gb);
            if (done.word && done.flags_src) {   // status words as the last kernel saw them: the usual case is "all clear"
                const volatile unsigned* f = ctx->done_host + 4;
                if (f[0] == 0u && f[2] == 0u) return TBK_OK;
            } else if (done.word) {
                return TBK_OK;                   // (n <= 2: closed forms, nothing iterates)
            }
        } else {
            rc = tbk_small_d2h(ctx, min_gaps, out_dev, gb);
            if (rc) return rc;
        }
        return check_noconv(ctx, w->view.nsta);
    }
    const size_t half = (size_t)TBK_GAP_SHARDS * w->view.ncomp;
    std::vector<unsigned long long> bits(half);
    if (w->gap_part_n == 0 && w->gaps_n > 0 && min_gaps) {
        TBK_HIP(hipMemcpyAsync(bits.data(), w->gaps_dev + (size_t)(1 - w->gaps_parity) * half,
                               half * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    if (w->gaps_n > 0 && min_gaps) {
        for (int b = 0; b < w->gaps_n; ++b) {
            unsigned long long best = bits[b];
            for (int s = 1; s < TBK_GAP_SHARDS; ++s) best = std::min(best, bits[(size_t)s * w->view.ncomp + b]);
            memcpy(&min_gaps[b], &best, sizeof(double));
        }
    }
    return check_noconv(ctx, w->view.nsta, can_retry);
}

extern "C" int tbk_wfs_solve_grid_result(tbk_wfs* w, double* min_gaps) {
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_wfs_solve_grid_result: null wfs");
    int rc = solve_grid_result_once(w, min_gaps, w->last_model != nullptr);
    if (rc == TBK_ERETRY_JACOBI) {   // the QL rotation record overflowed: the same launch again on the Jacobi kernels
        tbk_ctx* ctx = w->ctx;
        ctx->qlw_off = true;
        rc = tbk_wfs_solve_window_async(w, w->last_model, w->last_start, w->last_pbc.data(), w->last_off, w->last_gmesh);
        if (rc == TBK_OK) rc = solve_grid_result_once(w, min_gaps, false);
        ctx->qlw_off = false;
    }
    return rc;
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
    const int64_t per_band = nface * v.ncomp;
    if (idx >= per_band * v.nsta) return;
    const int band = (int)(idx / per_band);
    const int64_t r = idx - band * per_band;
    const int64_t f = r / v.ncomp;
    const int o = (int)(r - f * v.ncomp);
    // face index f enumerates all mesh points with index 0 along dir
    const int64_t inner = v.stride[dir];           // points per unit step of dir
    const int64_t outer = f / inner, in = f - outer * inner;
    const int64_t p0 = outer * inner * v.mesh[dir] + in;
    const int64_t p1 = p0 + (int64_t)(v.mesh[dir] - 1) * inner;
    cd val = wf_at(v, band, p0)[o];
    if (phase) val = cmul(val, phase[o]);
    wf_at(v, band, p1)[o] = val;
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
