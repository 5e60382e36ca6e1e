// tbk_solve_dev.h -- device-side pieces shared by the translation units of the eigen-solver (tbk_solve.hip and
// tbk_solve_e16.hip): the argument blocks of the list / mesh kernels, exp(2 pi i x), the decoding of a mesh index into
// its k-point, 1/sqrt to full precision, the DPP helpers of the 16-lanes-per-matrix kernels and one position of the
// implicit-QL sweep on a tridiagonal matrix held by ONE lane.  Everything here is `__device__ __forceinline__` or a
// template: including it in two translation units defines nothing twice.
#ifndef TBK_SOLVE_DEV_H
#define TBK_SOLVE_DEV_H
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
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
    double* gap_part;               // k_grid_rows: [ntiles][n-1] per-tile minima (no atomics)
    // per-axis tables of the regular mesh: z[d][i] = exp(2 pi i k_d(i)),
    // f[d][i*n+o] = exp(-2 pi i k_d(i) tau_o,d) * (pbc phase if i is the periodic image)
    const cd* tz[TBK_MAX_DIM];
    const cd* tf[TBK_MAX_DIM];
    int last;                // index of the last (fastest) mesh axis
    int cpr;                 // 64-point chunks per mesh row (row = all leading axes)
    int64_t nchunks;
    int wnchunk;             // k_solve_wave: aligned chains per mesh row that touch the window
    int64_t wcfirst;         // ... and the global number of the first of them
    int* flags;              // the context's sticky status words ([0]: an eigen-solver ran into its iteration cap)
    int img_last;            // k_grid_rows: the window's last column is the periodic image of its first (the whole last axis is
                             // inside the window): it is not chunked, the lane that solves column 0 stores it too
    int img_win;             // k_grid_rows, n = 3, 4: the window's last column is the global mesh's periodic image but column 0 is
                             // not solved with it (a window that starts past column 0, or TBK_GRID_IMG=0): the lane that solves
                             // it forms the image exactly as the lane of column 0 would -- column 0's vector under column 0's
                             // phases (tf0), then x (tf_image conj tf_0) -- so windows stay bit-identical to the whole array
    const cd* tf0;           // [nsta]: the last axis' orbital phases at GLOBAL index 0 (= tf[last][0..n) of a window that holds column 0)
    int seg;                 // chunks per wave tile (k_grid_rows)
    int tpr;                 // wave tiles per row
    int64_t ntiles;
    int reg_cells;           // k_solve_regd on a mesh: S(k) from the row's coefficient cells (reg_assemble_cells; TBK_REG_CELLS=0: the tiled sum)
#ifdef TBK_DIAG
    int ablate;              // diagnostic build only (TBK_ABLATE_GRID): 1 = no stores, 2 = no eigen-solve, ...
#endif
};

struct ListArgs {
    const double* k;  // [nk][dim_k]
    const cd* ham;    // [nk][n][n] (eigh of supplied matrices) or null
    double* eval;     // [n][nk]
    cd* evec;         // [n][nk][n] or null
    int* flags;       // the context's sticky status words ([0]: an eigen-solver ran into its iteration cap); null: not reported
    int natural;      // k_solve_row16 only: leave the eigenpairs in Jacobi's own order (column j grown out of e_j, so the
                      // eigenvector matrix stays as close to the identity as the rotations allow) instead of sorting
    DoneArgs done;    // k_solve_small only (k lists of <= 4 states whose results lie in mapped host memory): completion word for
                      // the host to poll (tbk_done_wait); word == null: not armed
};

// "entry j (value o) comes before entry x (value mine)" in an ascending order that is TOTAL even with NaNs (they sort last,
// ties by index): the ranks counted with it are a permutation whatever the input -- a NaN Hamiltonian must end in
// TBK_ENOCONV, not in a scatter through a half-filled permutation array
__device__ __forceinline__ bool tbk_before(const double o, const double mine, const int j, const int x) {
    const bool on = o != o, mn = mine != mine;
    return on ? (mn && j < x) : (mn || o < mine || (o == mine && j < x));
}

// exp(2 pi i x).  Exact argument reduction (2x - rint(2x) and the quadrant are exact in binary floating point, like sincospi's),
// then sin(pi t) / t and cos(pi t) on |t| <= 1/4 as polynomials in t^2 (Chebyshev fits computed with mpmath at 60 digits: fit
// errors 3e-18 and 3e-20 relative; measured against mpmath on 10^4 arguments: <= 1 ulp each).  32 VALU instructions where the
// library's sincospi is 90 -- 50 of them copies of its coefficients into vector registers; the k-list kernels of 2..4 states
// were VALU-bound on exactly that (configs[1]: 455 instructions per k-point, two calls).
__device__ __forceinline__ cd expi2pi(const double x) {
    const double f = x - __builtin_rint(x);            // [-1/2, 1/2]: the angle is 2 pi f
    const double q = __builtin_rint(4.0 * f);          // -2 .. 2: the quadrant
    const double h = fma(q, -0.25, f);                 // [-1/8, 1/8]
    const double t = h + h;                            // the angle is pi t + q pi / 2, |t| <= 1/4
    const int m = (int)q & 3;
    const double u = t * t;
    double sp = tbk_add_vs(tbk_mul_vs(u, 0.00046153185538358102), -0.0073700215869077707);
    sp = tbk_fma_vs(sp, u, 0.082145869180001746);
    sp = tbk_fma_vs(sp, u, -0.59926452893964488);
    sp = tbk_fma_vs(sp, u, 2.5501640398733763);
    sp = tbk_fma_vs(sp, u, -5.1677127800499543);
    sp = tbk_fma_vs(sp, u, 3.1415926535897931);
    sp *= t;
    double cp = tbk_add_vs(tbk_mul_vs(u, -0.00010356747255199479), 0.0019294657440800042);
    cp = tbk_fma_vs(cp, u, -0.025806885652951306);
    cp = tbk_fma_vs(cp, u, 0.23533063019088787);
    cp = tbk_fma_vs(cp, u, -1.3352627688519174);
    cp = tbk_fma_vs(cp, u, 4.0587121264167472);
    cp = tbk_fma_vs(cp, u, -4.934802200544679);
    cp = tbk_fma_vs(cp, u, 1.0);
    // quadrant m = q mod 4:  0: (c, s) = (cp, sp);  1: (-sp, cp);  2: (-cp, -sp);  3: (sp, -cp)
    const bool swap = (m & 1) != 0;
    double c = swap ? sp : cp, s = swap ? cp : sp;
    c = ((m + 1) & 2) ? -c : c;
    s = (m & 2) ? -s : s;
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

// 1 / sqrt(t) to full double precision from the hardware estimate (relative error 5e-8 measured,
// profiles/microbench/rsq_precision.hip) by one cubically convergent step:
// with e = 1 - t y^2,  1/sqrt(t) = y (1 - e)^(-1/2) = y (1 + e/2 + 3 e^2/8 + O(e^3)),  e^3 ~ 1e-21
__device__ __forceinline__ double rsqrt_full(const double t) {
    const double y = __builtin_amdgcn_rsq(t);
    const double e = fma(-t * y, y, 1.0);
    const double ye = y * e;
    return fma(ye, fma(e, 0.375, 0.5), y);
}

// ---- one DPP row of 16 lanes per matrix (tbk_solve_row16.inl, tbk_solve_ql16.inl, tbk_solve_tw16.inl, tbk_solve_e16.hip)
struct I2 {
    int lo, hi;
};
__device__ __forceinline__ double bperm_d(const int addr, const double v) {
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_ds_bpermute(addr, i.lo), __builtin_amdgcn_ds_bpermute(addr, i.hi)};
    return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ cd bperm_c(const int addr, const cd v) { return cd{bperm_d(addr, v.x), bperm_d(addr, v.y)}; }
__device__ __forceinline__ double perm_push_d(const int addr, const double v) {   // lane sends v to lane addr/4
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_ds_permute(addr, i.lo), __builtin_amdgcn_ds_permute(addr, i.hi)};
    return __builtin_bit_cast(double, o);
}
template <int SRC>
__device__ __forceinline__ double rowbcast_d(const double v) {   // value of lane SRC of each 16-lane row
    const I2 i = __builtin_bit_cast(I2, v);
    // (bound_ctrl with full row/bank masks: the old value is dead, so the compiler emits ONE v_mov_b32_dpp per dword --
    // with a live old operand it was a plain v_mov to seed the destination plus the DPP move)
    const I2 o{__builtin_amdgcn_update_dpp(0, i.lo, 0x150 + SRC, 0xf, 0xf, true),
               __builtin_amdgcn_update_dpp(0, i.hi, 0x150 + SRC, 0xf, 0xf, true)};
    return __builtin_bit_cast(double, o);
}
template <int SRC>
__device__ __forceinline__ int rowbcast_i(const int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + SRC, 0xf, 0xf, true); }

template <int J>
__device__ __forceinline__ cd sel16(const cd (&a)[16], const int idx, const cd acc) {
    const cd r = idx == J ? a[J] : acc;
    if constexpr (J + 1 < 16) return sel16<J + 1>(a, idx, r);
    else return r;
}

template <int N>
__device__ __forceinline__ double row_ror_d(const double v) {   // value of lane (x + N) mod 16 of the same row
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_update_dpp(0, i.lo, 0x120 + N, 0xf, 0xf, true),
               __builtin_amdgcn_update_dpp(0, i.hi, 0x120 + N, 0xf, 0xf, true)};
    return __builtin_bit_cast(double, o);
}
// sum over the 16 lanes of a row, the same bits in every lane (each step adds a value to its mirror image)
__device__ __forceinline__ double row_allsum(double v) {
    v += row_ror_d<8>(v);
    v += row_ror_d<4>(v);
    v += row_ror_d<2>(v);
    v += row_ror_d<1>(v);
    return v;
}
template <int SRC>
__device__ __forceinline__ cd rowbcast_c(const cd v) { return cd{rowbcast_d<SRC>(v.x), rowbcast_d<SRC>(v.y)}; }

#define TBK_QL_MAX_ITER 480   // 30 shifts per eigenvalue, LAPACK's limit

// ---- implicit QL on (d, e) held by one lane (static register indices, EXEC-masked full-range sweeps)
// REC: the rotation of position I goes to rot[I * stride]; lstop = lowest position rotated so far
template <int I, bool REC = false>
__device__ __forceinline__ void qle_pos(double (&d)[16], double (&e)[16], double& sn, double& cs, double& pp, double& g, bool& alive,
                                        const bool live, const int l, const int m, double2* __restrict__ rot = nullptr,
                                        int* lstop = nullptr, const int64_t stride = 0) {
    if (live && alive && I >= l && I < m) {
        const double f = sn * e[I], b = cs * e[I];
        const double t = f * f + g * g;
        if (t > 0.0) {
            const double inv = rsqrt_full(t), r = t * inv;
            e[I + 1] = I + 1 == m ? 0.0 : r;     // (e_m is zeroed at the end of a sweep)
            sn = f * inv;
            cs = g * inv;
            const double gg = d[I + 1] - pp;
            const double r2 = (d[I] - gg) * sn + 2.0 * cs * b;
            pp = sn * r2;
            d[I + 1] = gg + pp;
            g = cs * r2 - b;
            if (I == l) {                        // last position of the sweep
                d[I] -= pp;
                e[I] = g;
            }
            if constexpr (REC) {
                rot[(int64_t)I * stride] = double2{cs, sn};
                *lstop = I;
            }
        } else {                                 // r == 0 (underflow): tql2's recovery
            d[I + 1] -= pp;
            e[I + 1] = 0.0;                          // (e_{i+1} = r = 0)
            alive = false;
        }
    }
    if constexpr (I > 0) qle_pos<I - 1, REC>(d, e, sn, cs, pp, g, alive, live, l, m, rot, lstop, stride);
}
template <int J>
__device__ __forceinline__ double qle_pick(const double (&a)[16], const int idx, const double acc) {
    const double r = idx == J ? a[J] : acc;
    if constexpr (J + 1 < 16) return qle_pick<J + 1>(a, idx, r);
    else return r;
}

// tbk_solve_e16.hip: the fused n = 9..16 solver with eigenvectors (one kernel, reflectors in LDS)
// `form`: E16_F_* bits.  They select a FORM of the kernel and must depend on the model and the knobs alone, never on the window:
// the bit-identity of windows and shards rests on every window of a mesh taking the same form.
enum : int {
    E16_F_NS_FULL = 1,    // TBK_E16_NS_FULL=1: the full Newton-Schulz step (matrix cores) for EVERY matrix, as round 4 did
    E16_F_NO_CELLS = 2,   // TBK_E16_CELLS=0: S(k) from the tiled sum over the lattice vectors, not from the row's coefficient cells
};
int tbk_e16_launch(int mode, hipStream_t stream, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G, int64_t id0,
                   int64_t nc, int* list, int* count, double gaptol, int form);
int tbk_e16_launch_evals(int mode, hipStream_t stream, const ModelView& mv, int64_t nk, const ListArgs& L, int64_t id0, int64_t nc);

#endif  // TBK_SOLVE_DEV_H
