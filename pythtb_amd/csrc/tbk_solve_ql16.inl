// tbk_solve_ql16.inl -- included by tbk_solve.hip (after tbk_solve_row16.inl, whose DPP helpers it uses).
//
// n = 9..16 states per k: a DIRECT Hermitian eigen-solver in registers -- Householder
// tridiagonalisation followed by implicit-shift QL with accumulated eigenvectors (what LAPACK's
// zhetrd + zsteqr do for the reference's numpy.linalg.eigh, pythtb.py:939-947) -- one DPP row of 16
// lanes per matrix, four matrices per wavefront, no LDS storage, no barriers.
//
// Why not Jacobi here: a cyclic Jacobi sweep of a 16 x 16 matrix is 120 rotations of O(n) work on A and
// V each, and 4-7 sweeps are needed (~1e6 flops per matrix); tridiagonalisation is (16/3) n^3 flops once
// and QL converges in ~1.7 shifts per eigenvalue with O(n) rotations each (~1e5 flops per matrix).  The
// LDS wavefront-per-matrix Jacobi kernel issues 18.5 k VALU + 2.7 k LDS wave-instructions per matrix
// (profiles/r02a); this kernel ~7 k VALU per matrix and nothing else.
//
// Layout.  Lane x of a row holds ROW x of A (a[16]) and ROW x of the accumulated transformation Z
// (z[b] = component x of eigenvector b at the end).  The tridiagonal (d, e) is DISTRIBUTED: lane j keeps
// d_j and e_j (the coupling between j and j+1; e_15 = 0).
//
//  1. Householder, step K = 0..13 (unrolled: register indices are static).  The column below the diagonal
//     is one register per lane (a[K] of lanes x > K), so forming u needs one row reduction; p = beta A u and
//     w = beta Z u are local dot products of the lane's rows with u, whose elements arrive one at a time by
//     DPP row broadcast; the rank-2 update A -= u q^+ + q u^+ and Z -= w u^+ are local too.  H_K = I - beta
//     u u^+ is Hermitian and maps the column onto -phase * norm * e_{K+1}; the complex subdiagonal is made
//     real by a diagonal unitary D folded into Z column by column (Z = H_0 ... H_13 D).
//  2. Implicit QL (EISPACK tql2 / LAPACK zsteqr recurrences).  All four matrices of a wavefront run the
//     same instruction stream: every iteration is one shift and one FULL-range sweep i = 14 .. 0 in which a
//     position takes part only if it lies in that matrix's active block [l, m) (EXEC-masked).  The
//     rotation recurrence is scalar work replicated over the 16 lanes of a row; each lane applies the
//     rotation to its own row of Z (columns i, i+1: static registers).  d_i, d_{i+1}, e_i are read by row
//     broadcast -- a sweep only ever reads values from before the sweep -- and the new d_{i+1}, e_{i+1} are
//     kept by their owner lane.  Negligible couplings are found by one compare + ballot per iteration.
//
// Every point is solved on its own (no warm start), so periodic images, halo rows and shard windows are
// bit-identical by construction.  Matrices smaller than 16 are padded with decoupled zero rows (they
// never mix and are ranked last).

// (row_ror_d, row_allsum, rowbcast_c: tbk_solve_dev.h)

// ---- Householder step K: pass 1 (p = A u, w = Z u over the columns c > K) and pass 2 (rank-2 updates)
template <int K, int CIDX, bool VEC>
__device__ __forceinline__ void ql16_pass1(const cd (&a)[16], const cd (&z)[16], const cd u, cd& p, cd& w) {
    const cd uc = rowbcast_c<CIDX>(u);
    cfma(p, a[CIDX], uc);
    if (VEC) cfma(w, z[CIDX], uc);
    if constexpr (CIDX + 1 < 16) ql16_pass1<K, CIDX + 1, VEC>(a, z, u, p, w);
}
template <int K, int CIDX, bool VEC>
__device__ __forceinline__ void ql16_pass2(cd (&a)[16], cd (&z)[16], const cd u, const cd q, const cd w) {
    const cd uc = rowbcast_c<CIDX>(u), qc = rowbcast_c<CIDX>(q);
    // A[x][c] -= u_x conj(q_c) + q_x conj(u_c)
    a[CIDX].x -= (u.x * qc.x + u.y * qc.y) + (q.x * uc.x + q.y * uc.y);
    a[CIDX].y -= (u.y * qc.x - u.x * qc.y) + (q.y * uc.x - q.x * uc.y);
    if (VEC) {   // Z[x][c] -= w_x conj(u_c)
        z[CIDX].x -= w.x * uc.x + w.y * uc.y;
        z[CIDX].y -= w.y * uc.x - w.x * uc.y;
    }
    if constexpr (CIDX + 1 < 16) ql16_pass2<K, CIDX + 1, VEC>(a, z, u, q, w);
}

// returns the complex subdiagonal element t_K = T[K+1][K] (the same in every lane of the row)
template <int K, bool VEC>
__device__ __forceinline__ cd ql16_house(cd (&a)[16], cd (&z)[16], const int x) {
    const bool below = x > K;
    const cd xk = below ? a[K] : cd{0.0, 0.0};
    // |rows > K+1 of the column|^2: a reflection is needed iff this is non-zero.  (Decided on this part ALONE, like LAPACK's
    // zlarfg: through the sum with |alpha|^2, entries below ~1e-8 |alpha| would be dropped -- an eigenvalue error of up to
    // their size; met on ribbon Hamiltonians near k = 0, whose imaginary parts are that small.)
    const double rest = row_allsum(x > K + 1 ? cabs2(xk) : 0.0);
    const cd alpha = rowbcast_c<K + 1>(a[K]);            // A[K+1][K]
    const double absa2 = cabs2(alpha);
    const double sigma = rest + absa2;                   // |column below the diagonal|^2
    cd tK{0.0, 0.0};
    if (rest > 0.0) {                                    // row-uniform
        const double inv_n = rsqrt_full(sigma), nrm = sigma * inv_n;
        double absa = 0.0;
        cd ph{1.0, 0.0};
        if (absa2 > 0.0) {
            const double inv_a = rsqrt_full(absa2);
            absa = absa2 * inv_a;
            ph = cd{alpha.x * inv_a, alpha.y * inv_a};
        }
        // u = column + phase * norm * e_{K+1};  H u-reflection maps the column to -phase * norm * e_{K+1}
        const cd u = x == K + 1 ? cd{ph.x * (absa + nrm), ph.y * (absa + nrm)} : xk;
        const double beta = 1.0 / (nrm * (nrm + absa));  // 2 / (u^+ u)
        tK = cd{-ph.x * nrm, -ph.y * nrm};
        cd p{0.0, 0.0}, w{0.0, 0.0};
        ql16_pass1<K, K + 1, VEC>(a, z, u, p, w);
        p = cd{p.x * beta, p.y * beta};
        w = cd{w.x * beta, w.y * beta};
        // kappa = beta/2 u^+ p  (real: u^+ A u of a Hermitian A; u_x = 0 for x <= K masks those lanes' p)
        const double kappa = 0.5 * beta * row_allsum(u.x * p.x + u.y * p.y);
        const cd q = below ? cd{p.x - kappa * u.x, p.y - kappa * u.y} : cd{0.0, 0.0};
        ql16_pass2<K, K + 1, VEC>(a, z, u, q, w);
    } else {
        tK = alpha;
    }
    return tK;
}

struct Ql16State {
    double s, c, p, g;   // rotation recurrence (replicated over the lanes of a row)
    bool alive;          // false once this matrix's sweep hit an exact-zero rotation (underflow guard)
};

// position I of a sweep: rotation in the (I, I+1) plane
// (di1 = d_{I+1} as it was before the sweep: the caller's own d_I broadcast, taken before position I+1 could change it --
// it cannot: position I+1 writes lane I+2)
template <int I, bool VEC>
__device__ __forceinline__ void ql16_pos(Ql16State& S, cd (&z)[16], double& dd, double& ee, const int x, const int l, const int m,
                                         const double di1) {
    const double di = rowbcast_d<I>(dd), ei = rowbcast_d<I>(ee);   // pre-sweep values
    if (S.alive && I >= l && I < m) {
        const double f = S.s * ei, b = S.c * ei;
        const double t = f * f + S.g * S.g;
        if (t > 0.0) {
            const double inv = rsqrt_full(t), r = t * inv;
            S.s = f * inv;
            S.c = S.g * inv;
            double g = di1 - S.p;
            const double r2 = (di - g) * S.s + 2.0 * S.c * b;
            S.p = S.s * r2;
            if (x == I + 1) {
                ee = r;
                dd = g + S.p;
            }
            S.g = S.c * r2 - b;
            if (VEC) {
                const cd zi = z[I], zj = z[I + 1];
                z[I + 1] = cd{S.s * zi.x + S.c * zj.x, S.s * zi.y + S.c * zj.y};
                z[I] = cd{S.c * zi.x - S.s * zj.x, S.c * zi.y - S.s * zj.y};
            }
        } else {   // r == 0 (underflow): tql2's recovery -- d[i+1] -= p, e[m] = 0, start the block over
            if (x == I + 1) {
                dd -= S.p;
                ee = 0.0;                              // (e_{i+1} = r = 0)
            }
            S.alive = false;
        }
    }
    if constexpr (I > 0) ql16_pos<I - 1, VEC>(S, z, dd, ee, x, l, m, di);
}

template <int J>
__device__ __forceinline__ void ql16_bcast_ev(const double dd, double (&ev)[16]) {
    ev[J] = rowbcast_d<J>(dd);
    if constexpr (J + 1 < 16) ql16_bcast_ev<J + 1>(dd, ev);
}

// (TBK_QL_MAX_ITER, qle_pos, qle_pick: tbk_solve_dev.h)

// ---- eigenvalues only: after the tridiagonalisation nothing is left to do on Z, and the QL recurrence replicated over the
// 16 lanes of a matrix would be ALL of the remaining work (~5 k wave-instructions per matrix).  So the eigenvalue-only solve
// is two kernels: k_solve_ql16<.., false, 1> stops after the tridiagonalisation and leaves (d_j, e_j) in a workspace laid out
// [j][matrix]; k_tridiag_eigvals then gives every LANE one matrix -- static register indices, nothing replicated, coalesced
// loads and stores -- ~0.4 k wave-instructions per matrix.
// de[j * nk + id] = (d_j, e_j) of matrix id  ->  eval[rank][id], ascending (the n real rows; padding rows rank last)
__global__ __launch_bounds__(256) void k_tridiag_eigvals(const int n, const int64_t nk, const double2* __restrict__ de,
                                                         double* __restrict__ eval, int* noconv_flag) {
    const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool has = id < nk;
    const int64_t ic = has ? id : nk - 1;
    double d[16], e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double2 v = de[(int64_t)j * nk + ic];
        d[j] = v.x;
        e[j] = v.y;
    }
    e[15] = 0.0;
    int l = 0;
    bool done = !has;
    for (int iter = 0;; ++iter) {
        int m = 15;
        if (!done) {
            unsigned negl = 0x8000u;
#pragma unroll
            for (int j = 0; j < 15; ++j)
                negl |= fabs(e[j]) <= 2.220446049250313e-16 * (fabs(d[j]) + fabs(d[j + 1])) ? (1u << j) : 0u;
            const unsigned open = ~negl & (0xffffu << l) & 0xffffu;
            if (open == 0) {
                done = true;
            } else {
                l = __builtin_ctz(open);
                m = __builtin_ctz(negl & (0xffffu << l));
            }
        }
        if (__all(done)) break;
        if (iter >= TBK_QL_MAX_ITER) {
            if (!done) atomicExch(noconv_flag, 1);
            break;
        }
        double sn = 1.0, cs = 1.0, pp = 0.0, g = 0.0;
        bool alive = true;
        if (!done) {
            const double dl = qle_pick<0>(d, l, 0.0), dl1 = qle_pick<0>(d, l + 1, 0.0);
            const double el = qle_pick<0>(e, l, 1.0), dmm = qle_pick<0>(d, m, 0.0);
            const double gs = (dl1 - dl) * (0.5 * __builtin_amdgcn_rcp(el));
            const double r = __builtin_amdgcn_sqrt(fma(gs, gs, 1.0));
            g = dmm - dl + el * __builtin_amdgcn_rcp(gs + copysign(r, gs));
        }
        qle_pos<14>(d, e, sn, cs, pp, g, alive, !done, l, m);
    }
    // stable ascending ranks among the n real entries, then one coalesced store per rank
    int rk[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) r += (b < n) && (d[b] < d[a] || (d[b] == d[a] && b < a)) ? 1 : 0;
        rk[a] = r;
    }
    if (!has) return;
    for (int r = 0; r < n; ++r) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 16; ++a) v = (a < n && rk[a] == r) ? d[a] : v;
        eval[(int64_t)r * nk + id] = v;
    }
}

// ---- with eigenvectors, three-kernel form (TBK_QL16_SPLIT): the rotation recurrence replicated over the 16 lanes of a matrix
// is ~60 % of k_solve_ql16's instructions.  Here it runs ONCE per matrix -- one lane per matrix, like k_tridiag_eigvals --
// and leaves the rotations of every sweep in a record; k_ql16_replay then applies them to the rows of Z with the 16 lanes of
// a matrix (no recurrence, one coalesced load of the sweep's rotations per iteration).
// The record is laid out [sweep][position][matrix]: the lanes of the QL kernel (consecutive matrices, in step through sweeps
// and positions) write consecutive 16-byte entries, and the four matrices of a replay wavefront share each 64-byte sector
// they read.  Only the entries of positions a sweep rotated are ever touched.
struct Ql16Rec {
    double2* rot;      // [scap][16][nc]  (c, s) of sweep it at position i
    unsigned* swp;     // [scap][nc]      per sweep: lowest position rotated | m << 8   (rotations m-1 .. lowest)
    int* nit;          // [nc]
    signed char* rank; // [16][nc]        ascending rank of column b (padding columns rank last)
    int scap;
};

// LIST MODE (list != nullptr; tbk_solve_tw16.inl's fallback): the launch covers the *count matrices id0 + list[0 .. *count) of
// the chunk.  count is read on the device, so the grid is a fixed small one whose blocks stride over the list (an empty
// list costs a few idle blocks); workspace entries are indexed by the position in the list; the mesh's minimum gaps have
// been taken already.
template <int MODE>
__device__ __forceinline__ void ql16_lanes_body(const int n, const int64_t nk, const int64_t id0, const int64_t nc,
                                                const double2* __restrict__ de, double* __restrict__ eval, const GridArgs& G,
                                                const Ql16Rec& R, int* flags, const int* __restrict__ list, const int64_t nhave,
                                                const int64_t idc, const bool list_gaps = false) {
    const bool has = idc < nhave;
    const int64_t ic = has ? idc : nhave - 1;
    const int64_t mat_out = id0 + (list ? (int64_t)list[ic] : idc);
    double d[16], e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double2 v = de[(int64_t)j * nc + ic];
        d[j] = v.x;
        e[j] = v.y;
    }
    e[15] = 0.0;
    int it = 0;
    bool overflow = false;
    int l = 0;
    bool done = !has;
    for (int iter = 0;; ++iter) {
        int m = 15;
        if (!done) {
            unsigned negl = 0x8000u;
#pragma unroll
            for (int j = 0; j < 15; ++j)
                negl |= fabs(e[j]) <= 2.220446049250313e-16 * (fabs(d[j]) + fabs(d[j + 1])) ? (1u << j) : 0u;
            const unsigned open = ~negl & (0xffffu << l) & 0xffffu;
            if (open == 0) {
                done = true;
            } else {
                l = __builtin_ctz(open);
                m = __builtin_ctz(negl & (0xffffu << l));
            }
        }
        if (__all(done)) break;
        if (iter >= TBK_QL_MAX_ITER) {
            if (!done) atomicExch(flags, 1);
            break;
        }
        double sn = 1.0, cs = 1.0, pp = 0.0, g = 0.0;
        bool alive = true;
        if (!done) {
            const double dl = qle_pick<0>(d, l, 0.0), dl1 = qle_pick<0>(d, l + 1, 0.0);
            const double el = qle_pick<0>(e, l, 1.0), dmm = qle_pick<0>(d, m, 0.0);
            const double gs = (dl1 - dl) * (0.5 * __builtin_amdgcn_rcp(el));
            const double r = __builtin_amdgcn_sqrt(fma(gs, gs, 1.0));
            g = dmm - dl + el * __builtin_amdgcn_rcp(gs + copysign(r, gs));
        }
        // (a sweep that would not fit the record is not written at all: the lane flags the overflow and goes on unrecorded)
        const bool fits = it < R.scap;   // (a sweep that does not fit is not recorded: the lane flags the overflow and goes on)
        int lstop = m;
        qle_pos<14, true>(d, e, sn, cs, pp, g, alive, !done && fits, l, m, R.rot + (int64_t)it * 16 * nc + ic, &lstop, nc);
        if (!done && !fits) {
            overflow = true;
            qle_pos<14, false>(d, e, sn, cs, pp, g, alive, true, l, m);
        }
        if (!done && fits) {
            R.swp[(int64_t)it * nc + ic] = (unsigned)lstop | ((unsigned)m << 8);
            ++it;
        }
    }
    if (has) R.nit[idc] = it;
    if (overflow) atomicExch(flags + 2, 1);
    // stable ascending ranks among the n real entries (padding columns rank last, in place)
    double prev = 0.0;
    int rk[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const bool before = a < n ? (b < n && (d[b] < d[a] || (d[b] == d[a] && b < a))) : (b < n || b < a);
            r += before ? 1 : 0;
        }
        rk[a] = r;
        if (has) R.rank[(int64_t)a * nc + idc] = (signed char)r;
    }
    for (int r = 0; r < n; ++r) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 16; ++a) v = (a < n && rk[a] == r) ? d[a] : v;
        if constexpr (MODE == 1) {
            if (r > 0 && (list == nullptr || list_gaps)) {   // (list_gaps: the caller left the listed matrices' gaps to this kernel -- k_e16)
                double gap = has ? v - prev : INFINITY;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gap = fmin(gap, __shfl_xor(gap, o));
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* slot = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + (r - 1);
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
                    if (bits < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, bits);
                }
            }
            prev = v;
        } else {
            if (has) eval[(int64_t)r * nk + mat_out] = v;
        }
    }
}

template <int MODE, bool LIST = false>
__global__ __launch_bounds__(256) void k_ql16_lanes(const int n, const int64_t nk, const int64_t id0, const int64_t nc,
                                                    const double2* __restrict__ de, double* __restrict__ eval, const GridArgs G,
                                                    const Ql16Rec R, int* flags, const int* __restrict__ list = nullptr,
                                                    const int* __restrict__ count = nullptr, const bool list_gaps = false) {
    if constexpr (!LIST) {
        ql16_lanes_body<MODE>(n, nk, id0, nc, de, eval, G, R, flags, nullptr, nc, (int64_t)blockIdx.x * 256 + threadIdx.x);
        return;
    }
    const int64_t nhave = *count;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < nhave; base += (int64_t)gridDim.x * 256)
        ql16_lanes_body<MODE>(n, nk, id0, nc, de, eval, G, R, flags, list, nhave, base + threadIdx.x, list_gaps);
}

// position I of a recorded sweep over [lo, m): lane x of the matrix loaded the rotation of position x into `mine`
template <int I>
__device__ __forceinline__ void ql16_replay_pos(cd (&z)[16], const double2 mine, const bool on, const int lo, const int m) {
    const double c = rowbcast_d<I>(mine.x), s = rowbcast_d<I>(mine.y);
    if (on && I >= lo && I < m) {
        const cd zi = z[I], zj = z[I + 1];
        z[I + 1] = cd{s * zi.x + c * zj.x, s * zi.y + c * zj.y};
        z[I] = cd{c * zi.x - s * zj.x, c * zi.y - s * zj.y};
    }
    if constexpr (I > 0) ql16_replay_pos<I - 1>(z, mine, on, lo, m);
}

template <int MODE>
__device__ __forceinline__ void ql16_replay_body(const int n, const int64_t nk, const int64_t id0, const int64_t nc, const Ql16Rec& R,
                                                 cd* __restrict__ evec, const WfsView& wv, const int* __restrict__ list,
                                                 const int64_t nhave, const int64_t idc0) {
    const int lane = threadIdx.x & 63;
    const int x = lane & 15;
    const bool live = idc0 < nhave;
    const int64_t idc = live ? idc0 : nhave - 1, id = id0 + (list ? (int64_t)list[idc] : idc);
    const bool real_row = x < n;
    const int xr = real_row ? x : n - 1;
    cd z[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        const int bb = b < n ? b : n - 1;
        if constexpr (MODE == 1) z[b] = wf_at(wv, bb, id)[xr];
        else z[b] = evec[((int64_t)bb * nk + id) * n + xr];
    }
    const double2* __restrict__ rot = R.rot + idc;     // entry (it, i): rot[(it * 16 + i) * nc]
    const unsigned* __restrict__ swp = R.swp + idc;    // sweep it: swp[it * nc]
    const int nit = live ? R.nit[idc] : 0;
    int nmax = nit;
    nmax = max(nmax, __shfl_xor(nmax, 16));
    nmax = max(nmax, __shfl_xor(nmax, 32));
    // the sweep word and this lane's rotation of the NEXT sweep are loaded while the current one is applied
    unsigned w_nx = nit > 0 ? swp[0] : 0u;
    double2 r_nx{1.0, 0.0};
    {
        const int lo = (int)(w_nx & 0xffu), m = (int)(w_nx >> 8);
        if (nit > 0 && x >= lo && x < m) r_nx = rot[(int64_t)x * nc];
    }
    for (int it = 0; it < nmax; ++it) {
        const bool on = it < nit;
        const unsigned w = w_nx;
        const double2 mine = r_nx;
        const int lo = (int)(w & 0xffu), m = (int)(w >> 8);
        if (it + 1 < nit) {
            w_nx = swp[(int64_t)(it + 1) * nc];
            const int lo2 = (int)(w_nx & 0xffu), m2 = (int)(w_nx >> 8);
            r_nx = (x >= lo2 && x < m2) ? rot[((int64_t)(it + 1) * 16 + x) * nc] : double2{1.0, 0.0};
        }
        ql16_replay_pos<14>(z, mine, on, lo, m);
    }
    if (!live || !real_row) return;
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        if (b < n) {
            const int r = R.rank[(int64_t)b * nc + idc];
            if constexpr (MODE == 1) wf_at(wv, r, id)[x] = z[b];
            else evec[((int64_t)r * nk + id) * n + x] = z[b];
        }
    }
}

template <int MODE, bool LIST = false>
__global__ __launch_bounds__(256) void k_ql16_replay(const int n, const int64_t nk, const int64_t id0, const int64_t nc,
                                                      const Ql16Rec R, cd* __restrict__ evec, const WfsView wv,
                                                      const int* __restrict__ list = nullptr, const int* __restrict__ count = nullptr,
                                                      unsigned long long* __restrict__ listed_total = nullptr) {
    if constexpr (!LIST) {
        ql16_replay_body<MODE>(n, nk, id0, nc, R, evec, wv, nullptr, nc, ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4);
        return;
    }
    const int64_t nhave = *count;
    // tbk_ctx_solver_stats: matrices the direct kernels handed to this fallback since the last reset
    if (listed_total && nhave > 0 && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(listed_total, (unsigned long long)nhave);
    for (int64_t base = (int64_t)blockIdx.x * 16; base < nhave; base += (int64_t)gridDim.x * 16)
        ql16_replay_body<MODE>(n, nk, id0, nc, R, evec, wv, list, nhave, base + (threadIdx.x >> 4));
}

// MODE 0: k list, 1: regular mesh into a wf_array (+ min gaps), 2: supplied matrices
// STAGE 0: the whole solve.  STAGE 1 (eigenvalues only): stop after the tridiagonalisation, de[j * nc + (id - id0)] = (d_j, e_j).
// STAGE 2 (with eigenvectors, three-kernel form): the same, and Z = H_0 ... H_13 D goes to the output array in column order
// (band slot b = column b) for k_ql16_replay.  The launch covers the matrices [id0, id0 + nc) of the batch of nk.
template <int MODE, bool VEC, int STAGE>
__device__ __forceinline__ void ql16_solve_body(const ModelView& mv, const int64_t nk, const ListArgs& Lst, const GridArgs& G,
                                                int* noconv_flag, double2* __restrict__ de, const int64_t id0, const int64_t nc,
                                                const int64_t wslot, const bool live, const int64_t id) {
    static_assert(STAGE != 1 || !VEC, "k_solve_ql16: stage 1 is the eigenvalue-only form");
    static_assert(STAGE != 2 || VEC, "k_solve_ql16: stage 2 is the eigenvector form");
    const int lane = threadIdx.x & 63;
    const int x = lane & 15;
    const int rowbase4 = (lane & 48) * 4;
    const int n = mv.nsta;
    const bool real_row = x < n;
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    cd a[16], z[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        a[c] = cd{0.0, 0.0};
        z[c] = cd{c == x ? 1.0 : 0.0, 0.0};
    }
    if constexpr (MODE == 2) {
        const cd* h = Lst.ham + id * (int64_t)n * n;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (real_row && c < n) {   // upper triangle, mirrored (the reference's eigh reads one triangle)
                cd t = c >= x ? h[x * n + c] : cconj(h[c * n + x]);
                if (c == x) t.y = 0.0;
                a[c] = t;
            }
        }
    } else {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < mv.dim_k) kk[d] = Lst.k[id * mv.dim_k + d];
        } else {
            grid_point(G, id, kk, wrap);
        }
        cd zk[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) zk[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        // S[x][c] = sum_R U_R[slot(min,max)] e^{2 pi i k.R}  (conjugated below the diagonal)
        int sidx[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int lo = x < c ? x : c, hi = x < c ? c : x;
            sidx[c] = real_row && c < n ? lo * n - lo * (lo - 1) / 2 + (hi - lo) : -1;
        }
        for (int base = 0; base < mv.nR; base += 16) {
            const int mine = base + x;
            const cd ph = mine < mv.nR ? phase_of_R(zk, mv.rvec[mine]) : cd{0.0, 0.0};
            cd phs[16];
            row16_bcast_phase<0>(ph, phs);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (base + j < mv.nR) {
                    const cd* u = mv.rblock + (size_t)(base + j) * mv.nslot;
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (sidx[c] >= 0) cfma(a[c], u[sidx[c]], phs[j]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < x) a[c].y = -a[c].y;
            if (c == x) a[c].y = 0.0;
        }
    }

    // ---- 1. tridiagonalisation; lane j ends up with d_j, e_j = |T[j+1][j]| and Z = H_0 ... H_13 D
    double ee = 0.0;
    {
        cd delta{1.0, 0.0};   // D_{K+1} = D_K t_K / |t_K|
        auto step_phase = [&](const cd t, const int col) {
            const double t2 = cabs2(t);
            double mag = 0.0;
            if (t2 > 0.0) {
                const double inv = rsqrt_full(t2);
                mag = t2 * inv;
                delta = cmul(delta, cd{t.x * inv, t.y * inv});
            }
            if (x == col - 1) ee = mag;
            return mag;
        };
#define TBK_QL_HOUSE(KK)                                              \
    {                                                                 \
        const cd t = ql16_house<KK, VEC>(a, z, x);                    \
        step_phase(t, KK + 1);                                        \
        if (VEC) z[KK + 1] = cmul(z[KK + 1], delta);                  \
    }
        TBK_QL_HOUSE(0) TBK_QL_HOUSE(1) TBK_QL_HOUSE(2) TBK_QL_HOUSE(3) TBK_QL_HOUSE(4) TBK_QL_HOUSE(5) TBK_QL_HOUSE(6)
        TBK_QL_HOUSE(7) TBK_QL_HOUSE(8) TBK_QL_HOUSE(9) TBK_QL_HOUSE(10) TBK_QL_HOUSE(11) TBK_QL_HOUSE(12) TBK_QL_HOUSE(13)
#undef TBK_QL_HOUSE
        const cd t14 = rowbcast_c<15>(a[14]);            // T[15][14]: never reflected
        step_phase(t14, 15);
        if (VEC) z[15] = cmul(z[15], delta);
    }
    double dd = sel16<0>(a, x, cd{0.0, 0.0}).x;          // d_x = A[x][x]

    if constexpr (STAGE != 0) {
        if (live) de[(int64_t)x * nc + wslot] = double2{dd, x < 15 ? ee : 0.0};
        if constexpr (STAGE == 2) {
            if (live && real_row) {
                cd f{1.0, 0.0};
                if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[x])));
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (wrap[d]) f = cmul(f, G.pbc[d * n + x]);
                }
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    if (b < n) {
                        const cd val = cmul(z[b], f);
                        if constexpr (MODE == 1) wf_at(G.wv, b, id)[x] = val;
                        else Lst.evec[((int64_t)b * nk + id) * n + x] = val;
                    }
                }
            }
        }
        return;
    }

    // ---- 2. implicit QL; l = first row of the block being worked on, m = its last row
    int l = 0;
    bool done = false;
    for (int iter = 0; iter <= TBK_QL_MAX_ITER; ++iter) {
        // negligible couplings: |e_j| <= eps (|d_j| + |d_j+1|)   (e_15 = 0: always)
        const I2 di = __builtin_bit_cast(I2, dd);
        const I2 dn{__builtin_amdgcn_update_dpp(0, di.lo, 0x101, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, di.hi, 0x101, 0xf, 0xf, true)};
        const double dnext = __builtin_bit_cast(double, dn);                       // row_shl:1 = d_{x+1} (0 for x = 15)
        const bool negl = fabs(ee) <= 2.220446049250313e-16 * (fabs(dd) + fabs(dnext));
        const unsigned long long bal = __ballot(negl);
        const unsigned mask16 = (unsigned)(bal >> (lane & 48)) & 0xffffu;
        int m = 15;
        if (!done) {
            const unsigned open = ~mask16 & (0xffffu << l) & 0xffffu;
            if (open == 0) {
                done = true;
            } else {
                l = __builtin_ctz(open);
                m = __builtin_ctz(mask16 & (0xffffu << l));
            }
        }
        if (__all(done)) break;
        if (iter == TBK_QL_MAX_ITER) {
            if (!done && x == 0) atomicExch(noconv_flag, 1);
            break;
        }
        Ql16State S{1.0, 1.0, 0.0, 0.0, !done};
        if (!done) {
            // Wilkinson-type shift from the leading 2 x 2 of the block (its accuracy only affects the
            // speed of convergence, so hardware reciprocal / square root estimates are good enough)
            const double dl = bperm_d(rowbase4 + 4 * l, dd), dl1 = bperm_d(rowbase4 + 4 * l + 4, dd);
            const double el = bperm_d(rowbase4 + 4 * l, ee), dm = bperm_d(rowbase4 + 4 * m, dd);
            double g = (dl1 - dl) * (0.5 * __builtin_amdgcn_rcp(el));
            const double r = __builtin_amdgcn_sqrt(fma(g, g, 1.0));
            S.g = dm - dl + el * __builtin_amdgcn_rcp(g + copysign(r, g));
        }
        ql16_pos<14, VEC>(S, z, dd, ee, x, l, m, rowbcast_d<15>(dd));
        if (!done) {
            if (S.alive && x == l) {
                dd -= S.p;
                ee = S.g;
            }
            if (x == m) ee = 0.0;
        }
    }

    // ---- eigenvalues in stable ascending order (padding rows rank last)
    double ev[16];
    ql16_bcast_ev<0>(dd, ev);
    const double mine = dd;
    int rk = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool jr = j < n;
        const bool before = real_row ? (jr && (ev[j] < mine || (ev[j] == mine && j < x))) : (jr || j < x);
        rk += before ? 1 : 0;
    }
    const double sorted_here = perm_push_d(rowbase4 + 4 * rk, mine);   // lane r now holds the r-th eigenvalue
    if constexpr (MODE == 1) {
        const I2 sh = __builtin_bit_cast(I2, sorted_here);
        const I2 nx{__builtin_amdgcn_update_dpp(0, sh.lo, 0x101, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0, sh.hi, 0x101, 0xf, 0xf, false)};
        double gap = live && x + 1 < n ? __builtin_bit_cast(double, nx) - sorted_here : INFINITY;   // row_shl:1 = next lane's value
        gap = fmin(gap, __shfl_xor(gap, 16));
        gap = fmin(gap, __shfl_xor(gap, 32));
        if (lane < 16 && x + 1 < n) {
            unsigned long long* slot = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + x;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
            if (bits < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, bits);
        }
    } else {
        if (live && real_row) Lst.eval[(int64_t)x * nk + id] = sorted_here;
    }
    if (VEC) {
        int rks[16];
        row16_bcast_rank<0>(rk, rks);
        if (live && real_row) {
            cd f{1.0, 0.0};
            if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[x])));
            if constexpr (MODE == 1) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (wrap[d]) f = cmul(f, G.pbc[d * n + x]);
            }
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                if (b < n) {
                    const cd val = cmul(z[b], f);
                    if constexpr (MODE == 1) wf_at(G.wv, rks[b], id)[x] = val;
                    else Lst.evec[((int64_t)rks[b] * nk + id) * n + x] = val;
                }
            }
        }
    }
}

template <int MODE, bool VEC, int STAGE = 0, bool LIST = false>
__global__ __launch_bounds__(256) void k_solve_ql16(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G,
                                                     int* noconv_flag, double2* __restrict__ de = nullptr, const int64_t id0 = 0,
                                                     const int64_t nc = 0, const int* __restrict__ list = nullptr,
                                                     const int* __restrict__ count = nullptr) {
    if constexpr (!LIST) {
        const int64_t mat = id0 + (((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4);
        const int64_t nend = STAGE == 0 ? nk : id0 + nc;
        const bool live = mat < nend;
        const int64_t id = live ? mat : nend - 1;   // idle tail rows shadow the last point
        ql16_solve_body<MODE, VEC, STAGE>(mv, nk, Lst, G, noconv_flag, de, id0, nc, id - id0, live, id);
        return;
    }
    // list mode (see k_ql16_lanes): the matrices id0 + list[0 .. *count), workspace entries by position in the list
    const int64_t cnt = *count;
    for (int64_t base = (int64_t)blockIdx.x * 16; base < cnt; base += (int64_t)gridDim.x * 16) {
        const int64_t w0 = base + (threadIdx.x >> 4);
        const bool live = w0 < cnt;
        const int64_t w = live ? w0 : cnt - 1;
        ql16_solve_body<MODE, VEC, STAGE>(mv, nk, Lst, G, noconv_flag, de, id0, nc, w, live, id0 + list[w]);
    }
}

// tbk_solve_tw16.inl: tridiagonalise | eigenvalues | twisted-factorisation eigenvectors + back-transformation
template <int MODE>
static int launch_tw16(tbk_ctx* ctx, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G);

template <int MODE, bool VEC>
static int launch_ql16(tbk_ctx* ctx, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G, int64_t nk_eff) {
    TBK_REQUIRE(nk * 16 < (int64_t)0x7fffffff * 256, TBK_EUNSUPPORTED, "too many k-points for one launch");
    const unsigned blocks = (unsigned)((nk * 16 + 255) / 256);
    if constexpr (!VEC && MODE != 1) {
        // round 5: eigenvalues only through the fused kernel (k_e16<MODE, false>: tridiagonalisation on DPP, every lane its own
        // eigenvalue, nothing listed) at every count -- the pair of kernels below was slower than the fused kernel WITH its
        // eigenvectors (cubic16: 5.1 against 3.7 ns per point; TBK_E16_EVALS=0 keeps the pair)
        if (tbk_knobs().e16 != 0 && tbk_knobs().e16_evals != 0 && !ctx->qlw_off && (MODE == 2 || mv.nR > 0)) {
            const int64_t cmax = ((int64_t)0x7fffffff / 16 - 4) & ~(int64_t)3;
            for (int64_t id0 = 0; id0 < nk; id0 += cmax) {
                const int rc = tbk_e16_launch_evals(MODE, ctx->stream, mv, nk, L, id0, std::min<int64_t>(cmax, nk - id0));
                if (rc) return rc;
            }
            return TBK_OK;
        }
        // (from ~8 k matrices on: below that the second kernel's one-matrix-per-lane QL is a single under-filled wavefront
        // and its latency exceeds what the replication costs; TBK_QL16_EVONLY=0: always the single replicated kernel)
        if (tbk_knobs().ql16_evonly != 0 && nk >= 8192) {
            const size_t wbytes = (size_t)nk * 16 * sizeof(double2);
            if (wbytes > ctx->work_bytes) {
                TBK_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->work) TBK_HIP(hipFree(ctx->work));
                ctx->work = nullptr;
                ctx->work_bytes = 0;
                hipError_t e = hipMalloc(&ctx->work, wbytes);
                TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tridiagonal workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
                ctx->work_bytes = wbytes;
            }
            double2* de = (double2*)ctx->work;
            hipLaunchKernelGGL((k_solve_ql16<MODE, false, 1>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G, ctx->flags_dev, de,
                               (int64_t)0, nk);
            hipLaunchKernelGGL(k_tridiag_eigvals, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, mv.nsta, nk,
                               (const double2*)de, L.eval, ctx->flags_dev);
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
    }
    if constexpr (VEC) {
        // three-kernel form (TBK_QL16_SPLIT): tridiagonalise + Z | one lane per matrix: QL, recording | replay on the rows of Z
        const TbkKnobs& K = tbk_knobs();
        // (decided on the size of the GLOBAL mesh: every window of an array takes the same route)
        if (K.ql16_split != 0 && !ctx->qlw_off && nk_eff >= (K.ql16_split_min >= 0 ? K.ql16_split_min : 8192)) {
            if (K.tw16 != 0) return launch_tw16<MODE>(ctx, mv, nk, L, G);
            // sweeps recorded per matrix (~35 are typical at n = 16; LAPACK gives up at 480).  A matrix that needs more makes the
            // caller repeat the batch on the single kernel (TBK_QLW_CAP: tests provoke that)
            const int scap = K.qlw_cap > 0 ? (int)std::max<long long>(1, std::min<long long>(64, K.qlw_cap / 16)) : 64;
            auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
            const size_t per = 16 * sizeof(double2) + (size_t)scap * 16 * sizeof(double2) + (size_t)scap * sizeof(unsigned) + sizeof(int) + 16;
            const size_t budget = (size_t)(K.qlw_ws_mb > 0 ? K.qlw_ws_mb : 4096) << 20;
            int64_t chunk = std::max<int64_t>(4096, (int64_t)(budget / per));
            chunk = std::min<int64_t>(chunk, nk);
            chunk = (nk + (nk + chunk - 1) / chunk - 1) / ((nk + chunk - 1) / chunk);
            const size_t wbytes = al((size_t)chunk * 16 * sizeof(double2)) + al((size_t)chunk * scap * 16 * sizeof(double2)) +
                                  al((size_t)chunk * scap * sizeof(unsigned)) + al((size_t)chunk * sizeof(int)) + al((size_t)chunk * 16) + 1024;
            if (wbytes > ctx->work_bytes) {
                TBK_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->work) TBK_HIP(hipFree(ctx->work));
                ctx->work = nullptr;
                ctx->work_bytes = 0;
                hipError_t e = hipMalloc(&ctx->work, wbytes);
                TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "QL record workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
                ctx->work_bytes = wbytes;
            }
            unsigned char* p = (unsigned char*)ctx->work;
            double2* de = (double2*)p;
            p += al((size_t)chunk * 16 * sizeof(double2));
            Ql16Rec R{};
            R.rot = (double2*)p;
            p += al((size_t)chunk * scap * 16 * sizeof(double2));
            R.swp = (unsigned*)p;
            p += al((size_t)chunk * scap * sizeof(unsigned));
            R.nit = (int*)p;
            p += al((size_t)chunk * sizeof(int));
            R.rank = (signed char*)p;
            R.scap = scap;
            cd* evec = MODE == 1 ? nullptr : L.evec;
            for (int64_t id0 = 0; id0 < nk; id0 += chunk) {
                const int64_t nc = std::min<int64_t>(chunk, nk - id0);
                const unsigned b16 = (unsigned)((nc * 16 + 255) / 256);
                hipLaunchKernelGGL((k_solve_ql16<MODE, true, 2>), dim3(b16), dim3(256), 0, ctx->stream, mv, nk, L, G, ctx->flags_dev, de, id0, nc);
                hipLaunchKernelGGL((k_ql16_lanes<MODE>), dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream, mv.nsta, nk, id0, nc,
                                   (const double2*)de, L.eval, G, R, ctx->flags_dev);
                hipLaunchKernelGGL((k_ql16_replay<MODE>), dim3(b16), dim3(256), 0, ctx->stream, mv.nsta, nk, id0, nc, R, evec, G.wv);
            }
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
    }
    hipLaunchKernelGGL((k_solve_ql16<MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G, ctx->flags_dev);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
