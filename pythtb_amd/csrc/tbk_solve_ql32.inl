// tbk_solve_ql32.inl -- included by tbk_solve_qlw.inl (after QlwWork).  Stage 2 of the direct solver for n = 17..32 states (round 6):
// the implicit-shift QL iteration on (d, e), ONE lane per matrix like k_tridiag_ql_lanes, but with d and e in REGISTERS and every
// position of a sweep unrolled under a per-lane predicate -- the form k_tw16_eigvals has for 9..16 states (qle_pos, tbk_solve_dev.h).
// k_tridiag_ql_lanes keeps d, e in LDS and walks the union of the 64 lanes' ranges with dynamic positions: ~450 cycles per position
// (36 k matrices of 32 states: 0.39 ms on half a wavefront per SIMD, a pure latency chain); here a position is its ~12 dependent
// double-precision operations.  No rotation record: eigenvalues only, or also what k_tw32_vectors needs (W.lam set: ranks, the
// eigenvalue of every position, the splitting of T, close pairs listed) -- the matrices k_tw32_vectors leaves to the replay get
// their record from k_tridiag_ql_lanes<.., 1, LIST>.  Restates the reference's numpy.linalg.eigh (pythtb.py:939-947); same
// recurrences as EISPACK tql2, LAPACK's limit of 30 shifts per eigenvalue.  TBK_QL32=0: k_tridiag_ql_lanes for every matrix.

template <int I, int NM>
__device__ __forceinline__ void ql32_pos(double (&d)[NM], double (&e)[NM], double& sn, double& cs, double& pp, double& g, bool& alive,
                                         const bool live, const int l, const int m) {
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
        } else {                                 // r == 0 (underflow): tql2's recovery
            d[I + 1] -= pp;
            e[I + 1] = 0.0;
            alive = false;
        }
    }
    if constexpr (I > 0) ql32_pos<I - 1, NM>(d, e, sn, cs, pp, g, alive, live, l, m);
}
template <int J, int NM>
__device__ __forceinline__ double ql32_pick(const double (&a)[NM], const int idx, const double acc) {
    const double r = idx == J ? a[J] : acc;
    if constexpr (J + 1 < NM) return ql32_pick<J + 1, NM>(a, idx, r);
    else return r;
}

// ONE instantiation serves the eigenvalue-only calls and the calls with eigenvectors (W.lam != nullptr: the extra outputs), so that the
// eigenvalues of the two forms of a call are the same bits (two instantiations differed in the last place on an already tridiagonal
// chain: the compiler contracts multiply-adds per instantiation).
template <int MODE, int NM>
__global__ __launch_bounds__(64) void k_ql32_lanes(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk, const QlwWork W,
                                                   double* __restrict__ eval, const GridArgs G, int* flags) {
    const bool tw = W.lam != nullptr;                    // (uniform)
    __shared__ double S[NM][64], T[NM][64];              // the eigenvalues by position / by rank, on their way out
    const int lane = threadIdx.x;
    const int64_t idc = (int64_t)blockIdx.x * 64 + lane;
    const bool has = idc < nchunk;
    const int64_t ic = has ? idc : nchunk - 1;
    double d[NM], e[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) {
        double2 v{0.0, 0.0};
        if (j < n) v = W.de[(int64_t)j * nchunk + ic];
        d[j] = v.x;
        e[j] = j + 1 < n ? v.y : 0.0;
    }
    // T splits where a coupling is negligible; those are zeroed for good, so that no rotation ever mixes two blocks and the
    // eigenvalue left at position j belongs to the block of T that contains j
    const double eps = 2.220446049250313e-16;
    constexpr unsigned topbit = 1u << (NM - 1), all = NM == 32 ? ~0u : (1u << NM) - 1u;
    unsigned split = topbit;
#pragma unroll
    for (int j = 0; j < NM - 1; ++j) {
        const bool ng = fabs(e[j]) <= eps * (fabs(d[j]) + fabs(d[j + 1]));
        split |= ng ? (1u << j) : 0u;
        e[j] = ng ? 0.0 : e[j];
    }
    int l = 0;
    bool done = !has;
    const int max_iter = 30 * n;
    for (int iter = 0;; ++iter) {
        int m = NM - 1;
        if (!done) {
            unsigned negl = topbit;
#pragma unroll
            for (int j = 0; j < NM - 1; ++j) negl |= fabs(e[j]) <= eps * (fabs(d[j]) + fabs(d[j + 1])) ? (1u << j) : 0u;
            const unsigned open = ~negl & (all << l) & all;
            if (open == 0) {
                done = true;
            } else {
                l = __builtin_ctz(open);
                m = __builtin_ctz(negl & (all << l));
            }
        }
        if (__all(done)) break;
        if (iter >= max_iter) {
            if (!done) atomicExch(flags, 1);
            break;
        }
        double sn = 1.0, cs = 1.0, pp = 0.0, g = 0.0;
        bool alive = true;
        if (!done) {
            // Wilkinson-type shift from the leading 2 x 2 of the block (only the speed of convergence depends on its accuracy)
            const double dl = ql32_pick<0, NM>(d, l, 0.0), dl1 = ql32_pick<0, NM>(d, l + 1, 0.0);
            const double el = ql32_pick<0, NM>(e, l, 1.0), dmm = ql32_pick<0, NM>(d, m, 0.0);
            const double gs = (dl1 - dl) * (0.5 * __builtin_amdgcn_rcp(el));
            const double r = __builtin_amdgcn_sqrt(fma(gs, gs, 1.0));
            g = dmm - dl + el * __builtin_amdgcn_rcp(gs + copysign(r, gs));
        }
        ql32_pos<NM - 2, NM>(d, e, sn, cs, pp, g, alive, !done, l, m);
    }
    // stable ascending ranks among the n real entries; T[r] <- the eigenvalue of rank r.  (Through LDS with rolled loops: the
    // unrolled n^2 comparisons on registers were 3 k instructions, and the compiler put d in scratch memory for them once the loop
    // could not be unrolled fully.)
#pragma unroll
    for (int a = 0; a < NM; ++a) S[a][lane] = d[a];
    asm volatile("" ::: "memory");                      // (a lane reads back its own column: no barrier)
    double tnorm = 0.0;
    for (int a = 0; a < n; ++a) tnorm = fmax(tnorm, fabs(S[a][lane]));
    const double thr = tw ? W.gaptol * tnorm : 0.0;
    bool flagged = false;
    for (int a = 0; a < n; ++a) {
        const double da = S[a][lane];
        int r = 0;
        for (int b = 0; b < n; ++b) {
            const double db = S[b][lane];
            r += (db < da || (db == da && b < a)) ? 1 : 0;
            // two eigenvalues of one unreduced block (no split between positions a and b) closer than gaptol |T|: their
            // twisted-factorisation vectors would be nearly parallel
            if (tw && b > a) flagged = flagged || (((split >> a) & ((1u << (b - a)) - 1u)) == 0 && !(fabs(da - db) >= thr));
        }
        T[r][lane] = da;
        if (tw && has) {
            W.rank[(int64_t)a * nchunk + idc] = r;
            W.lam[(int64_t)a * nchunk + idc] = da;
        }
    }
    if (tw && has) {
        W.meta[idc] = uint2{split, flagged ? 1u : 0u};
        if (flagged) W.list[atomicAdd(W.count, 1)] = (int)idc;
    }
    asm volatile("" ::: "memory");                      // (a lane reads back its own column of S: no barrier)
    double prev = 0.0;
    for (int r = 0; r < n; ++r) {
        const double v = T[r][lane];
        if constexpr (MODE == 1) {
            if (r > 0) {
                double gap = has ? v - prev : INFINITY;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gap = fmin(gap, __shfl_xor(gap, o));
                if (lane == 0) {
                    unsigned long long* slot = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + (r - 1);
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
                    if (bits < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, bits);
                }
            }
            prev = v;
        } else {
            if (has) eval[(int64_t)r * nk + id0 + idc] = v;
        }
    }
}
