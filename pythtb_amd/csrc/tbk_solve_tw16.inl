// tbk_solve_tw16.inl -- included by tbk_solve.hip after tbk_solve_ql16.inl (whose helpers and fallback kernels it uses).
//
// n = 9..16 states per k WITH eigenvectors, chip-filling batches (config E: cubic16 on 257^3 points): the direct solver
// numpy.linalg.eigh runs for the reference (pythtb.py:939-947) -- Householder tridiagonalisation, eigenvalues of the real
// tridiagonal T, eigenvectors of T, back-transformation -- cut where its parallelism changes, with every stage doing
// only the work its stage needs:
//
//  1. k_tw16_tridiag    16 lanes per matrix (lane x = row x of A).  H(k) assembly and the 14 Householder reflections.
//                       Nothing is accumulated: the reflectors themselves go out (sqrt(beta) u_K packed by column, the
//                       diagonal unitary D that makes the subdiagonal real) -- 2.2 KB per matrix instead of the 4 KB of
//                       Z = H_0 .. H_13 D, and neither Z's registers nor its ~40 % of the instructions.
//  2. k_tw16_eigvals    ONE lane per matrix: implicit QL on (d, e), eigenvalues only (the rotation recurrence once per
//                       matrix, nothing recorded).  Leaves the eigenvalue of every position, the ascending ranks and the
//                       splitting of T into unreduced blocks; flags matrices with two eigenvalues of one block closer
//                       than gaptol |T|.
//  3. k_tw16_vectors    16 lanes per matrix (lane j = eigenvector j).  (a) The eigenvector of T for lambda_j from the
//                       TWISTED FACTORISATION (Fernando 1997; Parlett & Dhillon; LAPACK dlar1v): top-down and bottom-up
//                       pivots of T - lambda, glued at the position r of the smallest |gamma_r| -- O(n) per vector, no
//                       iteration, residual |gamma_r| / |z| ~ eps |T|.  Vectors computed independently are orthogonal
//                       to ~20 eps |T| / gap only, so (b) ONE Newton-Schulz step V <- V (3 I - V^T V) / 2 on the real
//                       16 x 16 matrix V makes them orthonormal to rounding (error E -> E^2) -- two dense 16^3 products,
//                       the one contraction on this path, on the matrix cores (v_mfma_f64_16x16x4_f64).  (c) z_j =
//                       H_0 .. H_13 D v_j by applying the stored reflectors (broadcast reads from LDS), (d) transposed
//                       through LDS so that stores are contiguous along the orbital index like every other solver's.
//
// The QL rotations are never applied to anything: the ~35 sweeps x 15 positions of 2 x 2 rotations on 16 complex rows of Z
// per matrix (k_ql16_replay, 1.9 k wave-instructions per matrix) and their [sweep][position][matrix] record (743 MB per
// 137 k matrices) are gone from the main path.
//
// Matrices the twisted factorisation cannot serve -- two eigenvalues of one unreduced block closer than gaptol |T|
// (default 1e-5: the Newton-Schulz step then no longer reaches rounding level), or a residual check that fails -- are
// put on a list and solved again by the three kernels of tbk_solve_ql16.inl (k_solve_ql16<.., 2>, k_ql16_lanes,
// k_ql16_replay), which take the list instead of a range.  The decision depends on the matrix alone, so periodic images,
// halo rows and shard windows stay bit-identical.  TBK_TW16=0 restores the old three-kernel form for every matrix;
// TBK_TW16_GAPTOL sets the threshold (tests force every matrix onto the list with a huge one).

#define TW16_REC 136   // 16-byte entries of a matrix's reflector record: 119 reflector elements, 16 phases, 1 spare
__host__ __device__ constexpr int tw16_off(const int K) { return 15 * K - K * (K - 1) / 2; }   // first element of u_K

typedef double tw_d4 __attribute__((ext_vector_type(4)));
// the LDS regions of these kernels are private to a wavefront, whose LDS operations execute in order: no barriers, no waits;
// only the compiler has to keep the order (the views of a region differ in type)
#define TW_LDS_ORDER() asm volatile("" ::: "memory")

// ---- Householder step K on the rows of A alone; us = sqrt(beta) u_x of this lane (0 for x <= K or when nothing is reflected).
// u and q reach the other rows through LDS: every lane stores its element once (16 bytes) and reads the elements of the
// columns c > K as 16-lane broadcasts -- 1 LDS read where a DPP row broadcast of a complex number takes 4 VALU moves
// (960 of the kernel's 4.9 k instructions per wavefront were those moves).  lu / lq: this matrix's 16 slots; the region is
// private to the wavefront and a wavefront's LDS operations execute in order, so there is no barrier anywhere.
template <int K, int C>
__device__ __forceinline__ void tw16_pass1(const cd (&a)[16], const cd* __restrict__ lu, cd& p) {
    const cd uc = lu[C];
    p.x = fma(a[C].x, uc.x, p.x);
    p.x = fma(-a[C].y, uc.y, p.x);
    p.y = fma(a[C].x, uc.y, p.y);
    p.y = fma(a[C].y, uc.x, p.y);
    if constexpr (C + 1 < 16) tw16_pass1<K, C + 1>(a, lu, p);
}
template <int K, int C>
__device__ __forceinline__ void tw16_pass2(cd (&a)[16], const cd* __restrict__ lu, const cd* __restrict__ lq, const cd u, const cd q) {
    const cd uc = lu[C], qc = lq[C];
    // A[x][c] -= u_x conj(q_c) + q_x conj(u_c)
    a[C].x = fma(-u.x, qc.x, fma(-u.y, qc.y, fma(-q.x, uc.x, fma(-q.y, uc.y, a[C].x))));
    a[C].y = fma(-u.y, qc.x, fma(u.x, qc.y, fma(-q.y, uc.x, fma(q.x, uc.y, a[C].y))));
    if constexpr (C + 1 < 16) tw16_pass2<K, C + 1>(a, lu, lq, u, q);
}
template <int K>
__device__ __forceinline__ cd tw16_house(cd (&a)[16], const int x, cd& us, cd* __restrict__ lu, cd* __restrict__ lq) {
    const bool below = x > K;
    const cd xk = below ? a[K] : cd{0.0, 0.0};
    // (decided on the entries below the subdiagonal alone, like LAPACK's zlarfg: see ql16_house)
    const double rest = row_allsum(x > K + 1 ? cabs2(xk) : 0.0);
    const cd alpha = rowbcast_c<K + 1>(a[K]);            // A[K+1][K]
    const double absa2 = cabs2(alpha);
    const double sigma = rest + absa2;
    cd tK = alpha;
    us = cd{0.0, 0.0};
    if (rest > 0.0) {                                    // row-uniform
        const double inv_n = rsqrt_full(sigma), nrm = sigma * inv_n;
        double absa = 0.0;
        cd ph{1.0, 0.0};
        if (absa2 > 0.0) {
            const double inv_a = rsqrt_full(absa2);
            absa = absa2 * inv_a;
            ph = cd{alpha.x * inv_a, alpha.y * inv_a};
        }
        const cd u = x == K + 1 ? cd{ph.x * (absa + nrm), ph.y * (absa + nrm)} : xk;
        const double sb = rsqrt_full(nrm * (nrm + absa)), beta = sb * sb;   // beta = 2 / (u^+ u)
        tK = cd{-ph.x * nrm, -ph.y * nrm};
        lu[x] = u;
        TW_LDS_ORDER();
        cd p{0.0, 0.0};
        tw16_pass1<K, K + 1>(a, lu, p);
        p = cd{p.x * beta, p.y * beta};
        const double kappa = 0.5 * beta * row_allsum(u.x * p.x + u.y * p.y);
        const cd q = below ? cd{fma(-kappa, u.x, p.x), fma(-kappa, u.y, p.y)} : cd{0.0, 0.0};
        lq[x] = q;
        TW_LDS_ORDER();
        tw16_pass2<K, K + 1>(a, lu, lq, u, q);
        TW_LDS_ORDER();
        us = cd{u.x * sb, u.y * sb};
    }
    return tK;
}

// MODE 0: k list, 1: regular mesh into a wf_array, 2: supplied matrices.  The launch covers the matrices [id0, id0 + nc).
template <int MODE>
__global__ __launch_bounds__(256) void k_tw16_tridiag(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G,
                                                      double2* __restrict__ de, cd* __restrict__ refl, const int64_t id0,
                                                      const int64_t nc) {
    // per wavefront: u and q of its four matrices, 17 slots of 16 bytes per matrix (the four broadcasts of a read then fall
    // on different banks)
    __shared__ __attribute__((aligned(16))) cd lds_uq[4 * 2 * 4 * 17];
    const int lane = threadIdx.x & 63;
    const int x = lane & 15;
    cd* const lu = lds_uq + ((threadIdx.x >> 6) * 2 * 4 + (lane >> 4)) * 17;
    cd* const lq = lu + 4 * 17;
    const int64_t slot0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = slot0 < nc;
    const int64_t slot = live ? slot0 : nc - 1;          // idle tail rows shadow the last matrix
    const int64_t id = id0 + slot;
    const int n = mv.nsta;
    const bool real_row = x < n;
    cd a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = cd{0.0, 0.0};
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
        cd zk[4] = {cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < mv.dim_k) zk[d] = expi2pi(Lst.k[id * mv.dim_k + d]);
        } else {
            // exp(2 pi i k_d) of a mesh point: the per-axis tables (k_grid_tables: the same expression as grid_point + expi2pi,
            // so the same bits whichever window the point is solved in)
            int64_t rem = id;
#pragma unroll
            for (int d = 3; d >= 0; --d) {
                if (d < G.wv.dim_arr) {
                    const int64_t md = G.wv.mesh[d];
                    const int64_t qd = d > 0 ? rem / md : 0;
                    zk[d] = G.tz[d][d > 0 ? rem - qd * md : rem];
                    rem = qd;
                }
            }
        }
        // S[x][c] = sum_R U_R[slot(min,max)] e^{2 pi i k.R}  (conjugated below the diagonal)
        int sidx[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int lo = x < c ? x : c, hi = x < c ? c : x;
            sidx[c] = real_row && c < n ? lo * n - lo * (lo - 1) / 2 + (hi - lo) : -1;
        }
        // (every lane forms the phase of every lattice vector itself -- a few complex products with wave-uniform loop counts --
        // instead of sixteen lanes forming one each and broadcasting them: the broadcast array cost 64 registers, the
        // difference between two and three wavefronts per SIMD for this kernel)
        for (int r = 0; r < mv.nR; ++r) {
            const cd ph = phase_of_R(zk, mv.rvec[r]);
            const cd* u = mv.rblock + (size_t)r * mv.nslot;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (sidx[c] >= 0) cfma(a[c], u[sidx[c]], ph);
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < x) a[c].y = -a[c].y;
            if (c == x) a[c].y = 0.0;
        }
    }

    // ---- tridiagonalisation; lane j ends up with d_j and e_j = |T[j+1][j]|, lane c with the phase of column c of D
    cd* __restrict__ rec = refl + slot * TW16_REC;
    double ee = 0.0;
    cd delta{1.0, 0.0}, dx{1.0, 0.0};
    auto step_phase = [&](const cd t, const int col) {
        const double t2 = cabs2(t);
        double mag = 0.0;
        if (t2 > 0.0) {
            const double inv = rsqrt_full(t2);
            mag = t2 * inv;
            delta = cmul(delta, cd{t.x * inv, t.y * inv});
        }
        if (x == col - 1) ee = mag;
        if (x == col) dx = delta;
    };
#define TBK_TW_HOUSE(KK)                                                          \
    {                                                                             \
        cd us;                                                                    \
        const cd t = tw16_house<KK>(a, x, us, lu, lq);                            \
        step_phase(t, KK + 1);                                                    \
        if (live && x > KK) rec[tw16_off(KK) + x - KK - 1] = us;                  \
    }
    TBK_TW_HOUSE(0) TBK_TW_HOUSE(1) TBK_TW_HOUSE(2) TBK_TW_HOUSE(3) TBK_TW_HOUSE(4) TBK_TW_HOUSE(5) TBK_TW_HOUSE(6)
    TBK_TW_HOUSE(7) TBK_TW_HOUSE(8) TBK_TW_HOUSE(9) TBK_TW_HOUSE(10) TBK_TW_HOUSE(11) TBK_TW_HOUSE(12) TBK_TW_HOUSE(13)
#undef TBK_TW_HOUSE
    const cd t14 = rowbcast_c<15>(a[14]);                // T[15][14]: never reflected
    step_phase(t14, 15);
    const double dd = sel16<0>(a, x, cd{0.0, 0.0}).x;    // d_x = A[x][x]
    if (live) {
        de[(int64_t)x * nc + slot] = double2{dd, x < 15 ? ee : 0.0};
        rec[119 + x] = dx;
    }
}

// ---- 2. eigenvalues of T, one lane per matrix.  lam[j * nc + slot] = eigenvalue left at position j;
// meta[slot] = {ranks of positions 0..7 (4 bits each), of positions 8..15, split mask (bit i: e_i negligible in T), flagged}
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_tw16_eigvals(const int n, const int64_t nk, const int64_t id0, const int64_t nc,
                                                      const double2* __restrict__ de, double* __restrict__ eval, const GridArgs G,
                                                      double* __restrict__ lam, uint4* __restrict__ meta, int* __restrict__ list,
                                                      int* __restrict__ count, int* flags, const double gaptol) {
    const int64_t idc = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool has = idc < nc;
    const int64_t ic = has ? idc : nc - 1;
    double d[16], e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double2 v = de[(int64_t)j * nc + ic];
        d[j] = v.x;
        e[j] = v.y;
    }
    e[15] = 0.0;
    // T splits where a coupling is negligible; those are zeroed for good, so that no rotation ever mixes two blocks and
    // the eigenvalue left at position j belongs to the block of T that contains j
    unsigned split = 0x8000u;
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        const bool ng = fabs(e[j]) <= 2.220446049250313e-16 * (fabs(d[j]) + fabs(d[j + 1]));
        split |= ng ? (1u << j) : 0u;
        e[j] = ng ? 0.0 : e[j];
    }
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
        qle_pos<14>(d, e, sn, cs, pp, g, alive, !done, l, m);
    }
    // stable ascending ranks among the n real entries (padding positions rank last, in place)
    int rk[16];
    unsigned rlo = 0, rhi = 0;
    double tnorm = 0.0;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        int r = 0;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const bool before = a < n ? (b < n && (d[b] < d[a] || (d[b] == d[a] && b < a))) : (b < n || b < a);
            r += before ? 1 : 0;
        }
        rk[a] = r;
        if (a < 8) rlo |= (unsigned)r << (4 * a);
        else rhi |= (unsigned)r << (4 * (a - 8));
        if (a < n) tnorm = fmax(tnorm, fabs(d[a]));
    }
    // two eigenvalues of one block closer than gaptol |T|: their twisted-factorisation vectors would be nearly parallel
    bool flagged = false;
    const double thr = gaptol * tnorm;
#pragma unroll
    for (int a = 0; a < 15; ++a) {
        bool same = true;                               // no split between a and b so far
#pragma unroll
        for (int b = a + 1; b < 16; ++b) {
            same = same && ((split >> (b - 1)) & 1u) == 0;
            flagged = flagged || (same && b < n && !(fabs(d[a] - d[b]) >= thr));
        }
    }
    if (has) {
#pragma unroll
        for (int j = 0; j < 16; ++j) lam[(int64_t)j * nc + idc] = d[j];
        meta[idc] = uint4{rlo, rhi, split, flagged ? 1u : 0u};
        if (flagged) list[atomicAdd(count, 1)] = (int)idc;
    }
    double prev = 0.0;
    for (int r = 0; r < n; ++r) {
        double v = 0.0;
#pragma unroll
        for (int a = 0; a < 16; ++a) v = (a < n && rk[a] == r) ? d[a] : v;
        if constexpr (MODE == 1) {
            if (r > 0) {
                double gap = has ? v - prev : INFINITY;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gap = fmin(gap, __shfl_xor(gap, o));
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* slotp = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + (r - 1);
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
                    if (bits < __hip_atomic_load(slotp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slotp, bits);
                }
            }
            prev = v;
        } else {
            if (has) eval[(int64_t)r * nk + id0 + idc] = v;
        }
    }
}

// ---- 3. eigenvectors
// 1 / p to full precision from the hardware estimate (two Newton steps), p already kept away from zero
__device__ __forceinline__ double tw_rcp(const double p) {
    double y = __builtin_amdgcn_rcp(p);
    y = fma(fma(-p, y, 1.0), y, y);
    y = fma(fma(-p, y, 1.0), y, y);
    return y;
}
#define TW_TINY 1e-290
__device__ __forceinline__ double tw_guard(const double p) { return fabs(p) < TW_TINY ? -TW_TINY : p; }

#define TW16_WAVE_LDS 9216   // bytes of LDS per wavefront: V of 4 matrices at a row stride of 18 doubles; the same region
                             // then stages the 4 reflector records (8704 B) and the output transposition (4 x 16 x 17 doubles)

template <int MODE>
__global__ __launch_bounds__(256, 3) void k_tw16_vectors(const int n, const int64_t nk, const int64_t id0, const int64_t nc,
                                                      const ModelView mv, const ListArgs Lst, const GridArgs G,
                                                      const double2* __restrict__ de, const double* __restrict__ lam,
                                                      const uint4* __restrict__ meta, const cd* __restrict__ refl,
                                                      int* __restrict__ list, int* __restrict__ count) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4 * TW16_WAVE_LDS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = lane & 15, mat = lane >> 4, g = lane >> 4;
    double* const Vs = reinterpret_cast<double*>(lds_all + wv * TW16_WAVE_LDS);     // [4][16][18]
    cd* const Rs = reinterpret_cast<cd*>(lds_all + wv * TW16_WAVE_LDS);             // [4][137] (odd stride: the four broadcasts of a read hit different banks)
    const int64_t wslot0 = ((int64_t)blockIdx.x * 4 + wv) * 4;                        // first matrix of this wavefront
    if (wslot0 >= nc) return;                                                         // (wave-uniform)
    const int64_t slot_u = wslot0 + mat;
    const bool live = slot_u < nc;
    const int64_t slot = live ? slot_u : nc - 1;
    const int64_t id = id0 + slot;

    // ---- (a) twisted factorisation of T - lambda_j within the unreduced block of position j
    const uint4 mt = meta[slot];
    const unsigned split = mt.z;
    double v[16];
    bool bad = false;
    {
        double d[16], e[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double2 t = de[(int64_t)i * nc + slot];
            d[i] = t.x;
            e[i] = (split >> i) & 1u ? 0.0 : t.y;      // (e_15 = 0: bit 15 is always set)
        }
        const double lj = lam[(int64_t)j * nc + slot];
        double tnorm = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            d[i] -= lj;                                // s_i = d_i - lambda
            tnorm = fmax(tnorm, fabs(d[i]));
        }
        // top-down pivots dp_{i+1} = s_{i+1} - e_i lp_i, lp_i = e_i / dp_i; bottom-up dm_i = s_i - e_i um_i, um_i = e_i / dm_{i+1}
        double lp[15], um[15];
        {
            double dp = d[0], dm = d[15];
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                lp[i] = e[i] * tw_rcp(tw_guard(dp));
                dp = fma(-e[i], lp[i], d[i + 1]);
                const int k = 14 - i;
                um[k] = e[k] * tw_rcp(tw_guard(dm));
                dm = fma(-e[k], um[k], d[k]);
            }
        }
        // gamma_k = s_k - e_{k-1} lp_{k-1} - e_k um_k; r = position of the smallest |gamma| inside the block of j
        const unsigned below_j = split & ((1u << j) - 1u);
        const int bl = below_j ? 32 - __builtin_clz(below_j) : 0;           // first position of the block
        const int bh = __builtin_ctz(split >> j) + j;                       // last position of the block
        double gmin = INFINITY, gam_r = 0.0;
        int r = j;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            double gk = d[k];
            if (k > 0) gk = fma(-e[k - 1], lp[k - 1], gk);
            if (k < 15) gk = fma(-e[k], um[k], gk);
            const bool in = k >= bl && k <= bh;
            if (in && fabs(gk) < gmin) {
                gmin = fabs(gk);
                gam_r = gk;
                r = k;
            }
        }
        // z_r = 1; downwards z_i = -lp_i z_{i+1} (i < r), upwards z_{i+1} = -um_i z_i (i >= r)
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = i == r ? 1.0 : 0.0;
#pragma unroll
        for (int i = 14; i >= 0; --i) v[i] = i < r ? -lp[i] * v[i + 1] : v[i];
#pragma unroll
        for (int i = 0; i < 15; ++i) v[i + 1] = i >= r ? -um[i] * v[i] : v[i + 1];
        double nz2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) nz2 = fma(v[i], v[i], nz2);
        const double inz = rsqrt_full(nz2);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= inz;
        // residual |(T - lambda) z| / |z| = |gamma_r| / |z|: a few eps |T| for an eigenvalue that accurate
        const double tn = tnorm + fabs(lj);
        bad = !(fabs(gam_r) * inz <= 1e-11 * tn) && j < n;
    }
    // a matrix whose vectors failed the residual test joins the list (once; not if step 2 listed it already)
    {
        const unsigned long long bal = __ballot(bad && live);
        const unsigned mine = (unsigned)(bal >> (lane & 48)) & 0xffffu;
        if (mine != 0 && mt.w == 0 && j == 0 && live) list[atomicAdd(count, 1)] = (int)slot;
    }

    // the reflector records of the four matrices: in flight while the matrix cores work
    double2 rr[9];
    {
        const double2* src = reinterpret_cast<const double2*>(refl + wslot0 * TW16_REC);
        const int64_t avail = (nc - wslot0 < 4 ? nc - wslot0 : 4) * TW16_REC;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int idx = t * 64 + lane;
            rr[t] = idx < avail ? src[idx] : double2{0.0, 0.0};
        }
    }

    // ---- (b) one Newton-Schulz step V <- V (1.5 I - 0.5 V^T V), per matrix two 16 x 16 x 16 real products on the matrix cores
    // v_mfma_f64_16x16x4_f64: lane l supplies A[l & 15][4 kb + (l >> 4)] and B[4 kb + (l >> 4)][l & 15], and holds
    // D[(l >> 4) + 4 r][l & 15] in register r (profiles/microbench/mfma_f64_layout.hip)
    {
        double* mine = Vs + (mat * 16 + j) * 18;
#pragma unroll
        for (int i = 0; i < 16; i += 2) *reinterpret_cast<double2*>(mine + i) = double2{v[i], v[i + 1]};
    }
    // (the region is private to this wavefront and a wavefront's LDS operations execute in order: no barrier, no wait --
    // only the compiler must keep the order, the views of the region differing in type)
    TW_LDS_ORDER();
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
        const double* Vm = Vs + m4 * 16 * 18;
        tw_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double a = Vm[j * 18 + 4 * kb + g];                // V[4 kb + g][j] = A^T and B alike: G = V^T V
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc, 0, 0, 0);
        }
        tw_d4 xr;                                                     // X[g + 4 r][j] = 1.5 delta - 0.5 G
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = fma(-0.5, acc[r], (g + 4 * r) == j ? 1.5 : 0.0);
        tw_d4 vn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const double a = Vm[(4 * kb + g) * 18 + j];              // A[x = j][k = 4 kb + g] = V[j][4 kb + g]
            vn = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xr[kb], vn, 0, 0, 0);   // B[k][col] = X[4 kb + g][col]: own register kb
        }
        // vn[r] = V'[x = g + 4 r][column j]: back into the column-major image, in place (all reads of this matrix are done)
        TW_LDS_ORDER();
#pragma unroll
        for (int r = 0; r < 4; ++r) Vs[(m4 * 16 + j) * 18 + g + 4 * r] = vn[r];
    }
    TW_LDS_ORDER();
    {
        const double* mine = Vs + (mat * 16 + j) * 18;
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const double2 t = *reinterpret_cast<const double2*>(mine + i);
            v[i] = t.x;
            v[i + 1] = t.y;
        }
    }
    TW_LDS_ORDER();

    // ---- (c) z = H_0 ( H_1 ( ... H_13 (D v))), H_K = I - us_K us_K^+; the reflectors are read from LDS by broadcast
    {
        double2* dst = reinterpret_cast<double2*>(Rs);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int idx = t * 64 + lane;
            if (idx < 4 * TW16_REC) dst[idx + idx / TW16_REC] = rr[t];      // record m at LDS entry 137 m
        }
    }
    TW_LDS_ORDER();
    const cd* __restrict__ R = Rs + mat * (TW16_REC + 1);
    cd y[16];
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) {
        const cd ph = R[119 + xx];
        y[xx] = cd{ph.x * v[xx], ph.y * v[xx]};
    }
    auto reflect = [&](auto KC) {
        constexpr int K = decltype(KC)::value;
        cd u[16];
        cd w{0.0, 0.0};
#pragma unroll
        for (int xx = K + 1; xx < 16; ++xx) {
            u[xx] = R[tw16_off(K) + xx - K - 1];
            // w += conj(u_x) y_x
            w.x = fma(u[xx].x, y[xx].x, w.x);
            w.x = fma(u[xx].y, y[xx].y, w.x);
            w.y = fma(u[xx].x, y[xx].y, w.y);
            w.y = fma(-u[xx].y, y[xx].x, w.y);
        }
#pragma unroll
        for (int xx = K + 1; xx < 16; ++xx) {
            // y_x -= u_x w
            y[xx].x = fma(-u[xx].x, w.x, y[xx].x);
            y[xx].x = fma(u[xx].y, w.y, y[xx].x);
            y[xx].y = fma(-u[xx].x, w.y, y[xx].y);
            y[xx].y = fma(-u[xx].y, w.x, y[xx].y);
        }
    };
    reflect(std::integral_constant<int, 13>{});
    reflect(std::integral_constant<int, 12>{});
    reflect(std::integral_constant<int, 11>{});
    reflect(std::integral_constant<int, 10>{});
    reflect(std::integral_constant<int, 9>{});
    reflect(std::integral_constant<int, 8>{});
    reflect(std::integral_constant<int, 7>{});
    reflect(std::integral_constant<int, 6>{});
    reflect(std::integral_constant<int, 5>{});
    reflect(std::integral_constant<int, 4>{});
    reflect(std::integral_constant<int, 3>{});
    reflect(std::integral_constant<int, 2>{});
    reflect(std::integral_constant<int, 1>{});
    reflect(std::integral_constant<int, 0>{});
    TW_LDS_ORDER();

    // ---- (d) transpose through LDS (row stride 17 doubles): lane c of a matrix receives component c of every vector
    double* const Ts = reinterpret_cast<double*>(lds_all + wv * TW16_WAVE_LDS);     // [4][16][17]
    cd zt[16];
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) Ts[(mat * 16 + j) * 17 + xx] = y[xx].x;
    TW_LDS_ORDER();
#pragma unroll
    for (int b = 0; b < 16; ++b) zt[b].x = Ts[(mat * 16 + b) * 17 + j];
    TW_LDS_ORDER();
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) Ts[(mat * 16 + j) * 17 + xx] = y[xx].y;
    TW_LDS_ORDER();
#pragma unroll
    for (int b = 0; b < 16; ++b) zt[b].y = Ts[(mat * 16 + b) * 17 + j];

    const int c = j;                                   // from here on the lane owns orbital component c
    if (!live || c >= n) return;
    cd f{1.0, 0.0};
    if constexpr (MODE != 2) {
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
        bool wrap[4] = {false, false, false, false};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
                if (dd < mv.dim_k) kk[dd] = Lst.k[id * mv.dim_k + dd];
        } else {
            grid_point(G, id, kk, wrap);
        }
        f = cconj(expi2pi(kdot(kk, mv.orb[c])));
        if constexpr (MODE == 1) {
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
                if (wrap[dd]) f = cmul(f, G.pbc[dd * n + c]);
        }
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        if (b < n) {
            const int r = (int)(((b < 8 ? mt.x : mt.y) >> (4 * (b & 7))) & 15u);
            const cd val = cmul(zt[b], f);
            if constexpr (MODE == 1) wf_at(G.wv, r, id)[c] = val;
            else Lst.evec[((int64_t)r * nk + id) * n + c] = val;
        }
    }
}

#ifndef TBK_TW16_KERNELS_ONLY   // (profiles/microbench/e16_bench.hip times the three kernels beside the fused one)
// ---- host side: chunks of the batch through the three kernels, then the listed matrices through the QL-replay kernels.
// Up to three chunks are in flight, each on a stream and a workspace of its own: the eigenvalue kernel is a dependent chain
// per lane (2 wavefronts per SIMD's worth of work at 43 % of the VALU issue rate, profiles/r03*), the other two are
// throughput kernels, and side by side they fill each other's gaps and tails.  The context's stream forks into the side
// streams at entry and waits for all of them at exit, so to every caller this is still one stream-ordered launch.
struct Tw16Work {
    double2* de;
    cd* refl;
    double* lam;
    uint4* meta;
    int* list;
    int* count;
    Ql16Rec R;
};

template <int MODE>
static int launch_tw16(tbk_ctx* ctx, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G) {
    const TbkKnobs& K = tbk_knobs();
    // sweeps recorded per LISTED matrix by the fallback (see launch_ql16)
    const int scap = K.qlw_cap > 0 ? (int)std::max<long long>(1, std::min<long long>(64, K.qlw_cap / 16)) : 64;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // per matrix: (d, e) | reflector record | eigenvalues by position | meta | list entry  +  the fallback's rotation record,
    // sweep words, sweep count and ranks (touched for listed matrices only)
    const bool fused = K.e16 != 0;   // TBK_E16 (default): the three main kernels are ONE, k_e16 (tbk_solve_e16.hip)
    // (fused: no reflector record, eigenvalue or meta array -- only the fallback's (d, e), list entry and rotation record)
    const size_t per_main = 16 * sizeof(double2) + (fused ? 0 : TW16_REC * sizeof(cd) + 16 * sizeof(double) + sizeof(uint4)) + sizeof(int);
    const size_t per_fb = (size_t)scap * 16 * sizeof(double2) + (size_t)scap * sizeof(unsigned) + sizeof(int) + 16;
    const size_t budget = (size_t)(K.qlw_ws_mb > 0 ? K.qlw_ws_mb : 8192) << 20;
    // fused: no (d, e) / reflector / eigenvalue arrays between kernels and no latency-bound lane-per-matrix kernel to hide behind
    // its neighbours, so one chunk in flight unless asked
    int ns = K.tw16_streams >= 1 ? std::min(K.tw16_streams, 3) : (fused ? 1 : 3);
    // (per-kernel HIP-event brackets exist with TBK_TW16_STREAMS=1 only; otherwise the caller's bracket on the context's
    // stream covers the fork and the join)
    if (nk < 4 * 16384) ns = 1;
    int64_t chunk = std::max<int64_t>(4096, (int64_t)(budget / ns / (per_main + per_fb)));
    chunk = std::min<int64_t>(chunk, (nk + ns - 1) / ns);
    {
        const int64_t nch = (nk + chunk - 1) / chunk;
        chunk = (nk + nch - 1) / nch;                                                // equal chunks
    }
    chunk = (chunk + 3) & ~(int64_t)3;                                               // whole wavefronts of four matrices
    TBK_REQUIRE(chunk < (int64_t)0x7fffffff / 16, TBK_EUNSUPPORTED, "chunk of %lld matrices", (long long)chunk);
    const size_t nmain = fused ? 0 : (size_t)chunk;     // entries of the arrays only the three-kernel form passes between its kernels
    const size_t wone = al((size_t)chunk * 16 * sizeof(double2)) + al(nmain * TW16_REC * sizeof(cd)) +
                        al(nmain * 16 * sizeof(double)) + al(nmain * sizeof(uint4)) + al((size_t)chunk * sizeof(int)) + 256 +
                        al((size_t)chunk * scap * 16 * sizeof(double2)) + al((size_t)chunk * scap * sizeof(unsigned)) +
                        al((size_t)chunk * sizeof(int)) + al((size_t)chunk * 16) + 1024;
    const size_t wbytes = wone * ns;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tridiagonal workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    Tw16Work W[3];
    for (int s = 0; s < ns; ++s) {
        unsigned char* p = (unsigned char*)ctx->work + (size_t)s * wone;
        W[s].de = (double2*)p;
        p += al((size_t)chunk * 16 * sizeof(double2));
        W[s].refl = (cd*)p;
        p += al(nmain * TW16_REC * sizeof(cd));
        W[s].lam = (double*)p;
        p += al(nmain * 16 * sizeof(double));
        W[s].meta = (uint4*)p;
        p += al(nmain * sizeof(uint4));
        W[s].list = (int*)p;
        p += al((size_t)chunk * sizeof(int));
        W[s].count = (int*)p;
        p += 256;
        W[s].R = Ql16Rec{};
        W[s].R.rot = (double2*)p;
        p += al((size_t)chunk * scap * 16 * sizeof(double2));
        W[s].R.swp = (unsigned*)p;
        p += al((size_t)chunk * scap * sizeof(unsigned));
        W[s].R.nit = (int*)p;
        p += al((size_t)chunk * sizeof(int));
        W[s].R.rank = (signed char*)p;
        W[s].R.scap = scap;
    }
    hipStream_t st[3] = {ctx->stream, ctx->stream, ctx->stream};
    if (ns > 1) {
        if (!ctx->side_ev[0])
            for (int i = 0; i < 4; ++i) TBK_HIP(hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming));
        TBK_HIP(hipEventRecord(ctx->side_ev[0], ctx->stream));
        for (int s = 0; s < ns; ++s) {
            if (!ctx->side[s]) TBK_HIP(hipStreamCreateWithFlags(&ctx->side[s], hipStreamNonBlocking));
            st[s] = ctx->side[s];
            TBK_HIP(hipStreamWaitEvent(st[s], ctx->side_ev[0], 0));
        }
    }
    cd* evec = MODE == 1 ? nullptr : L.evec;
    // the chunks; an error inside leaves through the join below like success does -- the side streams still use ctx->work, and
    // work issued later on ctx->stream (a retry on the Jacobi kernels, a workspace regrow) must be ordered after them
    auto run_chunks = [&]() -> int {
        int which = 0;
        for (int64_t id0 = 0; id0 < nk; id0 += chunk, which = (which + 1) % ns) {
            const int64_t nc = std::min<int64_t>(chunk, nk - id0);
            const unsigned b16 = (unsigned)((nc * 16 + 255) / 256), b1 = (unsigned)((nc + 255) / 256);
            const Tw16Work& w = W[which];
            hipStream_t sq = st[which];
            const bool brackets = ns == 1;
            TBK_HIP(hipMemsetAsync(w.count, 0, sizeof(int), sq));
            if (fused) {
                ProfScope ps(brackets ? ctx : nullptr, "e16");
                const int rc = tbk_e16_launch(MODE, sq, mv, nk, L, G, id0, nc, w.list, w.count, K.tw16_gaptol, (K.e16_ns_full ? E16_F_NS_FULL : 0) | (K.e16_cells ? 0 : E16_F_NO_CELLS));
                if (rc) return rc;
            } else {
            {
                ProfScope ps(brackets ? ctx : nullptr, "tw16_tridiag");
                hipLaunchKernelGGL((k_tw16_tridiag<MODE>), dim3(b16), dim3(256), 0, sq, mv, nk, L, G, w.de, w.refl, id0, nc);
            }
            {
                ProfScope ps(brackets ? ctx : nullptr, "tw16_eigvals");
                hipLaunchKernelGGL((k_tw16_eigvals<MODE>), dim3(b1), dim3(256), 0, sq, mv.nsta, nk, id0, nc, (const double2*)w.de, L.eval, G, w.lam,
                                   w.meta, w.list, w.count, ctx->flags_dev, K.tw16_gaptol);
            }
            {
                ProfScope ps(brackets ? ctx : nullptr, "tw16_vectors");
                hipLaunchKernelGGL((k_tw16_vectors<MODE>), dim3(b16), dim3(256), 0, sq, mv.nsta, nk, id0, nc, mv, L, G, (const double2*)w.de,
                                   (const double*)w.lam, (const uint4*)w.meta, (const cd*)w.refl, w.list, w.count);
            }
            }
            // the listed matrices once more, by QL with replayed rotations: the count stays on the device, so these are small fixed
            // grids whose blocks stride over the list (an empty list costs three launches of idle blocks)
            {
                ProfScope ps(brackets ? ctx : nullptr, "tw16_fallback");
                const unsigned f16 = std::min<unsigned>(b16, 4u * (unsigned)ctx->cus), f1 = std::min<unsigned>(b1, 4u * (unsigned)ctx->cus);
                hipLaunchKernelGGL((k_solve_ql16<MODE, true, 2, true>), dim3(f16), dim3(256), 0, sq, mv, nk, L, G, ctx->flags_dev, w.de, id0, nc,
                                   (const int*)w.list, (const int*)w.count);
                // (the fused kernel leaves the minimal gaps of the matrices it lists to this kernel; round 3's eigenvalue kernel took them itself)
                hipLaunchKernelGGL((k_ql16_lanes<MODE, true>), dim3(f1), dim3(256), 0, sq, mv.nsta, nk, id0, nc, (const double2*)w.de, L.eval, G, w.R,
                                   ctx->flags_dev, (const int*)w.list, (const int*)w.count, fused);
                hipLaunchKernelGGL((k_ql16_replay<MODE, true>), dim3(f16), dim3(256), 0, sq, mv.nsta, nk, id0, nc, w.R, evec, G.wv, (const int*)w.list,
                                   (const int*)w.count, (unsigned long long*)(ctx->flags_dev + TBK_FLAG_LISTED));
            }
        }
        return TBK_OK;
    };
    const int rc_chunks = run_chunks();
    int rc_join = TBK_OK;
    if (ns > 1) {
        for (int s = 0; s < ns; ++s) {
            hipError_t e = hipEventRecord(ctx->side_ev[1 + s], st[s]);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->side_ev[1 + s], 0);
            if (e != hipSuccess && rc_join == TBK_OK) {
                (void)hipStreamSynchronize(st[s]);                                   // (the ordering could not be expressed: wait here)
                tbk_set_error("joining the side streams of the n = 9..16 solver: %s", hipGetErrorString(e));
                rc_join = TBK_EHIP;
            }
        }
    }
    if (rc_chunks) return rc_chunks;
    if (rc_join) return rc_join;
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
#endif  // TBK_TW16_KERNELS_ONLY
