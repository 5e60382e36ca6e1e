#ifndef HH32_SYNC
#define HH32_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif
// tbk_solve_hh32.inl -- included by tbk_solve_qlw.inl (in front of its launcher).  Stage 1 of the direct solver for n = 17..32 states
// (pythtb.py:939-947 `_sol_ham`; the models of cut_piece / make_supercell, :1105-1637) with the matrix in REGISTERS (round 6).
//
// k_tridiag_lds (round 2) keeps A in LDS and works on it with a 128-thread workgroup: thread = (row, column strip), partial sums
// through LDS, three workgroup barriers per reflection and two more per reflector of the accumulation.  At 17..32 states that is
// all latency: 36 k matrices of 32 states took 2.3 ms of the path's 4.3 (17 states: 0.71 of 1.4 ms; profiles/n17_probe.py, round 6)
// -- 80 us per matrix for 0.6 M multiply-adds.  Here ONE wavefront owns a matrix and nothing but the broadcasts goes through LDS:
//   * tridiagonalisation: lane (x, h) = row x, columns h, h + 2, h + 4, ... of A in registers (the two halves of the wavefront
//     take the columns in turn, so the live ones stay evenly split as the reflections advance and a step's loops start at slot K / 2).  Reflection K (a compile-time constant: every register index is static): the column
//     below the diagonal is a[K % HC] of the half that owns column K, handed to the other half by one cross-half shuffle; u and
//     q reach the columns as broadcast LDS reads (every lane of a half reads the same address); p = A u and the rank-2 update
//     run over a lane's own columns, the halves' partial sums meet in one shuffle.  No workgroup barrier (one wavefront per
//     workgroup; its LDS operations execute in order).
//   * Z = H_0 (H_1 (... (H_{n-3} D))) (eigenvectors wanted): lane (c, h) = column c, rows h HC .. of Z in registers; a
//     reflector's entries arrive as broadcast reads of the record stage 1 left in LDS (rows above the reflector hold zeros, so
//     the loops need no masks).  Z leaves through LDS, transposed to the band-major layout, orbital phases applied on the way --
//     what k_tridiag_lds writes.
// Same mathematics, same conventions as k_tridiag_lds (zlarfg-style reflectors, beta = 2 / u^+ u, the subdiagonal made real by the
// diagonal unitary D): (d, e) go to the same workspace, Z to the same place, and stages 2 and 3 (k_tridiag_ql_lanes, k_ql_replay_reg)
// follow unchanged.  Not the same bits as k_tridiag_lds (the partial sums are cut differently); every point is still solved on its
// own, so periodic images, halo rows and shard windows stay bit-identical.  TBK_HH32=0: k_tridiag_lds as before.
//
// Measured (profiles/hh32_sweep.py, 33^3 points, the whole three-stage call): n = 18 1.42 ms (k_tridiag_lds: 1.46), 20 1.57 (1.72),
// 24 1.87 (2.24), 28 3.00 (3.18), 32 3.68 (4.05) -- 3 to 17 %; at 17 states a tie, left on the old kernel.  The stage itself: 1.76
// against 2.32 ms at 32 states.  It is not the 2-3 x its arithmetic allows: a wavefront's step is a chain of ~6 dependent LDS round
// trips, two 32-lane sums and two reciprocal square roots (~1.5 k cycles of latency for ~130 instructions), and 19 KB of LDS (the
// staged matrix / the record / Z on its way out) leave two wavefronts per SIMD to hide it behind (k_e16 keeps twelve matrices per
// SIMD in flight).  The first version, with the columns split in two blocks instead of in turn, did 60 % more multiply-adds on dead
// columns and was no faster than k_tridiag_lds at all (profiles/HISTORY.md).

// the value the lane 32 away holds: two v_permlane32_swap_b32 (vector ALU; __shfl_xor(v, 32) is a ds_bpermute round trip)
__device__ __forceinline__ double hh32_xhalf(const double v, const bool upper) {
    const uint2 w = __builtin_bit_cast(uint2, v);
    const auto lo = __builtin_amdgcn_permlane32_swap(w.x, w.x, false, false);    // [0]: lower half's value in both halves, [1]: upper half's
    const auto hi = __builtin_amdgcn_permlane32_swap(w.y, w.y, false, false);
    return __builtin_bit_cast(double, uint2{upper ? lo[0] : lo[1], upper ? hi[0] : hi[1]});
}
// the value of lane SRC (a constant) in every lane: scalar registers
__device__ __forceinline__ double hh32_lane(const double v, const int src) {
    const uint2 w = __builtin_bit_cast(uint2, v);
    return __builtin_bit_cast(double, uint2{(unsigned)__builtin_amdgcn_readlane((int)w.x, src), (unsigned)__builtin_amdgcn_readlane((int)w.y, src)});
}
// sum over the 32 lanes of a half (both halves hold the same data: the sum over all 64 lanes, halved exactly, would differ in rounding)
__device__ __forceinline__ double hh32_allsum(double v) {
    v = row_allsum(v);
    const uint2 w = __builtin_bit_cast(uint2, v);
    const auto lo = __builtin_amdgcn_permlane16_swap(w.x, w.x, false, false);    // rows 1 <-> 0 and 3 <-> 2 exchanged between the two results
    const auto hi = __builtin_amdgcn_permlane16_swap(w.y, w.y, false, false);
    const bool odd = (threadIdx.x & 16) != 0;
    const double o = __builtin_bit_cast(double, uint2{odd ? lo[0] : lo[1], odd ? hi[0] : hi[1]});
    return v + o;
}

#ifndef HH32_OCC
#define HH32_OCC __attribute__((amdgpu_waves_per_eu(3, 8)))
#endif
// the reflector record k_hh32<.., 2, NMAX> leaves per matrix for k_tw32_vectors<.., NMAX, true> (16-byte entries): sqrt(beta_K) u_K[r],
// r = K + 1 .. NMAX - 1, packed reflector after reflector (zeros past n), then the phases D[NMAX], then the orbital phases [NMAX]
__host__ __device__ constexpr int hh32_rec_off(const int K, const int NMAX) { return (NMAX - 1) * K - K * (K - 1) / 2; }
__host__ __device__ constexpr int hh32_rec_size(const int NMAX) { return hh32_rec_off(NMAX - 2, NMAX) + 2 * NMAX; }

// VEC 0: (d, e) only; 1: Z = H_0 .. H_{n-3} D with the orbital phases on its rows into the output array; 2: the reflector record
// into `refl` (round 6: the accumulation of Z was 0.67 of this kernel's 1.75 ms per 36 k matrices of 32 states)
template <int MODE, int VEC, int NMAX>
__global__ __launch_bounds__(64) HH32_OCC void k_hh32(const ModelView mv, const int64_t nk, const ListArgs L, const GridArgs G, const int64_t id0,
                                              const int64_t nchunk, double2* __restrict__ de, cd* __restrict__ refl = nullptr) {
    static_assert(NMAX == 20 || NMAX == 24 || NMAX == 28 || NMAX == 32, "k_hh32: 17..20, 21..24, 25..28 or 29..32 states");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int HC = NMAX / 2;
    const int n = mv.nsta, ld = n | 1;
    const int tid = threadIdx.x, x = tid & 31, h = tid >> 5;
    // VEC == 1: [n][ld]: H(k) as assembled; then the reflector record U[K][r] (stride n); then Z on its way out.  Otherwise only the upper
    // triangle of H(k), packed (8.4 instead of 16.9 KB at 32 states: three wavefronts per SIMD instead of two)
    constexpr bool TRI = VEC != 1;
    cd* A = (cd*)lds_raw;
    cd* ubuf = A + (TRI ? n * (n + 1) / 2 : n * ld);   // [32]
    cd* qbuf = ubuf + 32;                  // [32]
    cd* eo = qbuf + 32;                    // [max(n, nR)]: assembly scratch, then the orbital phases
    cd* dphase = eo + (n > mv.nR ? n : mv.nR);
    cd* tsub = dphase + n;
    double* tau = (double*)(tsub + n);
    double* eb = tau + n;
    const int64_t idc = blockIdx.x, id = id0 + idc;

    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
    } else if constexpr (MODE == 1) {
        grid_point(G, id, kk, wrap);
    }
    assemble_lds<MODE, 64, TRI>(mv, L, id, kk, A, ld, eo, tid);
    __syncthreads();
    if (VEC && tid < n) {
        cd f{1.0, 0.0};
        if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[tid])));
        if constexpr (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (wrap[d]) f = cmul(f, G.pbc[d * n + tid]);
        }
        eo[tid] = f;
    }
    // this lane's part of row x
    cd a[HC];
#pragma unroll
    for (int j = 0; j < HC; ++j) {
        const int c = 2 * j + h;
        if constexpr (TRI) a[j] = (x < n && c < n) ? (c <= x ? cconj(A[x * (x + 1) / 2 + c]) : A[c * (c + 1) / 2 + x]) : cd{0.0, 0.0};
        else a[j] = (x < n && c < n) ? A[x * ld + c] : cd{0.0, 0.0};
    }
    cd* const rec_out = VEC == 2 ? refl + idc * hh32_rec_size(NMAX) : nullptr;
    HH32_SYNC();                       // (the region takes the reflector record from here on)
    if (tid < n) {
        tau[tid] = 0.0;
        tsub[tid] = cd{0.0, 0.0};
    }

    // ---- 1. reflections K = 0 .. n-3.  Every step is written WITHOUT a branch on n: rows and columns past n hold zeros, so a step
    // past n - 3 finds nothing below the subdiagonal (rest = 0), takes the subdiagonal entry as it stands (K = n - 2: the entry
    // that is never reflected) and changes nothing.  (Steps under `if (K + 2 < n)` -- a chain of correlated uniform conditions --
    // sent the compiler's running time through the roof: 8 steps 4 minutes, 31 steps did not end.)
    static_for<0, NMAX - 1>([&](auto kt) __attribute__((always_inline)) {
        constexpr int K = decltype(kt)::value;
        if (K + 1 >= n) return;            // (uniform) nothing below the diagonal any more
        // (the row passes through an empty assembly statement once per step: InstCombine's floating-point class analysis walks the
        // dependence TREE of a value -- sixteen operands per level here -- and took a minute for six steps, exponentially more beyond)
#pragma unroll
        for (int j = 0; j < HC; ++j) asm("" : "+v"(a[j].x), "+v"(a[j].y));
        const cd own = a[K / 2];
        const cd oth{hh32_xhalf(own.x, h != 0), hh32_xhalf(own.y, h != 0)};
        const bool below = x > K && x < n;
        const bool mine = h == (K & 1);
        const cd colx = below ? cd{mine ? own.x : oth.x, mine ? own.y : oth.y} : cd{0.0, 0.0};
        const cd alpha{hh32_lane(colx.x, K + 1), hh32_lane(colx.y, K + 1)};
        // |rows > K+1 of the column|^2: a reflection is needed iff this is non-zero (like LAPACK's zlarfg)
        const double rest = hh32_allsum(x > K + 1 ? cabs2(colx) : 0.0);
        const double absa2 = cabs2(alpha);
        const double sigma = rest + absa2;
        // (no uniform branch around the step either: a step that has nothing to reflect runs with u = 0 and changes nothing)
        const bool refl = rest > 0.0;
        const double sig1 = refl ? sigma : 1.0;
        const double inv_n = rsqrt_full(sig1), nrm = sig1 * inv_n;
        const bool hasa = absa2 > 0.0;
        const double inv_a = rsqrt_full(hasa ? absa2 : 1.0);
        const double absa = hasa ? absa2 * inv_a : 0.0;
        const cd ph{hasa ? alpha.x * inv_a : 1.0, hasa ? alpha.y * inv_a : 0.0};
        const cd u0 = x == K + 1 ? cd{ph.x * (absa + nrm), ph.y * (absa + nrm)} : colx;   // (zero above the reflector)
        const cd u{refl ? u0.x : 0.0, refl ? u0.y : 0.0};
        const double beta = refl ? 1.0 / (nrm * (nrm + absa)) : 0.0;
        const cd tK{refl ? -ph.x * nrm : alpha.x, refl ? -ph.y * nrm : alpha.y};
        if (h == 0) {
            ubuf[x] = u;
            if constexpr (VEC == 1) {
                if (x < n && K + 2 < n) A[K * n + x] = u;  // the record the accumulation reads ((n - 2) n entries: the region holds n (n | 1))
            }
            if constexpr (VEC == 2) {                  // sqrt(beta) u_K straight into the record of k_tw32_vectors (zeros past n)
                if (K + 2 < NMAX && x > K && x < NMAX) {
                    const double sb = beta > 0.0 ? beta * rsqrt_full(beta) : 0.0;
                    rec_out[hh32_rec_off(K, NMAX) + x - K - 1] = cd{u.x * sb, u.y * sb};
                }
            }
        }
        HH32_SYNC();
        cd p{0.0, 0.0};
#pragma unroll
        for (int j = 0; j < HC; ++j) {
            if (2 * j + 1 > K) cfma(p, a[j], ubuf[2 * j + h]);          // (columns <= K: u is zero there; both halves' dead ones skipped)
        }
        p = cd{p.x + hh32_xhalf(p.x, h != 0), p.y + hh32_xhalf(p.y, h != 0)};
        const cd ps{p.x * beta, p.y * beta};
        const double kappa = 0.5 * beta * hh32_allsum(u.x * ps.x + u.y * ps.y);
        const cd q = below ? cd{ps.x - kappa * u.x, ps.y - kappa * u.y} : cd{0.0, 0.0};
        if (h == 0) qbuf[x] = q;
        HH32_SYNC();
#pragma unroll
        for (int j = 0; j < HC; ++j) {
            if (2 * j + 1 > K) {   // A[x][c] -= u_x conj(q_c) + q_x conj(u_c)
                const cd uc = ubuf[2 * j + h], qc = qbuf[2 * j + h];
                a[j].x -= (u.x * qc.x + u.y * qc.y) + (q.x * uc.x + q.y * uc.y);
                a[j].y -= (u.y * qc.x - u.x * qc.y) + (q.y * uc.x - q.x * uc.y);
            }
        }
        HH32_SYNC();           // (ubuf / qbuf are rewritten by the next reflection)
        if (tid == 0 && K + 1 < n) {
            tau[K] = beta;
            tsub[K] = tK;
        }
    });
    // the diagonal: A[x][x] sits in the half that owns column x
    double dx = 0.0;
#pragma unroll
    for (int j = 0; j < HC; ++j) dx = (2 * j + h == x) ? a[j].x : dx;
    dx += hh32_xhalf(dx, h != 0);
    HH32_SYNC();
    // subdiagonal moduli and the phases D_{k+1} = D_k t_k / |t_k|
    if (tid < n) {
        double mag = 0.0;
        cd f{1.0, 0.0};
        if (tid + 1 < n) {
            const cd t = tsub[tid];
            const double t2 = cabs2(t);
            if (t2 > 0.0) {
                const double inv = rsqrt_full(t2);
                mag = t2 * inv;
                f = cd{t.x * inv, t.y * inv};
            }
        }
        eb[tid] = mag;
        tsub[tid] = f;
    }
    HH32_SYNC();
    if (tid == 0) {
        cd delta{1.0, 0.0};
        dphase[0] = delta;
        for (int k = 0; k + 1 < n; ++k) {
            delta = cmul(delta, tsub[k]);
            dphase[k + 1] = delta;
        }
    }
    if (tid < n) de[(int64_t)tid * nchunk + idc] = double2{dx, eb[tid]};
    if constexpr (VEC == 0) return;
    HH32_SYNC();
    if constexpr (VEC == 2) {
        constexpr int P = hh32_rec_off(NMAX - 2, NMAX);
        cd* out = refl + idc * hh32_rec_size(NMAX);
        for (int K = (n > 1 ? n - 1 : 0) + h; K < NMAX - 2; K += 2) {   // the reflectors that do not exist (steps the loop above left out): zeros
            const int r = K + 1 + x;
            if (r < NMAX) out[hh32_rec_off(K, NMAX) + x] = cd{0.0, 0.0};
        }
        if (x < NMAX) out[P + h * NMAX + x] = x < n ? (h == 0 ? dphase[x] : eo[x]) : cd{0.0, 0.0};
        return;
    }

    // ---- 2. Z = H_0 (H_1 ( ... (H_{n-3} D))): lane (c, h) holds rows h HC .. of column c = x
    cd z[HC];
#pragma unroll
    for (int j = 0; j < HC; ++j) z[j] = (2 * j + h == x && x < n) ? dphase[x] : cd{0.0, 0.0};
    static_for<0, NMAX - 2>([&](auto it) __attribute__((always_inline)) {
        constexpr int K = NMAX - 3 - decltype(it)::value;
        if (K + 2 >= n) return;            // (uniform) no reflector with this number
#pragma unroll
        for (int j = 0; j < HC; ++j) asm("" : "+v"(z[j].x), "+v"(z[j].y));
        {
            const double beta = K + 2 < n ? tau[K] : 0.0;        // (no branch: a step without a reflector has beta = 0 and a zero record)
            cd t{0.0, 0.0};
#pragma unroll
            for (int j = 0; j < HC; ++j) {
                if (2 * j + 1 > K) {       // (rows <= K of the record are zero; the dead rows of BOTH halves are skipped)
                    const int r = 2 * j + h;
                    const cd ur = r < n && K + 2 < n ? A[K * n + r] : cd{0.0, 0.0};
                    cfmac(t, ur, z[j]);                        // t_c += conj(u_r) Z[r][c]
                }
            }
            t = cd{t.x + hh32_xhalf(t.x, h != 0), t.y + hh32_xhalf(t.y, h != 0)};
            const cd ts{t.x * beta, t.y * beta};
#pragma unroll
            for (int j = 0; j < HC; ++j) {
                if (2 * j + 1 > K) {       // Z[r][c] -= u_r (beta t_c)
                    const int r = 2 * j + h;
                    const cd ur = r < n && K + 2 < n ? A[K * n + r] : cd{0.0, 0.0};
                    z[j].x -= ur.x * ts.x - ur.y * ts.y;
                    z[j].y -= ur.x * ts.y + ur.y * ts.x;
                }
            }
        }
    });
    HH32_SYNC();                       // the record has been read: the region takes Z, row-major with stride ld
#pragma unroll
    for (int j = 0; j < HC; ++j) {
        const int r = 2 * j + h;
        if (r < n && x < n) A[r * ld + x] = z[j];
    }
    HH32_SYNC();
    // rows carry the orbital phase (and periodic-image phases) from here on: the rotations of step 3 act on columns
    for (int e = tid; e < n * n; e += 64) {
        const int b = e / n, o = e - b * n;
        const cd val = cmul(A[o * ld + b], eo[o]);
        if constexpr (MODE == 1) wf_at(G.wv, b, id)[o] = val;
        else L.evec[((int64_t)b * nk + id) * n + o] = val;
    }
}
