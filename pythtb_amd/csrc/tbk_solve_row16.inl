// tbk_solve_row16.inl -- included by tbk_solve.hip.
//
// n = 9..16 states per k: two-sided Jacobi with the matrix in REGISTERS, one DPP row (16 lanes) per
// matrix, four matrices per wavefront, no LDS storage and no barriers.
//
//   lane x of a row holds ROW x of A (16 complex) and COLUMN x of V^T (v[b] = component x of
//   eigenvector b).  The round-robin schedule is unrolled at compile time, so the column
//   rotations A <- A J and V <- V J of a round touch only registers with static indices, the
//   same ones in every lane.  The row half A <- J^+ A mixes the rows of the two partners of a
//   pair: each lane fetches its partner's row with ds_bpermute (the LDS crossbar, no LDS memory)
//   and forms its new row  c a + w b.  Rotation parameters are computed by the two lanes of a
//   pair from their own registers (the element facing the partner is picked with a 16-way
//   select) and handed to the whole row with DPP row broadcasts.
//
// The wavefront-per-matrix LDS kernel moves ~22 KB through LDS per round and matrix and is bound by
// LDS bandwidth; this layout exchanges 1 KB per round and matrix through the crossbar.
// Matrices smaller than 16 are padded with decoupled zero rows (never rotated, ranked last).
// Every point starts cold, so a point's result depends on the point alone (periodic images, halo
// rows and shard windows are bit-identical by construction).

// (I2, bperm_*, rowbcast_d / rowbcast_i, sel16: tbk_solve_dev.h)

// round R of the 15-round tournament on 16 players: pair l = 0 is (15, R), pair l = 1..7 is
// ((R+l) mod 15, (R-l) mod 15)
template <int R, int L>
struct Pair16 {
    static constexpr int p = L == 0 ? 15 : (R + L) % 15;
    static constexpr int q = L == 0 ? R : (R - L + 15) % 15;
};

// column rotations of one round on the registers of a row (A) and of a column of V^T
template <int R, int L, bool VEC>
__device__ __forceinline__ void row16_cols(cd (&a)[16], cd (&v)[16], const double c_own, const cd sw_own) {
    constexpr int P = Pair16<R, L>::p, Q = Pair16<R, L>::q;
    const double c = rowbcast_d<P>(c_own);
    const cd sw{rowbcast_d<P>(sw_own.x), rowbcast_d<P>(sw_own.y)};
    {   // a'_rp = c a_rp - conj(s) a_rq ;  a'_rq = s a_rp + c a_rq
        const cd x = a[P], y = a[Q];
        a[P] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
        a[Q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
    }
    if (VEC) {
        const cd x = v[P], y = v[Q];
        v[P] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
        v[Q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
        // Nothing reads V until the end, so the scheduler would postpone these updates and keep every
        // round's broadcast parameters alive instead (~50 registers per round, 1.8 KB of scratch per
        // lane over a sweep).  Pin them here: 252 VGPRs, no spills.
        asm volatile("" : "+v"(v[P].x), "+v"(v[P].y), "+v"(v[Q].x), "+v"(v[Q].y));
    }
    __builtin_amdgcn_sched_barrier(0);   // one pair's parameters live at a time
    if constexpr (L + 1 < 8) row16_cols<R, L + 1, VEC>(a, v, c_own, sw_own);
}

template <int R, bool VEC>
__device__ __forceinline__ void row16_round(cd (&a)[16], cd (&v)[16], const int x_in, const int rowbase4) {
    // The pairing below depends only on (lane, R), so the optimiser would hoist all 15 rounds' worth of
    // it (indices, 16-way select masks) out of the sweep loop and drown in registers: launder the lane
    // index once per round so that it is recomputed where it is used (a dozen integer instructions).
    int x = x_in;
    asm volatile("" : "+v"(x));
    // partner of this lane in round R, and whether this lane is the "p" of its pair
    int y, l = x - R;
    l += l < 0 ? 15 : 0;                       // (x - R) mod 15 for x < 15
    if (x == 15) {
        y = R;
    } else if (x == R) {
        y = 15;
    } else {
        y = 2 * R - x;
        y += y < 0 ? 15 : 0;
        y -= y >= 15 ? 15 : 0;
    }
    const bool is_p = x == 15 || (l >= 1 && l <= 7);
    const int paddr = rowbase4 + 4 * y;
    // ---- rotation parameters of this lane's pair (both lanes compute the same numbers)
    const cd gy = sel16<0>(a, y, cd{0.0, 0.0});          // A[x][y]
    const double d_own = sel16<0>(a, x, cd{0.0, 0.0}).x; // A[x][x]
    const double d_par = bperm_d(paddr, d_own);
    const cd g = is_p ? gy : cd{gy.x, -gy.y};            // A[p][q]
    const double dp = is_p ? d_own : d_par, dq = is_p ? d_par : d_own;
    double c = 1.0;
    cd sw{0.0, 0.0};
    const double g2 = cabs2(g);
    if (g2 > 0.0) {   // same division-free parameters as rotate<>
        const double h = 0.5 * (dq - dp), ah = fabs(h);
        const double r = sqrt(h * h + g2);
        const double inv = rsqrt(2.0 * r * (r + ah));
        const double sg = copysign(1.0, h);
        c = (ah + r) * inv;
        sw = cd{sg * g.x * inv, sg * g.y * inv};
    }
    // The q lane must use EXACTLY the numbers of the p lane.  Its own copy comes from A[q][p], which
    // differs from conj(A[p][q]) by rounding; a row rotated with parameters that differ by d(theta)
    // from the column rotation picks up d(theta) * A[p][p] -- an asymmetry amplified by
    // |diagonal| / |gap| per rotation, which blows up within a few sweeps (measured).  Three
    // doubles through the crossbar fix it.
    {
        const double c_p = bperm_d(paddr, c);
        const cd sw_p = bperm_c(paddr, sw);
        if (!is_p) {
            c = c_p;
            sw = sw_p;
        }
    }
    // ---- columns (all pairs of the round, static registers), V likewise
    row16_cols<R, 0, VEC>(a, v, c, sw);
    // ---- rows: new row = c * own + w * partner's, w = -s for p, conj(s) for q
    const cd w = is_p ? cd{-sw.x, -sw.y} : cd{sw.x, -sw.y};
    // (four columns at a time: 16 crossbar reads in flight, then their FMAs -- keeps the live set small)
#pragma unroll
    for (int j0 = 0; j0 < 16; j0 += 4) {
        cd b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = bperm_c(paddr, a[j0 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const cd t{c * a[j0 + j].x + (w.x * b[j].x - w.y * b[j].y), c * a[j0 + j].y + (w.x * b[j].y + w.y * b[j].x)};
            // the element facing the partner is the one this rotation annihilates: exactly zero, so the
            // off-diagonal norm can fall below the rounding floor of the diagonal (convergence test)
            a[j0 + j] = y == j0 + j ? cd{0.0, 0.0} : t;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int R, bool VEC>
__device__ __forceinline__ void row16_sweep(cd (&a)[16], cd (&v)[16], const int x, const int rowbase4) {
    row16_round<R, VEC>(a, v, x, rowbase4);
    if constexpr (R + 1 < 15) row16_sweep<R + 1, VEC>(a, v, x, rowbase4);
}

template <int J>
__device__ __forceinline__ void row16_bcast_diag(const cd (&a)[16], double (&ev)[16]) {
    ev[J] = rowbcast_d<J>(a[J].x);
    if constexpr (J + 1 < 16) row16_bcast_diag<J + 1>(a, ev);
}
template <int J>
__device__ __forceinline__ void row16_bcast_rank(const int rk, int (&rks)[16]) {
    rks[J] = rowbcast_i<J>(rk);
    if constexpr (J + 1 < 16) row16_bcast_rank<J + 1>(rk, rks);
}
template <int J>
__device__ __forceinline__ void row16_bcast_phase(const cd ph, cd (&phs)[16]) {
    phs[J] = cd{rowbcast_d<J>(ph.x), rowbcast_d<J>(ph.y)};
    if constexpr (J + 1 < 16) row16_bcast_phase<J + 1>(ph, phs);
}

// MODE 0: k list, 1: regular mesh into a wf_array (+ min gaps), 2: supplied matrices
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_solve_row16(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G,
                                                      int* noconv_flag) {
    const int lane = threadIdx.x & 63;
    const int x = lane & 15;
    const int rowbase4 = (lane & 48) * 4;
    const int64_t mat = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = mat < nk;
    const int64_t id = live ? mat : nk - 1;   // idle tail rows shadow the last point
    const int n = mv.nsta;
    const bool real_row = x < n;
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    cd a[16], v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        a[c] = cd{0.0, 0.0};
        v[c] = cd{c == x ? 1.0 : 0.0, 0.0};
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
        cd z[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        // S[x][c] = sum_R U_R[slot(min,max)] e^{2 pi i k.R}  (conjugated below the diagonal)
        int sidx[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int lo = x < c ? x : c, hi = x < c ? c : x;
            sidx[c] = real_row && c < n ? lo * n - lo * (lo - 1) / 2 + (hi - lo) : -1;
        }
        for (int base = 0; base < mv.nR; base += 16) {
            const int mine = base + x;
            const cd ph = mine < mv.nR ? phase_of_R(z, mv.rvec[mine]) : cd{0.0, 0.0};
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
    // ---- Jacobi sweeps: a row group stops rotating once ITS matrix has converged
    bool done = false;
    for (int sweep = 0; sweep <= TBK_JACOBI_MAX_SWEEPS; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const double t = cabs2(a[c]);
            off += c == x ? 0.0 : t;
            dia += c == x ? a[c].x * a[c].x : 0.0;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            off += __shfl_xor(off, o);
            dia += __shfl_xor(dia, o);
        }
        done = off <= 2.0e-32 * (dia + off);
        if (__all(done)) break;
        if (sweep == TBK_JACOBI_MAX_SWEEPS) {
            if (!done && x == 0) atomicExch(noconv_flag, 1);
            break;
        }
        if (!done) row16_sweep<0, VEC>(a, v, x, rowbase4);
    }
    // ---- eigenvalues in stable ascending order (padding rows rank last)
    double ev[16];
    row16_bcast_diag<0>(a, ev);
    const double mine = sel16<0>(a, x, cd{0.0, 0.0}).x;
    int rk = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool jr = j < n;
        const bool before = real_row ? (jr && (ev[j] < mine || (ev[j] == mine && j < x))) : (jr || j < x);
        rk += before ? 1 : 0;
    }
    if (Lst.natural) rk = x;
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
                    const cd val = cmul(v[b], f);
                    if constexpr (MODE == 1) wf_at(G.wv, rks[b], id)[x] = val;
                    else Lst.evec[((int64_t)rks[b] * nk + id) * n + x] = val;
                }
            }
        }
    }
}

template <int MODE, bool VEC>
static int launch_row16(tbk_ctx* ctx, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G) {
    TBK_REQUIRE(nk * 16 < (int64_t)0x7fffffff * 256, TBK_EUNSUPPORTED, "too many k-points for one launch");
    const unsigned blocks = (unsigned)((nk * 16 + 255) / 256);
    hipLaunchKernelGGL((k_solve_row16<MODE, VEC>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G, ctx->flags_dev);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
