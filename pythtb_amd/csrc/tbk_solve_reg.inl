// tbk_solve_reg.inl -- included by tbk_solve.hip.
//
// n = 5..8 states per k: cyclic Jacobi with the matrix in REGISTERS, like the n <= 4 kernels,
// instead of one wavefront per matrix through LDS (which costs ~1000 cycles of barrier and LDS
// latency per Jacobi round no matter how small the matrix is).
//
// One thread per matrix (L = 1).  Eigenvalues only: n = 8 needs 162 VGPRs, no spills.  With
// eigenvectors V does not fit next to A in 256 VGPRs (n = 8: 128 more doubles); the compiler
// parks the overflow in AGPRs (no scratch), one wavefront per SIMD -- measured on silicon
// (8 Wannier functions, 65^3 mesh): 1.83 ms, against 7.77 ms for the wavefront-per-matrix kernel.
//
// The kernel is also written for L = 2 or 4 neighbouring lanes sharing a matrix: every lane
// of the group keeps the whole A and performs the same rotations on it (identical arithmetic
// -> identical parameters, no communication) but owns only the rows o = sub, sub+L, ... of V
// (V <- V J touches each row on its own), and H(k) assembly is split over the group.  That
// trades the AGPR traffic for redundant A updates; it measured slower for n <= 8 (1.97 / 2.46
// ms), so those instantiations are only built with -DTBK_REG_MULTILANE (TBK_REG_LANES=2|4
// then selects them); the scheme is what would carry the register approach to n = 9, 10.
//
// H(k) assembly: when the model carries the R-grouped table (ModelView::nR > 0) one phase is
// formed per lattice vector and every slot gets one complex FMA with a wave-uniform
// coefficient; otherwise the slot-major term table is walked like in the other kernels.

template <int N, int NR>
struct RegMat {
    double dg[N];       // diagonal (real)
    cd up[N][N];        // strict upper triangle (p < q) used
    cd v[NR][N];        // v[i][b]: component o = sub + L*i of eigenvector b
};

template <int N, int NR, int P, int Q, bool VEC>
__device__ __forceinline__ void rotate_reg(RegMat<N, NR>& M) {
    const cd g = M.up[P][Q];
    const double g2 = cabs2(g);
    if (g2 > 0.0) {   // same division-free parameters as rotate<>
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
            cd x = r < P ? M.up[r][P] : cconj(M.up[P][r]);
            cd y = r < Q ? M.up[r][Q] : cconj(M.up[Q][r]);
            cd xn{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
            cd yn{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
            if (r < P) M.up[r][P] = xn; else M.up[P][r] = cconj(xn);
            if (r < Q) M.up[r][Q] = yn; else M.up[Q][r] = cconj(yn);
        }
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const cd x = M.v[i][P], y = M.v[i][Q];
                M.v[i][P] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
                M.v[i][Q] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
            }
        }
    }
}

template <int N, int NR, int P, int Q, bool VEC>
struct SweepReg {
    __device__ static __forceinline__ void run(RegMat<N, NR>& M) {
        rotate_reg<N, NR, P, Q, VEC>(M);
        if constexpr (Q + 1 < N)
            SweepReg<N, NR, P, Q + 1, VEC>::run(M);
        else if constexpr (P + 2 < N)
            SweepReg<N, NR, P + 1, P + 2, VEC>::run(M);
    }
};

// ---- S(k) from the R-grouped table, the table STREAMED THROUGH LDS by the wavefront in tiles of REG_TR lattice vectors (round 5).
// The table (silicon: 93 lattice vectors x 36 slots = 53 KB) used to arrive by scalar loads, a block of slots at a time because
// 36 complex numbers do not fit the scalar registers: ~10 dependent load -> wait -> use rounds per lattice vector, and at the one
// or two wavefronts per SIMD these kernels run at nothing hides them -- a wavefront spent ~80 % of its life waiting (0.9 ms per
// 65^3 for ~100 k cycles of arithmetic per wavefront).  Here tiles go from L2 straight into LDS (global_load_lds_dwordx4: no vector
// registers in between -- staged through registers the allocator parked them in scratch next to the 36 accumulators), two tiles
// ahead of the one being consumed as broadcast reads; the lattice vector itself comes the same way and goes into scalar registers
// (readfirstlane) so that the loops over |R_d| stay scalar-controlled.  vmcnt counts the transfers in order, so "tile t has landed"
// is "at most one tile's worth of transfers outstanding".  Same terms, same order, same operations as the scalar-load form: the
// same bits.  One wavefront per workgroup (no workgroup barrier; the LDS operations of a wavefront execute in order).
#define REG_TR 8
template <int N>
__host__ __device__ constexpr int reg_tile_nld() { return (REG_TR * (N * (N + 1) / 2) + 63) / 64; }        // transfers of 64 x 16 B per tile
template <int N>
__host__ __device__ constexpr int reg_tile_slots() { return (reg_tile_nld<N>() + 1) * 64; }               // + one row for the R's

template <int N>
__device__ __forceinline__ void reg_assemble_tiled(const ModelView& mv, const cd (&z)[4], cd* const tiles /* [2][reg_tile_slots] */,
                                                   const int lane, cd (&acc)[N * (N + 1) / 2]) {
    constexpr int NS = N * (N + 1) / 2;
    constexpr int NLD = reg_tile_nld<N>();
    constexpr int TS = reg_tile_slots<N>();
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int nR = mv.nR;
    const int ntile = (nR + REG_TR - 1) / REG_TR;
    const int64_t total = (int64_t)nR * NS;
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = cd{0.0, 0.0};
#define REG_ISSUE(T, BUF)                                                                                         \
    {                                                                                                             \
        const int64_t e0_ = (int64_t)(T) * REG_TR * NS;                                                           \
        _Pragma("unroll") for (int j = 0; j < NLD; ++j) {                                                         \
            const int64_t e_ = e0_ + j * 64 + lane;                                                               \
            __builtin_amdgcn_global_load_lds((const void*)(mv.rblock + (e_ < total ? e_ : total - 1)), (lds_ptr)((BUF) + j * 64), 16, 0, 0); \
        }                                                                                                         \
        const int r_ = (T) * REG_TR + (lane < REG_TR ? lane : REG_TR - 1);                                       \
        __builtin_amdgcn_global_load_lds((const void*)(mv.rvec + (r_ < nR ? r_ : nR - 1)), (lds_ptr)((BUF) + NLD * 64), 16, 0, 0);          \
    }
    REG_ISSUE(0, tiles)
    if (ntile > 1) REG_ISSUE(1, tiles + TS)
    for (int t = 0; t < ntile; ++t) {
        cd* const buf = tiles + (t & 1) * TS;
        if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int nhere = min(REG_TR, nR - t * REG_TR);
        for (int rr = 0; rr < nhere; ++rr) {
            const int4 Rv = *reinterpret_cast<const int4*>(buf + NLD * 64 + rr);
            const int4 Rs{__builtin_amdgcn_readfirstlane(Rv.x), __builtin_amdgcn_readfirstlane(Rv.y),
                          __builtin_amdgcn_readfirstlane(Rv.z), __builtin_amdgcn_readfirstlane(Rv.w)};
            const cd ph = phase_of_R(z, Rs);
            const cd* u = buf + rr * NS;
#pragma unroll
            for (int s = 0; s < NS; ++s) cfma_x(acc[s], u[s], ph);
        }
        // (every read of this buffer has been consumed by the multiply-adds above: it can take the tile after next)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (t + 2 < ntile) REG_ISSUE(t + 2, buf)
    }
}

// ---- the same sum on a MESH ROW (round 5): along the last mesh axis only z_last = exp(2 pi i k_last) changes, so
//     S_slot(k) = sum_p C_p[slot] z_last^p,     C_p[slot] = sum over the R with R_last = p of U_R[slot] exp(2 pi i k_lead . R_lead),
// the row's coefficient cells (what k_grid_rows does for n <= 4 from the per-slot term lists).  A wavefront's 64 consecutive points
// lie in at most three rows when the last axis holds >= 32 of them: the wavefront forms their cells TOGETHER -- lane = slot, the
// table streamed through LDS exactly as above, the (row, R) phases computed once by the lanes in parallel (lane = R) and parked in
// LDS -- 3 complex multiply-adds per lattice vector and lane instead of NS, and then every lane sums its point's 2 pmax + 1 cells per
// slot.  Silicon's 93 lattice vectors x 36 slots: ~2 k instead of 14.5 k vector instructions per wavefront (the kernel runs at one
// wavefront per SIMD -- the reflectors' LDS -- so its time follows its instruction count).  work: the reflector region of the
// wavefront, idle until the tridiagonalisation: [3][nR] phases, then [3][np][NS] cells.  Returns false (nothing touched but LDS)
// when the geometry does not fit; the caller then takes the tiled sum.  The terms of a point are added in an order that depends on
// the model and the point alone, and the choice between this form and the tiled one on the GLOBAL mesh, so windows cut anywhere and
// shards stay bit-identical (a narrow window takes more rounds of three rows); against the tiled sum and the k-list path the sums
// differ by rounding.
#define REG_CELLS_NPMAX 9
__device__ __forceinline__ double rowv_d(const double v, const int src) {   // the value of lane src (wave-uniform), in every lane
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_readlane(i.lo, src), __builtin_amdgcn_readlane(i.hi, src)};
    return __builtin_bit_cast(double, o);
}
// NRW: rows per round -- a tuning parameter only (a wavefront with more rows takes more rounds; the sums of a point are the same
// whatever NRW): 2 when the rows hold >= 64 points (never more than two rows per wavefront then), else 3.
template <int N, int NRW>
__device__ __forceinline__ bool reg_assemble_cells(const ModelView& mv, const GridArgs& G, const cd (&z)[4], const int64_t id, cd* const tiles,
                                                   cd* const work, const int work_entries, const int lane, cd (&acc)[N * (N + 1) / 2]) {
    constexpr int NS = N * (N + 1) / 2;
    constexpr int NLD = reg_tile_nld<N>();
    constexpr int TS = reg_tile_slots<N>();
    constexpr int NPM = REG_CELLS_NPMAX;
    static_assert(NS <= 64, "reg_assemble_cells: one lane per slot");
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int nR = mv.nR, last = G.last;
    const int np = 2 * mv.pmax + 1;
    const int mlast = G.wv.mesh[last];
    if (last != mv.dim_k - 1 || last > 2 || np > NPM || NRW * (nR + np * NS) > work_entries || G.wv.npts >= (int64_t)0x7fffffff)
        return false;                                    // (kernel-uniform)
    const int row = (int)((unsigned)id / (unsigned)mlast);
    const int row0 = __builtin_amdgcn_readfirstlane(row);
    const int rw = row - row0;                           // 0 .. : 64 consecutive points (ascending with the lane)
    const int nrows_all = __builtin_amdgcn_readlane(rw, 63) + 1;
    // z_last by 0 / 1 weights (a select on `last` turns z[] into an indexed array in scratch memory)
    const double w0 = last == 0 ? 1.0 : 0.0, w1 = last == 1 ? 1.0 : 0.0, w2 = last == 2 ? 1.0 : 0.0;
    const cd zl{fma(w2, z[2].x, fma(w1, z[1].x, w0 * z[0].x)), fma(w2, z[2].y, fma(w1, z[1].y, w0 * z[0].y))};
    const int ntile = (nR + REG_TR - 1) / REG_TR;
    const int64_t total = (int64_t)nR * NS;
    const int sl = lane < NS ? lane : NS - 1;
    cd* const cells = work + NRW * nR;                   // [NRW][np][NS]
    // three rows at a time (rows of >= 32 points: one round; a narrow WINDOW of such an array takes more rounds and the same sums,
    // so it stays bit-identical to the whole array)
    for (int g0 = 0; g0 < nrows_all; g0 += NRW) {
        const int nrows = min(NRW, nrows_all - g0);
        // exp(2 pi i k_d) of the leading axes, per row (the same bits in every lane of a row: its first lane speaks for it)
        cd zr[NRW][4];
#pragma unroll
        for (int w = 0; w < NRW; ++w) {
            const unsigned long long mw = __builtin_amdgcn_ballot_w64(rw == g0 + w);
            const int lf = mw != 0 ? (int)__builtin_ctzll(mw) : 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const cd zd = d < last ? z[d] : cd{1.0, 0.0};
                zr[w][d] = cd{rowv_d(zd.x, lf), rowv_d(zd.y, lf)};
            }
        }
        // phases of (row, R), lane = R
        for (int r = lane; r < nR; r += 64) {
            int4 R = mv.rvec[r];
            if (last == 0) R.x = 0; else if (last == 1) R.y = 0; else R.z = 0;
            R.w = 0;
#pragma unroll
            for (int w = 0; w < NRW; ++w)
                if (w < nrows) work[w * nR + r] = phase_of_R(zr[w], R);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // cells, lane = slot: the table through the same two LDS tiles
        cd c[NRW][NPM];
#pragma unroll
        for (int w = 0; w < NRW; ++w)
#pragma unroll
            for (int p = 0; p < NPM; ++p) c[w][p] = cd{0.0, 0.0};
        REG_ISSUE(0, tiles)
        if (ntile > 1) REG_ISSUE(1, tiles + TS)
        for (int t = 0; t < ntile; ++t) {
            cd* const buf = tiles + (t & 1) * TS;
            if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const int nhere = min(REG_TR, nR - t * REG_TR);
            // (the tile's LDS reads all at once: one exposed latency per tile instead of one per lattice vector -- this kernel runs at
            // one wavefront per SIMD)
            cd u[REG_TR], ph[REG_TR][NRW];
            int pl[REG_TR];
#pragma unroll
            for (int rr = 0; rr < REG_TR; ++rr) {
                const int rc = rr < nhere ? rr : 0;
                const int4 Rv = *reinterpret_cast<const int4*>(buf + NLD * 64 + rc);
                pl[rr] = rr < nhere ? __builtin_amdgcn_readfirstlane(last == 0 ? Rv.x : last == 1 ? Rv.y : Rv.z) + mv.pmax : -1;   // 0 .. np - 1
                u[rr] = buf[rc * NS + sl];
#pragma unroll
                for (int w = 0; w < NRW; ++w) ph[rr][w] = work[(w < nrows ? w : 0) * nR + t * REG_TR + rc];
            }
#pragma unroll
            for (int rr = 0; rr < REG_TR; ++rr) {
#pragma unroll
                for (int p = 0; p < NPM; ++p) {
                    if (p == pl[rr]) {                   // (scalar branch: the registers stay statically indexed)
#pragma unroll
                        for (int w = 0; w < NRW; ++w)
                            if (w < nrows) cfma_x(c[w][p], u[rr], ph[rr][w]);   // (scalar branch)
                    }
                }
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (t + 2 < ntile) REG_ISSUE(t + 2, buf)
        }
        if (lane < NS) {
#pragma unroll
            for (int w = 0; w < NRW; ++w)
#pragma unroll
                for (int p = 0; p < NPM; ++p)
                    if (p < np) cells[(w * np + p) * NS + lane] = c[w][p];
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // the point's sum: z_last^p for p = -pmax .. pmax
        if (rw >= g0 && rw < g0 + NRW) {                 // (every lane passes here in exactly one round)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = cd{0.0, 0.0};
            cd zc{1.0, 0.0};
            for (int q = 0; q < mv.pmax; ++q) zc = cmul(zc, cd{zl.x, -zl.y});
            const cd* mine = cells + (rw - g0) * np * NS;
            for (int p = 0; p < np; ++p) {
                const cd* cp = mine + p * NS;
#pragma unroll
                for (int s = 0; s < NS; ++s) cfma_x(acc[s], cp[s], zc);
                zc = cmul(zc, zl);
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();                 // (the next round overwrites phases and cells)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();                     // (the region goes back to the reflectors)
    return true;
}
#undef REG_ISSUE

// MODE 0: k list, 1: regular mesh into a wf_array (+ min gaps), 2: supplied matrices
template <int N, int L, int MODE, bool VEC>
__global__ __launch_bounds__(256) void k_solve_reg(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G) {
    constexpr int NR = VEC ? (N + L - 1) / L : 1;
    constexpr int NS = N * (N + 1) / 2;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int sub = (int)(tid & (L - 1));
    const int64_t id_raw = tid / L;
    const bool live = id_raw < nk;
    const int64_t id = live ? id_raw : nk - 1;   // idle tail lanes shadow the last point (shuffles stay defined)
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    RegMat<N, NR> M;
    if constexpr (MODE == 2) {
        const cd* h = Lst.ham + id * (int64_t)(N * N);
#pragma unroll
        for (int a = 0; a < N; ++a) {
            M.dg[a] = h[a * N + a].x;
#pragma unroll
            for (int b = a + 1; b < N; ++b) M.up[a][b] = h[a * N + b];
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
        if (mv.nR > 0) {
            // terms grouped by lattice vector: one phase per R, then NS complex FMAs whose
            // coefficients are wave-uniform (scalar loads) when L == 1
            cd acc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = cd{0.0, 0.0};
            // (the eigenvalue-only kernel runs three wavefronts per SIMD, which hide the scalar loads: the LDS tiles of
            // reg_assemble_tiled measured 0.154 against 0.144 ms per 48^3 of silicon here and are used by k_solve_regd only)
            for (int r = sub; r < mv.nR; r += L) {
                const cd ph = phase_of_R(z, mv.rvec[r]);
                const cd* u = mv.rblock + (size_t)r * NS;
#pragma unroll
                for (int s = 0; s < NS; ++s) cfma_x(acc[s], u[s], ph);      // (two fused operations per component; `cfma` as written is three)
            }
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    cd v = acc[slot];
                    if constexpr (L >= 2) {
                        v.x += __shfl_xor(v.x, 1);
                        v.y += __shfl_xor(v.y, 1);
                    }
                    if constexpr (L >= 4) {
                        v.x += __shfl_xor(v.x, 2);
                        v.y += __shfl_xor(v.y, 2);
                    }
                    if (b == a) M.dg[a] = v.x; else M.up[a][b] = v;
                }
            }
        } else {
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a) {
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
                    cd acc{0.0, 0.0};
                    for (int t = t0 + sub; t < t1; t += L) cfma(acc, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
                    if constexpr (L >= 2) {
                        acc.x += __shfl_xor(acc.x, 1);
                        acc.y += __shfl_xor(acc.y, 1);
                    }
                    if constexpr (L >= 4) {
                        acc.x += __shfl_xor(acc.x, 2);
                        acc.y += __shfl_xor(acc.y, 2);
                    }
                    if (b == a) M.dg[a] = acc.x; else M.up[a][b] = acc;
                }
            }
        }
    }
    if (VEC) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int b = 0; b < N; ++b) M.v[i][b] = cd{sub + L * i == b ? 1.0 : 0.0, 0.0};
    }
    if constexpr (!VEC && L == 1) {
        // Eigenvalues only (band structures, DOS: what a Wannier-interpolated model is mostly asked for): the DIRECT solver of the
        // n <= 4 kernels -- Householder on the upper triangle, then implicit QL on (d, e), one loop per deflation index
        // (tridiag_small / ql_deflate_small, tbk_solve.hip) -- instead of cyclic Jacobi on the complex matrix: ~3.5 k instead of
        // ~20 k instructions per 8 x 8 matrix (6 sweeps x 28 rotations x ~120).  Eigenvectors keep the Jacobi sweeps below: the real
        // Q of the direct solver would need N^2 more registers than a lane has.
        SmallFact<N> F;
        const bool ok = ql_small_core<N, false>(M.dg, M.up, F);
#pragma unroll
        for (int b = 0; b < N; ++b) M.dg[b] = F.d[b];
        if (!ok) {
            int* fl = MODE == 1 ? G.flags : Lst.flags;
            if (fl) fl[0] = 1;
        }
    } else {
    // The loop leaves on a WAVE-UNIFORM condition (no lane has off-diagonal weight left); a lane that is done idles behind its
    // EXEC bit.  After a per-lane `break` the compiler keeps every live-out value of the divergent loop twice -- the running one
    // and "the value of the lanes that have left": here the whole of A and V (tbk_solve.hip, ql_deflate_small, measured it:
    // 48 registers at n = 4).
    int sweep = 0;
    bool work = true;
    for (; sweep < TBK_JACOBI_MAX_SWEEPS; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int p = 0; p < N; ++p) {
            dia += M.dg[p] * M.dg[p];
#pragma unroll
            for (int q = p + 1; q < N; ++q) off += cabs2(M.up[p][q]);
        }
        work = !(off <= 1.0e-32 * (dia + off));
        if (__builtin_amdgcn_ballot_w64(work) == 0) break;
        if (work) SweepReg<N, NR, 0, 1, VEC>::run(M);
    }
    // (ran into the sweep cap: NaN input, or no convergence -- the reference's eigh raises there, pythtb.py:939,944)
    if (work) {
        int* fl = MODE == 1 ? G.flags : Lst.flags;
        if (fl) fl[0] = 1;
    }
    }
    int rk[N];
    double sorted[N];
    ranks_small<N>(M.dg, rk, sorted);
    if constexpr (MODE == 1) {
        unsigned long long* shard = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * N;
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) gap_min_wave(shard, b, live && sub == 0 ? sorted[b + 1] - sorted[b] : INFINITY);
    } else {
        if (live && sub == 0) {
#pragma unroll
            for (int b = 0; b < N; ++b) Lst.eval[(int64_t)b * nk + id] = sorted[b];
        }
    }
    if (VEC && live) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int o = sub + L * i;
            if (o >= N) continue;
            cd f{1.0, 0.0};
            if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[o])));
            if constexpr (MODE == 1) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    if (wrap[d]) f = cmul(f, G.pbc[d * N + o]);
            }
#pragma unroll
            for (int b = 0; b < N; ++b) {
                const cd val = cmul(M.v[i][b], f);
                if constexpr (MODE == 1) wf_at(G.wv, rk[b], id)[o] = val;
                else Lst.evec[((int64_t)rk[b] * nk + id) * N + o] = val;
            }
        }
    }
}

// ---- n = 5..8 WITH eigenvectors by the direct method (round 5): Householder on the upper triangle, implicit QL with the rotations
// accumulated in the REAL Q, then one band at a time Z[:, b] = H_0 .. H_{N-3} D Q[:, b] (the factored solver of the n <= 4 mesh kernels,
// tbk_solve.hip).  Cyclic Jacobi on the complex matrix with V -- the kernel above -- spends ~6 sweeps x 28 rotations x ~250
// instructions on an 8 x 8 matrix (41 k) and needs 256 VGPRs + AGPRs for A and V; this is ~8 k.  Q takes N^2 registers, so the
// normalised reflectors -- N (N - 1) / 2 - 1 complex numbers per matrix -- wait in LDS, [quantity][lane]: 27 KB per wavefront at
// n = 8, hence 64-thread workgroups (five of them share a compute unit).  One lane per matrix; bands go out to the plane of
// their rank like in the kernel above.
template <int N>
__host__ __device__ constexpr int regd_nu() { return N * (N - 1) / 2 - 1; }

template <int N, int MODE>
__global__ __launch_bounds__(64) void k_solve_regd(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G) {
    extern __shared__ __align__(16) unsigned char lds_regd[];
    cd* const Rf = reinterpret_cast<cd*>(lds_regd);           // [regd_nu<N>()][64], then two table tiles (reg_assemble_tiled)
    cd* const tiles = Rf + regd_nu<N>() * 64;
    constexpr int NS = N * (N + 1) / 2;
    const int lane = threadIdx.x;
    const int64_t id_raw = (int64_t)blockIdx.x * 64 + lane;
    const bool live = id_raw < nk;
    const int64_t id = live ? id_raw : nk - 1;                // idle tail lanes shadow the last point
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    double dg[N];
    cd up[N][N];
    if constexpr (MODE == 2) {
        const cd* h = Lst.ham + id * (int64_t)(N * N);
#pragma unroll
        for (int a = 0; a < N; ++a) {
            dg[a] = h[a * N + a].x;
#pragma unroll
            for (int b = a + 1; b < N; ++b) up[a][b] = h[a * N + b];
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
        if (mv.nR > 0) {                                      // R-grouped table: one phase per lattice vector, the table through LDS
            cd acc[NS];
            bool done = false;
            if constexpr (MODE == 1) {
                if (G.reg_cells) {
                    if (G.wv.mesh[G.last] >= 64) done = reg_assemble_cells<N, 2>(mv, G, z, id, tiles, Rf, regd_nu<N>() * 64, lane, acc);
                    else done = reg_assemble_cells<N, 3>(mv, G, z, id, tiles, Rf, regd_nu<N>() * 64, lane, acc);
                }
            }
            if (!done) reg_assemble_tiled<N>(mv, z, tiles, lane, acc);
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a)
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    if (b == a) dg[a] = acc[slot].x; else up[a][b] = acc[slot];
                }
        } else {
            int slot = 0;
#pragma unroll
            for (int a = 0; a < N; ++a)
#pragma unroll
                for (int b = a; b < N; ++b, ++slot) {
                    const cd s = slot_sum(mv, slot, z);
                    if (b == a) dg[a] = s.x; else up[a][b] = s;
                }
        }
    }
    SmallFact<N> F;
    double e[N];
    // reflector K, entry r > K  ->  row K (N - 1) - K (K - 1) / 2 + (r - K - 1) of this wavefront's LDS table
    tridiag_small_to<N, true>(dg, up, F, e, [&](const int K, const int r, const cd w) __attribute__((always_inline)) {
        Rf[(K * (N - 1) - K * (K - 1) / 2 + (r - K - 1)) * 64 + lane] = w;
    });
    asm volatile("" ::: "memory");
    const bool ok = ql_iterate_small<N, true>(F, e);
    if (!ok) {
        int* fl = MODE == 1 ? G.flags : Lst.flags;
        if (fl) fl[0] = 1;
    }
    int rk[N];
    double sorted[N];
    ranks_small<N>(F.d, rk, sorted);
    if constexpr (MODE == 1) {
        unsigned long long* shard = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * N;
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) gap_min_wave(shard, b, live ? sorted[b + 1] - sorted[b] : INFINITY);
    } else {
        if (live) {
#pragma unroll
            for (int b = 0; b < N; ++b) Lst.eval[(int64_t)b * nk + id] = sorted[b];
        }
    }
    // the orbital phases go into D once: Z = F H_0 .. H_{N-3} D Q needs them on every component of every band, and
    // F H = (F H F^+) F with F H F^+ = 1 - (F w)(F w)^+ ... but the reflectors sit in LDS; cheaper here: N products per band
    cd fo[N];
#pragma unroll
    for (int o = 0; o < N; ++o) {
        cd f{1.0, 0.0};
        if constexpr (MODE != 2) {
            if (mv.nspin == 2 && (o & 1)) f = fo[o - (o > 0)];           // both spin components of an orbital share its phase
            else f = cconj(expi2pi(kdot(kk, mv.orb[o])));
        }
        fo[o] = f;
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int o = 0; o < N; ++o)
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (wrap[d]) fo[o] = cmul(fo[o], G.pbc[d * N + o]);
    }
    asm volatile("" ::: "memory");
    if (live) {
        static_for<0, N>([&](auto bt) __attribute__((always_inline)) {
            constexpr int b = decltype(bt)::value;
            // (every band reads the reflectors from LDS again: left to itself the compiler reads them once and keeps all
            // N (N - 1) / 2 - 1 of them -- 108 registers at n = 8, through AGPR copies -- for the sake of 27 LDS reads per band)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            cd z[N];
#pragma unroll
            for (int r = 0; r < N; ++r) z[r] = cd{F.dph[r].x * F.Q[r][b], F.dph[r].y * F.Q[r][b]};
#pragma unroll
            for (int K = N - 3; K >= 0; --K) {
                const cd* uK = Rf + (K * (N - 1) - K * (K - 1) / 2 - (K + 1)) * 64 + lane;      // entry r at uK[r * 64]
                cd w{0.0, 0.0};
#pragma unroll
                for (int r = K + 1; r < N; ++r) {
                    const cd u = uK[r * 64];
                    w.x = fma(u.y, z[r].y, fma(u.x, z[r].x, w.x));
                    w.y = fma(-u.y, z[r].x, fma(u.x, z[r].y, w.y));
                }
#pragma unroll
                for (int r = K + 1; r < N; ++r) {
                    const cd u = uK[r * 64];
                    z[r].x = fma(u.y, w.y, fma(-u.x, w.x, z[r].x));
                    z[r].y = fma(-u.y, w.x, fma(-u.x, w.y, z[r].y));
                }
            }
#pragma unroll
            for (int o = 0; o < N; ++o) {
                const cd val = cmul_x(z[o], fo[o]);
                if constexpr (MODE == 1) wf_at(G.wv, rk[b], id)[o] = val;
                else Lst.evec[((int64_t)rk[b] * nk + id) * N + o] = val;
            }
        });
    }
}

// lanes per matrix when eigenvectors are wanted
static int reg_lanes(int) {
#ifdef TBK_REG_MULTILANE
    const int v = tbk_knobs().reg_lanes;
    if (v == 1 || v == 2 || v == 4) return v;
#endif
    return 1;
}

template <int N, int MODE, bool VEC>
static int launch_reg_n(tbk_ctx* ctx, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G) {
    const int lanes = VEC ? reg_lanes(N) : 1;
    const unsigned blocks = (unsigned)((nk * lanes + 255) / 256);
    if constexpr (VEC) {
        if (lanes == 1 && tbk_knobs().reg_direct != 0) {     // the direct method (TBK_REG_DIRECT=0: the Jacobi kernel, for A/B runs)
            const size_t lds = ((size_t)regd_nu<N>() * 64 + 2 * (size_t)reg_tile_slots<N>()) * sizeof(cd);
            hipLaunchKernelGGL((k_solve_regd<N, MODE>), dim3((unsigned)((nk + 63) / 64)), dim3(64), lds, ctx->stream, mv, nk, L, G);
            TBK_HIP(hipGetLastError());
            return TBK_OK;
        }
    }
    if constexpr (VEC) {
        switch (lanes) {
#ifdef TBK_REG_MULTILANE
            case 2: hipLaunchKernelGGL((k_solve_reg<N, 2, MODE, true>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
            case 4: hipLaunchKernelGGL((k_solve_reg<N, 4, MODE, true>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
#endif
            default: hipLaunchKernelGGL((k_solve_reg<N, 1, MODE, true>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G); break;
        }
    } else {
        hipLaunchKernelGGL((k_solve_reg<N, 1, MODE, false>), dim3(blocks), dim3(256), 0, ctx->stream, mv, nk, L, G);
    }
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

template <int MODE, bool VEC>
static int launch_reg(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L, const GridArgs& G) {
    TBK_REQUIRE(nk * 4 < (int64_t)0x7fffffff * 256, TBK_EUNSUPPORTED, "too many k-points for one launch");
    switch (n) {
        case 5: return launch_reg_n<5, MODE, VEC>(ctx, mv, nk, L, G);
        case 6: return launch_reg_n<6, MODE, VEC>(ctx, mv, nk, L, G);
        case 7: return launch_reg_n<7, MODE, VEC>(ctx, mv, nk, L, G);
        case 8: return launch_reg_n<8, MODE, VEC>(ctx, mv, nk, L, G);
        default: tbk_set_error("launch_reg: n=%d", n); return TBK_EINVAL;
    }
}
