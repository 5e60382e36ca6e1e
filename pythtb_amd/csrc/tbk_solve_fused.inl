// tbk_solve_fused.inl -- included by tbk_solve.hip after k_grid_rows.
//
// solve_on_grid + berry_flux of a 2-D mesh in ONE pass (wf_array.solve_on_grid, pythtb.py:2421-2532, followed by
// berry_flux(occ), :3068-3205 / _one_flux_plane :3840-3865): the plaquette phases are formed from the eigenvectors while they
// are still in registers, so the array is written once and never read back.  The two-kernel step moves 16 n^2 B per point out
// and 16 nocc n B back in (config C: 268 + 134 MB; the flux kernel then waits on a memory system that is still writing the
// solve's output back, 0.45 of the HBM roofline at 4096^2); this kernel moves the 16 n^2 B only.
//
// Tile = R consecutive mesh rows x `seg` 64-point chunks, one wavefront.  Like k_grid_rows the lanes first build the row
// coefficient cells C_ab,p -- here of R + 1 rows: the row after the tile's last one is its HALO, solved again (the kernels are
// deterministic functions of the point, so the halo's vectors are the very bits its owner tile stores) and not stored.
// Chunks outside, rows inside: at (chunk, row) a lane solves its point, stages and stores it exactly like k_grid_rows, and then
//   Uy(row; j-1 -> j) = det <u(row, j-1) | u(row, j)>      left neighbour by a one-lane wavefront shift (DPP wave_shr:1);
//                                                           lane 0 takes lane 63 of the previous chunk from an LDS carry
//   Ux(row-1, j)       = det <u(row-1, j) | u(row, j)>      the previous row's vectors stay in registers
//   F(row-1, j-1)      = -arg[ Ux(row-1, j-1) Uy(row; j-1 -> j) conj Ux(row-1, j) conj Uy(row-1; j-1 -> j) ]
// (the reference's loop (i,j) -> (i+1,j) -> (i+1,j+1) -> (i,j+1) -> (i,j), pythtb.py:3855-3861, as link determinants; the lane
// on column j owns the plaquette to its LEFT).  Per-lane sums in a fixed order, a wave tree, one partial per tile.
// The plaquette column between two tiles of a row group (1 in 64 seg) is left to k_flux_seams, which reads its four corners
// from the finished array and also adds everything up in a fixed order: bit-reproducible totals, no float atomics.
// (Tried: the seam plaquettes inside this launch -- wavefronts of seam x 32 rows, one point per lane solved again, links by
// lane exchange.  Per step at 2048^2 / 4096^2: at the head of the launch 59.3 / 270 us, at its tail 58.4 / 264, the separate
// kernel 59.3 / 252-258, no seams at all 56.8 / 255: a seam wavefront is a chain of table loads in front of one eigen-solve,
// 1072 (8.5 k at 4096^2) of them cost what the second launch costs.)
//
// Extra arithmetic: (R + 1) / R eigen-solves per point and the link / phase arithmetic of k_flux_rows -- on a kernel whose
// VALU was 37 % busy (n = 2) behind its stores.

struct FusedArgs {
    int R;              // rows per tile
    int nrg;            // row groups = ceil(mesh[0] / R)
    int occ[2];
    double* partial;    // [ntiles + nseam_blocks]
};

// value of lane - 1; lane 0 receives `first` (DPP wave_shr:1 leaves lane 0's destination untouched)
__device__ __forceinline__ double fused_shr1(const double v, const double first) {
    const I2 i = __builtin_bit_cast(I2, v), f = __builtin_bit_cast(I2, first);
    const I2 o{__builtin_amdgcn_update_dpp(f.lo, i.lo, 0x138, 0xf, 0xf, false),
               __builtin_amdgcn_update_dpp(f.hi, i.hi, 0x138, 0xf, 0xf, false)};
    return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ cd fused_shr1(const cd v, const cd first) { return cd{fused_shr1(v.x, first.x), fused_shr1(v.y, first.y)}; }

// det <p_a | q_b>, a, b < NOCC (conjugate on the first, sum over the N components: _wf_dpr, pythtb.py:3793-3796)
template <int N, int NOCC>
__device__ __forceinline__ cd fused_link(const cd (&p)[NOCC][N], const cd (&q)[NOCC][N]) {
    cd M[NOCC][NOCC];
#pragma unroll
    for (int a = 0; a < NOCC; ++a)
#pragma unroll
        for (int b = 0; b < NOCC; ++b) {
            cd acc{0.0, 0.0};
#pragma unroll
            for (int o = 0; o < N; ++o) cfmac(acc, p[a][o], q[b][o]);
            M[a][b] = acc;
        }
    if constexpr (NOCC == 1) return M[0][0];
    else return csub(cmul(M[0][0], M[1][1]), cmul(M[0][1], M[1][0]));
}


template <int N, int PM, int NOCC>
// (90 VGPRs at N = 2, five wavefronts per SIMD -- with MachineLICM off for this file (Makefile).  With it the compiler parked the
// 20 coefficients of atan2's rarely taken branch in 40 registers for the whole kernel: 154 VGPRs, three wavefronts; forcing four
// with __launch_bounds__(256, 4) kept the constants and spilled 31 live values to scratch -- and a scratch access is a
// vector-memory operation that waits for every store issued before it: never in this loop)
__global__ __launch_bounds__(256) void k_grid_rows_flux(const ModelView mv, const GridArgs G, const FusedArgs F) {
    extern __shared__ __align__(16) unsigned char lds_rows[];
    static_assert(PM >= 0, "k_grid_rows_flux: static range of the last lattice component");
    constexpr int NSLOT = N * (N + 1) / 2;
    constexpr int npow = 2 * PM + 1;
    constexpr int ncell = NSLOT * npow;
    constexpr int NCAR = NOCC * N + 1;              // carried per row: the occupied vectors and Ux of lane 63
    // bands staged at once.  (Tried NB = 2 at N = 2 -- one LDS round trip per row instead of two, 2 KB more LDS per wavefront:
    // 2048^2 58.6 against 58.4 us per step, 4096^2 265 against 265; on a slower box 66.4 against 64.4 and 257 against 270)
    constexpr int NB = 1;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int R1 = F.R + 1;
    const int per_wave = R1 * (ncell + N + NCAR) + 64 * N * NB + (G.seg > 1 ? G.seg * 64 * (1 + N) : 0);
    cd* const base = reinterpret_cast<cd*>(lds_rows) + wib * per_wave;
    cd* const C = base;                             // [R1][ncell]
    cd* const frowL = C + R1 * ncell;               // [R1][N]
    cd* const carry = frowL + R1 * N;               // [R1][NCAR]
    cd* const stage = carry + R1 * NCAR;            // [NB][64 N]
    // the tile's per-column table entries z_last(j), f_last(j, o): fetched ONCE, so that the main loop holds no vector-memory
    // load at all -- vmcnt counts loads and stores in issue order, and any load consumed inside the loop would wait for every
    // store issued before it (k_grid_rows keeps its prefetch legal with a static store count; here the row count is data)
    cd* const ztab = stage + 64 * N * NB;           // [seg][64]
    cd* const ftab = ztab + G.seg * 64;             // [seg][64][N]
    const int64_t tile = (int64_t)blockIdx.x * 4 + wib;
    const bool live = tile < G.ntiles;
    const int nlast = G.wv.mesh[1], M0 = G.wv.mesh[0];
    int r0 = 0, jc0 = 0, jc1 = 0, nrows = 0;
    if (live) {
        const int rg = (int)(tile / G.tpr);
        const int ts = (int)(tile - (int64_t)rg * G.tpr);
        r0 = rg * F.R;
        nrows = min(R1, M0 - r0);                   // rows solved by this tile (the last one is the halo when it exists)
        jc0 = ts * G.seg;
        jc1 = min(jc0 + G.seg, G.cpr);
        for (int cell = lane; cell < nrows * ncell; cell += 64) {
            const int rr = cell / ncell, c = cell - rr * ncell;
            const cd z[4] = {G.tz[0][r0 + rr], cd{1.0, 0.0}, cd{1.0, 0.0}, cd{1.0, 0.0}};
            const int t0 = mv.cell_ptr[c], t1 = mv.cell_ptr[c + 1];
            cd acc{0.0, 0.0};
            for (int t = t0; t < t1; ++t) {
                int4 Rv = mv.term_R[t];
                Rv.y = 0;
                cfma(acc, mv.term_amp[t], phase_of_R(z, Rv));
            }
            C[cell] = acc;
        }
        for (int e = lane; e < nrows * N; e += 64) {
            const int rr = e / N, o = e - rr * N;
            frowL[e] = G.tf[0][(int64_t)(r0 + rr) * N + o];
        }
        if (G.seg > 1) {
            for (int e = lane; e < (jc1 - jc0) * 64; e += 64) {
                const int jj = min(jc0 * 64 + e, nlast - 1);
                ztab[e] = G.tz[1][jj];
#pragma unroll
                for (int o = 0; o < N; ++o) ftab[e * N + o] = G.tf[1][(int64_t)jj * N + o];
            }
        }
    }
    // one chunk per tile (seg = 1): its table entries go straight to registers -- fetched here, before any store is issued, and
    // consumed here (the empty asm), so that no wait on them lands inside the row loop
    cd zl{1.0, 0.0}, tfl[N];
#pragma unroll
    for (int o = 0; o < N; ++o) tfl[o] = cd{0.0, 0.0};
    if (live && G.seg == 1) {
        const int jj = min(jc0 * 64 + lane, nlast - 1);
        zl = G.tz[1][jj];
#pragma unroll
        for (int o = 0; o < N; ++o) tfl[o] = G.tf[1][(int64_t)jj * N + o];
        asm volatile("" : "+v"(zl.x), "+v"(zl.y));
#pragma unroll
        for (int o = 0; o < N; ++o) asm volatile("" : "+v"(tfl[o].x), "+v"(tfl[o].y));
    }
    __syncthreads();
    if (!live) return;

    double gmin[N > 1 ? N - 1 : 1];
#pragma unroll
    for (int b = 0; b + 1 < N; ++b) gmin[b] = __longlong_as_double(0x7ff0000000000000ll);
    auto slot = [](const int s) { return (N & 1) ? s : (s ^ ((s >> 3) & 3)); };   // (see k_grid_rows)
    int wslot[N], rslot[N];
#pragma unroll
    for (int o = 0; o < N; ++o) {
        wslot[o] = slot(lane * N + o);
        rslot[o] = slot(o * 64 + lane);
    }
    double wsel0[N], wsel1[N];
#pragma unroll
    for (int b = 0; b < N; ++b) {
        wsel0[b] = F.occ[0] == b ? 1.0 : 0.0;
        wsel1[b] = F.occ[NOCC > 1 ? 1 : 0] == b ? 1.0 : 0.0;
    }
    double psum = 0.0;
    for (int jc = jc0; jc < jc1; ++jc) {
        if (G.seg > 1) {
            zl = ztab[(jc - jc0) * 64 + lane];
#pragma unroll
            for (int o = 0; o < N; ++o) tfl[o] = ftab[((jc - jc0) * 64 + lane) * N + o];
        }
        const int j = jc * 64 + lane;
        const int nvalid_pts = min(64, nlast - jc * 64);
        const bool full = nvalid_pts == 64;
        // the plaquette to the left of column j exists for 1 <= j <= nlast - 1; the first column of the tile has no left
        // neighbour inside the tile (k_flux_seams)
        const bool col_ok = j >= 1 && j < nlast && (lane > 0 || jc > jc0);
        cd vprev[NOCC][N], uy_prev{1.0, 0.0};
#pragma unroll
        for (int a = 0; a < NOCC; ++a)
#pragma unroll
            for (int o = 0; o < N; ++o) vprev[a][o] = cd{0.0, 0.0};
        for (int rr = 0; rr < nrows; ++rr) {
            const cd* Crow = C + rr * ncell;
            const bool stored = rr < F.R;              // (the halo row belongs to the next row group)
            cd psi[NOCC][N];
            if constexpr (N > 2) {
                // n = 3, 4: the factored solver of k_grid_rows' chunk_fact (the same assembly, hence the same eigenvalues bit for
                // bit): (d, Q) sorted, then one band at a time formed, phased, picked for the links and staged / stored
                SmallFact<N> Fc;
                {
                    double dg[N];
                    cd up[N][N];
                    rows_assemble<N, PM>(Crow, npow, PM, zl, dg, up);
                    if (!ql_small_core<N, true>(dg, up, Fc) && G.flags) G.flags[0] = 1;
                }
                sort_fact<N>(Fc);
                if (stored) {
#pragma unroll
                    for (int b = 0; b + 1 < N; ++b) gmin[b] = fmin(gmin[b], Fc.d[b + 1] - Fc.d[b]);
                }
                cd fo[N];
#pragma unroll
                for (int o = 0; o < N; ++o) fo[o] = cmul_x(frowL[rr * N + o], tfl[o]);
#pragma unroll
                for (int a = 0; a < NOCC; ++a)
#pragma unroll
                    for (int o = 0; o < N; ++o) psi[a][o] = cd{0.0, 0.0};
                bool do_store = stored;
#ifdef TBK_DIAG
                if (G.ablate == 1 || G.ablate == 4) do_store = do_store && Fc.d[0] == 1.2345e300;
#endif
                const int nvalid = nvalid_pts * N;
                const int64_t point0 = (int64_t)(r0 + rr) * nlast + (int64_t)jc * 64;
                static_for<0, N>([&](auto rt) __attribute__((always_inline)) {
                    constexpr int r = decltype(rt)::value;
                    cd val[N];
                    small_vector<N, r>(Fc, val);
#pragma unroll
                    for (int o = 0; o < N; ++o) {
                        val[o] = cmul_x(val[o], fo[o]);
                        // (0 / 1 weights pick the occupied bands: see the note in the n <= 2 branch)
                        psi[0][o].x = fma(wsel0[r], val[o].x, psi[0][o].x);
                        psi[0][o].y = fma(wsel0[r], val[o].y, psi[0][o].y);
                        if constexpr (NOCC > 1) {
                            psi[1][o].x = fma(wsel1[r], val[o].x, psi[1][o].x);
                            psi[1][o].y = fma(wsel1[r], val[o].y, psi[1][o].y);
                        }
                    }
                    if (do_store) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int o = 0; o < N; ++o) stage[wslot[o]] = val[o];
                        asm volatile("" ::: "memory");
                        cd* dst = G.wv.data + ((int64_t)r * G.wv.npts + point0) * N;
#pragma unroll
                        for (int i = 0; i < N; ++i) {
                            const int e = i * 64 + lane;
                            if (full || e < nvalid) dst[e] = stage[rslot[i]];
                        }
                    }
                });
            } else {
                SmallMat<N> M;
                int sl = 0;
#pragma unroll
                for (int a = 0; a < N; ++a) {
#pragma unroll
                    for (int b = a; b < N; ++b, ++sl) {
                        const cd* Cs = Crow + sl * npow + PM;
                        cd acc = Cs[0];
                        cd zp = zl;
#pragma unroll
                        for (int p = 1; p <= PM; ++p) {
                            cfma(acc, Cs[p], zp);
                            cfma(acc, Cs[-p], cconj(zp));
                            if (p < PM) zp = cmul(zp, zl);
                        }
                        if (b == a) M.dg[a] = acc.x; else M.up[a][b] = acc;
                    }
                }
                init_vectors<N, true>(M);
                if constexpr (N > 2) {
                    if (!jacobi_small<N, true>(M) && G.flags) G.flags[0] = 1;
                    sort_small<N>(M);
                } else {
                    jacobi_small<N, true>(M);
                }
                if (stored) {
#pragma unroll
                    for (int b = 0; b + 1 < N; ++b) gmin[b] = fmin(gmin[b], M.dg[b + 1] - M.dg[b]);
                }
                cd fo[N];
#pragma unroll
                for (int o = 0; o < N; ++o) fo[o] = cmul(frowL[rr * N + o], tfl[o]);
                // the stored components of the occupied states: what berry_flux would read back
#pragma unroll
                for (int o = 0; o < N; ++o) {
                    // (the band is picked with 0 / 1 weights: written as "M.v[o][occ]", or as a chain of selects on occ == b, the
                    // compiler parks the candidates in SCRATCH memory and indexes them -- and a scratch load is a vector-memory
                    // operation, so every row of every chunk then waited for all of its own stores to land: 2.4 x slower)
                    cd c0{0.0, 0.0}, c1{0.0, 0.0};
#pragma unroll
                    for (int b = 0; b < N; ++b) {
                        c0.x = fma(wsel0[b], M.v[o][b].x, c0.x);
                        c0.y = fma(wsel0[b], M.v[o][b].y, c0.y);
                        if constexpr (NOCC > 1) {
                            c1.x = fma(wsel1[b], M.v[o][b].x, c1.x);
                            c1.y = fma(wsel1[b], M.v[o][b].y, c1.y);
                        }
                    }
                    psi[0][o] = cmul(c0, fo[o]);
                    if constexpr (NOCC > 1) psi[1][o] = cmul(c1, fo[o]);
                }
                bool do_store = stored;
#ifdef TBK_DIAG
                // diagnostic build (TBK_ABLATE_GRID): 1 = no global stores, 3 = no links / phases, 4 = neither
                if (G.ablate == 1 || G.ablate == 4) do_store = do_store && M.dg[0] == 1.2345e300;
#endif
                if (do_store) {
                    const int nvalid = nvalid_pts * N;
                    const int64_t point0 = (int64_t)(r0 + rr) * nlast + (int64_t)jc * 64;
                    // NB bands go through the staging tile at once
#pragma unroll
                    for (int rb = 0; rb < N; rb += NB) {
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int r = rb; r < rb + NB; ++r)
#pragma unroll
                            for (int o = 0; o < N; ++o) stage[(r - rb) * 64 * N + wslot[o]] = cmul(M.v[o][r], fo[o]);
                        asm volatile("" ::: "memory");
                        cd out[NB][N];
#pragma unroll
                        for (int r = 0; r < NB; ++r)
#pragma unroll
                            for (int i = 0; i < N; ++i) out[r][i] = stage[r * 64 * N + rslot[i]];
#pragma unroll
                        for (int r = 0; r < NB; ++r) {
                            cd* dst = G.wv.data + ((int64_t)(rb + r) * G.wv.npts + point0) * N;
#pragma unroll
                            for (int i = 0; i < N; ++i) {
                                const int e = i * 64 + lane;
                                if (full || e < nvalid) dst[e] = out[r][i];
                            }
                        }
                    }
                }
            }
#ifdef TBK_DIAG
            if (G.ablate >= 3) {
                psum += psi[0][0].x;
                continue;
            }
#endif
            // ---- links and the plaquette row between rr - 1 and rr
            const cd* car = carry + rr * NCAR;
            cd left[NOCC][N];
#pragma unroll
            for (int a = 0; a < NOCC; ++a)
#pragma unroll
                for (int o = 0; o < N; ++o) left[a][o] = fused_shr1(psi[a][o], car[a * N + o]);
            const cd uy = fused_link<N, NOCC>(left, psi);
            cd ux{1.0, 0.0};
            if (rr > 0) {
                ux = fused_link<N, NOCC>(vprev, psi);
                const cd uxl = fused_shr1(ux, car[NOCC * N]);
                const cd z = cmul(cmul(uxl, uy), cconj(cmul(ux, uy_prev)));
                const double pha = -arg_small_first(z.y, z.x);
                psum += col_ok ? pha : 0.0;
            }
            if (lane == 63) {
                cd* cw = carry + rr * NCAR;
#pragma unroll
                for (int a = 0; a < NOCC; ++a)
#pragma unroll
                    for (int o = 0; o < N; ++o) cw[a * N + o] = psi[a][o];
                cw[NOCC * N] = ux;
            }
#pragma unroll
            for (int a = 0; a < NOCC; ++a)
#pragma unroll
                for (int o = 0; o < N; ++o) vprev[a][o] = psi[a][o];
            uy_prev = uy;
        }
    }
    if constexpr (N > 1) {
#pragma unroll
        for (int b = 0; b + 1 < N; ++b) {
            double g = gmin[b];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) g = fmin(g, __shfl_xor(g, off));
            if (lane == 0) G.gap_part[tile * (N - 1) + b] = g;
        }
    }
    // the tile's flux: a fixed-shape tree over the lanes' ordered sums
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) psum += __shfl_xor(psum, off);
    if (lane == 0) F.partial[tile] = psum;
}

// ---- the plaquette columns between two tiles of a row group, and the total.  Thread = (plaquette row i, seam s), column
// j = seam * seg * 64 - 1: reads its four corners from the finished array (band-major planes) -- 1 / (64 seg) of the plaquettes.
// Block b also adds slice b of the tiles' partials; the block that arrives last (two-level ticket, as in k_flux_rows) adds the
// blocks' sums in a fixed order: the total is bit-reproducible and there is no third launch (k_sum_fixed as a kernel of its own
// was 4.8 us bracketed, ~3.7 us of a back-to-back step).  The last block reads gridDim.x values (128 at 2048^2), one per
// thread -- handing it all 5.8 k tile partials instead was the 14 us of the first attempt.
template <int N, int NOCC>
__global__ __launch_bounds__(256) void k_flux_seams_sum(const WfsView v, const int occ0, const int occ1, const int seg, const int nseam,
                                                        const double* __restrict__ tile_partial, const int64_t ntiles,
                                                        double* blk_partial, unsigned* counters, double* __restrict__ total) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int nlast = v.mesh[1];
    const int64_t nthreads = (int64_t)(v.mesh[0] - 1) * nseam;
    double pha = 0.0;
    if (t < nthreads) {
        const int i = (int)(t / nseam), s = (int)(t - (int64_t)i * nseam);
        const int j = (s + 1) * seg * 64 - 1;                  // plaquette between columns j and j + 1
        if (j + 1 < nlast) {
            auto load = [&](const int ii, const int jj, cd (&u)[NOCC][N]) {
                const int64_t pt = (int64_t)ii * nlast + jj;
#pragma unroll
                for (int o = 0; o < N; ++o) {
                    u[0][o] = wf_at(v, occ0, pt)[o];
                    if constexpr (NOCC > 1) u[1][o] = wf_at(v, occ1, pt)[o];
                }
            };
            cd a[NOCC][N], b[NOCC][N], c[NOCC][N], d[NOCC][N];
            load(i, j, a);
            load(i + 1, j, b);
            load(i + 1, j + 1, c);
            load(i, j + 1, d);
            const cd ux0 = fused_link<N, NOCC>(a, b), uy1 = fused_link<N, NOCC>(b, c);
            const cd ux1 = fused_link<N, NOCC>(d, c), uy0 = fused_link<N, NOCC>(a, d);
            const cd z = cmul(cmul(ux0, uy1), cconj(cmul(ux1, uy0)));
            pha = -arg_small_first(z.y, z.x);
        }
    }
    // slice of the tiles' partials (written by the launch before this one)
    const int64_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * per, i1 = min(i0 + per, ntiles);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) pha += tile_partial[i];
    // block sum in a fixed order
    __shared__ double red[256];
    __shared__ int last_flag;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pha += __shfl_xor(pha, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pha;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mine = (red[0] + red[1]) + (red[2] + red[3]);
        const unsigned nb = gridDim.x, blk = blockIdx.x;
        int last = 0;
        // hand-off without cache-wide fences: one 8-byte agent-scope store, drained before the ticket
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(blk_partial) + blk, (unsigned long long)__double_as_longlong(mine),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (nb > 1u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned shard = blk & 7u;
            const unsigned shard_size = (nb - shard + 7u) / 8u;
            if (atomicAdd(counters + shard, 1u) == shard_size - 1u) {
                const unsigned nshards = min(nb, 8u);
                if (atomicAdd(counters + 8, 1u) == nshards - 1u) last = 1;
            }
        } else {
            *total = mine;
        }
        last_flag = last;
    }
    __syncthreads();
    if (last_flag) {
        const unsigned long long* pb = reinterpret_cast<const unsigned long long*>(blk_partial);
        auto ld = [&](const unsigned idx) {
            return __longlong_as_double((long long)__hip_atomic_load(pb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        };
        const unsigned nb = gridDim.x;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        unsigned i = threadIdx.x;
        for (; i + 3u * 256u < nb; i += 4u * 256u) {
            s0 += ld(i);
            s1 += ld(i + 256u);
            s2 += ld(i + 512u);
            s3 += ld(i + 768u);
        }
        for (; i < nb; i += 256u) s0 += ld(i);
        __syncthreads();
        red[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
#pragma unroll
        for (int w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) *total = red[0];
        if (threadIdx.x < 9) counters[threadIdx.x] = 0u;       // re-arm for the next launch
    }
}

// fixed-shape sum of n partials into *total (one block; the same order every run)
__global__ __launch_bounds__(1024) void k_sum_fixed(const double* __restrict__ p, const int64_t n, double* __restrict__ total) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        s0 += p[i];
        s1 += p[i + 1024];
        s2 += p[i + 2048];
        s3 += p[i + 3072];
    }
    for (; i < n; i += 1024) s0 += p[i];
    double s = (s0 + s1) + (s2 + s3);
    __shared__ double red[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        *total = tot;
    }
}
