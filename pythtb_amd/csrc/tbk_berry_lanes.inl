// tbk_berry_lanes.inl -- Wilson-loop eigenphases of 3 and 4 bands, the link stage (pythtb.py:3823-3838: per link
// `U, s, Vh = svd(M); P = P (U Vh)`).  Round 6, replacing the thread-per-segment kernel of round 4 (k_wilson_seg_reg, kept behind
// TBK_WILSON_REG=1) on arrays whose occupied vectors fit the tile below.
//
// What was wrong with a thread per segment (profiles/r05fcfg): every lane fetched its own points 16 bytes at a time, 96 bytes from
// its neighbour's -- 7.7 M L2 requests for 76 MB of vectors (a whole 128-byte line per 16 bytes used: the 36 KB a wavefront touches
// per link do not survive in a 32 KB L1), 77 % of a wavefront's life waiting at one wavefront per SIMD; and the segments of a string
// were then multiplied by ONE thread (257 threads on the chip, a serial chain of dependent loads: 167 us of the 281).
//
// Here a wavefront (one per workgroup: no barriers, the LDS operations of a wavefront execute in order) moves the occupied vectors
// of a TILE of 64 mesh points into LDS with global_load_lds_dwordx4 -- the 64 lanes of a transfer fetch 64 CONSECUTIVE 16-byte
// units of a band's plane, so every line is requested once -- and each lane then reads its own points from LDS:
//   * S form (strings across the lanes; the array's fastest axis is not the string axis): lane = string, the wavefront walks a
//     SEGMENT of links; two row buffers, the row of link i + 2 is requested as soon as the overlap matrix of link i has been formed
//     and lands behind the polar iteration; each lane keeps the ordered product of its segment.
//   * L form (the string along the lanes; the string axis is the fastest one, or there are too few strings to fill lanes): lane =
//     link of one 64-link tile of a string, 65 points in LDS; the 64 polar factors are multiplied by an ORDERED shuffle tree (the
//     left operand always from the lower lane): two levels per tile, the rest per string in the combine.
// k_wilson_lanes_combine multiplies a string's segment products the same way: a wavefront per string, a contiguous run of segments
// per lane, the same tree.  The polar iteration (Newton-Schulz, wilson_polar_reg) and the tail of the pipeline (Cayley transform,
// Hermitian eigen-solve, phases) are those of tbk_berry_big.inl.

struct WilsonLanesArgs {
    WilsonBigArgs W;
    int seg_len;       // S: links a lane multiplies in order; L: 64
    int nseg;          // matrices per string in `segs`: segments (S) or tiles (L)
    int64_t ntile;     // S: tiles of 64 strings (per segment)
    unsigned magic;    // ceil(65536 / ncomp): u / ncomp = (u * magic) >> 16 for the unit numbers of a tile
    int swz;           // >= 0 (ncomp = 4, 8, 16): component c of tile point j sits at unit j ncomp + (c ^ ((j >> swz) & (ncomp - 1))) --
                       // a lane reads its point with a stride of ncomp 16-byte units, 4- to 16-way bank conflicts of ds_read_b128 for
                       // these ncomp (k_wilson_lanes_s<4> on 8 components: 87 % of its LDS cycles, 41 % of a wavefront's life); with
                       // the exclusive-or the 16 lanes of a read group hit 16 different units.  The transfer fetches the permuted
                       // component (the same 16 ncomp bytes of the point: coalescing is untouched), LDS stays linear.  -1: none
    cd* segs;          // [ns][nseg][nocc^2]
    cd* prod;          // combine: string s at prod + s * pstride
    size_t pstride;
    int occ_inl[8];    // the occupied bands when W.occ is null (the determinant form: the list travels with the arguments)
    cd* dets;          // POLAR = false: [nseg][det_stride] products of the link determinants, what k_chain_final(_wave) reads
    int64_t det_stride;
    cd* herm;          // combine, non-null: [ns][nocc^2] Cayley transform of e^{-i alpha} (string product), Hermitian part (k_wilson_cayley's
    double ca, sa;     // output for the pipeline's FIRST angle alpha: cos, sin) -- one launch less in front of the eigen-solve
};

typedef __attribute__((address_space(3))) void* lanes_lds_ptr;
#define LANES_L_SPAN 4       // L form: links per matrix it leaves in `segs` (64 / LANES_L_SPAN matrices per tile; 4 bands of 8 components,
                             // 257 x 1024: kernel + combine 103 us at 16, 101 at 8, 97 at 4, 100 at 2)

// One band's components at the tile's points -> row[u], u = j * ncomp + c (j: point of the tile, c: component).  `pt(j)`: the mesh
// point of tile point j.  CONTIG: pt(j) = pt0 + j, the units are consecutive in memory and no division is needed.  A lane whose
// unit lies past the tile fetches the last unit again (every transfer complete: vmcnt counts them).
template <bool CONTIG, class PT>
__device__ __forceinline__ void lanes_issue_row(const WfsView& v, const int band, const int64_t pt0, PT pt, const int npt, cd* row,
                                                const int lane, const unsigned magic, const int swz) {
    const int ncomp = v.ncomp;
    const int nunit = npt * ncomp;
    const cd* const plane = v.data + (int64_t)band * v.npts * ncomp;
    for (int u0 = 0; u0 < nunit; u0 += 64) {
        const int u = min(u0 + lane, nunit - 1);
        const int j = (int)(((unsigned)u * magic) >> 16);
        const int us = swz >= 0 ? u ^ ((j >> swz) & (ncomp - 1)) : u;     // (ncomp a power of two: the exclusive-or stays inside the point)
        const cd* src;
        if constexpr (CONTIG) src = plane + pt0 * ncomp + us;
        else src = plane + pt(j) * ncomp + (us - j * ncomp);
        __builtin_amdgcn_global_load_lds((const void*)src, (lanes_lds_ptr)(row + u0), 16, 0, 0);
    }
}

template <int M>
__device__ __forceinline__ void lanes_identity(cd (&R)[M][M]) {
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) R[a][b] = cd{a == b ? 1.0 : 0.0, 0.0};
}
template <int M>
__device__ __forceinline__ void lanes_mul(cd (&R)[M][M], const cd (&X)[M][M]) {      // R <- R X
    cd T[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) {
            cd acc{0.0, 0.0};
#pragma unroll
            for (int k = 0; k < M; ++k) cfma(acc, R[a][k], X[k][b]);
            T[a][b] = acc;
        }
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) R[a][b] = T[a][b];
}
// X_ab = <u_a(p) | u_b(q)> from two row sets in LDS: band a of tile point j at rows[a * rowsz + j * ncomp + c]
template <int M>
__device__ __forceinline__ void lanes_overlap(const cd* P, const int jp, const cd* Q, const int jq, const int rowsz, const int ncomp,
                                              const int swz, cd (&X)[M][M]) {
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) X[a][b] = cd{0.0, 0.0};
    const cd* p = P + jp * ncomp;
    const cd* q = Q + jq * ncomp;
    const int fp = swz >= 0 ? (jp >> swz) & (ncomp - 1) : 0, fq = swz >= 0 ? (jq >> swz) & (ncomp - 1) : 0;
    for (int c = 0; c < ncomp; ++c) {
        cd pc[M], qc[M];
#pragma unroll
        for (int a = 0; a < M; ++a) {
            pc[a] = p[a * rowsz + (c ^ fp)];
            qc[a] = q[a * rowsz + (c ^ fq)];
        }
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = 0; b < M; ++b) cfmac(X[a][b], pc[a], qc[b]);
    }
}
// ordered product over the lanes: lane 0 ends with R_0 R_1 ... R_63 (the left operand always from the lower lane; lanes that hold
// nothing carry the identity)
template <int M, int SPAN = 64>
__device__ __forceinline__ void lanes_tree(cd (&R)[M][M]) {      // SPAN < 64: the lanes that are multiples of SPAN end with the product of their SPAN lanes
#pragma unroll
    for (int off = 1; off < SPAN; off <<= 1) {
        cd O[M][M];
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = 0; b < M; ++b) O[a][b] = cd{__shfl_down(R[a][b].x, off), __shfl_down(R[a][b].y, off)};
        lanes_mul<M>(R, O);        // (lanes that are not a multiple of 2 off form products nobody reads)
    }
}

// ---- S form: lane = string, a wavefront walks one segment of links for 64 neighbouring strings
// POLAR = false (round 6): the DETERMINANT form of berry_phase (pythtb.py:3829-3831: -angle(det of the product) = the product of the
// link determinants) for 1..4 bands of states with fewer than 8 components -- the same tiles, no polar iteration: a lane multiplies
// the determinants of its links and leaves ONE complex number per (segment, string) where k_chain_final(_wave) expects it.  The
// thread-per-(string, segment) kernel it replaces there (k_chain_partial) fetched its points 16 bytes per lane like round 4's
// Wilson kernel: 3 bands of 6 components, 257 x 1024 links: 52-74 us, 4 bands 80-124 (profiles/det_narrow_probe.py).
// OUT = 2 (berry_flux of the same bands): every link's determinant on its own, at dets[mesh point the link starts from] -- the
// array k_flux_from_dets combines into plaquette phases (the thread-per-plaquette kernel took 78 (1 band) to 327 us (4 bands) for
// the 263 k plaquettes of a 1025 x 257 array of 6-component states: profiles/flux_narrow_probe.py).
template <int M, int OUT = 0>
__global__ __launch_bounds__(64) void k_wilson_lanes_s(const WilsonLanesArgs S) {
    constexpr bool POLAR = OUT == 0;
    extern __shared__ __align__(16) unsigned char lds_lanes[];
    const WilsonBigArgs& A = S.W;
    const int lane = threadIdx.x;
    const int ncomp = A.v.ncomp;
    const int64_t seg = blockIdx.x / S.ntile, ts = blockIdx.x - seg * S.ntile;    // (neighbouring workgroups: neighbouring tiles of one segment)
    const int64_t s = ts * 64 + lane;
    const bool live = s < A.ns;
    const int npt = (int)min((int64_t)64, A.ns - ts * 64);
    const int i0 = (int)seg * S.seg_len, len = min(S.seg_len, A.nlinks - i0);    // wave-uniform
    const int64_t base = axis_offset(A.other, A.s0 + (live ? s : A.ns - 1)) + (int64_t)i0 * A.sdir;
    const int64_t base0 = (int64_t)__shfl((long long)base, 0);
    const bool contig = __builtin_amdgcn_ballot_w64(live && base != base0 + lane) == 0;
    const int rowsz = 64 * ncomp;
    cd* const buf = reinterpret_cast<cd*>(lds_lanes);            // [2][M][rowsz]
    int occ[M];
#pragma unroll
    for (int a = 0; a < M; ++a) occ[a] = A.occ ? A.occ[a] : S.occ_inl[a];
    auto issue = [&](const int li, const int b) __attribute__((always_inline)) {       // row of string point i0 + li -> buffer b
        const int64_t shift = (int64_t)li * A.sdir;
#pragma unroll
        for (int a = 0; a < M; ++a) {
            cd* row = buf + (b * M + a) * rowsz;
            if (contig) lanes_issue_row<true>(A.v, occ[a], base0 + shift, [](int) { return (int64_t)0; }, npt, row, lane, S.magic, S.swz);
            else lanes_issue_row<false>(A.v, occ[a], 0, [&](const int j) { return (int64_t)__shfl((long long)base, j) + shift; }, npt, row, lane, S.magic, S.swz);
        }
    };
    issue(0, 0);
    issue(1, 1);
    cd R[M][M];
    lanes_identity<M>(R);
    bool all_ok = true;
    for (int li = 0; li < len; ++li) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        cd X[M][M];
        lanes_overlap<M>(buf + (li & 1) * M * rowsz, lane, buf + ((li + 1) & 1) * M * rowsz, lane, rowsz, ncomp, S.swz, X);
        // the row of link li is not needed any more: its buffer takes the row of link li + 2, which lands behind the polar iteration
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (li + 2 <= len) issue(li + 2, li & 1);
        if (!live) lanes_identity<M>(X);           // (a lane past the last string reads repeated units: not a link matrix)
        if constexpr (POLAR) {
            all_ok = wilson_polar_reg<M>(X) && all_ok;
            lanes_mul<M>(R, X);
        } else if constexpr (OUT == 1) {
            R[0][0] = cmul(R[0][0], det_small<M>(X));      // (the running product of the link determinants sits in R[0][0])
        } else {
            if (live) S.dets[base + (int64_t)li * A.sdir] = det_small<M>(X);
        }
    }
    if (!live) return;
    if constexpr (OUT == 2) return;
    if constexpr (OUT == 1) {
        S.dets[seg * S.det_stride + A.s0 + s] = R[0][0];
        return;
    }
    if (!all_ok) atomicExch(A.flags + 1, 1);
    cd* const o = S.segs + ((size_t)s * S.nseg + seg) * (M * M);
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) o[a * M + b] = R[a][b];
}

// ---- L form: lane = link, a wavefront takes 64 consecutive links of one string
template <int M, int OUT = 0>
__global__ __launch_bounds__(64) void k_wilson_lanes_l(const WilsonLanesArgs S) {
    constexpr bool POLAR = OUT == 0;
    extern __shared__ __align__(16) unsigned char lds_lanes[];
    const WilsonBigArgs& A = S.W;
    const int lane = threadIdx.x;
    const int ncomp = A.v.ncomp;
    const int ntile = POLAR ? S.nseg / (64 / LANES_L_SPAN) : S.nseg;
    const int64_t s = blockIdx.x / ntile;
    const int t = (int)(blockIdx.x - s * ntile);
    const int i0 = t * 64, nl = min(64, A.nlinks - i0), npt = nl + 1;
    const int64_t base = axis_offset(A.other, A.s0 + s) + (int64_t)i0 * A.sdir;
    const int rowsz = (65 * ncomp + 63) & ~63;                   // (whole transfers: the repeated units of a row's last transfer stay inside it)
    cd* const buf = reinterpret_cast<cd*>(lds_lanes);            // [M][rowsz]
#pragma unroll
    for (int a = 0; a < M; ++a) {
        const int band = A.occ ? A.occ[a] : S.occ_inl[a];
        if (A.sdir == 1) lanes_issue_row<true>(A.v, band, base, [](int) { return (int64_t)0; }, npt, buf + a * rowsz, lane, S.magic, S.swz);
        else lanes_issue_row<false>(A.v, band, 0, [&](const int j) { return base + (int64_t)j * A.sdir; }, npt, buf + a * rowsz, lane, S.magic, S.swz);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    cd X[M][M];
    const bool act = lane < nl;
    lanes_overlap<M>(buf, act ? lane : 0, buf, act ? lane + 1 : 0, rowsz, ncomp, S.swz, X);
    if (!act) lanes_identity<M>(X);
    if constexpr (OUT == 2) {                      // every link's determinant at the mesh point it starts from
        if (act) S.dets[base + (int64_t)lane * A.sdir] = det_small<M>(X);
        return;
    }
    if constexpr (OUT == 1) {                      // determinant form: the product of the tile's link determinants (any order)
        cd dt = det_small<M>(X);
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) dt = cmul(dt, cd{__shfl_xor(dt.x, off), __shfl_xor(dt.y, off)});
        if (lane == 0) S.dets[(int64_t)t * S.det_stride + A.s0 + s] = dt;
        return;
    }
    if constexpr (!POLAR) return;                  // (nothing below is instantiated for use; kept out of the determinant form's code)
    if (!wilson_polar_reg<M>(X)) atomicExch(A.flags + 1, 1);
    // two levels of the tree here (the products of 4 links: lanes 0, 4, 8, ...), the rest with the string's other tiles in
    // k_wilson_lanes_combine -- a level costs the wavefront a whole matrix product however few lanes still need it, and the
    // combine pays its levels once per string, not once per tile
    lanes_tree<M, LANES_L_SPAN>(X);
    if ((lane & (LANES_L_SPAN - 1)) != 0) return;
    cd* const o = S.segs + ((size_t)s * S.nseg + (size_t)t * (64 / LANES_L_SPAN) + lane / LANES_L_SPAN) * (M * M);
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) o[a * M + b] = X[a][b];
}

// ---- the segments of a string in order: a wavefront per string, a contiguous run of segments per lane, the ordered tree
template <int M>
__global__ __launch_bounds__(64) void k_wilson_lanes_combine(const WilsonLanesArgs S) {
    const int64_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const int run = (S.nseg + 63) / 64;
    const int g0 = lane * run, g1 = min(g0 + run, S.nseg);
    const cd* const segs = S.segs + (size_t)s * S.nseg * (M * M);
    cd R[M][M];
    lanes_identity<M>(R);
    for (int g = g0; g < g1; ++g) {
        cd X[M][M];
        const cd* x = segs + (size_t)g * (M * M);
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = 0; b < M; ++b) X[a][b] = x[a * M + b];
        lanes_mul<M>(R, X);
    }
    lanes_tree<M>(R);
    if (lane != 0) return;
    cd* const o = S.prod + (size_t)s * S.pstride;
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) o[a * M + b] = R[a][b];
    if (S.herm == nullptr) return;
    // H = Hermitian part of i (I - Q) (I + Q)^-1, Q = e^{-i alpha} P (k_wilson_cayley's arithmetic, in the registers of the lane
    // that holds P): Gauss-Jordan on [I + Q | i (I - Q)] with partial pivoting, row exchanges as selects
    cd A[M][M], B[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) {
            const cd pv = R[a][b];
            const cd q{S.ca * pv.x + S.sa * pv.y, S.ca * pv.y - S.sa * pv.x};
            const double d = a == b ? 1.0 : 0.0;
            A[a][b] = cd{d + q.x, q.y};
            B[a][b] = cd{q.y, d - q.x};
        }
#pragma unroll
    for (int k = 0; k < M; ++k) {
        int r = k;
        double best = cabs2(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            const double v = cabs2(A[i][k]);
            if (v > best) {
                best = v;
                r = i;
            }
        }
#pragma unroll
        for (int i = k + 1; i < M; ++i) {
            const bool sw = r == i;
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const cd ta = A[k][j], tb = B[k][j];
                A[k][j] = cd{sw ? A[i][j].x : ta.x, sw ? A[i][j].y : ta.y};
                A[i][j] = cd{sw ? ta.x : A[i][j].x, sw ? ta.y : A[i][j].y};
                B[k][j] = cd{sw ? B[i][j].x : tb.x, sw ? B[i][j].y : tb.y};
                B[i][j] = cd{sw ? tb.x : B[i][j].x, sw ? tb.y : B[i][j].y};
            }
        }
        const cd piv = A[k][k];
        const double ip = 1.0 / fmax(cabs2(piv), 1e-300);
        const cd inv{piv.x * ip, -piv.y * ip};
#pragma unroll
        for (int j = 0; j < M; ++j) {
            if (j > k) A[k][j] = cmul(A[k][j], inv);
            B[k][j] = cmul(B[k][j], inv);
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
            if (i == k) continue;
            const cd f = A[i][k];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                if (j > k) {
                    const cd t = cmul(f, A[k][j]);
                    A[i][j] = cd{A[i][j].x - t.x, A[i][j].y - t.y};
                }
                const cd t = cmul(f, B[k][j]);
                B[i][j] = cd{B[i][j].x - t.x, B[i][j].y - t.y};
            }
        }
    }
    cd* const H = S.herm + (size_t)s * (M * M);
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) H[a * M + b] = cd{0.5 * (B[a][b].x + B[b][a].x), 0.5 * (B[a][b].y - B[b][a].y)};
}
