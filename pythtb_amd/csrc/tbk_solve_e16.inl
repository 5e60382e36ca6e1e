// tbk_solve_e16.inl -- n = 9..16 states per k WITH eigenvectors, chip-filling batches (config E: cubic16 on 257^3 points), in
// ONE kernel with nothing but the eigenvectors leaving the compute unit (round 4).  Included by tbk_solve_e16.hip (the
// product's translation unit) and by profiles/microbench/e16_bench.hip.
//
// The direct solver numpy.linalg.eigh runs for the reference (pythtb.py:939-947) -- Householder tridiagonalisation, eigenvalues
// of the real tridiagonal T, eigenvectors of T, back-transformation -- on 16 lanes per matrix, four matrices per wavefront, like
// round 3's three kernels (tbk_solve_tw16.inl), but with the stage that forced the cut redone so that it fits the same lanes:
//
//   round 3:  k_tw16_tridiag (16 lanes / matrix)  ->  k_tw16_eigvals (implicit QL, ONE lane per matrix: 64 matrices per
//             wavefront, a 200 us dependent chain)  ->  k_tw16_vectors (16 lanes / matrix).  The reflectors (2.2 KB per matrix)
//             and (d, e), the eigenvalues and the ranks went through HBM in between: 9.5 KB of traffic per matrix for 4 KB of
//             eigenvectors (2.39 x).  QL cannot come into the 16-lane kernels: it is one sequential chain per matrix.
//   here:     lane j of a matrix computes eigenvalue j itself -- (a) Sturm counts of T - x (LDL^T pivots with IEEE infinities, the
//             hardware reciprocal estimate: a count is exact for a matrix within 1e-8 |T| of T, which is all isolation needs):
//             one 16-point multisection shared by the lanes of a matrix, then bisection; (b) Newton's iteration on the
//             characteristic polynomial of the UNREDUCED BLOCK of T that owns the eigenvalue (three-term recurrence with its
//             derivative; the bracket kept by the same recurrence's sign changes), quadratic from a bracket that isolates the
//             root; (c) the Rayleigh-quotient correction the twisted factorisation gives for free.  ~0.5 k wave-instructions per
//             matrix, what the lane-per-matrix QL cost, with all 64 lanes busy and no second kernel: the reflectors stay in
//             the wavefront's LDS region from the reflection that makes them to the back-transformation that applies them.
//
// HBM traffic: 16 n^2 bytes per matrix written, the model's tables read (L2) -- 1.0 x algorithmic.  No workspace.
//
// Matrices this cannot serve -- two eigenvalues of one unreduced block closer than gaptol |T| (the Newton-Schulz step no longer
// reaches rounding level), a Newton iteration that has not converged, a twisted-factorisation residual that fails -- are put on
// a list and solved again by the QL-replay kernels of tbk_solve_ql16.inl (LIST mode), exactly as round 3's path did; the
// decision depends on the matrix alone, so periodic images, halo rows and shard windows stay bit-identical.

#define E16_REC 143                       // 16-byte entries of a matrix's reflector record: 119 elements of w_K packed by column,
                                          // 16 phases of the diagonal unitary, 8 spare (round 4 kept beta_K there); the stride keeps
                                          // the four broadcasts of a read on different banks
#define E16_XCH 160                       // 16-byte entries of the wavefront's exchange region (2560 B): the staged U_R, (d, e) of T
                                          // (64), Sturm counts (32), (g, d g) of the scaled recurrence (64), one matrix's V for the
                                          // matrix cores (144)
#define E16_WAVE_LDS ((4 * E16_REC + E16_XCH) * 16)   // 11 712 B per wavefront: 3 wavefronts per SIMD fit 160 KB
__host__ __device__ constexpr int e16_off(const int K) { return 15 * K - K * (K - 1) / 2; }   // first element of u_K

// LDS through pointers that carry their address space (the loads and stores are ds_read / ds_write whatever the compiler can
// prove about the pointer) and plain vector element types (struct copies across address spaces do not compile on the host pass)
typedef double e16_d2 __attribute__((ext_vector_type(2)));
typedef double e16_d4 __attribute__((ext_vector_type(4)));
typedef unsigned e16_u2 __attribute__((ext_vector_type(2)));
typedef unsigned e16_u4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) e16_d2 e16_lcd;     // one complex number / one (d, e) pair: 16 bytes
typedef __attribute__((address_space(3))) double e16_ld;
typedef __attribute__((address_space(3))) e16_u2 e16_lu2;
typedef __attribute__((address_space(3))) e16_u4 e16_lu4;
__device__ __forceinline__ cd e16_get(const e16_lcd* p) {
    const e16_d2 t = *p;
    return cd{t.x, t.y};
}
__device__ __forceinline__ void e16_put(e16_lcd* p, const cd v) { *p = e16_d2{v.x, v.y}; }
// the LDS regions are private to a wavefront, whose LDS operations execute in order: no barriers, no waits; only the
// compiler has to keep the order (the views of a region differ in type)
#define E16_ORDER() asm volatile("" ::: "memory")

// Complex multiply-add and the phase e^{2 pi i k.R} with every rounding spelled out: H(k) is accumulated at two places of the
// kernel (by the lanes together for the part that is constant along a mesh row, lane by lane otherwise) and both must give the
// same bits -- the shared helpers' `a * b - c * d` leaves the choice of which product is fused to the compiler, site by site.
__device__ __forceinline__ void e16_cfma(cd& acc, const cd a, const cd b) {
    acc.x = fma(-a.y, b.y, fma(a.x, b.x, acc.x));
    acc.y = fma(a.y, b.x, fma(a.x, b.y, acc.y));
}
__device__ __forceinline__ cd e16_phase_of_R(const cd (&z)[4], const int4 R) {
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
        for (int q = 0; q < m; ++q) e = cd{fma(e.x, zz.x, -(e.y * zz.y)), fma(e.x, zz.y, e.y * zz.x)};
    }
    return e;
}

// ---------------------------------------------------------------- 1. Householder tridiagonalisation, reflectors into LDS
// acc += (value of `src` in lane C of this 16-lane row) * mul, one instruction: the fp64 ALU takes a DPP row broadcast on its
// first operand (v_fmac_f64_dpp ... row_newbcast:C; the only DPP control the 64-bit ALU has).  u and q of a reflection reach
// the other rows of the matrix this way -- no LDS round trip (round 3: a 16-byte ds_write per lane and (15 - K) broadcast
// ds_read_b128 twice per reflection, 357 LDS reads per wavefront and two exposed LDS latencies per step) and no separate
// broadcast moves (round 2: four v_mov_b32_dpp per complex number).  Inline assembly because the compiler forms the 64-bit
// broadcast move (v_mov_b64_dpp) but does not fold it into the multiply-add.
template <int C>
__device__ __forceinline__ void e16_fmac_bc(double& acc, const double src, const double mul) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "i"(C));
}
template <int C>
__device__ __forceinline__ void e16_fmac_nbc(double& acc, const double src, const double mul) {   // acc -= bcast(src) * mul
    asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "i"(C));
}
// value of lane SRC of each 16-lane row: ONE v_mov_b64_dpp (the builtin is typed for doubles in the device pass only)
template <int SRC>
__device__ __forceinline__ double e16_bcast(const double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + SRC, 0xf, 0xf, true);
#else
    return v;
#endif
}
// the value of lane `src` (wave-uniform index) in every lane
__device__ __forceinline__ double e16_lane_d(const double v, const int src) {
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_readlane(i.lo, src), __builtin_amdgcn_readlane(i.hi, src)};
    return __builtin_bit_cast(double, o);
}
#ifndef E16_CELLS_MIN_NR
#define E16_CELLS_MIN_NR 16               // models with more lattice vectors than this take the row's coefficient cells on a mesh (cubic16, 7 vectors: 1-2 % slower with them; 27: even; 125: 2 x faster)
#endif
// (a VGPR written by the vector ALU may be read through DPP two wait states later at the earliest; the compiler does not see
// into the assembly above, so the values it broadcasts pass through here once)
__device__ __forceinline__ void e16_dpp_ready(cd& v) { asm volatile("s_nop 1" : "+v"(v.x), "+v"(v.y)); }

// p += sum over the columns C > K of A[x][C] u_C.  (One assembly statement per column: between separate statements the compiler
// pads the dependent accumulators with s_nop -- 677 of them per wavefront, one issue cycle each.)
#define E16_BC " row_newbcast:%[c] row_mask:0xf bank_mask:0xf\n\t"
template <int K, int C>
__device__ __forceinline__ void e16_pass1(const cd (&a)[16], const cd u, cd& p) {
    asm("v_fmac_f64_dpp %[px], %[ux], %[ax]" E16_BC
        "v_fmac_f64_dpp %[py], %[uy], %[ax]" E16_BC
        "v_fmac_f64_dpp %[px], -%[uy], %[ay]" E16_BC
        "v_fmac_f64_dpp %[py], %[ux], %[ay]" E16_BC
        : [px] "+v"(p.x), [py] "+v"(p.y)
        : [ux] "v"(u.x), [uy] "v"(u.y), [ax] "v"(a[C].x), [ay] "v"(a[C].y), [c] "i"(C));
    if constexpr (C + 1 < 16) e16_pass1<K, C + 1>(a, u, p);
}
// A[x][C] -= u_x conj(q_C) + q_x conj(u_C) for the columns C > K
template <int K, int C>
__device__ __forceinline__ void e16_pass2(cd (&a)[16], const cd u, const cd q) {
    asm("v_fmac_f64_dpp %[ax], -%[qx], %[ux]" E16_BC
        "v_fmac_f64_dpp %[ay], -%[qx], %[uy]" E16_BC
        "v_fmac_f64_dpp %[ax], -%[qy], %[uy]" E16_BC
        "v_fmac_f64_dpp %[ay], %[qy], %[ux]" E16_BC
        "v_fmac_f64_dpp %[ax], -%[ux], %[qx]" E16_BC
        "v_fmac_f64_dpp %[ay], -%[ux], %[qy]" E16_BC
        "v_fmac_f64_dpp %[ax], -%[uy], %[qy]" E16_BC
        "v_fmac_f64_dpp %[ay], %[uy], %[qx]" E16_BC
        : [ax] "+v"(a[C].x), [ay] "+v"(a[C].y)
        : [ux] "v"(u.x), [uy] "v"(u.y), [qx] "v"(q.x), [qy] "v"(q.y), [c] "i"(C));
    if constexpr (C + 1 < 16) e16_pass2<K, C + 1>(a, u, q);
}
#undef E16_BC
// Step K on the rows of A (lane x = row x).  mag = |T[K+1][K]|, unit = T[K+1][K] / |T[K+1][K]| (1 when it vanishes): with a
// reflection they are |x| and -alpha / |alpha|, both at hand -- no second square root for the phase fix.  rec: this matrix's
// record, which keeps the NORMALISED reflector w_K = u_K sqrt(beta_K) (H_K = 1 - w w^+: no beta in the products here or in the
// back-transformation).
// Rows <= K are finished (row x is last read at step x - 1, its diagonal entry is never a column > K): nothing below masks them
// out of the products -- what the passes leave in their columns > K is never read -- only out of the two sums over the rows.
template <int K, bool VEC = true>
__device__ __forceinline__ void e16_house(cd (&a)[16], const int x, e16_lcd* rec, double& mag, cd& unit) {
    const cd ak = a[K];
    // (decided on the entries below the subdiagonal alone, like LAPACK's zlarfg: see ql16_house)
    const double rest = row_allsum(x > K + 1 ? cabs2(ak) : 0.0);
    const cd alpha = cd{e16_bcast<K + 1>(ak.x), e16_bcast<K + 1>(ak.y)};   // A[K+1][K]
    const double absa2 = cabs2(alpha);
    e16_lcd* const ru = rec + (e16_off(K) - (K + 1));
    double absa = absa2;                                 // (0 -- or a NaN, which must reach the coupling and not turn into a split of T)
    cd ph{1.0, 0.0};
    if (absa2 > 0.0) {
        const double inv_a = rsqrt_full(absa2);
        absa = absa2 * inv_a;
        ph = cd{alpha.x * inv_a, alpha.y * inv_a};
    }
    mag = absa;
    unit = ph;
    if (rest > 0.0) {                                    // row-uniform
        const double sigma = rest + absa2;
        const double inv_n = rsqrt_full(sigma), nrm = sigma * inv_n;
        const double sb = rsqrt_full(nrm * (nrm + absa));                   // sqrt(beta), beta = 2 / (u^+ u)
        cd w{ak.x * sb, ak.y * sb};
        if (x == K + 1) {
            const double t = (absa + nrm) * sb;
            w = cd{ph.x * t, ph.y * t};
        }
        mag = nrm;                                       // T[K+1][K] = -ph |x|
        unit = cd{-ph.x, -ph.y};
        if constexpr (VEC) {
            if (x > K) e16_put(ru + x, w);
        }
        e16_dpp_ready(w);
        cd p{0.0, 0.0};
        e16_pass1<K, K + 1>(a, w, p);
        const double ws = fma(w.x, p.x, w.y * p.y);
        const double kappa = 0.5 * row_allsum(x > K ? ws : 0.0);
        cd q{fma(-kappa, w.x, p.x), fma(-kappa, w.y, p.y)};
        e16_dpp_ready(q);
        e16_pass2<K, K + 1>(a, w, q);
    } else {                                             // nothing to reflect: H_K = I
        if constexpr (VEC) {
            if (x > K) e16_put(ru + x, cd{0.0, 0.0});
        }
    }
}

// ---------------------------------------------------------------- 2. eigenvalue j of T on lane j
// Number of eigenvalues of the (scaled) T below x, and which pivots were negative (bit i: q_i < 0).  q_i = (d_i - x) -
// e_{i-1}^2 / q_{i-1} with IEEE arithmetic: a zero pivot gives an infinite next one and the recurrence recovers, the count
// being one of the two one-sided limits (which is all a bracket needs); e2 holds 1e-300 instead of 0 at the splits of T so
// that 0 x inf never appears.  v_rcp_f64 without refinement: the count is exact for a matrix within ~1e-8 |T| of T.
template <int I>
__device__ __forceinline__ void e16_count_step(const double (&d)[16], const double (&e2)[16], const int n, const double x, double& q,
                                               unsigned& acc) {
    {
        const double r = __builtin_amdgcn_rcp(q);
        q = I == 0 ? d[0] - x : fma(-e2[I > 0 ? I - 1 : 0], r, d[I] - x);
        acc = __builtin_amdgcn_alignbit(acc, __double2hiint(q), 31);   // (acc << 1) | sign(q)
    }
    if constexpr (I + 1 < 16) e16_count_step<I + 1>(d, e2, n, x, q, acc);
}
// The same count with the reciprocal refined to full precision (dstebz's recurrence: backward stable) -- the eigenvalue-only form's
// last resort, a plain bisection to rounding level for a lane whose Newton iteration did not settle (e16_eigenvalue, RESCUE).
template <int I>
__device__ __forceinline__ void e16_count_step_exact(const double (&d)[16], const double (&e2)[16], const double x, double& q, unsigned& acc) {
    {
        const double r0 = __builtin_amdgcn_rcp(q);       // (a vanishing pivot: infinite, and it stays that way -- the refinement of an
        double r = fma(fma(-q, r0, 1.0), r0, r0);        // infinity is a NaN, and so is that of the zero after it; unrefined both behave)
        r = fma(fma(-q, r, 1.0), r, r);
        r = fabs(r0) < INFINITY && fabs(q) < INFINITY ? r : r0;
        q = I == 0 ? d[0] - x : fma(-e2[I > 0 ? I - 1 : 0], r, d[I] - x);
        acc += (unsigned)__double2hiint(q) >> 31;
    }
    if constexpr (I + 1 < 16) e16_count_step_exact<I + 1>(d, e2, x, q, acc);
}
// -> bit (15 - i) of the result: pivot i negative
__device__ __forceinline__ unsigned e16_signs(const double (&d)[16], const double (&e2)[16], const int n, const double x) {
    double q = 1.0;
    unsigned acc = 0;
    e16_count_step<0>(d, e2, n, x, q, acc);
    return acc & 0xffffu;
}

// The same bits from the three-term recurrence p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2} for a T that does not split, in the
// SCALED form s_i = p_i / m_i with m_i = e_{i-1}^2 m_{i-2} > 0 (m_{-1} = m_0 = 1): the coefficient of s_{i-2} becomes exactly one,
//     s_i = (d_i - x) g_i s_{i-1} - s_{i-2},   g_i = m_{i-1} / m_i,
// two multiply-adds per position on the tables g_i and d_i g_i (e16_eigenvalue forms them once per matrix, lane i its own entry)
// where the plain recurrence takes three full-rate instructions and the pivots a quarter-rate reciprocal; the signs of s_i are those
// of p_i (bit: p_i and p_{i-1} differ in sign; collected raw, the differences taken once at the end).  With every e^2 > 0 a
// vanishing p_i has neighbours of opposite signs and counts once, as it should; |T| <= 1 and m_i >= 1e-250 here (a smaller m_i
// sends the wavefront to the pivots, like a split T: after a zero p_i of one block the rest would vanish), so |s_i| <= 3^16 1e250.
// Rows n.. (the padding of a matrix smaller than 16) are left out: the branch is scalar.
template <int I>
__device__ __forceinline__ void e16_count_scaled_step(const double (&g)[16], const double (&dg)[16], const int n, const double x, double& s1,
                                                      double& s2, unsigned& acc) {
    const double s = fma(fma(-x, g[I], dg[I]), s1, -s2);
    acc = __builtin_amdgcn_alignbit(acc, (unsigned)__double2hiint(s), 31);
    s2 = s1;
    s1 = s;
    if constexpr (I + 1 < 16) {
        if (I + 1 < 2 || I + 1 < n) e16_count_scaled_step<I + 1>(g, dg, n, x, s1, s2, acc);   // (nested: one way out, no copies per position)
    }
}
__device__ __forceinline__ unsigned e16_signs_scaled(const double (&g)[16], const double (&dg)[16], const int n, const double x) {
    double s1 = 1.0, s2 = 0.0;
    unsigned acc = 0;
    e16_count_scaled_step<0>(g, dg, n, x, s1, s2, acc);
    const unsigned raw = acc & ((1u << n) - 1u);         // sign of p_0 in bit n - 1 .. of p_{n-1} in bit 0; p_{-1} = 1 is positive
    return ((raw ^ (raw >> 1)) << (16 - n)) & 0xffffu;
}
// ... and with the derivative, for the Newton steps on an unsplit T: s'_i = (d_i - x) g_i s'_{i-1} - g_i s_{i-1} - s'_{i-2}
// (p / p' = s / s': the scale cancels).  sgn collects the signs of s_0 .. s_{n-1}, newest in bit 0.
template <int I>
__device__ __forceinline__ void e16_newton_scaled_step(const double (&g)[16], const double (&dg)[16], const int n, const double x, double& s1,
                                                       double& s2, double& ds1, double& ds2, unsigned& sgn) {
    const double tg = fma(-x, g[I], dg[I]);
    const double s = fma(tg, s1, -s2);
    const double ds = fma(-g[I], s1, fma(tg, ds1, -ds2));
    sgn = __builtin_amdgcn_alignbit(sgn, (unsigned)__double2hiint(s), 31);
    s2 = s1;
    ds2 = ds1;
    s1 = s;
    ds1 = ds;
    if constexpr (I + 1 < 16) {
        if (I + 1 < 2 || I + 1 < n) e16_newton_scaled_step<I + 1>(g, dg, n, x, s1, s2, ds1, ds2, sgn);
    }
}

// One evaluation of the characteristic polynomial of the block [bl, bh] of T at x with its derivative and the number of sign
// changes of the sequence (= eigenvalues of the block below x): p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}.
// BLOCK = false: the block is all of T's n positions (no per-lane predicate).  sgn collects the signs of p_bl .. p_bh (newest in
// bit 0); the sign changes are counted from it afterwards.
template <int I, bool BLOCK>
__device__ __forceinline__ void e16_poly_step(const double (&d)[16], const double (&e2)[16], const int n, const int bl, const int bh,
                                              const double x, double& p1, double& p2, double& dp1, double& dp2, unsigned& sgn, unsigned& len) {
    if (!BLOCK || (I >= bl && I <= bh)) {
        const double t = d[I] - x;
        const double ee = I > 0 ? ((!BLOCK || I > bl) ? e2[I > 0 ? I - 1 : 0] : 0.0) : 0.0;
        const double p = fma(t, p1, -ee * p2);
        const double dp = fma(t, dp1, -ee * dp2) - p1;
        sgn = __builtin_amdgcn_alignbit(sgn, (unsigned)__double2hiint(p), 31);
        if constexpr (BLOCK) ++len;
        p2 = p1;
        dp2 = dp1;
        p1 = p;
        dp1 = dp;
    }
    if constexpr (I + 1 < 16) e16_poly_step<I + 1, BLOCK>(d, e2, n, bl, bh, x, p1, p2, dp1, dp2, sgn, len);
}

// value of the NEXT lane of the 16-lane row (DPP row_shl:1 -- data moves towards lane 0; lane 15 reads 0) and of the PREVIOUS one
// (row_shr:1; lane 0 reads 0)
__device__ __forceinline__ double e16_next(const double v) {
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_update_dpp(0, i.lo, 0x101, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, i.hi, 0x101, 0xf, 0xf, true)};
    return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ double e16_prev(const double v) {
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_update_dpp(0, i.lo, 0x111, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, i.hi, 0x111, 0xf, 0xf, true)};
    return __builtin_bit_cast(double, o);
}

__device__ __forceinline__ double e16_rcp(const double p) {
    double y = __builtin_amdgcn_rcp(p);
    y = fma(fma(-p, y, 1.0), y, y);
    y = fma(fma(-p, y, 1.0), y, y);
    return y;
}
// value of lane x - N of the 16-lane row (row_shr:N); the lanes 0 .. N-1, which have no such neighbour, read 1.0
template <int N>
__device__ __forceinline__ double e16_shr_one(const double v) {
    const I2 i = __builtin_bit_cast(I2, v);
    const I2 o{__builtin_amdgcn_update_dpp(0, i.lo, 0x110 + N, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(0x3ff00000, i.hi, 0x110 + N, 0xf, 0xf, false)};
    return __builtin_bit_cast(double, o);
}

#ifdef E16_DEBUG
__device__ double e16_dbg[64 * 16];
#define E16_DBG(slotv, j, k, val) do { if ((slotv) == E16_DEBUG) e16_dbg[(j) * 16 + (k)] = (double)(val); } while (0)
#else
#define E16_DBG(slotv, j, k, val) do { } while (0)
#endif

// E16_SKIP (profiles/microbench/e16_bench.hip only; the results are wrong on purpose, only the timing is read): a bit mask of
// phases to leave out -- 1 Householder steps, 2 eigenvalue, 4 twisted factorisation, 8 Newton-Schulz, 16 back-transformation,
// 32 the eigenvector stores
#ifndef E16_SKIP
#define E16_SKIP 0
#endif
#ifdef E16_MARKS   // (ISA inspection only: phase boundaries visible in the assembly listing)
#define E16_MARK(n) asm volatile("s_nop 0 ; E16_MARK " #n ::: "memory")
#else
#define E16_MARK(n) do { } while (0)
#endif

#ifndef E16_NBISECT
#define E16_NBISECT 8
#endif
#ifdef E16_ITER_HIST
__device__ unsigned e16_iter_hist[32];
#endif
#ifndef E16_NEWTON_MAX
#define E16_NEWTON_MAX 12
#endif
#ifndef E16_NEWTON_TOL
#define E16_NEWTON_TOL 5.820766091346741e-11   // 2^-34
#endif
#ifndef E16_NS_TRIPLE
#define E16_NS_TRIPLE 1e-2                     // three eigenvalues within this much of |T|, or two within E16_NS_PAIR |T|: the full
#define E16_NS_PAIR 1e-3                       // Newton-Schulz step (stage 4)
#endif

// d, e: T of this lane's matrix (replicated over its 16 lanes); on return scaled by `scale` (a power of two: exact) with the
// negligible couplings zeroed, lam = eigenvalue j of the scaled T, [bl, bh] its unreduced block.  flag: this lane could not do
// its part (no convergence).  n: real states (rows n.. of T are decoupled padding and take no part).
template <bool RESCUE = false>
__device__ __forceinline__ void e16_eigenvalue(const double dd, const double ee_in, double (&d)[16], double (&e)[16], const int n, const int j,
                                               e16_lcd* xd /* [16] of this matrix */, double& scale, double& lam, int& bl, int& bh, unsigned& split,
                                               bool& flag, const int64_t dbg_slot = -1) {
    // ---- splits, scale and Gershgorin bounds, each lane for its own position j of T (then row reductions), the scaled
    // (d_j, e_j) to every lane of the matrix through LDS.  (Every lane doing all 16 positions was 465 instructions.)
    const int lane = threadIdx.x & 63;
    const double dnext = e16_next(dd);                   // d_{j+1} (0 in lane 15)
    const bool ng = !(fabs(ee_in) > 2.220446049250313e-16 * (fabs(dd) + fabs(dnext))) || j >= n - 1 || j >= 15;
    split = (unsigned)(__builtin_amdgcn_ballot_w64(ng) >> (lane & 48)) & 0xffffu;   // bit i: e_i negligible in T (bit 15 always)
    const double ee = ng ? 0.0 : ee_in;
    const double rad = fabs(e16_prev(ee)) + fabs(ee);    // |e_{j-1}| + |e_j|
    double tn = j < n ? fabs(dd) + rad : 0.0;
    tn = fmax(tn, row_ror_d<8>(tn));
    tn = fmax(tn, row_ror_d<4>(tn));
    tn = fmax(tn, row_ror_d<2>(tn));
    tn = fmax(tn, row_ror_d<1>(tn));
    {
        int ex = 0;
        (void)frexp(tn, &ex);                            // tn = m 2^ex, m in [1/2, 1)
        scale = tn > 0.0 && tn < INFINITY ? ldexp(1.0, -ex) : 1.0;
    }
    // (the padding rows of a matrix smaller than 16 sit at 2, above the scaled spectrum: decoupled, they never add to a
    // count below 2 and the recurrences run over all 16 positions without asking)
    const double ds = j < n ? dd * scale : 2.0, es = ee * scale;
    double gl = j < n ? ds - rad * scale : INFINITY, gu = j < n ? ds + rad * scale : -INFINITY;
    gl = fmin(gl, row_ror_d<8>(gl));
    gu = fmax(gu, row_ror_d<8>(gu));
    gl = fmin(gl, row_ror_d<4>(gl));
    gu = fmax(gu, row_ror_d<4>(gu));
    gl = fmin(gl, row_ror_d<2>(gl));
    gu = fmax(gu, row_ror_d<2>(gu));
    gl = fmin(gl, row_ror_d<1>(gl));
    gu = fmax(gu, row_ror_d<1>(gu));
    // (a hair wider than Gershgorin's discs, so that the counts at the ends are 0 and n whatever the rounding)
    gl -= 1e-13;
    gu += 1e-13;
    double e2[16];
    // tables of the scaled recurrence (e16_signs_scaled): m_j = product of the e_{k-1}^2 over k = j, j - 2, .. >= 1 by a stride-2
    // scan along the row, g_j = m_{j-1} / m_j; lane j forms entry j
    const double ep = e16_prev(es);                      // scaled e_{j-1} (0 in lane 0)
    double mj = j >= 1 && j < n ? ep * ep : 1.0;
    mj *= e16_shr_one<2>(mj);
    mj *= e16_shr_one<4>(mj);
    mj *= e16_shr_one<8>(mj);
    const double gj = e16_shr_one<1>(mj) * e16_rcp(mj);
    // does THIS T split, or scale out of range?  No: the scaled recurrence; yes: the pivots.  (Row-uniform, not wave-uniform as in
    // round 4: a matrix's arithmetic must not depend on what shares its wavefront -- the same k-point in another list, window or
    // chunk has to come out with the same bits.  A wavefront with both kinds runs both forms behind EXEC masks; that is rare.)
    const unsigned long long sp_b = __builtin_amdgcn_ballot_w64((split & 0x7fffu & ((1u << (n - 1)) - 1u)) != 0 || (!(mj >= 1e-250) && j < n));
    const bool splits = ((unsigned)(sp_b >> (lane & 48)) & 0xffffu) != 0;
    e16_lcd* const xg = xd + 96;                         // (g, d g): past the four (d, e) blocks and the counts
    E16_ORDER();
    xd[j] = e16_d2{ds, es};
    xg[j] = e16_d2{gj, ds * gj};
    E16_ORDER();
    // (the scaled e_i themselves stay in LDS until the eigenvalue is known: 32 registers the isolation and the Newton steps do not
    // need -- they work on (d, e^2), or on (g, d g) held in the same registers -- and that were being spilled to scratch here)
    if (splits) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const e16_d2 t = xd[i];
            d[i] = t.x;
            e2[i] = fmax(t.y * t.y, 1e-300);             // (1e-300 at the splits: 0 x inf never appears among the pivots)
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const e16_d2 t = xg[i];
            d[i] = t.x;                                  // g_i
            e2[i] = t.y;                                 // d_i g_i
        }
    }
    E16_ORDER();
    e16_lu2* const xch = reinterpret_cast<e16_lu2*>(xd - 16 * ((threadIdx.x & 63) >> 4) + 64) + 16 * ((threadIdx.x & 63) >> 4);   // counts: past the four (d, e) blocks

    E16_MARK(21);
    // ---- one multisection shared by the 16 lanes: lane j looks at point j of 16 inside (gl, gu)
    const double w = (gu - gl) * (1.0 / 17.0);
    const double tj = fma(w, (double)(j + 1), gl);
    unsigned sj;
    if (splits) sj = e16_signs(d, e2, n, tj);
    else sj = e16_signs_scaled(d, e2, n, tj);
    xch[j] = e16_u2{(unsigned)__builtin_popcount(sj), sj};
    E16_ORDER();
    double lo = gl, hi = gu;
    unsigned clo = 0, chi = (unsigned)n, slo = 0, shi = n >= 16 ? 0xffffu : (((1u << n) - 1u) << (16 - n));   // (pivot i negative: bit 15 - i)
    {
        // m = points whose count is <= j (the counts do not decrease): eigenvalue j lies between points m-1 and m
        int m = 0;
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            const e16_u4 two = *reinterpret_cast<const e16_lu4*>(xch + k);
            const bool b0 = two.x <= (unsigned)j, b1 = two.z <= (unsigned)j;
            // the last point with count <= j gives lo; the first with count > j gives hi
            clo = b0 ? two.x : clo;
            slo = b0 ? two.y : slo;
            clo = b1 ? two.z : clo;
            slo = b1 ? two.w : slo;
            m += (b0 ? 1 : 0) + (b1 ? 1 : 0);
        }
        unsigned ch2 = chi, sh2 = shi;
#pragma unroll
        for (int k = 14; k >= 0; k -= 2) {
            const e16_u4 two = *reinterpret_cast<const e16_lu4*>(xch + k);
            const bool b1 = two.z > (unsigned)j, b0 = two.x > (unsigned)j;
            ch2 = b1 ? two.z : ch2;
            sh2 = b1 ? two.w : sh2;
            ch2 = b0 ? two.x : ch2;
            sh2 = b0 ? two.y : sh2;
        }
        chi = ch2;
        shi = sh2;
        lo = m > 0 ? fma(w, (double)m, gl) : gl;
        hi = m < 16 ? fma(w, (double)(m + 1), gl) : gu;
    }
    E16_ORDER();
    E16_MARK(22);
    // ---- bisection on the global count
#pragma unroll 1
    for (int it = 0; it < E16_NBISECT; ++it) {
        const double mid = 0.5 * (lo + hi);
        unsigned s;
        if (splits) s = e16_signs(d, e2, n, mid);
        else s = e16_signs_scaled(d, e2, n, mid);
        const unsigned c = (unsigned)__builtin_popcount(s);
        const bool left = c <= (unsigned)j;              // eigenvalue j is at or above mid
        lo = left ? mid : lo;
        clo = left ? c : clo;
        slo = left ? s : slo;
        hi = left ? hi : mid;
        chi = left ? chi : c;
        shi = left ? shi : s;
    }
    E16_MARK(23);
    // ---- the block that owns eigenvalue j: it is member r = j - clo of the chi - clo eigenvalues inside (lo, hi]; walking the
    // blocks in order, block b holds inc_b = count_b(hi) - count_b(lo) of them (Kramers pairs of a cleanly split T live in
    // different blocks).  kb = index of the eigenvalue inside its block.
    bl = 0;
    bh = n - 1;
    int kb = (int)clo + ((int)j - (int)clo);             // (= j: the whole T is one block)
    int c_lo = (int)clo, c_hi = (int)chi;                // eigenvalues of the block below lo / at or below hi
    if (splits) {                                        // (row-uniform: this T splits)
        const int r = j - (int)clo;
        int cum = 0, cl = 0, ch = 0, start = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            cl += (int)((slo >> (15 - i)) & 1u);
            ch += (int)((shi >> (15 - i)) & 1u);
            const bool endb = ((split >> i) & 1u) != 0 && i < n;
            if (endb) {
                const int inc = ch - cl;
                const bool here = !found && r < cum + inc;
                bl = here ? start : bl;
                bh = here ? i : bh;
                kb = here ? cl + (r - cum) : kb;
                c_lo = here ? cl : c_lo;
                c_hi = here ? ch : c_hi;
                found = found || here;
                cum += inc > 0 ? inc : 0;
                cl = 0;
                ch = 0;
                start = i + 1;
            }
        }
        // (inconsistent counts -- possible only within the 1e-8 |T| of the reciprocal estimate of an eigenvalue of a leading
        // block: the matrix goes to the list)
        flag = flag || !found;
        if (!found) {
            bl = 0;
            bh = n - 1;
            kb = j;
        }
    }
    E16_MARK(24);
    // ---- Newton on the block's characteristic polynomial, bracket kept by the block's own Sturm count
    double x = 0.5 * (lo + hi);
    bool conv = j >= n;
    E16_DBG(dbg_slot, j, 0, lo);
    E16_DBG(dbg_slot, j, 1, hi);
    E16_DBG(dbg_slot, j, 2, clo);
    E16_DBG(dbg_slot, j, 3, chi);
    E16_DBG(dbg_slot, j, 4, kb);
    E16_DBG(dbg_slot, j, 5, bl * 100 + bh);
    int dbg_it = 0;
#pragma unroll 1
    for (int it = 0; it < E16_NEWTON_MAX; ++it) {
        if (__builtin_amdgcn_ballot_w64(!conv) == 0) break;
        double p1 = 1.0, p2 = 0.0, dp1 = 0.0, dp2 = 0.0;
        unsigned sgn = 0, len = 16u;
        if (splits) {
            len = 0;
            e16_poly_step<0, true>(d, e2, n, bl, bh, x, p1, p2, dp1, dp2, sgn, len);
        } else {
            len = (unsigned)n;
            e16_newton_scaled_step<0>(d, e2, n, x, p1, p2, dp1, dp2, sgn);
        }
        // sign changes of 1, p_bl, .., p_bh: bit k of sgn = sign of the (len - k)-th value; the leading 1 is positive
        const unsigned chg = (unsigned)__builtin_popcount((sgn ^ (sgn >> 1)) & ((1u << len) - 1u) & 0xffffu);
        const bool left = (int)chg <= kb;                // the root is at or above x
        const double nlo = left ? x : lo, nhi = left ? hi : x;
        const int nclo = left ? (int)chg : c_lo, nchi = left ? c_hi : (int)chg;
        // Newton for a root of multiplicity m = eigenvalues of the block still inside the bracket: x - m p / p'.  One eigenvalue
        // (every bracket but those of pairs closer than the 2^-12 |T| the bisection leaves): m = 1 and the same bits as before.
        // Twins -- Kramers pairs, spin-degenerate bands: both lanes hold the pair in their bracket, plain Newton on the (near-)
        // double root halves its error per step and twelve steps ended 2.5e-8 |T| away; with m = 2 three steps reach rounding.
        const double mult = (double)(nchi - nclo > 1 ? nchi - nclo : 1);
        double y = __builtin_amdgcn_rcp(dp1);
        y = fma(fma(-dp1, y, 1.0), y, y);
        double xn = fma(-(p1 * mult), y, x);
        // A step is taken if it stays inside the bracket -- with a slack of 64 eps |T|: from the far side of a root whose near
        // bound has already closed in to rounding level, the (correct) step lands a few 1e-15 beyond that bound, and the midpoint
        // that a strict test would take instead is half the old error away (12 such steps ended 1e-7 from the root).
        const bool inside = xn > nlo - 1.4210854715202004e-14 && xn < nhi + 1.4210854715202004e-14;   // (false for NaN / inf: a vanishing derivative)
        xn = inside ? xn : 0.5 * (nlo + nhi);
        // converged: a Newton step of at most 2^-34 |T| is the last one needed -- from there the error after it is step^2 x (sum of
        // 1 / distance to the other eigenvalues of the block) <= 3.4e-21 x 2e5 at the gap where matrices are listed anyway
        const bool done = p1 == 0.0 || (inside && fabs(xn - x) <= E16_NEWTON_TOL) || !(nhi - nlo > 0.0);
        if (!conv) {
            lo = nlo;
            hi = nhi;
            c_lo = nclo;
            c_hi = nchi;
            x = p1 == 0.0 ? x : xn;
            conv = done;
            ++dbg_it;
            E16_DBG(dbg_slot, j, 9, inside ? 1.0 : 0.0);
            E16_DBG(dbg_slot, j, 10, p1);
            E16_DBG(dbg_slot, j, 11, dp1);
            E16_DBG(dbg_slot, j, 12, (double)chg);
        }
    }
#ifdef E16_ITER_HIST   // (profiles/microbench/e16_bench.hip only: Newton steps per lane [0..15] and per wavefront [16..31])
    atomicAdd(&e16_iter_hist[dbg_it < 15 ? dbg_it : 15], 1u);
    {
        int wmax = dbg_it;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off));
        if ((threadIdx.x & 63) == 0) atomicAdd(&e16_iter_hist[16 + (wmax < 15 ? wmax : 15)], 1u);
    }
#endif
    if constexpr (RESCUE) {
        // eigenvalues only: nothing is listed, so nothing may be left to the list.  EVERY lane's result is checked against the exact
        // pivot count (reciprocal refined to full precision, dstebz's recurrence: backward stable): eigenvalue j must lie inside
        // (x - 3e-15, x + 3e-15] |T|.  Newton's answer passes unless the spectrum crowds (a pair merged to its mean, a cluster
        // the bracket did not separate, a graded T) or the iteration did not settle; a lane that fails bisects on the same count
        // from the Gershgorin interval down to rounding level.  Two counts per lane, 7 % of this form; the rest is rare.
        E16_ORDER();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const e16_d2 t = xd[i];
            d[i] = t.x;
            e2[i] = fmax(t.y * t.y, 1e-300);
        }
        unsigned c_below = 0, c_above = 0;
        {
            double q = 1.0;
            e16_count_step_exact<0>(d, e2, x - 3e-15, q, c_below);
            q = 1.0;
            e16_count_step_exact<0>(d, e2, x + 3e-15, q, c_above);
        }
        const bool need = (flag || !conv || !(c_below <= (unsigned)j && (unsigned)j < c_above)) && j < n;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
            double blo = gl, bhi = gu;
#pragma unroll 1
            for (int it = 0; it < 64; ++it) {
                const double mid = 0.5 * (blo + bhi);
                double q = 1.0;
                unsigned cnt = 0;
                e16_count_step_exact<0>(d, e2, mid, q, cnt);
                const bool left = cnt <= (unsigned)j;
                blo = left ? mid : blo;
                bhi = left ? bhi : mid;
            }
            if (need) x = 0.5 * (blo + bhi);
        }
        conv = true;
        flag = false;
        // (a matrix with a NaN or an infinity in it has no eigenvalues to bisect for: the caller raises like the reference's eigh.
        // fmax drops NaNs, so the norm above does not show them: ask d and e themselves, any position of the row)
        const unsigned long long nb = __builtin_amdgcn_ballot_w64(!(fabs(dd) < INFINITY) || !(fabs(ee_in) < INFINITY));
        flag = flag || !(tn < INFINITY) || ((unsigned)(nb >> (lane & 48)) & 0xffffu) != 0;
    }
    flag = flag || !conv;
    lam = x;
    E16_ORDER();
#pragma unroll
    for (int i = 0; i < 16; ++i) {                       // (d and the couplings, for the twisted factorisation)
        const e16_d2 t = xd[i];
        d[i] = t.x;
        e[i] = t.y;
    }
    E16_ORDER();
    if (j >= n) {                                        // a padding row: decoupled, its eigenvector is e_j (V stays orthogonal)
        bl = j;
        bh = j;
        lam = qle_pick<0>(d, j, 0.0);
    }
    E16_DBG(dbg_slot, j, 6, x);
    E16_DBG(dbg_slot, j, 7, dbg_it);
    E16_DBG(dbg_slot, j, 8, conv ? 1.0 : 0.0);
}

// ---------------------------------------------------------------- 3. eigenvector of T for lam by the twisted factorisation
#define E16_TINY 1e-290
__device__ __forceinline__ double e16_guard(const double p) { return fabs(p) < E16_TINY ? -E16_TINY : p; }

// v = unit eigenvector of the (scaled, split) T for its eigenvalue lam inside the block [bl, bh]; dlam = the Rayleigh-quotient
// correction gamma_r / |z|^2; returns true when the residual |gamma_r| / |z| is not at rounding level
__device__ __forceinline__ bool e16_twisted(const double (&d0)[16], const double (&e)[16], const double lam, const int bl, const int bh,
                                            double (&v)[16], double& dlam) {
    double d[16];
    double tnorm = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        d[i] = d0[i] - lam;                            // s_i = d_i - lambda
        tnorm = fmax(tnorm, fabs(d[i]));
    }
    // top-down pivots dp_{i+1} = s_{i+1} - e_i lp_i, lp_i = e_i / dp_i; bottom-up dm_i = s_i - e_i um_i, um_i = e_i / dm_{i+1}
    double lp[15], um[15];
    {
        double dp = d[0], dm = d[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            lp[i] = e[i] * e16_rcp(e16_guard(dp));
            dp = fma(-e[i], lp[i], d[i + 1]);
            const int k = 14 - i;
            um[k] = e[k] * e16_rcp(e16_guard(dm));
            dm = fma(-e[k], um[k], d[k]);
        }
    }
    // gamma_k = s_k - e_{k-1} lp_{k-1} - e_k um_k; r = position of the smallest |gamma| inside the block
    double gmin = INFINITY, gam_r = 0.0;
    int r = bl;
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
    dlam = gam_r * inz * inz;
    // residual |(T - lambda) z| / |z| = |gamma_r| / |z|: a few eps |T| for an eigenvalue that accurate
    const double tn = tnorm + fabs(lam);
    return !(fabs(gam_r) * inz <= 1e-13 * tn);   // (1e-11 until round 5: pairs split by a little more than gaptol kept vectors with residuals of 1e-12, profiles/evecs_stress.py)
}

// The SECOND eigenvector of a pair of eigenvalues of one block closer than gaptol |T| (twins: Kramers pairs, spin-degenerate bands,
// or merely close levels).  The twisted factorisation gives both lanes of such a pair (nearly) the same vector; xSTEIN's remedy:
// inverse iteration with reorthogonalisation against the first member.  w = the first member's unit vector (lane j - 1's, fetched
// by the caller), v = this lane's start vector (its own twisted-factorisation vector) and result; T - lam = L D L^T top-down with
// d, e streamed from LDS (guarded pivots: the solve only has to amplify the eigenspace, the checks at the end decide).  Two rounds of
// orthogonalise | solve | orthogonalise | normalise, then the Rayleigh quotient (dlam) and the residual of (lam + dlam, v).
// Returns false when the result is not an eigenvector to 1e-14 |T| -- the matrix then goes to the list like before.
__device__ __forceinline__ bool e16_twin(const e16_lcd* xd, const double lam, const int bl, const int bh, const double (&w)[16], double (&v)[16],
                                         double& dlam, double* dbg = nullptr) {
    // start: a fixed generic vector on the rows of the block (the lane's own twisted-factorisation vector is the first member's
    // bit for bit when the two eigenvalues came out equal: nothing would be left of it after the orthogonalisation)
    {
        const double c[16] = {0.61, -0.37, 0.93, 0.28, -0.75, 0.49, 0.17, -0.88, 0.55, -0.23, 0.71, 0.39, -0.64, 0.82, -0.12, 0.45};
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = i >= bl && i <= bh ? c[i] : 0.0;
    }
    double l[15], rdp[16];
    {
        e16_d2 t = xd[0];
        double dp = t.x - lam;
#pragma unroll
        for (int i = 0; i < 15; ++i) {
            rdp[i] = e16_rcp(e16_guard(dp));
            l[i] = t.y * rdp[i];
            const e16_d2 tn = xd[i + 1];
            dp = fma(-t.y, l[i], tn.x - lam);
            t = tn;
        }
        rdp[15] = e16_rcp(e16_guard(dp));
    }
    auto orth = [&]() {
        double c = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) c = fma(w[i], v[i], c);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = fma(-c, w[i], v[i]);
    };
    auto normalise = [&]() {
        double m = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) m = fmax(m, fabs(v[i]));
        int ex = 0;
        (void)frexp(m, &ex);
        const double s = m > 0.0 && m < INFINITY ? ldexp(1.0, -ex) : 1.0;      // (the solve grows by up to 1 / pivot)
        double nz2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            v[i] *= s;
            nz2 = fma(v[i], v[i], nz2);
        }
        const double inz = nz2 > 0.0 ? rsqrt_full(nz2) : 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= inz;
    };
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        orth();
        normalise();
#pragma unroll
        for (int i = 0; i < 15; ++i) v[i + 1] = fma(-l[i], v[i], v[i + 1]);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] *= rdp[i];
#pragma unroll
        for (int i = 14; i >= 0; --i) v[i] = fma(-l[i], v[i + 1], v[i]);
        normalise();
        orth();
        normalise();
    }
    // r = (T - lam) v, rq = v.r (Rayleigh correction), residual of (lam + rq, v)
    double rq = 0.0, r2 = 0.0, cw = 0.0, nv2 = 0.0;
    {
        double eprev = 0.0, vprev = 0.0;
        double r[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const e16_d2 t = xd[i];
            r[i] = fma(t.x - lam, v[i], eprev * vprev);
            if (i < 15) r[i] = fma(t.y, v[i + 1], r[i]);
            eprev = t.y;
            vprev = v[i];
            rq = fma(v[i], r[i], rq);
            cw = fma(w[i], v[i], cw);
            nv2 = fma(v[i], v[i], nv2);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const double q = fma(-rq, v[i], r[i]);
            r2 = fma(q, q, r2);
        }
    }
    dlam = rq;
    if (dbg) {
        dbg[0] = r2;
        dbg[1] = cw;
        dbg[2] = nv2;
    }
    return r2 <= 1e-28 && fabs(cw) <= 1e-12 && fabs(rq) <= 1e-9 && fabs(nv2 - 1.0) <= 1e-12;
}

// ---------------------------------------------------------------- the kernel
// MODE 0: k list, 1: regular mesh into a wf_array, 2: supplied matrices.  The launch covers the matrices [id0, id0 + nc);
// list / count: the matrices (relative to id0) left to the QL-replay kernels.
// VEC = false (round 5): eigenvalues only -- the reflector record, the phases, the eigenvector stages and the stores are left out
// and nothing is listed (e16_eigenvalue<RESCUE>); eigenvalue-only k lists of 9..16 states used to take an older pair of kernels
// that was SLOWER than this kernel with its eigenvectors (cubic16: 5.1 against 3.7 ns per point).
template <int MODE, bool VEC = true>
__global__ __launch_bounds__(256, 3) void k_e16(const ModelView mv, const int64_t nk, const ListArgs Lst, const GridArgs G, const int64_t id0,
                                                const int64_t nc, int* __restrict__ list, int* __restrict__ count, const double gaptol,
                                                const int form) {
#ifndef E16_LDS_PAD   // (profiles/microbench/e16_bench.hip only: extra LDS per block, to run the kernel at a lower occupancy)
#define E16_LDS_PAD 0
#endif
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4 * E16_WAVE_LDS + E16_LDS_PAD];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = lane & 15, mat = lane >> 4, g = lane >> 4;
    const int j = x;
    e16_lcd* const wrec = (e16_lcd*)(lds_all + wv * E16_WAVE_LDS);     // [4][E16_REC]
    e16_lcd* const wxch = wrec + 4 * E16_REC;                                          // [E16_XCH]
    e16_lcd* const rec = wrec + mat * E16_REC;
    const int64_t wslot0 = ((int64_t)blockIdx.x * 4 + wv) * 4;                          // first matrix of this wavefront
    if (wslot0 >= nc) return;                                                           // (wave-uniform)
    const int64_t slot_u = wslot0 + mat;
    const bool live = slot_u < nc;
    const int64_t slot = live ? slot_u : nc - 1;                                        // idle tail rows shadow the last matrix
    const int64_t id = id0 + slot;
    const int n = mv.nsta;
    const bool real_row = x < n;

    // mesh indices of a point (MODE 1), for the tables of exp(2 pi i k_d) here and of the orbital phases at the end
    auto mesh_indices = [&](const int64_t pid, int (&mi_)[4]) {
#pragma unroll
        for (int d = 0; d < 4; ++d) mi_[d] = 0;
        if (G.wv.npts < (int64_t)0xffffffffu) {
            unsigned rem = (unsigned)pid;
#pragma unroll
            for (int d = 3; d >= 1; --d) {
                const unsigned md = (unsigned)G.wv.mesh[d];
                if (d < G.wv.dim_arr && md > 1) {
                    const unsigned q = rem / md;
                    mi_[d] = (int)(rem - q * md);
                    rem = q;
                }
            }
            mi_[0] = (int)rem;
        } else {
            int64_t rem = pid;
#pragma unroll
            for (int d = 3; d >= 1; --d) {
                const int64_t md = G.wv.mesh[d];
                if (d < G.wv.dim_arr && md > 1) {
                    const int64_t q = rem / md;
                    mi_[d] = (int)(rem - q * md);
                    rem = q;
                }
            }
            mi_[0] = (int)rem;
        }
    };
    int mi[4] = {0, 0, 0, 0};
    if constexpr (MODE == 1) mesh_indices(id, mi);
    // ---- H(k), lane x = row x
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
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < G.wv.dim_arr) zk[d] = G.tz[d][mi[d]];
        }
        // S[x][c] = sum_R U_R[slot(min,max)] e^{2 pi i k.R}  (conjugated below the diagonal)
        int sidx[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int lo = x < c ? x : c, hi = x < c ? c : x;
            sidx[c] = real_row && c < n ? lo * n - lo * (lo - 1) / 2 + (hi - lo) : -1;
        }
        // U_R (nslot <= 136 entries) goes through the wavefront's exchange region: three coalesced loads per R (the next R's are
        // in flight while this one is accumulated), then every lane picks its 16 slots by ds_read_b128 -- the 16 lanes of a row
        // used to load their 16 scattered 16-byte entries of the table straight from L2, 112 one-KB load instructions per
        // wavefront of which three quarters were duplicates (the four matrices of a wavefront need the same entries): 19 % of
        // the kernel (profiles/microbench/e16_bench mesh, E16_SKIP=64)
        const int nsl = mv.nslot;
        const int nRr = (E16_SKIP & 64) ? 1 : mv.nR;
        // On a mesh the lattice vectors with NO component along the last axis give the same sum for every point of a mesh row:
        // S = C(k_0 .. k_{last-1}) + sum over the others of U_R e^{2 pi i k.R}.  When the four points of the wavefront lie in one
        // row, the lanes form C together -- slot t, t + 64, t + 128 each, every U_R read once, coalesced -- and pick their 16
        // entries of it from the exchange region; for cubic16 that is 3 passes over the slots instead of 7.  A wavefront that
        // straddles two rows accumulates the same terms in the same order lane by lane: C first, then the others, each term one
        // complex multiply-add from zero -- the same bits either way, so windows and halo planes stay bit-identical.
        const int lastax = MODE == 1 ? G.last : -1;
        auto in_row_part = [&](const int4 R) {             // (wave-uniform)
            return lastax >= 0 && (lastax == 0 ? R.x : lastax == 1 ? R.y : lastax == 2 ? R.z : R.w) == 0;
        };
        bool shared = false;
        bool same = true;                                  // this lane's point lies in the mesh row of lane 0's
        if constexpr (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 3; ++d)
                if (d < lastax) same = same && mi[d] == __builtin_amdgcn_readfirstlane(mi[d]);
            shared = lastax >= 1 && __builtin_amdgcn_ballot_w64(!same && live) == 0;
        }
        // ---- models with MANY lattice vectors (a Wannier-interpolated model of 9..16 functions: ~100, most of them with a component
        // along the last axis): the per-vector staging below -- a scalar load of R ahead of every branch, three vector loads waited
        // for, an LDS round trip and 64 multiply-adds per lane, each vector on its own -- made a dense 16-function model 10 x slower per
        // point than cubic16 (profiles/e16_many_R_probe.py: 83 against 7.8 ns).  Such models take the ROW's coefficient cells,
        //     S = sum_p C_p z_last^p,   C_p = sum over the R with R_last = p of U_R exp(2 pi i k_lead . R_lead),
        // C_p formed by the lanes together like the row part below (slot t, t + 64, t + 128 each: three multiply-adds per lattice
        // vector), the lattice vectors held in the LANES (one coalesced load per 64; which of them belong to p is one ballot, the
        // walk a bit scan, their lead phases computed by the lanes in parallel and fetched by v_readlane: no memory in the control
        // flow), one LDS hand-over and 64 multiply-adds per lane PER p.  A wavefront that straddles two rows does all of it once per
        // row, the lanes of the other row idling through the last step: the same bits for a point whatever shares its wavefront.
        // The choice depends on the model alone (nR), so windows and shards stay bit-identical; cubic16 (7 vectors) keeps the form below.
        bool cells_done = false;
        if constexpr (MODE == 1) {
            if (nRr > E16_CELLS_MIN_NR && lastax >= 0 && (form & E16_F_NO_CELLS) == 0) {
                cells_done = true;
                const int pmax = mv.pmax;
                // z_last by 0 / 1 weights (a select on `lastax` would turn zk[] into an indexed array in scratch memory)
                const double w0 = lastax == 0 ? 1.0 : 0.0, w1 = lastax == 1 ? 1.0 : 0.0, w2 = lastax == 2 ? 1.0 : 0.0, w3 = lastax == 3 ? 1.0 : 0.0;
                const cd zl{fma(w3, zk[3].x, fma(w2, zk[2].x, fma(w1, zk[1].x, w0 * zk[0].x))),
                            fma(w3, zk[3].y, fma(w2, zk[2].y, fma(w1, zk[1].y, w0 * zk[0].y)))};
                const unsigned long long other = __builtin_amdgcn_ballot_w64(!same);
                const int nrow = other != 0 ? 2 : 1;
                for (int rowi = 0; rowi < nrow; ++rowi) {
                    const bool mine = rowi == 0 ? same : !same;
                    const int lsrc = rowi == 0 ? 0 : (int)__builtin_ctzll(other);
                    cd zrow[4];                            // exp(2 pi i k_d) of the row's leading axes (wave-uniform)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const cd zd = d < lastax ? zk[d] : cd{1.0, 0.0};
                        zrow[d] = cd{e16_lane_d(zd.x, lsrc), e16_lane_d(zd.y, lsrc)};
                    }
                    cd zc{1.0, 0.0};                       // z_last^p, from p = -pmax upwards
                    for (int q = 0; q < pmax; ++q) zc = cd{fma(zc.x, zl.x, zc.y * zl.y), fma(zc.y, zl.x, -(zc.x * zl.y))};
                    for (int p = -pmax; p <= pmax; ++p) {
                        cd cs[3] = {cd{0.0, 0.0}, cd{0.0, 0.0}, cd{0.0, 0.0}};
                        bool any = false;
                        for (int base = 0; base < nRr; base += 64) {
                            const bool have = base + lane < nRr;
                            int4 Rl = have ? mv.rvec[base + lane] : int4{0, 0, 0, 0};
                            const int rl = lastax == 0 ? Rl.x : lastax == 1 ? Rl.y : lastax == 2 ? Rl.z : Rl.w;
                            unsigned long long bits = __builtin_amdgcn_ballot_w64(have && rl == p);
                            if (bits == 0) continue;
                            any = true;
                            if (lastax == 0) Rl.x = 0; else if (lastax == 1) Rl.y = 0; else if (lastax == 2) Rl.z = 0; else Rl.w = 0;
                            const cd phl = e16_phase_of_R(zrow, Rl);       // lane l: the lead phase of R_{base + l}
                            while (bits != 0) {
                                int rr[2];
                                cd u[2][3];
#pragma unroll
                                for (int q = 0; q < 2; ++q) {
                                    rr[q] = bits != 0 ? (int)__builtin_ctzll(bits) : -1;
                                    bits &= bits - 1;      // (0 stays 0)
                                    if (rr[q] >= 0) {
                                        const cd* un = mv.rblock + (size_t)(base + rr[q]) * nsl;
#pragma unroll
                                        for (int t = 0; t < 3; ++t) u[q][t] = t * 64 + lane < nsl ? un[t * 64 + lane] : cd{0.0, 0.0};
                                    }
                                }
#pragma unroll
                                for (int q = 0; q < 2; ++q) {
                                    if (rr[q] >= 0) {
                                        const cd ph{e16_lane_d(phl.x, rr[q]), e16_lane_d(phl.y, rr[q])};
#pragma unroll
                                        for (int t = 0; t < 3; ++t)
                                            if (t * 64 + lane < nsl) e16_cfma(cs[t], u[q][t], ph);
                                    }
                                }
                            }
                        }
                        if (any) {                         // (wave-uniform)
                            E16_ORDER();
#pragma unroll
                            for (int t = 0; t < 3; ++t)
                                if (t * 64 + lane < nsl) e16_put(wxch + t * 64 + lane, cs[t]);
                            E16_ORDER();
                            if (mine) {
#pragma unroll
                                for (int c = 0; c < 16; ++c)
                                    if (sidx[c] >= 0) e16_cfma(a[c], e16_get(wxch + sidx[c]), zc);
                            }
                        }
                        zc = cd{fma(zc.x, zl.x, -(zc.y * zl.y)), fma(zc.x, zl.y, zc.y * zl.x)};
                    }
                }
            }
        }
        if (!cells_done) {
        auto stage_and_add = [&](const int r) {            // a[c] += U_r[slot(x, c)] e^{2 pi i k.R_r}, U_r through the exchange region
            const cd ph = e16_phase_of_R(zk, mv.rvec[r]);
            const cd* un = mv.rblock + (size_t)r * nsl;
            cd nx[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) nx[t] = t * 64 + lane < nsl ? un[t * 64 + lane] : cd{0.0, 0.0};
            E16_ORDER();
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t * 64 + lane < nsl) e16_put(wxch + t * 64 + lane, nx[t]);
            E16_ORDER();
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (sidx[c] >= 0) e16_cfma(a[c], e16_get(wxch + sidx[c]), ph);
        };
        if (shared) {
            cd cs[3] = {cd{0.0, 0.0}, cd{0.0, 0.0}, cd{0.0, 0.0}};
            for (int r = 0; r < nRr; ++r) {
                const int4 R = mv.rvec[r];
                if (!in_row_part(R)) continue;
                const cd ph = e16_phase_of_R(zk, R);        // (no factor of the last axis: the same in every lane)
                const cd* un = mv.rblock + (size_t)r * nsl;
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t * 64 + lane < nsl) e16_cfma(cs[t], un[t * 64 + lane], ph);
            }
            E16_ORDER();
#pragma unroll
            for (int t = 0; t < 3; ++t)
                if (t * 64 + lane < nsl) e16_put(wxch + t * 64 + lane, cs[t]);
            E16_ORDER();
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (sidx[c] >= 0) a[c] = e16_get(wxch + sidx[c]);
        } else if (lastax >= 0) {                           // (k lists have no row part: no walk over the table for nothing)
            for (int r = 0; r < nRr; ++r)
                if (in_row_part(mv.rvec[r])) stage_and_add(r);
        }
        for (int r = 0; r < nRr; ++r)
            if (!in_row_part(mv.rvec[r])) stage_and_add(r);
        }
        E16_ORDER();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < x) a[c].y = -a[c].y;
            if (c == x) a[c].y = 0.0;
        }
    }

    E16_MARK(1);
    // ---- 1. tridiagonalisation.  d_x, e_x = |T[x+1][x]| and the phase of column x of D are final after step x - 1: the lane that
    // owns them then (x == K + 1: one exec mask) parks them in LDS -- (d, e) in the matrix's slots of the exchange region, idle
    // during the reflections, the phase in the record -- and every lane picks its own up afterwards.  (Kept in registers, the
    // three were six conditional moves per step, and d_x a 15-deep select chain at the end.)
    cd delta{1.0, 0.0};
    e16_ld* const xde = reinterpret_cast<e16_ld*>(wxch + mat * 16);                     // [16] x (d, e)
    if (x == 0) {
        if constexpr (VEC) e16_put(rec + 119, cd{1.0, 0.0});
        xde[0] = a[0].x;
    }
    {
#define TBK_E16_HOUSE(KK)                                     \
    {                                                         \
        double mag;                                           \
        cd unit;                                              \
        e16_house<KK, VEC>(a, x, rec, mag, unit);             \
        if constexpr (VEC) delta = cmul(delta, unit);         \
        if (x == KK + 1) {                                    \
            if constexpr (VEC) e16_put(rec + 119 + (KK + 1), delta); \
            xde[2 * KK + 1] = mag;                            \
            xde[2 * KK + 2] = a[KK + 1].x;                    \
        }                                                     \
    }
        if constexpr (!(E16_SKIP & 1)) {
        TBK_E16_HOUSE(0) TBK_E16_HOUSE(1) TBK_E16_HOUSE(2) TBK_E16_HOUSE(3) TBK_E16_HOUSE(4) TBK_E16_HOUSE(5) TBK_E16_HOUSE(6)
        TBK_E16_HOUSE(7) TBK_E16_HOUSE(8) TBK_E16_HOUSE(9) TBK_E16_HOUSE(10) TBK_E16_HOUSE(11) TBK_E16_HOUSE(12) TBK_E16_HOUSE(13)
        }
#undef TBK_E16_HOUSE
    }
    {
        const cd t14 = rowbcast_c<15>(a[14]);            // T[15][14]: never reflected
        const double t2 = cabs2(t14);
        double mag = t2;                                 // (0, or a NaN that must stay one)
        if (t2 > 0.0) {
            const double inv = rsqrt_full(t2);
            mag = t2 * inv;
            if constexpr (VEC) delta = cmul(delta, cd{t14.x * inv, t14.y * inv});
        }
        if (x == 15) {
            if constexpr (VEC) e16_put(rec + 119 + 15, delta);
            xde[29] = mag;
            xde[30] = a[15].x;
            xde[31] = 0.0;
        }
    }
    E16_ORDER();
    const e16_d2 de_x = wxch[mat * 16 + x];
    const double dd = de_x.x, ee = de_x.y;               // d_x = A[x][x], e_x (0 in lane 15)
    E16_ORDER();
    E16_MARK(2);
    double d[16], e[16];
    E16_ORDER();
    // ---- 2. eigenvalue j (scaled T), 3. its eigenvector of T
    double scale = 1.0, lam = 0.0;
    int bl = 0, bh = 15;
    unsigned split = 0;
    bool flag = false;
    e16_eigenvalue<!VEC>(dd, ee, d, e, n, j, wxch + mat * 16, scale, lam, bl, bh, split, flag, slot_u);
    E16_MARK(3);
    if constexpr (!VEC) {
        // ---- eigenvalues only: ascending by construction up to ties between the blocks of a split T (neighbours exchange), out
        int tid3 = threadIdx.x;
        asm volatile("" : "+v"(tid3));
        const int64_t slot_u3 = ((int64_t)blockIdx.x * 4 + (tid3 >> 6)) * 4 + ((tid3 & 63) >> 4);
        const bool live3 = slot_u3 < nc;
        const int64_t id3 = id0 + (live3 ? slot_u3 : nc - 1);
        double lam_o = lam;
        if (__builtin_amdgcn_ballot_w64((split & 0x7fffu & ((1u << (n - 1)) - 1u)) != 0) != 0) {
            // (the eigenvalues of different blocks of a split T that agree to rounding may come out in any order: a full odd-even
            // transposition sort along the row -- values only, there are no vectors to keep in step)
#pragma unroll 1
            for (int pass = 0; pass < 16; ++pass) {
                const double up = e16_next(lam_o), dn = e16_prev(lam_o);
                if ((j & 1) == (pass & 1)) {
                    if (j + 1 < n && up < lam_o) lam_o = up;
                } else {
                    if (j >= 1 && j < n && dn > lam_o) lam_o = dn;
                }
            }
        }
        if constexpr (MODE != 1) {
            if (live3 && j < n) Lst.eval[(int64_t)j * nk + id3] = lam_o * e16_rcp(scale);   // (scale is a power of two: exact)
            if (flag && j < n && live3 && Lst.flags) Lst.flags[0] = 1;                              // (NaN / infinite input: TBK_ENOCONV)
        }
        return;
    }
    double v[16], dlam = 0.0;
    bool bad = false;
    if constexpr (!(E16_SKIP & 4)) bad = e16_twisted(d, e, lam, bl, bh, v, dlam) && j < n;
    else {
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = i == j ? 1.0 : d[i] * 1e-3;
    }
    // ---- twins: the second member of an isolated pair of close eigenvalues of one block gets its vector by inverse iteration
    // against the first (e16_twin); pairs inside longer clusters stay for the list (close_pair below)
    bool twin_ok = false;
    {
        const double lam_dn = e16_prev(lam), lam_up = e16_next(lam);
        const int bl_dn = __builtin_amdgcn_update_dpp(0, bl, 0x111, 0xf, 0xf, true);     // lane j - 1
        const int bl_up = __builtin_amdgcn_update_dpp(0, bl, 0x101, 0xf, 0xf, true);     // lane j + 1
        const bool cdn = j >= 1 && j < n && bl_dn == bl && !(lam - lam_dn >= gaptol);    // (scaled: |T| < 1, so no narrower than the test below)
        const bool cup = j + 1 < n && bl_up == bl && !(lam_up - lam >= gaptol);
        const bool cdn_dn = __builtin_amdgcn_update_dpp(0, cdn ? 1 : 0, 0x111, 0xf, 0xf, true) != 0;
        const bool second = cdn && !cup && !cdn_dn && !bad && !flag;
        if (__builtin_amdgcn_ballot_w64(second) != 0) {     // (wave-uniform)
            double w[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = e16_prev(v[i]);     // the first member's vector, before this lane's changes
            if (second) {
                double dl2 = 0.0;
#ifdef E16_DEBUG
                double tdbg[3] = {0.0, 0.0, 0.0};
                const bool ok = e16_twin(wxch + mat * 16, lam, bl, bh, w, v, dl2, tdbg);
                E16_DBG(slot_u, j, 9, tdbg[0]);
                E16_DBG(slot_u, j, 10, tdbg[1]);
                E16_DBG(slot_u, j, 11, tdbg[2]);
                E16_DBG(slot_u, j, 12, ok ? 1.0 : 0.0);
#else
                const bool ok = e16_twin(wxch + mat * 16, lam, bl, bh, w, v, dl2);
#endif
                dlam = dl2;
                twin_ok = ok;
                bad = bad || !ok;
            }
        }
    }
    E16_MARK(4);
    // From here on the matrix this lane works for, whether it exists, and its mesh indices are derived AGAIN from the thread
    // index (through a register the compiler cannot see through): held from the top of the kernel, those eight registers were
    // spilled to scratch across the eigenvalue and eigenvector stages -- 72 bytes per lane, 0.3 GB of stores and 0.2 GB of loads per
    // 65^3 points on a kernel whose whole output is 1.1 GB (profiles/r04acfg).
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int64_t slot_u2 = ((int64_t)blockIdx.x * 4 + (tid2 >> 6)) * 4 + ((tid2 & 63) >> 4);
    const bool live2 = slot_u2 < nc;
    const int64_t slot2 = live2 ? slot_u2 : nc - 1;
    const int64_t id2 = id0 + slot2;
    const double lam_s = lam + dlam;                    // Rayleigh-quotient correction
    E16_DBG(slot_u, j, 13, dlam);
    E16_DBG(slot_u, j, 14, bad ? 1.0 : 0.0);
    E16_DBG(slot_u, j, 15, scale);
    // |T| = the largest eigenvalue in magnitude, as in round 3's kernels
    double tmax = j < n ? fabs(lam_s) : 0.0;
    tmax = fmax(tmax, row_ror_d<8>(tmax));
    tmax = fmax(tmax, row_ror_d<4>(tmax));
    tmax = fmax(tmax, row_ror_d<2>(tmax));
    tmax = fmax(tmax, row_ror_d<1>(tmax));
    // Eigenvalues of DIFFERENT blocks that agree to rounding (Kramers pairs of a cleanly split T) may come out in either order:
    // neighbours exchange their values (not their vectors) so that the bands ascend; a disorder beyond rounding lists the matrix
    double lam_o = lam_s;
    bool disorder = false;
    if (__builtin_amdgcn_ballot_w64((split & 0x7fffu & ((1u << (n - 1)) - 1u)) != 0) != 0) {   // (wave-uniform: some T of the wavefront splits)
        const double tol = 1.4210854715202004e-14 * tmax;   // 64 eps |T|
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const double up = e16_next(lam_o), dn = e16_prev(lam_o);   // lanes j + 1 and j - 1
            const bool lead = (j & 1) == pass;
            if (lead) {
                if (j + 1 < n && up < lam_o) {
                    disorder = disorder || lam_o - up > tol;
                    lam_o = up;
                }
            } else {
                if (j >= 1 && j < n && dn > lam_o) {
                    disorder = disorder || dn - lam_o > tol;
                    lam_o = dn;
                }
            }
        }
    }
    {   // a repaired twin must not come out below its partner (their Rayleigh corrections differ by ~1e-17 |T|): the bands ascend
        const double dn = e16_prev(lam_o);
        if (twin_ok && dn > lam_o) lam_o = dn;
    }
    const double lam_out = lam_o * e16_rcp(scale);      // (scale is a power of two: exact)
    // two eigenvalues of ONE block closer than gaptol |T| (the Newton-Schulz step below would not reach rounding level)
    bool close_pair = disorder;
    {
        const double up = e16_next(lam_o);              // eigenvalue j + 1 (lane 15: excluded)
        const int bl_up = __builtin_amdgcn_update_dpp(0, bl, 0x101, 0xf, 0xf, true);
        const bool up_fixed = __builtin_amdgcn_update_dpp(0, twin_ok ? 1 : 0, 0x101, 0xf, 0xf, true) != 0;   // lane j + 1 is the repaired twin of this one
        close_pair = close_pair || (j + 1 < n && ((bl_up == bl && !(up - lam_o >= gaptol * tmax) && !up_fixed) || up < lam_o));
    }
    const unsigned long long fb = __builtin_amdgcn_ballot_w64((flag || bad || close_pair) && live2 && j < n);
    const bool listed = ((unsigned)(fb >> (lane & 48)) & 0xffffu) != 0;
    if (listed && j == 0 && live2) list[atomicAdd(count, 1)] = (int)slot2;

    // eigenvalues out / minimal gaps of the mesh (listed matrices: the QL-replay kernels report theirs)
    if constexpr (MODE == 1) {
        double gap = e16_next(lam_out) - lam_out;
        gap = (j + 1 < n && live2 && !listed) ? gap : INFINITY;
        gap = fmin(gap, __shfl_xor(gap, 16));
        gap = fmin(gap, __shfl_xor(gap, 32));
        if (lane < 15 && lane + 1 < n) {
            unsigned long long* slotp = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + lane;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
            if (bits < __hip_atomic_load(slotp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slotp, bits);
        }
    } else {
        if (live2 && j < n) Lst.eval[(int64_t)j * nk + id2] = lam_out;
    }

    E16_MARK(5);
    // ---- 4. one Newton-Schulz step V <- V (1.5 I - 0.5 V^T V) = V (I - E / 2), E = V^T V - I.
    // The vectors of the twisted factorisation are orthogonal to ~eps |T| / (distance of their eigenvalues): what is not at rounding
    // level already sits beside the diagonal of E.  A matrix whose T does not split and that has NO three eigenvalues inside
    // E16_NS_TRIPLE |T| and no two inside E16_NS_PAIR |T| takes the step
    // with E cut to its tridiagonal part -- the neighbours' vectors by DPP, 150 instructions, no LDS.  The others take the full
    // step, per matrix two 16 x 16 x 16 real products on the matrix cores (v_mfma_f64_16x16x4_f64: lane l supplies
    // A[l & 15][4 kb + (l >> 4)] and B[4 kb + (l >> 4)][l & 15], and holds D[(l >> 4) + 4 r][l & 15] in register r,
    // profiles/microbench/mfma_f64_layout.hip), one matrix at a time through the exchange region (16 x 18 doubles): on gfx950 a
    // v_mfma_f64_16x16x4 holds the SIMD's double-precision pipe for 64 cycles (profiles/microbench/valu_rates.hip) -- the 32 of a
    // wavefront were 9 % of the kernel, for every matrix (E16_SKIP=8).  The choice depends on the matrix alone.
    unsigned long long ns_full = 0;
    if constexpr (!(E16_SKIP & 8)) {
        const double lam_up2 = e16_next(e16_next(lam_o));                 // eigenvalue j + 2
        // (a T that splits interleaves the eigenvalues of its blocks: neighbours in the spectrum are then exactly orthogonal and
        // the overlaps that matter lie further from the diagonal -- the full step)
        const bool splits_here = (split & 0x7fffu & ((1u << (n - 1)) - 1u)) != 0;
        // (... and a close pair leaves its vectors -- the second one of a twin comes from inverse iteration -- less accurate towards the
        // rest of the spectrum, not only towards each other)
        const double lam_up1 = e16_next(lam_o);
        const bool near = (j + 2 < n && !(lam_up2 - lam_o >= E16_NS_TRIPLE * tmax)) || (j + 1 < n && !(lam_up1 - lam_o >= E16_NS_PAIR * tmax));
        ns_full = __builtin_amdgcn_ballot_w64(live2 && (splits_here || near || (form & E16_F_NS_FULL) != 0));
        if ((((unsigned)(ns_full >> (lane & 48))) & 0xffffu) == 0) {      // (row-uniform) this lane's matrix: the tridiagonal part
            double vn[16];
            double s_up = 0.0, s_self = 0.0;                              // v_j . v_{j+1} (0 in lane 15), v_j . v_j
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                vn[i] = e16_next(v[i]);
                s_up = fma(v[i], vn[i], s_up);
                s_self = fma(v[i], v[i], s_self);
            }
            const double h_dn = -0.5 * e16_prev(s_up), h_up = -0.5 * s_up, cs = fma(-0.5, s_self, 1.5);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double vp = e16_prev(v[i]);                         // (lane j - 1 has not touched its v[i] yet: same instruction)
                v[i] = fma(h_dn, vp, fma(h_up, vn[i], cs * v[i]));
            }
        }
    }
    if constexpr (!(E16_SKIP & 8)) {
        e16_ld* const Vm = reinterpret_cast<e16_ld*>(wxch);
#pragma unroll
        for (int m4 = 0; m4 < 4; ++m4) {
            if (((unsigned)(ns_full >> (16 * m4)) & 0xffffu) == 0) continue;   // (wave-uniform)
            E16_ORDER();
            if (mat == m4) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) *reinterpret_cast<e16_lcd*>(Vm + j * 18 + i) = e16_d2{v[i], v[i + 1]};
            }
            E16_ORDER();
            e16_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double av = Vm[j * 18 + 4 * kb + g];               // V[4 kb + g][j] = A^T and B alike: G = V^T V
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, av, acc, 0, 0, 0);
            }
            e16_d4 xr;                                                    // X[g + 4 r][j] = 1.5 delta - 0.5 G
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = fma(-0.5, acc[r], (g + 4 * r) == j ? 1.5 : 0.0);
            e16_d4 vn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                const double av = Vm[(4 * kb + g) * 18 + j];              // A[x = j][k = 4 kb + g] = V[j][4 kb + g]
                vn = __builtin_amdgcn_mfma_f64_16x16x4f64(av, xr[kb], vn, 0, 0, 0);   // B[k][col] = X[4 kb + g][col]: own register kb
            }
            // vn[r] = V'[x = g + 4 r][column j]: back into the column-major image, in place (all reads of this matrix are done)
            E16_ORDER();
#pragma unroll
            for (int r = 0; r < 4; ++r) Vm[j * 18 + g + 4 * r] = vn[r];
            E16_ORDER();
            if (mat == m4) {
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const e16_d2 t = *reinterpret_cast<const e16_lcd*>(Vm + j * 18 + i);
                    v[i] = t.x;
                    v[i + 1] = t.y;
                }
            }
        }
        E16_ORDER();
    }

    E16_MARK(6);
    // ---- 5. z = H_0 ( H_1 ( ... H_13 (D v))), H_K = I - w_K w_K^+; the reflectors are read from LDS by broadcast
    cd y[16];
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) {
        const cd ph = e16_get(rec + 119 + xx);
        y[xx] = cd{ph.x * v[xx], ph.y * v[xx]};
    }
    if constexpr (!(E16_SKIP & 16)) {
        auto reflect = [&](auto KC) {
            constexpr int K = decltype(KC)::value;
            const e16_lcd* const ru = rec + (e16_off(K) - (K + 1));
            cd u[16];
            cd w{0.0, 0.0};
#pragma unroll
            for (int xx = K + 1; xx < 16; ++xx) {
                u[xx] = e16_get(ru + xx);
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
    }
    E16_ORDER();

    E16_MARK(7);
    // ---- 6. transpose through LDS (row stride 17 doubles, over the records: all reads of them are done): lane c of a matrix
    // receives component c of every vector
    e16_ld* const Ts = reinterpret_cast<e16_ld*>(wrec);                                 // [4][16][17]
    cd zt[16];
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) Ts[(mat * 16 + j) * 17 + xx] = y[xx].x;
    E16_ORDER();
#pragma unroll
    for (int b = 0; b < 16; ++b) zt[b].x = Ts[(mat * 16 + b) * 17 + j];
    E16_ORDER();
#pragma unroll
    for (int xx = 0; xx < 16; ++xx) Ts[(mat * 16 + j) * 17 + xx] = y[xx].y;
    E16_ORDER();
#pragma unroll
    for (int b = 0; b < 16; ++b) zt[b].y = Ts[(mat * 16 + b) * 17 + j];

    const int c = j;                                   // from here on the lane owns orbital component c
    if (!live2 || c >= n) return;
    cd f{1.0, 0.0};
    if constexpr (MODE == 0) {
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int dd2 = 0; dd2 < 4; ++dd2)
            if (dd2 < mv.dim_k) kk[dd2] = Lst.k[id2 * mv.dim_k + dd2];
        f = cconj(expi2pi(kdot(kk, mv.orb[c])));
    } else if constexpr (MODE == 1) {
        // exp(-2 pi i k.tau_c) x (pbc phase on the periodic images) as the product of the per-axis tables k_grid_tables wrote for
        // this window (one entry per axis and orbital, computed from the GLOBAL index: the same bits in every window)
        int mj[4];
        mesh_indices(id2, mj);
        f = G.tf[0][(int64_t)mj[0] * n + c];
#pragma unroll
        for (int dd2 = 1; dd2 < 4; ++dd2)
            if (dd2 < G.wv.dim_arr) f = cmul(f, G.tf[dd2][(int64_t)mj[dd2] * n + c]);
    }
    // (lane b of the matrix computed eigenvalue b: the bands are in ascending order by construction)
    // One 64-bit address per lane, stepped by the (wave-uniform) band stride: formed band by band from (band, point, component) it
    // was three 64-bit multiplies per store, 250 instructions of this stage.
    cd* outp;
    int64_t bstride;
    if constexpr (MODE == 1) {
        outp = wf_at(G.wv, 0, id2) + c;
        bstride = G.wv.npts * G.wv.ncomp;
    } else {
        outp = Lst.evec + (id2 * n + c);
        bstride = nk * n;
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        if (b < n) {
            const cd val = cmul(zt[b], f);
            if constexpr (E16_SKIP & 32) {
                if (val.x == 1.2345e-300) Lst.evec[0] = val;     // (keeps the value alive)
            } else {
                *outp = val;
                outp += bstride;
            }
        }
    }
}
