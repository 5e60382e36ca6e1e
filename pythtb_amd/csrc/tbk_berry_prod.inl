// tbk_berry_prod.inl -- included by tbk_berry.hip.  det-type Berry phase of 5..8 bands of wide states (config E:
// berry_phase(range(8), dir=2) on 16 components) WITHOUT the link-matrix workspace (round 4).
//
// The reference multiplies the overlap matrices of a string and takes ONE determinant at the end (pythtb.py:3813-3831:
// `prd = np.dot(prd, ovr_m)` over the links, then `det(prd)`).  Round 2/3 took the determinant of every link instead (det of a
// product = product of dets) with a second kernel that needs a whole link matrix per lane: k_chain_links_tile wrote nocc^2 c128
// per link to a <= 1 GiB workspace and k_chain_lu_wave read it back -- 2.0 x the algorithmic bytes of the pair.
//
// Here the ordered product P = M_1 M_2 ... of a (string, segment) stays in the wavefront that forms the link matrices, on the
// MATRIX CORES: a complex nocc x nocc matrix is a real 16 x 16 one (every entry re + i im the 2 x 2 block [[re, -im], [im, re]];
// nocc < 8 padded with the identity), and P^T <- M^T P^T is one 16 x 16 x 16 real product = four v_mfma_f64_16x16x4_f64 whose
// B operand is the accumulator of the previous product as it stands (lane l holds D[(l >> 4) + 4 r][l & 15] in register r,
// which is B[4 kb + (l >> 4)][l & 15] for kb = r: profiles/microbench/mfma_f64_layout.hip).  The link matrices leave the
// vector ALU through 2 KB of LDS each.  One nocc x nocc matrix per (string, segment) goes out (1 KB for 256 links) and
// k_chain_prod_det takes its determinant.  The vector ALU does exactly what k_chain_links_tile did.  (Round 4 counted the matrix
// cores as idle capacity; round 5 measured a v_mfma_f64_16x16x4 at 64 cycles of the SIMD's own fp64 pipe -- the vector rate,
// profiles/microbench/valu_rates.hip.  The four per link are 18 % of this kernel's pipe time, affordable for a product that needs
// no data movement between lanes; the polar form below, 40 per link, was cut to 28.)
#define TBK_CHAINP_G 4   // links per wavefront step (sixteen lanes per link)
// row stride (doubles) of a link's 16 x 16 real image in LDS: 16 for the plain product (its one read pattern, rows of 16 lanes, is
// conflict-free); 18 with polar factors, whose iteration also reads the image and its own iterate TRANSPOSED (stride-16 columns put
// the 16 lanes of a row on two banks: 43 % of that kernel's LDS cycles were conflicts, profiles/r05ecfg)
#define TBK_CHAINP_LD(POLAR) ((POLAR) ? 18 : 16)
// POLAR (Wilson-loop eigenphases of 5..8 wide bands, round 4): every link matrix is replaced by its polar factor U = M (M^+ M)^(-1/2)
// (the reference's U Vh of svd(M), pythtb.py:3820-3826) before it is multiplied on -- by the Newton-Schulz iteration
// X <- X (3 I - X^T X) / 2 on the real image, ON THE MATRIX CORES: with X and X^T both held in the accumulator layout every operand
// of the two products of a step, Y = X^T X and X Y, is a register as it stands (Y is symmetric bit for bit, so its accumulator
// doubles as its own A operand): 8 v_mfma_f64_16x16x4_f64 per step; X^T of the new iterate comes back through the link's own LDS
// image (round 4 kept it current with a third product, Y X^T: 12 per step, no LDS).  The loop ends when ||Y - I||_F^2 < 1e-14
// (wave-uniform: one link at a time), like k_link_polar_big; a link that does not converge in 200 steps raises the singular-link
// status.  Output: the unitary product of the (string, segment), nocc x nocc compact at pw[(string * nseg + segment) * nocc^2],
// the layout the product tree / Cayley tail of the Wilson pipeline reads (tbk_berry_big.inl).
template <int NOCC, int NLD, bool POLAR = false>
__global__ __launch_bounds__(128) void k_chain_prod_tile(const ChainArgs A, const int64_t s0, const int64_t ns, cd* __restrict__ pw) {
    static_assert(NOCC >= 3 && NOCC <= 8 && (POLAR || NOCC >= 5), "k_chain_prod_tile: 5..8 bands (3..8 with polar factors)");
    extern __shared__ __align__(16) unsigned char chainw_lds[];
    constexpr int G = TBK_CHAINP_G, NT = (NOCC + 1) / 2, LD = TBK_CHAINP_LD(POLAR), IMG = 16 * LD;
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + wib;
    if (t >= ns * A.nseg) return;                    // (no workgroup barrier below: waves are independent)
    const int64_t seg = t / ns, sl = t - seg * ns, s = s0 + sl;
    const int ncomp = A.v.ncomp;
    const int ldp = ncomp + 1;
    const int pbuf = NOCC * ldp + 1;
    // nocc = 8: every entry of the real images is written at every step, so they can lie OVER the staged points (which are dead
    // once the overlaps are in registers): 10.7 instead of 18.7 KB per wavefront, 14 instead of 8 wavefronts per CU -- the kernel
    // waits on HBM latency, not on the vector ALU (43 % busy at two wavefronts per SIMD)
    constexpr bool ALIAS = NOCC == 8;
    const size_t pts_bytes = (size_t)(G + 1) * pbuf * sizeof(cd), img_bytes = (size_t)G * IMG * sizeof(double);
    const size_t wave_bytes = ALIAS ? (pts_bytes > img_bytes ? pts_bytes : img_bytes) : pts_bytes + img_bytes;
    cd* const buf = reinterpret_cast<cd*>(chainw_lds + (size_t)wib * wave_bytes);        // slots 0 .. G: points i .. i + G
    double* const Me = ALIAS ? reinterpret_cast<double*>(buf)                            // [G][16][16]: the links' real images
                             : reinterpret_cast<double*>(buf + (size_t)(G + 1) * pbuf);
    const int64_t plane = A.v.npts * ncomp;
    const int i0 = (int)seg * A.seg_len;
    const int i1 = min(i0 + A.seg_len, A.nlinks);
    const int np = i1 - i0;                          // links of this segment; its points are 0 .. np
    const cd* P = A.v.data + (axis_offset(A.other, s) + (int64_t)i0 * A.sdir) * ncomp;
    const int64_t step = A.sdir * ncomp;
    const int nel = NOCC * ncomp;
    int64_t goff[NLD];
    int dst[NLD];
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int e = j * 64 + lane;
        const bool ok = e < nel;
        const int a = ok ? e / ncomp : 0, c = ok ? e - a * ncomp : 0;
        goff[j] = (int64_t)A.occ[a] * plane + c;
        dst[j] = ok ? a * ldp + c : NOCC * ldp;
    }
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef double v4d __attribute__((ext_vector_type(4)));
    v2d x[G][NLD], xl[NLD];
    auto load_group = [&](const int first) {         // points first .. first + G - 1 of the segment (clamped to its last point)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const cd* p = P + (int64_t)min(first + g, np) * step;
#pragma unroll
            for (int j = 0; j < NLD; ++j) x[g][j] = *reinterpret_cast<const v2d*>(p + goff[j]);
        }
    };
#pragma unroll
    for (int j = 0; j < NLD; ++j) xl[j] = *reinterpret_cast<const v2d*>(P + goff[j]);   // point 0
    load_group(1);
    // the real images start as the identity (the padding of nocc < 8 stays that way: nobody writes there)
    if constexpr (!ALIAS) {
#pragma unroll
        for (int q = 0; q < G * 256 / 64; ++q) {
            const int e = q * 64 + lane, rc = e & 255;
            Me[(e >> 8) * IMG + (rc >> 4) * LD + (rc & 15)] = (rc >> 4) == (rc & 15) ? 1.0 : 0.0;
        }
    }
    // this lane's block of M (as in k_chain_links_tile)
    const int g4 = lane >> 4, tl = lane & 15;
    const int ta = tl / NT, tb = tl - ta * NT;
    const bool active = tl < NT * NT;
    const int a0 = min(2 * ta, NOCC - 1), a1 = min(2 * ta + 1, NOCC - 1), b0 = min(2 * tb, NOCC - 1), b1 = min(2 * tb + 1, NOCC - 1);
    const bool va1 = 2 * ta + 1 < NOCC, vb1 = 2 * tb + 1 < NOCC;
    // P^T as a 16 x 16 real matrix in the accumulator layout: the identity
    const int gl = lane >> 4, cl = lane & 15;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (gl + 4 * r) == cl ? 1.0 : 0.0;
    bool singular = false;
    for (int i = 0; i < np; i += G) {
        // points i (kept from the previous step) and i + 1 .. i + G (arriving) go to LDS; the registers take the next group
#pragma unroll
        for (int j = 0; j < NLD; ++j) *reinterpret_cast<v2d*>(buf + dst[j]) = xl[j];
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
#pragma unroll
            for (int j = 0; j < NLD; ++j) *reinterpret_cast<v2d*>(buf + (gg + 1) * pbuf + dst[j]) = x[gg][j];
#pragma unroll
        for (int j = 0; j < NLD; ++j) xl[j] = x[G - 1][j];
        load_group(i + G + 1);
        lds_sync_wave();
        const bool mine = active && i + g4 < np;
        cd m00 = cd{0.0, 0.0}, m01 = m00, m10 = m00, m11 = m00;
        if (mine) {
            const int g = g4;
            const cd* pa = buf + g * pbuf;
            const cd* pb = pa + pbuf;
            const cd *ua0 = pa + a0 * ldp, *ua1 = pa + a1 * ldp, *ub0 = pb + b0 * ldp, *ub1 = pb + b1 * ldp;
            m00 = cmulc(ua0[0], ub0[0]);
            m01 = cmulc(ua0[0], ub1[0]);
            m10 = cmulc(ua1[0], ub0[0]);
            m11 = cmulc(ua1[0], ub1[0]);
            cd n00 = cmulc(ua0[1], ub0[1]), n01 = cmulc(ua0[1], ub1[1]), n10 = cmulc(ua1[1], ub0[1]), n11 = cmulc(ua1[1], ub1[1]);
            int c = 2;
            for (; c + 1 < ncomp; c += 2) {
                const cd p0 = ua0[c], p1 = ua1[c], q0 = ub0[c], q1 = ub1[c];
                cfmac(m00, p0, q0);
                cfmac(m01, p0, q1);
                cfmac(m10, p1, q0);
                cfmac(m11, p1, q1);
                const cd r0 = ua0[c + 1], r1 = ua1[c + 1], t0 = ub0[c + 1], t1 = ub1[c + 1];
                cfmac(n00, r0, t0);
                cfmac(n01, r0, t1);
                cfmac(n10, r1, t0);
                cfmac(n11, r1, t1);
            }
            if (c < ncomp) {
                const cd p0 = ua0[c], p1 = ua1[c], q0 = ub0[c], q1 = ub1[c];
                cfmac(m00, p0, q0);
                cfmac(m01, p0, q1);
                cfmac(m10, p1, q0);
                cfmac(m11, p1, q1);
            }
            m00 = cadd(m00, n00);
            m01 = cadd(m01, n01);
            m10 = cadd(m10, n10);
            m11 = cadd(m11, n11);
        }
        if constexpr (ALIAS) lds_sync_wave();        // every lane has read its points before the images overwrite them
        if (mine) {
            // entry (a, b) = re + i im  ->  rows 2a, 2a + 1 x columns 2b, 2b + 1 of the link's real image: [[re, -im], [im, re]]
            double* const me = Me + g4 * IMG;
            {
                double* r0 = me + (2 * a0) * LD + 2 * b0;
                *reinterpret_cast<v2d*>(r0) = v2d{m00.x, -m00.y};
                *reinterpret_cast<v2d*>(r0 + LD) = v2d{m00.y, m00.x};
                if (vb1) {
                    *reinterpret_cast<v2d*>(r0 + 2) = v2d{m01.x, -m01.y};
                    *reinterpret_cast<v2d*>(r0 + LD + 2) = v2d{m01.y, m01.x};
                }
            }
            if (va1) {
                double* r1 = me + (2 * a1) * LD + 2 * b0;
                *reinterpret_cast<v2d*>(r1) = v2d{m10.x, -m10.y};
                *reinterpret_cast<v2d*>(r1 + LD) = v2d{m10.y, m10.x};
                if (vb1) {
                    *reinterpret_cast<v2d*>(r1 + 2) = v2d{m11.x, -m11.y};
                    *reinterpret_cast<v2d*>(r1 + LD + 2) = v2d{m11.y, m11.x};
                }
            }
        }
        lds_sync_wave();
        // P^T <- M_g^T P^T for the links of this step, in order: A operand = M^T, i.e. lane (gl, cl) supplies
        // M_img[4 kb + gl][cl]; B operand = the old accumulator's register kb
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (i + g < np) {                        // (wave-uniform)
                double* me = Me + g * IMG;
                if constexpr (!POLAR) {
                    v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) nw = __builtin_amdgcn_mfma_f64_16x16x4f64(me[(4 * kb + gl) * LD + cl], acc[kb], nw, 0, 0, 0);
                    acc = nw;
                } else {
                    v4d X, Xt;                       // X[4 kb + gl][cl] and X^T[4 kb + gl][cl] = X[cl][4 kb + gl]
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) {
                        X[kb] = me[(4 * kb + gl) * LD + cl];
                        Xt[kb] = me[cl * LD + 4 * kb + gl];
                    }
                    bool ok = false;
                    for (int it = 0; it < 200; ++it) {
                        v4d Y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) Y = __builtin_amdgcn_mfma_f64_16x16x4f64(X[kb], X[kb], Y, 0, 0, 0);     // X^T X
                        double r2 = 0.0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double dv = Y[r] - ((gl + 4 * r) == cl ? 1.0 : 0.0);
                            r2 = fma(dv, dv, r2);
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) r2 += __shfl_xor(r2, off);
                        v4d XY = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) XY = __builtin_amdgcn_mfma_f64_16x16x4f64(Xt[kb], Y[kb], XY, 0, 0, 0);    // X Y
#pragma unroll
                        for (int r = 0; r < 4; ++r) X[r] = fma(1.5, X[r], -0.5 * XY[r]);
                        // X^T of the new iterate through the link's own image slot (X came from there; nobody reads it again): a third
                        // product Y X^T kept it current in round 4 -- on gfx950 four more v_mfma_f64 are 256 cycles of the fp64 pipe, the
                        // eight LDS accesses of a transposition none
                        if (r2 < 1e-14) {            // residual 1e-7 before this update, its square after it
                            ok = true;
                            break;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) me[(4 * r + gl) * LD + cl] = X[r];
                        lds_sync_wave();
#pragma unroll
                        for (int kb = 0; kb < 4; ++kb) Xt[kb] = me[cl * LD + 4 * kb + gl];
                        lds_sync_wave();
                    }
                    singular = singular || !ok;
                    v4d nw = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) nw = __builtin_amdgcn_mfma_f64_16x16x4f64(X[kb], acc[kb], nw, 0, 0, 0);
                    acc = nw;
                }
            }
        }
        lds_sync_wave();
    }
    if constexpr (POLAR) {
        if (singular && lane == 0) atomicExch(A.flags + 1, 1);
        if ((gl & 1) == 0) {
            double* const o = reinterpret_cast<double*>(pw + ((size_t)sl * A.nseg + seg) * (NOCC * NOCC));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = (gl + 4 * r) >> 1, a = cl >> 1;
                if (a < NOCC && b < NOCC) o[(a * NOCC + b) * 2 + (cl & 1)] = acc[r];
            }
        }
        return;
    }
    // P[a][b] = (P_img[2a][2b], P_img[2a + 1][2b]); register r of lane (gl, cl) holds P^T_img[gl + 4 r][cl] = P_img[cl][gl + 4 r]
    if ((gl & 1) == 0) {
        double* const o = reinterpret_cast<double*>(pw + t * 64);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int b = (gl + 4 * r) >> 1, a = cl >> 1;
            o[(a * 8 + b) * 2 + (cl & 1)] = acc[r];
        }
    }
}

// determinant of every (string, segment)'s product -> A.partial (what k_chain_lu_wave left there: the segment's factor of det P)
template <int NOCC>
__global__ __launch_bounds__(64) void k_chain_prod_det(const ChainArgs A, const int64_t s0, const int64_t ns, const cd* __restrict__ pw) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= ns * A.nseg) return;
    const int64_t seg = t / ns, sl = t - seg * ns, s = s0 + sl;
    cd M[NOCC][NOCC];
#pragma unroll
    for (int a = 0; a < NOCC; ++a)
#pragma unroll
        for (int b = 0; b < NOCC; ++b) M[a][b] = pw[t * 64 + a * 8 + b];
    A.partial[seg * A.nstrings + s] = det_small<NOCC>(M);
}
