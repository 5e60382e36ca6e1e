// tbk_solve_tw32.inl -- included by tbk_solve_qlw.inl before launch_qlw (round 6).
//
// n = 17..32 states per k WITH eigenvectors: what replaces the replay of the recorded QL rotations on Z (k_ql_replay_reg: ~0.85 n^2
// rotations of two columns of Z each, every sweep visiting every position under a predicate -- 0.43 / 0.67 / 1.28 ms per 36 k
// matrices of n = 17 / 24 / 32, a third of the call).  The same cut k_tw16_vectors / k_e16 make for 9..16 states, on 32 lanes per
// matrix (pythtb.py:939-947 -- numpy.linalg.eigh -- is what all of this restates):
//
//   (a) lane j of a matrix: the eigenvector of T for the eigenvalue the QL kernel left at POSITION j, from the twisted
//       factorisation of T - lambda_j inside the unreduced block of T that holds j (d, e streamed from LDS as broadcasts; the two
//       pivot chains interleaved) -- O(n) per vector, residual |gamma_r| / |z| checked against 1e-13 |T|;
//   (b) one Newton-Schulz step V <- V (3 I - V^T V) / 2 on the real 32 x 32 matrix of a matrix's vectors (LDS image, column stride
//       34 doubles) with v_mfma_f64_16x16x4_f64: 2 x 2 tiles of G = V^T V, then 2 x 2 tiles of V X;
//   (c) Z^T = V^T Q^T on the matrix cores as well: A operand = the LDS image of V, B operand = Q as k_hh32 / k_tridiag_lds left it in
//       the output array (row b = column b of Q with the orbital phases on it, 2 n doubles), read from HBM ONCE straight into the B
//       registers (the lanes of a 16-lane row read 128 contiguous bytes), the accumulators stored as 128-byte runs of the output
//       rows in ascending order of the eigenvalues.  No Z in LDS, no row of Z in registers.
//
// Matrices with two eigenvalues of one unreduced block closer than gaptol |T| (flagged by the QL kernel) or with a vector that
// fails the residual test are left exactly as they were (Q in the output array) and go on a list; k_ql_replay_reg then replays the
// recorded rotations for the listed matrices only.  The decision depends on the matrix alone: periodic images, halo rows and shard
// windows stay bit-identical.  TBK_TW32=0 restores the replay for every matrix.

#define TW32_LD 34            // doubles per column of the V image (even: double2 stores; 2-way bank conflicts at most)
#define TW32_WAVE_LDS (2 * 32 * TW32_LD * 8 + 2 * 32 * 16 + 2 * 32 * 4)

// y <- H_K y for K = K0 .. 0, H_K = I - us_K us_K^+, for ONE matrix on the whole wavefront: lane (j, h) holds rows 2 i + h of vector j
// (the rows dealt in turn, like the columns in k_hh32: the live ones stay evenly split); the entries of us_K are broadcast reads of the
// staged record (two addresses per read), the halves' partial sums meet in one cross-half exchange.
template <int K, int NM>
__device__ __forceinline__ void tw32_reflect(cd (&y)[NM / 2], const cd* R, const int h) {
    constexpr int off = hh32_rec_off(K, NM);
    TW_LDS_ORDER();                                       // (keeps the reads of later reflectors from being hoisted: registers)
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) asm("" : "+v"(y[i].x), "+v"(y[i].y));   // (opaque per step: see k_hh32 on InstCombine)
    cd w{0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) {
        if (2 * i + 1 > K) {                              // row 2 i + h > K in at least one half
            cd u = R[off + (2 * i - K - 1 >= 0 ? 2 * i - K - 1 + h : 0)];
            if (2 * i <= K && h == 0) u = cd{0.0, 0.0};  // (row 2 i = K: no entry)
            // w += conj(u_x) y_x
            w.x = fma(u.x, y[i].x, w.x);
            w.x = fma(u.y, y[i].y, w.x);
            w.y = fma(u.x, y[i].y, w.y);
            w.y = fma(-u.y, y[i].x, w.y);
        }
    }
    w = cd{w.x + hh32_xhalf(w.x, h != 0), w.y + hh32_xhalf(w.y, h != 0)};
#pragma unroll
    for (int i = 0; i < NM / 2; ++i) {
        if (2 * i + 1 > K) {
            cd u = R[off + (2 * i - K - 1 >= 0 ? 2 * i - K - 1 + h : 0)];
            if (2 * i <= K && h == 0) u = cd{0.0, 0.0};
            // y_x -= u_x w
            y[i].x = fma(-u.x, w.x, y[i].x);
            y[i].x = fma(u.y, w.y, y[i].x);
            y[i].y = fma(-u.x, w.y, y[i].y);
            y[i].y = fma(-u.y, w.x, y[i].y);
        }
    }
    if constexpr (K > 0) tw32_reflect<K - 1, NM>(y, R, h);
}

// REFL: stage 1 left the reflector record (k_hh32<.., 2, NM>) instead of Q: (c) becomes z_j = H_0 (H_1 (.. H_{n-3} (D v_j))) with lane j
// holding its vector (the form k_tw16_vectors has), transposed through LDS on the way out; listed matrices get V = I, i.e. Q in the
// output array for the replay.
template <int MODE, int NM, bool REFL>
__global__ __launch_bounds__(64) void k_tw32_vectors(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk, const QlwWork W,
                                                     cd* evec, const WfsView wv) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[TW32_WAVE_LDS];
    double* const Vs = reinterpret_cast<double*>(lds_all);                                  // [2][32][TW32_LD]
    double2* const Xd = reinterpret_cast<double2*>(lds_all + 2 * 32 * TW32_LD * 8);          // [2][32] (d_i, e_i | 0 at a split)
    int* const Rk = reinterpret_cast<int*>(lds_all + 2 * 32 * TW32_LD * 8 + 2 * 32 * 16);   // [2][32] ascending rank of position j
    const int lane = threadIdx.x, mat = lane >> 5, j = lane & 31;
    const int64_t slot0 = (int64_t)blockIdx.x * 2;
    const int64_t slot_u = slot0 + mat;
    const bool live = slot_u < nchunk;
    const int64_t slot = live ? slot_u : nchunk - 1;

    // ---- (a) twisted factorisation of T - lambda_j within the unreduced block of position j
    // (W.all_q: only Q is wanted -- every matrix of the model goes through the rotation replay, ModelView::pairs_hint; stages (a), (b)
    // are skipped and V = I in position order, as for a listed matrix)
    const bool allq = REFL && W.all_q != 0;
    uint2 mt{~0u, 0u};
    if (!allq) mt = W.meta[slot];                        // {split mask (bit i: e_i negligible in T; bit n-1 set), flagged}
    const unsigned split = mt.x | (n < 32 ? ~0u << n : 0u);
    {
        double2 t{0.0, 0.0};
        int rk = j;
        if (j < n && !allq) {
            t = W.de[(int64_t)j * nchunk + slot];
            rk = W.rank[(int64_t)j * nchunk + slot];
        }
        if ((split >> j) & 1u) t.y = 0.0;
        Xd[mat * 32 + j] = t;
        Rk[mat * 32 + j] = rk;
    }
    const double lam = j < n && !allq ? W.lam[(int64_t)j * nchunk + slot] : 0.0;
    TW_LDS_ORDER();
    bool bad = false;
    double lamv = lam;
    // (a second pass with the Rayleigh-quotient correction lambda + gamma_r / |z|^2 when a vector of the wavefront fails the residual
    // test: a listed matrix costs a whole lane-per-matrix QL chain with its record, ~0.15 ms per chunk whatever their number)
    if (!allq)
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const double lam = lamv;
        const double2* xd = Xd + mat * 32;
        // lp in registers; um_k goes to slot k + 1 of this lane's column of the V image -- where z_{k+1} = -um_k z_k will stand
        // (lp AND um in registers, then z: 218 VGPRs, one wavefront per SIMD)
        double* const mine = Vs + (mat * 32 + j) * TW32_LD;
        constexpr bool UM_LDS = NM > 24;                   // (NM = 24 fits two wavefronts per SIMD with um in registers: 207 against 225 us per 36 k matrices)
        double lp[NM - 1], umr[UM_LDS ? 1 : NM - 1];
        auto um_at = [&](const int k) { return UM_LDS ? mine[k + 1] : umr[UM_LDS ? 0 : k]; };
        double tnorm = 0.0;
        {
            double dp = xd[0].x - lam, dm = xd[NM - 1].x - lam;
#pragma unroll
            for (int i = 0; i < NM - 1; ++i) {
                const double2 a = xd[i], an = xd[i + 1];
                lp[i] = a.y * tw_rcp(tw_guard(dp));
                dp = fma(-a.y, lp[i], an.x - lam);
                const int k = NM - 2 - i;
                const double2 b = xd[k];
                const double um = b.y * tw_rcp(tw_guard(dm));
                if constexpr (UM_LDS) mine[k + 1] = um;
                else umr[k] = um;
                dm = fma(-b.y, um, b.x - lam);
            }
        }
        TW_LDS_ORDER();
        const unsigned below_j = split & ((1u << j) - 1u);
        const int bl = below_j ? 32 - __builtin_clz(below_j) : 0;           // first position of the block
        const int bh = __builtin_ctz(split >> j) + j;                       // last position of the block
        double gmin = INFINITY, gam_r = 0.0;
        int r = j;
        {
            double eprev = 0.0;
#pragma unroll
            for (int k = 0; k < NM; ++k) {
                const double2 a = xd[k];
                const double s = a.x - lam;
                tnorm = fmax(tnorm, fabs(s));
                double gk = s;
                if (k > 0) gk = fma(-eprev, lp[k - 1], gk);
                if (k < NM - 1) gk = fma(-a.y, um_at(k), gk);
                eprev = a.y;
                const bool in = k >= bl && k <= bh;
                if (in && fabs(gk) < gmin) {
                    gmin = fabs(gk);
                    gam_r = gk;
                    r = k;
                }
            }
        }
        TW_LDS_ORDER();
        // z_r = 1; downwards z_i = -lp_i z_{i+1} (i < r) into slot i; upwards z_{i+1} = -um_i z_i (i >= r) in place
        double nz2 = 1.0;
        {
            double zn = 1.0;
#pragma unroll
            for (int i = NM - 2; i >= 0; --i) {
                if (i < r) {
                    zn = -lp[i] * zn;
                    mine[i] = zn;
                    nz2 = fma(zn, zn, nz2);
                }
            }
            mine[r] = 1.0;
            TW_LDS_ORDER();
            double zp = 1.0;
#pragma unroll
            for (int i = 0; i < NM - 1; ++i) {
                if (i >= r) {
                    zp = -um_at(i) * zp;
                    mine[i + 1] = zp;
                    nz2 = fma(zp, zp, nz2);
                }
            }
        }
        TW_LDS_ORDER();
        const double inz = rsqrt_full(nz2);
        const double tn = tnorm + fabs(lam);
        bad = !(fabs(gam_r) * inz <= 1e-13 * tn) && j < n;
        if (pass == 0 && __any(bad && live && mt.y == 0)) {
            lamv = bad ? fma(gam_r * inz, inz, lam) : lam;
            continue;
        }
        if (j < NM) {
#pragma unroll
            for (int i = 0; i < NM; i += 2) {
                const double2 t = *reinterpret_cast<const double2*>(mine + i);
                *reinterpret_cast<double2*>(mine + i) = double2{t.x * inz, t.y * inz};
            }
        }
#pragma unroll
        for (int i = 0; i < 32; i += 2)                   // rows past NM; the whole column of a lane past NM (a unit vector: V stays orthogonal)
            if (i >= NM || j >= NM) *reinterpret_cast<double2*>(mine + i) = double2{i == j ? 1.0 : 0.0, i + 1 == j ? 1.0 : 0.0};
        break;
    }
    // a matrix one of whose vectors failed the residual test joins the list (once; not if the QL kernel listed it already)
    const unsigned long long bal = __ballot(bad && live);
    bool skip_m[2];
    {
        const unsigned m0 = (unsigned)bal, m1 = (unsigned)(bal >> 32);
        const unsigned fl0 = (unsigned)__builtin_amdgcn_readlane((int)mt.y, 0), fl1 = (unsigned)__builtin_amdgcn_readlane((int)mt.y, 32);
        if (j == 0 && live && (mat ? m1 : m0) != 0 && mt.y == 0) W.list[atomicAdd(W.count, 1)] = (int)slot;
        skip_m[0] = m0 != 0 || fl0 != 0 || allq;
        skip_m[1] = m1 != 0 || fl1 != 0 || slot0 + 1 >= nchunk || allq;
    }
    constexpr int RSZ = hh32_rec_size(NM), NL = (RSZ + 63) / 64;
    if constexpr (REFL) {
        // a listed matrix: V = I in position order (Q itself goes out, the replay takes it from there)
        if (skip_m[mat]) {
            double* const mine = Vs + (mat * 32 + j) * TW32_LD;
#pragma unroll
            for (int i = 0; i < 32; i += 2) *reinterpret_cast<double2*>(mine + i) = double2{i == j ? 1.0 : 0.0, i + 1 == j ? 1.0 : 0.0};
            Rk[mat * 32 + j] = j;
        }
    }
    TW_LDS_ORDER();

    const int j16 = lane & 15, g = lane >> 4;
    constexpr int KS = NM / 4;                            // k-steps of 4 over the NM rows / columns that can be real
    constexpr int NT = (NM + 7) / 8;                      // 16-double tiles of a row of Q (2 NM doubles)
    // (REFL: matrix 1 first -- its record is staged over its own image and the (d, e) region behind it, then matrix 0's over image 0
    // and the start of image 1, which is dead by then)
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
        const int m2 = REFL ? 1 - mm : mm;
        if (REFL ? (m2 == 1 && slot0 + 1 >= nchunk) : skip_m[m2]) continue;   // (wave-uniform)
        const int64_t id = id0 + slot0 + m2;
        double* const Vm = Vs + m2 * 32 * TW32_LD;
        // Q of this matrix, straight into the B operands of (c): row b = 4 kk + g, doubles 16 tn + j16
        double qb[REFL ? 1 : NT][REFL ? 1 : KS];
        if constexpr (!REFL) {
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int b = 4 * kk + g, bb = b < n ? b : n - 1;
                const double* src;
                if constexpr (MODE == 1) src = reinterpret_cast<const double*>(wf_at(wv, bb, id));
                else src = reinterpret_cast<const double*>(evec + ((int64_t)bb * nk + id) * n);
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    const int col = 16 * tn + j16;
                    qb[tn][kk] = src[col < 2 * n ? col : 0];
                }
            }
        }
        // the reflector record of this matrix: in flight while the matrix cores work
        double2 rr[REFL ? NL : 1];
        if constexpr (REFL) {
            const double2* src = reinterpret_cast<const double2*>(W.refl + (slot0 + m2) * RSZ);
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const int idx = t * 64 + lane;
                rr[t] = idx < RSZ ? src[idx] : double2{0.0, 0.0};
            }
        }
        // ---- (b) Newton-Schulz: X = 1.5 I - 0.5 V^T V, V <- V X  (not for V = I)
        if (!(REFL && skip_m[m2])) {
            tw_d4 X[2][2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    tw_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        const double a = Vm[(16 * ti + j16) * TW32_LD + 4 * kk + g];     // A[a][k] = V[k][16 ti + a]
                        const double b = Vm[(16 * tj + j16) * TW32_LD + 4 * kk + g];     // B[k][b] = V[k][16 tj + b]
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) X[ti][tj][r] = fma(-0.5, acc[r], (16 * ti + g + 4 * r) == (16 * tj + j16) ? 1.5 : 0.0);
                }
            tw_d4 Vn[2][2];
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) {
                    tw_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        const double a = Vm[(4 * kk + g) * TW32_LD + 16 * tx + j16];     // A[x][k] = V[16 tx + x][i = 4 kk + g]
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[kk >> 2][tc][kk & 3], acc, 0, 0, 0);   // B[k][c] = X[4 kk + g][16 tc + c]
                    }
                    Vn[tx][tc] = acc;
                }
            TW_LDS_ORDER();
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Vm[(16 * tc + j16) * TW32_LD + 16 * tx + g + 4 * r] = Vn[tx][tc][r];
            TW_LDS_ORDER();
        }
        const int* rk = Rk + m2 * 32;
        if constexpr (!REFL) {
            // ---- (c) Z^T = V^T Q^T: rows = vectors (16 tm + g + 4 r), columns = the 2 n doubles of an output row
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                double av[KS];
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) av[kk] = Vm[(16 * tm + j16) * TW32_LD + 4 * kk + g];   // A[a][k] = V[k][16 tm + a]
                double* dst[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int vec = 16 * tm + g + 4 * r;
                    const int rr1 = rk[vec];
                    if constexpr (MODE == 1) dst[r] = reinterpret_cast<double*>(wf_at(wv, vec < n ? rr1 : 0, id));
                    else dst[r] = reinterpret_cast<double*>(evec + ((int64_t)(vec < n ? rr1 : 0) * nk + id) * n);
                }
#pragma unroll
                for (int tn = 0; tn < NT; ++tn) {
                    tw_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], qb[tn][kk], acc, 0, 0, 0);
                    const int col = 16 * tn + j16;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * tm + g + 4 * r < n && col < 2 * n) dst[r][col] = acc[r];
                }
            }
            TW_LDS_ORDER();
        } else {
            // ---- (c') the whole wavefront on this matrix: lane (j, h) takes rows 2 i + h of vector j out of the image
            const int h = mat;
            cd y[NM / 2];
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) y[i].x = Vm[j * TW32_LD + 2 * i + h];
            TW_LDS_ORDER();
            // the record: over this matrix's image and what lies behind it
            cd* const R = reinterpret_cast<cd*>(Vm);
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const int idx = t * 64 + lane;
                if (idx < RSZ) R[idx] = cd{rr[t].x, rr[t].y};
            }
            TW_LDS_ORDER();
            constexpr int P = hh32_rec_off(NM - 2, NM);
            const cd eoc = R[P + NM + (j < NM ? j : 0)];  // the orbital phase of component j (this lane's after the transposition)
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) {
                const cd ph = R[P + 2 * i + h];
                const double vx = y[i].x;
                y[i] = cd{ph.x * vx, ph.y * vx};
            }
            tw32_reflect<NM - 3, NM>(y, R, h);
            TW_LDS_ORDER();
            // ---- (d) transpose through LDS (row stride 33 doubles): lane (c, h) receives component c of the vectors 2 i + h
            double* const Ts = reinterpret_cast<double*>(Vm);                           // [32][33]
            cd zt[NM / 2];
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) Ts[j * 33 + 2 * i + h] = y[i].x;
            TW_LDS_ORDER();
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) zt[i].x = Ts[(2 * i + h) * 33 + j];
            TW_LDS_ORDER();
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) Ts[j * 33 + 2 * i + h] = y[i].y;
            TW_LDS_ORDER();
#pragma unroll
            for (int i = 0; i < NM / 2; ++i) zt[i].y = Ts[(2 * i + h) * 33 + j];
            TW_LDS_ORDER();
            if (j < n) {
#pragma unroll
                for (int i = 0; i < NM / 2; ++i) {
                    const int b = 2 * i + h;
                    if (b < n) {
                        const int r = rk[b];
                        const cd val = cmul(zt[i], eoc);
                        if constexpr (MODE == 1) wf_at(wv, r, id)[j] = val;
                        else evec[((int64_t)r * nk + id) * n + j] = val;
                    }
                }
            }
        }
    }
}
