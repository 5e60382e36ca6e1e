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

template <int MODE, int NM>
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
    const uint2 mt = W.meta[slot];                       // {split mask (bit i: e_i negligible in T; bit n-1 set), flagged}
    const unsigned split = mt.x | (n < 32 ? ~0u << n : 0u);
    {
        double2 t{0.0, 0.0};
        int rk = j;
        if (j < n) {
            t = W.de[(int64_t)j * nchunk + slot];
            rk = W.rank[(int64_t)j * nchunk + slot];
        }
        if ((split >> j) & 1u) t.y = 0.0;
        Xd[mat * 32 + j] = t;
        Rk[mat * 32 + j] = rk;
    }
    const double lam = j < n ? W.lam[(int64_t)j * nchunk + slot] : 0.0;
    TW_LDS_ORDER();
    bool bad = false;
    {
        const double2* __restrict__ xd = Xd + mat * 32;
        double lp[NM - 1], um[NM - 1];
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
                um[k] = b.y * tw_rcp(tw_guard(dm));
                dm = fma(-b.y, um[k], b.x - lam);
            }
        }
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
                if (k < NM - 1) gk = fma(-a.y, um[k], gk);
                eprev = a.y;
                const bool in = k >= bl && k <= bh;
                if (in && fabs(gk) < gmin) {
                    gmin = fabs(gk);
                    gam_r = gk;
                    r = k;
                }
            }
        }
        double v[NM];
#pragma unroll
        for (int i = 0; i < NM; ++i) v[i] = i == r ? 1.0 : 0.0;
#pragma unroll
        for (int i = NM - 2; i >= 0; --i) v[i] = i < r ? -lp[i] * v[i + 1] : v[i];
#pragma unroll
        for (int i = 0; i < NM - 1; ++i) v[i + 1] = i >= r ? -um[i] * v[i] : v[i + 1];
        double nz2 = 0.0;
#pragma unroll
        for (int i = 0; i < NM; ++i) nz2 = fma(v[i], v[i], nz2);
        const double inz = rsqrt_full(nz2);
        const double tn = tnorm + fabs(lam);
        bad = !(fabs(gam_r) * inz <= 1e-13 * tn) && j < n;
        double* mine = Vs + (mat * 32 + j) * TW32_LD;
        if (j < NM) {
#pragma unroll
            for (int i = 0; i < NM; i += 2) *reinterpret_cast<double2*>(mine + i) = double2{v[i] * inz, v[i + 1] * inz};
        }
#pragma unroll
        for (int i = 0; i < 32; i += 2)                   // rows past NM; the whole column of a lane past NM (a unit vector: V stays orthogonal)
            if (i >= NM || j >= NM) *reinterpret_cast<double2*>(mine + i) = double2{i == j ? 1.0 : 0.0, i + 1 == j ? 1.0 : 0.0};
    }
    // a matrix one of whose vectors failed the residual test joins the list (once; not if the QL kernel listed it already)
    const unsigned long long bal = __ballot(bad && live);
    bool skip_m[2];
    {
        const unsigned m0 = (unsigned)bal, m1 = (unsigned)(bal >> 32);
        const unsigned fl0 = (unsigned)__builtin_amdgcn_readlane((int)mt.y, 0), fl1 = (unsigned)__builtin_amdgcn_readlane((int)mt.y, 32);
        if (j == 0 && live && (mat ? m1 : m0) != 0 && mt.y == 0) W.list[atomicAdd(W.count, 1)] = (int)slot;
        skip_m[0] = m0 != 0 || fl0 != 0;
        skip_m[1] = m1 != 0 || fl1 != 0 || slot0 + 1 >= nchunk;
    }
    TW_LDS_ORDER();

    const int j16 = lane & 15, g = lane >> 4;
    constexpr int KS = NM / 4;                            // k-steps of 4 over the NM rows / columns that can be real
    constexpr int NT = NM / 8;                            // 16-double tiles of a row of Q (2 NM doubles)
#pragma unroll 1
    for (int m2 = 0; m2 < 2; ++m2) {
        if (skip_m[m2]) continue;                         // (wave-uniform)
        const int64_t id = id0 + slot0 + m2;
        // Q of this matrix, straight into the B operands of (c): row b = 4 kk + g, doubles 16 tn + j16
        double qb[NT][KS];
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
        double* const Vm = Vs + m2 * 32 * TW32_LD;
        // ---- (b) Newton-Schulz: X = 1.5 I - 0.5 V^T V, V <- V X
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
        // ---- (c) Z^T = V^T Q^T: rows = vectors (16 tm + g + 4 r), columns = the 2 n doubles of an output row
        const int* rk = Rk + m2 * 32;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            if (16 * tm >= n) continue;
            double av[KS];
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) av[kk] = Vm[(16 * tm + j16) * TW32_LD + 4 * kk + g];   // A[a][k] = V[k][16 tm + a]
            double* dst[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int vec = 16 * tm + g + 4 * r;
                const int rr = rk[vec];
                if constexpr (MODE == 1) dst[r] = reinterpret_cast<double*>(wf_at(wv, vec < n ? rr : 0, id));
                else dst[r] = reinterpret_cast<double*>(evec + ((int64_t)(vec < n ? rr : 0) * nk + id) * n);
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
    }
}
