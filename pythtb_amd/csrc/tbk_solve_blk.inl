// tbk_solve_blk.inl -- included by tbk_solve.hip (after tbk_solve_big.inl and tbk_solve_row16.inl).
//
// Batches of wide matrices (ribbon / slab band structures: hundreds of k-points, n in the hundreds): two-sided
// BLOCK Jacobi.  The whole-chip solver of tbk_solve_big.inl streams every matrix through HBM once per round of
// n/2 scalar rotations, n - 1 rounds per sweep; here the index range is cut into blocks of 8, a round rotates
// nb/2 disjoint block pairs and there are nb - 1 = n/8 - 1 rounds per sweep -- 8x fewer passes over A and V.
//
//   per round   1. gather the 16x16 diagonal subproblem S = A[{I,J}][{I,J}] of every pair (I,J) of every matrix
//               2. diagonalise all of them at once: S = U diag(e) U^+   (k_solve_row16, registers + DPP)
//               3. A_out[{I,J}][{K,L}] = U_IJ^+  A_in[{I,J}][{K,L}]  U_KL  for all pairs of pairs (one 16x16 tile
//                  per workgroup, two 16^3 products through LDS), V^T[{I,J}][:] <- U_IJ^T V^T[{I,J}][:]
//
// The subproblem's eigenvectors must stay in Jacobi's own order (ListArgs::natural: U as close to the identity as
// its rotations allow).  Sorted eigenpairs make U permute, and with this position-based round-robin schedule the
// permutations conspire: the method stalls at a fixed off-norm after one sweep (measured, and reproduced in NumPy).
// n is padded to a multiple of 16 with decoupled diagonal entries above the spectrum (Frobenius
// norm + 1 + index): they are exact fixed points of every rotation and sort last in the final ordering.
// Initialisation, the off-norm test, the final sort and the output kernels are those of tbk_solve_big.inl.

constexpr int TBK_BLK = 8;

struct BlkArgs {
    BigWs W;
    int nbk;        // blocks per matrix (even)
    int npairs;     // nbk / 2 = ld / 16
    int64_t nsub;   // subproblems per round = matrices * npairs
    cd* S;          // [nsub][16][16]
    double* sev;    // [16][nsub]
    cd* U;          // [16][nsub][16]    U[i][j] of subproblem t at U[(j * nsub + t) * 16 + i]
};

__global__ __launch_bounds__(64) void k_blk_setpad(const BigWs W) {
    const int mat = blockIdx.x;
    cd* A = W.A0 + (size_t)mat * W.ld * W.ld;
    for (int x = W.n + threadIdx.x; x < W.ld; x += 64) A[(int64_t)x * W.ld + x] = cd{W.padval[mat] + (x - W.n), 0.0};
}

// global index of subproblem coordinate x of the pair (p, q)
__device__ __forceinline__ int blk_index(const int p, const int q, const int x) {
    return x < TBK_BLK ? p * TBK_BLK + x : q * TBK_BLK + x - TBK_BLK;
}

__global__ __launch_bounds__(256) void k_blk_gather(const BlkArgs B, const int round, const int par) {
    const int mat = blockIdx.y, l = blockIdx.x;
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    cd* S = B.S + ((size_t)mat * B.npairs + l) * 256;
    if (B.W.done[mat]) {      // converged: a trivial subproblem (its tiles are not applied)
        S[threadIdx.x] = cd{r == c ? (double)r : 0.0, 0.0};
        return;
    }
    const int ld = B.W.ld;
    const cd* A = (par ? B.W.A1 : B.W.A0) + (size_t)mat * ld * ld;
    int p, q;
    big_pair(l, round, B.nbk, p, q);
    S[threadIdx.x] = A[(int64_t)blk_index(p, q, r) * ld + blk_index(p, q, c)];
}

__global__ __launch_bounds__(256) void k_blk_apply(const BlkArgs B, const int round, const int par) {
    const int mat = blockIdx.z;
    if (B.W.done[mat]) return;
    __shared__ cd sU[2][16][17];
    __shared__ cd sT[16][17];
    __shared__ cd sX[16][17];
    const int ld = B.W.ld, np = B.npairs;
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const bool isA = (int)blockIdx.y < np;
    const int lr = isA ? (int)blockIdx.y : (int)blockIdx.y - np;
    const int64_t tr = (int64_t)mat * np + lr;
    int pr, qr;
    big_pair(lr, round, B.nbk, pr, qr);
    sU[0][r][c] = B.U[((int64_t)c * B.nsub + tr) * 16 + r];
    const int gr = blk_index(pr, qr, r);
    if (isA) {
        const int lc = blockIdx.x;
        const int64_t tc = (int64_t)mat * np + lc;
        int pc, qc;
        big_pair(lc, round, B.nbk, pc, qc);
        sU[1][r][c] = B.U[((int64_t)c * B.nsub + tc) * 16 + r];
        const int gc = blk_index(pc, qc, c);
        const cd* Ain = (par ? B.W.A1 : B.W.A0) + (size_t)mat * ld * ld;
        cd* Aout = (par ? B.W.A0 : B.W.A1) + (size_t)mat * ld * ld;
        sT[r][c] = Ain[(int64_t)gr * ld + gc];
        __syncthreads();
        cd x{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 16; ++k) cfma(x, sT[r][k], sU[1][k][c]);          // T U_KL
        sX[r][c] = x;
        __syncthreads();
        cd y{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 16; ++k) cfmac(y, sU[0][k][r], sX[k][c]);         // U_IJ^+ (T U_KL)
        if (lr == lc) y = cd{r == c ? B.sev[(int64_t)r * B.nsub + tr] : 0.0, 0.0};   // the subproblem itself: exactly diagonal
        Aout[(int64_t)gr * ld + gc] = y;
    } else {
        cd* Vt = B.W.Vt + (size_t)mat * ld * ld;
        const int col = (int)blockIdx.x * 16 + c;
        sT[r][c] = Vt[(int64_t)gr * ld + col];
        __syncthreads();
        cd y{0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 16; ++k) cfma(y, sU[0][k][r], sT[k][c]);          // V <- V U :  V^T rows <- U^T V^T
        Vt[(int64_t)gr * ld + col] = y;
    }
}

template <int MODE, bool VEC>
static int launch_blocked(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L, const GridArgs& G) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int ld = (n + 2 * TBK_BLK - 1) / (2 * TBK_BLK) * (2 * TBK_BLK);
    const int nbk = ld / TBK_BLK, npairs = nbk / 2;
    const size_t ll = (size_t)ld * ld;
    const int nbn = (int)std::min<size_t>(64, (ll + 256 * 16 - 1) / (256 * 16));
    const size_t sub = (size_t)npairs * (256 * sizeof(cd) * 2 + 16 * sizeof(double));     // S, U, sev per matrix
    const size_t per = 3 * al(ll * sizeof(cd)) + al((size_t)n * sizeof(cd)) + al((size_t)ld * sizeof(double)) +
                       al((size_t)ld * sizeof(int)) + al((size_t)nbn * 2 * sizeof(double)) + al(sub);
    size_t free_b = 0, total_b = 0;
    TBK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>(per, std::min<size_t>((size_t)8 << 30, (free_b + ctx->work_bytes) / 2));
    int64_t B = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nk, 32768), (int64_t)(budget / per)));
    if (tbk_knobs().big_batch >= 0) B = std::max<int64_t>(1, std::min<int64_t>(B, tbk_knobs().big_batch));   // test hook: matrices per batch
    const size_t wbytes = (size_t)B * per + al((size_t)B * sizeof(int)) * 2 + al((size_t)B * sizeof(double)) + 4096;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "eigen-solver workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    BlkArgs A{};
    BigWs& W = A.W;
    unsigned char* p = (unsigned char*)ctx->work;
    W.pending = (int*)p;
    p += 256;
    W.A0 = (cd*)p;
    p += (size_t)B * al(ll * sizeof(cd));
    W.A1 = (cd*)p;
    p += (size_t)B * al(ll * sizeof(cd));
    W.Vt = (cd*)p;
    p += (size_t)B * al(ll * sizeof(cd));
    W.eo = (cd*)p;
    p += al((size_t)B * n * sizeof(cd));
    W.ev = (double*)p;
    p += al((size_t)B * ld * sizeof(double));
    W.perm = (int*)p;
    p += al((size_t)B * ld * sizeof(int));
    W.partial = (double*)p;
    p += al((size_t)B * nbn * 2 * sizeof(double));
    W.done = (int*)p;
    p += al((size_t)B * sizeof(int));
    W.fpar = (int*)p;
    p += al((size_t)B * sizeof(int));
    W.padval = (double*)p;
    p += al((size_t)B * sizeof(double));
    A.S = (cd*)p;
    p += al((size_t)B * npairs * 256 * sizeof(cd));
    A.U = (cd*)p;
    p += al((size_t)B * npairs * 256 * sizeof(cd));
    A.sev = (double*)p;
    W.n = n;
    W.nbn = nbn;
    W.ld = ld;
    A.nbk = nbk;
    A.npairs = npairs;
    ModelView sub_mv{};
    sub_mv.nsta = 2 * TBK_BLK;
    sub_mv.nspin = 1;
    sub_mv.nslot = 2 * TBK_BLK * (2 * TBK_BLK + 1) / 2;
    const unsigned init_x = (unsigned)std::min<size_t>(1024, (ll + 255) / 256);
    for (int64_t base = 0; base < nk; base += B) {
        const int nb = (int)std::min<int64_t>(B, nk - base);
        A.nsub = (int64_t)nb * npairs;
        const ListArgs Ls{nullptr, A.S, A.sev, A.U, nullptr, 1};
        hipLaunchKernelGGL((k_big_init<MODE>), dim3(init_x, nb), dim3(256), 0, ctx->stream, mv, base, L, G, W, VEC ? 1 : 0);
        int par = 0;
        for (int sweep = 0; sweep <= TBK_JACOBI_MAX_SWEEPS; ++sweep) {
            TBK_HIP(hipMemsetAsync(W.pending, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(k_big_norm1, dim3(nbn, nb), dim3(256), 0, ctx->stream, W, par);
            hipLaunchKernelGGL(k_big_norm2, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, W, nb, par,
                               sweep == TBK_JACOBI_MAX_SWEEPS ? 1 : 0, ctx->flags_dev, sweep > 0 ? 1 : 0);
            if (sweep == 0 && ld > n) hipLaunchKernelGGL(k_blk_setpad, dim3(nb), dim3(64), 0, ctx->stream, W);
            int pending = 0;
            TBK_HIP(hipMemcpyAsync(&pending, W.pending, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (pending == 0) break;
            for (int round = 0; round < nbk - 1; ++round) {
                hipLaunchKernelGGL(k_blk_gather, dim3(npairs, nb), dim3(256), 0, ctx->stream, A, round, par);
                int rc = launch_row16<2, true>(ctx, sub_mv, A.nsub, Ls, G);
                if (rc) return rc;
                hipLaunchKernelGGL(k_blk_apply, dim3(npairs, (VEC ? 2 : 1) * npairs, nb), dim3(256), 0, ctx->stream, A, round, par);
                par ^= 1;
            }
            TBK_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL((k_big_sort<MODE>), dim3(nb), dim3(256), (size_t)ld * sizeof(double), ctx->stream, W, base, nk, L, G);
        if (VEC)
            hipLaunchKernelGGL((k_big_write<MODE>), dim3(init_x, nb), dim3(256), 0, ctx->stream, W, base, nk, L, G);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}
