// tbk_solve_trig.inl -- included by tbk_solve.hip.
//
// EIGENVALUES ONLY, n = 65..1024 states per k (ribbon and slab band structures: solve_all(k_path) without eigenvectors,
// pythtb.py:939 numpy.linalg.eigvalsh per k): the direct method where the Jacobi solvers (tbk_solve_big.inl,
// tbk_solve_blk.inl) spend 7-9 sweeps of n^2/2 rotations on a result that needs no eigenvectors.
//
//  1. k_tridiag_glb   one 1024-thread workgroup per matrix, A in a global workspace (it stays in L2 / the last-level cache
//                     for the sizes this path takes), u, p, q in LDS.  Householder step k: the column below the diagonal
//                     is read as the conjugate of row k (contiguous); the rank-2 update A -= u q^+ + q u^+ of step k and
//                     the product p = A u of step k+1 share ONE pass over the trailing block (the next reflector only needs
//                     the updated column k+1, which is formed ahead from row k+1): every element is read once and written
//                     once per step, one wavefront per row with the lanes along the columns, both triangles kept.
//                     (16/3) n^3 flops, 32 n^3 / 3 bytes of L2 traffic per matrix.
//  2. k_tridiag_bisect one thread per EIGENVALUE: bisection on the Sturm count of the real symmetric tridiagonal (d, e)
//                     (LAPACK dstebz's recurrence).  The QL iteration of the smaller sizes is a sequential chain of ~n^2
//                     rotations per matrix -- 11 ms at n = 300 whatever the batch -- whereas all n bisections of a matrix
//                     run side by side: 64 halvings x n recurrence steps each, d and e^2 broadcast from LDS.  Eigenvalue j
//                     is the j-th smallest by construction, so nothing is sorted.
//
// Every matrix is solved on its own.  Eigenvectors (and mesh solves, which always want them) keep the Jacobi kernels.

// sum over the workgroup of NT threads, the same bits in every thread (fixed order); red: 16 doubles of LDS; two barriers
template <int NT>
__device__ __forceinline__ double trig_block_sum(double v, double* red, const int tid) {
    v = row_allsum(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    __syncthreads();   // (red may still be read from the previous call)
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += red[w];
    return s;
}

// MODE 0: k list, 2: supplied matrices.  Block b works on matrix id0 + b; work holds nc matrices of n x n.
// ALDS: the matrix fits the LDS of a CU next to the vectors (n <= 96): the same steps on an LDS-resident A -- a row costs
// ~0.1 us of latency instead of ~1 us from L2, and the small sizes are all latency.
// KEEP (tbk_solve_trigv.inl: eigenvectors wanted): the reflectors stay behind for the back-transformation -- row k of A keeps
// conj(u_k[c]) for c >= k + 2 (no later step touches it), aux[(idc n + k) 3 + 0 .. 2] = (u_k[k+1]), (beta_k, reflected?), (t_k =
// T[k+1][k], complex: the diagonal unitary D that makes the subdiagonal real is D_{k+1} = D_k t_k / |t_k|).  MODE 1 (a mesh
// window, KEEP only): the point's k from grid_point.
template <int MODE, bool ALDS, int NT, bool KEEP = false>
__global__ __launch_bounds__(NT) void k_tridiag_glb(const ModelView mv, const int64_t nk, const ListArgs L, const int64_t id0,
                                                              const int64_t nc, cd* __restrict__ work, double2* __restrict__ de,
                                                              const GridArgs G = GridArgs{}, double2* __restrict__ aux = nullptr) {
    static_assert(!KEEP || !ALDS, "k_tridiag_glb: the reflectors are kept in the global workspace");
    static_assert(MODE != 1 || KEEP, "k_tridiag_glb: mesh windows always want eigenvectors");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int n = mv.nsta, ld = ALDS ? (n | 1) : n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    cd* ub = (cd*)lds_raw;                    // [n]   u of the current step
    cd* pb = ub + n;                          // [n]   p = beta A u of the current step
    cd* ub2 = pb + n;                         // [n]   the next step's
    cd* pb2 = ub2 + n;                        // [n]
    cd* qb = pb2 + n;                         // [n]
    cd* ph = qb + n;                          // [max(nR, 1)] assembly phases
    double* eb = (double*)(ph + (mv.nR > 1 ? mv.nR : 1));   // [n]
    double* red = eb + n;                     // [16]
    cd* shr = (cd*)(red + 16 + (n & 1));      // [2]: alpha (16-byte aligned)
    const int64_t idc = blockIdx.x, id = id0 + idc;
    cd* A = ALDS ? shr + 2 : work + (size_t)idc * n * ld;

    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
    } else if constexpr (MODE == 1) {
        bool wrap[4];
        grid_point(G, id, kk, wrap);
    }
    assemble_lds<MODE == 1 ? 0 : MODE, NT>(mv, L, id, kk, A, ld, ph, tid);
    __syncthreads();

    const int x = tid;                        // this thread's row in the per-row steps (n <= NT)
    // The reflector of a step: from the column below the diagonal (colx: this thread's entry, 0 outside) to u_x, beta and
    // |t_k|; reflect = something to annihilate (decided on the annihilated part alone, LAPACK zlarfg).  Uniform results.
    struct Refl {
        cd u;
        double beta, mag;
        bool on;
        cd t;                                 // t_k = T[k+1][k]
    };
    auto reflector = [&](const cd colx, const int k) {
        if (x == k + 1) shr[0] = colx;
        const double rest = trig_block_sum<NT>(x > k + 1 && x < n ? cabs2(colx) : 0.0, red, tid);   // (its barriers publish shr[0])
        const cd alpha = shr[0];
        const double absa2 = cabs2(alpha);
        Refl R{cd{0.0, 0.0}, 0.0, sqrt(absa2), rest > 0.0, alpha};
        if (R.on) {
            const double nrm = sqrt(rest + absa2);
            double absa = 0.0;
            cd phs{1.0, 0.0};
            if (absa2 > 0.0) {
                absa = sqrt(absa2);
                phs = cd{alpha.x / absa, alpha.y / absa};
            }
            R.u = x == k + 1 ? cd{phs.x * (absa + nrm), phs.y * (absa + nrm)} : colx;   // u = column + phase * norm * e_{k+1}
            R.beta = 1.0 / (nrm * (nrm + absa));
            R.mag = nrm;                                                               // t_k = -phase * nrm
            R.t = cd{-phs.x * nrm, -phs.y * nrm};
        }
        return R;
    };
    // step 0: its reflector and p = beta A u by a pass of their own; every later step gets both from the previous step's pass
    Refl cur;
    {
        cd colx{0.0, 0.0};
        if (x > 0 && x < n) colx = cconj(A[x]);          // column 0 below the diagonal = conj(row 0): contiguous
        cur = reflector(colx, 0);
        if (cur.on) {
            if (x < n) ub[x] = cur.u;
            __syncthreads();
            for (int r = 1 + wv; r < n; r += NW) {
                const cd* row = A + (size_t)r * ld;
                cd acc{0.0, 0.0};
                for (int c = 1 + lane; c < n; c += 64) cfma(acc, row[c], ub[c]);
                acc.x = row_allsum(acc.x);
                acc.y = row_allsum(acc.y);
                acc.x += __shfl_xor(acc.x, 16);
                acc.y += __shfl_xor(acc.y, 16);
                acc.x += __shfl_xor(acc.x, 32);
                acc.y += __shfl_xor(acc.y, 32);
                if (lane == 0) pb[r] = cd{acc.x * cur.beta, acc.y * cur.beta};
            }
        }
        __syncthreads();
    }
    for (int k = 0; k + 2 < n; ++k) {
        const bool below = x > k && x < n;
        cd q{0.0, 0.0};
        if (cur.on) {                             // (uniform)
            const cd p = below ? pb[x] : cd{0.0, 0.0};
            const double kappa = 0.5 * cur.beta * trig_block_sum<NT>(cur.u.x * p.x + cur.u.y * p.y, red, tid);   // beta/2 u^+ p (real)
            if (below) {
                q = cd{p.x - kappa * cur.u.x, p.y - kappa * cur.u.y};
                qb[x] = q;
            }
            __syncthreads();
        }
        // the NEXT step's reflector: column k+1 below the diagonal as it will be after this step's update,
        // conj(A[k+1][x] - (u_{k+1} conj(q_x) + q_{k+1} conj(u_x)))  (row k+1: contiguous)
        if constexpr (KEEP) {
            if (x == k + 1) {
                double2* a3 = aux + ((int64_t)idc * n + k) * 3;
                a3[0] = double2{cur.u.x, cur.u.y};
                a3[1] = double2{cur.beta, cur.on ? 1.0 : 0.0};
                a3[2] = double2{cur.t.x, cur.t.y};
            }
        }
        Refl nxt{cd{0.0, 0.0}, 0.0, 0.0, false, cd{0.0, 0.0}};
        const bool more = k + 3 < n;
        if (more) {
            cd colx{0.0, 0.0};
            if (x > k + 1 && x < n) {
                cd a = A[(size_t)(k + 1) * ld + x];
                if (cur.on) {
                    const cd u1 = ub[k + 1], q1 = qb[k + 1];
                    a.x -= (u1.x * q.x + u1.y * q.y) + (q1.x * cur.u.x + q1.y * cur.u.y);
                    a.y -= (u1.y * q.x - u1.x * q.y) + (q1.y * cur.u.x - q1.x * cur.u.y);
                }
                colx = cconj(a);
            }
            nxt = reflector(colx, k + 1);
            if (nxt.on && x < n) ub2[x] = nxt.u;
        }
        __syncthreads();
        // ONE pass over the trailing block: A -= u q^+ + q u^+ (both triangles kept) and, on the fly, the next step's
        // p' = beta' A_new u' over rows and columns >= k+2 -- every element is read once and written once per step
        if (cur.on || nxt.on) {
            for (int r = k + 1 + wv; r < n; r += NW) {
                cd* row = A + (size_t)r * ld;
                const cd ur = cur.on ? ub[r] : cd{0.0, 0.0}, qr = cur.on ? qb[r] : cd{0.0, 0.0};
                cd acc{0.0, 0.0};
                for (int c = k + 1 + lane; c < n; c += 64) {
                    cd a = row[c];
                    if (cur.on) {
                        const cd uc = ub[c], qc = qb[c];
                        a.x -= (ur.x * qc.x + ur.y * qc.y) + (qr.x * uc.x + qr.y * uc.y);
                        a.y -= (ur.y * qc.x - ur.x * qc.y) + (qr.y * uc.x - qr.x * uc.y);
                        row[c] = a;
                    }
                    if (nxt.on && c > k + 1) cfma(acc, a, ub2[c]);
                }
                if (nxt.on && r > k + 1) {        // (wave-uniform)
                    acc.x = row_allsum(acc.x);
                    acc.y = row_allsum(acc.y);
                    acc.x += __shfl_xor(acc.x, 16);
                    acc.y += __shfl_xor(acc.y, 16);
                    acc.x += __shfl_xor(acc.x, 32);
                    acc.y += __shfl_xor(acc.y, 32);
                    if (lane == 0) pb2[r] = cd{acc.x * nxt.beta, acc.y * nxt.beta};
                }
            }
        }
        if (tid == 0) eb[k] = cur.mag;            // e_k = |t_k| (a diagonal unitary makes the subdiagonal real and non-negative)
        __syncthreads();
        cd* t = ub;
        ub = ub2;
        ub2 = t;
        t = pb;
        pb = pb2;
        pb2 = t;
        cur = nxt;
    }
    if (tid == 0) {
        if (n >= 2) {
            const cd t = A[(size_t)(n - 2) * ld + (n - 1)];
            eb[n - 2] = sqrt(cabs2(t));
            if constexpr (KEEP) {                 // T[n-1][n-2] = conj(A[n-2][n-1]): never reflected
                double2* a3 = aux + ((int64_t)idc * n + (n - 2)) * 3;
                a3[0] = double2{0.0, 0.0};
                a3[1] = double2{0.0, 0.0};
                a3[2] = double2{t.x, -t.y};
            }
        }
        eb[n - 1] = 0.0;
    }
    __syncthreads();
    if (tid < n) de[(int64_t)idc * n + tid] = double2{A[(size_t)tid * ld + tid].x, eb[tid]};
}

// (d_j, e_j) of matrix idc at de[idc * si + j * sj]  ->  eval[j][id] ascending.  One block per matrix, thread j <-> eigenvalue j (+ 256, ...).
// threads per matrix of k_tridiag_bisect: one per eigenvalue up to 1024 (round 6: with 256 threads a matrix of 257..512 states ran its
// bisections in two rounds -- 101 x n = 300: 2.8 ms of the 17 ms call)
static inline int trig_bisect_nt(const int n) { return n <= 256 ? 256 : std::min(1024, (n + 63) & ~63); }
__global__ __launch_bounds__(1024) void k_tridiag_bisect(const int n, const int64_t nk, const int64_t id0, const double2* __restrict__ de,
                                                        double* __restrict__ eval, const int64_t si, const int64_t sj, int* flags) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double2* T = (double2*)lds_raw;           // [n] (d_j, e_{j-1}^2)   (e_{-1} = 0)
    double* red = (double*)(T + n);           // [32]: (lo, hi) per wavefront
    const int tid = threadIdx.x;
    const int64_t idc = blockIdx.x, id = id0 + idc;
    const double2* src = de + idc * si;
    double glo = INFINITY, ghi = -INFINITY, emax = 0.0;
    const int nt = blockDim.x;                // 64 | 256 | ... | 1024
    for (int j = tid; j < n; j += nt) {
        const double2 v = src[j * sj];
        const double em = j > 0 ? src[(j - 1) * sj].y : 0.0;
        T[j] = double2{v.x, em * em};
        // a NaN or an infinity in T (from the Hamiltonian: the reference's eigvalsh raises "Eigenvalues did not converge" on a
        // NaN, pythtb.py:939): bisection always "converges", so say it here (fmin / fmax below would drop a NaN silently)
        if (!(fabs(v.x) < INFINITY) || !(fabs(v.y) < INFINITY)) flags[0] = 1;
        const double rad = em + v.y;          // Gershgorin: |e_{j-1}| + |e_j|   (e_{n-1} = 0)
        glo = fmin(glo, v.x - rad);
        ghi = fmax(ghi, v.x + rad);
        emax = fmax(emax, v.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        glo = fmin(glo, __shfl_xor(glo, o));
        ghi = fmax(ghi, __shfl_xor(ghi, o));
    }
    if ((tid & 63) == 0) {
        red[2 * (tid >> 6)] = glo;
        red[2 * (tid >> 6) + 1] = ghi;
    }
    __syncthreads();
    glo = red[0];
    ghi = red[1];
    for (int w = 1; w < (nt >> 6); ++w) {
        glo = fmin(glo, red[2 * w]);
        ghi = fmax(ghi, red[2 * w + 1]);
    }
    const double span = fmax(fabs(glo), fabs(ghi));
    const double pivmin = 2.2250738585072014e-308 * fmax(1.0, span * span);   // safe minimum pivot (dstebz)
    glo -= 2.220446049250313e-16 * span * n + pivmin;
    ghi += 2.220446049250313e-16 * span * n + pivmin;
    for (int j = tid; j < n; j += nt) {
        // eigenvalue j (0-based): the smallest x with count(x) > j, count(x) = number of eigenvalues below x
        double lo = glo, hi = ghi;
        for (int it = 0; it < 120; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (mid <= lo || mid >= hi) break;                 // the interval is down to neighbouring doubles
            int cnt = 0;
            double q = 1.0;
            for (int i = 0; i < n; ++i) {
                const double2 t = T[i];
                q = t.x - mid - t.y / q;                       // (t.y = 0 at i = 0)
                if (fabs(q) < pivmin) q = -pivmin;
                cnt += q < 0.0 ? 1 : 0;
            }
            if (cnt > j) hi = mid;
            else lo = mid;
            if (hi - lo <= 2.0 * 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
        }
        eval[(int64_t)j * nk + id] = 0.5 * (lo + hi);
    }
}

template <int MODE>
static int launch_trig(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L) {
    static_assert(MODE != 1, "launch_trig: eigenvalues only, k lists and supplied matrices");
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t per = al((size_t)n * n * sizeof(cd)) + al((size_t)n * sizeof(double2));
    size_t free_b = 0, total_b = 0;
    TBK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>(per, std::min<size_t>((size_t)4 << 30, (free_b + ctx->work_bytes) / 2));
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nk, (int64_t)(budget / per)));
    chunk = (nk + (nk + chunk - 1) / chunk - 1) / ((nk + chunk - 1) / chunk);
    const size_t wbytes = (size_t)chunk * per + 256;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tridiagonalisation workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    cd* work = (cd*)ctx->work;
    double2* de = (double2*)((unsigned char*)ctx->work + (size_t)chunk * al((size_t)n * n * sizeof(cd)));
    const int nR = MODE == 2 ? 0 : mv.nR;
    const size_t lds_vec = ((size_t)5 * n + std::max(nR, 1) + 2) * sizeof(cd) + (((size_t)n + 16 + 1) & ~(size_t)1) * sizeof(double);
    const size_t lds_a = (size_t)n * (n | 1) * sizeof(cd);
    const bool alds = lds_vec + lds_a <= 160 * 1024 && tbk_knobs().use_trig != 3;   // (TBK_TRIG=3: A in L2 at every size)
    const size_t lds1 = lds_vec + (alds ? lds_a : 0);
    TBK_REQUIRE(lds1 <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d with %d lattice vectors needs %zu bytes of LDS", n, nR, lds1);
    // threads per matrix: a row per thread in the per-row steps, so at least n.  Measured (ms, 256 / 512 / 1024 threads): 512 x n=96
    // 2.50 / 2.11 / 2.19, 512 x n=128 4.59 / 4.18 / 4.27, 256 x n=200 11.2 / 7.96 / 7.31, 128 x n=256 20.3 / 12.5 / 9.1
    // (TBK_TRIG_NT forces 256 | 512 | 1024)
    int nt = n <= 160 ? 512 : 1024;
    if (tbk_knobs().trig_nt > 0) {
        const int want = tbk_knobs().trig_nt >= 1024 ? 1024 : (tbk_knobs().trig_nt >= 512 ? 512 : 256);
        nt = std::max(want, n <= 256 ? 256 : (n <= 512 ? 512 : 1024));
    }
    const void* f1 = nullptr;
#define TBK_TRIG_K1(AL_, NT_) \
    if (alds == AL_ && nt == NT_) f1 = (const void*)k_tridiag_glb<MODE, AL_, NT_>;
    TBK_TRIG_K1(false, 256) TBK_TRIG_K1(false, 512) TBK_TRIG_K1(false, 1024) TBK_TRIG_K1(true, 256) TBK_TRIG_K1(true, 512) TBK_TRIG_K1(true, 1024)
#undef TBK_TRIG_K1
    if (lds1 > 64 * 1024) TBK_HIP(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds2 = (size_t)n * sizeof(double2) + 32 * sizeof(double);
    for (int64_t id0 = 0; id0 < nk; id0 += chunk) {
        const int64_t nc = std::min<int64_t>(chunk, nk - id0);
        // (the workspace stride is n x n complex, rounded: keep the kernels' own stride n * n -- chunks are packed)
#define TBK_TRIG_K1(AL_, NT_)            \
    if (alds == AL_ && nt == NT_)        \
        hipLaunchKernelGGL((k_tridiag_glb<MODE, AL_, NT_>), dim3((unsigned)nc), dim3(NT_), lds1, ctx->stream, mv, nk, L, id0, nc, work, de);
        TBK_TRIG_K1(false, 256) TBK_TRIG_K1(false, 512) TBK_TRIG_K1(false, 1024) TBK_TRIG_K1(true, 256) TBK_TRIG_K1(true, 512) TBK_TRIG_K1(true, 1024)
#undef TBK_TRIG_K1
        hipLaunchKernelGGL(k_tridiag_bisect, dim3((unsigned)nc), dim3(trig_bisect_nt(n)), lds2, ctx->stream, n, nk, id0, (const double2*)de, L.eval,
                           (int64_t)n, (int64_t)1, ctx->flags_dev);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}
