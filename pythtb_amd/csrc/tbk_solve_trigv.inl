// tbk_solve_trigv.inl -- included by tbk_solve.hip after tbk_solve_trig.inl.
//
// 65..1024 states per k WITH eigenvectors by the direct method numpy.linalg.eigh runs for the reference (pythtb.py:944-952:
// reduce to tridiagonal, solve T, transform back) -- the ribbon / slab regime (cut_piece, make_supercell: examples/edge.py,
// haldane_fin, every solve_on_grid of a finite-width model) that the Jacobi kernels served at 10 x the cost of the
// eigenvalue-only path:
//
//  1. k_tridiag_glb<.., KEEP>   Householder with the matrix in L2, one workgroup per matrix (tbk_solve_trig.inl); the reflectors
//                               stay in the rows of A.
//  2. k_tridiag_bisect          one thread per eigenvalue, ascending by construction.
//  3. k_trigv_twisted           one thread per EIGENVECTOR of T: twisted factorisation (bottom-up pivots, then top-down pivots
//                               with gamma_k on the fly, then the two recurrences from the twist) -- O(n) per vector, all n of a
//                               matrix side by side.  T may split (couplings below eps |T|): the pivots restart there by
//                               themselves, the twist lands in a block that holds lambda, and equal eigenvalues of DIFFERENT
//                               blocks (Kramers pairs of a cleanly split T) take different blocks by their rank in the cluster.
//  4. k_trigv_gram / _apply     Newton-Schulz  V <- V (3 I - V^T V) / 2  on the real n x n matrix: independently computed
//                               vectors are orthogonal to ~20 eps |T| / gap only; each step squares that.  Two steps, the
//                               second skipped per matrix where the first Gram matrix was already the identity to 1e-15;
//                               a final Gram matrix decides: off by more than 1e-13 (eigenvalues of ONE block closer than
//                               ~eps |T| -- twisted vectors then coincide) raises the retry flag and the call is repeated
//                               on the Jacobi kernels.
//  5. k_trigv_back              z_j = H_0 .. H_{n-3} D v_j on column strips kept in LDS: the columns are independent, so a
//                               workgroup owns 4..16 of them (n x 16 complex in LDS) and streams the reflectors past -- the
//                               strip is read and written 2 (n - 2) times at LDS bandwidth instead of L2's.
//
// Every matrix is solved on its own (bit-identical windows and images).  MODE 0 k list, 1 mesh window, 2 supplied matrices.

// ---- 3. twisted-factorisation eigenvectors.  Block = one matrix, thread j <-> eigenvalue j.
// V[k * n + j] (component k of vector j) doubles as the store of the top-down pivots, W[k * n + j] of the bottom-up ones.
template <int NT>
__global__ __launch_bounds__(NT) void k_trigv_twisted(const int n, const double2* __restrict__ de, const double* __restrict__ lam,
                                                      const int64_t lam_stride_j, const int64_t lam_stride_m,
                                                      double* __restrict__ Vall, double* __restrict__ Wall) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double2* T = (double2*)lds_raw;            // [n]  (d_k, e_k), couplings below the splitting threshold zeroed
    double* ls = (double*)(T + n);             // [n]  the eigenvalues, ascending
    const int tid = threadIdx.x;
    const int64_t mat = blockIdx.x;
    const double2* src = de + mat * n;
    double* V = Vall + (size_t)mat * n * n;
    double* W = Wall + (size_t)mat * n * n;
    for (int k = tid; k < n; k += NT) {
        const double2 v = src[k];
        const double dn = k + 1 < n ? src[k + 1].x : 0.0;
        const bool ng = k + 1 >= n || fabs(v.y) <= 2.220446049250313e-16 * (fabs(v.x) + fabs(dn));
        T[k] = double2{v.x, ng ? 0.0 : v.y};
        ls[k] = lam[(int64_t)k * lam_stride_j + mat * lam_stride_m];
    }
    __syncthreads();
    double tnorm = fmax(fabs(ls[0]), fabs(ls[n - 1]));
    const double tiny = 1e-290;
    for (int j = tid; j < n; j += NT) {
        const double lj = ls[j];
        // rank of lambda_j among the eigenvalues that coincide with it to the accuracy of the bisection
        const double tolc = 8.0 * 2.220446049250313e-16 * tnorm;
        int m = 0;
        for (int i = j - 1; i >= 0 && lj - ls[i] <= tolc; --i) ++m;
        // bottom-up pivots dm_k = s_k - e_k^2 / dm_{k+1}
        double dm = T[n - 1].x - lj;
        W[(size_t)(n - 1) * n + j] = dm;
        for (int k = n - 2; k >= 0; --k) {
            const double2 t = T[k];
            const double p = fabs(dm) < tiny ? -tiny : dm;
            dm = (t.x - lj) - t.y * (t.y / p);
            W[(size_t)k * n + j] = dm;
        }
        // top-down pivots dp_k, gamma_k = dp_k + dm_k - s_k; the twist goes to the smallest |gamma| of a block; blocks whose
        // minimum is an eigenvalue-sized residual are candidates, and the m-th candidate (by position) is taken
        double dp = T[0].x - lj;
        double best = INFINITY, bmin = INFINITY;
        int r = 0, rb = 0, cand = 0, rsel = -1;
        const double gtol = 64.0 * n * 2.220446049250313e-16 * fmax(tnorm, 1e-300);
        for (int k = 0; k < n; ++k) {
            const double2 t = T[k];
            const double s = t.x - lj;
            const double g = fabs(dp + W[(size_t)k * n + j] - s);
            V[(size_t)k * n + j] = dp;
            if (g < bmin) {
                bmin = g;
                rb = k;
            }
            if (g < best) {
                best = g;
                r = k;
            }
            if (t.y == 0.0) {                  // end of a block
                if (bmin <= gtol) {
                    if (cand == m) rsel = rb;
                    ++cand;
                }
                bmin = INFINITY;
            }
            const double p = fabs(dp) < tiny ? -tiny : dp;
            if (k + 1 < n) dp = (T[k + 1].x - lj) - t.y * (t.y / p);
        }
        if (rsel >= 0) r = rsel;
        // z_r = 1; downwards z_k = -(e_k / dp_k) z_{k+1}; upwards z_{k+1} = -(e_k / dm_{k+1}) z_k
        double z = 1.0, nz2 = 1.0;
        for (int k = r - 1; k >= 0; --k) {
            const double pv = V[(size_t)k * n + j];
            const double p = fabs(pv) < tiny ? -tiny : pv;
            z = -(T[k].y / p) * z;
            V[(size_t)k * n + j] = z;
            nz2 = fma(z, z, nz2);
        }
        V[(size_t)r * n + j] = 1.0;
        z = 1.0;
        for (int k = r; k + 1 < n; ++k) {
            const double pv = W[(size_t)(k + 1) * n + j];
            const double p = fabs(pv) < tiny ? -tiny : pv;
            z = -(T[k].y / p) * z;
            V[(size_t)(k + 1) * n + j] = z;
            nz2 = fma(z, z, nz2);
        }
        const double inz = 1.0 / sqrt(nz2);
        for (int k = 0; k < n; ++k) V[(size_t)k * n + j] *= inz;
    }
    // ---- twins.  Two eigenvalues of ONE block closer than ~1e-6 |T| (bands degenerate on the whole mesh -- a spinful ribbon
    // without spin-orbit coupling, Kramers pairs: T splits only to ~1e-13 |T| -- or merely close levels) get (nearly) the same
    // twisted-factorisation vector; the final Gram matrix then failed and the WHOLE call was repeated on the Jacobi kernels (9 x
    // the time, profiles/degenerate_regimes_probe.py).  xSTEIN's remedy for the second member of an isolated pair: two rounds of
    // inverse iteration on T - lambda = L D L^T (pivots recomputed on the fly, multipliers parked in W) with reorthogonalisation
    // against the first member.  Pairs whose vectors are orthogonal already (different blocks) are left alone; longer clusters
    // and anything this does not repair are caught by the Gram matrix as before.
    __syncthreads();
    const double ctol = 1e-6 * fmax(tnorm, 1e-300);
    for (int j = tid; j < n; j += NT) {
        if (j == 0) continue;
        const double lj = ls[j];
        const bool cdn = lj - ls[j - 1] <= ctol;
        const bool cup = j + 1 < n && ls[j + 1] - lj <= ctol;
        const bool cdd = j >= 2 && ls[j - 1] - ls[j - 2] <= ctol;
        if (!cdn || cup || cdd) continue;
        const size_t cw = (size_t)j - 1, cv = (size_t)j;
        double c0 = 0.0;
        for (int k = 0; k < n; ++k) c0 = fma(V[(size_t)k * n + cw], V[(size_t)k * n + cv], c0);
        if (fabs(c0) <= 1e-9) continue;                // (independent already: the Newton-Schulz steps finish the job)
        for (int k = 0; k < n; ++k) V[(size_t)k * n + cv] = 1.0 + 0.37 * (double)(((k * 7919) % 13) - 6);   // a fixed generic start
        for (int round = 0; round < 2; ++round) {
            for (int half = 0; half < 2; ++half) {
                // orthogonalise against the first member, normalise (the solve grows by up to 1 / pivot: scale by the largest entry first)
                double c = 0.0, mx = 0.0;
                for (int k = 0; k < n; ++k) c = fma(V[(size_t)k * n + cw], V[(size_t)k * n + cv], c);
                for (int k = 0; k < n; ++k) {
                    const double x = fma(-c, V[(size_t)k * n + cw], V[(size_t)k * n + cv]);
                    V[(size_t)k * n + cv] = x;
                    mx = fmax(mx, fabs(x));
                }
                const double sc = mx > 0.0 && mx < INFINITY ? 1.0 / mx : 1.0;
                double nz2 = 0.0;
                for (int k = 0; k < n; ++k) {
                    const double x = V[(size_t)k * n + cv] * sc;
                    nz2 = fma(x, x, nz2);
                }
                const double inz = nz2 > 0.0 ? sc / sqrt(nz2) : 0.0;
                for (int k = 0; k < n; ++k) V[(size_t)k * n + cv] *= inz;
                if (half == 1) break;
                // solve (T - lj) x = v: forward with the multipliers l_k = e_k / dp_k (kept in W), then backward
                double dp = T[0].x - lj, y = 0.0, lprev = 0.0;
                for (int k = 0; k < n; ++k) {
                    const double2 t = T[k];
                    const double p = fabs(dp) < tiny ? -tiny : dp;
                    y = fma(-lprev, y, V[(size_t)k * n + cv]);
                    V[(size_t)k * n + cv] = y / p;
                    lprev = t.y / p;
                    W[(size_t)k * n + cv] = lprev;
                    if (k + 1 < n) dp = (T[k + 1].x - lj) - t.y * lprev;
                }
                double x = V[(size_t)(n - 1) * n + cv];
                for (int k = n - 2; k >= 0; --k) {
                    x = fma(-W[(size_t)k * n + cv], x, V[(size_t)k * n + cv]);
                    V[(size_t)k * n + cv] = x;
                }
            }
        }
    }
}

// ---- 4. Newton-Schulz.  C = A^T B (gram: C = V^T V) or C = A B (apply), batched real n x n, 64 x 64 tiles, 256 threads, 4 x 4
// per thread.  Epilogues: gram -> X = 1.5 I - 0.5 G and the matrix's max |G - I| (bit pattern, atomicMax);
// apply -> plain store.
template <bool TRANS_A, bool GRAM>
__global__ __launch_bounds__(256) void k_trigv_gemm(const int n, const double* __restrict__ Aall, const double* __restrict__ Ball,
                                                    double* __restrict__ Call, unsigned long long* __restrict__ err) {
    __shared__ double As[16][65], Bs[16][65];
    const int64_t mat = blockIdx.z;
    const double* A = Aall + (size_t)mat * n * n;
    const double* B = Ball + (size_t)mat * n * n;
    double* Cm = Call + (size_t)mat * n * n;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int k0 = 0; k0 < n; k0 += 16) {
        // stage As[kk][i] = op(A)[i0 + i][k0 + kk], Bs[kk][j] = B[k0 + kk][j0 + j]
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e >> 6, c = e & 63;
            const int k = k0 + kk;
            double av = 0.0, bv = 0.0;
            if (k < n) {
                if (TRANS_A) {
                    if (i0 + c < n) av = A[(size_t)k * n + i0 + c];          // A^T[i][k] = A[k][i]
                } else {
                    // A[i][k]: read transposed below (kk fastest) to keep the loads coalesced
                }
                if (j0 + c < n) bv = B[(size_t)k * n + j0 + c];
            }
            if (TRANS_A) As[kk][c] = av;
            Bs[kk][c] = bv;
        }
        if (!TRANS_A) {
            for (int e = threadIdx.x; e < 16 * 64; e += 256) {
                const int c = e >> 4, kk = e & 15;                            // A[i0 + c][k0 + kk]: kk contiguous
                const int k = k0 + kk;
                As[kk][c] = (k < n && i0 + c < n) ? A[(size_t)(i0 + c) * n + k] : 0.0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = As[kk][ty * 4 + q];
                b[q] = Bs[kk][tx * 4 + q];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[p][q] = fma(a[p], b[q], acc[p][q]);
        }
        __syncthreads();
    }
    double emax = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = i0 + ty * 4 + p, j = j0 + tx * 4 + q;
            if (i < n && j < n) {
                double v = acc[p][q];
                if (GRAM) {
                    const double d = i == j ? 1.0 : 0.0;
                    const double dev = fabs(v - d);
                    emax = dev > emax || dev != dev ? (dev != dev ? INFINITY : dev) : emax;   // (a NaN counts as infinite)
                    v = fma(-0.5, v, 1.5 * d);
                }
                Cm[(size_t)i * n + j] = v;
            }
        }
    if (GRAM) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) emax = fmax(emax, __shfl_xor(emax, off));
        if ((threadIdx.x & 63) == 0) atomicMax(err + mat, (unsigned long long)__double_as_longlong(emax));
    }
}

// max over the matrices of the final Gram error -> the retry flag (flags[2]) when the vectors could not be made orthonormal
__global__ void k_trigv_verdict(const unsigned long long* __restrict__ err, const int64_t nmat, const double tol, int* flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nmat && !(__longlong_as_double((long long)err[i]) <= tol)) flags[2] = 1;
}

// ---- 5. back-transformation and output.  Block = (column strip, matrix): NC columns of V in LDS as complex Y[x][c]; for
// k = n-3 .. 0: w_c = beta_k sum_x conj(u_k[x]) Y[x][c], Y[x][c] -= u_k[x] w_c.  Thread = (column c, row group g).
template <int MODE, int NC>
__global__ __launch_bounds__(256) void k_trigv_back(const ModelView mv, const int64_t nk, const ListArgs L, const GridArgs G, const int64_t id0,
                                                    const cd* __restrict__ work, const double2* __restrict__ aux,
                                                    const double* __restrict__ Vall) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int NG = 256 / NC;               // row groups
    const int n = mv.nsta;
    cd* Y = (cd*)lds_raw;                      // [n][NC]
    cd* us = Y + (size_t)n * NC;               // [n]   the current reflector
    cd* wred = us + n;                         // [NG][NC] partial dot products
    cd* dph = wred + NG * NC;                  // [n]   D
    const int tid = threadIdx.x, c = tid % NC, g = tid / NC;
    const int64_t idc = blockIdx.y, id = id0 + idc;
    const int j0 = blockIdx.x * NC;
    const cd* A = work + (size_t)idc * n * n;
    const double2* a3 = aux + (int64_t)idc * n * 3;
    const double* V = Vall + (size_t)idc * n * n;
    // D_0 = 1, D_{k+1} = D_k t_k / |t_k|
    if (tid == 0) {
        cd d{1.0, 0.0};
        dph[0] = d;
        for (int k = 0; k + 1 < n; ++k) {
            const double2 t = a3[k * 3 + 2];
            const double t2 = t.x * t.x + t.y * t.y;
            if (t2 > 0.0) {
                const double inv = 1.0 / sqrt(t2);
                d = cmul(d, cd{t.x * inv, t.y * inv});
            }
            dph[k + 1] = d;
        }
    }
    __syncthreads();
    const bool col_ok = j0 + c < n;
    for (int x = g; x < n; x += NG) {
        const double v = col_ok ? V[(size_t)x * n + j0 + c] : 0.0;
        const cd d = dph[x];
        Y[(size_t)x * NC + c] = cd{d.x * v, d.y * v};
    }
    __syncthreads();
    for (int k = n - 3; k >= 0; --k) {
        const double2 bo = a3[k * 3 + 1];
        if (bo.y == 0.0) continue;              // (uniform) nothing was reflected at this step
        // u_k: zero up to k, aux at k+1, conj(row k of A) beyond
        for (int x = k + 1 + tid; x < n; x += 256) {
            cd u;
            if (x == k + 1) {
                const double2 t = a3[k * 3];
                u = cd{t.x, t.y};
            } else {
                u = cconj(A[(size_t)k * n + x]);
            }
            us[x] = u;
        }
        __syncthreads();
        cd acc{0.0, 0.0};
        for (int x = k + 1 + g; x < n; x += NG) cfmac(acc, us[x], Y[(size_t)x * NC + c]);
        wred[g * NC + c] = acc;
        __syncthreads();
        cd w{0.0, 0.0};
#pragma unroll 4
        for (int gg = 0; gg < NG; ++gg) w = cadd(w, wred[gg * NC + c]);   // (fixed order: the same bits in every group)
        w = cd{w.x * bo.x, w.y * bo.x};
        for (int x = k + 1 + g; x < n; x += NG) {
            cd y = Y[(size_t)x * NC + c];
            const cd u = us[x];
            y.x -= u.x * w.x - u.y * w.y;
            y.y -= u.x * w.y + u.y * w.x;
            Y[(size_t)x * NC + c] = y;
        }
        __syncthreads();
    }
    // ---- output: eigenvector j0 + c of H = conj(e_x) z_x (pbc phase on periodic images), band-major
    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
    } else if constexpr (MODE == 1) {
        grid_point(G, id, kk, wrap);
    }
    for (int e = tid; e < n * NC; e += 256) {
        const int cc = e / n, x = e - cc * n;   // x fastest: contiguous stores along the orbital index
        if (j0 + cc >= n) continue;
        cd f{1.0, 0.0};
        if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[x])));
        if constexpr (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (wrap[d]) f = cmul(f, G.pbc[d * n + x]);
        }
        const cd val = cmul(Y[(size_t)x * NC + cc], f);
        if constexpr (MODE == 1) wf_at(G.wv, j0 + cc, id)[x] = val;
        else L.evec[((int64_t)(j0 + cc) * nk + id) * n + x] = val;
    }
}

// mesh windows: min over the window's points of E[b+1] - E[b] from the ascending eigenvalues ev[b * nc + idc]
__global__ __launch_bounds__(256) void k_trigv_gaps(const int n, const int64_t nc, const double* __restrict__ ev, const GridArgs G) {
    const int b = blockIdx.x;                   // gap b
    double gmin = INFINITY;
    for (int64_t i = threadIdx.x; i < nc; i += 256) gmin = fmin(gmin, ev[(int64_t)(b + 1) * nc + i] - ev[(int64_t)b * nc + i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) gmin = fmin(gmin, __shfl_xor(gmin, off));
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* slot = G.gaps + (size_t)((threadIdx.x >> 6) & (TBK_GAP_SHARDS - 1)) * n + b;
        atomicMin(slot, (unsigned long long)__double_as_longlong(fmax(gmin, 0.0)));
    }
}

template <int MODE>
static int launch_trigv(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L, const GridArgs& G) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nn = (size_t)n * n;
    // per matrix: A (complex: the reflectors), three real n x n arrays (vectors / pivots and X / products), (d, e), aux,
    // eigenvalues, Gram errors
    const size_t per = al(nn * sizeof(cd)) + 3 * al(nn * sizeof(double)) + al((size_t)n * sizeof(double2)) + al((size_t)n * 3 * sizeof(double2)) +
                       al((size_t)n * sizeof(double)) + 64;
    size_t free_b = 0, total_b = 0;
    TBK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>(per, std::min<size_t>((size_t)6 << 30, (free_b + ctx->work_bytes) / 2));
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(nk, (int64_t)(budget / per)));
    chunk = (nk + (nk + chunk - 1) / chunk - 1) / ((nk + chunk - 1) / chunk);
    const size_t wbytes = al((size_t)chunk * nn * sizeof(cd)) + 3 * al((size_t)chunk * nn * sizeof(double)) + al((size_t)chunk * n * sizeof(double2)) +
                          al((size_t)chunk * n * 3 * sizeof(double2)) + al((size_t)chunk * n * sizeof(double)) + al((size_t)chunk * 2 * sizeof(unsigned long long)) + 256;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tridiagonalisation workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    unsigned char* p = (unsigned char*)ctx->work;
    cd* work = (cd*)p;
    p += al((size_t)chunk * nn * sizeof(cd));
    double* V = (double*)p;
    p += al((size_t)chunk * nn * sizeof(double));
    double* W = (double*)p;
    p += al((size_t)chunk * nn * sizeof(double));
    double* U = (double*)p;
    p += al((size_t)chunk * nn * sizeof(double));
    double2* de = (double2*)p;
    p += al((size_t)chunk * n * sizeof(double2));
    double2* aux = (double2*)p;
    p += al((size_t)chunk * n * 3 * sizeof(double2));
    double* evb = (double*)p;                   // [n][chunk] eigenvalues (mesh windows; lists write L.eval directly)
    p += al((size_t)chunk * n * sizeof(double));
    unsigned long long* err = (unsigned long long*)p;   // [2][chunk]
    const int nR = MODE == 2 ? 0 : mv.nR;
    const size_t lds1 = ((size_t)5 * n + std::max(nR, 1) + 2) * sizeof(cd) + (((size_t)n + 16 + 1) & ~(size_t)1) * sizeof(double);
    TBK_REQUIRE(lds1 <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d with %d lattice vectors needs %zu bytes of LDS", n, nR, lds1);
    const int nt = n <= 160 ? 512 : 1024;
    const void* f1 = nt == 512 ? (const void*)k_tridiag_glb<MODE, false, 512, true> : (const void*)k_tridiag_glb<MODE, false, 1024, true>;
    if (lds1 > 64 * 1024) TBK_HIP(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds2 = (size_t)n * sizeof(double2) + 32 * sizeof(double);
    const size_t lds3 = (size_t)n * (sizeof(double2) + sizeof(double));
    // columns per strip of the back-transformation: n x NC complex (+ reflector, D, partial sums) within 160 KB of LDS
    // (ms for the back-transformation, 16 | 8 | 4 columns: 512 x n=128 1.3 | 2.0 | 4.9, 101 x n=300 6.6 | 4.1 | 7.0 -- a strip should
    // leave room for two workgroups per CU)
    int ncs = n <= 224 ? 16 : (n <= 980 ? 8 : 4);
    if (tbk_knobs().trigv_nc > 0) ncs = tbk_knobs().trigv_nc >= 16 ? 16 : (tbk_knobs().trigv_nc >= 8 ? 8 : 4);
    while (ncs > 4 && ((size_t)n * ncs + 2 * (size_t)n + 256) * sizeof(cd) > 160 * 1024) ncs /= 2;
    const size_t lds5 = ((size_t)n * ncs + 2 * (size_t)n + 256) * sizeof(cd);
    TBK_REQUIRE(lds5 <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d: back-transformation strip of %zu bytes", n, lds5);
    const void* f5 = ncs == 16 ? (const void*)k_trigv_back<MODE, 16> : ncs == 8 ? (const void*)k_trigv_back<MODE, 8> : (const void*)k_trigv_back<MODE, 4>;
    if (lds5 > 64 * 1024) TBK_HIP(hipFuncSetAttribute(f5, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const unsigned tiles = (unsigned)((n + 63) / 64);
    for (int64_t id0 = 0; id0 < nk; id0 += chunk) {
        const int64_t nc = std::min<int64_t>(chunk, nk - id0);
        {
            ProfScope ps(ctx, "trigv_tridiag");
            if (nt == 512)
                hipLaunchKernelGGL((k_tridiag_glb<MODE, false, 512, true>), dim3((unsigned)nc), dim3(512), lds1, ctx->stream, mv, nk, L, id0, nc, work, de, G, aux);
            else
                hipLaunchKernelGGL((k_tridiag_glb<MODE, false, 1024, true>), dim3((unsigned)nc), dim3(1024), lds1, ctx->stream, mv, nk, L, id0, nc, work, de, G, aux);
        }
        double* lam = MODE == 1 ? evb : L.eval;
        const int64_t lam_nk = MODE == 1 ? nc : nk, lam_id0 = MODE == 1 ? 0 : id0;
        {
            ProfScope ps(ctx, "trigv_bisect");
            hipLaunchKernelGGL(k_tridiag_bisect, dim3((unsigned)nc), dim3(trig_bisect_nt(n)), lds2, ctx->stream, n, lam_nk, lam_id0, (const double2*)de, lam,
                               (int64_t)n, (int64_t)1, ctx->flags_dev);
        }
        if constexpr (MODE == 1) hipLaunchKernelGGL(k_trigv_gaps, dim3((unsigned)(n - 1)), dim3(256), 0, ctx->stream, n, nc, (const double*)evb, G);
        {
            ProfScope ps(ctx, "trigv_twisted");
            hipLaunchKernelGGL((k_trigv_twisted<256>), dim3((unsigned)nc), dim3(256), lds3, ctx->stream, n, (const double2*)de,
                               (const double*)(lam + lam_id0), lam_nk, (int64_t)1, V, W);
        }
        {
            // two Newton-Schulz steps: X1 = 1.5 I - 0.5 V^T V, U = V X1; X2 from U, V = U X2.  err[1] = max |U^T U - I|: the final
            // deviation is its square, so 1e-7 there means orthonormal to rounding; more (vectors of one block that
            // coincide) raises the retry flag
            ProfScope ps(ctx, "trigv_newton_schulz");
            TBK_HIP(hipMemsetAsync(err, 0, (size_t)2 * chunk * sizeof(unsigned long long), ctx->stream));
            const dim3 grid(tiles, tiles, (unsigned)nc);
            hipLaunchKernelGGL((k_trigv_gemm<true, true>), grid, dim3(256), 0, ctx->stream, n, (const double*)V, (const double*)V, W, err);
            hipLaunchKernelGGL((k_trigv_gemm<false, false>), grid, dim3(256), 0, ctx->stream, n, (const double*)V, (const double*)W, U,
                               (unsigned long long*)nullptr);
            hipLaunchKernelGGL((k_trigv_gemm<true, true>), grid, dim3(256), 0, ctx->stream, n, (const double*)U, (const double*)U, W, err + chunk);
            hipLaunchKernelGGL((k_trigv_gemm<false, false>), grid, dim3(256), 0, ctx->stream, n, (const double*)U, (const double*)W, V,
                               (unsigned long long*)nullptr);
            hipLaunchKernelGGL(k_trigv_verdict, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const unsigned long long*)(err + chunk), nc, 1e-7, ctx->flags_dev);
        }
        {
            ProfScope ps(ctx, "trigv_back");
            const dim3 grid((unsigned)((n + ncs - 1) / ncs), (unsigned)nc);
            if (ncs == 16)
                hipLaunchKernelGGL((k_trigv_back<MODE, 16>), grid, dim3(256), lds5, ctx->stream, mv, nk, L, G, id0, (const cd*)work, (const double2*)aux,
                                   (const double*)V);
            else if (ncs == 8)
                hipLaunchKernelGGL((k_trigv_back<MODE, 8>), grid, dim3(256), lds5, ctx->stream, mv, nk, L, G, id0, (const cd*)work, (const double2*)aux,
                                   (const double*)V);
            else
                hipLaunchKernelGGL((k_trigv_back<MODE, 4>), grid, dim3(256), lds5, ctx->stream, mv, nk, L, G, id0, (const cd*)work, (const double2*)aux,
                                   (const double*)V);
        }
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}
