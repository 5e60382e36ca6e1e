// tbk_berry_big.inl -- included by tbk_berry.hip.
//
// Occupied sets larger than TBK_MAX_NOCC (ribbons and slabs: tens to hundreds of occupied bands).
// The determinant of a product is the product of the determinants, so both
//   berry_phase(occ, dir, berry_evals=False) = -arg prod_i det M(i, i+1)          (pythtb.py:3829-3831)
//   berry_flux plaquette = -arg det[M(00,10) M(10,11) M(11,01) M(01,00)]          (pythtb.py:3840-3865)
// need nothing but the determinants of the link matrices M_mn = <u_m(p)|u_n(p + e_dir)>.  One
// workgroup per link forms M and reduces it by LU with partial pivoting (what numpy.linalg.det
// does); two small kernels then combine the link determinants.
//
// The Wilson-loop eigenphases (berry_evals=True) need the polar factor of every link and the spectrum of
// their ordered product (pythtb.py:3825-3838): second half of this file, also workgroup-level.

struct LinkDetArgs {
    WfsView v;
    const int* occ;    // [nocc] device
    int nocc;
    int ndir;          // mesh points along the link direction
    int64_t sdir;      // its point stride
    cd* dets;          // [npts]  det of the link starting at each point (1 where there is none)
    cd* work;          // [gridDim.x][nocc*nocc] when the matrix does not fit in LDS, else null
};

__global__ __launch_bounds__(256) void k_link_det_big(const LinkDetArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    __shared__ cd s_det;
    __shared__ int s_piv;
    const int tid = threadIdx.x, nocc = A.nocc, ncomp = A.v.ncomp;
    const int ld = A.work ? nocc : nocc + 1;
    cd* M = A.work ? A.work + (size_t)blockIdx.x * nocc * nocc : reinterpret_cast<cd*>(lds_raw);
    for (int64_t p = blockIdx.x; p < A.v.npts; p += gridDim.x) {
        const int64_t idir = (p / A.sdir) % A.ndir;
        if (idir == A.ndir - 1) {   // uniform per workgroup: the last point along dir starts no link
            if (tid == 0) A.dets[p] = cd{1.0, 0.0};
            continue;
        }
        const int64_t q = p + A.sdir;
        __syncthreads();
        for (int e = tid; e < nocc * nocc; e += 256) {
            const int m = e / nocc, n2 = e - m * nocc;
            const cd* up = wf_at(A.v, A.occ[m], p);
            const cd* uq = wf_at(A.v, A.occ[n2], q);
            cd acc{0.0, 0.0};
            for (int c = 0; c < ncomp; ++c) cfmac(acc, up[c], uq[c]);
            M[m * ld + n2] = acc;
        }
        if (tid == 0) s_det = cd{1.0, 0.0};
        __syncthreads();
        bool singular = false;
        for (int k = 0; k < nocc; ++k) {
            // pivot: largest |M[i][k]|, i >= k (first one on ties)
            double best = -1.0;
            int bi = k;
            for (int i = k + tid; i < nocc; i += 256) {
                const double a = cabs2(M[i * ld + k]);
                if (a > best) {
                    best = a;
                    bi = i;
                }
            }
            s_val[tid] = best;
            s_idx[tid] = bi;
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if (tid < w) {
                    const double o = s_val[tid + w];
                    const int oi = s_idx[tid + w];
                    if (o > s_val[tid] || (o == s_val[tid] && oi < s_idx[tid])) {
                        s_val[tid] = o;
                        s_idx[tid] = oi;
                    }
                }
                __syncthreads();
            }
            if (tid == 0) s_piv = s_idx[0];
            __syncthreads();
            const int r = s_piv;
            if (s_val[0] <= 0.0) {   // exactly singular: det = 0
                singular = true;
                break;
            }
            if (r != k)
                for (int j = tid; j < nocc; j += 256) {
                    const cd t = M[k * ld + j];
                    M[k * ld + j] = M[r * ld + j];
                    M[r * ld + j] = t;
                }
            __syncthreads();
            const cd piv = M[k * ld + k];
            if (tid == 0) {
                cd d = cmul(s_det, piv);
                if (r != k) d = cd{-d.x, -d.y};
                s_det = d;
            }
            const double ip = 1.0 / cabs2(piv);
            const cd inv{piv.x * ip, -piv.y * ip};
            const int wdt = nocc - k - 1;
            for (int e = tid; e < wdt * wdt; e += 256) {
                const int i = k + 1 + e / wdt, j = k + 1 + e % wdt;
                const cd f = cmul(M[i * ld + k], inv);
                const cd t = cmul(f, M[k * ld + j]);
                M[i * ld + j] = cd{M[i * ld + j].x - t.x, M[i * ld + j].y - t.y};
            }
            __syncthreads();
        }
        if (tid == 0) A.dets[p] = singular ? cd{0.0, 0.0} : s_det;
    }
}

// ordered product of the link determinants of every string along dir -> -arg
struct StringDetArgs {
    const cd* dets;
    int nlinks;
    int64_t sdir;
    AxisSet other;
    int64_t nstrings;
    double* out;
};
__global__ __launch_bounds__(256) void k_string_from_dets(const StringDetArgs A) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= A.nstrings) return;
    const int64_t base = axis_offset(A.other, s);
    cd acc{1.0, 0.0};
    for (int i = 0; i < A.nlinks; ++i) acc = cmul(acc, A.dets[base + (int64_t)i * A.sdir]);
    A.out[s] = -atan2(acc.y, acc.x);
}

// plaquette phases from the two link-determinant arrays; same indexing, partial sums and
// plaquette layout as k_flux
struct PlaqDetArgs {
    const cd* d0;      // links along dir0
    const cd* d1;      // links along dir1
    int n0, n1;
    int64_t s0, s1;
    AxisSet other;
    int bps;
    double* plaq;      // nullable
    double* partial;
};
__global__ __launch_bounds__(256) void k_flux_from_dets(const PlaqDetArgs A) {
    const int64_t slice = blockIdx.x / A.bps;
    const int blk = (int)(blockIdx.x - slice * A.bps);
    const int64_t per = (int64_t)A.n0 * A.n1;
    const int64_t idx = (int64_t)blk * 256 + threadIdx.x;
    double phase = 0.0;
    if (idx < per) {
        const int i = (int)(idx / A.n1), j = (int)(idx - (int64_t)i * A.n1);
        const int64_t p00 = axis_offset(A.other, slice) + (int64_t)i * A.s0 + (int64_t)j * A.s1;
        // (i,j) -> (i+1,j) -> (i+1,j+1) -> (i,j+1) -> (i,j): the two backward links are the adjoints
        cd acc = cmul(A.d0[p00], A.d1[p00 + A.s0]);
        acc = cmul(acc, cconj(A.d0[p00 + A.s1]));
        acc = cmul(acc, cconj(A.d1[p00]));
        phase = -atan2(acc.y, acc.x);
        if (A.plaq) A.plaq[slice * per + idx] = phase;
    }
    __shared__ double red[256];
    red[threadIdx.x] = phase;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.partial[slice * A.bps + blk] = red[0];
}

// dets[p] for every mesh point along `dir`; scratch layout decided by the caller
static int launch_link_dets(tbk_wfs* w, const int* occ_dev, int nocc, int dir, cd* dets, void* work, size_t work_bytes) {
    tbk_ctx* ctx = w->ctx;
    const WfsView& v = w->view;
    LinkDetArgs A{};
    A.v = v;
    A.occ = occ_dev;
    A.nocc = nocc;
    A.ndir = v.mesh[dir];
    A.sdir = v.stride[dir];
    A.dets = dets;
    const size_t lds = (size_t)nocc * (nocc + 1) * sizeof(cd);
    const bool in_lds = lds <= 96 * 1024;
    static bool attr_set = false;
    if (in_lds && lds > 48 * 1024 && !attr_set) {
        TBK_HIP(hipFuncSetAttribute((const void*)k_link_det_big, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set = true;
    }
    unsigned blocks = (unsigned)std::min<int64_t>(v.npts, (int64_t)ctx->cus * 4);
    if (!in_lds) {
        const size_t per = (size_t)nocc * nocc * sizeof(cd);
        blocks = (unsigned)std::max<size_t>(1, std::min<size_t>(blocks, work_bytes / per));
        A.work = (cd*)work;
    }
    ProfScope ps(ctx, "link_det_big");
    hipLaunchKernelGGL(k_link_det_big, dim3(blocks), dim3(256), in_lds ? lds : 0, ctx->stream, A);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// ---- Wilson-loop eigenphases for large occupied sets, workgroup level.
// Every step is a dense nocc x nocc operation shared by the 256 threads of a workgroup, matrices in global
// memory (L2-resident: 78 KB at nocc = 70):
//   1. link polar factors  W = M (M^H M)^(-1/2)  (= U Vh of the SVD, pythtb.py:3825-3826) by the Newton-Schulz
//      iteration X <- X (3 I - X^H X) / 2, which needs only products and converges quadratically from the singular
//      values of an overlap matrix (all in (0, 1]);
//   2. the ordered product of a string's factors by a pairwise tree, one launch per level;
//   3. the spectrum of the unitary product P through the Cayley transform of Q = e^{-i alpha} P,
//      H = i (I - Q)(I + Q)^(-1), Hermitian with eigenvalues tan((theta - alpha) / 2): one Gauss-Jordan solve,
//      then the batched Hermitian eigen-solver of tbk_solve.hip; theta -> sort(-angle)  (pythtb.py:3834-3838).
//      The map has a pole at theta = alpha + pi; an eigenphase closer than ~2e-4 to it (|h| > 1e4) would lose
//      digits, so the host repeats step 3 with another alpha for such strings.

// workgroup-cooperative C = sAB * op(A) B + sD * D (m x m, row-major); op(A) = A^H when HA.  2 x 2 register tiles.
// Returns this thread's share of ||C - I||_F^2 when RES.
template <bool HA, bool RES>
__device__ double wg_matmul(const int m, const cd* __restrict__ A, const cd* __restrict__ B, cd* __restrict__ C,
                            const double sAB, const cd* __restrict__ D, const double sD) {
    const int mt = (m + 1) >> 1;
    double res = 0.0;
    for (int e = threadIdx.x; e < mt * mt; e += blockDim.x) {
        const int ta = e / mt, tb = e - ta * mt;
        const int a0 = 2 * ta, b0 = 2 * tb;
        const int a1 = min(a0 + 1, m - 1), b1 = min(b0 + 1, m - 1);
        cd c00{0.0, 0.0}, c01{0.0, 0.0}, c10{0.0, 0.0}, c11{0.0, 0.0};
        for (int j = 0; j < m; ++j) {
            const cd y0 = B[j * m + b0], y1 = B[j * m + b1];
            if (HA) {
                const cd x0 = A[j * m + a0], x1 = A[j * m + a1];
                cfmac(c00, x0, y0);
                cfmac(c01, x0, y1);
                cfmac(c10, x1, y0);
                cfmac(c11, x1, y1);
            } else {
                const cd x0 = A[a0 * m + j], x1 = A[a1 * m + j];
                cfma(c00, x0, y0);
                cfma(c01, x0, y1);
                cfma(c10, x1, y0);
                cfma(c11, x1, y1);
            }
        }
        auto put = [&](int a, int b, cd c) {
            cd o = cscale(c, sAB);
            if (D) {
                const cd d = D[a * m + b];
                o.x += sD * d.x;
                o.y += sD * d.y;
            }
            C[a * m + b] = o;
            if (RES) {
                const double dx = o.x - (a == b ? 1.0 : 0.0);
                res += dx * dx + o.y * o.y;
            }
        };
        put(a0, b0, c00);
        if (b0 + 1 < m) put(a0, b1, c01);
        if (a0 + 1 < m) {
            put(a1, b0, c10);
            if (b0 + 1 < m) put(a1, b1, c11);
        }
    }
    return res;
}

struct WilsonBigArgs {
    WfsView v;
    const int* occ;
    int nocc, nlinks;
    int64_t sdir;
    AxisSet other;
    int64_t s0;      // first string of this batch
    int64_t ns;      // strings in this batch
    cd* buf0;        // [ns][nlinks][nocc^2]  link factors (result of step 1), then tree levels
    cd* buf1;        // same size, ping-pong partner
    cd* ywork;       // [gridDim.x][nocc^2]
    int* flags;      // ctx->flags_dev: word 1 is set when a polar iteration fails to converge (singular link matrix)
};

__global__ __launch_bounds__(256) void k_link_polar_big(const WilsonBigArgs A) {
    __shared__ double red[256];
    const int tid = threadIdx.x, m = A.nocc, nn = m * m, ncomp = A.v.ncomp;
    cd* Y = A.ywork + (size_t)blockIdx.x * nn;
    for (int64_t item = blockIdx.x; item < A.ns * A.nlinks; item += gridDim.x) {
        const int64_t s = item / A.nlinks;
        const int i = (int)(item - s * A.nlinks);
        cd* const home = A.buf0 + (size_t)item * nn;
        cd* X = home;
        cd* X2 = A.buf1 + (size_t)item * nn;
        const int64_t p = axis_offset(A.other, A.s0 + s) + (int64_t)i * A.sdir, q = p + A.sdir;
        for (int e = tid; e < nn; e += 256) {
            const int a = e / m, b = e - a * m;
            const cd* up = wf_at(A.v, A.occ[a], p);
            const cd* uq = wf_at(A.v, A.occ[b], q);
            cd acc{0.0, 0.0};
            for (int c = 0; c < ncomp; ++c) cfmac(acc, up[c], uq[c]);
            X[e] = acc;
        }
        __syncthreads();
        bool converged = false;
        for (int it = 0; it < 200; ++it) {
            red[tid] = wg_matmul<true, true>(m, X, X, Y, 1.0, nullptr, 0.0);      // Y = X^H X, residual ||Y - I||_F^2
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if (tid < w) red[tid] += red[tid + w];
                __syncthreads();
            }
            const double r2 = red[0];
            wg_matmul<false, false>(m, X, Y, X2, -0.5, X, 1.5);                    // X <- (3 X - X Y) / 2
            __syncthreads();
            cd* t = X;
            X = X2;
            X2 = t;
            if (r2 < 1e-14) {           // residual 1e-7 before this update, its square after it
                converged = true;
                break;
            }
        }
        // A singular overlap matrix (orthogonal occupied subspaces at neighbouring points) has no polar factor: the
        // iteration keeps its zero singular values at zero.  Report it instead of handing a non-unitary factor on.
        if (!converged && tid == 0) atomicExch(A.flags + 1, 1);
        if (X != home) {
            for (int e = tid; e < nn; e += 256) home[e] = X[e];
        }
        __syncthreads();
    }
}

// 3 or 4 bands, the whole string in registers: a thread owns a SEGMENT of a string, forms each link matrix, iterates it to its polar
// factor and multiplies it onto the segment's ordered product; k_wilson_lanes_combine (tbk_berry_lanes.inl) multiplies a string's segments in order.  No
// link factors in memory, no product tree (log2 L launches of workgroup matmuls: 0.5 of the 1.16 ms that were left after
// k_link_polar_reg).  The polar iteration of a link ends when every lane of the wavefront has converged.
struct WilsonSegArgs {
    WilsonBigArgs W;
    int seg_len, nseg;
    cd* segs;        // [ns][nseg][nocc^2]
    cd* prod;        // string s: prod + s * pstride
    size_t pstride;
};
template <int M>
__device__ __forceinline__ bool wilson_polar_reg(cd (&X)[M][M]) {
    bool converged = false;
    for (int it = 0; it < 200; ++it) {
        cd Y[M][M];
        double r2 = 0.0;
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = a; b < M; ++b) {
                cd acc{0.0, 0.0};
#pragma unroll
                for (int k = 0; k < M; ++k) cfmac(acc, X[k][a], X[k][b]);
                if (a == b) acc.y = 0.0;
                Y[a][b] = acc;
                Y[b][a] = cconj(acc);
                const double dx = acc.x - (a == b ? 1.0 : 0.0);
                r2 += (a == b ? 1.0 : 2.0) * (dx * dx + acc.y * acc.y);
            }
        if (!converged) {
            cd Z[M][M];
#pragma unroll
            for (int a = 0; a < M; ++a)
#pragma unroll
                for (int b = 0; b < M; ++b) {
                    cd acc{0.0, 0.0};
#pragma unroll
                    for (int k = 0; k < M; ++k) cfma(acc, X[a][k], Y[k][b]);
                    Z[a][b] = cd{1.5 * X[a][b].x - 0.5 * acc.x, 1.5 * X[a][b].y - 0.5 * acc.y};
                }
#pragma unroll
            for (int a = 0; a < M; ++a)
#pragma unroll
                for (int b = 0; b < M; ++b) X[a][b] = Z[a][b];
            if (r2 < 1e-14) converged = true;     // residual 1e-7 before this update, its square after it
        }
        if (__builtin_amdgcn_ballot_w64(!converged) == 0) break;
    }
    return converged;
}
template <int M>
__global__ __launch_bounds__(256) void k_wilson_seg_reg(const WilsonSegArgs S) {
    const WilsonBigArgs& A = S.W;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < A.ns * S.nseg;
    const int64_t tt = live ? t : A.ns * S.nseg - 1;
    const int64_t seg = tt / A.ns, s = tt - seg * A.ns;          // (neighbouring lanes: neighbouring strings)
    const int ncomp = A.v.ncomp;
    const int i0 = (int)seg * S.seg_len, i1 = min(i0 + S.seg_len, A.nlinks);
    const int64_t base = axis_offset(A.other, A.s0 + s);
    cd R[M][M];
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) R[a][b] = cd{a == b ? 1.0 : 0.0, 0.0};
    bool all_ok = true;
    // (a wavefront can straddle two segments, and the last segment of a string is shorter: a lane past the end of its segment
    // multiplies by the identity while its neighbours finish)
    const int my_len = i1 - i0;
    for (int li = 0; __builtin_amdgcn_ballot_w64(li < my_len) != 0; ++li) {
        const bool act = li < my_len;
        const int64_t p = base + (int64_t)(i0 + (act ? li : 0)) * A.sdir, q = p + A.sdir;
        cd X[M][M];
#pragma unroll
        for (int a = 0; a < M; ++a)
#pragma unroll
            for (int b = 0; b < M; ++b) X[a][b] = cd{0.0, 0.0};
        {
            const cd* up[M];
            const cd* uq[M];
#pragma unroll
            for (int a = 0; a < M; ++a) {
                up[a] = wf_at(A.v, A.occ[a], p);
                uq[a] = wf_at(A.v, A.occ[a], q);
            }
            for (int c = 0; c < ncomp; ++c) {
                cd pc[M], qc[M];
#pragma unroll
                for (int a = 0; a < M; ++a) {
                    pc[a] = up[a][c];
                    qc[a] = uq[a][c];
                }
#pragma unroll
                for (int a = 0; a < M; ++a)
#pragma unroll
                    for (int b = 0; b < M; ++b) cfmac(X[a][b], pc[a], qc[b]);
            }
        }
        if (!act) {
#pragma unroll
            for (int a = 0; a < M; ++a)
#pragma unroll
                for (int b = 0; b < M; ++b) X[a][b] = cd{a == b ? 1.0 : 0.0, 0.0};
        }
        all_ok = wilson_polar_reg<M>(X) && all_ok;
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
    if (!live) return;
    if (!all_ok) atomicExch(A.flags + 1, 1);
    cd* const o = S.segs + ((size_t)s * S.nseg + seg) * (M * M);
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
        for (int b = 0; b < M; ++b) o[a * M + b] = R[a][b];
}
struct WilsonTreeArgs {
    const cd* in;
    cd* out;
    int m, nlinks, st;
    int64_t ns;
};
// one level of the ordered pairwise product: out[j] = in[j] in[j + st] for j = 0, 2 st, 4 st, ...
__global__ __launch_bounds__(256) void k_wilson_tree(const WilsonTreeArgs A) {
    const int nn = A.m * A.m;
    const int npair = (A.nlinks + 2 * A.st - 1) / (2 * A.st);
    for (int64_t item = blockIdx.x; item < A.ns * npair; item += gridDim.x) {
        const int64_t s = item / npair;
        const int j = (int)(item - s * npair) * 2 * A.st;
        const cd* a = A.in + ((size_t)s * A.nlinks + j) * nn;
        cd* o = A.out + ((size_t)s * A.nlinks + j) * nn;
        if (j + A.st < A.nlinks)
            wg_matmul<false, false>(A.m, a, A.in + ((size_t)s * A.nlinks + j + A.st) * nn, o, 1.0, nullptr, 0.0);
        else
            for (int e = threadIdx.x; e < nn; e += 256) o[e] = a[e];
    }
}

struct CayleyArgs {
    const cd* prod;      // string s: prod + s * pstride
    size_t pstride;
    cd* ab;              // [ns][2][nocc^2]
    cd* herm;            // [ns][nocc^2]
    int m;
    double ca, sa;       // cos / sin of alpha
};
__global__ __launch_bounds__(256) void k_wilson_cayley(const CayleyArgs C) {
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    const int tid = threadIdx.x, m = C.m, nn = m * m;
    const cd* P = C.prod + (size_t)blockIdx.x * C.pstride;
    cd* A = C.ab + (size_t)blockIdx.x * 2 * nn;
    cd* B = A + nn;
    for (int e = tid; e < nn; e += 256) {
        const cd pv = P[e];
        const cd q{C.ca * pv.x + C.sa * pv.y, C.ca * pv.y - C.sa * pv.x};      // e^{-i alpha} P
        const double d = (e / m == e % m) ? 1.0 : 0.0;
        A[e] = cd{d + q.x, q.y};              // I + Q
        B[e] = cd{q.y, d - q.x};              // i (I - Q)
    }
    __syncthreads();
    for (int k = 0; k < m; ++k) {             // Gauss-Jordan on [A | B] with partial pivoting
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < m; i += 256) {
            const double a = cabs2(A[i * m + k]);
            if (a > best) {
                best = a;
                bi = i;
            }
        }
        s_val[tid] = best;
        s_idx[tid] = bi;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) {
                const double o = s_val[tid + w];
                const int oi = s_idx[tid + w];
                if (o > s_val[tid] || (o == s_val[tid] && oi < s_idx[tid])) {
                    s_val[tid] = o;
                    s_idx[tid] = oi;
                }
            }
            __syncthreads();
        }
        const int r = s_idx[0];
        const cd piv = A[r * m + k];
        __syncthreads();
        const double ip = 1.0 / fmax(cabs2(piv), 1e-300);
        const cd inv{piv.x * ip, -piv.y * ip};
        // row k <- (row r) / pivot, row r <- old row k   (columns k.. of A, all of B)
        for (int e = tid; e < 2 * m; e += 256) {
            cd* row = e < m ? A : B;
            const int j = e < m ? e : e - m;
            if (e < m && j < k) continue;
            const cd top = row[k * m + j], low = row[r * m + j];
            row[r * m + j] = top;
            row[k * m + j] = cmul(low, inv);
        }
        __syncthreads();
        const int wa = m - k - 1, width = wa + m;
        for (int e = tid; e < m * width; e += 256) {
            const int i = e / width, jj = e - i * width;
            if (i == k) continue;
            const cd f = A[i * m + k];
            cd* row = jj < wa ? A : B;
            const int j = jj < wa ? k + 1 + jj : jj - wa;
            const cd t = cmul(f, row[k * m + j]);
            row[i * m + j] = cd{row[i * m + j].x - t.x, row[i * m + j].y - t.y};
        }
        __syncthreads();
    }
    cd* H = C.herm + (size_t)blockIdx.x * nn;
    for (int e = tid; e < nn; e += 256) {
        const int a = e / m, b = e - a * m;
        const cd u = B[a * m + b], w = B[b * m + a];
        H[e] = cd{0.5 * (u.x + w.x), 0.5 * (u.y - w.y)};
    }
}

// eigenvalues h[j][s] of the Cayley transforms -> sort(-angle) per string, and max |h| per string
__global__ __launch_bounds__(64) void k_wilson_phases(const double* __restrict__ h, const int64_t ns, const int m,
                                                     const double alpha, double* __restrict__ out,
                                                     double* __restrict__ hmax) {
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= ns) return;
    double* o = out + s * m;
    double big = 0.0;
    for (int j = 0; j < m; ++j) {
        const double hv = h[(int64_t)j * ns + s];
        big = fmax(big, fabs(hv));
        const double th = alpha + 2.0 * atan(hv);
        const double ph = -atan2(sin(th), cos(th));
        int pos = j;
        while (pos > 0 && o[pos - 1] > ph) {
            o[pos] = o[pos - 1];
            --pos;
        }
        o[pos] = ph;
    }
    hmax[s] = big;
}
