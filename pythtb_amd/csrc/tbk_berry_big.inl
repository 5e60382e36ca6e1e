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
// their ordered product (pythtb.py:3825-3838).  Above TBK_MAX_NOCC the same per-thread routines as for small
// sets (one-sided Jacobi polar factor, Householder-Hessenberg + shifted QR) run on matrices kept in a global
// per-thread workspace instead of registers/local memory: thread = (string, segment of links), then one
// thread per string.  Slow per thread, but this is the rarely used corner of the path.

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

// ---- Wilson-loop eigenphases for large occupied sets: matrices in a per-thread global workspace
struct ChainBigArgs {
    WfsView v;
    const int* occ;
    int nocc;
    int nlinks;
    int64_t sdir;
    AxisSet other;
    int64_t nstrings;
    int seg_len, nseg;
    cd* partial;   // [nseg][nstrings][nocc*nocc]
    cd* work;      // [threads][4*nocc*nocc]
    double* out;   // [nstrings][nocc]
    int* flags;
};

__global__ __launch_bounds__(64) void k_chain_partial_big(const ChainBigArgs A) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= A.nstrings * A.nseg) return;
    const int64_t seg = t / A.nstrings, s = t - seg * A.nstrings;
    const int nocc = A.nocc, ncomp = A.v.ncomp, nn = nocc * nocc;
    const int64_t plane = A.v.npts * ncomp;
    cd* R = A.work + (size_t)t * 4 * nn;
    cd *M = R + nn, *V = M + nn, *T = V + nn;
    const int i0 = (int)seg * A.seg_len, i1 = min(i0 + A.seg_len, A.nlinks);
    const cd* P = A.v.data + (axis_offset(A.other, s) + (int64_t)i0 * A.sdir) * ncomp;
    const int64_t step = A.sdir * ncomp;
    for (int a = 0; a < nocc; ++a)
        for (int b = 0; b < nocc; ++b) R[a * nocc + b] = cd{a == b ? 1.0 : 0.0, 0.0};
    for (int i = i0; i < i1; ++i, P += step) {
        link_matrix_dyn(P, P + step, A.occ, nocc, ncomp, plane, M);
        polar_dyn(nocc, M, V, T);                       // M <- U Vh of its SVD   (pythtb.py:3825-3826)
        for (int a = 0; a < nocc; ++a)
            for (int b = 0; b < nocc; ++b) {
                cd acc{0.0, 0.0};
                for (int j = 0; j < nocc; ++j) cfma(acc, R[a * nocc + j], M[j * nocc + b]);
                T[a * nocc + b] = acc;
            }
        for (int e = 0; e < nn; ++e) R[e] = T[e];
    }
    cd* o = A.partial + ((int64_t)seg * A.nstrings + s) * nn;
    for (int e = 0; e < nn; ++e) o[e] = R[e];
}

__global__ __launch_bounds__(64) void k_chain_final_big(const ChainBigArgs A) {
    const int64_t s = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (s >= A.nstrings) return;
    const int nocc = A.nocc, nn = nocc * nocc;
    cd* R = A.work + (size_t)s * 4 * nn;
    cd *T = R + nn, *ev = T + nn, *rc = ev + nocc, *rs = rc + nocc;
    for (int e = 0; e < nn; ++e) R[e] = A.partial[s * nn + e];
    for (int g = 1; g < A.nseg; ++g) {
        const cd* M = A.partial + ((int64_t)g * A.nstrings + s) * nn;
        for (int a = 0; a < nocc; ++a)
            for (int b = 0; b < nocc; ++b) {
                cd acc{0.0, 0.0};
                for (int j = 0; j < nocc; ++j) cfma(acc, R[a * nocc + j], M[j * nocc + b]);
                T[a * nocc + b] = acc;
            }
        for (int e = 0; e < nn; ++e) R[e] = T[e];
    }
    if (!eigvals_dyn(nocc, R, ev, rc, rs)) atomicExch(A.flags + 1, 1);
    double* o = A.out + s * nocc;   // sort(-angle(eigvals))   (pythtb.py:3834-3838)
    for (int j = 0; j < nocc; ++j) {
        const double ph = -atan2(ev[j].y, ev[j].x);
        int pos = j;
        while (pos > 0 && o[pos - 1] > ph) {
            o[pos] = o[pos - 1];
            --pos;
        }
        o[pos] = ph;
    }
}

