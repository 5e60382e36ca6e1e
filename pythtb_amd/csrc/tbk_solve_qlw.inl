// tbk_solve_qlw.inl -- included by tbk_solve.hip (after tbk_solve_ql16.inl, whose row sums it uses).
//
// n = 17..64 states per k, large batches: the DIRECT Hermitian eigen-solver (what LAPACK's zhetrd + zsteqr/zungtr
// do for the reference's numpy.linalg.eigh / eigvalsh, pythtb.py:939-947) as three kernels, because its three parts
// parallelise differently:
//
//  1. k_tridiag_lds      one workgroup per matrix, A in LDS.  Householder reflections H_k = I - beta u u^+ bring A to a
//                        (complex) tridiagonal form; u_k is kept where LAPACK keeps it, in the dead column k below the
//                        diagonal.  With eigenvectors wanted the unitary Z = H_0 ... H_{n-3} D (D: the diagonal unitary
//                        that makes the subdiagonal real) is then accumulated BACKWARDS, IN PLACE of A (each H_k only
//                        touches the trailing block it was built from: n^3/3 work instead of n^3/2, and no second
//                        matrix in LDS) and written to the output array, band slot b = column b.
//                        O(n^3) work, n^2-way parallel, LDS-bandwidth bound.
//  2. k_tridiag_ql_lanes the implicit-shift QL iteration on (d, e) is a SEQUENTIAL scalar recurrence (~n^2 plane
//                        rotations, each a dozen dependent fp64 operations): one LANE per matrix, d and e in LDS laid
//                        out [j][lane] (conflict-free whatever j each lane is at), all lanes of a wavefront stepping
//                        through the same positions under EXEC masks.  It writes the sorted eigenvalues (or the mesh's
//                        minimum gaps), the sorting permutation and -- with eigenvectors wanted -- the rotation
//                        sequence (c, s) of every sweep into a workspace.
//  3. k_ql_replay_reg / k_ql_replay_reg64 / k_ql_backtransform: the recorded rotations are replayed on Z.  They are real
//                        and act on columns, so the 2n real rows are independent.  A lane keeps its row in REGISTERS
//                        (static indices: every position of the range unrolled, taken or not under a per-lane
//                        predicate) -- 32 lanes per matrix up to n = 32, one wavefront per part (real | imaginary)
//                        from n = 40; n = 33..39 keep Z in LDS with dynamic positions (k_ql_backtransform).  Columns
//                        leave in ascending order of their eigenvalue.
//
// The cyclic Jacobi kernels these replace do 7-9 sweeps of n(n-1)/2 rotations over A and V (~80 n^3 flops, all of it
// LDS traffic); this path is ~(16/3 + 8/3 + 6) n^3.  Every point is solved on its own: periodic images, halo rows and
// shard windows are bit-identical by construction.  Batches are processed in chunks so that the rotation workspace
// stays below ~4 GiB.

template <int RW>
__device__ __forceinline__ double rw_allsum(double v) {   // sum over the RW (32 | 64) lanes x of a strip, same bits in every lane
    v = row_allsum(v);
    v += __shfl_xor(v, 16);
    if constexpr (RW == 64) v += __shfl_xor(v, 32);
    return v;
}

// lanes of ONE wavefront exchanging data through LDS: its LDS operations execute in order; this only keeps the compiler from
// moving accesses across the exchange
__device__ __forceinline__ void qlw_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct QlwWork {           // chunk workspace (device pointers; see launch_qlw)
    double2* de;           // [n][nchunk]      (d_j, e_j)
    double2* rot;          // [nchunk][cap]    (c, s) in the order they were applied
    unsigned* swp;         // [nchunk][scap]   one word per sweep: first position | rotations << 8
    int* nsw;              // [nchunk]
    int* rank;             // [n][nchunk]      rank[b][id] = ascending rank of the eigenvalue of column b
    int64_t cap;
    int scap;
    // n <= 32 with eigenvectors (tbk_solve_tw32.inl): what k_tw32_vectors needs from the QL kernel; lam == nullptr: not asked for
    double* lam;           // [n][nchunk]      the eigenvalue left at position j
    uint2* meta;           // [nchunk]         {split mask of T (bit i: e_i negligible), 1 = two eigenvalues of one block closer than gaptol |T|}
    int* list;             // [nchunk]         matrices left to the rotation replay
    int* count;            //                  their number
    double gaptol;
    unsigned long long* listed;   // the context's count of listed matrices (tbk_ctx_solver_stats)
    int all_q;             //                  k_tw32_vectors<.., true>: Q only (every matrix of the call takes the rotation replay)
    cd* refl;              // [nchunk][hh32_rec_size(NM)]  the reflector records of k_hh32<.., 2, NM> (nullptr: Q sits in the output array)
};

static size_t qlw_lds1_bytes(int n, int nR, int rw, int nt) {
    const int ld = n | 1, hs = nt / rw;
    size_t b = (size_t)n * ld * sizeof(cd);                       // A / Z
    const int nw = nt / 64;
    b += (size_t)(nw * rw + rw + hs * rw) * sizeof(cd);           // ubuf (a private copy per wavefront), qbuf, pbuf
    b += (size_t)(std::max(n, nR) + 2 * n) * sizeof(cd);          // eo / phases, dphase, tsub
    b += (size_t)2 * n * sizeof(double);                          // tau, eb
    return (b + 15) & ~(size_t)15;
}

// MODE 0: k list, 1: regular mesh into a wf_array, 2: supplied matrices.  Block b works on matrix id0 + b of the batch
// (chunk-local index b).
template <int MODE, bool VEC, int RW, int NT>
__global__ __launch_bounds__(NT) void k_tridiag_lds(const ModelView mv, const int64_t nk, const ListArgs L, const GridArgs G,
                                                     const int64_t id0, const int64_t nchunk, double2* __restrict__ de) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int HS = NT / RW;
    const int n = mv.nsta, ld = n | 1;
    const int tid = threadIdx.x, x = tid & (RW - 1), h = tid / RW;
    cd* A = (cd*)lds_raw;
    // Every wavefront holds all RW rows (once or twice), so u is computed by each wavefront for itself and passed between
    // its lanes through a PRIVATE LDS copy: a wavefront's LDS operations execute in order, no workgroup barrier.  (q the same
    // way would save another barrier per reflection but costs the LDS that lets two 64 x 64 matrices share a CU: slower.)
    constexpr int NW = NT / 64;
    const int wv = tid >> 6;
    const bool first = (tid & 63) < RW;    // the lanes of a wavefront that write its private copy
    cd* ubuf = A + n * ld + wv * RW;       // [NW][RW]
    cd* qbuf = A + n * ld + NW * RW;       // [RW]
    cd* pbuf = qbuf + RW;                  // [HS][RW]
    cd* eo = pbuf + HS * RW;               // [max(n, nR)]
    cd* dphase = eo + (n > mv.nR ? n : mv.nR);
    cd* tsub = dphase + n;
    double* tau = (double*)(tsub + n);
    double* eb = tau + n;
    const int64_t idc = blockIdx.x, id = id0 + idc;

    double kk[4] = {0.0, 0.0, 0.0, 0.0};
    bool wrap[4] = {false, false, false, false};
    if constexpr (MODE == 0) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
    } else if constexpr (MODE == 1) {
        grid_point(G, id, kk, wrap);
    }
    assemble_lds<MODE, NT>(mv, L, id, kk, A, ld, eo, tid);
    __syncthreads();
    if (VEC && tid < n) {
        cd f{1.0, 0.0};
        if constexpr (MODE != 2) f = cconj(expi2pi(kdot(kk, mv.orb[tid])));
        if constexpr (MODE == 1) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (wrap[d]) f = cmul(f, G.pbc[d * n + tid]);
        }
        eo[tid] = f;
    }

    // ---- 1. reflections k = 0 .. n-3.  Thread (x, h): row x, columns c = k+1+h, k+1+h+HS, ...
    for (int k = 0; k + 2 < n; ++k) {
        const bool below = x > k && x < n;
        const cd colx = below ? A[x * ld + k] : cd{0.0, 0.0};
        // |rows > k+1 of the column|^2: a reflection is needed iff this is non-zero (decided on this part alone, like LAPACK's
        // zlarfg: through the sum with |alpha|^2 entries below ~1e-8 |alpha| would be dropped)
        const double rest = rw_allsum<RW>(x > k + 1 ? cabs2(colx) : 0.0);
        const cd alpha = A[(k + 1) * ld + k];
        const double absa2 = cabs2(alpha);
        const double sigma = rest + absa2;                   // |column below the diagonal|^2
        cd tK = alpha;
        double beta = 0.0;
        if (rest > 0.0) {      // (uniform) something to annihilate below the subdiagonal
            const double inv_n = rsqrt_full(sigma), nrm = sigma * inv_n;
            double absa = 0.0;
            cd ph{1.0, 0.0};
            if (absa2 > 0.0) {
                const double inv_a = rsqrt_full(absa2);
                absa = absa2 * inv_a;
                ph = cd{alpha.x * inv_a, alpha.y * inv_a};
            }
            // u = column + phase * norm * e_{k+1};  H maps the column onto -phase * norm * e_{k+1}
            const cd u = x == k + 1 ? cd{ph.x * (absa + nrm), ph.y * (absa + nrm)} : colx;
            beta = 1.0 / (nrm * (nrm + absa));               // 2 / (u^+ u)
            tK = cd{-ph.x * nrm, -ph.y * nrm};
            if (first) ubuf[x] = u;
            qlw_wave_sync();
            cd p{0.0, 0.0};
            if (below)
                for (int c = k + 1 + h; c < n; c += HS) cfma(p, A[x * ld + c], ubuf[c]);
            cd* pb = pbuf;
            pb[h * RW + x] = p;
            __syncthreads();
            cd ps{0.0, 0.0};
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) {
                const cd t = pb[hh * RW + x];
                ps.x += t.x;
                ps.y += t.y;
            }
            ps = cd{ps.x * beta, ps.y * beta};
            // kappa = beta/2 u^+ p  (real: u^+ A u of a Hermitian A)
            const double kappa = 0.5 * beta * rw_allsum<RW>(u.x * ps.x + u.y * ps.y);
            const cd q = below ? cd{ps.x - kappa * u.x, ps.y - kappa * u.y} : cd{0.0, 0.0};
            if (h == 0) {
                qbuf[x] = q;
                if (below) A[x * ld + k] = u;                // the reflector stays in its column (LAPACK's convention)
            }
            __syncthreads();
            if (below)
                for (int c = k + 1 + h; c < n; c += HS) {    // A[x][c] -= u_x conj(q_c) + q_x conj(u_c)
                    const cd uc = ubuf[c], qc = qbuf[c];
                    cd a = A[x * ld + c];
                    a.x -= (u.x * qc.x + u.y * qc.y) + (q.x * uc.x + q.y * uc.y);
                    a.y -= (u.y * qc.x - u.x * qc.y) + (q.y * uc.x - q.x * uc.y);
                    A[x * ld + c] = a;
                }
            __syncthreads();
        }
        if (tid == 0) {
            tau[k] = beta;
            tsub[k] = tK;
        }
    }
    if (tid == 0) {
        if (n >= 2) tsub[n - 2] = A[(n - 1) * ld + (n - 2)];   // never reflected
        if (n >= 2) tau[n - 2] = 0.0;
    }
    __syncthreads();
    // subdiagonal moduli and the phases D_{k+1} = D_k t_k / |t_k|
    if (tid < n) {
        double mag = 0.0;
        cd f{1.0, 0.0};
        if (tid + 1 < n) {
            const cd t = tsub[tid];
            const double t2 = cabs2(t);
            if (t2 > 0.0) {
                const double inv = rsqrt_full(t2);
                mag = t2 * inv;
                f = cd{t.x * inv, t.y * inv};
            }
        }
        eb[tid] = mag;
        tsub[tid] = f;
    }
    __syncthreads();
    if (tid == 0) {
        cd delta{1.0, 0.0};
        dphase[0] = delta;
        for (int k = 0; k + 1 < n; ++k) {
            delta = cmul(delta, tsub[k]);
            dphase[k + 1] = delta;
        }
    }
    if (tid < n) de[(int64_t)tid * nchunk + idc] = double2{A[tid * ld + tid].x, eb[tid]};
    if constexpr (!VEC) return;
    __syncthreads();

    // ---- 2. Z = H_0 (H_1 ( ... (H_{n-3} D))) in place: before step k the block [k+2.., k+2..] holds the product so far;
    // row and column k+1 join it as D's entry, then H_k acts from the left on rows k+1.. (its reflector sits in column k,
    // which the block has not reached yet).  Thread (c, h): column c = x, rows r = k+1+h, k+1+h+HS, ...
    if (tid == 0) A[(n - 1) * ld + (n - 1)] = dphase[n - 1];
    for (int k = n - 2; k >= 0; --k) {
        cd* ub = ubuf;                       // (private to the wavefront)
        cd* pb = pbuf;
        const double beta = tau[k];
        // the reflector of step k, for this wavefront; the new row / column k+1 of the block (nobody reads these places
        // before the barrier: column k+1 below the diagonal held the reflector of step k+1, consumed one step ago)
        if (first && x > k && x < n) ub[x] = A[x * ld + k];
        if (h == 1 % HS && x > k + 1 && x < n) A[(k + 1) * ld + x] = cd{0.0, 0.0};
        if (h == 0 && x > k + 1 && x < n) A[x * ld + (k + 1)] = cd{0.0, 0.0};
        if (tid == 0) A[(k + 1) * ld + (k + 1)] = dphase[k + 1];
        __syncthreads();
        if (beta != 0.0) {   // (uniform)
            const bool col = x > k && x < n;
            cd t{0.0, 0.0};
            if (col)
                for (int r = k + 1 + h; r < n; r += HS) cfmac(t, ub[r], A[r * ld + x]);   // t_c += conj(u_r) Z[r][c]
            pb[h * RW + x] = t;
            __syncthreads();
            cd ts{0.0, 0.0};
#pragma unroll
            for (int hh = 0; hh < HS; ++hh) {
                const cd v = pb[hh * RW + x];
                ts.x += v.x;
                ts.y += v.y;
            }
            ts = cd{ts.x * beta, ts.y * beta};
            if (col)
                for (int r = k + 1 + h; r < n; r += HS) {    // Z[r][c] -= u_r (beta t_c)
                    const cd ur = ub[r];
                    cd z = A[r * ld + x];
                    z.x -= ur.x * ts.x - ur.y * ts.y;
                    z.y -= ur.x * ts.y + ur.y * ts.x;
                    A[r * ld + x] = z;
                }
        }
    }
    __syncthreads();
    if (h == 0 && x > 0 && x < n) A[x * ld] = cd{0.0, 0.0};          // column 0 held the reflector of step 0
    if (h == 1 % HS && x > 0 && x < n) A[x] = cd{0.0, 0.0};
    if (tid == 0) A[0] = cd{1.0, 0.0};
    __syncthreads();
    // rows carry the orbital phase (and periodic-image phases) from here on: the rotations of step 3 act on columns
    for (int e = tid; e < n * n; e += NT) {
        const int b = e / n, o = e - b * n;
        const cd val = cmul(A[o * ld + b], eo[o]);
        if constexpr (MODE == 1) wf_at(G.wv, b, id)[o] = val;
        else L.evec[((int64_t)b * nk + id) * n + o] = val;
    }
}

// ---- one lane per matrix: implicit-shift QL (EISPACK tql2 recurrences) on (d, e)
// RECM 0: eigenvalues only; 1: the rotations recorded for the replay; 2: no record, but what k_tw32_vectors needs (ranks, the eigenvalue
// of every position, the splitting of T, close pairs listed).  LIST: the matrices W.list[0 .. *W.count) instead of the chunk (the record
// for the matrices k_tw32_vectors left to the replay; the grid is sized for the chunk, blocks past the count leave at once).
template <int MODE, int RECM, bool LIST = false>
__global__ __launch_bounds__(64) void k_tridiag_ql_lanes(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk,
                                                         const QlwWork W, double* __restrict__ eval, const GridArgs G,
                                                         int* flags) {
    constexpr bool REC = RECM == 1;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* D = (double*)lds_raw;          // [n][64]
    double* E = D + n * 64;                // [n][64]
    const int lane = threadIdx.x;
    int64_t nlive = nchunk;
    if constexpr (LIST) {
        nlive = *W.count;
        if ((int64_t)blockIdx.x * 64 >= nlive) return;
    }
    const int64_t it = (int64_t)blockIdx.x * 64 + lane;
    const bool has = it < nlive;
    const int64_t ic = LIST ? (int64_t)W.list[has ? it : nlive - 1] : (has ? it : nchunk - 1);
    const int64_t idc = ic;                // (every write below is under `has`)
    for (int j = 0; j < n; ++j) {
        const double2 v = W.de[(int64_t)j * nchunk + ic];
        D[j * 64 + lane] = v.x;
        E[j * 64 + lane] = j + 1 < n ? v.y : 0.0;
    }
    double2* rot = W.rot + ic * W.cap;
    unsigned* swp = W.swp + ic * W.scap;
    int64_t seq = 0;
    int isw = 0;
    bool overflow = false;
    int l = 0;
    bool done = !has;
    const int max_iter = 30 * n;
    const unsigned long long top = 1ull << (n - 1);
    // negligible couplings: bit j of `negl` <=> |e_j| <= eps (|d_j| + |d_j+1|)   (e_{n-1} = 0: always).  Scanned once; a sweep
    // over [l, m) changes e_l .. e_{m-1} and d_l .. d_m only, and those bits are renewed as the new values appear.
    const double eps = 2.220446049250313e-16;
    unsigned long long negl = top;
    auto rescan = [&]() {
        negl = top;
        double dj = D[lane];
        for (int j = 0; j + 1 < n; ++j) {
            const double dn = D[(j + 1) * 64 + lane];
            if (fabs(E[j * 64 + lane]) <= eps * (fabs(dj) + fabs(dn))) negl |= 1ull << j;
            dj = dn;
        }
    };
    rescan();
    const unsigned split0 = (unsigned)negl;              // (n <= 32 where it is used) T as it splits before the first sweep
    for (int iter = 0;; ++iter) {
        int m = n - 1;
        if (!done) {
            const unsigned long long open = ~negl & (~0ull << l) & (top - 1);
            if (open == 0) {
                done = true;
            } else {
                l = __builtin_ctzll(open);
                m = __builtin_ctzll(negl & (~0ull << l));
            }
        }
        if (__all(done)) break;
        if (iter >= max_iter) {
            if (!done) atomicExch(flags, 1);
            break;
        }
        double sn = 1.0, cs = 1.0, pp = 0.0, g = 0.0;
        bool alive = !done;
        if (!done) {
            // Wilkinson-type shift from the leading 2 x 2 of the block (only the speed of convergence depends on its
            // accuracy: hardware reciprocal / square root estimates are good enough)
            const double dl = D[l * 64 + lane], dl1 = D[(l + 1) * 64 + lane], el = E[l * 64 + lane], dm = D[m * 64 + lane];
            const double gs = (dl1 - dl) * (0.5 * __builtin_amdgcn_rcp(el));
            const double r = __builtin_amdgcn_sqrt(fma(gs, gs, 1.0));
            g = dm - dl + el * __builtin_amdgcn_rcp(gs + copysign(r, gs));
        }
        // the wavefront walks i = max(m) - 1 .. min(l); a lane takes part inside its own block [l, m)
        int hi = done ? 1 : m, lo = done ? n - 1 : l;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            hi = max(hi, __shfl_xor(hi, o));
            lo = min(lo, __shfl_xor(lo, o));
        }
        int cnt = 0;
        // e_i, d_i of the position are loaded one position ahead; d_{i+1} as it was BEFORE the sweep is the d_i of the
        // position above (position i+1 wrote d_{i+2})
        double e_cur = E[(hi - 1) * 64 + lane], d_cur = D[(hi - 1) * 64 + lane], d_up = D[hi * 64 + lane];
        double d_new_up = 0.0;                            // the NEW d_{i+2} while position i is worked on
        for (int i = hi - 1; i >= lo; --i) {
            double e_nx = 0.0, d_nx = 0.0;
            if (i > lo) {
                e_nx = E[(i - 1) * 64 + lane];
                d_nx = D[(i - 1) * 64 + lane];
            }
            if (alive && i >= l && i < m) {
                const double ei = e_cur, di = d_cur, di1 = d_up;
                const double f = sn * ei, b = cs * ei;
                const double t = f * f + g * g;
                if (t > 0.0) {
                    const double inv = rsqrt_full(t), r = t * inv;
                    E[(i + 1) * 64 + lane] = r;
                    sn = f * inv;
                    cs = g * inv;
                    const double gg = di1 - pp;
                    const double r2 = (di - gg) * sn + 2.0 * cs * b;
                    pp = sn * r2;
                    const double dn = gg + pp;
                    D[(i + 1) * 64 + lane] = dn;
                    g = cs * r2 - b;
                    if (i + 1 < m) {                      // bit i+1: e_{i+1} = r, d_{i+1} = dn, d_{i+2} = d_new_up, all final
                        const unsigned long long bit = 1ull << (i + 1);
                        negl = r <= eps * (fabs(dn) + fabs(d_new_up)) ? negl | bit : negl & ~bit;
                    }
                    d_new_up = dn;
                    if (REC) {
                        if (seq + cnt < W.cap) rot[seq + cnt] = double2{cs, sn};
                        else overflow = true;
                    }
                    ++cnt;
                } else {                                 // r == 0 (underflow): tql2's recovery
                    D[(i + 1) * 64 + lane] = di1 - pp;
                    E[(i + 1) * 64 + lane] = 0.0;          // (e_{i+1} = r = 0)
                    alive = false;
                }
            }
            d_up = d_cur;
            e_cur = e_nx;
            d_cur = d_nx;
        }
        bool again = false;
        if (!done) {
            if (alive) {
                const double dl = D[l * 64 + lane] - pp;
                D[l * 64 + lane] = dl;
                E[l * 64 + lane] = g;
                const unsigned long long bit = 1ull << l;  // bit l: e_l = g, d_l = dl, d_{l+1} = d_new_up
                negl = fabs(g) <= eps * (fabs(dl) + fabs(d_new_up)) ? negl | bit : negl & ~bit;
            } else {
                again = true;                            // (underflow stop: renew every bit)
            }
            E[m * 64 + lane] = 0.0;
            if (REC) {
                if (isw < W.scap) swp[isw] = (unsigned)(m - 1) | ((unsigned)cnt << 8);
                else overflow = true;
                ++isw;
                seq += cnt;
            }
        }
        if (__any(again)) {
            const unsigned long long keep = negl;
            rescan();
            if (!again) negl = keep;
        }
    }
    if (REC) {
        // (a matrix whose record overflowed is replayed as "no sweeps": its sweep words would otherwise send the register
        // replay kernels past the end of the truncated record -- past the workspace for the last matrix of a chunk; the
        // flag below makes the caller repeat the call on the Jacobi kernels anyway)
        if (has) W.nsw[idc] = overflow ? 0 : (isw < W.scap ? isw : W.scap);
        if (overflow) atomicExch(flags + 2, 1);
    }
    // stable ascending ranks; E is free now: E[r] <- the column of rank r (as a double)
    constexpr bool tw = RECM == 2;                       // positions, splitting and close pairs for k_tw32_vectors
    bool flagged = false;
    double thr = 0.0;
    if (tw) {
        double tnorm = 0.0;
        for (int a = 0; a < n; ++a) tnorm = fmax(tnorm, fabs(D[a * 64 + lane]));
        thr = W.gaptol * tnorm;
    }
    for (int a = 0; a < n; ++a) {
        const double da = D[a * 64 + lane];
        int r = 0;
        for (int b = 0; b < n; ++b) {
            const double db = D[b * 64 + lane];
            r += (db < da || (db == da && b < a)) ? 1 : 0;
            // two eigenvalues of one unreduced block (no split between positions a and b) closer than gaptol |T|: their
            // twisted-factorisation vectors would be nearly parallel
            if (tw && b > a) flagged = flagged || (((split0 >> a) & ((1u << (b - a)) - 1u)) == 0 && !(fabs(da - db) >= thr));
        }
        E[r * 64 + lane] = (double)a;
        if (tw && has) W.lam[(int64_t)a * nchunk + idc] = da;
    }
    if (tw && has) {
        W.meta[idc] = uint2{split0, flagged ? 1u : 0u};
        if (flagged) W.list[atomicAdd(W.count, 1)] = (int)idc;
    }
    double prev = 0.0;
    for (int r = 0; r < n; ++r) {
        const int a = (int)E[r * 64 + lane];
        const double v = D[a * 64 + lane];
        if (RECM != 0 && has) W.rank[(int64_t)a * nchunk + idc] = r;
        if constexpr (LIST) continue;                    // (the eigenvalues and the gaps went out with the main launch: both forms of a call keep the same bits)
        if constexpr (MODE == 1) {
            if (r > 0) {
                double gap = has ? v - prev : INFINITY;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gap = fmin(gap, __shfl_xor(gap, o));
                if (lane == 0) {
                    unsigned long long* slot = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n + (r - 1);
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(gap, 0.0));
                    if (bits < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, bits);
                }
            }
            prev = v;
        } else {
            if (has) eval[(int64_t)r * nk + id0 + idc] = v;
        }
    }
}

// ---- replay the rotations on the rows of Z, leave in eigenvalue order.  Block = one matrix, Z in LDS, thread = (row x,
// real | imaginary part).  rot holds, sweep after sweep, the rotations of positions ihi, ihi-1, ..., ihi-cnt+1 in that
// order; the stream comes from HBM, so it passes through a small LDS ring that is refilled NT entries at a time, the next
// refill already in flight (in a register per thread) while the current entries are applied.
// Two matrices of n = 64 fit the LDS of a CU, i.e. one wavefront per SIMD: the kernel is bound by the issue latency of one
// wavefront's dependent instructions (PMC: ~21 instructions per rotation, ~7 cycles each; profiles/qlw_pmc.sh).
// The register-resident kernels below (k_ql_replay_reg*) are the default outside n = 33..39.  What does NOT work for them is
// any WAVE-UNIFORM way of skipping the positions a sweep does not touch -- scalar branches per position or per group of four
// (sweeps recorded padded to whole groups), a jump into the unrolled chain at the sweep's first position, one if / else per
// sweep between range variants, or sweeps padded to the full range as straight-line code under an occupancy bound: the
// register allocator then splits the row's live ranges around the regions (256 VGPRs + AGPR copies + hundreds of moves for a
// 64-register row, or spills).  Per-lane predicates (EXEC-masked, in place) compile to exactly the row plus temporaries.
#define TBK_QLW_RING 256   // entries (4 KB); a power of two, at least 2 NT
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void k_ql_backtransform(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk,
                                                          const QlwWork W, cd* __restrict__ evec, const WfsView wv) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    cd* Z = (cd*)lds_raw;
    double* Zr = (double*)lds_raw;
    const int ld = n | 1;
    double2* ring = (double2*)(Z + n * ld);             // [TBK_QLW_RING]
    unsigned* swl = (unsigned*)(ring + TBK_QLW_RING);   // [scap] the sweep words of this matrix
    const int tid = threadIdx.x;
    const int64_t idc = blockIdx.x, id = id0 + idc;
    const double2* __restrict__ rot = W.rot + idc * W.cap;
    const int cap = (int)W.cap;
    const int nsw = W.nsw[idc];
    int filled = 0;
    double2 pre = tid < cap ? rot[tid] : double2{1.0, 0.0};   // the first refill, in flight while Z is loaded
    for (int s = tid; s < nsw; s += NT) swl[s] = W.swp[idc * W.scap + s];
    for (int e = tid; e < n * n; e += NT) {
        const int b = e / n, o = e - b * n;
        cd v;
        if constexpr (MODE == 1) v = wf_at(wv, b, id)[o];
        else v = evec[((int64_t)b * nk + id) * n + o];
        Z[o * ld + b] = v;
    }
    const int x = tid >> 1, part = tid & 1;
    const bool rowok = x < n;
    double* row = Zr + (size_t)(rowok ? x : 0) * ld * 2 + part;      // element i of the row: row[2 i]
    int seq = 0, s = 0;
    while (s < nsw) {
        if (filled + NT <= seq + TBK_QLW_RING && filled < cap) {   // (uniform) room for the refill that is in flight
            ring[(filled + tid) & (TBK_QLW_RING - 1)] = pre;
            filled += NT;
            pre = filled + tid < cap ? rot[filled + tid] : double2{1.0, 0.0};
        }
        __syncthreads();
        while (s < nsw) {
            const unsigned desc = (unsigned)__builtin_amdgcn_readfirstlane((int)swl[s]);
            const int ihi = (int)(desc & 0xffu), cnt = (int)(desc >> 8);
            if (seq + cnt > filled) {
                if (filled >= cap) s = nsw;   // the recording overflowed (flagged by the QL kernel): give up
                break;
            }
            ++s;
            if (cnt == 0) continue;
            if (rowok) {
                // z_{i+1}' = s z_i + c z,  z' = c z_i - s z  with z carried down the sweep
                double z = row[2 * (ihi + 1)];
                int j = 0;
                for (; j + 4 <= cnt; j += 4) {
                    const int i = ihi - j, e = seq + j;
                    const double2 r0 = ring[e & (TBK_QLW_RING - 1)], r1 = ring[(e + 1) & (TBK_QLW_RING - 1)];
                    const double2 r2 = ring[(e + 2) & (TBK_QLW_RING - 1)], r3 = ring[(e + 3) & (TBK_QLW_RING - 1)];
                    const double z0 = row[2 * i], z1 = row[2 * (i - 1)], z2 = row[2 * (i - 2)], z3 = row[2 * (i - 3)];
                    row[2 * (i + 1)] = fma(r0.x, z, r0.y * z0);
                    z = fma(-r0.y, z, r0.x * z0);
                    row[2 * i] = fma(r1.x, z, r1.y * z1);
                    z = fma(-r1.y, z, r1.x * z1);
                    row[2 * (i - 1)] = fma(r2.x, z, r2.y * z2);
                    z = fma(-r2.y, z, r2.x * z2);
                    row[2 * (i - 2)] = fma(r3.x, z, r3.y * z3);
                    z = fma(-r3.y, z, r3.x * z3);
                }
                for (; j < cnt; ++j) {
                    const int i = ihi - j;
                    const double2 r = ring[(seq + j) & (TBK_QLW_RING - 1)];
                    const double zi = row[2 * i];
                    row[2 * (i + 1)] = fma(r.x, z, r.y * zi);
                    z = fma(-r.y, z, r.x * zi);
                }
                row[2 * (ihi - cnt + 1)] = z;
            }
            seq += cnt;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) {
        const int b = e / n, o = e - b * n;
        const int r = W.rank[(int64_t)b * nchunk + idc];
        const cd v = Z[o * ld + b];
        if constexpr (MODE == 1) wf_at(wv, r, id)[o] = v;
        else evec[((int64_t)r * nk + id) * n + o] = v;
    }
}

// ---- the replay with the rows of Z in REGISTERS, n <= 32 (TBK_QLW_REPLAY_REG): the form k_ql16_replay has -- 32 lanes per
// matrix, lane = row, z[NMAX] complex with static indices, EVERY position of the range unrolled and taken or not under a
// per-lane predicate (EXEC-masked, in place).  The lane that sits on position x loads that sweep's rotation; the others get
// it through a 1 KB LDS exchange (a DPP row broadcast spans 16 lanes only).  Twice the positions of the LDS form are
// visited (a sweep covers [l, m) of 0..n-2), but there is no Z in LDS, so occupancy is set by the registers.
template <int I, int NMAX>
__device__ __forceinline__ void qlw_replay_pos(cd (&z)[NMAX], const double2* __restrict__ xg, const bool on, const int ilo, const int ihi) {
    const double2 r = xg[I];
    if (on && I >= ilo && I <= ihi) {
        const cd zi = z[I], zj = z[I + 1];
        z[I + 1] = cd{r.y * zi.x + r.x * zj.x, r.y * zi.y + r.x * zj.y};
        z[I] = cd{r.x * zi.x - r.y * zj.x, r.x * zi.y - r.y * zj.y};
    }
    if constexpr (I > 0) qlw_replay_pos<I - 1, NMAX>(z, xg, on, ilo, ihi);
}

// LIST: the matrices W.list[0 .. *W.count) instead of the whole chunk (what k_tw32_vectors left: tbk_solve_tw32.inl); the grid is
// sized for the chunk and the blocks past the count leave at once.
template <int MODE, int NMAX, bool LIST = false>
__global__ __launch_bounds__(256) void k_ql_replay_reg(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk,
                                                        const QlwWork W, cd* __restrict__ evec, const WfsView wv) {
    __shared__ double2 xch[4][2][32];
    const int lane = threadIdx.x & 63, x = lane & 31;
    double2* xg = xch[threadIdx.x >> 6][lane >> 5];
    const int64_t idc0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 5;
    int64_t nlive = nchunk;
    if constexpr (LIST) {
        nlive = *W.count;
        if (blockIdx.x == 0 && threadIdx.x == 0 && nlive > 0) atomicAdd(W.listed, (unsigned long long)nlive);
        if ((int64_t)blockIdx.x * 8 >= nlive) return;
    }
    const bool live = idc0 < nlive;
    int64_t idc = live ? idc0 : nlive - 1;
    if constexpr (LIST) idc = W.list[idc];
    const int64_t id = id0 + idc;
    const bool real_row = x < n;
    const int xr = real_row ? x : n - 1;
    cd z[NMAX];
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
        const int bb = b < n ? b : n - 1;
        if constexpr (MODE == 1) z[b] = wf_at(wv, bb, id)[xr];
        else z[b] = evec[((int64_t)bb * nk + id) * n + xr];
    }
    const double2* __restrict__ rot = W.rot + idc * W.cap;
    const unsigned* __restrict__ swp = W.swp + idc * W.scap;
    const int nsw = live ? W.nsw[idc] : 0;
    int nmax = nsw;
    nmax = max(nmax, __shfl_xor(nmax, 32));
    int seq = 0;
    // the sweep word and this lane's rotation of the NEXT sweep are loaded while the current one is applied
    unsigned w_nx = nsw > 0 ? swp[0] : 0u;
    double2 r_nx{1.0, 0.0};
    {
        const int ihi = (int)(w_nx & 0xffu), cnt = (int)(w_nx >> 8);
        if (nsw > 0 && x <= ihi && x > ihi - cnt) r_nx = rot[ihi - x];
    }
    for (int s = 0; s < nmax; ++s) {
        const bool on = s < nsw;
        const unsigned w = w_nx;
        const double2 mine = r_nx;
        const int ihi = (int)(w & 0xffu), cnt = on ? (int)(w >> 8) : 0, ilo = ihi - cnt + 1;
        seq += cnt;
        if (s + 1 < nsw) {
            w_nx = swp[s + 1];
            const int ihi2 = (int)(w_nx & 0xffu), cnt2 = (int)(w_nx >> 8);
            r_nx = (x <= ihi2 && x > ihi2 - cnt2) ? rot[seq + (ihi2 - x)] : double2{1.0, 0.0};
        }
        xg[x] = mine;
        qlw_wave_sync();
        qlw_replay_pos<NMAX - 2, NMAX>(z, xg, on && cnt > 0, ilo, ihi);
        qlw_wave_sync();
    }
    if (!live || !real_row) return;
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
        if (b < n) {
            const int r = W.rank[(int64_t)b * nchunk + idc];
            if constexpr (MODE == 1) wf_at(wv, r, id)[x] = z[b];
            else evec[((int64_t)r * nk + id) * n + x] = z[b];
        }
    }
}

// ---- the same for n = 33..64: one matrix per 128-thread block, wavefront 0 holds the real parts of the 64 rows and wavefront 1
// the imaginary parts (the rotations are real), z[NMAX] doubles per lane; each wavefront loads the sweep's rotations for
// itself.  The sweep word is loaded through an address the compiler cannot prove uniform: as a wave-uniform value the
// per-position tests become scalar branches, and at their merge points the register allocator copies the row (see above).
template <int I, int NMAX, int LOW = 0>
__device__ __forceinline__ void qlw_replay_pos_re(double (&z)[NMAX], const double2* __restrict__ xg, const int ilo, const int ihi) {
    const double2 r = xg[I];
    if (I >= ilo && I <= ihi) {
        const double zi = z[I], zj = z[I + 1];
        z[I + 1] = r.y * zi + r.x * zj;
        z[I] = r.x * zi - r.y * zj;
    }
    if constexpr (I > LOW) qlw_replay_pos_re<I - 1, NMAX, LOW>(z, xg, ilo, ihi);
}

template <int MODE, int NMAX>
__global__ __launch_bounds__(128) void k_ql_replay_reg64(const int n, const int64_t nk, const int64_t id0, const int64_t nchunk,
                                                          const QlwWork W, cd* __restrict__ evec, const WfsView wv) {
    __shared__ double2 xch[2][64];
    const int x = threadIdx.x & 63, part = threadIdx.x >> 6;
    double2* xg = xch[part];
    const int64_t idc = blockIdx.x, id = id0 + idc;
    const bool real_row = x < n;
    const int xr = real_row ? x : n - 1;
    double z[NMAX];
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
        const int bb = b < n ? b : n - 1;
        const double* src;
        if constexpr (MODE == 1) src = (const double*)(wf_at(wv, bb, id) + xr);
        else src = (const double*)(evec + ((int64_t)bb * nk + id) * n + xr);
        z[b] = src[part];
    }
    int lane0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(lane0));          // (0 in every lane; opaque)
    const double2* __restrict__ rot = W.rot + idc * W.cap;
    const unsigned* __restrict__ swp = W.swp + idc * W.scap + lane0;
    const int nsw = W.nsw[idc];
    int seq = 0;
    unsigned w_nx = nsw > 0 ? swp[0] : 0u;
    double2 r_nx{1.0, 0.0};
    {
        const int ihi = (int)(w_nx & 0xffu), cnt = (int)(w_nx >> 8);
        if (nsw > 0 && x <= ihi && x > ihi - cnt) r_nx = rot[ihi - x];
    }
    for (int s = 0; s < nsw; ++s) {
        const unsigned w = w_nx;
        const double2 mine = r_nx;
        const int ihi = (int)(w & 0xffu), cnt = (int)(w >> 8), ilo = ihi - cnt + 1;
        seq += __builtin_amdgcn_readfirstlane(cnt);
        if (s + 1 < nsw) {
            w_nx = swp[s + 1];
            const int ihi2 = (int)(w_nx & 0xffu), cnt2 = (int)(w_nx >> 8);
            r_nx = (x <= ihi2 && x > ihi2 - cnt2) ? rot[seq + (ihi2 - x)] : double2{1.0, 0.0};
        }
        xg[x] = mine;
        qlw_wave_sync();
        // (tried: l only grows during the QL iteration, so late sweeps could run a half-range variant -- chosen by a
        // wave-uniform branch, which again makes the allocator split the row: 256 VGPRs instead of 166)
        qlw_replay_pos_re<NMAX - 2, NMAX>(z, xg, ilo, ihi);
        qlw_wave_sync();
    }
    if (!real_row) return;
#pragma unroll
    for (int b = 0; b < NMAX; ++b) {
        if (b < n) {
            const int r = W.rank[(int64_t)b * nchunk + idc];
            double* dst;
            if constexpr (MODE == 1) dst = (double*)(wf_at(wv, r, id) + x);
            else dst = (double*)(evec + ((int64_t)r * nk + id) * n + x);
            dst[part] = z[b];
        }
    }
}

#include "tbk_solve_hh32.inl"   // k_hh32: stage 1 for n <= 32 with the matrix in registers
#include "tbk_solve_ql32.inl"   // k_ql32_lanes: stage 2 for n <= 32 with (d, e) in registers
#include "tbk_solve_tw32.inl"   // k_tw32_vectors: stage 3 for n <= 32 without the rotation replay
static size_t hh32_lds_bytes(int n, int nR, bool tri) {
    size_t b = (size_t)(tri ? n * (n + 1) / 2 : n * (n | 1)) * sizeof(cd);   // H(k) | reflector record | Z; or the packed triangle of H(k)
    b += (size_t)(64 + std::max(n, nR) + 2 * n) * sizeof(cd);     // ubuf, qbuf, eo / phases, dphase, tsub
    b += (size_t)2 * n * sizeof(double);                          // tau, eb
    return (b + 15) & ~(size_t)15;
}

template <int MODE, bool VEC>
static int launch_qlw(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L, const GridArgs& G) {
    const TbkKnobs& K = tbk_knobs();
    const int64_t cap = VEC ? (K.qlw_cap > 0 ? (int64_t)K.qlw_cap : (int64_t)3 * n * n + 64) : 0;   // rotations recorded per matrix (~1.2 n^2 are typical)
    const int scap = VEC ? 8 * n : 0;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t per = (size_t)n * sizeof(double2) + (size_t)cap * sizeof(double2) + (size_t)scap * sizeof(unsigned) + sizeof(int) + (size_t)n * sizeof(int) +
                       (VEC && n <= 32 ? (size_t)n * sizeof(double) + 16 + (size_t)hh32_rec_size(n <= 20 ? 20 : n <= 24 ? 24 : n <= 28 ? 28 : 32) * sizeof(cd) : 0);
    const size_t budget = (size_t)(K.qlw_ws_mb > 0 ? K.qlw_ws_mb : 4096) << 20;
    // n <= 32 with eigenvectors: the vectors from the twisted factorisation (k_tw32_vectors), the replay for the listed matrices only
    // (not for models whose levels come in pairs at a generic k -- ModelView::pairs_hint -- unless TBK_TW32=3: every matrix would be listed)
    const bool tw32 = VEC && n <= 32 && K.tw32 != 0 && K.qlw_replay_reg != 0 && (MODE == 2 || !mv.pairs_hint || K.tw32 == 3);
    // n <= 32: the tridiagonalisation with the matrix in the registers of ONE wavefront (k_hh32, tbk_solve_hh32.inl; round 6)
    // (33^3 points, ms per call, k_hh32 / k_tridiag_lds: n = 17 1.41 / 1.37, 18 1.42 / 1.46, 20 1.57 / 1.72, 24 1.87 / 2.24, 28 3.00 / 3.18, 32 3.68 / 4.05
    // -- profiles/hh32_sweep.py; TBK_HH32=2 forces it at 17 too, 0 never).  With k_tw32_vectors behind it k_hh32 leaves the reflector
    // record instead of Z (TBK_TW32=2: Z as before, the back-transformation as a matrix product) and is used at 17 as well.
    // (at 17 states with AND without eigenvectors: the eigenvalues of the two forms of a call stay the same bits)
    const bool hh32 = n <= 32 && (K.hh32 == 2 || (K.hh32 != 0 && (n >= 18 || ((K.tw32 == 1 || K.tw32 == 3) && K.qlw_replay_reg != 0))));
    const bool refl = tw32 && hh32 && K.tw32 != 2;
    // models with paired levels at every k (pairs_hint): the replay for every matrix, but Q from the reflector record (k_hh32<.., 2>, then
    // k_tw32_vectors with W.all_q) instead of accumulated in k_hh32 -- 0.67 of that kernel's 1.75 ms per 36 k matrices of 32 states
    // against ~0.25 ms
    const bool hintq = VEC && n <= 32 && MODE != 2 && mv.pairs_hint && K.tw32 == 1 && K.qlw_replay_reg != 0 && hh32;
    // (17..20 and 25..28 states: forms of the three kernels with 20 / 28 rows / positions / lanes' worth of unrolled work -- the record format follows)
    const int nm32 = n <= 20 ? 20 : n <= 24 ? 24 : n <= 28 ? 28 : 32;
    // n <= 32 on k_ql32_lanes: the QL kernel is ONE dependent chain per wavefront on half a wavefront per SIMD -- a quarter of the call during
    // which the chip idles.  Chunks on the context's side streams (the scheme of launch_tw16): a chunk's QL runs beside its
    // neighbours' tridiagonalisation and vectors.  (Round 6 tried this with the replay kernels and gained nothing: every stage was
    // throughput-bound then.)  Measured (profiles/qlw_streams_probe.py, 1 / 2 / 3 chunks in flight, ms): 33^3 points with vectors n = 17
    // 1.03 / 0.98 / 0.95, 24: 1.22 / 1.10 / 1.09, 32: 1.81 / 1.64 / 1.66; 49^3: 2.52 / 2.43 / 2.46, 2.97 / 2.86 / 2.99, 5.01 / 4.60 / 4.66;
    // eigenvalues only LOSE (32 states: 1.48 / 1.75 / 1.98 -- two stages only, and every chunk's QL lasts as long as the whole batch's).
    // So: two chunks with eigenvectors, one without.  TBK_QLW_STREAMS=1: one chunk after the other on the context's stream.
    const bool ql32 = n <= 32 && K.ql32 != 0 && (tw32 || !VEC);
    int ns = ql32 && tw32 && nk >= 16384 ? 2 : 1;
    if (K.qlw_streams >= 1) ns = ql32 ? std::min(K.qlw_streams, 3) : 1;
    int64_t chunk = std::max<int64_t>(1024, (int64_t)(budget / ns / per));
    chunk = std::min<int64_t>(chunk, (nk + ns - 1) / ns);
    chunk = (nk + (nk + chunk - 1) / chunk - 1) / ((nk + chunk - 1) / chunk);   // equal chunks (the QL kernel's time hardly depends on the count)
    const size_t wone = al((size_t)chunk * n * sizeof(double2)) + 256 + al((size_t)chunk * cap * sizeof(double2)) +
                          al((size_t)chunk * scap * sizeof(unsigned)) + al((size_t)chunk * sizeof(int)) + al((size_t)chunk * n * sizeof(int)) + 1024 +
                          (tw32 ? al((size_t)chunk * n * sizeof(double)) + al((size_t)chunk * sizeof(uint2)) + al((size_t)chunk * sizeof(int)) + 256 : 0) +
                          (refl || hintq ? al((size_t)chunk * hh32_rec_size(nm32) * sizeof(cd)) : 0);
    const size_t wbytes = wone * ns;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tridiagonal-path workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    QlwWork Wsl[3];
    for (int sl = 0; sl < ns; ++sl) {
    QlwWork W{};
    unsigned char* p = (unsigned char*)ctx->work + (size_t)sl * wone;
    W.de = (double2*)p;
    p += al((size_t)chunk * n * sizeof(double2));
    p += 256;                                  // (the back-transform's eight-entry loads may start before a sweep's first entry)
    W.rot = (double2*)p;
    p += al((size_t)chunk * cap * sizeof(double2));
    W.swp = (unsigned*)p;
    p += al((size_t)chunk * scap * sizeof(unsigned));
    W.nsw = (int*)p;
    p += al((size_t)chunk * sizeof(int));
    W.rank = (int*)p;
    p += al((size_t)chunk * n * sizeof(int));
    W.cap = cap;
    W.scap = scap;
    if (tw32) {
        W.lam = (double*)p;
        p += al((size_t)chunk * n * sizeof(double));
        W.meta = (uint2*)p;
        p += al((size_t)chunk * sizeof(uint2));
        W.list = (int*)p;
        p += al((size_t)chunk * sizeof(int));
        W.count = (int*)p;
        p += 256;
        W.gaptol = K.tw16_gaptol;
        W.listed = (unsigned long long*)(ctx->flags_dev + TBK_FLAG_LISTED);
        if (refl) W.refl = (cd*)p;
    }
    if (hintq) {
        W.refl = (cd*)p;
        W.all_q = 1;
    }
    Wsl[sl] = W;
    }
    hipStream_t st[3] = {ctx->stream, ctx->stream, ctx->stream};
    if (ns > 1) {
        if (!ctx->side_ev[0])
            for (int i = 0; i < 4; ++i) TBK_HIP(hipEventCreateWithFlags(&ctx->side_ev[i], hipEventDisableTiming));
        TBK_HIP(hipEventRecord(ctx->side_ev[0], ctx->stream));
        for (int sl = 0; sl < ns; ++sl) {
            if (!ctx->side[sl]) TBK_HIP(hipStreamCreateWithFlags(&ctx->side[sl], hipStreamNonBlocking));
            st[sl] = ctx->side[sl];
            TBK_HIP(hipStreamWaitEvent(st[sl], ctx->side_ev[0], 0));
        }
    }

    const int rw = n <= 32 ? 32 : 64;
    int nt = rw == 32 ? 128 : 256;
    if (K.qlw_nt > 0) nt = K.qlw_nt >= 512 ? 512 : (K.qlw_nt >= 256 ? 256 : (K.qlw_nt >= 128 ? 128 : 64));
    if (rw == 64 && nt < 128) nt = 128;
    if (rw == 32 && nt > 256) nt = 256;
    const size_t lds1 = qlw_lds1_bytes(n, MODE == 2 ? 0 : mv.nR, rw, nt);
    TBK_REQUIRE(lds1 <= 160 * 1024, TBK_EUNSUPPORTED, "nsta=%d with %d lattice vectors needs %zu bytes of LDS", n, mv.nR, lds1);
    const size_t lds2 = (size_t)2 * n * 64 * sizeof(double);
    const size_t lds3 = (size_t)n * (n | 1) * sizeof(cd) + TBK_QLW_RING * sizeof(double2) + (size_t)scap * sizeof(unsigned);
    const void* f1 = nullptr;
#define TBK_QLW_K1(RW_, NT_)                                                                                                    \
    if (rw == RW_ && nt == NT_) {                                                                                               \
        f1 = (const void*)k_tridiag_lds<MODE, VEC, RW_, NT_>;                                                                   \
    }
    TBK_QLW_K1(32, 64) TBK_QLW_K1(32, 128) TBK_QLW_K1(32, 256) TBK_QLW_K1(64, 128) TBK_QLW_K1(64, 256) TBK_QLW_K1(64, 512)
#undef TBK_QLW_K1
    TBK_REQUIRE(f1, TBK_EINVAL, "launch_qlw: no kernel for %d rows x %d threads", rw, nt);
    if (lds1 > 64 * 1024) TBK_HIP(hipFuncSetAttribute(f1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (lds2 > 64 * 1024)
        TBK_HIP(hipFuncSetAttribute((const void*)k_tridiag_ql_lanes<MODE, VEC ? 1 : 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (VEC && lds3 > 64 * 1024) {
        TBK_HIP(hipFuncSetAttribute((const void*)k_ql_backtransform<MODE, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        TBK_HIP(hipFuncSetAttribute((const void*)k_ql_backtransform<MODE, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    cd* evec = MODE == 1 ? nullptr : L.evec;
    const size_t lds_hh = hh32_lds_bytes(n, MODE == 2 ? 0 : mv.nR, !VEC || refl || hintq);
    // the chunks; an error inside leaves through the join below like success does (see launch_tw16)
    auto run_chunks = [&]() -> int {
    int which = 0;
    for (int64_t id0 = 0; id0 < nk; id0 += chunk, which = (which + 1) % ns) {
        const int64_t nc = std::min<int64_t>(chunk, nk - id0);
        const QlwWork& W = Wsl[which];
        hipStream_t sq = st[which];
        if (hh32 && (refl || hintq)) {
            if (n <= 20) hipLaunchKernelGGL((k_hh32<MODE, 2, 20>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, W.refl);
            else if (n <= 24) hipLaunchKernelGGL((k_hh32<MODE, 2, 24>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, W.refl);
            else if (n <= 28) hipLaunchKernelGGL((k_hh32<MODE, 2, 28>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, W.refl);
            else hipLaunchKernelGGL((k_hh32<MODE, 2, 32>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, W.refl);
        } else if (hh32 && !VEC && n <= 20) {
            if constexpr (!VEC) hipLaunchKernelGGL((k_hh32<MODE, 0, 20>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, (cd*)nullptr);
        } else if (hh32 && !VEC && n > 24 && n <= 28) {
            if constexpr (!VEC) hipLaunchKernelGGL((k_hh32<MODE, 0, 28>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, (cd*)nullptr);
        } else if (hh32) {
            if (n <= 24) hipLaunchKernelGGL((k_hh32<MODE, VEC ? 1 : 0, 24>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, (cd*)nullptr);
            else hipLaunchKernelGGL((k_hh32<MODE, VEC ? 1 : 0, 32>), dim3((unsigned)nc), dim3(64), lds_hh, sq, mv, nk, L, G, id0, nc, W.de, (cd*)nullptr);
        } else {
#define TBK_QLW_K1(RW_, NT_)                                                                                                    \
    if (rw == RW_ && nt == NT_)                                                                                                 \
        hipLaunchKernelGGL((k_tridiag_lds<MODE, VEC, RW_, NT_>), dim3((unsigned)nc), dim3(NT_), lds1, sq, mv, nk, L, G, id0, nc, W.de);
        TBK_QLW_K1(32, 64) TBK_QLW_K1(32, 128) TBK_QLW_K1(32, 256) TBK_QLW_K1(64, 128) TBK_QLW_K1(64, 256) TBK_QLW_K1(64, 512)
#undef TBK_QLW_K1
        }
        // eigenvalues only, batches that leave the lane-per-matrix QL kernel a few wavefronts of pure latency: one thread per
        // EIGENVALUE instead (bisection, tbk_solve_trig.inl); TBK_QLW_BISECT=0 | 1 forces either
        bool bisect = false;
        // (with k_ql32_lanes the crossover sits lower for 17..32 states -- profiles/bisect_crossover_probe.py, ms bisection | QL: n = 17 x 2048
        // 0.20 | 0.16, n = 24 x 2048 0.29 | 0.26, x 4096 0.48 | 0.30, n = 32 x 2048 0.39 | 0.44, x 4096 0.57 | 0.48)
        const int bis_per_cu = ql32 ? (n <= 20 ? 6 : n <= 24 ? 8 : 12) : 16;
        if constexpr (!VEC && MODE != 1) bisect = K.qlw_bisect >= 0 ? K.qlw_bisect == 1 : nc < (int64_t)ctx->cus * bis_per_cu;
        if (bisect)
            hipLaunchKernelGGL(k_tridiag_bisect, dim3((unsigned)nc), dim3(64), (size_t)n * sizeof(double2) + 32 * sizeof(double), sq, n,
                               nk, id0, (const double2*)W.de, L.eval, (int64_t)1, nc, ctx->flags_dev);
        else {
            if (tw32) TBK_HIP(hipMemsetAsync(W.count, 0, sizeof(int), sq));
            // (tw32: no rotation record here -- 0.60 against 0.42 ms per 36 k matrices of 32 states -- the listed matrices get theirs below)
            const dim3 gq((unsigned)((nc + 63) / 64));
            if (ql32 && n <= 20) hipLaunchKernelGGL((k_ql32_lanes<MODE, 20>), gq, dim3(64), 0, sq, n, nk, id0, nc, W, L.eval, G, ctx->flags_dev);
            else if (ql32 && n <= 24) hipLaunchKernelGGL((k_ql32_lanes<MODE, 24>), gq, dim3(64), 0, sq, n, nk, id0, nc, W, L.eval, G, ctx->flags_dev);
            else if (ql32 && n <= 28) hipLaunchKernelGGL((k_ql32_lanes<MODE, 28>), gq, dim3(64), 0, sq, n, nk, id0, nc, W, L.eval, G, ctx->flags_dev);
            else if (ql32) hipLaunchKernelGGL((k_ql32_lanes<MODE, 32>), gq, dim3(64), 0, sq, n, nk, id0, nc, W, L.eval, G, ctx->flags_dev);
            else if (tw32)
                hipLaunchKernelGGL((k_tridiag_ql_lanes<MODE, 2>), dim3((unsigned)((nc + 63) / 64)), dim3(64), lds2, sq, n, nk, id0, nc, W,
                                   L.eval, G, ctx->flags_dev);
            else
                hipLaunchKernelGGL((k_tridiag_ql_lanes<MODE, VEC ? 1 : 0>), dim3((unsigned)((nc + 63) / 64)), dim3(64), lds2, sq, n, nk,
                                   id0, nc, W, L.eval, G, ctx->flags_dev);
        }
        if (VEC) {
            if (tw32) {
                const unsigned b2 = (unsigned)((nc + 1) / 2), b32 = (unsigned)((nc * 32 + 255) / 256);
                auto listed_ql = [&]() {
                    hipLaunchKernelGGL((k_tridiag_ql_lanes<MODE, 1, true>), dim3((unsigned)((nc + 63) / 64)), dim3(64), lds2, sq, n, nk, id0, nc,
                                       W, L.eval, G, ctx->flags_dev);
                };
                if (n <= 20 && refl) {
                    hipLaunchKernelGGL((k_tw32_vectors<MODE, 20, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    listed_ql();
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 24, true>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                } else if (n > 24 && n <= 28 && refl) {
                    hipLaunchKernelGGL((k_tw32_vectors<MODE, 28, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    listed_ql();
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 32, true>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                } else if (n <= 24) {
                    if (refl) hipLaunchKernelGGL((k_tw32_vectors<MODE, 24, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    else hipLaunchKernelGGL((k_tw32_vectors<MODE, 24, false>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    listed_ql();
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 24, true>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                } else {
                    if (refl) hipLaunchKernelGGL((k_tw32_vectors<MODE, 32, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    else hipLaunchKernelGGL((k_tw32_vectors<MODE, 32, false>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    listed_ql();
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 32, true>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                }
            } else if (rw == 32 && K.qlw_replay_reg != 0) {
                const unsigned b32 = (unsigned)((nc * 32 + 255) / 256);
                if (hintq) {                             // Q = H_0 .. H_{n-3} D from the record into the output array, for the replay
                    const unsigned b2 = (unsigned)((nc + 1) / 2);
                    if (n <= 20) hipLaunchKernelGGL((k_tw32_vectors<MODE, 20, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    else if (n <= 24) hipLaunchKernelGGL((k_tw32_vectors<MODE, 24, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    else if (n <= 28) hipLaunchKernelGGL((k_tw32_vectors<MODE, 28, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                    else hipLaunchKernelGGL((k_tw32_vectors<MODE, 32, true>), dim3(b2), dim3(64), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                }
                if (n <= 24)
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 24>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                else
                    hipLaunchKernelGGL((k_ql_replay_reg<MODE, 32>), dim3(b32), dim3(256), 0, sq, n, nk, id0, nc, W, evec, G.wv);
            } else if (rw == 64 && K.qlw_replay_reg != 0 && n >= 40) {   // (33..39: the 40-row form wastes more than it gains)
                if (n <= 40)
                    hipLaunchKernelGGL((k_ql_replay_reg64<MODE, 40>), dim3((unsigned)nc), dim3(128), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                else if (n <= 48)
                    hipLaunchKernelGGL((k_ql_replay_reg64<MODE, 48>), dim3((unsigned)nc), dim3(128), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                else if (n <= 56)
                    hipLaunchKernelGGL((k_ql_replay_reg64<MODE, 56>), dim3((unsigned)nc), dim3(128), 0, sq, n, nk, id0, nc, W, evec, G.wv);
                else
                    hipLaunchKernelGGL((k_ql_replay_reg64<MODE, 64>), dim3((unsigned)nc), dim3(128), 0, sq, n, nk, id0, nc, W, evec, G.wv);
            } else if (rw == 32)
                hipLaunchKernelGGL((k_ql_backtransform<MODE, 64>), dim3((unsigned)nc), dim3(64), lds3, sq, n, nk, id0, nc, W, evec, G.wv);
            else
                hipLaunchKernelGGL((k_ql_backtransform<MODE, 128>), dim3((unsigned)nc), dim3(128), lds3, sq, n, nk, id0, nc, W, evec,
                                   G.wv);
        }
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
    };
    const int rc_chunks = run_chunks();
    int rc_join = TBK_OK;
    if (ns > 1) {
        for (int sl = 0; sl < ns; ++sl) {
            hipError_t e = hipEventRecord(ctx->side_ev[1 + sl], st[sl]);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->side_ev[1 + sl], 0);
            if (e != hipSuccess && rc_join == TBK_OK) {
                (void)hipStreamSynchronize(st[sl]);                                  // (the ordering could not be expressed: wait here)
                tbk_set_error("joining the side streams of the n = 17..32 solver: %s", hipGetErrorString(e));
                rc_join = TBK_EHIP;
            }
        }
    }
    if (rc_chunks) return rc_chunks;
    return rc_join;
}
