// tbk_solve_big.inl -- included by tbk_solve.hip.
//
// n > 256 states per k (ribbons, slabs, finite flakes made by cut_piece / make_supercell):
// the whole chip works on each matrix.  Same parallel-ordered two-sided Jacobi as
// k_solve_wave, but one ROUND is one kernel launch over every 2x2 block of every matrix of
// the batch:
//
//     A_out[{p_i,q_i}][{p_j,q_j}] = J_i^+  A_in[{p_i,q_i}][{p_j,q_j}]  J_j
//
// for all pairs (i,j) of the round's n/2 disjoint rotations.  The 2x2 blocks are disjoint,
// A ping-pongs between two buffers, so a round needs no grid-wide barrier other than the
// kernel boundary; the rotation parameters are recomputed from A_in by every workgroup that
// needs them (identical arithmetic -> identical values).  V^T rows are rotated in place by
// extra workgroups of the same launch.  Lanes run along j: in the round-robin order the
// column indices p_j (ascending) and q_j (descending) are consecutive, so all four loads and
// stores of a wavefront are coalesced.
//
// Per sweep: a deterministic two-stage off-norm reduction marks converged matrices (their
// workgroups exit at once in later rounds, and the buffer parity that holds their result is
// recorded), and the host reads back one int: how many matrices are still rotating.

struct BigWs {
    cd* A0;           // [B][n][n]
    cd* A1;           // [B][n][n]
    cd* Vt;           // [B][n][n]   Vt[c][r] = V[r][c]
    cd* eo;           // [B][n]      conj(exp(2 pi i k.tau_o)) (* pbc phase)
    double* ev;       // [B][n]
    int* perm;        // [B][n]
    double* partial;  // [B][nbn][2]
    int* done;        // [B]
    int* fpar;        // [B]  buffer parity that holds the converged A
    int* pending;     // [1]
    int n;
    int nbn;          // norm blocks per matrix
    int ld;           // row pitch of A0/A1/Vt and per-matrix stride of ev/perm: n here, n rounded up to the block
                      // grid in the blocked solver (tbk_solve_blk.inl), whose padding rows/columns are decoupled
    double* padval;   // [B] blocked solver: diagonal value of the first padding index (null otherwise)
};

template <int MODE>
__global__ __launch_bounds__(256) void k_big_init(const ModelView mv, const int64_t base, const ListArgs L,
                                                  const GridArgs G, const BigWs W, const int vec) {
    const int mat = blockIdx.y;
    const int64_t id = base + mat;
    const int n = W.n, ld = W.ld;
    const int64_t nn = (int64_t)n * n, ll = (int64_t)ld * ld;
    cd* A = W.A0 + (size_t)mat * ll;
    cd* Vt = W.Vt + (size_t)mat * ll;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nth = (int64_t)gridDim.x * 256;
    if (tid == 0) {
        W.done[mat] = 0;
        W.fpar[mat] = 0;
    }
    for (int64_t e = tid; e < ll; e += nth) {
        const int a = (int)(e / ld), b = (int)(e - (int64_t)a * ld);
        Vt[e] = cd{a == b ? 1.0 : 0.0, 0.0};
        if (a >= n || b >= n) A[e] = cd{0.0, 0.0};     // padding (blocked solver): decoupled, diagonal set later
    }
    if constexpr (MODE == 2) {
        const cd* h = L.ham + id * nn;
        for (int64_t e = tid; e < nn; e += nth) {
            const int a = (int)(e / n), b = (int)(e - (int64_t)a * n);
            cd v = a <= b ? h[e] : cconj(h[(int64_t)b * n + a]);   // one triangle, like the reference's eigh
            if (a == b) v.y = 0.0;
            A[(int64_t)a * ld + b] = v;
        }
        if (vec)
            for (int64_t o = tid; o < n; o += nth) W.eo[(size_t)mat * n + o] = cd{1.0, 0.0};
    } else {
        double kk[4] = {0.0, 0.0, 0.0, 0.0};
        bool wrap[4] = {false, false, false, false};
        if constexpr (MODE == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
                if (d < mv.dim_k) kk[d] = L.k[id * mv.dim_k + d];
        } else {
            grid_point(G, id, kk, wrap);
        }
        cd z[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) z[d] = d < mv.dim_k ? expi2pi(kk[d]) : cd{1.0, 0.0};
        for (int64_t slot = tid; slot < mv.nslot; slot += nth) {
            const int ab = mv.slot_ab[slot];
            const int a = ab & 0xffff, b = ab >> 16;
            const int t0 = mv.slot_ptr[slot], t1 = mv.slot_ptr[slot + 1];
            cd acc{0.0, 0.0};
            for (int t = t0; t < t1; ++t) cfma(acc, mv.term_amp[t], phase_of_R(z, mv.term_R[t]));
            if (a == b) {
                A[(int64_t)a * ld + a] = cd{acc.x, 0.0};
            } else {
                A[(int64_t)a * ld + b] = acc;
                A[(int64_t)b * ld + a] = cconj(acc);
            }
        }
        if (vec)
            for (int64_t o = tid; o < n; o += nth) {
                cd f = cconj(expi2pi(kdot(kk, mv.orb[o])));
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (wrap[d]) f = cmul(f, G.pbc[d * n + o]);
                }
                W.eo[(size_t)mat * n + o] = f;
            }
    }
}

// stage 1: per-workgroup partial sums of |a|^2 on and off the diagonal (fixed order)
__global__ __launch_bounds__(256) void k_big_norm1(const BigWs W, const int par) {
    const int mat = blockIdx.y;
    if (W.done[mat]) return;
    const int n = W.ld;        // padding: exact zeros off the diagonal; its diagonal is taken out again in stage 2
    const int64_t nn = (int64_t)n * n;
    const cd* A = (par ? W.A1 : W.A0) + (size_t)mat * nn;
    double off = 0.0, dia = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (int64_t)gridDim.x * 256) {
        const int a = (int)(e / n), b = (int)(e - (int64_t)a * n);
        const double v2 = cabs2(A[e]);
        if (a == b) dia += v2; else off += v2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        off += __shfl_xor(off, o);
        dia += __shfl_xor(dia, o);
    }
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) {
        red[2 * (threadIdx.x >> 6)] = off;
        red[2 * (threadIdx.x >> 6) + 1] = dia;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = W.partial + ((size_t)mat * W.nbn + blockIdx.x) * 2;
        p[0] = (red[0] + red[2]) + (red[4] + red[6]);
        p[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

// stage 2: one thread per matrix decides convergence (same criterion as k_solve_wave)
__global__ __launch_bounds__(64) void k_big_norm2(const BigWs W, const int nmat, const int par, const int force,
                                                  int* noconv_flag, const int pads_set = 0) {
    const int mat = blockIdx.x * 64 + threadIdx.x;
    if (mat >= nmat || W.done[mat]) return;
    double off = 0.0, dia = 0.0;
    const double* p = W.partial + (size_t)mat * W.nbn * 2;
    for (int b = 0; b < W.nbn; ++b) {
        off += p[2 * b];
        dia += p[2 * b + 1];
    }
    if (W.padval) {
        if (pads_set) {      // the padding diagonal is not part of the matrix
            for (int x = 0; x < W.ld - W.n; ++x) {
                const double v = W.padval[mat] + x;
                dia -= v * v;
            }
            dia = fmax(dia, 0.0);
        } else {
            W.padval[mat] = sqrt(dia + off) + 1.0;      // above every eigenvalue (Frobenius norm + 1)
        }
    }
    if (off <= 2.0e-32 * (dia + off)) {
        W.done[mat] = 1;
        W.fpar[mat] = par;
    } else if (force) {
        W.done[mat] = 1;
        W.fpar[mat] = par;
        atomicExch(noconv_flag, 1);
    } else {
        atomicAdd(W.pending, 1);
    }
}

// pair l of round `round` among m players (m even; player m-1 is fixed, and is the bye when
// n is odd).  p ascends and q descends with l, which is what makes the loads coalesce.
__device__ __forceinline__ void big_pair(const int l, const int round, const int m, int& p, int& q) {
    if (l == 0) {
        p = m - 1;
        q = round;
    } else {
        p = round + l;
        if (p >= m - 1) p -= m - 1;
        q = round - l;
        if (q < 0) q += m - 1;
    }
}

__global__ __launch_bounds__(256) void k_big_rotate(const BigWs W, const int round, const int par, const int tilesA) {
    const int mat = blockIdx.z;
    if (W.done[mat]) return;
    const int n = W.n;
    const int m = (n + 1) & ~1, half = m >> 1;
    const int64_t nn = (int64_t)n * n;
    const cd* Ain = (par ? W.A1 : W.A0) + (size_t)mat * nn;
    cd* Aout = (par ? W.A0 : W.A1) + (size_t)mat * nn;
    __shared__ double s_c[68];
    __shared__ cd s_sw[68];
    __shared__ int s_p[68], s_q[68];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bool isA = (int)blockIdx.y < tilesA;
    const int ibase = (isA ? (int)blockIdx.y : (int)blockIdx.y - tilesA) * 4;
    if (threadIdx.x < 68) {
        const bool colslot = threadIdx.x < 64;
        const int l = colslot ? (int)blockIdx.x * 64 + (int)threadIdx.x : ibase + (int)threadIdx.x - 64;
        int p = -1, q = -1;
        double c = 1.0;
        cd sw{0.0, 0.0};
        if ((isA || !colslot) && l < half) {
            big_pair(l, round, m, p, q);
            if (p < n) {   // not the bye: same division-free parameters as rotate<>
                const cd g = Ain[(int64_t)p * n + q];
                const double g2 = cabs2(g);
                if (g2 > 0.0) {
                    const double a = 0.5 * (Ain[(int64_t)q * n + q].x - Ain[(int64_t)p * n + p].x), aa = fabs(a);
                    const double r = sqrt(a * a + g2);
                    const double inv = rsqrt(2.0 * r * (r + aa));
                    const double sg = copysign(1.0, a);
                    c = (aa + r) * inv;
                    sw = cd{sg * g.x * inv, sg * g.y * inv};
                }
            }
        }
        s_p[threadIdx.x] = p;
        s_q[threadIdx.x] = q;
        s_c[threadIdx.x] = c;
        s_sw[threadIdx.x] = sw;
    }
    __syncthreads();
    if (isA) {
        const int pi = s_p[64 + ty], qi = s_q[64 + ty], pj = s_p[tx], qj = s_q[tx];
        if (pi < 0 || pj < 0) return;
        const bool vi = pi < n, vj = pj < n;   // false: the bye "player" (does not exist)
        const double ci = s_c[64 + ty], cj = s_c[tx];
        const cd si = s_sw[64 + ty], sj = s_sw[tx];
        const cd zero{0.0, 0.0};
        cd x00 = vi && vj ? Ain[(int64_t)pi * n + pj] : zero;
        cd x01 = vi ? Ain[(int64_t)pi * n + qj] : zero;
        cd x10 = vj ? Ain[(int64_t)qi * n + pj] : zero;
        cd x11 = Ain[(int64_t)qi * n + qj];
        // columns: A <- A J_j     a'_rp = c a_rp - conj(s) a_rq ;  a'_rq = s a_rp + c a_rq
        {
            const cd a = x00, b = x01;
            x00 = cd{cj * a.x - (sj.x * b.x + sj.y * b.y), cj * a.y - (sj.x * b.y - sj.y * b.x)};
            x01 = cd{(sj.x * a.x - sj.y * a.y) + cj * b.x, (sj.x * a.y + sj.y * a.x) + cj * b.y};
        }
        {
            const cd a = x10, b = x11;
            x10 = cd{cj * a.x - (sj.x * b.x + sj.y * b.y), cj * a.y - (sj.x * b.y - sj.y * b.x)};
            x11 = cd{(sj.x * a.x - sj.y * a.y) + cj * b.x, (sj.x * a.y + sj.y * a.x) + cj * b.y};
        }
        // rows: A <- J_i^+ A      a'_pc = c a_pc - s a_qc ;  a'_qc = conj(s) a_pc + c a_qc
        {
            const cd a = x00, b = x10;
            x00 = cd{ci * a.x - (si.x * b.x - si.y * b.y), ci * a.y - (si.x * b.y + si.y * b.x)};
            x10 = cd{(si.x * a.x + si.y * a.y) + ci * b.x, (si.x * a.y - si.y * a.x) + ci * b.y};
        }
        {
            const cd a = x01, b = x11;
            x01 = cd{ci * a.x - (si.x * b.x - si.y * b.y), ci * a.y - (si.x * b.y + si.y * b.x)};
            x11 = cd{(si.x * a.x + si.y * a.y) + ci * b.x, (si.x * a.y - si.y * a.x) + ci * b.y};
        }
        if (pi == pj) {   // the rotated pair itself: exactly diagonal, real
            x01 = zero;
            x10 = zero;
            x00.y = 0.0;
            x11.y = 0.0;
        }
        if (vi && vj) Aout[(int64_t)pi * n + pj] = x00;
        if (vi) Aout[(int64_t)pi * n + qj] = x01;
        if (vj) Aout[(int64_t)qi * n + pj] = x10;
        Aout[(int64_t)qi * n + qj] = x11;
    } else {
        const int p = s_p[64 + ty], q = s_q[64 + ty];
        if (p < 0 || p >= n) return;
        const double c = s_c[64 + ty];
        const cd sw = s_sw[64 + ty];
        if (c == 1.0 && sw.x == 0.0 && sw.y == 0.0) return;
        cd* Vt = W.Vt + (size_t)mat * nn;
        // V <- V J :  v'_rp = c v_rp - conj(s) v_rq ;  v'_rq = s v_rp + c v_rq      (rows p, q of V^T)
        for (int col = (int)blockIdx.x * 64 + tx; col < n; col += (int)gridDim.x * 64) {
            const cd x = Vt[(int64_t)p * n + col], y = Vt[(int64_t)q * n + col];
            Vt[(int64_t)p * n + col] = cd{c * x.x - (sw.x * y.x + sw.y * y.y), c * x.y - (sw.x * y.y - sw.y * y.x)};
            Vt[(int64_t)q * n + col] = cd{(sw.x * x.x - sw.y * x.y) + c * y.x, (sw.x * x.y + sw.y * x.x) + c * y.y};
        }
    }
}

// eigenvalues in stable ascending order; list modes write them out, the mesh mode folds the
// gaps between neighbouring bands into the sharded minimum
template <int MODE>
__global__ __launch_bounds__(256) void k_big_sort(const BigWs W, const int64_t base, const int64_t nk,
                                                  const ListArgs L, const GridArgs G) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    double* ev = (double*)lds_raw;
    const int mat = blockIdx.x;
    const int64_t id = base + mat;
    const int n = W.n, ld = W.ld;     // the padding entries (blocked solver) exceed every eigenvalue: they sort last
    const int64_t nn = (int64_t)ld * ld;
    const cd* A = (W.fpar[mat] ? W.A1 : W.A0) + (size_t)mat * nn;
    for (int x = threadIdx.x; x < ld; x += 256) ev[x] = A[(int64_t)x * ld + x].x;
    __syncthreads();
    int* perm = W.perm + (size_t)mat * ld;
    double* sorted = W.ev + (size_t)mat * ld;
    for (int x = threadIdx.x; x < ld; x += 256) {
        const double mine = ev[x];
        int r = 0;
        for (int j = 0; j < ld; ++j) r += tbk_before(ev[j], mine, j, x) ? 1 : 0;
        perm[r] = x;
        sorted[r] = mine;
    }
    __syncthreads();
    if constexpr (MODE == 1) {
        unsigned long long* shard = G.gaps + (size_t)(blockIdx.x & (TBK_GAP_SHARDS - 1)) * n;
        for (int b = threadIdx.x; b + 1 < n; b += 256) {
            const double g = sorted[b + 1] - sorted[b];
            const unsigned long long bits = (unsigned long long)__double_as_longlong(fmax(g, 0.0));
            if (bits < __hip_atomic_load(shard + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(shard + b, bits);
        }
    } else {
        for (int b = threadIdx.x; b < n; b += 256) L.eval[(int64_t)b * nk + id] = sorted[b];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_big_write(const BigWs W, const int64_t base, const int64_t nk,
                                                   const ListArgs L, const GridArgs G) {
    const int mat = blockIdx.y;
    const int64_t id = base + mat;
    const int n = W.n, ld = W.ld;
    const int64_t nn = (int64_t)n * n;
    const cd* Vt = W.Vt + (size_t)mat * ld * ld;
    const cd* eo = W.eo + (size_t)mat * n;
    const int* perm = W.perm + (size_t)mat * ld;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nn; e += (int64_t)gridDim.x * 256) {
        const int rb = (int)(e / n), o = (int)(e - (int64_t)rb * n);
        const cd v = cmul(Vt[(int64_t)perm[rb] * ld + o], eo[o]);
        if constexpr (MODE == 1) wf_at(G.wv, rb, id)[o] = v;
        else L.evec[((int64_t)rb * nk + id) * n + o] = v;
    }
}

template <int MODE, bool VEC>
static int launch_big(tbk_ctx* ctx, const ModelView& mv, int n, int64_t nk, const ListArgs& L, const GridArgs& G) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nn = (size_t)n * n;
    const int nbn = (int)std::min<size_t>(64, (nn + 256 * 16 - 1) / (256 * 16));
    const size_t per = 3 * al(nn * sizeof(cd)) + al((size_t)n * sizeof(cd)) + al((size_t)n * sizeof(double)) +
                       al((size_t)n * sizeof(int)) + al((size_t)nbn * 2 * sizeof(double));
    size_t free_b = 0, total_b = 0;
    TBK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t budget = std::max<size_t>(per, std::min<size_t>((size_t)8 << 30, (free_b + ctx->work_bytes) / 2));
    int64_t B = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(nk, 32768), (int64_t)(budget / per)));
    if (tbk_knobs().big_batch >= 0) B = std::max<int64_t>(1, std::min<int64_t>(B, tbk_knobs().big_batch));   // test hook: matrices per batch
    const size_t wbytes = (size_t)B * per + al((size_t)B * sizeof(int)) * 2 + 256;
    if (wbytes > ctx->work_bytes) {
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->work) TBK_HIP(hipFree(ctx->work));
        ctx->work = nullptr;
        ctx->work_bytes = 0;
        hipError_t e = hipMalloc(&ctx->work, wbytes);
        TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "eigen-solver workspace of %zu bytes: %s", wbytes, hipGetErrorString(e));
        ctx->work_bytes = wbytes;
    }
    BigWs W{};
    unsigned char* p = (unsigned char*)ctx->work;
    W.pending = (int*)p;
    p += 256;
    W.A0 = (cd*)p;
    p += (size_t)B * al(nn * sizeof(cd));
    W.A1 = (cd*)p;
    p += (size_t)B * al(nn * sizeof(cd));
    W.Vt = (cd*)p;
    p += (size_t)B * al(nn * sizeof(cd));
    W.eo = (cd*)p;
    p += al((size_t)B * n * sizeof(cd));
    W.ev = (double*)p;
    p += al((size_t)B * n * sizeof(double));
    W.perm = (int*)p;
    p += al((size_t)B * n * sizeof(int));
    W.partial = (double*)p;
    p += al((size_t)B * nbn * 2 * sizeof(double));
    W.done = (int*)p;
    p += al((size_t)B * sizeof(int));
    W.fpar = (int*)p;
    W.n = n;
    W.nbn = nbn;
    W.ld = n;
    W.padval = nullptr;
    // al() of a per-matrix block only pads the END of each region; matrices inside a region are
    // contiguous (stride n*n), which is what the kernels index with.
    const int m = (n + 1) & ~1, half = m >> 1;
    const int tiles_x = (half + 63) / 64, tiles_a = (half + 3) / 4;
    const unsigned init_x = (unsigned)std::min<size_t>(1024, (nn + 255) / 256);
    for (int64_t base = 0; base < nk; base += B) {
        const int nb = (int)std::min<int64_t>(B, nk - base);
        hipLaunchKernelGGL((k_big_init<MODE>), dim3(init_x, nb), dim3(256), 0, ctx->stream, mv, base, L, G, W, VEC ? 1 : 0);
        int par = 0;
        for (int sweep = 0; sweep <= TBK_JACOBI_MAX_SWEEPS; ++sweep) {
            TBK_HIP(hipMemsetAsync(W.pending, 0, sizeof(int), ctx->stream));
            hipLaunchKernelGGL(k_big_norm1, dim3(nbn, nb), dim3(256), 0, ctx->stream, W, par);
            hipLaunchKernelGGL(k_big_norm2, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, W, nb, par,
                               sweep == TBK_JACOBI_MAX_SWEEPS ? 1 : 0, ctx->flags_dev);
            int pending = 0;
            TBK_HIP(hipMemcpyAsync(&pending, W.pending, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            TBK_HIP(hipStreamSynchronize(ctx->stream));
            if (pending == 0) break;
            for (int round = 0; round < m - 1; ++round) {
                hipLaunchKernelGGL(k_big_rotate, dim3(tiles_x, (VEC ? 2 : 1) * tiles_a, nb), dim3(256), 0, ctx->stream, W, round, par, tiles_a);
                par ^= 1;
            }
            TBK_HIP(hipGetLastError());
        }
        hipLaunchKernelGGL((k_big_sort<MODE>), dim3(nb), dim3(256), (size_t)n * sizeof(double), ctx->stream, W, base, nk, L, G);
        if (VEC)
            hipLaunchKernelGGL((k_big_write<MODE>), dim3(init_x, nb), dim3(256), 0, ctx->stream, W, base, nk, L, G);
        TBK_HIP(hipGetLastError());
    }
    return TBK_OK;
}
