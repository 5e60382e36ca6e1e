// tbk_position.hip -- position-operator matrix elements and hybrid Wannier functions
// (the first "next" row): tb_model.position_matrix / position_expectation /
// position_hwf (pythtb.py:2034-2279), batched over k-points.
//
//   X_mn(k) = sum_j conj(C_mj) r_j C_nj          r_j = reduced orbital coordinate along `dir`
//   hwfc    = ascending eigenvalues of X, hwf = its eigenvectors (rows), optionally expanded
//             back on the orbitals: hwf_orb[i][j] = sum_b hwf[i][b] C_bj   (:2262-2277)
#include <algorithm>
#include "tbk_internal.h"

// Where the input states live: element (point ik, state b, component j) sits at
//   base[(pts ? pts[ik] : ik) * sk + (occ ? occ[b] : b) * sb + j]
// -- a packed host-supplied batch [k][b][j] (sk = nsub*ncomp, sb = ncomp) or the band-major planes of a
// resident wf_array (sk = ncomp, sb = npts*ncomp) with an occupied-band list and, optionally, a point list.
struct EvecSrc {
    const cd* base;
    int64_t sk, sb;
    const int32_t* occ;
    const int64_t* pts;
    __device__ __forceinline__ const cd* at(int64_t ik, int b) const {
        return base + (pts ? pts[ik] : ik) * sk + (int64_t)(occ ? occ[b] : b) * sb;
    }
};

// one thread per (k, m, n)
__global__ __launch_bounds__(256) void k_position_matrix(const EvecSrc ev, const double* __restrict__ pos,
                                                         int64_t nk, int nsub, int ncomp, cd* __restrict__ xmat) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)nsub * nsub;
    if (idx >= nk * per) return;
    const int64_t ik = idx / per;
    const int e = (int)(idx - ik * per);
    const int m = e / nsub, n = e - m * nsub;
    const cd* a = ev.at(ik, m);
    const cd* b = ev.at(ik, n);
    cd acc{0.0, 0.0};
    for (int j = 0; j < ncomp; ++j) cfmac(acc, a[j], cscale(b[j], pos[j]));
    xmat[idx] = acc;
}

// The same for up to 8 states on batches (position_hwf_mesh: a 513^2 array of slabs, 8 of 16 bands): the kernel above has every
// thread walk two 256-byte rows of its own -- 64 threads per point re-read the point's 2 KB sixteen times through the caches, and
// the launch took 0.59 ms of the call's 0.89 (profiles/r04f).  Here a wavefront stages FOUR points (their nsub rows, coalesced
// 256-byte runs) in LDS and sixteen lanes per point form X in 2 x 2 blocks (two rows of the bra, two of the ket per lane and
// component: four LDS reads per sixteen multiply-adds).  The sum over the components runs in the same order, one chain per entry.
__device__ __forceinline__ void pos_lds_sync_wave() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
}
template <int NT>   // NT = ceil(nsub / 2) = 1..8: up to 8 states sixteen lanes and four points per wavefront, 9..16 states all 64 lanes on one point
__global__ __launch_bounds__(256) void k_position_matrix_tile(const EvecSrc ev, const double* __restrict__ pos, const int64_t nk,
                                                              const int nsub, const int ncomp, cd* __restrict__ xmat) {
    extern __shared__ __align__(16) unsigned char pos_lds[];
    constexpr int LP = NT <= 4 ? 16 : 64, PW = 64 / LP;   // lanes per point, points per wavefront
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p0 = ((int64_t)blockIdx.x * 4 + wib) * PW;
    if (p0 >= nk) return;                              // (no workgroup barrier below: the wavefronts are independent)
    const int ldp = ncomp + 1, pbuf = nsub * ldp;
    const int wstride = PW * pbuf + (ncomp + 1) / 2;  // the points + the coordinates
    cd* const buf = reinterpret_cast<cd*>(pos_lds) + (size_t)wib * wstride;
    {   // rows (point, state) of ncomp contiguous components: sixteen lanes per row.  All the loads of a 16-component slice are
        // issued before the first goes to LDS (a load -> ds_write pair per loop trip serialises eight memory latencies: 0.24 ms
        // for the 513^2 array where this form takes 0.1)
        const int j0 = lane & 15, nrow = PW * nsub;     // (at most 32 rows: 4 x 8 or 1 x 16)
        for (int jc = 0; jc < ncomp; jc += 16) {
            const int j = jc + j0;
            cd r[8];
            int dsto[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = (lane >> 4) + 4 * i;
                const int pt = row / nsub, b = row - pt * nsub;
                const bool ok = row < nrow && j < ncomp && p0 + pt < nk;
                dsto[i] = ok ? pt * pbuf + b * ldp + j : -1;
                r[i] = ok ? ev.at(p0 + pt, b)[j] : cd{0.0, 0.0};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (dsto[i] >= 0) buf[dsto[i]] = r[i];
        }
    }
    {
        double* const pl = reinterpret_cast<double*>(buf + PW * pbuf);
        for (int j = lane; j < ncomp; j += 64) pl[j] = pos[j];
    }
    pos_lds_sync_wave();
    const int g = lane / LP, tl = lane - g * LP;
    const int ta = tl / NT, tb = tl - ta * NT;
    if (tl >= NT * NT || p0 + g >= nk) return;
    const int m0 = min(2 * ta, nsub - 1), m1 = min(2 * ta + 1, nsub - 1), n0 = min(2 * tb, nsub - 1), n1 = min(2 * tb + 1, nsub - 1);
    const cd* const pa = buf + g * pbuf;
    const cd *a0 = pa + m0 * ldp, *a1 = pa + m1 * ldp, *b0 = pa + n0 * ldp, *b1 = pa + n1 * ldp;
    cd x00{0.0, 0.0}, x01{0.0, 0.0}, x10{0.0, 0.0}, x11{0.0, 0.0};
    const double* const posl = reinterpret_cast<const double*>(buf + PW * pbuf);  // (staged behind the points)
    for (int j = 0; j < ncomp; ++j) {
        const double r = posl[j];
        const cd u0 = a0[j], u1 = a1[j], v0 = cscale(b0[j], r), v1 = cscale(b1[j], r);
        cfmac(x00, u0, v0);
        cfmac(x01, u0, v1);
        cfmac(x10, u1, v0);
        cfmac(x11, u1, v1);
    }
    cd* const o = xmat + (p0 + g) * (int64_t)nsub * nsub;
    o[m0 * nsub + n0] = x00;
    if (2 * tb + 1 < nsub) o[m0 * nsub + n1] = x01;
    if (2 * ta + 1 < nsub) {
        o[m1 * nsub + n0] = x10;
        if (2 * tb + 1 < nsub) o[m1 * nsub + n1] = x11;
    }
}

// band-major eigen-solver outputs -> [k][i] / [k][i][x] in the requested basis
// The same for 17..32 states (round 6; profiles/position_cliff_sweep.py: the thread-per-entry kernel took 5.6 x the time of 16 states
// for 20, 28 x for 32 -- 2.3 ms per 33 k points): one point per wavefront, a 4 x 4 block of X per lane (NT4 = ceil(nsub / 4) = 5..8:
// NT4^2 <= 64 lanes), eight LDS reads for sixteen complex multiply-adds per component; two wavefronts per workgroup (17 KB of LDS each).
template <int NT4>
__global__ __launch_bounds__(128) void k_position_matrix_tile4(const EvecSrc ev, const double* __restrict__ pos, const int64_t nk,
                                                               const int nsub, const int ncomp, cd* __restrict__ xmat) {
    extern __shared__ __align__(16) unsigned char pos_lds[];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t p0 = (int64_t)blockIdx.x * 2 + wib;
    if (p0 >= nk) return;                              // (no workgroup barrier below: the wavefronts are independent)
    const int ldp = ncomp + 1, pbuf = nsub * ldp;
    const int wstride = pbuf + (ncomp + 1) / 2;        // the point + the coordinates
    cd* const buf = reinterpret_cast<cd*>(pos_lds) + (size_t)wib * wstride;
    {   // rows (state) of ncomp contiguous components: sixteen lanes per row, all loads of a 16-component slice before the first write
        const int j0 = lane & 15;
        for (int jc = 0; jc < ncomp; jc += 16) {
            const int j = jc + j0;
            cd r[8];
            int dsto[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int b = (lane >> 4) + 4 * i;
                const bool ok = b < nsub && j < ncomp;
                dsto[i] = ok ? b * ldp + j : -1;
                r[i] = ok ? ev.at(p0, b)[j] : cd{0.0, 0.0};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (dsto[i] >= 0) buf[dsto[i]] = r[i];
        }
    }
    {
        double* const pl = reinterpret_cast<double*>(buf + pbuf);
        for (int j = lane; j < ncomp; j += 64) pl[j] = pos[j];
    }
    pos_lds_sync_wave();
    const int ta = lane / NT4, tb = lane - ta * NT4;
    if (lane >= NT4 * NT4) return;
    const cd* ar[4];
    const cd* br[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ar[i] = buf + min(4 * ta + i, nsub - 1) * ldp;
        br[i] = buf + min(4 * tb + i, nsub - 1) * ldp;
    }
    cd x[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) x[i][k] = cd{0.0, 0.0};
    const double* const posl = reinterpret_cast<const double*>(buf + pbuf);
    for (int j = 0; j < ncomp; ++j) {
        const double r = posl[j];
        cd u[4], v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u[i] = ar[i][j];
            v[i] = cscale(br[i][j], r);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) cfmac(x[i][k], u[i], v[k]);
    }
    cd* const o = xmat + p0 * (int64_t)nsub * nsub;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * ta + i < nsub && 4 * tb + k < nsub) o[(4 * ta + i) * nsub + 4 * tb + k] = x[i][k];
}

__global__ __launch_bounds__(256) void k_hwf_out(const double* __restrict__ ev, const cd* __restrict__ vw,
                                                 const EvecSrc src, int64_t nk, int nsub, int ncomp,
                                                 int orbital, double* __restrict__ hwfc, cd* __restrict__ hwf) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int width = orbital ? ncomp : nsub;
    const int64_t per = (int64_t)nsub * width;
    if (idx >= nk * per) return;
    const int64_t ik = idx / per;
    const int e = (int)(idx - ik * per);
    const int i = e / width, x = e - i * width;
    if (hwfc && x == 0) hwfc[ik * nsub + i] = ev[(int64_t)i * nk + ik];
    if (!hwf) return;
    const cd* row = vw + ((int64_t)i * nk + ik) * nsub;        // HWF i on the input states
    if (!orbital) {
        hwf[idx] = row[x];
    } else {
        cd acc{0.0, 0.0};
        for (int b = 0; b < nsub; ++b) cfma(acc, row[b], src.at(ik, b)[x]);
        hwf[idx] = acc;
    }
}

// Shared driver.  host_evec != null: the packed batch is uploaded first; else `src` already points into
// device memory (a resident wf_array) and occ/pts (host lists, nullable) are uploaded into the scratch.
static int position_run(tbk_ctx* ctx, const double* host_evec, EvecSrc src, const int32_t* occ, const int64_t* pts,
                        int64_t nk, int nsub, int ncomp, const double* pos, double* xmat, double* hwfc, double* hwf,
                        int orbital_basis) {
    if (nk == 0) return TBK_OK;
    TBK_HIP(hipSetDevice(ctx->device));
    const bool eig = hwfc || hwf;
    const int width = orbital_basis ? ncomp : nsub;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_evec = host_evec ? al((size_t)nk * nsub * ncomp * sizeof(cd)) : 0;
    const size_t b_occ = occ ? al((size_t)nsub * sizeof(int32_t)) : 0;
    const size_t b_pts = pts ? al((size_t)nk * sizeof(int64_t)) : 0;
    const size_t b_pos = al((size_t)ncomp * sizeof(double));
    const size_t b_x = al((size_t)nk * nsub * nsub * sizeof(cd));
    const size_t b_ev = al((size_t)nk * nsub * sizeof(double));
    const size_t b_vw = hwf ? b_x : 0;
    const size_t b_c = hwfc ? b_ev : 0;
    const size_t b_h = hwf ? al((size_t)nk * nsub * width * sizeof(cd)) : 0;
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + b_evec + b_occ + b_pts + b_pos + b_x + b_ev + b_vw + b_c + b_h, &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;
    cd* d_evec = (cd*)p;                  p += b_evec;
    int32_t* d_occ = (int32_t*)p;         p += b_occ;
    int64_t* d_pts = (int64_t*)p;         p += b_pts;
    double* d_pos = (double*)p;           p += b_pos;
    cd* d_x = (cd*)p;                     p += b_x;
    double* d_ev = (double*)p;            p += b_ev;
    cd* d_vw = hwf ? (cd*)p : nullptr;    p += b_vw;
    double* d_c = hwfc ? (double*)p : nullptr;  p += b_c;
    cd* d_h = hwf ? (cd*)p : nullptr;
    if (host_evec) {
        TBK_HIP(hipMemcpyAsync(d_evec, host_evec, (size_t)nk * nsub * ncomp * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
        src = EvecSrc{d_evec, (int64_t)nsub * ncomp, (int64_t)ncomp, nullptr, nullptr};
    }
    if (occ) {
        TBK_HIP(hipMemcpyAsync(d_occ, occ, (size_t)nsub * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        src.occ = d_occ;
    }
    if (pts) {
        TBK_HIP(hipMemcpyAsync(d_pts, pts, (size_t)nk * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        src.pts = d_pts;
    }
    TBK_HIP(hipMemcpyAsync(d_pos, pos, (size_t)ncomp * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "position_matrix");
        const int64_t total = nk * nsub * nsub;
        const int pw = nsub <= 8 ? 4 : 1;               // points per wavefront (k_position_matrix_tile)
        const size_t lds_t = (size_t)4 * (pw * nsub * (ncomp + 1) + (ncomp + 1) / 2) * sizeof(cd);   // four wavefronts x (their points + the coordinates)
        // (TBK_POS_TILE=0: the thread-per-entry kernel at any size)
        if (nsub <= 16 && nk >= 64 && lds_t <= 64 * 1024 && tbk_knobs().pos_tile != 0) {
            const dim3 g((unsigned)((nk + 4 * pw - 1) / (4 * pw))), b(256);
#define TBK_PT(NN) hipLaunchKernelGGL((k_position_matrix_tile<NN>), g, b, lds_t, ctx->stream, src, (const double*)d_pos, nk, nsub, ncomp, d_x)
            switch ((nsub + 1) / 2) {
                case 1: TBK_PT(1); break;
                case 2: TBK_PT(2); break;
                case 3: TBK_PT(3); break;
                case 4: TBK_PT(4); break;
                case 5: TBK_PT(5); break;
                case 6: TBK_PT(6); break;
                case 7: TBK_PT(7); break;
                default: TBK_PT(8); break;
            }
#undef TBK_PT
        } else if (nsub > 16 && nsub <= 32 && nk >= 64 && (size_t)2 * (nsub * (ncomp + 1) + (ncomp + 1) / 2) * sizeof(cd) <= 64 * 1024 &&
                   tbk_knobs().pos_tile != 0) {
            const size_t lds4 = (size_t)2 * (nsub * (ncomp + 1) + (ncomp + 1) / 2) * sizeof(cd);
            const dim3 g((unsigned)((nk + 1) / 2)), b(128);
#define TBK_PT4(NN) hipLaunchKernelGGL((k_position_matrix_tile4<NN>), g, b, lds4, ctx->stream, src, (const double*)d_pos, nk, nsub, ncomp, d_x)
            switch ((nsub + 3) / 4) {
                case 5: TBK_PT4(5); break;
                case 6: TBK_PT4(6); break;
                case 7: TBK_PT4(7); break;
                default: TBK_PT4(8); break;
            }
#undef TBK_PT4
        } else {
            hipLaunchKernelGGL(k_position_matrix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                               src, (const double*)d_pos, nk, nsub, ncomp, d_x);
        }
        TBK_HIP(hipGetLastError());
    }
    if (xmat) TBK_HIP(hipMemcpyAsync(xmat, d_x, (size_t)nk * nsub * nsub * sizeof(cd), hipMemcpyDeviceToHost, ctx->stream));
    if (eig) {
        rc = tbk_eigh_dev_checked(ctx, nsub, d_x, nk, d_ev, d_vw, "position_eigh");
        if (rc) return rc;
        {
            ProfScope ps(ctx, "hwf_out");
            hipLaunchKernelGGL(k_hwf_out, dim3((unsigned)((nk * nsub * (int64_t)width + 255) / 256)), dim3(256), 0,
                               ctx->stream, (const double*)d_ev, (const cd*)d_vw, src, nk, nsub, ncomp,
                               orbital_basis, d_c, d_h);
            TBK_HIP(hipGetLastError());
        }
        if (hwfc) TBK_HIP(hipMemcpyAsync(hwfc, d_c, (size_t)nk * nsub * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (hwf) TBK_HIP(hipMemcpyAsync(hwf, d_h, (size_t)nk * nsub * width * sizeof(cd), hipMemcpyDeviceToHost, ctx->stream));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

extern "C" int tbk_position_hwf(tbk_ctx* ctx, const double* evec, int64_t nk, int nsub, int ncomp,
                                const double* pos, double* xmat, double* hwfc, double* hwf, int orbital_basis) {
    TBK_REQUIRE(ctx && evec && pos && nk >= 0, TBK_EINVAL, "tbk_position_hwf: bad argument");
    TBK_REQUIRE(nsub >= 1 && ncomp >= 1, TBK_EINVAL, "tbk_position_hwf: nsub=%d ncomp=%d", nsub, ncomp);
    TBK_REQUIRE(nsub <= TBK_MAX_NSTA, TBK_EUNSUPPORTED, "tbk_position_hwf: %d states exceeds this build's limit of %d",
                nsub, TBK_MAX_NSTA);
    return position_run(ctx, evec, EvecSrc{}, nullptr, nullptr, nk, nsub, ncomp, pos, xmat, hwfc, hwf, orbital_basis);
}

extern "C" int tbk_wfs_position_hwf(tbk_wfs* w, const int64_t* point_index, int64_t npoints, const int32_t* occ, int nocc,
                                    const double* pos, double* xmat, double* hwfc, double* hwf, int orbital_basis) {
    TBK_REQUIRE(w && occ && pos && npoints >= 0, TBK_EINVAL, "tbk_wfs_position_hwf: bad argument");
    const WfsView& v = w->view;
    TBK_REQUIRE(nocc >= 1 && nocc <= TBK_MAX_NSTA, TBK_EUNSUPPORTED, "tbk_wfs_position_hwf: nocc=%d (limit %d)", nocc, TBK_MAX_NSTA);
    for (int b = 0; b < nocc; ++b)
        TBK_REQUIRE(occ[b] >= 0 && occ[b] < v.nsta, TBK_EINVAL, "tbk_wfs_position_hwf: state %d outside the %d stored", occ[b], v.nsta);
    int64_t nk = v.npts;
    if (point_index) {
        nk = npoints;
        for (int64_t i = 0; i < nk; ++i)
            TBK_REQUIRE(point_index[i] >= 0 && point_index[i] < v.npts, TBK_EINVAL,
                        "tbk_wfs_position_hwf: point %lld outside the mesh of %lld points", (long long)point_index[i], (long long)v.npts);
    }
    const EvecSrc src{v.data, (int64_t)v.ncomp, v.npts * (int64_t)v.ncomp, nullptr, nullptr};
    return position_run(w->ctx, nullptr, src, occ, point_index, nk, nocc, v.ncomp, pos, xmat, hwfc, hwf, orbital_basis);
}
