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

// band-major eigen-solver outputs -> [k][i] / [k][i][x] in the requested basis
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
        hipLaunchKernelGGL(k_position_matrix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           src, (const double*)d_pos, nk, nsub, ncomp, d_x);
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
