// tbk_position.hip -- position-operator matrix elements and hybrid Wannier functions
// (the first "next" row): tb_model.position_matrix / position_expectation /
// position_hwf (pythtb.py:2034-2279), batched over k-points.
//
//   X_mn(k) = sum_j conj(C_mj) r_j C_nj          r_j = reduced orbital coordinate along `dir`
//   hwfc    = ascending eigenvalues of X, hwf = its eigenvectors (rows), optionally expanded
//             back on the orbitals: hwf_orb[i][j] = sum_b hwf[i][b] C_bj   (:2262-2277)
#include <algorithm>
#include "tbk_internal.h"

// one thread per (k, m, n)
__global__ __launch_bounds__(256) void k_position_matrix(const cd* __restrict__ evec, const double* __restrict__ pos,
                                                         int64_t nk, int nsub, int ncomp, cd* __restrict__ xmat) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t per = (int64_t)nsub * nsub;
    if (idx >= nk * per) return;
    const int64_t ik = idx / per;
    const int e = (int)(idx - ik * per);
    const int m = e / nsub, n = e - m * nsub;
    const cd* a = evec + (ik * nsub + m) * ncomp;
    const cd* b = evec + (ik * nsub + n) * ncomp;
    cd acc{0.0, 0.0};
    for (int j = 0; j < ncomp; ++j) cfmac(acc, a[j], cscale(b[j], pos[j]));
    xmat[idx] = acc;
}

// band-major eigen-solver outputs -> [k][i] / [k][i][x] in the requested basis
__global__ __launch_bounds__(256) void k_hwf_out(const double* __restrict__ ev, const cd* __restrict__ vw,
                                                 const cd* __restrict__ evec, int64_t nk, int nsub, int ncomp,
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
        for (int b = 0; b < nsub; ++b) cfma(acc, row[b], evec[(ik * nsub + b) * ncomp + x]);
        hwf[idx] = acc;
    }
}

extern "C" int tbk_position_hwf(tbk_ctx* ctx, const double* evec, int64_t nk, int nsub, int ncomp,
                                const double* pos, double* xmat, double* hwfc, double* hwf, int orbital_basis) {
    TBK_REQUIRE(ctx && evec && pos && nk >= 0, TBK_EINVAL, "tbk_position_hwf: bad argument");
    TBK_REQUIRE(nsub >= 1 && ncomp >= 1, TBK_EINVAL, "tbk_position_hwf: nsub=%d ncomp=%d", nsub, ncomp);
    TBK_REQUIRE(nsub <= TBK_MAX_NSTA, TBK_EUNSUPPORTED, "tbk_position_hwf: %d states exceeds this build's limit of %d",
                nsub, TBK_MAX_NSTA);
    if (nk == 0) return TBK_OK;
    TBK_HIP(hipSetDevice(ctx->device));
    const bool eig = hwfc || hwf;
    const int width = orbital_basis ? ncomp : nsub;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t b_evec = al((size_t)nk * nsub * ncomp * sizeof(cd));
    const size_t b_pos = al((size_t)ncomp * sizeof(double));
    const size_t b_x = al((size_t)nk * nsub * nsub * sizeof(cd));
    const size_t b_ev = al((size_t)nk * nsub * sizeof(double));
    const size_t b_vw = hwf ? b_x : 0;
    const size_t b_c = hwfc ? b_ev : 0;
    const size_t b_h = hwf ? al((size_t)nk * nsub * width * sizeof(cd)) : 0;
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + b_evec + b_pos + b_x + b_ev + b_vw + b_c + b_h, &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;
    cd* d_evec = (cd*)p;                  p += b_evec;
    double* d_pos = (double*)p;           p += b_pos;
    cd* d_x = (cd*)p;                     p += b_x;
    double* d_ev = (double*)p;            p += b_ev;
    cd* d_vw = hwf ? (cd*)p : nullptr;    p += b_vw;
    double* d_c = hwfc ? (double*)p : nullptr;  p += b_c;
    cd* d_h = hwf ? (cd*)p : nullptr;
    TBK_HIP(hipMemcpyAsync(d_evec, evec, (size_t)nk * nsub * ncomp * sizeof(cd), hipMemcpyHostToDevice, ctx->stream));
    TBK_HIP(hipMemcpyAsync(d_pos, pos, (size_t)ncomp * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "position_matrix");
        const int64_t total = nk * nsub * nsub;
        hipLaunchKernelGGL(k_position_matrix, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const cd*)d_evec, (const double*)d_pos, nk, nsub, ncomp, d_x);
        TBK_HIP(hipGetLastError());
    }
    if (xmat) TBK_HIP(hipMemcpyAsync(xmat, d_x, (size_t)nk * nsub * nsub * sizeof(cd), hipMemcpyDeviceToHost, ctx->stream));
    if (eig) {
        rc = tbk_eigh_dev(ctx, nsub, d_x, nk, d_ev, d_vw, "position_eigh");
        if (rc) return rc;
        {
            ProfScope ps(ctx, "hwf_out");
            const int64_t total = nk * nsub * (hwf ? width : 1);
            hipLaunchKernelGGL(k_hwf_out, dim3((unsigned)((nk * nsub * (int64_t)width + 255) / 256)), dim3(256), 0,
                               ctx->stream, (const double*)d_ev, (const cd*)d_vw, (const cd*)d_evec, nk, nsub, ncomp,
                               orbital_basis, d_c, d_h);
            (void)total;
            TBK_HIP(hipGetLastError());
        }
        if (hwfc) TBK_HIP(hipMemcpyAsync(hwfc, d_c, (size_t)nk * nsub * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (hwf) TBK_HIP(hipMemcpyAsync(hwf, d_h, (size_t)nk * nsub * width * sizeof(cd), hipMemcpyDeviceToHost, ctx->stream));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return eig ? tbk_eigh_check(ctx, nsub) : TBK_OK;
}
