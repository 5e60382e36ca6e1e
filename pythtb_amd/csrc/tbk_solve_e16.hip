// tbk_solve_e16.hip -- translation unit of the fused n = 9..16 eigen-solver (k_e16, tbk_solve_e16.inl): H(k) assembly, Householder
// tridiagonalisation, eigenvalues, eigenvectors and back-transformation of a matrix in ONE wavefront pass with the reflectors kept
// in LDS (pythtb.py:927-953 `_sol_ham` with eig_vectors, :2499-2511 the mesh loop).  Its own file so that the 100-second compile of
// tbk_solve.hip is not in the loop of this kernel's development; the dispatch (launch_tw16, tbk_solve_tw16.inl) calls in here.
#include "tbk_solve_dev.h"
#include "tbk_solve_e16.inl"

// The matrices [id0, id0 + nc) of a k list (mode 0), a mesh window (1) or supplied matrices (2) on `stream`; the matrices left to
// the QL-replay kernels come back as list[0 .. *count) (positions relative to id0; *count must be zero on the stream before).  `form`: E16_F_* bits (tbk_solve_dev.h).
int tbk_e16_launch(int mode, hipStream_t stream, const ModelView& mv, int64_t nk, const ListArgs& L, const GridArgs& G, int64_t id0,
                   int64_t nc, int* list, int* count, double gaptol, int form) {
    TBK_REQUIRE(mode >= 0 && mode <= 2 && nc >= 1 && nc < (int64_t)0x7fffffff / 16 && mv.nsta >= 2 && mv.nsta <= 16, TBK_EINVAL,
                "tbk_e16_launch: mode %d, %lld matrices of %d states", mode, (long long)nc, mv.nsta);
    const unsigned blocks = (unsigned)((nc * 16 + 255) / 256);
    if (mode == 0) hipLaunchKernelGGL((k_e16<0>), dim3(blocks), dim3(256), 0, stream, mv, nk, L, G, id0, nc, list, count, gaptol, form);
    else if (mode == 1) hipLaunchKernelGGL((k_e16<1>), dim3(blocks), dim3(256), 0, stream, mv, nk, L, G, id0, nc, list, count, gaptol, form);
    else hipLaunchKernelGGL((k_e16<2>), dim3(blocks), dim3(256), 0, stream, mv, nk, L, G, id0, nc, list, count, gaptol, form);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

// Eigenvalues only (k list: mode 0, supplied matrices: mode 2) -> L.eval[b][id0 .. id0 + nc); nothing is listed, nothing else written.
int tbk_e16_launch_evals(int mode, hipStream_t stream, const ModelView& mv, int64_t nk, const ListArgs& L, int64_t id0, int64_t nc) {
    TBK_REQUIRE((mode == 0 || mode == 2) && nc >= 1 && nc < (int64_t)0x7fffffff / 16 && mv.nsta >= 2 && mv.nsta <= 16 && L.eval, TBK_EINVAL,
                "tbk_e16_launch_evals: mode %d, %lld matrices of %d states", mode, (long long)nc, mv.nsta);
    const unsigned blocks = (unsigned)((nc * 16 + 255) / 256);
    const GridArgs G{};
    if (mode == 0) hipLaunchKernelGGL((k_e16<0, false>), dim3(blocks), dim3(256), 0, stream, mv, nk, L, G, id0, nc, (int*)nullptr, (int*)nullptr, 0.0, 0);
    else hipLaunchKernelGGL((k_e16<2, false>), dim3(blocks), dim3(256), 0, stream, mv, nk, L, G, id0, nc, (int*)nullptr, (int*)nullptr, 0.0, 0);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}
