// tbk_dos.hip -- k generators on the device and eigenvalue reductions.
//
// tb_model.k_uniform_mesh (pythtb.py:1792-1861) and the interpolation step of k_path
// (:1986-1996) as kernels, so that  solve_all(k_uniform_mesh(mesh))  needs no 16 B/k upload, and
// the histogram the reference's DOS example builds on the host (examples/haldane.py:96-121,
// matplotlib hist == np.histogram) as a reduction over the device-resident eigenvalues, so
// that it needs no 8 n B/k download either.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "tbk_internal.h"

// k[idx] = (i_0/N_0, ..., i_{d-1}/N_{d-1}),  idx row-major, last index fastest (pythtb.py:1828-1859)
// (k[0] is point `first` of the mesh: a rank of a sharded solve_all generates only its chunk of the list)
__global__ __launch_bounds__(256) void k_gen_mesh(const int d, const int n0, const int n1, const int n2, const int64_t first,
                                                  const int64_t nk, double* __restrict__ k) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= nk) return;
    int64_t rem = first + idx;
    int i2 = 0, i1 = 0;
    if (d >= 3) {
        i2 = (int)(rem % n2);
        rem /= n2;
    }
    if (d >= 2) {
        i1 = (int)(rem % n1);
        rem /= n1;
    }
    const int i0 = (int)rem;
    k[idx * d] = (double)i0 / (double)n0;
    if (d >= 2) k[idx * d + 1] = (double)i1 / (double)n1;
    if (d >= 3) k[idx * d + 2] = (double)i2 / (double)n2;
}

// linear interpolation between consecutive path nodes, both ends included (pythtb.py:1986-1996):
// point j of segment s (node_index[s] <= j <= node_index[s+1]) is
//   node[s] + (node[s+1]-node[s]) * frac,  frac = (j - n_i)/(n_f - n_i)
// Later segments overwrite the shared node point, like the reference's loop does.
__global__ __launch_bounds__(256) void k_gen_path(const int d, const int nseg, const double* __restrict__ nodes,
                                                  const int32_t* __restrict__ node_index, const int64_t nk,
                                                  double* __restrict__ k) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nk) return;
    int s = nseg - 1;   // last segment that contains j
    while (s > 0 && j < node_index[s]) --s;
    const int n_i = node_index[s], n_f = node_index[s + 1];
    // the reference builds frac with np.linspace-free arithmetic: float(j-n_i)/float(n_f-n_i)
    const double frac = (double)(j - n_i) / (double)(n_f - n_i);
    for (int c = 0; c < d; ++c) {
#pragma clang fp contract(off)   // separate multiply and add: bit-equal to the reference's k_i+frac*(k_f-k_i)
        const double a = nodes[s * d + c], b = nodes[(s + 1) * d + c];
        const double step = frac * (b - a);
        k[j * d + c] = a + step;
    }
}

// ---- per-band min / max of eval[n][nk] (ordered reduction: exact anyway)
__global__ __launch_bounds__(256) void k_minmax(const double* __restrict__ eval, const int64_t nk, double* __restrict__ part) {
    const int band = blockIdx.y;
    const double* e = eval + (int64_t)band * nk;
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += (int64_t)gridDim.x * 256) {
        const double v = e[i];
        lo = fmin(lo, v);
        hi = fmax(hi, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, o));
        hi = fmax(hi, __shfl_xor(hi, o));
    }
    __shared__ double red[8];
    if ((threadIdx.x & 63) == 0) {
        red[2 * (threadIdx.x >> 6)] = lo;
        red[2 * (threadIdx.x >> 6) + 1] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = part + ((int64_t)band * gridDim.x + blockIdx.x) * 2;
        p[0] = fmin(fmin(red[0], red[2]), fmin(red[4], red[6]));
        p[1] = fmax(fmax(red[1], red[3]), fmax(red[5], red[7]));
    }
}

// ---- np.histogram with equal-width bins: index guess from the affine map, then the same
// edge corrections numpy applies (numpy/lib/_histograms_impl.py: `decrement` / `increment`), so
// the bin of every value is decided by comparisons with the SAME edges array numpy would use.
// counts[band][bin]; LDS histogram per workgroup, then one integer atomic per non-empty bin.
__global__ __launch_bounds__(256) void k_hist(const double* __restrict__ eval, const int64_t nk, const int nbins,
                                              const double* __restrict__ edges, unsigned long long* __restrict__ counts) {
    extern __shared__ unsigned int h[];
    const int band = blockIdx.y;
    for (int b = threadIdx.x; b < nbins; b += 256) h[b] = 0u;
    __syncthreads();
    const double first = edges[0], last = edges[nbins];
    const double denom = last - first;
    const double* e = eval + (int64_t)band * nk;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nk; i += (int64_t)gridDim.x * 256) {
        const double x = e[i];
        if (!(x >= first && x <= last)) continue;
        int b = (int)(((x - first) / denom) * (double)nbins);
        b = b < 0 ? 0 : (b >= nbins ? nbins - 1 : b);
        while (b > 0 && x < edges[b]) --b;
        while (b < nbins - 1 && x >= edges[b + 1]) ++b;
        atomicAdd(&h[b], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256)
        if (h[b]) atomicAdd(&counts[(int64_t)band * nbins + b], (unsigned long long)h[b]);
}

extern "C" int tbk_k_uniform_mesh_range_dev(tbk_ctx* ctx, int dim_k, const int32_t* mesh, int64_t first, int64_t count,
                                            double* k_dev) {
    TBK_REQUIRE(ctx && mesh && (k_dev || count == 0), TBK_EINVAL, "tbk_k_uniform_mesh_dev: null argument");
    TBK_REQUIRE(dim_k >= 1 && dim_k <= 3, TBK_EINVAL, "tbk_k_uniform_mesh_dev: dim_k=%d (the reference supports 1..3)", dim_k);
    int64_t nk = 1;
    for (int d = 0; d < dim_k; ++d) {
        TBK_REQUIRE(mesh[d] >= 1, TBK_EINVAL, "tbk_k_uniform_mesh_dev: mesh[%d]=%d", d, mesh[d]);
        nk *= mesh[d];
    }
    TBK_REQUIRE(first >= 0 && count >= 0 && first + count <= nk, TBK_EINVAL,
                "tbk_k_uniform_mesh_range_dev: points [%lld, %lld) of a mesh of %lld", (long long)first,
                (long long)(first + count), (long long)nk);
    if (count == 0) return TBK_OK;
    TBK_HIP(hipSetDevice(ctx->device));
    ProfScope ps(ctx, "k_uniform_mesh");
    hipLaunchKernelGGL(k_gen_mesh, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, dim_k, mesh[0],
                       dim_k > 1 ? mesh[1] : 1, dim_k > 2 ? mesh[2] : 1, first, count, k_dev);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

extern "C" int tbk_k_uniform_mesh_dev(tbk_ctx* ctx, int dim_k, const int32_t* mesh, double* k_dev) {
    TBK_REQUIRE(ctx && mesh && k_dev, TBK_EINVAL, "tbk_k_uniform_mesh_dev: null argument");
    int64_t nk = 1;
    for (int d = 0; d < dim_k && d < 3; ++d) nk *= mesh[d] > 0 ? mesh[d] : 0;
    return tbk_k_uniform_mesh_range_dev(ctx, dim_k, mesh, 0, nk, k_dev);
}

extern "C" int tbk_k_path_dev(tbk_ctx* ctx, int dim_k, int n_nodes, const double* nodes, const int32_t* node_index,
                              int64_t nk, double* k_dev) {
    TBK_REQUIRE(ctx && nodes && node_index && k_dev, TBK_EINVAL, "tbk_k_path_dev: null argument");
    TBK_REQUIRE(dim_k >= 1 && dim_k <= TBK_MAX_DIM && n_nodes >= 2 && nk >= n_nodes, TBK_EINVAL,
                "tbk_k_path_dev: dim_k=%d n_nodes=%d nk=%lld", dim_k, n_nodes, (long long)nk);
    TBK_REQUIRE(node_index[0] == 0 && node_index[n_nodes - 1] == nk - 1, TBK_EINVAL,
                "tbk_k_path_dev: node_index must run from 0 to nk-1");
    for (int s = 0; s + 1 < n_nodes; ++s)
        TBK_REQUIRE(node_index[s] < node_index[s + 1], TBK_EINVAL, "tbk_k_path_dev: node_index not increasing at %d", s);
    TBK_HIP(hipSetDevice(ctx->device));
    const size_t nb = (size_t)n_nodes * dim_k * sizeof(double), ib = (size_t)n_nodes * sizeof(int32_t);
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + ((nb + 255) & ~(size_t)255) + ib, &base);
    if (rc) return rc;
    double* nodes_dev = (double*)((unsigned char*)base + 256);
    int32_t* idx_dev = (int32_t*)((unsigned char*)nodes_dev + ((nb + 255) & ~(size_t)255));
    TBK_HIP(hipMemcpyAsync(nodes_dev, nodes, nb, hipMemcpyHostToDevice, ctx->stream));
    TBK_HIP(hipMemcpyAsync(idx_dev, node_index, ib, hipMemcpyHostToDevice, ctx->stream));
    {
        ProfScope ps(ctx, "k_path");
        hipLaunchKernelGGL(k_gen_path, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, ctx->stream, dim_k, n_nodes - 1,
                           nodes_dev, idx_dev, nk, k_dev);
        TBK_HIP(hipGetLastError());
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));   // the staging area is scratch: done before anyone reuses it
    return TBK_OK;
}

// shared front end: eigenvalues (and optionally vectors) of the model on k_uniform_mesh(mesh),
// everything generated and kept on the device.  Scratch layout: [flags | k | eval | evec].
static int mesh_solve_dev(tbk_model* m, const int32_t* mesh, bool vec, size_t extra, int64_t* nk_out, double** k_dev,
                          double** e_dev, double** v_dev, void** extra_dev) {
    tbk_ctx* ctx = m->ctx;
    const int d = m->dim_k, n = m->nsta;
    TBK_REQUIRE(d >= 1 && d <= 3, TBK_EINVAL, "k_uniform_mesh: dim_k=%d (the reference supports 1..3)", d);
    int64_t nk = 1;
    for (int c = 0; c < d; ++c) {
        TBK_REQUIRE(mesh[c] >= 1, TBK_EINVAL, "k_uniform_mesh: mesh[%d]=%d", c, mesh[c]);
        nk *= mesh[c];
    }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t kb = (size_t)nk * d * sizeof(double), eb = (size_t)nk * n * sizeof(double);
    const size_t vb = vec ? (size_t)nk * n * n * sizeof(cd) : 0;
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + al(kb) + al(eb) + al(vb) + al(extra), &base);
    if (rc) return rc;
    unsigned char* p = (unsigned char*)base + 256;
    *k_dev = (double*)p;
    *e_dev = (double*)(p + al(kb));
    *v_dev = vec ? (double*)(p + al(kb) + al(eb)) : nullptr;
    if (extra_dev) *extra_dev = p + al(kb) + al(eb) + al(vb);
    *nk_out = nk;
    if (!vec) {          // eigenvalues only, up to 4 states: straight from the mesh (k_mesh_evals), no k list at all
        bool done = false;
        rc = tbk_mesh_evals_rows(m, mesh, *e_dev, &done);
        if (rc) return rc;
        if (done) return tbk_eigh_check(ctx, n);
    }
    rc = tbk_k_uniform_mesh_dev(ctx, d, mesh, *k_dev);
    if (rc) return rc;
    return tbk_solve_list_dev_checked(m, *k_dev, nk, *e_dev, *v_dev);
}

extern "C" int tbk_solve_mesh(tbk_model* m, const int32_t* mesh, double* eval, double* evec) {
    TBK_REQUIRE(m && mesh && eval, TBK_EINVAL, "tbk_solve_mesh: null argument");
    tbk_ctx* ctx = m->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    int64_t nk = 0;
    double *k_dev = nullptr, *e_dev = nullptr, *v_dev = nullptr;
    int rc = mesh_solve_dev(m, mesh, evec != nullptr, 0, &nk, &k_dev, &e_dev, &v_dev, nullptr);
    if (rc) return rc;
    const int n = m->nsta;
    TBK_HIP(hipMemcpyAsync(eval, e_dev, (size_t)nk * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (evec) TBK_HIP(hipMemcpyAsync(evec, v_dev, (size_t)nk * n * n * sizeof(cd), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

extern "C" int tbk_dos_mesh(tbk_model* m, const int32_t* mesh, int nbins, const double* edges, int64_t* counts,
                            double* band_min, double* band_max) {
    TBK_REQUIRE(m && mesh, TBK_EINVAL, "tbk_dos_mesh: null argument");
    TBK_REQUIRE((nbins == 0 && !edges && !counts) || (nbins >= 1 && nbins <= 8192 && edges && counts), TBK_EINVAL,
                "tbk_dos_mesh: nbins=%d (1..8192 with edges and counts, or 0 for the band extrema alone)", nbins);
    tbk_ctx* ctx = m->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const int n = m->nsta;
    int64_t nk = 1;
    for (int c = 0; c < m->dim_k && c < 3; ++c) nk *= std::max(mesh[c], 1);
    double *k_dev = nullptr, *e_dev = nullptr, *v_dev = nullptr;
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nk + 256 * 8 - 1) / (256 * 8), 1024));
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t part_b = al((size_t)n * gx * 2 * sizeof(double));
    const size_t edge_b = al((size_t)(nbins + 1) * sizeof(double));
    const size_t cnt_b = al((size_t)n * std::max(nbins, 1) * sizeof(unsigned long long));
    void* red = nullptr;   // reduction buffers ride behind the eigenvalues in the same scratch block
    int rc = mesh_solve_dev(m, mesh, false, part_b + edge_b + cnt_b, &nk, &k_dev, &e_dev, &v_dev, &red);
    if (rc) return rc;
    double* part_dev = (double*)red;
    double* edges_dev = (double*)((unsigned char*)red + part_b);
    unsigned long long* cnt_dev = (unsigned long long*)((unsigned char*)red + part_b + edge_b);
    if (band_min || band_max) {
        ProfScope ps(ctx, "eval_minmax");
        hipLaunchKernelGGL(k_minmax, dim3(gx, n), dim3(256), 0, ctx->stream, e_dev, nk, part_dev);
        TBK_HIP(hipGetLastError());
    }
    if (nbins > 0) {
        TBK_HIP(hipMemcpyAsync(edges_dev, edges, (size_t)(nbins + 1) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        TBK_HIP(hipMemsetAsync(cnt_dev, 0, (size_t)n * nbins * sizeof(unsigned long long), ctx->stream));
        ProfScope ps(ctx, "eval_hist");
        hipLaunchKernelGGL(k_hist, dim3(gx, n), dim3(256), (size_t)nbins * sizeof(unsigned int), ctx->stream, e_dev, nk, nbins,
                           edges_dev, cnt_dev);
        TBK_HIP(hipGetLastError());
    }
    std::vector<double> part;
    if (band_min || band_max) {
        part.resize((size_t)n * gx * 2);
        TBK_HIP(hipMemcpyAsync(part.data(), part_dev, part.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (nbins > 0)
        TBK_HIP(hipMemcpyAsync(counts, cnt_dev, (size_t)n * nbins * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < n && !part.empty(); ++b) {
        double lo = INFINITY, hi = -INFINITY;
        for (unsigned g = 0; g < gx; ++g) {
            lo = std::min(lo, part[((size_t)b * gx + g) * 2]);
            hi = std::max(hi, part[((size_t)b * gx + g) * 2 + 1]);
        }
        if (band_min) band_min[b] = lo;
        if (band_max) band_max[b] = hi;
    }
    return TBK_OK;
}
