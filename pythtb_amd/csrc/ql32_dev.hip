#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <type_traits>
#include "tbk_internal.h"
#include "tbk_solve_dev.h"
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
    cd* refl;              // [nchunk][hh32_rec_size(NM)]  the reflector records of k_hh32<.., 2, NM> (nullptr: Q sits in the output array)
};
#include "tbk_solve_ql32.inl"
template __global__ void k_ql32_lanes<1, 28>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, double*, const GridArgs, int*);
template __global__ void k_ql32_lanes<1, 24>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, double*, const GridArgs, int*);
template __global__ void k_ql32_lanes<1, 32>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, double*, const GridArgs, int*);
template __global__ void k_ql32_lanes<0, 32>(const int, const int64_t, const int64_t, const int64_t, const QlwWork, double*, const GridArgs, int*);
