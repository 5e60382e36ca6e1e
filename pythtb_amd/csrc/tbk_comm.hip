// tbk_comm.hip -- one process per GPU, one gather: thin RCCL wrappers.
//
// k-points shard with no data-path collective (every k, plaquette and string is
// computed from local data; slabs recompute their one halo row).  The only
// exchange is the final gather of eigenvalues / phases / partial flux sums,
// done with ncclAllGather over xGMI.  librccl is dlopen'ed on first use so a
// single-GPU user never loads it.
#include <dlfcn.h>
#include <string.h>
#include "tbk_internal.h"

namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void** comm, int nranks, nccl_uid id, int rank);
typedef int (*fn_destroy)(void* comm);
typedef int (*fn_allgather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t s);
typedef int (*fn_sendrecv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s);
typedef int (*fn_group)(void);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* lib = nullptr;
    fn_get_uid get_uid = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_sendrecv send = nullptr, recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_errstr errstr = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return TBK_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* lib = nullptr;
    for (const char* n : names) {
        lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    TBK_REQUIRE(lib, TBK_ECOMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.get_uid = (fn_get_uid)dlsym(lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.allgather = (fn_allgather)dlsym(lib, "ncclAllGather");
    g_rccl.send = (fn_sendrecv)dlsym(lib, "ncclSend");
    g_rccl.recv = (fn_sendrecv)dlsym(lib, "ncclRecv");
    g_rccl.group_start = (fn_group)dlsym(lib, "ncclGroupStart");
    g_rccl.group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
    g_rccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    TBK_REQUIRE(g_rccl.get_uid && g_rccl.init_rank && g_rccl.destroy && g_rccl.allgather, TBK_ECOMM,
                "librccl.so lacks an expected symbol");
    g_rccl.lib = lib;
    return TBK_OK;
}
const char* nccl_msg(int rc) { return g_rccl.errstr ? g_rccl.errstr(rc) : "?"; }
}  // namespace

extern "C" int tbk_comm_unique_id(unsigned char id_out[128]) {
    TBK_REQUIRE(id_out, TBK_EINVAL, "tbk_comm_unique_id: null id");
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid id;
    memset(&id, 0, sizeof(id));
    const int nrc = g_rccl.get_uid(&id);
    TBK_REQUIRE(nrc == 0, TBK_ECOMM, "ncclGetUniqueId: %s", nccl_msg(nrc));
    memcpy(id_out, &id, 128);
    return TBK_OK;
}

extern "C" int tbk_comm_init(tbk_ctx* ctx, const unsigned char id[128], int nranks, int rank) {
    TBK_REQUIRE(ctx && id && nranks >= 1 && rank >= 0 && rank < nranks, TBK_EINVAL, "tbk_comm_init: bad argument");
    TBK_REQUIRE(!ctx->comm, TBK_EINVAL, "tbk_comm_init: communicator already initialised");
    int rc = load_rccl();
    if (rc) return rc;
    TBK_HIP(hipSetDevice(ctx->device));
    nccl_uid uid;
    memcpy(&uid, id, 128);
    void* comm = nullptr;
    const int nrc = g_rccl.init_rank(&comm, nranks, uid, rank);
    TBK_REQUIRE(nrc == 0, TBK_ECOMM, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, nccl_msg(nrc));
    ctx->comm = comm;
    ctx->comm_nranks = nranks;
    ctx->comm_rank = rank;
    return TBK_OK;
}

extern "C" int tbk_comm_destroy(tbk_ctx* ctx) {
    TBK_REQUIRE(ctx, TBK_EINVAL, "tbk_comm_destroy: null ctx");
    if (ctx->comm && g_rccl.destroy) {
        (void)hipStreamSynchronize(ctx->stream);
        g_rccl.destroy(ctx->comm);
    }
    ctx->comm = nullptr;
    ctx->comm_nranks = 0;
    ctx->comm_rank = -1;
    return TBK_OK;
}

extern "C" int tbk_comm_allgather_f64(tbk_ctx* ctx, const double* send_dev, double* recv_dev, int64_t count) {
    TBK_REQUIRE(ctx && ctx->comm, TBK_ECOMM, "tbk_comm_allgather_f64: communicator not initialised");
    TBK_REQUIRE(send_dev && recv_dev && count >= 0, TBK_EINVAL, "tbk_comm_allgather_f64: bad argument");
    const int nccl_float64 = 8;  // ncclDouble
    const int nrc = g_rccl.allgather(send_dev, recv_dev, (size_t)count, nccl_float64, ctx->comm, ctx->stream);
    TBK_REQUIRE(nrc == 0, TBK_ECOMM, "ncclAllGather: %s", nccl_msg(nrc));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

// Ranks contribute different counts (513 strings over 8 ranks, slabs of 257 planes, k lists that do not divide): one
// grouped exchange in which every rank sends its block to every rank and receives block r at displs[r] -- the standard
// all-gather-v on point-to-point links, which is what xGMI is.  counts / displs are host arrays of nranks entries (in
// doubles).  The ROWS form gathers a band-major array in place: rank r holds send[nrows][counts[r]] (its k-chunk of
// eval[band][k], pythtb.py:1040,1053-1067) and every receiver gets recv[nrows][row_stride] with rank r's piece of row b
// at b*row_stride + displs[r] -- nrows messages per pair of ranks, no relayout pass afterwards.
//
// root < 0: every rank receives (all-gather-v).  root >= 0: only `root` receives (gather-v, SURVEY.md 8e: the reference's
// ret_eval is ONE array on ONE caller, pythtb.py:1040): the other ranks send their rows to the root and neither allocate
// nor receive the (nrows, row_stride) array -- 1/nranks of the bytes on every link of the all-gather form.
//
// The point-to-point operations go out in groups of at most TBK_COMM_GROUP_ROWS rows (x nranks sends + receives each): a
// ribbon's eigenvalue array has hundreds of rows, and one ncclGroup of thousands of operations runs into RCCL's per-group
// work limits; consecutive groups on one stream keep the order.
#define TBK_COMM_GROUP_ROWS 32
static int gatherv_rows(tbk_ctx* ctx, const double* send_dev, int64_t nrows, int64_t count, double* recv_dev,
                        const int64_t* counts, const int64_t* displs, int64_t row_stride, int root, const char* who) {
    TBK_REQUIRE(ctx && ctx->comm, TBK_ECOMM, "%s: communicator not initialised", who);
    TBK_REQUIRE(root < ctx->comm_nranks, TBK_EINVAL, "%s: root %d of %d ranks", who, root, ctx->comm_nranks);
    const bool receives = root < 0 || root == ctx->comm_rank;
    TBK_REQUIRE((recv_dev || !receives) && counts && displs && count >= 0 && nrows >= 1 && (send_dev || count == 0), TBK_EINVAL,
                "%s: bad argument", who);
    TBK_REQUIRE(g_rccl.send && g_rccl.recv && g_rccl.group_start && g_rccl.group_end, TBK_ECOMM,
                "librccl.so lacks ncclSend/ncclRecv/ncclGroupStart/ncclGroupEnd");
    TBK_REQUIRE(ctx->comm_rank >= 0 && ctx->comm_rank < ctx->comm_nranks && counts[ctx->comm_rank] == count, TBK_EINVAL,
                "%s: counts[rank] (%lld) != count (%lld)", who,
                (long long)(ctx->comm_rank >= 0 && ctx->comm_rank < ctx->comm_nranks ? counts[ctx->comm_rank] : -1), (long long)count);
    for (int r = 0; r < ctx->comm_nranks; ++r)
        TBK_REQUIRE(counts[r] >= 0 && displs[r] >= 0 && (nrows == 1 || displs[r] + counts[r] <= row_stride), TBK_EINVAL,
                    "%s: block %d (displacement %lld, count %lld) does not fit a row of %lld", who, r, (long long)displs[r],
                    (long long)counts[r], (long long)row_stride);
    const int nccl_float64 = 8;  // ncclDouble
    for (int64_t b0 = 0; b0 < nrows; b0 += TBK_COMM_GROUP_ROWS) {
        const int64_t b1 = b0 + TBK_COMM_GROUP_ROWS < nrows ? b0 + TBK_COMM_GROUP_ROWS : nrows;
        int nrc = g_rccl.group_start();
        TBK_REQUIRE(nrc == 0, TBK_ECOMM, "ncclGroupStart: %s", nccl_msg(nrc));
        for (int64_t b = b0; b < b1 && nrc == 0; ++b) {
            for (int r = 0; r < ctx->comm_nranks && nrc == 0; ++r) {
                const bool to_r = root < 0 || r == root;          // does rank r receive?
                if (to_r && count > 0)
                    nrc = g_rccl.send(const_cast<double*>(send_dev) + b * count, (size_t)count, nccl_float64, r, ctx->comm, ctx->stream);
                if (nrc == 0 && receives && counts[r] > 0)
                    nrc = g_rccl.recv(recv_dev + b * row_stride + displs[r], (size_t)counts[r], nccl_float64, r, ctx->comm, ctx->stream);
            }
        }
        const int erc = g_rccl.group_end();
        TBK_REQUIRE(nrc == 0, TBK_ECOMM, "ncclSend/ncclRecv: %s", nccl_msg(nrc));
        TBK_REQUIRE(erc == 0, TBK_ECOMM, "ncclGroupEnd: %s", nccl_msg(erc));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

extern "C" int tbk_comm_allgatherv_f64(tbk_ctx* ctx, const double* send_dev, int64_t count, double* recv_dev,
                                       const int64_t* counts, const int64_t* displs) {
    return gatherv_rows(ctx, send_dev, 1, count, recv_dev, counts, displs, 0, -1, "tbk_comm_allgatherv_f64");
}

extern "C" int tbk_comm_allgatherv_rows_f64(tbk_ctx* ctx, const double* send_dev, int64_t nrows, int64_t count,
                                            double* recv_dev, const int64_t* counts, const int64_t* displs,
                                            int64_t row_stride) {
    return gatherv_rows(ctx, send_dev, nrows, count, recv_dev, counts, displs, row_stride, -1, "tbk_comm_allgatherv_rows_f64");
}

extern "C" int tbk_comm_gatherv_rows_f64(tbk_ctx* ctx, const double* send_dev, int64_t nrows, int64_t count,
                                         double* recv_dev, const int64_t* counts, const int64_t* displs,
                                         int64_t row_stride, int root) {
    TBK_REQUIRE(root >= 0, TBK_EINVAL, "tbk_comm_gatherv_rows_f64: root %d", root);
    return gatherv_rows(ctx, send_dev, nrows, count, recv_dev, counts, displs, row_stride, root, "tbk_comm_gatherv_rows_f64");
}
