// tbk_internal.h -- shared host/device definitions for libtbk (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/tbk.h"

// ---------------------------------------------------------------- errors
void tbk_set_error(const char* fmt, ...);

#define TBK_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            tbk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),        \
                          __FILE__, __LINE__);                                          \
            return TBK_EHIP;                                                            \
        }                                                                               \
    } while (0)

#define TBK_REQUIRE(cond, code, ...)                                                    \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            tbk_set_error(__VA_ARGS__);                                                 \
            return (code);                                                              \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------- run-time knobs
// Every TBK_* environment variable the library honours, parsed ONCE (first use) into this struct;
// tbk_knobs_reload() re-reads the environment (tests and A/B scripts that change a knob mid-process).
// -1 / NaN = "not set: use the measured default at the point of use".  DESIGN.md section 8a lists them.
struct TbkKnobs {
    int big_from = -1;          // TBK_BIG_FROM      smallest n sent to the whole-chip solver regardless of batch size
    int blocked = -1;           // TBK_BLOCKED       1 forces the block-Jacobi solver, 0 forbids it
    int use_reg = 1;            // TBK_REG           0: n = 5..8 through the wavefront kernel
    int use_row16 = 1;          // TBK_ROW16         0: no DPP-row Jacobi kernel
    int use_ql16 = 1;           // TBK_QL16          0: n = 9..16 through the Jacobi kernels
    long long ql16_min = -1;    // TBK_QL16_MIN      batches of at most this many matrices stay on the workgroup-per-matrix Jacobi (default 8 x CUs)
    int trig_nt = -1;           // TBK_TRIG_NT       threads per matrix of the L2 Householder kernel (256 | 512 | 1024; at least n)
    int use_trig = 1;           // TBK_TRIG          0: eigenvalue-only n = 65..1024 through the Jacobi solvers instead of tridiagonalise + bisection
    int use_trigv = 1;          // TBK_TRIGV         0: eigenvectors of 65..1024 states through the Jacobi solvers instead of the direct method (tbk_solve_trigv.inl)
    int trigv_nc = -1;          // TBK_TRIGV_NC      columns per LDS strip of the back-transformation (4 | 8 | 16)
    int use_qlw = 1;            // TBK_QLW           0: n = 17..64 through the Jacobi kernels whatever the batch size
    long long qlw_min = -1;     // TBK_QLW_MIN       batches of at most this many matrices stay on the Jacobi kernels (default 8 x CUs)
    int qlw_replay_reg = 1;     // TBK_QLW_REPLAY_REG  0: n <= 32 replays the rotations on Z in LDS (the form n = 33..64 uses) instead of in registers
    int qlw_bisect = -1;        // TBK_QLW_BISECT    eigenvalue-only n = 17..64: 1 bisection, 0 lane-per-matrix QL (default: bisection below 16 x CUs matrices)
    int hh32 = 1;               // TBK_HH32          0: n = 18..32 tridiagonalised by the LDS workgroup kernel k_tridiag_lds instead of k_hh32 (matrix in registers); 2: k_hh32 from 17
    int tw32 = 1;               // TBK_TW32          0: eigenvectors of 17..32 states by replaying the QL rotations on every matrix (k_ql_replay_reg) instead of k_tw32_vectors (what models with paired levels at a generic k get by themselves; 3: not even those); 2: k_tw32_vectors on Q (k_hh32 accumulates Z) instead of on the reflector record
    int ql32 = 1;               // TBK_QL32          0: the QL iteration of 17..32 states with (d, e) in LDS and dynamic positions (k_tridiag_ql_lanes) instead of registers (k_ql32_lanes)
    int qlw_streams = 0;        // TBK_QLW_STREAMS   chunks of a 17..32-state batch in flight on side streams (default: 2 from 16384 matrices with eigenvectors; 1: one after the other)
    int qlw_nt = -1;            // TBK_QLW_NT        threads per matrix of the tridiagonalisation kernel (64 | 128 | 256 | 512)
    long long qlw_cap = -1;     // TBK_QLW_CAP       tests: rotations recorded per matrix (default 3 n^2 + 64)
    int qlw_ws_mb = -1;         // TBK_QLW_WS_MB     workspace budget of the tridiagonal paths in MiB (default 4096; 8192 for the n = 9..16 eigenvector path)
    int ql16_split = 1;         // TBK_QL16_SPLIT    0: n = 9..16 with eigenvectors in the single kernel instead of three (tridiagonalise | lane-per-matrix QL, recorded | replay)
    long long ql16_split_min = -1;  // TBK_QL16_SPLIT_MIN  smallest batch that takes the three-kernel form (default 8192)
    int tw16_streams = -1;      // TBK_TW16_STREAMS  chunks of the twisted-factorisation path in flight at once, 1..3 (default 3; 1 = the context's own stream alone, with per-kernel brackets)
    int e16 = 1;                // TBK_E16           0: n = 9..16 with eigenvectors through round 3's three kernels (tridiagonalise | QL eigenvalues | twisted vectors) instead of the ONE fused kernel k_e16
    int tw16 = 1;               // TBK_TW16          0: n = 9..16 with eigenvectors through the QL-replay three-kernel form instead of twisted-factorisation vectors
    int e16_evals = 1;          // TBK_E16_EVALS     0: eigenvalue-only lists of 9..16 states take round 3's pair of kernels instead of k_e16<.., false>
    int e16_cells = 1;          // TBK_E16_CELLS     0: k_e16 on a mesh stages every lattice vector per point even for models with many of them
    int e16_ns_full = 0;        // TBK_E16_NS_FULL   1: k_e16 takes the full Newton-Schulz step (matrix cores) for EVERY matrix, as round 4 did
    double tw16_gaptol = 1e-5;  // TBK_TW16_GAPTOL   relative eigenvalue gap (of one unreduced block) below which a matrix is solved again by QL replay
    int ql16_evonly = 1;        // TBK_QL16_EVONLY   0: eigenvalue-only n = 9..16 lists through the single replicated kernel instead of tridiagonalise + lane-per-matrix QL
    long long few_max = -1;     // TBK_FEW_MAX       largest n < 22 batch that gets a workgroup per matrix
    int few_warm = 1;           // TBK_FEW_WARM      0: workgroup solver always starts cold
    int few_nt = -1;            // TBK_FEW_NT        threads of the LDS workgroup solver
    int wg_nt = 1024;           // TBK_WG_NT         threads of the global-workspace workgroup solver
    int wave_run = -1;          // TBK_WAVE_RUN      chain length of the wavefront solver (1 = always cold)
    int fused_rows = -1;        // TBK_FUSED_ROWS    mesh rows per wave tile of the fused solve + flux kernel (default 6; 10 beyond the LLC)
    int zero_copy_kb = 1024;    // TBK_ZERO_COPY_KB  k list + results of a solve_all / solve_one call up to this size go through mapped host memory (0: always copy;
                                //                   64 until round 4: 500 k-points of a 2-band model with eigenvectors 57 -> 21 us, profiles/zero_copy_size_probe.py)
    int small_kpt = -1;         // TBK_SMALL_KPT     k points per lane of the n <= 4 list kernels: 1, 2; default 2 from 2^19 points
    int grid_occ = -1;          // TBK_GRID_OCC      cap on the resident wavefronts per SIMD of k_grid_rows (diagnostic; default none)
    int fused_occ = -1;         // TBK_FUSED_OCC     cap on the resident wavefronts per SIMD of the fused kernel (default: none inside the LLC, 3 beyond)
    int fused_sum = 1;          // TBK_FUSED_SUM     0: the flux total of the fused pass by a kernel of its own (k_sum_fixed)
    int grid_seg = -1;          // TBK_GRID_SEG      chunks per wave tile of k_grid_rows
    int grid_kernel = 0;        // TBK_GRID_KERNEL   1: term-walking mesh kernel instead of the row-polynomial one
    int flux_ti = -1;           // TBK_FLUX_TI       rows per flux tile
    int grid_img = 1;           // TBK_GRID_IMG      0: k_grid_rows solves the periodic-image column of a closed mesh row like any other (A/B)
    int pos_tile = 1;           // TBK_POS_TILE      0: position matrices of <= 8 states by the thread-per-entry kernel (A/B)
    int poll_done = 1;          // TBK_POLL_DONE     0: small calls wait with hipStreamSynchronize instead of polling the completion word
    int flux_slices = 1;        // TBK_FLUX_SLICES   0: planes without the fastest mesh axis on the row kernel instead of k_flux_slices (lane = slice)
    int flux_fused = 0;         // TBK_FLUX_FUSED    1: final flux sum inside the kernel
    int flux_order = -1;        // TBK_FLUX_ORDER    0: row tiles newest rows first, 1: oldest first (default: by the array's size)
    int trigv_from = -1;        // TBK_TRIGV_FROM    smallest n of the workgroup-scale direct eigenvector path (default 65; A/B runs down to 17)
    int chain_ws_mb = 1024;     // TBK_CHAIN_WS_MB   link-matrix workspace per batch of strings, MiB
    int mesh_rows = 1;          // TBK_MESH_ROWS     0: eigenvalues on a generated uniform mesh by the list kernel on the generated list instead of the row kernel k_mesh_evals
    int chain_prod = 1;         // TBK_CHAIN_PROD    0: det-type berry_phase of 5..8 wide bands by link determinants (k_chain_links_tile + k_chain_lu_wave, a link-matrix workspace) instead of the string's matrix product on the matrix cores
    int chain_tile = 1;         // TBK_CHAIN_TILE    0: one link per wavefront step (k_chain_links_wave) also for 5..8 bands
    int chain_wave = 1;         // TBK_CHAIN_WAVE    0: thread-per-string link determinants also for 1..8 bands of wide (>= 8 component) states
    int chain_wave_from = -1;   // TBK_CHAIN_WAVE_FROM  smallest band count of the wave-per-string link kernels (default 1)
    int det_big_from = -1;      // TBK_DET_BIG_FROM  smallest band count of the workgroup-level link determinants
    int wilson_mfma = 1;        // TBK_WILSON_MFMA   0: Wilson loops of 5..8 wide bands on the workgroup-per-link kernels instead of k_chain_prod_tile<.., POLAR>
    int wilson_reg = 3;         // TBK_WILSON_REG    Wilson loops of 3-4 bands: 3 a lane per string / link, vectors through LDS (tbk_berry_lanes.inl);
                                //                   1 round 4's thread per segment (still what wide states that fit neither tile take); 0 the workgroup-per-link kernel
    int wilson_seg = -1;        // TBK_WILSON_SEG    test / probe hook: links per lane segment of the S form of tbk_berry_lanes.inl
    int wilson_swz = 1;         // TBK_WILSON_SWZ    0: no LDS swizzle of the components in tbk_berry_lanes.inl (A/B of the bank conflicts)
    int wilson_form = -1;       // TBK_WILSON_FORM   test / probe hook: 0 forces the S form (lane = string), 1 the L form (lane = link)
    int wilson_big_from = -1;   // TBK_WILSON_BIG_FROM  ... of the workgroup-level Wilson-loop pipeline
    long long wilson_batch_bytes = -1;   // TBK_WILSON_BATCH_BYTES  test hook: workspace bound per batch of strings
    double wilson_alpha = 0.0;  // TBK_WILSON_ALPHA  test hook: first Cayley angle
    bool wilson_alpha_set = false;
    long long big_batch = -1;   // TBK_BIG_BATCH     test hook: matrices per workspace batch
    int reg_lanes = 0;          // TBK_REG_LANES     (multilane build only)
    int reg_cells = 1;          // TBK_REG_CELLS     0: n = 5..8 meshes sum every lattice vector per point instead of the row's coefficient cells
    int reg_direct = 1;         // TBK_REG_DIRECT    0: n = 5..8 with eigenvectors by the register Jacobi kernel instead of the direct method
    int ablate_grid = 0;        // TBK_ABLATE_GRID   -DTBK_DIAG builds only
    int ablate_flux = 0;        // TBK_ABLATE_FLUX   -DTBK_DIAG builds only
};
const TbkKnobs& tbk_knobs();

// Ablation branches of the hot kernels exist only in the diagnostic build (make -C pythtb_amd/csrc diag ->
// libtbk_diag.so); in the shipped library the expression is never evaluated and the branches fold away.
#ifdef TBK_DIAG
#define TBK_ABLATE(expr) (expr)
#else
#define TBK_ABLATE(expr) 0
#endif

// ---------------------------------------------------------------- complex
// c128 as a plain pair; every operation spelled out so the compiler emits
// straight v_fma_f64 sequences (no library complex-multiply NaN fix-ups).
struct cd {
    double x, y;
};
__host__ __device__ inline cd cmake(double x, double y) { return cd{x, y}; }
__host__ __device__ inline cd cconj(cd a) { return cd{a.x, -a.y}; }
__host__ __device__ inline cd cadd(cd a, cd b) { return cd{a.x + b.x, a.y + b.y}; }
__host__ __device__ inline cd csub(cd a, cd b) { return cd{a.x - b.x, a.y - b.y}; }
__host__ __device__ inline cd cmul(cd a, cd b) {
    return cd{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// conj(a) * b
// the same product with the fused operations spelled out: two places of one kernel that must give the same bits (cmul leaves the
// choice of which product is fused to the compiler, site by site)
__host__ __device__ inline cd cmul_x(cd a, cd b) { return cd{fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)}; }
// acc += a b and acc += a conj(b), every operation a spelled-out fused multiply-add (see cmul_x)
__host__ __device__ inline void cfma_x(cd& acc, const cd a, const cd b) {
    acc.x = fma(-a.y, b.y, fma(a.x, b.x, acc.x));
    acc.y = fma(a.y, b.x, fma(a.x, b.y, acc.y));
}
__host__ __device__ inline void cfmac_x(cd& acc, const cd a, const cd b) {
    acc.x = fma(a.y, b.y, fma(a.x, b.x, acc.x));
    acc.y = fma(a.y, b.x, fma(-a.x, b.y, acc.y));
}
__host__ __device__ inline cd cmulc(cd a, cd b) {
    return cd{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};
}
__host__ __device__ inline cd cscale(cd a, double s) { return cd{a.x * s, a.y * s}; }
// acc += a*b
__host__ __device__ inline void cfma(cd& acc, cd a, cd b) {
    acc.x += a.x * b.x - a.y * b.y;
    acc.y += a.x * b.y + a.y * b.x;
}
// acc += conj(a)*b
__host__ __device__ inline void cfmac(cd& acc, cd a, cd b) {
    acc.x += a.x * b.x + a.y * b.y;
    acc.y += a.x * b.y - a.y * b.x;
}
__host__ __device__ inline double cabs2(cd a) { return a.x * a.x + a.y * a.y; }

// arg(z) = atan2(y, x).  Plaquette and link phases on a fine mesh are tiny, so the
// common case |y| <= 2^-6 x (x > 0) takes the odd Taylor series of atan through
// t^13 (truncation < 2^-90 |t|, i.e. below half an ulp of the result); anything
// else -- large phases, x <= 0, zeros, non-finite -- goes to the library atan2.
// a * b + c, a * c, a + c with the UNIFORM constant c in a scalar register pair.  Spelled out because the compiler otherwise
// copies every polynomial coefficient into a vector register first (v_mov_b32 x 2, or v_mov_b64 + v_fmac_f64): one to two extra
// VALU issue slots per Horner step, and -- hoisted out of a loop -- two VGPRs per coefficient for the whole kernel.  An s_mov
// issues on the scalar unit beside the other wavefronts' vector work.
__device__ __forceinline__ double tbk_fma_vs(const double a, const double b, const double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ double tbk_mul_vs(const double a, const double c) {
    double d;
    asm("v_mul_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c));
    return d;
}
__device__ __forceinline__ double tbk_add_vs(const double a, const double c) {
    double d;
    asm("v_add_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(c));
    return d;
}

__device__ __forceinline__ double arg_small_first(double y, double x) {
    if (x > 0.0 && fabs(y) <= 0.015625 * x) {
        const double t = y / x, t2 = t * t;
        double p = tbk_add_vs(tbk_mul_vs(t2, -1.0 / 13.0), 1.0 / 11.0);
        p = tbk_fma_vs(p, t2, -1.0 / 9.0);
        p = tbk_fma_vs(p, t2, 1.0 / 7.0);
        p = tbk_fma_vs(p, t2, -1.0 / 5.0);
        p = tbk_fma_vs(p, t2, 1.0 / 3.0);
        return fma(-(t * t2), p, t);
    }
    return atan2(y, x);
}

// ---------------------------------------------------------------- handles
struct ProfRec {
    const char* name;
    hipEvent_t t0, t1;
};
struct ProfAgg {
    std::string name;
    int64_t launches;
    double ms;
};

struct tbk_ctx {
    int device = 0;
    int cus = 0;
    bool qlw_off = false;  // set while a solve is repeated on the Jacobi kernels (the QL rotation record overflowed)
    hipStream_t stream = nullptr;
    // side streams of the chunked n = 9..16 solve (tbk_solve_tw16.inl): chunks alternate between them so that one chunk's
    // latency-bound eigenvalue kernel runs beside its neighbours' throughput-bound ones; created on first use
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [0]: fork point on `stream`; [1..3]: each side stream's end
    hipEvent_t timer0 = nullptr, timer1 = nullptr;
    int prof_period = 0;   // 0 off, 1 bracket every launch, N bracket every Nth launch
    unsigned prof_tick = 0;
    std::vector<ProfRec> prof_pending;
    std::vector<hipEvent_t> event_pool;
    std::vector<ProfAgg> prof_agg;
    // scratch reused across calls (grown on demand, stream-ordered use only)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    // host memory the device reads and writes directly (hipHostMalloc, mapped): small calls hand their k list over and take their
    // eigenvalues / eigenvectors back through it -- one stream synchronisation instead of a copy each way (tbk_solve_list)
    // table blobs of freed models (<= 1 MiB each, at most 8): a parameter sweep edits and re-uploads a small model per point, and a
    // hipMalloc + hipFree pair costs more than the solve it serves
    struct Blob { void* p; size_t bytes; };
    std::vector<Blob> blob_pool;
    void* zc_host = nullptr;
    void* zc_dev = nullptr;
    size_t zc_bytes = 0;
    void* pinned = nullptr;    // TBK_PINNED_BYTES (1 MiB) of pinned host memory for small results
    void* pinned_dev = nullptr;  // ... its device address (k_copy_small_signal writes small results there itself)
    int* flags_dev = nullptr;  // [64] sticky kernel status words (0: eigen no-convergence); [TBK_FLAG_LISTED, +1]: a 64-bit count of
                               // the matrices the direct n = 9..16 kernels listed for their QL-replay fallback (tbk_ctx_solver_stats)
    // completion word of small calls (tbk_done_arm / tbk_done_wait): 64 B of mapped host memory -- [0] the sequence number the last
    // kernel of the call stores after its results, [4..7] a copy of flags_dev[0..3] taken by that kernel; and its device-side
    // arrival counter.  Null when the mapped allocation failed or TBK_POLL_DONE=0: callers then synchronise the stream.
    unsigned* done_host = nullptr;
    unsigned* done_dev = nullptr;
    unsigned* done_cnt_dev = nullptr;
    unsigned done_seq = 0;
    void* work = nullptr;      // workspace of the workgroup-per-matrix eigen-solver (n > 64)
    size_t work_bytes = 0;
    // whole-array / per-point wf_array transfers across PCIe (tbk_ctx_transfer_stats)
    int64_t xfer_h2d_bytes = 0, xfer_d2h_bytes = 0, xfer_h2d_calls = 0, xfer_d2h_calls = 0;
    // RCCL
    void* rccl_lib = nullptr;
    void* comm = nullptr;
    int comm_nranks = 0, comm_rank = -1;
};

int tbk_ctx_scratch(tbk_ctx* ctx, size_t bytes, void** out);
int tbk_ctx_zero_copy(tbk_ctx* ctx, size_t bytes, void** host, void** dev);
// small device-to-host result (min gaps, flux totals, status words): through a pinned staging buffer -- an async copy into
// pageable memory is staged by the runtime and cost ~10 us more per call on the Python-API path -- then stream sync
int tbk_small_d2h(tbk_ctx* ctx, void* dst, const void* src_dev, size_t bytes);
// the same with the status words (flags_out[4] = ctx->flags_dev[0..3], nullable) in the same round trip
int tbk_small_result(tbk_ctx* ctx, void* dst, const void* src_dev, size_t bytes, int* flags_out);

// ---- "the call is finished" without hipStreamSynchronize.  Waiting on the runtime's completion signal costs ~12 us for a kernel
// of ~2 us; polling a word of mapped host memory that the call's LAST kernel stores after its results costs ~6.6 us
// (profiles/microbench/sync_latency.hip; hipDeviceScheduleSpin does not change the former).  The pair of calls of the headline
// step waits twice.  Use: `DoneArgs d = tbk_done_arm(ctx)` -> pass d to the last kernel, which calls tbk_signal_done(d) from ONE
// thread per workgroup after that workgroup's result stores -> `tbk_done_wait(ctx, d)` on the host (polls for at most ~1 ms,
// then falls back to hipStreamSynchronize, which is also what happens when d.word is null).  Results the host reads after the
// wait must have been stored by that last kernel (or sit in device memory).
#define TBK_PINNED_BYTES ((size_t)1 << 20)
#define TBK_FLAG_LISTED 16
struct DoneArgs {
    unsigned* word;        // device pointer of done_host[0]; null = not armed
    unsigned* cnt;         // arrival counter (device memory), zero between launches
    const int* flags_src;  // ctx->flags_dev, or null: copied to word[4..7] by the last workgroup
    unsigned seq;
};
DoneArgs tbk_done_arm(tbk_ctx* ctx, bool with_flags);
int tbk_done_wait(tbk_ctx* ctx, const DoneArgs& d);
#ifdef __HIPCC__
__device__ inline void tbk_signal_done(const DoneArgs& d) {
    if (!d.word) return;
    __threadfence_system();                       // this workgroup's results (mapped host memory or HBM) before its ticket
    const unsigned t = atomicAdd(d.cnt, 1u);
    if (t == gridDim.x * gridDim.y * gridDim.z - 1u) {
        __hip_atomic_store(d.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d.flags_src) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                d.word[4 + i] = (unsigned)__hip_atomic_load(d.flags_src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence_system();
        __hip_atomic_store(d.word, d.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#endif

// RAII bracket recording HIP events around one kernel launch when profiling.
struct ProfScope {
    tbk_ctx* ctx;
    const char* name;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    ProfScope(tbk_ctx* c, const char* n);
    ~ProfScope();
};

// Device-side view of the model (passed to kernels by value).
struct ModelView {
    int dim_k;   // 0..4
    int nsta;    // states per k  (norb*nspin)
    int nspin;   // 1|2
    int nslot;   // nsta*(nsta+1)/2 upper-triangular slots, row-major
    const int32_t* slot_ptr;  // [nslot+1] term ranges
    const int32_t* slot_ab;   // [nslot]  a | (b<<16)
    const cd* term_amp;       // [nterm]
    const int4* term_R;       // [nterm]  lattice vector (periodic comps, 0-padded)
    const double4* orb;       // [nsta]   reduced position of each state's orbital
    // the same terms seen as a polynomial in z_last = exp(2 pi i k_last): cell
    // (slot, p) holds the terms of `slot` whose last lattice component is p,
    // p in [-pmax, pmax]; terms are ordered (slot, p, ...) so cells are ranges
    int pmax;
    const int32_t* cell_ptr;  // [nslot*(2*pmax+1) + 1]
    // the same terms grouped by lattice vector (built for n = 5..16 always, up to 64 when the blocks
    // are dense; nR = 0 when not built):
    // S_slot(k) = sum_r rblock[r][slot] * exp(2 pi i k.rvec[r])
    int nR;
    const int4* rvec;         // [nR]
    const cd* rblock;         // [nR][nslot]
    // the non-empty slots only: {a | b<<16, t0, t1, 0}
    int nnz;
    const int4* nz;           // [nnz]
    // 17..32 states: 1 when the spectrum at a fixed generic k holds two levels closer than gaptol |H| -- spin-degenerate or Kramers-
    // paired bands: degenerate at EVERY k, and every matrix would go on k_tw32_vectors' list.  Found once, at upload (one k-point
    // solved); such models take the rotation replay from the start (launch_qlw).  A property of the model alone: every window and
    // every shard of a mesh decides the same way.
    int pairs_hint;
};

struct tbk_model {
    tbk_ctx* ctx = nullptr;
    int dim_k = 0, norb = 0, nspin = 1, nsta = 0, nslot = 0;
    int64_t nterm = 0;
    int64_t upload_id = 0; // unique per tbk_model_upload (cache keys must not use the blob address: hipMalloc reuses it)
    void* blob = nullptr;  // one device allocation holding all tables
    size_t blob_bytes = 0; // its size (small ones are parked in the context's pool when the model is freed)
    ModelView view{};
};

// Device layout of a wf_array: BAND-MAJOR planes, data[band][point][comp] with
// point = row-major mesh index.  (The reference / NumPy layout is
// [point][band][comp]; tbk_wfs_upload/download convert.)  Band-major keeps the
// bytes of an occupied subset contiguous, so berry_flux/berry_phase with
// nocc < nsta read only the occupied planes, and the solve kernels' stores are
// contiguous across lanes within each plane.
struct WfsView {
    int dim_arr;
    int nsta;   // states stored per mesh point (nsta_arr)
    int ncomp;  // components per state (norb*nspin)
    int mesh[TBK_MAX_DIM];
    int64_t stride[TBK_MAX_DIM];  // in mesh points
    int64_t npts;
    cd* data;
};
__device__ __forceinline__ cd* wf_at(const WfsView& v, int band, int64_t point) {
    return v.data + ((int64_t)band * v.npts + point) * v.ncomp;
}

#define TBK_GAP_SHARDS 64

struct tbk_wfs {
    tbk_ctx* ctx = nullptr;
    WfsView view{};
    int64_t bytes = 0;
    // solve_grid: min gaps as ordered bit patterns, [2 parities][TBK_GAP_SHARDS][ncomp];
    // each launch min-reduces into its parity and re-arms the other one for the next launch
    unsigned long long* gaps_dev = nullptr;
    int gaps_n = 0, gaps_parity = 0;
    double* gap_part_dev = nullptr;          // per-tile minima of the row kernel (n <= 4), [ntiles][n-1]
    int64_t gap_part_cap = 0, gap_part_n = 0; // gap_part_n > 0: the last solve wrote partials, not shards
    cd* pbc_dev = nullptr;                   // [TBK_MAX_DIM][nsta]
    // the last solve_grid launch, kept so that tbk_wfs_solve_grid_result can repeat it on other kernels
    struct tbk_model* last_model = nullptr;
    double last_start[TBK_MAX_DIM] = {0.0, 0.0, 0.0, 0.0};
    int64_t last_off[TBK_MAX_DIM] = {0, 0, 0, 0}, last_gmesh[TBK_MAX_DIM] = {1, 1, 1, 1};
    std::vector<double> last_pbc;
    // per-axis phase tables of the regular mesh (rebuilt only when their inputs change)
    cd* tab_dev = nullptr;                   // z[d][i] then f[d][i][n]
    int64_t tab_cap = 0;
    std::vector<double> tab_key;
    // flux results
    double* flux_totals_dev = nullptr;   // device pointer of the per-slice totals; mapped HOST memory when small (flux_totals_host != null)
    double* flux_totals_host = nullptr;  // ... its host address: the result is read after one synchronisation, no copy operation
    DoneArgs flux_done{nullptr, nullptr, nullptr, 0u};   // completion word armed by the pending berry_flux launch (word null: synchronise)
    unsigned* flux_cnt_dev = nullptr;        // [slices][16] arrival tickets of the row kernel
    int64_t flux_nslices = 0, flux_nslices_cap = 0;
    double* flux_plaq_dev = nullptr;
    int64_t flux_plaq_cap = 0, flux_plaq_n = 0;
    double* flux_partial_dev = nullptr;
    int64_t flux_partial_cap = 0;
};

// tbk_solve.hip: eigenvalues on k_uniform_mesh(mesh) by the row kernel (n <= 4); *done = false where it does not apply
int tbk_mesh_evals_rows(struct tbk_model* m, const int32_t* mesh, double* e_dev, bool* done);
// tbk_core.hip: (re)allocate / release the per-slice flux totals (mapped host memory up to 64 KB, device memory beyond)
int tbk_wfs_totals_alloc(struct tbk_wfs* w, int64_t nslices);
void tbk_wfs_totals_free(struct tbk_wfs* w);

// implemented in tbk_solve.hip: batched Hermitian eigen-solve, all pointers on the device
// (eval[n][nk], evec[n][nk][n]); tbk_eigh_check reports Jacobi non-convergence
int tbk_eigh_dev(tbk_ctx* ctx, int n, const cd* ham_dev, int64_t nk, double* eval_dev, cd* evec_dev,
                 const char* name);
int tbk_eigh_check(tbk_ctx* ctx, int n);
// solve + check in one (one host synchronisation); a rotation-record overflow of the direct solvers is repeated on the Jacobi
// kernels instead of being reported (the inputs must still be on the device)
int tbk_eigh_dev_checked(tbk_ctx* ctx, int n, const cd* ham_dev, int64_t nk, double* eval_dev, cd* evec_dev, const char* name);
// (tbk_solve_list_dev_checked: the same for k lists, exported -- include/tbk.h)
