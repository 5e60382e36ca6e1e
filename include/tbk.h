/* tbk.h -- C ABI of libtbk: the MI355X (gfx950) k-mesh tight-binding kernels.
 *
 * This is the drop-in boundary for PythTB's k-space hot path.  The reference
 * (PythTB 1.8.0, one pure-Python file) has no FFI of its own, so each entry
 * point below names the reference routine whose results it reproduces
 * (file:line in /root/reference/pythtb.py).  The Python side that binds these
 * with ctypes lives in pythtb_amd/_lib.py; INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a TBK_E* code otherwise;
 *     tbk_last_error() gives a thread-local message for the last failure.
 *   - plain pointers and sizes only.  "c128" = interleaved (re,im) doubles.
 *     Pointers are HOST pointers unless the parameter name ends in _dev.
 *   - state index n = 2*orbital+spin for nspin=2 (the reshape of pythtb.py:933).
 *   - a handle is bound to one device and one HIP stream; calls on a handle are
 *     synchronous from the caller's view unless the name ends in _async.
 *     Handles are not thread-safe (neither are the reference's objects).
 */
#ifndef TBK_H
#define TBK_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TBK_OK 0
#define TBK_EINVAL 1   /* bad argument (shape, range, null)            */
#define TBK_EHIP 2     /* a HIP runtime call failed                    */
#define TBK_ENOMEM 3   /* device or host allocation failed             */
#define TBK_EUNSUPPORTED 4 /* size outside what this build handles     */
#define TBK_ECOMM 5    /* RCCL not loadable / communicator failure     */
#define TBK_ENOCONV 6  /* an iterative kernel hit its sweep limit      */

#define TBK_MAX_DIM 4      /* dim_k, dim_arr <= 4      (pythtb.py:99)   */
#define TBK_MAX_NSTA 2048  /* states per k in this build                */
#define TBK_MAX_NOCC 16    /* largest band set of the per-thread Berry kernels (used up to 8 bands, 2 for Wilson-loop eigenphases); larger sets take the workgroup-level paths, any size */

typedef struct tbk_ctx tbk_ctx;     /* device + stream + workspaces          */
typedef struct tbk_model tbk_model; /* flattened hopping table on the device */
typedef struct tbk_wfs tbk_wfs;     /* device-resident wf_array._wfs         */

const char* tbk_last_error(void);
int tbk_version(void);
int tbk_device_count(int* count);
/* TBK_* environment knobs are parsed once, on first use; re-read them after changing one */
int tbk_knobs_reload(void);
/* 1 if this library was built with -DTBK_DIAG (ablation branches compiled into the kernels), else 0 */
int tbk_build_has_diagnostics(void);

/* ---- context -------------------------------------------------------- */
int tbk_ctx_create(int device, tbk_ctx** out);
int tbk_ctx_destroy(tbk_ctx* ctx);
int tbk_ctx_sync(tbk_ctx* ctx);
int tbk_ctx_device_info(tbk_ctx* ctx, char* name, int name_cap, int* compute_units,
                        int64_t* hbm_bytes);

/* raw device memory for callers that keep inputs/outputs resident (bench) */
int tbk_dev_alloc(tbk_ctx* ctx, int64_t bytes, void** ptr_dev);
int tbk_dev_free(tbk_ctx* ctx, void* ptr_dev);
int tbk_dev_upload(tbk_ctx* ctx, void* dst_dev, const void* src, int64_t bytes);
int tbk_dev_download(tbk_ctx* ctx, void* dst, const void* src_dev, int64_t bytes);

/* HIP-event timing on the context's stream (the stream kernels run on).
 * tbk_timer_*: one bracket around arbitrary calls.  tbk_prof_*: per-kernel
 * brackets recorded inside the library around every launch while enabled. */
int tbk_timer_begin(tbk_ctx* ctx);
int tbk_timer_end(tbk_ctx* ctx, double* elapsed_ms);
/* period: 0 = off, 1 = bracket every launch, N = bracket every N-th launch (an event
 * record costs about 3 us of stream time, so timed loops sample)                     */
int tbk_prof_enable(tbk_ctx* ctx, int period);
int tbk_prof_reset(tbk_ctx* ctx);
/* median duration of an EMPTY bracket (two event records, nothing between them)        */
int tbk_prof_calibrate(tbk_ctx* ctx, int reps, double* median_ms);
int tbk_prof_count(tbk_ctx* ctx, int* n_kernels);
int tbk_prof_get(tbk_ctx* ctx, int index, char* name, int name_cap, int64_t* launches,
                 double* total_ms);

/* ---- model tables (tb_model._site_energies/_hoppings, pythtb.py:171-180,
 *      :475-478; consumed by _gen_ham :874-925) ------------------------ */
/* orb:     norb x dim_k  reduced orbital coordinates, periodic components only
 *          (_orb[:, _per], :912-914)
 * onsite:  norb x nspin x nspin c128 (nspin=1: the real site energy as c128)
 * hop_R:   nhop x dim_k  integer lattice vectors, periodic components only
 * hop_amp: nhop x nspin x nspin c128 (the block of _val_to_block :517-560)  */
int tbk_model_upload(tbk_ctx* ctx, int dim_k, int norb, int nspin, const double* orb,
                     const double* onsite, int64_t nhop, const int32_t* hop_i,
                     const int32_t* hop_j, const int32_t* hop_R, const double* hop_amp,
                     tbk_model** out);
int tbk_model_free(tbk_model* model);
/* Host-only introspection of what tbk_model_upload builds (no device needed): the merged term list of
 * S_ab(k) = sum_t amp_t exp(2 pi i k.R_t), a <= b, slot = a*n - a(a-1)/2 + (b-a), in the order the kernels
 * read it.  term_cap = capacity of term_slot[], term_R[][4], term_amp[] (c128); *nterm = number of terms
 * (call with term_cap = 0 to size).  info[4] = {pmax, nR, nnz, nslot}.  For tests and sanitizer builds.   */
int tbk_model_flatten_host(int dim_k, int norb, int nspin, const double* orb, const double* onsite,
                           int64_t nhop, const int32_t* hop_i, const int32_t* hop_j, const int32_t* hop_R,
                           const double* hop_amp, int64_t term_cap, int64_t* nterm, int32_t* term_slot,
                           int32_t* term_R, double* term_amp, int32_t* info);
int tbk_model_info(tbk_model* model, int* dim_k, int* nsta, int64_t* nterm);

/* Host-only: continuity of Berry phases along the first index of a result array (berry_phase(contin=True), pythtb.py:2980-3036).
 * tbk_one_phase_cont  = _one_phase_cont  (pythtb.py:3876-3889): pha[n] unwrapped by 2 pi steps, the first entry anchored at clos.
 * tbk_array_phases_cont = _array_phases_cont (pythtb.py:3891-3921): arr[n0][nb] eigenphase sets; set i is matched greedily to the
 * previous (already continuous) set by distance on the unit circle -- the LAST index among equal minima, like the reference's <= --
 * and unwrapped; clos[nb] anchors set 0.  `stride` = doubles between consecutive sets / entries (a column of a larger array).
 * No device call, no context.                                                                                                   */
int tbk_one_phase_cont(const double* pha, int64_t n, int64_t stride, double clos, double* out, int64_t out_stride);
int tbk_array_phases_cont(const double* arr, int64_t n0, int nb, int64_t stride, const double* clos, double* out,
                          int64_t out_stride);

/* ---- H(k) and eigen-solve ------------------------------------------- */
/* _gen_ham (pythtb.py:874-925) for nk points: ham_out[nk][nsta][nsta] c128.
 * k: nk x dim_k reduced coordinates (ignored when dim_k == 0).            */
int tbk_gen_ham(tbk_model* model, const double* k, int64_t nk, double* ham_out);

/* solve_all (pythtb.py:955-1079) = _gen_ham + _sol_ham (:927-953) fused:
 * eval[nsta][nk] ascending per k; evec[nsta][nk][nsta] c128 (rows are
 * eigenvectors, band-major) or NULL for eigenvalues only.                  */
int tbk_solve_list(tbk_model* model, const double* k, int64_t nk, double* eval, double* evec);
/* same, all buffers already on the device (no PCIe in the call)           */
int tbk_solve_list_dev(tbk_model* model, const double* k_dev, int64_t nk, double* eval_dev,
                       double* evec_dev);
/* tbk_solve_list_dev followed by the read (and reset) of the context's sticky solver status -- one host synchronisation:
 * TBK_ENOCONV where numpy.linalg.eigh would raise (pythtb.py:939-944: an iteration limit, a NaN model); a rotation-record
 * overflow of the direct solvers is repeated once on the Jacobi kernels.  What every caller that hands results on (a gather,
 * a download) must use: the unchecked form leaves the status set for the next checked call on the context.               */
int tbk_solve_list_dev_checked(tbk_model* model, const double* k_dev, int64_t nk, double* eval_dev,
                               double* evec_dev);

/* _sol_ham (pythtb.py:927-953) on caller-supplied Hermitian matrices
 * ham[nk][n][n] c128 -> eval[n][nk], evec[n][nk][n] (or NULL).            */
int tbk_eigh_batch(tbk_ctx* ctx, int n, const double* ham, int64_t nk, double* eval,
                   double* evec);

/* Which of the eigen-solver's regimes a batch would take (host only, no device): the dispatch of _sol_ham's replacement is
 * ONE table of (states, eigenvectors?, input form, batch window) rows with the measured crossovers as data
 * (tbk_solve.hip, kRegimeRules).  form: 0 k list, 1 regular mesh, 2 supplied matrices; nk: matrices of the call; batch:
 * matrices of the global mesh (= nk for lists); compute_units <= 0: 256.  Returns a static name ("trig", "blocked", "big",
 * "reg", "ql16", "qlw", "row16", "wg_lds", "wg_global", "wave", "ql_small", "closed_form"); *note_out (nullable) the
 * measurement behind the matching row.                                                                                  */
const char* tbk_solver_regime(int n, int with_vectors, int form, int64_t nk, int64_t batch,
                              int compute_units, int has_rblocks, const char** note_out);

/* ---- wf_array storage (pythtb.py:2388-2419): _wfs[k1..kD][state][comp] */
int tbk_wfs_create(tbk_ctx* ctx, int dim_arr, const int32_t* mesh, int nsta_arr, int ncomp,
                   tbk_wfs** out);
int tbk_wfs_free(tbk_wfs* wfs);
int tbk_wfs_upload(tbk_wfs* wfs, const double* host_c128);
int tbk_wfs_download(tbk_wfs* wfs, double* host_c128);
int tbk_wfs_device_ptr(tbk_wfs* wfs, void** ptr_dev, int64_t* bytes);
/* wf_array.choose_states (pythtb.py:2568-2608) on the device: dst (same mesh and ncomp, nsta_arr = nb) receives the
 * states bands[0..nb) of src.  Band planes are contiguous in HBM, so this is nb device-to-device copies.            */
int tbk_wfs_copy_bands(tbk_wfs* dst, tbk_wfs* src, const int32_t* bands, int nb);
/* wf_array.__getitem__/__setitem__ (pythtb.py:2644-2672) on a resident array: copy the states of
 * npoints mesh points (row-major mesh indices) to/from host[npoints][nsta_arr][ncomp] c128 without
 * moving the rest of the array.                                                              */
int tbk_wfs_download_points(tbk_wfs* wfs, const int64_t* point_index, int64_t npoints, double* host_c128);
int tbk_wfs_upload_points(tbk_wfs* wfs, const int64_t* point_index, int64_t npoints, const double* host_c128);
/* bytes and calls of wf_array traffic across PCIe since the last reset (whole-array upload/download
 * and the per-point forms): lets callers and tests assert that a script causes no re-uploads. */
int tbk_ctx_transfer_stats(tbk_ctx* ctx, int64_t* h2d_bytes, int64_t* d2h_bytes, int64_t* h2d_calls,
                           int64_t* d2h_calls, int reset);
/* diagnostics of the eigen-solver since the last reset: `listed_matrices` = matrices of 9..16 states that the direct kernels
 * (k_e16 / k_tw16_*) could not finish themselves (three or more eigenvalues of a block within gaptol, a failed residual)
 * and handed to the QL-replay fallback.  Results are the same either way; the count tells a regression of the in-kernel
 * repairs from a healthy launch without timing anything.  Synchronises the context's stream.  (No reference counterpart:
 * numpy.linalg.eigh, pythtb.py:939-947, has one path.)                                                              */
int tbk_ctx_solver_stats(tbk_ctx* ctx, int64_t* listed_matrices, int reset);

/* solve_on_grid (pythtb.py:2421-2532): every mesh point i_d < N_d-1 solved at
 * start_k[d] + i_d/(N_d-1); the points with i_d == N_d-1 are the impose_pbc
 * images (:2729-2747), produced in the same launch by solving k(i_d=0) again
 * and multiplying by pbc_phase[d][comp] (c128, = exp(-2 pi i orb[:,per[d]])).
 * min_gaps[nsta-1] = min over solved points of E[b+1]-E[b] (:2495,:2530).
 * Slab form for k-sharding along mesh axis 0: the handle holds rows
 * [row0, row0+mesh[0]) of a global mesh whose axis-0 size is global_n0
 * (pass row0=0, global_n0=mesh[0] for the whole mesh).                     */
int tbk_wfs_solve_grid(tbk_wfs* wfs, tbk_model* model, const double* start_k,
                       const double* pbc_phase, int64_t row0, int64_t global_n0,
                       double* min_gaps);
/* Launch-only forms (no host synchronisation, no PCIe traffic): results stay
 * in device buffers owned by the handle until the matching *_result call.  */
int tbk_wfs_solve_grid_async(tbk_wfs* wfs, tbk_model* model, const double* start_k,
                             const double* pbc_phase, int64_t row0, int64_t global_n0);
int tbk_wfs_solve_grid_result(tbk_wfs* wfs, double* min_gaps);
/* General sharding window: the handle holds the points [offset[d], offset[d]+mesh[d]) of a
 * global mesh of global_mesh[d] points along every axis d (k-sharding of Berry strings
 * cuts an axis other than 0).  Results through tbk_wfs_solve_grid_result.              */
int tbk_wfs_solve_window_async(tbk_wfs* wfs, tbk_model* model, const double* start_k,
                               const double* pbc_phase, const int64_t* offset,
                               const int64_t* global_mesh);
/* solve_on_grid (:2421-2532) followed by berry_flux(occ, dirs=[0,1]) (:3068-3205, _one_flux_plane :3840-3865) in ONE pass over
 * a 2-D array of 2 or 4 states: the plaquette phases are formed from the eigenvectors while they are in registers, so the
 * array is written once and never read back.  occ: 1 or 2 bands.  Launch only; the results through
 * tbk_wfs_solve_grid_result (min gaps) and tbk_berry_flux_result (the total: same value as the two separate calls up to the
 * order of the sum).  TBK_EUNSUPPORTED where the fused kernel does not apply (other dimensions, state counts, long-ranged
 * models along the last axis): issue the two calls then.                                                              */
int tbk_wfs_solve_grid_flux_async(tbk_wfs* wfs, tbk_model* model, const double* start_k,
                                  const double* pbc_phase, int64_t row0, int64_t global_n0,
                                  const int32_t* occ, int nocc);
/* impose_pbc (:2674-2749) / impose_loop (:2751-2791) on a filled array:
 * last slice along mesh_dir = first slice * phase[comp] (phase NULL: copy) */
int tbk_wfs_impose(tbk_wfs* wfs, int mesh_dir, const double* phase_c128);

/* ---- Berry quantities ----------------------------------------------- */
/* berry_flux (pythtb.py:3068-3205) / _one_flux_plane (:3840-3865):
 * plaquette phases on the (dir0,dir1) planes for bands occ[nocc].
 * totals[n_slices]: sum over each plane (slices = remaining axes, row-major
 * in original axis order).  plaq (nullable): [n_slices][N_dir0-1][N_dir1-1].
 * The sum is a fixed-shape tree (no float atomics): bit-reproducible.       */
int tbk_berry_flux(tbk_wfs* wfs, const int32_t* occ, int nocc, int dir0, int dir1,
                   double* totals, double* plaq);

int tbk_berry_flux_async(tbk_wfs* wfs, const int32_t* occ, int nocc, int dir0, int dir1,
                         int want_plaq);
int tbk_berry_flux_result(tbk_wfs* wfs, double* totals, double* plaq);

/* berry_phase (pythtb.py:2863-3066) / _one_berry_loop (:3798-3838) for every
 * string along `dir` (strings = remaining axes, row-major in original order).
 * berry_evals == 0: out[n_strings]        = -arg det prod_i M_i
 * berry_evals != 0: out[n_strings][nocc]  = sorted -arg eig prod_i polar(M_i)
 * (the contin post-processing :3036-3065 is O(n_strings) host work).        */
int tbk_berry_phase(tbk_wfs* wfs, const int32_t* occ, int nocc, int dir, int berry_evals,
                    double* out);

/* ---- position operator / hybrid Wannier functions (first "next" row) ----
 * tb_model.position_matrix (pythtb.py:2034-2098), position_expectation (:2100-2141) and
 * position_hwf (:2143-2279), batched over nk points.
 * evec[nk][nsub][ncomp] c128: the states at each point; pos[ncomp]: reduced coordinate of
 * each component's orbital along the chosen non-periodic direction (repeated per spin).
 * xmat (nullable)  [nk][nsub][nsub] c128      X_mn = <u_m| r |u_n>
 * hwfc (nullable)  [nk][nsub]                 eigenvalues of X, ascending
 * hwf  (nullable)  [nk][nsub][nsub]  rows = eigenvectors of X on the input states, or with
 *                  orbital_basis != 0 [nk][nsub][ncomp] expanded on the orbitals (:2262-2277) */
int tbk_position_hwf(tbk_ctx* ctx, const double* evec, int64_t nk, int nsub, int ncomp,
                     const double* pos, double* xmat, double* hwfc, double* hwf, int orbital_basis);

/* The same on the states `occ[nocc]` of a RESIDENT wf_array (wf_array.position_matrix / _expectation /
 * _hwf, pythtb.py:2793-2861): nothing but the results crosses PCIe.  point_index (nullable): row-major
 * mesh indices of the npoints points wanted; NULL = every mesh point in order (npoints ignored).
 * Output shapes as above with nk = number of points, nsub = nocc.                                   */
int tbk_wfs_position_hwf(tbk_wfs* wfs, const int64_t* point_index, int64_t npoints, const int32_t* occ, int nocc,
                         const double* pos, double* xmat, double* hwfc, double* hwf, int orbital_basis);

/* ---- k generators on the device, eigenvalue reductions (third "next" row) ----
 * tb_model.k_uniform_mesh (pythtb.py:1792-1861): k_dev[prod(mesh)][dim_k], point
 * (i_0,..) row-major = (i_0/N_0, ...); dim_k 1..3 like the reference.                  */
int tbk_k_uniform_mesh_dev(tbk_ctx* ctx, int dim_k, const int32_t* mesh, double* k_dev);
/* points [first, first+count) of that list only: the chunk one rank of a k-sharded solve_all owns  */
int tbk_k_uniform_mesh_range_dev(tbk_ctx* ctx, int dim_k, const int32_t* mesh, int64_t first,
                                 int64_t count, double* k_dev);
/* interpolation step of tb_model.k_path (pythtb.py:1978-1996): nodes[n_nodes][dim_k] and
 * node_index[n_nodes] (0 .. nk-1, increasing; both computed on the host, :1926-1976)
 * -> k_dev[nk][dim_k], bit-equal to the reference's k_vec.                             */
int tbk_k_path_dev(tbk_ctx* ctx, int dim_k, int n_nodes, const double* nodes,
                   const int32_t* node_index, int64_t nk, double* k_dev);
/* solve_all(k_uniform_mesh(mesh)) with the k list generated on the device (no upload):
 * eval[n][nk], evec[n][nk][n] or NULL.                                                  */
int tbk_solve_mesh(tbk_model* model, const int32_t* mesh, double* eval, double* evec);
/* The reduction of the reference's DOS example (examples/haldane.py:96-121: histogram of
 * solve_all over a uniform mesh) without downloading the eigenvalues: counts[n][nbins] per
 * band with np.histogram's bin rule for the given edges[nbins+1] (equal-width bins, last
 * bin closed), and/or the band extrema band_min[n], band_max[n] (each nullable;
 * nbins = 0 with edges = counts = NULL computes the extrema alone).                     */
int tbk_dos_mesh(tbk_model* model, const int32_t* mesh, int nbins, const double* edges,
                 int64_t* counts, double* band_min, double* band_max);

/* ---- multi-GPU: one process per GPU, k-points sharded, one gather ------
 * Thin RCCL wrappers (librccl is dlopen'ed on first use).  The 128-byte id is
 * created on rank 0 and distributed by the launcher (any out-of-band channel). */
int tbk_comm_unique_id(unsigned char id_out[128]);
int tbk_comm_init(tbk_ctx* ctx, const unsigned char id[128], int nranks, int rank);
int tbk_comm_destroy(tbk_ctx* ctx);
/* all ranks contribute count doubles from send_dev; recv_dev[nranks*count]   */
int tbk_comm_allgather_f64(tbk_ctx* ctx, const double* send_dev, double* recv_dev,
                           int64_t count);

/* uneven contributions (513 Berry strings over 8 ranks; slabs of a 257-plane mesh): rank r contributes
 * counts[r] doubles, received at recv_dev + displs[r] on every rank; counts/displs are HOST arrays of nranks
 * entries and count must equal counts[own rank].  One grouped ncclSend/ncclRecv exchange.                  */
int tbk_comm_allgatherv_f64(tbk_ctx* ctx, const double* send_dev, int64_t count, double* recv_dev,
                            const int64_t* counts, const int64_t* displs);
/* the eigenvalue gather of a sharded solve_all (ret_eval (nsta, nkp) band-major, pythtb.py:1040,1053-1067): rank r
 * holds send_dev[nrows][counts[r]] -- its contiguous chunk of the k list for every band -- and every rank receives
 * recv_dev[nrows][row_stride] with that chunk of row b at b*row_stride + displs[r].  Same single grouped exchange,
 * nrows messages per pair of ranks, so the result lands band-major with no relayout pass (config E: 16 messages of
 * 16.8 MB per pair).                                                                                              */
int tbk_comm_allgatherv_rows_f64(tbk_ctx* ctx, const double* send_dev, int64_t nrows, int64_t count,
                                 double* recv_dev, const int64_t* counts, const int64_t* displs,
                                 int64_t row_stride);
/* the ROOTED form of the same gather (the reference's ret_eval is one array on one caller,
 * pythtb.py:1040,1053-1067): only rank `root` receives recv_dev[nrows][row_stride]; on every other rank recv_dev may
 * be NULL -- those ranks send their rows to the root and allocate nothing of size nrows x row_stride (config E's
 * solve_all leg: 268 MB sent per rank instead of 2.1 GB received by each).  Grouped ncclSend / ncclRecv, at most 32
 * rows per ncclGroup (both rows forms).                                                                            */
int tbk_comm_gatherv_rows_f64(tbk_ctx* ctx, const double* send_dev, int64_t nrows, int64_t count,
                              double* recv_dev, const int64_t* counts, const int64_t* displs,
                              int64_t row_stride, int root);

#ifdef __cplusplus
}
#endif
#endif /* TBK_H */
