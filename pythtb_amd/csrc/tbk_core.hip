// tbk_core.hip -- contexts, device memory, timing, model flattening, wf storage.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <map>
#include <new>
#include "tbk_internal.h"
#include <cmath>

static thread_local char g_err[1024] = "";

void tbk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tbk_last_error(void) { return g_err; }

// ------------------------------------------------------------------ knobs
static TbkKnobs g_knobs;
static bool g_knobs_parsed = false;
static void knobs_parse() {
    TbkKnobs k;
    auto geti = [](const char* name, int& dst) { if (const char* e = getenv(name)) dst = atoi(e); };
    auto getl = [](const char* name, long long& dst) { if (const char* e = getenv(name)) dst = atoll(e); };
    geti("TBK_BIG_FROM", k.big_from);
    geti("TBK_BLOCKED", k.blocked);
    geti("TBK_REG", k.use_reg);
    geti("TBK_ROW16", k.use_row16);
    geti("TBK_QL16", k.use_ql16);
    geti("TBK_QLW", k.use_qlw);
    geti("TBK_TRIG", k.use_trig);
    geti("TBK_TRIGV", k.use_trigv);
    geti("TBK_TRIGV_NC", k.trigv_nc);
    geti("TBK_TRIG_NT", k.trig_nt);
    getl("TBK_QLW_MIN", k.qlw_min);
    geti("TBK_QLW_NT", k.qlw_nt);
    geti("TBK_HH32", k.hh32);
    geti("TBK_TW32", k.tw32);
    geti("TBK_QL32", k.ql32);
    geti("TBK_QLW_STREAMS", k.qlw_streams);
    geti("TBK_QLW_BISECT", k.qlw_bisect);
    geti("TBK_QLW_REPLAY_REG", k.qlw_replay_reg);
    geti("TBK_QLW_WS_MB", k.qlw_ws_mb);
    getl("TBK_QLW_CAP", k.qlw_cap);
    geti("TBK_QL16_EVONLY", k.ql16_evonly);
    geti("TBK_TW16", k.tw16);
    geti("TBK_E16", k.e16);
    geti("TBK_E16_NS_FULL", k.e16_ns_full);
    geti("TBK_E16_CELLS", k.e16_cells);
    geti("TBK_E16_EVALS", k.e16_evals);
    geti("TBK_TW16_STREAMS", k.tw16_streams);
    if (const char* e = getenv("TBK_TW16_GAPTOL")) k.tw16_gaptol = atof(e);
    geti("TBK_QL16_SPLIT", k.ql16_split);
    getl("TBK_QL16_SPLIT_MIN", k.ql16_split_min);
    getl("TBK_QL16_MIN", k.ql16_min);
    getl("TBK_FEW_MAX", k.few_max);
    geti("TBK_FEW_WARM", k.few_warm);
    geti("TBK_FEW_NT", k.few_nt);
    geti("TBK_WG_NT", k.wg_nt);
    geti("TBK_WAVE_RUN", k.wave_run);
    geti("TBK_GRID_SEG", k.grid_seg);
    geti("TBK_FUSED_ROWS", k.fused_rows);
    geti("TBK_FUSED_SUM", k.fused_sum);
    geti("TBK_FUSED_OCC", k.fused_occ);
    geti("TBK_GRID_OCC", k.grid_occ);
    geti("TBK_SMALL_KPT", k.small_kpt);
    geti("TBK_ZERO_COPY_KB", k.zero_copy_kb);
    geti("TBK_GRID_KERNEL", k.grid_kernel);
    geti("TBK_FLUX_TI", k.flux_ti);
    geti("TBK_FLUX_FUSED", k.flux_fused);
    geti("TBK_FLUX_ORDER", k.flux_order);
    geti("TBK_REG_DIRECT", k.reg_direct);
    geti("TBK_REG_CELLS", k.reg_cells);
    geti("TBK_FLUX_SLICES", k.flux_slices);
    geti("TBK_POLL_DONE", k.poll_done);
    geti("TBK_POS_TILE", k.pos_tile);
    geti("TBK_GRID_IMG", k.grid_img);
    geti("TBK_CHAIN_WAVE", k.chain_wave);
    geti("TBK_CHAIN_TILE", k.chain_tile);
    geti("TBK_CHAIN_PROD", k.chain_prod);
    geti("TBK_MESH_ROWS", k.mesh_rows);
    geti("TBK_TRIGV_FROM", k.trigv_from);
    geti("TBK_CHAIN_WS_MB", k.chain_ws_mb);
    geti("TBK_CHAIN_WAVE_FROM", k.chain_wave_from);
    geti("TBK_DET_BIG_FROM", k.det_big_from);
    geti("TBK_WILSON_BIG_FROM", k.wilson_big_from);
    geti("TBK_WILSON_REG", k.wilson_reg);
    geti("TBK_WILSON_SEG", k.wilson_seg);
    geti("TBK_WILSON_SWZ", k.wilson_swz);
    geti("TBK_WILSON_FORM", k.wilson_form);
    geti("TBK_WILSON_MFMA", k.wilson_mfma);
    getl("TBK_WILSON_BATCH_BYTES", k.wilson_batch_bytes);
    if (const char* e = getenv("TBK_WILSON_ALPHA")) {
        k.wilson_alpha = atof(e);
        k.wilson_alpha_set = true;
    }
    getl("TBK_BIG_BATCH", k.big_batch);
#ifdef TBK_REG_MULTILANE
    geti("TBK_REG_LANES", k.reg_lanes);          // (the multi-lane instantiations of k_solve_reg exist in that build only)
#endif
#ifdef TBK_DIAG
    geti("TBK_ABLATE_GRID", k.ablate_grid);      // (the ablation branches are compiled into diagnostic builds only)
    geti("TBK_ABLATE_FLUX", k.ablate_flux);
#endif
    g_knobs = k;
    g_knobs_parsed = true;
}
const TbkKnobs& tbk_knobs() {
    if (!g_knobs_parsed) knobs_parse();
    return g_knobs;
}
extern "C" int tbk_knobs_reload(void) {
    knobs_parse();
    return TBK_OK;
}
extern "C" int tbk_build_has_diagnostics(void) {
#ifdef TBK_DIAG
    return 1;
#else
    return 0;
#endif
}
extern "C" int tbk_version(void) { return 100; }

extern "C" int tbk_device_count(int* count) {
    TBK_REQUIRE(count, TBK_EINVAL, "tbk_device_count: null argument");
    TBK_HIP(hipGetDeviceCount(count));
    return TBK_OK;
}

// ------------------------------------------------------------------ context
extern "C" int tbk_ctx_create(int device, tbk_ctx** out) {
    TBK_REQUIRE(out, TBK_EINVAL, "tbk_ctx_create: null out");
    int n = 0;
    TBK_HIP(hipGetDeviceCount(&n));
    TBK_REQUIRE(device >= 0 && device < n, TBK_EINVAL, "tbk_ctx_create: device %d of %d", device, n);
    TBK_HIP(hipSetDevice(device));
    tbk_ctx* c = new (std::nothrow) tbk_ctx();
    TBK_REQUIRE(c, TBK_ENOMEM, "tbk_ctx_create: out of host memory");
    c->device = device;
    hipDeviceProp_t prop;
    TBK_HIP(hipGetDeviceProperties(&prop, device));
    c->cus = prop.multiProcessorCount;
    TBK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    TBK_HIP(hipEventCreate(&c->timer0));
    TBK_HIP(hipEventCreate(&c->timer1));
    if (hipHostMalloc(&c->pinned, TBK_PINNED_BYTES, hipHostMallocMapped) != hipSuccess) c->pinned = nullptr;   // (optional: falls back to plain copies)
    if (c->pinned && hipHostGetDevicePointer(&c->pinned_dev, c->pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        c->pinned_dev = nullptr;
    }
    TBK_HIP(hipMalloc((void**)&c->flags_dev, 64 * sizeof(int)));
    TBK_HIP(hipMemsetAsync(c->flags_dev, 0, 64 * sizeof(int), c->stream));
    {   // completion word (tbk_done_arm): optional -- without it small calls synchronise the stream
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess &&
            hipMalloc((void**)&c->done_cnt_dev, 64) == hipSuccess) {
            memset(h, 0, 64);
            c->done_host = (unsigned*)h;
            c->done_dev = (unsigned*)d;
            TBK_HIP(hipMemsetAsync(c->done_cnt_dev, 0, 64, c->stream));
        } else {
            (void)hipGetLastError();
            if (h) hipHostFree(h);
            c->done_cnt_dev = nullptr;
        }
    }
    TBK_HIP(hipStreamSynchronize(c->stream));
    *out = c;
    return TBK_OK;
}

extern "C" int tbk_ctx_destroy(tbk_ctx* c) {
    if (!c) return TBK_OK;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->comm) tbk_comm_destroy(c);
    for (auto& r : c->prof_pending) {
        hipEventDestroy(r.t0);
        hipEventDestroy(r.t1);
    }
    for (auto e : c->event_pool) hipEventDestroy(e);
    if (c->scratch) hipFree(c->scratch);
    if (c->zc_host) hipHostFree(c->zc_host);
    for (auto& b : c->blob_pool) hipFree(b.p);
    c->blob_pool.clear();
    if (c->flags_dev) hipFree(c->flags_dev);
    if (c->done_host) hipHostFree(c->done_host);
    if (c->done_cnt_dev) hipFree(c->done_cnt_dev);
    if (c->pinned) hipHostFree(c->pinned);
    if (c->work) hipFree(c->work);
    hipEventDestroy(c->timer0);
    hipEventDestroy(c->timer1);
    for (int i = 0; i < 3; ++i)
        if (c->side[i]) hipStreamDestroy(c->side[i]);
    for (int i = 0; i < 4; ++i)
        if (c->side_ev[i]) hipEventDestroy(c->side_ev[i]);
    hipStreamDestroy(c->stream);
    delete c;
    return TBK_OK;
}

extern "C" int tbk_ctx_sync(tbk_ctx* c) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_ctx_sync: null ctx");
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}

extern "C" int tbk_ctx_device_info(tbk_ctx* c, char* name, int cap, int* cus, int64_t* hbm) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_ctx_device_info: null ctx");
    hipDeviceProp_t prop;
    TBK_HIP(hipGetDeviceProperties(&prop, c->device));
    if (name && cap > 0) snprintf(name, cap, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    return TBK_OK;
}

// a small result out of device memory: ONE kernel copies it into the context's pinned (device-visible) buffer, takes the status
// words along and stores the completion word; the host polls that (a hipMemcpyAsync + hipStreamSynchronize pair is ~17 us, this is
// ~9).  flags_out (nullable): ctx->flags_dev[0..3] as that kernel saw them.
__global__ __launch_bounds__(256) void k_copy_small_signal(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           const unsigned bytes, const DoneArgs done) {
    // (one workgroup per 32 KB; the last one to arrive stores the completion word: tbk_signal_done)
    const unsigned lo = blockIdx.x * 32768u, hi = min(bytes, lo + 32768u);
    if ((((size_t)src | (size_t)dst) & 7) == 0) {
        const unsigned n8 = hi >> 3;
        for (unsigned i = (lo >> 3) + threadIdx.x; i < n8; i += 256)
            reinterpret_cast<unsigned long long*>(dst)[i] = reinterpret_cast<const unsigned long long*>(src)[i];
        for (unsigned i = max(lo, n8 << 3) + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
    } else {
        for (unsigned i = lo + threadIdx.x; i < hi; i += 256) dst[i] = src[i];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) tbk_signal_done(done);
}

int tbk_small_result(tbk_ctx* c, void* dst, const void* src_dev, size_t bytes, int* flags_out) {
    if (bytes == 0 && !flags_out) return TBK_OK;
    if (c->pinned && c->pinned_dev && bytes + 64 <= TBK_PINNED_BYTES) {
        const DoneArgs d = tbk_done_arm(c, flags_out != nullptr);
        if (d.word) {
            hipLaunchKernelGGL(k_copy_small_signal, dim3((unsigned)std::max<size_t>(1, (bytes + 32767) / 32768)), dim3(256), 0, c->stream, (const unsigned char*)src_dev,
                               (unsigned char*)c->pinned_dev + 64, (unsigned)bytes, d);
            TBK_HIP(hipGetLastError());
            const int rc = tbk_done_wait(c, d);
            if (rc) return rc;
            memcpy(dst, (unsigned char*)c->pinned + 64, bytes);
            if (flags_out)
                for (int i = 0; i < 4; ++i) flags_out[i] = (int)c->done_host[4 + i];
            return TBK_OK;
        }
    }
    if (c->pinned && bytes + 64 <= TBK_PINNED_BYTES) {
        unsigned char* pin = (unsigned char*)c->pinned;
        if (bytes) TBK_HIP(hipMemcpyAsync(pin + 64, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        if (flags_out) TBK_HIP(hipMemcpyAsync(pin, c->flags_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        TBK_HIP(hipStreamSynchronize(c->stream));
        memcpy(dst, pin + 64, bytes);
        if (flags_out) memcpy(flags_out, pin, 4 * sizeof(int));
        return TBK_OK;
    }
    if (bytes) TBK_HIP(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    if (flags_out) TBK_HIP(hipMemcpyAsync(flags_out, c->flags_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}

int tbk_small_d2h(tbk_ctx* c, void* dst, const void* src_dev, size_t bytes) { return tbk_small_result(c, dst, src_dev, bytes, nullptr); }

DoneArgs tbk_done_arm(tbk_ctx* c, bool with_flags) {
    DoneArgs d{nullptr, nullptr, nullptr, 0u};
    // (profiling brackets the launch with events and reads them back after a synchronisation: keep that path as it was)
    if (!c->done_host || tbk_knobs().poll_done == 0 || c->prof_period != 0) return d;
    c->done_seq += 1u;
    if (c->done_seq == 0u) c->done_seq = 1u;
    d.word = c->done_dev;
    d.cnt = c->done_cnt_dev;
    d.flags_src = with_flags ? c->flags_dev : nullptr;
    d.seq = c->done_seq;
    return d;
}

int tbk_done_wait(tbk_ctx* c, const DoneArgs& d) {
    if (d.word) {
        const volatile unsigned* w = c->done_host;
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            for (int i = 0; i < 512; ++i) {
                if ((int)(__atomic_load_n(w, __ATOMIC_ACQUIRE) - d.seq) >= 0) return TBK_OK;   // (launches complete in stream order)
                __builtin_ia32_pause();
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;   // a long launch: sleep on the signal
        }
    }
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}

// The totals of a berry_flux launch (one double per slice): small ones live in mapped host memory, so that the kernel's own
// stores are the transfer and tbk_berry_flux_result is one synchronisation and a read (a hipMemcpyAsync of 8 bytes is 10 us of
// the 125 us two-call step at 2048^2)
void tbk_wfs_totals_free(tbk_wfs* w) {
    if (w->flux_totals_host) hipHostFree(w->flux_totals_host);
    else if (w->flux_totals_dev) hipFree(w->flux_totals_dev);
    w->flux_totals_host = nullptr;
    w->flux_totals_dev = nullptr;
}
int tbk_wfs_totals_alloc(tbk_wfs* w, int64_t nslices) {
    tbk_wfs_totals_free(w);
    const size_t bytes = (size_t)std::max<int64_t>(nslices, 1) * sizeof(double);
    if (bytes <= 64 * 1024 && tbk_knobs().zero_copy_kb > 0) {
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, bytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            w->flux_totals_host = (double*)h;
            w->flux_totals_dev = (double*)d;
            return TBK_OK;
        }
        (void)hipGetLastError();
        if (h) hipHostFree(h);
    }
    TBK_HIP(hipMalloc((void**)&w->flux_totals_dev, bytes));
    return TBK_OK;
}

int tbk_ctx_scratch(tbk_ctx* c, size_t bytes, void** out) {
    if (bytes > c->scratch_bytes) {
        TBK_HIP(hipStreamSynchronize(c->stream));
        if (c->scratch) TBK_HIP(hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
        size_t want = std::max(bytes, (size_t)1 << 20);
        hipError_t e = hipMalloc(&c->scratch, want);
        if (e != hipSuccess) {
            tbk_set_error("device scratch of %zu bytes: %s", want, hipGetErrorString(e));
            return TBK_ENOMEM;
        }
        c->scratch_bytes = want;
    }
    *out = c->scratch;
    return TBK_OK;
}

// mapped host memory of at least `bytes` (grown on demand; null when the platform refuses: callers then copy)
int tbk_ctx_zero_copy(tbk_ctx* c, size_t bytes, void** host, void** dev) {
    if (bytes > c->zc_bytes) {
        TBK_HIP(hipStreamSynchronize(c->stream));
        if (c->zc_host) hipHostFree(c->zc_host);
        c->zc_host = c->zc_dev = nullptr;
        c->zc_bytes = 0;
        const size_t want = std::max(bytes, (size_t)256 << 10);
        void* h = nullptr;
        void* d = nullptr;
        if (hipHostMalloc(&h, want, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (h) hipHostFree(h);
            *host = *dev = nullptr;
            return TBK_OK;
        }
        c->zc_host = h;
        c->zc_dev = d;
        c->zc_bytes = want;
    }
    *host = c->zc_host;
    *dev = c->zc_dev;
    return TBK_OK;
}

extern "C" int tbk_dev_alloc(tbk_ctx* c, int64_t bytes, void** p) {
    TBK_REQUIRE(c && p && bytes >= 0, TBK_EINVAL, "tbk_dev_alloc: bad argument");
    TBK_HIP(hipSetDevice(c->device));
    hipError_t e = hipMalloc(p, (size_t)std::max<int64_t>(bytes, 16));
    if (e != hipSuccess) {
        tbk_set_error("hipMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
        return TBK_ENOMEM;
    }
    return TBK_OK;
}
extern "C" int tbk_dev_free(tbk_ctx* c, void* p) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_dev_free: null ctx");
    if (p) {
        TBK_HIP(hipStreamSynchronize(c->stream));
        TBK_HIP(hipFree(p));
    }
    return TBK_OK;
}
extern "C" int tbk_dev_upload(tbk_ctx* c, void* dst, const void* src, int64_t bytes) {
    TBK_REQUIRE(c && dst && src && bytes >= 0, TBK_EINVAL, "tbk_dev_upload: bad argument");
    TBK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}
extern "C" int tbk_dev_download(tbk_ctx* c, void* dst, const void* src, int64_t bytes) {
    TBK_REQUIRE(c && dst && src && bytes >= 0, TBK_EINVAL, "tbk_dev_download: bad argument");
    TBK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    TBK_HIP(hipStreamSynchronize(c->stream));
    return TBK_OK;
}

// ------------------------------------------------------------------ timing
extern "C" int tbk_timer_begin(tbk_ctx* c) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_timer_begin: null ctx");
    TBK_HIP(hipEventRecord(c->timer0, c->stream));
    return TBK_OK;
}
extern "C" int tbk_timer_end(tbk_ctx* c, double* ms) {
    TBK_REQUIRE(c && ms, TBK_EINVAL, "tbk_timer_end: bad argument");
    TBK_HIP(hipEventRecord(c->timer1, c->stream));
    TBK_HIP(hipEventSynchronize(c->timer1));
    float f = 0.f;
    TBK_HIP(hipEventElapsedTime(&f, c->timer0, c->timer1));
    *ms = f;
    return TBK_OK;
}

static hipEvent_t prof_event(tbk_ctx* c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(tbk_ctx* c, const char* n) : ctx(c), name(n) {
    if (ctx && ctx->prof_period > 0 && (ctx->prof_tick++ % (unsigned)ctx->prof_period) == 0) {   // (null ctx: no bracket)
        t0 = prof_event(ctx);
        t1 = prof_event(ctx);
        hipEventRecord(t0, ctx->stream);
    }
}
ProfScope::~ProfScope() {
    if (t0) {
        hipEventRecord(t1, ctx->stream);
        ctx->prof_pending.push_back(ProfRec{name, t0, t1});
    }
}

static int prof_collect(tbk_ctx* c) {
    if (c->prof_pending.empty()) return TBK_OK;
    TBK_HIP(hipStreamSynchronize(c->stream));
    for (auto& r : c->prof_pending) {
        float f = 0.f;
        TBK_HIP(hipEventElapsedTime(&f, r.t0, r.t1));
        bool found = false;
        for (auto& a : c->prof_agg)
            if (a.name == r.name) {
                a.launches++;
                a.ms += f;
                found = true;
                break;
            }
        if (!found) c->prof_agg.push_back(ProfAgg{r.name, 1, (double)f});
        c->event_pool.push_back(r.t0);
        c->event_pool.push_back(r.t1);
    }
    c->prof_pending.clear();
    return TBK_OK;
}

extern "C" int tbk_prof_enable(tbk_ctx* c, int on) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_prof_enable: null ctx");
    c->prof_period = on < 0 ? 0 : on;
    c->prof_tick = 0;
    return TBK_OK;
}
// cost of an empty bracket (two event records back to back, nothing between): what every
// bracketed duration contains on top of the kernel itself
extern "C" int tbk_prof_calibrate(tbk_ctx* c, int reps, double* median_ms) {
    TBK_REQUIRE(c && median_ms && reps > 0, TBK_EINVAL, "tbk_prof_calibrate: bad argument");
    std::vector<float> v;
    hipEvent_t a, b;
    TBK_HIP(hipEventCreate(&a));
    TBK_HIP(hipEventCreate(&b));
    for (int i = 0; i < reps; ++i) {
        TBK_HIP(hipEventRecord(a, c->stream));
        TBK_HIP(hipEventRecord(b, c->stream));
        TBK_HIP(hipEventSynchronize(b));
        float f = 0.f;
        TBK_HIP(hipEventElapsedTime(&f, a, b));
        v.push_back(f);
    }
    TBK_HIP(hipEventDestroy(a));
    TBK_HIP(hipEventDestroy(b));
    std::sort(v.begin(), v.end());
    *median_ms = v[v.size() / 2];
    return TBK_OK;
}
extern "C" int tbk_prof_reset(tbk_ctx* c) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_prof_reset: null ctx");
    int rc = prof_collect(c);
    c->prof_agg.clear();
    return rc;
}
extern "C" int tbk_prof_count(tbk_ctx* c, int* n) {
    TBK_REQUIRE(c && n, TBK_EINVAL, "tbk_prof_count: bad argument");
    int rc = prof_collect(c);
    *n = (int)c->prof_agg.size();
    return rc;
}
extern "C" int tbk_prof_get(tbk_ctx* c, int i, char* name, int cap, int64_t* launches, double* ms) {
    TBK_REQUIRE(c && i >= 0 && i < (int)c->prof_agg.size(), TBK_EINVAL, "tbk_prof_get: index");
    if (name && cap > 0) snprintf(name, cap, "%s", c->prof_agg[i].name.c_str());
    if (launches) *launches = c->prof_agg[i].launches;
    if (ms) *ms = c->prof_agg[i].ms;
    return TBK_OK;
}

// ------------------------------------------------------------------ model
// Flatten the reference's hopping list into upper-triangular "slots":
//   S_ab(k) = sum_t amp_t * exp(2 pi i k.R_t)       (a <= b, state indices)
//   H_ab(k) = conj(e_a) e_b S_ab(k),  e_a = exp(2 pi i k.tau_a)
// which is _gen_ham's  amp*exp(2 pi i k.(R + tau_b - tau_a))  (pythtb.py:912-924)
// with the orbital phases factored out (H = D^+ S D, D unitary diagonal).  The
// conjugate entries H_ba are implied; a hop whose state pair is below the
// diagonal is stored as (b,a,-R,conj amp); an i==j hop contributes both
// (amp,R) and (conj amp,-R) to its diagonal slot (:919-924).
namespace {
struct TermKey {
    int slot;
    int last;  // index of the last periodic component (ordered first within a slot)
    int R[4];
    bool operator<(const TermKey& o) const {
        if (slot != o.slot) return slot < o.slot;
        if (R[last] != o.R[last]) return R[last] < o.R[last];
        for (int d = 0; d < 4; ++d)
            if (R[d] != o.R[d]) return R[d] < o.R[d];
        return false;
    }
};
inline int slot_of(int n, int a, int b) { return a * n - a * (a - 1) / 2 + (b - a); }
}  // namespace

namespace {
// Everything tbk_model_upload derives from the reference's tables, on the host: the slot-major term list and
// the packed blob the kernels read, with the offsets of its pieces.
struct FlatModel {
    int n = 0, nslot = 0, pmax = 0, nR = 0, nnz = 0;
    int64_t nterm = 0;
    std::vector<int32_t> slot_ptr, R4;
    std::vector<cd> amp;
    std::vector<unsigned char> host;
    size_t o_orb = 0, o_amp = 0, o_R = 0, o_ptr = 0, o_ab = 0, o_cell = 0, o_rvec = 0, o_rblk = 0, o_nz = 0, total = 0;
};
}  // namespace

static int model_flatten(int dim_k, int norb, int nspin, const double* orb, const double* onsite, int64_t nhop,
                         const int32_t* hop_i, const int32_t* hop_j, const int32_t* hop_R, const double* hop_amp,
                         FlatModel& F) {
    TBK_REQUIRE(dim_k >= 0 && dim_k <= TBK_MAX_DIM, TBK_EINVAL, "tbk_model_upload: dim_k=%d", dim_k);
    TBK_REQUIRE(nspin == 1 || nspin == 2, TBK_EINVAL, "tbk_model_upload: nspin=%d", nspin);
    TBK_REQUIRE(norb >= 1, TBK_EINVAL, "tbk_model_upload: norb=%d", norb);
    const int n = norb * nspin;
    TBK_REQUIRE(n <= TBK_MAX_NSTA, TBK_EUNSUPPORTED,
                "tbk_model_upload: %d states per k exceeds this build's limit of %d", n,
                TBK_MAX_NSTA);
    TBK_REQUIRE(onsite && (dim_k == 0 || orb), TBK_EINVAL, "tbk_model_upload: null table");
    TBK_REQUIRE(nhop == 0 || (hop_i && hop_j && hop_amp && (dim_k == 0 || hop_R)), TBK_EINVAL,
                "tbk_model_upload: null hopping table");
    const int ns = nspin;
    std::map<TermKey, cd> acc;
    const int last = dim_k > 0 ? dim_k - 1 : 0;
    auto add = [&](int a, int b, const int* R, cd amp) {
        TermKey key;
        key.last = last;
        if (a <= b) {
            key.slot = slot_of(n, a, b);
            for (int d = 0; d < 4; ++d) key.R[d] = d < dim_k ? R[d] : 0;
        } else {
            key.slot = slot_of(n, b, a);
            for (int d = 0; d < 4; ++d) key.R[d] = d < dim_k ? -R[d] : 0;
            amp = cconj(amp);
        }
        auto it = acc.find(key);
        if (it == acc.end())
            acc[key] = amp;
        else
            it->second = cadd(it->second, amp);
    };
    const int zeroR[4] = {0, 0, 0, 0};
    for (int o = 0; o < norb; ++o)
        for (int s = 0; s < ns; ++s)
            for (int t = s; t < ns; ++t) {  // upper part of the hermitian on-site block
                const double* p = onsite + 2 * ((o * ns + s) * ns + t);
                add(o * ns + s, o * ns + t, zeroR, cd{p[0], s == t ? 0.0 : p[1]});
            }
    for (int64_t h = 0; h < nhop; ++h) {
        const int i = hop_i[h], j = hop_j[h];
        TBK_REQUIRE(i >= 0 && i < norb && j >= 0 && j < norb, TBK_EINVAL,
                    "tbk_model_upload: hop %lld has orbital index out of range", (long long)h);
        int R[4] = {0, 0, 0, 0}, mR[4] = {0, 0, 0, 0};
        for (int d = 0; d < dim_k; ++d) {
            R[d] = hop_R[h * dim_k + d];
            mR[d] = -R[d];
        }
        for (int s = 0; s < ns; ++s)
            for (int t = 0; t < ns; ++t) {
                const double* p = hop_amp + 2 * ((h * ns + s) * ns + t);
                const cd amp{p[0], p[1]};
                const int a = i * ns + s, b = j * ns + t;
                // ham[i,s,j,t] += amp E_R ; ham[j,t,i,s] += conj(amp) E_-R   (:919-924).
                // Off the diagonal the second is the hermitian mirror of the first and
                // is implied by the slot storage; on it both land in the same slot.
                if (a == b) {
                    add(a, a, R, amp);
                    add(a, a, mR, cconj(amp));
                } else {
                    add(a, b, R, amp);
                }
            }
    }
    // CSR over slots (std::map iterates in slot-major order)
    const int nslot = n * (n + 1) / 2;
    std::vector<int32_t>& slot_ptr = F.slot_ptr;
    slot_ptr.assign(nslot + 1, 0);
    std::vector<int32_t> slot_ab(nslot);
    std::vector<cd>& amp = F.amp;
    std::vector<int32_t>& R4 = F.R4;
    for (int a = 0; a < n; ++a)
        for (int b = a; b < n; ++b) slot_ab[slot_of(n, a, b)] = a | (b << 16);
    for (auto& kv : acc) {
        if (kv.second.x == 0.0 && kv.second.y == 0.0) continue;
        slot_ptr[kv.first.slot + 1]++;
        amp.push_back(kv.second);
        for (int d = 0; d < 4; ++d) R4.push_back(kv.first.R[d]);
    }
    for (int s = 0; s < nslot; ++s) slot_ptr[s + 1] += slot_ptr[s];
    const int64_t nterm = (int64_t)amp.size();
    // the same ordered terms as cells (slot, p = R_last): a polynomial in z_last per slot
    int pmax = 0;
    for (int64_t t = 0; t < nterm; ++t) pmax = std::max(pmax, std::abs(R4[t * 4 + last]));
    const int npow = 2 * pmax + 1;
    // (only the n <= 4 mesh-row kernel reads it: a 2048-state model would carry 6 M useless entries)
    std::vector<int32_t> cell_ptr(n <= 4 ? (size_t)nslot * npow + 1 : 1, 0);
    if (n <= 4) {
        for (int s = 0; s < nslot; ++s)
            for (int t = slot_ptr[s]; t < slot_ptr[s + 1]; ++t) cell_ptr[(size_t)s * npow + (R4[t * 4 + last] + pmax) + 1]++;
        for (size_t c = 0; c + 1 < cell_ptr.size(); ++c) cell_ptr[c + 1] += cell_ptr[c];
    }
    std::vector<double> orb4((size_t)n * 4, 0.0);
    for (int a = 0; a < n; ++a)
        for (int d = 0; d < dim_k; ++d) orb4[a * 4 + d] = orb[(a / ns) * dim_k + d];

    // the same terms grouped by lattice vector: S(k) = sum_R e^{2 pi i k.R} U_R with U_R dense over the
    // slots -- one phase per R instead of one per term, and coalesced coefficient loads.  Built for
    // the register (n = 5..8) and wavefront (n <= 64) solvers when the blocks are reasonably full
    // (Wannier-type long-ranged tables, dense synthetic models).
    std::vector<int32_t> rvec;
    std::vector<cd> rblock;
    int nR = 0;
    // the non-empty slots {a | b<<16, t0, t1, 0}: sparse models (ribbons, slabs) have ~3n of n^2/2
    std::vector<int32_t> nz;
    for (int s = 0; s < nslot; ++s)
        if (slot_ptr[s + 1] > slot_ptr[s]) {
            nz.push_back(slot_ab[s]);
            nz.push_back(slot_ptr[s]);
            nz.push_back(slot_ptr[s + 1]);
            nz.push_back(0);
        }
    const int nnz = (int)(nz.size() / 4);
    // (n <= 4, round 5: the k-list kernels stage this table in LDS instead of walking the term table with scalar loads)
    if (n <= 64 && nterm > 0) {
        std::map<std::array<int, 4>, int> rid;
        for (int64_t t = 0; t < nterm; ++t) {
            std::array<int, 4> key{R4[t * 4], R4[t * 4 + 1], R4[t * 4 + 2], R4[t * 4 + 3]};
            if (rid.emplace(key, (int)rid.size()).second)
                for (int d = 0; d < 4; ++d) rvec.push_back(key[d]);
        }
        // n <= 16: up to 64 lattice vectors always (the k-list kernels of n <= 4 stage the whole table in LDS; the register kernels
        // of 5..8 and k_e16 assemble from it); 9 <= n <= 16: up to 256 whatever the density (round 5: k_e16 assembles from this table
        // ONLY -- a sparse model of 16 functions with 125 vectors used to fall back to the wavefront Jacobi kernel, 10 x the time per
        // point); else when dense enough -- for 5..8 states, whose kernels have the term walk as well, from an eighth of the table
        // filled (profiles/list_5_8_dense_probe.py / list_5_8_sparse_probe.py: 60 n hoppings over 125 vectors 10-20 % faster on
        // lists and up to 1.65 x on meshes WITH the table, 100 hoppings over ~150 vectors 1.5-2 x faster on lists WITHOUT it)
        const int64_t fill = n <= 8 ? 8 : 4;
        if (rid.size() <= 256 && ((n <= 16 && rid.size() <= 64) || (n >= 9 && n <= 16) || (n >= 5 && (int64_t)rid.size() * nslot <= fill * nterm + 64))) {
            nR = (int)rid.size();
            rblock.assign((size_t)nR * nslot, cd{0.0, 0.0});
            for (int s = 0; s < nslot; ++s)
                for (int t = slot_ptr[s]; t < slot_ptr[s + 1]; ++t) {
                    std::array<int, 4> key{R4[t * 4], R4[t * 4 + 1], R4[t * 4 + 2], R4[t * 4 + 3]};
                    rblock[(size_t)rid[key] * nslot + s] = amp[t];
                }
        } else {
            rvec.clear();
        }
    }

    // one blob: [orb4 | amp | R4 | slot_ptr | slot_ab | cell_ptr | rvec | rblock], 32-byte aligned pieces
    auto al = [](size_t x) { return (x + 31) & ~(size_t)31; };
    const size_t o_orb = 0;
    const size_t o_amp = al(o_orb + orb4.size() * sizeof(double));
    const size_t o_R = al(o_amp + std::max<size_t>(amp.size(), 1) * sizeof(cd));
    const size_t o_ptr = al(o_R + std::max<size_t>(R4.size(), 4) * sizeof(int32_t));
    const size_t o_ab = al(o_ptr + slot_ptr.size() * sizeof(int32_t));
    const size_t o_cell = al(o_ab + slot_ab.size() * sizeof(int32_t));
    const size_t o_rvec = al(o_cell + cell_ptr.size() * sizeof(int32_t));
    const size_t o_rblk = al(o_rvec + std::max<size_t>(rvec.size(), 4) * sizeof(int32_t));
    const size_t o_nz = al(o_rblk + std::max<size_t>(rblock.size(), 1) * sizeof(cd));
    const size_t total = al(o_nz + std::max<size_t>(nz.size(), 4) * sizeof(int32_t));
    std::vector<unsigned char>& host = F.host;
    host.assign(total, 0);
    if (nnz > 0) memcpy(host.data() + o_nz, nz.data(), nz.size() * sizeof(int32_t));
    if (nR > 0) {
        memcpy(host.data() + o_rvec, rvec.data(), rvec.size() * sizeof(int32_t));
        memcpy(host.data() + o_rblk, rblock.data(), rblock.size() * sizeof(cd));
    }
    memcpy(host.data() + o_cell, cell_ptr.data(), cell_ptr.size() * sizeof(int32_t));
    memcpy(host.data() + o_orb, orb4.data(), orb4.size() * sizeof(double));
    if (!amp.empty()) memcpy(host.data() + o_amp, amp.data(), amp.size() * sizeof(cd));
    if (!R4.empty()) memcpy(host.data() + o_R, R4.data(), R4.size() * sizeof(int32_t));
    memcpy(host.data() + o_ptr, slot_ptr.data(), slot_ptr.size() * sizeof(int32_t));
    memcpy(host.data() + o_ab, slot_ab.data(), slot_ab.size() * sizeof(int32_t));

    F.n = n;
    F.nslot = nslot;
    F.pmax = pmax;
    F.nR = nR;
    F.nnz = nnz;
    F.nterm = nterm;
    F.o_orb = o_orb; F.o_amp = o_amp; F.o_R = o_R; F.o_ptr = o_ptr; F.o_ab = o_ab; F.o_cell = o_cell;
    F.o_rvec = o_rvec; F.o_rblk = o_rblk; F.o_nz = o_nz; F.total = total;
    return TBK_OK;
}

// Host-only view of the flatten step (no device needed): the merged slot-major term list exactly as the
// kernels will read it.  Lets the CPU test-suite (and a sanitizer build) check the table construction
// against the reference's _gen_ham without a GPU.  term_cap = capacity of the three output arrays;
// *nterm receives the number of terms (call with term_cap = 0 to size).  info[4] = {pmax, nR, nnz, nslot}.
// ---- continuity of Berry phases (host only): the reference's _one_phase_cont / _array_phases_cont, which a Python loop made the
// largest part of a small berry_phase(contin=True, berry_evals=True) call (250 of 310 us for 41 strings of two bands)
extern "C" int tbk_one_phase_cont(const double* pha, int64_t n, int64_t stride, double clos, double* out, int64_t out_stride) {
    TBK_REQUIRE(pha && out && n >= 0, TBK_EINVAL, "tbk_one_phase_cont: bad argument");
    const double two_pi = 2.0 * M_PI;
    double ref = clos;
    for (int64_t i = 0; i < n; ++i) {
        double v = pha[i * stride];
        if (std::isfinite(v) && std::isfinite(ref))
            while (std::fabs(ref - v) > M_PI) v += ref - v > M_PI ? two_pi : -two_pi;
        out[i * out_stride] = v;
        ref = v;
    }
    return TBK_OK;
}
extern "C" int tbk_array_phases_cont(const double* arr, int64_t n0, int nb, int64_t stride, const double* clos, double* out,
                                     int64_t out_stride) {
    TBK_REQUIRE(arr && clos && out && n0 >= 0 && nb >= 1, TBK_EINVAL, "tbk_array_phases_cont: bad argument");
    const double two_pi = 2.0 * M_PI;
    std::vector<double> ref(clos, clos + nb), cx(nb), cy(nb);
    std::vector<int> free_idx;
    for (int64_t i = 0; i < n0; ++i) {
        const double* cur = arr + i * stride;
        double* o = out + i * out_stride;
        free_idx.resize(nb);
        for (int b = 0; b < nb; ++b) {
            free_idx[b] = b;
            cx[b] = std::cos(cur[b]);
            cy[b] = std::sin(cur[b]);
        }
        for (int j = 0; j < nb; ++j) {
            const double rx = std::cos(ref[j]), ry = std::sin(ref[j]);
            // distance on the unit circle to every entry still free; the LAST index among equal minima
            size_t best = 0;
            double dbest = 0.0;
            for (size_t f = 0; f < free_idx.size(); ++f) {
                const double d = std::hypot(rx - cx[free_idx[f]], ry - cy[free_idx[f]]);
                if (f == 0 || d <= dbest) {
                    best = f;
                    dbest = d;
                }
            }
            const int pick = free_idx[best];
            free_idx.erase(free_idx.begin() + (long)best);
            double v = cur[pick];
            if (std::isfinite(v) && std::isfinite(ref[j]))
                while (std::fabs(ref[j] - v) > M_PI) v += ref[j] - v > M_PI ? two_pi : -two_pi;
            o[j] = v;
        }
        for (int j = 0; j < nb; ++j) ref[j] = o[j];
    }
    return TBK_OK;
}

extern "C" int tbk_model_flatten_host(int dim_k, int norb, int nspin, const double* orb, const double* onsite, int64_t nhop,
                                      const int32_t* hop_i, const int32_t* hop_j, const int32_t* hop_R, const double* hop_amp,
                                      int64_t term_cap, int64_t* nterm, int32_t* term_slot, int32_t* term_R, double* term_amp,
                                      int32_t* info) {
    TBK_REQUIRE(nterm, TBK_EINVAL, "tbk_model_flatten_host: null nterm");
    FlatModel F;
    int rc = model_flatten(dim_k, norb, nspin, orb, onsite, nhop, hop_i, hop_j, hop_R, hop_amp, F);
    if (rc) return rc;
    *nterm = F.nterm;
    if (info) {
        info[0] = F.pmax;
        info[1] = F.nR;
        info[2] = F.nnz;
        info[3] = F.nslot;
    }
    if (term_cap >= F.nterm && term_slot && term_R && term_amp) {
        for (int s = 0; s < F.nslot; ++s)
            for (int t = F.slot_ptr[s]; t < F.slot_ptr[s + 1]; ++t) term_slot[t] = s;
        for (int64_t t = 0; t < F.nterm; ++t) {
            for (int d = 0; d < 4; ++d) term_R[t * 4 + d] = F.R4[t * 4 + d];
            term_amp[2 * t] = F.amp[t].x;
            term_amp[2 * t + 1] = F.amp[t].y;
        }
    }
    return TBK_OK;
}

static int64_t g_model_uploads = 0;

extern "C" int tbk_model_upload(tbk_ctx* ctx, int dim_k, int norb, int nspin, const double* orb,
                                const double* onsite, int64_t nhop, const int32_t* hop_i,
                                const int32_t* hop_j, const int32_t* hop_R, const double* hop_amp,
                                tbk_model** out) {
    TBK_REQUIRE(ctx && out, TBK_EINVAL, "tbk_model_upload: null ctx/out");
    FlatModel F;
    {
        int rc = model_flatten(dim_k, norb, nspin, orb, onsite, nhop, hop_i, hop_j, hop_R, hop_amp, F);
        if (rc) return rc;
    }
    const int n = F.n, nslot = F.nslot, pmax = F.pmax, nR = F.nR, nnz = F.nnz;
    const int64_t nterm = F.nterm;
    const size_t total = F.total, o_orb = F.o_orb, o_amp = F.o_amp, o_R = F.o_R, o_ptr = F.o_ptr, o_ab = F.o_ab,
                 o_cell = F.o_cell, o_rvec = F.o_rvec, o_rblk = F.o_rblk, o_nz = F.o_nz;
    std::vector<unsigned char>& host = F.host;
    tbk_model* m = new (std::nothrow) tbk_model();
    TBK_REQUIRE(m, TBK_ENOMEM, "tbk_model_upload: out of host memory");
    m->ctx = ctx;
    m->dim_k = dim_k;
    m->norb = norb;
    m->nspin = nspin;
    m->nsta = n;
    m->nslot = nslot;
    m->nterm = nterm;
    m->upload_id = ++g_model_uploads;
    TBK_HIP(hipSetDevice(ctx->device));
    m->blob_bytes = total;
    for (size_t i = 0; i < ctx->blob_pool.size(); ++i) {      // a freed model's blob that fits (its owner synchronised before parking it)
        if (ctx->blob_pool[i].bytes >= total && ctx->blob_pool[i].bytes <= 4 * total + 4096) {
            m->blob = ctx->blob_pool[i].p;
            m->blob_bytes = ctx->blob_pool[i].bytes;
            ctx->blob_pool.erase(ctx->blob_pool.begin() + (long)i);
            break;
        }
    }
    if (!m->blob) {
        hipError_t e = hipMalloc(&m->blob, total);
        if (e != hipSuccess) {
            delete m;
            tbk_set_error("tbk_model_upload: hipMalloc(%zu): %s", total, hipGetErrorString(e));
            return TBK_ENOMEM;
        }
    }
    TBK_HIP(hipMemcpyAsync(m->blob, host.data(), total, hipMemcpyHostToDevice, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    unsigned char* base = (unsigned char*)m->blob;
    m->view.dim_k = dim_k;
    m->view.nsta = n;
    m->view.nspin = nspin;
    m->view.nslot = nslot;
    m->view.orb = (const double4*)(base + o_orb);
    m->view.term_amp = (const cd*)(base + o_amp);
    m->view.term_R = (const int4*)(base + o_R);
    m->view.slot_ptr = (const int32_t*)(base + o_ptr);
    m->view.slot_ab = (const int32_t*)(base + o_ab);
    m->view.pmax = pmax;
    m->view.cell_ptr = (const int32_t*)(base + o_cell);
    m->view.nR = nR;
    m->view.rvec = (const int4*)(base + o_rvec);
    m->view.rblock = (const cd*)(base + o_rblk);
    m->view.nnz = nnz;
    m->view.nz = (const int4*)(base + o_nz);
    // 17..32 states: does the spectrum hold paired levels at a generic k?  (ModelView::pairs_hint)
    m->view.pairs_hint = 0;
    if (n >= 17 && n <= 32 && dim_k >= 1) {
        const double kprobe[4] = {0.1234567, 0.2718281, 0.3141592, 0.4142135};
        double ev[32];
        if (tbk_solve_list(m, kprobe, 1, ev, nullptr) == TBK_OK) {
            double tmax = 0.0;
            for (int i = 0; i < n; ++i) tmax = std::max(tmax, std::fabs(ev[i]));
            const double thr = 1e-5 * tmax;       // (the default of TBK_TW16_GAPTOL -- not the knob: tests set that to list every matrix)
            for (int i = 0; i + 1 < n; ++i)
                if (!(ev[i + 1] - ev[i] >= thr)) m->view.pairs_hint = 1;
        }
    }
    *out = m;
    return TBK_OK;
}

extern "C" int tbk_model_free(tbk_model* m) {
    if (!m) return TBK_OK;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    if (m->blob) {
        if (m->blob_bytes <= ((size_t)1 << 20) && m->ctx->blob_pool.size() < 8) m->ctx->blob_pool.push_back(tbk_ctx::Blob{m->blob, m->blob_bytes});
        else hipFree(m->blob);
    }
    delete m;
    return TBK_OK;
}

extern "C" int tbk_model_info(tbk_model* m, int* dim_k, int* nsta, int64_t* nterm) {
    TBK_REQUIRE(m, TBK_EINVAL, "tbk_model_info: null model");
    if (dim_k) *dim_k = m->dim_k;
    if (nsta) *nsta = m->nsta;
    if (nterm) *nterm = m->nterm;
    return TBK_OK;
}

// ------------------------------------------------------------------ wfs
// external (NumPy) layout [point][band][comp]  <->  device layout [band][point][comp]
__global__ __launch_bounds__(256) void k_relayout(cd* __restrict__ dst, const cd* __restrict__ src,
                                                  const int64_t npts, const int nsta, const int ncomp,
                                                  const int to_device) {
    const int64_t total = npts * nsta * ncomp;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        // e enumerates the DESTINATION contiguously
        if (to_device) {
            const int o = (int)(e % ncomp);
            const int64_t r = e / ncomp;
            const int64_t p = r % npts;
            const int b = (int)(r / npts);
            dst[e] = src[(p * nsta + b) * ncomp + o];
        } else {
            const int o = (int)(e % ncomp);
            const int64_t r = e / ncomp;
            const int b = (int)(r % nsta);
            const int64_t p = r / nsta;
            dst[e] = src[((int64_t)b * npts + p) * ncomp + o];
        }
    }
}

__global__ void k_fill_u64(unsigned long long* p, int n, unsigned long long v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

extern "C" int tbk_wfs_create(tbk_ctx* ctx, int dim_arr, const int32_t* mesh, int nsta_arr,
                              int ncomp, tbk_wfs** out) {
    TBK_REQUIRE(ctx && mesh && out, TBK_EINVAL, "tbk_wfs_create: null argument");
    TBK_REQUIRE(dim_arr >= 1 && dim_arr <= TBK_MAX_DIM, TBK_EINVAL, "tbk_wfs_create: dim_arr=%d", dim_arr);
    TBK_REQUIRE(nsta_arr >= 1 && ncomp >= 1, TBK_EINVAL, "tbk_wfs_create: nsta_arr=%d ncomp=%d",
                nsta_arr, ncomp);
    tbk_wfs* w = new (std::nothrow) tbk_wfs();
    TBK_REQUIRE(w, TBK_ENOMEM, "tbk_wfs_create: out of host memory");
    w->ctx = ctx;
    w->view.dim_arr = dim_arr;
    w->view.nsta = nsta_arr;
    w->view.ncomp = ncomp;
    int64_t npts = 1;
    for (int d = 0; d < TBK_MAX_DIM; ++d) w->view.mesh[d] = 1;
    for (int d = 0; d < dim_arr; ++d) {
        if (mesh[d] < 2) {  // pythtb.py:2409
            delete w;
            tbk_set_error("tbk_wfs_create: mesh[%d]=%d, every mesh dimension must be >= 2", d, mesh[d]);
            return TBK_EINVAL;
        }
        w->view.mesh[d] = mesh[d];
        npts *= mesh[d];
    }
    int64_t st = 1;
    for (int d = TBK_MAX_DIM - 1; d >= 0; --d) {  // trailing padded axes have size 1
        w->view.stride[d] = st;
        st *= w->view.mesh[d];
    }
    w->view.npts = npts;
    w->bytes = npts * (int64_t)nsta_arr * ncomp * (int64_t)sizeof(cd);
    TBK_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc((void**)&w->view.data, (size_t)w->bytes);
    if (e != hipSuccess) {
        tbk_set_error("tbk_wfs_create: hipMalloc(%lld bytes) failed: %s", (long long)w->bytes,
                      hipGetErrorString(e));
        delete w;
        return TBK_ENOMEM;
    }
    TBK_HIP(hipMemsetAsync(w->view.data, 0, (size_t)w->bytes, ctx->stream));
    const int ngap = 2 * TBK_GAP_SHARDS * ncomp;   // [parity][shard][ncomp]
    TBK_HIP(hipMalloc((void**)&w->gaps_dev, ngap * sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_fill_u64, dim3((ngap + 255) / 256), dim3(256), 0, ctx->stream, w->gaps_dev, ngap,
                       0x7ff0000000000000ull);
    TBK_HIP(hipMalloc((void**)&w->pbc_dev, TBK_MAX_DIM * (size_t)ncomp * sizeof(cd)));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    *out = w;
    return TBK_OK;
}

extern "C" int tbk_wfs_free(tbk_wfs* w) {
    if (!w) return TBK_OK;
    hipSetDevice(w->ctx->device);
    hipStreamSynchronize(w->ctx->stream);
    if (w->view.data) hipFree(w->view.data);
    if (w->gaps_dev) hipFree(w->gaps_dev);
    if (w->gap_part_dev) hipFree(w->gap_part_dev);
    if (w->pbc_dev) hipFree(w->pbc_dev);
    if (w->tab_dev) hipFree(w->tab_dev);
    tbk_wfs_totals_free(w);
    if (w->flux_cnt_dev) hipFree(w->flux_cnt_dev);
    if (w->flux_plaq_dev) hipFree(w->flux_plaq_dev);
    if (w->flux_partial_dev) hipFree(w->flux_partial_dev);
    delete w;
    return TBK_OK;
}

static int relayout(tbk_wfs* w, cd* dst, const cd* src, int to_device) {
    const WfsView& v = w->view;
    const int64_t total = v.npts * v.nsta * v.ncomp;
    const unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)w->ctx->cus * 16);
    ProfScope ps(w->ctx, "wfs_relayout");
    hipLaunchKernelGGL(k_relayout, dim3(blocks), dim3(256), 0, w->ctx->stream, dst, src, v.npts, v.nsta, v.ncomp,
                       to_device);
    TBK_HIP(hipGetLastError());
    return TBK_OK;
}

extern "C" int tbk_wfs_upload(tbk_wfs* w, const double* host) {
    TBK_REQUIRE(w && host, TBK_EINVAL, "tbk_wfs_upload: null argument");
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    if (w->view.nsta == 1) {  // the two layouts coincide
        TBK_HIP(hipMemcpyAsync(w->view.data, host, (size_t)w->bytes, hipMemcpyHostToDevice, ctx->stream));
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->xfer_h2d_bytes += w->bytes;
        ctx->xfer_h2d_calls += 1;
        return TBK_OK;
    }
    cd* tmp = nullptr;
    hipError_t e = hipMalloc((void**)&tmp, (size_t)w->bytes);
    TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tbk_wfs_upload: staging buffer of %lld bytes: %s", (long long)w->bytes,
                hipGetErrorString(e));
    int rc = TBK_OK;
    if (hipMemcpyAsync(tmp, host, (size_t)w->bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = TBK_EHIP;
    if (!rc) rc = relayout(w, w->view.data, tmp, 1);
    hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    if (rc == TBK_EHIP) tbk_set_error("tbk_wfs_upload: host to device copy failed");
    ctx->xfer_h2d_bytes += w->bytes;
    ctx->xfer_h2d_calls += 1;
    return rc;
}

extern "C" int tbk_wfs_download(tbk_wfs* w, double* host) {
    TBK_REQUIRE(w && host, TBK_EINVAL, "tbk_wfs_download: null argument");
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    if (w->view.nsta == 1) {
        TBK_HIP(hipMemcpyAsync(host, w->view.data, (size_t)w->bytes, hipMemcpyDeviceToHost, ctx->stream));
        TBK_HIP(hipStreamSynchronize(ctx->stream));
        ctx->xfer_d2h_bytes += w->bytes;
        ctx->xfer_d2h_calls += 1;
        return TBK_OK;
    }
    cd* tmp = nullptr;
    hipError_t e = hipMalloc((void**)&tmp, (size_t)w->bytes);
    TBK_REQUIRE(e == hipSuccess, TBK_ENOMEM, "tbk_wfs_download: staging buffer of %lld bytes: %s", (long long)w->bytes,
                hipGetErrorString(e));
    int rc = relayout(w, tmp, w->view.data, 0);
    if (!rc && hipMemcpyAsync(host, tmp, (size_t)w->bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) {
        rc = TBK_EHIP;
        tbk_set_error("tbk_wfs_download: device to host copy failed");
    }
    hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    ctx->xfer_d2h_bytes += w->bytes;
    ctx->xfer_d2h_calls += 1;
    return rc;
}

// ---- single mesh points of a resident array (wf[i,j] reads and writes, pythtb.py:2644-2672,
// without moving the whole array): host layout [point][band][comp] <-> device planes
__global__ __launch_bounds__(256) void k_points_copy(const WfsView v, const int64_t* __restrict__ idx, const int64_t np,
                                                     cd* __restrict__ buf, const int to_device) {
    const int64_t per = (int64_t)v.nsta * v.ncomp;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= np * per) return;
    const int64_t p = e / per;
    const int r = (int)(e - p * per);
    const int b = r / v.ncomp, o = r - b * v.ncomp;
    cd* dev = v.data + ((int64_t)b * v.npts + idx[p]) * v.ncomp + o;
    if (to_device) *dev = buf[e]; else buf[e] = *dev;
}

static int points_copy(tbk_wfs* w, const int64_t* point_index, int64_t np, double* host, int to_device, const char* who) {
    TBK_REQUIRE(w && (np == 0 || (point_index && host)) && np >= 0, TBK_EINVAL, "%s: bad argument", who);
    if (np == 0) return TBK_OK;
    const WfsView& v = w->view;
    for (int64_t i = 0; i < np; ++i)
        TBK_REQUIRE(point_index[i] >= 0 && point_index[i] < v.npts, TBK_EINVAL, "%s: point %lld outside the mesh of %lld points",
                    who, (long long)point_index[i], (long long)v.npts);
    tbk_ctx* ctx = w->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const size_t ib = (((size_t)np * sizeof(int64_t)) + 255) & ~(size_t)255;
    const size_t vb = (size_t)np * v.nsta * v.ncomp * sizeof(cd);
    void* base = nullptr;
    int rc = tbk_ctx_scratch(ctx, 256 + ib + vb, &base);
    if (rc) return rc;
    int64_t* idx_dev = (int64_t*)((unsigned char*)base + 256);
    cd* buf_dev = (cd*)((unsigned char*)base + 256 + ib);
    TBK_HIP(hipMemcpyAsync(idx_dev, point_index, (size_t)np * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    if (to_device) TBK_HIP(hipMemcpyAsync(buf_dev, host, vb, hipMemcpyHostToDevice, ctx->stream));
    const int64_t total = np * v.nsta * v.ncomp;
    hipLaunchKernelGGL(k_points_copy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, v, idx_dev, np, buf_dev,
                       to_device);
    TBK_HIP(hipGetLastError());
    if (!to_device) TBK_HIP(hipMemcpyAsync(host, buf_dev, vb, hipMemcpyDeviceToHost, ctx->stream));
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    (to_device ? ctx->xfer_h2d_bytes : ctx->xfer_d2h_bytes) += (int64_t)vb;
    (to_device ? ctx->xfer_h2d_calls : ctx->xfer_d2h_calls) += 1;
    return TBK_OK;
}

extern "C" int tbk_wfs_download_points(tbk_wfs* w, const int64_t* point_index, int64_t npoints, double* host) {
    return points_copy(w, point_index, npoints, host, 0, "tbk_wfs_download_points");
}

extern "C" int tbk_wfs_upload_points(tbk_wfs* w, const int64_t* point_index, int64_t npoints, const double* host) {
    return points_copy(w, point_index, npoints, const_cast<double*>(host), 1, "tbk_wfs_upload_points");
}

extern "C" int tbk_ctx_transfer_stats(tbk_ctx* c, int64_t* h2d_bytes, int64_t* d2h_bytes, int64_t* h2d_calls, int64_t* d2h_calls,
                                      int reset) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_ctx_transfer_stats: null context");
    if (h2d_bytes) *h2d_bytes = c->xfer_h2d_bytes;
    if (d2h_bytes) *d2h_bytes = c->xfer_d2h_bytes;
    if (h2d_calls) *h2d_calls = c->xfer_h2d_calls;
    if (d2h_calls) *d2h_calls = c->xfer_d2h_calls;
    if (reset) c->xfer_h2d_bytes = c->xfer_d2h_bytes = c->xfer_h2d_calls = c->xfer_d2h_calls = 0;
    return TBK_OK;
}

extern "C" int tbk_ctx_solver_stats(tbk_ctx* c, int64_t* listed_matrices, int reset) {
    TBK_REQUIRE(c, TBK_EINVAL, "tbk_ctx_solver_stats: null context");
    TBK_HIP(hipSetDevice(c->device));
    unsigned long long v = 0;
    TBK_HIP(hipMemcpyAsync(&v, c->flags_dev + TBK_FLAG_LISTED, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    if (reset) TBK_HIP(hipMemsetAsync(c->flags_dev + TBK_FLAG_LISTED, 0, sizeof(v), c->stream));
    TBK_HIP(hipStreamSynchronize(c->stream));
    if (listed_matrices) *listed_matrices = (int64_t)v;
    return TBK_OK;
}

// wf_array.choose_states (pythtb.py:2568-2608) between two resident arrays of the same mesh: band planes are contiguous on
// the device, so a subset of the states is nb device-to-device copies -- nothing crosses PCIe.
extern "C" int tbk_wfs_copy_bands(tbk_wfs* dst, tbk_wfs* src, const int32_t* bands, int nb) {
    TBK_REQUIRE(dst && src && bands && nb >= 1, TBK_EINVAL, "tbk_wfs_copy_bands: bad argument");
    TBK_REQUIRE(dst->ctx == src->ctx, TBK_EINVAL, "tbk_wfs_copy_bands: arrays live on different contexts");
    const WfsView &d = dst->view, &s = src->view;
    TBK_REQUIRE(d.npts == s.npts && d.ncomp == s.ncomp && d.dim_arr == s.dim_arr && d.nsta == nb, TBK_EINVAL,
                "tbk_wfs_copy_bands: destination must hold %d states on the same mesh", nb);
    for (int i = 0; i < d.dim_arr; ++i) TBK_REQUIRE(d.mesh[i] == s.mesh[i], TBK_EINVAL, "tbk_wfs_copy_bands: mesh mismatch");
    tbk_ctx* ctx = src->ctx;
    TBK_HIP(hipSetDevice(ctx->device));
    const size_t plane = (size_t)s.npts * s.ncomp * sizeof(cd);
    for (int b = 0; b < nb; ++b) {
        TBK_REQUIRE(bands[b] >= 0 && bands[b] < s.nsta, TBK_EINVAL, "tbk_wfs_copy_bands: state %d outside the %d stored", bands[b], s.nsta);
        TBK_HIP(hipMemcpyAsync((unsigned char*)d.data + (size_t)b * plane, (const unsigned char*)s.data + (size_t)bands[b] * plane, plane,
                               hipMemcpyDeviceToDevice, ctx->stream));
    }
    TBK_HIP(hipStreamSynchronize(ctx->stream));
    return TBK_OK;
}

extern "C" int tbk_wfs_device_ptr(tbk_wfs* w, void** p, int64_t* bytes) {
    TBK_REQUIRE(w, TBK_EINVAL, "tbk_wfs_device_ptr: null wfs");
    if (p) *p = w->view.data;
    if (bytes) *bytes = w->bytes;
    return TBK_OK;
}
