// msq_ctx.hip -- context, error text, device-memory helpers, log-factorial table.
#include "msq_internal.h"

#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

#include <cmath>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";

void msq_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int msq_fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* msq_last_error(void) { return g_err; }

static void timing_release(msq_ctx* ctx);

extern "C" int msq_ctx_create(int device, msq_ctx** out)
{
    if (!out) return msq_fail(MSQ_ERR_ARG, "msq_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return msq_fail(MSQ_ERR_NODEVICE, "msq_ctx_create: no HIP device (this layer has no CPU fallback)");
    if (device < 0 || device >= n) return msq_fail(MSQ_ERR_ARG, "msq_ctx_create: device %d of %d", device, n);
    MSQ_HIP(hipSetDevice(device));
    msq_ctx* c = new msq_ctx;
    c->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->n_cu = prop.multiProcessorCount;
    hipError_t e = hipMalloc(&c->d_err, 64 * sizeof(uint32_t));  // the error word + the small accumulators
    if (e == hipSuccess) e = hipMemset(c->d_err, 0, 64 * sizeof(uint32_t));
    c->d_small = c->d_err + 16;
    if (e != hipSuccess) {
        delete c;
        return msq_fail(MSQ_ERR_HIP, "msq_ctx_create: %s", hipGetErrorString(e));
    }
    if (int rc = msq_noalloc_init(c)) {  // this GPU's address of the built-in scratch (msq_pileup_noalloc)
        (void)hipFree(c->d_err);
        delete c;
        return rc;
    }
    c->no_merge = std::getenv("MSQ_NO_MERGE") != nullptr;
    c->no_haps = std::getenv("MSQ_NO_HAPS") != nullptr;
    if (const char* v = std::getenv("MSQ_BAM_FIRST_SHARE")) c->bam_first_share_pct = (uint64_t)std::min(99ll, std::max(1ll, strtoll(v, nullptr, 10)));
    if (const char* v = std::getenv("MSQ_BAM_UPLOAD_THREADS")) c->bam_upload_threads = (uint64_t)std::min(16ll, std::max(0ll, strtoll(v, nullptr, 10)));
    if (const char* v = std::getenv("MSQ_BAM_PIECES")) c->bam_pieces = (uint64_t)std::max(1ll, strtoll(v, nullptr, 10));
    if (const char* v = std::getenv("MSQ_CARRIER_MIN_BATCHES")) c->carrier_min_batches = (int64_t)strtoll(v, nullptr, 10);
    *out = c;
    return MSQ_OK;
}

extern "C" int msq_debug_tune(msq_ctx* ctx, int key, long long value)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_debug_tune: ctx is NULL");
    switch (key) {
    case MSQ_TUNE_NO_MERGE: ctx->no_merge = value != 0; return MSQ_OK;
    case MSQ_TUNE_CARRIER_MIN_BATCHES: ctx->carrier_min_batches = (int64_t)value; return MSQ_OK;
    case MSQ_TUNE_NO_HAPS: ctx->no_haps = value != 0; return MSQ_OK;
    case MSQ_TUNE_BAM_PIECES: ctx->bam_pieces = value < 1 ? 1 : (uint64_t)value; return MSQ_OK;
    case MSQ_TUNE_BAM_PIECE_MIN_BYTES: ctx->bam_piece_min_bytes = value < 0 ? 0 : (uint64_t)value; return MSQ_OK;
    case MSQ_TUNE_BAM_SEGMENT: ctx->bam_segment_bytes = value < 64 ? 64 : (uint64_t)value; return MSQ_OK;
    default: return msq_fail(MSQ_ERR_ARG, "msq_debug_tune: unknown key %d", key);
    }
}

int msq_ensure_dynamic_lds(int device, const void* kernel, size_t lds)
{
    if (lds <= 48 * 1024) return MSQ_OK;
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> set;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = set[std::make_pair(device, kernel)];
    if (lds > have) {
        MSQ_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        have = lds;
    }
    return MSQ_OK;
}

int msq_validate_reads(const msq_packed_reads* d_reads, const msq_qv_thresholds* thr, const char* who)
{
    if (!d_reads->ref_start || !d_reads->group_off || !d_reads->bop) return msq_fail(MSQ_ERR_ARG, "%s: NULL read arrays", who);
    if (((uintptr_t)d_reads->bop & 15u) != 0) return msq_fail(MSQ_ERR_ARG, "%s: bop must be 16-byte aligned", who);
    if ((((uintptr_t)d_reads->qual | (uintptr_t)d_reads->delqv | (uintptr_t)d_reads->subqv | (uintptr_t)d_reads->insqv) & 15u) != 0)
        return msq_fail(MSQ_ERR_ARG, "%s: QV planes must be 16-byte aligned", who);
    for (int k = 0; k < 4; ++k) {
        const int32_t t = (&thr->qual)[k];
        if (t < -1 || t > 255) return msq_fail(MSQ_ERR_ARG, "%s: QV threshold %d out of range", who, t);
    }
    return MSQ_OK;
}

extern "C" int msq_ctx_destroy(msq_ctx* ctx)
{
    if (!ctx) return MSQ_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)msq_comm_destroy(ctx);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->d_lfact) (void)hipFree(ctx->d_lfact);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_dev) (void)hipFree(ctx->d_dev);
    for (int k = 0; k < 2; ++k) {
        if (ctx->h_stage[k]) (void)hipHostFree(ctx->h_stage[k]);
        if (ctx->stage_done[k]) (void)hipEventDestroy(ctx->stage_done[k]);
    }
    msq_bgzf_drop_staged(ctx);
    if (ctx->d_bam_tables) (void)hipFree(ctx->d_bam_tables);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->pinned_ring) (void)hipHostFree(ctx->pinned_ring);
    for (hipStream_t st : ctx->side_stream)
        if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->side_event)
        if (ev) (void)hipEventDestroy(ev);
    timing_release(ctx);
    delete ctx;
    return MSQ_OK;
}

extern "C" int msq_ctx_set_stream(msq_ctx* ctx, void* hip_stream)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_ctx_set_stream: ctx is NULL");
    ctx->stream = (hipStream_t)hip_stream;
    return MSQ_OK;
}

extern "C" int msq_ctx_sync(msq_ctx* ctx)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_ctx_sync: ctx is NULL");
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_dev_malloc(msq_ctx* ctx, size_t bytes, void** d_out)
{
    if (!ctx || !d_out) return msq_fail(MSQ_ERR_ARG, "msq_dev_malloc: NULL argument");
    MSQ_HIP(hipSetDevice(ctx->device));
    *d_out = nullptr;
    if (bytes == 0) bytes = 4;
    hipError_t e = hipMalloc(d_out, bytes);
    if (e != hipSuccess) return msq_fail(MSQ_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return MSQ_OK;
}
extern "C" int msq_dev_free(msq_ctx* ctx, void* d_ptr)
{
    if (!ctx) return msq_fail(MSQ_ERR_ARG, "msq_dev_free: ctx is NULL");
    if (d_ptr) {
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        MSQ_HIP(hipFree(d_ptr));
    }
    return MSQ_OK;
}
extern "C" int msq_dev_memset(msq_ctx* ctx, void* d_ptr, int value, size_t bytes)
{
    if (!ctx || (!d_ptr && bytes)) return msq_fail(MSQ_ERR_ARG, "msq_dev_memset: NULL argument");
    if (bytes) MSQ_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return MSQ_OK;
}
extern "C" int msq_memcpy_h2d(msq_ctx* ctx, void* d_dst, const void* h_src, size_t bytes)
{
    if (!ctx || ((!d_dst || !h_src) && bytes)) return msq_fail(MSQ_ERR_ARG, "msq_memcpy_h2d: NULL argument");
    if (bytes) MSQ_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return MSQ_OK;
}
extern "C" int msq_memcpy_d2h(msq_ctx* ctx, void* h_dst, const void* d_src, size_t bytes)
{
    if (!ctx || ((!h_dst || !d_src) && bytes)) return msq_fail(MSQ_ERR_ARG, "msq_memcpy_d2h: NULL argument");
    if (bytes) MSQ_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    return MSQ_OK;
}

extern "C" int msq_memcpy_d2d(msq_ctx* ctx, void* d_dst, const void* d_src, size_t bytes)
{
    if (!ctx || ((!d_dst || !d_src) && bytes)) return msq_fail(MSQ_ERR_ARG, "msq_memcpy_d2d: NULL argument");
    if (bytes) MSQ_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return MSQ_OK;
}

static void timing_release(msq_ctx* ctx)
{
    for (auto e : ctx->ev_begin) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_end) (void)hipEventDestroy(e);
    for (auto e : ctx->ev_path_end) (void)hipEventDestroy(e);
    ctx->ev_begin.clear();
    ctx->ev_end.clear();
    ctx->ev_path_end.clear();
    ctx->ev_used = 0;
    ctx->ev_calls = 0;
    ctx->ev_open = ctx->ev_path_open = false;
}

extern "C" int msq_timing_enable_strided(msq_ctx* ctx, int32_t max_records, int32_t stride)
{
    if (!ctx || max_records < 0 || stride < 1) return msq_fail(MSQ_ERR_ARG, "msq_timing_enable: bad argument");
    MSQ_HIP(hipSetDevice(ctx->device));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    timing_release(ctx);
    ctx->ev_stride = stride;
    for (int i = 0; i < max_records; ++i) {
        hipEvent_t a, b, c;
        MSQ_HIP(hipEventCreate(&a));
        MSQ_HIP(hipEventCreate(&b));
        MSQ_HIP(hipEventCreate(&c));
        ctx->ev_begin.push_back(a);
        ctx->ev_end.push_back(b);
        ctx->ev_path_end.push_back(c);
    }
    return MSQ_OK;
}
extern "C" int msq_timing_enable(msq_ctx* ctx, int32_t max_records) { return msq_timing_enable_strided(ctx, max_records, 1); }

// kernel_ms_out: begin .. end of the dominant kernel; path_ms_out: begin .. end of msq_call_cells (either may be NULL)
static int timing_read(msq_ctx* ctx, float* kernel_ms_out, float* path_ms_out, int32_t capacity, int32_t* n_out)
{
    float* ms_out = kernel_ms_out ? kernel_ms_out : path_ms_out;
    if (!ctx || !n_out || (capacity > 0 && !ms_out)) return msq_fail(MSQ_ERR_ARG, "msq_timing_read: bad argument");
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    int n = ctx->ev_used < capacity ? ctx->ev_used : capacity;
    for (int i = 0; i < n; ++i)
        MSQ_HIP(hipEventElapsedTime(&ms_out[i], ctx->ev_begin[i], kernel_ms_out ? ctx->ev_end[i] : ctx->ev_path_end[i]));
    *n_out = n;
    if (!path_ms_out) ctx->ev_used = 0;
    return MSQ_OK;
}

extern "C" int msq_timing_read(msq_ctx* ctx, float* ms_out, int32_t capacity, int32_t* n_out)
{
    return timing_read(ctx, ms_out, nullptr, capacity, n_out);
}
extern "C" int msq_timing_read_path(msq_ctx* ctx, float* ms_out, int32_t capacity, int32_t* n_out)
{
    if (ctx && ctx->ev_path_open) return msq_fail(MSQ_ERR_ARG, "msq_timing_read_path: the last timed pileup has no msq_call_cells behind it");
    return timing_read(ctx, nullptr, ms_out, capacity, n_out);
}

void msq_timing_mark_begin(msq_ctx* ctx)
{
    ctx->ev_open = false;
    if (ctx->ev_used < (int)ctx->ev_begin.size() && (ctx->ev_calls++ % ctx->ev_stride) == 0) {
        (void)hipEventRecord(ctx->ev_begin[ctx->ev_used], ctx->stream);
        ctx->ev_open = true;
    }
}
void msq_timing_mark_end(msq_ctx* ctx)
{
    if (ctx->ev_open) {
        (void)hipEventRecord(ctx->ev_end[ctx->ev_used], ctx->stream);
        ++ctx->ev_used;
        ctx->ev_open = false;
        ctx->ev_path_open = true;
    }
}
void msq_timing_mark_path_end(msq_ctx* ctx)  // behind msq_call_cells' last kernel
{
    if (ctx->ev_path_open) {
        (void)hipEventRecord(ctx->ev_path_end[ctx->ev_used - 1], ctx->stream);
        ctx->ev_path_open = false;
    }
}

// Diagnostics: device memory the context has allocated so far and still holds (scratch, the step's deviation matrix,
// the log-factorial table) -- a test of msq_pileup_noalloc asks that a call leaves it as it was.
extern "C" uint64_t msq_debug_ctx_device_bytes(const msq_ctx* ctx)
{
    if (!ctx) return 0;
    return (uint64_t)ctx->scratch_bytes + (uint64_t)ctx->dev_words * sizeof(uint32_t) +
           (ctx->lfact_n >= 0 ? ((uint64_t)ctx->lfact_n + 1) * sizeof(double) : 0);
}

int msq_scratch(msq_ctx* ctx, size_t bytes, void** out)
{
    if (bytes > ctx->scratch_bytes) {
        MSQ_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->d_scratch) MSQ_HIP(hipFree(ctx->d_scratch));
        ctx->d_scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + (bytes >> 2) + 4096;
        hipError_t e = hipMalloc(&ctx->d_scratch, want);
        if (e != hipSuccess) return msq_fail(MSQ_ERR_NOMEM, "scratch hipMalloc(%zu): %s", want, hipGetErrorString(e));
        ctx->scratch_bytes = want;
    }
    *out = ctx->d_scratch;
    return MSQ_OK;
}

// QvThresholds::QvThresholds(), src/QvThresholds.cpp:45-56: getenv + std::stoi, SUBQV defaults to 42.
extern "C" int msq_qv_thresholds_from_env(msq_qv_thresholds* out)
{
    if (!out) return msq_fail(MSQ_ERR_ARG, "msq_qv_thresholds_from_env: out is NULL");
    auto get = [](const char* name, int dflt, int32_t* dst) -> int {
        const char* v = std::getenv(name);
        if (!v) {
            *dst = dflt;
            return MSQ_OK;
        }
        try {
            // boost::optional<uint8_t> = std::stoi(...): the int narrows to uint8_t
            *dst = (int32_t)(uint8_t)std::stoi(std::string(v));
        } catch (const std::exception& e) {
            return msq_fail(MSQ_ERR_ARG, "%s: std::stoi failed on '%s'", name, v);
        }
        return MSQ_OK;
    };
    int rc;
    if ((rc = get("DELQV", -1, &out->del)) != MSQ_OK) return rc;
    if ((rc = get("SUBQV", 42, &out->sub)) != MSQ_OK) return rc;
    if ((rc = get("INSQV", -1, &out->ins)) != MSQ_OK) return rc;
    if ((rc = get("QUALQV", -1, &out->qual)) != MSQ_OK) return rc;
    return MSQ_OK;
}

// ErrorEstimates(chemistry), src/ErrorEstimates.cpp:44-49 (same numbers for every chemistry).
extern "C" int msq_error_model_default(msq_error_model* out)
{
    if (!out) return msq_fail(MSQ_ERR_ARG, "msq_error_model_default: out is NULL");
    out->match = 0.9956844883;
    out->substitution = 0.0005244257 / 3.0;
    out->deletion = 0.003791086;
    return MSQ_OK;
}
// ErrorEstimates(substitutionRate, deletionRate), src/ErrorEstimates.cpp:61-67.
extern "C" int msq_error_model_from_rates(double substitution_rate, double deletion_rate, msq_error_model* out)
{
    if (!out) return msq_fail(MSQ_ERR_ARG, "msq_error_model_from_rates: out is NULL");
    out->match = 1 - substitution_rate - deletion_rate;
    out->substitution = substitution_rate / 3.0;
    out->deletion = deletion_rate;
    return MSQ_OK;
}

// factln table, src/Fisher.cpp:115-125: 0 for n <= 1, else lgamma(n + 1.0) from the host libm (the very
// function the reference calls, so the values are bit-identical to the reference's on the same host).
extern "C" int msq_lfact_reserve(msq_ctx* ctx, int64_t max_n)
{
    if (!ctx || max_n < 0) return msq_fail(MSQ_ERR_ARG, "msq_lfact_reserve: bad argument");
    if (max_n <= ctx->lfact_n) return MSQ_OK;
    if (max_n < 4096) max_n = 4096;
    MSQ_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)max_n + 1;
    std::vector<double> tab(n);
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 16) nt = 16;
    if (n < 1u << 16) nt = 1;
    auto fill = [&tab](size_t lo, size_t hi) {
        int sign;
        for (size_t i = lo; i < hi; ++i) tab[i] = i <= 1 ? 0.0 : lgamma_r((double)((int)i + 1.0), &sign);
    };
    if (nt == 1) {
        fill(0, n);
    } else {
        std::vector<std::thread> th;
        const size_t per = (n + nt - 1) / nt;
        for (unsigned t = 0; t < nt; ++t) {
            const size_t lo = t * per, hi = lo + per < n ? lo + per : n;
            if (lo < hi) th.emplace_back(fill, lo, hi);
        }
        for (auto& t : th) t.join();
    }
    double* d = nullptr;
    hipError_t e = hipMalloc(&d, n * sizeof(double));
    if (e != hipSuccess) return msq_fail(MSQ_ERR_NOMEM, "lfact hipMalloc(%zu): %s", n * sizeof(double), hipGetErrorString(e));
    MSQ_HIP(hipStreamSynchronize(ctx->stream));
    MSQ_HIP(hipMemcpy(d, tab.data(), n * sizeof(double), hipMemcpyHostToDevice));
    if (ctx->d_lfact) MSQ_HIP(hipFree(ctx->d_lfact));
    ctx->d_lfact = d;
    ctx->lfact_n = max_n;
    return MSQ_OK;
}
