// Context, device memory, streams, events and kernel timing behind the C-ABI (include/ssdseg.h).
#include <stdarg.h>

#include "common.h"

#include <stdlib.h>
#include <utility>

static thread_local char g_err[512] = "";

void ssdseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ssdseg_hip_fail(hipError_t e, const char* what) {
    ssdseg_set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return -(int)e;
}

// A composite entry point (one that calls other entry points) keeps its own scratch at the front of the workspace and sets
// ctx->ws_reserved to its size for the duration of the nested calls: they get the region BEHIND it.  The composite sizes the
// buffer up front (growing here would free the region it is using), so a nested request that does not fit is an error.
int ssdseg_workspace(ssdseg_ctx* ctx, size_t bytes, void** out) {
    bytes += ctx->ws_reserved;
    if (bytes > ctx->workspace_bytes) {
        if (ctx->ws_reserved != 0) {
            ssdseg_set_error("nested workspace request of %zu bytes exceeds the composite call's reservation (%zu)", bytes, ctx->workspace_bytes);
            return SSDSEG_EINVAL(0);
        }
        // the old workspace may still be in use by queued kernels
        SSDSEG_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->workspace) SSDSEG_HIP(hipFree(ctx->workspace));
        ctx->workspace = nullptr;
        ctx->workspace_bytes = 0;
        size_t want = bytes + bytes / 4;
        SSDSEG_HIP(hipMalloc(&ctx->workspace, want));
        ctx->workspace_bytes = want;
    }
    *out = (char*)ctx->workspace + ctx->ws_reserved;
    return 0;
}

// ---------------------------------------------------------------------------------------------- kernel timing
#include <map>
#include <string>
#include <vector>

struct ssdseg_timing {
    struct Pending {
        const char* kernel;
        hipEvent_t start, stop;
        double bytes, flops, view_bytes;
    };
    struct Stat {
        long long count = 0;
        double ms = 0, bytes = 0, flops = 0, view_bytes = 0;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    std::map<std::string, Stat> stats;
    std::string filter;     // non-empty: only launches of this kernel symbol are bracketed (low-overhead mode)
    bool open = false;      // the last begin() recorded a start event that still needs its stop
};

#include <mutex>
#include <set>
const char* ssdseg_intern(const char* name) {
    static std::mutex mu;
    static std::set<std::string> pool;
    std::lock_guard<std::mutex> lock(mu);
    return pool.insert(name).first->c_str();
}

static hipEvent_t timing_event(ssdseg_timing* t) {
    if (!t->pool.empty()) {
        hipEvent_t e = t->pool.back();
        t->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

static void timing_fold(ssdseg_ctx* ctx) {
    ssdseg_timing* t = ctx->timing;
    if (!t || t->pending.empty()) return;
    // events may sit on either stream (a weight-gradient launch is bracketed on the side stream): wait for both, otherwise a
    // pair that is still "not ready" would be dropped from the statistics without a trace
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (auto& p : t->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            auto& s = t->stats[p.kernel];
            s.count += 1;
            s.ms += ms;
            s.bytes += p.bytes;
            s.flops += p.flops;
            s.view_bytes += p.view_bytes;
        }
        t->pool.push_back(p.start);
        t->pool.push_back(p.stop);
    }
    t->pending.clear();
}

void ssdseg_timing_begin(ssdseg_ctx* ctx, const char* kernel, double bytes, double flops) {
    ssdseg_timing* t = ctx->timing;
    const double view_bytes = ctx->timing_view_bytes;
    ctx->timing_view_bytes = 0.0;
    if (!t) return;
    t->open = false;
    if (!t->filter.empty() && t->filter != kernel) return;
    t->open = true;
    if (t->pending.size() >= 8192) timing_fold(ctx);
    ssdseg_timing::Pending p{kernel, timing_event(t), timing_event(t), bytes, flops, view_bytes};
    (void)hipEventRecord(p.start, ctx->stream);
    t->pending.push_back(p);
}

void ssdseg_timing_end(ssdseg_ctx* ctx) {
    ssdseg_timing* t = ctx->timing;
    if (!t || !t->open || t->pending.empty()) return;
    t->open = false;
    (void)hipEventRecord(t->pending.back().stop, ctx->stream);
}

extern "C" {

int ssdseg_timing_enable(ssdseg_ctx* ctx, int enable) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (enable && !ctx->timing) ctx->timing = new ssdseg_timing();
    if (!enable && ctx->timing) {
        timing_fold(ctx);
        for (auto e : ctx->timing->pool) (void)hipEventDestroy(e);
        delete ctx->timing;
        ctx->timing = nullptr;
    }
    return 0;
}

// restrict the event bracketing to one kernel symbol (NULL or "" = every kernel)
int ssdseg_timing_filter(ssdseg_ctx* ctx, const char* kernel) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ctx->timing != nullptr, 1);
    timing_fold(ctx);
    ctx->timing->filter = kernel ? kernel : "";
    return 0;
}

int ssdseg_timing_reset(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ctx->timing) {
        timing_fold(ctx);
        ctx->timing->stats.clear();
    }
    return 0;
}

// Writes "kernel\tcount\ttotal_ms\talgorithmic_bytes\tflops\tview_bytes\n" lines; returns SSDSEG_EINVAL(3) if buf is too small.
int ssdseg_timing_report(ssdseg_ctx* ctx, char* buf_host, size_t buf_len) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(buf_host != nullptr && buf_len > 0, 2);
    buf_host[0] = 0;
    if (!ctx->timing) return 0;
    timing_fold(ctx);
    size_t used = 0;
    for (auto& kv : ctx->timing->stats) {
        int n = snprintf(buf_host + used, buf_len - used, "%s\t%lld\t%.6f\t%.0f\t%.0f\t%.0f\n", kv.first.c_str(), kv.second.count, kv.second.ms,
                         kv.second.bytes, kv.second.flops, kv.second.view_bytes);
        if (n < 0 || (size_t)n >= buf_len - used) {
            ssdseg_set_error("ssdseg_timing_report: buffer of %zu bytes is too small", buf_len);
            return SSDSEG_EINVAL(3);
        }
        used += (size_t)n;
    }
    return 0;
}

const char* ssdseg_last_error(void) { return g_err; }
int ssdseg_version(void) { return SSDSEG_VERSION; }

int ssdseg_device_count(int* count_host) {
    SSDSEG_ARG(count_host != nullptr, 1);
    SSDSEG_HIP(hipGetDeviceCount(count_host));
    return 0;
}

int ssdseg_ctx_create(int device, void* stream, ssdseg_ctx** out_host) {
    SSDSEG_ARG(out_host != nullptr, 3);
    int count = 0;
    SSDSEG_HIP(hipGetDeviceCount(&count));
    SSDSEG_ARG(device >= 0 && device < count, 1);
    SSDSEG_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SSDSEG_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        ssdseg_set_error("libssdseg_hip is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
        return SSDSEG_EINVAL(1);
    }
    ssdseg_ctx* c = new ssdseg_ctx();
    c->device = device;
    c->workspace = nullptr;
    c->workspace_bytes = 0;
    c->num_cus = prop.multiProcessorCount;
    c->timing = nullptr;
    c->timing_view_bytes = 0.0;
    c->side_stream = nullptr;
    c->ev_fork = c->ev_join = nullptr;
    c->side_workspace = nullptr;
    c->side_workspace_bytes = 0;
    c->side_ok = c->side_on = c->side_pending = false;
    c->ev_mark = nullptr;
    c->mark_set = false;
    c->copy_stream = nullptr;
    c->ev_copy_fork = c->ev_copy_join = nullptr;
    c->comm = nullptr;
    c->comm_rank = 0;
    c->comm_world = 1;
    c->defer = nullptr;
    if (stream) {
        c->stream = (hipStream_t)stream;
        c->owns_stream = false;
    } else {
        // highest priority: the main stream carries the critical path; side-stream work (lowest priority) fills the gaps
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi);
        if (e != hipSuccess) {
            delete c;
            return ssdseg_hip_fail(e, "hipStreamCreateWithFlags");
        }
        c->owns_stream = true;
    }
    // side stream (SSDSEG_NO_SIDE_STREAM=1 keeps everything on the one stream)
    const char* noside = getenv("SSDSEG_NO_SIDE_STREAM");
    if (!(noside && noside[0] == '1')) {
        int plo = 0, phi = 0;
        (void)hipDeviceGetStreamPriorityRange(&plo, &phi);
        // SSDSEG_SIDE_PRIORITY=high|low (A/B; read when the context is made).  Default LOW: the side stream fills what the main
        // stream leaves.
        const char* sp = getenv("SSDSEG_SIDE_PRIORITY");
        const int sprio = (sp != nullptr && !strcmp(sp, "high")) ? phi : plo;
        if (hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, sprio) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) == hipSuccess)
            c->side_ok = true;
    }
    *out_host = c;
    return 0;
}

bool ssdseg_side_begin(ssdseg_ctx* c) {
    if (!c->side_ok || c->side_on) return false;
    if (hipEventRecord(c->ev_fork, c->stream) != hipSuccess) return false;
    if (hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) return false;
    std::swap(c->stream, c->side_stream);
    std::swap(c->workspace, c->side_workspace);
    std::swap(c->workspace_bytes, c->side_workspace_bytes);
    c->side_on = true;
    return true;
}

void ssdseg_side_end(ssdseg_ctx* c) {
    if (!c->side_on) return;
    std::swap(c->stream, c->side_stream);
    std::swap(c->workspace, c->side_workspace);
    std::swap(c->workspace_bytes, c->side_workspace_bytes);
    c->side_on = false;
    c->side_pending = true;
}

int ssdseg_join(ssdseg_ctx* c) {
    if (c->side_on) ssdseg_side_end(c);
    if (c->side_pending) {
        SSDSEG_HIP(hipEventRecord(c->ev_join, c->side_stream));
        SSDSEG_HIP(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        c->side_pending = false;
    }
    // whoever joins is about to read gradients: the deferred column sums of the weight-gradient slabs go out now, as one launch
    return c->defer != nullptr ? ssdseg_colsum_flush(c) : 0;
}

int ssdseg_ctx_destroy(ssdseg_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)ssdseg_join(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)ssdseg_comm_destroy(ctx);
    ssdseg_timing_enable(ctx, 0);
    ssdseg_defer_destroy(ctx);
    if (ctx->workspace) (void)hipFree(ctx->workspace);
    if (ctx->side_workspace) (void)hipFree(ctx->side_workspace);
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_mark) (void)hipEventDestroy(ctx->ev_mark);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
        (void)hipEventDestroy(ctx->ev_copy_fork);
        (void)hipEventDestroy(ctx->ev_copy_join);
    }
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

int ssdseg_ctx_sync(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    int rc = ssdseg_join(ctx);
    if (rc) return rc;
    SSDSEG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ssdseg_ctx_join(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    return ssdseg_join(ctx);
}

// A point on the side stream: everything queued there so far.  ssdseg_ctx_side_wait_mark makes the MAIN stream wait for that point
// only -- not for side-stream work queued after it (a full join would also wait for the weight gradients that were queued behind
// the detection branch).  No-ops without a side stream.
int ssdseg_ctx_side_mark(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ctx->side_on) ssdseg_side_end(ctx);
    ctx->mark_set = false;
    if (!ctx->side_ok) return 0;
    if (ctx->ev_mark == nullptr) SSDSEG_HIP(hipEventCreateWithFlags(&ctx->ev_mark, hipEventDisableTiming));
    SSDSEG_HIP(hipEventRecord(ctx->ev_mark, ctx->side_stream));
    ctx->mark_set = true;
    return 0;
}

int ssdseg_ctx_side_wait_mark(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(!ctx->side_on, 1);
    if (!ctx->mark_set) return 0;
    SSDSEG_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_mark, 0));
    return 0;
}

int ssdseg_ctx_side_enable(ssdseg_ctx* ctx, int enabled) {
    SSDSEG_ARG(ctx != nullptr, 1);
    int rc = ssdseg_join(ctx);
    if (rc) return rc;
    ctx->side_ok = enabled != 0 && ctx->side_stream != nullptr && ctx->ev_fork != nullptr && ctx->ev_join != nullptr;
    return 0;
}

int ssdseg_ctx_side(ssdseg_ctx* ctx, int on) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (on) (void)ssdseg_side_begin(ctx);
    else ssdseg_side_end(ctx);
    return 0;
}

int ssdseg_ctx_reserve(ssdseg_ctx* ctx, size_t workspace_bytes) {
    SSDSEG_ARG(ctx != nullptr, 1);
    void* p;
    return ssdseg_workspace(ctx, workspace_bytes, &p);
}

int ssdseg_ctx_device_name(ssdseg_ctx* ctx, char* buf_host, size_t buf_len) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(buf_host != nullptr && buf_len > 0, 2);
    hipDeviceProp_t prop;
    SSDSEG_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf_host, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

int ssdseg_malloc(ssdseg_ctx* ctx, size_t bytes, void** out_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(out_host != nullptr, 3);
    SSDSEG_HIP(hipSetDevice(ctx->device));
    *out_host = nullptr;
    if (bytes == 0) return 0;
    SSDSEG_HIP(hipMalloc(out_host, bytes));
    return 0;
}

int ssdseg_free(ssdseg_ctx* ctx, void* ptr) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ptr) SSDSEG_HIP(hipFree(ptr));
    return 0;
}

int ssdseg_memcpy_h2d(ssdseg_ctx* ctx, void* dst, const void* src_host, size_t bytes) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (bytes == 0) return 0;
    SSDSEG_ARG(dst != nullptr, 2);
    SSDSEG_ARG(src_host != nullptr, 3);
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }
    SSDSEG_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    SSDSEG_HIP(hipStreamSynchronize(ctx->stream));  // pageable host memory: the caller may reuse it
    return 0;
}

int ssdseg_memcpy_d2h(ssdseg_ctx* ctx, void* dst_host, const void* src, size_t bytes) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (bytes == 0) return 0;
    SSDSEG_ARG(dst_host != nullptr, 2);
    SSDSEG_ARG(src != nullptr, 3);
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }
    SSDSEG_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SSDSEG_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

int ssdseg_memcpy_d2d(ssdseg_ctx* ctx, void* dst, const void* src, size_t bytes) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (bytes == 0) return 0;
    SSDSEG_ARG(dst != nullptr, 2);
    SSDSEG_ARG(src != nullptr, 3);
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }
    SSDSEG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// ---- overlapped uploads: pinned host memory + a copy stream ordered against the main stream by two events
static int copy_stream_of(ssdseg_ctx* ctx) {
    if (ctx->copy_stream != nullptr) return 0;
    SSDSEG_HIP(hipSetDevice(ctx->device));
    SSDSEG_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    SSDSEG_HIP(hipEventCreateWithFlags(&ctx->ev_copy_fork, hipEventDisableTiming));
    SSDSEG_HIP(hipEventCreateWithFlags(&ctx->ev_copy_join, hipEventDisableTiming));
    return 0;
}

int ssdseg_host_alloc(ssdseg_ctx* ctx, size_t bytes, void** out_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(bytes > 0, 2);
    SSDSEG_ARG(out_host != nullptr, 3);
    SSDSEG_HIP(hipSetDevice(ctx->device));
    SSDSEG_HIP(hipHostMalloc(out_host, bytes, hipHostMallocDefault));
    return 0;
}

int ssdseg_host_free(ssdseg_ctx* ctx, void* ptr_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ptr_host) SSDSEG_HIP(hipHostFree(ptr_host));
    return 0;
}

int ssdseg_upload_fence(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    int rc = copy_stream_of(ctx);
    if (rc) return rc;
    SSDSEG_HIP(hipEventRecord(ctx->ev_copy_fork, ctx->stream));
    return 0;
}

int ssdseg_upload_async(ssdseg_ctx* ctx, void* dst, const void* src_host, size_t bytes, int after_fence) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (bytes == 0) return 0;
    SSDSEG_ARG(dst != nullptr, 2);
    SSDSEG_ARG(src_host != nullptr, 3);
    int rc = copy_stream_of(ctx);
    if (rc) return rc;
    // the destination may still be read by main-stream work queued BEFORE the last fence (not by what came after it)
    if (after_fence) SSDSEG_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_copy_fork, 0));
    SSDSEG_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    return 0;
}

int ssdseg_upload_join(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ctx->copy_stream == nullptr) return 0;
    SSDSEG_HIP(hipEventRecord(ctx->ev_copy_join, ctx->copy_stream));
    SSDSEG_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_copy_join, 0));
    return 0;
}

int ssdseg_upload_sync(ssdseg_ctx* ctx) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ctx->copy_stream != nullptr) SSDSEG_HIP(hipStreamSynchronize(ctx->copy_stream));
    return 0;
}

int ssdseg_memset(ssdseg_ctx* ctx, void* dst, int value, size_t bytes) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (bytes == 0) return 0;
    SSDSEG_ARG(dst != nullptr, 2);
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }
    SSDSEG_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
    return 0;
}

int ssdseg_event_create(ssdseg_ctx* ctx, void** out_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(out_host != nullptr, 2);
    hipEvent_t ev;
    SSDSEG_HIP(hipEventCreate(&ev));
    *out_host = (void*)ev;
    return 0;
}

int ssdseg_event_destroy(ssdseg_ctx* ctx, void* ev) {
    SSDSEG_ARG(ctx != nullptr, 1);
    if (ev) SSDSEG_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}

int ssdseg_event_record(ssdseg_ctx* ctx, void* ev) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ev != nullptr, 2);
    { int jrc = ssdseg_join(ctx); if (jrc) return jrc; }
    SSDSEG_HIP(hipEventRecord((hipEvent_t)ev, ctx->stream));
    return 0;
}

int ssdseg_event_elapsed_ms(ssdseg_ctx* ctx, void* ev_start, void* ev_stop, float* ms_host) {
    SSDSEG_ARG(ctx != nullptr, 1);
    SSDSEG_ARG(ev_start != nullptr, 2);
    SSDSEG_ARG(ev_stop != nullptr, 3);
    SSDSEG_ARG(ms_host != nullptr, 4);
    SSDSEG_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    SSDSEG_HIP(hipEventElapsedTime(ms_host, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}

}  // extern "C"
