// Context, memory, error and profiling plumbing of libhydrodem_hip.so.
#include "hdem_internal.h"

#include <sys/mman.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";

void hdem_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *hdem_last_error(void) { return g_err; }
extern "C" int hdem_version(void) { return 100; }

extern "C" int hdem_device_count(int *count)
{
    HDEM_REQUIRE(count, HDEM_ERR_BAD_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        hdem_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return HDEM_ERR_NO_DEVICE;
    }
    *count = n;
    return HDEM_OK;
}

extern "C" int hdem_init(int device, hdem_ctx **out)
{
    HDEM_REQUIRE(out, HDEM_ERR_BAD_ARG, "ctx out-pointer is null");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        hdem_set_error("no HIP device visible");
        return HDEM_ERR_NO_DEVICE;
    }
    HDEM_REQUIRE(device >= 0 && device < n, HDEM_ERR_BAD_ARG,
                 "device %d out of range [0, %d)", device, n);
    HDEM_HIP_CHECK(hipSetDevice(device));
    hdem_ctx *ctx = new (std::nothrow) hdem_ctx();
    HDEM_REQUIRE(ctx, HDEM_ERR_OOM, "out of host memory");
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess)
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        hdem_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return HDEM_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return HDEM_OK;
}

extern "C" int hdem_shutdown(hdem_ctx *ctx)
{
    if (!ctx) return HDEM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &t : ctx->pending) {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto &b : ctx->pool_free) {
        (void)hipFree(b.p);
        (void)hipEventDestroy(b.ready);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    for (auto ev : ctx->pool_events) (void)hipEventDestroy(ev);
    hdem_fourier_release(ctx);
    for (void *b : ctx->fill_ws)
        if (b) (void)hipFree(b);
    if (ctx->coarse_buf) (void)hipFree(ctx->coarse_buf);
    for (void *b : ctx->hub_buf)
        if (b) (void)hipFree(b);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->host_counts) (void)hipHostFree(ctx->host_counts);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return HDEM_OK;
}

void *hdem_arena(hdem_ctx *ctx, size_t bytes)
{
    if (ctx->arena_bytes >= bytes) return ctx->arena;
    if (ctx->arena) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
    if (hdem_raw_alloc(ctx, bytes, &ctx->arena) != HDEM_OK) {
        ctx->arena = nullptr;
        return nullptr;
    }
    ctx->arena_bytes = bytes;
    return ctx->arena;
}

extern "C" int hdem_set_stream(hdem_ctx *ctx, void *hip_stream)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return HDEM_OK;
}

extern "C" int hdem_synchronize(hdem_ctx *ctx)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return HDEM_OK;
}

namespace {

void pool_drop(hdem_ctx *ctx, size_t keep_bytes)
{
    // oldest first; hipFree waits for the device, which also covers `ready`
    size_t i = 0;
    for (; i < ctx->pool_free.size() && ctx->pool_bytes > keep_bytes; ++i) {
        hdem_cached_block &b = ctx->pool_free[i];
        (void)hipFree(b.p);
        ctx->pool_events.push_back(b.ready);
        ctx->pool_bytes -= b.bytes;
    }
    ctx->pool_free.erase(ctx->pool_free.begin(), ctx->pool_free.begin() + (long)i);
}

}  // namespace

// Every device allocation the library makes for itself (sink-fill workspace, coarse rasters,
// the arena of the chains, rocFFT work buffers) comes through here: a request the device
// cannot serve empties this context's block cache and is tried once more, so memory parked
// by hdem_free never stands between the library and an allocation it needs.
int hdem_raw_alloc(hdem_ctx *ctx, size_t bytes, void **dptr)
{
    *dptr = nullptr;
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> guard(ctx->pool_lock);
            pool_drop(ctx, 0);
        }
        e = hipMalloc(dptr, bytes ? bytes : 1);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *dptr = nullptr;
        hdem_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? HDEM_ERR_OOM : HDEM_ERR_HIP;
    }
    return HDEM_OK;
}

extern "C" int hdem_trim(hdem_ctx *ctx, size_t *released)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> guard(ctx->pool_lock);
        bytes = ctx->pool_bytes;
        pool_drop(ctx, 0);
    }
    // the scratch the context keeps between calls goes too; it grows back on demand
    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->arena) {
        bytes += ctx->arena_bytes;
        (void)hipFree(ctx->arena);
        ctx->arena = nullptr;
        ctx->arena_bytes = 0;
    }
    if (ctx->coarse_buf) {
        bytes += ctx->coarse_bytes;
        (void)hipFree(ctx->coarse_buf);
        ctx->coarse_buf = nullptr;
        ctx->coarse_bytes = 0;
    }
    for (int k = 0; k < 2; ++k)
        if (ctx->hub_buf[k]) {
            bytes += ctx->hub_bytes[k];
            (void)hipFree(ctx->hub_buf[k]);
            ctx->hub_buf[k] = nullptr;
            ctx->hub_bytes[k] = 0;
        }
    if (released) *released = bytes;
    return HDEM_OK;
}

extern "C" int hdem_malloc(hdem_ctx *ctx, size_t bytes, void **dptr)
{
    HDEM_REQUIRE(ctx && dptr, HDEM_ERR_BAD_ARG, "null argument");
    *dptr = nullptr;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    if (!bytes) bytes = 1;
    std::lock_guard<std::mutex> guard(ctx->pool_lock);
    // smallest cached block that holds the request without wasting more than an eighth
    int best = -1;
    for (int i = 0; i < (int)ctx->pool_free.size(); ++i) {
        const size_t have = ctx->pool_free[i].bytes;
        if (have >= bytes && have - bytes <= bytes / 8 &&
            (best < 0 || have < ctx->pool_free[best].bytes))
            best = i;
    }
    if (best >= 0) {
        const hdem_cached_block b = ctx->pool_free[best];
        ctx->pool_free.erase(ctx->pool_free.begin() + best);
        ctx->pool_bytes -= b.bytes;
        // whatever still runs on the block was enqueued before `ready`
        HDEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, b.ready, 0));
        ctx->pool_events.push_back(b.ready);
        ctx->pool_live[b.p] = {b.bytes, ctx->stream};
        *dptr = b.p;
        return HDEM_OK;
    }
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory && !ctx->pool_free.empty()) {
        (void)hipGetLastError();
        pool_drop(ctx, 0);
        e = hipMalloc(dptr, bytes);
    }
    if (e != hipSuccess) {
        *dptr = nullptr;
        hdem_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? HDEM_ERR_OOM : HDEM_ERR_HIP;
    }
    ctx->pool_live[*dptr] = {bytes, ctx->stream};
    return HDEM_OK;
}

extern "C" int hdem_free(hdem_ctx *ctx, void *dptr)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (!dptr) return HDEM_OK;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> guard(ctx->pool_lock);
    const auto it = ctx->pool_live.find(dptr);
    const size_t bytes = it == ctx->pool_live.end() ? 0 : it->second.bytes;
    // The cache orders a block's next use behind an event on the context's stream.  That
    // covers everything this context enqueued on it -- provided the stream is still the one
    // the block was handed out under.  If hdem_set_stream has changed it since, work on the
    // old stream may still use the block: wait for the whole device first, as hipFree would.
    const bool stream_changed = it != ctx->pool_live.end() && it->second.stream != ctx->stream;
    if (it != ctx->pool_live.end()) ctx->pool_live.erase(it);
    if (stream_changed) HDEM_HIP_CHECK(hipDeviceSynchronize());
    if (!ctx->pool_cap) {
        // a quarter of the device, at most 64 GiB (HDEM_POOL_MIB: another figure, 0 = keep nothing)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        ctx->pool_cap = std::min<size_t>(total_b / 4, (size_t)64 << 30);
        if (const char *e = getenv("HDEM_POOL_MIB")) ctx->pool_cap = (size_t)atoll(e) << 20;
        if (!ctx->pool_cap) ctx->pool_cap = 1;          // "looked up"; nothing fits
    }
    if (bytes && bytes <= ctx->pool_cap) {
        hipEvent_t ev = nullptr;
        if (!ctx->pool_events.empty()) {
            ev = ctx->pool_events.back();
            ctx->pool_events.pop_back();
        } else if (hipEventCreateWithFlags(&ev, hipEventDefault) != hipSuccess) {
            ev = nullptr;
        }
        if (ev && hipEventRecord(ev, ctx->stream) == hipSuccess) {
            if (ctx->pool_bytes + bytes > ctx->pool_cap) pool_drop(ctx, ctx->pool_cap - bytes);
            ctx->pool_free.push_back({dptr, bytes, ev});
            ctx->pool_bytes += bytes;
            return HDEM_OK;
        }
        if (ev) ctx->pool_events.push_back(ev);
    }
    // not one of ours, larger than the cache, or no event to be had: the plain way
    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HDEM_HIP_CHECK(hipFree(dptr));
    return HDEM_OK;
}

// A device-to-host copy into memory the process has never touched (a fresh np.empty) runs
// at 10-17 GB/s instead of 56: the copy engine's staging path takes the page faults one by
// one.  Taking them first, on many threads, costs a few milliseconds per GiB -- the whole
// destination range is about to be overwritten; one byte per page is rewritten with itself.
static void prefault_host(void *dst, size_t bytes)
{
    constexpr size_t PAGE = 4096, MIN_BYTES = (size_t)32 << 20;
    if (bytes < MIN_BYTES) return;
    unsigned nthreads = std::thread::hardware_concurrency();
    nthreads = nthreads ? (nthreads > 16 ? 16 : nthreads) : 4;
    char *base = static_cast<char *>(dst);
    {
        // 2 MiB pages where the host hands them out on request: 512 x fewer faults here, and
        // the unmapping of the array when its owner drops it (50 ms per GiB in 4 KiB pages)
        // shrinks with them.  Whole pages inside the range only; a refusal changes nothing.
        const uintptr_t lo = ((uintptr_t)base + PAGE - 1) & ~(uintptr_t)(PAGE - 1);
        const uintptr_t hi = ((uintptr_t)base + bytes) & ~(uintptr_t)(PAGE - 1);
        if (hi > lo) (void)madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    }
    const size_t per = (bytes / nthreads + PAGE - 1) / PAGE * PAGE;
    std::vector<std::thread> pool;
    try {
        for (unsigned t = 0; t < nthreads; ++t) {
            const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
            if (lo >= hi) break;
            pool.emplace_back([base, lo, hi] {
                // (each byte written back as it was: the destination may be an array the caller
                // still reads if the copy fails, e.g. an in-place repair of its input)
                auto touch = [](volatile char *p) { const char c = *p; *p = c; };
                for (size_t o = lo; o < hi; o += PAGE) touch(base + o);
                touch(base + hi - 1);
            });
        }
    } catch (...) {
        // no more threads to be had: the copy takes the remaining faults itself
    }
    for (auto &th : pool) th.join();
}

static int copy_sync(hdem_ctx *ctx, void *dst, const void *src, size_t bytes,
                     hipMemcpyKind kind)
{
    HDEM_REQUIRE(ctx && (bytes == 0 || (dst && src)), HDEM_ERR_BAD_ARG,
                 "null argument");
    if (!bytes) return HDEM_OK;
    if (kind == hipMemcpyDeviceToHost) prefault_host(dst, bytes);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    if (kind != hipMemcpyDeviceToDevice)
        HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return HDEM_OK;
}

extern "C" int hdem_memcpy_h2d(hdem_ctx *c, void *d, const void *s, size_t n)
{ return copy_sync(c, d, s, n, hipMemcpyHostToDevice); }
extern "C" int hdem_memcpy_d2h(hdem_ctx *c, void *d, const void *s, size_t n)
{ return copy_sync(c, d, s, n, hipMemcpyDeviceToHost); }
extern "C" int hdem_memcpy_d2d(hdem_ctx *c, void *d, const void *s, size_t n)
{ return copy_sync(c, d, s, n, hipMemcpyDeviceToDevice); }

// Pinned host memory and stream-ordered copies: the raster I/O seam.  A
// caller that reads a raster band by band (GDAL ReadAsArray of windows) reads straight
// into a pinned buffer, queues copy -> kernels -> copy on this context's stream and goes
// on to the next band on another context; hdem_synchronize() is the only wait.
extern "C" int hdem_host_alloc(hdem_ctx *ctx, size_t bytes, void **hptr)
{
    HDEM_REQUIRE(ctx && hptr, HDEM_ERR_BAD_ARG, "null argument");
    *hptr = nullptr;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipHostMalloc(hptr, bytes ? bytes : 1, hipHostMallocDefault));
    return HDEM_OK;
}

extern "C" int hdem_host_free(hdem_ctx *ctx, void *hptr)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (!hptr) return HDEM_OK;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    HDEM_HIP_CHECK(hipHostFree(hptr));
    return HDEM_OK;
}

static int copy_async(hdem_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
{
    HDEM_REQUIRE(ctx && (bytes == 0 || (dst && src)), HDEM_ERR_BAD_ARG, "null argument");
    if (!bytes) return HDEM_OK;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, ctx->stream));
    return HDEM_OK;
}

// A byte fill on the context's stream (HydroConditioning.apply_batch makes its canvas of
// nodata with it: 0xff bytes are a NaN).
extern "C" int hdem_memset_dev(hdem_ctx *ctx, void *dptr, int byte, size_t bytes)
{
    HDEM_REQUIRE(ctx && (bytes == 0 || dptr), HDEM_ERR_BAD_ARG, "null argument");
    if (!bytes) return HDEM_OK;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipMemsetAsync(dptr, byte, bytes, ctx->stream));
    return HDEM_OK;
}

extern "C" int hdem_memcpy_h2d_async(hdem_ctx *c, void *d, const void *s, size_t n)
{ return copy_async(c, d, s, n, hipMemcpyHostToDevice); }
extern "C" int hdem_memcpy_d2h_async(hdem_ctx *c, void *d, const void *s, size_t n)
{ return copy_async(c, d, s, n, hipMemcpyDeviceToHost); }

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
static hipEvent_t take_event(hdem_ctx *ctx)
{
    if (!ctx->event_pool.empty()) {
        hipEvent_t ev = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return ev;
    }
    hipEvent_t ev = nullptr;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return ev;
}

hdem_scoped_timer::hdem_scoped_timer(hdem_ctx *c, int kernel_id, int64_t units)
    : ctx(c), on(c && c->profiling)
{
    if (!on) return;
    t.kernel_id = kernel_id;
    t.units = units;
    t.start = take_event(ctx);
    t.stop = take_event(ctx);
    if (!t.start || !t.stop) { on = false; return; }
    (void)hipEventRecord(t.start, ctx->stream);
}

hdem_scoped_timer::~hdem_scoped_timer()
{
    if (!on) return;
    (void)hipEventRecord(t.stop, ctx->stream);
    ctx->pending.push_back(t);
}

int hdem_fold_profile(hdem_ctx *ctx)
{
    if (ctx->pending.empty()) return HDEM_OK;
    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &t : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            hdem_kernel_stat &s = ctx->stats[t.kernel_id];
            s.launches += 1;
            s.ms += ms;
            s.units += t.units;
        }
        ctx->event_pool.push_back(t.start);
        ctx->event_pool.push_back(t.stop);
    }
    ctx->pending.clear();
    return HDEM_OK;
}

extern "C" int hdem_profile_enable(hdem_ctx *ctx, int on)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    ctx->profiling = on != 0;
    return HDEM_OK;
}

extern "C" int hdem_profile_reset(hdem_ctx *ctx)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    int rc = hdem_fold_profile(ctx);
    std::memset(ctx->stats, 0, sizeof(ctx->stats));
    return rc;
}

extern "C" int hdem_profile_get(hdem_ctx *ctx, int kernel_id, hdem_kernel_stat *out)
{
    HDEM_REQUIRE(ctx && out, HDEM_ERR_BAD_ARG, "null argument");
    HDEM_REQUIRE(kernel_id >= 0 && kernel_id < HDEM_K_COUNT, HDEM_ERR_BAD_ARG,
                 "kernel id %d out of range", kernel_id);
    int rc = hdem_fold_profile(ctx);
    *out = ctx->stats[kernel_id];
    return rc;
}
