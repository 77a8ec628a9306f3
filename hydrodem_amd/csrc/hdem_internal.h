// Internal declarations shared by the .hip translation units of
// libhydrodem_hip.so.  Nothing here is part of the C ABI (include/hydrodem_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/hydrodem_hip.h"

void hdem_set_error(const char *fmt, ...);

#define HDEM_HIP_CHECK(expr)                                                  \
    do {                                                                      \
        hipError_t e_ = (expr);                                               \
        if (e_ != hipSuccess) {                                               \
            hdem_set_error("%s failed: %s (%s:%d)", #expr,                    \
                           hipGetErrorString(e_), __FILE__, __LINE__);        \
            return e_ == hipErrorOutOfMemory ? HDEM_ERR_OOM : HDEM_ERR_HIP;   \
        }                                                                     \
    } while (0)

#define HDEM_REQUIRE(cond, code, ...)                                         \
    do {                                                                      \
        if (!(cond)) {                                                        \
            hdem_set_error(__VA_ARGS__);                                      \
            return (code);                                                    \
        }                                                                     \
    } while (0)

struct hdem_timed_launch {
    hipEvent_t start, stop;
    int kernel_id;
    int64_t units;
};

struct hdem_fourier_state;            // hdem_fourier.hip: rocFFT plans of one raster shape

// A block of device memory that hdem_free has taken back: `ready` was recorded on the stream
// the context ran on at that moment, the next owner's stream waits for it.
struct hdem_cached_block {
    void *p;
    size_t bytes;
    hipEvent_t ready;
};

struct hdem_live_block {
    size_t bytes;
    hipStream_t stream;               // the context's stream when the block was handed out
};

struct hdem_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;      // stream in use (own or caller's)
    int num_cus = 256;
    bool profiling = false;
    std::vector<hdem_timed_launch> pending;   // events not yet folded in
    std::vector<hipEvent_t> event_pool;
    hdem_kernel_stat stats[HDEM_K_COUNT] = {};
    // sink-fill workspace (grown on demand, reused across calls)
    void *fill_ws[3] = {nullptr, nullptr, nullptr};   // [hub_depth]
    size_t fill_ws_bytes[3] = {0, 0, 0};
    int fill_slice_us = 0;             // 0: run the asynchronous phase to convergence
    int fill_last_h = 0, fill_last_w = 0;   // problem the worklist in fill_ws belongs to
    const void *fill_last_z = nullptr, *fill_last_out = nullptr;
    bool fill_resumable = false;       // state words hold a consistent asynchronous worklist
    bool fill_quiescent = false;       // ... and the last call left nothing to do
    int *fill_seam_words = nullptr;    // device: [0] queued after a deferred call, [1]/[2] ghost row changed
    bool fill_stats_carry = false;     // deferred calls' counters are still in the workspace
    // coarse start of the sink fill: a caller's filled coarse raster for the next INIT call
    // (hdem_set_fill_coarse_start), and the buffer of the library's own coarse pre-solve
    const float *start_coarse = nullptr;
    const int32_t *start_row_map = nullptr;
    int start_cw = 0, start_shift = 0;
    void *coarse_buf = nullptr;
    bool in_coarse_presolve = false;   // the fill in progress is that pre-solve
    uint8_t *fill_d8 = nullptr;        // D8 raster the certifying pass of the next fill writes
    bool fill_d8_done = false;         // ... and whether it did
    bool fill_d8_ring_done = false;    // ... the raster ring included (the certifying stream)
    size_t coarse_bytes = 0;
    void *hub_buf[2] = {nullptr, nullptr};   // hub start of the sink fill: rim lines, hub raster
    size_t hub_bytes[2] = {0, 0};            // ([1]: of the hub raster's own fill)
    int hub_depth = 0;                       // 0: a caller's fill, 1: of a hub raster, 2: of its raster
    // ... prepared for a row-block partition (hdem_fill_hub_prepare_dev), and the levels the
    // partition worked out for the next INIT fill of that block
    const float *hub_prep_z = nullptr;
    float *hub_prep_w = nullptr;
    int hub_prep_h = 0, hub_prep_cols = 0, hub_prep_flags = 0;
    const float *hub_levels_given = nullptr;
    void *arena = nullptr;             // scratch of the multi-kernel chains, grown on demand
    size_t arena_bytes = 0;
    hdem_fourier_state *fourier = nullptr;
    int32_t *host_counts = nullptr;    // pinned: convergence counters
    size_t host_counts_len = 0;
    // hdem_malloc / hdem_free: blocks handed back are kept (up to pool_cap bytes) and handed
    // out again for a request of about their size -- a chain of device-resident operators
    // allocates its intermediates at every call, and hipMalloc + hipFree of a 1 GiB raster
    // cost 0.3 ms + a device-wide wait
    std::mutex pool_lock;
    std::unordered_map<void *, hdem_live_block> pool_live;   // blocks handed out
    std::vector<hdem_cached_block> pool_free;          // oldest first
    std::vector<hipEvent_t> pool_events;               // spare `ready` events
    size_t pool_bytes = 0, pool_cap = 0;
};

// Brackets one kernel launch with events when profiling is on.
struct hdem_scoped_timer {
    hdem_ctx *ctx;
    bool on;
    hdem_timed_launch t;
    hdem_scoped_timer(hdem_ctx *c, int kernel_id, int64_t units);
    ~hdem_scoped_timer();
};

int hdem_fold_profile(hdem_ctx *ctx);   // sync + accumulate pending events
void hdem_fourier_release(hdem_ctx *ctx);
// A device buffer of at least `bytes` that stays with the context (one user at a time:
// the chains carve it up themselves).  nullptr + error set on failure.
void *hdem_arena(hdem_ctx *ctx, size_t bytes);
// hipMalloc for the library's own long-lived buffers: retried once after emptying the
// context's block cache.  HDEM_OK / HDEM_ERR_OOM / HDEM_ERR_HIP, error text set.
int hdem_raw_alloc(hdem_ctx *ctx, size_t bytes, void **dptr);
// hdem_stencil.hip: streaming certification of a filled surface (+ its D8 codes)
int hdem_certify_d8_launch(hdem_ctx *ctx, const float *z, const float *w, int H, int W, float eps,
                           uint8_t *d8, int *flag);

static inline int hdem_check_raster(const void *in, const void *out, int H,
                                    int W)
{
    if (!in || !out) {
        hdem_set_error("null raster pointer");
        return HDEM_ERR_BAD_ARG;
    }
    if (H <= 0 || W <= 0) {
        hdem_set_error("raster dimensions must be positive, got %d x %d", H, W);
        return HDEM_ERR_BAD_ARG;
    }
    return HDEM_OK;
}

// Host-pointer wrapper helper: a device buffer from the context's block cache that hands
// itself back (hdem_malloc / hdem_free).
struct hdem_dbuf {
    hdem_ctx *ctx = nullptr;
    void *p = nullptr;
    ~hdem_dbuf() { if (p) (void)hdem_free(ctx, p); }
    int alloc(hdem_ctx *c, size_t bytes)
    {
        ctx = c;
        return hdem_malloc(c, bytes, &p);
    }
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
#define HDEM_INF (__builtin_huge_valf())

// 16-byte vector of 4 floats with only 4-byte alignment assumed, so that a
// row whose pitch is not a multiple of 16 B can still be moved with one
// global_load_dwordx4 / global_store_dwordx4 per lane.
typedef float hdem_f4 __attribute__((ext_vector_type(4)));
typedef float hdem_f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ hdem_f4 hdem_ld4u(const float *p)
{
    return *reinterpret_cast<const hdem_f4u *>(p);
}
__device__ __forceinline__ void hdem_st4u(float *p, hdem_f4 v)
{
    *reinterpret_cast<hdem_f4u *>(p) = v;
}
