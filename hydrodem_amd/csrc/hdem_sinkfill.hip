// A1: sink fill (new operator; the reference has none -- SURVEY F2).
//
// Fixed point.  W* = greatest fixed point of
//     T(W)[c] = max(Z[c], min(W[c], min_{8 nbrs n} (W[n] + eps)))
// with the one-cell ring of the raster (and nodata cells and their
// neighbours) pinned.  T is monotone, so ANY schedule of cell updates that
// starts from an upper bound of W* and keeps visiting every cell converges to
// the same bits: every value ever written is T applied to upper bounds, hence
// itself an upper bound, and a state that no update changes is a fixed point
// <= W*.  That licence is what the kernel below uses: tiles are relaxed
// asynchronously, halos may be stale, in-register Gauss-Seidel replaces Jacobi.
//
// Schedule (gfx950):
//   * the raster is cut into 64 x 64 cell tiles; a worklist holds the tiles
//     whose halo changed since they were last relaxed;
//   * one 256-thread workgroup relaxes one tile per visit: every lane owns a
//     4 x 4 cell block -- its Z and W stay in VGPRs for the whole visit (32
//     registers), loaded/stored as 16-byte row pieces (a wave moves 4 x 256 B
//     contiguous runs per instruction);
//   * lanes exchange only block perimeters through LDS, laid out as 16 planes
//     (one per position in the 4 x 4 block) of 18 x 18 blocks (16 + ghost
//     ring), so every halo read is a unit-stride ds_read_b32 across the wave;
//   * a pass = read 20 halo cells, one forward and one backward Gauss-Seidel
//     sweep of the block in registers (v_min3/v_max), publish, vote; passes
//     repeat inside LDS until the tile stops changing (information crosses a
//     tile without touching HBM);
//   * on exit the tile is written back once, and the up-to-8 neighbour tiles
//     whose shared edge changed are appended to the next round's worklist
//     (stamp-deduplicated, no clearing pass).
// HBM traffic per tile visit is the algorithmic 12 B/cell (Z in, W in, W out)
// + the 260-cell halo ring; unchanged tiles skip the write.
#include "hdem_internal.h"

#include <algorithm>

namespace {

constexpr int FT = 64;             // tile edge (cells)
constexpr int NB = FT / 4;         // 4x4 blocks per tile edge
constexpr int GS = NB + 2;         // block grid incl. ghost ring
constexpr int PLANE = GS * GS;
constexpr int NT = 256;
constexpr int PASS_MAX = 64;       // in-LDS passes before the tile re-queues

struct fill_ws {
    int *list[2];
    int *flag;          // last round stamp for which the tile was queued
    int *tile_pinned;   // init only
    int *counts;        // counts[r] = entries in the list consumed by round r
    int ntiles, tiles_x, tiles_y, max_rounds;
};

__device__ __forceinline__ void enqueue(int t, int stamp, int *flag, int *list_next,
                                        int *count_next)
{
    if (atomicExch(&flag[t], stamp) != stamp) {
        int i = atomicAdd(count_next, 1);
        list_next[i] = t;
    }
}

__device__ __forceinline__ float ld_w(const float *w, int H, int W, int y, int x)
{   // halo read: outside the raster and nodata act as +inf walls
    if (y < 0 || y >= H || x < 0 || x >= W) return HDEM_INF;
    float v = w[(size_t)y * W + x];
    return v != v ? HDEM_INF : v;
}

__device__ __forceinline__ float min9(float a, float b, float c, float d, float e,
                                      float f, float g, float h)
{
    float m = fminf(fminf(a, b), c);
    m = fminf(fminf(m, d), e);
    m = fminf(fminf(m, f), g);
    return fminf(m, h);
}

template <bool HAS_EPS>
__global__ __launch_bounds__(NT) void fill_tile_kernel(const float *__restrict__ z,
                                                      float *w, int H, int W, float eps,
                                                      int tiles_x, int tiles_y,
                                                      const int *__restrict__ list_cur,
                                                      const int *__restrict__ count_cur,
                                                      int *list_next, int *count_next,
                                                      int *flag, int stamp)
{
    __shared__ float P[16 * PLANE];
    __shared__ int edge_bits;

    const int tid = threadIdx.x;
    const int bx = tid & (NB - 1), by = tid >> 4;
    const int n_cur = *count_cur;

    for (int li = blockIdx.x; li < n_cur; li += gridDim.x) {
        const int t = list_cur[li];
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int y0 = ty * FT, x0 = tx * FT;
        const int gy = y0 + 4 * by, gx = x0 + 4 * bx;
        if (tid == 0) edge_bits = 0;

        // ---- load the lane's 4x4 block of Z and W -------------------------
        float zz[4][4], e[6][6];
        unsigned nanmask = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = gy + r;
            hdem_f4 vz = {HDEM_INF, HDEM_INF, HDEM_INF, HDEM_INF}, vw = vz;
            if (y < H) {
                const size_t o = (size_t)y * W + gx;
                if (gx + 4 <= W) {
                    vz = hdem_ld4u(z + o);
                    vw = hdem_ld4u(w + o);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (gx + c < W) { vz[c] = z[o + c]; vw[c] = w[o + c]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int x = gx + c;
                float zc = vz[c], wc = vw[c];
                const bool nan = zc != zc;
                const bool ring = y == 0 || y == H - 1 || x == 0 || x == W - 1;
                if (nan) nanmask |= 1u << (r * 4 + c);
                if (wc != wc) wc = HDEM_INF;
                // pinned cells: the update max(zz, min(w, .)) leaves w alone
                zz[r][c] = nan ? HDEM_INF : (ring ? wc : zc);
                e[r + 1][c + 1] = wc;
            }
        }

        // ---- publish the block, fetch the tile's halo ring ----------------
        const int me = (by + 1) * GS + bx + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) P[(r * 4 + c) * PLANE + me] = e[r + 1][c + 1];
        {
            const int i = tid & 63, q = i >> 2, s = i & 3;
            if (tid < 64)        // row above the tile -> bottom row of ghost blocks
                P[(12 + s) * PLANE + 0 * GS + q + 1] = ld_w(w, H, W, y0 - 1, x0 + i);
            else if (tid < 128)  // row below
                P[(0 + s) * PLANE + (NB + 1) * GS + q + 1] = ld_w(w, H, W, y0 + FT, x0 + i);
            else if (tid < 192)  // column left
                P[(s * 4 + 3) * PLANE + (q + 1) * GS + 0] = ld_w(w, H, W, y0 + i, x0 - 1);
            else                 // column right
                P[(s * 4 + 0) * PLANE + (q + 1) * GS + NB + 1] = ld_w(w, H, W, y0 + i, x0 + FT);
            if (tid == 0) P[15 * PLANE + 0] = ld_w(w, H, W, y0 - 1, x0 - 1);
            if (tid == 1) P[12 * PLANE + NB + 1] = ld_w(w, H, W, y0 - 1, x0 + FT);
            if (tid == 2) P[3 * PLANE + (NB + 1) * GS] = ld_w(w, H, W, y0 + FT, x0 - 1);
            if (tid == 3) P[0 * PLANE + (NB + 1) * GS + NB + 1] = ld_w(w, H, W, y0 + FT, x0 + FT);
        }
        __syncthreads();

        // ---- in-LDS passes -------------------------------------------------
        unsigned chmask = 0;
        int pass = 0;
        bool more = true;
        for (; pass < PASS_MAX && more; ++pass) {
            // 20 halo cells: rows above/below (with corners), columns left/right
            e[0][0] = P[15 * PLANE + me - GS - 1];
            e[0][5] = P[12 * PLANE + me - GS + 1];
            e[5][0] = P[3 * PLANE + me + GS - 1];
            e[5][5] = P[0 * PLANE + me + GS + 1];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e[0][c + 1] = P[(12 + c) * PLANE + me - GS];
                e[5][c + 1] = P[(0 + c) * PLANE + me + GS];
                e[c + 1][0] = P[(c * 4 + 3) * PLANE + me - 1];
                e[c + 1][5] = P[(c * 4 + 0) * PLANE + me + 1];
            }
            float old[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) old[r][c] = e[r + 1][c + 1];
            // forward then backward Gauss-Seidel sweep of the block
#pragma unroll
            for (int dir = 0; dir < 2; ++dir)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const int r = dir ? 4 - rr : rr + 1, c = dir ? 4 - cc : cc + 1;
                        float m = min9(e[r - 1][c - 1], e[r - 1][c], e[r - 1][c + 1],
                                       e[r][c - 1], e[r][c + 1], e[r + 1][c - 1],
                                       e[r + 1][c], e[r + 1][c + 1]);
                        if (HAS_EPS) m = m + eps;
                        e[r][c] = fmaxf(zz[r - 1][c - 1], fminf(e[r][c], m));
                    }
            unsigned bits = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (e[r + 1][c + 1] != old[r][c]) {
                        bits |= 1u << (r * 4 + c);
                        P[(r * 4 + c) * PLANE + me] = e[r + 1][c + 1];
                    }
            chmask |= bits;
            more = __syncthreads_or(bits != 0) != 0;
        }
        const bool converged = !more;

        // ---- write back, wake the neighbours whose edge moved --------------
        if (__syncthreads_or(chmask != 0)) {
            if (chmask) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = gy + r;
                    if (y >= H || !((chmask >> (4 * r)) & 0xFu)) continue;
                    const size_t o = (size_t)y * W + gx;
                    hdem_f4 v;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        v[c] = (nanmask >> (r * 4 + c)) & 1u ? __builtin_nanf("") : e[r + 1][c + 1];
                    if (gx + 4 <= W) {
                        hdem_st4u(w + o, v);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (gx + c < W) w[o + c] = v[c];
                    }
                }
                unsigned d = 0;
                if (by == 0 && (chmask & 0x000Fu)) d |= 1u << 1;            // N
                if (by == NB - 1 && (chmask & 0xF000u)) d |= 1u << 6;       // S
                if (bx == 0 && (chmask & 0x1111u)) d |= 1u << 3;            // W
                if (bx == NB - 1 && (chmask & 0x8888u)) d |= 1u << 4;       // E
                if (by == 0 && bx == 0 && (chmask & 0x0001u)) d |= 1u << 0;             // NW
                if (by == 0 && bx == NB - 1 && (chmask & 0x0008u)) d |= 1u << 2;        // NE
                if (by == NB - 1 && bx == 0 && (chmask & 0x1000u)) d |= 1u << 5;        // SW
                if (by == NB - 1 && bx == NB - 1 && (chmask & 0x8000u)) d |= 1u << 7;   // SE
                if (d) atomicOr(&edge_bits, (int)d);
            }
            __syncthreads();
            if (tid < 8 && ((edge_bits >> tid) & 1)) {
                const int dy = tid < 3 ? -1 : (tid < 5 ? 0 : 1);
                const int dx = (tid == 0 || tid == 3 || tid == 5) ? -1
                               : ((tid == 1 || tid == 6) ? 0 : 1);
                const int ny = ty + dy, nx = tx + dx;
                if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x)
                    enqueue(ny * tiles_x + nx, stamp, flag, list_next, count_next);
            }
        }
        if (!converged && tid == 8) enqueue(t, stamp, flag, list_next, count_next);
        __syncthreads();   // LDS is reused by the next tile of this workgroup
    }
}

// W0: pinned cells <- Z (ring, nodata, neighbours of nodata), the rest +inf.
// One lane per cell; the 3x3 nodata probe is served by L1/L2.
__global__ __launch_bounds__(NT) void fill_init_kernel(const float *__restrict__ z,
                                                      float *__restrict__ w, int H, int W,
                                                      int tiles_x, int *tile_pinned,
                                                      int ghost_top, int ghost_bottom)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const float zc = z[i];
    // a ghost row belongs to the neighbouring row block: only its two border
    // cells are pinned; the rest waits at +inf for the first halo exchange
    const bool ghost = (ghost_top && y == 0) || (ghost_bottom && y == H - 1);
    if (ghost && x != 0 && x != W - 1) { w[i] = zc != zc ? zc : HDEM_INF; return; }
    bool pin = y == 0 || y == H - 1 || x == 0 || x == W - 1 || zc != zc;
    if (!pin) {
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const float zn = z[(size_t)(y + dy) * W + (x + dx)];
                pin |= zn != zn;   // (a nodata cell in a ghost row pins too: it is nodata for its owner as well)
            }
    }
    w[i] = pin ? zc : HDEM_INF;
    if (pin && zc == zc) tile_pinned[(y / FT) * tiles_x + x / FT] = 1;
}

// Round-0 worklist for INIT: tiles that hold a pinned cell, and their 8
// neighbours (a pinned cell never changes, so it cannot wake a neighbour).
__global__ __launch_bounds__(NT) void fill_seed_kernel(const int *__restrict__ tile_pinned,
                                                      int tiles_x, int tiles_y, int *flag,
                                                      int *list0, int *count0, int stamp)
{
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    bool act = false;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int ny = ty + dy, nx = tx + dx;
            if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x)
                act |= tile_pinned[ny * tiles_x + nx] != 0;
        }
    if (act) enqueue(t, stamp, flag, list0, count0);
}

// Round-0 worklist for WARM starts: all tiles, or the tile rows next to a
// ghost row that a halo exchange just replaced.
__global__ __launch_bounds__(NT) void fill_seed_rows_kernel(int tiles_x, int tiles_y, int H,
                                                           int mode, int *flag, int *list0,
                                                           int *count0, int stamp)
{
    const int t = blockIdx.x * NT + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    const int ty = t / tiles_x;
    // a replaced ghost row can only move the row next to it: row 1 / row H-2
    // (the ghost row itself may sit alone in the last tile row)
    const bool act = mode == 0 || ((mode & HDEM_FILL_ACT_TOP) && ty == 1 / FT) ||
                     ((mode & HDEM_FILL_ACT_BOTTOM) && ty == max(H - 2, 0) / FT);
    if (act) enqueue(t, stamp, flag, list0, count0);
}

int ensure_ws(hdem_ctx *ctx, int H, int W, int max_rounds, fill_ws *ws)
{
    ws->tiles_x = (W + FT - 1) / FT;
    ws->tiles_y = (H + FT - 1) / FT;
    ws->ntiles = ws->tiles_x * ws->tiles_y;
    ws->max_rounds = max_rounds;
    const size_t ints = (size_t)ws->ntiles * 4 + (size_t)max_rounds + 16;
    const size_t bytes = ints * sizeof(int);
    if (ctx->fill_ws_bytes < bytes) {
        if (ctx->fill_ws) {
            HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            HDEM_HIP_CHECK(hipFree(ctx->fill_ws));
            ctx->fill_ws = nullptr;
            ctx->fill_ws_bytes = 0;
        }
        HDEM_HIP_CHECK(hipMalloc(&ctx->fill_ws, bytes));
        ctx->fill_ws_bytes = bytes;
    }
    if (ctx->host_counts_len < (size_t)max_rounds + 16) {
        if (ctx->host_counts) HDEM_HIP_CHECK(hipHostFree(ctx->host_counts));
        ctx->host_counts = nullptr;
        HDEM_HIP_CHECK(hipHostMalloc((void **)&ctx->host_counts,
                                     ((size_t)max_rounds + 16) * sizeof(int32_t)));
        ctx->host_counts_len = (size_t)max_rounds + 16;
    }
    int *base = (int *)ctx->fill_ws;
    ws->list[0] = base;
    ws->list[1] = base + ws->ntiles;
    ws->flag = base + 2 * (size_t)ws->ntiles;
    ws->tile_pinned = base + 3 * (size_t)ws->ntiles;
    ws->counts = base + 4 * (size_t)ws->ntiles;
    // flags, pinned map and counters start at zero; stamps are >= 1
    HDEM_HIP_CHECK(hipMemsetAsync(ws->flag, 0,
                                  (2 * (size_t)ws->ntiles + max_rounds + 16) * sizeof(int),
                                  ctx->stream));
    return HDEM_OK;
}

}  // namespace

extern "C" int hdem_sinkfill_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, float eps,
                                     int max_rounds, int flags, float *w,
                                     hdem_fill_stats *stats)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, w, H, W)) return rc;
    HDEM_REQUIRE(z != w, HDEM_ERR_BAD_ARG, "sink fill cannot run in place");
    HDEM_REQUIRE(eps >= 0.0f && eps == eps, HDEM_ERR_BAD_ARG, "eps must be >= 0, got %g",
                 (double)eps);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    if (max_rounds <= 0) max_rounds = 1 << 16;
    const int K = 8;   // rounds enqueued between convergence checks
    max_rounds = (max_rounds + K - 1) / K * K;

    fill_ws ws;
    if (int rc = ensure_ws(ctx, H, W, max_rounds, &ws)) return rc;
    hipStream_t st = ctx->stream;
    const unsigned tile_blocks = (unsigned)((ws.ntiles + NT - 1) / NT);

    if (!(flags & HDEM_FILL_WARM)) {
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FILL_INIT, (int64_t)H * W);
            const size_t n = (size_t)H * W;
            hipLaunchKernelGGL(fill_init_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT),
                               0, st, z, w, H, W, ws.tiles_x, ws.tile_pinned,
                               flags & HDEM_FILL_GHOST_TOP, flags & HDEM_FILL_GHOST_BOTTOM);
        }
        hipLaunchKernelGGL(fill_seed_kernel, dim3(tile_blocks), dim3(NT), 0, st,
                           ws.tile_pinned, ws.tiles_x, ws.tiles_y, ws.flag, ws.list[0],
                           ws.counts, 1);
    } else {
        hipLaunchKernelGGL(fill_seed_rows_kernel, dim3(tile_blocks), dim3(NT), 0, st,
                           ws.tiles_x, ws.tiles_y, H,
                           flags & (HDEM_FILL_ACT_TOP | HDEM_FILL_ACT_BOTTOM), ws.flag,
                           ws.list[0], ws.counts, 1);
    }
    HDEM_HIP_CHECK(hipGetLastError());

    const int grid = std::max(1, std::min(ws.ntiles, ctx->num_cus * 8));
    int round = 0, converged = 0;
    int64_t visits = 0;
    while (round < max_rounds && !converged) {
        for (int k = 0; k < K; ++k) {
            const int r = round + k;
            hdem_scoped_timer tm(ctx, HDEM_K_FILL_TILE, 0);
            if (eps != 0.0f)
                hipLaunchKernelGGL(fill_tile_kernel<true>, dim3(grid), dim3(NT), 0, st, z, w, H,
                                   W, eps, ws.tiles_x, ws.tiles_y, ws.list[r & 1],
                                   ws.counts + r, ws.list[(r + 1) & 1], ws.counts + r + 1,
                                   ws.flag, r + 2);
            else
                hipLaunchKernelGGL(fill_tile_kernel<false>, dim3(grid), dim3(NT), 0, st, z, w,
                                   H, W, eps, ws.tiles_x, ws.tiles_y, ws.list[r & 1],
                                   ws.counts + r, ws.list[(r + 1) & 1], ws.counts + r + 1,
                                   ws.flag, r + 2);
        }
        HDEM_HIP_CHECK(hipGetLastError());
        HDEM_HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ws.counts + round,
                                      (K + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
        HDEM_HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < K; ++k) {
            if (ctx->host_counts[k] == 0) { converged = 1; break; }
            visits += ctx->host_counts[k];
            ++round;
        }
        if (!converged && ctx->host_counts[K] == 0) converged = 1;
    }
    ctx->stats[HDEM_K_FILL_TILE].units += visits * FT * FT;
    if (stats) {
        stats->rounds = round;
        stats->converged = converged;
        stats->tile_visits = visits;
        stats->tiles = ws.ntiles;
        stats->tile_h = FT;
        stats->tile_w = FT;
        stats->scans = 0;
        stats->reserved = 0;
    }
    if (!converged) {
        hdem_set_error("sink fill did not converge in %d rounds", max_rounds);
        return HDEM_ERR_NOT_CONVERGED;
    }
    return HDEM_OK;
}

extern "C" int hdem_sinkfill_f32(hdem_ctx *ctx, const float *z, int H, int W, float eps,
                                 int max_rounds, float *w, hdem_fill_stats *stats)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, w, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)H * W * sizeof(float);
    hdem_dbuf dz, dw;
    if (int rc = dz.alloc(bytes)) return rc;
    if (int rc = dw.alloc(bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dz.p, z, bytes)) return rc;
    int rc = hdem_sinkfill_f32_dev(ctx, (const float *)dz.p, H, W, eps, max_rounds,
                                   HDEM_FILL_INIT, (float *)dw.p, stats);
    if (rc != HDEM_OK && rc != HDEM_ERR_NOT_CONVERGED) return rc;
    if (int rc2 = hdem_memcpy_d2h(ctx, w, dw.p, bytes)) return rc2;
    return rc;
}
