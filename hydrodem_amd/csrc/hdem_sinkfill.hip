// A1: sink fill (new operator; the reference has none -- SURVEY F2).
//
// Fixed point.  W* = greatest fixed point of
//     T(W)[c] = max(Z[c], min(W[c], min_{8 nbrs n} (W[n] + eps)))
// with the one-cell ring of the raster (and nodata cells and their
// neighbours) pinned.  T is monotone, so ANY schedule of cell updates that
// starts from an upper bound of W* and keeps visiting every cell converges to
// the same bits: every value ever written is T applied to upper bounds, hence
// itself an upper bound, and a state that no update changes is a fixed point
// <= W*.  That licence is what this file uses: tiles are relaxed
// asynchronously, halos may be stale, directional Gauss-Seidel scans replace
// Jacobi sweeps -- and a final round-synchronous pass certifies the fixed point.
//
// The visit (gfx950):
//   * the interior of the raster is cut into 62 x 62 cell tiles; a tile is
//     relaxed inside its 64 x 64 window (tile + one-cell halo ring, the ring
//     pinned for the duration of the visit);
//   * ONE WAVE relaxes one tile: lane c holds column c of the window, 64 rows
//     of Z and 64 of W, in VGPRs (a wave-level load moves one 256-byte row
//     piece; all 128 are in flight together).  A north->south scan walks the
//     rows in registers: w[r] <- med3(z[r], w[r], min3 of row r-1 at lanes
//     c-1,c,c+1) -- two v_min_f32_dpp (the wave shifts ride on the mins) and a
//     v_med3: 3 VALU per row, no memory.  South->north runs interleaved with it.  The window is then transposed
//     through a 64 x 65 LDS buffer (conflict-free both ways) so that lane = row,
//     and the same code scans west->east and east->west.  The four scans cover
//     all eight neighbours; a Jacobi check (one full T step, a third of the
//     cost) decides whether another round of scans is needed;
//   * the wave writes the tile back once (62 rows x 248 B) and wakes only those
//     of its 8 neighbour tiles that one of its changed edge cells can still
//     lower: a halo cell (as held in registers) above the new edge value AND
//     above its own terrain.
//
// The schedule:
//   * every tile has a fixed owner workgroup (tile t -> workgroup t % G, slot
//     t / G; one state word per tile, [owner][slot]), so there are no queues and
//     no contended counters -- a single hot atomic costs ~12 ns per arrival and
//     was measured to dominate everything else here;
//   * asynchronous driver (the work horse): one persistent launch; a workgroup
//     keeps relaxing whichever of its queued tiles was woken first; waking a
//     neighbour is one compare-and-swap on its state word;
//   * round-synchronous driver: one launch per round, state word = stamp of the
//     round the tile is due in, waking is a plain store.  Used behind the
//     asynchronous launch to certify (or finish) the fixed point, and on its own
//     with HDEM_FILL_SYNC_ONLY.
// HBM traffic per tile visit: Z in + W in (2 x 64 x 256 B) + W out when changed;
// the algorithmic figure is 12 B per tile cell.
#include "hdem_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int FT = 62;             // tile edge (cells)
constexpr int WN = 64;             // window edge = tile + halo ring = wave width
constexpr int TS = WN + 1;         // LDS row stride (floats): conflict-free transposes
constexpr int NT = 64;             // one wave per workgroup
constexpr int ITER_MAX = 8;        // 4-scan iterations before the tile re-queues
constexpr int INIT_NT = 256;
constexpr int INIT_ROWS = 4;        // output rows per lane of fill_init_kernel
constexpr int KEY_NONE = 0x7fffffff;   // "no key": above every float_key()
constexpr int SEED_KEYS = 1 << 20;     // queue keys below this: seeds (flood order)
constexpr int AUX_SC1 = 16;            // buffer-instruction cache policy: sc1 (agent scope)
constexpr int COARSE_SHIFT = 4;    // own coarse start: 16 x 16 blocks ...
constexpr int HUB_EDGE = 4 * WN;        // hub start, floats per tile: N row, S row (lane = column),
                                        // W column, E column (lane = row) of d on the tile's rim
constexpr float HUB_BIG = 3.0e38f;      // a wall of the hub raster (finite: not nodata)
constexpr int HUB_MIN_TILES = 64;       // hub start from this many tiles on (512^2 cells: measured)
constexpr int HUB_MIN_TILES_NESTED = 36;   // ... of a hub raster's own fill     // hub start (below) from this many tiles on (~1000^2 cells)
constexpr int COARSE_MIN_CELLS = 6000 * 6000;   // ... from this raster size on (below, the
                                   // two extra launches cost what they save)
constexpr int PEND_SHARDS = 64;    // one per lane of the polling wave
constexpr int PEND_STRIDE = 32;    // ints: one 128-byte line per shard
#ifdef HDEM_VISIT_PROF
constexpr int STAT_WORDS = 16;
#else
constexpr int STAT_WORDS = 9;      // per workgroup: visits, iterations, unchanged, re-queued,
                                   // busy ticks, idle ticks (100 MHz, async driver),
                                   // visits made by the round driver, visits of flat tiles,
                                   // of those: unchanged
#endif
constexpr int STAT_FLAT = 7;
constexpr int STAT_FLAT_SAME = 8;
#ifdef HDEM_VISIT_PROF
constexpr int STAT_PROF = 9;
#endif
constexpr int HEAD_INTS = 32;          // head of the workspace: [0] budget / error flag,
                                       // [1] residency census, [2] soft-budget flag,
                                       // [3] partial residency, [4] certifying pass's flag
constexpr int FLAT_NONE = 0x7f7f7f7f;  // "not flat" / "unknown" (what the 0x7f memset leaves)
enum { ST_IDLE = 0, ST_QUEUED = 1, ST_RUNNING = 2, ST_DIRTY = 3, ST_ROUND0 = 16 };

struct fill_ws {
    int *tile_key;      // INIT: lowest pinned elevation next to the tile
    int *state;         // [G][S] async: ST_*; round driver: stamp of the round the tile is due in
    int *prio;          // [G][S] lowest key offered while queued
    int *pend;          // sharded count of non-idle tiles (async)
    int *any;           // any[r] != 0: round r has work
    int *error;
    unsigned long long *stats;   // [G][STAT_WORDS], written only by the owner
    int *flat;          // [ntiles] float bits: level of a tile whose interior is one level
    int *zmax;          // [ntiles] float bits: highest terrain of that interior (FLAT_NONE: -)
    int *applied;       // [ntiles] hub start: 1 once the tile has had its first visit
    int ntiles, tiles_x, tiles_y, max_rounds, G, S;
};

// lane i <- lane i-1 / lane i+1 across the whole wave (gfx9 wave_shr / wave_shl).
// bound_ctrl: the lane without a source (0 resp. 63) reads 0 -- harmless where
// those two lanes are the pinned window ring and ignore their candidate.
__device__ __forceinline__ float lane_prev(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_next(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}

// acc |= lanes where a != b.  Written as asm so that the compare result is
// folded into ONE scalar pair at once (left to itself hipcc keeps 62 compare
// masks per scan alive and spills ~550 SGPRs).
__device__ __forceinline__ void or_changed(unsigned long long &acc, float a, float b)
{
    asm volatile("v_cmp_neq_f32 vcc, %1, %2\n\ts_or_b64 %0, %0, vcc"
                 : "+s"(acc) : "v"(a), "v"(b) : "vcc", "scc");   // s_or_b64 writes SCC
}
// acc |= lanes where a < b
__device__ __forceinline__ void or_less(unsigned long long &acc, float a, float b)
{
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_or_b64 %0, %0, vcc"
                 : "+s"(acc) : "v"(a), "v"(b) : "vcc", "scc");
}

// bit `shift` of west / east |= (a < b) in lane 0 / lane 63.  One block, so that the compare
// mask is consumed at once (see or_changed).
// acc <- 2 acc + (a < b), per lane: a shift register of compare results that costs two vector
// instructions per step and no scalar one (the carry-in of v_addc is the compare's mask).
__device__ __forceinline__ void shift_in_less(unsigned &acc, float a, float b)
{
    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\t"
                 "v_addc_co_u32 %0, vcc, %0, %0, vcc"
                 : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}

__device__ __forceinline__ void column_bits(unsigned &west, unsigned &east, float a, float b,
                                            int shift)
{
    unsigned t;
    asm volatile("v_cmp_lt_f32 vcc, %3, %4\n\t"
                 "s_and_b32 %2, vcc_lo, 1\n\t"
                 "s_lshl_b32 %2, %2, %5\n\t"
                 "s_or_b32 %0, %0, %2\n\t"
                 "s_lshr_b32 %2, vcc_hi, 31\n\t"
                 "s_lshl_b32 %2, %2, %5\n\t"
                 "s_or_b32 %1, %1, %2"
                 : "+s"(west), "+s"(east), "=&s"(t)
                 : "v"(a), "v"(b), "i"(shift)
                 : "vcc", "scc");
}

struct scan_masks {                 // wave-uniform (SGPR) lane masks of changed cells
    unsigned long long first, last, all;
};

// Two opposite directional scans over the 64 lines held in registers, interleaved
// instruction by instruction: the forward chain walks lines 1..62 while the
// backward chain walks 62..1.  Each step takes the three neighbours on the line its
// own chain visited just before: two v_mov_dpp wave shifts, v_min3, v_med3, v_cmp.
// A chain is latency-bound (each step waits on the previous one), so running the
// two together costs little more than one.  Once the chains have crossed, each
// walks over lines the other already lowered and simply uses the newer values.
// Lines 0 and 63 and lanes 0 and 63 are the pinned ring (z == w there, so med3
// returns w).  Needs z[i] <= w[i], which every valid upper bound of W* satisfies.
template <bool HAS_EPS>
__device__ __forceinline__ void scan_lines(const float (&z)[WN], float (&w)[WN], float eps)
{
    // (no change detection in here: a compare per row step is a fifth of the scan's
    // vector instructions, and the visit only needs to know which of its four edge lines
    // moved -- it compares those around the scans instead)
    // One step of both chains as one hand-scheduled block.  The lane shifts ride on the
    // mins (v_min_f32_dpp; the intrinsic route is v_mov_dpp x2 + v_min3, and neither the
    // DPP combiner nor the instruction selector folds wave shifts): 6 vector instructions
    // per step pair instead of 8 plus hazard nops.  A DPP read needs two wait states
    // after a vector write of its source; the two chains are interleaved so that there is
    // always one full instruction of the other chain in between -- the compiler cannot
    // see into the block, so the order below is the hazard management.  Lanes 0 / 63
    // take the min with 0 (bound_ctrl): they are pinned ring lanes, z = w clamps them.
    float pf = w[0], pb = w[WN - 1];
    // (whatever wrote w[0] / w[63] last, the first DPP read below is two wait states away)
    asm volatile("s_nop 1" : "+v"(pf), "+v"(pb));
#pragma unroll
    for (int k = 1; k <= WN - 2; ++k) {
        const int i = k, j = WN - 1 - k;
        float t1, t2, cf, cb, nf, nb;
        if (HAS_EPS)
            asm("v_min_f32_dpp %0, %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %1, %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %2, %6, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %3, %7, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_add_f32 %2, %2, %12\n\t"
                "v_add_f32 %3, %3, %12\n\t"
                "v_med3_f32 %4, %8, %9, %2\n\t"
                "v_med3_f32 %5, %10, %11, %3"
                : "=&v"(t1), "=&v"(t2), "=&v"(cf), "=&v"(cb), "=&v"(nf), "=&v"(nb)
                : "v"(pf), "v"(pb), "v"(z[i]), "v"(w[i]), "v"(z[j]), "v"(w[j]), "v"(eps));
        else
            asm("v_min_f32_dpp %0, %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %1, %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %2, %6, %0 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_min_f32_dpp %3, %7, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                "v_med3_f32 %4, %8, %9, %2\n\t"
                "v_med3_f32 %5, %10, %11, %3"
                : "=&v"(t1), "=&v"(t2), "=&v"(cf), "=&v"(cb), "=&v"(nf), "=&v"(nb)
                : "v"(pf), "v"(pb), "v"(z[i]), "v"(w[i]), "v"(z[j]), "v"(w[j]));
        w[i] = nf;
        pf = nf;
        w[j] = nb;
        pb = nb;
    }
}

// One full application of T in the column layout (Jacobi order): row r takes
// the 3x3 minimum around each cell.  Costs a third of a 4-scan iteration and
// is the convergence test: if it lowers nothing, the window is at its fixed
// point for the current halo.
template <bool HAS_EPS>
__device__ __forceinline__ void check_rows(const float (&z)[WN], float (&w)[WN], float eps,
                                           scan_masks &m)
{
    // h[r] = min of row r at lanes c-1, c, c+1 (self included: harmless, med3 clamps at w)
    float h_prev = fminf(fminf(w[0], lane_prev(w[0])), lane_next(w[0]));
    float h_cur = fminf(fminf(w[1], lane_prev(w[1])), lane_next(w[1]));
#pragma unroll
    for (int r = 1; r <= WN - 2; ++r) {
        const float h_next = fminf(fminf(w[r + 1], lane_prev(w[r + 1])), lane_next(w[r + 1]));
        float c = fminf(fminf(h_prev, h_cur), h_next);
        if (HAS_EPS) c = c + eps;
        const float n = __builtin_amdgcn_fmed3f(z[r], w[r], c);
        if (r == 1) or_changed(m.first, n, w[r]);
        else if (r == WN - 2) or_changed(m.last, n, w[r]);
        else or_changed(m.all, n, w[r]);
        w[r] = n;
        h_prev = h_cur;
        h_cur = h_next;
    }
}

// registers (lane = column, index = row)  <->  registers (lane = row, index = column)
__device__ __forceinline__ void transpose(float (&v)[WN], float *T, int lane)
{
#pragma unroll
    for (int i = 0; i < WN; ++i) T[i * TS + lane] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WN; ++i) v[i] = T[lane * TS + i];
    __syncthreads();
}

__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// order-preserving float -> int key (so that atomicMin on ints orders floats)
__device__ __forceinline__ int float_key(float f)
{
    int i = __builtin_bit_cast(int, f);
    return i ^ ((i >> 31) & 0x7fffffff);
}

#ifdef HDEM_VISIT_PROF
#define PROF_MARK(k) do { long long now_ = wall_clock64(); out.ticks[k] += now_ - tprof_; tprof_ = now_; } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif

// Hub start: a tile holds d (cost to its hub) until its first visit makes max(d, level of the
// hub) of it.  lv[3 * (dy + 1) + dx + 1]: that level for the tile at (ty + dy, tx + dx) while
// it has not had its first visit, -inf once it has (or when there is no such tile).
struct hub_bounds {
    float lv[9];
    bool mine;         // this tile's own first visit: write the tile back whatever happens
};

struct visit_result {
#ifdef HDEM_VISIT_PROF
    long long ticks[6];   // load, first check, zt transpose, iterations, store, wake tests
#endif
    bool changed;      // some interior cell was lowered (the tile was written back)
    bool more;         // still changing at the iteration cap: visit again
    unsigned dirs;     // bit k set: neighbour k (NW,N,NE,W,E,SW,S,SE) can use the new edge
    int iters;
    int flat_bits;     // float bits of the level the whole interior ended at, or FLAT_NONE
    int zmax_bits;     // ... and of the highest terrain cell under it
};

// Relax one tile to its fixed point for the current halo.  Whole wave, wave-uniform
// result.  T: the wave's 64 x 65 float LDS buffer.
// COHERENT (asynchronous driver): W is read and written with agent-scope (sc1)
// accesses -- loads bypass L1, stores write through -- so that a tile handed from
// one workgroup to another inside the launch needs no cache-wide fence
// (MI355X_MICROARCH.md, "Valid forms").  The round driver uses plain accesses.
template <bool HAS_EPS, bool COHERENT>
__device__ __forceinline__ visit_result tile_visit(const float *__restrict__ zg, float *wg,
                                                   int H, int W, float eps, int ty, int tx,
                                                   float *T, uint8_t *d8 = nullptr,
                                                   float flat_level = HDEM_INF,
                                                   const hub_bounds *hb = nullptr)
{
    const int lane = threadIdx.x;
    const int y0 = ty * FT, x0 = tx * FT;              // window origin (= halo row/col)
    const int x = x0 + lane;
#ifdef HDEM_VISIT_PROF
    const long long tprof_entry_ = wall_clock64();
#endif

    // ---- load the window: lane = column -------------------------------------
    // Addresses are clamped into the raster instead of predicated, so that all 128
    // row loads of the lane are in flight together.
    float z[WN], w[WN];
    const int xc = min(x, W - 1);
    // Buffer instructions take 32-bit byte offsets: the resource is based at the window's
    // first row (wave-uniform), so offsets stay below 64 rows whatever the raster size.
    float *wbase = wg + (size_t)y0 * W;
    const size_t wbytes = (size_t)(H - y0) * W * sizeof(float);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        wbase, 0, COHERENT ? (int)(unsigned)(wbytes < 0xffffffffull ? wbytes : 0xffffffffull) : 0,
        0x00020000);
    // window row of the raster's last row (uniform; >= 63 unless the window overhangs)
    const int hbrow = H - 1 - y0;
    if (COHERENT) {
        // Z through a resource based at the same row: one scalar row offset (s_min, s_mul) serves
        // both loads of a row and the lane's column offset is one register for all 128 -- the
        // 64-bit address arithmetic of 64 global loads was 300 scalar instructions per visit
        const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(zg) + (size_t)y0 * W, 0,
            (int)(unsigned)(wbytes < 0xffffffffull ? wbytes : 0xffffffffull), 0x00020000);
        const unsigned col = (unsigned)xc * (unsigned)sizeof(float);
        const unsigned pitch = (unsigned)W * (unsigned)sizeof(float);
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            const unsigned row = (unsigned)min(r, hbrow) * pitch;
            z[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(zrsrc, col, row, 0));
            // (an agent-scope __hip_atomic_load is waited for one by one -- 64 serial round
            // trips; a buffer load with the sc1 bit is an ordinary, pipelined load)
            w[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, col, row, AUX_SC1));
        }
    } else {
#pragma unroll
        for (int r = 0; r < WN; ++r) {
            const size_t o = (size_t)min(y0 + r, H - 1) * W + xc;
            z[r] = zg[o];
            w[r] = wg[o];
        }
    }
    // Normalise: nodata (NaN) becomes the wall +inf in both arrays (fminf returns the
    // operand that is a number), cells outside the raster likewise, and the cells that
    // are pinned for this visit -- window ring, raster ring -- get z = w so that the
    // relaxation leaves them alone.  Every visit pays this, so it is kept to ~3 vector
    // operations per row: what depends on the lane is decided once, what depends on the
    // row is uniform.
    const bool lane_pin = lane == 0 || lane == WN - 1 || x == 0 || x >= W - 1;
    const bool lane_out = x >= W;
    const bool partial = x0 + WN > W || y0 + WN > H;   // uniform: the window overhangs
    // Halo cells that sit above their own floor (z < w): only those can be lowered by a
    // new edge of mine, so only those justify waking the tile that owns them (7 % fewer
    // visits).  Halo rows and halo columns are both tested here, while their Z is in
    // registers (lane masks for the rows, lane = row masks for the columns).  (Testing
    // against halo values re-read after the write-back instead of the copies held since
    // the load saves another 3 % of the visits but puts 8 loads on the path to the
    // hand-off: 6.2 against 5.9 ms.)
    unsigned long long free_n = ~0ull, free_s = ~0ull;
    unsigned free_lo = 0, free_hi = 0;          // per lane: bit 31 - r of lo / hi <- row r / 32 + r free
    // (hub start) cells of tiles that have not had their first visit still hold d: raise them
    // to their tile's start bound -- per lane the bound of the tile above / beside / below
    float hub_t = -HDEM_INF, hub_m = -HDEM_INF, hub_b = -HDEM_INF;
    if (COHERENT && hb) {
        const int k = lane == 0 ? 0 : (lane == WN - 1 ? 2 : 1);
        hub_t = k == 0 ? hb->lv[0] : (k == 1 ? hb->lv[1] : hb->lv[2]);
        hub_m = k == 0 ? hb->lv[3] : (k == 1 ? hb->lv[4] : hb->lv[5]);
        hub_b = k == 0 ? hb->lv[6] : (k == 1 ? hb->lv[7] : hb->lv[8]);
    }
    // What depends on the row is decided per lane against a row number held in a register --
    // one vector compare with the row as a literal -- not per row in scalar code: the uniform
    // form cost ten scalar instructions per row (compare, select, combine the masks), 640 per
    // visit, for conditions that only the raster's last tiles ever meet.
    //   pin_from: rows from here on are pinned in this lane (pinned lane: all; else the raster's
    //             last row and what lies beyond it -- copies of that row, see the loads)
    //   raise_to: the hub bound applies to rows below this one (tile interiors only: on the
    //             raster's last tiles the window also holds the raster ring, which is the
    //             boundary condition, not a start value)
    const int pin_from = lane_pin ? 0 : min(hbrow, WN - 1);
    const int raise_to = x >= W - 1 ? 0 : min(hbrow, WN);
#pragma unroll
    for (int r = 0; r < WN; ++r) {
        float wc = fminf(w[r], HDEM_INF), zc = fminf(z[r], HDEM_INF);
        if (COHERENT) {
            const float bound = r == 0 ? hub_t : (r == WN - 1 ? hub_b : hub_m);
            wc = fmaxf(wc, r < raise_to ? bound : -HDEM_INF);
        }
        if (!COHERENT && partial && (lane_out || y0 + r >= H)) { wc = HDEM_INF; zc = HDEM_INF; }
        if (COHERENT) {
            // rows 0 / 63: the halo rows, one bit per lane; every row: its two halo-column
            // cells sit in lanes 0 and 63 right now -- every lane shifts its own compare into a
            // register (two vector instructions, no scalar one) and lanes 0 and 63 are read
            // out after the loop: no second, strided fetch of Z later
            if (r == 0) free_n = __ballot(zc < wc);
            if (r == WN - 1) free_s = __ballot(zc < wc);
            if (r < 32) shift_in_less(free_lo, zc, wc);
            else shift_in_less(free_hi, zc, wc);
        }
        // (a pinned nodata cell has w = NaN -> +inf as well, so z = w is the wall there too)
        z[r] = (r == 0 || r == WN - 1 || r >= pin_from) ? wc : zc;
        w[r] = wc;
    }
    if (COHERENT && partial) {
        // Cells beyond the raster are copies of its ring (clamped loads), pinned like it: nothing
        // inside the tile can see them past the ring.  They must not count as halo cells that
        // a new edge could lower, though -- their bits leave the masks here, once.
        const unsigned long long in_lanes = __ballot(!lane_out);
        free_n &= in_lanes;
        free_s = hbrow >= WN - 1 ? (free_s & in_lanes) : 0ull;
    }

    if (COHERENT && flat_level < HDEM_INF) {
        // the tile was last left as "flat at this level" with only its edge lines written
        // (flat_visit): its interior in memory is stale, the level is the truth
        const bool inner_lane = lane >= 2 && lane <= FT - 1;
#pragma unroll
        for (int r = 2; r <= FT - 1; ++r) w[r] = inner_lane ? flat_level : w[r];
    }

    // the halo columns are lanes 0 and 63: their registers, bit-reversed, are lane = row masks
    unsigned long long free_w = 0, free_e = 0;
    if (COHERENT) {
        const unsigned wl = __builtin_amdgcn_readlane(free_lo, 0), wh = __builtin_amdgcn_readlane(free_hi, 0);
        const unsigned el = __builtin_amdgcn_readlane(free_lo, WN - 1), eh = __builtin_amdgcn_readlane(free_hi, WN - 1);
        free_w = ((unsigned long long)__builtin_bitreverse32(wh) << 32) | __builtin_bitreverse32(wl);
        free_e = ((unsigned long long)__builtin_bitreverse32(eh) << 32) | __builtin_bitreverse32(el);
        if (partial && hbrow < WN - 1) {
            const unsigned long long in_rows = (2ull << hbrow) - 1;         // rows 0..hbrow
            free_w &= in_rows;
            free_e &= in_rows;
        }
    }
    visit_result out;
    out.flat_bits = out.zmax_bits = FLAT_NONE;
#ifdef HDEM_VISIT_PROF
    for (int k = 0; k < 6; ++k) out.ticks[k] = 0;
    long long tprof_ = tprof_entry_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PROF_MARK(0);
#endif
    out.dirs = 0;
    out.iters = 0;
    scan_masks V = {0, 0, 0};
    check_rows<HAS_EPS>(z, w, eps, V);
    V.all |= V.first | V.last;
    bool more = V.all != 0;
    out.changed = more;
    const bool force_store = COHERENT && hb && hb->mine;
    // lanes whose cell of row 1 / row 62 / column 1 / column 62 moved during the visit
    // (columns: lane = row) -- what the wake tests below are gated on
    const unsigned long long b1 = 1ull << 1, bl = 1ull << FT;
    unsigned long long mv_r1 = V.first, mv_r62 = V.last, mv_c1 = V.all & b1, mv_c62 = V.all & bl;
    PROF_MARK(1);
    if (more || force_store) {
        float zt[WN];
#pragma unroll
        for (int r = 0; r < WN; ++r) zt[r] = z[r];
        transpose(zt, T, lane);                        // zt: lane = row
        // the four edge lines as they are now; the columns sit in lanes 1 and 62 and come
        // back through LDS as lane = row vectors
        float r1_prev = w[1], r62_prev = w[FT];
        if (lane == 1 || lane == FT) {
            float *dst = T + (lane == 1 ? 0 : WN);
#pragma unroll
            for (int r = 0; r < WN; ++r) dst[r] = w[r];
        }
        __syncthreads();
        float c1_prev = T[lane], c62_prev = T[WN + lane];
        __syncthreads();
        float hl = HDEM_INF, hr = HDEM_INF;            // the halo columns as lane = row vectors
        PROF_MARK(2);
        for (; out.iters < ITER_MAX && more; ++out.iters) {
            scan_lines<HAS_EPS>(z, w, eps);            // north -> south and south -> north
            transpose(w, T, lane);
            hl = w[0];                                 // (pinned: the same every time)
            hr = w[WN - 1];
            or_changed(mv_c1, w[1], c1_prev);          // columns: what the vertical scans (and
            or_changed(mv_c62, w[FT], c62_prev);       // the last check) did to them
            c1_prev = w[1];
            c62_prev = w[FT];
            scan_lines<HAS_EPS>(zt, w, eps);           // west -> east and east -> west
            or_changed(mv_c1, w[1], c1_prev);
            or_changed(mv_c62, w[FT], c62_prev);
            c1_prev = w[1];
            c62_prev = w[FT];
            // back to lane = column
#pragma unroll
            for (int i = 0; i < WN; ++i) T[lane * TS + i] = w[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < WN; ++i) w[i] = T[i * TS + lane];
            __syncthreads();
            or_changed(mv_r1, w[1], r1_prev);          // rows: both scans of this iteration
            or_changed(mv_r62, w[FT], r62_prev);
            scan_masks c = {0, 0, 0};
            check_rows<HAS_EPS>(z, w, eps, c);         // convergence test (and one more T step)
            c.all |= c.first | c.last;
            mv_r1 |= c.first;
            mv_r62 |= c.last;
            mv_c1 |= c.all & b1;
            mv_c62 |= c.all & bl;
            r1_prev = w[1];
            r62_prev = w[FT];
            more = c.all != 0;
        }

        PROF_MARK(3);
        // ---- write back --------------------------------------------------------
        if (lane >= 1 && lane <= FT && x <= W - 2) {
#pragma unroll
            for (int r = 1; r <= FT; ++r) {
                const int y = y0 + r;
                // a nodata cell was loaded as the wall z = w = +inf; it goes back as NaN
                // (an interior cell that no path reaches keeps w = +inf but a finite z)
                if (y <= H - 2) {
                    const float val = z[r] == HDEM_INF ? __builtin_nanf("") : w[r];
                    if (COHERENT)
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __builtin_bit_cast(unsigned, val), wrsrc,
                            (unsigned)(((size_t)r * W + x) * sizeof(float)), 0, AUX_SC1);
                    else
                        wg[(size_t)y * W + x] = val;
                }
            }
        }

#ifdef HDEM_VISIT_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        PROF_MARK(4);
        // ---- which neighbours can use the new edge, and how low it is -------------
        // A neighbour cell n next to my edge cell a can only drop if w[a] + eps < w[n];
        // w[n] is the halo value held since the load (stale means higher: the test
        // errs towards waking).  Rows first: lane c tests halo cell (0, c) / (63, c)
        // against my row 1 / 62 at lanes c-1, c, c+1.
        const unsigned long long mid = ((1ull << FT) - 1) << 1;
        const bool inner = lane >= 1 && lane <= FT;
        const float e1 = inner ? w[1] : HDEM_INF, e62 = inner ? w[FT] : HDEM_INF;
        // lanes 0/63 of the shifted copies read 0 under bound_ctrl: rebuild them as +inf.
        // The shifts are made by ALL lanes and patched afterwards: written as
        // `lane == 0 ? inf : lane_prev(v)` the shift lands under an exec mask without lane 0,
        // and a DPP read of a masked-off lane returns 0 -- lane 1 then sees a candidate of 0
        // and the neighbour is woken whatever the values are (rounds 1-2 ran that way).
        const float p1s = lane_prev(e1), n1s = lane_next(e1);
        const float p62s = lane_prev(e62), n62s = lane_next(e62);
        const float p1 = lane == 0 ? HDEM_INF : p1s;
        const float n1 = lane == WN - 1 ? HDEM_INF : n1s;
        const float p62 = lane == 0 ? HDEM_INF : p62s;
        const float n62 = lane == WN - 1 ? HDEM_INF : n62s;
        float cn = fminf(fminf(e1, p1), n1), cs = fminf(fminf(e62, p62), n62);
        if (HAS_EPS) { cn += eps; cs += eps; }
        unsigned long long north = 0, south = 0;
        or_less(north, cn, w[0]);
        or_less(south, cs, w[WN - 1]);
        north &= free_n;
        south &= free_s;
        const bool row1_moved = mv_r1 != 0, row62_moved = mv_r62 != 0;
        if (row1_moved) {
            if (north & mid) out.dirs |= 1u << 1;                                  // N
            if (north & 1ull) out.dirs |= 1u << 0;                                 // NW
            if (north & (1ull << (WN - 1))) out.dirs |= 1u << 2;                   // NE
        }
        if (row62_moved) {
            if (south & mid) out.dirs |= 1u << 6;                                  // S
            if (south & 1ull) out.dirs |= 1u << 5;                                 // SW
            if (south & (1ull << (WN - 1))) out.dirs |= 1u << 7;                   // SE
        }
        // Columns, as vectors with lane = row: c1_prev / c62_prev are my final edge columns
        // (the loop ends on a check that changed nothing, so the copies taken after the
        // last horizontal scan are final), hl / hr the halo columns.  Row r of the halo is
        // tested against my rows r-1, r, r+1; rows 0 and 63 of my columns are halo cells
        // of other tiles and do not count.
        const bool col1_moved = mv_c1 != 0, col62_moved = mv_c62 != 0;
        if (more) {
            // cut short by the iteration cap: the columns may have moved after the copies
            // were taken -- wake without the test (rare, and the tile runs again anyway)
            if (col1_moved) out.dirs |= 1u << 3;
            if (col62_moved) out.dirs |= 1u << 4;
        } else if (col1_moved || col62_moved) {
            const float m1 = inner ? c1_prev : HDEM_INF, m62 = inner ? c62_prev : HDEM_INF;
            const float pws = lane_prev(m1), nxs = lane_next(m1);        // (all lanes: see above)
            const float pes = lane_prev(m62), nes = lane_next(m62);
            const float pw = lane == 0 ? HDEM_INF : pws;
            const float nx = lane == WN - 1 ? HDEM_INF : nxs;
            const float pe = lane == 0 ? HDEM_INF : pes;
            const float ne = lane == WN - 1 ? HDEM_INF : nes;
            float cwest = fminf(fminf(m1, pw), nx), ceast = fminf(fminf(m62, pe), ne);
            if (HAS_EPS) { cwest += eps; ceast += eps; }
            unsigned long long west = 0, east = 0;
            or_less(west, cwest, hl);
            or_less(east, ceast, hr);
            if (COHERENT) {
                west &= free_w;                            // (as held since the load, like hl / hr:
                east &= free_e;                            // stale errs towards waking)
            }
            if (col1_moved && (west & mid)) out.dirs |= 1u << 3;                  // W
            if (col62_moved && (east & mid)) out.dirs |= 1u << 4;                 // E
        }
        PROF_MARK(5);
    }
    // ---- is the interior one flat level now?  (asynchronous driver, eps = 0) ----------------
    // A tile under a lake: every later visit only has to compare its halo with that level
    // (flat_visit) until the halo dips below the tile's highest terrain.  Row 31 decides
    // cheaply for nearly every tile that is not.
    if (COHERENT && !HAS_EPS && !partial && !more) {
        const bool inner_lane = lane >= 1 && lane <= FT;
        const float level = __builtin_bit_cast(
            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w[31]), 31));
        if ((__ballot(inner_lane && w[31] != level) == 0) && level < HDEM_INF) {
            unsigned long long off = 0;
            float zm = -HDEM_INF;
#pragma unroll
            for (int r = 1; r <= FT; ++r) {
                or_changed(off, w[r], level);
                zm = fmaxf(zm, z[r]);
            }
            if ((off & ((((1ull << FT) - 1) << 1))) == 0) {
                zm = inner_lane ? zm : -HDEM_INF;
                zm = -wave_min(-zm);
                out.flat_bits = __builtin_bit_cast(int, level);
                out.zmax_bits = __builtin_bit_cast(int, zm);
            }
        }
    }
    // ---- D8 of a certified tile (round driver, on request) ------------------------------
    // A visit that changes nothing has the final surface of the tile and of its halo in
    // registers: the flow directions cost no second pass over W.  Same arithmetic and tie
    // rule as d8_kernel (hdem_stencil.hip): drops (c - n) * w in float32, first maximum in
    // the order NW, N, NE, W, E, SW, S, SE; a nodata centre has no direction.
    if (!COHERENT && d8 && !out.changed) {
        // (all 64 lanes take part: a lane shift reads nothing from a lane that is masked off)
        const bool mine = lane >= 1 && lane <= FT && x <= W - 2;
        const float dg = 0.70710678f;
        float nw = lane_prev(w[0]), ne = lane_next(w[0]);
        float cw = lane_prev(w[1]), ce = lane_next(w[1]);
#pragma unroll
        for (int r = 1; r <= FT; ++r) {
            const float sw = lane_prev(w[r + 1]), se = lane_next(w[r + 1]);
            const float c = w[r];
            float best = 0.0f, d;
            unsigned code = 0;
            d = (c - nw) * dg;   if (d > best) { best = d; code = 32; }
            d = (c - w[r - 1]);  if (d > best) { best = d; code = 64; }
            d = (c - ne) * dg;   if (d > best) { best = d; code = 128; }
            d = (c - cw);        if (d > best) { best = d; code = 16; }
            d = (c - ce);        if (d > best) { best = d; code = 1; }
            d = (c - sw) * dg;   if (d > best) { best = d; code = 8; }
            d = (c - w[r + 1]);  if (d > best) { best = d; code = 4; }
            d = (c - se) * dg;   if (d > best) { best = d; code = 2; }
            if (z[r] == HDEM_INF) code = 0;
            const int y = y0 + r;
            if (mine && y <= H - 2) d8[(size_t)y * W + x] = (uint8_t)code;
            nw = cw; ne = ce; cw = sw; ce = se;
        }
    }
    out.more = more;
    return out;
}

__device__ __forceinline__ int neighbour_tile(int k, int ty, int tx, int tiles_x, int tiles_y)
{   // k: NW,N,NE,W,E,SW,S,SE ; -1 when outside the tile grid
    const int dy = k < 3 ? -1 : (k < 5 ? 0 : 1);
    const int dx = (k == 0 || k == 3 || k == 5) ? -1 : ((k == 1 || k == 6) ? 0 : 1);
    const int ny = ty + dy, nx = tx + dx;
    return (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x) ? ny * tiles_x + nx : -1;
}

__device__ __forceinline__ void add_stats(unsigned long long *stats, int b, bool changed,
                                          bool more, int iters)
{   // one writer per workgroup: plain read-modify-write, no atomics
    unsigned long long *s = stats + (size_t)b * STAT_WORDS;
    s[0] += 1;
    s[1] += (unsigned long long)iters;
    s[2] += changed ? 0 : 1;
    s[3] += more ? 1 : 0;
}

// ---------------------------------------------------------------------------
// Round-synchronous driver: one launch per round.  A tile is due in round r when
// its state word holds stamp r; waking a tile for the next round is a plain store
// of stamp r+1 (every writer stores the same value), and *any_next is set the
// same way -- no atomics.  Used to certify / finish the fixed point behind the
// asynchronous launch, and alone with HDEM_FILL_SYNC_ONLY.
// ---------------------------------------------------------------------------
template <bool HAS_EPS>
__global__ __launch_bounds__(NT, 2) void fill_round_kernel(const float *__restrict__ zg,
                                                          float *wg, int H, int W, float eps,
                                                          int tiles_x, int tiles_y, int ntiles,
                                                          int S, int *state, int stamp,
                                                          int *any_next,
                                                          unsigned long long *stats,
                                                          uint8_t *d8)
{
    __shared__ float T[WN * TS];
    const int lane = threadIdx.x;
    const int G = gridDim.x, b = blockIdx.x;
    int *my_state = state + (size_t)b * S;
    for (int base = 0; base < S; base += NT) {
        const int slot_l = base + lane;
        const bool due_l = slot_l < S && slot_l * G + b < ntiles && my_state[slot_l] == stamp;
        unsigned long long due = __ballot(due_l);
        while (due) {
            const int slot = base + __builtin_ctzll(due);
            due &= due - 1;
            const int t = slot * G + b;
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            visit_result v = tile_visit<HAS_EPS, false>(zg, wg, H, W, eps, ty, tx, T, d8);
            // with flow directions asked for, a tile that still moved is seen again, and so
            // are all its neighbours (their directions read its edge): every tile's last
            // visit is one that changed nothing and wrote its codes from final values
            if (d8 && v.changed) { v.dirs = 0xffu; v.more = true; }
            if (lane < 8 && ((v.dirs >> lane) & 1u)) {
                const int t2 = neighbour_tile(lane, ty, tx, tiles_x, tiles_y);
                if (t2 >= 0) {
                    state[(size_t)(t2 % G) * S + t2 / G] = stamp + 1;
                    *any_next = 1;
                }
            }
            if (lane == 8) {
                if (v.more) {
                    my_state[slot] = stamp + 1;
                    *any_next = 1;
                }
                add_stats(stats, b, v.changed, v.more, v.iters);
                stats[(size_t)b * STAT_WORDS + 6] += 1;
#ifdef HDEM_VISIT_PROF
                for (int k = 0; k < 6; ++k) stats[(size_t)b * STAT_WORDS + STAT_PROF + k] += v.ticks[k];
#endif
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Asynchronous driver.  state: IDLE -> QUEUED (by a neighbour) -> RUNNING (by the
// owner) -> IDLE; RUNNING -> DIRTY when a neighbour's edge moves during the
// visit; a DIRTY tile goes back to QUEUED when its visit ends.
//   pend[]: tiles in a non-IDLE state, sharded over 64 cache lines.  A workgroup
//          leaves when it has nothing queued and reads an all-zero sum twice;
//          leaving early is harmless -- the round-synchronous pass behind this
//          kernel finishes whatever is left and is what certifies the result.
//   memory: the writer drains its stores, releases at agent scope, then wakes;
//          the owner marks RUNNING, acquires at agent scope, then loads
//          (cdna_hip_programming.md, Guideline 16).
//   exit:  every wait is bounded by a wall-clock budget; on expiry the kernel
//          sets *error and drains.
// The scheduler half is kept out of line: inlined around the visit it pushes the
// visit body over 256 VGPRs; as calls made while no window is live it is free.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int ld_relaxed(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void wake_async(int t2, int key, int G, int S, int *state,
                                           int *prio, int *pend)
{
    const int owner = t2 % G, idx = owner * S + t2 / G;
    atomicMin(&prio[idx], key);
    for (int tries = 0; tries < 8; ++tries) {
        const int s = atomicCAS(&state[idx], ST_IDLE, ST_QUEUED);
        if (s == ST_IDLE) {
            atomicAdd(pend + (owner % PEND_SHARDS) * PEND_STRIDE, 1);
            return;
        }
        if (s == ST_QUEUED || s == ST_DIRTY) return;
        const int s2 = atomicCAS(&state[idx], ST_RUNNING, ST_DIRTY);
        if (s2 != ST_IDLE) return;                      // RUNNING->DIRTY done, or already marked
    }
}

// Returns the tile to relax next, -1 to stop (drained or out of budget).
__device__ __attribute__((noinline)) int async_pick(int b, int G, int S, int ntiles, int *state,
                                                    int *prio, const int *pend, int *error,
                                                    long long t_begin, long long budget_ticks)
{
    const int lane = threadIdx.x;
    int *my_state = state + (size_t)b * S, *my_prio = prio + (size_t)b * S;
    int zero_reads = 0;
    for (;;) {
        // the queued slot with the lowest key
        long long best = 0x7fffffffffffffffll;
        for (int base = 0; base < S; base += NT) {
            const int slot = base + lane;
            long long cand = 0x7fffffffffffffffll;
            if (slot < S && slot * G + b < ntiles && ld_relaxed(&my_state[slot]) == ST_QUEUED)
                cand = ((long long)ld_relaxed(&my_prio[slot]) << 32) | (unsigned)slot;
            best = cand < best ? cand : best;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const long long other = __shfl_xor(best, o);
            best = other < best ? other : best;
        }
        if (best != 0x7fffffffffffffffll) {
            const int slot = (int)(best & 0xffffffffll);
            // claim it (a thief may be after the same tile)
            int got = 0;
            if (lane == 0) got = atomicCAS(&my_state[slot], ST_QUEUED, ST_RUNNING) == ST_QUEUED;
            got = __shfl(got, 0);
            if (!got) continue;
            if (lane == 0) atomicExch(&my_prio[slot], KEY_NONE);
            return slot * G + b;
        }
        // Nothing of my own is queued: steal.  Ownership only says where a tile's state
        // word lives; any workgroup may run a queued tile once it has won the CAS.
        for (int tries = 0; tries < 4; ++tries) {
            const int victim = (int)((unsigned)(b + 1 + (unsigned)(wall_clock64() >> 3) % (unsigned)G +
                                                tries * 997u) % (unsigned)G);
            const int *vs = state + (size_t)victim * S;
            int found = -1;
            for (int base = 0; base < S && found < 0; base += NT) {
                const int slot = base + lane;
                const bool q = slot < S && slot * G + victim < ntiles &&
                               ld_relaxed(&vs[slot]) == ST_QUEUED;
                const unsigned long long m = __ballot(q);
                if (m) found = base + __builtin_ctzll(m);
            }
            if (found >= 0) {
                int got = 0;
                if (lane == 0)
                    got = atomicCAS(&state[(size_t)victim * S + found], ST_QUEUED, ST_RUNNING) ==
                          ST_QUEUED;
                got = __shfl(got, 0);
                if (got) {
                    if (lane == 0) atomicExch(&prio[(size_t)victim * S + found], KEY_NONE);
                    return found * G + victim;
                }
            }
        }
        // nothing queued here: leave once the whole raster looks drained
        int p = ld_relaxed(&pend[lane * PEND_STRIDE]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o);
        if (p <= 0) { if (++zero_reads >= 2) return -1; } else zero_reads = 0;
        if (wall_clock64() - t_begin > budget_ticks) {
            if (lane == 0) atomicCAS(error, 0, 1);
            return -1;
        }
        if (ld_relaxed(error) != 0) return -1;
        __builtin_amdgcn_s_sleep(32);
    }
}

// Publish a finished visit: release the written tile, wake the neighbours that can
// use the new edge, retire (or re-queue) the tile.
__device__ __attribute__((noinline)) void async_finish(int t, int b, int G, int S, int tiles_x,
                                                       int tiles_y, int *state, int *prio,
                                                       int *pend, unsigned long long *stats,
                                                       bool changed, bool more, unsigned dirs,
                                                       int iters)
{
    const int lane = threadIdx.x;
    const int ty = t / tiles_x, tx = t - ty * tiles_x, slot = t / G, owner = t % G;
    int *my_state = state + (size_t)owner * S, *my_prio = prio + (size_t)owner * S;
    // queue order: first come, first served (key = time of the first wake, 100 MHz
    // ticks).  Flood order (key = lowest elevation offered) was tried and is worse: a
    // tile woken early and low runs before its other neighbours have spoken.
    const int now_key = SEED_KEYS + (int)(wall_clock64() & 0x3fffffff);
    // A tile that wakes nobody needs no release: nobody depends on seeing it before
    // the launch ends (its edge cannot lower any neighbour cell).
    if (changed && dirs) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through stores have landed
        if (lane < 8 && ((dirs >> lane) & 1u)) {
            const int t2 = neighbour_tile(lane, ty, tx, tiles_x, tiles_y);
            if (t2 >= 0) wake_async(t2, now_key, G, S, state, prio, pend);
        }
    }
    if (lane == 8) {
        if (more) atomicMin(&my_prio[slot], now_key);
        const int target = more ? ST_QUEUED : ST_IDLE;
        const int old = atomicCAS(&my_state[slot], ST_RUNNING, target);
        if (old == ST_DIRTY) atomicExch(&my_state[slot], ST_QUEUED);
        else if (target == ST_IDLE) atomicAdd(pend + (owner % PEND_SHARDS) * PEND_STRIDE, -1);
        add_stats(stats, b, changed, more, iters);
    }
}

// A visit of a tile whose interior is one flat level L (a tile under a lake) and whose
// highest terrain cell is zmax.  Its fixed point for the halo it finds is
//     min(L, lowest halo cell)  on every interior cell,
// provided that minimum is not below zmax: all interior cells are connected below it, and no
// way in is lower.  So the visit reads the 252 halo cells instead of two 64 x 64 windows and,
// when the level drops, writes the four edge lines the neighbours read as their halo; the
// rest of the interior is written once, behind the launch (flat_store_kernel).  Returns 0 when
// the halo dips below the terrain -- the tile then takes the ordinary visit, which starts
// from the level instead of the stale interior in memory.
struct flat_result {
    int handled;
    bool changed;
    unsigned dirs;
    float level;
};

__device__ __attribute__((noinline)) flat_result flat_visit(float *wg, int H, int W, int ty, int tx,
                                                            float L, float zmax,
                                                            const hub_bounds *hb)
{
    const int lane = threadIdx.x;
    const int y0 = ty * FT, x0 = tx * FT;
    float *wbase = wg + (size_t)y0 * W;
    const size_t wbytes = (size_t)(H - y0) * W * sizeof(float);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        wbase, 0, (int)(unsigned)(wbytes < 0xffffffffull ? wbytes : 0xffffffffull), 0x00020000);
    auto at = [&](int r, int c) { return (unsigned)(((size_t)r * W + x0 + c) * sizeof(float)); };
    auto ld = [&](unsigned off) {
        return fminf(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                         wrsrc, off, 0, AUX_SC1)), HDEM_INF);          // nodata: a wall
    };
    // (only tiles whose window lies inside the raster are ever flat)
    float top = ld(at(0, lane)), bottom = ld(at(WN - 1, lane));
    float left = ld(at(lane, 0)), right = ld(at(lane, WN - 1));             // lane = row
    if (hb) {
        // (hub start) halo cells of tiles that still hold d: raised to their start bound
        const int k = lane == 0 ? 0 : (lane == WN - 1 ? 2 : 1);
        top = fmaxf(top, k == 0 ? hb->lv[0] : (k == 1 ? hb->lv[1] : hb->lv[2]));
        bottom = fmaxf(bottom, k == 0 ? hb->lv[6] : (k == 1 ? hb->lv[7] : hb->lv[8]));
        left = fmaxf(left, k == 0 ? hb->lv[0] : (k == 1 ? hb->lv[3] : hb->lv[6]));
        right = fmaxf(right, k == 0 ? hb->lv[2] : (k == 1 ? hb->lv[5] : hb->lv[8]));
    }
    const float low = wave_min(fminf(fminf(top, bottom), fminf(left, right)));
    flat_result out = {1, false, 0u, L};
    if (low >= L) return out;                       // nothing new
    if (low < zmax) { out.handled = 0; return out; }
    out.changed = true;
    out.level = low;
    const unsigned bits = __builtin_bit_cast(unsigned, low);
    if (lane >= 1 && lane <= FT) {
        __builtin_amdgcn_raw_buffer_store_b32(bits, wrsrc, at(1, lane), 0, AUX_SC1);
        __builtin_amdgcn_raw_buffer_store_b32(bits, wrsrc, at(FT, lane), 0, AUX_SC1);
        if (lane >= 2 && lane <= FT - 1) {
            __builtin_amdgcn_raw_buffer_store_b32(bits, wrsrc, at(lane, 1), 0, AUX_SC1);
            __builtin_amdgcn_raw_buffer_store_b32(bits, wrsrc, at(lane, FT), 0, AUX_SC1);
        }
    }
    // wake whoever holds a cell above the new level next to it (the terrain of those cells
    // is not at hand here: no floor filter, a few wakes too many)
    const unsigned long long mid = ((1ull << FT) - 1) << 1, hi = 1ull << (WN - 1);
    const unsigned long long n = __ballot(top > low), s_ = __ballot(bottom > low);
    const unsigned long long w_ = __ballot(left > low), e = __ballot(right > low);
    if (n & 1ull) out.dirs |= 1u << 0;
    if (n & mid) out.dirs |= 1u << 1;
    if (n & hi) out.dirs |= 1u << 2;
    if (w_ & mid) out.dirs |= 1u << 3;
    if (e & mid) out.dirs |= 1u << 4;
    if (s_ & 1ull) out.dirs |= 1u << 5;
    if (s_ & mid) out.dirs |= 1u << 6;
    if (s_ & hi) out.dirs |= 1u << 7;
    return out;
}

// The interiors of the tiles that ended the launch flat (their edge lines are in place).
// Hub start: a tile the launch never got to (a launch cut short by its budget) still holds d;
// it gets its start bound here, so that whatever runs next finds upper bounds everywhere.
__global__ __launch_bounds__(NT) void flat_store_kernel(float *wg, int H, int W, int tiles_x,
                                                        int ntiles, const int *__restrict__ flat,
                                                        const float *__restrict__ hub_lev,
                                                        const int *__restrict__ applied,
                                                        const int *__restrict__ seam_words)
{
    const int t = blockIdx.x;
    if (t >= ntiles) return;
    if (seam_words && (seam_words[0] | seam_words[1] | seam_words[2]) == 0) return;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int lane = threadIdx.x;
    if (hub_lev && !applied[t]) {
        float l = hub_lev[(size_t)(2 * ty + 1) * (2 * tiles_x + 1) + 2 * tx + 1];
        if (l >= HUB_BIG) l = HDEM_INF;
        const int x = tx * FT + lane;
        if (l == l && lane >= 1 && lane <= FT && x <= W - 2) {
            float *p = wg + (size_t)(ty * FT + 1) * W + x;
            for (int r = 1; r <= FT && ty * FT + r <= H - 2; ++r, p += W) {
                const float v = *p;
                if (v == v) *p = fmaxf(v, l);
            }
        }
        return;
    }
    if (flat[t] == FLAT_NONE) return;
    const float level = __builtin_bit_cast(float, flat[t]);
    if (lane < 2 || lane > FT - 1) return;
    float *p = wg + (size_t)(ty * FT + 2) * W + tx * FT + lane;
    for (int r = 2; r <= FT - 1; ++r, p += W) *p = level;
}

// ---------------------------------------------------------------------------
// Hub start (INIT, eps = 0): start values from a graph of tile hubs.
//
// Any upper bound of the fill is a legal start of the relaxation, and how many visits the
// relaxation then needs is decided by how exact the bound is along the drainage lines.  The
// block-maximum raster of rounds 1-2 bounds a lake from above by the highest *cell* of every
// block on the way out: 1.5 m too high on this terrain, walked down pass by pass.  This one
// bounds it by actual paths:
//   * every tile gets a hub, its lowest cell, and d(c) = the minimax cost of the best path from
//     c to the hub that stays inside the tile interior: a single-source relaxation, two rounds
//     of the four directional scans of a visit from +inf (any state of that relaxation is the
//     cost of some path, so stopping early only loosens the bound; a third round changes 2 %
//     of the cells);  a tile that holds pinned cells (nodata fringe) takes those as its
//     sources instead and is an outlet of the graph;
//   * hubs of tiles that share a seam are joined by the cheapest crossing,
//     min over adjacent cells a | b of max(d(a), d(b));  tiles on the raster ring are joined
//     to it the same way (d(a) against the ring cell's elevation);
//   * the hub graph is a node-weighted raster of (2 ty + 1) x (2 tx + 1) cells -- hubs at
//     odd/odd, crossings between them, walls at even/even, the crossings to the raster ring on
//     its own ring -- and is filled exactly by this same solver (1/3700 of the cells);
//   * start value of a free cell: max(d(c), level(hub of its tile)): c -> hub inside the tile,
//     hub -> raster ring along the graph.
// Measured on the bench raster (16384^2 "rough"): 67 % of the hub levels and 53 % of the raised
// cells are exact from the start, mean excess 0.13 m; 3.5 visits per tile instead of 6.6.
// ---------------------------------------------------------------------------

template <int ITERS>
__global__ __launch_bounds__(NT, 2) void hub_dist_kernel(const float *__restrict__ zg,
                                                        float *__restrict__ wg, int H, int W,
                                                        int tiles_x, int ntiles,
                                                        float *__restrict__ edge,
                                                        float *__restrict__ node)
{
    __shared__ float T[WN * TS];
    const int lane = threadIdx.x;
    // XCD-aware tile order: workgroups go to the 8 XCDs round-robin by index, each XCD with an
    // L2 of its own.  Workgroup b takes tile (b % 8) * per + b / 8, so every XCD walks one
    // contiguous eighth of the tiles and the tiles in flight on it are neighbours in a tile row:
    // the 128-byte lines that two adjacent windows share (a window row is 256 bytes at a
    // multiple of 248: three lines, the outer two shared) are fetched once per XCD, and the two
    // halves of a line that two tiles write meet in one L2.
    const int per = (ntiles + 7) / 8;
    for (int b = blockIdx.x; b < 8 * per; b += gridDim.x) {
    const int t = (b & 7) * per + (b >> 3);
    if (t >= ntiles) continue;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * FT, x0 = tx * FT, x = x0 + lane, xc = min(x, W - 1);
    float z[WN], w[WN];
    // Z and W through buffer resources based at the window's first row: one literal multiple of
    // the row pitch per row for both, no 64-bit address arithmetic.  Z's ends behind the
    // raster's last row (window row hbrow; rows past it read 0, and nothing looks at them), W's
    // before it: a store to a row that is not tile interior falls off the end and is dropped.
    const int hbrow = H - 1 - y0;
    const unsigned pitch = (unsigned)W * (unsigned)sizeof(float), col = (unsigned)xc * (unsigned)sizeof(float);
    const size_t zbytes = (size_t)min(hbrow + 1, WN) * pitch, dbytes = (size_t)min(hbrow, WN) * pitch;
    const __amdgpu_buffer_rsrc_t zrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(zg) + (size_t)y0 * W, 0, (int)(unsigned)zbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(
        wg + (size_t)y0 * W, 0, (int)(unsigned)dbytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < WN; ++r)
        z[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            zrsrc, col, (unsigned)r * pitch, 0));
    // any nodata in the window?  A sum is NaN as soon as one term is (one vector add per row,
    // no scalar work; +inf next to -inf would say yes as well and only costs the slower path)
    float zsum = 0.0f;
#pragma unroll
    for (int r = 0; r < WN; ++r) zsum += z[r];
    const bool nan_any = __ballot(zsum != zsum) != 0;
    // last interior row / column of the window (62, less on the raster's last tiles)
    const int lr = min(FT, H - 2 - y0), lc = min(FT, W - 2 - x0);
    const bool lane_in = lane >= 1 && lane <= lc;
    // (per lane: the last row that is tile interior in this lane, 0 = none -- what depends on
    // the row is one vector compare against it, not scalar code per row)
    const int row_in = lane_in ? lr : 0;
    bool outlet = false;
#pragma unroll
    for (int r = 0; r < WN; ++r) w[r] = HDEM_INF;
    if (nan_any) {
        // pinned cells of the fill -- nodata's 8 neighbours -- are the sources of this tile.
        // All in vector registers (flags 1.0 / 0.0, three rolling rows as in check_rows): with
        // lane masks the compiler computed 64 ballots ahead of the branch and spilled them --
        // 400 instructions on every tile for a path that few tiles take.
        auto near_row = [&](float v) {
            const float n = v != v ? 1.0f : 0.0f;
            return fmaxf(fmaxf(n, lane_prev(n)), lane_next(n));
        };
        float h_prev = near_row(z[0]), h_cur = near_row(z[1]), any = 0.0f;
#pragma unroll
        for (int r = 1; r <= FT; ++r) {
            const float h_next = near_row(z[r + 1]);
            const bool src = r <= row_in && fmaxf(fmaxf(h_prev, h_cur), h_next) > 0.0f && z[r] == z[r];
            w[r] = src ? z[r] : HDEM_INF;
            any = src ? 1.0f : any;
            h_prev = h_cur;
            h_cur = h_next;
        }
        outlet = __ballot(any != 0.0f) != 0;
    }
    // everything outside the tile interior, and nodata, is a wall (fminf drops the NaN)
    z[0] = z[WN - 1] = HDEM_INF;
#pragma unroll
    for (int r = 1; r <= FT; ++r) z[r] = r <= row_in ? fminf(z[r], HDEM_INF) : HDEM_INF;
    float hz = HDEM_INF;
    if (!outlet) {
        float m = HDEM_INF;
#pragma unroll
        for (int r = 1; r <= FT; ++r) m = fminf(m, z[r]);
        hz = wave_min(m);
        const int L = __builtin_ctzll(__ballot(m == hz) | (1ull << 63));
        int rr = 0;
#pragma unroll
        for (int r = FT; r >= 1; --r) rr = z[r] == hz ? r : rr;
        rr = lane == L ? rr : -1;                   // (the first lane with the minimum, its first row)
#pragma unroll
        for (int r = 1; r <= FT; ++r) w[r] = rr == r ? hz : HDEM_INF;
    }
    float zt[WN];
#pragma unroll
    for (int r = 0; r < WN; ++r) zt[r] = z[r];
    transpose(zt, T, lane);
    float col_w = HDEM_INF, col_e = HDEM_INF;
#pragma unroll
    for (int k = 0; k < ITERS; ++k) {
        scan_lines<false>(z, w, 0.0f);
        transpose(w, T, lane);
        scan_lines<false>(zt, w, 0.0f);
        if (k == ITERS - 1) {
            col_w = w[1];
            col_e = w[FT];
            if (lc != FT) {
#pragma unroll
                for (int i = 1; i < FT; ++i) col_e = i == lc ? w[i] : col_e;
            }
        }
#pragma unroll
        for (int i = 0; i < WN; ++i) T[lane * TS + i] = w[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WN; ++i) w[i] = T[i * TS + lane];
        __syncthreads();
    }
    float row_s = w[FT];
    if (lr != FT) {
#pragma unroll
        for (int i = 1; i < FT; ++i) row_s = i == lr ? w[i] : row_s;
    }
    float *e = edge + (size_t)t * HUB_EDGE;
    e[lane] = w[1];
    e[WN + lane] = row_s;
    e[2 * WN + lane] = col_w;
    e[3 * WN + lane] = col_e;
    if (lane == 0) node[t] = outlet ? __builtin_nanf("") : hz;
    if (lane_in) {
        // (rows past the tile interior: beyond the resource's end, dropped)
#pragma unroll
        for (int r = 1; r <= FT; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(
                __builtin_bit_cast(unsigned, z[r] == HDEM_INF ? __builtin_nanf("") : w[r]), drsrc,
                col, (unsigned)r * pitch, 0);
    }
    }
}

// min over cells a of my rim line and the (up to three) cells b next to it on the other side
// of max(a, b); both lines indexed alike by lane, +inf where there is no cell
// (lanes 0 and 63 of the shifted copies read 0 under bound_ctrl; a is +inf there -- rim lines
// start at lane 1 and end at 62 -- so those two lanes drop out of the minimum by themselves.
// No `lane == 0 ? inf : lane_prev(b)` here: that puts the shift under an exec mask, and a DPP
// read of a masked-off lane returns 0.)
__device__ __forceinline__ float seam_cost(float a, float b, int lane)
{
    const float p = lane_prev(b), n = lane_next(b);
    const float edge = (lane == 0 || lane == WN - 1) ? HDEM_INF : a;
    return wave_min(fmaxf(edge, fminf(fminf(p, b), n)));
}

// The hub raster: node, east and south crossing of every tile (and the crossings to the raster
// ring for the tiles next to it).
__global__ __launch_bounds__(NT) void hub_edges_kernel(const float *__restrict__ zg,
                                                       float *__restrict__ wg, int H, int W,
                                                       int tiles_x, int tiles_y,
                                                       const float *__restrict__ edge,
                                                       const float *__restrict__ node,
                                                       float *__restrict__ cr, int ghost_top,
                                                       int ghost_bottom)
{
    const int t = blockIdx.x, lane = threadIdx.x;
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int y0 = ty * FT, x0 = tx * FT, cw = 2 * tiles_x + 1;
    const float *e = edge + (size_t)t * HUB_EDGE;
    const float mine_n = e[lane], mine_s = e[WN + lane], mine_w = e[2 * WN + lane],
                mine_e = e[3 * WN + lane];
    // ring cells next to my rim (walls where nodata or outside the raster)
    // (and W's ring <- Z: the fill's Dirichlet boundary, written by the tiles next to it)
    auto ring_col = [&](int xr) {
        const int y = y0 + lane;
        if (y > H - 1) return HDEM_INF;
        const float v = zg[(size_t)y * W + xr];
        wg[(size_t)y * W + xr] = v;
        return fminf(v, HDEM_INF);
    };
    // Row-block partition: a ghost row is a row of the neighbouring block, and the caller has
    // put that block's d there (its path costs to ITS hubs): the crossing then joins my hub to
    // the hub of the neighbour's tile with the same column range -- the two tilings share their
    // columns -- so the two cells that belong to the tile columns beside mine stay out.
    auto ring_row = [&](int yr, int ghost) {
        const int xx = x0 + lane;
        if (xx > W - 1) return HDEM_INF;
        if (ghost) {
            // (nor do its two end cells, raster ring: a crossing to them is an outlet of MY
            // tile, and this crossing stands for both hubs it joins)
            const bool out = lane == 0 || lane == WN - 1 || xx == 0 || xx == W - 1;
            return out ? HDEM_INF : fminf(wg[(size_t)yr * W + xx], HDEM_INF);
        }
        const float v = zg[(size_t)yr * W + xx];
        wg[(size_t)yr * W + xx] = v;
        return fminf(v, HDEM_INF);
    };
    auto big = [](float v) { return v < HUB_BIG ? v : HUB_BIG; };
    const float other_e = tx + 1 < tiles_x ? edge[(size_t)(t + 1) * HUB_EDGE + 2 * WN + lane]
                                           : ring_col(W - 1);
    const float other_s = ty + 1 < tiles_y ? edge[(size_t)(t + tiles_x) * HUB_EDGE + lane]
                                           : ring_row(H - 1, ghost_bottom);
    const float ce = seam_cost(mine_e, other_e, lane), cs = seam_cost(mine_s, other_s, lane);
    float cwest = HDEM_INF, cnorth = HDEM_INF;
    if (tx == 0) cwest = seam_cost(mine_w, ring_col(0), lane);
    if (ty == 0) cnorth = seam_cost(mine_n, ring_row(0, ghost_top), lane);
    if (lane == 0) {
        float *row = cr + (size_t)(2 * ty + 1) * cw + 2 * tx + 1;
        const float nz = node[t];
        row[0] = nz != nz ? nz : big(nz);
        row[1] = big(ce);
        row[cw] = big(cs);
        row[cw + 1] = HUB_BIG;
        if (tx == 0) { row[-1] = big(cwest); row[cw - 1] = HUB_BIG; }
        if (ty == 0) { row[-cw] = big(cnorth); row[-cw + 1] = HUB_BIG; }
        if (tx == 0 && ty == 0) row[-cw - 1] = HUB_BIG;
    }
}

// ROLE only names the launch (0: the raster itself, 1: the coarse pre-solve of a larger
// raster) so that profilers list the two apart.
template <bool HAS_EPS, int ROLE>
__global__ __launch_bounds__(NT, 2) void fill_async_kernel(const float *__restrict__ zg,
                                                          float *wg, int H, int W, float eps,
                                                          int tiles_x, int tiles_y, int ntiles,
                                                          int S, int *state, int *prio,
                                                          int *pend, int *error,
                                                          unsigned long long *stats,
                                                          long long budget_ticks, int soft,
                                                          int *flat, int *zmax,
                                                          const float *__restrict__ hub_lev,
                                                          int *applied,
                                                          const int *__restrict__ seam_words)
{
    __shared__ float T[WN * TS];
    // (deferred call: no tile queued and no ghost row replaced -- known only here -- is no launch)
    if (seam_words && (seam_words[0] | seam_words[1] | seam_words[2]) == 0) return;
    const int G = gridDim.x, b = blockIdx.x;
    const long long t_begin = wall_clock64();
    // Residency census.  The launch is sized so that every workgroup is resident at once on
    // an otherwise idle MI355X (8 single-wave workgroups per CU) and they start together.  When
    // the GPU is shared -- another rank's launch, RCCL's kernels -- some of them are not: the
    // others wait 200 us for them and then start without them.  Nothing depends on an owner
    // being there: its queued tiles are stolen by workgroups that have run out of their own,
    // the drain test counts tiles, not workgroups, and a workgroup that is scheduled late finds
    // nothing queued and leaves.  (Round 1 gave up here and let the host fall back to the
    // round-synchronous driver, 2-20 x slower; error[3] only records that it happened.)
    if (threadIdx.x == 0) atomicAdd(error + 1, 1);
    while (ld_relaxed(error + 1) < G && ld_relaxed(error) == 0) {
        if (wall_clock64() - t_begin > 20000) {
            if (threadIdx.x == 0) atomicExch(error + 3, 1);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (ld_relaxed(error) != 0) return;
    long long t_mark = t_begin, busy = 0, idle = 0;
    for (;;) {
        // readfirstlane: the tile index is wave-uniform; say so, or every row address
        // of the visit is computed (and kept) per lane
        // soft budget = time slice: just stop taking tiles, everything stays consistent
        if (soft && wall_clock64() - t_begin > budget_ticks) break;
        const int t = __builtin_amdgcn_readfirstlane(
            async_pick(b, G, S, ntiles, state, prio, pend, soft ? error + 2 : error, t_begin,
                       budget_ticks));
        long long now = wall_clock64();
        idle += now - t_mark;
        t_mark = now;
        if (t < 0) break;
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        // flat[t] / zmax[t] belong to whoever runs the tile; agent-scope accesses, handed on
        // with the tile's state word like the tile itself
        float flat_level = HDEM_INF;
        // (hub start) which of the nine tiles under the window have had their first visit?
        // Read BEFORE the window: a tile found visited had its cells in memory before its
        // mark; a tile found unvisited has its cells raised to its start bound, which is right
        // for d and harmless for newer values.  Rides on the round trip of the flat level.
        hub_bounds hb;
        bool use_hb = false, first_visit = false;
        const int fb_raw = HAS_EPS ? FLAT_NONE : ld_relaxed(flat + t);
        if (!HAS_EPS && hub_lev) {
            const int k9 = threadIdx.x < 9 ? (int)threadIdx.x : 4;
            const int ny = ty + k9 / 3 - 1, nx = tx + k9 % 3 - 1;
            const bool there = threadIdx.x < 9 && ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x;
            float l = -HDEM_INF;
            int seen = 1;
            if (there) {
                seen = ld_relaxed(applied + ny * tiles_x + nx);
                l = hub_lev[(size_t)(2 * ny + 1) * (2 * tiles_x + 1) + 2 * nx + 1];
            }
            // (a NaN level: the tile is an outlet of the hub graph, its d is its bound)
            l = (seen || l != l) ? -HDEM_INF : (l >= HUB_BIG ? HDEM_INF : l);
            first_visit = __builtin_amdgcn_readlane(seen, 4) == 0;
            use_hb = first_visit || __ballot(l > -HDEM_INF) != 0;
            if (use_hb) {
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    hb.lv[k] = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, l), k));
                hb.mine = first_visit;
            }
            asm volatile("" ::: "memory");      // (the window loads stay behind the marks)
        }
        if (!HAS_EPS) {
            const int fb = __builtin_amdgcn_readfirstlane(fb_raw);
            if (fb != FLAT_NONE) {
                flat_level = __builtin_bit_cast(float, fb);
                const float zm = __builtin_bit_cast(
                    float, __builtin_amdgcn_readfirstlane(ld_relaxed(zmax + t)));
                const flat_result f = flat_visit(wg, H, W, ty, tx, flat_level, zm,
                                                 use_hb ? &hb : nullptr);
                if (f.handled) {
                    if (f.changed) {
                        if (threadIdx.x == 0)
                            __hip_atomic_store(flat + t, __builtin_bit_cast(int, f.level),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        // the level travels with the tile: it has landed before the state word
                        // is released (async_finish), whoever picks the tile up next
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    async_finish(t, b, G, S, tiles_x, tiles_y, state, prio, pend, stats, f.changed,
                                 false, f.dirs, 0);
                    now = wall_clock64();
                    if (threadIdx.x == 8) {
                        stats[(size_t)b * STAT_WORDS + STAT_FLAT] += 1;
                        stats[(size_t)b * STAT_WORDS + STAT_FLAT_SAME] += f.changed ? 0 : 1;
                    }
                    busy += now - t_mark;
                    t_mark = now;
                    continue;
                }
            }
        }
        const visit_result v = tile_visit<HAS_EPS, true>(zg, wg, H, W, eps, ty, tx, T, nullptr,
                                                         flat_level, use_hb ? &hb : nullptr);
        if (first_visit) {
            // the tile's cells are in memory before its mark, the mark before its state word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0)
                __hip_atomic_store(applied + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!HAS_EPS && (v.flat_bits != FLAT_NONE || flat_level < HDEM_INF)) {
            if (threadIdx.x == 0) {
                __hip_atomic_store(flat + t, v.flat_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v.flat_bits != FLAT_NONE)
                    __hip_atomic_store(zmax + t, v.zmax_bits, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
            // (as above: flat level and zmax are in memory before the tile's state word moves)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef HDEM_VISIT_PROF
        const long long t_v = wall_clock64();
#endif
        async_finish(t, b, G, S, tiles_x, tiles_y, state, prio, pend, stats, v.changed, v.more,
                     v.dirs, v.iters);
        now = wall_clock64();
#ifdef HDEM_VISIT_PROF
        if (threadIdx.x == 8) {
            for (int k = 0; k < 6; ++k) stats[(size_t)b * STAT_WORDS + STAT_PROF + k] += v.ticks[k];
            stats[(size_t)b * STAT_WORDS + STAT_PROF + 6] += (unsigned long long)(now - t_v);   // finish
        }
#endif
        busy += now - t_mark;
        t_mark = now;
    }
    if (threadIdx.x == 0) {
        stats[(size_t)b * STAT_WORDS + 4] += (unsigned long long)busy;
        stats[(size_t)b * STAT_WORDS + 5] += (unsigned long long)idle;
    }
}

// W0: pinned cells <- Z (ring, nodata, neighbours of nodata), the rest +inf.
// One lane per 4 consecutive cells of a row: the three rows around them come in
// as one 16-byte load plus two edge cells each, the result leaves as one 16-byte
// store (8 B/cell of HBM traffic).  tile_key[t] receives the lowest pinned
// elevation next to tile t (KEY_NONE: none); pinned cells are few (ring + nodata
// fringe), so that atomic is cold.
__global__ __launch_bounds__(INIT_NT) void fill_init_kernel(const float *__restrict__ z,
                                                      float *__restrict__ w, int H, int W,
                                                      int tiles_x, int *tile_key,
                                                      int ghost_top, int ghost_bottom,
                                                      int ghost_given,
                                                      const float *__restrict__ coarse, int cw,
                                                      int shift,
                                                      const int *__restrict__ row_map,
                                                      float level_add)
{
    // A lane makes 4 columns x INIT_ROWS rows: INIT_ROWS + 2 rows of loads (all in flight
    // together) instead of 3 per output row.
    const int quads = (W + 3) / 4;
    const size_t q = (size_t)blockIdx.x * INIT_NT + threadIdx.x;
    if (q >= (size_t)((H + INIT_ROWS - 1) / INIT_ROWS) * quads) return;
    const int y0 = (int)(q / quads) * INIT_ROWS, x = (int)(q % quads) * 4;
    // rows y0-1 .. y0+INIT_ROWS, columns x-1 .. x+4 (clamped: a clamped duplicate cannot
    // add a NaN that is not already in the neighbourhood)
    float v[INIT_ROWS + 2][6];
#pragma unroll
    for (int r = 0; r < INIT_ROWS + 2; ++r) {
        const float *row = z + (size_t)min(max(y0 + r - 1, 0), H - 1) * W;
        v[r][0] = row[max(x - 1, 0)];
        if (x + 4 <= W) {
            const hdem_f4 m = hdem_ld4u(row + x);
            v[r][1] = m[0]; v[r][2] = m[1]; v[r][3] = m[2]; v[r][4] = m[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[r][1 + k] = row[min(x + k, W - 1)];
        }
        v[r][5] = row[min(x + 4, W - 1)];
    }
#pragma unroll
    for (int rr = 0; rr < INIT_ROWS; ++rr) {
        const int y = y0 + rr;
        if (y >= H) break;
        // start value of the free cells: +inf, or the filled level of the cell's block in a
        // coarse (block maximum) raster -- an upper bound of the fill (hdem_coarsen.hip); the
        // 4 cells of a lane share a block (blocks are >= 4 wide, x is a multiple of 4)
        float level = HDEM_INF;
        if (coarse) {
            level = coarse[(size_t)(row_map ? row_map[y] : (y >> shift)) * cw + (x >> shift)];
            // a nodata wall stays a wall; with a gradient the block's level is only reached
            // after up to a block's width of steps (level_add, see the host side)
            level = level >= 3.0e38f ? HDEM_INF : level + level_add;
        }
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int xx = x + k;
            const float zc = v[rr + 1][1 + k];
            // a ghost row belongs to the neighbouring row block: only its two border
            // cells are pinned; the rest waits at +inf for the first halo exchange
            const bool ghost = ((ghost_top && y == 0) || (ghost_bottom && y == H - 1)) &&
                               xx != 0 && xx != W - 1;
            bool pin = y == 0 || y == H - 1 || xx == 0 || xx >= W - 1 || zc != zc;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) pin |= v[rr + r][k + c] != v[rr + r][k + c];
            if (ghost && ghost_given) {
                // the caller's upper bound: acts like a pinned cell until the first exchange
                // (never below the terrain, whatever the caller wrote)
                const float g = xx < W ? w[(size_t)y * W + xx] : zc;
                o[k] = zc != zc ? zc : fmaxf(g, zc);
                if (zc == zc && g == g && tiles_x > 0 && xx < W) {
                    const int ty = min(max(y - 1, 0), H - 3) / FT, tx = min(max(xx - 1, 0), W - 3) / FT;
                    atomicMin(&tile_key[ty * tiles_x + tx], float_key(o[k]));
                }
            } else if (ghost) {
                o[k] = zc != zc ? zc : fmaxf(level, zc);
            } else {
                o[k] = pin ? zc : fmaxf(level, zc);
                if (pin && zc == zc && tiles_x > 0 && xx < W) {
                    // the tile whose interior is nearest (ring cells belong to no interior)
                    const int ty = min(max(y - 1, 0), H - 3) / FT, tx = min(max(xx - 1, 0), W - 3) / FT;
                    atomicMin(&tile_key[ty * tiles_x + tx], float_key(zc));
                }
            }
        }
        float *dst = w + (size_t)y * W + x;
        if (x + 4 <= W) {
            const hdem_f4 m = {o[0], o[1], o[2], o[3]};
            hdem_st4u(dst, m);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < W) dst[k] = o[k];
        }
    }
}

// Seeds a driver.  mode < 0: INIT -- a tile starts due when it or one of its 8
// neighbours holds a pinned cell (a pinned cell never changes, so it cannot wake a
// neighbour later).  mode >= 0: WARM -- all tiles (0) or the tile rows next to a
// ghost row that a halo exchange just replaced (ACT_TOP / ACT_BOTTOM bits).
// async != 0: state QUEUED + key + pend; else state = stamp and *any0 = 1.
__global__ __launch_bounds__(INIT_NT) void fill_seed_kernel(const int *__restrict__ tile_key,
                                                      int tiles_x, int tiles_y, int H, int mode,
                                                      int G, int S, int async, int *state,
                                                      int *prio, int *pend, int stamp, int *any0,
                                                      const float *__restrict__ coarse, int cw,
                                                      int shift, const int *__restrict__ row_map,
                                                      int W, const float *__restrict__ hub_lev,
                                                      const int *__restrict__ seam_words)
{
    const int t = blockIdx.x * INIT_NT + threadIdx.x;
    if (t >= tiles_x * tiles_y) return;
    if (seam_words && mode > 0) {
        // (deferred call: whether a ghost row was replaced is only known on the device)
        mode &= (seam_words[1] ? HDEM_FILL_ACT_TOP : 0) | (seam_words[2] ? HDEM_FILL_ACT_BOTTOM : 0);
        if (mode == 0) return;
    }
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    int key = KEY_NONE;
    if (mode < 0) {
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int ny = ty + dy, nx = tx + dx;
                if (ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x)
                    key = min(key, tile_key[ny * tiles_x + nx]);
            }
    } else if (mode == 0 || ((mode & HDEM_FILL_ACT_TOP) && ty == 0) ||
               ((mode & HDEM_FILL_ACT_BOTTOM) && ty == max(H - 3, 0) / FT)) {
        key = 0;
    }
    if (key == KEY_NONE) return;
    const int owner = t % G, idx = owner * S + t / G;
    if (async) {
        // (works on a fresh worklist and on one resumed from the previous slice)
        if (atomicCAS(&state[idx], ST_IDLE, ST_QUEUED) == ST_IDLE) {
            // seeds go first (queue order: oldest).  From a coarse start every tile is a
            // seed: they go in flood order, lowest coarse level first -- every workgroup
            // walks its own tiles upwards, so the whole raster is swept roughly from the
            // outlets up and a tile's first visit already finds its downstream neighbours
            // lowered.  (Keys below SEED_KEYS; wake keys are clock ticks above it.)
            int k = 0;
            if (hub_lev && mode == 0) {
                const float level = hub_lev[(size_t)(2 * ty + 1) * (2 * tiles_x + 1) + 2 * tx + 1];
                if (level == level)          // (an outlet tile goes first)
                    k = min(max((float_key(level) >> 12) + (SEED_KEYS >> 1), 0), SEED_KEYS - 1);
            } else if (coarse && mode == 0) {
                const int y = min(ty * FT + FT / 2, H - 1), x = min(tx * FT + FT / 2, W - 1);
                const float level = coarse[(size_t)(row_map ? row_map[y] : (y >> shift)) * cw +
                                           (x >> shift)];
                k = min(max((float_key(level) >> 12) + (SEED_KEYS >> 1), 0), SEED_KEYS - 1);
            }
            prio[idx] = k;
            // 64 shards and one arrival per tile: ~1k arrivals per shard at 16384^2, once
            atomicAdd(pend + (owner % PEND_SHARDS) * PEND_STRIDE, 1);
        }
    } else {
        state[idx] = stamp;
        *any0 = 1;
    }
}

// Counters left by deferred calls, summed into the head of the workspace (64-bit words from
// int CARRY_INT on: visits, unchanged) before the call that reports them adds its own.
constexpr int CARRY_INT = 8;
__global__ __launch_bounds__(INIT_NT) void fill_carry_sum_kernel(const unsigned long long *__restrict__ stats,
                                                                 int G, int *__restrict__ head)
{
    unsigned long long v = 0, u = 0;
    for (int b = threadIdx.x; b < G; b += INIT_NT) {
        v += stats[(size_t)b * STAT_WORDS + 0];
        u += stats[(size_t)b * STAT_WORDS + 2];
    }
    unsigned long long *out = reinterpret_cast<unsigned long long *>(head + CARRY_INT);
    if (v) atomicAdd(out, v);
    if (u) atomicAdd(out + 1, u);
}

// Behind a deferred launch: words[0] <- are tiles still queued (or did the launch give up)?
__global__ __launch_bounds__(NT) void fill_seam_busy_kernel(const int *__restrict__ pend,
                                                            const int *__restrict__ error,
                                                            int *__restrict__ words)
{
    int p = pend[threadIdx.x * PEND_STRIDE];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) p += __shfl_xor(p, o);
    if (threadIdx.x == 0) words[0] = (p > 0 || error[0] != 0) ? 1 : 0;
}

size_t stat_ints_of(const fill_ws &ws) { return (size_t)ws.G * STAT_WORDS * 2; }

int ensure_ws(hdem_ctx *ctx, int H, int W, int max_rounds, int G, bool *resume, fill_ws *ws,
              bool *keep_stats_io = nullptr)
{
    bool keep_stats = keep_stats_io && *keep_stats_io;
    // tiles cover the interior (rows 1..H-2, cols 1..W-2); none if there is no interior
    ws->tiles_x = W >= 3 && H >= 3 ? (W - 2 + FT - 1) / FT : 0;
    ws->tiles_y = W >= 3 && H >= 3 ? (H - 2 + FT - 1) / FT : 0;
    ws->ntiles = ws->tiles_x * ws->tiles_y;
    ws->max_rounds = max_rounds;
    ws->G = std::max(1, std::min(G, ws->ntiles));
    ws->S = std::max(1, (ws->ntiles + ws->G - 1) / ws->G);
    const size_t n = (size_t)std::max(ws->ntiles, 1), gs = (size_t)ws->G * ws->S;
    const size_t stat_ints = (size_t)ws->G * STAT_WORDS * 2;
    const size_t head = HEAD_INTS;                                 // error + pad (128 B)
    const size_t ints = head + stat_ints + PEND_SHARDS * PEND_STRIDE + n + 2 * gs +
                        (size_t)max_rounds + 32 + 3 * n;
    const size_t bytes = ints * sizeof(int);
    // (a workspace per nesting depth: the fill of a hub raster runs in the middle of the fill it
    // starts, whose workspace is set up already)
    void *&ws_buf = ctx->fill_ws[ctx->hub_depth];
    size_t &ws_bytes = ctx->fill_ws_bytes[ctx->hub_depth];
    if (ws_bytes < bytes) {
        if (ws_buf) {
            HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            HDEM_HIP_CHECK(hipFree(ws_buf));
            ws_buf = nullptr;
            ws_bytes = 0;
        }
        if (int rc = hdem_raw_alloc(ctx, bytes, &ws_buf)) return rc;
        ws_bytes = bytes;
        *resume = false;                                           // the worklist went with it
        keep_stats = false;                                        // (and the counters)
    }
    const size_t host_ints = std::max(head + stat_ints + (size_t)PEND_SHARDS * PEND_STRIDE,
                                       (size_t)max_rounds + 32);
    if (ctx->host_counts_len < host_ints) {
        if (ctx->host_counts) HDEM_HIP_CHECK(hipHostFree(ctx->host_counts));
        ctx->host_counts = nullptr;
        HDEM_HIP_CHECK(hipHostMalloc((void **)&ctx->host_counts, host_ints * sizeof(int32_t)));
        ctx->host_counts_len = host_ints;
    }
    // Layout: [error | stats | any | applied] [pend | state] [tile_key | prio] [flat | zmax] -- what is
    // zeroed sits together and what is set to 0x7f.. sits together, so that a call costs two
    // fill launches, not seven (each one is a kernel of its own, ~4 us on the stream).
    int *base = (int *)ws_buf;
    const size_t any_ints = (size_t)max_rounds + 32, pend_ints = PEND_SHARDS * PEND_STRIDE;
    ws->error = base;
    ws->stats = (unsigned long long *)(base + head);               // 8-byte aligned
    ws->any = base + head + stat_ints;
    ws->applied = ws->any + any_ints;
    ws->pend = ws->applied + n;
    ws->state = ws->pend + pend_ints;
    ws->tile_key = ws->state + gs;
    ws->prio = ws->tile_key + n;
    ws->flat = ws->prio + gs;
    ws->zmax = ws->flat + n;
    // error, stats, any = 0 and the flat levels reset (they live inside one asynchronous
    // launch and are written out behind it); a resumed worklist keeps pend / state /
    // tile_key / prio of the last slice
    const size_t zeros = head + stat_ints + any_ints + n + (*resume ? 0 : pend_ints + gs);
    if (keep_stats_io) *keep_stats_io = keep_stats;
    if (keep_stats) {
        // (the counters of deferred calls stay: the first call that waits reports them too)
        HDEM_HIP_CHECK(hipMemsetAsync(base, 0, head * sizeof(int), ctx->stream));
        HDEM_HIP_CHECK(hipMemsetAsync(base + head + stat_ints, 0,
                                      (zeros - head - stat_ints) * sizeof(int), ctx->stream));
    } else {
        HDEM_HIP_CHECK(hipMemsetAsync(base, 0, zeros * sizeof(int), ctx->stream));
    }
    int *sevens = *resume ? ws->flat : ws->tile_key;
    HDEM_HIP_CHECK(hipMemsetAsync(sevens, 0x7f, (size_t)(ws->zmax + n - sevens) * sizeof(int),
                                  ctx->stream));
    return HDEM_OK;
}

// largest |value| among the cells that are not walls (gradient fill: the rounding allowance of
// its coarse start is a multiple of the ulp up there).  Non-negative floats order like ints.
__global__ __launch_bounds__(INIT_NT) void absmax_kernel(const float *__restrict__ v, size_t n,
                                                         int *__restrict__ out)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * INIT_NT + threadIdx.x; i < n; i += (size_t)gridDim.x * INIT_NT) {
        const float a = fabsf(v[i]);
        if (a < HUB_BIG) m = fmaxf(m, a);          // (NaN compares false)
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(int, m));
}

struct hub_bufs {
    float *edge, *node, *cr, *lev;
    int ch, cw, tiles_x, tiles_y;
};

// The hub start's device buffers (kept in the context, grown on demand): rim lines and hub
// elevation of every tile, the hub raster and its filled twin.
int hub_alloc(hdem_ctx *ctx, int depth, int H, int W, hub_bufs *hb)
{
    void *&buf = ctx->hub_buf[depth];
    size_t &buf_bytes = ctx->hub_bytes[depth];
    hb->tiles_x = (W - 2 + FT - 1) / FT;
    hb->tiles_y = (H - 2 + FT - 1) / FT;
    hb->ch = 2 * hb->tiles_y + 1;
    hb->cw = 2 * hb->tiles_x + 1;
    const size_t nt = (size_t)hb->tiles_x * hb->tiles_y, cells = (size_t)hb->ch * hb->cw;
    const size_t need = (nt * (HUB_EDGE + 1) + 2 * cells) * sizeof(float);
    if (buf_bytes < need) {
        if (buf) {
            HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            HDEM_HIP_CHECK(hipFree(buf));
            buf = nullptr;
            buf_bytes = 0;
        }
        if (int rc = hdem_raw_alloc(ctx, need, &buf)) return rc;
        buf_bytes = need;
    }
    hb->edge = (float *)buf;
    hb->node = hb->edge + nt * HUB_EDGE;
    hb->cr = hb->node + nt;
    hb->lev = hb->cr + cells;
    return HDEM_OK;
}

void hub_launch_dist(hdem_ctx *ctx, const float *z, float *w, int H, int W, const hub_bufs &hb)
{
    const int ntiles = hb.tiles_x * hb.tiles_y;
    // (HDEM_HUB_ITERS / HDEM_HUB_WGS: experiments)
    const int iters = getenv("HDEM_HUB_ITERS") ? atoi(getenv("HDEM_HUB_ITERS")) : 3;
    // one workgroup per tile: a resident grid of 8 per CU walking the tiles was measured
    // at 0.77 against 0.64 ms (HDEM_HUB_WGS: that grid, per CU)
    const int grid = getenv("HDEM_HUB_WGS") ? std::min(ntiles, ctx->num_cus * atoi(getenv("HDEM_HUB_WGS")))
                                            : (ntiles + 7) / 8 * 8;     // (whole rounds of the 8 XCDs)
    hipStream_t st = ctx->stream;
    if (iters >= 4)
        hipLaunchKernelGGL(hub_dist_kernel<4>, dim3(grid), dim3(NT), 0, st, z, w, H, W, hb.tiles_x,
                           ntiles, hb.edge, hb.node);
    else if (iters == 3)
        hipLaunchKernelGGL(hub_dist_kernel<3>, dim3(grid), dim3(NT), 0, st, z, w, H, W, hb.tiles_x,
                           ntiles, hb.edge, hb.node);
    else if (iters <= 1)
        hipLaunchKernelGGL(hub_dist_kernel<1>, dim3(grid), dim3(NT), 0, st, z, w, H, W, hb.tiles_x,
                           ntiles, hb.edge, hb.node);
    else
        hipLaunchKernelGGL(hub_dist_kernel<2>, dim3(grid), dim3(NT), 0, st, z, w, H, W, hb.tiles_x,
                           ntiles, hb.edge, hb.node);
}

void hub_launch_edges(hdem_ctx *ctx, const float *z, float *w, int H, int W, int flags,
                      const hub_bufs &hb)
{
    hipLaunchKernelGGL(hub_edges_kernel, dim3(hb.tiles_x * hb.tiles_y), dim3(NT), 0, ctx->stream, z,
                       w, H, W, hb.tiles_x, hb.tiles_y, hb.edge, hb.node, hb.cr,
                       flags & HDEM_FILL_GHOST_TOP, flags & HDEM_FILL_GHOST_BOTTOM);
}

}  // namespace

// ---- hub start of a row-block partition (hydrodem_amd/partition.py) -------------------------
// The ranks build ONE hub graph: every rank makes the d of its block (prepare), swaps the seam
// rows of d with its neighbours into its ghost rows, makes its hub raster (crossings to a ghost
// row join its hubs to the neighbour's), the rasters are stacked and filled by every rank, and
// the local fill then starts from the levels of the rank's own part (set_fill_hub_levels).
extern "C" int hdem_fill_hub_prepare_dev(hdem_ctx *ctx, const float *z, int H, int W, int flags,
                                         float *w)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, w, H, W)) return rc;
    HDEM_REQUIRE(H >= 3 && W >= 3, HDEM_ERR_BAD_ARG, "hub start needs an interior, got %d x %d", H, W);
    HDEM_REQUIRE(z != w, HDEM_ERR_BAD_ARG, "sink fill cannot run in place");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    hub_bufs hb;
    if (int rc = hub_alloc(ctx, 0, H, W, &hb)) return rc;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FILL_HUB, (int64_t)H * W);
        hub_launch_dist(ctx, z, w, H, W, hb);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    ctx->hub_prep_z = z;
    ctx->hub_prep_w = w;
    ctx->hub_prep_h = H;
    ctx->hub_prep_cols = W;
    ctx->hub_prep_flags = flags & (HDEM_FILL_GHOST_TOP | HDEM_FILL_GHOST_BOTTOM);
    ctx->hub_levels_given = nullptr;
    return HDEM_OK;
}

extern "C" int hdem_fill_hub_raster_dev(hdem_ctx *ctx, float *raster)
{
    HDEM_REQUIRE(ctx && raster, HDEM_ERR_BAD_ARG, "null argument");
    HDEM_REQUIRE(ctx->hub_prep_z, HDEM_ERR_BAD_ARG, "no prepared hub start (hdem_fill_hub_prepare_dev)");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    hub_bufs hb;
    if (int rc = hub_alloc(ctx, 0, ctx->hub_prep_h, ctx->hub_prep_cols, &hb)) return rc;
    hub_launch_edges(ctx, ctx->hub_prep_z, ctx->hub_prep_w, ctx->hub_prep_h, ctx->hub_prep_cols,
                     ctx->hub_prep_flags, hb);
    HDEM_HIP_CHECK(hipGetLastError());
    HDEM_HIP_CHECK(hipMemcpyAsync(raster, hb.cr, (size_t)hb.ch * hb.cw * sizeof(float),
                                  hipMemcpyDeviceToDevice, ctx->stream));
    return HDEM_OK;
}

extern "C" int hdem_set_fill_hub_levels(hdem_ctx *ctx, const float *levels)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    HDEM_REQUIRE(!levels || ctx->hub_prep_z, HDEM_ERR_BAD_ARG,
                 "no prepared hub start (hdem_fill_hub_prepare_dev)");
    ctx->hub_levels_given = levels;
    return HDEM_OK;
}

// One seam exchange's bookkeeping in one launch: the received rows replace the ghost rows;
// words[1] / words[2] <- the top / bottom ghost row got different bits (NaN included: bit
// patterns, not values); words[3] <- any of words[0..2] (what the ranks vote on).
__global__ __launch_bounds__(INIT_NT) void fill_seam_apply_kernel(float *__restrict__ w, int H, int W,
                                                                  const float *__restrict__ recv_top,
                                                                  const float *__restrict__ recv_bot,
                                                                  int pending, int *__restrict__ words)
{
    const int x = blockIdx.x * INIT_NT + threadIdx.x;
    if (x == 0) {
        const int p = pending >= 0 ? (pending > 0) : (words[0] != 0);
        words[0] = p;
        if (p) atomicMax(words + 3, 1);
    }
    if (x >= W) return;
    int *wi = reinterpret_cast<int *>(w);
    bool top = false, bot = false;
    if (recv_top) {
        const int v = reinterpret_cast<const int *>(recv_top)[x];
        top = v != wi[x];
        wi[x] = v;
    }
    if (recv_bot) {
        const size_t o = (size_t)(H - 1) * W + x;
        const int v = reinterpret_cast<const int *>(recv_bot)[x];
        bot = v != wi[o];
        wi[o] = v;
    }
    if (__any(top)) { if ((threadIdx.x & 63) == 0) { atomicMax(words + 1, 1); atomicMax(words + 3, 1); } }
    if (__any(bot)) { if ((threadIdx.x & 63) == 0) { atomicMax(words + 2, 1); atomicMax(words + 3, 1); } }
}

extern "C" int hdem_fill_seam_apply_dev(hdem_ctx *ctx, float *w, int H, int W,
                                        const float *recv_top, const float *recv_bot,
                                        int64_t pending, int *words)
{
    HDEM_REQUIRE(ctx && w && words, HDEM_ERR_BAD_ARG, "null argument");
    HDEM_REQUIRE(H >= 1 && W >= 1, HDEM_ERR_BAD_ARG, "bad shape %d x %d", H, W);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipMemsetAsync(words + 1, 0, 3 * sizeof(int), ctx->stream));
    hipLaunchKernelGGL(fill_seam_apply_kernel, dim3((W + INIT_NT - 1) / INIT_NT), dim3(INIT_NT), 0,
                       ctx->stream, w, H, W, recv_top, recv_bot,
                       pending < 0 ? -1 : (pending > 0 ? 1 : 0), words);
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_set_fill_seam_words(hdem_ctx *ctx, int *words)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    ctx->fill_seam_words = words;
    return HDEM_OK;
}

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
    // (64 rows of a window must be addressable with 32-bit byte offsets)
    const bool use_async = !(flags & HDEM_FILL_SYNC_ONLY) && getenv("HDEM_FILL_SYNC") == nullptr &&
                           (size_t)64 * W * sizeof(float) < (size_t)0xffffffffu;
    const bool trace = getenv("HDEM_FILL_TRACE") != nullptr;
    // (taken here: the coarse pre-solve below re-enters this function)
    uint8_t *const d8_request = ctx->fill_d8;
    ctx->fill_d8 = nullptr;
    ctx->fill_d8_done = false;
    const int async_id = ctx->in_coarse_presolve ? HDEM_K_FILL_COARSE : HDEM_K_FILL_TILE;

    // a worklist can only be resumed for the problem it was built for; when it cannot,
    // every tile is due again (correct, just slower)
    const bool want_resume = (flags & HDEM_FILL_RESUME) && (flags & HDEM_FILL_WARM);
    const bool same = ctx->fill_last_h == H && ctx->fill_last_w == W && ctx->fill_last_z == z &&
                      ctx->fill_last_out == w;
    bool resume = want_resume && use_async && same && ctx->fill_resumable;
    if (want_resume) flags |= HDEM_FILL_NO_VERIFY;
    // ---- coarse start (INIT, eps = 0): the caller's coarse fill, or our own -------------
    const float *coarse = nullptr;
    const int *row_map = nullptr;
    int coarse_cw = 0, coarse_shift = 0;
    // hub start (above): the single-GPU INIT default from HUB_MIN_TILES tiles on; or the levels
    // a row-block partition worked out for this block (hdem_set_fill_hub_levels)
    const float *hub_lev = nullptr;
    hub_bufs hubs = {};
    const int hub_depth = ctx->hub_depth;
    const float *const hub_given = ctx->hub_levels_given;
    ctx->hub_levels_given = nullptr;
    if (hub_given) {
        HDEM_REQUIRE(!(flags & HDEM_FILL_WARM) && eps == 0.0f && use_async &&
                         ctx->hub_prep_z == z && ctx->hub_prep_w == w && ctx->hub_prep_h == H &&
                         ctx->hub_prep_cols == W,
                     HDEM_ERR_BAD_ARG, "hub levels were set for another fill than this one");
        ctx->hub_prep_z = nullptr;
        hub_lev = hub_given;
    } else if (!(flags & HDEM_FILL_WARM) && eps == 0.0f && !ctx->start_coarse && use_async &&
               !(flags & (HDEM_FILL_GHOST_TOP | HDEM_FILL_GHOST_BOTTOM)) &&
               // (NO_COARSE is the caller's "start from +inf"; the fill of a hub raster -- depth
               // 1 -- carries it only to keep the block-maximum start out, and may take a hub
               // start of its own, whose raster -- depth 2 -- is filled plainly)
               (hub_depth == 1 || (hub_depth == 0 && !(flags & HDEM_FILL_NO_COARSE))) &&
               !(getenv("HDEM_FILL_HUB") && atoi(getenv("HDEM_FILL_HUB")) == 0) && H >= 3 && W >= 3) {
        const int txs = (W - 2 + FT - 1) / FT, tys = (H - 2 + FT - 1) / FT;
        const int min_tiles = hub_depth ? (getenv("HDEM_HUB_MIN_TILES_NESTED")
                                               ? atoi(getenv("HDEM_HUB_MIN_TILES_NESTED"))
                                               : HUB_MIN_TILES_NESTED)
                                        : (getenv("HDEM_HUB_MIN_TILES") ? atoi(getenv("HDEM_HUB_MIN_TILES"))
                                                                        : HUB_MIN_TILES);
        if ((int64_t)txs * tys >= min_tiles) {
            if (int rc = hub_alloc(ctx, hub_depth, H, W, &hubs)) return rc;
            hub_lev = hubs.lev;
        }
    }
    float coarse_add = 0.0f;
    // (eps > 0: the block-maximum start below is valid and keeps the bits -- tested -- but the
    // launch behind it takes 14 instead of 9 ms at 16384^2: a bound that is flat inside every
    // block, under a surface that climbs cell by cell, has every tile relax many times while
    // the values from the outlets are still on their way; from +inf the tiles run in the order
    // the flood reaches them.  Off unless HDEM_FILL_EPS_COARSE is set.)
    const bool eps_coarse = eps != 0.0f && !ctx->start_coarse && getenv("HDEM_FILL_EPS_COARSE");
    if (!(flags & HDEM_FILL_WARM) && !hub_lev && (eps == 0.0f || eps_coarse)) {
        if (ctx->start_coarse) {
            coarse = ctx->start_coarse;
            row_map = ctx->start_row_map;
            coarse_cw = ctx->start_cw;
            coarse_shift = ctx->start_shift;
        } else if (use_async && !(flags & (HDEM_FILL_NO_COARSE | HDEM_FILL_GHOST_TOP |
                                           HDEM_FILL_GHOST_BOTTOM)) &&
                   (int64_t)H * W >= (getenv("HDEM_COARSE_MIN_CELLS")
                                         ? atoll(getenv("HDEM_COARSE_MIN_CELLS"))
                                         : (int64_t)COARSE_MIN_CELLS)) {
            // fill the block-maximum raster first: 1/256 of the cells, and every level in
            // it bounds the fine fill of its block from above
            const int cshift = getenv("HDEM_COARSE_SHIFT") ? atoi(getenv("HDEM_COARSE_SHIFT")) : COARSE_SHIFT;
            const int b = 1 << cshift, ch = (H + b - 1) / b, cwid = (W + b - 1) / b;
            const size_t need = ((size_t)2 * ch * cwid + 4) * sizeof(float);
            if (ctx->coarse_bytes < need) {
                if (ctx->coarse_buf) {
                    HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    HDEM_HIP_CHECK(hipFree(ctx->coarse_buf));
                    ctx->coarse_buf = nullptr;
                    ctx->coarse_bytes = 0;
                }
                if (int rc = hdem_raw_alloc(ctx, need, &ctx->coarse_buf)) return rc;
                ctx->coarse_bytes = need;
            }
            float *cz = (float *)ctx->coarse_buf, *cfill = cz + (size_t)ch * cwid;
            if (int rc = hdem_blockmax_f32_dev(ctx, z, H, W, b, cz)) return rc;
            float coarse_eps = 0.0f;
            if (eps != 0.0f) {
                // Gradient fill (eps > 0): a fine path that follows a chain of blocks makes at
                // most b steps per block of the chain, every step adds eps -- in float32: at
                // most eps + half an ulp of the value, and every value of the fill is below
                // twice the largest |elevation| + what the path adds -- so the coarse raster
                // filled with  eps_c = b (eps + ulp) + ulp  (the last ulp: its own additions
                // round too) bounds a cell of block B by  level(B) + b (eps + ulp) + 2 ulp:
                // up to b steps from the cell to the chain, one rounding of that sum.
                int *amax = (int *)(cfill + (size_t)ch * cwid);
                HDEM_HIP_CHECK(hipMemsetAsync(amax, 0, sizeof(int), ctx->stream));
                hipLaunchKernelGGL(absmax_kernel, dim3(64), dim3(INIT_NT), 0, ctx->stream, cz,
                                   (size_t)ch * cwid, amax);
                float top = 0.0f;
                HDEM_HIP_CHECK(hipMemcpyAsync(&top, amax, sizeof(float), hipMemcpyDeviceToHost,
                                              ctx->stream));
                HDEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                // (values may climb above the highest block by the gradient itself: twice the
                // top, and never less than 1, is a generous binade)
                const float span = 2.0f * std::max(top, 1.0f) + (float)(ch + cwid) * b * eps;
                const float ulp = std::nextafter(span, HDEM_INF) - span;
                coarse_eps = (float)b * (eps + ulp) + ulp;
                coarse_add = (float)b * (eps + ulp) + 2.0f * ulp;
            }
            ctx->in_coarse_presolve = true;
            const int rc = hdem_sinkfill_f32_dev(ctx, cz, ch, cwid, coarse_eps, 0,
                                                 HDEM_FILL_INIT | HDEM_FILL_NO_VERIFY |
                                                     HDEM_FILL_NO_COARSE,
                                                 cfill, nullptr);
            ctx->in_coarse_presolve = false;
            if (rc) return rc;
            coarse = cfill;
            coarse_cw = cwid;
            coarse_shift = cshift;
        }
    }
    ctx->start_coarse = nullptr;                // a caller's coarse raster is used once
    ctx->start_row_map = nullptr;
    fill_ws ws;
    // (HDEM_FILL_WGS_PER_CU: experiments and tests; the machine holds 8 of these one-wave
    // workgroups per CU -- more than 8 is a launch that cannot be resident at once)
    const int wgs_per_cu = getenv("HDEM_FILL_WGS_PER_CU") ? atoi(getenv("HDEM_FILL_WGS_PER_CU")) : 8;
    bool keep_stats = ctx->fill_stats_carry && same && !ctx->in_coarse_presolve;
    if (int rc = ensure_ws(ctx, H, W, max_rounds, ctx->num_cus * std::max(1, std::min(wgs_per_cu, 16)),
                           &resume, &ws, &keep_stats))
        return rc;
    if (!ctx->in_coarse_presolve) ctx->fill_stats_carry = false;
    if (keep_stats && !(flags & HDEM_FILL_DEFER))
        hipLaunchKernelGGL(fill_carry_sum_kernel, dim3(1), dim3(INIT_NT), 0, ctx->stream, ws.stats,
                           ws.G, ws.error);
    // a deferred call (see the header): only as a resumed correcting solve
    const bool defer = (flags & HDEM_FILL_DEFER) != 0;
    HDEM_REQUIRE(!defer || (want_resume && use_async && ctx->fill_seam_words && !d8_request),
                 HDEM_ERR_BAD_ARG,
                 "HDEM_FILL_DEFER needs WARM | RESUME, seam words and the asynchronous driver");
    // not resumable: fine if the last call on this problem left nothing queued (then the ACT
    // flags describe all there is to do), otherwise every tile is due again
    if (want_resume && !resume && !(same && ctx->fill_quiescent))
        flags &= ~(HDEM_FILL_ACT_TOP | HDEM_FILL_ACT_BOTTOM);
    ctx->fill_last_h = H;
    ctx->fill_last_w = W;
    ctx->fill_last_z = z;
    ctx->fill_last_out = w;
    ctx->fill_resumable = ctx->fill_quiescent = false;   // until this call has ended well
    hipStream_t st = ctx->stream;
    const unsigned tile_blocks = (unsigned)std::max(1, (ws.ntiles + INIT_NT - 1) / INIT_NT);
    const bool warm = (flags & HDEM_FILL_WARM) != 0;
    // (from a coarse start every tile has something to lower: all of them are due)
    int mode = warm ? (flags & (HDEM_FILL_ACT_TOP | HDEM_FILL_ACT_BOTTOM))
                    : ((coarse || hub_lev) ? 0 : -1);
    // resuming with no replaced ghost row: nothing to add to the worklist (mode 0 would
    // mean "all tiles")
    const bool seed_async = !(resume && mode == 0);
    // (a deferred call whose words are all clear has nothing to do -- unless every tile is due)
    const int *idle_words = (defer && mode != 0) ? ctx->fill_seam_words : nullptr;
    const int slice_us = (flags & HDEM_FILL_NO_VERIFY) ? ctx->fill_slice_us : 0;
    int64_t pending = 0;

    if (hub_lev && !hub_given) {
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FILL_HUB, (int64_t)H * W);
            hub_launch_dist(ctx, z, w, H, W, hubs);
            hub_launch_edges(ctx, z, w, H, W, 0, hubs);
        }
        HDEM_HIP_CHECK(hipGetLastError());
        // the hub raster is filled by this same function (in a workspace of its own)
        const bool was_presolve = ctx->in_coarse_presolve;
        ctx->in_coarse_presolve = true;
        ctx->hub_depth = hub_depth + 1;
        const int rc = hdem_sinkfill_f32_dev(ctx, hubs.cr, hubs.ch, hubs.cw, 0.0f, 0,
                                             HDEM_FILL_INIT | HDEM_FILL_NO_VERIFY |
                                                 HDEM_FILL_NO_COARSE,
                                             hubs.lev, nullptr);
        ctx->hub_depth = hub_depth;
        ctx->in_coarse_presolve = was_presolve;
        if (rc) return rc;
        if (hub_depth == 0) {                      // (the inner call wrote these for its own raster)
            ctx->fill_last_h = H;
            ctx->fill_last_w = W;
            ctx->fill_last_z = z;
            ctx->fill_last_out = w;
            ctx->fill_resumable = ctx->fill_quiescent = false;
        }
        // (no pass over the raster for max(d, level): every tile makes it on its first visit)
    } else if (hub_lev) {
        // the caller's levels: d is in w already (hdem_fill_hub_prepare_dev), the ghost rows are
        // the caller's start values
    } else if (!warm) {
        hdem_scoped_timer tm(ctx, HDEM_K_FILL_INIT, (int64_t)H * W);
        const size_t n = (size_t)((H + INIT_ROWS - 1) / INIT_ROWS) * ((W + 3) / 4);
        hipLaunchKernelGGL(fill_init_kernel, dim3((unsigned)((n + INIT_NT - 1) / INIT_NT)),
                           dim3(INIT_NT), 0, st, z, w, H, W, ws.tiles_x, ws.tile_key,
                           flags & HDEM_FILL_GHOST_TOP, flags & HDEM_FILL_GHOST_BOTTOM,
                           flags & HDEM_FILL_GHOST_GIVEN, coarse, coarse_cw, coarse_shift,
                           row_map, coarse_add);
    }
    int converged = ws.ntiles == 0 ? 1 : 0, round = 0, async_error = 0;
    bool have_counts = false;          // head words + counters already on the host
    bool d8_by_stream = false;         // the certifying stream wrote the codes (every cell)
    const bool did_async = use_async && ws.ntiles > 0;
    if (did_async) {
        // ---- asynchronous phase: does (nearly) all of the work -------------------
        if (seed_async)
            hipLaunchKernelGGL(fill_seed_kernel, dim3(tile_blocks), dim3(INIT_NT), 0, st,
                               ws.tile_key, ws.tiles_x, ws.tiles_y, H, mode, ws.G, ws.S, 1,
                               ws.state, ws.prio, ws.pend, 0, ws.any, coarse, coarse_cw,
                               coarse_shift, row_map, W, hub_lev, defer ? ctx->fill_seam_words : nullptr);
        // wall-clock budget (100 MHz ticks): generous against the ~0.15 us per tile a
        // 16384^2 fill takes, small enough that a stuck launch costs a fraction of a second;
        // or the caller's time slice (soft: the launch just stops taking tiles)
        long long budget = slice_us > 0 ? (long long)slice_us * 100ll
                                        : 20000000ll + (long long)ws.ntiles * 200ll;
        int soft = slice_us > 0;
        // (tests: cut the asynchronous phase short so that the passes behind it have work)
        // (not the pre-solve of a start raster: that one is to run as it always does)
        if (const char *tb = ctx->in_coarse_presolve ? nullptr : getenv("HDEM_FILL_TEST_BUDGET_US")) {
            budget = atoll(tb) * 100ll;
            soft = 1;
        }
        hdem_scoped_timer tm(ctx, async_id, 0);
        if (eps != 0.0f)
            hipLaunchKernelGGL((fill_async_kernel<true, 0>), dim3(ws.G), dim3(NT), 0, st, z, w, H,
                               W, eps, ws.tiles_x, ws.tiles_y, ws.ntiles, ws.S, ws.state,
                               ws.prio, ws.pend, ws.error, ws.stats, budget, soft, ws.flat, ws.zmax,
                               hub_lev, ws.applied, idle_words);
        else if (ctx->in_coarse_presolve)
            hipLaunchKernelGGL((fill_async_kernel<false, 1>), dim3(ws.G), dim3(NT), 0, st, z, w,
                               H, W, eps, ws.tiles_x, ws.tiles_y, ws.ntiles, ws.S, ws.state,
                               ws.prio, ws.pend, ws.error, ws.stats, budget, soft, ws.flat, ws.zmax,
                               hub_lev, ws.applied, idle_words);
        else
            hipLaunchKernelGGL((fill_async_kernel<false, 0>), dim3(ws.G), dim3(NT), 0, st, z, w,
                               H, W, eps, ws.tiles_x, ws.tiles_y, ws.ntiles, ws.S, ws.state,
                               ws.prio, ws.pend, ws.error, ws.stats, budget, soft, ws.flat, ws.zmax,
                               hub_lev, ws.applied, idle_words);
    }
    if (did_async && eps == 0.0f) {
        // tiles that ended the launch flat have only their edge lines in memory
        hdem_scoped_timer tm(ctx, HDEM_K_FILL_FLAT, 0);
        hipLaunchKernelGGL(flat_store_kernel, dim3(ws.ntiles), dim3(NT), 0, st, w, H, W, ws.tiles_x,
                           ws.ntiles, ws.flat, hub_lev, ws.applied, idle_words);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    // ---- round-synchronous phase: certifies (or finishes) the fixed point --------
    // behind the asynchronous phase every tile is checked once (mode 0 = all tiles);
    // on its own it starts from the same seeds
    bool verify = !(did_async && (flags & HDEM_FILL_NO_VERIFY));
    // flow directions on request (hdem_sinkfill_d8_f32_dev): written by the certifying
    // pass when that pass sees every tile, i.e. behind the asynchronous phase
    uint8_t *d8 = d8_request;
    // The coarse pre-solve needs no host round trip at all: whatever state its launch ends
    // in -- even one cut short -- is an upper bound of the coarse fill, which is all the
    // fine solve asks of it; its counters are only read when somebody is looking.
    if (ctx->in_coarse_presolve && did_async && !verify && !trace && !stats) {
        ctx->fill_resumable = ctx->fill_quiescent = false;
        return HDEM_OK;
    }
    if (defer && did_async && !verify) {
        hipLaunchKernelGGL(fill_seam_busy_kernel, dim3(1), dim3(NT), 0, st, ws.pend, ws.error,
                           ctx->fill_seam_words);
        HDEM_HIP_CHECK(hipGetLastError());
        if (stats) { *stats = hdem_fill_stats{}; stats->pending = -1; stats->tiles = ws.ntiles; stats->tile_h = stats->tile_w = FT; }
        ctx->fill_stats_carry = true;
        ctx->fill_resumable = true;          // (the worklist is whatever the launch leaves)
        ctx->fill_quiescent = false;
        return HDEM_OK;
    }
    if (!verify) {
        // trust the asynchronous phase unless it gave up; after a time slice, report how
        // many tiles are still queued (the worklist stays in the workspace for RESUME)
        // (one wait for all of it: head words, counters and -- after a slice -- the shards of
        // the pending count, which sit behind the counters' neighbours in the workspace)
        HDEM_HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ws.error,
                                      (HEAD_INTS + stat_ints_of(ws)) * sizeof(int),
                                      hipMemcpyDeviceToHost, st));
        int *host_pend = ctx->host_counts + HEAD_INTS + stat_ints_of(ws);
        if (slice_us > 0)
            HDEM_HIP_CHECK(hipMemcpyAsync(host_pend, ws.pend, PEND_SHARDS * PEND_STRIDE * sizeof(int),
                                          hipMemcpyDeviceToHost, st));
        HDEM_HIP_CHECK(hipStreamSynchronize(st));
        async_error = ctx->host_counts[0];
        if (slice_us > 0)
            for (int i = 0; i < PEND_SHARDS; ++i) pending += host_pend[i * PEND_STRIDE];
        if (async_error) { verify = true; pending = 0; }
        else { converged = pending == 0; have_counts = true; }
    }
    // (the pass must see every tile: behind the asynchronous phase, or a WARM round-driver
    // call with all tiles due -- the verifying call of the row-block loop)
    if (!(verify && (did_async || (warm && mode == 0)))) d8 = nullptr;
    // Behind the asynchronous phase -- and in the verifying call of the row-block loop --
    // the surface is normally final: certify it with one streaming pass (hdem_stencil.hip;
    // it also writes the flow directions) and only fall back to certifying rounds of tile
    // visits when that pass finds a cell to lower.
    // (HDEM_FILL_CERTIFY_ROUNDS: always the rounds.)
    const bool sees_all = did_async || (warm && mode == 0);      // (as for d8 above)
    if (ws.ntiles > 0 && verify && sees_all && !getenv("HDEM_FILL_CERTIFY_ROUNDS")) {
        int *flag = ws.error + 4;                          // (zeroed with the other head words)
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FILL_ROUND, (int64_t)H * W);
            if (int rc = hdem_certify_d8_launch(ctx, z, w, H, W, eps, d8, flag)) return rc;
        }
        // one read-back for everything the host wants to know: the head words (budget flag,
        // partial residency, the certifying pass's flag) and the per-workgroup counters sit
        // next to each other.  Normally the flag is clear and this is the call's only wait.
        HDEM_HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ws.error, (HEAD_INTS + stat_ints_of(ws)) * sizeof(int),
                                      hipMemcpyDeviceToHost, st));
        HDEM_HIP_CHECK(hipStreamSynchronize(st));
        if (ctx->host_counts[4] == 0) { converged = 1; have_counts = true; d8_by_stream = d8 != nullptr; }
    }
    if (ws.ntiles > 0 && verify && !converged)
        hipLaunchKernelGGL(fill_seed_kernel, dim3(tile_blocks), dim3(INIT_NT), 0, st,
                           ws.tile_key, ws.tiles_x, ws.tiles_y, H, did_async ? 0 : mode, ws.G,
                           ws.S, 0, ws.state, ws.prio, ws.pend, (int)ST_ROUND0, ws.any, nullptr, 0,
                           0, nullptr, W, nullptr, nullptr);
    // rounds per host check: behind the asynchronous phase the first round is expected to
    // find nothing, so only one more is queued with it (an empty launch costs ~9 us)
    const int KB = did_async ? 2 : K;
    while (ws.ntiles > 0 && verify && round < max_rounds && !converged) {
        for (int k = 0; k < KB; ++k) {
            const int r = round + k;
            hdem_scoped_timer tm(ctx, HDEM_K_FILL_ROUND, 0);
            if (eps != 0.0f)
                hipLaunchKernelGGL(fill_round_kernel<true>, dim3(ws.G), dim3(NT), 0, st, z, w, H,
                                   W, eps, ws.tiles_x, ws.tiles_y, ws.ntiles, ws.S, ws.state,
                                   ST_ROUND0 + r, ws.any + r + 1, ws.stats, d8);
            else
                hipLaunchKernelGGL(fill_round_kernel<false>, dim3(ws.G), dim3(NT), 0, st, z, w,
                                   H, W, eps, ws.tiles_x, ws.tiles_y, ws.ntiles, ws.S, ws.state,
                                   ST_ROUND0 + r, ws.any + r + 1, ws.stats, d8);
        }
        HDEM_HIP_CHECK(hipGetLastError());
        HDEM_HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ws.any + round, (KB + 1) * sizeof(int),
                                      hipMemcpyDeviceToHost, st));
        HDEM_HIP_CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < KB; ++k) {
            if (ctx->host_counts[k] == 0) { converged = 1; break; }
            ++round;
        }
        if (!converged && ctx->host_counts[KB] == 0) converged = 1;
    }
    // ---- statistics ----------------------------------------------------------------
    const size_t stat_words = (size_t)ws.G * STAT_WORDS;
    if (!have_counts) {
        HDEM_HIP_CHECK(hipMemcpyAsync(ctx->host_counts, ws.error,
                                      (HEAD_INTS + stat_ints_of(ws)) * sizeof(int),
                                      hipMemcpyDeviceToHost, st));
        HDEM_HIP_CHECK(hipStreamSynchronize(st));
    }
    const unsigned long long *hs = (const unsigned long long *)(ctx->host_counts + HEAD_INTS);
    async_error = ctx->host_counts[0];
    const int partial_residency = ctx->host_counts[3];
    unsigned long long tot[STAT_WORDS] = {};
    for (size_t i = 0; i < stat_words; ++i) tot[i % STAT_WORDS] += hs[i];
    // (visits of flat tiles touch ~500 cells, not a window: counted apart, stats->visits_flat)
    ctx->stats[async_id].units += (int64_t)(tot[0] - tot[6] - tot[STAT_FLAT]) * FT * FT;
    ctx->stats[HDEM_K_FILL_ROUND].units += (int64_t)tot[6] * FT * FT;
    if (trace)
        fprintf(stderr, "sink fill: visits %llu iterations %llu unchanged %llu requeued %llu, "
                        "flat %llu, sync rounds %d, async_error %d; async busy %.3f ms idle %.3f ms per "
                        "workgroup (G=%d)\n", tot[0], tot[1], tot[2], tot[3], tot[STAT_FLAT], round, async_error,
                tot[4] / 1e5 / ws.G, tot[5] / 1e5 / ws.G, ws.G);
#ifdef HDEM_VISIT_PROF
    if (trace && tot[0])
        fprintf(stderr, "  per-visit us (sync visits): load %.2f check %.2f zt %.2f iterate %.2f "
                        "store %.2f wake-tests %.2f finish (per visit) %.2f (changed visits %llu)\n",
                tot[9] / 100.0 / tot[0], tot[10] / 100.0 / tot[0],
                tot[11] / 100.0 / (tot[0] - tot[2] + 1), tot[12] / 100.0 / (tot[0] - tot[2] + 1),
                tot[13] / 100.0 / (tot[0] - tot[2] + 1), tot[14] / 100.0 / (tot[0] - tot[2] + 1),
                tot[15] / 100.0 / tot[0], tot[0] - tot[2]);
#endif
    if (stats) {
        stats->rounds = round;
        stats->converged = converged;
        stats->tile_visits = (int64_t)tot[0];
        stats->tiles = ws.ntiles;
        stats->tile_h = FT;
        stats->tile_w = FT;
        stats->visits_flat = (int32_t)tot[STAT_FLAT];
        stats->async_timed_out = async_error;
        stats->partial_residency = partial_residency;
        stats->flat_unchanged = (int32_t)tot[STAT_FLAT_SAME];
        stats->iterations = (int64_t)tot[1];
        stats->visits_unchanged = (int64_t)tot[2];
        stats->visits_requeued = (int64_t)tot[3];
        stats->round_visits = (int64_t)tot[6];
    }
    if (stats) {
        stats->pending = pending;
        const unsigned long long *carry = (const unsigned long long *)(ctx->host_counts + CARRY_INT);
        stats->deferred_visits = (int64_t)carry[0];
        stats->deferred_unchanged = (int64_t)carry[1];
    }
    if (!converged && pending == 0) {
        hdem_set_error("sink fill did not converge in %d rounds", max_rounds);
        return HDEM_ERR_NOT_CONVERGED;
    }
    ctx->fill_d8_done = d8 != nullptr && converged;
    ctx->fill_d8_ring_done = ctx->fill_d8_done && d8_by_stream;
    // the round driver leaves round stamps in the state words: no worklist to resume
    ctx->fill_resumable = did_async && !verify;
    ctx->fill_quiescent = converged != 0;
    return HDEM_OK;
}

// rows 0 and H-1, columns 0 and W-1 of a D8 raster: no direction on the raster ring
__global__ __launch_bounds__(INIT_NT) void d8_ring_kernel(uint8_t *d8, int H, int W)
{
    const int i = blockIdx.x * INIT_NT + threadIdx.x;
    if (i < W) { d8[i] = 0; d8[(size_t)(H - 1) * W + i] = 0; }
    if (i < H) { d8[(size_t)i * W] = 0; d8[(size_t)i * W + W - 1] = 0; }
}

extern "C" int hdem_sinkfill_d8_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, float eps,
                                        int max_rounds, int flags, float *w, uint8_t *d8,
                                        hdem_fill_stats *stats)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, d8, H, W)) return rc;
    ctx->fill_d8 = d8;
    const int rc = hdem_sinkfill_f32_dev(ctx, z, H, W, eps, max_rounds, flags, w, stats);
    ctx->fill_d8 = nullptr;
    if (rc) return rc;
    if (!ctx->fill_d8_done)              // no certifying pass over every tile: the plain kernel
        return hdem_d8_f32_dev(ctx, w, H, W, d8);
    if (!ctx->fill_d8_ring_done)        // (tile visits write tile interiors only)
        hipLaunchKernelGGL(d8_ring_kernel, dim3((std::max(H, W) + INIT_NT - 1) / INIT_NT),
                           dim3(INIT_NT), 0, ctx->stream, d8, H, W);
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_set_fill_coarse_start(hdem_ctx *ctx, const float *coarse_filled, int ch,
                                          int cw, int block, const int32_t *row_map)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (!coarse_filled) {
        ctx->start_coarse = nullptr;
        ctx->start_row_map = nullptr;
        return HDEM_OK;
    }
    HDEM_REQUIRE(ch > 0 && cw > 0 && block >= 4 && block <= 256 && (block & (block - 1)) == 0,
                 HDEM_ERR_BAD_ARG, "coarse raster %d x %d with block %d is not usable", ch, cw,
                 block);
    int shift = 0;
    while ((1 << shift) < block) ++shift;
    ctx->start_coarse = coarse_filled;
    ctx->start_row_map = row_map;
    ctx->start_cw = cw;
    ctx->start_shift = shift;
    return HDEM_OK;
}

extern "C" int hdem_set_fill_slice_us(hdem_ctx *ctx, int microseconds)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    HDEM_REQUIRE(microseconds >= 0, HDEM_ERR_BAD_ARG, "slice must be >= 0 us");
    ctx->fill_slice_us = microseconds;
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
    if (int rc = dz.alloc(ctx, bytes)) return rc;
    if (int rc = dw.alloc(ctx, bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dz.p, z, bytes)) return rc;
    int rc = hdem_sinkfill_f32_dev(ctx, (const float *)dz.p, H, W, eps, max_rounds,
                                   HDEM_FILL_INIT, (float *)dw.p, stats);
    if (rc != HDEM_OK && rc != HDEM_ERR_NOT_CONVERGED) return rc;
    if (int rc2 = hdem_memcpy_d2h(ctx, w, dw.p, bytes)) return rc2;
    return rc;
}
