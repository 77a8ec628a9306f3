// HydroSHEDS / lagoon branch: CorrectNANValues, MajorityFilter,
// TidyingLagoons, LagoonsDetection (custom_filters.py:260-317, 22-73, 564-661) and
// the SciPy morphology wrappers they use (extension_filters.py:187-345:
// binary_erosion, binary_closing, grey_dilation).  Small-window stencils on float32
// rasters and byte masks; all results are selections, comparisons and (one) small
// fixed-order float32 mean, so every kernel is bit-exact against the reference.
//
// HBM-bound except the majority vote, which is VALU-bound (a separable Boyer-Moore vote
// out of LDS, then a count pass where the vote allows the share).
#include <cstring>

#include "hdem_internal.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_STRUCT = 7;

inline dim3 grid2(int w, int h) { return dim3((unsigned)((w + NT - 1) / NT), (unsigned)h); }

// CorrectNANValues.apply (:287-317), window 3: an interior cell < 0 becomes the mean of
// its 8 neighbours that are >= 0 (NaN fails the test), summed in float32 the way
// NumPy's add.reduce does for n <= 8 -- sequentially from 0 below 8 values, as the
// 8-leaf tree for exactly 8 -- and divided in double (float32 / intp) before the cast.
__device__ __forceinline__ float correct_nan_cell(const float *__restrict__ in, int h, int w, int y,
                                                  int x, float v)
{
    if (!(v < 0.0f && y >= 1 && y < h - 1 && x >= 1 && x < w - 1)) return v;
    float a[8];
    int n = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if (dy == 0 && dx == 0) continue;
            const float t = in[(size_t)(y + dy) * w + x + dx];
            if (t >= 0.0f) a[n++] = t;
        }
    float s;
    if (n == 8) {
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    } else {
        s = 0.0f;
        for (int k = 0; k < n; ++k) s += a[k];
    }
    return n ? (float)((double)s / (double)n) : __builtin_nanf("");
}

// 4 cells per lane (unaligned 16-byte accesses); voids are rare, so the neighbour reads
// of the slow path hardly ever run.
__global__ __launch_bounds__(NT) void correct_nan_kernel(const float *__restrict__ in, int h, int w,
                                                         float *__restrict__ out)
{
    const int x = (blockIdx.x * NT + threadIdx.x) * 4, y = blockIdx.y;
    if (x >= w) return;
    const size_t at = (size_t)y * w + x;
    if (x + 4 <= w) {
        hdem_f4 v = hdem_ld4u(in + at);
        if (v[0] < 0.0f || v[1] < 0.0f || v[2] < 0.0f || v[3] < 0.0f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = correct_nan_cell(in, h, w, y, x + k, v[k]);
        }
        hdem_st4u(out + at, v);
    } else {
        for (int k = 0; x + k < w; ++k) out[at + k] = correct_nan_cell(in, h, w, y, x + k, in[at + k]);
    }
}

// Any odd window (the reference's own pipeline only uses 3): the neighbours >= 0 of the
// ws x ws window without its centre, in row-major order, summed the way NumPy sums a
// contiguous float32 vector of n < 128 values -- sequentially from 0 below 8 values; else
// eight running sums over the blocks of 8, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)),
// then the n mod 8 values left over one by one -- and divided in double.  One cell per lane;
// voids are rare.
__global__ __launch_bounds__(NT) void correct_nan_ws_kernel(const float *__restrict__ in, int h,
                                                            int w, int R, float *__restrict__ out)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const float v = in[(size_t)y * w + x];
    float res = v;
    if (v < 0.0f && y >= R && y < h - R && x >= R && x < w - R) {
        float r[8], held[8];
        int n = 0, nh = 0;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) {
                if (dy == 0 && dx == 0) continue;
                const float t = in[(size_t)(y + dy) * w + x + dx];
                if (!(t >= 0.0f)) continue;
                held[nh++] = t;
                ++n;
                if (nh == 8) {                               // a whole block of 8
                    for (int k = 0; k < 8; ++k) r[k] = n == 8 ? held[k] : r[k] + held[k];
                    nh = 0;
                }
            }
        float s = 0.0f;
        if (n >= 8) s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int k = 0; k < nh; ++k) s += held[k];
        res = n ? (float)((double)s / (double)n) : __builtin_nanf("");
    }
    out[(size_t)y * w + x] = res;
}

// MajorityFilter.apply (:44-73): the value held by more than 70 % of (ws^2 - 1) cells of
// the ws x ws window minus its corners, else 0; only centres whose window fits.  A value
// with that share is a strict majority, so a Boyer-Moore vote finds it and a count pass
// confirms it.  The vote is separable: a (candidate, votes) summary stands for "votes
// copies of candidate plus pairs of distinct values", two summaries merge into one of
// their union, so each row segment is voted once (ws steps) and shared by the ws windows
// above and below it, which merge ws row summaries each: ~2.3 ws steps per cell instead
// of ws^2.  A value present c times among n cells leaves 2c - n <= votes <= c, so the
// count pass only runs where the votes neither prove the share nor rule it out.
//
// Values are compared as bit patterns after -0 -> +0 and every NaN -> one pattern.  In
// the reference every NaN is its own Counter key and never wins; here the NaNs vote
// together, which cannot unseat a value with the share (they are at most n - need), and
// a NaN winner becomes "no majority" at the end.
constexpr int MTX = 64, MMAX = 15;
constexpr unsigned NAN_KEY = 0x7fc00000u;

__device__ __forceinline__ unsigned vote_key(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x7fffffffu) > 0x7f800000u ? NAN_KEY : b == 0x80000000u ? 0u : b;
}

// the summary (c2, v2) merged into (cand, votes)
__device__ __forceinline__ void vote_merge(unsigned c2, unsigned v2, unsigned &cand, unsigned &votes)
{
    const bool eq = c2 == cand, take = !eq && v2 > votes;
    votes = eq ? votes + v2 : __usad(votes, v2, 0u);
    cand = take ? c2 : cand;
}

// Kernel shape (64 staged rows x 64 + ws - 1 columns per 256-thread block):
//   phase 1  a thread votes WRUN = 16 consecutive segments of one staged row out of the
//            WRUN + ws - 1 keys it holds in registers (7 ds_read_b128 for 16 segments) and
//            writes their 16-bit summaries -- candidate column and votes, of the full and
//            of the inner segment -- as two 16-byte stores.  One Boyer-Moore step with
//            the K-th cell, K known at compile time: with q = (votes + K) / 2 a match is
//            q + 1 and a mismatch leaves q alone, and votes can only be 0 (the newcomer
//            becomes the candidate) when K is even: 3-4 operations per step;
//   phase 2  a thread walks down one output column with the summaries of its rows in
//            registers; the sum of the rows' votes slides (one row in, one out).  A value
//            present c times leaves at least 2 c_row - n_row votes in every row it leads
//            and no row has negative votes, so the rows' votes add up to >= 2c - n: where
//            they do not reach 2 need - n nothing has the share -- all of rough terrain;
//   phase 3  the cells that pass are few and scattered (2.7 % of the cells of the bench
//            raster, but at least one in 45 % of its wave-rows): they go to a queue in LDS
//            and are merged / counted one per lane, instead of a whole wave running the
//            merge for one lane.
// Round 1's form (11 ds_read_b32 per segment, 11 summary reads per cell, merges where
// they fell) took 1.67 ms at 16384^2; registers instead of LDS reads alone made it no
// faster (the time was in the divergent merges), the queue did: 0.97 ms.
constexpr int WTH = 64, WRUN = 16;

template <int WS, int K, int END>
struct vote_walk {
    static constexpr int COL = K < WS - 2 ? K + 1 : K == WS - 2 ? 0 : WS - 1;
    static __device__ __forceinline__ void go(const unsigned *row, unsigned &cand, unsigned &col,
                                              unsigned &q)
    {
        if (K % 2 == 0) {
            const bool fresh = q == K / 2;
            cand = fresh ? row[COL] : cand;
            col = fresh ? COL : col;
        }
        q += row[COL] == cand;
        vote_walk<WS, K + 1, END>::go(row, cand, col, q);
    }
};
template <int WS, int END>
struct vote_walk<WS, END, END> {
    static __device__ __forceinline__ void go(const unsigned *, unsigned &, unsigned &, unsigned &) {}
};

template <int WS>
__global__ __launch_bounds__(NT) void majority_walk_kernel(const float *__restrict__ in, int h,
                                                           int w, int need, float *__restrict__ out)
{
    static_assert(WS <= 15, "columns and votes are packed in 4 bits");
    constexpr int R = WS / 2, TW = MTX + 2 * R, TS = (TW + 3) / 4 * 4, ROWS = WTH - 2 * R;
    constexpr int CELLS = WS * WS - 4, RPW = (ROWS + 3) / 4;
    constexpr int NK = WRUN + WS - 1, NQ = (NK + 3) / 4;
    static_assert(3 * WRUN + NQ * 4 <= TS, "the last run reads past its row");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned s[WTH * TS];
    __shared__ __attribute__((aligned(16))) unsigned short summary[WTH * MTX];
    __shared__ unsigned short queue[ROWS * MTX];
    __shared__ unsigned queued;
    const int x0 = blockIdx.x * MTX, y0 = blockIdx.y * ROWS;
    {
        constexpr int LOADS = (TS * WTH + NT - 1) / NT;
        float v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int k = threadIdx.x + j * NT, ly = k / TS, lx = k - ly * TS;
            const int gy = y0 - R + ly, gx = x0 - R + lx;
            v[j] = (k < TS * WTH && gy >= 0 && gy < h && gx >= 0 && gx < w)
                       ? in[(size_t)gy * w + gx] : __builtin_nanf("");
        }
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int k = threadIdx.x + j * NT;
            if (k < TS * WTH) s[k] = vote_key(v[j]);
        }
    }
    __syncthreads();
    {
        const int ly = threadIdx.x >> 2, run = threadIdx.x & 3;
        unsigned keys[NQ * 4];
        const u4 *src = reinterpret_cast<const u4 *>(s + ly * TS + run * WRUN);
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            const u4 q4 = src[k];
            keys[4 * k] = q4[0]; keys[4 * k + 1] = q4[1]; keys[4 * k + 2] = q4[2]; keys[4 * k + 3] = q4[3];
        }
        unsigned packed[WRUN / 2];
#pragma unroll
        for (int i = 0; i < WRUN; ++i) {
            unsigned cand = 0, col = 0, q = 0;
            vote_walk<WS, 0, WS - 2>::go(keys + i, cand, col, q);
            const unsigned inner = col | ((2 * q - (WS - 2)) << 4);
            vote_walk<WS, WS - 2, WS>::go(keys + i, cand, col, q);
            const unsigned sum16 = col | ((2 * q - WS) << 4) | (inner << 8);
            if (i % 2 == 0) packed[i / 2] = sum16; else packed[i / 2] |= sum16 << 16;
        }
        u4 *dst = reinterpret_cast<u4 *>(summary + ly * MTX + run * WRUN);
        dst[0] = (u4){packed[0], packed[1], packed[2], packed[3]};
        dst[1] = (u4){packed[4], packed[5], packed[6], packed[7]};
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, r0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * RPW;
    const int x = x0 + lx;
    unsigned sm[RPW + WS - 1];
#pragma unroll
    for (int k = 0; k < RPW + WS - 1; ++k) sm[k] = summary[min(r0 + k, WTH - 1) * MTX + lx];
    int mid = 0;                                    // full-segment votes of rows 1 .. ws-2
#pragma unroll
    for (int dy = 1; dy < WS - 1; ++dy) mid += (sm[dy] >> 4) & 15u;
    unsigned pass = 0;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int y = y0 + r0 + i;
        const int vote_sum = mid + (int)(sm[i] >> 12) + (int)(sm[i + WS - 1] >> 12);
        const bool inside = r0 + i < ROWS && y >= R && y < h - R && x >= R && x < w - R;
        pass |= (unsigned)(inside && vote_sum >= 2 * need - CELLS) << i;
        mid += (int)((sm[i + WS - 1] >> 4) & 15u) - (int)((sm[i + 1] >> 4) & 15u);
    }
    if (threadIdx.x == 0) queued = 0;
    __syncthreads();
    if (pass) {
        unsigned at = atomicAdd(&queued, (unsigned)__builtin_popcount(pass));
        for (unsigned m = pass; m; m &= m - 1)
            queue[at++] = (unsigned short)((r0 + __builtin_ctz(m)) << 6 | lx);
    }
    if (x < w) {
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int y = y0 + r0 + i;
            if (r0 + i < ROWS && y < h && !(pass >> i & 1u)) out[(size_t)y * w + x] = 0.0f;
        }
    }
    __syncthreads();
    const unsigned total = queued;
    for (unsigned q = threadIdx.x; q < total; q += NT) {
        const int ly = queue[q] >> 6, cx = queue[q] & 63;
        const unsigned *col0 = s + ly * TS + cx;
        unsigned rows[WS];
        rows[0] = summary[ly * MTX + cx] >> 8;
#pragma unroll
        for (int dy = 1; dy < WS - 1; ++dy) rows[dy] = summary[(ly + dy) * MTX + cx] & 0xffu;
        rows[WS - 1] = summary[(ly + WS - 1) * MTX + cx] >> 8;
        unsigned cand = col0[rows[0] & 15u], votes = rows[0] >> 4;
#pragma unroll
        for (int dy = 1; dy < WS; ++dy)
            vote_merge(col0[dy * TS + (rows[dy] & 15u)], rows[dy] >> 4, cand, votes);
        unsigned result = 0;
        if (cand == NAN_KEY) {
            // no value has the share
        } else if ((int)votes >= need) {
            result = cand;                          // the summary holds `votes` copies of it
        } else if ((int)votes >= 2 * need - CELLS) {
            int count = 0;
#pragma unroll
            for (int dy = 0; dy < WS; ++dy)
#pragma unroll
                for (int dx = 0; dx < WS; ++dx) {
                    if ((dy == 0 || dy == WS - 1) && (dx == 0 || dx == WS - 1)) continue;
                    count += col0[dy * TS + dx] == cand;
                }
            if (count >= need) result = cand;
        }
        out[(size_t)(y0 + ly) * w + x0 + cx] = __uint_as_float(result);
    }
}

// scipy.ndimage binary erosion / dilation with a small centred structure, cells outside
// the array = 0 (border_value).  erode: all of in[p + s]; dilate: any of in[p - s].
struct morph_struct {
    int sh, sw;
    unsigned char bits[MAX_STRUCT * MAX_STRUCT];
};

__global__ __launch_bounds__(NT) void morph_kernel(const uint8_t *__restrict__ in, int h, int w,
                                                   morph_struct st, int dilate,
                                                   uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const int cy = st.sh / 2, cx = st.sw / 2;
    bool acc = !dilate;
    for (int sy = 0; sy < st.sh; ++sy)
        for (int sx = 0; sx < st.sw; ++sx) {
            if (!st.bits[sy * st.sw + sx]) continue;
            const int dy = sy - cy, dx = sx - cx;
            const int py = dilate ? y - dy : y + dy, px = dilate ? x - dx : x + dx;
            const bool v = py >= 0 && py < h && px >= 0 && px < w && in[(size_t)py * w + px] != 0;
            if (dilate) acc = acc || v; else acc = acc && v;
        }
    out[(size_t)y * w + x] = acc ? 1 : 0;
}

// The 3 x 3 cross (scipy's default structure) on 0/1 byte masks, 4 cells per lane as one
// 32-bit word: out = mid & up & down & left-shifted & right-shifted (little endian: byte
// i-1 is one byte shift left).  W must be a multiple of 4.
__device__ __forceinline__ unsigned to01(unsigned v)
{
    v = (v | (v >> 4)) & 0x0f0f0f0fu;
    v = (v | (v >> 2)) & 0x03030303u;
    return (v | (v >> 1)) & 0x01010101u;
}

__global__ __launch_bounds__(NT) void erode_cross4_kernel(const uint8_t *__restrict__ in, int h,
                                                          int w, uint8_t *__restrict__ out)
{
    const int q = blockIdx.x * NT + threadIdx.x, y = blockIdx.y, x = q * 4;
    if (x >= w) return;
    const unsigned *row = (const unsigned *)(in + (size_t)y * w);
    const unsigned mid = to01(row[q]);
    const unsigned up = y > 0 ? to01(((const unsigned *)(in + (size_t)(y - 1) * w))[q]) : 0u;
    const unsigned dn = y < h - 1 ? to01(((const unsigned *)(in + (size_t)(y + 1) * w))[q]) : 0u;
    const unsigned left = x > 0 ? (in[(size_t)y * w + x - 1] != 0) : 0u;
    const unsigned right = x + 4 < w ? (in[(size_t)y * w + x + 4] != 0) : 0u;
    ((unsigned *)(out + (size_t)y * w))[q] =
        mid & up & dn & ((mid << 8) | left) & ((mid >> 8) | (right << 24));
}

// 4 cells per lane for the byte <-> float point kernels (16-byte load, 4-byte store)
__global__ __launch_bounds__(NT) void nonzero_kernel(const float *__restrict__ in, size_t n,
                                                     uint8_t *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const hdem_f4 v = hdem_ld4u(in + i);
        // NaN != 0 is true, as bool(nan) is
        *(unsigned *)(out + i) = (unsigned)(v[0] != 0.0f) | ((unsigned)(v[1] != 0.0f) << 8) |
                                 ((unsigned)(v[2] != 0.0f) << 16) | ((unsigned)(v[3] != 0.0f) << 24);
    } else {
        for (size_t k = i; k < n; ++k) out[k] = in[k] != 0.0f;
    }
}

// img * mask (ProductFilter with the byte mask of ExpandFilter)
__device__ __forceinline__ int reflect(int i, int n)
{   // scipy mode='reflect': d c b a | a b c d | d c b a
    if (i >= 0 && i < n) return i;
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// scipy.ndimage.grey_dilation(size=(sy, sx)), odd sizes: maximum over the centred window,
// mode='reflect'.  64 x 32 outputs per block from an LDS tile; the reflection is resolved
// once per loaded cell.  The maximum is separable: row maxima of the tile first (sx LDS
// reads per tile cell), then sy reads down the columns.  A NaN is never "greater", so it
// only survives at the centre, like the one-pass form.  CSY / CSX > 0: sizes known at
// compile time (7 x 7 is the only size the reference uses), 0: run-time sizes.
constexpr int GTX = 64, GTY = 32;

template <int CSY, int CSX, typename T>
__global__ __launch_bounds__(NT) void grey_dilation_kernel(const T *__restrict__ in, int h,
                                                           int w, int sy_, int sx_,
                                                           T *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(8))) unsigned char tile_bytes[];
    T *tile = reinterpret_cast<T *>(tile_bytes);
    const int sy = CSY ? CSY : sy_, sx = CSX ? CSX : sx_;
    const int ry = sy / 2, rx = sx / 2, tw = GTX + 2 * rx, th = GTY + 2 * ry;
    T *rowmax = tile + tw * th;                     // th x GTX
    const int x0 = blockIdx.x * GTX, y0 = blockIdx.y * GTY;
    if (CSY && CSX) {
        // compiled-in size: all loads of the thread in flight together (a load -> ds_write
        // loop waits out one memory round trip per cell: 0.78 against 0.50 ms at 16384^2)
        constexpr int CELLS = (GTX + (CSX ? CSX : 1) - 1) * (GTY + (CSY ? CSY : 1) - 1);
        constexpr int LOADS = (CELLS + NT - 1) / NT;
        T v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int k = min((int)threadIdx.x + j * NT, CELLS - 1), ly = k / tw, lx = k - ly * tw;
            v[j] = in[(size_t)reflect(y0 - ry + ly, h) * w + reflect(x0 - rx + lx, w)];
        }
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            if ((int)threadIdx.x + j * NT < CELLS) tile[threadIdx.x + j * NT] = v[j];
    } else {
        for (int k = threadIdx.x; k < tw * th; k += NT) {
            const int ly = k / tw, lx = k - ly * tw;
            tile[k] = in[(size_t)reflect(y0 - ry + ly, h) * w + reflect(x0 - rx + lx, w)];
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % GTX;
    for (int ly = threadIdx.x / GTX; ly < th; ly += NT / GTX) {
        T m = -(T)__builtin_inff();
#pragma unroll
        for (int dx = 0; dx < sx; ++dx) {
            const T v = tile[ly * tw + lx + dx];
            m = v > m ? v : m;
        }
        rowmax[ly * GTX + lx] = m;
    }
    __syncthreads();
    for (int ly = threadIdx.x / GTX; ly < GTY; ly += NT / GTX) {
        const int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) continue;
        T m = tile[(ly + ry) * tw + lx + rx];
#pragma unroll
        for (int dy = 0; dy < sy; ++dy) {
            const T v = rowmax[(ly + dy) * GTX + lx];
            m = v > m ? v : m;
        }
        out[(size_t)y * w + x] = m;
    }
}

__global__ __launch_bounds__(NT) void positive_kernel(const float *__restrict__ in, size_t n,
                                                      uint8_t *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const hdem_f4 v = hdem_ld4u(in + i);
        *(unsigned *)(out + i) = (unsigned)(v[0] > 0.0f) | ((unsigned)(v[1] > 0.0f) << 8) |
                                 ((unsigned)(v[2] > 0.0f) << 16) | ((unsigned)(v[3] > 0.0f) << 24);
    } else {
        for (size_t k = i; k < n; ++k) out[k] = in[k] > 0.0f;
    }
}

// TidyingLagoons.apply (:564-610) in one pass: erode (img != 0) twice with the 3 x 3
// cross, ExpandFilter(7), multiply with img, 7 x 7 grey dilation -- reach 2 + 3 + 3 = 8
// cells.  A block owns 48 x 64 outputs inside a 64 x 80 tile, one wave lane per tile
// column, so a tile row of a binary stage is one 64-bit word (the wave's ballot) and the
// morphology is shifts and ands of whole rows, one thread per row:
//   two erosions by the cross = one by the radius-2 diamond (cells outside the raster 0),
//   the expand window = rows -3, +3 five wide, rows -2..+2 seven wide (square minus its
//   corners), only centres whose window fits.
// Cells outside the raster hold the reflected cell for the dilation; the expand mask is
// 0 within 3 cells of the border, so their product is img * 0 like the cell they mirror.
// positive != NULL: MaskPositives of the result as well (LagoonsDetection).
constexpr int FTX = 48, FTY = 64, FREACH = 8, FTW = 64, FTH = FTY + 2 * FREACH;
typedef unsigned long long rowbits;

__device__ __forceinline__ rowbits all3(rowbits m) { return m & (m << 1) & (m >> 1); }
__device__ __forceinline__ rowbits any5(rowbits m)
{
    return m | (m << 1) | (m >> 1) | (m << 2) | (m >> 2);
}

// lane i <- lane i -+ 1 across the wave (gfx9 wave_shr / wave_shl; the lane without a source
// reads 0 -- tile columns 0..2 and 61..63, which are never outputs)
__device__ __forceinline__ float tidy_prev(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float tidy_next(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
// max of two numbers that are not NaN, as ONE instruction (fmaxf brings a canonicalisation of
// each operand with it)
__device__ __forceinline__ float tidy_max(float a, float b)
{
    return __builtin_amdgcn_fmed3f(a, b, __builtin_inff());
}

__global__ __launch_bounds__(NT) void tidy_fused_kernel(const float *__restrict__ in, int h, int w,
                                                        float *__restrict__ out,
                                                        uint8_t *__restrict__ positive)
{
    static_assert(FTW == 64 && FTX + 2 * FREACH == FTW, "one wave lane per tile column");
    __shared__ float ti[FTH * FTW];
    __shared__ rowbits nz[FTH], eroded[FTH], expanded[FTH], nans[FTH];
    const int x0 = blockIdx.x * FTX - FREACH, y0 = blockIdx.y * FTY - FREACH;
    // (the wave index as a scalar: the rows' reflections -- a modulo each -- are then scalar code
    // instead of ~35 vector instructions per row and lane, which was half of this kernel)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int gx = x0 + lane;
    {   // all loads of the thread in flight before the first ballot waits for one
        constexpr int ROWS = FTH / (NT / 64);
        static_assert(FTH % (NT / 64) == 0, "tile rows split evenly over the waves");
        const int rx = reflect(gx, w);
        float v[ROWS];
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            v[k] = in[(size_t)reflect(y0 + wave + k * (NT / 64), h) * w + rx];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const int ly = wave + k * (NT / 64), gy = y0 + ly;
            ti[ly * FTW + lane] = v[k];
            const rowbits m = __ballot(gy >= 0 && gy < h && gx >= 0 && gx < w && v[k] != 0.0f);
            if (lane == 0) nz[ly] = m;
        }
    }
    __syncthreads();
    const int r = threadIdx.x;
    if (r < FTH) {
        rowbits e = 0;
        if (r >= 2 && r < FTH - 2) {
            const rowbits c = nz[r];
            e = nz[r - 2] & all3(nz[r - 1]) & all3(c) & (c << 2) & (c >> 2) & all3(nz[r + 1]) &
                nz[r + 2];
        }
        eroded[r] = e;
    }
    __syncthreads();
    if (r < FTH) {
        rowbits x = 0;
        const int gy = y0 + r;
        if (r >= 5 && r < FTH - 5 && gy >= 3 && gy < h - 3) {
            rowbits wide = eroded[r - 2] | eroded[r - 1] | eroded[r] | eroded[r + 1] | eroded[r + 2];
            wide = any5(wide) | (wide << 3) | (wide >> 3);
            x = wide | any5(eroded[r - 3] | eroded[r + 3]);
            const int lo = max(0, 3 - x0), hi = min(64, w - 3 - x0);      // window fits: [lo, hi)
            const rowbits upto_hi = hi >= 64 ? ~0ull : hi <= 0 ? 0ull : (1ull << hi) - 1;
            const rowbits below_lo = lo >= 64 ? ~0ull : (1ull << lo) - 1;
            x &= upto_hi & ~below_lo;
        }
        expanded[r] = x;
    }
    __syncthreads();
    // Product and row maxima.  The kernel is bound by its vector instructions (SQ counters,
    // round 3: 511 M per launch at 16384^2 = the 0.87 ms it took), so the two running maxima are
    // written for few of them: a NaN is never "greater" and only survives at the centre, which
    // nans[] remembers -- it becomes -inf here and plain maxima do the rest;
    //   rows: the 7-wide maximum as three doublings of wave shifts ([-1..1], [-2..2], [-3..3]:
    //         6 maxima on registers instead of 7 clamped LDS reads and 7 compare-selects);
    //   columns: m2[k] = max(x[k], x[k+1]), m4[k] = max(m2[k], m2[k+2]), m7[k] = max(m4[k],
    //         m4[k+3]) over the 22 row maxima a lane needs for its 16 outputs: 3.5 maxima per
    //         output instead of 14 operations.
    for (int ly = wave; ly < FTH; ly += NT / 64) {
        float *row = ti + ly * FTW;
        const float p = row[lane] * ((expanded[ly] >> lane) & 1 ? 1.0f : 0.0f);
        const bool gone = p != p;
        const rowbits isnan = __ballot(gone);
        if (lane == 0) nans[ly] = isnan;
        const float q = gone ? -__builtin_inff() : p;
        const float a = tidy_max(tidy_max(q, tidy_prev(q)), tidy_next(q));
        const float b = tidy_max(tidy_prev(a), tidy_next(a));
        row[lane] = tidy_max(tidy_prev(b), tidy_next(b));
    }
    __syncthreads();
    if (lane < FREACH || lane >= FTW - FREACH || gx >= w) return;
    // each wave takes FTY / 4 consecutive rows
    constexpr int PER = FTY / (NT / 64);
    const int first = FREACH + wave * PER;
    float x[PER + 6];
#pragma unroll
    for (int k = 0; k < PER + 6; ++k) x[k] = ti[(first - 3 + k) * FTW + lane];
#pragma unroll
    for (int k = 0; k < PER + 5; ++k) x[k] = tidy_max(x[k], x[k + 1]);          // rows k .. k+1
#pragma unroll
    for (int k = 0; k < PER + 3; ++k) x[k] = tidy_max(x[k], x[k + 2]);          // rows k .. k+3
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int ly = first + k, gy = y0 + ly;
        if (gy >= h) break;
        const float m7 = tidy_max(x[k], x[k + 3]);                             // rows k .. k+6
        const float m = (nans[ly] >> lane) & 1 ? __builtin_nanf("") : m7;
        out[(size_t)gy * w + gx] = m;
        if (positive) positive[(size_t)gy * w + gx] = m > 0.0f;
    }
}

int window_ok(int window, int h, int w)
{
    if (window > h || window > w) {
        hdem_set_error("Window size: %d cannot be higher than grid dimensions: (%d, %d)", window,
                       h, w);
        return HDEM_ERR_WINDOW_HIGH;
    }
    if (window % 2 != 1) {
        hdem_set_error("Window size: %d cannot be an even number", window);
        return HDEM_ERR_WINDOW_EVEN;
    }
    return HDEM_OK;
}

int make_struct(const uint8_t *structure, int sh, int sw, morph_struct *st)
{
    HDEM_REQUIRE(sh >= 1 && sw >= 1 && sh <= MAX_STRUCT && sw <= MAX_STRUCT && (sh & 1) && (sw & 1),
                 HDEM_ERR_BAD_ARG, "structure must be odd-sized, at most %d x %d, got %d x %d",
                 MAX_STRUCT, MAX_STRUCT, sh, sw);
    st->sh = sh;
    st->sw = sw;
    for (int i = 0; i < sh * sw; ++i) st->bits[i] = structure[i] ? 1 : 0;
    return HDEM_OK;
}

const uint8_t CROSS[9] = {0, 1, 0, 1, 1, 1, 0, 1, 0};     // generate_binary_structure(2, 1)

int erode_n(hdem_ctx *ctx, const uint8_t *in, int h, int w, const morph_struct &st, int iterations,
            uint8_t *tmp, uint8_t *out)
{
    // ping-pong so that the last iteration lands in `out`
    const uint8_t *src = in;
    for (int it = 0; it < iterations; ++it) {
        uint8_t *dst = ((iterations - it) & 1) ? out : tmp;
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)h * w);
        const bool cross = st.sh == 3 && st.sw == 3 && !memcmp(st.bits, CROSS, 9);
        if (cross && w % 4 == 0 && ((uintptr_t)src | (uintptr_t)dst) % 4 == 0)
            hipLaunchKernelGGL(erode_cross4_kernel, grid2(w / 4, h), dim3(NT), 0, ctx->stream, src,
                               h, w, dst);
        else
            hipLaunchKernelGGL(morph_kernel, grid2(w, h), dim3(NT), 0, ctx->stream, src, h, w, st,
                               0, dst);
        src = dst;
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int hdem_correct_nan_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W, int window,
                                        float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    HDEM_REQUIRE(dem != out, HDEM_ERR_BAD_ARG, "the NaN correction cannot run in place");
    HDEM_REQUIRE(window >= 3, HDEM_ERR_BAD_ARG, "window must be >= 3, got %d", window);
    if (int rc = window_ok(window, H, W)) return rc;
    // (NumPy's sum changes shape again at 128 values: windows up to 11)
    HDEM_REQUIRE(window <= 11, HDEM_ERR_BAD_ARG, "NaN correction window must be 3..11, got %d",
                 window);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W);
        if (window == 3)
            hipLaunchKernelGGL(correct_nan_kernel, grid2((W + 3) / 4, H), dim3(NT), 0, ctx->stream,
                               dem, H, W, out);
        else
            hipLaunchKernelGGL(correct_nan_ws_kernel, grid2(W, H), dim3(NT), 0, ctx->stream, dem, H,
                               W, window / 2, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_majority_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, int window,
                                     float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "the majority filter cannot run in place");
    if (int rc = window_ok(window, H, W)) return rc;
    HDEM_REQUIRE(window >= 3 && window <= MMAX, HDEM_ERR_BAD_ARG,
                 "majority window must be 3..%d, got %d", MMAX, window);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    // count > (ws^2 - 1) * 0.7, in the double arithmetic of the reference (:71)
    const double thr = (double)(window * window - 1) * 0.7;
    int need = (int)thr;
    while ((double)need <= thr) ++need;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_MAJORITY, (int64_t)H * W);
        switch (window) {
#define HDEM_MAJORITY(WS_)                                                                     \
    case WS_:                                                                                  \
        hipLaunchKernelGGL(majority_walk_kernel<WS_>,                                          \
                           dim3((W + MTX - 1) / MTX,                                           \
                                (H + WTH - 2 * (WS_ / 2) - 1) / (WTH - 2 * (WS_ / 2))),        \
                           dim3(NT), 0, ctx->stream, img, H, W, need, out);                    \
        break;
            HDEM_MAJORITY(3) HDEM_MAJORITY(5) HDEM_MAJORITY(7) HDEM_MAJORITY(9)
            HDEM_MAJORITY(11) HDEM_MAJORITY(13) HDEM_MAJORITY(15)
#undef HDEM_MAJORITY
        }
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_binary_erosion_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int H, int W,
                                          const uint8_t *structure, int sh, int sw, int iterations,
                                          uint8_t *tmp, uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(mask, out, H, W)) return rc;
    HDEM_REQUIRE(iterations >= 1, HDEM_ERR_BAD_ARG, "iterations must be >= 1, got %d", iterations);
    HDEM_REQUIRE(mask != out && (iterations == 1 || (tmp && tmp != out && tmp != mask)),
                 HDEM_ERR_BAD_ARG, "erosion needs distinct in / tmp / out buffers");
    morph_struct st;
    if (int rc = make_struct(structure ? structure : CROSS, structure ? sh : 3, structure ? sw : 3,
                             &st))
        return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    return erode_n(ctx, mask, H, W, st, iterations, tmp, out);
}

extern "C" int hdem_binary_closing_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int H, int W,
                                          const uint8_t *structure, int sh, int sw, uint8_t *tmp,
                                          uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(mask, out, H, W)) return rc;
    HDEM_REQUIRE(tmp && mask != out && tmp != out && tmp != mask, HDEM_ERR_BAD_ARG,
                 "closing needs distinct in / tmp / out buffers");
    morph_struct st;
    if (int rc = make_struct(structure ? structure : CROSS, structure ? sh : 3, structure ? sw : 3,
                             &st))
        return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W * 2);
        hipLaunchKernelGGL(morph_kernel, grid2(W, H), dim3(NT), 0, ctx->stream, mask, H, W, st, 1,
                           tmp);
        hipLaunchKernelGGL(morph_kernel, grid2(W, H), dim3(NT), 0, ctx->stream,
                           (const uint8_t *)tmp, H, W, st, 0, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

template <typename T>
static int grey_dilation_dev(hdem_ctx *ctx, const T *img, int H, int W, int sy, int sx, T *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "grey dilation cannot run in place");
    HDEM_REQUIRE(sy >= 1 && sx >= 1 && (sy & 1) && (sx & 1) && sy <= 31 && sx <= 31,
                 HDEM_ERR_BAD_ARG, "grey dilation size must be odd and <= 31, got (%d, %d)", sy, sx);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W);
        const dim3 grid((W + GTX - 1) / GTX, (H + GTY - 1) / GTY);
        const size_t lds = (size_t)(GTX + sx - 1 + GTX) * (GTY + sy - 1) * sizeof(T);
        if (sy == 7 && sx == 7)
            hipLaunchKernelGGL((grey_dilation_kernel<7, 7, T>), grid, dim3(NT), lds, ctx->stream, img,
                               H, W, sy, sx, out);
        else
            hipLaunchKernelGGL((grey_dilation_kernel<0, 0, T>), grid, dim3(NT), lds, ctx->stream, img,
                               H, W, sy, sx, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_grey_dilation_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, int sy,
                                          int sx, float *out)
{ return grey_dilation_dev<float>(ctx, img, H, W, sy, sx, out); }
// (float64 rasters keep their values: scipy.ndimage.grey_dilation works in the input's type,
// extension_filters.py:345)
extern "C" int hdem_grey_dilation_f64_dev(hdem_ctx *ctx, const double *img, int H, int W, int sy,
                                          int sx, double *out)
{ return grey_dilation_dev<double>(ctx, img, H, W, sy, sx, out); }

static size_t round16(size_t n) { return (n + 15) / 16 * 16; }

static int tidying(hdem_ctx *ctx, const float *img, int H, int W, float *out, uint8_t *positive)
{
    hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W);
    hipLaunchKernelGGL(tidy_fused_kernel, dim3((W + FTX - 1) / FTX, (H + FTY - 1) / FTY), dim3(NT),
                       0, ctx->stream, img, H, W, out, positive);
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_tidying_lagoons_f32_dev(hdem_ctx *ctx, const float *img, int H, int W,
                                            float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "tidying cannot run in place");
    if (int rc = window_ok(7, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    return tidying(ctx, img, H, W, out, nullptr);
}

// LagoonsDetection.apply (:613-661): CorrectNANValues -> MajorityFilter(11) ->
// TidyingLagoons -> MaskPositives.  fixed / values are the intermediate results the
// reference keeps (hsheds_nan_fixed, lagoons_values); either may be NULL.
extern "C" int hdem_lagoons_detection_f32_dev(hdem_ctx *ctx, const float *hsheds, int H, int W,
                                              float *fixed, float *values, uint8_t *mask)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(hsheds, mask, H, W)) return rc;
    if (int rc = window_ok(11, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W, fbytes = round16(n * sizeof(float));
    char *scratch = (char *)hdem_arena(ctx, 3 * fbytes);
    if (!scratch) return HDEM_ERR_OOM;
    float *major = (float *)scratch;
    if (!fixed) fixed = (float *)(scratch + fbytes);
    if (!values) values = (float *)(scratch + 2 * fbytes);
    if (int rc = hdem_correct_nan_f32_dev(ctx, hsheds, H, W, 3, fixed)) return rc;
    if (int rc = hdem_majority_f32_dev(ctx, fixed, H, W, 11, major)) return rc;
    if (int rc = tidying(ctx, major, H, W, values, mask)) return rc;
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}
