// Radius-1 stencils of the path: D8 flow direction (A2) and the 3x3 box mean
// with optional round-half-even (A5), plus the general small-kernel Convolve.
//
// Shape shared by the radius-1 kernels (gfx950):
//   * one 256-thread workgroup (4 waves) per 256 x 32 cell tile;
//   * the tile plus a one-cell halo is staged in LDS by coalesced 16-byte row
//     loads (a wave moves 1 KiB of one raster row per instruction); the halo
//     columns sit at LDS columns 3 and 4+TW so the interior stays 16-byte
//     aligned for ds_read_b128;
//   * a wave owns 8 consecutive rows x 256 columns; each lane produces 4
//     adjacent cells per row from three rolling rows held in registers
//     (1 ds_read_b128 + 2 ds_read_b32 per new row), and stores its 4 results
//     as one 4-byte (D8) or 16-byte (box mean) word.
// HBM traffic is the algorithmic 5 B/cell (D8) or 8 B/cell (box mean) plus
// the halo: (34*258)/(32*256) = 1.07x on the read side.
#include "hdem_internal.h"

namespace {

constexpr int TW = 256;          // tile width  (cells)
constexpr int TH = 32;           // tile height (cells), 4-byte elements
// 8-byte elements use half the rows so the tile stays under 64 KiB of LDS
template <typename T> struct tile_rows { static constexpr int value = sizeof(T) == 8 ? TH / 2 : TH; };
constexpr int LS = TW + 8;       // LDS row stride in elements, interior at col 4
constexpr int NT = 256;          // threads per workgroup

__device__ __forceinline__ int reflect(int i, int n)
{   // SciPy 'reflect': d c b a | a b c d | d c b a
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - 1 - i;
    }
    return i;
}

// Stage rows y0-1..y0+TH, cols x0-1..x0+TW of `src` into `t`.
// REFLECT: out-of-raster cells mirror (box mean); otherwise they read 0 (D8:
// only border cells see them and those are forced to code 0).
template <typename T, bool REFLECT>
__device__ __forceinline__ void stage_tile(const T *__restrict__ src, int H, int W,
                                           int y0, int x0, T *t)
{
    constexpr int THT = tile_rows<T>::value;
    constexpr int V = 16 / sizeof(T);            // elements per 16-byte load
    constexpr int CPR = TW / V;                  // chunks per row
    typedef T vec_t __attribute__((ext_vector_type(V)));
    typedef T vecu_t __attribute__((ext_vector_type(V), aligned(sizeof(T))));
    const int tid = threadIdx.x;
    // all of the thread's row chunks are requested before the first one is stored: a
    // load -> ds_write loop waits out one memory round trip per iteration
    constexpr int CHUNKS = ((THT + 2) * CPR + NT - 1) / NT;
    vec_t v[CHUNKS];
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
        const int i = tid + j * NT;
        const int r = i / CPR, c = (i % CPR) * V;
        int gy = y0 - 1 + r;
        const int gx = x0 + c;
        const bool row_ok = gy >= 0 && gy < H;
        if (REFLECT) gy = reflect(gy, H);
        if (i >= (THT + 2) * CPR) {
            v[j] = vec_t(T(0));
        } else if ((row_ok || REFLECT) && gx + V <= W) {
            v[j] = *reinterpret_cast<const vecu_t *>(src + (size_t)gy * W + gx);
        } else {
            for (int k = 0; k < V; ++k) {
                int xx = gx + k;
                T e = T(0);
                if (REFLECT) e = src[(size_t)gy * W + reflect(xx, W)];
                else if (row_ok && xx < W) e = src[(size_t)gy * W + xx];
                v[j][k] = e;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CHUNKS; ++j) {
        const int i = tid + j * NT;
        if (i < (THT + 2) * CPR)
            *reinterpret_cast<vec_t *>(t + (i / CPR) * LS + 4 + (i % CPR) * V) = v[j];
    }
    for (int i = tid; i < (THT + 2) * 2; i += NT) {
        int r = i >> 1, side = i & 1;
        int gy = y0 - 1 + r, gx = side ? x0 + TW : x0 - 1;
        T e = T(0);
        if (REFLECT) e = src[(size_t)reflect(gy, H) * W + reflect(gx, W)];
        else if (gy >= 0 && gy < H && gx >= 0 && gx < W) e = src[(size_t)gy * W + gx];
        t[r * LS + (side ? 4 + TW : 3)] = e;
    }
}

// one LDS row -> 6 consecutive values (cols 4*cg-1 .. 4*cg+4 of the tile)
template <typename T>
__device__ __forceinline__ void read_row6(const T *t, int r, int cg, T (&o)[6])
{
    const T *p = t + r * LS + 4 + 4 * cg;
    o[0] = p[-1];
    o[1] = p[0]; o[2] = p[1]; o[3] = p[2]; o[4] = p[3];
    o[5] = p[4];
}

// ---------------------------------------------------------------------------
// A2  D8 flow direction.  Window order NW,N,NE,W,E,SW,S,SE; ESRI codes
// 32,64,128,16,1,8,4,2; drop = (zc - zk) * wk evaluated in float32 with one
// rounding per operation (the library is built with -ffp-contract=off);
// strict '>' keeps the first maximum; NaN never compares greater.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned d8_code(float nw, float n, float ne, float w,
                                            float c, float e, float sw, float s,
                                            float se)
{
    const float dg = 0.70710678f;
    float best = 0.0f;
    unsigned code = 0;
    float d;
    d = (c - nw) * dg; if (d > best) { best = d; code = 32; }
    d = (c - n);       if (d > best) { best = d; code = 64; }
    d = (c - ne) * dg; if (d > best) { best = d; code = 128; }
    d = (c - w);       if (d > best) { best = d; code = 16; }
    d = (c - e);       if (d > best) { best = d; code = 1; }
    d = (c - sw) * dg; if (d > best) { best = d; code = 8; }
    d = (c - s);       if (d > best) { best = d; code = 4; }
    d = (c - se) * dg; if (d > best) { best = d; code = 2; }
    return code;
}

__global__ __launch_bounds__(NT) void d8_kernel(const float *__restrict__ z, int H,
                                               int W, uint8_t *__restrict__ out,
                                               int tiles_x, int vec_store)
{
    __shared__ __attribute__((aligned(16))) float t[(TH + 2) * LS];
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x0 = bx * TW, y0 = by * TH;
    stage_tile<float, false>(z, H, W, y0, x0, t);
    __syncthreads();

    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int x = x0 + 4 * cg;
    if (x >= W) return;
    float a[6], b[6], c[6];
    read_row6(t, rg * 8 + 0, cg, a);
    read_row6(t, rg * 8 + 1, cg, b);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int y = y0 + rg * 8 + rr;
        read_row6(t, rg * 8 + rr + 2, cg, c);
        if (y < H) {
            unsigned code[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned cd = d8_code(a[k], a[k + 1], a[k + 2], b[k], b[k + 1],
                                      b[k + 2], c[k], c[k + 1], c[k + 2]);
                int xx = x + k;
                bool interior = y > 0 && y < H - 1 && xx > 0 && xx < W - 1;
                code[k] = interior ? cd : 0u;
            }
            uint8_t *o = out + (size_t)y * W + x;
            if (vec_store && x + 4 <= W) {
                *reinterpret_cast<uint32_t *>(o) =
                    code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k < W) o[k] = (uint8_t)code[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[k] = b[k]; b[k] = c[k]; }
    }
}

// ---------------------------------------------------------------------------
// Certifying pass of the sink fill as a stream (hdem_sinkfill.hip runs it behind the
// asynchronous phase): W is at the fixed point iff no interior cell can be lowered,
//     med3(z, w, min over the 3 x 3 window of w (+ eps)) == w      (NaN = the wall +inf),
// the test the tile visit's check_rows makes, here without tiles or a worklist: the
// d8_kernel shape (W staged in LDS, three rolling rows), one 16-byte load of Z per lane
// and row on top, and the flow directions of the certified surface on request.  *flag
// is set when some cell could still be lowered; the caller then falls back to the round
// driver.  9 B/cell with directions, 8 without.
// ---------------------------------------------------------------------------
template <bool HAS_EPS>
__global__ __launch_bounds__(NT) void certify_d8_kernel(const float *__restrict__ z,
                                                       const float *__restrict__ w, int H, int W,
                                                       float eps, uint8_t *__restrict__ out,
                                                       int tiles_x, int vec_store, int *flag)
{
    __shared__ __attribute__((aligned(16))) float t[(TH + 2) * LS];
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x0 = bx * TW, y0 = by * TH;
    stage_tile<float, false>(w, H, W, y0, x0, t);
    __syncthreads();

    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int x = x0 + 4 * cg;
    if (x >= W) return;
    // the lane's Z: 8 rows x 4 cells, requested before the first is used
    float zc[8][4];
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int y = min(y0 + rg * 8 + rr, H - 1);
        if (x + 4 <= W) {
            const hdem_f4 m = hdem_ld4u(z + (size_t)y * W + x);
            zc[rr][0] = m[0]; zc[rr][1] = m[1]; zc[rr][2] = m[2]; zc[rr][3] = m[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) zc[rr][k] = z[(size_t)y * W + min(x + k, W - 1)];
        }
    }
    bool lower = false;
    float a[6], b[6], c[6];
    read_row6(t, rg * 8 + 0, cg, a);
    read_row6(t, rg * 8 + 1, cg, b);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int y = y0 + rg * 8 + rr;
        read_row6(t, rg * 8 + rr + 2, cg, c);
        if (y < H) {
            unsigned code[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x + k;
                const bool interior = y > 0 && y < H - 1 && xx > 0 && xx < W - 1;
                // fminf skips a NaN operand: nodata neighbours are walls
                float m = fminf(fminf(fminf(a[k], a[k + 1]), fminf(a[k + 2], b[k])),
                                fminf(fminf(b[k + 1], b[k + 2]), fminf(c[k], fminf(c[k + 1], c[k + 2]))));
                m = fminf(m, HDEM_INF);
                if (HAS_EPS) m += eps;
                const float wc = fminf(b[k + 1], HDEM_INF), zz = fminf(zc[rr][k], HDEM_INF);
                lower |= interior && __builtin_amdgcn_fmed3f(zz, wc, m) != wc;
                const unsigned cd = d8_code(a[k], a[k + 1], a[k + 2], b[k], b[k + 1], b[k + 2], c[k],
                                            c[k + 1], c[k + 2]);
                code[k] = interior ? cd : 0u;
            }
            if (out) {
                uint8_t *o = out + (size_t)y * W + x;
                if (vec_store && x + 4 <= W) {
                    *reinterpret_cast<uint32_t *>(o) =
                        code[0] | (code[1] << 8) | (code[2] << 16) | (code[3] << 24);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) o[k] = (uint8_t)code[k];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[k] = b[k]; b[k] = c[k]; }
    }
    if (lower) *flag = 1;                           // (every writer stores the same value)
}

// ---------------------------------------------------------------------------
// A5  3x3 box mean (+ round).  Reference arithmetic: SciPy accumulates the
// nine window values in double in raster order starting from 0, casts the sum
// to the array dtype; the filter divides by 9 in that dtype
// (extension_filters.py:183-184) and np.around rounds half to even.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T mean9(const T (&a)[6], const T (&b)[6], const T (&c)[6],
                                   int k, int do_round)
{
    double acc = 0.0;
    acc += (double)a[k]; acc += (double)a[k + 1]; acc += (double)a[k + 2];
    acc += (double)b[k]; acc += (double)b[k + 1]; acc += (double)b[k + 2];
    acc += (double)c[k]; acc += (double)c[k + 1]; acc += (double)c[k + 2];
    T s = (T)acc;
    T m = s / T(9);
    if (do_round) m = sizeof(T) == 4 ? (T)rintf((float)m) : (T)rint((double)m);
    return m;
}

template <typename T>
__global__ __launch_bounds__(NT) void boxmean3_kernel(const T *__restrict__ x, int H,
                                                     int W, T *__restrict__ out,
                                                     int tiles_x, int do_round)
{
    constexpr int THT = tile_rows<T>::value;
    constexpr int RPW = THT / 4;                 // rows per wave
    __shared__ __attribute__((aligned(16))) T t[(THT + 2) * LS];
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x0 = bx * TW, y0 = by * THT;
    stage_tile<T, true>(x, H, W, y0, x0, t);
    __syncthreads();

    const int cg = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int xc = x0 + 4 * cg;
    if (xc >= W) return;
    T a[6], b[6], c[6];
    read_row6(t, rg * RPW + 0, cg, a);
    read_row6(t, rg * RPW + 1, cg, b);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int y = y0 + rg * RPW + rr;
        read_row6(t, rg * RPW + rr + 2, cg, c);
        if (y < H) {
            T m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = mean9<T>(a, b, c, k, do_round);
            T *o = out + (size_t)y * W + xc;
            if (xc + 4 <= W) {
                if (sizeof(T) == 4) {
                    hdem_f4 v = {(float)m[0], (float)m[1], (float)m[2], (float)m[3]};
                    hdem_st4u(reinterpret_cast<float *>(o), v);
                } else {
                    o[0] = m[0]; o[1] = m[1]; o[2] = m[2]; o[3] = m[3];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (xc + k < W) o[k] = m[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[k] = b[k]; b[k] = c[k]; }
    }
}

// ---------------------------------------------------------------------------
// General Convolve(weights): scipy.ndimage.convolve == correlation with the
// flipped weights, footprint visited in raster order, double accumulator,
// cast to float32, then / weights.size.  Small kernels only (<= 15 x 15);
// one thread per cell, L2-served re-reads -- this is the generic path, the
// 3x3 ones() default takes boxmean3_kernel.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void convolve_kernel(const T *__restrict__ x, int H,
                                                     int W, const double *__restrict__ wt,
                                                     int kh, int kw, T *__restrict__ out)
{
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= (size_t)H * W) return;
    int y = (int)(i / W), xx = (int)(i % W);
    int py = kh / 2, px = kw / 2;
    double acc = 0.0;
    for (int j = 0; j < kh; ++j) {
        int gy = reflect(y + j - py, H);
        for (int k = 0; k < kw; ++k) {
            double w = wt[(kh - 1 - j) * kw + (kw - 1 - k)];
            if (w != 0.0)
                acc += (double)x[(size_t)gy * W + reflect(xx + k - px, W)] * w;
        }
    }
    // SciPy writes the double sum in the array's type; the reference then divides by
    // weights.size in that type (extension_filters.py:183)
    const T s = (T)acc;
    out[i] = s / (T)(kh * kw);
}

template <typename T>
__global__ __launch_bounds__(NT) void around_kernel(const T *x, int64_t n, T *out)
{
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) out[i] = sizeof(T) == 4 ? (T)rintf((float)x[i]) : (T)rint((double)x[i]);
}

inline int tiles_of(int H, int W, int th, int *tx)
{
    *tx = (W + TW - 1) / TW;
    return *tx * ((H + th - 1) / th);
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
// (internal: hdem_sinkfill.hip) one streaming certification of W against Z, with the D8
// codes of W when d8 != NULL; *flag (device, zeroed by the caller) != 0 afterwards: not a
// fixed point yet
int hdem_certify_d8_launch(hdem_ctx *ctx, const float *z, const float *w, int H, int W, float eps,
                           uint8_t *d8, int *flag)
{
    int tx, nt = tiles_of(H, W, TH, &tx);
    const int vec = d8 && (W % 4 == 0) && ((uintptr_t)d8 % 4 == 0);
    if (eps != 0.0f)
        hipLaunchKernelGGL(certify_d8_kernel<true>, dim3(nt), dim3(NT), 0, ctx->stream, z, w, H, W,
                           eps, d8, tx, vec, flag);
    else
        hipLaunchKernelGGL(certify_d8_kernel<false>, dim3(nt), dim3(NT), 0, ctx->stream, z, w, H, W,
                           eps, d8, tx, vec, flag);
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_d8_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, out, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    int tx, nt = tiles_of(H, W, TH, &tx);
    int vec = (W % 4 == 0) && ((uintptr_t)out % 4 == 0);
    {
        hdem_scoped_timer tm(ctx, HDEM_K_D8, (int64_t)H * W);
        hipLaunchKernelGGL(d8_kernel, dim3(nt), dim3(NT), 0, ctx->stream, z, H, W, out,
                           tx, vec);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_d8_f32(hdem_ctx *ctx, const float *z, int H, int W, uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, out, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t n = (size_t)H * W;
    hdem_dbuf dz, dout;
    if (int rc = dz.alloc(ctx, n * sizeof(float))) return rc;
    if (int rc = dout.alloc(ctx, n)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dz.p, z, n * sizeof(float))) return rc;
    if (int rc = hdem_d8_f32_dev(ctx, (const float *)dz.p, H, W, (uint8_t *)dout.p)) return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, n);
}

template <typename T>
static int boxmean_dev(hdem_ctx *ctx, const T *x, int H, int W, int do_round, T *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(x, out, H, W)) return rc;
    HDEM_REQUIRE((const void *)x != (const void *)out, HDEM_ERR_BAD_ARG,
                 "box mean cannot run in place");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    int tx, nt = tiles_of(H, W, tile_rows<T>::value, &tx);
    {
        hdem_scoped_timer tm(ctx, HDEM_K_BOXMEAN, (int64_t)H * W);
        hipLaunchKernelGGL(boxmean3_kernel<T>, dim3(nt), dim3(NT), 0, ctx->stream, x, H, W,
                           out, tx, do_round);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

template <typename T>
static int boxmean_host(hdem_ctx *ctx, const T *x, int H, int W, int do_round, T *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(x, out, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t bytes = (size_t)H * W * sizeof(T);
    hdem_dbuf dx, dout;
    if (int rc = dx.alloc(ctx, bytes)) return rc;
    if (int rc = dout.alloc(ctx, bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dx.p, x, bytes)) return rc;
    if (int rc = boxmean_dev<T>(ctx, (const T *)dx.p, H, W, do_round, (T *)dout.p)) return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, bytes);
}

extern "C" int hdem_boxmean3_f32_dev(hdem_ctx *c, const float *x, int H, int W, int r, float *o)
{ return boxmean_dev<float>(c, x, H, W, r, o); }
extern "C" int hdem_boxmean3_f64_dev(hdem_ctx *c, const double *x, int H, int W, int r, double *o)
{ return boxmean_dev<double>(c, x, H, W, r, o); }
extern "C" int hdem_boxmean3_f32(hdem_ctx *c, const float *x, int H, int W, int r, float *o)
{ return boxmean_host<float>(c, x, H, W, r, o); }
extern "C" int hdem_boxmean3_f64(hdem_ctx *c, const double *x, int H, int W, int r, double *o)
{ return boxmean_host<double>(c, x, H, W, r, o); }

template <typename T>
static int convolve_host(hdem_ctx *ctx, const T *x, int H, int W, const double *weights, int kh,
                         int kw, T *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(x, out, H, W)) return rc;
    HDEM_REQUIRE(weights, HDEM_ERR_BAD_ARG, "weights is null");
    HDEM_REQUIRE(kh > 0 && kw > 0 && kh <= 15 && kw <= 15, HDEM_ERR_BAD_ARG,
                 "weights shape %d x %d not supported (1..15 per axis)", kh, kw);
    HDEM_REQUIRE(kh % 2 == 1 && kw % 2 == 1, HDEM_ERR_WINDOW_EVEN,
                 "weights shape %d x %d must be odd on both axes", kh, kw);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t n = (size_t)H * W;
    hdem_dbuf dx, dout, dw;
    if (int rc = dx.alloc(ctx, n * sizeof(T))) return rc;
    if (int rc = dout.alloc(ctx, n * sizeof(T))) return rc;
    if (int rc = dw.alloc(ctx, sizeof(double) * kh * kw)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dx.p, x, n * sizeof(T))) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dw.p, weights, sizeof(double) * kh * kw)) return rc;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_CONVOLVE, (int64_t)n);
        hipLaunchKernelGGL(convolve_kernel<T>, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                           ctx->stream, (const T *)dx.p, H, W, (const double *)dw.p, kh, kw,
                           (T *)dout.p);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return hdem_memcpy_d2h(ctx, out, dout.p, n * sizeof(T));
}

extern "C" int hdem_convolve_f32(hdem_ctx *ctx, const float *x, int H, int W,
                                 const double *weights, int kh, int kw, float *out)
{ return convolve_host<float>(ctx, x, H, W, weights, kh, kw, out); }
// (a float64 raster is convolved in float64, as scipy.ndimage.convolve does:
// extension_filters.py:183)
extern "C" int hdem_convolve_f64(hdem_ctx *ctx, const double *x, int H, int W,
                                 const double *weights, int kh, int kw, double *out)
{ return convolve_host<double>(ctx, x, H, W, weights, kh, kw, out); }

template <typename T>
static int around_host(hdem_ctx *ctx, const T *x, int64_t n, T *out)
{
    HDEM_REQUIRE(ctx && x && out, HDEM_ERR_BAD_ARG, "null argument");
    HDEM_REQUIRE(n > 0, HDEM_ERR_BAD_ARG, "n must be positive");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    hdem_dbuf dx;
    if (int rc = dx.alloc(ctx, (size_t)n * sizeof(T))) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dx.p, x, (size_t)n * sizeof(T))) return rc;
    hipLaunchKernelGGL(around_kernel<T>, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                       ctx->stream, (const T *)dx.p, n, (T *)dx.p);
    HDEM_HIP_CHECK(hipGetLastError());
    return hdem_memcpy_d2h(ctx, out, dx.p, (size_t)n * sizeof(T));
}

extern "C" int hdem_around_f32(hdem_ctx *c, const float *x, int64_t n, float *o)
{ return around_host<float>(c, x, n, o); }
extern "C" int hdem_around_f64(hdem_ctx *c, const double *x, int64_t n, double *o)
{ return around_host<double>(c, x, n, o); }
