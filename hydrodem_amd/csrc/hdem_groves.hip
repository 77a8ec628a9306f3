// A3/A4: QuadraticFilter (custom_filters.py:226-257) and the fused
// GrovesCorrection pass (custom_filters.py:708-732, MaskTallGroves :533-534).
//
// Arithmetic.  The reference's closed form
//     out = ((s2 + s3) r1 - s1 (r2 + r3)) / (2 r1^2 - r0 (r2 + r3))
// with s1 = sum w, s2 = sum w xx^2, s3 = sum w yy^2 over the ws x ws window is
// the fixed correlation kernel K = a (xx^2 + yy^2) + b, sum K = 1, with
// v_k = -ws/2 + 1 + k (np.linspace(-ws/2+1, ws/2, ws); asymmetric: 0.5 at the
// centre), a = r1/den, b = -(r2+r3)/den.  It is separable:
//     out = sum_y ( a * R2[y] + (a v_y^2 + b) * R0[y] ),
//     R0[y] = sum_x d[y][x],  R2[y] = sum_x d[y][x] v_x^2,
// and because sum K = 1 it is evaluated on d = w - c0 (c0 = one cell near the
// tile centre), which keeps float32 accumulation ~1e-6 m from exact math (the
// reference itself sits 5e-5 m from exact math: float32 s1).
//
// Kernel shape (gfx950): one 256-thread workgroup per 64 x 32 output tile.
//   phase 0  stage the (32+2p) x (64+2p) input window in LDS as d = w - c0;
//   phase 1  row sums R0/R2 for every staged row, 4 adjacent columns per item
//            from aligned ds_read_b128 runs, results to two LDS planes;
//   phase 2  each lane walks 8 output rows of one column (conflict-free
//            ds_read_b32 down the planes), 2 fma per tap;
//   epilogue hl = img - smooth; m = groves && hl > thr;
//            out = m ? smooth : hl + smooth; ring of p cells = img unchanged.
// Algorithmic HBM bytes: 4 (img) + 1 (mask) + 4 (out) = 9 B/cell/iteration
// (8 for the plain quadratic filter).
#include "hdem_internal.h"

#include <cmath>

namespace {

constexpr int GTW = 64;
constexpr int GTH = 32;
constexpr int NT = 256;
constexpr int WS_MAX = 31;

struct quad_coef {
    float a;
    float v2[WS_MAX];   // v_k^2
    float cy[WS_MAX];   // a v_k^2 + b
};

template <int WS>
__global__ __launch_bounds__(NT) void groves_kernel(const float *__restrict__ img,
                                                   const uint8_t *__restrict__ groves,
                                                   int H, int W, float thr, int tiles_x,
                                                   quad_coef cf, float *__restrict__ out)
{
    constexpr int P = WS / 2;
    constexpr int IR = GTH + 2 * P;                   // staged rows
    constexpr int IC = GTW + 2 * P;                   // staged cols
    constexpr int IS = (IC + 3) / 4 * 4 + 4;          // row stride (16 B multiple)
    __shared__ __attribute__((aligned(16))) float in[IR * IS];
    __shared__ __attribute__((aligned(16))) float r0p[IR * GTW];
    __shared__ __attribute__((aligned(16))) float r2p[IR * GTW];

    const int tid = threadIdx.x;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x0 = bx * GTW, y0 = by * GTH;

    // tile offset: any finite value near the window works (sum K = 1)
    float c0 = img[(size_t)min(y0 + GTH / 2, H - 1) * W + min(x0 + GTW / 2, W - 1)];
    if (!(fabsf(c0) < HDEM_INF)) c0 = 0.0f;

    // phase 0: stage d = w - c0 (coordinates clamped; cells that would read
    // outside the raster only feed ring outputs, which are overwritten by img)
    for (int i = tid; i < IR * IC; i += NT) {
        int r = i / IC, c = i % IC;
        int gy = min(max(y0 - P + r, 0), H - 1);
        int gx = min(max(x0 - P + c, 0), W - 1);
        in[r * IS + c] = img[(size_t)gy * W + gx] - c0;
    }
    for (int i = tid; i < IR * (IS - IC); i += NT) {  // pad columns: defined values
        int r = i / (IS - IC), c = IC + i % (IS - IC);
        in[r * IS + c] = 0.0f;
    }
    __syncthreads();

    // phase 1: row sums, items of 4 adjacent output columns
    constexpr int NV = (4 + 2 * P + 3) / 4;           // b128 reads per item
    for (int i = tid; i < IR * (GTW / 4); i += NT) {
        int r = i / (GTW / 4), c4 = (i % (GTW / 4)) * 4;
        float d[NV * 4];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            hdem_f4 v = *reinterpret_cast<const hdem_f4 *>(&in[r * IS + c4 + 4 * k]);
            d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
        }
        hdem_f4 s0, s2;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float a0 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                a0 += d[o + k];
                a2 = fmaf(d[o + k], cf.v2[k], a2);
            }
            s0[o] = a0; s2[o] = a2;
        }
        *reinterpret_cast<hdem_f4 *>(&r0p[r * GTW + c4]) = s0;
        *reinterpret_cast<hdem_f4 *>(&r2p[r * GTW + c4]) = s2;
    }
    __syncthreads();

    // phase 2: column sums + epilogue; lane = column, wave = 8 output rows
    const int c = tid & 63, rg = tid >> 6;
    const int x = x0 + c;
    constexpr int RPT = GTH / 4;                      // rows per thread
    float q0[RPT + 2 * P], q2[RPT + 2 * P];
#pragma unroll
    for (int k = 0; k < RPT + 2 * P; ++k) {
        q0[k] = r0p[(rg * RPT + k) * GTW + c];
        q2[k] = r2p[(rg * RPT + k) * GTW + c];
    }
    if (x >= W) return;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int y = y0 + rg * RPT + rr;
        if (y >= H) break;
        float acc = 0.0f, acc2 = 0.0f;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            acc = fmaf(cf.cy[k], q0[rr + k], acc);
            acc2 += q2[rr + k];
        }
        float smooth = c0 + fmaf(cf.a, acc2, acc);
        size_t gi = (size_t)y * W + x;
        float w = img[gi];                                // L2-served re-read
        bool ring = y < P || y >= H - P || x < P || x >= W - P;
        float o;
        if (ring) {
            o = w;
        } else if (groves) {
            float hl = w - smooth;
            bool m = groves[gi] != 0 && hl > thr;
            o = m ? smooth : hl + smooth;
        } else {
            o = smooth;
        }
        out[gi] = o;
    }
}

int make_coef(int ws, quad_coef *cf)
{
    double v[WS_MAX], r0 = (double)ws * ws, r1 = 0, r2 = 0, r3 = 0;
    for (int k = 0; k < ws; ++k) v[k] = -ws / 2.0 + 1.0 + k;
    for (int j = 0; j < ws; ++j)
        for (int i = 0; i < ws; ++i) {
            double xx = v[i] * v[i], yy = v[j] * v[j];
            r1 += xx; r2 += xx * xx; r3 += xx * yy;
        }
    double den = 2.0 * r1 * r1 - r0 * (r2 + r3);
    double a = r1 / den, b = -(r2 + r3) / den;
    cf->a = (float)a;
    for (int k = 0; k < WS_MAX; ++k) {
        cf->v2[k] = k < ws ? (float)(v[k] * v[k]) : 0.0f;
        cf->cy[k] = k < ws ? (float)(a * v[k] * v[k] + b) : 0.0f;
    }
    return 0;
}

template <int WS>
void launch_ws(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H, int W,
               float thr, const quad_coef &cf, float *out)
{
    int tx = (W + GTW - 1) / GTW, ty = (H + GTH - 1) / GTH;
    hipLaunchKernelGGL(groves_kernel<WS>, dim3(tx * ty), dim3(NT), 0, ctx->stream, img,
                       groves, H, W, thr, tx, cf, out);
}

int launch_pass(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H, int W,
                int ws, float thr, const quad_coef &cf, float *out)
{
    hdem_scoped_timer tm(ctx, HDEM_K_GROVES, (int64_t)H * W);
    switch (ws) {
#define HDEM_WS_CASE(N) case N: launch_ws<N>(ctx, img, groves, H, W, thr, cf, out); break;
        HDEM_WS_CASE(3) HDEM_WS_CASE(5) HDEM_WS_CASE(7)
        HDEM_WS_CASE(9) HDEM_WS_CASE(11) HDEM_WS_CASE(13) HDEM_WS_CASE(15)
        HDEM_WS_CASE(17) HDEM_WS_CASE(19) HDEM_WS_CASE(21) HDEM_WS_CASE(23)
        HDEM_WS_CASE(25) HDEM_WS_CASE(27) HDEM_WS_CASE(29) HDEM_WS_CASE(31)
#undef HDEM_WS_CASE
        default: return HDEM_ERR_BAD_ARG;
    }
    return HDEM_OK;
}

// window validation, same two failure classes and the same order of checks as
// the SlidingWindow constructor (sliding_window.py:150-156)
int check_window(int ws, int H, int W)
{
    HDEM_REQUIRE(ws > 0, HDEM_ERR_BAD_ARG, "window size must be positive, got %d", ws);
    HDEM_REQUIRE(ws != 1, HDEM_ERR_BAD_ARG,
                 "window size 1 is degenerate (the reference's closed form is 0/0)");
    HDEM_REQUIRE(ws <= H && ws <= W, HDEM_ERR_WINDOW_HIGH,
                 "Window size: %d cannot be higher than grid dimensions: (%d, %d)", ws, H, W);
    HDEM_REQUIRE(ws % 2 == 1, HDEM_ERR_WINDOW_EVEN,
                 "Window size: %d cannot be an even number", ws);
    HDEM_REQUIRE(ws <= WS_MAX, HDEM_ERR_BAD_ARG,
                 "window size %d not supported by the HIP kernel (max %d)", ws, WS_MAX);
    return HDEM_OK;
}

}  // namespace

extern "C" int hdem_groves_f32_dev(hdem_ctx *ctx, const float *img, const uint8_t *groves,
                                   int H, int W, int ws, float thr, int iters,
                                   float *scratch, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(groves, HDEM_ERR_BAD_ARG, "groves mask is null");
    HDEM_REQUIRE(iters >= 1, HDEM_ERR_BAD_ARG, "iterations must be >= 1, got %d", iters);
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "groves cannot run in place");
    HDEM_REQUIRE(iters == 1 || (scratch && scratch != out && scratch != img),
                 HDEM_ERR_BAD_ARG, "iterations > 1 need a distinct scratch buffer");
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    quad_coef cf;
    make_coef(ws, &cf);
    // ping-pong so that the last pass lands in `out`
    const float *src = img;
    for (int it = 0; it < iters; ++it) {
        float *dst = ((iters - 1 - it) % 2 == 0) ? out : scratch;
        if (int rc = launch_pass(ctx, src, groves, H, W, ws, thr, cf, dst)) return rc;
        src = dst;
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_quadratic_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W, int ws,
                                      float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    HDEM_REQUIRE(dem != out, HDEM_ERR_BAD_ARG, "quadratic filter cannot run in place");
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    quad_coef cf;
    make_coef(ws, &cf);
    if (int rc = launch_pass(ctx, dem, nullptr, H, W, ws, 0.0f, cf, out)) return rc;
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_quadratic_f32(hdem_ctx *ctx, const float *dem, int H, int W, int ws,
                                  float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t bytes = (size_t)H * W * sizeof(float);
    hdem_dbuf din, dout;
    if (int rc = din.alloc(bytes)) return rc;
    if (int rc = dout.alloc(bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, din.p, dem, bytes)) return rc;
    if (int rc = hdem_quadratic_f32_dev(ctx, (const float *)din.p, H, W, ws, (float *)dout.p))
        return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, bytes);
}

extern "C" int hdem_groves_f32(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H,
                               int W, int ws, float thr, int iters, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(groves, HDEM_ERR_BAD_ARG, "groves mask is null");
    HDEM_REQUIRE(iters >= 1, HDEM_ERR_BAD_ARG, "iterations must be >= 1, got %d", iters);
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t n = (size_t)H * W, bytes = n * sizeof(float);
    hdem_dbuf din, dout, dscr, dg;
    if (int rc = din.alloc(bytes)) return rc;
    if (int rc = dout.alloc(bytes)) return rc;
    if (iters > 1) if (int rc = dscr.alloc(bytes)) return rc;
    if (int rc = dg.alloc(n)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, din.p, img, bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dg.p, groves, n)) return rc;
    if (int rc = hdem_groves_f32_dev(ctx, (const float *)din.p, (const uint8_t *)dg.p, H, W,
                                     ws, thr, iters, (float *)dscr.p, (float *)dout.p))
        return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, bytes);
}
