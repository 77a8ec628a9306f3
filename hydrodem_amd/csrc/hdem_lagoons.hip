// HydroSHEDS / lagoon branch: CorrectNANValues, MajorityFilter,
// TidyingLagoons, LagoonsDetection (custom_filters.py:260-317, 22-73, 564-661) and
// the SciPy morphology wrappers they use (extension_filters.py:187-345:
// binary_erosion, binary_closing, grey_dilation).  Small-window stencils on float32
// rasters and byte masks; all results are selections, comparisons and (one) small
// fixed-order float32 mean, so every kernel is bit-exact against the reference.
//
// HBM-bound except the majority vote, which is LDS-bound: ws^2 - 4 reads per vote
// pass, two passes (Boyer-Moore candidate, then its count).
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
__global__ __launch_bounds__(NT) void correct_nan_kernel(const float *__restrict__ in, int h, int w,
                                                         float *__restrict__ out)
{
    const int x = blockIdx.x * NT + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    float v = in[(size_t)y * w + x];
    if (v < 0.0f && y >= 1 && y < h - 1 && x >= 1 && x < w - 1) {
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
        v = n ? (float)((double)s / (double)n) : __builtin_nanf("");
    }
    out[(size_t)y * w + x] = v;
}

// MajorityFilter.apply (:44-73): the value held by more than 70 % of (ws^2 - 1) cells of
// the ws x ws window minus its corners, else 0; only centres whose window fits.  A value
// with that share is a strict majority, so Boyer-Moore finds it; the second pass counts.
constexpr int MTX = 64, MTY = 16, MMAX = 15;

__global__ __launch_bounds__(NT) void majority_kernel(const float *__restrict__ in, int h, int w,
                                                      int ws, int need, float *__restrict__ out)
{
    __shared__ float s[(MTY + MMAX - 1) * (MTX + MMAX - 1)];
    const int r = ws / 2, tw = MTX + 2 * r, th = MTY + 2 * r;
    const int x0 = blockIdx.x * MTX, y0 = blockIdx.y * MTY;
    for (int k = threadIdx.x; k < tw * th; k += NT) {
        const int ly = k / tw, lx = k - ly * tw;
        const int gy = y0 - r + ly, gx = x0 - r + lx;
        s[k] = (gy >= 0 && gy < h && gx >= 0 && gx < w) ? in[(size_t)gy * w + gx]
                                                        : __builtin_nanf("");
    }
    __syncthreads();
    const int lx = threadIdx.x % MTX;
    for (int ly = threadIdx.x / MTX; ly < MTY; ly += NT / MTX) {
        const int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) continue;
        float result = 0.0f;
        if (y >= r && y < h - r && x >= r && x < w - r) {
            float cand = 0.0f;
            int votes = 0;
            for (int dy = 0; dy < ws; ++dy)
                for (int dx = 0; dx < ws; ++dx) {
                    if ((dy == 0 || dy == ws - 1) && (dx == 0 || dx == ws - 1)) continue;
                    const float v = s[(ly + dy) * tw + lx + dx];
                    if (votes == 0) { cand = v; votes = 1; }
                    else votes += (v == cand) ? 1 : -1;
                }
            // a value present c times leaves at least 2c - cells votes: no count pass
            // where the vote already rules the share out
            if (votes >= 2 * need - (ws * ws - 4)) {
                int count = 0;
                for (int dy = 0; dy < ws; ++dy)
                    for (int dx = 0; dx < ws; ++dx) {
                        if ((dy == 0 || dy == ws - 1) && (dx == 0 || dx == ws - 1)) continue;
                        count += s[(ly + dy) * tw + lx + dx] == cand;
                    }
                if (count >= need) result = cand;
            }
        }
        out[(size_t)y * w + x] = result;
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
__global__ __launch_bounds__(NT) void mask_product_kernel(const float *__restrict__ img,
                                                          const uint8_t *__restrict__ m, size_t n,
                                                          float *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const hdem_f4 v = hdem_ld4u(img + i);
        const unsigned b = *(const unsigned *)(m + i);
        const hdem_f4 r = {v[0] * ((b & 0xffu) ? 1.0f : 0.0f), v[1] * ((b & 0xff00u) ? 1.0f : 0.0f),
                           v[2] * ((b & 0xff0000u) ? 1.0f : 0.0f),
                           v[3] * ((b & 0xff000000u) ? 1.0f : 0.0f)};
        hdem_st4u(out + i, r);
    } else {
        for (size_t k = i; k < n; ++k) out[k] = img[k] * (m[k] ? 1.0f : 0.0f);
    }
}

__device__ __forceinline__ int reflect(int i, int n)
{   // scipy mode='reflect': d c b a | a b c d | d c b a
    if (n == 1) return 0;
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// scipy.ndimage.grey_dilation(size=(sy, sx)), odd sizes: maximum over the centred window,
// mode='reflect'.  64 x 16 outputs per block from an LDS tile; the reflection is resolved
// once per loaded cell, the sy * sx taps are plain LDS reads.
constexpr int GTX = 64, GTY = 16;

__global__ __launch_bounds__(NT) void grey_dilation_kernel(const float *__restrict__ in, int h,
                                                           int w, int sy, int sx,
                                                           float *__restrict__ out)
{
    extern __shared__ float tile[];
    const int ry = sy / 2, rx = sx / 2, tw = GTX + 2 * rx, th = GTY + 2 * ry;
    const int x0 = blockIdx.x * GTX, y0 = blockIdx.y * GTY;
    for (int k = threadIdx.x; k < tw * th; k += NT) {
        const int ly = k / tw, lx = k - ly * tw;
        tile[k] = in[(size_t)reflect(y0 - ry + ly, h) * w + reflect(x0 - rx + lx, w)];
    }
    __syncthreads();
    const int lx = threadIdx.x % GTX;
    for (int ly = threadIdx.x / GTX; ly < GTY; ly += NT / GTX) {
        const int x = x0 + lx, y = y0 + ly;
        if (x >= w || y >= h) continue;
        float m = tile[ly * tw + lx];
        for (int dy = 0; dy < sy; ++dy)
            for (int dx = 0; dx < sx; ++dx) {
                const float v = tile[(ly + dy) * tw + lx + dx];
                if (v > m) m = v;
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
extern "C" int hdem_correct_nan_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    HDEM_REQUIRE(dem != out, HDEM_ERR_BAD_ARG, "the NaN correction cannot run in place");
    if (int rc = window_ok(3, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W);
        hipLaunchKernelGGL(correct_nan_kernel, grid2(W, H), dim3(NT), 0, ctx->stream, dem, H, W,
                           out);
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
        hipLaunchKernelGGL(majority_kernel, dim3((W + MTX - 1) / MTX, (H + MTY - 1) / MTY), dim3(NT),
                           0, ctx->stream, img, H, W, window, need, out);
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

extern "C" int hdem_grey_dilation_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, int sy,
                                          int sx, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "grey dilation cannot run in place");
    HDEM_REQUIRE(sy >= 1 && sx >= 1 && (sy & 1) && (sx & 1) && sy <= 31 && sx <= 31,
                 HDEM_ERR_BAD_ARG, "grey dilation size must be odd and <= 31, got (%d, %d)", sy, sx);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)H * W);
        hipLaunchKernelGGL(grey_dilation_kernel, dim3((W + GTX - 1) / GTX, (H + GTY - 1) / GTY),
                           dim3(NT), (GTX + sx - 1) * (GTY + sy - 1) * sizeof(float), ctx->stream,
                           img, H, W, sy, sx, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

// TidyingLagoons.apply (:564-610): erode (img != 0) twice, expand 7, multiply with img,
// 7 x 7 grey dilation.  scratch: 2 byte rasters + 1 float raster (6 bytes per cell).
static size_t round16(size_t n) { return (n + 15) / 16 * 16; }

static int tidying(hdem_ctx *ctx, const float *img, int H, int W, float *out, char *scratch)
{
    const size_t n = (size_t)H * W;
    uint8_t *a = (uint8_t *)scratch, *b = a + round16(n);
    float *f = (float *)(b + round16(n));
    morph_struct st;
    if (int rc = make_struct(CROSS, 3, 3, &st)) return rc;
    hipStream_t s = ctx->stream;
    const unsigned blocks = (unsigned)((n + 4 * NT - 1) / (4 * NT));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)n);
        hipLaunchKernelGGL(nonzero_kernel, dim3(blocks), dim3(NT), 0, s, img, n, a);
    }
    // a -> b -> a : two erosions, the result back in a
    if (int rc = erode_n(ctx, a, H, W, st, 1, nullptr, b)) return rc;
    if (int rc = erode_n(ctx, b, H, W, st, 1, nullptr, a)) return rc;
    if (int rc = hdem_expand_u8_dev(ctx, a, H, W, 7, b)) return rc;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)n * 2);
        hipLaunchKernelGGL(mask_product_kernel, dim3(blocks), dim3(NT), 0, s, img,
                           (const uint8_t *)b, n, f);
        hipLaunchKernelGGL(grey_dilation_kernel, dim3((W + GTX - 1) / GTX, (H + GTY - 1) / GTY),
                           dim3(NT), (GTX + 6) * (GTY + 6) * sizeof(float), s, (const float *)f, H,
                           W, 7, 7, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

static size_t tidying_scratch(size_t n) { return 2 * round16(n) + round16(n * sizeof(float)); }

extern "C" int hdem_tidying_lagoons_f32_dev(hdem_ctx *ctx, const float *img, int H, int W,
                                            float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "tidying cannot run in place");
    if (int rc = window_ok(7, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    char *scratch = (char *)hdem_arena(ctx, tidying_scratch((size_t)H * W));
    if (!scratch) return HDEM_ERR_OOM;
    return tidying(ctx, img, H, W, out, scratch);
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
    char *scratch = (char *)hdem_arena(ctx, 3 * fbytes + tidying_scratch(n));
    if (!scratch) return HDEM_ERR_OOM;
    float *major = (float *)scratch;
    if (!fixed) fixed = (float *)(scratch + fbytes);
    if (!values) values = (float *)(scratch + 2 * fbytes);
    if (int rc = hdem_correct_nan_f32_dev(ctx, hsheds, H, W, fixed)) return rc;
    if (int rc = hdem_majority_f32_dev(ctx, fixed, H, W, 11, major)) return rc;
    if (int rc = tidying(ctx, major, H, W, values, scratch + 3 * fbytes)) return rc;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_LAGOON, (int64_t)n);
        hipLaunchKernelGGL(positive_kernel, dim3((unsigned)((n + 4 * NT - 1) / (4 * NT))), dim3(NT), 0,
                           ctx->stream, (const float *)values, n, mask);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}
