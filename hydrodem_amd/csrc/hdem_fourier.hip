// Fourier destripe: DetectApplyFourier and its parts,
// custom_filters.py:369-462 (BlanksFourier, DetectBlanksFourier), :320-366
// (IsolatedPoints), :76-125 (ExpandFilter), :537-561 (MaskFourier), :834-1101
// (FourierInitial, FourierProcessQuarters, DetectApplyFourier); FFT wrappers
// extension_filters.py:348-480.
//
//   dem - mean -> real-to-complex FFT (rocFFT single precision, like scipy.fftpack on float32
//   input) into the left half of the spectrum, right half from the Hermitian symmetry
//   -> |F| of the two upper quadrants in *shifted* coordinates, 10-cell margin
//   -> 2 x { hollow 55x55 mean (inner 5x5 left out, cells past the quadrant edge left
//            out), cells > 4 x mean are peaks and are zeroed for the next pass }
//   -> peaks without a neighbour dropped, the rest dilated by 13x13 minus corners
//   -> those frequencies and their point mirrors zeroed in F -> inverse FFT -> |.|/N
//
// fftshift / ifftshift never move data: every kernel that needs shifted coordinates
// maps them (shifted i <-> unshifted (i - n/2) mod n).  The full-raster mask is only
// materialised when the caller asks for it.
//
// Kernels are HBM-bound stencils.  The hollow mean is separable and done in one pass per
// detection: lanes slide float64 column sums (spectra span ten decades) down their
// columns, a block-wide prefix turns them into the 55- and 5-wide window sums
// (hollow_detect_kernel).  Algorithmic bytes per quadrant cell and pass: 4 + 4 + 1.
#include <dlfcn.h>

#include <cmath>
#include <mutex>

#include "hdem_internal.h"

// ---------------------------------------------------------------------------
// rocFFT, loaded on first use (the rest of the library does not need it)
// ---------------------------------------------------------------------------
namespace {

typedef struct rocfft_plan_t *rocfft_plan;
typedef struct rocfft_execution_info_t *rocfft_execution_info;
typedef struct rocfft_plan_description_t *rocfft_plan_description;
enum { ROCFFT_COMPLEX_FORWARD = 0, ROCFFT_COMPLEX_INVERSE = 1, ROCFFT_REAL_FORWARD = 2 };
enum { ROCFFT_INPLACE = 0, ROCFFT_NOTINPLACE = 1 };
enum { ROCFFT_ARRAY_REAL = 2, ROCFFT_ARRAY_HERMITIAN_INTERLEAVED = 3 };
enum { ROCFFT_SINGLE = 0, ROCFFT_DOUBLE = 1 };

struct rocfft_api {
    void *handle = nullptr;
    int (*setup)() = nullptr;
    int (*plan_create)(rocfft_plan *, int, int, int, size_t, const size_t *, size_t,
                       const void *) = nullptr;
    int (*plan_destroy)(rocfft_plan) = nullptr;
    int (*plan_get_work_buffer_size)(const rocfft_plan, size_t *) = nullptr;
    int (*execute)(const rocfft_plan, void **, void **, rocfft_execution_info) = nullptr;
    int (*info_create)(rocfft_execution_info *) = nullptr;
    int (*info_destroy)(rocfft_execution_info) = nullptr;
    int (*info_set_work_buffer)(rocfft_execution_info, void *, size_t) = nullptr;
    int (*info_set_stream)(rocfft_execution_info, void *) = nullptr;
    int (*desc_create)(rocfft_plan_description *) = nullptr;
    int (*desc_destroy)(rocfft_plan_description) = nullptr;
    int (*desc_set_data_layout)(rocfft_plan_description, int, int, const size_t *, const size_t *,
                                size_t, const size_t *, size_t, size_t, const size_t *,
                                size_t) = nullptr;
};

rocfft_api g_fft;
std::mutex g_fft_lock;      // contexts of different threads may get here together

int load_rocfft()
{
    std::lock_guard<std::mutex> guard(g_fft_lock);
    if (g_fft.handle) return HDEM_OK;
    const char *names[] = {"librocfft.so.0", "/opt/rocm/lib/librocfft.so.0", "librocfft.so"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    HDEM_REQUIRE(h, HDEM_ERR_HIP, "cannot load rocFFT: %s", dlerror());
#define HDEM_SYM(field, name)                                                         \
    do {                                                                              \
        *(void **)(&g_fft.field) = dlsym(h, name);                                    \
        HDEM_REQUIRE(g_fft.field, HDEM_ERR_HIP, "rocFFT lacks %s", name);             \
    } while (0)
    HDEM_SYM(setup, "rocfft_setup");
    HDEM_SYM(plan_create, "rocfft_plan_create");
    HDEM_SYM(plan_destroy, "rocfft_plan_destroy");
    HDEM_SYM(plan_get_work_buffer_size, "rocfft_plan_get_work_buffer_size");
    HDEM_SYM(execute, "rocfft_execute");
    HDEM_SYM(info_create, "rocfft_execution_info_create");
    HDEM_SYM(info_destroy, "rocfft_execution_info_destroy");
    HDEM_SYM(info_set_work_buffer, "rocfft_execution_info_set_work_buffer");
    HDEM_SYM(info_set_stream, "rocfft_execution_info_set_stream");
    HDEM_SYM(desc_create, "rocfft_plan_description_create");
    HDEM_SYM(desc_destroy, "rocfft_plan_description_destroy");
    HDEM_SYM(desc_set_data_layout, "rocfft_plan_description_set_data_layout");
#undef HDEM_SYM
    HDEM_REQUIRE(g_fft.setup() == 0, HDEM_ERR_HIP, "rocfft_setup failed");
    g_fft.handle = h;
    return HDEM_OK;
}

}  // namespace

// per-context state: plans and the work buffer for one raster shape
struct hdem_fourier_state {
    int H = 0, W = 0;
    rocfft_plan fwd = nullptr, inv = nullptr, r2c = nullptr;
    // the destripe's inverse as two batched 1-D passes with own transposes in between
    // and behind (the second one fused with the final abs): rows of W, then rows of H
    rocfft_plan inv_rows = nullptr, inv_cols = nullptr;
    void *tr = nullptr;             // H x W complex: the transposed intermediate
    bool split_needs_work = false;  // the 1-D plans use rocFFT's work buffer themselves
    rocfft_execution_info info = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    // scratch of hdem_fourier_destripe_f32_dev, kept for the next raster of this shape
    // (a 16384^2 destripe spends ~2 of 17 ms in hipMalloc / hipFree otherwise)
    void *scratch = nullptr;
};

void hdem_fourier_release(hdem_ctx *ctx)
{
    hdem_fourier_state *s = ctx->fourier;
    if (!s) return;
    if (s->fwd) g_fft.plan_destroy(s->fwd);
    if (s->inv) g_fft.plan_destroy(s->inv);
    if (s->r2c) g_fft.plan_destroy(s->r2c);
    if (s->inv_rows) g_fft.plan_destroy(s->inv_rows);
    if (s->inv_cols) g_fft.plan_destroy(s->inv_cols);
    if (s->tr) (void)hipFree(s->tr);
    if (s->info) g_fft.info_destroy(s->info);
    if (s->work) (void)hipFree(s->work);
    if (s->scratch) (void)hipFree(s->scratch);
    delete s;
    ctx->fourier = nullptr;
}

namespace {

int ensure_plans(hdem_ctx *ctx, int H, int W)
{
    if (int rc = load_rocfft()) return rc;
    hdem_fourier_state *s = ctx->fourier;
    if (s && s->H == H && s->W == W) return HDEM_OK;
    hdem_fourier_release(ctx);
    s = ctx->fourier = new hdem_fourier_state;
    const size_t lengths[2] = {(size_t)W, (size_t)H};        // fastest dimension first
    HDEM_REQUIRE(g_fft.plan_create(&s->fwd, ROCFFT_INPLACE, ROCFFT_COMPLEX_FORWARD, ROCFFT_SINGLE,
                                   2, lengths, 1, nullptr) == 0,
                 HDEM_ERR_HIP, "rocfft_plan_create (forward %d x %d) failed", H, W);
    HDEM_REQUIRE(g_fft.plan_create(&s->inv, ROCFFT_INPLACE, ROCFFT_COMPLEX_INVERSE, ROCFFT_SINGLE,
                                   2, lengths, 1, nullptr) == 0,
                 HDEM_ERR_HIP, "rocfft_plan_create (inverse %d x %d) failed", H, W);
    // real -> Hermitian half, written straight into the left half of the full complex
    // spectrum (row stride W): the destripe fills the right half from the symmetry
    {
        rocfft_plan_description d = nullptr;
        HDEM_REQUIRE(g_fft.desc_create(&d) == 0, HDEM_ERR_HIP, "rocfft description_create failed");
        const size_t strides[2] = {1, (size_t)W};
        const int rc = g_fft.desc_set_data_layout(d, ROCFFT_ARRAY_REAL,
                                                  ROCFFT_ARRAY_HERMITIAN_INTERLEAVED, nullptr,
                                                  nullptr, 2, strides, (size_t)H * W, 2, strides,
                                                  (size_t)H * W);
        const int rc2 = rc ? rc
                           : g_fft.plan_create(&s->r2c, ROCFFT_NOTINPLACE, ROCFFT_REAL_FORWARD,
                                               ROCFFT_SINGLE, 2, lengths, 1, d);
        g_fft.desc_destroy(d);
        HDEM_REQUIRE(rc2 == 0, HDEM_ERR_HIP, "rocfft_plan_create (real forward %d x %d) failed",
                     H, W);
    }
    // (powers of two only -- 8192^2: 2.4 against 2.6 ms for the whole destripe, 16384^2: 8.9
    // against 9.5; at 6000^2 and 12000^2 rocFFT's 2-D plan takes a route without the
    // transposes and is as fast or faster: 3.1 against 3.4 ms at 6000^2)
    if ((W & (W - 1)) == 0 && (H & (H - 1)) == 0) {
        const size_t lw[1] = {(size_t)W}, lh[1] = {(size_t)H};
        if (g_fft.plan_create(&s->inv_rows, ROCFFT_INPLACE, ROCFFT_COMPLEX_INVERSE, ROCFFT_SINGLE, 1,
                              lw, (size_t)H, nullptr) != 0)
            s->inv_rows = nullptr;
        if (g_fft.plan_create(&s->inv_cols, ROCFFT_INPLACE, ROCFFT_COMPLEX_INVERSE, ROCFFT_SINGLE, 1,
                              lh, (size_t)W, nullptr) != 0)
            s->inv_cols = nullptr;
    }
    size_t a = 0, b = 0, c3 = 0;
    g_fft.plan_get_work_buffer_size(s->fwd, &a);
    g_fft.plan_get_work_buffer_size(s->inv, &b);
    g_fft.plan_get_work_buffer_size(s->r2c, &c3);
    s->work_bytes = a > b ? a : b;
    if (c3 > s->work_bytes) s->work_bytes = c3;
    for (rocfft_plan p : {s->inv_rows, s->inv_cols}) {
        size_t w1 = 0;
        if (p) g_fft.plan_get_work_buffer_size(p, &w1);
        if (w1 > s->work_bytes) s->work_bytes = w1;
        if (w1) s->split_needs_work = true;          // (lengths that take Bluestein's route)
    }
    if (s->work_bytes)
        if (int rc = hdem_raw_alloc(ctx, s->work_bytes, &s->work)) return rc;
    HDEM_REQUIRE(g_fft.info_create(&s->info) == 0, HDEM_ERR_HIP, "rocfft info_create failed");
    if (s->work_bytes)
        HDEM_REQUIRE(g_fft.info_set_work_buffer(s->info, s->work, s->work_bytes) == 0,
                     HDEM_ERR_HIP, "rocfft set_work_buffer failed");
    s->H = H;
    s->W = W;
    return HDEM_OK;
}

int run_fft(hdem_ctx *ctx, bool inverse, float2 *data)
{
    hdem_fourier_state *s = ctx->fourier;
    HDEM_REQUIRE(g_fft.info_set_stream(s->info, ctx->stream) == 0, HDEM_ERR_HIP,
                 "rocfft set_stream failed");
    void *in[1] = {data};
    hdem_scoped_timer tm(ctx, HDEM_K_FFT, (int64_t)s->H * s->W);
    HDEM_REQUIRE(g_fft.execute(inverse ? s->inv : s->fwd, in, nullptr, s->info) == 0,
                 HDEM_ERR_HIP, "rocfft_execute failed");
    return HDEM_OK;
}

int run_r2c(hdem_ctx *ctx, float *real_in, float2 *spectrum)
{
    hdem_fourier_state *s = ctx->fourier;
    HDEM_REQUIRE(g_fft.info_set_stream(s->info, ctx->stream) == 0, HDEM_ERR_HIP,
                 "rocfft set_stream failed");
    void *in[1] = {real_in}, *out[1] = {spectrum};
    hdem_scoped_timer tm(ctx, HDEM_K_FFT, (int64_t)s->H * s->W);
    HDEM_REQUIRE(g_fft.execute(s->r2c, in, out, s->info) == 0, HDEM_ERR_HIP,
                 "rocfft_execute (real forward) failed");
    return HDEM_OK;
}

constexpr int NT = 256;

// geometry of the reference's quadrants (FourierProcessQuarters.__init__, :906-925)
struct quad_geom {
    int ny, nx, mid_y, mid_x, y_odd, x_odd, qh, qw;
    __host__ __device__ int col0(int second) const { return second ? mid_x + 10 + x_odd : 0; }
};

__host__ quad_geom make_geom(int H, int W)
{
    quad_geom g;
    g.ny = H; g.nx = W;
    g.mid_y = H / 2; g.y_odd = H % 2;
    g.mid_x = W / 2; g.x_odd = W % 2;
    g.qh = g.mid_y - 10;
    g.qw = g.mid_x - 10;
    return g;
}

// shifted index -> index in the transform as rocFFT / fftpack lay it out
__device__ __forceinline__ int unshift(int i, int n) { const int u = i - n / 2; return u < 0 ? u + n : u; }

// A reference level of the raster: the mean of every step-th cell (about a million of
// them; any level near the elevations serves), deterministic (fixed tree): partial[b] per
// block, then one block.  The transform runs on dem - level and the level is added back
// before the final |.|:
// the same numbers as the reference's (only the zero frequency moves, which no quadrant
// contains), at a tenth of the complex64 rounding error for elevations around 100 m.
constexpr int SUM_BLOCKS = 1024;

__device__ __forceinline__ double block_sum(double v, double *s)
{
    s[threadIdx.x] = v;
    __syncthreads();
    for (int k = NT / 2; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) s[threadIdx.x] += s[threadIdx.x + k];
        __syncthreads();
    }
    return s[0];
}

__global__ __launch_bounds__(NT) void sum_partial_kernel(const float *__restrict__ x, size_t n,
                                                         size_t step,
                                                         double *__restrict__ partial)
{
    __shared__ double s[NT];
    double acc = 0.0;
    for (size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * step; i < n;
         i += (size_t)SUM_BLOCKS * NT * step)
        acc += (double)x[i];
    const double t = block_sum(acc, s);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ __launch_bounds__(NT) void sum_final_kernel(const double *__restrict__ partial, size_t n,
                                                       double *__restrict__ mean)
{
    __shared__ double s[NT];
    double acc = 0.0;
    for (int i = threadIdx.x; i < SUM_BLOCKS; i += NT) acc += partial[i];
    const double t = block_sum(acc, s);
    if (threadIdx.x == 0) *mean = t / (double)n;
}

__global__ __launch_bounds__(NT) void to_real_kernel(const float *__restrict__ x, size_t n,
                                                     const double *__restrict__ mean,
                                                     float *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4;
    const double m = *mean;
    if (i + 4 <= n) {
        const hdem_f4 v = hdem_ld4u(x + i);
        const hdem_f4 r = {(float)((double)v[0] - m), (float)((double)v[1] - m),
                           (float)((double)v[2] - m), (float)((double)v[3] - m)};
        hdem_st4u(out + i, r);
    } else {
        for (size_t k = i; k < n; ++k) out[k] = (float)((double)x[k] - m);
    }
}

// The transform of a real raster is Hermitian: F[u][v] = conj(F[-u][-v]).  The real
// forward transform fills columns 0 .. W/2 of every row; this fills the rest.
__global__ __launch_bounds__(NT) void fill_right_half_kernel(float2 *F, int H, int W)
{
    const int first = W / 2 + 1, v = first + blockIdx.x * NT + threadIdx.x, u = blockIdx.y;
    if (v >= W) return;
    const float2 s = F[(size_t)(u ? H - u : 0) * W + (W - v)];
    F[(size_t)u * W + v] = make_float2(s.x, -s.y);
}

// |F| of one upper quadrant (FourierInitial :857 + _get_firsts_quarters :953-966)
__global__ __launch_bounds__(NT) void quadrant_abs_kernel(const float2 *__restrict__ F, quad_geom g,
                                                          int second, float *__restrict__ q)
{
    const int j = blockIdx.x * NT + threadIdx.x, i = blockIdx.y;
    if (j >= g.qw) return;
    const float2 v = F[(size_t)unshift(i, g.ny) * g.nx + unshift(g.col0(second) + j, g.nx)];
    q[(size_t)i * g.qw + j] = hypotf(v.x, v.y);
}

// BlanksFourier.apply (:417-429) / DetectBlanksFourier (:457-461): q > factor x hollow mean.
// (An earlier version made float64 row sums in one kernel and slid them down the columns
// in a second: 45 B per cell and pass through HBM, 1.0 ms per pass at 8182^2.)
// ONE pass over the quadrant, ~10 B per cell.  A block of 256 lanes owns 256 columns
// (the 202 in the middle are its outputs, R on each side are context) and walks down
// `seg` rows.  Each lane slides the two *column* sums of its column (float64; two
// 4-byte row loads per step, the second a re-read from cache); per row the block turns
// the 256 column sums into an inclusive prefix (wave scan by shuffles + the three wave
// totals) and every output is P[c + R] - P[c - R - 1] -- two LDS reads instead of 55.
// Out of place: q_out <- q * (1 - hit) (may be NULL: the second pass only needs the mask).

// Inclusive prefix sum over the 64 lanes of a wave, float64, with DPP lane moves instead
// of LDS-crossbar shuffles: four shifts inside each row of 16 lanes (lanes without a source
// add 0), then lane 15 of rows 0 and 2 broadcast into rows 1 and 3, then lane 31 into the
// upper half (gfx9 row_shr / row_bcast).  A double moves as its two 32-bit halves.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xF,
                                               ROW_MASK == 0xF);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double wave_scan_f64(double p)
{
    p += dpp_move_f64<0x111, 0xF>(p);      // row_shr:1
    p += dpp_move_f64<0x112, 0xF>(p);      // row_shr:2
    p += dpp_move_f64<0x114, 0xF>(p);      // row_shr:4
    p += dpp_move_f64<0x118, 0xF>(p);      // row_shr:8
    p += dpp_move_f64<0x142, 0xA>(p);      // row_bcast:15 -> rows 1 and 3
    p += dpp_move_f64<0x143, 0xC>(p);      // row_bcast:31 -> rows 2 and 3
    return p;
}

// Four columns per lane (RT: half the window, or 0 = taken from `reach` at run time --
// BlanksFourier windows other than the 55 of the reference's pipeline).  Round 1's form held
// one column per lane and was instruction-bound: ~250 vector instructions per row and wave,
// most of them per-lane overhead that does not depend on how many cells the lane holds (five
// row loads, two float64 wave scans, the LDS hand-off).  Here the row loads are 16-byte, the
// scans run on the lanes' totals (a lane-local prefix in front, the exclusive wave prefix
// added) and the inner window is a prefix difference too.  128 lanes = 512 columns per block,
// 512 - 2R of them outputs; one row per barrier, two LDS buffers; the start-up column sums
// are read eight rows at a time.
constexpr int H4T = 128, H4W = 4 * H4T;

// FULL: every column of the block lies inside the quadrant (all block columns but the first and
// the last one or two): every lane loads 16 bytes from a clamped row WITHOUT a condition and a
// row that does not exist is masked off afterwards.  With the loads under per-lane and per-row
// conditions (the general form) the compiler cannot count what is in flight and waits for every
// row load before it issues the next -- five round trips to L2 per row, which is what the "2.6 us
// per row" of the note at the launch was made of.
struct hollow_lds {
    double pre[2][H4W + 1], sml[2][H4W + 1];   // inclusive prefixes inside each wave; [0] = 0
    double wtot[2][2], wsml[2][2];             // the waves' totals
};

template <int RT, int r, bool FULL>
__device__ __forceinline__ void hollow_detect4_body(hollow_lds &L, const float *__restrict__ q, int h,
                                                    int w, float factor, int seg,
                                                    float *__restrict__ q_out, uint8_t *found,
                                                    uint8_t *total, uint8_t *occ_mark,
                                                    const uint8_t *__restrict__ occ_skip, int occ_w,
                                                    int reach)
{
    const int R = RT ? RT : reach;
    auto &pre = L.pre;
    auto &sml = L.sml;
    auto &wtot = L.wtot;
    auto &wsml = L.wsml;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int outw = H4W - 2 * R;
    const int c_first = (int)blockIdx.x * outw - R, c0 = c_first + 4 * tid;
    const int y0 = blockIdx.y * seg, y1 = min(y0 + seg, h);
    if (occ_skip) {
        const int c_lo = max(c_first, 0) >> 5, c_hi = min(c_first + H4W - 1, w - 1) >> 5;
        const int r_lo = max(y0 - R, 0) >> 5, r_hi = min(y1 - 1 + R, h - 1) >> 5;
        const int nc = c_hi - c_lo + 1, cells = nc * (r_hi - r_lo + 1);
        int any = 0;
        for (int k = tid; k < cells; k += H4T)
            any |= occ_skip[(size_t)(r_lo + k / nc) * occ_w + c_lo + k % nc];
        if (!__syncthreads_or(any)) return;
    }
    if (tid == 0) { pre[0][0] = pre[1][0] = 0.0; sml[0][0] = sml[1][0] = 0.0; }
    const bool all4 = c0 >= 0 && c0 + 3 < w;
    bool live[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) live[k] = c0 + k >= 0 && c0 + k < w;
    // one row of my four columns (0 where the column or the row does not exist)
    auto load4 = [&](int y, bool want) -> hdem_f4 {
        if (FULL) {
            const unsigned keep = (want && y >= 0 && y < h) ? 0xffffffffu : 0u;     // uniform
            hdem_f4 v = hdem_ld4u(q + (size_t)min(max(y, 0), h - 1) * w + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __uint_as_float(__float_as_uint(v[k]) & keep);
            return v;
        }
        hdem_f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!want || y < 0 || y >= h) return v;
        const float *row = q + (size_t)y * w;
        if (all4) return hdem_ld4u(row + c0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (live[k]) v[k] = row[c0 + k];
        return v;
    };
    // the column sums of the first row's windows: eight row loads in flight at a time (one at
    // a time, each waited for, this start-up was a third of the kernel), added in row order
    double cb[4] = {0.0, 0.0, 0.0, 0.0}, cs[4] = {0.0, 0.0, 0.0, 0.0};
    for (int ya = max(y0 - R, 0), ye = min(y0 + R, h - 1); ya <= ye; ya += 8) {
        hdem_f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = load4(ya + j, ya + j <= ye);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) cb[k] += (double)v[j][k];
    }
    {
        hdem_f4 v[2 * r + 1];
#pragma unroll
        for (int j = 0; j < 2 * r + 1; ++j) v[j] = load4(y0 - r + j, y0 - r + j <= min(y0 + r, h - 1));
#pragma unroll
        for (int j = 0; j < 2 * r + 1; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) cs[k] += (double)v[j][k];
    }
    bool outs[4];
    int cols_b[4], cols_s[4];
    bool any_out = false, all_out = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = 4 * tid + k, c = c0 + k;
        outs[k] = i >= R && i < H4W - R && live[k];
        any_out = any_out || outs[k];
        all_out = all_out && outs[k];
        cols_b[k] = min(c + R, w - 1) - max(c - R, 0) + 1;
        cols_s[k] = min(c + r, w - 1) - max(c - r, 0) + 1;
    }
    // (Issuing the next rows' loads before working on the current ones -- software pipelining
    // at 180 registers and two waves per SIMD -- is slower than this form at three: 0.37
    // against 0.28 ms per launch.  So are four rows per loop body, 0.27 against 0.26, and two
    // rows' scans ahead of ONE barrier with four prefix buffers, 0.27: 32 KB of LDS per block
    // leave two waves per SIMD.  What remains is float64 arithmetic at half rate: ~40 operations
    // per cell are 0.15 ms of the 0.26.)
    constexpr int UN = 2;
    for (int yb = y0; yb < y1; yb += UN) {
        hdem_f4 in_b[UN], out_b[UN], in_s[UN], out_s[UN], cell[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int y = yb + u;
            in_b[u] = load4(y + R + 1, true);
            out_b[u] = load4(y - R, y < h);
            in_s[u] = load4(y + r + 1, true);
            out_s[u] = load4(y - r, y < h);
            cell[u] = load4(y, (FULL || any_out) && y < y1);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int y = yb + u;
            if (y >= y1) break;                               // uniform over the block
            const int buf = u & 1;
            // lane-local inclusive prefixes, then the wave's exclusive prefix of the lane totals
            double lb[4], ls[4];
            lb[0] = cb[0]; ls[0] = cs[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) { lb[k] = lb[k - 1] + cb[k]; ls[k] = ls[k - 1] + cs[k]; }
            const double pb = wave_scan_f64(lb[3]), ps = wave_scan_f64(ls[3]);
            const double eb = pb - lb[3], es = ps - ls[3];
            if (lane == 63) { wtot[buf][wave] = pb; wsml[buf][wave] = ps; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pre[buf][1 + 4 * tid + k] = eb + lb[k];
                sml[buf][1 + 4 * tid + k] = es + ls[k];
            }
            __syncthreads();
            if (any_out) {
                const double t0 = wtot[buf][0], u0 = wsml[buf][0];
                const int rows_b = min(y + R, h - 1) - max(y - R, 0) + 1;
                const int rows_s = min(y + r, h - 1) - max(y - r, 0) + 1;
                unsigned hits = 0;
                float keep[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = 4 * tid + k;
                    // window = columns b+1 .. a of the block; a column >= 256 sits in wave 1
                    const int a = min(i + R, H4W - 1), b = max(i - R - 1, -1);
                    const int a2 = min(i + r, H4W - 1), b2 = max(i - r - 1, -1);
                    const double hi = pre[buf][a + 1] + (a >= 256 ? t0 : 0.0);
                    const double lo = pre[buf][b + 1] + (b >= 256 ? t0 : 0.0);
                    const double hs = sml[buf][a2 + 1] + (a2 >= 256 ? u0 : 0.0);
                    const double lw = sml[buf][b2 + 1] + (b2 >= 256 ? u0 : 0.0);
                    const double sb = hi - lo, ss = hs - lw;
                    const int cnt = rows_b * cols_b[k] - rows_s * cols_s[k];
                    const float v = cell[u][k];
                    const bool hit = outs[k] && cnt > 0 &&
                                     (double)v * (double)cnt > (double)factor * (sb - ss);
                    hits |= (unsigned)hit << k;
                    keep[k] = hit ? v * 0.0f : v;
                }
                const size_t o = (size_t)y * w + c0;
                if (all_out) {
                    if (found) {
                        // (4 bytes at any alignment: one unaligned dword store)
                        typedef unsigned u1 __attribute__((aligned(1)));
                        *reinterpret_cast<u1 *>(found + o) =
                            (hits & 1u) | (hits & 2u) << 7 | (hits & 4u) << 14 | (hits & 8u) << 21;
                    }
                    if (q_out) hdem_st4u(q_out + o, (hdem_f4){keep[0], keep[1], keep[2], keep[3]});
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!outs[k]) continue;
                        if (found) found[o + k] = (hits >> k) & 1u;
                        if (q_out) q_out[o + k] = keep[k];
                    }
                }
                if (hits) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (!((hits >> k) & 1u)) continue;
                        if (occ_mark) occ_mark[(size_t)(y >> 5) * occ_w + ((c0 + k) >> 5)] = 1;
                        if (total) total[o + k] = (uint8_t)(total[o + k] + 1);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cb[k] += (double)in_b[u][k] - (double)out_b[u][k];
                cs[k] += (double)in_s[u][k] - (double)out_s[u][k];
            }
        }
    }
}

// (one launch, both forms: a block's rows are a serial chain, and three launches -- first block
// column, the full ones, the last -- cost three chains: 0.38 instead of 0.28 ms per pass)
template <int RT, int r>
__global__ __launch_bounds__(H4T) void hollow_detect4_kernel(const float *__restrict__ q, int h, int w,
                                                             float factor, int seg,
                                                             float *__restrict__ q_out,
                                                             uint8_t *found, uint8_t *total,
                                                             uint8_t *occ_mark,
                                                             const uint8_t *__restrict__ occ_skip,
                                                             int occ_w, int reach)
{
    __shared__ hollow_lds L;
    const int R = RT ? RT : reach;
    const int c_first = (int)blockIdx.x * (H4W - 2 * R) - R;
    if (c_first >= 0 && c_first + H4W <= w)
        hollow_detect4_body<RT, r, true>(L, q, h, w, factor, seg, q_out, found, total, occ_mark,
                                         occ_skip, occ_w, reach);
    else
        hollow_detect4_body<RT, r, false>(L, q, h, w, factor, seg, q_out, found, total, occ_mark,
                                          occ_skip, occ_w, reach);
}

// The detection masks are sparse (a handful of spectral peaks among 10^7..10^8 cells), so
// the three mask kernels walk the flat byte array 16 cells per lane and only stop at
// non-zero bytes.  vec: the mask pointer is 16-byte aligned (one dwordx4 load per lane).
template <typename Fn>
__device__ __forceinline__ void for_nonzero(const uint8_t *__restrict__ m, size_t n, bool vec, Fn fn)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 16;
    if (i >= n) return;
    if (vec && i + 16 <= n) {
        const uint4 v = *(const uint4 *)(m + i);
        if (!(v.x | v.y | v.z | v.w)) return;
        const unsigned wds[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned b = (wds[k >> 2] >> (8 * (k & 3))) & 0xffu;
            if (b) fn(i + k, (uint8_t)b);
        }
    } else {
        const size_t end = i + 16 < n ? i + 16 : n;
        for (size_t k = i; k < end; ++k)
            if (m[k]) fn(k, m[k]);
    }
}

inline dim3 grid16(size_t n) { return dim3((unsigned)((n + 16 * NT - 1) / (16 * NT))); }
inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// IsolatedPoints.apply (:344-366): interior cells equal to 1 survive only next to
// another non-zero cell (Jacobi on the input); everything else is copied.  out holds a
// copy of m on entry; the lonely ones are cleared.
__global__ __launch_bounds__(NT) void isolated_kernel(const uint8_t *__restrict__ m, int h, int w,
                                                      int reach, bool vec, uint8_t *__restrict__ out)
{
    for_nonzero(m, (size_t)h * w, vec, [&](size_t idx, uint8_t v) {
        const int y = (int)(idx / (size_t)w), x = (int)(idx - (size_t)y * w);
        if (v != 1 || y < reach || y >= h - reach || x < reach || x >= w - reach) return;
        bool any = false;
        for (int dy = -reach; dy <= reach; ++dy)
            for (int dx = -reach; dx <= reach; ++dx)
                if ((dy || dx) && m[(size_t)(y + dy) * w + x + dx]) any = true;
        if (!any) out[idx] = 0;
    });
}

// ExpandFilter.apply (:103-125) as a scatter: every non-zero cell marks the centres whose
// window -- the square minus its four corners -- holds it; only centres whose window
// fits in the array exist.  out must be zeroed.  (Detected peaks come in clusters -- lines
// along the spectrum's axes -- so each lane scattering its own cells beats a wave taking
// the wave's cells one at a time: 0.26 against 0.6 ms on a quarter of 16384^2.)
// REACH > 0: the window size is known at compile time (13 for the Fourier masks, 7 for the
// lagoons): a mark whose whole window lies among valid centres writes it as straight-line
// code, one pointer per row and immediate column offsets -- a lane with many marks (every
// other cell of a row: the harmonics of a stripe) runs them one after the other, and the
// generic loops cost ~15 instructions per store (0.29 -> 0.15 ms on the bench quadrant).
template <int REACH>
__global__ __launch_bounds__(NT) void expand_kernel(const uint8_t *__restrict__ m, int h, int w,
                                                    int reach_, bool vec, uint8_t *out)
{
    const int reach = REACH ? REACH : reach_;
    for_nonzero(m, (size_t)h * w, vec, [&](size_t idx, uint8_t) {
        const int y = (int)(idx / (size_t)w), x = (int)(idx - (size_t)y * w);
        auto mark = [&](int cy, int cx) {
            if (cy >= reach && cy < h - reach && cx >= reach && cx < w - reach)
                out[(size_t)cy * w + cx] = 1;
        };
        // Peaks come in runs along the spectrum's axes: a cell whose left (upper) neighbour
        // is marked too only adds what that neighbour's window does not hold -- the column
        // (row) that enters, without its corner cells, and the two cells that were the
        // neighbour's cut corners: 2 reach + 1 stores instead of (2 reach + 1)^2 - 4.
        const bool inside = REACH && y >= 2 * REACH && y < h - 2 * REACH && x >= 2 * REACH &&
                            x < w - 2 * REACH;             // every centre of the window is valid
        if (x > 0 && m[idx - 1]) {
            if (inside) {
                uint8_t *col = out + (size_t)(y - REACH) * w + x + REACH;
                col[-1] = 1;
#pragma unroll
                for (int dy = -REACH + 1; dy <= REACH - 1; ++dy) { col += w; col[0] = 1; }
                col[(size_t)w - 1] = 1;
                return;
            }
            for (int dy = -reach + 1; dy <= reach - 1; ++dy) mark(y + dy, x + reach);
            mark(y - reach, x + reach - 1);
            mark(y + reach, x + reach - 1);
            return;
        }
        if (y > 0 && m[idx - (size_t)w]) {
            if (inside) {
                uint8_t *row = out + (size_t)(y + REACH) * w + x;
#pragma unroll
                for (int dx = -REACH + 1; dx <= REACH - 1; ++dx) row[dx] = 1;
                row[-(ptrdiff_t)w - REACH] = 1;
                row[-(ptrdiff_t)w + REACH] = 1;
                return;
            }
            for (int dx = -reach + 1; dx <= reach - 1; ++dx) mark(y + reach, x + dx);
            mark(y + reach - 1, x - reach);
            mark(y + reach - 1, x + reach);
            return;
        }
        if (inside) {
            uint8_t *row = out + (size_t)(y - REACH) * w + x;
#pragma unroll
            for (int dy = -REACH; dy <= REACH; ++dy) {
#pragma unroll
                for (int dx = -REACH; dx <= REACH; ++dx)
                    if (!((dy == -REACH || dy == REACH) && (dx == -REACH || dx == REACH))) row[dx] = 1;
                row += w;
            }
            return;
        }
        for (int dy = -reach; dy <= reach; ++dy)
            for (int dx = -reach; dx <= reach; ++dx)
                if (!((dy == -reach || dy == reach) && (dx == -reach || dx == reach)))
                    mark(y + dy, x + dx);
    });
}

// _fill_complete_quarters ... _fill_complete_mask (:985-1050) + (1 - mask) * spectrum
// (:1097-1098): a marked quadrant cell zeroes its frequency and the point mirror
// (ny-1-i, nx-1-j in shifted coordinates -- the reference mirrors array positions, which
// is the conjugate frequency only for odd sizes).
__global__ __launch_bounds__(NT) void apply_mask_kernel(const uint8_t *__restrict__ m, quad_geom g,
                                                        int second, bool vec, float2 *F,
                                                        uint8_t *full)
{
    for_nonzero(m, (size_t)g.qh * g.qw, vec, [&](size_t idx, uint8_t) {
        const int i = (int)(idx / (size_t)g.qw), j = (int)(idx - (size_t)i * g.qw);
        const int X = g.col0(second) + j;
        const int mi = g.ny - 1 - i, mX = g.nx - 1 - X;
        F[(size_t)unshift(i, g.ny) * g.nx + unshift(X, g.nx)] = make_float2(0.0f, 0.0f);
        F[(size_t)unshift(mi, g.ny) * g.nx + unshift(mX, g.nx)] = make_float2(0.0f, 0.0f);
        if (full) {
            full[(size_t)i * g.nx + X] = 1;
            full[(size_t)mi * g.nx + mX] = 1;
        }
    });
}

// |F * scale + mean|: the complex inverse transform back to metres, 4 cells per lane
// where the count allows (F is 16-byte aligned: the library's own buffer).
__global__ __launch_bounds__(NT) void abs_scale_kernel(const float2 *__restrict__ F, size_t n,
                                                       double scale,
                                                       const double *__restrict__ mean,
                                                       float *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * NT + threadIdx.x) * 4;
    if (i >= n) return;
    const double mu = *mean;
    auto mag = [&](float re_, float im_) {
        const double re = (double)re_ * scale + mu, im = (double)im_ * scale;
        return (float)sqrt(re * re + im * im);
    };
    if (i + 4 <= n && ((uintptr_t)out & 15) == 0) {
        const float4 a = *(const float4 *)(F + i), b = *(const float4 *)(F + i + 2);
        *(float4 *)(out + i) = make_float4(mag(a.x, a.y), mag(a.z, a.w), mag(b.x, b.y), mag(b.z, b.w));
    } else {
        for (size_t k = i; k < n && k < i + 4; ++k) out[k] = mag(F[k].x, F[k].y);
    }
}

// The inverse transform of the destripe, second half.  rocFFT's 2-D plan is rows,
// transpose, rows, transpose; the last transpose only feeds a point-wise kernel, so the
// destripe runs the two row passes as batched 1-D plans and does the data movement
// itself: transpose_kernel between them, and transpose_abs_kernel behind them, which
// reads the transposed result and writes |F / N + mean| in raster order (the two routes
// agree bit for bit; 3.35 against 4.3 ms at 16384^2, tools/micro/fft_split.hip).
constexpr int TT = 32;

// G[x][y] = F[y][x]   (F: H rows of W, G: W rows of H)
__global__ __launch_bounds__(NT) void transpose_kernel(const float2 *__restrict__ F, int H, int W,
                                                       float2 *__restrict__ G)
{
    __shared__ float2 t[TT][TT + 1];
    const int x0 = blockIdx.x * TT, y0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8 threads
#pragma unroll
    for (int k = 0; k < TT; k += NT / 32) {
        const int y = y0 + ty + k, x = x0 + tx;
        if (y < H && x < W) t[ty + k][tx] = F[(size_t)y * W + x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TT; k += NT / 32) {
        const int x = x0 + ty + k, y = y0 + tx;
        if (y < H && x < W) G[(size_t)x * H + y] = t[tx][ty + k];
    }
}

// out[y][x] = |G[x][y] * scale + mean|, evaluated in double like abs_scale_kernel
__global__ __launch_bounds__(NT) void transpose_abs_kernel(const float2 *__restrict__ G, int H, int W,
                                                           double scale,
                                                           const double *__restrict__ mean,
                                                           float *__restrict__ out)
{
    __shared__ float2 t[TT][TT + 1];
    const int x0 = blockIdx.x * TT, y0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < TT; k += NT / 32) {
        const int x = x0 + ty + k, y = y0 + tx;
        if (y < H && x < W) t[ty + k][tx] = G[(size_t)x * H + y];
    }
    __syncthreads();
    const double mu = *mean;
#pragma unroll
    for (int k = 0; k < TT; k += NT / 32) {
        const int y = y0 + ty + k, x = x0 + tx;
        if (y < H && x < W) {
            const float2 v = t[tx][ty + k];
            const double re = (double)v.x * scale + mu, im = (double)v.y * scale;
            out[(size_t)y * W + x] = (float)sqrt(re * re + im * im);
        }
    }
}

inline dim3 grid2(int w, int h) { return dim3((unsigned)((w + NT - 1) / NT), (unsigned)h); }

int check_window(int window, int h, int w)
{
    // SlidingWindow.window_size setter (sliding_window.py:150-156): same order of checks
    if (window > h || window > w) {
        hdem_set_error("Window size: %d cannot be higher than grid dimensions: (%d, %d)", window, h, w);
        return HDEM_ERR_WINDOW_HIGH;
    }
    if (window % 2 != 1) {
        hdem_set_error("Window size: %d cannot be an even number", window);
        return HDEM_ERR_WINDOW_EVEN;
    }
    return HDEM_OK;
}

// One BlanksFourier pass on a device quadrant: q_out <- q with the peaks zeroed (NULL: not
// needed), found / total as in detect_kernel.
// occ_mark / occ_skip: ((h + 31) / 32) x ((w + 31) / 32) bytes, see the kernel; either may be NULL.
int blanks_pass(hdem_ctx *ctx, const float *q, int h, int w, float *q_out, uint8_t *found,
                uint8_t *total, uint8_t *occ_mark = nullptr, const uint8_t *occ_skip = nullptr,
                int window = 55)
{
    const int R = window / 2;
    const int per_block4 = H4W - 2 * R, bx4 = (w + per_block4 - 1) / per_block4;
    // rows one block walks.  A block's rows are a serial chain (scan, barrier, LDS, ~2.6 us per
    // row with three waves per SIMD sharing the issue slots), so what counts is the number of
    // chains: the launch takes 0.22 ms + 1.8 us per row of a block at 8182^2 -- 0.27 ms at 32
    // rows, 0.33 at 64, 0.68 at 256 -- although a short block reads its 2R start-up rows for few
    // outputs (0.31 ms at 16 rows).
    int seg4 = 256;
    while (seg4 > 32 && (int64_t)bx4 * ((h + seg4 - 1) / seg4) < 4096) seg4 /= 2;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_DETECT, (int64_t)h * w);
        const dim3 grid(bx4, (h + seg4 - 1) / seg4);
        if (window == 55)
            hipLaunchKernelGGL((hollow_detect4_kernel<27, 2>), grid, dim3(H4T), 0, ctx->stream, q, h,
                               w, 4.0f, seg4, q_out, found, total, occ_mark, occ_skip, (w + 31) / 32, 27);
        else
            hipLaunchKernelGGL((hollow_detect4_kernel<0, 2>), grid, dim3(H4T), 0, ctx->stream, q, h,
                               w, 4.0f, seg4, q_out, found, total, occ_mark, occ_skip, (w + 31) / 32, R);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int hdem_blanks_fourier_f32_dev(hdem_ctx *ctx, float *q, int h, int w, int window,
                                           uint8_t *found)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(q, found, h, w)) return rc;
    if (int rc = check_window(window, h, w)) return rc;
    // (the inner window left out is 5 x 5; a block of 256 columns keeps `window - 1` of them
    // as context)
    HDEM_REQUIRE(window >= 7 && window <= 201, HDEM_ERR_BAD_ARG,
                 "BlanksFourier window must be 7..201, got %d", window);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)h * w * sizeof(float);
    float *tmp = (float *)hdem_arena(ctx, bytes);          // the pass is out of place
    if (!tmp) return HDEM_ERR_OOM;
    if (int rc = blanks_pass(ctx, q, h, w, tmp, found, nullptr, nullptr, nullptr, window)) return rc;
    HDEM_HIP_CHECK(hipMemcpyAsync(q, tmp, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return HDEM_OK;
}

extern "C" int hdem_isolated_points_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int h, int w,
                                           int window, uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(mask, out, h, w)) return rc;
    HDEM_REQUIRE(mask != out, HDEM_ERR_BAD_ARG, "isolated points cannot run in place");
    if (int rc = check_window(window, h, w)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_MASK, (int64_t)h * w);
        HDEM_HIP_CHECK(hipMemcpyAsync(out, mask, (size_t)h * w, hipMemcpyDeviceToDevice,
                                      ctx->stream));
        hipLaunchKernelGGL(isolated_kernel, grid16((size_t)h * w), dim3(NT), 0, ctx->stream, mask,
                           h, w, window / 2, aligned16(mask), out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_expand_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int h, int w, int window,
                                  uint8_t *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(mask, out, h, w)) return rc;
    HDEM_REQUIRE(mask != out, HDEM_ERR_BAD_ARG, "expand cannot run in place");
    if (int rc = check_window(window, h, w)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    HDEM_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)h * w, ctx->stream));
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_MASK, (int64_t)h * w);
        const int reach = window / 2;
        if (reach == 6)
            hipLaunchKernelGGL(expand_kernel<6>, grid16((size_t)h * w), dim3(NT), 0, ctx->stream,
                               mask, h, w, reach, aligned16(mask), out);
        else if (reach == 3)
            hipLaunchKernelGGL(expand_kernel<3>, grid16((size_t)h * w), dim3(NT), 0, ctx->stream,
                               mask, h, w, reach, aligned16(mask), out);
        else
            hipLaunchKernelGGL(expand_kernel<0>, grid16((size_t)h * w), dim3(NT), 0, ctx->stream,
                               mask, h, w, reach, aligned16(mask), out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_fft2_c2c_f32_dev(hdem_ctx *ctx, float *data, int H, int W, int inverse)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(data, data, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = ensure_plans(ctx, H, W)) return rc;
    return run_fft(ctx, inverse != 0, (float2 *)data);
}

// The same transform in double precision, for float64 / complex128 input -- what
// scipy.fftpack.fft2 / ifft2 compute for it (extension_filters.py:379,414).  A standalone
// operator off the pipeline's path (the pipeline transforms float32 rasters): plan and work
// buffer are made for the call and released behind it.
extern "C" int hdem_fft2_c2c_f64_dev(hdem_ctx *ctx, double *data, int H, int W, int inverse)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(data, data, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = load_rocfft()) return rc;
    const size_t lengths[2] = {(size_t)W, (size_t)H};
    rocfft_plan plan = nullptr;
    HDEM_REQUIRE(g_fft.plan_create(&plan, ROCFFT_INPLACE,
                                   inverse ? ROCFFT_COMPLEX_INVERSE : ROCFFT_COMPLEX_FORWARD,
                                   ROCFFT_DOUBLE, 2, lengths, 1, nullptr) == 0,
                 HDEM_ERR_HIP, "rocfft_plan_create (double, %d x %d) failed", H, W);
    size_t work_bytes = 0;
    g_fft.plan_get_work_buffer_size(plan, &work_bytes);
    rocfft_execution_info info = nullptr;
    hdem_dbuf work;
    int rc = HDEM_OK;
    if (g_fft.info_create(&info) != 0) rc = HDEM_ERR_HIP;
    if (!rc && work_bytes) rc = work.alloc(ctx, work_bytes);
    if (!rc && work_bytes && g_fft.info_set_work_buffer(info, work.p, work_bytes) != 0) rc = HDEM_ERR_HIP;
    if (!rc && g_fft.info_set_stream(info, ctx->stream) != 0) rc = HDEM_ERR_HIP;
    if (!rc) {
        void *in[1] = {data};
        hdem_scoped_timer tm(ctx, HDEM_K_FFT, (int64_t)H * W);
        if (g_fft.execute(plan, in, nullptr, info) != 0) rc = HDEM_ERR_HIP;
    }
    // (the plan's kernels are on the stream; the plan and its work buffer go behind them)
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) rc = HDEM_ERR_HIP;
    if (info) g_fft.info_destroy(info);
    g_fft.plan_destroy(plan);
    if (rc == HDEM_ERR_HIP) hdem_set_error("rocFFT double-precision transform (%d x %d) failed", H, W);
    return rc;
}

extern "C" int hdem_fourier_destripe_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W,
                                             float *out, uint8_t *mask)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    const quad_geom g = make_geom(H, W);
    // what the reference's window constructors would raise on the quadrants
    if (g.qh < 1 || g.qw < 1) {
        hdem_set_error("Window size: %d cannot be higher than grid dimensions: (%d, %d)", 55,
                       g.qh > 0 ? g.qh : 0, g.qw > 0 ? g.qw : 0);
        return HDEM_ERR_WINDOW_HIGH;
    }
    if (int rc = check_window(55, g.qh, g.qw)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    if (int rc = ensure_plans(ctx, H, W)) return rc;
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)H * W, qn = (size_t)g.qh * g.qw;
    // one allocation, carved up: spectrum | quadrant and its copy without the first pass's
    // peaks | 4 byte masks | partial sums + mean
    const size_t qn8 = (qn + 15) / 16 * 16;     // (keeps every view 16-byte aligned)
    const size_t occ_bytes = ((size_t)((g.qh + 31) / 32) * ((g.qw + 31) / 32) + 15) / 16 * 16;
    const size_t bytes = n * sizeof(float2) + 2 * qn8 * sizeof(float) + 4 * qn8 + occ_bytes +
                         (SUM_BLOCKS + 1) * sizeof(double);
    hdem_fourier_state *fs = ctx->fourier;
    if (!fs->scratch)
        if (int rc = hdem_raw_alloc(ctx, bytes, &fs->scratch)) return rc;
    char *base = (char *)fs->scratch;
    struct view { void *p; } F{base}, q{base + n * sizeof(float2)},
        q1{(char *)q.p + qn8 * sizeof(float)}, det{(char *)q1.p + qn8 * sizeof(float)},
        iso{(char *)det.p + qn8},
        exp{(char *)iso.p + qn8}, exp2{(char *)exp.p + qn8};
    uint8_t *occ = (uint8_t *)exp2.p + qn8;                 // 32 x 32 blocks with first-pass peaks
    double *partial = (double *)(occ + occ_bytes), *mean = partial + SUM_BLOCKS;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_POINT, (int64_t)n);
        const size_t step = n >> 20 ? n >> 20 : 1;
        hipLaunchKernelGGL(sum_partial_kernel, dim3(SUM_BLOCKS), dim3(NT), 0, st, dem, n, step,
                           partial);
        hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(NT), 0, st, (const double *)partial,
                           (n + step - 1) / step, mean);
        // dem - mean goes through the caller's output raster, free until the last kernel
        hipLaunchKernelGGL(to_real_kernel, dim3((unsigned)((n + 4 * NT - 1) / (4 * NT))), dim3(NT),
                           0, st, dem, n, (const double *)mean, out);
    }
    if (int rc = run_r2c(ctx, out, (float2 *)F.p)) return rc;
    if (W > 2) {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_POINT, (int64_t)n);
        hipLaunchKernelGGL(fill_right_half_kernel, grid2(W - W / 2 - 1, H), dim3(NT), 0, st,
                           (float2 *)F.p, H, W);
    }
    if (mask) HDEM_HIP_CHECK(hipMemsetAsync(mask, 0, n, st));
    // both quadrants are detected on the untouched spectrum, then both are applied
    for (int second = 0; second < 2; ++second) {
        uint8_t *e = (uint8_t *)(second ? exp2.p : exp.p);
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_POINT, (int64_t)qn);
            hipLaunchKernelGGL(quadrant_abs_kernel, grid2(g.qw, g.qh), dim3(NT), 0, st,
                               (const float2 *)F.p, g, second, (float *)q.p);
        }
        HDEM_HIP_CHECK(hipMemsetAsync(det.p, 0, qn, st));
        HDEM_HIP_CHECK(hipMemsetAsync(occ, 0, occ_bytes, st));
        if (int rc = blanks_pass(ctx, (const float *)q.p, g.qh, g.qw, (float *)q1.p, nullptr,
                                 (uint8_t *)det.p, occ, nullptr))
            return rc;
        if (int rc = blanks_pass(ctx, (const float *)q1.p, g.qh, g.qw, nullptr, nullptr,
                                 (uint8_t *)det.p, nullptr, occ))
            return rc;
        HDEM_HIP_CHECK(hipMemsetAsync(e, 0, qn, st));
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_MASK, (int64_t)qn * 2);
            HDEM_HIP_CHECK(hipMemcpyAsync(iso.p, det.p, qn, hipMemcpyDeviceToDevice, st));
            hipLaunchKernelGGL(isolated_kernel, grid16(qn), dim3(NT), 0, st,
                               (const uint8_t *)det.p, g.qh, g.qw, 1, aligned16(det.p),
                               (uint8_t *)iso.p);
            hipLaunchKernelGGL(expand_kernel<6>, grid16(qn), dim3(NT), 0, st,
                               (const uint8_t *)iso.p, g.qh, g.qw, 6, aligned16(iso.p), e);
        }
    }
    for (int second = 0; second < 2; ++second) {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_MASK, (int64_t)qn);
        const uint8_t *e = (const uint8_t *)(second ? exp2.p : exp.p);
        hipLaunchKernelGGL(apply_mask_kernel, grid16(qn), dim3(NT), 0, st, e, g, second,
                           aligned16(e), (float2 *)F.p, mask);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    const double scale = 1.0 / ((double)H * (double)W);
    if (fs->inv_rows && fs->inv_cols && !getenv("HDEM_FFT_2D_INVERSE")) {   // (env: experiments)
        // the transposed intermediate: rocFFT's work buffer when it is large enough (the
        // 1-D plans need none of it), else a buffer of its own
        float2 *G = !fs->split_needs_work && fs->work_bytes >= n * sizeof(float2) ? (float2 *)fs->work
                                                                                  : (float2 *)fs->tr;
        if (!G) {
            if (int rc = hdem_raw_alloc(ctx, n * sizeof(float2), (void **)&fs->tr)) return rc;
            G = (float2 *)fs->tr;
        }
        HDEM_REQUIRE(g_fft.info_set_stream(fs->info, st) == 0, HDEM_ERR_HIP, "rocfft set_stream failed");
        const dim3 tiles((W + TT - 1) / TT, (H + TT - 1) / TT);
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FFT, (int64_t)n);
            void *rows[1] = {F.p}, *cols[1] = {G};
            HDEM_REQUIRE(g_fft.execute(fs->inv_rows, rows, nullptr, fs->info) == 0, HDEM_ERR_HIP,
                         "rocfft_execute (inverse, rows) failed");
            hipLaunchKernelGGL(transpose_kernel, tiles, dim3(NT), 0, st, (const float2 *)F.p, H, W, G);
            HDEM_REQUIRE(g_fft.execute(fs->inv_cols, cols, nullptr, fs->info) == 0, HDEM_ERR_HIP,
                         "rocfft_execute (inverse, columns) failed");
        }
        {
            hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_POINT, (int64_t)n);
            hipLaunchKernelGGL(transpose_abs_kernel, tiles, dim3(NT), 0, st, (const float2 *)G, H, W,
                               scale, (const double *)mean, out);
        }
        HDEM_HIP_CHECK(hipGetLastError());
        return HDEM_OK;
    }
    if (int rc = run_fft(ctx, true, (float2 *)F.p)) return rc;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_FOURIER_POINT, (int64_t)n);
        hipLaunchKernelGGL(abs_scale_kernel, dim3((unsigned)((n + 4 * NT - 1) / (4 * NT))),
                           dim3(NT), 0, st, (const float2 *)F.p, n, scale, (const double *)mean, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_fourier_destripe_f32(hdem_ctx *ctx, const float *dem, int H, int W, float *out,
                                         uint8_t *mask)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    hdem_dbuf d_in, d_out, d_mask;
    if (int rc = d_in.alloc(ctx, n * sizeof(float))) return rc;
    if (int rc = d_out.alloc(ctx, n * sizeof(float))) return rc;
    if (mask)
        if (int rc = d_mask.alloc(ctx, n)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, d_in.p, dem, n * sizeof(float))) return rc;
    if (int rc = hdem_fourier_destripe_f32_dev(ctx, (const float *)d_in.p, H, W, (float *)d_out.p,
                                               mask ? (uint8_t *)d_mask.p : nullptr))
        return rc;
    if (mask)
        if (int rc = hdem_memcpy_d2h(ctx, mask, d_mask.p, n)) return rc;
    return hdem_memcpy_d2h(ctx, out, d_out.p, n * sizeof(float));
}
