// Element-wise operators of the seam on device rasters: the bodies of
// LowerThan / GreaterThan / BooleanToInteger / ProductFilter / AdditionFilter /
// SubtractionFilter.apply (`cguerrero/hydrodem/filters/simple_filters.py:7-275`), which the
// orchestration composes directly (`hydro_dem_process.py:60-91`: mask addition, 1 - mask,
// two products, then the three-term sum in front of PostProcessingFinal, `:148-149`).
//
// One kernel: out[i] = op(image[i], operand[i] or scalar).  Rasters are float32, float64,
// uint8 (masks) or int64 (what NumPy makes of `mask * 1`; exact below 2^53); the arithmetic is done in double -- what NumPy does for the float64 rasters
// the pipeline holds at that point (float32 * int64 promotes), exact for float32 and mask
// inputs -- and stored in the type the caller asks for.  HBM-bound: 4 cells per lane, one
// vector load per operand.
#include "hdem_internal.h"

namespace {

constexpr int NT = 256;

struct ew_args {
    const void *image, *operand;
    void *out;
    int image_type, operand_type, out_type, op;
    double scalar;
    int64_t n;
};

__device__ __forceinline__ void load4(const void *p, int type, int64_t i, int64_t n, double (&v)[4])
{
    if (i + 4 <= n) {
        if (type == HDEM_T_F32) {
            const hdem_f4 q = hdem_ld4u(static_cast<const float *>(p) + i);
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
        } else if (type == HDEM_T_F64) {
            const double *d = static_cast<const double *>(p) + i;
            v[0] = d[0]; v[1] = d[1]; v[2] = d[2]; v[3] = d[3];
        } else if (type == HDEM_T_I64) {
            const int64_t *d = static_cast<const int64_t *>(p) + i;
            v[0] = (double)d[0]; v[1] = (double)d[1]; v[2] = (double)d[2]; v[3] = (double)d[3];
        } else {
            const uint8_t *b = static_cast<const uint8_t *>(p) + i;
            v[0] = b[0]; v[1] = b[1]; v[2] = b[2]; v[3] = b[3];
        }
        return;
    }
    for (int k = 0; k < 4; ++k) {
        const int64_t j = i + k < n ? i + k : n - 1;
        v[k] = type == HDEM_T_F32   ? (double)static_cast<const float *>(p)[j]
               : type == HDEM_T_F64 ? static_cast<const double *>(p)[j]
               : type == HDEM_T_I64 ? (double)static_cast<const int64_t *>(p)[j]
                                    : (double)static_cast<const uint8_t *>(p)[j];
    }
}

__global__ __launch_bounds__(NT) void elementwise_kernel(ew_args a)
{
    const int64_t i = ((int64_t)blockIdx.x * NT + threadIdx.x) * 4;
    if (i >= a.n) return;
    double x[4], y[4];
    load4(a.image, a.image_type, i, a.n, x);
    if (a.operand) load4(a.operand, a.operand_type, i, a.n, y);
    else y[0] = y[1] = y[2] = y[3] = a.scalar;
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        switch (a.op) {
        case HDEM_EW_MUL: r[k] = y[k] * x[k]; break;          // factor * image
        case HDEM_EW_ADD: r[k] = y[k] + x[k]; break;          // addend + image
        case HDEM_EW_RSUB: r[k] = y[k] - x[k]; break;         // minuend - image
        case HDEM_EW_GT: r[k] = x[k] > y[k] ? 1.0 : 0.0; break;
        case HDEM_EW_LT: r[k] = x[k] < y[k] ? 1.0 : 0.0; break;
        default: r[k] = x[k] != 0.0 ? 1.0 : 0.0; break;       // HDEM_EW_NONZERO
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i + k >= a.n) break;
        if (a.out_type == HDEM_T_F32) static_cast<float *>(a.out)[i + k] = (float)r[k];
        else if (a.out_type == HDEM_T_F64) static_cast<double *>(a.out)[i + k] = r[k];
        else if (a.out_type == HDEM_T_I64) static_cast<int64_t *>(a.out)[i + k] = (int64_t)r[k];
        else static_cast<uint8_t *>(a.out)[i + k] = (uint8_t)r[k];
    }
}

bool known_type(int t) { return t >= HDEM_T_F32 && t <= HDEM_T_I64; }

}  // namespace

extern "C" int hdem_elementwise_dev(hdem_ctx *ctx, int op, const void *image, int image_type,
                                    const void *operand, int operand_type, double scalar,
                                    int64_t n, void *out, int out_type)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    HDEM_REQUIRE(image && out, HDEM_ERR_BAD_ARG, "null raster pointer");
    HDEM_REQUIRE(n > 0, HDEM_ERR_BAD_ARG, "element count must be positive, got %lld", (long long)n);
    HDEM_REQUIRE(op >= HDEM_EW_MUL && op <= HDEM_EW_NONZERO, HDEM_ERR_BAD_ARG,
                 "unknown element-wise operator %d", op);
    HDEM_REQUIRE(known_type(image_type) && known_type(out_type) &&
                     (!operand || known_type(operand_type)),
                 HDEM_ERR_BAD_ARG, "unknown raster type (%d, %d, %d)", image_type, operand_type,
                 out_type);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    ew_args a = {image, operand, out, image_type, operand_type, out_type, op, scalar, n};
    const int64_t quads = (n + 3) / 4;
    HDEM_REQUIRE((quads + NT - 1) / NT < 0x7fffffffll, HDEM_ERR_BAD_ARG, "raster too large");
    {
        hdem_scoped_timer tm(ctx, HDEM_K_ELEMENTWISE, n);
        hipLaunchKernelGGL(elementwise_kernel, dim3((unsigned)((quads + NT - 1) / NT)), dim3(NT),
                           0, ctx->stream, a);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}
