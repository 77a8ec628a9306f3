// Block-maximum coarsening of a DEM: out[i][j] = max of z over fine rows
// [i*b, (i+1)*b) and columns [j*b, (j+1)*b), clipped to the raster.  New work (the
// reference is single process): the multi-GPU sink fill solves the global problem on
// this coarse grid first -- every fine path from a cell to the raster border that stays
// inside a chain of adjacent blocks has a maximum <= the maximum over those blocks, so
// the filled coarse surface is an upper bound of the filled fine surface, which is all a
// start value of the relaxation has to be (hydrodem_amd/partition.py).
// A block holding a nodata cell (NaN) is a wall: FLT_MAX, finite so that the coarse
// solve treats it as an ordinary (very high) cell, not as nodata with pinned neighbours.
//
// HBM-bound: reads z once (4 B/cell), writes 4/b^2 B/cell.  One lane = 4 adjacent
// columns of one band of b rows: b 16-byte loads, then a max across the b/4 lanes of
// the block with DPP-friendly shuffles.
#include <cfloat>

#include "hdem_internal.h"

namespace {

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void blockmax_kernel(const float *__restrict__ z, int H, int W,
                                                      int b, int cw, float *__restrict__ out)
{
    const int quads = (W + 3) / 4;
    const int q = blockIdx.x * NT + threadIdx.x;          // which 4 columns
    const int band = blockIdx.y;                          // which b rows
    const int x = min(q, quads - 1) * 4;                  // lanes past the edge repeat the last
    const int y0 = band * b, y1 = min(y0 + b, H);
    float m = -FLT_MAX;
    for (int y = y0; y < y1; ++y) {
        const float *row = z + (size_t)y * W + x;
        float v[4];
        if (x + 4 <= W) {
            const hdem_f4 t = hdem_ld4u(row);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = row[min(k, W - 1 - x)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, v[k] != v[k] ? FLT_MAX : v[k]);
    }
    // lanes of one block are b/4 neighbours (b = 4 .. 256, a power of two: at most one wave)
    const int group = b / 4;
    for (int o = 1; o < group; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (q < quads && (threadIdx.x % group) == 0) {
        const int j = x / b;
        if (j < cw) out[(size_t)band * cw + j] = m;
    }
}

// the copy roof: one 16-byte streaming load and store per lane, one lane per vector (measured
// against grid-stride loops and 4 or 8 vectors in flight per lane: 6.6 TB/s of traffic this
// way, 4.7 - 6.2 the others)
__global__ __launch_bounds__(NT) void copy_kernel(const hdem_f4 *__restrict__ src,
                                                  hdem_f4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

extern "C" int hdem_copy_rate_dev(hdem_ctx *ctx, const void *src, void *dst, size_t bytes)
{
    HDEM_REQUIRE(ctx && src && dst, HDEM_ERR_BAD_ARG, "null argument");
    HDEM_REQUIRE(bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0,
                 HDEM_ERR_BAD_ARG, "copy rate: size and pointers must be multiples of 16");
    HDEM_REQUIRE(bytes / 16 / NT < 0x7fffffffull, HDEM_ERR_BAD_ARG, "copy rate: too large");
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = bytes / 16;
    if (!n) return HDEM_OK;
    {
        hdem_scoped_timer tm(ctx, HDEM_K_COPY, (int64_t)bytes);
        hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                           ctx->stream, (const hdem_f4 *)src, (hdem_f4 *)dst, n);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_blockmax_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, int b,
                                     float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(z, out, H, W)) return rc;
    HDEM_REQUIRE(b >= 4 && b <= 256 && (b & (b - 1)) == 0, HDEM_ERR_BAD_ARG,
                 "block size must be a power of two in 4..256, got %d", b);
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    const int quads = (W + 3) / 4, ch = (H + b - 1) / b, cw = (W + b - 1) / b;
    HDEM_REQUIRE(ch <= 65535, HDEM_ERR_BAD_ARG, "too many block rows (%d)", ch);
    {
        hdem_scoped_timer tm(ctx, HDEM_K_BLOCKMAX, (int64_t)H * W);
        hipLaunchKernelGGL(blockmax_kernel, dim3((quads + NT - 1) / NT, ch), dim3(NT), 0,
                           ctx->stream, z, H, W, b, cw, out);
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}
