// Which form of write-back suits the tile visit (exploration).  Every wave writes
// the 62 x 62 interior of "its" tile of a 16384^2 float raster (tiles at multiples of 62 cells,
// as in the fill), in four forms:
//   dword      one dword per lane, a 248-byte row per instruction (what the visit does), sc1;
//   dword_pl   the same, plain stores;
//   x4         16 bytes per lane: 16 lanes cover a row (15 quads + one pair), 4 rows per
//              instruction, sc1;
//   x4_pl      the same, plain.
// build: hipcc -O3 --offload-arch=gfx950 -o store_form store_form.hip ; run: ./store_form
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int FT = 62, AUX_SC1 = 16;
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

template <int AUX>
__global__ __launch_bounds__(64) void k_dword(float *w, int W, int tiles_x, int ntiles)
{
    const int t = blockIdx.x, ty = t / tiles_x, tx = t - ty * tiles_x, lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        w + (size_t)ty * FT * W, 0, (int)(unsigned)((size_t)64 * W * 4), 0x00020000);
    if (lane >= 1 && lane <= FT)
#pragma unroll
        for (int r = 1; r <= FT; ++r)
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + r), rs, (unsigned)((tx * FT + lane) * 4),
                                                  (unsigned)r * (unsigned)W * 4u, AUX);
}

// the same 62 rows, but 256 bytes each at multiples of 256 bytes: whole 128-byte lines
template <int AUX>
__global__ __launch_bounds__(64) void k_aligned(float *w, int W, int tiles_x, int ntiles)
{
    const int t = blockIdx.x, ty = t / tiles_x, tx = t - ty * tiles_x, lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        w + (size_t)ty * FT * W, 0, (int)(unsigned)((size_t)64 * W * 4), 0x00020000);
#pragma unroll
    for (int r = 1; r <= FT; ++r)
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)(t + r), rs, (unsigned)((tx * 64 + lane) * 4),
                                              (unsigned)r * (unsigned)W * 4u, AUX);
}

template <int AUX>
__global__ __launch_bounds__(64) void k_x4(float *w, int W, int tiles_x, int ntiles)
{
    const int t = blockIdx.x, ty = t / tiles_x, tx = t - ty * tiles_x, lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        w + (size_t)ty * FT * W, 0, (int)(unsigned)((size_t)64 * W * 4), 0x00020000);
    const int q = lane & 15, sub = lane >> 4;           // quad of the row, row of the group of 4
    const unsigned col = (unsigned)((tx * FT + 1 + 4 * q) * 4);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const int r = 1 + 4 * g + sub;
        if (r > FT) break;
        const unsigned so = (unsigned)r * (unsigned)W * 4u;
        const u4 v = {(unsigned)t, (unsigned)r, (unsigned)q, 0u};
        if (q < 15) __builtin_amdgcn_raw_buffer_store_b128(v, rs, col, so, AUX);
        else __builtin_amdgcn_raw_buffer_store_b64((u2){(unsigned)t, (unsigned)r}, rs, col, so, AUX);
    }
}

int main()
{
    const int W = 16384, H = 16384, tiles_x = (W - 2 + FT - 1) / FT, tiles_y = (H - 2 + FT - 1) / FT;
    const int ntiles = (tiles_x - 1) * (tiles_y - 1);          // (whole tiles only)
    float *w;
    CK(hipMalloc(&w, (size_t)H * W * 4 + (1 << 20)));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)ntiles * FT * FT * 4;
    for (int rep = 0; rep < 3; ++rep) {
        float ms[4];
#define RUN(i, K) hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(ntiles), dim3(64), 0, 0, w, W, tiles_x - 1, ntiles); \
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[i], e0, e1);
        RUN(0, k_dword<AUX_SC1>) RUN(1, k_dword<0>) RUN(2, k_x4<AUX_SC1>) RUN(3, k_x4<0>)
        {
            // aligned: 256 tiles per row of tiles (16384 / 64), same number of rows of tiles
            float a, b;
            const int atx = W / 64, an = atx * (tiles_y - 1);
            hipEventRecord(e0); hipLaunchKernelGGL(k_aligned<AUX_SC1>, dim3(an), dim3(64), 0, 0, w, W, atx, an);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&a, e0, e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k_aligned<0>, dim3(an), dim3(64), 0, 0, w, W, atx, an);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&b, e0, e1);
            const double ab = (double)an * FT * 256;
            printf("   aligned 256-byte rows: sc1 %.3f ms (%.0f GB/s)  plain %.3f (%.0f)\n", a, ab / a / 1e6, b, ab / b / 1e6);
        }
        printf("dword sc1 %.3f ms (%.0f GB/s)  dword plain %.3f (%.0f)  x4 sc1 %.3f (%.0f)  x4 plain %.3f (%.0f)\n",
               ms[0], bytes / ms[0] / 1e6, ms[1], bytes / ms[1] / 1e6, ms[2], bytes / ms[2] / 1e6, ms[3], bytes / ms[3] / 1e6);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
