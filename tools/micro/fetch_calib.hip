// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two load shapes this library uses
// (exploration; DESIGN 3.1b "The FETCH_SIZE factor").  Both kernels read exactly 1 GiB, once:
//   rows_dword   one dword per lane, a 256-byte row piece per wave instruction, 64 such rows
//                in flight per wave -- the shape of the sink fill's window loads;
//   stream_x4    one 16-byte vector per lane, grid-stride -- the shape of the streaming kernels.
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
// run:   rocprofv3 --output-format csv --pmc FETCH_SIZE -d out -o c -- ./fetch_calib
//        (FETCH_SIZE is in KB; bytes read = 1073741824 per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void rows_dword(const float *__restrict__ z, float *out, int W, int rows_per_wave)
{
    // wave b reads columns [64 (b % (W / 64)), +64) of rows_per_wave consecutive rows
    const int strips = W / 64, b = blockIdx.x, sx = b % strips, sy = b / strips;
    const float *p = z + (size_t)sy * rows_per_wave * W + sx * 64 + threadIdx.x;
    float acc = 0.f;
    for (int r0 = 0; r0 < rows_per_wave; r0 += 64) {
        float a[64];
#pragma unroll
        for (int r = 0; r < 64; ++r) a[r] = p[(size_t)(r0 + r) * W];
#pragma unroll
        for (int r = 0; r < 64; ++r) acc += a[r];
    }
    if (acc == 12345.678f) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_x4(const float4 *__restrict__ z, float *out, size_t n)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = z[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main()
{
    const int W = 16384, H = 16384;                    // 1 GiB of float32
    float *z, *out;
    CK(hipMalloc(&z, (size_t)H * W * 4));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(z, 0, (size_t)H * W * 4));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rows_dword, dim3((W / 64) * (H / 64)), dim3(64), 0, 0, z, out, W, 64);
        hipLaunchKernelGGL(stream_x4, dim3(8192), dim3(256), 0, 0, (const float4 *)z, out, (size_t)H * W / 4);
    }
    CK(hipDeviceSynchronize());
    printf("each kernel read %zu bytes\n", (size_t)H * W * 4);
    return 0;
}
