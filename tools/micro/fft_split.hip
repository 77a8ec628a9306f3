// The complex inverse of the destripe as two batched 1-D rocFFT passes with own transposes:
//   rows (in place) -> transpose F -> G -> rows of G (in place) -> fused transpose + |.| -> float
// against rocFFT's 2-D in-place plan followed by the abs kernel.  Exploration.
// build: hipcc -O3 --offload-arch=gfx950 -o fft_split fft_split.hip -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { if ((x) != 0) { printf("failed: %s (line %d)\n", #x, __LINE__); exit(1); } } while (0)

constexpr int TT = 32;

// G[x][y] = F[y][x]   (F: H rows of W, G: W rows of H)
__global__ __launch_bounds__(256) void transpose_c(const float2 *__restrict__ F, int H, int W,
                                                   float2 *__restrict__ G)
{
    __shared__ float2 t[TT][TT + 1];
    const int x0 = blockIdx.x * TT, y0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
    for (int k = 0; k < TT; k += 8) {
        const int y = y0 + ty + k, x = x0 + tx;
        if (y < H && x < W) t[ty + k][tx] = F[(size_t)y * W + x];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TT; k += 8) {
        const int x = x0 + ty + k, y = y0 + tx;
        if (y < H && x < W) G[(size_t)x * H + y] = t[tx][ty + k];
    }
}

// out[y][x] = |G[x][y] * scale + mean|
__global__ __launch_bounds__(256) void transpose_abs(const float2 *__restrict__ G, int H, int W,
                                                     double scale, double mean, float *__restrict__ out)
{
    __shared__ float2 t[TT][TT + 1];
    const int x0 = blockIdx.x * TT, y0 = blockIdx.y * TT;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < TT; k += 8) {
        const int x = x0 + ty + k, y = y0 + tx;
        if (y < H && x < W) t[ty + k][tx] = G[(size_t)x * H + y];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TT; k += 8) {
        const int y = y0 + ty + k, x = x0 + tx;
        if (y < H && x < W) {
            const float2 v = t[tx][ty + k];
            const double re = (double)v.x * scale + mean, im = (double)v.y * scale;
            out[(size_t)y * W + x] = (float)sqrt(re * re + im * im);
        }
    }
}

__global__ void abs_scale(const float2 *F, size_t n, double scale, double mean, float *out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double re = (double)F[i].x * scale + mean, im = (double)F[i].y * scale;
    out[i] = (float)sqrt(re * re + im * im);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const size_t N = (size_t)n * n;
    CK(rocfft_setup());
    float2 *F, *G, *F2; float *out, *out2;
    hipMalloc(&F, N * 8); hipMalloc(&G, N * 8); hipMalloc(&F2, N * 8); hipMalloc(&out, N * 4); hipMalloc(&out2, N * 4);
    std::vector<float2> h(N);
    unsigned s = 12345;
    for (size_t i = 0; i < N; ++i) { s = s * 1664525u + 1013904223u; h[i].x = (s >> 8) * 1e-7f; s = s * 1664525u + 1013904223u; h[i].y = (s >> 8) * 1e-7f; }
    hipMemcpy(F, h.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(F2, h.data(), N * 8, hipMemcpyHostToDevice);
    const size_t len2[2] = {(size_t)n, (size_t)n}, len1[1] = {(size_t)n};
    rocfft_plan p2d, p1d;
    CK(rocfft_plan_create(&p2d, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single, 2, len2, 1, nullptr));
    CK(rocfft_plan_create(&p1d, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single, 1, len1, n, nullptr));
    size_t w2 = 0, w1 = 0;
    rocfft_plan_get_work_buffer_size(p2d, &w2); rocfft_plan_get_work_buffer_size(p1d, &w1);
    printf("work buffers: 2-D %.0f MB, 1-D batched %.0f MB\n", w2 / 1e6, w1 / 1e6);
    void *work = nullptr; hipMalloc(&work, w2 > w1 ? w2 : (w1 ? w1 : 16));
    rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
    CK(rocfft_execution_info_set_work_buffer(info, work, w2 > w1 ? w2 : (w1 ? w1 : 16)));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((n + TT - 1) / TT, (n + TT - 1) / TT);
    const double scale = 1.0 / ((double)n * n), mean = 100.0;
    for (int rep = 0; rep < 3; ++rep) {
        void *io[1] = {F2};
        hipMemcpy(F2, h.data(), N * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        CK(rocfft_execute(p2d, io, nullptr, info));
        abs_scale<<<(unsigned)((N + 255) / 256), 256>>>(F2, N, scale, mean, out2);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(F, h.data(), N * 8, hipMemcpyHostToDevice);
        void *a[1] = {F}, *b[1] = {G};
        float t[5];
        hipEvent_t ev[5]; for (auto &e : ev) hipEventCreate(&e);
        hipEventRecord(ev[0]);
        CK(rocfft_execute(p1d, a, nullptr, info));
        hipEventRecord(ev[1]);
        transpose_c<<<grid, 256>>>(F, n, n, G);
        hipEventRecord(ev[2]);
        CK(rocfft_execute(p1d, b, nullptr, info));
        hipEventRecord(ev[3]);
        transpose_abs<<<grid, 256>>>(G, n, n, scale, mean, out);
        hipEventRecord(ev[4]); hipEventSynchronize(ev[4]);
        for (int k = 0; k < 4; ++k) hipEventElapsedTime(&t[k], ev[k], ev[k + 1]);
        printf("2-D plan + abs: %.3f ms | split: rows %.3f + transpose %.3f + rows %.3f + transpose-abs %.3f = %.3f ms\n",
               ms, t[0], t[1], t[2], t[3], t[0] + t[1] + t[2] + t[3]);
    }
    std::vector<float> r1(N), r2(N);
    hipMemcpy(r1.data(), out, N * 4, hipMemcpyDeviceToHost); hipMemcpy(r2.data(), out2, N * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < N; i += 997) md = fmax(md, fabs((double)r1[i] - r2[i]));
    printf("max difference between the two routes (sampled): %g\n", md);
    return 0;
}
