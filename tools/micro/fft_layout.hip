// Does rocFFT drop a transpose when the 2-D spectrum may stay transposed between the
// forward and the inverse transform?  Times four plans on an n x n raster (exploration):
//   r2c normal      real -> left half of the full spectrum, row stride n      (what the destripe uses)
//   r2c transposed  real -> the same half, stored x-major (stride n along x)
//   c2c inverse, in place, normal layout                                       (what the destripe uses)
//   c2c inverse, transposed input -> normal output, out of place
// build: hipcc -O2 --offload-arch=gfx950 -o fft_layout fft_layout.hip -lrocfft
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { if ((x) != 0) { printf("failed: %s (line %d)\n", #x, __LINE__); exit(1); } } while (0)

static float run(rocfft_plan p, void *in, void *out, void *work, size_t wbytes, int reps)
{
    rocfft_execution_info info;
    CK(rocfft_execution_info_create(&info));
    if (wbytes) CK(rocfft_execution_info_set_work_buffer(info, work, wbytes));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    void *ins[1] = {in}, *outs[1] = {out};
    CK(rocfft_execute(p, ins, outs, info));
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) CK(rocfft_execute(p, ins, outs, info));
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    rocfft_execution_info_destroy(info);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t n = argc > 1 ? atoi(argv[1]) : 16384;
    CK(rocfft_setup());
    float *real; float2 *F, *G;
    hipMalloc(&real, n * n * 4); hipMalloc(&F, n * n * 8); hipMalloc(&G, n * n * 8);
    hipMemset(real, 0, n * n * 4); hipMemset(F, 0, n * n * 8);
    const size_t len[2] = {n, n};
    void *work = nullptr; size_t wcap = 0;
    auto bench = [&](const char *name, rocfft_plan p, void *in, void *out) {
        size_t wb = 0;
        CK(rocfft_plan_get_work_buffer_size(p, &wb));
        if (wb > wcap) { if (work) hipFree(work); hipMalloc(&work, wb); wcap = wb; }
        printf("%-46s %.3f ms  (work buffer %.0f MB)\n", name, run(p, in, out, work, wb, 5), wb / 1e6);
        rocfft_plan_destroy(p);
    };
    rocfft_plan p;
    rocfft_plan_description d;
    {   // r2c normal
        const size_t is[2] = {1, n}, os[2] = {1, n};
        CK(rocfft_plan_description_create(&d));
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                   nullptr, nullptr, 2, is, n * n, 2, os, n * n));
        CK(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                              rocfft_precision_single, 2, len, 1, d));
        rocfft_plan_description_destroy(d);
        bench("r2c, normal output (row stride n)", p, real, F);
    }
    {   // r2c transposed output: element (x, y) at x * n + y
        const size_t is[2] = {1, n}, os[2] = {n, 1};
        CK(rocfft_plan_description_create(&d));
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved,
                                                   nullptr, nullptr, 2, is, n * n, 2, os, n * n));
        if (rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_real_forward,
                               rocfft_precision_single, 2, len, 1, d) == 0)
            bench("r2c, transposed output (stride n along x)", p, real, F);
        else printf("r2c transposed: plan refused\n");
        rocfft_plan_description_destroy(d);
    }
    {   // c2c inverse in place
        CK(rocfft_plan_create(&p, rocfft_placement_inplace, rocfft_transform_type_complex_inverse,
                              rocfft_precision_single, 2, len, 1, nullptr));
        bench("c2c inverse, in place, normal", p, F, F);
    }
    {   // c2c inverse normal, out of place
        CK(rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                              rocfft_precision_single, 2, len, 1, nullptr));
        bench("c2c inverse, out of place, normal", p, F, G);
    }
    {   // c2c inverse, transposed input -> normal output
        const size_t is[2] = {n, 1}, os[2] = {1, n};
        CK(rocfft_plan_description_create(&d));
        CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved,
                                                   rocfft_array_type_complex_interleaved, nullptr, nullptr,
                                                   2, is, n * n, 2, os, n * n));
        if (rocfft_plan_create(&p, rocfft_placement_notinplace, rocfft_transform_type_complex_inverse,
                               rocfft_precision_single, 2, len, 1, d) == 0)
            bench("c2c inverse, transposed input, out of place", p, F, G);
        else printf("c2c transposed input: plan refused\n");
        rocfft_plan_description_destroy(d);
    }
    rocfft_cleanup();
    return 0;
}
