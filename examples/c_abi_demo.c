/*
 * The drop-in boundary from plain C: no Python, no torch, nothing but
 * include/hydrodem_hip.h and libhydrodem_hip.so.
 *
 *     gcc -std=c11 -O2 -I include examples/c_abi_demo.c \
 *         -L hydrodem_amd/csrc -lhydrodem_hip -Wl,-rpath,$PWD/hydrodem_amd/csrc -lm -o c_abi_demo
 *     ./c_abi_demo [rows cols [dump-the-input-raster-here.f32]]
 *
 * Builds a small deterministic DEM with a pit, a lake behind a dam and a nodata cell, runs
 * SinkFill (host-pointer entry point), D8 and the 3 x 3 mean + rounding, checks a few facts
 * any correct implementation satisfies (W >= Z, border untouched, the pit is gone, codes are
 * ESRI codes) and prints FNV-1a checksums of the three outputs; tests/test_c_abi_demo.py
 * compares those with the checksums of the same calls made through the Python binding.
 * Error behaviour: every call returns an hdem_status; hdem_last_error() has the text.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hydrodem_hip.h"

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != HDEM_OK) {                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hdem_last_error());       \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const int H = argc > 2 ? atoi(argv[1]) : 300, W = argc > 2 ? atoi(argv[2]) : 421;
    const size_t n = (size_t)H * W;
    float *z = malloc(n * sizeof *z), *w = malloc(n * sizeof *w), *m = malloc(n * sizeof *m);
    uint8_t *d8 = malloc(n);
    if (!z || !w || !m || !d8) return 3;
    /* a tilted plane with ripples (a 32-bit LCG, no libc rand) ... */
    uint32_t s = 12345u;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            s = s * 1664525u + 1013904223u;
            z[(size_t)y * W + x] = 100.0f + 0.05f * (float)x + 0.02f * (float)y +
                                   2.0f * sinf(0.07f * (float)x) * cosf(0.05f * (float)y) +
                                   0.3f * (float)(s >> 8) / 16777216.0f;
        }
    /* ... a pit, a dam across a valley, one nodata cell */
    z[(size_t)(H / 2) * W + W / 2] -= 25.0f;
    for (int y = H / 4; y < H / 4 + 40 && y < H; ++y) z[(size_t)y * W + W / 3] += 15.0f;
    z[(size_t)(H / 3) * W + 2 * W / 3] = NAN;

    if (argc > 3) {                                      /* for a checker in another language */
        FILE *f = fopen(argv[3], "wb");
        if (!f || fwrite(z, sizeof *z, n, f) != n) return 6;
        fclose(f);
    }
    int ndev = 0;
    CHECK(hdem_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no GPU: %s\n", hdem_last_error()); return 4; }
    hdem_ctx *ctx = NULL;
    CHECK(hdem_init(0, &ctx));
    hdem_fill_stats st;
    memset(&st, 0, sizeof st);
    CHECK(hdem_sinkfill_f32(ctx, z, H, W, 0.0f, 0, w, &st));
    CHECK(hdem_d8_f32(ctx, w, H, W, d8));
    CHECK(hdem_boxmean3_f32(ctx, w, H, W, 1, m));
    /* the error path: an even window is the reference's WindowSizeEvenError */
    if (hdem_quadratic_f32(ctx, z, H, W, 4, m) != HDEM_ERR_WINDOW_EVEN) {
        fprintf(stderr, "even window was not refused\n");
        return 5;
    }
    CHECK(hdem_boxmean3_f32(ctx, w, H, W, 1, m));

    int bad = 0;
    for (size_t i = 0; i < n; ++i) {
        if (isnan(z[i])) { bad += !isnan(w[i]); continue; }
        bad += !(w[i] >= z[i]);
        const uint8_t c = d8[i];
        bad += !(c == 0 || c == 1 || c == 2 || c == 4 || c == 8 || c == 16 || c == 32 || c == 64 ||
                 c == 128);
    }
    for (int x = 0; x < W; ++x) bad += w[x] != z[x] && !isnan(z[x]);
    const size_t pit = (size_t)(H / 2) * W + W / 2;
    bad += !(w[pit] > z[pit] + 20.0f);                    /* the pit is filled */
    bad += !st.converged;
    printf("%dx%d converged=%d tile_visits=%lld fill=%016llx d8=%016llx mean=%016llx bad=%d\n", H,
           W, st.converged, (long long)st.tile_visits, (unsigned long long)fnv1a(w, n * sizeof *w),
           (unsigned long long)fnv1a(d8, n), (unsigned long long)fnv1a(m, n * sizeof *m), bad);
    CHECK(hdem_shutdown(ctx));
    free(z); free(w); free(m); free(d8);
    return bad ? 1 : 0;
}
