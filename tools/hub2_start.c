// Host prototype (exploration; not product, not oracle): TWO hubs per 62 x 62 tile.
// Hub A = the tile's lowest cell, dA = minimax cost to it inside the tile; hub B = the lowest cell
// that A cannot reach below the level at which A first reaches the tile's rim (the bottom of the
// "other basin"), dB likewise.  Graph: nodes A_T, B_T; A_T - B_T costs dA(B); X_T - Y_T' across a
// seam costs min over adjacent cells a | b of max(dX(a), dY(b)); outlets: the raster ring.
// Levels by a minimax Dijkstra from the ring; start value u(c) = min over the two hubs of
// max(d_hub(c), level(hub)).
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#define FT 62
#define BIG 3.0e38f
typedef struct { float k; int i; } item;
static void push(item *h, int *n, float k, int i)
{
    int c = (*n)++;
    while (c > 0) { int p = (c - 1) >> 1; if (h[p].k <= k) break; h[c] = h[p]; c = p; }
    h[c].k = k; h[c].i = i;
}
static item pop(item *h, int *n)
{
    item top = h[0], last = h[--(*n)];
    int c = 0;
    for (;;) {
        int l = 2 * c + 1, r = l + 1, m = c; float mk = last.k;
        if (l < *n && h[l].k < mk) { m = l; mk = h[l].k; }
        if (r < *n && h[r].k < mk) { m = r; }
        if (m == c) break;
        h[c] = h[m]; c = m;
    }
    h[c] = last;
    return top;
}
static void tile_box(int H, int W, int ty, int tx, int *y0, int *y1, int *x0, int *x1)
{
    *y0 = 1 + ty * FT; *x0 = 1 + tx * FT;
    *y1 = *y0 + FT - 1; if (*y1 > H - 2) *y1 = H - 2;
    *x1 = *x0 + FT - 1; if (*x1 > W - 2) *x1 = W - 2;
}
// exact single-source minimax inside the tile box (heap)
static void relax(const float *z, int W, int y0, int y1, int x0, int x1, int sy, int sx, float *d, item *heap)
{
    for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) d[(size_t)y * W + x] = BIG;
    int n = 0;
    d[(size_t)sy * W + sx] = z[(size_t)sy * W + sx];
    push(heap, &n, d[(size_t)sy * W + sx], sy * W + sx);
    while (n) {
        item it = pop(heap, &n);
        const int y = it.i / W, x = it.i % W;
        if (it.k > d[it.i]) continue;
        for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy < y0 || yy > y1 || xx < x0 || xx > x1) continue;
            const size_t j = (size_t)yy * W + xx;
            const float nd = it.k > z[j] ? it.k : z[j];
            if (nd < d[j]) { d[j] = nd; push(heap, &n, nd, (int)j); }
        }
    }
}
// dA, dB: H x W; hubA, hubB: flat indices per tile (hubB = -1: none); ab: dA at hub B
void hub2_dist(const float *z, int H, int W, float *dA, float *dB, int64_t *hubA, int64_t *hubB, float *ab)
{
    const int tiles_y = (H - 2 + FT - 1) / FT, tiles_x = (W - 2 + FT - 1) / FT;
#pragma omp parallel
    {
        item *heap = (item *)malloc(sizeof(item) * 16 * (FT + 2) * (FT + 2));
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < tiles_y * tiles_x; ++t) {
            int y0, y1, x0, x1;
            tile_box(H, W, t / tiles_x, t % tiles_x, &y0, &y1, &x0, &x1);
            int hy = y0, hx = x0; float hz = INFINITY;
            for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x)
                if (z[(size_t)y * W + x] < hz) { hz = z[(size_t)y * W + x]; hy = y; hx = x; }
            hubA[t] = (int64_t)hy * W + hx;
            relax(z, W, y0, y1, x0, x1, hy, hx, dA, heap);
            // level at which A first reaches the rim
            float rim = INFINITY;
            for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x)
                if ((y == y0 || y == y1 || x == x0 || x == x1) && dA[(size_t)y * W + x] < rim) rim = dA[(size_t)y * W + x];
            int by = -1, bx = -1; float bz = INFINITY;
            for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
                const size_t i = (size_t)y * W + x;
                if (dA[i] > rim && dA[i] > z[i] && z[i] < bz) { bz = z[i]; by = y; bx = x; }
            }
            if (by < 0) { hubB[t] = -1; ab[t] = BIG; for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) dB[(size_t)y * W + x] = BIG; continue; }
            hubB[t] = (int64_t)by * W + bx;
            ab[t] = dA[hubB[t]];
            relax(z, W, y0, y1, x0, x1, by, bx, dB, heap);
        }
        free(heap);
    }
}
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float fmin2(float a, float b) { return a < b ? a : b; }
// levels of the 2 * ntiles hubs (A at 2t, B at 2t + 1) by minimax Dijkstra from the raster ring
void hub2_levels(const float *z, const float *dA, const float *dB, int H, int W, const int64_t *hubA,
                 const int64_t *hubB, const float *ab, float *lev)
{
    const int tiles_y = (H - 2 + FT - 1) / FT, tiles_x = (W - 2 + FT - 1) / FT, nt = tiles_y * tiles_x;
    // edge costs: per tile and side (E, S) a 2 x 2 matrix; ring costs per hub
    float *ce = (float *)malloc(sizeof(float) * nt * 4), *cs = (float *)malloc(sizeof(float) * nt * 4);
    float *ring = (float *)malloc(sizeof(float) * nt * 2);
    const float *D[2] = {dA, dB};
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < nt; ++t) {
        const int ty = t / tiles_x, tx = t % tiles_x;
        int y0, y1, x0, x1;
        tile_box(H, W, ty, tx, &y0, &y1, &x0, &x1);
        for (int k = 0; k < 4; ++k) { ce[t * 4 + k] = BIG; cs[t * 4 + k] = BIG; }
        ring[2 * t] = ring[2 * t + 1] = BIG;
        for (int p = 0; p < 2; ++p) {
            // ring crossings of this tile's hub p
            for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
                if (!(y == 1 || y == H - 2 || x == 1 || x == W - 2)) continue;
                const float dc = D[p][(size_t)y * W + x];
                for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy == 0 || yy == H - 1 || xx == 0 || xx == W - 1)
                        ring[2 * t + p] = fmin2(ring[2 * t + p], fmax2(dc, z[(size_t)yy * W + xx]));
                }
            }
            for (int q = 0; q < 2; ++q) {
                if (tx + 1 < tiles_x) {
                    float e = BIG;
                    for (int y = y0; y <= y1; ++y) for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = y + dy;
                        if (yy < y0 || yy > y1) continue;
                        e = fmin2(e, fmax2(D[p][(size_t)y * W + x1], D[q][(size_t)yy * W + x1 + 1]));
                    }
                    ce[t * 4 + p * 2 + q] = e;
                }
                if (ty + 1 < tiles_y) {
                    float e = BIG;
                    for (int x = x0; x <= x1; ++x) for (int dx = -1; dx <= 1; ++dx) {
                        const int xx = x + dx;
                        if (xx < x0 || xx > x1) continue;
                        e = fmin2(e, fmax2(D[p][(size_t)y1 * W + x], D[q][(size_t)(y1 + 1) * W + xx]));
                    }
                    cs[t * 4 + p * 2 + q] = e;
                }
            }
        }
    }
    item *heap = (item *)malloc(sizeof(item) * 64 * (size_t)nt);
    int n = 0;
    for (int i = 0; i < 2 * nt; ++i) {
        lev[i] = BIG;
        const int t = i / 2, p = i & 1;
        if (p == 1 && hubB[t] < 0) continue;
        const float hz = z[p ? hubB[t] : hubA[t]];
        if (ring[i] < BIG) { lev[i] = fmax2(ring[i], hz); push(heap, &n, lev[i], i); }
    }
    while (n) {
        item it = pop(heap, &n);
        if (it.k > lev[it.i]) continue;
        const int t = it.i / 2, p = it.i & 1, ty = t / tiles_x, tx = t % tiles_x;
#define TRY(j, cost) do { const int jj = (j); if ((jj & 1) == 0 || hubB[jj / 2] >= 0) { \
    const float hz_ = z[(jj & 1) ? hubB[jj / 2] : hubA[jj / 2]]; \
    const float nd_ = fmax2(fmax2(it.k, (cost)), hz_); \
    if (nd_ < lev[jj]) { lev[jj] = nd_; push(heap, &n, nd_, jj); } } } while (0)
        TRY(2 * t + (1 - p), ab[t]);
        for (int q = 0; q < 2; ++q) {
            if (tx + 1 < tiles_x) TRY(2 * (t + 1) + q, ce[t * 4 + p * 2 + q]);
            if (tx > 0) TRY(2 * (t - 1) + q, ce[(t - 1) * 4 + q * 2 + p]);
            if (ty + 1 < tiles_y) TRY(2 * (t + tiles_x) + q, cs[t * 4 + p * 2 + q]);
            if (ty > 0) TRY(2 * (t - tiles_x) + q, cs[(t - tiles_x) * 4 + q * 2 + p]);
        }
    }
    free(heap); free(ce); free(cs); free(ring);
}
void hub2_start(const float *z, const float *dA, const float *dB, int H, int W, const float *lev, float *u)
{
    const int tiles_x = (W - 2 + FT - 1) / FT;
#pragma omp parallel for
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            if (y == 0 || y == H - 1 || x == 0 || x == W - 1) { u[i] = z[i]; continue; }
            const int t = ((y - 1) / FT) * tiles_x + (x - 1) / FT;
            u[i] = fmin2(fmax2(dA[i], lev[2 * t]), fmax2(dB[i], lev[2 * t + 1]));
        }
}
