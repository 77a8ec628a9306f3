// Host prototype behind tools/hub_start.py (exploration; not part of the product or the oracle).
// Per 62 x 62 tile: minimax cost d(c) of the best path from c to the tile's hub (its lowest
// cell) that stays inside the tile interior -- exactly (heap) or as `iters` rounds of the four
// directional scans the GPU visit makes; edge costs between the hubs of neighbouring tiles.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef FT
#define FT 62
#endif
#define BIG 3.0e38f

typedef struct { float k; int i; } item;
static void push(item *h, int *n, float k, int i)
{
    int c = (*n)++;
    while (c > 0) {
        int p = (c - 1) >> 1;
        if (h[p].k <= k) break;
        h[c] = h[p];
        c = p;
    }
    h[c].k = k; h[c].i = i;
}
static item pop(item *h, int *n)
{
    item top = h[0], last = h[--(*n)];
    int c = 0;
    for (;;) {
        int l = 2 * c + 1, r = l + 1, m = c;
        float mk = last.k;
        if (l < *n && h[l].k < mk) { m = l; mk = h[l].k; }
        if (r < *n && h[r].k < mk) { m = r; }
        if (m == c) break;
        h[c] = h[m];
        c = m;
    }
    h[c] = last;
    return top;
}

// tile (ty, tx): interior rows y0..y1, cols x0..x1 (inclusive)
static void tile_box(int H, int W, int ty, int tx, int *y0, int *y1, int *x0, int *x1)
{
    *y0 = 1 + ty * FT; *x0 = 1 + tx * FT;
    *y1 = *y0 + FT - 1; if (*y1 > H - 2) *y1 = H - 2;
    *x1 = *x0 + FT - 1; if (*x1 > W - 2) *x1 = W - 2;
}

// d: H x W (only interior cells written); hub: per tile flat index of the hub cell.
// nhub hubs per tile side (1: one hub per tile; 2: four sub-blocks, each relaxed inside its own
// sub-block -- d then refers to the sub-block's hub).
static int g_margin = 0;
void hub_set_margin(int m) { g_margin = m; }

// exact minimax distances to the tile's hub over the tile grown by g_margin cells (paths may
// leave the tile by that much); written for the tile's own cells only
static void dist_margin(const float *z, int H, int W, int y0, int y1, int x0, int x1, int hy, int hx, float *d)
{
    const int m = g_margin;
    const int Y0 = y0 - m < 1 ? 1 : y0 - m, Y1 = y1 + m > H - 2 ? H - 2 : y1 + m;
    const int X0 = x0 - m < 1 ? 1 : x0 - m, X1 = x1 + m > W - 2 ? W - 2 : x1 + m;
    const int hh = Y1 - Y0 + 1, ww = X1 - X0 + 1;
    float *w = (float *)malloc(sizeof(float) * hh * ww);
    item *heap = (item *)malloc(sizeof(item) * 16 * hh * ww);
    for (int i = 0; i < hh * ww; ++i) w[i] = INFINITY;
    int n = 0;
    w[(hy - Y0) * ww + hx - X0] = z[(size_t)hy * W + hx];
    push(heap, &n, w[(hy - Y0) * ww + hx - X0], (hy - Y0) * ww + hx - X0);
    while (n) {
        item it = pop(heap, &n);
        const int r = it.i / ww, c = it.i % ww;
        if (it.k > w[it.i]) continue;
        for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if (rr < 0 || rr >= hh || cc < 0 || cc >= ww) continue;
            const float zz = z[(size_t)(Y0 + rr) * W + X0 + cc];
            if (!(zz == zz)) continue;
            const float nd = it.k > zz ? it.k : zz;
            if (nd < w[rr * ww + cc]) { w[rr * ww + cc] = nd; push(heap, &n, nd, rr * ww + cc); }
        }
    }
    for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
        const float v = w[(y - Y0) * ww + x - X0];
        d[(size_t)y * W + x] = v == INFINITY ? BIG : v;
    }
    free(w); free(heap);
}

void hub_dist(const float *z, int H, int W, int iters, float *d, int64_t *hub)
{
    const int tiles_y = (H - 2 + FT - 1) / FT, tiles_x = (W - 2 + FT - 1) / FT;
#pragma omp parallel
    {
        item *heap = (item *)malloc(sizeof(item) * 8 * (FT + 2) * (FT + 2));
        float w[FT + 2][FT + 2], zz[FT + 2][FT + 2];
#pragma omp for schedule(dynamic, 4)
        for (int t = 0; t < tiles_y * tiles_x; ++t) {
            const int ty = t / tiles_x, tx = t % tiles_x;
            int y0, y1, x0, x1;
            tile_box(H, W, ty, tx, &y0, &y1, &x0, &x1);
            const int h = y1 - y0 + 1, wd = x1 - x0 + 1;
            int hy = 0, hx = 0;
            float hz = INFINITY;
            for (int r = 0; r < FT + 2; ++r)
                for (int c = 0; c < FT + 2; ++c) { w[r][c] = INFINITY; zz[r][c] = INFINITY; }
            for (int r = 0; r < h; ++r)
                for (int c = 0; c < wd; ++c) {
                    const float v = z[(size_t)(y0 + r) * W + x0 + c];
                    zz[r + 1][c + 1] = v;
                    if (v < hz) { hz = v; hy = r; hx = c; }
                }
            hub[t] = (int64_t)(y0 + hy) * W + x0 + hx;
            w[hy + 1][hx + 1] = hz;
            if (g_margin > 0) {
                dist_margin(z, H, W, y0, y1, x0, x1, y0 + hy, x0 + hx, d);
                continue;
            }
            if (iters == 0) {
                int n = 0;
                push(heap, &n, hz, (hy + 1) * (FT + 2) + hx + 1);
                while (n) {
                    item it = pop(heap, &n);
                    const int r = it.i / (FT + 2), c = it.i % (FT + 2);
                    if (it.k > w[r][c]) continue;
                    for (int dr = -1; dr <= 1; ++dr)
                        for (int dc = -1; dc <= 1; ++dc) {
                            const int rr = r + dr, cc = c + dc;
                            if (zz[rr][cc] == INFINITY) continue;
                            const float nd = it.k > zz[rr][cc] ? it.k : zz[rr][cc];
                            if (nd < w[rr][cc]) { w[rr][cc] = nd; push(heap, &n, nd, rr * (FT + 2) + cc); }
                        }
                }
            } else {
                for (int k = 0; k < iters; ++k) {
#define STEP(r, c, pr0, pc0, pr1, pc1, pr2, pc2) do { \
    float m = w[pr0][pc0]; if (w[pr1][pc1] < m) m = w[pr1][pc1]; if (w[pr2][pc2] < m) m = w[pr2][pc2]; \
    if (m < w[r][c]) w[r][c] = m > zz[r][c] ? m : zz[r][c]; } while (0)
                    for (int r = 1; r <= h; ++r) for (int c = 1; c <= wd; ++c) STEP(r, c, r - 1, c - 1, r - 1, c, r - 1, c + 1);
                    for (int r = h; r >= 1; --r) for (int c = 1; c <= wd; ++c) STEP(r, c, r + 1, c - 1, r + 1, c, r + 1, c + 1);
                    for (int c = 1; c <= wd; ++c) for (int r = 1; r <= h; ++r) STEP(r, c, r - 1, c - 1, r, c - 1, r + 1, c - 1);
                    for (int c = wd; c >= 1; --c) for (int r = 1; r <= h; ++r) STEP(r, c, r - 1, c + 1, r, c + 1, r + 1, c + 1);
                }
            }
            for (int r = 0; r < h; ++r)
                for (int c = 0; c < wd; ++c) {
                    float v = w[r + 1][c + 1];
                    d[(size_t)(y0 + r) * W + x0 + c] = v == INFINITY ? BIG : v;
                }
        }
        free(heap);
    }
}

static inline float fmax2(float a, float b) { return a > b ? a : b; }

// cr: (2 tiles_y + 1) x (2 tiles_x + 1) node-weighted coarse raster: nodes at odd/odd, edges
// between them, +BIG at even/even; the ring holds the edges to the raster's own ring.
void hub_edges(const float *z, const float *d, int H, int W, const int64_t *hub, float *cr)
{
    const int tiles_y = (H - 2 + FT - 1) / FT, tiles_x = (W - 2 + FT - 1) / FT;
    const int ch = 2 * tiles_y + 1, cw = 2 * tiles_x + 1;
    for (int i = 0; i < ch * cw; ++i) cr[i] = BIG;
#pragma omp parallel for schedule(dynamic, 4)
    for (int t = 0; t < tiles_y * tiles_x; ++t) {
        const int ty = t / tiles_x, tx = t % tiles_x;
        int y0, y1, x0, x1;
        tile_box(H, W, ty, tx, &y0, &y1, &x0, &x1);
        cr[(2 * ty + 1) * cw + 2 * tx + 1] = z[hub[t]];
        // east seam (or the raster's last column)
        {
            float e = BIG;
            const int xa = x1, xb = x1 + 1;
            const int ring = xb == W - 1;
            for (int y = y0; y <= y1; ++y)
                for (int dy = -1; dy <= 1; ++dy) {
                    const int yy = y + dy;
                    if (yy < 0 || yy > H - 1) continue;
                    if (!ring && (yy < y0 || yy > y1)) continue;      // (diagonal tile pairs left out)
                    const float other = ring || yy == 0 || yy == H - 1 ? z[(size_t)yy * W + xb] : d[(size_t)yy * W + xb];
                    if (!ring && (yy == 0 || yy == H - 1)) continue;
                    const float c = fmax2(d[(size_t)y * W + xa], other);
                    if (c < e) e = c;
                }
            cr[(2 * ty + 1) * cw + 2 * tx + 2] = e;
        }
        // south seam (or the raster's last row)
        {
            float e = BIG;
            const int ya = y1, yb = y1 + 1;
            const int ring = yb == H - 1;
            for (int x = x0; x <= x1; ++x)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx > W - 1) continue;
                    if (!ring && (xx < x0 || xx > x1)) continue;
                    const float c = fmax2(d[(size_t)ya * W + x], ring ? z[(size_t)yb * W + xx] : d[(size_t)yb * W + xx]);
                    if (c < e) e = c;
                }
            cr[(2 * ty + 2) * cw + 2 * tx + 1] = e;
        }
        if (tx == 0) {   // west: the raster's column 0
            float e = BIG;
            for (int y = y0; y <= y1; ++y)
                for (int dy = -1; dy <= 1; ++dy) {
                    const float c = fmax2(d[(size_t)y * W + x0], z[(size_t)(y + dy) * W]);
                    if (c < e) e = c;
                }
            cr[(2 * ty + 1) * cw] = e;
        }
        if (ty == 0) {   // north: the raster's row 0
            float e = BIG;
            for (int x = x0; x <= x1; ++x)
                for (int dx = -1; dx <= 1; ++dx) {
                    const float c = fmax2(d[(size_t)y0 * W + x], z[x + dx]);
                    if (c < e) e = c;
                }
            cr[2 * tx + 1] = e;
        }
    }
}

// u = max(d, level of the cell's tile) on interior cells, z on the raster ring
void hub_start(const float *z, const float *d, int H, int W, const float *lev, float *u)
{
    const int tiles_x = (W - 2 + FT - 1) / FT;
#pragma omp parallel for
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t i = (size_t)y * W + x;
            if (y == 0 || y == H - 1 || x == 0 || x == W - 1) { u[i] = z[i]; continue; }
            const float l = lev[((y - 1) / FT) * tiles_x + (x - 1) / FT];
            u[i] = fmax2(d[i], l);
        }
}
