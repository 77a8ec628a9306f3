// A3/A4: QuadraticFilter (custom_filters.py:226-257) and the fused
// GrovesCorrection pass (custom_filters.py:708-732, MaskTallGroves :533-534).
//
// Arithmetic.  The reference's closed form
//     out = ((s2 + s3) r1 - s1 (r2 + r3)) / (2 r1^2 - r0 (r2 + r3))
// with s1 = sum w, s2 = sum w xx^2, s3 = sum w yy^2 over the ws x ws window is
// the fixed correlation kernel K = a (xx^2 + yy^2) + b, sum K = 1, with
// v_k = -ws/2 + 1 + k (np.linspace(-ws/2+1, ws/2, ws); asymmetric: 0.5 at the
// centre), a = r1/den, b = -(r2+r3)/den.  It is separable:
//     out = sum_y ( a * R2[y] + (a v_y^2 + b) * R0[y] ),
//     R0[y] = sum_x d[y][x],  R2[y] = sum_x d[y][x] v_x^2,
// and because sum K = 1 it is evaluated on d = w - c0 (c0 = one cell near the
// tile centre), which keeps float32 accumulation ~1e-6 m from exact math (the
// reference itself sits 5e-5 m from exact math: float32 s1).
//
// Two kernels evaluate it: groves_stream_kernel (ws = 3, 9, 15: the production
// window) and the tiled groves_kernel below for the other odd sizes up to 31.
//
// Tiled kernel shape (gfx950): one 256-thread workgroup per 64 x 32 output tile.
//   phase 0  stage the (32+2p) x (64+2p) input window in LDS as d = w - c0;
//   phase 1  row sums R0/R2 for every staged row, 4 adjacent columns per item
//            from aligned ds_read_b128 runs, results to two LDS planes;
//   phase 2  each lane walks 8 output rows of one column (conflict-free
//            ds_read_b32 down the planes), 2 fma per tap;
//   epilogue hl = img - smooth; m = groves && hl > thr;
//            out = m ? smooth : hl + smooth; ring of p cells = img unchanged.
// Algorithmic HBM bytes: 4 (img) + 1 (mask) + 4 (out) = 9 B/cell/iteration
// (8 for the plain quadratic filter).
#include "hdem_internal.h"

#include <cmath>
#include <cstdlib>

namespace {

constexpr int GTW = 64;
constexpr int GTH = 32;
constexpr int NT = 256;
constexpr int WS_MAX = 31;

struct quad_coef {
    float a;
    float v2[WS_MAX];   // v_k^2
    float cy[WS_MAX];   // a v_k^2 + b
};

template <int WS>
__global__ __launch_bounds__(NT) void groves_kernel(const float *__restrict__ img,
                                                   const uint8_t *__restrict__ groves,
                                                   int H, int W, float thr, int tiles_x,
                                                   quad_coef cf, float *__restrict__ out)
{
    constexpr int P = WS / 2;
    constexpr int IR = GTH + 2 * P;                   // staged rows
    constexpr int IC = GTW + 2 * P;                   // staged cols
    constexpr int IS = (IC + 3) / 4 * 4 + 4;          // row stride (16 B multiple)
    __shared__ __attribute__((aligned(16))) float in[IR * IS];
    __shared__ __attribute__((aligned(16))) float r0p[IR * GTW];
    __shared__ __attribute__((aligned(16))) float r2p[IR * GTW];

    const int tid = threadIdx.x;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int x0 = bx * GTW, y0 = by * GTH;

    // tile offset: any finite value near the window works (sum K = 1)
    float c0 = img[(size_t)min(y0 + GTH / 2, H - 1) * W + min(x0 + GTW / 2, W - 1)];
    if (!(fabsf(c0) < HDEM_INF)) c0 = 0.0f;

    // phase 0: stage d = w - c0 (coordinates clamped; cells that would read
    // outside the raster only feed ring outputs, which are overwritten by img)
    for (int i = tid; i < IR * IC; i += NT) {
        int r = i / IC, c = i % IC;
        int gy = min(max(y0 - P + r, 0), H - 1);
        int gx = min(max(x0 - P + c, 0), W - 1);
        in[r * IS + c] = img[(size_t)gy * W + gx] - c0;
    }
    for (int i = tid; i < IR * (IS - IC); i += NT) {  // pad columns: defined values
        int r = i / (IS - IC), c = IC + i % (IS - IC);
        in[r * IS + c] = 0.0f;
    }
    __syncthreads();

    // phase 1: row sums, items of 4 adjacent output columns
    constexpr int NV = (4 + 2 * P + 3) / 4;           // b128 reads per item
    for (int i = tid; i < IR * (GTW / 4); i += NT) {
        int r = i / (GTW / 4), c4 = (i % (GTW / 4)) * 4;
        float d[NV * 4];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            hdem_f4 v = *reinterpret_cast<const hdem_f4 *>(&in[r * IS + c4 + 4 * k]);
            d[4 * k] = v[0]; d[4 * k + 1] = v[1]; d[4 * k + 2] = v[2]; d[4 * k + 3] = v[3];
        }
        hdem_f4 s0, s2;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float a0 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int k = 0; k < WS; ++k) {
                a0 += d[o + k];
                a2 = fmaf(d[o + k], cf.v2[k], a2);
            }
            s0[o] = a0; s2[o] = a2;
        }
        *reinterpret_cast<hdem_f4 *>(&r0p[r * GTW + c4]) = s0;
        *reinterpret_cast<hdem_f4 *>(&r2p[r * GTW + c4]) = s2;
    }
    __syncthreads();

    // phase 2: column sums + epilogue; lane = column, wave = 8 output rows
    const int c = tid & 63, rg = tid >> 6;
    const int x = x0 + c;
    constexpr int RPT = GTH / 4;                      // rows per thread
    float q0[RPT + 2 * P], q2[RPT + 2 * P];
#pragma unroll
    for (int k = 0; k < RPT + 2 * P; ++k) {
        q0[k] = r0p[(rg * RPT + k) * GTW + c];
        q2[k] = r2p[(rg * RPT + k) * GTW + c];
    }
    if (x >= W) return;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int y = y0 + rg * RPT + rr;
        if (y >= H) break;
        float acc = 0.0f, acc2 = 0.0f;
#pragma unroll
        for (int k = 0; k < WS; ++k) {
            acc = fmaf(cf.cy[k], q0[rr + k], acc);
            acc2 += q2[rr + k];
        }
        float smooth = c0 + fmaf(cf.a, acc2, acc);
        size_t gi = (size_t)y * W + x;
        float w = img[gi];                                // L2-served re-read
        bool ring = y < P || y >= H - P || x < P || x >= W - P;
        float o;
        if (ring) {
            o = w;
        } else if (groves) {
            float hl = w - smooth;
            bool m = groves[gi] != 0 && hl > thr;
            o = m ? smooth : hl + smooth;
        } else {
            o = smooth;
        }
        out[gi] = o;
    }
}

// ---------------------------------------------------------------------------
// Streaming form of the same arithmetic (used for ws <= 15).  One WAVE owns a
// 256-column strip of 96..192 rows (launch_ws picks the height) and walks down its input rows once:
//   * lane l holds columns 4l..4l+3 as one float4 (a wave-level load is 1 KiB of
//     one raster row); lanes 0..2p-1 also fetch one halo column each; rows are
//     prefetched three ahead so ~3 KiB per wave are always in flight;
//   * the row (as d = w - c0) goes through a wave-private LDS row buffer (two
//     alternating slots, no workgroup barrier) from which each lane reads the
//     4+2p values around its columns as aligned ds_read_b128;
//   * the row sums R0/R2 feed a ring of `ws` vertical accumulators per column held
//     in registers (slot = output row mod ws, static after unrolling by ws): the
//     output row that receives its last term is finished, blended and stored.
// HBM reads are the raster once plus (rows+2p)/rows row overlap (~1.1x at ws = 15)
// instead of the 1.75x of the tiled kernel; no tile is staged twice.
// ---------------------------------------------------------------------------
constexpr int SW_COLS = 256;     // strip width  (cells) = 64 lanes x 4
constexpr int SR_ROWS = 128;     // strip height (output rows) unless the launch picks another

// (two waves per SIMD.  Rounds 1-2 ran three at 168 registers with a few of them spilled, which
// was 6 % faster than 183 registers and two waves THEN; with the vertical box sum through a
// prefix -- 10 % fewer vector instructions -- the three-wave build spills 47 scratch accesses
// per 15 rows and gains nothing, the two-wave build is 8 % faster: 1.85 against 2.02 ms)
// FULL: every strip of the launch lies inside the raster's columns and W is a multiple of 4 --
// no lane is ever out of range, every load and store is the vector form, and above all NO load
// sits in a branch: with loads under per-lane conditions the compiler cannot count them and
// waits for ALL outstanding memory operations (s_waitcnt vmcnt(0)) five times per row, which
// drains the rows that were fetched ahead -- the general form runs that way, on the raster's
// last strip column only.
template <int WS, bool FULL>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2))) void groves_stream_kernel(const float *__restrict__ img,
                                                          const uint8_t *__restrict__ groves,
                                                          int H, int W, float thr,
                                                          int strips_x, int nstrips, int strip_rows,
                                                          int sx0, quad_coef cf, float *__restrict__ out)
{
    constexpr int P = WS / 2;
    constexpr int PADL = (P + 3) / 4 * 4;                 // interior starts 16-byte aligned
    constexpr int OFF = PADL - P;                         // first needed float, from the aligned read
    constexpr int NRD = (OFF + 4 + 2 * P + 3) / 4;        // ds_read_b128 per lane per row
    constexpr int RB = (PADL + SW_COLS + P + 3) / 4 * 4 + 4;   // row buffer (floats)
    constexpr int PF = WS % 5 == 0 ? 5 : 3;               // rows in flight (divides the unroll)
    static_assert(WS % PF == 0 || WS < PF, "prefetch ring must divide the unroll");
    constexpr int NSLOT = 8;                              // >= p + 1 rows of history (ws <= 15)
    static_assert(P + 1 <= NSLOT, "row ring too short");
    __shared__ __attribute__((aligned(16))) float rows[4][NSLOT][RB];
    // the lane's own four raw cells of the same rows: the epilogue needs the untouched centre
    // value (the row buffer holds d = w - c0, subtracted once by the lane that wrote the cell
    // instead of by every lane that reads it: 5 instead of 20 subtractions per lane-row)
    __shared__ __attribute__((aligned(16))) float raw[4][NSLOT][SW_COLS];

    // (the wave index as a scalar: strip origin, row addresses and range tests are then scalar
    // code instead of per-lane vector arithmetic)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = blockIdx.x * 4 + wave;
    if (strip >= nstrips) return;
    const int sy = strip / strips_x, sx = sx0 + strip - sy * strips_x;
    const int x0 = sx * SW_COLS, y0 = sy * strip_rows;
    const int x = x0 + 4 * lane;
    float *rb = &rows[wave][0][0];
    float *rw = &raw[wave][0][0];
    const bool vec_ok = FULL || x + 4 <= W;

    float c0 = img[(size_t)min(y0 + strip_rows / 2, H - 1) * W + min(x0 + SW_COLS / 2, W - 1)];
    if (!(fabsf(c0) < HDEM_INF)) c0 = 0.0f;

    // halo column of this lane (lanes 0..2p-1): left halo x0-p+lane, right x0+256+(lane-p)
    const bool has_halo = lane < 2 * P;
    const int hx = min(max(lane < P ? x0 - P + lane : x0 + SW_COLS + lane - P, 0), W - 1);
    const int hidx = lane < P ? OFF + lane : PADL + SW_COLS + lane - P;

    const int hxa = has_halo ? hx : x;                   // FULL: every lane loads a "halo" cell
    auto load_row = [&](int i, hdem_f4 &v, float &hv) {
        const int y = min(max(y0 - P + i, 0), H - 1);
        const float *row = img + (size_t)y * W;
        if (FULL) {
            v = hdem_ld4u(row + x);
            hv = row[hxa];
            return;
        }
        if (x + 4 <= W) {
            v = hdem_ld4u(row + x);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = row[min(x + k, W - 1)];
        }
        hv = has_halo ? row[hx] : 0.0f;
    };

    // accumulators as float2 pairs: the vertical update is v_pk_add + v_pk_fma
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 acc[WS][2];
#pragma unroll
    for (int j = 0; j < WS; ++j) { acc[j][0] = (f2){0.0f, 0.0f}; acc[j][1] = (f2){0.0f, 0.0f}; }

    // groves bytes of the output row that completes at step i (row y0 + i - 2p), fetched
    // PF steps ahead like the image rows: nothing in the loop waits on a fresh load
    auto load_mask = [&](int i) -> unsigned {
        const int y = y0 + i - 2 * P;
        if (FULL)       // (a row outside the raster: any valid word, the epilogue skips the row)
            return groves ? *reinterpret_cast<const unsigned *>(
                                groves + (size_t)min(max(y, 0), H - 1) * W + x)
                          : 0u;
        unsigned g = 0;
        if (groves && y >= 0 && y < H && x < W) {
            const size_t gi = (size_t)y * W + x;
            if (vec_ok && (gi & 3) == 0) g = *reinterpret_cast<const unsigned *>(groves + gi);
            else
                for (int k = 0; k < 4; ++k)
                    if (x + k < W) g |= (unsigned)groves[gi + k] << (8 * k);
        }
        return g;
    };

    // R2 enters all ws pending output rows with weight 1 -- a vertical box sum.  Instead of ws
    // packed adds per row: one running prefix PT of the rows' R2 inside the current block of ws
    // rows (the unrolled loop body).  A slot is opened at -PT, takes +PT(end of block) when the
    // block ends (ws adds once per ws rows) and +PT once more when it completes: 4 packed
    // operations per row and column pair instead of ws, and every sum still has <= ws terms.
    f2 pt[2] = {(f2){0.0f, 0.0f}, (f2){0.0f, 0.0f}};

    hdem_f4 pre[PF];
    float preh[PF];
    unsigned preg[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) { load_row(k, pre[k], preh[k]); preg[k] = load_mask(k); }

    // One row ahead through LDS: iteration i puts row i + 1 into its slot and asks for the
    // 4 + 2p values around the lane's columns, and works on row i, whose values arrived during
    // iteration i - 1 -- the LDS round trip is off the row's critical path.
    auto stage_row = [&](int i, const hdem_f4 &v, float hv) {
        float *slot = rb + (i & (NSLOT - 1)) * RB;          // d = w - c0 of image row i
        *reinterpret_cast<hdem_f4 *>(rw + (i & (NSLOT - 1)) * SW_COLS + 4 * lane) = v;
        const hdem_f4 dv = {v[0] - c0, v[1] - c0, v[2] - c0, v[3] - c0};
        *reinterpret_cast<hdem_f4 *>(slot + PADL + 4 * lane) = dv;
        if (has_halo) slot[hidx] = hv - c0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto read_row = [&](int i, float (&d)[NRD * 4]) {
        const float *slot = rb + (i & (NSLOT - 1)) * RB;
#pragma unroll
        for (int k = 0; k < NRD; ++k) {
            const hdem_f4 q = *reinterpret_cast<const hdem_f4 *>(slot + 4 * lane + 4 * k);
            d[4 * k] = q[0]; d[4 * k + 1] = q[1];
            d[4 * k + 2] = q[2]; d[4 * k + 3] = q[3];
        }
    };
    float d[NRD * 4];
    stage_row(0, pre[0], preh[0]);
    load_row(PF, pre[0], preh[0]);
    read_row(0, d);

    const int NROWS = strip_rows + 2 * P;
    for (int base = 0; base < NROWS; base += WS) {
#pragma unroll
        for (int u = 0; u < WS; ++u) {
            const int i = base + u;
            const unsigned g4 = preg[u % PF];
            preg[u % PF] = load_mask(i + PF);
            // the raw centre row of the output that completes now (input row i - p): read
            // before row i + 1 takes that slot of the ring
            const hdem_f4 wq = *reinterpret_cast<const hdem_f4 *>(
                rw + ((i - P) & (NSLOT - 1)) * SW_COLS + 4 * lane);
            // ---- row i + 1: registers -> LDS, its values on their way, next prefetch ----
            float dn[NRD * 4];
            stage_row(i + 1, pre[(u + 1) % PF], preh[(u + 1) % PF]);
            load_row(i + 1 + PF, pre[(u + 1) % PF], preh[(u + 1) % PF]);
            read_row(i + 1, dn);
            // ---- row sums, then into the ring of vertical accumulators --------------
            // Moments instead of four 15-tap sums: with v_k = k + v0 the weighted row sum is
            // sum v_k^2 d = M2 + 2 v0 M1 + v0^2 M0 (M_i = sum k^i d), and the three moments
            // slide from one output column to the next in 7 operations
            //   M0' = M0 - a + e,  M1' = M1 + n e - M0',  M2' = M2 - 2 M1 + (n^2 - 2n) e + M0'
            // (a leaves the window, e enters): 78 operations per lane-row instead of 120.
            // The offsets d = w - c0 are a few metres, M2 ~ 1e5: 1e-5 m after the a = 3e-4.
            float r0[4], r2[4];
            {
                constexpr float V0 = 1.0f - WS / 2.0f, NN = (float)WS, N2 = (float)(WS * WS - 2 * WS);
                const float a0 = cf.a * (V0 * V0), a1 = cf.a * (2.0f * V0);
                float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
#pragma unroll
                for (int k = 0; k < WS; ++k) {
                    const float x = d[OFF + k];
                    m0 += x;
                    m1 = fmaf((float)k, x, m1);
                    m2 = fmaf((float)(k * k), x, m2);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    r0[o] = m0;
                    r2[o] = fmaf(a0, m0, fmaf(a1, m1, cf.a * m2));
                    if (o < 3) {
                        const float lv = d[OFF + o], en = d[OFF + o + WS];
                        const float m0n = m0 - lv + en;
                        const float m1n = fmaf(NN, en, m1) - m0n;
                        m2 = fmaf(N2, en, fmaf(-2.0f, m1, m2)) + m0n;
                        m1 = m1n;
                        m0 = m0n;
                    }
                }
            }
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
                const f2 r = {r0[2 * p2], r0[2 * p2 + 1]}, t = {r2[2 * p2], r2[2 * p2 + 1]};
                // (all the adds, then all the fmas: written as add-then-fma per slot the
                // compiler reuses one temporary, and a packed fma that consumes the packed
                // add just before it needs a wait state -- 30 s_nop per row)
                pt[p2] += t;
#pragma unroll
                for (int j = 0; j < WS; ++j) {
                    // output row (i - j) sits in slot (u - j) mod WS and takes weight cy[j]
                    const int sl = ((u - j) % WS + WS) % WS;
                    const f2 cy = {cf.cy[j], cf.cy[j]};
                    acc[sl][p2] = __builtin_elementwise_fma(cy, r, acc[sl][p2]);
                }
            }
            // ---- output row i - 2p is complete: blend and store ------------------------
            const int done = (u + 1) % WS;            // slot of output row i - (WS - 1)
            if (u == WS - 1) {
                // the block ends: its R2 total goes to every open slot (the completing one
                // opened with the block), and the prefix starts again
#pragma unroll
                for (int j = 0; j < WS; ++j) { acc[j][0] += pt[0]; acc[j][1] += pt[1]; }
                pt[0] = (f2){0.0f, 0.0f};
                pt[1] = (f2){0.0f, 0.0f};
            } else {
                acc[done][0] += pt[0];
                acc[done][1] += pt[1];
            }
            const int oy = i - 2 * P, y = y0 + oy;
            if (oy >= 0 && oy < strip_rows && y < H && (FULL || x < W)) {
                const size_t gi = (size_t)y * W + x;
                float o4[4];
                const float wv[4] = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float smooth = c0 + acc[done][k >> 1][k & 1];
                    const int xx = x + k;
                    const bool ring = y < P || y >= H - P || xx < P || xx >= W - P;
                    float o;
                    if (ring) {
                        o = wv[k];
                    } else if (groves) {
                        const float hl = wv[k] - smooth;
                        const bool m = ((g4 >> (8 * k)) & 0xffu) != 0 && hl > thr;
                        o = m ? smooth : hl + smooth;
                    } else {
                        o = smooth;
                    }
                    o4[k] = o;
                }
                if (vec_ok) {
                    hdem_f4 q = {o4[0], o4[1], o4[2], o4[3]};
                    hdem_st4u(out + gi, q);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (x + k < W) out[gi + k] = o4[k];
                }
            }
            // the slot opens again for the output row that starts with the next input row
            acc[done][0] = -pt[0];
            acc[done][1] = -pt[1];
#pragma unroll
            for (int k = 0; k < NRD * 4; ++k) d[k] = dn[k];
        }
    }
}

int make_coef(int ws, quad_coef *cf)
{
    double v[WS_MAX], r0 = (double)ws * ws, r1 = 0, r2 = 0, r3 = 0;
    for (int k = 0; k < ws; ++k) v[k] = -ws / 2.0 + 1.0 + k;
    for (int j = 0; j < ws; ++j)
        for (int i = 0; i < ws; ++i) {
            double xx = v[i] * v[i], yy = v[j] * v[j];
            r1 += xx; r2 += xx * xx; r3 += xx * yy;
        }
    double den = 2.0 * r1 * r1 - r0 * (r2 + r3);
    double a = r1 / den, b = -(r2 + r3) / den;
    cf->a = (float)a;
    for (int k = 0; k < WS_MAX; ++k) {
        cf->v2[k] = k < ws ? (float)(v[k] * v[k]) : 0.0f;
        cf->cy[k] = k < ws ? (float)(a * v[k] * v[k] + b) : 0.0f;
    }
    return 0;
}

template <int WS>
void launch_ws(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H, int W,
               float thr, const quad_coef &cf, float *out)
{
    if constexpr (WS == 15 || WS == 9 || WS == 3) {      // streaming form (prefetch ring | ws)
        // Strip height: every strip is one wave's serial walk, and the machine holds 8 of them
        // per CU (two waves per SIMD) -- a count of strips just above a multiple of that leaves
        // the last round nearly empty (16384^2 at 128 rows: 8192 strips on 3072 places, 2.67
        // rounds).  Among the heights from 96 to 192 rows take the one whose last round is
        // fullest, ties to the taller (less overlap between strips).
        const int sx = (W + SW_COLS - 1) / SW_COLS, places = ctx->num_cus * (getenv("HDEM_GROVES_PLACES") ? atoi(getenv("HDEM_GROVES_PLACES")) : 8);
        int rows = SR_ROWS;
        double best = -1.0;
        for (int cand = 96; cand <= 192; ++cand) {
            const int64_t strips = (int64_t)sx * ((H + cand - 1) / cand);
            const int64_t rounds = (strips + places - 1) / places;
            // useful share of the machine-time: work / (rounds x places x strip length)
            const double eff = (double)H * sx / ((double)rounds * places * (cand + 2 * (WS / 2)));
            if (eff >= best) { best = eff; rows = cand; }
        }
        const int sy = (H + rows - 1) / rows;
        // strip columns that lie inside the raster take the branch-free form (see the kernel)
        const int fx = (W % 4 == 0 && ((uintptr_t)groves % 4 == 0)) ? W / SW_COLS : 0;
        if (fx > 0)
            hipLaunchKernelGGL((groves_stream_kernel<WS, true>), dim3((fx * sy + 3) / 4), dim3(NT), 0,
                               ctx->stream, img, groves, H, W, thr, fx, fx * sy, rows, 0, cf, out);
        if (sx > fx)
            hipLaunchKernelGGL((groves_stream_kernel<WS, false>), dim3(((sx - fx) * sy + 3) / 4),
                               dim3(NT), 0, ctx->stream, img, groves, H, W, thr, sx - fx,
                               (sx - fx) * sy, rows, fx, cf, out);
        return;
    }
    int tx = (W + GTW - 1) / GTW, ty = (H + GTH - 1) / GTH;
    hipLaunchKernelGGL(groves_kernel<WS>, dim3(tx * ty), dim3(NT), 0, ctx->stream, img,
                       groves, H, W, thr, tx, cf, out);
}

int launch_pass(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H, int W,
                int ws, float thr, const quad_coef &cf, float *out)
{
    hdem_scoped_timer tm(ctx, HDEM_K_GROVES, (int64_t)H * W);
    switch (ws) {
#define HDEM_WS_CASE(N) case N: launch_ws<N>(ctx, img, groves, H, W, thr, cf, out); break;
        HDEM_WS_CASE(3) HDEM_WS_CASE(5) HDEM_WS_CASE(7)
        HDEM_WS_CASE(9) HDEM_WS_CASE(11) HDEM_WS_CASE(13) HDEM_WS_CASE(15)
        HDEM_WS_CASE(17) HDEM_WS_CASE(19) HDEM_WS_CASE(21) HDEM_WS_CASE(23)
        HDEM_WS_CASE(25) HDEM_WS_CASE(27) HDEM_WS_CASE(29) HDEM_WS_CASE(31)
#undef HDEM_WS_CASE
        default:
            hdem_set_error("window size %d has no kernel (odd sizes 3..%d)", ws, WS_MAX);
            return HDEM_ERR_BAD_ARG;
    }
    return HDEM_OK;
}

// window validation, same two failure classes and the same order of checks as
// the SlidingWindow constructor (sliding_window.py:150-156)
int check_window(int ws, int H, int W)
{
    HDEM_REQUIRE(ws > 0, HDEM_ERR_BAD_ARG, "window size must be positive, got %d", ws);
    HDEM_REQUIRE(ws != 1, HDEM_ERR_BAD_ARG,
                 "window size 1 is degenerate (the reference's closed form is 0/0)");
    HDEM_REQUIRE(ws <= H && ws <= W, HDEM_ERR_WINDOW_HIGH,
                 "Window size: %d cannot be higher than grid dimensions: (%d, %d)", ws, H, W);
    HDEM_REQUIRE(ws % 2 == 1, HDEM_ERR_WINDOW_EVEN,
                 "Window size: %d cannot be an even number", ws);
    HDEM_REQUIRE(ws <= WS_MAX, HDEM_ERR_BAD_ARG,
                 "window size %d not supported by the HIP kernel (max %d)", ws, WS_MAX);
    return HDEM_OK;
}

}  // namespace

extern "C" int hdem_groves_f32_dev(hdem_ctx *ctx, const float *img, const uint8_t *groves,
                                   int H, int W, int ws, float thr, int iters,
                                   float *scratch, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(groves, HDEM_ERR_BAD_ARG, "groves mask is null");
    HDEM_REQUIRE(iters >= 1, HDEM_ERR_BAD_ARG, "iterations must be >= 1, got %d", iters);
    HDEM_REQUIRE(img != out, HDEM_ERR_BAD_ARG, "groves cannot run in place");
    HDEM_REQUIRE(iters == 1 || (scratch && scratch != out && scratch != img),
                 HDEM_ERR_BAD_ARG, "iterations > 1 need a distinct scratch buffer");
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    quad_coef cf;
    make_coef(ws, &cf);
    // ping-pong so that the last pass lands in `out`
    const float *src = img;
    for (int it = 0; it < iters; ++it) {
        float *dst = ((iters - 1 - it) % 2 == 0) ? out : scratch;
        if (int rc = launch_pass(ctx, src, groves, H, W, ws, thr, cf, dst)) return rc;
        src = dst;
    }
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_quadratic_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W, int ws,
                                      float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    HDEM_REQUIRE(dem != out, HDEM_ERR_BAD_ARG, "quadratic filter cannot run in place");
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    quad_coef cf;
    make_coef(ws, &cf);
    if (int rc = launch_pass(ctx, dem, nullptr, H, W, ws, 0.0f, cf, out)) return rc;
    HDEM_HIP_CHECK(hipGetLastError());
    return HDEM_OK;
}

extern "C" int hdem_quadratic_f32(hdem_ctx *ctx, const float *dem, int H, int W, int ws,
                                  float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(dem, out, H, W)) return rc;
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t bytes = (size_t)H * W * sizeof(float);
    hdem_dbuf din, dout;
    if (int rc = din.alloc(ctx, bytes)) return rc;
    if (int rc = dout.alloc(ctx, bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, din.p, dem, bytes)) return rc;
    if (int rc = hdem_quadratic_f32_dev(ctx, (const float *)din.p, H, W, ws, (float *)dout.p))
        return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, bytes);
}

extern "C" int hdem_groves_f32(hdem_ctx *ctx, const float *img, const uint8_t *groves, int H,
                               int W, int ws, float thr, int iters, float *out)
{
    HDEM_REQUIRE(ctx, HDEM_ERR_BAD_ARG, "ctx is null");
    if (int rc = hdem_check_raster(img, out, H, W)) return rc;
    HDEM_REQUIRE(groves, HDEM_ERR_BAD_ARG, "groves mask is null");
    HDEM_REQUIRE(iters >= 1, HDEM_ERR_BAD_ARG, "iterations must be >= 1, got %d", iters);
    if (int rc = check_window(ws, H, W)) return rc;
    HDEM_HIP_CHECK(hipSetDevice(ctx->device));
    size_t n = (size_t)H * W, bytes = n * sizeof(float);
    hdem_dbuf din, dout, dscr, dg;
    if (int rc = din.alloc(ctx, bytes)) return rc;
    if (int rc = dout.alloc(ctx, bytes)) return rc;
    if (iters > 1) if (int rc = dscr.alloc(ctx, bytes)) return rc;
    if (int rc = dg.alloc(ctx, n)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, din.p, img, bytes)) return rc;
    if (int rc = hdem_memcpy_h2d(ctx, dg.p, groves, n)) return rc;
    if (int rc = hdem_groves_f32_dev(ctx, (const float *)din.p, (const uint8_t *)dg.p, H, W,
                                     ws, thr, iters, (float *)dscr.p, (float *)dout.p))
        return rc;
    return hdem_memcpy_d2h(ctx, out, dout.p, bytes);
}
