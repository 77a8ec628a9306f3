/*
 * hydrodem_hip.h -- C ABI of the MI355X (gfx950) raster hot path.
 *
 * This is the drop-in boundary: plain C, caller-owned pointers and sizes, no
 * C++ or torch types, no exceptions.  Every operator entry point replaces the
 * body of one `Filter.apply(ndarray) -> ndarray` of the reference package
 * (`cguerrero/hydrodem/filters/__init__.py:23-39`); the reference-side ctypes
 * stub that binds it is shown in INTEGRATION.md.
 *
 * Conventions
 *   - rasters are C-contiguous row-major, H rows x W columns;
 *   - every function returns an hdem_status; on failure hdem_last_error()
 *     (thread local) holds a message.  Window validation uses the same two
 *     failure classes as `sliding_window.py:150-156`
 *     (HDEM_ERR_WINDOW_HIGH / HDEM_ERR_WINDOW_EVEN);
 *   - `*_f32(...)` entry points take HOST pointers, run synchronously
 *     (upload, kernels, download) and never keep or free caller memory;
 *   - `*_dev(...)` entry points take DEVICE pointers, enqueue on the context's
 *     stream and return without synchronising (except sink fill, which has to
 *     read its convergence counter);
 *   - there is no CPU fallback anywhere behind this header.
 */
#ifndef HYDRODEM_HIP_H
#define HYDRODEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hdem_ctx hdem_ctx;

typedef enum hdem_status {
    HDEM_OK = 0,
    HDEM_ERR_BAD_ARG = 1,       /* null pointer, non-positive size, ...      */
    HDEM_ERR_WINDOW_EVEN = 2,   /* -> WindowSizeEvenError                    */
    HDEM_ERR_WINDOW_HIGH = 3,   /* -> WindowSizeHighError                    */
    HDEM_ERR_HIP = 4,           /* a HIP runtime call failed                 */
    HDEM_ERR_NOT_CONVERGED = 5, /* sink fill hit max_rounds                  */
    HDEM_ERR_NO_DEVICE = 6,
    HDEM_ERR_OOM = 7
} hdem_status;

/* ---- life cycle -------------------------------------------------------- */
const char *hdem_last_error(void);
int hdem_version(void);                       /* 100*major + minor          */
int hdem_device_count(int *count);
int hdem_init(int device, hdem_ctx **ctx);    /* own stream + workspace     */
int hdem_shutdown(hdem_ctx *ctx);
/* Run on a caller-provided hipStream_t (e.g. torch's current stream);
 * NULL restores the context's own stream. */
int hdem_set_stream(hdem_ctx *ctx, void *hip_stream);
int hdem_synchronize(hdem_ctx *ctx);

/* ---- device memory (so that a host needs nothing but this library) ------
 * hdem_free keeps the block (up to a quarter of the device's memory, at most 64 GiB;
 * HDEM_POOL_MIB in the environment: another figure, 0 = none) and hdem_malloc hands it out
 * again for a request of that size or up to an eighth less; the new owner's stream waits
 * for whatever the context's stream held at the time of the free -- no host wait, no
 * hipMalloc / hipFree in a chain of operators that allocates at every call.  A block is
 * handed back on the context it came from; hdem_shutdown returns everything.
 * Contract (single stream): when hdem_free is called, every use of the block must have been
 * enqueued on THIS context's stream (or be complete).  A block that another context, or a
 * stream of the caller's own, still works on must be synchronised by the caller before it is
 * freed (hdem_synchronize on that context); a context whose stream was changed with
 * hdem_set_stream since the block was handed out waits for the whole device in hdem_free.
 * hdem_trim gives the cached blocks (and the scratch buffers the context keeps between calls)
 * back to the device -- for a process that shares the GPU with another allocator (a second
 * context, torch); every allocation the library makes for itself does the same before it
 * reports HDEM_ERR_OOM.  *released (may be NULL): bytes returned. */
int hdem_malloc(hdem_ctx *ctx, size_t bytes, void **dptr);
int hdem_free(hdem_ctx *ctx, void *dptr);
int hdem_trim(hdem_ctx *ctx, size_t *released);
int hdem_memcpy_h2d(hdem_ctx *ctx, void *dst, const void *src, size_t bytes);
int hdem_memcpy_d2h(hdem_ctx *ctx, void *dst, const void *src, size_t bytes);
int hdem_memcpy_d2d(hdem_ctx *ctx, void *dst, const void *src, size_t bytes);
/* A byte fill on the context's stream (0xff bytes make a raster of NaN: the canvas of
 * HydroConditioning.apply_batch). */
int hdem_memset_dev(hdem_ctx *ctx, void *dptr, int byte, size_t bytes);
/* Raster I/O seam (utils_dem.py:17-40 reads / writes whole arrays):
 * page-locked host buffers and copies ordered on the context's stream, so that a band
 * of a raster can be read into pinned memory, processed and written back while the
 * next band is in flight on another context.  hdem_synchronize() waits. */
int hdem_host_alloc(hdem_ctx *ctx, size_t bytes, void **hptr);
int hdem_host_free(hdem_ctx *ctx, void *hptr);
int hdem_memcpy_h2d_async(hdem_ctx *ctx, void *dst, const void *src, size_t bytes);
int hdem_memcpy_d2h_async(hdem_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- per-kernel timing (HIP events on the stream the kernels run on) ---- */
typedef enum hdem_kernel_id {
    HDEM_K_D8 = 0,
    HDEM_K_FILL_INIT = 1,
    HDEM_K_FILL_TILE = 2,     /* sink fill: asynchronous tile relaxation (dominant) */
    HDEM_K_BOXMEAN = 3,
    HDEM_K_GROVES = 4,        /* fused quadratic + groves epilogue          */
    HDEM_K_CONVOLVE = 5,
    HDEM_K_COPY = 6,          /* hdem_copy_rate_dev: the measured copy roof       */
    HDEM_K_FILL_ROUND = 7,    /* certifying stream + finishing rounds of tile visits */
    HDEM_K_BLOCKMAX = 8,      /* block-maximum coarsening (multi-GPU start values) */
    HDEM_K_FFT = 9,           /* rocFFT 2-D complex transform (forward or inverse) */
    HDEM_K_FOURIER_ROWSUM = 10, /* hollow mean, row pass                       */
    HDEM_K_FOURIER_DETECT = 11, /* hollow mean, column pass + peak decision    */
    HDEM_K_FOURIER_MASK = 12,   /* isolated points, expand, apply to spectrum  */
    HDEM_K_FOURIER_POINT = 13,  /* real->complex, |F| of a quadrant, |f|/N     */
    HDEM_K_LAGOON = 14,         /* lagoon branch: NaN repair, morphology, dilation */
    HDEM_K_MAJORITY = 15,       /* majority vote over the circular window       */
    HDEM_K_FILL_COARSE = 16,    /* sink fill: asynchronous launch of the coarse pre-solve */
    HDEM_K_FILL_FLAT = 17,      /* sink fill: interiors of the tiles that ended flat      */
    HDEM_K_ELEMENTWISE = 18,    /* element-wise operators on device rasters               */
    HDEM_K_FILL_HUB = 19,       /* sink fill: hub start (in-tile path costs + hub raster)  */
    HDEM_K_COUNT = 20
} hdem_kernel_id;

typedef struct hdem_kernel_stat {
    int64_t launches;
    double  ms;               /* sum of event-timed launch durations        */
    int64_t units;            /* cells (or tile-visit cells) processed      */
} hdem_kernel_stat;

int hdem_profile_enable(hdem_ctx *ctx, int on);   /* default off            */
int hdem_profile_reset(hdem_ctx *ctx);
/* synchronises the stream, then reports the totals since the last reset */
int hdem_profile_get(hdem_ctx *ctx, int kernel_id, hdem_kernel_stat *out);

/* ---- A2  D8FlowDirection.apply  (new operator; no reference body --------
 * SURVEY F2; tie rule of custom_filters.py:193-195).  ESRI codes, uint8. */
int hdem_d8_f32(hdem_ctx *ctx, const float *z, int H, int W, uint8_t *out);
int hdem_d8_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, uint8_t *out);

/* ---- A1  SinkFill.apply  (new operator; no reference body -- SURVEY F2) -- */
typedef struct hdem_fill_stats {
    int32_t rounds;           /* round-synchronous launches that had work (0: the
                                 certifying stream found the surface final)     */
    int32_t converged;        /* 1: certified fixed point                      */
    int64_t tile_visits;      /* tile visits, both drivers                     */
    int64_t tiles;            /* tiles in the raster                         */
    int32_t tile_h, tile_w;   /* tile shape in cells                         */
    int32_t visits_flat;      /* of tile_visits: tiles under one flat level, settled from
                                 their halo ring alone (no window load)         */
    int32_t async_timed_out;  /* != 0: the asynchronous launch gave up (wall-clock
                                 budget); the round driver finished the fill     */
    int64_t iterations;       /* 4-scan iterations, summed over tile visits  */
    int64_t visits_unchanged; /* visits that found nothing to lower          */
    int64_t visits_requeued;  /* visits that hit the iteration cap           */
    int64_t round_visits;     /* of tile_visits: made by the round driver    */
    int64_t pending;          /* tiles still queued when a time slice ended  */
    int32_t partial_residency;/* 1: some workgroups of the asynchronous launch were not
                                 resident within 200 us (the GPU is shared); the others
                                 started without them and took their tiles       */
    int32_t flat_unchanged;   /* of visits_flat: those that found nothing to lower (they are
                                 part of visits_unchanged as well)               */
    int64_t deferred_visits;  /* of tile_visits / visits_unchanged: made by HDEM_FILL_DEFER  */
    int64_t deferred_unchanged; /* calls before this one, which is the first to report them */
} hdem_fill_stats;

#define HDEM_FILL_INIT        0x0  /* w is output only: pinned ring <- z, rest from above */
#define HDEM_FILL_WARM        0x1  /* w holds a valid upper bound; the one-cell ring of w
                                      is the Dirichlet boundary (never written)           */
#define HDEM_FILL_ACT_ALL     0x0  /* WARM: every tile starts active                      */
#define HDEM_FILL_ACT_TOP     0x2  /* WARM: only tiles touching row 1 ...                 */
#define HDEM_FILL_ACT_BOTTOM  0x4  /* ... and/or row H-2 start active (after a halo       */
                                   /* exchange replaced ghost row 0 / H-1)                */
#define HDEM_FILL_SYNC_ONLY   0x40 /* skip the asynchronous phase (round-synchronous only) */
#define HDEM_FILL_NO_VERIFY   0x80 /* skip the certifying pass behind the asynchronous
                                      phase: for intermediate solves of a halo-exchange loop
                                      whose last solve is a verifying one                  */
#define HDEM_FILL_RESUME      0x100 /* WARM: keep the worklist the previous call on this context
                                      left (a time slice that ended with tiles queued);
                                      ACT_TOP / ACT_BOTTOM add the tile rows next to a replaced
                                      ghost row.  Implies NO_VERIFY.                      */
#define HDEM_FILL_GHOST_TOP   0x10 /* INIT: row 0 / row H-1 is a ghost row owned by the   */
#define HDEM_FILL_GHOST_BOTTOM 0x20 /* neighbouring row block: starts at +inf, not at Z    */
#define HDEM_FILL_NO_COARSE   0x400 /* INIT: start every free cell at +inf.  Without this flag a
                                      raster of >= 6000^2 cells (eps = 0, no ghost rows) is first
                                      filled on its 16 x 16 block maxima -- 1/256 of the cells --
                                      and the free cells start at their block's filled level, an
                                      upper bound of the fill: same bits, ~15 % less time.     */
#define HDEM_FILL_GHOST_GIVEN 0x200 /* INIT with GHOST_TOP / GHOST_BOTTOM: the ghost rows of w
                                      already hold upper bounds of the filled surface (the
                                      caller's guess, e.g. from a coarse solve): start from
                                      those instead of +inf.  Must be >= the true fill.    */

#define HDEM_FILL_DEFER       0x800 /* WARM | RESUME: enqueue the solve and return without
                                      waiting for it and without reading anything back: for the
                                      correcting solves of a halo-exchange loop that keeps its
                                      decisions on the device (hdem_set_fill_seam_words).  The
                                      call's counters are added to those of the next call on this
                                      problem that does wait; stats->pending is -1.          */

/* eps = 0 gives flats (exact, order-independent, bit-reproducible);
 * eps > 0 is the Planchon-Darboux gradient.  max_rounds <= 0 -> default. */
int hdem_sinkfill_f32(hdem_ctx *ctx, const float *z, int H, int W, float eps,
                      int max_rounds, float *w, hdem_fill_stats *stats);
int hdem_sinkfill_f32_dev(hdem_ctx *ctx, const float *z, int H, int W,
                          float eps, int max_rounds, int flags, float *w,
                          hdem_fill_stats *stats);
/* Sink fill and D8 of the filled surface in one call (what HydroConditioning and the
 * headline benchmark run).  The certifying pass of the fill streams over Z and W once;
 * with this entry point it writes the flow directions on the way instead of a second pass
 * over w (same codes as hdem_d8_f32_dev, bit for bit; falls back to that kernel when the
 * call has no certifying pass over the whole raster). */
int hdem_sinkfill_d8_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, float eps,
                             int max_rounds, int flags, float *w, uint8_t *d8,
                             hdem_fill_stats *stats);
/* Time slice of the asynchronous phase in microseconds (0 = run to convergence, the
 * default).  With a slice, an INIT/WARM call that has HDEM_FILL_NO_VERIFY returns
 * HDEM_OK with stats->pending > 0 when the slice ended first; continue with
 * HDEM_FILL_WARM | HDEM_FILL_RESUME.  This is what lets row blocks on different
 * GPUs trade ghost rows every millisecond instead of once per local convergence. */
int hdem_set_fill_slice_us(hdem_ctx *ctx, int microseconds);
/* Seam words of a halo-exchange loop that does not come back to the host between a seam
 * exchange and the solve behind it (new work; partition.py).  `words`: three ints in
 * device memory, owned by the caller, valid until the next call replaces them (NULL: none):
 *   words[1], words[2]   written by the caller on the stream before a HDEM_FILL_DEFER call:
 *                        non-zero = ghost row 0 / H-1 was replaced by different values.  The
 *                        call's HDEM_FILL_ACT_TOP / ACT_BOTTOM then only act when the word is set;
 *   words[0]             written by every HDEM_FILL_DEFER call, on the stream, when its launch has
 *                        ended: 1 = tiles are still queued (a time slice ended first), else 0.
 * The words are read and written by kernels on the context's stream, never by the host. */
int hdem_set_fill_seam_words(hdem_ctx *ctx, int *words);
/* The bookkeeping of one seam exchange as one launch on the context's stream (device pointers):
 * recv_top / recv_bot (W floats each, NULL: no such neighbour) replace row 0 / row H-1 of w;
 * words[1] / words[2] <- that row's bits changed; words[0] <- pending > 0 when pending >= 0
 * (pending < 0: words[0] is what the last deferred call left there); words[3] <- any of
 * words[0..2], the word the ranks MAX-reduce.  `words`: four ints, the first three as above. */
int hdem_fill_seam_apply_dev(hdem_ctx *ctx, float *w, int H, int W, const float *recv_top,
                             const float *recv_bot, int64_t pending, int *words);

/* Hub start of a row-block partition (new work; no reference counterpart -- SURVEY 8e).  The
 * single-GPU fill starts from a graph of tile hubs (one hub per 62 x 62 tile, path costs d to
 * it inside the tile, cheapest crossings between hubs of neighbouring tiles, the graph filled
 * exactly as a small raster: hdem_sinkfill.hip).  A partition builds ONE such graph:
 *   hdem_fill_hub_prepare_dev   d of this block into the interior of w (flags: GHOST_TOP /
 *                               GHOST_BOTTOM as for the fill); the caller then puts the rows of
 *                               the neighbours' d that are its ghost rows into w's ghost rows;
 *   hdem_fill_hub_raster_dev    this block's hub raster, (2 ty + 1) x (2 tx + 1) floats with
 *                               ty x tx = ceil((H - 2) / 62) x ceil((W - 2) / 62) tiles: hubs at
 *                               odd/odd, crossings between them; the first (last) row holds the
 *                               crossings to the raster's first (last) row, or -- ghost row -- to
 *                               the neighbouring block's hubs;
 *   hdem_set_fill_hub_levels    the filled levels of this block's part of the stacked raster
 *                               (same shape), for the next INIT fill of the same z / w: that fill
 *                               skips its own start-value work; with GHOST_GIVEN the ghost rows
 *                               of w are the caller's upper bounds.  Used once. */
int hdem_fill_hub_prepare_dev(hdem_ctx *ctx, const float *z, int H, int W, int flags, float *w);
int hdem_fill_hub_raster_dev(hdem_ctx *ctx, float *raster);
int hdem_set_fill_hub_levels(hdem_ctx *ctx, const float *levels);
/* Start values for the NEXT hdem_sinkfill_f32_dev INIT call with eps = 0 on this context:
 * coarse_filled is the sink fill of a block-maximum raster (hdem_blockmax_f32_dev) that
 * covers this raster -- ch x cw floats on the device, block a power of two in 4..256;
 * free cells (and ghost rows) start at max(z, coarse level of their block).  row_map
 * (device, H ints, or NULL for y / block) gives the coarse row of each raster row: a row
 * block of a partitioned raster points into the stacked coarse raster of all ranks.
 * Used once; NULL clears it. */
int hdem_set_fill_coarse_start(hdem_ctx *ctx, const float *coarse_filled, int ch, int cw,
                               int block, const int32_t *row_map);

/* Block maximum: out[i][j] = max of z over rows [i*b, (i+1)*b) x columns [j*b, (j+1)*b)
 * (clipped to the raster; a block with a NaN cell gives FLT_MAX), b a power of two in
 * 4..256, out is ceil(H/b) x ceil(W/b).  The sink fill of this coarse raster bounds the
 * sink fill of z from above cell by cell -- the multi-GPU path uses it as the start value
 * of the ghost rows (new work; the reference is single process). */
int hdem_blockmax_f32_dev(hdem_ctx *ctx, const float *z, int H, int W, int b, float *out);

/* Measurement aid ("report % of measured copy bandwidth as well as % of
 * nominal"): a plain 16-byte-per-lane device copy of `bytes` bytes (a multiple of 16,
 * both pointers 16-byte aligned), timed under HDEM_K_COPY (units = bytes copied); it moves
 * 2 * bytes through HBM.  bench.py quotes every kernel against the rate it reaches. */
int hdem_copy_rate_dev(hdem_ctx *ctx, const void *src, void *dst, size_t bytes);

/* ---- element-wise operators ------------------------------------
 * LowerThan / GreaterThan / BooleanToInteger / ProductFilter / AdditionFilter /
 * SubtractionFilter.apply, simple_filters.py:7-275, on device rasters, so that the
 * orchestration's algebra (hydro_dem_process.py:60-91, and the three-term sum in front
 * of PostProcessingFinal, :148-149) stays in HBM between the stencils:
 *     out[i] = op(image[i], operand ? operand[i] : scalar),  i < n.
 * Rasters are HDEM_T_F32 / HDEM_T_F64 / HDEM_T_U8 (masks); evaluated in double, stored
 * as out_type (comparisons and NONZERO give 0 / 1). */
typedef enum hdem_ew_op {
    HDEM_EW_MUL = 0,      /* operand * image      ProductFilter     :165-180 */
    HDEM_EW_ADD = 1,      /* operand + image      AdditionFilter    :214-229 */
    HDEM_EW_RSUB = 2,     /* operand - image      SubtractionFilter :261-275 */
    HDEM_EW_GT = 3,       /* image > operand      GreaterThan       :81-96   */
    HDEM_EW_LT = 4,       /* image < operand      LowerThan         :35-50   */
    HDEM_EW_NONZERO = 5   /* image != 0           BooleanToInteger  :113-131 (bool * 1) */
} hdem_ew_op;
typedef enum hdem_dtype { HDEM_T_F32 = 0, HDEM_T_F64 = 1, HDEM_T_U8 = 2, HDEM_T_I64 = 3 } hdem_dtype;
int hdem_elementwise_dev(hdem_ctx *ctx, int op, const void *image, int image_type,
                         const void *operand, int operand_type, double scalar, int64_t n,
                         void *out, int out_type);

/* ---- A8  Fourier destripe ------------------------------------
 * DetectApplyFourier.apply, custom_filters.py:1083-1101, with FourierInitial
 * (:834-877), FourierProcessQuarters (:880-1050) and MaskFourier (:537-561)
 * inside: out = | ifft2( (1 - mask) * fft2(dem) ) |, the mask found on the
 * magnitude of the two upper quadrants of the shifted spectrum.  float32 /
 * complex64 throughout (scipy.fftpack keeps float32 input single; the reference
 * then drifts to complex128 for the inverse: values agree to ~1e-5 m).
 * mask (optional, H*W bytes): the reference's masks_fourier in shifted
 * coordinates.  Quadrants smaller than the 55-cell window give
 * HDEM_ERR_WINDOW_HIGH, as the reference's window constructor does. */
int hdem_fourier_destripe_f32(hdem_ctx *ctx, const float *dem, int H, int W,
                              float *out, uint8_t *mask);
int hdem_fourier_destripe_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W,
                                  float *out, uint8_t *mask);
/* BlanksFourier.apply (:400-429), `window` odd in 7..201 (the reference's pipeline: 55),
 * inner 5, factor 4: found[i] = 1 where q > 4 x hollow mean; q is rewritten as
 * q * (1 - found). */
int hdem_blanks_fourier_f32_dev(hdem_ctx *ctx, float *q, int h, int w, int window,
                                uint8_t *found);
/* IsolatedPoints.apply (:344-366) and ExpandFilter.apply (:103-125) on byte masks. */
int hdem_isolated_points_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int h, int w,
                                int window, uint8_t *out);
int hdem_expand_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int h, int w, int window,
                       uint8_t *out);
/* FourierTransform / FourierITransform (extension_filters.py:348-414) on an
 * interleaved complex64 H x W array, in place; the inverse is unnormalised
 * (scipy's ifft2 = this / (H*W)). */
int hdem_fft2_c2c_f32_dev(hdem_ctx *ctx, float *data, int H, int W, int inverse);
/* ... and on an interleaved complex128 array: what scipy.fftpack computes for float64 /
 * complex128 input (extension_filters.py:379,414).  Plan made and released per call. */
int hdem_fft2_c2c_f64_dev(hdem_ctx *ctx, double *data, int H, int W, int inverse);

/* ---- SURVEY 8f-3  HydroSHEDS / lagoon branch --------------------------------
 * CorrectNANValues.apply (custom_filters.py:287-317): cells < 0 whose `window` (odd, 3..11;
 * the reference's pipeline: 3) fits <- float32 mean of the window's other cells >= 0, summed
 * as NumPy sums them (NaN when there is none). */
int hdem_correct_nan_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W, int window,
                             float *out);
/* MajorityFilter.apply (:44-73): value held by > 70 % of (window^2 - 1) cells of the
 * window minus its corners, else 0; window odd, 3..15. */
int hdem_majority_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, int window,
                          float *out);
/* scipy.ndimage.binary_erosion(iterations) / binary_closing(structure)
 * (extension_filters.py:187-293): byte masks, border_value 0; structure = sh x sw bytes,
 * odd sizes up to 7, NULL = the 3 x 3 cross.  tmp: a scratch mask of the same size
 * (may be NULL for a single erosion). */
int hdem_binary_erosion_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int H, int W,
                               const uint8_t *structure, int sh, int sw, int iterations,
                               uint8_t *tmp, uint8_t *out);
int hdem_binary_closing_u8_dev(hdem_ctx *ctx, const uint8_t *mask, int H, int W,
                               const uint8_t *structure, int sh, int sw, uint8_t *tmp,
                               uint8_t *out);
/* scipy.ndimage.grey_dilation(size=(sy, sx)) (extension_filters.py:296-345): flat
 * maximum filter, mode 'reflect', odd sizes. */
int hdem_grey_dilation_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, int sy, int sx,
                               float *out);
int hdem_grey_dilation_f64_dev(hdem_ctx *ctx, const double *img, int H, int W, int sy, int sx,
                               double *out);
/* TidyingLagoons.apply (:564-610) and LagoonsDetection.apply (:613-661), device
 * resident.  fixed / values (may be NULL): the reference's hsheds_nan_fixed and
 * lagoons_values; mask: 1 where lagoons_values > 0. */
int hdem_tidying_lagoons_f32_dev(hdem_ctx *ctx, const float *img, int H, int W, float *out);
int hdem_lagoons_detection_f32_dev(hdem_ctx *ctx, const float *hsheds, int H, int W,
                                   float *fixed, float *values, uint8_t *mask);

/* ---- A5  Convolve.apply + Around.apply -----------------------------------
 * extension_filters.py:166-184 (scipy.ndimage.convolve, mode='reflect',
 * double accumulation, result / weights.size) and :113-130 (np.around).
 * boxmean3 is the ones((3,3)) default fused with the optional rounding
 * (PostProcessingFinal, custom_filters.py:1124-1125). */
int hdem_boxmean3_f32(hdem_ctx *ctx, const float *x, int H, int W,
                      int do_round, float *out);
int hdem_boxmean3_f64(hdem_ctx *ctx, const double *x, int H, int W,
                      int do_round, double *out);
int hdem_boxmean3_f32_dev(hdem_ctx *ctx, const float *x, int H, int W,
                          int do_round, float *out);
int hdem_boxmean3_f64_dev(hdem_ctx *ctx, const double *x, int H, int W,
                          int do_round, double *out);
/* general odd kh x kw weights (host pointer, row-major doubles), <= 15x15 */
int hdem_convolve_f32(hdem_ctx *ctx, const float *x, int H, int W,
                      const double *weights, int kh, int kw, float *out);
int hdem_convolve_f64(hdem_ctx *ctx, const double *x, int H, int W,
                      const double *weights, int kh, int kw, double *out);
int hdem_around_f32(hdem_ctx *ctx, const float *x, int64_t n, float *out);
int hdem_around_f64(hdem_ctx *ctx, const double *x, int64_t n, double *out);

/* ---- A3  QuadraticFilter.apply  (custom_filters.py:226-257) -------------
 * ws odd and <= min(H, W) (sliding_window.py:150-156); border ring of ws/2
 * cells is returned unchanged (custom_filters.py:249). */
int hdem_quadratic_f32(hdem_ctx *ctx, const float *dem, int H, int W, int ws,
                       float *out);
int hdem_quadratic_f32_dev(hdem_ctx *ctx, const float *dem, int H, int W,
                           int ws, float *out);

/* ---- A4  GrovesCorrection / GrovesCorrectionsIter.apply -----------------
 * custom_filters.py:708-732,755-767 with MaskTallGroves :533-534 fused:
 *   smooth = quadratic(img); hl = img - smooth; m = groves && hl > thr;
 *   out = m ? smooth : hl + smooth;   repeated `iters` times.
 * groves: uint8, non-zero = groves class.  scratch_dev (iters > 1) is a
 * second H*W float device buffer for the ping-pong; may be NULL for the
 * host-pointer variant. */
int hdem_groves_f32(hdem_ctx *ctx, const float *img, const uint8_t *groves,
                    int H, int W, int ws, float thr, int iters, float *out);
int hdem_groves_f32_dev(hdem_ctx *ctx, const float *img, const uint8_t *groves,
                        int H, int W, int ws, float thr, int iters,
                        float *scratch_dev, float *out);

#ifdef __cplusplus
}
#endif
#endif /* HYDRODEM_HIP_H */
