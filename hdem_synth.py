"""
Synthetic workloads: the rasters `bench.py`, the tests, the tools and the
oracle's self-checks all draw from, so that the GPU run and every CPU checker see the
same cells.  Pure NumPy data generation -- no part of the hot path and no restatement of
it; `oracle/` re-exports these names for the tests.

Block-seeded: rows [r0, r1) of an H x W raster are identical whatever partition asks
for them.
"""
import os

import numpy as np

GEN_SEED = 20240607
GEN_BLOCK = 1024


def synth_dem(h, w_, row0=0, rows=None, variant="rough", pits=True,
              total_rows=None):
    """Rows [row0, row0+rows) of the synthetic H x W float32 DEM.

    Z = 100 + 0.002 x + 0.001 y + sum_k A_k sin(2 pi (x u_k + y v_k))
        + 0.5 * N(0,1),  A = 8,4,2,1 m at wavelengths 4096,1024,256,64 cells,
    plus ~0.1 % single-cell pits (-5 m).  ``variant='srtm'`` rounds to
    integer metres (large flats and ties, the regime of ``final_dem.tif``).
    """
    rows = h - row0 if rows is None else rows
    out = np.empty((rows, w_), dtype=np.float32)
    x = np.arange(w_, dtype=np.float64)[None, :]
    amps = (8.0, 4.0, 2.0, 1.0)
    lams = (4096.0, 1024.0, 256.0, 64.0)
    angs = (0.3, 1.1, 2.0, 2.9)

    base = 100.0 + 0.002 * x
    # sin(px + py) expanded so that only 1-D sines are evaluated
    waves = [(a, np.sin(2 * np.pi * np.cos(ang) / lam * x), np.cos(2 * np.pi * np.cos(ang) / lam * x),
              2 * np.pi * np.sin(ang) / lam) for a, lam, ang in zip(amps, lams, angs)]
    step = max(1, (1 << 19) // max(w_, 1))          # rows per chunk: ~4 MB of float64

    def block(blk):
        b0 = blk * GEN_BLOCK
        b1 = min(b0 + GEN_BLOCK, h)
        rng = np.random.default_rng([GEN_SEED, blk, w_])
        noise = rng.standard_normal((b1 - b0, w_), dtype=np.float32)
        pit = rng.random((b1 - b0, w_), dtype=np.float32) < 0.001
        np.multiply(noise, np.float32(0.5), out=noise)
        lo, hi = max(row0, b0), min(row0 + rows, b1)
        # chunks through two reused buffers: threads that allocate block-sized temporaries
        # spend their time in page faults, which one process takes one at a time
        zz_buf, t_buf, u_buf = (np.empty((step, w_)) for _ in range(3))
        for c0 in range(lo, hi, step):
            c1 = min(c0 + step, hi)
            n = c1 - c0
            zz, t, u = zz_buf[:n], t_buf[:n], u_buf[:n]
            y = np.arange(c0, c1, dtype=np.float64)[:, None]
            np.add(base, 0.001 * y, out=zz)
            for a, sin_px, cos_px, ky in waves:
                py = ky * y
                np.multiply(sin_px, np.cos(py), out=t)
                np.multiply(cos_px, np.sin(py), out=u)
                np.add(t, u, out=t)
                np.multiply(t, a, out=t)
                np.add(zz, t, out=zz)
            np.add(zz, noise[c0 - b0:c1 - b0], out=zz)
            if pits:
                np.subtract(zz, 5.0, out=zz, where=pit[c0 - b0:c1 - b0])
            out[c0 - row0:c1 - row0] = zz

    blocks = range(row0 // GEN_BLOCK, (row0 + rows - 1) // GEN_BLOCK + 1) if rows > 0 else ()
    workers = max(1, min(len(blocks), os.cpu_count() or 1, 16))
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(block, blocks))
    else:
        for blk in blocks:
            block(blk)
    if variant == "srtm":
        out = np.round(out).astype(np.float32)
    return out


def synth_groves(h, w_, seed=7):
    """Bernoulli(0.05) mask closed with a 3x3 structuring element (the
    reference closes its class raster the same way, image_srtm.py:177-178);
    uint8 0/1."""
    from scipy.ndimage import binary_closing
    rng = np.random.default_rng([GEN_SEED, seed, h, w_])
    raw = rng.random((h, w_), dtype=np.float32) < 0.05
    return binary_closing(raw, structure=np.ones((3, 3))).astype(np.uint8)


def synth_hsheds(h, w_, seed=11):
    """HydroSHEDS-like raster: integer-metre terrain, flat lagoons (constant
    elevation discs), and a few voids coded as large negative numbers."""
    rng = np.random.default_rng([20240607, seed, h, w_])
    z = np.round(synth_dem(h, w_, pits=False)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w_]
    for _ in range(max(2, h * w_ // 6000)):
        cy, cx = rng.integers(0, h), rng.integers(0, w_)
        rad = rng.integers(5, 14)
        disc = (yy - cy) ** 2 + (xx - cx) ** 2 <= rad ** 2
        z[disc] = z[min(cy, h - 1), min(cx, w_ - 1)]
    void = rng.random((h, w_)) < 0.004
    z[void] = -32768.0
    z[h // 3:h // 3 + 3, w_ // 4:w_ // 4 + 4] = -32768.0        # voids with only void neighbours
    return z


def synth_striped_dem(h, w_, seed=5, stripes=((0.31, 0.07, 1.2), (0.12, 0.38, 0.8))):
    """Synthetic DEM with a few plane waves of 1 m scale on top (the SRTM striping
    artefact the destripe removes): each (fy, fx, amplitude), cycles per cell."""
    z = synth_dem(h, w_, pits=False).astype(np.float64)
    y, x = np.mgrid[0:h, 0:w_]
    rng = np.random.default_rng([20240607, seed])
    for fy, fx, amp in stripes:
        z += amp * np.sin(2 * np.pi * (fy * y + fx * x) + rng.uniform(0, 2 * np.pi))
    return z.astype(np.float32)
