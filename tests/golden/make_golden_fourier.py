"""
Regenerates tests/golden/fourier.npz.  RUNS ONLY IN THE BUILD CONTAINER, where
the reference checkout is mounted read-only at /root/reference (see
make_golden.py).  Stores seeded inputs and what the *imported reference
operators* of the Fourier destripe chain return for them, stage by stage.

    python tests/golden/make_golden_fourier.py
"""
import os
import sys

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from filters.custom_filters import (BlanksFourier, DetectBlanksFourier,  # noqa: E402
                                    IsolatedPoints, ExpandFilter, MaskFourier,
                                    FourierInitial, FourierProcessQuarters,
                                    DetectApplyFourier)
from oracle.hdem_oracle_fourier import synth_striped_dem, quarter_slices  # noqa: E402


def main():
    out = {}
    for tag, (h, w_) in {"even": (150, 168), "odd": (141, 155)}.items():
        stripes = ((0.31, 0.07, 1.2), (0.12, -0.38, 0.8)) if tag == "odd" else \
            ((0.31, 0.07, 1.2), (0.12, 0.38, 0.8), (0.2, -0.21, 0.5))
        dem = synth_striped_dem(h, w_, seed=len(tag), stripes=stripes)
        init = FourierInitial()
        mag = init.apply(dem)
        s1, s2 = quarter_slices(h, w_)
        q1 = mag[s1].copy()
        found1, q1_mod = BlanksFourier(window_size=55).apply(q1.copy())
        det1 = DetectBlanksFourier().apply(q1.copy())
        det2 = DetectBlanksFourier().apply(mag[s2].copy())
        iso1 = IsolatedPoints(window_size=3).apply(det1.copy())
        exp1 = ExpandFilter(window_size=13).apply(iso1.copy())
        m1 = MaskFourier().apply(q1.copy())
        assert np.array_equal(m1, exp1)
        full = FourierProcessQuarters(mag).apply(None)
        result = DetectApplyFourier().apply(dem)
        print(tag, dem.shape, "spectrum", init.fourier_shift.dtype, "mag", mag.dtype,
              "detected", int(det1.sum()), int(det2.sum()), "isolated kept", int(iso1.sum()),
              "expanded", int(exp1.sum()), "mask", int(full.sum()), "result", result.dtype,
              "max |dem - result|", float(np.abs(result - dem).max()))
        out.update({f"{tag}_dem": dem, f"{tag}_mag": mag, f"{tag}_found1": found1.astype(np.uint8),
                    f"{tag}_q1_mod": q1_mod, f"{tag}_det1": det1.astype(np.uint8),
                    f"{tag}_det2": det2.astype(np.uint8), f"{tag}_iso1": iso1.astype(np.uint8),
                    f"{tag}_exp1": exp1.astype(np.uint8), f"{tag}_mask": full.astype(np.uint8),
                    f"{tag}_result": result})
    # the mask stencils alone, on a mask with isolated points, clusters and border cells
    rng = np.random.default_rng(99)
    m = (rng.random((40, 47)) < 0.03).astype(np.float64)
    m[10:12, 20:22] = 1
    m[0, 5] = m[39, 46] = m[17, 0] = 1
    iso = IsolatedPoints(window_size=3).apply(m.copy())
    out.update(st_mask=m.astype(np.uint8), st_iso=iso.astype(np.uint8),
               st_exp=ExpandFilter(window_size=13).apply(iso.copy()).astype(np.uint8),
               st_exp5=ExpandFilter(window_size=5).apply(m.copy()).astype(np.uint8))
    path = os.path.join(HERE, "fourier.npz")
    np.savez_compressed(path, **out)
    print("fourier.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
