"""
Regenerates tests/golden/windows.npz.  RUNS ONLY IN THE BUILD CONTAINER (see
make_golden.py): outputs of the imported reference for the constructor arguments its own
pipeline never uses -- CorrectNANValues(window_size=5, 7) and BlanksFourier(window_size=
15, 21, 35) -- so that the kernels that take any odd window are pinned too.

    python tests/golden/make_golden_windows.py
"""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from filters.custom_filters import CorrectNANValues, BlanksFourier  # noqa: E402
from oracle.hdem_oracle_lagoons import synth_hsheds  # noqa: E402


def main():
    warnings.simplefilter("ignore")
    out = {}
    hs = synth_hsheds(70, 96)
    rng = np.random.default_rng(12)
    hs[rng.random(hs.shape) < 0.03] = -32768.0             # more voids, some side by side
    hs[30:34, 40:45] = -32768.0                            # a void larger than a 3 x 3 window
    hs[10, 10] = np.nan
    out["hs"] = hs
    for ws in (3, 5, 7):
        out[f"fixed{ws}"] = CorrectNANValues(window_size=ws).apply(hs.copy())
        print("CorrectNANValues", ws, "changed", int((out[f"fixed{ws}"] != hs).sum()),
              "NaN", int(np.isnan(out[f"fixed{ws}"]).sum()))
    # a spectrum-like magnitude raster: ten decades, a few peaks, a NaN
    q = np.exp(rng.normal(0.0, 2.0, (90, 120))).astype(np.float32)
    for y, x in ((20, 30), (21, 31), (50, 80), (5, 5), (88, 118), (45, 60)):
        q[y, x] *= 300.0
    out["q"] = q
    for ws in (15, 21, 35):
        found, modified = BlanksFourier(window_size=ws).apply(q.copy())
        out[f"found{ws}"] = np.asarray(found).astype(np.uint8)
        out[f"modified{ws}"] = np.asarray(modified)
        print("BlanksFourier", ws, "found", int(found.sum()), modified.dtype)
    np.savez_compressed(os.path.join(HERE, "windows.npz"), **out)


if __name__ == "__main__":
    main()
