"""
Regenerates tests/golden/lagoons.npz.  RUNS ONLY IN THE BUILD CONTAINER (see
make_golden.py).  Holds (1) three rasters of the reference's own test suite
(tests/resources/tests_expected.zip, read with Pillow) that form a chain the
imported reference reproduces exactly, and (2) seeded inputs with the outputs of
the imported reference operators of the HydroSHEDS / lagoon branch.

    python tests/golden/make_golden_lagoons.py
"""
import io
import os
import sys
import warnings
import zipfile

import numpy as np

REF = "/root/reference/cguerrero"
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "hydrodem"))
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from filters.custom_filters import (CorrectNANValues, MajorityFilter,  # noqa: E402
                                    TidyingLagoons, LagoonsDetection, ExpandFilter,
                                    MaskPositives, MaskNegatives)
from filters.extension_filters import (BinaryErosion, BinaryClosing,  # noqa: E402
                                       GreyDilation, BitwiseXOR)
from oracle.hdem_oracle_lagoons import synth_hsheds  # noqa: E402


def tif(zf, name):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(zf.read(name))))


def main():
    warnings.simplefilter("ignore")
    out = {}
    zf = zipfile.ZipFile(os.path.join(REF, "tests/resources/tests_expected.zip"))
    nanv = tif(zf, "expected/hsheds_nan_values_expected.tif")
    maj = tif(zf, "expected/hsheds_majority_11_expected.tif")
    lag = tif(zf, "expected/lagoons_expected.tif")
    # the chain closes with the imported reference: check before storing
    assert np.array_equal(MajorityFilter(window_size=11).apply(nanv), maj)
    assert np.array_equal(TidyingLagoons().apply(maj), lag)
    print("reference rasters", nanv.shape, nanv.dtype, "distinct", len(np.unique(nanv)),
          "majority cells", int((maj != 0).sum()), "lagoon cells", int((lag != 0).sum()))
    out.update(ref_nan_values=nanv, ref_majority_11=maj, ref_lagoons=lag)

    hs = synth_hsheds(90, 110)
    fixed = CorrectNANValues().apply(hs.copy())
    det = LagoonsDetection()
    mask = det.apply(hs.copy())
    out.update(hs=hs, hs_fixed=fixed, hs_majority=det.results["MajorityFilter"],
               hs_tidy=det.results["TidyingLagoons"], hs_mask=np.asarray(mask).astype(np.uint8),
               hs_majority5=MajorityFilter(window_size=5).apply(fixed.copy()))
    assert np.array_equal(det.results["CorrectNANValues"], fixed, equal_nan=True)
    print("synthetic", hs.shape, "voids", int((hs < 0).sum()), "fixed NaN", int(np.isnan(fixed).sum()),
          "majority cells", int((det.results["MajorityFilter"] != 0).sum()),
          "lagoon cells", int(np.asarray(mask).sum()))

    rng = np.random.default_rng(77)
    m = rng.random((60, 70)) < 0.55
    m[20:40, 25:50] = True
    m[0, :] = True
    st = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], dtype=bool)
    img = (rng.random((41, 37)) * 50).astype(np.float32)
    out.update(morph_in=m.astype(np.uint8),
               erosion1=BinaryErosion(iterations=1).apply(m).astype(np.uint8),
               erosion2=BinaryErosion(iterations=2).apply(m).astype(np.uint8),
               closing_default=BinaryClosing().apply(m).astype(np.uint8),
               closing_ones3=BinaryClosing(structure=np.ones((3, 3))).apply(m).astype(np.uint8),
               closing_cross=BinaryClosing(structure=st).apply(m).astype(np.uint8),
               grey_in=img, grey77=GreyDilation(size=(7, 7)).apply(img),
               grey35=GreyDilation(size=(3, 5)).apply(img),
               expand7=ExpandFilter(window_size=7).apply(m.astype(np.float64)).astype(np.uint8),
               xor=BitwiseXOR(operand=m.astype(np.int64)).apply((~m).astype(np.int64) * 3),
               positives=MaskPositives().apply(img - 25), negatives=MaskNegatives().apply(img - 25))
    path = os.path.join(HERE, "lagoons.npz")
    np.savez_compressed(path, **out)
    print("lagoons.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
