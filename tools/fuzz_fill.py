"""Random differential test of the sink fill + D8 against the C oracle (exploration; the suite's
own random test is tests/test_gpu_parity.py): odd shapes, nodata sprinkles and blocks, eps = 0 and
> 0, the hub start forced on tiny rasters and off.
usage: python tools/fuzz_fill.py [cases] [seed] [largest edge]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
edge = int(sys.argv[3]) if len(sys.argv) > 3 else 420
os.environ["HDEM_HUB_MIN_TILES"] = "1"
os.environ["HDEM_HUB_MIN_TILES_NESTED"] = "1"
from hydrodem_amd import backend as B
from oracle import c_oracle
bad = 0
for k in range(cases):
    h, w = int(rng.integers(3, edge)), int(rng.integers(3, edge))
    kind = k % 4
    z = (rng.normal(0, 1, (h, w)) * rng.choice([0.01, 1.0, 30.0]) + rng.choice([0.0, 500.0, -40.0])).astype(np.float32)
    if kind == 1:
        z = np.round(z)                                 # ties and flats
    if kind >= 2:
        z[rng.random((h, w)) < 0.01] = np.nan
        if kind == 3 and h > 20 and w > 20:
            y, x = int(rng.integers(0, h - 10)), int(rng.integers(0, w - 10))
            z[y:y + int(rng.integers(1, 10)), x:x + int(rng.integers(1, 10))] = np.nan
    eps = float(rng.choice([0.0, 0.0, 1e-3, 1e-4]))
    os.environ["HDEM_FILL_HUB"] = "1" if k % 3 else "0"
    want = c_oracle.sinkfill_pflood(z, eps)
    want_d8 = c_oracle.d8(want)
    zd = B.DeviceRaster.from_host(z)
    wd, dd, st = B.sinkfill_d8_dev(zd, eps=eps)
    got, got_d8 = wd.to_host(), dd.to_host()
    for r in (zd, wd, dd): r.free()
    ok = np.array_equal(got, want, equal_nan=True) and np.array_equal(got_d8, want_d8)
    if not ok:
        bad += 1
        m = ~((got == want) | (np.isnan(got) & np.isnan(want)))
        print(f"case {k}: {h} x {w} kind {kind} eps {eps} hub {os.environ['HDEM_FILL_HUB']}: {int(m.sum())} cells, "
              f"{int((got_d8 != want_d8).sum())} codes differ; first {np.argwhere(m)[:3].tolist()}", flush=True)
print(f"{cases} cases, {bad} mismatching", flush=True)
