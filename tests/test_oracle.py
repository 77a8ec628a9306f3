"""
CPU suite, part 1: the oracle itself.

* pinned pieces (quadratic / groves / box mean) against the golden vectors
  produced by the imported reference (tests/golden/make_golden.py) and against
  rasters of the reference's own test suite;
* unpinned pieces (sink fill / D8: the reference has neither) through
  properties, hand grids and two independent algorithms agreeing bit for bit.
"""
import numpy as np
import pytest

import oracle
from oracle import c_oracle


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    yield


# ---- pinned: quadratic ----------------------------------------------------
def test_quadratic_restatement_is_bit_exact(golden):
    g = golden("quadratic.npz")
    for a, b, ws in (("dem", "q15", 15), ("dem_srtm", "q15s", 15), ("q5_in", "q5", 5),
                     ("dem64", "q15_64", 15)):
        got = c_oracle.quadratic_ref(g[a], ws)
        assert got.dtype == g[b].dtype
        assert np.array_equal(got, g[b]), a


def test_quadratic_exact64_within_reference_rounding(golden):
    g = golden("quadratic.npz")
    assert np.abs(oracle.quadratic_exact64(g["dem"], 15) - g["q15"]).max() < 1e-4
    # K = a(xx^2+yy^2)+b sums to one and reproduces the closed form
    v, (r0, r1, r2, r3, den), (a, b) = oracle.quadratic_constants(15)
    xx, yy = np.meshgrid(v, v)
    k = a * (xx * xx + yy * yy) + b
    assert abs(k.sum() - 1) < 1e-12
    assert abs(a - (-2.862691686844229e-4)) < 1e-15 and abs(b - 1.5274961326338444e-2) < 1e-14
    win = g["dem"][:15, :15].astype(np.float64)
    assert abs((k * win).sum() - oracle.quadratic_exact64(g["dem"], 15)[7, 7]) < 1e-9


# ---- pinned: groves ---------------------------------------------------------
def test_groves_restatement_is_bit_exact(golden):
    g = golden("groves.npz")
    out1, m = c_oracle.groves_ref(g["img"], g["groves"], 1, masks=True)
    assert np.array_equal(out1, g["out1"]) and np.array_equal(m[0], g["mask1"])
    assert g["mask1"].sum() > 10
    assert np.array_equal(c_oracle.groves_ref(g["img"], g["groves"], 3), g["out3"])
    e3, _ = oracle.groves_exact64(g["img"], g["groves"], 3)
    assert np.abs(e3 - g["out3"]).max() < 1e-4


def test_groves_known_answer_from_reference_test_rasters(golden):
    g = golden("ref_rasters.npz")
    fc, sp = g["fourier_corrected"], g["srtm_processed"]
    mask = np.abs(fc.astype(np.float64) - sp) > 1e-3
    for it in (1, 3):
        out = c_oracle.groves_ref(fc, mask, it)
        r = 7 * it
        assert np.abs(out - sp)[r:-r, r:-r].max() < 2e-4
    assert mask[7:-7, 7:-7].sum() > 900


# ---- pinned: box mean + round ----------------------------------------------
def test_boxmean_restatements_are_bit_exact(golden):
    g = golden("boxmean.npz")
    for k in ("32", "int", "64"):
        x = g["x" + k]
        assert np.array_equal(c_oracle.boxmean3(x, False), g["conv" + k])
        assert np.array_equal(c_oracle.boxmean3(x, True), g["final" + k])
        assert np.array_equal(oracle.boxmean3(x), g["conv" + k])
        assert np.array_equal(oracle.boxmean3_round(x), g["final" + k])


def test_boxmean_against_scipy_here():
    from scipy.ndimage import convolve
    for shape in [(1, 1), (1, 6), (2, 2), (40, 33)]:
        x = oracle.synth_dem(*shape)
        want = convolve(x, weights=np.ones((3, 3))) / 9
        assert np.array_equal(c_oracle.boxmean3(x, False), want)
        assert np.array_equal(c_oracle.boxmean3(x, True), np.around(want))


# ---- unpinned: sink fill -----------------------------------------------------
@pytest.mark.parametrize("shape", [(3, 3), (5, 9), (40, 41), (128, 96)])
@pytest.mark.parametrize("variant", ["rough", "srtm"])
def test_sinkfill_two_algorithms_agree(shape, variant):
    z = oracle.synth_dem(*shape, variant=variant)
    wj, _ = oracle.sinkfill_jacobi(z)
    assert np.array_equal(wj, c_oracle.sinkfill_pflood(z))


def test_sinkfill_properties():
    z = oracle.synth_dem(160, 200)
    w = c_oracle.sinkfill_pflood(z)
    assert (w >= z).all()
    assert np.array_equal(w[0], z[0]) and np.array_equal(w[-1], z[-1])
    assert np.array_equal(w[:, 0], z[:, 0]) and np.array_equal(w[:, -1], z[:, -1])
    m = oracle.hdem_oracle_np._min8(w)
    wi, zi = w[1:-1, 1:-1], z[1:-1, 1:-1]
    assert (m <= wi).all()                    # no strict pit remains
    raised = wi > zi
    assert raised.any() and np.array_equal(wi[raised], m[raised])   # minimality
    assert np.array_equal(c_oracle.sinkfill_pflood(w), w)           # idempotent
    assert oracle.sinkfill_is_fixed_point(z, w)
    # order independence: Gauss-Seidel row sweeps from the same start
    g = oracle.sinkfill_init(z)
    for _ in range(10000):
        before = g.copy()
        for y in list(range(1, g.shape[0] - 1)) + list(range(g.shape[0] - 2, 0, -1)):
            mm = np.minimum(np.minimum(g[y - 1, :-2], g[y - 1, 1:-1]), g[y - 1, 2:])
            mm = np.minimum(mm, np.minimum(np.minimum(g[y + 1, :-2], g[y + 1, 1:-1]), g[y + 1, 2:]))
            mm = np.minimum(mm, np.minimum(g[y, :-2], g[y, 2:]))
            g[y, 1:-1] = np.maximum(z[y, 1:-1], np.minimum(g[y, 1:-1], mm))
        if np.array_equal(before, g):
            break
    assert np.array_equal(g, w)


def test_sinkfill_spill_elevation_by_brute_force():
    # closed form: min over 8-connected paths to the border of the path maximum
    rng = np.random.default_rng(5)
    z = rng.integers(0, 9, (9, 10)).astype(np.float32)
    w = c_oracle.sinkfill_pflood(z)
    h, w_ = z.shape
    for level in np.unique(z):
        # cells connected to the border through cells <= level are exactly {W <= level}
        ok = z <= level
        reach = np.zeros_like(ok)
        reach[0], reach[-1], reach[:, 0], reach[:, -1] = ok[0], ok[-1], ok[:, 0], ok[:, -1]
        while True:
            grow = reach.copy()
            p = np.pad(reach, 1)
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    grow |= p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w_]
            grow &= ok
            if np.array_equal(grow, reach):
                break
            reach = grow
        assert np.array_equal(reach, w <= level)


def test_sinkfill_epsilon_and_nodata():
    z = oracle.synth_dem(70, 80)
    for eps in (1e-3, 0.05):
        wj, _ = oracle.sinkfill_jacobi(z, eps=eps)
        assert np.array_equal(wj, c_oracle.sinkfill_pflood(z, eps=eps))
        m = oracle.hdem_oracle_np._min8(wj)
        raised = wj[1:-1, 1:-1] > z[1:-1, 1:-1]
        assert (m[raised] < wj[1:-1, 1:-1][raised]).all()   # strictly draining
    z[30:35, 40:50] = np.nan
    wj, _ = oracle.sinkfill_jacobi(z)
    wp = c_oracle.sinkfill_pflood(z)
    assert np.array_equal(np.isnan(wj), np.isnan(z))
    assert np.array_equal(np.nan_to_num(wj, nan=-1), np.nan_to_num(wp, nan=-1))
    assert np.array_equal(wj[29, 39:51], z[29, 39:51])      # neighbours of nodata pinned


# ---- unpinned: D8 ------------------------------------------------------------
def test_d8_two_implementations_agree_and_properties(golden):
    for z in (oracle.synth_dem(90, 70), oracle.synth_dem(64, 64, variant="srtm"),
              golden("ref_rasters.npz")["final_dem"]):
        d = c_oracle.d8(z)
        assert np.array_equal(d, oracle.d8_flow_direction(z))
        assert set(np.unique(d)) <= {0, 1, 2, 4, 8, 16, 32, 64, 128}
        assert not d[0].any() and not d[-1].any() and not d[:, 0].any() and not d[:, -1].any()
        off = {32: (-1, -1), 64: (-1, 0), 128: (-1, 1), 16: (0, -1), 1: (0, 1),
               8: (1, -1), 4: (1, 0), 2: (1, 1)}
        ys, xs = np.nonzero(d)
        for y, x in list(zip(ys, xs))[:2000]:
            dy, dx = off[int(d[y, x])]
            assert z[y + dy, x + dx] < z[y, x]              # points strictly downhill


def test_d8_tie_rule_and_diagonal_weight():
    z = np.full((3, 3), 5, np.float32)
    z[1, 1] = 6
    assert c_oracle.d8(z)[1, 1] == 64          # diagonals are scaled by 0.7071: first cardinal (N) wins
    z = np.full((3, 3), 6, np.float32); z[1, 1] = 7; z[0, 1] = 5; z[1, 0] = 5
    assert c_oracle.d8(z)[1, 1] == 64          # N and W tie at drop 2 -> N first in window order
    z = np.full((3, 3), 11, np.float32); z[1, 1] = 10; z[0, 0] = 8.6; z[1, 2] = 9
    # NW drop 1.4*0.7071 = 0.99 < E drop 1.0
    assert c_oracle.d8(z)[1, 1] == 1


def test_d8_after_fill_has_no_interior_pits():
    z = oracle.synth_dem(120, 130)
    w = c_oracle.sinkfill_pflood(z, eps=1e-3)
    d = c_oracle.d8(w)
    assert (d[1:-1, 1:-1] != 0).all()          # epsilon fill drains everywhere


def test_synth_dem_is_partition_independent():
    full = oracle.synth_dem(2500, 300)
    a = oracle.synth_dem(2500, 300, row0=0, rows=1100)
    b = oracle.synth_dem(2500, 300, row0=1099, rows=1401)
    assert np.array_equal(full[:1100], a) and np.array_equal(full[1099:], b)
