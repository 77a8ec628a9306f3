"""ctypes binding of ``liboracle_c.so`` (the C half of the CPU oracle).
TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile ``hdem_oracle.c`` with gcc (a second or two)."""
    so = os.path.join(_HERE, "liboracle_c.so")
    src = os.path.join(_HERE, "hdem_oracle.c")
    if force or not os.path.exists(so) or \
            os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B",
                               "liboracle_c.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        ci = ctypes.c_int
        _LIB.oracle_sinkfill_pflood_f32.argtypes = [f32p, ci, ci,
                                                    ctypes.c_float, f32p]
        _LIB.oracle_sinkfill_pflood_f32.restype = ci
        _LIB.oracle_d8_f32.argtypes = [f32p, ci, ci, u8p]
        _LIB.oracle_d8_f32.restype = None
        _LIB.oracle_boxmean3_f32.argtypes = [f32p, ci, ci, f32p, ci]
        _LIB.oracle_boxmean3_f32.restype = None
        _LIB.oracle_boxmean3_f64.argtypes = [f64p, ci, ci, f64p, ci]
        _LIB.oracle_boxmean3_f64.restype = None
        _LIB.oracle_quadratic_ref_f32.argtypes = [f32p, ci, ci, ci, f32p]
        _LIB.oracle_quadratic_ref_f32.restype = ci
        _LIB.oracle_quadratic_ref_f64.argtypes = [f64p, ci, ci, ci, f64p]
        _LIB.oracle_quadratic_ref_f64.restype = ci
        _LIB.oracle_groves_ref.argtypes = [f32p, u8p, ci, ci, ci,
                                           ctypes.c_double, ci, f64p, u8p]
        _LIB.oracle_groves_ref.restype = ci
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def sinkfill_pflood(z, eps=0.0):
    z = np.ascontiguousarray(z, dtype=np.float32)
    w = np.empty_like(z)
    rc = lib().oracle_sinkfill_pflood_f32(_p(z, ctypes.c_float), z.shape[0],
                                          z.shape[1], float(eps),
                                          _p(w, ctypes.c_float))
    if rc:
        raise MemoryError("oracle_sinkfill_pflood_f32")
    return w


def d8(z):
    z = np.ascontiguousarray(z, dtype=np.float32)
    out = np.empty(z.shape, dtype=np.uint8)
    lib().oracle_d8_f32(_p(z, ctypes.c_float), z.shape[0], z.shape[1],
                        _p(out, ctypes.c_uint8))
    return out


def boxmean3(x, do_round=True):
    x = np.ascontiguousarray(x)
    if x.dtype == np.float32:
        out = np.empty_like(x)
        lib().oracle_boxmean3_f32(_p(x, ctypes.c_float), x.shape[0],
                                  x.shape[1], _p(out, ctypes.c_float),
                                  int(do_round))
        return out
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().oracle_boxmean3_f64(_p(x, ctypes.c_double), x.shape[0], x.shape[1],
                              _p(out, ctypes.c_double), int(do_round))
    return out


def quadratic_ref(dem, ws=15):
    """Bit-faithful QuadraticFilter: float32 in -> float32 out, anything
    else -> float64 out (as the reference's ``dem.copy()`` would be)."""
    dem = np.ascontiguousarray(dem)
    if dem.dtype == np.float32:
        out = np.empty_like(dem)
        rc = lib().oracle_quadratic_ref_f32(_p(dem, ctypes.c_float),
                                            dem.shape[0], dem.shape[1], ws,
                                            _p(out, ctypes.c_float))
    else:
        dem = np.ascontiguousarray(dem, dtype=np.float64)
        out = np.empty_like(dem)
        rc = lib().oracle_quadratic_ref_f64(_p(dem, ctypes.c_double),
                                            dem.shape[0], dem.shape[1], ws,
                                            _p(out, ctypes.c_double))
    if rc:
        raise MemoryError("oracle_quadratic_ref")
    return out


def groves_ref(img, groves, iterations=3, ws=15, thr=1.5, masks=False):
    img = np.ascontiguousarray(img, dtype=np.float32)
    g = np.ascontiguousarray(np.asarray(groves) != 0, dtype=np.uint8)
    out = np.empty(img.shape, dtype=np.float64)
    m = np.empty((max(iterations, 1),) + img.shape, dtype=np.uint8)
    rc = lib().oracle_groves_ref(_p(img, ctypes.c_float),
                                 _p(g, ctypes.c_uint8), img.shape[0],
                                 img.shape[1], ws, float(thr), iterations,
                                 _p(out, ctypes.c_double),
                                 _p(m, ctypes.c_uint8))
    if rc:
        raise MemoryError("oracle_groves_ref")
    return (out, m) if masks else out
