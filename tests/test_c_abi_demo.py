"""
The boundary from plain C (`examples/c_abi_demo.c`): compiles against nothing but
`include/hydrodem_hip.h`, links against nothing but `libhydrodem_hip.so` (no Python, no
torch in its dependencies), and -- on the GPU -- produces the same bytes as the same calls
made through the Python binding, checked against the oracle.
"""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hydrodem_amd", "csrc")


@pytest.fixture(scope="module")
def demo(tmp_path_factory, built):
    exe = str(tmp_path_factory.mktemp("c_abi") / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c11", "-O2", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", CSRC,
                           "-lhydrodem_hip", f"-Wl,-rpath,{CSRC}", "-lm", "-o", exe])
    return exe


def fnv1a(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).view(np.uint8).ravel().tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_the_c_demo_needs_only_the_header_and_the_library(demo):
    deps = subprocess.run(["ldd", demo], capture_output=True, text=True, check=True).stdout
    assert "libhydrodem_hip.so" in deps
    assert not re.search(r"python|torch|numpy", deps, re.I), deps
    header = open(os.path.join(ROOT, "include", "hydrodem_hip.h")).read()
    # the header itself pulls in nothing but the two freestanding C headers
    assert sorted(re.findall(r"^#include\s+[<\"]([^>\"]+)[>\"]", header, re.M)) == ["stddef.h", "stdint.h"]


@pytest.mark.gpu
def test_the_c_demo_and_the_python_binding_produce_the_same_bytes(demo, tmp_path):
    from hydrodem_amd import backend
    from oracle import c_oracle
    raw = tmp_path / "z.f32"
    out = subprocess.run([demo, "140", "203", str(raw)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    got = dict(re.findall(r"(\w+)=([0-9a-f]+)", out.stdout))
    z = np.fromfile(raw, dtype=np.float32).reshape(140, 203)
    w = backend.sinkfill(z)
    codes = backend.d8(w)
    mean = backend.boxmean3(w, do_round=True)
    assert int(got["fill"], 16) == fnv1a(w) and int(got["d8"], 16) == fnv1a(codes)
    assert int(got["mean"], 16) == fnv1a(mean) and got["bad"] == "0" and got["converged"] == "1"
    want = c_oracle.sinkfill_pflood(z)
    assert np.array_equal(w, want, equal_nan=True) and np.array_equal(codes, c_oracle.d8(want))
