import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    m = importlib.import_module("webgpu-msm-twisted-edwards_amd")
    if not os.path.exists(m.library_path()):
        m.build_library()
    return m


@pytest.fixture(scope="session")
def ora():
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def model():
    from oracle import model
    return model


@pytest.fixture(scope="session")
def fpcheck():
    """Host build of the product's device arithmetic headers (tests/csrc/fpcheck.cpp)."""
    import ctypes
    d = os.path.join(ROOT, "tests", "csrc")
    so, src = os.path.join(d, "libfpcheck.so"), os.path.join(d, "fpcheck.cpp")
    hdr_dir = os.path.join(ROOT, "webgpu-msm-twisted-edwards_amd", "csrc")
    deps = [src] + [os.path.join(hdr_dir, f) for f in ("fp.hpp", "fq377.hpp", "field.hpp", "curve.hpp", "fp_constants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    L = ctypes.CDLL(so)
    L.fpc_partial_rows.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p]
    L.fpc_partial_rows.restype = ctypes.c_int
    return L


@pytest.fixture(scope="session")
def fq377check():
    """Host build of the BLS12-377 device arithmetic with column-overflow checking (tests/csrc/fq377check.cpp)."""
    import ctypes
    d = os.path.join(ROOT, "tests", "csrc")
    so, src = os.path.join(d, "libfq377check.so"), os.path.join(d, "fq377check.cpp")
    hdr_dir = os.path.join(ROOT, "webgpu-msm-twisted-edwards_amd", "csrc")
    deps = [src] + [os.path.join(hdr_dir, f) for f in ("fp.hpp", "fq377.hpp", "field.hpp", "curve.hpp", "fq377_constants.inc")]
    if not os.path.exists(so) or any(os.path.getmtime(x) > os.path.getmtime(so) for x in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return ctypes.CDLL(so)


@pytest.fixture(scope="session")
def kats():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))


@pytest.fixture(scope="session")
def wasm_golden():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "msm_wasm_golden.json")))
