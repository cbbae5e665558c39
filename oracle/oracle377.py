"""ctypes wrapper around oracle/libbls377_oracle.so (BLS12-377 G1, BASELINE config 5).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Parity unpinned
by the reference (see oracle/bls377_oracle.c); checked against oracle/model377.py in tests/test_oracle_bls377.py.
"""
from __future__ import annotations

import ctypes
import os

from . import model377
from .oracle import build as _build_all

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
POINT_BYTES, SCALAR_BYTES = 96, 48


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        _build_all()
        L = ctypes.CDLL(os.path.join(_HERE, "libbls377_oracle.so"))
        u8p, u64, ci = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int
        L.ora377_msm.argtypes = [u8p, u8p, u64, ci, ci, u8p]
        L.ora377_msm.restype = ci
        L.ora377_msm_naive.argtypes = [u8p, u8p, u64, u8p]
        L.ora377_msm_naive.restype = ci
        L.ora377_add_affine.argtypes = [u8p, u8p, u8p]
        L.ora377_scalar_mul_affine.argtypes = [u8p, u8p, u8p]
        L.ora377_on_curve.argtypes = [u8p]
        L.ora377_on_curve.restype = ci
        L.ora377_gen_scalars.argtypes = [u64, u64, u8p]
        L.ora377_gen_points.argtypes = [u64, u64, u8p, u8p]
        L.ora377_gen_points.restype = ci
        _LIB = L
    return _LIB


def msm(points: bytes, scalars: bytes, c: int = 16, threads: int = 1) -> bytes:
    n = len(scalars) // SCALAR_BYTES
    assert len(points) == POINT_BYTES * n
    out = ctypes.create_string_buffer(96)
    rc = lib().ora377_msm(points, scalars, n, c, threads, out)
    if rc:
        raise ValueError("final carry is 1" if rc == -3 else f"ora377_msm failed: {rc}")
    return out.raw


def msm_naive(points: bytes, scalars: bytes) -> bytes:
    n = len(scalars) // SCALAR_BYTES
    out = ctypes.create_string_buffer(96)
    lib().ora377_msm_naive(points, scalars, n, out)
    return out.raw


def point_add(a: bytes, b: bytes) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().ora377_add_affine(a, b, out)
    return out.raw


def scalar_mul(a: bytes, k: int) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().ora377_scalar_mul_affine(a, model377.le48(k), out)
    return out.raw


def on_curve(a: bytes) -> bool:
    return bool(lib().ora377_on_curve(a))


def gen_scalars(seed: int, n: int) -> bytes:
    out = ctypes.create_string_buffer(48 * n)
    lib().ora377_gen_scalars(seed, n, out)
    return out.raw


def gen_points(seed: int, n: int) -> bytes:
    out = ctypes.create_string_buffer(96 * n)
    rc = lib().ora377_gen_points(seed, n, model377.le48(model377.GX) + model377.le48(model377.GY), out)
    assert rc == 0
    return out.raw
