"""ctypes wrapper around oracle/libte_oracle.so (the C restatement of the reference pipeline).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

from . import model

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libte_oracle.so")
    src = os.path.join(_HERE, "te_oracle.c")
    so2, src2 = os.path.join(_HERE, "libbls377_oracle.so"), os.path.join(_HERE, "bls377_oracle.c")
    stale2 = not os.path.exists(so2) or os.path.getmtime(so2) < os.path.getmtime(src2)
    if force or stale2 or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        u8p, u64, c_int = ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int
        L.ora_msm.argtypes = [u8p, u8p, u64, c_int, c_int, c_int, u8p]
        L.ora_msm.restype = c_int
        L.ora_msm_naive.argtypes = [u8p, u8p, u64, u8p]
        L.ora_msm_naive.restype = c_int
        L.ora_point_add_affine.argtypes = [u8p, u8p, u8p]
        L.ora_point_double_affine.argtypes = [u8p, u8p]
        L.ora_scalar_mul_affine.argtypes = [u8p, u8p, u8p]
        L.ora_field_op.argtypes = [c_int, u8p, u8p, u8p]
        L.ora_on_curve.argtypes = [u8p]
        L.ora_on_curve.restype = c_int
        L.ora_gen_scalars.argtypes = [u64, u64, u8p]
        L.ora_set_generator.argtypes = [u8p]
        L.ora_gen_points.argtypes = [u64, u64, u8p]
        L.ora_gen_points.restype = c_int
        L.ora_gen_points_random.argtypes = [u64, u64, u8p, c_int]
        L.ora_gen_points_random.restype = c_int
        L.ora_decompose_scalars_signed.argtypes = [u8p, u64, c_int, c_int, ctypes.c_void_p]
        L.ora_decompose_scalars_signed.restype = c_int
        L.ora_transpose.argtypes = [ctypes.c_void_p, u64, ctypes.c_uint32, c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ora_set_generator(model.le32(model.GX) + model.le32(model.GY))
        _LIB = L
    return _LIB


def reference_window_bits(n: int) -> int:
    """submission.ts:80: chunk_size = input_size >= 65536 ? 16 : 4"""
    return 16 if n >= 65536 else 4


def msm(points: bytes, scalars: bytes, c: int | None = None, bpr_mode: int = 1, threads: int = 1) -> bytes:
    n = len(scalars) // 32
    assert len(points) == 64 * n and len(scalars) == 32 * n
    if c is None:
        c = reference_window_bits(n)
    out = ctypes.create_string_buffer(64)
    rc = lib().ora_msm(points, scalars, n, c, bpr_mode, threads, out)
    if rc == -1:
        raise ValueError("final carry is 1")
    if rc:
        raise RuntimeError(f"ora_msm failed: {rc}")
    return out.raw


def msm_naive(points: bytes, scalars: bytes) -> bytes:
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(64)
    lib().ora_msm_naive(points, scalars, n, out)
    return out.raw


def point_add(a: bytes, b: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    lib().ora_point_add_affine(a, b, out)
    return out.raw


def point_double(a: bytes) -> bytes:
    out = ctypes.create_string_buffer(64)
    lib().ora_point_double_affine(a, out)
    return out.raw


def scalar_mul(a: bytes, k: int) -> bytes:
    out = ctypes.create_string_buffer(64)
    lib().ora_scalar_mul_affine(a, model.le32(k), out)
    return out.raw


def field_op(op: str, a: int, b: int = 0) -> int:
    out = ctypes.create_string_buffer(32)
    lib().ora_field_op({"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], model.le32(a), model.le32(b), out)
    return int.from_bytes(out.raw, "little")


def on_curve(a: bytes) -> bool:
    return bool(lib().ora_on_curve(a))


def gen_scalars(seed: int, n: int) -> bytes:
    out = ctypes.create_string_buffer(32 * n) if n else ctypes.create_string_buffer(1)
    lib().ora_gen_scalars(seed, n, out)
    return out.raw[: 32 * n]


def gen_points(seed: int, n: int) -> bytes:
    out = ctypes.create_string_buffer(64 * n) if n else ctypes.create_string_buffer(1)
    rc = lib().ora_gen_points(seed, n, out)
    if rc:
        raise RuntimeError(f"ora_gen_points failed: {rc}")
    return out.raw[: 64 * n]


def gen_points_random(seed: int, n: int, threads: int = 8) -> bytes:
    """SURVEY 8d set (R): n independent points a_i * G, a_i seeded-random (plain double-and-add per point: ~0.1 ms each)."""
    out = ctypes.create_string_buffer(64 * n) if n else ctypes.create_string_buffer(1)
    rc = lib().ora_gen_points_random(seed, n, out, threads)
    if rc:
        raise RuntimeError(f"ora_gen_points_random failed: {rc}")
    return out.raw[: 64 * n]


def gen_points_fixed(n: int) -> bytes:
    """'harness mode': the UI's random mode replicates ONE point n times (ui/AllBenchmarks.tsx:105-112)."""
    return (model.le32(model.HX) + model.le32(model.HY)) * n


def decompose_scalars_signed(scalars: bytes, c: int):
    import numpy as np
    n = len(scalars) // 32
    nw = -(-256 // c)
    out = np.zeros((nw, n), dtype=np.uint32)
    rc = lib().ora_decompose_scalars_signed(scalars, n, c, nw, out.ctypes.data)
    if rc:
        raise ValueError("final carry is 1")
    return out


def transpose(chunks, c: int):
    import numpy as np
    nw, n = chunks.shape
    ncols = 1 << c
    col_ptr = np.zeros((nw, ncols + 1), dtype=np.uint32)
    val_idx = np.zeros((nw, n), dtype=np.uint32)
    ch = np.ascontiguousarray(chunks, dtype=np.uint32)
    lib().ora_transpose(ch.ctypes.data, n, ncols, nw, col_ptr.ctypes.data, val_idx.ctypes.data)
    return col_ptr, val_idx
