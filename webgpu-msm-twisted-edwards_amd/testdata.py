"""ZPrize test-vector ingestion: the host-side twin of `loadTestCase`
(test-data/testCases.ts:35-52).  The official input files are NOT in the reference tree
(README.md:22-33); if they are supplied, this turns them into the wire format compute_msm takes.

  points file : one JSON object per line  {"x": "...", "y": "...", "t": "...", "z": "..."}  decimal strings
  scalars file: one decimal integer per line
"""
from __future__ import annotations

import json

# expected affine results, test-data/testCases.ts:11-32 (kept as data in tests/golden/reference_kats.json too)
EXPECTED_POWERS = (16, 17, 18, 19, 20)


def parse_points_text(text: str) -> bytes:
    out = bytearray()
    for line in text.strip().split("\n"):
        if not line.strip():
            continue
        o = json.loads(line)
        if int(o.get("z", "1")) != 1:
            raise ValueError("test-vector points must be affine (z = 1)")
        out += int(o["x"]).to_bytes(32, "little") + int(o["y"]).to_bytes(32, "little")
    return bytes(out)


def parse_scalars_text(text: str) -> bytes:
    return b"".join(int(line).to_bytes(32, "little") for line in text.strip().split("\n") if line.strip())


def load_test_case(points_path: str, scalars_path: str):
    """Returns (bufferPoints, bufferScalars) in compute_msm's wire format."""
    with open(points_path) as f:
        pts = parse_points_text(f.read())
    with open(scalars_path) as f:
        sc = parse_scalars_text(f.read())
    if len(pts) // 64 != len(sc) // 32:
        raise ValueError("points / scalars count mismatch")
    return pts, sc
