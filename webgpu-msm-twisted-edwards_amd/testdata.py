"""ZPrize test-vector ingestion: the host-side twin of `loadTestCase`
(test-data/testCases.ts:35-52).  The official input files are NOT in the reference tree
(README.md:22-33); if they are supplied, this turns them into the wire format compute_msm takes.

  points file : one JSON object per line  {"x": "...", "y": "...", "t": "...", "z": "..."}  decimal strings
  scalars file: one decimal integer per line
"""
from __future__ import annotations

import json

EXPECTED_POWERS = (16, 17, 18, 19, 20)
# expected affine results of the five official cases, test-data/testCases.ts:11-32 (data, decimal)
EXPECTED = {
    16: {"x": 4490298471131273381350715833932091894064554978284853693957586604825823442429,
         "y": 207233051598812890797414182362695316831408959017076683749810755208551572458},
    17: {"x": 405755281347735151880827575059343698498813029460786026451708154294960743560,
         "y": 7112985356832152643523650125935205310677117771129806490701829425450717492869},
    18: {"x": 4020134989704514076121556080357844499902614818105934254331815581426895427831,
         "y": 2694327822589008080344499645494473764166611881342421427746308662023437975766},
    19: {"x": 3856727778963570638772781884183843350150969534777451295534564482755471873113,
         "y": 1398750101296346671684024297455637342909036274728274942667983346895370713922},
    20: {"x": 5201851187583570844529445080011852189038251929148722905178398320328749074909,
         "y": 3586360219804356686204324370397321114669962278596135149389460948678051407803},
}


def expected_result(power: int):
    """{"x": int, "y": int} for 2^power of the official test data, or None"""
    return EXPECTED.get(power)


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
