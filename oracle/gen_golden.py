"""Generates tests/golden/msm_wasm_golden.json by running the reference's CPU MSM (Aleo WASM
Address.msm, reference/reference.ts:29-39) on deterministic synthetic inputs.  BUILD CONTAINER ONLY
(needs /root/reference and node); the fixture it writes holds data only: {name, seed, n, mode, x, y}.

Inputs are regenerated in tests from (seed, n, mode) by oracle.gen_points / gen_scalars /
edge_scalars, so no input data is stored.  y is recovered from the WASM's x exactly as the reference
does (FieldMath.getPointFromX, reference/utils/FieldMath.ts:31-55).

usage: python -m oracle.gen_golden [--max-n 65536] [--only chain_n1048576,fixed_n262144]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import tempfile

from . import model, oracle

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "msm_wasm_golden.json")

CASES = [  # (name, seed, n, mode)
    ("chain_n1", 0x5EED0001, 1, "chain"),
    ("chain_n2", 0x5EED0002, 2, "chain"),
    ("chain_n3", 0x5EED0003, 3, "chain"),
    ("chain_n16", 0x5EED0010, 16, "chain"),
    ("chain_n255", 0x5EED00FF, 255, "chain"),
    ("chain_n256", 0x5EED0100, 256, "chain"),
    ("chain_n1000", 0x5EED03E8, 1000, "chain"),
    ("chain_n4096", 0x5EED1000, 4096, "chain"),
    ("fixed_n256", 0x5EED0101, 256, "fixed"),      # harness mode: one point replicated
    ("edge_n64", 0x5EED0040, 64, "edge"),          # scalars 0, 1, p-1, l, l-1, 2^k boundaries ...
    ("chain_n65536", 0x5EED0010000, 65536, "chain"),
    ("fixed_n65536", 0x5EED0010001, 65536, "fixed"),
    # round 4: the headline size itself and the harness mode at 2^18 (about 35 + 5 minutes of WASM; run with
    # --only <names>, which keeps every record already in the fixture)
    ("chain_n1048576", 0x5EED0100000, 1 << 20, "chain"),
    ("fixed_n262144", 0x5EED0040001, 1 << 18, "fixed"),
    ("random_n1000", 0x5EED03E9, 1000, "random"),              # SURVEY 8d set (R): independent seeded-random a_i * G
    ("random_n65536", 0x5EED0010002, 65536, "random"),
    # every harness size 2^16 .. 2^20 (full_benchmarks.ts:13-15) has a reference-generated point, and the harness mode at the headline size
    ("chain_n131072", 0x5EED0020000, 1 << 17, "chain"),
    ("random_n262144", 0x5EED0040002, 1 << 18, "random"),
    ("chain_n524288", 0x5EED0080000, 1 << 19, "chain"),
    ("fixed_n1048576", 0x5EED0100001, 1 << 20, "fixed"),
    # SURVEY 8f rank 3, skewed scalars as a prover feeds them: a quarter zeros, a quarter ones, a quarter small values, the rest uniform
    ("witness_n4096", 0x5EED1001, 4096, "witness"),
    ("witness_n65536", 0x5EED0010003, 65536, "witness"),
    # round 5: edge scalars (0, 1, p - 1, l, l - 1, 2^k boundaries, then uniform ones) at a size the engine runs with its 15-bit plan --
    # 17 windows since round 5 (scalars below p < 2^253 need no 18th)
    ("edge_n4096", 0x5EED1002, 4096, "edge"),
    ("edge_n65536", 0x5EED0010004, 65536, "edge"),
]


def make_inputs(seed: int, n: int, mode: str):
    pts = oracle.gen_points_fixed(n) if mode == "fixed" else oracle.gen_points_random(seed, n) if mode == "random" else oracle.gen_points(seed, n)
    sc = oracle.gen_scalars(seed, n)
    if mode == "edge":
        sc = model.scalars_to_bytes(edge_scalars(seed, n))
    if mode == "witness":
        a = bytearray(sc)
        for i in range(n):
            r = i & 3
            if r < 3:
                v = 0 if r == 0 else 1 if r == 1 else (i * 2654435761 + seed) % (1 << 20)
                a[32 * i:32 * i + 32] = v.to_bytes(32, "little")
        sc = bytes(a)
    return pts, sc


def edge_scalars(seed: int, n: int):
    p, l = model.P, model.L
    special = [0, 1, p - 1, l, l - 1, l + 1, 2, (1 << 15), (1 << 15) - 1, (1 << 16) - 1, (1 << 16),
               0x8000800080008000800080008000800080008000800080008000800080008000 % p,
               0x7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF7FFF % p,
               (1 << 252), (1 << 253) % p, p - 2, p // 2, 0xFFFF0000FFFF0000FFFF, 3 * l, 4 * l - 1]
    rnd = model.gen_scalars(seed, n)
    return [special[i] if i < len(special) else rnd[i] for i in range(n)]


def special_point_inputs(seed: int, n: int):
    """Inputs that exercise the exceptional cases of the group law (the reference's add-2008-hwcd with a = -1, d = 3021 a
    non-square, is complete -- wgsl/curve/ec.template.wgsl:34-66 -- so every one of them has a defined result): the neutral
    element (0, 1), the point of order two (0, -1), the two points of order four (+-sqrt(-1), 0), P and -P and P again under the
    SAME scalar (one bucket receives P, -P, P: a sum that passes through the neutral element and a doubling), and subgroup
    points shifted by low-order points.  Returns (points: list of (x, y), scalars: list of int); n >= 32."""
    p = model.P
    i4 = model.sqrt_mod_p(p - 1)
    low = [(0, 1), (0, p - 1), (i4, 0), (p - i4, 0)]
    g = (model.GX, model.GY)
    rnd = model.gen_scalars(seed, n)
    pts, ks = [], []
    for j, a in enumerate((1, 2, 3, 5, 7)):
        b = model.scalar_mul(a, g)
        pts += [b, model.neg(b), b]
        ks += [rnd[j]] * 3
    for j, q in enumerate(low):
        pts.append(q); ks.append(rnd[8 + j])
        pts.append(model.add(model.scalar_mul(11 + j, g), q)); ks.append(rnd[12 + j])
    pts += [low[0], low[1]]; ks += [0, p - 1]
    base = model.gen_points(seed, n)
    while len(pts) < n:
        k = len(pts)
        pts.append(base[k] if k % 5 else model.add(base[k], low[k % 4]))
        ks.append(rnd[k])
    return pts[:n], ks[:n]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-n", type=int, default=65536)
    ap.add_argument("--only", default="", help="comma-separated case names: run these (any size) and merge them "
                                               "into the existing fixture instead of rewriting it")
    args = ap.parse_args()
    only = [s for s in args.only.split(",") if s]
    cases, meta = [], []
    for name, seed, n, mode in CASES:
        if (name not in only) if only else (n > args.max_n):
            continue
        pts, sc = make_inputs(seed, n, mode)
        xs = [str(int.from_bytes(pts[64 * i:64 * i + 32], "little")) for i in range(n)]
        ks = [str(int.from_bytes(sc[32 * i:32 * i + 32], "little")) for i in range(n)]
        cases.append({"name": name, "xs": xs, "ks": ks})
        meta.append((name, seed, n, mode))
    with tempfile.TemporaryDirectory() as td:
        cin, cout = os.path.join(td, "cases.json"), os.path.join(td, "out.json")
        json.dump(cases, open(cin, "w"))
        subprocess.check_call(["node", os.path.join(HERE, "wasm_msm.js"), cin, cout])
        res = {r["name"]: r for r in json.load(open(cout))}
    golden = []
    for name, seed, n, mode in meta:
        x = int(res[name]["x"])
        if x == 0:
            y = 1
        else:
            _, y = model.point_from_x(x)
        golden.append({"name": name, "seed": seed, "n": n, "mode": mode, "x": str(x), "y": str(y),
                       "source": "aleo-wasm Address.msm (reference/reference.ts:29-39), node " +
                                 subprocess.check_output(["node", "--version"]).decode().strip(),
                       "wasm_ms": res[name]["ms"]})
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if only and os.path.exists(OUT):
        new = {g["name"]: g for g in golden}
        golden = [g for g in json.load(open(OUT)) if g["name"] not in new] + golden
        rank = {c[0]: i for i, c in enumerate(CASES)}
        golden.sort(key=lambda g: rank.get(g["name"], len(rank)))
    json.dump(golden, open(OUT, "w"), indent=1)
    print("wrote", OUT, len(golden), "cases")


if __name__ == "__main__":
    main()
