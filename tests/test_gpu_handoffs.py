"""The in-launch hand-offs of the engine ("the block that arrives last does the next step": kernels.hip.hpp, handoff_arrive)
against BOTH builds of the library: libtemsm.so, whose arrival is bracketed by agent-scope release / acquire fences -- the memory
model's own form, the default since round 5 --, and libtemsm_relaxed.so (-DTE_HANDOFF_RELAXED), whose arrival is a relaxed
device-scope atomic behind s_waitcnt: outside the HIP / LLVM memory model, correct on this hardware by test and soak (the
default of rounds 3-4, kept as the A/B twin).  Each build runs in a process of
its own (TE_MSM_LIB names the library; one HIP library per process) on inputs that force
  * multi-piece partitions (more than TE_L2_CAP = 9208 entries in one level-1 partition): k_l2_local counts the pieces into
    bucket_count with global atomics, part_ticket counts the arrivals, the last piece plans the partition;
  * giant buckets (more than 16 parts): k_seg_combine_all sums runs of 256 parts, bucket_cursor counts the arrivals, the last
    run's block adds the runs up (block_sum_points<COHERENT>).
Both must return the oracle's point, bit for bit; the stage times of both are printed (profiles/r05_handoff_fenced_twin.txt, taken when the fenced form still was the twin).
Reference: the reference has no such step -- its transpose is one thread per window (wgsl/cuzk/transpose.wgsl:32-76)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "webgpu-msm-twisted-edwards_amd")

_CHILD = r"""
import importlib, json, os, random, sys, time
sys.path.insert(0, {root!r})
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle import oracle as ora, model          # input generators only; the parent compares with the oracle's results
assert os.path.samefile(pkg.library_path(), os.environ["TE_MSM_LIB"])
out = {{"lib": os.path.basename(pkg.library_path()), "cases": {{}}, "stage_us": {{}}}}
def inputs(name):
    if name == "all_equal":                       # ONE bucket per window holds every point: 200000 / 64 = 3125 parts, 13 runs; one partition holds all entries
        n = 200000
        return ora.gen_points(71, n), model.scalars_to_bytes([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % model.P] * n), {{}}
    if name == "one_bit_top_window":              # 253-bit scalars, 12-bit windows: the top window has ONE occupied bucket (n / 2 entries) in one partition
        n = 200000
        return ora.gen_points(71, n), ora.gen_scalars(72, n), {{"window_bits": 12}}
    if name == "many_parts":                      # segment length 1, 64 buckets per window: thousands of parts per bucket, more runs than buckets
        n = 158508
        return ora.gen_points(5, n), ora.gen_scalars(5, n), {{"window_bits": 7, "segment_len": 1}}
    if name == "witness":                         # zeros, ones, small values among uniform scalars: bucket 0 of window 0 is giant
        n = 300000
        rnd = random.Random(31)
        ks = model.gen_scalars(n, n)
        for i in range(n):
            q = rnd.random()
            if q < 0.5:
                ks[i] = 0 if q < 0.2 else 1 if q < 0.4 else rnd.randrange(1 << 20)
        return ora.gen_points(n, n), model.scalars_to_bytes(ks), {{}}
    if name == "general_entries_equal":           # the same skew through the general (key + index) level-1 entries
        n = 200000
        return ora.gen_points(71, n), model.scalars_to_bytes([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % model.P] * n), {{"packed_sort": 0}}
    if name == "uniform_2_20":                    # the headline shape: only the top window has multi-piece partitions (~115 pieces)
        n = 1 << 20
        p, s = pkg.synth_inputs(0x5EED0000 + 20, n)
        return p, s, {{"window_bits": 16}}
    raise KeyError(name)
with pkg.MsmContext((0,)) as c:
    for name in {names!r}:
        pts, sc, opts = inputs(name)
        for k, v in opts.items():
            c.set_option(k, v)
        res = [c.run(pts, sc).hex()]
        ts = [c.submit(pts, sc) for _ in range(3)]            # several in flight: the hand-offs of different MSMs side by side
        res += [c.collect(t).hex() for t in ts]
        c.set_option("profile", 2)
        acc = {{}}
        for _ in range(5):
            assert c.run(pts, sc).hex() == res[0]
            for k, v in c.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v * 1e3 / 5
        c.set_option("profile", 0)
        out["cases"][name] = sorted(set(res))
        out["stage_us"][name] = {{k: round(v, 1) for k, v in acc.items() if not k.endswith("ghz")}}
        for k in opts:
            c.set_option(k, 1 if k == "packed_sort" else 0)
    # a slice of the differential soak: random sizes, window bits and skews
    rnd = random.Random(20251005)
    soak = []
    for it in range(24):
        n = int(2 ** rnd.uniform(4, 17.5))
        c.set_option("window_bits", rnd.choice([0, 7, 10, 13, 16]))
        c.set_option("segment_len", rnd.choice([0, 0, 1, 7]))
        pts, sc = ora.gen_points(3000 + it, n), ora.gen_scalars(3000 + it, n)
        if rnd.random() < 0.5:
            sc = sc[:32] * n
        soak.append(c.run(pts, sc).hex())
    out["soak"] = soak
print(json.dumps(out))
"""

CASES = ["all_equal", "one_bit_top_window", "many_parts", "witness", "general_entries_equal", "uniform_2_20"]


def _expected(ora, model):
    """the oracle's answers for the child's inputs (same generators, same seeds)"""
    import random
    exp = {}
    n = 200000
    pts = ora.gen_points(71, n)
    same = model.scalars_to_bytes([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % model.P] * n)
    exp["all_equal"] = exp["general_entries_equal"] = ora.msm(pts, same, threads=16).hex()
    exp["one_bit_top_window"] = ora.msm(pts, ora.gen_scalars(72, n), threads=16).hex()
    n = 158508
    exp["many_parts"] = ora.msm(ora.gen_points(5, n), ora.gen_scalars(5, n), threads=16).hex()
    n = 300000
    rnd = random.Random(31)
    ks = model.gen_scalars(n, n)
    for i in range(n):
        q = rnd.random()
        if q < 0.5:
            ks[i] = 0 if q < 0.2 else 1 if q < 0.4 else rnd.randrange(1 << 20)
    exp["witness"] = ora.msm(ora.gen_points(n, n), model.scalars_to_bytes(ks), threads=16).hex()
    rnd = random.Random(20251005)
    soak = []
    for it in range(24):
        n = int(2 ** rnd.uniform(4, 17.5))
        rnd.choice([0, 7, 10, 13, 16]); rnd.choice([0, 0, 1, 7])
        pts, sc = ora.gen_points(3000 + it, n), ora.gen_scalars(3000 + it, n)
        if rnd.random() < 0.5:
            sc = sc[:32] * n
        soak.append(ora.msm(pts, sc, threads=16).hex())
    return exp, soak


def test_handoffs_under_both_builds(pkg, ora, model, wasm_golden, tmp_path):
    relaxed = os.path.join(PKG, "libtemsm_relaxed.so")
    if not os.path.exists(relaxed):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-s", "relaxed"])
    exp, soak = _expected(ora, model)
    p20, s20 = pkg.synth_inputs(0x5EED0000 + 20, 1 << 20)
    exp["uniform_2_20"] = ora.msm(p20, s20, c=16, threads=16).hex()
    script = tmp_path / "child.py"
    script.write_text(_CHILD.format(root=ROOT, names=CASES))
    outs = {}
    for lib in (os.path.join(PKG, "libtemsm.so"), relaxed):
        r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, TE_MSM_LIB=lib), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        o = json.loads(r.stdout.decode().strip().splitlines()[-1])
        outs[o["lib"]] = o
        for name in CASES:
            assert o["cases"][name] == [exp[name]], (o["lib"], name)
        assert o["soak"] == soak, o["lib"]
    a, b = outs["libtemsm.so"], outs["libtemsm_relaxed.so"]
    print("\nhand-offs, stage times in us (mean of 5 MSMs alone on the GPU): default build (fenced) | relaxed twin")
    for name in CASES:
        sa, sb = a["stage_us"][name], b["stage_us"][name]
        print("  %-24s bucket_sort %7.1f | %7.1f   marginal_sums(+combine) %7.1f | %7.1f   all stages %8.1f | %8.1f" % (
            name, sa.get("bucket_sort", 0), sb.get("bucket_sort", 0), sa.get("marginal_sums", 0), sb.get("marginal_sums", 0),
            sum(v for k, v in sa.items() if k != "accumulate_on_device"), sum(v for k, v in sb.items() if k != "accumulate_on_device")))
