"""GPU tests of whole-MSM TICKETS on contexts of several devices (te_msm_submit / te_msm_submit_async /
te_msm_submit_device + te_msm_ticket_wait / te_msm_collect on n_dev = D): one whole MSM per device, a ticket goes to the
device with the fewest in flight.  A one-GPU box names GPU 0 several times: D device records, D x TE_MSM_WORKSETS work
sets, D host threads on the one card -- the scheduling, the threads and every copy run; two PHYSICAL devices stay
unmeasured.  Reference side: the async call convention ui/Benchmark.tsx:32, multi-device as the
README's future work (README.md:551), the entry point submission.ts:73-78.  Nothing here reads /root/reference."""
import ctypes
import os

import pytest

from oracle.gen_golden import make_inputs

pytestmark = pytest.mark.gpu


def _dev(buf: bytes):
    import torch
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("ids", [(0, 0), (0, 0, 0, 0), (0,) * 8])
def test_tickets_on_a_multi_device_context(pkg, model, ora, wasm_golden, ids):
    """2 D tickets in flight (asynchronous and blocking submits mixed) against the oracle; every device gets two; collected
    out of order; a scalar-range error belongs to its ticket; te_msm_trim and a lone te_msm_run beside tickets; capacity"""
    D = len(ids)
    cases = [(1200 + i, [70001, 1000, 30000, 65536, 5, 40000, 12345, 257][i % 8]) for i in range(2 * D)]
    data = []
    for seed, n in cases:
        pts, sc = ora.gen_points(seed, n), ora.gen_scalars(seed, n)
        data.append((pts, sc, ora.msm(pts, sc, threads=8)))
    with pkg.MsmContext(ids) as c:
        assert c.get_option("num_devices") == D
        for signed in (1, 0):
            c.set_option("signed_digits", signed)
            ts = [(c.submit_async if i % 3 else c.submit)(data[i][0], data[i][1]) for i in range(2 * D)]
            assert c.get_option("in_flight") == 2 * D
            where = [c.ticket_device(t) for t in ts]
            assert sorted(i for i, _ in where) == sorted(list(range(D)) * 2), where       # the least loaded device, idle ones in turn
            assert all(dev == ids[i] for i, dev in where)
            order = list(reversed(range(2 * D)))
            order = order[1::2] + order[0::2]
            for i in order:
                if i % 2:
                    c.ticket_wait(ts[i])
                assert c.collect(ts[i]) == data[i][2], (signed, i)
            assert c.get_option("in_flight") == 0
            with pytest.raises(pkg.MsmError):
                c.collect(ts[0])                                                           # a ticket is consumed once
            with pytest.raises(pkg.MsmError):
                c.ticket_device(ts[0])
        c.set_option("signed_digits", 1)
        # an error in one ticket only; a lone call (point slices over all devices) and a trim beside tickets in flight
        n = 100003
        pts, sc = ora.gen_points(1300, n), ora.gen_scalars(1300, n)
        want = ora.msm(pts, sc, threads=8)
        bad = bytearray(sc); bad[32 * (n - 2):32 * (n - 2) + 32] = b"\xff" * 32
        c.set_option("window_bits", 16)                                                    # 16 x 16 bits: 2^256 - 1 leaves a final carry
        t = [c.submit_async(pts, sc), c.submit_async(pts, bytes(bad)), c.submit(pts, sc), c.submit_async(pts, sc)]
        assert c.run(pts, sc) == want                                                      # te_msm_run on the sets no ticket owns
        full = c.get_option("device_bytes")
        assert c.trim(0) >= 0 and c.get_option("device_bytes") <= full                     # owned sets are skipped, idle ones freed
        assert c.collect(t[3]) == want
        with pytest.raises(pkg.MsmError) as e:
            c.collect(t[1])
        assert e.value.code == -3
        assert c.collect(t[0]) == want and c.collect(t[2]) == want
        c.set_option("window_bits", 0)
        # capacity: D x WORKSETS tickets, then TE_MSM_ESTATE until one is collected
        small_p, small_s = data[1][0], data[1][1]
        ts = [c.submit_async(small_p, small_s) for _ in range(D * pkg.WORKSETS)]
        with pytest.raises(pkg.MsmError) as e:
            c.submit_async(small_p, small_s)
        assert e.value.code == -4
        with pytest.raises(pkg.MsmError):
            c.submit(small_p, small_s)
        with pytest.raises(pkg.MsmError):
            c.run(small_p, small_s)                                                        # no free work set for the slices either
        assert c.collect(ts.pop(3)) == data[1][2]
        ts.append(c.submit(small_p, small_s))                                              # takes the set that ticket left
        assert all(c.collect(t) == data[1][2] for t in ts)
        assert c.get_option("in_flight") == 0
        assert c.run(pts, sc) == want


@pytest.mark.parametrize("ids", [(0, 0), (0, 0, 0, 0), (0,) * 8])
def test_tickets_against_the_reference_outputs(pkg, model, wasm_golden, ids):
    """tickets on D devices against the points the reference's own CPU MSM (Aleo WASM) returned: every golden up to 2^16 in
    flight together, then the headline size n = 2^20 on every device at once"""
    D = len(ids)
    with pkg.MsmContext(ids) as c:
        small = [g for g in wasm_golden if g["n"] <= 65536]
        assert small
        while small:
            batch, small = small[:2 * D], small[2 * D:]
            ts = []
            for g in batch:
                pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
                ts.append(c.submit_async(pts, sc))
            for g, t in reversed(list(zip(batch, ts))):
                assert model.xy_from_bytes(c.collect(t)) == (int(g["x"]), int(g["y"])), g["name"]
        big = [g for g in wasm_golden if g["n"] == 1 << 20]
        assert big, "the reference-generated golden at the headline size is missing"
        g = big[0]
        pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
        ts = [c.submit_async(pts, sc) for _ in range(D)]
        assert sorted(c.ticket_device(t)[0] for t in ts) == list(range(D))                 # one whole MSM per device
        for t in ts:
            assert model.xy_from_bytes(c.collect(t)) == (int(g["x"]), int(g["y"])), D


def test_device_resident_tickets_on_a_multi_device_context(pkg, ora):
    """te_msm_submit_device on n_dev = 4: the inputs live on one device of the context, a ticket that lands on another device
    pulls them over its peer link first ("stage_device_inputs" forces the copy on a one-GPU box, where every device id names
    the holder); bytes and copies are counted; mixed with host-buffer tickets; single-device contexts keep their window shard"""
    import torch
    n = 50000
    pts, sc = ora.gen_points(1400, n), ora.gen_scalars(1400, n)
    want = ora.msm(pts, sc, threads=8)
    dp, ds = _dev(pts), _dev(sc)
    torch.cuda.synchronize()
    with pkg.MsmContext((0, 0, 0, 0)) as c:
        ts = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(4)]
        assert sorted(c.ticket_device(t)[0] for t in ts) == [0, 1, 2, 3]
        assert c.get_option("peer_copies") == 0                                            # every "device" is the holder: no copy
        assert all(c.collect(t) == want for t in ts)
        c.set_option("stage_device_inputs", 1)
        ts = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(6)] + [c.submit_async(pts, sc)]
        assert c.get_option("peer_copies") == 12 and c.get_option("peer_bytes") == 6 * n * 96
        for t in reversed(ts):
            assert c.collect(t) == want
        # the lone call on the same context: window shards, inputs as scatter + all-gather over the peer links
        before = c.get_option("peer_copies"), c.get_option("peer_bytes")
        assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == want
        assert c.get_option("peer_copies") - before[0] == 2 * 3 * 4                        # 2 (D - 1) D
        assert c.get_option("peer_bytes") - before[1] == 3 * n * 96                        # every other device ends up with all n points
        with pytest.raises(pkg.MsmError):
            c.submit_device(0, ds.data_ptr(), n)
        host = torch.empty(64 * n, dtype=torch.uint8).pin_memory()
        with pytest.raises(pkg.MsmError):
            c.submit_device(host.data_ptr(), ds.data_ptr(), n)                             # not resident on a device of the context
    with pkg.MsmContext((0,)) as c1:                                                       # one device: the ticket follows the window shard
        c1.set_window_shard(1, 2)
        t = c1.submit_device(dp.data_ptr(), ds.data_ptr(), n)
        part1 = c1.collect(t)
        c1.set_window_shard(0, 1)
        assert part1 != want and c1.collect(c1.submit_device(dp.data_ptr(), ds.data_ptr(), n)) == want


def test_pinned_host_buffers_are_released_at_return(pkg, ora):
    """te_msm_submit / te_msm_run promise that the caller's buffers are free when the call returns.  Copies from PINNED memory
    are truly asynchronous, so the call has to wait for its uploads: the buffers are overwritten right after it returns"""
    import torch
    n = 300000
    pts, sc = ora.gen_points(1500, n), ora.gen_scalars(1500, n)
    want = ora.msm(pts, sc, threads=8)
    hp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).pin_memory()
    hs = torch.frombuffer(bytearray(sc), dtype=torch.uint8).pin_memory()
    keep_p, keep_s = hp.clone(), hs.clone()
    from importlib import import_module
    L = import_module("webgpu-msm-twisted-edwards_amd.binding")._lib()
    as_cp = lambda t: ctypes.cast(ctypes.c_void_p(t.data_ptr()), ctypes.c_char_p)
    with pkg.MsmContext((0,)) as c:
        for chunks in (0, 1, 3):
            c.set_option("host_chunks", chunks)
            hp.copy_(keep_p); hs.copy_(keep_s)
            t = ctypes.c_uint64()
            assert L.te_msm_submit(c._h, as_cp(hp), as_cp(hs), n, ctypes.byref(t)) == 0
            hp.fill_(0xff); hs.fill_(0xff)                                                 # the call is over: the buffers are the caller's again
            assert c.collect(t.value) == want, chunks
        hp.copy_(keep_p); hs.copy_(keep_s)
        out = ctypes.create_string_buffer(96)
        assert L.te_msm_run(c._h, as_cp(hp), as_cp(hs), n, out) == 0 and out.raw[:64] == want


def test_host_staging_option(pkg, model, ora):
    """option "host_staging" = 1: host buffers travel through the work set's own pinned ring, filled 2 MB at a time by the
    engine's crew of host threads (independent of what the runtime remembers about the caller's pages).  Same results; the
    caller's buffers are free when a blocking call returns; pieces, tickets, several devices, both curves' record sizes"""
    import torch
    n = 300007                                                  # 18.3 MB of points: ten chunks, the last one short
    pts, sc = ora.gen_points(1800, n), ora.gen_scalars(1800, n)
    want = ora.msm(pts, sc, threads=8)
    small = (ora.gen_points(1801, 1000), ora.gen_scalars(1801, 1000))      # below the staging threshold: direct copies
    with pkg.MsmContext((0,)) as c:
        c.set_option("host_staging", 1)
        assert c.get_option("host_staging") == 1
        for chunks in (0, 1, 3):
            c.set_option("host_chunks", chunks)
            assert c.run(pts, sc) == want, chunks
            bp, bs = bytearray(pts), bytearray(sc)
            hp = torch.frombuffer(bp, dtype=torch.uint8); hs = torch.frombuffer(bs, dtype=torch.uint8)
            L = __import__("importlib").import_module("webgpu-msm-twisted-edwards_amd.binding")._lib()
            as_cp = lambda t: ctypes.cast(ctypes.c_void_p(t.data_ptr()), ctypes.c_char_p)
            tk = ctypes.c_uint64()
            assert L.te_msm_submit(c._h, as_cp(hp), as_cp(hs), n, ctypes.byref(tk)) == 0
            hp.fill_(0xff); hs.fill_(0xff)                      # the call is over: the buffers are the caller's again
            t2 = c.submit_async(pts, sc)
            assert c.collect(tk.value) == want and c.collect(t2) == want, chunks
        c.set_option("host_chunks", 0)
        assert c.run(*small) == ora.msm(*small)
        ts = [c.submit(pts, sc) for _ in range(pkg.WORKSETS)]   # every work set gets a ring
        assert all(c.collect(t) == want for t in ts)
        assert c.trim(0) == pkg.WORKSETS and c.run(pts, sc) == want            # rings are freed with the sets and come back
        c.set_option("host_staging", 0)
        assert c.run(pts, sc) == want
    with pkg.MsmContext((0, 0, 0, 0)) as c:
        c.set_option("host_staging", 1)
        assert c.run(pts, sc) == want                           # point slices: four device threads post to the one crew
        ts = [c.submit_async(pts, sc) for _ in range(8)]
        assert all(c.collect(t) == want for t in ts)


def test_window_count_of_15_and_5_bit_plans(pkg, model, ora):
    """signed digits run ceil(255 / c) windows (17 x 15 bits, 51 x 5): scalars of this boundary are below p < 2^253.  The
    error rule stays exact: s + sum_w 2^(c w + c - 1) >= 2^(c W) is the reference's "final carry is 1" (utils.ts:80-83)"""
    n = 64
    pts = ora.gen_points(1600, n)
    p = 8444461749428370424248824938781546531375899335154063827935233455917409239041
    with pkg.MsmContext((0,)) as c:
        for bits, W in ((15, 17), (5, 51), (16, 16), (13, 20), (8, 32)):
            c.set_option("window_bits", bits)
            assert c.plan(n) == (bits, W)
            H = sum(1 << (bits * w + bits - 1) for w in range(W))
            top = (1 << (bits * W)) - H                                                    # the first scalar that leaves a final carry
            ok = [p - 1, p - 2, (1 << 253) - 1, min(top, 1 << 254) - 1, 0, 1] + [(p - 1) >> k for k in range(1, n - 5)]     # (the oracle's own 16-bit windows hold anything below 2^255 - 2^239)
            sc = model.scalars_to_bytes(ok[:n])
            assert c.run(pts, sc) == ora.msm(pts, sc, threads=4), bits
            if top < (1 << 256):
                with pytest.raises(pkg.MsmError) as e:
                    c.run(pts, model.scalars_to_bytes([1, top] + [2] * (n - 2)))
                assert e.value.code == -3, bits
        c.set_option("signed_digits", 0)                                                   # unsigned digits: ceil(256 / c), any 256-bit scalar
        c.set_option("window_bits", 15)
        assert c.plan(n) == (15, 18)
        sc = model.scalars_to_bytes([(1 << 256) - 1, p - 1] + [3] * (n - 2))
        assert c.run(pts, sc) == ora.msm_naive(pts, sc)                                    # (the pipeline oracle has signed windows only)


def test_acceptance_of_non_canonical_scalars_follows_the_window_size(pkg, model, ora):
    """canonical scalars (< p) always fit; which NON-canonical ones are accepted depends on the window size,
    hence -- through the automatic plan -- on n: a scalar in [2^254, 2^255) leaves a final carry under the 17 x 15-bit plan that
    n = 2^16 .. 2^18 get by default (TE_MSM_ESCALAR, the reference's "final carry is 1", utils.ts:80-83) and is an ordinary scalar
    under 16 x 16 bits, the reference's own decomposition; "window_bits" = 16 gives its acceptance range at any n"""
    n = 1 << 16
    pts = ora.gen_points(1700, n)
    big = (1 << 254) + 12345
    sc = model.scalars_to_bytes([big, 7] + [3] * (n - 2))
    with pkg.MsmContext((0,)) as c:
        assert c.plan(n) == (15, 17)
        for call in (lambda: c.run(pts, sc), lambda: c.collect(c.submit(pts, sc))):
            with pytest.raises(pkg.MsmError) as e:
                call()
            assert e.value.code == -3
        c.set_option("window_bits", 16)
        assert c.plan(n) == (16, 16)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=8)          # the oracle's 16-bit windows accept it as the reference does
        c.set_option("window_bits", 0)
        small = 1 << 12
        assert c.plan(small)[0] == 11 and c.plan(3 << 17) == (16, 16)
        assert c.run(pts[:64 * small], sc[:32 * small]) == ora.msm(pts[:64 * small], sc[:32 * small], threads=4)      # 24 x 11 bits reach bit 263


def test_node_eight_promises_on_four_devices(pkg, model, ora, tmp_path):
    """From the reference's host language: eight compute_msm promises in flight on TE_MSM_DEVICES=0,0,0,0 become eight
    tickets, two per device (te_msm_submit_async from the JavaScript thread, wait + collect on libuv's pool); a lone call
    on the same device list runs as point slices; all equal the oracle"""
    import json
    import shutil
    import subprocess
    node = shutil.which("node")
    if not node:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    js = os.path.join(root, "webgpu-msm-twisted-edwards_amd", "js")
    if not os.path.exists(os.path.join(js, "te_msm_napi.node")):
        subprocess.check_call(["make", "-C", js, "-s"])
    n = 1 << 17
    pts, sc = ora.gen_points(1700, n), ora.gen_scalars(1700, n)
    (tmp_path / "p.bin").write_bytes(pts)
    (tmp_path / "s.bin").write_bytes(sc)
    exp = model.xy_from_bytes(ora.msm(pts, sc, threads=8))
    for k, env in ((8, {"TE_MSM_DEVICES": "0,0,0,0"}), (8, {"TE_MSM_DEVICES": "0,0,0,0", "UV_THREADPOOL_SIZE": "2"}), (12, {"TE_MSM_DEVICES": "0"})):
        r = subprocess.run([node, os.path.join(js, "run_concurrent.js"), str(tmp_path / "p.bin"), str(tmp_path / "s.bin"), str(k)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
        out = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert "x" in out, (out, r.stderr.decode()[-2000:])
        assert (int(out["x"]), int(out["y"])) == exp and out["all_equal"], (k, env)
        print("node %s: single %.3f ms, %d in flight %.3f ms" % (env, out["single_ms"], k, out["concurrent_ms"]))


def test_shared_record_slabs(pkg, ora):
    """round 6: calls in flight that name the same device-resident point buffer share ONE record slab -- every call still converts its
    points: the buffer may hold other points for the next call --; different buffers, different n and the other curve get slabs of
    their own; option "share_records" = 0 restores one slab per work set; the stage verifier's "records" come from the slab"""
    import torch
    n = 30000
    pa, pb = ora.gen_points(9100, n), ora.gen_points(9101, n)
    scs = [ora.gen_scalars(9200 + i, n) for i in range(4)]
    want_a = [ora.msm(pa, s, threads=8) for s in scs]
    want_b = [ora.msm(pb, s, threads=8) for s in scs]
    da, db = _dev(pa), _dev(pb)
    dsc = [_dev(s) for s in scs]
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        assert c.get_option("share_records") == 1 and c.get_option("record_slabs") == 0
        ts = [c.submit_device(da.data_ptr(), d.data_ptr(), n) for d in dsc]
        assert c.get_option("record_slabs") == 1                                             # four tickets, one point buffer, one slab
        assert [c.collect(t) for t in ts] == want_a
        # two point buffers in flight: a slab each; a shorter MSM over the first buffer: a third
        ts = [c.submit_device((da if i % 2 == 0 else db).data_ptr(), dsc[i].data_ptr(), n) for i in range(4)] + [c.submit_device(da.data_ptr(), dsc[0].data_ptr(), n // 2)]
        assert c.get_option("record_slabs") == 3
        got = [c.collect(t) for t in ts]
        assert got[:4] == [want_a[0], want_b[1], want_a[2], want_b[3]] and got[4] == ora.msm(pa[:64 * (n // 2)], scs[0][:32 * (n // 2)], threads=8)
        # nothing is remembered across calls: the same buffer with OTHER points in it
        da.copy_(db)
        torch.cuda.synchronize()
        ts = [c.submit_device(da.data_ptr(), d.data_ptr(), n) for d in dsc]
        assert [c.collect(t) for t in ts] == want_b
        assert c.run_device(da.data_ptr(), dsc[0].data_ptr(), n) == want_b[0]
        recs = c.debug_read("records", n * 128)                                              # the shared slab's records ...
        assert c.run(pb, scs[0]) == want_b[0]
        assert c.debug_read("records", n * 128) == recs                                      # ... are the ones the host path converts into the set's own slab
        held = c.get_option("device_bytes")
        assert c.trim(0) >= 1 and c.get_option("record_slabs") == 0 and c.get_option("device_bytes") < held
        c.set_option("share_records", 0)
        ts = [c.submit_device(db.data_ptr(), d.data_ptr(), n) for d in dsc]
        assert c.get_option("record_slabs") == 0
        assert [c.collect(t) for t in ts] == want_b
        c.set_option("share_records", 1)
        # the second curve on the same context: slabs are keyed by the curve as well
        from oracle import oracle377 as o
        m = 5000
        p3, s3 = o.gen_points(31, m), o.gen_scalars(31, m)
        d3, ds3 = _dev(p3), _dev(s3)
        torch.cuda.synchronize()
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        ts = [c.submit_device(d3.data_ptr(), ds3.data_ptr(), m) for _ in range(3)]
        assert [c.collect(t) for t in ts] == [o.msm(p3, s3, threads=4)] * 3


UPLOAD_SWITCH_CHILD = r'''
import importlib, json, sys
sys.path.insert(0, %(root)r)
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle.gen_golden import make_inputs
g = json.loads(sys.argv[1])
pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
want = int(g["x"]).to_bytes(32, "little") + int(g["y"]).to_bytes(32, "little")
for ids in ((0,), (0, 0)):
    with pkg.MsmContext(ids) as c:
        b = c.bind_points(pts)
        assert c.run(pts, sc) == want and c.run_scalars(b, sc) == want
        for rnd in range(3):                                   # work sets are reused: the second round's uploads follow a collected MSM
            tk = [c.submit_scalars(b, sc) for _ in range(3)] + [c.submit_async(pts, sc) for _ in range(3)] + [c.submit(pts, sc)]
            assert all(c.collect(t) == want for t in reversed(tk))
        c.release_points(b)
print("ok")
'''


def test_upload_path_switches_agree(pkg, wasm_golden):
    """Round 6 took three marker packets out of the upload path of host-buffer MSMs (csrc/te_msm.hip: copy_stream_behind_previous,
    lane_wait, gpu_t::scalar_link, caller_may_wait_on_host; profiles/r06_bound_host_tickets_gap.txt) and kept the old forms behind
    environment switches for A/B runs: every combination gives the reference's own output -- tickets over bound bases, host-buffer
    tickets (lane threads and the calling thread), lone calls, work sets reused across rounds, one and two "devices".
    Child processes: the switches are read once per process."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    g = next(x for x in wasm_golden if x["name"] == "random_n65536")
    arg = json.dumps({k: g[k] for k in ("seed", "n", "mode", "x", "y")})
    for env in ({}, {"TE_MSM_COPY_MARKER": "1", "TE_MSM_LANE_EVENT_WAITS": "1", "TE_MSM_SCALAR_UPLOADS_SERIAL": "0", "TE_MSM_CALLER_HOST_WAITS": "0"},
                {"TE_MSM_LANE_EVENT_WAITS": "1"}, {"TE_MSM_SCALAR_UPLOADS_SERIAL": "0"}, {"TE_MSM_CALLER_HOST_WAITS": "0", "TE_MSM_LANE_HOST_WAITS": "0"},
                {"TE_MSM_COPY_PRIORITY": "1"}, {"TE_MSM_HOST_STAGING": "1"}):
        r = subprocess.run([sys.executable, "-c", UPLOAD_SWITCH_CHILD % {"root": root}, arg], env=dict(os.environ, **env),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0 and r.stdout.decode().strip().endswith("ok"), (env, r.stderr.decode()[-2000:])
