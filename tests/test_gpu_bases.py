"""GPU parity of RESIDENT BASES (include/te_msm.h: te_msm_bind_points / te_msm_run_scalars[_device] / te_msm_submit_scalars[_device] /
te_msm_release_points): the points are uploaded and converted once, every MSM over them takes scalars alone.

Reference shape: the harness hands the SAME point buffer to compute_msm for every run of a size
(submission/miscellaneous/full_benchmarks.ts:63-68,100-105; ui/AllBenchmarks.tsx:213-222); compute_msm's own signature
(submission.ts:73-78) is untouched -- the bound path is an opt-in beside it.  Every result here is compared bit for bit with the
reference's own outputs (the WASM goldens) or with the oracle; the bound records with the bigint model.
One-GPU box: contexts of several "devices" name GPU 0 several times (every device holds its own copy of the records).
Nothing here reads /root/reference."""
import ctypes

import pytest

from oracle.gen_golden import make_inputs

pytestmark = pytest.mark.gpu


def _dev(buf: bytes):
    import torch
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


@pytest.mark.parametrize("ids", [(0,), (0, 0, 0, 0)])
def test_goldens_through_the_bound_path(pkg, model, wasm_golden, ids):
    """every reference-generated golden up to n = 2^20: host scalars (lone call, pieces), device scalars, asynchronous and
    device-resident tickets -- one bind per point set; the large ones also with 16-bit windows in both digit forms"""
    import torch
    D = len(ids)
    with pkg.MsmContext(ids) as c:
        assert c.get_option("bases_bound") == 0
        for g in wasm_golden:
            n = g["n"]
            if D > 1 and 4096 < n < (1 << 20) and g["mode"] != "random":
                continue                                         # the multi-device pass keeps the small cases, the random ones and the headline size
            pts, sc = make_inputs(g["seed"], n, g["mode"])
            want = (int(g["x"]), int(g["y"]))
            b = c.bind_points(pts)
            assert b.n == n and c.get_option("bases_bound") == 1 and c.get_option("bases_bytes") == D * n * 128
            assert model.xy_from_bytes(c.run_scalars(b, sc)) == want, g["name"]
            ds = _dev(sc)
            torch.cuda.synchronize()
            assert model.xy_from_bytes(c.run_scalars_device(b, ds.data_ptr())) == want, g["name"]
            ts = [c.submit_scalars(b, sc), c.submit_scalars_device(b, ds.data_ptr()), c.submit_scalars(b, sc)]
            assert c.get_option("in_flight") == 3
            with pytest.raises(pkg.MsmError):
                c.release_points(b)                              # a ticket over the set is in flight
            for t in (ts[1], ts[2], ts[0]):
                assert model.xy_from_bytes(c.collect(t)) == want, g["name"]
            if n >= 65536:
                c.set_option("window_bits", 16)
                for signed in (1, 0):
                    c.set_option("signed_digits", signed)
                    assert model.xy_from_bytes(c.run_scalars(b, sc)) == want, (g["name"], signed)
                    assert model.xy_from_bytes(c.run_scalars_device(b, ds.data_ptr())) == want, (g["name"], signed)
                c.set_option("signed_digits", 1)
                c.set_option("window_bits", 0)
            elif n <= 4096:
                for cb in (15, 5):
                    c.set_option("window_bits", cb)
                    assert model.xy_from_bytes(c.run_scalars(b, sc)) == want, (g["name"], cb)
                c.set_option("window_bits", 0)
            # the unbound call still works beside a bound set, and gives the same point
            if n <= 65536:
                assert model.xy_from_bytes(c.run(pts, sc)) == want, g["name"]
            c.release_points(b)
            assert c.get_option("bases_bound") == 0 and c.get_option("bases_bytes") == 0
            with pytest.raises(pkg.MsmError):
                c.run_scalars(b, sc)                             # released
            del ds


def test_bound_records_equal_the_per_call_conversion(pkg, ora, model):
    """the bound records of the Twisted-Edwards curve are the records of the per-call conversion, byte for byte, on every device,
    and ((y - x)/2, (y + x)/2, -d x y) in Montgomery form by the bigint model"""
    n = 1000
    pts, sc = ora.gen_points(77, n), ora.gen_scalars(77, n)
    with pkg.MsmContext((0, 0)) as c2, pkg.MsmContext((0,)) as c:
        c.set_option("prezero", 0)
        want = ora.msm(pts, sc, threads=4)
        assert c.run(pts, sc) == want
        recs = c.debug_read("records", n * 128)
        for ctx, D in ((c, 1), (c2, 2)):
            b = ctx.bind_points(pts)
            for di in range(D):
                rb, got = ctx.bases_read(b, 0, n, di)
                assert rb == 128 and got == recs, di
            rb, part = ctx.bases_read(b, 10, 5)
            assert part == recs[10 * 128:15 * 128]
            assert ctx.run_scalars(b, sc) == want
        P = model.P
        rinv = pow(1 << 261, -1, P)
        for i in (0, 1, 500, n - 1):
            x, y = int.from_bytes(pts[64 * i:64 * i + 32], "little"), int.from_bytes(pts[64 * i + 32:64 * i + 64], "little")
            hm, hp, dt = (sum(int.from_bytes(recs[128 * i + 36 * k + 4 * j:128 * i + 36 * k + 4 * j + 4], "little") << (29 * j) for j in range(9)) * rinv % P
                          for k in range(3))
            assert (hp - hm) % P == x and (hp + hm) % P == y and dt == (-3021 * x * y) % P, i
        # device-resident points give the same set
        dp = _dev(pts)
        import torch
        torch.cuda.synchronize()
        bd = c.bind_points_device(dp.data_ptr(), n)
        assert c.bases_read(bd, 0, n)[1] == recs and c.run_scalars(bd, sc) == want
        bd2 = c2.bind_points_device(dp.data_ptr(), n)
        assert c2.bases_read(bd2, 0, n, 1)[1] == recs and c2.run_scalars(bd2, sc) == want


@pytest.mark.parametrize("n", [1, 2, 3, 7, 63, 257, 4097, 70001, 200003])
def test_ragged_sizes_pieces_and_digit_forms(pkg, ora, n):
    """ragged n, fewer points than pieces or devices, every piece count, both digit forms, against the oracle"""
    pts, sc = ora.gen_points(5000 + n, n), ora.gen_scalars(5000 + n, n)
    want = ora.msm(pts, sc, threads=8)
    for ids in ((0,), (0, 0, 0)):
        with pkg.MsmContext(ids) as c:
            c.set_option("host_shard_min", 16)
            b = c.bind_points(pts)
            for chunks in (0, 1, 2, 3, 5):
                c.set_option("scalar_chunks", chunks)
                assert c.run_scalars(b, sc) == want, (ids, chunks)
                t = c.submit_scalars(b, sc)
                assert c.collect(t) == want, (ids, chunks)
            c.set_option("scalar_chunks", 0)
            c.set_option("signed_digits", 0)
            assert c.run_scalars(b, sc) == want
            c.set_option("signed_digits", 1)
            for cb in (8, 13, 16):
                c.set_option("window_bits", cb)
                assert c.run_scalars(b, sc) == want, cb
            c.set_option("window_bits", 0)
            if len(ids) == 1:                                   # a window-sharded single-device context folds its own rows only
                c.set_option("segment_len", 3)
                assert c.run_scalars(b, sc) == want
                c.set_option("segment_len", 0)


def test_errors_capacity_and_mixed_tickets(pkg, ora):
    """a scalar-range error belongs to its ticket; capacity; bound and unbound tickets mixed; two sets on one context; an empty set"""
    n = 50000
    pts, sc = ora.gen_points(610, n), ora.gen_scalars(610, n)
    pts2, sc2 = ora.gen_points(611, n // 2), ora.gen_scalars(611, n // 2)
    want, want2 = ora.msm(pts, sc, threads=8), ora.msm(pts2, sc2, threads=8)
    bad = bytearray(sc); bad[32 * 7:32 * 8] = b"\xff" * 32
    with pkg.MsmContext((0,)) as c:
        c.set_option("window_bits", 16)
        b, b2 = c.bind_points(pts), c.bind_points(pts2)
        assert c.get_option("bases_bound") == 2
        ts = [c.submit_scalars(b, sc), c.submit_scalars(b, bytes(bad)), c.submit_scalars(b2, sc2), c.submit(pts, sc), c.submit_async(pts2, sc2)]
        assert c.run_scalars(b2, sc2) == want2                    # a lone call beside tickets: a work set no ticket owns
        with pytest.raises(pkg.MsmError) as e:
            c.collect(ts[1])
        assert e.value.code == -3
        assert c.collect(ts[4]) == want2 and c.collect(ts[0]) == want and c.collect(ts[3]) == want and c.collect(ts[2]) == want2
        with pytest.raises(pkg.MsmError) as e:
            c.run_scalars(b, bytes(bad))
        assert e.value.code == -3
        ts = [c.submit_scalars(b, sc) for _ in range(pkg.WORKSETS)]
        with pytest.raises(pkg.MsmError) as e:
            c.submit_scalars(b, sc)
        assert e.value.code == -4                                  # every work set is taken
        for t in ts:
            assert c.collect(t) == want
        c.set_option("window_bits", 0)
        # wrong length, foreign set, empty set
        with pytest.raises(pkg.MsmError):
            c.run_scalars(b, sc[:-32])
        with pkg.MsmContext((0,)) as other:
            with pytest.raises(pkg.MsmError):
                other.run_scalars(b, sc)
        e0 = c.bind_points(b"")
        ident = (0).to_bytes(32, "little") + (1).to_bytes(32, "little")
        assert c.run_scalars(e0, b"") == ident
        c.release_points(e0)
        # trim beside bound sets: the records are not work-set buffers
        held = c.get_option("bases_bytes")
        c.trim(0)
        assert c.get_option("bases_bytes") == held and c.run_scalars(b, sc) == want
        b.release(); b2.release()
        assert c.get_option("bases_bound") == 0


def test_witness_like_scalars_and_giant_buckets_over_bound_points(pkg, ora, wasm_golden, model):
    """a prover's skewed scalars (the reference-generated witness goldens) and all-equal scalars over a bound set, in pieces: the
    giant-bucket combine runs once per piece onto the same buckets"""
    with pkg.MsmContext((0,)) as c:
        for g in wasm_golden:
            if g["mode"] != "witness":
                continue
            pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
            b = c.bind_points(pts)
            for chunks in (1, 3):
                c.set_option("scalar_chunks", chunks)
                assert model.xy_from_bytes(c.run_scalars(b, sc)) == (int(g["x"]), int(g["y"])), (g["name"], chunks)
            b.release()
        n = 30000
        pts = ora.gen_points(9, n)
        k = (123456789 << 64 | 987654321).to_bytes(32, "little")
        sc = k * n
        want = ora.msm(pts, sc, threads=8)
        b = c.bind_points(pts)
        for chunks in (1, 2, 4):
            c.set_option("scalar_chunks", chunks)
            assert c.run_scalars(b, sc) == want, chunks
        t = [c.submit_scalars(b, sc) for _ in range(3)]
        assert all(c.collect(x) == want for x in t)


def test_window_shard_of_a_single_device_context_over_bound_points(pkg, ora):
    """te_msm_set_window_shard + te_msm_run_scalars[_device]: the shards' rows are not exposed here, but the folded shard results
    sum to the MSM (window sums are points: shard results add up) -- checked through the oracle's point addition"""
    from oracle import model as m
    n = 20000
    pts, sc = ora.gen_points(808, n), ora.gen_scalars(808, n)
    want = m.xy_from_bytes(ora.msm(pts, sc, threads=8))
    import torch
    ds = _dev(sc)
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        b = c.bind_points(pts)
        for world in (2, 3):
            for how in ("host", "device"):
                acc = (0, 1)
                for r in range(world):
                    c.set_window_shard(r, world)
                    part = c.run_scalars(b, sc) if how == "host" else c.run_scalars_device(b, ds.data_ptr())
                    acc = m.add(acc, m.xy_from_bytes(part))
                assert acc == want, (world, how)
        c.set_window_shard(0, 1)
        with pytest.raises(pkg.MsmError):
            c.set_window_shard(1, 2); c.submit_scalars(b, sc)
        c.set_window_shard(0, 1)


# ------------------------------------------------------------------ BLS12-377 G1: affine records at bind time
def test_bls12_377_affine_records_against_the_model(pkg, fq377check):
    """te_msm_bind_points on BLS12-377: 168-byte AFFINE records (hm, hp, dt) = projective record / z (one inversion per point,
    Montgomery's trick in groups of eight): mapped back to y^2 = x^3 + 1 they are the input points, dt = -d x y; with
    bind_affine = 0 the set keeps the 224-byte records of the per-call conversion"""
    from oracle import model377 as m
    from oracle import oracle377 as o
    from test_oracle_bls377 import _edwards_consts, edwards_to_weierstrass
    n = 1003                                                     # not a multiple of the group of eight
    pts, sc = o.gen_points(21, n), o.gen_scalars(21, n)
    want = o.msm(pts, sc, threads=4)
    s_, f_, d_ = _edwards_consts(fq377check)
    rinv = pow(1 << 406, -1, m.Q)
    with pkg.MsmContext((0, 0)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        b = c.bind_points(pts)
        assert c.get_option("bases_bytes") == 2 * n * 168
        rb, recs = c.bases_read(b, 0, n, 1)
        assert rb == 168 and len(recs) == n * 168 and recs == c.bases_read(b, 0, n, 0)[1]
        for i in (0, 1, 7, 8, 9, 500, n - 4, n - 3, n - 2, n - 1):
            hm, hp, dt = (sum(int.from_bytes(recs[168 * i + 56 * k + 4 * j:168 * i + 56 * k + 4 * j + 4], "little") << (29 * j) for j in range(14)) * rinv % m.Q
                          for k in range(3))
            xa, ya = (hp - hm) % m.Q, (hp + hm) % m.Q
            assert edwards_to_weierstrass(xa, ya, s_, f_) == m.xy_from_bytes(pts[96 * i:96 * i + 96]), i
            assert dt == -d_ * xa * ya % m.Q, i
        assert c.run_scalars(b, sc) == want
        c.set_option("bind_affine", 0)
        bp = c.bind_points(pts)
        assert c.bases_read(bp, 0, 1)[0] == 224
        assert c.run_scalars(bp, sc) == want and c.run_scalars(b, sc) == want          # both kinds of set live side by side
        c.set_option("curve", pkg.CURVE_TE_BLS12)
        with pytest.raises(pkg.MsmError):
            c.run_scalars(b, sc)                                 # bound under the other curve


@pytest.mark.parametrize("n", [1, 5, 8, 9, 1000, 4097, 70001])
def test_bls12_377_bound_sizes_windows_and_tickets(pkg, n):
    from oracle import oracle377 as o
    pts, sc = o.gen_points(300 + n, n), o.gen_scalars(300 + n, n)
    want = o.msm(pts, sc, threads=8)
    import torch
    ds = _dev(sc)
    torch.cuda.synchronize()
    for ids in ((0,), (0, 0)):
        with pkg.MsmContext(ids) as c:
            c.set_option("curve", pkg.CURVE_BLS12_377_G1)
            c.set_option("host_shard_min", 16)
            for affine in (1, 0):
                c.set_option("bind_affine", affine)
                b = c.bind_points(pts)
                assert c.run_scalars(b, sc) == want, (ids, affine)
                assert c.run_scalars_device(b, ds.data_ptr()) == want, (ids, affine)
                ts = [c.submit_scalars(b, sc), c.submit_scalars_device(b, ds.data_ptr())]
                assert [c.collect(t) for t in ts] == [want, want]
                if n >= 1000:
                    for cb, signed, chunks in ((13, 1, 2), (16, 0, 3), (11, 1, 1)):
                        c.set_option("window_bits", cb); c.set_option("signed_digits", signed); c.set_option("scalar_chunks", chunks)
                        assert c.run_scalars(b, sc) == want, (cb, signed, chunks)
                    c.set_option("window_bits", 0); c.set_option("signed_digits", 1); c.set_option("scalar_chunks", 0)
                b.release()


def test_bls12_377_full_size_bound(pkg):
    """n = 2^20 over affine records, tickets in flight, against the oracle"""
    from oracle import oracle377 as o
    n = 1 << 20
    pts, sc = o.gen_points(0x377, n), o.gen_scalars(0x377, n)
    want = o.msm(pts, sc, threads=16)
    with pkg.MsmContext((0,)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        b = c.bind_points(pts)
        assert c.run_scalars(b, sc) == want
        ts = [c.submit_scalars(b, sc) for _ in range(4)]
        assert all(c.collect(t) == want for t in ts)


# ------------------------------------------------------------------ the Python mirror of the entry point
def test_compute_msm_with_set_bases(pkg, model, wasm_golden):
    """compute_msm(bufferPoints, bufferScalars) keeps its signature (submission.ts:73-78); after set_bases(buffer) the calls that
    pass that very buffer take the scalars-only path, any other buffer the ordinary one"""
    g = next(x for x in wasm_golden if x["name"] == "random_n65536")
    pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
    want = {"x": int(g["x"]), "y": int(g["y"])}
    assert pkg.compute_msm(pts, sc, log_result=False) == want
    pkg.set_bases(pts)
    ctx = pkg.binding._DEFAULT_CTX
    assert ctx.get_option("bases_bound") == 1
    for _ in range(3):
        assert pkg.compute_msm(pts, sc, log_result=False) == want
    other = bytes(bytearray(pts))                               # equal contents, another object: the ordinary path
    assert pkg.compute_msm(other, sc, log_result=False) == want
    assert pkg.compute_msm(pts, sc, log_result=False, force_recompile=True) == want      # a new context binds the buffer again
    assert pkg.binding._DEFAULT_CTX.get_option("bases_bound") == 1
    pkg.set_bases(None)
    assert pkg.binding._DEFAULT_CTX.get_option("bases_bound") == 0
    assert pkg.compute_msm(pts, sc, log_result=False) == want


def test_node_set_bases(pkg, model, wasm_golden, tmp_path):
    """from the reference's host language: setBases(points) once, then compute_msm(points, scalars) -- the same Buffer object --
    moves the scalars only (the addon counts those jobs); a copy of the buffer takes the ordinary path; both equal the reference's
    own output; eight promises in flight over the bound buffer on one and on four "devices"; a pool of ONE libuv thread with more
    promises than work sets (round 5's protocol hung there)"""
    import json
    import os
    import subprocess
    from test_gpu_parity import _node_js_dir
    node, js = _node_js_dir()
    g = next(x for x in wasm_golden if x["name"] == "random_n262144")
    pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
    (tmp_path / "p.bin").write_bytes(pts)
    (tmp_path / "s.bin").write_bytes(sc)
    want = (int(g["x"]), int(g["y"]))

    def run(args, env=None):
        r = subprocess.run([node, os.path.join(js, "run_concurrent.js"), str(tmp_path / "p.bin"), str(tmp_path / "s.bin")] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        out = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert "x" in out, (out, r.stderr.decode()[-2000:])
        assert (int(out["x"]), int(out["y"])) == want and out["all_equal"], out
        return out

    out = run(["8", "-", "bases"])
    print("node, bound bases: single %.3f ms, eight in flight %.3f ms" % (out["single_ms"], out["concurrent_ms"]))
    assert out["stats"]["boundJobs"] == 2 + 4 * 8 and out["stats"]["maxInFlight"] == 8, out["stats"]
    out = run(["8", "0,0,0,0", "bases"])
    assert out["devices"] == [0, 0, 0, 0] and out["stats"]["boundJobs"] == 2 + 4 * 8, out
    out = run(["20", "-", "bases"], env=dict(os.environ, UV_THREADPOOL_SIZE="1"))       # 20 promises, 8 work sets, one pool thread
    assert out["stats"]["boundJobs"] == 2 + 4 * 20 and out["stats"]["maxInFlight"] == 8, out["stats"]
    out = run(["20"], env=dict(os.environ, UV_THREADPOOL_SIZE="1"))                      # the same without bases
    assert out["stats"]["boundJobs"] == 0 and out["stats"]["maxInFlight"] == 8, out["stats"]


def test_node_awaited_calls_are_tickets_from_the_javascript_thread(pkg, wasm_golden, tmp_path):
    """the reference harness's call pattern (every compute_msm awaited, ui/Benchmark.tsx:29-39) through the addon: on one device a lone
    promise becomes a ticket in enter() -- its upload starts while libuv still hands the job to a pool thread --, with and without
    bound bases, and on several devices it stays the lone call of its pool thread (all devices on the one MSM); results equal the
    reference's own output either way"""
    import json
    import os
    import subprocess
    from test_gpu_parity import _node_js_dir
    node, js = _node_js_dir()
    g = next(x for x in wasm_golden if x["name"] == "random_n262144")
    pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
    (tmp_path / "p.bin").write_bytes(pts)
    (tmp_path / "s.bin").write_bytes(sc)
    want = (int(g["x"]), int(g["y"]))

    def run(args, env=None):
        r = subprocess.run([node, os.path.join(js, "run_awaited.js"), str(tmp_path / "p.bin"), str(tmp_path / "s.bin")] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
        out = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert "x" in out, (out, r.stderr.decode()[-2000:])
        assert (int(out["x"]), int(out["y"])) == want, out
        return out

    out = run(["6", "bases"])                                       # 5 warm-up calls + 6 timed ones, all over the bound buffer
    assert out["stats"]["boundJobs"] == 11 and out["stats"]["submittedInEnter"] == 11 and out["stats"]["maxInFlight"] == 1, out["stats"]
    print("node, awaited calls over bound bases at 2^18: median %.3f ms" % out["median_ms"])
    out = run(["6"])                                                # the very first call creates the context in its pool thread
    assert out["stats"]["submittedInEnter"] == 10 and out["stats"]["submittedInExecute"] == 1 and out["stats"]["boundJobs"] == 0, out["stats"]
    out = run(["6"], env=dict(os.environ, TE_MSM_LONE_IN_ENTER="0"))
    assert out["stats"]["submittedInEnter"] == 0 and out["stats"]["submittedInExecute"] == 11, out["stats"]
    out = run(["6"], env=dict(os.environ, TE_MSM_DEVICES="0,0"))
    assert out["stats"]["loneRuns"] == 11 and out["stats"]["submittedInEnter"] == 0, out["stats"]


# ------------------------------------------------------------------ fixed-base windows over a bound set (option "bind_fixed_base")
def test_fixed_base_table_against_the_model(pkg, ora, model):
    """table w of a fixed-base set holds the records of 2^(c w) P_i: ((y - x)/2, (y + x)/2, -d x y) of that multiple, by the bigint
    model; table 0 is the ordinary conversion byte for byte"""
    n, c = 37, 20
    pts = ora.gen_points(4242, n)
    P = model.P
    rinv = pow(1 << 261, -1, P)
    with pkg.MsmContext((0,)) as cx:
        plain = cx.bind_points(pts)
        ordinary = cx.bases_read(plain, 0, n)[1]
        cx.set_option("bind_fixed_base", c)
        b = cx.bind_points(pts)
        W = -(-255 // c)
        assert cx.get_option("bases_bytes") == n * 128 * (1 + W)
        rb, tab = cx.bases_read(b, 0, n * W)
        assert rb == 128 and len(tab) == n * W * 128 and tab[:n * 128] == ordinary
        for w in (1, 2, W - 1):
            for i in (0, 5, n - 1):
                x, y = int.from_bytes(pts[64 * i:64 * i + 32], "little"), int.from_bytes(pts[64 * i + 32:64 * i + 64], "little")
                qx, qy = model.scalar_mul(1 << (c * w), (x, y))
                o = (w * n + i) * 128
                hm, hp, dt = (sum(int.from_bytes(tab[o + 36 * k + 4 * j:o + 36 * k + 4 * j + 4], "little") << (29 * j) for j in range(9)) * rinv % P for k in range(3))
                assert ((hp - hm) % P, (hp + hm) % P) == (qx, qy) and dt == (-3021 * qx * qy) % P, (w, i)


@pytest.mark.parametrize("c", [16, 17, 18, 19, 20, 21])
def test_fixed_base_windows_sizes_and_forms(pkg, ora, c):
    """every table width, ragged and tiny n, host and device scalars, lone calls and tickets (also on four "devices"), against the oracle"""
    import torch
    for n in (1, 2, 9, 1000, 4097, 70001):
        pts, sc = ora.gen_points(7000 + n + c, n), ora.gen_scalars(7100 + n + c, n)
        want = ora.msm(pts, sc, threads=8)
        ds = _dev(sc)
        torch.cuda.synchronize()
        for ids in ((0,), (0, 0, 0, 0)):
            if len(ids) > 1 and n not in (9, 70001):
                continue
            with pkg.MsmContext(ids) as cx:
                cx.set_option("bind_fixed_base", c)
                b = cx.bind_points(pts)
                assert cx.run_scalars(b, sc) == want, (c, n, ids)
                assert cx.run_scalars_device(b, ds.data_ptr()) == want, (c, n, ids)
                ts = [cx.submit_scalars(b, sc), cx.submit_scalars_device(b, ds.data_ptr()), cx.submit_scalars(b, sc), cx.submit_scalars_device(b, ds.data_ptr())]
                assert [cx.collect(t) for t in reversed(ts)] == [want] * 4, (c, n, ids)
                assert cx.get_option("fixed_base_fallbacks") == 0
                b.release()


def test_fixed_base_windows_against_the_reference(pkg, model, wasm_golden):
    """the reference-generated goldens -- edge scalars (0, 1, p - 1, 2^k boundaries), witness-like skew, harness mode, random points --
    through the c = 20 table, the headline size included; skewed scalars may overflow a row: the engine then answers with the
    ordinary windows (counted), the result is the reference's either way"""
    import torch
    with pkg.MsmContext((0,)) as cx:
        cx.set_option("bind_fixed_base", 20)
        for g in wasm_golden:
            if 4096 < g["n"] < (1 << 20) and g["mode"] not in ("random", "witness", "edge"):
                continue
            pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
            want = (int(g["x"]), int(g["y"]))
            b = cx.bind_points(pts)
            before = cx.get_option("fixed_base_fallbacks")
            assert model.xy_from_bytes(cx.run_scalars(b, sc)) == want, g["name"]
            ds = _dev(sc)
            torch.cuda.synchronize()
            ts = [cx.submit_scalars_device(b, ds.data_ptr()), cx.submit_scalars(b, sc)]
            assert [model.xy_from_bytes(cx.collect(t)) for t in ts] == [want, want], g["name"]
            if g["mode"] in ("chain", "random") and g["n"] >= 256:
                assert cx.get_option("fixed_base_fallbacks") == before, g["name"]           # well-spread digits never overflow
            b.release()
            del ds


def test_fixed_base_fallback_errors_and_mixed_sets(pkg, ora, model):
    """all scalars equal: every window's n entries land in ONE bucket -- a row overflows, the MSM is answered by the ordinary windows;
    a scalar out of range is the ticket's error; fixed-base and ordinary sets and unbound calls side by side on one context"""
    n = 40000
    pts, sc = ora.gen_points(8100, n), ora.gen_scalars(8100, n)
    want = ora.msm(pts, sc, threads=8)
    k = (0x1234567 << 200 | 0xabcdef << 100 | 99).to_bytes(32, "little")
    same = k * n
    want_same = ora.msm(pts, same, threads=8)
    bad = bytearray(sc); bad[32 * 11:32 * 12] = b"\xff" * 32
    with pkg.MsmContext((0,)) as cx:
        plain = cx.bind_points(pts)
        cx.set_option("bind_fixed_base", 19)
        fb = cx.bind_points(pts)
        assert cx.run_scalars(fb, sc) == want and cx.run_scalars(plain, sc) == want and cx.run(pts, sc) == want
        assert cx.get_option("fixed_base_fallbacks") == 0
        assert cx.run_scalars(fb, same) == want_same
        assert cx.get_option("fixed_base_fallbacks") == 1
        # which non-canonical scalars are accepted follows the window size here as everywhere: 14 x 19 bits
        # reach bit 265 -- 2^256 - 1 is an ordinary scalar --, 15 x 17 bits = 255 do not: TE_MSM_ESCALAR, in that ticket only
        cx.set_option("bind_fixed_base", 17)
        fb17 = cx.bind_points(pts)
        ts = [cx.submit_scalars(fb, same), cx.submit_scalars(fb, sc), cx.submit_scalars(plain, same), cx.submit_scalars(fb17, bytes(bad)), cx.submit_scalars(fb, bytes(bad)),
              cx.submit_scalars(fb17, sc)]
        assert cx.collect(ts[1]) == want and cx.collect(ts[0]) == want_same and cx.collect(ts[2]) == want_same and cx.collect(ts[5]) == want
        assert cx.get_option("fixed_base_fallbacks") == 2
        with pytest.raises(pkg.MsmError) as e:
            cx.collect(ts[3])
        assert e.value.code == -3
        assert cx.collect(ts[4]) == ora.msm_naive(pts, bytes(bad))
        with pytest.raises(pkg.MsmError) as e:
            cx.run_scalars(fb17, bytes(bad))
        assert e.value.code == -3
        # the work sets go back and forth between the two kinds of plan
        for _ in range(2):
            assert cx.run_scalars(plain, sc) == want and cx.run_scalars(fb, sc) == want and cx.run(pts, sc) == want
