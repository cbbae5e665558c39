"""GPU tests added in round 2: per-stage verifiers against the bigint MODEL (not against the host build of the same
header), the ZPrize file format end to end, the RCCL path of ShardedPipeline, work-set ownership, and the plan that
te_msm_finalize folds with.  All through the C-ABI; `-m gpu`.  Nothing here reads /root/reference."""
import importlib
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R261 = 1 << 261


def _dev(buf: bytes):
    import torch
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


def _limbs_value(words):
    return sum(int(v) << (29 * i) for i, v in enumerate(words))


def _digits(ora, sc, c, signed=True):
    """signed digit d[w][i] of every scalar (reference walk, miscellaneous/utils.ts:52-95)"""
    B = 1 << (c - 1)
    return ora.decompose_scalars_signed(sc, c).astype(np.int64) - B


# ------------------------------------------------------------------ K1a against the model
def test_records_against_bigint_model(pkg, model, ora):
    """K1 (convert_point_coords...wgsl:37-77): the record of point i is ((y-x)/2, (y+x)/2, -d*x*y) in Montgomery form
    R = 2^261, lazily reduced -- compared with plain modular arithmetic, not with the host build of the device header."""
    n = 3001
    pts, sc = ora.gen_points(4001, n), ora.gen_scalars(4001, n)
    P = model.P
    with pkg.MsmContext((0,)) as c:
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=4)
        recs = np.frombuffer(c.debug_read("records", n * 128), dtype=np.uint32).reshape(n, 32)
    inv2 = pow(2, -1, P)
    for i in list(range(0, n, 13)) + [n - 1]:
        x, y = model.xy_from_bytes(pts[64 * i:64 * i + 64])
        w = recs[i]
        assert np.all(w[27:] == 0), "slot padding"
        for k, want in enumerate(((y - x) * inv2, (y + x) * inv2, -model.D * x * y)):
            limbs = w[9 * k:9 * k + 9]
            assert np.all(limbs[:8] < (1 << 29)), "limb class N"
            v = _limbs_value(limbs)
            assert v < 2 * P, "lazy bound"
            assert v % P == want * R261 % P, f"record {i} field {k}"


# ------------------------------------------------------------------ K4 / K5 against the model
def _row_points(model, row: bytes):
    """the 5 extended points of a 720-byte row as affine model points"""
    P = model.P
    rinv = pow(R261, -1, P)
    out = []
    for s in range(5):
        words = np.frombuffer(row[144 * s:144 * s + 144], dtype=np.uint32).reshape(4, 9)
        x, y, z, t = [_limbs_value(words[k]) * rinv % P for k in range(4)]
        zi = pow(z, -1, P)
        assert (x * zi % P) * (y * zi % P) % P == t * zi % P, "T = XY/Z"
        out.append((x * zi % P, y * zi % P))
    return out


@pytest.mark.parametrize("n,c,signed", [(1500, 9, 1), (5003, 13, 1), (2000, 8, 0), (9000, 16, 1)])
def test_bucket_reduction_rows_against_model(pkg, model, ora, n, c, signed):
    """The reference verifies K4 and K5 separately (submission.ts:1087-1261, :1263-1363).  Here the device reduces a
    window to the row [T | W0 | W1 | W2 | W3]: T = sum of the window's buckets, W_k = sum_v v * (sum of the buckets whose
    index has digit k equal to v).  Both are recomputed from the digits with the bigint model for two windows, and the
    row folded with bucket weights must be the window's sum_i digit_i * P_i (what K4+K5 of the reference produce)."""
    pts, sc = ora.gen_points(7000 + n, n), ora.gen_scalars(7000 + n, n)
    with pkg.MsmContext((0,)) as ctx:
        ctx.set_option("window_bits", c)
        ctx.set_option("signed_digits", signed)
        ctx.set_option("prezero", 0)       # the rows live in the block that is otherwise cleared behind the read-back
        assert ctx.run(pts, sc) == ora.msm(pts, sc, threads=4)
        W = (256 + c - 1) // c
        rows = ctx.debug_read("partials", W * 720)
    logB = c - 1 if signed else c
    dw = [(logB + 3 - k) // 4 for k in range(4)]
    if signed:
        d = _digits(ora, sc, c)
    else:
        ks = [int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(n)]
        d = np.array([[(k >> (c * w)) & ((1 << c) - 1) for k in ks] for w in range(W)], dtype=np.int64)
    plist = [model.xy_from_bytes(pts[64 * i:64 * i + 64]) for i in range(n)]
    for w in (0, W // 2, W - 2):
        got = _row_points(model, rows[720 * w:720 * w + 720])
        # per-bucket affine sums from the digits
        buckets = {}
        for i in np.nonzero(d[w])[0]:
            b = abs(int(d[w][i])) - 1
            p_i = plist[int(i)] if d[w][i] > 0 else model.neg(plist[int(i)])
            buckets[b] = model.add(buckets.get(b, model.ZERO), p_i)
        # digit marginals M_k[v] = sum of the buckets whose index has digit k equal to v, then W_k = sum_v v * M_k[v]
        T = model.ZERO
        M = [dict() for _ in range(4)]
        for b, s in buckets.items():
            T = model.add(T, s)
            sh = 0
            for k in range(4):
                v = (b >> sh) & ((1 << dw[k]) - 1)
                sh += dw[k]
                if v:
                    M[k][v] = model.add(M[k].get(v, model.ZERO), s)
        Wk = []
        for k in range(4):
            acc = model.ZERO
            for v, s in M[k].items():
                acc = model.add(acc, model.scalar_mul(v, s))
            Wk.append(acc)
        assert got[0] == T, f"window {w}: T"
        for k in range(4):
            assert got[1 + k] == Wk[k], f"window {w}: W{k}"
        # the row folded with the digit weights is the window sum  sum_i digit_i * P_i  (what K4 + K5 of the reference
        # produce); the C oracle computes it as an MSM over the window's digits (-d = l - d on the prime-order subgroup)
        V, sh = T, 0
        for k in range(4):
            V = model.add(V, model.scalar_mul(1 << sh, Wk[k]))
            sh += dw[k]
        wsc = model.scalars_to_bytes([int(x) % model.L for x in d[w]])
        assert V == model.xy_from_bytes(ora.msm(pts, wsc, threads=4))


# ------------------------------------------------------------------ BASELINE configs 2 and 3 at the full size
@pytest.mark.parametrize("signed", [1, 0])
def test_full_size_2_20_both_digit_forms(pkg, model, ora, signed):
    """n = 2^20, 16-bit windows, signed digits (BASELINE config 3, the reference's shipped form) and plain unsigned windows
    (config 2, 65 536 buckets per window): bit-exact against the oracle, plus the harness-mode closed form"""
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0x5EED0014, n)
    exp = ora.msm(pts, sc, c=16, threads=16)
    with pkg.MsmContext((0,)) as ctx:
        ctx.set_option("window_bits", 16)
        ctx.set_option("signed_digits", signed)
        assert ctx.run(pts, sc) == exp
        dp, ds = _dev(pts), _dev(sc)
        import torch
        torch.cuda.synchronize()
        tickets = [ctx.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(3)]
        assert all(ctx.collect(t) == exp for t in tickets)
        fixed, _ = pkg.synth_inputs(0, n, fixed_point=True, scalars=False)
        ks = np.frombuffer(sc, dtype="<u8").reshape(n, 4).astype(object)
        total = int(ks[:, 0].sum()) + (int(ks[:, 1].sum()) << 64) + (int(ks[:, 2].sum()) << 128) + (int(ks[:, 3].sum()) << 192)
        assert model.xy_from_bytes(ctx.run(fixed, sc)) == model.scalar_mul(total % model.L, (model.HX, model.HY))


# ------------------------------------------------------------------ ZPrize file format, end to end on the GPU
def test_zprize_format_case_end_to_end(pkg, model, ora, tmp_path):
    """test-data/testCases.ts:35-52: a case in the official file format (one JSON point {x,y,t,z} per line, one decimal
    scalar per line), written here from synthetic inputs because the official files are not in the reference tree
    (README.md:22-33): load_test_case -> te_msm_run -> oracle."""
    td = importlib.import_module(pkg.__name__ + ".testdata")
    n = 20011
    pts, sc = ora.gen_points(0x2A11, n), ora.gen_scalars(0x2A11, n)
    P = model.P
    with open(tmp_path / "14-power-points.txt", "w") as f:
        for i in range(n):
            x, y = model.xy_from_bytes(pts[64 * i:64 * i + 64])
            f.write('{"x": "%d", "y": "%d", "t": "%d", "z": "1"}\n' % (x, y, x * y % P))
    with open(tmp_path / "14-power-scalars.txt", "w") as f:
        for i in range(n):
            f.write("%d\n" % int.from_bytes(sc[32 * i:32 * i + 32], "little"))
    bp, bs = td.load_test_case(str(tmp_path / "14-power-points.txt"), str(tmp_path / "14-power-scalars.txt"))
    assert bp == pts and bs == sc
    out = pkg.compute_msm(bp, bs, log_result=False)
    assert (out["x"], out["y"]) == model.xy_from_bytes(ora.msm(pts, sc, threads=8))


# ------------------------------------------------------------------ RCCL path of the window-sharded pipeline
@pytest.fixture()
def nccl_world1():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_sharded_pipeline_over_rccl(pkg, model, ora, nccl_world1):
    """ShardedPipeline (sharding.py) over the `nccl` backend (= RCCL) with one rank: async all_gather_into_tensor, the copy
    stream, events and the gathered tail, several MSMs in flight, results against the oracle; a scalar-range error is
    reported by collect() for the slot that saw it and the pipeline goes on."""
    import torch
    dist = nccl_world1
    n = 60000
    cases = []
    for seed in (301, 302, 303, 304, 305):
        pts, sc = ora.gen_points(seed, n), ora.gen_scalars(seed, n)
        cases.append((_dev(pts), _dev(sc), ora.msm(pts, sc, threads=8)))
    torch.cuda.synchronize()
    for signed in (1, 0):
        with pkg.MsmContext((0,)) as ctx:
            ctx.set_option("signed_digits", signed)
            ctx.set_window_shard(*pkg.window_shard_for_rank(0, 1))
            pipe = pkg.ShardedPipeline(ctx, n, dist, depth=3)
            tickets, got = [], []
            for dp, ds, _ in cases:
                tickets.append(pipe.submit(dp, ds))
                if len(tickets) == 3:
                    got.append(pipe.collect(tickets.pop(0)))
            while tickets:
                got.append(pipe.collect(tickets.pop(0)))
            assert got == [e for _, _, e in cases]
            # launch sequences of up to three MSMs (te_msm_partial_device_batch), two in flight; the last batch is short
            bp = pkg.ShardedPipeline(ctx, n, dist, depth=2, batch=3)
            t0 = bp.submit_batch([(dp, ds) for dp, ds, _ in cases[:3]])
            t1 = bp.submit_batch([(dp, ds) for dp, ds, _ in cases[3:]])
            assert bp.collect_batch(t0) + bp.collect_batch(t1) == [e for _, _, e in cases]
            t2 = bp.submit(cases[2][0], cases[2][1])
            assert bp.collect(t2) == cases[2][2]
            # host buffers: the rank uploads its slice, one all-gather per buffer over RCCL assembles the whole (load_host)
            pts_h, sc_h = ora.gen_points(301, n), ora.gen_scalars(301, n)
            dp_h, ds_h = pipe.load_host(pts_h, sc_h)
            assert dp_h.is_cuda and bytes(dp_h.cpu().numpy()) == pts_h and bytes(ds_h.cpu().numpy()) == sc_h
            assert pipe.collect(pipe.submit()) == cases[0][2]
            # the synchronous form over the same backend
            part = torch.zeros(pipe.W * pkg.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
            assert pkg.compute_msm_sharded(ctx, cases[0][0], cases[0][1], n, part, dist) == cases[0][2]
    # scalar-range error in the middle of the pipeline (signed 16-bit windows: 2^256 - 1 leaves a final carry)
    with pkg.MsmContext((0,)) as ctx:
        ctx.set_option("window_bits", 16)
        ctx.set_window_shard(0, 1)
        m = 4096
        pts, sc = ora.gen_points(9, m), ora.gen_scalars(9, m)
        bad = bytearray(sc); bad[32 * 100:32 * 101] = b"\xff" * 32
        dp, dg, db = _dev(pts), _dev(sc), _dev(bytes(bad))
        torch.cuda.synchronize()
        pipe = pkg.ShardedPipeline(ctx, m, dist, depth=2)
        t0, t1 = pipe.submit(dp, dg), pipe.submit(dp, db)
        exp = ora.msm(pts, sc, c=16, threads=4)
        assert pipe.collect(t0) == exp
        with pytest.raises(pkg.MsmError) as e:
            pipe.collect(t1)
        assert e.value.code == -3
        assert pipe.collect(pipe.submit(dp, dg)) == exp


# ------------------------------------------------------------------ work-set ownership (te_msm.hip:580)
def test_worksets_owned_by_tickets_are_not_reused(pkg, ora):
    """every work set owned by a submitted MSM, then synchronous calls: they must not overwrite a work set whose ticket
    has not been collected (te_msm_run_device used to take work set 0 regardless)"""
    import torch
    K = pkg.WORKSETS
    data = []
    for i in range(K + 1):
        n = 20000 + 1000 * i
        pts, sc = ora.gen_points(800 + i, n), ora.gen_scalars(800 + i, n)
        data.append((_dev(pts), _dev(sc), n, ora.msm(pts, sc, threads=8), pts, sc))
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        t = [c.submit_device(d[0].data_ptr(), d[1].data_ptr(), d[2]) for d in data[:K]]
        last = data[K]
        with pytest.raises(pkg.MsmError) as e:                    # every set is owned
            c.run_device(last[0].data_ptr(), last[1].data_ptr(), last[2])
        assert e.value.code == -4
        part = torch.zeros(c.plan(last[2])[1] * 720, dtype=torch.uint8, device="cuda")
        with pytest.raises(pkg.MsmError) as e:
            c.partial_device(last[0].data_ptr(), last[1].data_ptr(), last[2], part.data_ptr())
        assert e.value.code == -4
        assert c.collect(t[0]) == data[0][3]                      # frees one work set
        # seven sets are still owned by tickets: the synchronous calls run on the free one
        assert c.run_device(last[0].data_ptr(), last[1].data_ptr(), last[2]) == last[3]
        assert c.run(last[4], last[5]) == last[3]
        for i in range(1, K):
            assert c.collect(t[i]) == data[i][3], i
        # profile switched on between submit and collect: the result still arrives
        t9 = c.submit_device(last[0].data_ptr(), last[1].data_ptr(), last[2])
        c.set_option("profile", 2)
        assert c.collect(t9) == last[3]
        c.set_option("profile", 0)


def test_finalize_folds_with_the_plan_that_made_the_rows(pkg, ora):
    """te_msm_finalize takes the digit form from the plan of the partial_device call, not from the option's current value;
    a window size that is not the plan's is refused"""
    import torch
    n = 30000
    pts, sc = ora.gen_points(55, n), ora.gen_scalars(55, n)
    exp = ora.msm(pts, sc, threads=8)
    dp, ds = _dev(pts), _dev(sc)
    with pkg.MsmContext((0,)) as c:
        c.set_option("window_bits", 12)
        c.set_option("signed_digits", 0)
        cb, W = c.plan(n)
        part = torch.zeros(W * 720, dtype=torch.uint8, device="cuda")
        c.partial_device(dp.data_ptr(), ds.data_ptr(), n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rows = part.cpu().numpy().tobytes()
        c.set_option("signed_digits", 1)                         # changed after the rows were made
        assert c.finalize(rows, cb, W) == exp
        with pytest.raises(pkg.MsmError) as e:
            c.finalize(rows, cb + 1, W)
        assert e.value.code == -4
        assert pkg.finalize_host(rows, cb, W, bucket_bits=cb) == exp
