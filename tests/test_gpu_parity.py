"""GPU parity tests: the HIP path (through the C-ABI of libtemsm.so) against the oracle, on the same
seeded inputs, bit-exact.  Run with `-m gpu` on an MI355X.  Nothing here reads /root/reference."""
import ctypes
import importlib
import os

import numpy as np
import pytest

from oracle.gen_golden import edge_scalars, make_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.MsmContext((0,))
    yield c
    c.close()


def _dev(buf: bytes):
    import torch
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).cuda()


# ------------------------------------------------------------------ stage verifiers
# (the reference's per-stage `debug` checks, submission.ts:892-1363)
@pytest.mark.parametrize("n,c,packed", [(1000, 8, 1), (5003, 13, 1), (70001, 16, 1), (70001, 16, 0), (5003, 13, 0)])
def test_stages_against_oracle(ctx, fpcheck, model, ora, n, c, packed):
    pts, sc = ora.gen_points(500 + n, n), ora.gen_scalars(500 + n, n)
    ctx.set_option("window_bits", c)
    ctx.set_option("sort_buckets", 1)
    ctx.set_option("packed_sort", packed)  # level-1 entries as one word (default where n <= 2^23) or as key + index (larger n)
    ctx.set_option("prezero", 0)           # keep counters and rows in the zeroed block for the checks below
    res = ctx.run(pts, sc)
    W, B = (256 + c - 1) // c, 1 << (c - 1)
    # K1a: records == the same limb code compiled for the host
    recs = ctx.debug_read("records", n * 128)
    for i in list(range(0, n, max(1, n // 97))) + [n - 1]:
        r = ctypes.create_string_buffer(128)
        fpcheck.fpc_prep_point(pts[64 * i:64 * i + 64], r)
        assert recs[128 * i:128 * i + 128] == r.raw, f"record {i}"
    # K1b: digits == decompose_scalars_signed (miscellaneous/utils.ts:52-95)
    nst = (n + 7) & ~7                                  # digit rows are padded to a multiple of 8 entries (digit 0)
    dig = np.frombuffer(ctx.debug_read("digits", W * nst * 2), dtype=np.uint16).reshape(W, nst)
    assert np.all(dig[:, n:] == B)
    dig = dig[:, :n]
    exp = ora.decompose_scalars_signed(sc, c)
    assert np.array_equal(dig.astype(np.uint32), exp)
    # K2 level 1: partitions of S = min(B, 256) buckets (order inside a partition is free)
    S = min(B, 256)
    P, logS = B // S, S.bit_length() - 1
    pstart = np.frombuffer(ctx.debug_read("part_start", W * P * 4), dtype=np.uint32).reshape(W, P)
    pcount = np.frombuffer(ctx.debug_read("part_count", W * P * 4), dtype=np.uint32).reshape(W, P)
    pkeys = np.frombuffer(ctx.debug_read("part_keys", W * nst * 2), dtype=np.uint16).reshape(W, nst)
    pidx = np.frombuffer(ctx.debug_read("part_idx", W * nst * 4), dtype=np.uint32).reshape(W, nst)
    for w in range(W):
        d = exp[w].astype(np.int64) - B
        bucket = np.abs(d) - 1
        nz = d != 0
        e_cnt = np.bincount(bucket[nz] >> logS, minlength=P)
        assert np.array_equal(pcount[w], e_cnt), f"window {w} partition counts"
        assert np.array_equal(pstart[w], np.concatenate([[0], np.cumsum(e_cnt)[:-1]])), f"window {w} partition starts"
        used = int(e_cnt.sum())
        idx, key = pidx[w][:used].astype(np.int64), pkeys[w][:used].astype(np.int64)
        assert np.array_equal(np.sort(idx), np.nonzero(nz)[0]), "entries are not a permutation of the non-zero digits"
        assert np.array_equal(bucket[idx] >> logS, np.repeat(np.arange(P), e_cnt)), "entry in the wrong partition"
        assert np.array_equal(key & 0x7FFF, bucket[idx] & (S - 1)), "key low bits"
        assert np.array_equal((key >> 15).astype(bool), d[idx] < 0), "sign bit"
    # K2 level 2: bucket_count/bucket_start == cpu_transpose's column pointers folded by sign (transpose.ts:14-62)
    cnt = np.frombuffer(ctx.debug_read("bucket_count", W * B * 4), dtype=np.uint32).reshape(W, B)
    start = np.frombuffer(ctx.debug_read("bucket_start", W * B * 4), dtype=np.uint32).reshape(W, B)
    srt = np.frombuffer(ctx.debug_read("sorted", W * n * 4), dtype=np.uint32).reshape(W, n)
    for w in range(W):
        d = exp[w].astype(np.int64) - B
        bucket = np.abs(d) - 1
        nz = d != 0
        e_cnt = np.bincount(bucket[nz], minlength=B)
        assert np.array_equal(cnt[w], e_cnt), f"window {w} counts"
        assert np.array_equal(start[w], np.concatenate([[0], np.cumsum(e_cnt)[:-1]])), f"window {w} starts"
        used = int(e_cnt.sum())
        ent = srt[w][:used]
        idx, neg = ent & 0x7FFFFFFF, ent >> 31
        assert np.array_equal(np.sort(idx), np.sort(np.nonzero(nz)[0])), "sorted is not a permutation of the non-zero digits"
        assert np.array_equal(bucket[idx], np.repeat(np.arange(B), e_cnt)), "entry in the wrong bucket"
        assert np.array_equal(neg.astype(bool), d[idx] < 0), "sign bit"
    # work segments: every bucket is cut into pieces of at most segment_len entries; the schedule is a permutation of
    # the segments in descending length
    seg_len = ctx.get_option("segment_len_used")
    nseg = int(np.frombuffer(ctx.debug_read("num_segments", 4), dtype=np.uint32)[0])
    per_bucket = np.maximum(1, -(-cnt.reshape(-1).astype(np.int64) // seg_len))
    assert nseg == int(per_bucket.sum())
    # segment ids: window k owns [k * capW, (k+1) * capW), capW = B + n // seg_len; ids are dense inside a level-1 partition,
    # in bucket order, and the ids a partition does not use are marked invalid (0xffffffff)
    cap_w = B + n // seg_len
    ids = W * cap_w
    seg_bucket_all = np.frombuffer(ctx.debug_read("seg_bucket", ids * 4), dtype=np.uint32)
    seg_lens_all = np.frombuffer(ctx.debug_read("seg_len", ids * 4), dtype=np.uint32)
    valid = seg_bucket_all != 0xFFFFFFFF
    assert np.array_equal(valid, seg_lens_all != 0xFFFFFFFF) and int(valid.sum()) == nseg
    seg_bucket, seg_lens = seg_bucket_all[valid], seg_lens_all[valid]
    assert np.array_equal(seg_bucket, np.repeat(np.arange(W * B), per_bucket)), "segments are not in bucket order"
    assert np.all(seg_bucket // B == np.nonzero(valid)[0] // cap_w), "segment id outside its window's range"
    assert np.array_equal(np.bincount(seg_bucket, weights=seg_lens, minlength=W * B).astype(np.int64), cnt.reshape(-1).astype(np.int64))
    assert seg_lens.max() <= seg_len
    order = np.frombuffer(ctx.debug_read("order", nseg * 4), dtype=np.uint32)
    assert np.array_equal(np.sort(order), np.nonzero(valid)[0]), "order is not a permutation of the valid segments"
    sizes = seg_lens_all[order]
    assert np.all(sizes[:-1] >= sizes[1:]), "order is not descending"
    # K3: a sample of bucket sums == affine sums of the model
    bk = ctx.debug_read("buckets", W * B * 144)
    P = model.P
    rinv = pow(1 << 261, -1, P)
    rng = np.random.default_rng(1)
    for w, b in [(0, 0), (W - 1, B - 1)] + [(int(rng.integers(W)), int(rng.integers(B))) for _ in range(6)]:
        raw = bk[(w * B + b) * 144:(w * B + b + 1) * 144]
        words = np.frombuffer(raw, dtype=np.uint32).reshape(4, 9)
        assert np.all(words[:, :8] < (1 << 29)), "limb class N violated"
        x, y, z, t = [sum(int(v) << (29 * i) for i, v in enumerate(words[k])) for k in range(4)]
        assert max(x, y, z, t) < 2 * P, "lazy bound < 2p violated"
        zi = pow(z * rinv % P, -1, P)
        got = (x * rinv * zi % P, y * rinv * zi % P)
        d = exp[w].astype(np.int64) - B
        e = model.ZERO
        for i in np.nonzero(np.abs(d) - 1 == b)[0]:
            p_i = model.xy_from_bytes(pts[64 * int(i):64 * int(i) + 64])
            e = model.add(e, model.neg(p_i) if d[i] < 0 else p_i)
        assert got == e, f"bucket ({w},{b})"
    # K4: partial rows -> product host tail == oracle; and the emulated rows agree after the tail
    assert res == ora.msm(pts, sc, threads=8)
    ctx.set_option("window_bits", 0)
    ctx.set_option("prezero", 1)
    ctx.set_option("packed_sort", 1)
    # with the default (the zeroed block is cleared behind the read-back) the same stages are refused, not read as zeros
    assert ctx.run(pts, sc) == res
    with pytest.raises(Exception):
        ctx.debug_read("bucket_count", 4)


# ------------------------------------------------------------------ end to end, golden fixtures
def test_wasm_golden_cases(ctx, wasm_golden, model):
    """the HIP path against outputs of the reference's own CPU MSM (Aleo WASM Address.msm, reference/reference.ts:29-39) -- up to
    the headline size n = 2^20, chain / harness-mode / random points; the large cases also with the 16-bit windows of BASELINE
    configs 2 and 3 (unsigned and signed digits) and from device-resident inputs"""
    import torch
    ctx.set_option("window_bits", 0)
    assert max(g["n"] for g in wasm_golden) == 1 << 20, "the reference-generated golden at the headline size is missing"
    for g in wasm_golden:
        pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
        want = (int(g["x"]), int(g["y"]))
        assert model.xy_from_bytes(ctx.run(pts, sc)) == want, g["name"]
        if g["n"] >= 65536:
            dp, ds = _dev(pts), _dev(sc)
            torch.cuda.synchronize()
            ctx.set_option("window_bits", 16)
            for signed in (1, 0):
                ctx.set_option("signed_digits", signed)
                assert model.xy_from_bytes(ctx.run_device(dp.data_ptr(), ds.data_ptr(), g["n"])) == want, (g["name"], signed)
                assert model.xy_from_bytes(ctx.run(pts, sc)) == want, (g["name"], signed)
            ctx.set_option("signed_digits", 1)
            ctx.set_option("window_bits", 0)
            del dp, ds
        elif g["n"] <= 4096:
            # round 5: the two plans whose window count changed (17 x 15 bits, 51 x 5 bits: ceil(255 / c) for signed digits) against
            # the reference's own outputs, edge scalars (0, 1, p - 1, ...) included
            for c in (15, 5):
                ctx.set_option("window_bits", c)
                assert ctx.plan(g["n"]) == (c, -(-255 // c))
                assert model.xy_from_bytes(ctx.run(pts, sc)) == want, (g["name"], c)
            ctx.set_option("window_bits", 0)


@pytest.mark.parametrize("c", [4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
def test_every_window_size(ctx, ora, c):
    n = 3000
    pts, sc = ora.gen_points(40 + c, n), ora.gen_scalars(40 + c, n)
    exp = ora.msm(pts, sc, threads=4)
    ctx.set_option("window_bits", c)
    for sort in (0, 1):
        ctx.set_option("sort_buckets", sort)
        assert ctx.run(pts, sc) == exp
    ctx.set_option("window_bits", 0)
    ctx.set_option("sort_buckets", 1)


# ------------------------------------------------------------------ unsigned windows, 2^c buckets
@pytest.mark.parametrize("c,n", [(4, 500), (8, 3000), (13, 20011), (15, 40000), (16, 70001)])
def test_unsigned_digits(pkg, model, ora, c, n):
    """option signed_digits = 0: plain windows (miscellaneous/utils.ts:34-50), bucket = digit - 1 of 2^c; same result as the
    signed path and as the oracle; digits, bucket counts and the pipelined / sharded tails agree"""
    import torch
    pts, sc = ora.gen_points(900 + c, n), ora.gen_scalars(900 + c, n)
    exp = ora.msm(pts, sc, threads=8)
    with pkg.MsmContext((0,)) as u:
        u.set_option("window_bits", c)
        u.set_option("signed_digits", 0)
        assert u.get_option("signed_digits") == 0
        u.set_option("prezero", 0)         # bucket_count is read back below
        assert u.run(pts, sc) == exp
        W, B = (256 + c - 1) // c, 1 << c
        nst = (n + 7) & ~7
        dig = np.frombuffer(u.debug_read("digits", W * nst * 2), dtype=np.uint16).reshape(W, nst)
        assert np.all(dig[:, n:] == 0)
        words = np.frombuffer(sc, dtype="<u8").reshape(n, 4)
        ints = [int(words[i, 0]) | int(words[i, 1]) << 64 | int(words[i, 2]) << 128 | int(words[i, 3]) << 192 for i in range(0, n, max(1, n // 211))]
        for j, i in enumerate(range(0, n, max(1, n // 211))):
            assert [int(dig[w, i]) for w in range(W)] == [(ints[j] >> (c * w)) & (B - 1) for w in range(W)]
        cnt = np.frombuffer(u.debug_read("bucket_count", W * B * 4), dtype=np.uint32).reshape(W, B)
        for w in (0, W // 2, W - 1):
            d = dig[w, :n].astype(np.int64)
            assert np.array_equal(cnt[w], np.bincount(d[d != 0] - 1, minlength=B))
        # pipelined tickets and the window-sharded tail use the same digit form
        dp, ds = _dev(pts), _dev(sc)
        torch.cuda.synchronize()
        t = u.submit_device(dp.data_ptr(), ds.data_ptr(), n)
        assert u.collect(t) == exp
        part = torch.zeros(W * 720, dtype=torch.uint8, device="cuda")
        u.partial_device(dp.data_ptr(), ds.data_ptr(), n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        rows = part.cpu().numpy().tobytes()
        assert u.finalize(rows, c, W) == exp
        assert pkg.finalize_host(rows, c, W, bucket_bits=c) == exp
    # unsigned windows accept any 256-bit scalar (no final carry): check against plain double-and-add
    pts4 = ora.gen_points(3, 4)
    ks = [(1 << 256) - 1, (1 << 255) + 12345, 1, 0]
    with pkg.MsmContext((0,)) as u:
        u.set_option("window_bits", c)
        u.set_option("signed_digits", 0)
        got = u.run(pts4, model.scalars_to_bytes(ks))
    pl = [model.xy_from_bytes(pts4[64 * i:64 * i + 64]) for i in range(4)]
    assert model.xy_from_bytes(got) == model.msm_naive(pl, ks)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 1000, 4097, 65535, 65537, 100003])
def test_ragged_sizes(ctx, ora, n):
    pts, sc = ora.gen_points(n, n), ora.gen_scalars(n, n)
    assert ctx.run(pts, sc) == ora.msm(pts, sc, threads=8)


@pytest.mark.parametrize("seg_len", [1, 3, 16, 64, 100000])
def test_segment_lengths(ctx, ora, seg_len):
    """bucket splitting: any segment length gives the same point (the top window of a 253-bit scalar has ~219-entry buckets at n = 2^20)"""
    n = 40000
    pts, sc = ora.gen_points(61, n), ora.gen_scalars(61, n)
    exp = ora.msm(pts, sc, threads=8)
    ctx.set_option("segment_len", seg_len)
    for c in (8, 13):
        ctx.set_option("window_bits", c)
        assert ctx.run(pts, sc) == exp
    ctx.set_option("window_bits", 0)
    ctx.set_option("segment_len", 0)


def test_empty_input(ctx):
    assert ctx.run(b"", b"") == bytes(32) + (1).to_bytes(32, "little")


def test_edge_scalars_and_harness_mode(ctx, model, ora):
    n = 4096
    sc = model.scalars_to_bytes(edge_scalars(99, n))
    for pts in (ora.gen_points(99, n), ora.gen_points_fixed(n)):
        for c in (0, 16, 13):
            ctx.set_option("window_bits", c)
            assert ctx.run(pts, sc) == ora.msm(pts, sc, threads=8)
    ctx.set_option("window_bits", 0)
    # all scalars zero -> identity; all scalars equal -> one giant bucket per window (thread-per-bucket worst case)
    pts = ora.gen_points(5, 2048)
    assert ctx.run(pts, bytes(32 * 2048)) == bytes(32) + (1).to_bytes(32, "little")
    same = model.scalars_to_bytes([0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % model.P] * 2048)
    assert ctx.run(pts, same) == ora.msm(pts, same, threads=8)


def test_exceptional_points(ctx, pkg, model, ora):
    """The cases a unified addition must get right without a branch: the neutral element, the points of order 2 and 4 as INPUTS,
    P, -P and P again in one bucket (a sum through the neutral element and a doubling), subgroup points shifted by low-order
    points -- through the first-entry conversion, the mixed addition, the folds and the tail, for window sizes that put them in
    one bucket or spread them; host buffers, device buffers and unsigned digits."""
    from oracle.gen_golden import special_point_inputs
    for seed, n in ((5, 32), (6, 200), (7, 5000)):
        pts, ks = special_point_inputs(seed, n)
        pb, sb = model.points_to_bytes(pts), model.scalars_to_bytes(ks)
        exp = ora.msm(pb, sb, threads=8)
        if n <= 200:
            assert model.xy_from_bytes(exp) == model.msm_naive(pts, ks)
        for c in (0, 4, 9, 13, 16):
            ctx.set_option("window_bits", c)
            assert ctx.run(pb, sb) == exp, (n, c)
        ctx.set_option("window_bits", 0)
        dp, ds = _dev(pb), _dev(sb)
        import torch
        torch.cuda.synchronize()
        assert ctx.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
        ctx.set_option("signed_digits", 0)
        assert ctx.run(pb, sb) == exp
        ctx.set_option("signed_digits", 1)
    # nothing but low-order points and the neutral element: every bucket sum is a low-order point
    p = model.P
    i4 = model.sqrt_mod_p(p - 1)
    low = [(0, 1), (0, p - 1), (i4, 0), (p - i4, 0)] * 16
    ks = model.gen_scalars(9, len(low))
    pb, sb = model.points_to_bytes(low), model.scalars_to_bytes(ks)
    assert model.xy_from_bytes(ctx.run(pb, sb)) == model.msm_naive(low, ks)


def test_non_canonical_coordinates_are_reduced(ctx, model, ora):
    """x + p and y + p are other 256-bit names of the same field elements."""
    n = 64
    pts = ora.gen_points(8, n)
    sc = ora.gen_scalars(8, n)
    shifted = b"".join(model.le32(int.from_bytes(pts[32 * i:32 * i + 32], "little") + (model.P if i % 3 == 0 else 0)) for i in range(2 * n))
    assert ctx.run(shifted, sc) == ora.msm(pts, sc)


def test_final_carry_is_an_error(ctx, pkg, model, ora):
    pts = ora.gen_points(3, 4)
    sc = model.scalars_to_bytes([1, 2, (1 << 256) - 1, 3])
    ctx.set_option("window_bits", 16)
    with pytest.raises(pkg.MsmError) as e:
        ctx.run(pts, sc)
    assert e.value.code == -3 and "final carry" in str(e.value)
    ctx.set_option("window_bits", 0)
    assert ctx.run(pts, model.scalars_to_bytes([1, 2, 3, 4])) == ora.msm(pts, model.scalars_to_bytes([1, 2, 3, 4]))


def test_device_resident_inputs(ctx, ora):
    import torch
    n = 50000
    pts, sc = ora.gen_points(77, n), ora.gen_scalars(77, n)
    dp, ds = _dev(pts), _dev(sc)
    torch.cuda.synchronize()
    assert ctx.run_device(dp.data_ptr(), ds.data_ptr(), n) == ora.msm(pts, sc, threads=8)


def test_compute_msm_mirror(pkg, model, ora):
    pts, sc = ora.gen_points(123, 777), ora.gen_scalars(123, 777)
    r = pkg.compute_msm(pts, sc, log_result=False)
    assert (r["x"], r["y"]) == model.xy_from_bytes(ora.msm(pts, sc))


# ------------------------------------------------------------------ window sharding on one GPU
@pytest.mark.parametrize("world", [2, 4, 8])
def test_window_shards_on_one_gpu(pkg, ora, world):
    import torch
    n = 30000
    pts, sc = ora.gen_points(31, n), ora.gen_scalars(31, n)
    dp, ds = _dev(pts), _dev(sc)
    exp = ora.msm(pts, sc, threads=8)
    rows, cW = [], None
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            part = torch.zeros(W * 720, dtype=torch.uint8, device="cuda")
            c.partial_device(dp.data_ptr(), ds.data_ptr(), n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            rows.append(part.cpu().numpy().tobytes())
            cW = (cbits, W)
            if r == world - 1:
                assert c.finalize(pkg.merge_partials(rows, W, world), cbits, W) == exp
    assert pkg.finalize_host(pkg.merge_partials(rows, cW[1], world), *cW) == exp


@pytest.mark.parametrize("world,count,n,opts", [(1, 3, 5000, {}), (4, 5, 30000, {}), (8, 8, 20000, {"window_bits": 16}),
                                                (2, 2, 3000, {"signed_digits": 0, "window_bits": 9}), (3, 4, 777, {"segment_len": 5})])
def test_batches_of_window_sharded_msms(pkg, ora, world, count, n, opts):
    """te_msm_partial_device_batch: `count` MSMs with different inputs share one launch sequence per rank; every MSM's rows,
    merged over the ranks, finalize to the oracle's result -- also when a scalar buffer repeats, and batch 1 = the plain call"""
    import torch
    ins = [(ora.gen_points(900 + m, n), ora.gen_scalars(950 + m, n)) for m in range(count)]
    ins[-1] = (ins[-1][0], ins[0][1])                                    # same scalars, other points
    dev = [(_dev(p), _dev(s)) for p, s in ins]
    exp = [ora.msm(p, s, threads=8) for p, s in ins]
    per_rank = []
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            for k, v in opts.items():
                c.set_option(k, v)
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            part = torch.zeros(count * W * 720, dtype=torch.uint8, device="cuda")
            c.partial_device_batch([p.data_ptr() for p, _ in dev], [s.data_ptr() for _, s in dev], n, part.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            c.partial_wait(0)
            per_rank.append(part.cpu().numpy().tobytes())
            if r == 0:                                                   # a batch of one is te_msm_partial_device
                one = torch.zeros(W * 720, dtype=torch.uint8, device="cuda")
                c.partial_device_batch([dev[1][0].data_ptr()], [dev[1][1].data_ptr()], n, one.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                blk = W * 720
                mine = [w for w in range(W) if w % world == 0]
                got1, ref1 = one.cpu().numpy().tobytes(), per_rank[0][blk:2 * blk]
                # rows are lazily reduced sums: compare as points through the host tail, window by window
                for w in mine:
                    z = bytearray(blk); z[w * 720:(w + 1) * 720] = got1[w * 720:(w + 1) * 720]
                    y = bytearray(blk); y[w * 720:(w + 1) * 720] = ref1[w * 720:(w + 1) * 720]
                    bb = cbits - 1 if c.get_option("signed_digits") else cbits
                    assert pkg.finalize_host(bytes(z), cbits, W, bb) == pkg.finalize_host(bytes(y), cbits, W, bb)
    bb = cbits - 1 if opts.get("signed_digits", 1) else cbits
    blk = W * 720
    for m in range(count):
        merged = pkg.merge_partials([rows[m * blk:(m + 1) * blk] for rows in per_rank], W, world)
        assert pkg.finalize_host(merged, cbits, W, bb) == exp[m], f"MSM {m} of the batch"


@pytest.mark.parametrize("world,count", [(8, 8), (4, 4)])
def test_full_size_window_sharded_batches(pkg, ora, world, count):
    """BASELINE config 4 at its full size, against the oracle: n = 2^20, the D ranks' window shards run one after another on
    the one GPU through te_msm_partial_device_batch (a batch of D MSMs per launch sequence, as bench.py --gpus D does), the
    rows of every MSM merged with te_msm_finalize_gathered.  The batch mixes shared and distinct point buffers: two point
    sets, `count` different scalar sets (a point buffer named by several MSMs of a call is converted once)."""
    import torch
    n = 1 << 20
    psets = [pkg.synth_inputs(0x5EED0040 + k, n, scalars=False)[0] for k in range(2)]
    ssets = [ora.gen_scalars(0x5EED0050 + m, n) for m in range(count)]
    dpts = [_dev(p) for p in psets]
    dscs = [_dev(sc) for sc in ssets]
    which = [0 if m % 3 != 1 else 1 for m in range(count)]                 # point set of MSM m: 0 1 0 0 1 0 0 1
    exp = [ora.msm(psets[which[m]], ssets[m], c=16, threads=16) for m in range(count)]
    torch.cuda.synchronize()
    gathered = None
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            c.set_option("window_bits", 16)
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            blk = W * 720
            if gathered is None:
                gathered = torch.zeros(world, count, blk, dtype=torch.uint8)
            part = torch.zeros(count * blk, dtype=torch.uint8, device="cuda")
            c.partial_device_batch([dpts[which[m]].data_ptr() for m in range(count)], [d.data_ptr() for d in dscs], n, part.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            c.partial_wait(0)
            gathered[r] = part.cpu().view(count, blk)
    for m in range(count):
        mine = gathered[:, m, :].contiguous()                              # [rank][W rows]: the layout an all-gather delivers
        assert pkg.finalize_gathered(mine.data_ptr(), world, cbits, W) == exp[m], f"MSM {m} of the batch"


def test_pipelined_full_size(pkg, ora):
    """n = 2^20, eight MSMs in flight on eight work sets (two alternating input sets), three rounds: the launch sequences
    overlap on the device for real at this size; every result against the oracle"""
    import torch
    n = 1 << 20
    ins = [pkg.synth_inputs(0x5EED0080 + k, n) for k in range(2)]
    exp = [ora.msm(p, s, c=16, threads=16) for p, s in ins]
    dev = [(_dev(p), _dev(s)) for p, s in ins]
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        for _ in range(3):
            tickets = [c.submit_device(dev[i & 1][0].data_ptr(), dev[i & 1][1].data_ptr(), n) for i in range(pkg.WORKSETS)]
            assert [c.collect(t) == exp[i & 1] for i, t in enumerate(tickets)] == [True] * pkg.WORKSETS


def test_batch_argument_checks(pkg, ora):
    import torch
    n = 100
    dp, ds = _dev(ora.gen_points(1, n)), _dev(ora.gen_scalars(1, n))
    with pkg.MsmContext((0,)) as c:
        _, W = c.plan(n)
        part = torch.zeros(9 * W * 720, dtype=torch.uint8, device="cuda")
        with pytest.raises(pkg.MsmError):
            c.partial_device_batch([dp.data_ptr()] * 9, [ds.data_ptr()] * 9, n, part.data_ptr())
        with pytest.raises(pkg.MsmError):
            c.partial_device_batch([dp.data_ptr(), 0], [ds.data_ptr()] * 2, n, part.data_ptr())
        bad = bytearray(ora.gen_scalars(2, n)); bad[-32:] = b"\xff" * 32          # a scalar above the decomposition's range
        db = _dev(bytes(bad))
        c.partial_device_batch([dp.data_ptr()] * 3, [ds.data_ptr(), db.data_ptr(), ds.data_ptr()], n, part.data_ptr())
        with pytest.raises(pkg.MsmError):
            c.partial_wait(0)


def test_multi_device_context_same_gpu(pkg, ora):
    """n_dev = 2 with the same device id twice: exercises the in-process window sharding."""
    n = 20000
    pts, sc = ora.gen_points(32, n), ora.gen_scalars(32, n)
    exp = ora.msm(pts, sc, threads=8)
    dp, ds = _dev(pts), _dev(sc)
    import torch
    torch.cuda.synchronize()
    with pkg.MsmContext((0, 0)) as c:
        assert c.get_option("num_devices") == 2
        assert c.run(pts, sc) == exp                                   # host buffers: every device uploads for itself
        assert c.get_option("peer_copies") == 0
        # device-resident inputs live on the first device: the second "device" gets them through hipMemcpyPeerAsync -- the
        # copies are issued and ordered in front of its kernels also when both ids name one physical GPU (the only form a
        # one-GPU box can run: whether the path is right across two physical devices stays unmeasured)
        # The inputs travel as a scatter + all-gather (SURVEY 8e "Inputs"): device i >= 1 pulls slice i from the source, then the
        # other slices from their holders -- 2 (D - 1) D copies per call (points and scalars), every device but the first receives
        # all n points once, and only 2 (D - 1) / D of the input leaves the first device's memory.
        for k in range(3):
            assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
            assert c.get_option("peer_copies") == 4 * (k + 1) and c.get_option("peer_bytes") == n * 96 * (k + 1)
    with pkg.MsmContext((0, 0, 0)) as c3:                               # 16 windows over three "devices": 6 + 5 + 5; ragged slices (6667 + 6667 + 6666)
        assert c3.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
        assert c3.get_option("peer_copies") == 12 and c3.get_option("peer_bytes") == 2 * n * 96
    with pkg.MsmContext((0,) * 8) as c8:                                # two windows per device; tiny n: some slices are empty
        assert c8.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
        assert c8.get_option("peer_bytes") == 7 * n * 96
        assert c8.run_device(dp.data_ptr(), ds.data_ptr(), 5) == ora.msm(pts[:64 * 5], sc[:32 * 5])


@pytest.mark.parametrize("ids", [(0, 0), (0, 0, 0, 0), (0,) * 8])
def test_multi_device_host_point_shards(pkg, model, ora, wasm_golden, ids):
    """compute_msm(Buffer, Buffer) on D devices (te_msm_run on an n_dev = D context, SURVEY 8e / README.md:551): points and
    scalars cut into D slices, one upload thread per device, all windows on every slice, rows summed in the host tail.
    Device ids repeat (a one-GPU box): D threads, D work sets, D staging areas on the one GPU.  Against the oracle AND the
    reference's own outputs (WASM goldens), both digit forms, ragged and tiny n, slices in pieces, a scalar-range error in a
    late slice."""
    D = len(ids)
    with pkg.MsmContext(ids) as c:
        assert c.get_option("num_devices") == D and c.get_option("host_shard_min") == 4096
        c.set_option("host_shard_min", 1)                                # every device gets a slice, however small
        for g in wasm_golden:
            if g["n"] > 65536:
                continue
            pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
            for signed in (1, 0):
                c.set_option("signed_digits", signed)
                assert model.xy_from_bytes(c.run(pts, sc)) == (int(g["x"]), int(g["y"])), (g["name"], signed)
        c.set_option("signed_digits", 1)
        for seed, n, chunks in [(51, 100003, 0), (52, 7, 0), (53, D, 0), (54, 300007, 2), (55, 70001, 3)]:
            pts, sc = ora.gen_points(seed, n), ora.gen_scalars(seed, n)
            c.set_option("host_chunks", chunks)
            assert c.run(pts, sc) == ora.msm(pts, sc, threads=8), (n, chunks)
        c.set_option("host_chunks", 0)
        n = 100003
        pts, sc = ora.gen_points(56, n), ora.gen_scalars(56, n)
        bad = bytearray(sc); bad[32 * (n - 3):32 * (n - 3) + 32] = b"\xff" * 32          # lands in the last device's slice
        c.set_option("window_bits", 16)                                  # 16 x 16 bits: 2^256 - 1 leaves a final carry
        with pytest.raises(pkg.MsmError) as e:
            c.run(pts, bytes(bad))
        assert e.value.code == -3
        c.set_option("window_bits", 0)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=8)              # the context is usable afterwards
        c.set_option("host_shard_min", 4096)                             # default: small inputs use fewer devices
        pts, sc = ora.gen_points(57, 5000), ora.gen_scalars(57, 5000)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=8)
        assert c.run(b"", b"") == bytes(32) + b"\x01" + bytes(31)
        # device-resident inputs on the same context: window shards with peer copies, as before
        dp, ds = _dev(pts), _dev(sc)
        import torch
        torch.cuda.synchronize()
        assert c.run_device(dp.data_ptr(), ds.data_ptr(), 5000) == ora.msm(pts, sc, threads=8)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=8)              # and point shards again (the window shards were restored in between)


def test_multi_device_full_size_against_the_reference(pkg, model, ora, wasm_golden):
    """n = 2^20 from host buffers over eight "devices" (2^17 points each, 15-bit windows) and over four: equal to the point the
    reference's own CPU MSM (Aleo WASM) returned for these inputs, and to the oracle"""
    g = [x for x in wasm_golden if x["n"] == 1 << 20]
    if not g:
        pytest.skip("no reference-generated golden at n = 2^20 in tests/golden/msm_wasm_golden.json")
    g = g[0]
    pts, sc = make_inputs(g["seed"], g["n"], g["mode"])
    for D in (8, 4):
        with pkg.MsmContext((0,) * D) as c:
            assert model.xy_from_bytes(c.run(pts, sc)) == (int(g["x"]), int(g["y"])), D


# ------------------------------------------------------------------ BASELINE.json's full size
def test_full_size_2_20_general_sort_entries(pkg, ora):
    """n = 2^20 with the sort's level-1 entries in the general form (u16 key + u32 index; the default packs them into one
    word where n <= 2^23): the form every n above 2^23 runs with, at the headline size, both digit forms, several in flight"""
    import torch
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0xFACE, n)
    exp = ora.msm(pts, sc, threads=16)
    dp, ds = _dev(pts), _dev(sc)
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        c.set_option("packed_sort", 0)
        assert c.get_option("packed_sort") == 0
        for signed in (1, 0):
            c.set_option("signed_digits", signed)
            ts = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(3)]
            assert [c.collect(t) for t in ts] == [exp] * 3, signed
        c.set_option("signed_digits", 1)
        assert c.run(pts, sc) == exp


def test_full_size_2_20(ctx, model, ora):
    n = 1 << 20
    pts, sc = ora.gen_points(0x5EED0014, n), ora.gen_scalars(0x5EED0014, n)
    ctx.set_option("window_bits", 16)
    got = ctx.run(pts, sc)
    assert got == ora.msm(pts, sc, c=16, threads=16)            # bit-exact against the oracle at full size
    # linearity: MSM(P, a) + MSM(P, b) == MSM(P, a + b mod l) for subgroup points
    a = np.frombuffer(sc, dtype=np.uint8)
    sc2 = ora.gen_scalars(0x5EED0015, n)
    L = model.L
    s1 = [int.from_bytes(sc[32 * i:32 * i + 32], "little") for i in range(0, n, 1)]
    s2 = [int.from_bytes(sc2[32 * i:32 * i + 32], "little") for i in range(0, n, 1)]
    s3 = model.scalars_to_bytes([(x + y) % L for x, y in zip(s1, s2)])
    r1, r2, r3 = (model.xy_from_bytes(ctx.run(pts, s)) for s in (sc, sc2, s3))
    assert model.add(r1, r2) == r3
    # harness mode (one point replicated, ui/AllBenchmarks.tsx:105-112): sum k_i * P = (sum k_i mod l) * P
    fixed = ora.gen_points_fixed(n)
    rh = model.xy_from_bytes(ctx.run(fixed, sc))
    assert rh == model.scalar_mul(sum(s1) % L, (model.HX, model.HY))
    ctx.set_option("window_bits", 0)


def test_beyond_the_harness_sizes_2_22(pkg, model, ora):
    """n = 2^22 (four times the largest harness case): 64-bit offsets, more chunks / slices / segments than any other test;
    bit-exact against the oracle, for the harness's fixed point against the closed form, and pipelined with 2^20 MSMs"""
    import torch
    n = 1 << 22
    pts, sc = pkg.synth_inputs(0x5EED0016, n)
    exp = ora.msm(pts, sc, c=16, threads=16)
    with pkg.MsmContext((0,)) as big:
        assert big.run(pts, sc) == exp
        dp, ds = _dev(pts), _dev(sc)
        torch.cuda.synchronize()
        small_n = 1 << 20
        exp_small = ora.msm(pts[:64 * small_n], sc[:32 * small_n], c=16, threads=16)
        t0 = big.submit_device(dp.data_ptr(), ds.data_ptr(), n)
        t1 = big.submit_device(dp.data_ptr(), ds.data_ptr(), small_n)       # a prefix of the same buffers
        t2 = big.submit_device(dp.data_ptr(), ds.data_ptr(), n)
        assert big.collect(t0) == exp and big.collect(t1) == exp_small and big.collect(t2) == exp
        fixed, _ = pkg.synth_inputs(0, n, fixed_point=True, scalars=False)
        ks = np.frombuffer(sc, dtype="<u8").reshape(n, 4).astype(object)
        total = int(ks[:, 0].sum()) + (int(ks[:, 1].sum()) << 64) + (int(ks[:, 2].sum()) << 128) + (int(ks[:, 3].sum()) << 192)
        assert model.xy_from_bytes(big.run(fixed, sc)) == model.scalar_mul(total % model.L, (model.HX, model.HY))


def test_zprize_vectors_if_supplied(ctx, pkg, kats):
    """test-data/testCases.ts:11-52: replays the official vectors when TE_ZPRIZE_DATA points at them."""
    import importlib
    root = os.environ.get("TE_ZPRIZE_DATA")
    if not root:
        pytest.skip("official ZPrize input files are not in the reference tree (README.md:22-33)")
    td = importlib.import_module("webgpu-msm-twisted-edwards_amd.testdata")
    for pw, e in kats["zprize_expected"].items():
        pp, sp = os.path.join(root, "points", f"{pw}-power-points.txt"), os.path.join(root, "scalars", f"{pw}-power-scalars.txt")
        if os.path.exists(pp):
            pts, sc = td.load_test_case(pp, sp)
            out = ctx.run(pts, sc)
            assert (int.from_bytes(out[:32], "little"), int.from_bytes(out[32:], "little")) == (int(e["x"]), int(e["y"]))


def test_full_benchmarks_protocol(pkg, ora):
    """the reference's benchmark loop (full_benchmarks.ts:6-163) over compute_msm: table shape, and every run is a real MSM"""
    import io
    fb = importlib.import_module(pkg.__name__ + ".full_benchmarks")
    buf = io.StringIO()
    res = fb.run([10, 12], None, num_runs=2, delay_ms=1, out=buf)
    text = buf.getvalue()
    assert "| MSM size | 1st run | Run 1 | Run 2 | Average (incl 1st) | Average (excl 1st) |" in text
    assert "| 2^10 |" in text and "| 2^12 |" in text
    for power in (10, 12):
        r = res[power]
        assert len(r["subsequent_runs"]) == 2 and r["first_run_elapsed"] > 0
        assert abs(r["full_average"] - (r["first_run_elapsed"] + sum(r["subsequent_runs"])) / 3) < 1e-9
    # the harness's CSV export (ui/CSVExportButton.tsx:8-23, rows of ui/AllBenchmarks.tsx:45-52): quoted cells, one row per timed call
    lines = fb.to_csv(fb.csv_rows(res)).split("\n")
    assert lines[0] == '"InputSize","MSM Func","Time (MS)"' and len(lines) == 1 + 2 * 3
    assert lines[1].startswith('"10","Submission","') and lines[4].startswith('"12","Submission","')
    pts, sc = pkg.synth_inputs(0x5EED0000 + 10, 1 << 10)
    out = pkg.compute_msm(pts, sc, log_result=False)
    assert out["x"] == int.from_bytes(ora.msm(pts, sc)[:32], "little")


def test_node_compute_msm_entry_point(pkg, model, ora, tmp_path):
    """The reference's own entry point, compute_msm(bufferPoints, bufferScalars) (submission.ts:73-78), through
    the N-API addon, against the oracle and a WASM golden case."""
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
    for seed, n, mode in [(0x5EED03E8, 1000, "chain"), (0x5EED0040, 64, "edge"), (4242, 70000, "chain")]:
        pts, sc = make_inputs(seed, n, mode)
        (tmp_path / "p.bin").write_bytes(pts)
        (tmp_path / "s.bin").write_bytes(sc)
        r = subprocess.run([node, os.path.join(js, "run_msm.js"), str(tmp_path / "p.bin"), str(tmp_path / "s.bin")],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        out = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert "x" in out, out
        assert (int(out["x"]), int(out["y"])) == model.xy_from_bytes(ora.msm(pts, sc, threads=8))


def _node_js_dir():
    import shutil
    import subprocess
    node = shutil.which("node")
    if not node:
        pytest.skip("node is not installed on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    js = os.path.join(root, "webgpu-msm-twisted-edwards_amd", "js")
    if not os.path.exists(os.path.join(js, "te_msm_napi.node")):
        subprocess.check_call(["make", "-C", js, "-s"])
    return node, js


def test_node_promises_in_flight_and_device_list(pkg, model, ora, tmp_path):
    """From the reference's host language: four compute_msm promises in flight at n = 2^18 overlap on the engine's work sets
    (te_msm_submit_async / ticket_wait / collect under the addon) -- all four equal the oracle and were in flight together (the
    addon's own count; the timings are printed, not asserted); setDevices([0, 0]) / TE_MSM_DEVICES shard one call over "two devices"
    with the same result"""
    import json
    import subprocess
    node, js = _node_js_dir()
    n = 1 << 18
    pts, sc = ora.gen_points(4343, n), ora.gen_scalars(4343, n)
    (tmp_path / "p.bin").write_bytes(pts)
    (tmp_path / "s.bin").write_bytes(sc)
    exp = model.xy_from_bytes(ora.msm(pts, sc, threads=8))

    def run(args, env=None):
        r = subprocess.run([node, os.path.join(js, "run_concurrent.js"), str(tmp_path / "p.bin"), str(tmp_path / "s.bin")] + args,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
        out = json.loads(r.stdout.decode().strip().splitlines()[-1])
        assert "x" in out, (out, r.stderr.decode()[-2000:])
        assert (int(out["x"]), int(out["y"])) == exp and out["all_equal"]
        return out

    out = run(["4"])
    # the timings are RECORDED, not asserted (round-5 verdict: one slow box must not take the parity suite down -- this very line went
    # red on a box whose copy engines were being brought up); that the four calls overlap is checked structurally: the addon saw
    # four tickets in flight at a submit, i.e. all four were on the engine's work sets before the first one was collected
    print("node: single %.3f ms, four in flight %.3f ms (not asserted)" % (out["single_ms"], out["concurrent_ms"]))
    if out["concurrent_ms"] >= 4 * out["single_ms"]:
        import warnings
        warnings.warn("four compute_msm promises in flight took %.3f ms against %.3f ms for one call: no overlap visible in the timing on this box" % (out["concurrent_ms"], out["single_ms"]))
    (tmp_path / "node_in_flight.json").write_text(json.dumps(out))
    assert out["devices"] == [0]
    assert out["stats"]["maxInFlight"] == 4, out["stats"]
    assert out["stats"]["boundJobs"] == 0
    out = run(["2", "0,0"])
    assert out["devices"] == [0, 0]
    out = run(["1"], env=dict(os.environ, TE_MSM_DEVICES="0,0,0"))
    assert out["devices"] == [0, 0, 0]


def test_pipelined_submit_collect(pkg, ora):
    """MSMs in flight on rotating work sets: results in submission order, each equal to the oracle; protocol errors are
    reported"""
    import torch
    sizes = [30000, 70000, 5000, 30000, 1000, 40000, 65536, 257, 12345, 50000, 3, 20000]
    K = pkg.WORKSETS
    cases = [(11 + i, sizes[i % len(sizes)]) for i in range(K + 2)]
    data = []
    for seed, n in cases:
        pts, sc = ora.gen_points(seed, n), ora.gen_scalars(seed, n)
        data.append((_dev(pts), _dev(sc), n, ora.msm(pts, sc, threads=8)))
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        sub = lambda i: c.submit_device(data[i][0].data_ptr(), data[i][1].data_ptr(), data[i][2])
        t = [sub(i) for i in range(K)]
        with pytest.raises(pkg.MsmError):
            sub(K)                                                                            # every work set is busy
        assert c.get_option("in_flight") == K
        assert c.collect(t[1]) == data[1][3]                                                  # any order (round 4)
        with pytest.raises(pkg.MsmError):
            c.collect(t[1])                                                                   # a ticket is consumed once
        with pytest.raises(pkg.MsmError):
            c.collect(10 ** 6)                                                                # never handed out
        t.append(sub(K))                                                                      # takes the work set ticket 1 left
        assert c.collect(t[0]) == data[0][3]
        t.append(sub(K + 1))
        for i in reversed(range(2, K + 2)):
            c.ticket_wait(t[i])
            assert c.collect(t[i]) == data[i][3]
        assert c.get_option("in_flight") == 0
        assert c.run_device(data[0][0].data_ptr(), data[0][1].data_ptr(), data[0][2]) == data[0][3]


def test_pipelined_host_submit(pkg, model, ora):
    """te_msm_submit: the pipelined form for HOST buffers (what concurrent compute_msm promises map onto): several in flight
    on their own work sets and staging areas, inputs released right after the call, collected in any order; mixed with
    device-resident tickets; a scalar-range error belongs to its ticket only; both digit forms; pieces"""
    import torch
    cases = [(901, 70001), (902, 1000), (903, 300000), (904, 65536), (905, 3), (906, 150000)]
    with pkg.MsmContext((0,)) as c:
        for signed in (1, 0):
            c.set_option("signed_digits", signed)
            tickets, exp = [], []
            for seed, n in cases:
                pts, sc = bytearray(ora.gen_points(seed, n)), bytearray(ora.gen_scalars(seed, n))
                exp.append(ora.msm(bytes(pts), bytes(sc), threads=8))
                tickets.append(c.submit(bytes(pts), bytes(sc)))
                pts[:] = b"\xff" * len(pts); sc[:] = b"\xff" * len(sc)          # the call has copied them out
            for i in (3, 0, 5, 1, 4, 2):
                assert c.collect(tickets[i]) == exp[i], (signed, i)
        c.set_option("signed_digits", 1)
        n = 100003
        pts, sc = ora.gen_points(907, n), ora.gen_scalars(907, n)
        want = ora.msm(pts, sc, threads=8)
        bad = bytearray(sc); bad[32 * (n - 2):32 * (n - 2) + 32] = b"\xff" * 32
        dp, ds = _dev(pts), _dev(sc)
        torch.cuda.synchronize()
        c.set_option("window_bits", 16)                                        # 16 x 16 bits: 2^256 - 1 leaves a final carry
        for chunks in (0, 1, 3):
            c.set_option("host_chunks", chunks)
            t1, t2, t3, t4 = c.submit(pts, sc), c.submit(pts, bytes(bad)), c.submit_device(dp.data_ptr(), ds.data_ptr(), n), c.submit(pts, sc)
            assert c.run(pts, sc) == want                                      # a synchronous call beside four tickets
            assert c.collect(t4) == want
            with pytest.raises(pkg.MsmError) as e:
                c.collect(t2)
            assert e.value.code == -3
            assert c.collect(t3) == want and c.collect(t1) == want
        c.set_option("host_chunks", 0)
        c.set_option("window_bits", 0)
        # capacity: WORKSETS tickets, then ESTATE until one is collected
        ts = [c.submit(pts, sc) for _ in range(pkg.WORKSETS)]
        with pytest.raises(pkg.MsmError):
            c.submit(pts, sc)
        assert all(c.collect(t) == want for t in ts)


def test_trim_gives_device_memory_back(pkg, ora):
    """te_msm_trim frees the buffers of idle work sets (staging areas included), skips sets owned by a ticket, and the next
    MSM simply allocates again"""
    import torch
    n = 50000
    pts, sc = ora.gen_points(31, n), ora.gen_scalars(31, n)
    exp = ora.msm(pts, sc, threads=8)
    with pkg.MsmContext((0,)) as c:
        assert c.get_option("device_bytes") == 0
        ts = [c.submit(pts, sc) for _ in range(3)]
        full = c.get_option("device_bytes")
        assert full > 3 * n * 96
        assert c.trim(0) == 0 and c.get_option("device_bytes") == full         # all three are owned by tickets
        assert c.collect(ts[0]) == exp and c.collect(ts[2]) == exp
        assert c.trim(1) == 1                                                   # set 2 is idle now; set 0 is kept, set 1 still owned
        assert full * 0.6 < c.get_option("device_bytes") < full
        assert c.collect(ts[1]) == exp
        assert c.trim(0) == 2 and c.get_option("device_bytes") == 0
        assert c.run(pts, sc) == exp and c.get_option("device_bytes") > 0
        dp, ds = _dev(pts), _dev(sc)
        torch.cuda.synchronize()
        assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
        c.trim(0)
        assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp


def test_many_parts_per_bucket(ctx, ora):
    """segment length 1 with few buckets: every bucket has thousands of parts, more runs of the giant-bucket combine than buckets
    (found by tools/soak.py: the chunk list was sized by the bucket count)"""
    n = 158508
    pts, sc = ora.gen_points(5, n), ora.gen_scalars(5, n)
    ctx.set_option("window_bits", 7)
    ctx.set_option("segment_len", 1)
    try:
        assert ctx.run(pts, sc) == ora.msm(pts, sc, threads=8)
    finally:
        ctx.set_option("window_bits", 0)
        ctx.set_option("segment_len", 0)


def test_schedule_slices_beyond_the_register_form(ctx, ora):
    """k_l2_place_order's schedule block keeps its slice of the segment ids in registers (32 per thread); with more than
    64 x 8192 ids per window -- n = 2^19 points in segments of ONE entry -- a slice does not fit and the two-pass loop runs.
    Also: buckets of 17..256 parts, i.e. giant buckets that are a single run of the combine."""
    n = 1 << 19
    pts, sc = ora.gen_points(19, n), ora.gen_scalars(19, n)
    exp = ora.msm(pts, sc, threads=8)
    ctx.set_option("window_bits", 16)
    try:
        for seg in (1, 2):
            ctx.set_option("segment_len", seg)
            assert ctx.run(pts, sc) == exp, seg
    finally:
        ctx.set_option("window_bits", 0)
        ctx.set_option("segment_len", 0)


def test_witness_like_scalars(ctx, model, ora):
    """SURVEY 8f rank 3, what a prover feeds: zeros, ones, small values and uniform scalars mixed -- one bucket of window 0 holds a
    large share of the points (runs of the giant-bucket combine), the high windows of many scalars are empty."""
    import random
    rnd = random.Random(31)
    for n in (3000, 70000, 300000):
        ks = model.gen_scalars(n, n)
        for i in range(n):
            q = rnd.random()
            if q < 0.5:
                ks[i] = 0 if q < 0.2 else 1 if q < 0.4 else rnd.randrange(1 << 20)
        pts, sc = ora.gen_points(n, n), model.scalars_to_bytes(ks)
        exp = ora.msm(pts, sc, threads=8)
        for c in (0, 16):
            ctx.set_option("window_bits", c)
            assert ctx.run(pts, sc) == exp, (n, c)
        ctx.set_option("window_bits", 0)
        ctx.set_option("signed_digits", 0)
        assert ctx.run(pts, sc) == exp, (n, "unsigned")
        ctx.set_option("signed_digits", 1)


def test_random_configurations(pkg, ora):
    """differential run over seeded random combinations of size, window bits, digit form, segment length, schedule and
    host pieces -- every one must equal the oracle"""
    import random
    rnd = random.Random(20241003)
    with pkg.MsmContext((0,)) as c:
        for it in range(40):
            n = rnd.choice([1, 2, 7, 64, 65, 300, 1023, 4096, 5000, 20011, 66000])
            cfg = {"window_bits": rnd.choice([0, 4, 6, 9, 12, 14, 15, 16]), "signed_digits": rnd.choice([0, 1]),
                   "segment_len": rnd.choice([1, 2, 5, 64, 300]), "sort_buckets": rnd.choice([0, 1]), "host_chunks": rnd.choice([0, 1, 2, 5]),
                   "packed_sort": rnd.choice([1, 1, 0]), "fold_pairs": rnd.choice([1, 1, 0])}
            for k, v in cfg.items():
                c.set_option(k, v)
            mode = rnd.choice(["uniform", "equal", "small", "fixed_point"])
            pts, sc = ora.gen_points(1000 + it, n), ora.gen_scalars(1000 + it, n)
            if mode == "equal":
                sc = sc[:32] * n
            elif mode == "small":
                sc = b"".join(sc[32 * i:32 * i + 3] + bytes(29) for i in range(n))
            elif mode == "fixed_point":
                pts = ora.gen_points_fixed(n)
            assert c.run(pts, sc) == ora.msm(pts, sc, threads=8), (it, n, cfg, mode)


def test_host_buffers_in_pieces(pkg, model, ora):
    """te_msm_run uploads and processes large host buffers in pieces (option host_chunks): same result for any split, a
    scalar-range error in a late piece is reported, and a ticket in flight makes it fall back to the whole-buffer path"""
    import torch
    n = 100003
    pts, sc = ora.gen_points(61, n), ora.gen_scalars(61, n)
    exp = ora.msm(pts, sc, threads=8)
    with pkg.MsmContext((0,)) as c:
        for k in (1, 2, 3, 8, 0):
            c.set_option("host_chunks", k)
            assert c.run(pts, sc) == exp, k
        c.set_option("host_chunks", 4)
        c.set_option("window_bits", 16)
        bad = bytearray(sc)
        bad[32 * (n - 5):32 * (n - 5) + 32] = b"\xff" * 32            # lands in the last piece
        with pytest.raises(pkg.MsmError) as e:
            c.run(pts, bytes(bad))
        assert e.value.code == -3
        c.set_option("window_bits", 0)
        dp, ds = _dev(pts), _dev(sc)
        torch.cuda.synchronize()
        t = c.submit_device(dp.data_ptr(), ds.data_ptr(), n)           # a ticket in flight: run() must not touch its work set
        assert c.run(pts, sc) == exp
        assert c.collect(t) == exp
        c.set_option("profile", 2)                                     # stage timing refers to one whole MSM
        assert c.run(pts, sc) == exp and "accumulate" in c.stage_ms()
        # the dominant kernel also reports its own device clock (first wave in .. last wave out): inside the event interval
        for level in (1, 2):
            c.set_option("profile", level)
            assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
            st = c.stage_ms()
            assert 0.0 < st["accumulate_on_device"] <= st["accumulate"] * 1.02 + 0.01, st
            assert 0.5 < st["accumulate_core_clock_ghz"] < 3.0, st                    # MI355X: up to 2.4 GHz
        c.set_option("profile", 0)
        assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp


def test_work_set_streams_are_spread_over_the_hardware_queues(pkg, ora):
    """The first te_msm_submit_device of a context (not te_msm_init: one-shot callers never pay the ~16 ms) measures which of
    its streams share a hardware queue, twice, and -- when both measurements agree -- re-deals them so that work sets 0..3
    (and 4..7) sit on as many different queues as the runtime has, whatever streams the process created before.  Results
    before and after the re-deal are the oracle's."""
    import torch
    extra = [torch.cuda.Stream() for _ in range(3)]                       # shift the runtime's stream counter
    n = 5000
    pts, sc = ora.gen_points(77, n), ora.gen_scalars(77, n)
    exp = ora.msm(pts, sc, threads=4)
    dp, ds = _dev(pts), _dev(sc)
    torch.cuda.synchronize()
    accepted = 0
    for rep in range(3):
        with pkg.MsmContext((0,)) as c:
            before = [c.workset_stream(i) for i in range(pkg.WORKSETS)]
            assert all(h != 0 for h, _ in before) and len({h for h, _ in before}) == pkg.WORKSETS
            assert all(k == -1 for _, k in before), "te_msm_init must not measure anything"
            assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp          # on a stream in creation order
            assert c.get_option("streams_final") == 0
            if rep == 2:
                nq_now = c.probe_queues()                                 # the explicit form: at a moment the caller chooses
                assert 0 <= nq_now <= pkg.WORKSETS
            tickets = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(4)]
            assert all(c.collect(t) == exp for t in tickets)
            assert c.get_option("streams_final") == 1
            got = [c.workset_stream(i) for i in range(pkg.WORKSETS)]
            assert {h for h, _ in got} == {h for h, _ in before}, "the same eight streams, re-dealt"
            cls = [k for _, k in got]
            if -1 in cls:
                # the two measurements disagreed: creation order kept -- identity and results must be what they were
                assert [h for h, _ in got] == [h for h, _ in before]
                assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
                continue
            accepted += 1
            nq = len(set(cls))
            assert 1 <= nq <= pkg.WORKSETS
            assert len(set(cls[:4])) == min(4, nq), cls                  # the first four work sets never share while they need not
            assert c.run_device(dp.data_ptr(), ds.data_ptr(), n) == exp
    with pkg.MsmContext((0,)) as c:                                       # option "queue_probe" = 0: nothing is measured, ever
        c.set_option("queue_probe", 0)
        before = [c.workset_stream(i) for i in range(pkg.WORKSETS)]
        tickets = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(4)]
        assert all(c.collect(t) == exp for t in tickets)
        assert [c.workset_stream(i) for i in range(pkg.WORKSETS)] == before and all(k == -1 for _, k in before)
    with pkg.MsmContext((0,)) as c:                                       # round 5: host-buffer tickets never trigger the measurement
        before = [c.workset_stream(i) for i in range(pkg.WORKSETS)]
        for t in [c.submit(pts, sc), c.submit_async(pts, sc)]:
            assert c.collect(t) == exp
        assert [c.workset_stream(i) for i in range(pkg.WORKSETS)] == before and all(k == -1 for _, k in before)
        assert c.get_option("streams_final") == 0                         # a later te_msm_submit_device may still re-deal them
        assert c.collect(c.submit_device(dp.data_ptr(), ds.data_ptr(), n)) == exp
        assert c.get_option("streams_final") == 1
    del extra
    if accepted == 0:
        pytest.skip("the hardware-queue measurement was not accepted in any of three contexts (its passes disagreed): the spreading property itself was not checked")


def test_two_work_sets_overlap_on_two_streams(pkg, model, ora):
    """te_msm_partial_device on alternating work sets and streams: many MSMs in flight pairwise, each equal to the oracle;
    a scalar-range error is reported by te_msm_partial_wait for the work set that saw it"""
    import torch
    n = 40000
    cases = []
    for seed in (21, 22, 23, 24, 25, 26):
        pts, sc = ora.gen_points(seed, n), ora.gen_scalars(seed, n)
        cases.append((_dev(pts), _dev(sc), ora.msm(pts, sc, threads=8)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        cbits, W = c.plan(n)
        parts = [torch.zeros(W * 720, dtype=torch.uint8, device="cuda") for _ in cases]
        torch.cuda.synchronize()
        for i, (dp, ds, _) in enumerate(cases):
            if i >= 2:
                c.partial_wait(i & 1)                   # the work set's previous MSM must be done before it is reused
            c.set_option("workset", i & 1)
            c.partial_device(dp.data_ptr(), ds.data_ptr(), n, parts[i].data_ptr(), streams[i & 1].cuda_stream)
        c.partial_wait(0); c.partial_wait(1)
        for i, (_, _, exp) in enumerate(cases):
            assert pkg.finalize_host(parts[i].cpu().numpy().tobytes(), cbits, W) == exp, i
        # error on work set 1 only
        bad = _dev(model.scalars_to_bytes([(1 << 256) - 1] * 4)); pts4 = _dev(ora.gen_points(3, 4))
        good = _dev(model.scalars_to_bytes([1, 2, 3, 4]))
        c.set_option("window_bits", 16)
        W16 = c.plan(4)[1]
        p0, p1 = (torch.zeros(W16 * 720, dtype=torch.uint8, device="cuda") for _ in range(2))
        torch.cuda.synchronize()
        c.set_option("workset", 0); c.partial_device(pts4.data_ptr(), good.data_ptr(), 4, p0.data_ptr(), streams[0].cuda_stream)
        c.set_option("workset", 1); c.partial_device(pts4.data_ptr(), bad.data_ptr(), 4, p1.data_ptr(), streams[1].cuda_stream)
        c.partial_wait(0)
        with pytest.raises(pkg.MsmError) as e:
            c.partial_wait(1)
        assert e.value.code == -3


def test_giant_buckets(ctx, model, ora):
    """skew: all scalars equal (one bucket per window holds every point) and a window size whose top window has a single
    occupied bucket -- buckets of 10^5 entries are split into thousands of parts and summed by the block-level combine"""
    n = 200000
    pts = ora.gen_points(71, n)
    same = model.scalars_to_bytes([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % model.P] * n)
    assert ctx.run(pts, same) == ora.msm(pts, same, threads=16)
    sc = ora.gen_scalars(72, n)
    exp = ora.msm(pts, sc, threads=16)
    for c in (12, 14, 13):                      # 253-bit scalars: 12- and 14-bit windows leave ONE bit for the top window
        ctx.set_option("window_bits", c)
        assert ctx.run(pts, sc) == exp
    ctx.set_option("window_bits", 0)
    # the same skew through the general (key + index) form of the sort's level-1 entries -- what n > 2^23 always uses: pieces of
    # over-long partitions, counted with atomics, planned by the last arriver, placed by the schedule launch
    ctx.set_option("packed_sort", 0)
    assert ctx.run(pts, same) == ora.msm(pts, same, threads=16) and ctx.run(pts, sc) == exp
    ctx.set_option("packed_sort", 1)
