"""GPU parity for BASELINE config 5: MSM on BLS12-377 G1 through the same C-ABI (option "curve" = 1), against the
BLS12-377 oracle (oracle/bls377_oracle.c, itself checked against the bigint model in tests/test_oracle_bls377.py).
The reference holds no vector for this curve: parity is UNPINNED by the reference and pinned to the group law only."""
import ctypes

import pytest

from oracle import model377 as m
from oracle import oracle377 as o

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bls(pkg):
    c = pkg.MsmContext((0,))
    c.set_option("curve", pkg.CURVE_BLS12_377_G1)
    yield c
    c.close()


def _dev(b):
    import torch
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()


def test_records_against_the_model_and_the_host_build(bls, fq377check):
    """the 224-byte records are projective points of the curve's twisted-Edwards form: mapped back to y^2 = x^3 + 1 they
    are the input points (bigint model); byte for byte they equal the host build of the same header"""
    from test_oracle_bls377 import _edwards_consts, edwards_to_weierstrass
    L = fq377check
    n = 300
    pts, sc = o.gen_points(1, n), o.gen_scalars(1, n)
    assert bls.run(pts, sc) == o.msm(pts, sc, threads=4)
    recs = bls.debug_read("records", n * 224)
    s_, f_, d_ = _edwards_consts(L)
    rinv = pow(1 << 406, -1, m.Q)
    for i in (0, 1, 2, 77, n - 1):
        r = ctypes.create_string_buffer(224)
        L.f377_prep_point(pts[96 * i:96 * i + 96], r)
        assert recs[224 * i:224 * i + 224] == r.raw, i
        hm, hp, dt, z = (sum(int.from_bytes(r.raw[56 * k + 4 * j:56 * k + 4 * j + 4], "little") << (29 * j) for j in range(14)) * rinv % m.Q
                         for k in range(4))
        zi = pow(z, -1, m.Q)
        xa, ya = (hp - hm) * zi % m.Q, (hp + hm) * zi % m.Q
        assert edwards_to_weierstrass(xa, ya, s_, f_) == m.xy_from_bytes(pts[96 * i:96 * i + 96])
        assert dt * zi % m.Q == -d_ * xa * ya % m.Q


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 1000, 4097, 70001])
def test_ragged_sizes(bls, n):
    pts, sc = o.gen_points(n, n), o.gen_scalars(n, n)
    assert bls.run(pts, sc) == o.msm(pts, sc, threads=8)


@pytest.mark.parametrize("c", [4, 7, 8, 11, 13, 15, 16])
def test_window_sizes_and_digit_forms(bls, c):
    n = 2500
    pts, sc = o.gen_points(50 + c, n), o.gen_scalars(50 + c, n)
    exp = o.msm(pts, sc, threads=4)
    bls.set_option("window_bits", c)
    for signed in (1, 0):
        bls.set_option("signed_digits", signed)
        assert bls.run(pts, sc) == exp
    bls.set_option("signed_digits", 1)
    bls.set_option("window_bits", 0)


def test_edge_scalars_empty_and_errors(bls, pkg):
    pts = o.gen_points(9, 8)
    ks = [0, 1, m.R_ORDER - 1, m.R_ORDER, 1 << 15, (1 << 16) - 1, (1 << 252) + 5, (1 << 253) + 12345]
    pl = [m.xy_from_bytes(pts[96 * i:96 * i + 96]) for i in range(8)]
    assert bls.run(pts, m.scalars_to_bytes(ks)) == m.result_to_bytes(m.msm_naive(pl, ks))
    assert bls.run(b"", b"") == bytes(96)                                              # empty sum: infinity
    assert bls.run(pts[:96], m.scalars_to_bytes([0])) == bytes(96)
    assert bls.run(pts[:192], m.scalars_to_bytes([5, m.R_ORDER - 5]) ) == m.result_to_bytes(
        m.add(m.scalar_mul(5, pl[0]), m.scalar_mul(m.R_ORDER - 5, pl[1])))
    same = pts[:96] * 2                                                                # P + (-P) = infinity through the buckets
    assert bls.run(same, m.scalars_to_bytes([7, m.R_ORDER - 7])) == bytes(96)
    bls.set_option("window_bits", 16)
    with pytest.raises(pkg.MsmError) as e:
        bls.run(pts[:96], m.scalars_to_bytes([(1 << 256) - 1]))                        # final carry
    assert e.value.code == -3
    with pytest.raises(pkg.MsmError) as e:
        bls.run(pts[:96], m.scalars_to_bytes([1 << 300]))                              # 48-byte record above 2^256
    assert e.value.code == -3
    bls.set_option("window_bits", 0)
    with pytest.raises(pkg.MsmError):
        bls.run(pts[:96], bytes(32))                                                   # wrong record size for this curve


def test_skew_and_split_buckets(bls):
    """all scalars equal (every window: one bucket holds every point), and a tiny segment length"""
    n = 6000
    pts = o.gen_points(17, n)
    sc = o.gen_scalars(17, 1) * n
    assert bls.run(pts, sc) == o.msm(pts, sc, threads=8)
    sc = o.gen_scalars(18, n)
    exp = o.msm(pts, sc, threads=8)
    for seg in (1, 3, 64):
        bls.set_option("segment_len", seg)
        assert bls.run(pts, sc) == exp
    bls.set_option("segment_len", 0)


def test_device_resident_and_pipelined(bls, pkg):
    import torch
    cases = []
    for seed, n in ((21, 30000), (22, 5000), (23, 65536)):
        pts, sc = o.gen_points(seed, n), o.gen_scalars(seed, n)
        cases.append((_dev(pts), _dev(sc), n, o.msm(pts, sc, threads=8)))
    torch.cuda.synchronize()
    assert bls.run_device(cases[0][0].data_ptr(), cases[0][1].data_ptr(), cases[0][2]) == cases[0][3]
    ts = [bls.submit_device(dp.data_ptr(), ds.data_ptr(), n) for dp, ds, n, _ in cases]
    for t, (_, _, _, exp) in zip(ts, cases):
        assert bls.collect(t) == exp


@pytest.mark.parametrize("world", [2, 8])
def test_window_shards_on_one_gpu(pkg, world):
    """window sharding for this curve: every rank's rows (1120 bytes per window) merged and folded give the oracle's point"""
    import torch
    n = 20000
    pts, sc = o.gen_points(31, n), o.gen_scalars(31, n)
    dp, ds = _dev(pts), _dev(sc)
    exp = o.msm(pts, sc, threads=8)
    rows, cW = [], None
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            c.set_option("curve", pkg.CURVE_BLS12_377_G1)
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            part = torch.zeros(W * c.row_bytes, dtype=torch.uint8, device="cuda")
            c.partial_device(dp.data_ptr(), ds.data_ptr(), n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            rows.append(part.cpu().numpy().tobytes())
            cW = (cbits, W)
            if r == world - 1:
                assert c.finalize(pkg.merge_partials(rows, W, world, c.row_bytes), cbits, W) == exp
    assert pkg.finalize_host(pkg.merge_partials(rows, cW[1], world, 1120), *cW, curve=pkg.CURVE_BLS12_377_G1) == exp


@pytest.mark.parametrize("world,count,n", [(1, 2, 3000), (4, 4, 9000)])
def test_batches_of_window_sharded_msms(pkg, world, count, n):
    """te_msm_partial_device_batch for this curve (224-byte record slabs, 1120-byte rows): every MSM of a launch sequence,
    merged over the ranks, against the oracle"""
    import torch
    ins = [(o.gen_points(500 + m, n), o.gen_scalars(520 + m, n)) for m in range(count)]
    dev = [(_dev(p), _dev(s)) for p, s in ins]
    per_rank = []
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            c.set_option("curve", pkg.CURVE_BLS12_377_G1)
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            blk = W * c.row_bytes
            part = torch.zeros(count * blk, dtype=torch.uint8, device="cuda")
            c.partial_device_batch([p.data_ptr() for p, _ in dev], [s.data_ptr() for _, s in dev], n, part.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            c.partial_wait(0)
            per_rank.append(part.cpu().numpy().tobytes())
    for m, (p_, s_) in enumerate(ins):
        merged = pkg.merge_partials([rows[m * blk:(m + 1) * blk] for rows in per_rank], W, world, 1120)
        assert pkg.finalize_host(merged, cbits, W, curve=pkg.CURVE_BLS12_377_G1) == o.msm(p_, s_, threads=8), f"MSM {m} of the batch"


def test_full_size_window_shards_d2(pkg):
    """n = 2^20 with the windows sharded over two ranks (run one after the other on the one GPU), two MSMs per launch
    sequence, merged with te_msm_finalize_gathered_curve -- against this curve's oracle (unpinned by the reference)"""
    import torch
    n, world, count = 1 << 20, 2, 2
    pts, _ = pkg.synth_inputs(0x5EED0060, n, scalars=False, curve=pkg.CURVE_BLS12_377_G1)
    ssets = [o.gen_scalars(0x5EED0061 + m, n) for m in range(count)]
    exp = [o.msm(pts, sc, c=16, threads=16) for sc in ssets]
    dp, dscs = _dev(pts), [_dev(sc) for sc in ssets]
    torch.cuda.synchronize()
    gathered = None
    for r in range(world):
        with pkg.MsmContext((0,)) as c:
            c.set_option("curve", pkg.CURVE_BLS12_377_G1)
            c.set_option("window_bits", 16)
            c.set_window_shard(*pkg.window_shard_for_rank(r, world))
            cbits, W = c.plan(n)
            blk = W * c.row_bytes
            if gathered is None:
                gathered = torch.zeros(world, count, blk, dtype=torch.uint8)
            part = torch.zeros(count * blk, dtype=torch.uint8, device="cuda")
            c.partial_device_batch([dp.data_ptr()] * count, [d.data_ptr() for d in dscs], n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            c.partial_wait(0)
            gathered[r] = part.cpu().view(count, blk)
    for m in range(count):
        mine = gathered[:, m, :].contiguous()
        assert pkg.finalize_gathered(mine.data_ptr(), world, cbits, W, curve=pkg.CURVE_BLS12_377_G1) == exp[m], f"MSM {m}"


def test_pipelined_full_size(pkg):
    """n = 2^20 with four MSMs in flight, three rounds: at this size the launch sequences really overlap on the device (the
    small pipelined cases finish before the next one starts).  Round 3 shipped a build step for an hour that broke exactly
    this and nothing else in the suite (profiles/r03_peephole_experiment.txt)."""
    import torch
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0x5EED0070, n, curve=pkg.CURVE_BLS12_377_G1)
    exp = o.msm(pts, sc, c=16, threads=16)
    dp, ds = _dev(pts), _dev(sc)
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        for _ in range(3):
            tickets = [c.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(4)]
            assert [c.collect(t) == exp for t in tickets] == [True] * 4


def test_giant_buckets_and_host_pieces(bls):
    """skew: all scalars equal (one bucket per window holds every point: thousands of parts summed by the block-level
    combine) and window sizes whose top window has a single occupied bucket; te_msm_run in pieces"""
    n = 120000
    pts = o.gen_points(71, n)
    same = o.gen_scalars(5, 1) * n
    assert bls.run(pts, same) == o.msm(pts, same, threads=16)
    sc = o.gen_scalars(72, n)
    exp = o.msm(pts, sc, threads=16)
    for c in (12, 14):
        bls.set_option("window_bits", c)
        assert bls.run(pts, sc) == exp
    bls.set_option("window_bits", 0)
    for k in (1, 3, 5):
        bls.set_option("host_chunks", k)
        assert bls.run(pts, sc) == exp
    bls.set_option("host_chunks", 0)


def test_multi_device_point_shards_and_host_tickets(pkg):
    """BLS12-377 through the round-4 host paths: te_msm_run on an n_dev = 3 context (point slices, rows of 1120 bytes summed in
    this curve's host tail) and te_msm_submit tickets (scalars travel piece by piece with their points on this curve)"""
    from oracle import oracle377
    n = 50021
    pts, sc = oracle377.gen_points(91, n), oracle377.gen_scalars(91, n)
    exp = oracle377.msm(pts, sc, threads=8)
    with pkg.MsmContext((0, 0, 0)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        c.set_option("host_shard_min", 1)
        for chunks in (0, 2):
            c.set_option("host_chunks", chunks)
            assert c.run(pts, sc) == exp, chunks
        assert c.run(pts[:96 * 2], sc[:48 * 2]) == oracle377.msm(pts[:96 * 2], sc[:48 * 2])        # fewer points than devices
    with pkg.MsmContext((0,)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        for chunks in (0, 3):
            c.set_option("host_chunks", chunks)
            ts = [c.submit(pts, sc) for _ in range(3)]
            assert [c.collect(t) for t in reversed(ts)] == [exp] * 3
        assert pkg.finalize_sum([bytes(16 * 1120)], 16, 16, curve=pkg.CURVE_BLS12_377_G1) == bytes(96)   # no rows at all: the point at infinity


def test_one_context_serves_both_curves(pkg, ora):
    n = 3000
    with pkg.MsmContext((0,)) as c:
        pts, sc = ora.gen_points(5, n), ora.gen_scalars(5, n)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=4)
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        p2, s2 = o.gen_points(5, n), o.gen_scalars(5, n)
        assert c.run(p2, s2) == o.msm(p2, s2, threads=4)
        c.set_option("curve", pkg.CURVE_TE_BLS12)
        assert c.run(pts, sc) == ora.msm(pts, sc, threads=4)


def test_full_size_2_20(bls):
    n = 1 << 20
    pts, sc = o.gen_points(0x5EED0014, n), o.gen_scalars(0x5EED0014, n)
    bls.set_option("window_bits", 16)
    assert bls.run(pts, sc) == o.msm(pts, sc, c=16, threads=16)
    bls.set_option("window_bits", 0)


def test_tickets_on_a_multi_device_context_bls12_377(pkg):
    """round 5: whole-MSM tickets of the second curve on four "devices" (96-byte points, 48-byte scalar records, 1 120-byte rows):
    blocking, asynchronous and staged submits, any collect order, the lone call beside them"""
    cases = []
    for seed, n in ((601, 30000), (602, 70001), (603, 257), (604, 9000)):
        pts, sc = o.gen_points(seed, n), o.gen_scalars(seed, n)
        cases.append((pts, sc, o.msm(pts, sc, threads=8)))
    with pkg.MsmContext((0, 0, 0, 0)) as c:
        c.set_option("curve", pkg.CURVE_BLS12_377_G1)
        for staging in (0, 1):
            c.set_option("host_staging", staging)
            ts = [(c.submit_async if i % 2 else c.submit)(p, s) for i, (p, s, _) in enumerate(cases * 2)]
            assert sorted(c.ticket_device(t)[0] for t in ts) == [0, 0, 1, 1, 2, 2, 3, 3]
            assert c.run(cases[1][0], cases[1][1]) == cases[1][2]
            for i in (5, 0, 7, 2, 1, 6, 3, 4):
                assert c.collect(ts[i]) == cases[i % 4][2], (staging, i)
