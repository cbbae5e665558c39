"""Differential soak: random MSM configurations, pipelined in random order, against the oracle.  Not part of the test
suite (minutes); run on the GPU box:  python tools/soak.py [seconds]"""
import importlib
import random
import sys
import time

import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle import oracle, oracle377    # noqa: E402  (checker)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
print('soak seed', seed0, flush=True)
rnd = random.Random(seed0)
t_end = time.time() + budget
done = 0
te = pkg.MsmContext((0,))
bls = pkg.MsmContext((0,))
bls.set_option("curve", pkg.CURVE_BLS12_377_G1)
# multi-"device" contexts (the one GPU named several times): te_msm_run shards the POINTS over them, one host thread each
multi = {d: pkg.MsmContext((0,) * d) for d in (2, 3, 4)}
multi_bls = pkg.MsmContext((0, 0))
multi_bls.set_option("curve", pkg.CURVE_BLS12_377_G1)
while time.time() < t_end:
    ctx, orc, pb, sb = (te, oracle, 64, 32) if rnd.random() < 0.75 else (bls, oracle377, 96, 48)
    opts = {}
    for k, v in (("window_bits", rnd.choice([0, 0, 4, 7, 10, 13, 15, 16])), ("signed_digits", rnd.choice([1, 1, 0])),
                 ("segment_len", rnd.choice([0, 0, 64, 1, 7, 500])), ("sort_buckets", rnd.choice([1, 1, 0])), ("host_chunks", rnd.choice([0, 1, 3, 5])),
                 ("graph", rnd.choice([0, 0, 1])), ("profile", rnd.choice([0, 0, 1, 2])), ("prezero", rnd.choice([1, 1, 0])),
                 ("fuse_prep", rnd.choice([1, 1, 0])), ("packed_sort", rnd.choice([1, 1, 0])), ("fold_pairs", rnd.choice([1, 1, 0])),
                 ("host_staging", rnd.choice([0, 0, 1]))):
        ctx.set_option(k, v)
        opts[k] = v
    mode = rnd.choice(["run", "run", "tickets", "tickets", "shards", "batch", "host_tickets", "host_tickets", "multi", "multi_tickets", "multi_tickets",
                       "bases", "bases", "bases", "multi_bases"])
    if mode in ("multi", "multi_tickets", "multi_bases"):
        ctx = multi_bls if ctx is bls else multi[rnd.choice([2, 3, 4])]
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_option("host_shard_min", rnd.choice([1, 1, 4096]))
    batch, n_common = [], int(2 ** rnd.uniform(0, 17.5))
    bound_mode = mode in ("bases", "multi_bases")
    for _ in range(rnd.randint(1, pkg.WORKSETS)):
        n = n_common if (mode == "batch" or bound_mode) else int(2 ** rnd.uniform(0, 20.2 if rnd.random() < 0.05 else 17.5))      # one launch sequence takes MSMs of one size
        seed = rnd.randrange(1 << 30)
        if bound_mode and batch:
            pts = batch[0][0]                                           # resident bases: every MSM of the round runs over the ONE bound point set
        elif mode == "batch" and batch and rnd.random() < 0.4:
            pts = batch[rnd.randrange(len(batch))][0]                  # a point buffer several MSMs of the sequence share (same object -> same device pointer below)
        else:
            pts = orc.gen_points(seed, n)
        sc = orc.gen_scalars(seed, n)
        r = rnd.random()
        if r < 0.2:
            sc = sc[:sb] * n                                            # all scalars equal: one giant bucket per window
        elif r < 0.35:
            # a prover's witness: zeros, ones and small values among uniform scalars (one giant bucket in window 0, empty digits)
            a = bytearray(sc)
            for i in range(n):
                q = rnd.random()
                if q < 0.5:
                    a[sb * i:sb * (i + 1)] = (0 if q < 0.2 else 1 if q < 0.4 else rnd.randrange(1 << 20)).to_bytes(sb, "little")
            sc = bytes(a)
        batch.append((pts, sc, n))
    exp = [orc.msm(p, s, threads=8) for p, s, _ in batch]
    if bound_mode:
        # round 6: resident bases -- bind once (ordinary records; BLS12-377: affine or projective; Twisted-Edwards: now and then a
        # fixed-base table), then lone calls and tickets from host and device scalars in random order, a release and a second bind in between
        ctx.set_option("graph", 0)
        ctx.set_option("scalar_chunks", rnd.choice([0, 0, 1, 2, 4]))
        ctx.set_option("bind_affine", rnd.choice([1, 1, 0]))
        ctx.set_option("bind_fixed_base", rnd.choice([0, 0, 0, 16, 17, 18, 19, 20, 21]) if ctx.curve == pkg.CURVE_TE_BLS12 else 0)
        ctx.set_option("stage_device_inputs", rnd.choice([0, 1]))
        n = batch[0][2]
        b = ctx.bind_points(batch[0][0])
        keep, tickets, got = [], [], [None] * len(batch)
        for i, (p, s, _) in enumerate(batch):
            kind = rnd.choice(["run", "run_dev", "submit", "submit", "submit_dev"])
            if kind in ("run_dev", "submit_dev") and n:
                d_s = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                keep.append(d_s)
            if n == 0 or kind == "run":
                got[i] = ctx.run_scalars(b, s); tickets.append(None)
            elif kind == "run_dev":
                got[i] = ctx.run_scalars_device(b, d_s.data_ptr()); tickets.append(None)
            elif kind == "submit":
                tickets.append(ctx.submit_scalars(b, s))
            else:
                tickets.append(ctx.submit_scalars_device(b, d_s.data_ptr()))
        for i in rnd.sample(range(len(batch)), len(batch)):
            if tickets[i] is not None:
                if rnd.random() < 0.5:
                    ctx.ticket_wait(tickets[i])
                got[i] = ctx.collect(tickets[i])
        if rnd.random() < 0.5:                                          # released and bound again (under other bind options): the same point
            ctx.release_points(b)
            ctx.set_option("bind_fixed_base", 0)
            ctx.set_option("bind_affine", rnd.choice([1, 0]))
            b = ctx.bind_points(batch[0][0])
            assert ctx.run_scalars(b, batch[0][1]) == exp[0], "second bind"
        ctx.release_points(b)
        for k in ("bind_fixed_base", "scalar_chunks"):
            ctx.set_option(k, 0)
        ctx.set_option("bind_affine", 1)
        if rnd.random() < 0.3:
            ctx.trim(rnd.choice([0, 1, 4]))
    elif mode in ("run", "multi"):
        got = [ctx.run(p, s) for p, s, _ in batch]
    elif mode == "host_tickets":
        # te_msm_submit: host buffers, tickets collected in random order, now and then beside a synchronous call
        tickets = [ctx.submit(p, s) if n else None for p, s, n in batch]
        got = [None] * len(batch)
        for i in rnd.sample(range(len(batch)), len(batch)):
            got[i] = ctx.collect(tickets[i]) if tickets[i] is not None else ctx.run(batch[i][0], batch[i][1])
        if rnd.random() < 0.3:
            ctx.trim(rnd.choice([0, 1, 4]))
    elif mode == "multi_tickets":
        # round 5: whole-MSM tickets on a context of several "devices" -- blocking, asynchronous and device-resident submits mixed,
        # collected in random order, now and then beside a lone call (point slices / window shards over all devices) and a trim
        ctx.set_option("graph", 0)
        ctx.set_option("stage_device_inputs", rnd.choice([0, 1]))
        keep, tickets = [], []
        for p, s, n in batch:
            if not n:
                tickets.append(None)
                continue
            kind = rnd.choice(["submit", "async", "async", "device"])
            if kind == "device":
                a = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda(); b = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                keep.append((a, b))
                tickets.append(ctx.submit_device(a.data_ptr(), b.data_ptr(), n))
            else:
                tickets.append((ctx.submit if kind == "submit" else ctx.submit_async)(p, s))
        got = [None] * len(batch)
        if rnd.random() < 0.3 and len(batch) < pkg.WORKSETS:
            assert ctx.run(batch[0][0], batch[0][1]) == exp[0]
        for i in rnd.sample(range(len(batch)), len(batch)):
            if tickets[i] is not None and rnd.random() < 0.5:
                ctx.ticket_wait(tickets[i])
            got[i] = ctx.collect(tickets[i]) if tickets[i] is not None else ctx.run(batch[i][0], batch[i][1])
        if rnd.random() < 0.3:
            ctx.trim(rnd.choice([0, 1, 4]))
    elif mode == "shards":
        world = rnd.choice([2, 3, 5, 8])
        got = []
        for p_, s_, n in batch:
            a = torch.frombuffer(bytearray(p_), dtype=torch.uint8).cuda(); b = torch.frombuffer(bytearray(s_), dtype=torch.uint8).cuda()
            cb, W = ctx.plan(n)
            rows = []
            for r in range(world):
                ctx.set_window_shard(r, world)
                ctx.set_option("workset", r % pkg.WORKSETS)
                part = torch.zeros(W * ctx.row_bytes, dtype=torch.uint8, device="cuda")
                ctx.partial_device(a.data_ptr(), b.data_ptr(), n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                ctx.partial_wait(r % pkg.WORKSETS)
                rows.append(part.cpu().numpy().tobytes())
            ctx.set_window_shard(0, 1)
            ctx.set_option("workset", 0)
            got.append(pkg.finalize_host(pkg.merge_partials(rows, W, world, ctx.row_bytes), cb, W, None if opts["signed_digits"] else cb, curve=ctx.curve))
    elif mode == "batch":
        world = rnd.choice([1, 2, 4, 8])
        n = batch[0][2]
        pdev = {}
        for p, _, _ in batch:
            if id(p) not in pdev:
                pdev[id(p)] = torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda()
        dev = [(pdev[id(p)], torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()) for p, s, _ in batch]
        cb, W = ctx.plan(n)
        blk = W * ctx.row_bytes
        per_rank = []
        for r in range(world):
            ctx.set_window_shard(r, world)
            ctx.set_option("workset", r % pkg.WORKSETS)
            part = torch.zeros(len(dev) * blk, dtype=torch.uint8, device="cuda")
            ctx.partial_device_batch([a.data_ptr() for a, _ in dev], [b.data_ptr() for _, b in dev], n, part.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ctx.partial_wait(r % pkg.WORKSETS)
            per_rank.append(part.cpu().numpy().tobytes())
        ctx.set_window_shard(0, 1)
        ctx.set_option("workset", 0)
        got = [pkg.finalize_host(pkg.merge_partials([rows[m * blk:(m + 1) * blk] for rows in per_rank], W, world, ctx.row_bytes), cb, W,
                                 None if opts["signed_digits"] else cb, curve=ctx.curve) for m in range(len(dev))]
    else:
        dev = [(torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda(), n) for p, s, n in batch]
        torch.cuda.synchronize()
        tickets = [ctx.submit_device(a.data_ptr(), b.data_ptr(), n) for a, b, n in dev]
        got = [ctx.collect(t) for t in tickets]
    if got != exp:
        print("MISMATCH", "bls" if ctx in (bls, multi_bls) else "te", opts, mode, [(n, g == e) for (_, _, n), g, e in zip(batch, got, exp)], flush=True)
        raise SystemExit(1)
    done += len(batch)
    if done % 50 < len(batch):
        print("ok", done, flush=True)
print("soak passed:", done, "MSMs")
