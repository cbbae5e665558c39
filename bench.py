"""bench.py -- MSM throughput / latency of the HIP path on N GPUs of one node.

A "step" is one full MSM (n = 2^20 Twisted-Edwards BLS12 points, 16-bit signed windows) over
synthetic inputs already resident in HBM.  N = 1: te_msm_submit_device / te_msm_collect (device stages + host
tail), several MSMs in flight.  N > 1: the MSM's 16 windows are sharded over the ranks (one process per GPU), the
11 KB of partial sums per MSM are exchanged with one RCCL all-gather per launch sequence -- a sequence carries one MSM,
or, from four ranks on, as many MSMs as there are ranks (te_msm_partial_device_batch, --batch) -- and every rank runs the
host tail ("scaling": "strong").

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # no launcher: bench.py starts its N ranks itself (launch_ranks below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            # under a launcher: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env

One JSON line on rank 0.  `value` / `ms_per_step` = pipelined throughput with inputs resident in HBM; `latency_ms` =
one synchronous MSM from resident inputs; `host_buffers_ms` = one te_msm_run from pageable host buffers (what the
reference's compute_msm(Buffer, Buffer) delivers, PCIe included; never `value`); `sizes` = the same three figures for
n = 2^16..2^20 (full_benchmarks.ts:13-15 runs 16..20); `configs` = short passes of BASELINE configs 2 (unsigned windows) and
5 (BLS12-377 G1) at n = 2^20, each with its own parity check, roofline and cpu_baseline.
"""
import argparse
import hashlib
import re
import importlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "webgpu-msm-twisted-edwards_amd"
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")
ISA_JSON = os.path.join(ROOT, "profiles", "isa_cycles.json")
KERNEL_SOURCES = ("kernels.hip.hpp", "curve.hpp", "fp.hpp")     # what decides k_accumulate's memory traffic and instruction count


def kernel_sources_sha():
    """hash of the CODE of the kernel sources: comments and white space do not take part, so that re-wording a comment does
    not declare a profile stale (tools/summarize_prof.py and tools/isa_hist.py compute the same)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, PKG, "csrc", f), "r", errors="replace") as fh:
            text = fh.read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def measured_traffic(log2n, c, world, default_workload=True, config=None):
    """HBM bytes per k_accumulate launch from the committed rocprofv3 PMC passes (tools/profile_bench.sh and, for the side
    configs, tools/profile_configs.sh -> profiles/pmc_traffic.json).  Counters cannot be read from inside the process, so
    the figure applies to the profiled workloads only (n = 2^20, c = 16, one GPU: the default line under "kernels", the
    side configs under "configs") and to the kernel sources they were profiled with: the JSON records their hash and the
    commit; when the sources have changed since, the figure is withheld (null) and flagged stale."""
    info = {"file": os.path.relpath(TRAFFIC_JSON, ROOT), "sources_sha_now": kernel_sources_sha()}
    if not (log2n == 20 and c == 16 and world == 1 and (default_workload or config) and os.path.exists(TRAFFIC_JSON)):
        return None, info
    try:
        j = json.load(open(TRAFFIC_JSON))
        k = (j["configs"][config]["kernels"] if config else j["kernels"])["k_accumulate"]
        if config:
            info["profiled_workload"] = j["configs"][config].get("workload")
    except (KeyError, ValueError):
        return None, info
    info.update({"profiled_at_commit": j.get("commit"), "sources_sha_profiled": j.get("kernel_sources_sha"),
                 "correction": j.get("correction"), "fetch_bytes_raw": k.get("fetch_bytes_raw"), "write_bytes": k.get("write_bytes")})
    if j.get("kernel_sources_sha") != info["sources_sha_now"]:
        info["stale"] = True
        info["stale_value"] = k.get("hbm_bytes_per_launch")
        return None, info
    info["stale"] = False
    return k.get("hbm_bytes_per_launch"), info


def isa_cycles(bls, affine=False):
    """estimated VALU issue cycles per 64 accumulated points, from the ISA listing of the build (tools/isa_hist.py --json ->
    profiles/isa_cycles.json, which records the hash of the sources it was taken from); affine: BLS12-377 over bound bases
    (k_accumulate<14, 1>: 7 products per gathered point)"""
    fallback = {"cycles": (13669.0 if affine else 15545.0) if bls else 6136.0, "note": "round-2 listing (profiles/r02_isa_hist_k_accumulate.txt)", "stale": True}
    try:
        j = json.load(open(ISA_JSON))
        k = j[("k_accumulate<14,affine>" if affine else "k_accumulate<14>") if bls else "k_accumulate<9>"]
    except (OSError, KeyError, ValueError):
        return fallback
    return {"cycles": float(k["valu_issue_cycles"]), "note": k.get("note", ""), "stale": j.get("kernel_sources_sha") != kernel_sources_sha()}


def algorithmic_bytes(n, W, B, bls=False, entries=None):
    """whole MSM, and the share of the dominant kernel (bucket accumulation).
    entries: the points the accumulation actually gathers = non-zero window digits, counted by the engine (option
    "entries_accumulated"): a zero digit contributes nothing (smvp.template.wgsl:128) and is dropped by the sort.  SURVEY's
    formula writes W * n for it -- exact to 2^-c for uniform scalars, but a prover's witness with half of its scalars 0 or 1
    has 8.25 non-zero digits of 16: pricing W * n there put the kernel ABOVE its own instruction floor (round 4: frac 1.49)."""
    e = W * n if entries is None else entries
    if bls:                                               # 48-B coordinates and scalar records, 144-B projective buckets
        return 144 * n + e * (96 + 4) + 2 * W * B * 144 + 96, e * (96 + 4) + W * B * 144
    whole = 96 * n + e * (64 + 4) + 2 * W * B * 128 + 64
    accumulate = e * (64 + 4) + W * B * 128              # gather each point + its 4-B index once, write each bucket once
    return whole, accumulate


def pipelined_pass(ctx, dp, ds, n, steps, depth, note=None):
    """`steps` MSMs back to back, `depth` in flight (te_msm_submit_device / te_msm_collect); returns (seconds, last result)"""
    import torch
    result, tickets = None, []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tickets.append(ctx.submit_device(dp, ds, n))
        if len(tickets) >= depth:
            result = ctx.collect(tickets.pop(0))
            if note:
                note()
    while tickets:
        result = ctx.collect(tickets.pop(0))
        if note:
            note()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, result


def alone_pass(ctx, step, reps):
    """`reps` synchronous MSMs at profile level 2 (an event at every stage boundary): mean stage times with ONE MSM on the GPU"""
    ctx.set_option("profile", 2)
    acc = {}
    for _ in range(reps):
        step()
        for k, v in ctx.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v
    ctx.set_option("profile", 1)
    return {k: v / reps for k, v in acc.items()}


def bases_resident_figures(ctx, pts, sc, expect, d_sc=None, n=None, depth=4, steps=96):
    """Resident bases (te_msm_bind_points + te_msm_run_scalars / te_msm_submit_scalars): the points are bound ONCE (timed: bind_ms),
    every MSM then moves its scalars only -- what the reference's harness could do with the one point buffer it passes to six calls
    per size (full_benchmarks.ts:63-68,100-105).  latency_ms: one te_msm_run_scalars at a time from pageable host scalars (best of
    7); in_flight_ms: te_msm_submit_scalars tickets, 8 in flight, per MSM (best of 3 passes of `steps` MSMs: 96, so that filling and draining
    the pipeline -- an upload plus an MSM, ~3 ms at n = 2^20 -- stay a few per cent of a pass); device_scalars_ms: scalars already in
    HBM, `depth` tickets in flight (te_msm_submit_scalars_device).  Every result is compared with `expect`."""
    import torch
    prof = ctx.get_option("profile")
    ctx.set_option("profile", 0)
    out = {}
    try:
        t1 = time.perf_counter()
        b = ctx.bind_points(pts)
        out["bind_ms"] = (time.perf_counter() - t1) * 1e3
        assert ctx.run_scalars(b, sc) == expect
        lat = []
        for _ in range(7):
            t1 = time.perf_counter()
            r = ctx.run_scalars(b, sc)
            lat.append((time.perf_counter() - t1) * 1e3)
        assert r == expect
        out["latency_ms"] = min(lat)
        infl = 8
        for t in [ctx.submit_scalars(b, sc) for _ in range(infl)]:
            assert ctx.collect(t) == expect
        passes = []
        for _ in range(3):
            t1 = time.perf_counter()
            tk = []
            for _ in range(steps):
                tk.append(ctx.submit_scalars(b, sc))
                if len(tk) >= infl:
                    assert ctx.collect(tk.pop(0)) == expect
            while tk:
                assert ctx.collect(tk.pop(0)) == expect
            passes.append((time.perf_counter() - t1) * 1e3 / steps)
        out["in_flight_ms"] = min(passes)
        out["in_flight_passes_ms"] = passes
        if d_sc is not None:
            for t in [ctx.submit_scalars_device(b, d_sc) for _ in range(depth)]:
                assert ctx.collect(t) == expect
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tk = []
            for _ in range(steps):
                tk.append(ctx.submit_scalars_device(b, d_sc))
                if len(tk) >= depth:
                    assert ctx.collect(tk.pop(0)) == expect
            while tk:
                assert ctx.collect(tk.pop(0)) == expect
            out["device_scalars_ms"] = (time.perf_counter() - t1) * 1e3 / steps
        out["bytes_per_msm_over_pcie"] = len(sc)
        out["bases_bytes_on_device"] = ctx.get_option("bases_bytes")
        ctx.release_points(b)
    finally:
        ctx.set_option("profile", prof)
    return out


def roofline_block(acc_bytes, alone, timed, bls, waves, traffic=None, traffic_info=None, msms_per_launch=1, entries=None):
    """roofline of the dominant kernel.  `achieved` = algorithmic bytes per launch / the kernel's mean duration with the GPU
    to itself (HIP events around the launch on the engine's stream, measured live in this run; agrees with the kernel trace
    of `bench.py --no-pipeline`): a per-launch cost that cannot exceed ms_per_step.  The duration seen in the timed region,
    where `depth` MSMs share the GPU and two accumulations usually overlap, is a concurrency-stretched span: side field."""
    alone_ms = alone.get("accumulate") or alone.get("accumulate_on_device")
    achieved = acc_bytes / (alone_ms * 1e-3) / 1e9 if alone_ms else 0.0
    isa = isa_cycles(bls)
    ghz = alone.get("accumulate_core_clock_ghz")
    floor_ms = waves * isa["cycles"] / 1024.0 / 2.4e6          # 1024 SIMDs at the 2.4 GHz peak clock
    valu = {"bound": "valu-issue", "floor_ms_at_2.4GHz": floor_ms, "kernel_ms": alone_ms, "frac": floor_ms / alone_ms if alone_ms else None,
            "core_clock_ghz": ghz, "frac_at_measured_clock": (floor_ms * 2.4 / ghz) / alone_ms if (ghz and alone_ms) else None,
            "valu_issue_cycles_per_64_points": isa["cycles"], "isa_listing_stale": isa["stale"], "note": isa["note"]}
    out = {"bound": "hbm", "kernel": "k_accumulate", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_info": traffic_info,
           "algorithmic_bytes_per_launch": acc_bytes, "entries_accumulated_per_launch": entries, "kernel_ms": alone_ms, "kernel_ms_device_clock": alone.get("accumulate_on_device"),
           "msms_per_launch": msms_per_launch,
           "duration": "mean of the launches of an untimed pass inside this run with one MSM on the GPU (HIP events on the engine's stream)",
           "timed_region": timed, "binding_roofline": valu,
           "note": ("VALU-bound: 8 products of 14-limb operands per gathered point" if bls else
                    "the north star names the HBM roofline; the kernel is VALU-issue bound (7 field products per gathered "
                    "point, see binding_roofline and DESIGN.md section 4)")}
    return out


def side_config(pkg, dev, name, curve, digits, log2n, depth, threads, steps, expect=None, inputs=None, workload_note=""):
    """a short pass of another BASELINE config on a fresh context: pipelined throughput, latency, the dominant kernel alone,
    parity against the config's oracle (timed: its cpu_baseline)"""
    import torch
    bls = curve == "bls12-377"
    n = 1 << log2n
    if inputs is None:
        inputs = pkg.synth_inputs(0x5EED0000 + log2n, n, curve=pkg.CURVE_BLS12_377_G1 if bls else pkg.CURVE_TE_BLS12)
    pts, sc = inputs
    dpt = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    dst = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    dp, ds = dpt.data_ptr(), dst.data_ptr()
    with pkg.MsmContext((dev,)) as cx:
        cx.set_option("window_bits", 16)
        cx.set_option("signed_digits", 1 if digits == "signed" else 0)
        if bls:
            cx.set_option("curve", pkg.CURVE_BLS12_377_G1)
        cx.set_option("profile", 1)
        c, W = cx.plan(n)
        B = 1 << (c - 1 if digits == "signed" else c)
        for t in [cx.submit_device(dp, ds, n) for _ in range(depth)]:      # every work set allocates its buffers outside the timed pass
            result = cx.collect(t)
        lat = []
        cx.set_option("profile", 0)
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cx.run_device(dp, ds, n)
            lat.append((time.perf_counter() - t1) * 1e3)
        cx.set_option("profile", 1)
        elapsed, result = pipelined_pass(cx, dp, ds, n, steps, depth)
        alone = alone_pass(cx, lambda: cx.run_device(dp, ds, n), 4)
        entries = cx.get_option("entries_accumulated")     # non-zero window digits of that last MSM, counted on the device
        bound = None
        if bls:
            # BLS12-377 over BOUND bases: affine records (one inversion per point at bind time): 7 products and 168 bytes per gathered
            # point instead of 8 and 224
            bound = bases_resident_figures(cx, pts, sc, result, ds, n, depth, 48)
            cx.set_option("profile", 1)
            bb = cx.bind_points(pts)
            st_b = alone_pass(cx, lambda: cx.run_scalars_device(bb, ds), 4)
            bound["accumulate_alone_ms"] = st_b.get("accumulate")
            isa_b = isa_cycles(True, affine=True)
            floor_b = (cx.get_option("entries_accumulated") / 64.0) * isa_b["cycles"] / 1024.0 / 2.4e6
            ghz_b = st_b.get("accumulate_core_clock_ghz")
            bound["binding_roofline"] = {"bound": "valu-issue", "kernel": "k_accumulate<14, affine records>", "floor_ms_at_2.4GHz": floor_b, "kernel_ms": st_b.get("accumulate"),
                                         "frac": floor_b / st_b["accumulate"] if st_b.get("accumulate") else None, "core_clock_ghz": ghz_b,
                                         "frac_at_measured_clock": (floor_b * 2.4 / ghz_b) / st_b["accumulate"] if (ghz_b and st_b.get("accumulate")) else None,
                                         "valu_issue_cycles_per_64_points": isa_b["cycles"], "isa_listing_stale": isa_b["stale"], "note": isa_b["note"]}
            cx.release_points(bb)
    whole, acc_bytes = algorithmic_bytes(n, W, B, bls, entries)
    traffic, traffic_info = measured_traffic(log2n, c, 1, False, name)
    out = {"workload": "n=2^%d %s MSM, %d-bit %s windows (%d windows x %d buckets)%s, inputs resident in HBM" % (
               log2n, "BLS12-377 G1" if bls else "TE-BLS12", c, digits, W, B, workload_note),
           "value": steps / elapsed, "unit": "MSM/s", "steps": steps, "ms_per_step": elapsed * 1e3 / steps, "latency_ms": min(lat),
           "roofline": roofline_block(acc_bytes, alone, None, bls, entries / 64.0, traffic, traffic_info, entries=entries),
           "msm_algorithmic_bytes": whole}
    if bound:
        bound["value_device_scalars"] = 1e3 / bound["device_scalars_ms"]
        bound["note"] = ("te_msm_bind_points once (affine records: 168 B and 7 products per gathered point; bind_ms), then scalars only: device_scalars_ms = "
                         "te_msm_submit_scalars_device tickets, %d in flight; in_flight_ms / latency_ms from pageable host scalars (48 B per point over PCIe)" % depth)
        out["bases_resident"] = bound
    if expect is None:
        t0 = time.perf_counter()
        if bls:
            from oracle import oracle377       # cpu_baseline leg (and result checker) only
            expect = oracle377.msm(pts, sc, c=16, threads=threads)
        else:
            from oracle import oracle          # cpu_baseline leg (and result checker) only
            expect = oracle.msm(pts, sc, c=16, bpr_mode=1, threads=threads)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "MSM/s", "cores": threads, "kind": "port",
                               "sample": "1 full MSM at n=2^%d (oracle/%s), %.1f s" % (
                                   log2n, "bls377_oracle.c; parity unpinned by the reference" if bls else "te_oracle.c", cpu_s)}
    else:
        out["cpu_baseline"] = "same inputs as the headline run: see its cpu_baseline (the oracle's result does not depend on the digit form)"
    out["parity"] = "bit-exact vs oracle" if expect == result else "MISMATCH vs oracle"
    if bls:
        out["parity"] += " (oracle unpinned by the reference: it holds no code or vector for this curve)"
    return out, expect == result


def launch_ranks(n, cmd, timeout=None, env=None):
    """`python bench.py --gpus N` WITHOUT a launcher: starts N fresh child processes of `cmd` (one rank per GPU) with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set -- plain subprocess.Popen from a process that has not imported
    torch nor touched HIP (never an exec of a process that has) --, relays rank 0's stdout (the JSON line) to ours and the other
    ranks' output to stderr, and returns the exit code: 0 when every rank returned 0, else the first non-zero code, after the
    remaining ranks have been terminated (a rank that died would otherwise leave the others in a collective until their own
    timeout).  Every child is its own process group: what it started goes with it."""
    timeout = float(os.environ.get("TE_BENCH_LAUNCH_TIMEOUT", "3000")) if timeout is None else timeout
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:          # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, pumps = [], []
    for r in range(n):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=None, start_new_session=True, text=True, bufsize=1)
        procs.append(p)

        def pump(p=p, r=r):
            for line in p.stdout:
                if r == 0:
                    sys.stdout.write(line); sys.stdout.flush()
                else:
                    sys.stderr.write("[rank %d] %s" % (r, line)); sys.stderr.flush()
        t = threading.Thread(target=pump, daemon=True)
        t.start()
        pumps.append(t)

    def stop_all():
        for q in procs:
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGTERM)
                except OSError:
                    pass
        t_end = time.time() + 5.0
        for q in procs:
            while q.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if q.poll() is None:
                try:
                    os.killpg(q.pid, signal.SIGKILL)
                except OSError:
                    pass
                q.wait()

    rc, t0 = 0, time.time()
    try:
        while True:
            codes = [q.poll() for q in procs]
            failed = [(i, c) for i, c in enumerate(codes) if c not in (None, 0)]
            if failed:
                rc = failed[0][1] if failed[0][1] > 0 else 128 - failed[0][1]         # a signal -N reads as 128 + N
                sys.stderr.write("bench.py: rank %d exited with %d: stopping the other ranks\n" % failed[0])
                break
            if all(c == 0 for c in codes):
                break
            if time.time() - t0 > timeout:
                rc = 124
                sys.stderr.write("bench.py: the ranks did not finish within %.0f s: stopping them\n" % timeout)
                break
            time.sleep(0.05)
    finally:
        stop_all()
        for t in pumps:
            t.join(timeout=2.0)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--window-bits", type=int, default=16)
    ap.add_argument("--points", choices=["random", "chain", "fixed"], default="random",
                    help="random: n independent seeded-random a_i*G (SURVEY 8d set (R), default); chain: the arithmetic progression (a+i*b)G; "
                         "fixed: harness mode (H), one point replicated (ui/AllBenchmarks.tsx:105-112)")
    ap.add_argument("--scalars", choices=["uniform", "equal", "small", "mixed"], default="uniform",
                    help="uniform: the harness distribution; equal: one scalar repeated (worst-case skew); small: 64-bit scalars; mixed: a quarter zeros, a quarter ones, the rest uniform")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sizes", action="store_true", help="skip the n = 2^16..2^19 side table")
    ap.add_argument("--no-configs", action="store_true", help="skip the short passes of BASELINE configs 2 and 5 (unsigned windows, BLS12-377 G1)")
    ap.add_argument("--no-host-buffers", action="store_true", help="skip the te_msm_run (host buffer) timing: kernel traces of one configuration only")
    ap.add_argument("--segment-len", type=int, default=0)
    ap.add_argument("--curve", choices=("te", "bls12-377"), default="te",
                    help="te: the Twisted-Edwards BLS12 curve (headline); bls12-377: G1 of BLS12-377, BASELINE config 5 (single GPU)")
    ap.add_argument("--digits", choices=("signed", "unsigned"), default="signed",
                    help="signed window digits, 2^(c-1) buckets (BASELINE config 3, the reference's shipped behaviour) or unsigned, 2^c buckets (config 2)")
    ap.add_argument("--inflight", type=int, default=0,
                    help="MSMs in flight in pipelined mode (1..8, each on its own stream / work set); 0 = 4 on one GPU, 8 when the windows are sharded (small per-rank kernels)")
    ap.add_argument("--batch", type=int, default=0,
                    help="window-sharded runs: MSMs per launch sequence (te_msm_partial_device_batch); 0 = as many as make a rank's sequence "
                         "carry a whole MSM's worth of windows (D ranks -> D, at most 8; 1 below D = 4); 1 = one MSM per sequence")
    ap.add_argument("--bases", choices=("shared", "distinct"), default="shared",
                    help="window-sharded batches: every MSM of a batch names the SAME point buffer (a prover's batch over one SRS: the "
                         "engine converts it once per launch sequence) or each its own copy")
    ap.add_argument("--inputs", choices=("resident", "host"), default=None,
                    help="window-sharded runs: host (the default there) = the inputs are distributed first -- rank r uploads its n/D slice, one "
                         "all-gather per buffer over RCCL assembles the whole on every GPU (ShardedPipeline.load_host) -- timed separately as "
                         "input_distribution_ms; the timed steps then run on those buffers (should the distribution fail, the run falls back to "
                         "resident inputs and says so: input_distribution_error); resident = every rank synthesises the full inputs on its GPU (untimed)")
    ap.add_argument("--no-pipeline", action="store_true", help="N=1: await every MSM before submitting the next (latency mode)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--repeats", type=int, default=3, help="timed passes of --steps steps each; value = the median pass")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("TE_BENCH_FORCE_DIST") != "1":
        # no launcher: start the N ranks ourselves -- BEFORE torch is imported or HIP is touched in this process
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("TE_BENCH_FORCE_DIST") == "1"      # rehearsal: run the N > 1 code path with one rank over RCCL
    cpu_group = None
    share = os.environ.get("TE_BENCH_SHARE_GPU") == "1"            # rehearsal on a one-GPU box: every rank on cuda:0, exchange over gloo
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d needs WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d); got WORLD_SIZE=%d"
                         % (args.gpus, args.gpus, args.gpus, world))
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_dist and world == 1:
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("MASTER_PORT", "29533")
        if share:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            if local_rank >= torch.cuda.device_count():
                raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible (set TE_BENCH_SHARE_GPU=1 to rehearse on one GPU)"
                                 % (rank, local_rank, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # a CPU-side group: the ranks that wait while rank 0 drives all GPUs from its own process (host-buffer figure
            # below) must wait on the host -- an RCCL barrier is a kernel spinning on their GPUs
            cpu_group = None
            if world > 1:
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node; the container's hostname may not resolve
                try:
                    cpu_group = dist.new_group(backend="gloo")
                except Exception as e:                                  # the host-buffer leg is skipped, nothing else depends on it
                    print("bench.py: no CPU-side group (%s): host_buffers_ms is not measured" % e, file=sys.stderr)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus says %d" % (dist.get_world_size(), args.gpus))
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    if world > 1 and not share:
        # one process per GPU: no two ranks may resolve to the same physical device
        props = torch.cuda.get_device_properties(dev)
        tag = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or "dev%d" % dev)
        ident = torch.tensor([int(hashlib.sha256(tag.encode()).hexdigest()[:12], 16)], dtype=torch.int64, device="cuda")
        allid = [torch.zeros_like(ident) for _ in range(world)]
        dist.all_gather(allid, ident)
        ids = [int(t.item()) for t in allid]
        if len(set(ids)) != world:
            raise SystemExit("two ranks share a GPU (device uuids %s): --gpus %d needs %d distinct devices, or TE_BENCH_SHARE_GPU=1 for a rehearsal" % (ids, world, world))

    pkg = importlib.import_module(PKG)
    bls = args.curve == "bls12-377"
    assert not (bls and (world > 1 or force_dist)), "BLS12-377 is single-GPU (window sharding is Twisted-Edwards only)"
    sharded = world > 1 or force_dist
    pipelined = not args.no_pipeline
    depth = max(1, min(args.inflight or (8 if sharded else 4), pkg.WORKSETS))
    sb = 48 if bls else 32
    threads = args.cpu_threads or min(16, os.cpu_count() or 1)

    def make_inputs(log2n):
        n = 1 << log2n
        pts, sc = pkg.synth_inputs(0x5EED0000 + log2n, n, fixed_point=("chain" if (bls and args.points == "random") else args.points),   # the engine's own harness inputs
                                   curve=pkg.CURVE_BLS12_377_G1 if bls else pkg.CURVE_TE_BLS12)
        if args.scalars == "equal":
            sc = sc[:sb] * n
        elif args.scalars == "small":
            sc = b"".join(sc[sb * i:sb * i + 8] + bytes(sb - 8) for i in range(n))
        elif args.scalars == "mixed":
            # a prover's witness: a quarter zeros, a quarter ones, the rest uniform -- bucket 0 of window 0
            # holds n / 4 points, every other bucket the usual few dozen
            import numpy as np
            a = np.frombuffer(sc, dtype=np.uint8).reshape(n, sb).copy()
            a[0::4] = 0
            a[1::4] = 0
            a[1::4, 0] = 1
            sc = a.tobytes()
        return pts, sc

    n = 1 << args.log2n
    pts, sc = make_inputs(args.log2n)
    d_pts = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    d_sc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    ctx = pkg.MsmContext((dev,))
    ctx.set_option("window_bits", args.window_bits)
    if args.segment_len:
        ctx.set_option("segment_len", args.segment_len)
    ctx.set_option("signed_digits", 1 if args.digits == "signed" else 0)
    if bls:
        ctx.set_option("curve", pkg.CURVE_BLS12_377_G1)
    ctx.set_option("profile", 1)          # two HIP events around the dominant kernel, on the engine's stream
    c, W = ctx.plan(n)
    B = 1 << (c - 1 if args.digits == "signed" else c)
    rehearse = 0
    if sharded:
        # TE_BENCH_REHEARSE_WORLD=D (with TE_BENCH_FORCE_DIST=1, one rank): this rank does the work of rank 0 of D -- the
        # per-rank step of a D-GPU run without the other D-1 GPUs.  The result is a partial sum: no parity claim is made.
        rehearse = int(os.environ.get("TE_BENCH_REHEARSE_WORLD", "0")) if (force_dist and world == 1) else 0
        ctx.set_window_shard(*pkg.window_shard_for_rank(rank, rehearse or world))
        partials = torch.zeros(W * pkg.PARTIAL_BYTES, dtype=torch.uint8, device="cuda")
        gather_list = [torch.empty_like(partials) for _ in range(world)]

    def step():
        if sharded:
            return pkg.compute_msm_sharded(ctx, d_pts, d_sc, n, partials, dist, None, gather_list)
        return ctx.run_device(d_pts.data_ptr(), d_sc.data_ptr(), n)

    # a rank of D owns ~W/D windows of every MSM: too little work for a launch sequence of its own (the reduction tail and the
    # sort are latency-bound), so D MSMs share one (their windows are sorted, accumulated and reduced together)
    batch = 1
    if sharded and pipelined:
        own = -(-W // (rehearse or world))              # windows of the busiest rank: every rank must arrive at the SAME batch size
        batch = args.batch or max(1, min(pkg.MAX_BATCH, W // max(own, 1)))
        if not args.batch and batch < 4:               # measured (profiles/r02_rehearsal_per_rank_step.txt): pays from D = 4 on
            batch = 1
        if batch > 1 and not args.inflight:
            depth = 4
    pipe = pkg.ShardedPipeline(ctx, n, dist, depth=depth, batch=batch) if (sharded and pipelined) else None
    distribution_ms = None
    distribution_error = None
    inputs_mode = args.inputs or ("host" if (world > 1 and not rehearse) else "resident")
    if pipe is not None and inputs_mode == "host" and not bls:
        # SURVEY 8e "Inputs": the full inputs reach every GPU once -- each rank's own PCIe link carries n / D points, xGMI the rest.
        # The default of an N > 1 run; guarded: should the distribution fail on any rank, EVERY rank falls back to the inputs it
        # synthesised itself (the timed steps do not depend on it) and the line says so
        got = None
        ts = []
        try:
            for _ in range(3):
                dist.barrier(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                got = pipe.load_host(pts, sc)
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda" if dist.get_backend() != "gloo" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ts.append(float(tt.item()) * 1e3)
            ok_here = bool(torch.equal(got[0], d_pts) and torch.equal(got[1], d_sc))
            if not ok_here:
                distribution_error = "the assembled buffers differ from the inputs"
        except Exception as e:
            ok_here = False
            distribution_error = "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            d_pts, d_sc = got
            distribution_ms = min(ts)
        else:
            distribution_error = distribution_error or "the distribution failed on another rank"
            inputs_mode = "resident (fallback)"
        del got
    # the point buffers the MSMs of one batch name: one shared buffer, or a copy per MSM (same bytes, distinct addresses)
    base_copies = [d_pts] + ([d_pts.clone() for _ in range(batch - 1)] if (args.bases == "distinct" and batch > 1) else [])

    def batch_inputs(k):
        return [(base_copies[m % len(base_copies)], d_sc) for m in range(k)]

    result = None
    for _ in range(args.warmup):
        result = step()
    # warm-up of the pipelined form as well: every work set allocates its device buffers on first use, which must not
    # fall into the timed region (one untimed round over all of them)
    if pipelined and sharded:
        for t in [pipe.submit_batch(batch_inputs(batch)) for _ in range(depth)]:
            result = pipe.collect_batch(t)[-1]
    elif pipelined:
        for t in [ctx.submit_device(d_pts.data_ptr(), d_sc.data_ptr(), n) for _ in range(depth)]:
            result = ctx.collect(t)
    stage_acc, stage_cnt = {}, [0]

    def sync():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    def note_stage():
        try:
            st = ctx.stage_ms()                       # HIP events recorded on the engine's own stream
        except pkg.MsmError:
            return                                    # no profiled launch sequence collected yet
        for k, v in st.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v
        stage_cnt[0] += 1

    # latency of ONE synchronous MSM (device stages + read-back + host tail), before the timed throughput region; as a caller
    # gets it: without the two profiling events around the dominant kernel (~5 us of idle stream time each)
    lat = []
    ctx.set_option("profile", 0)
    for _ in range(8):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        lat.append((time.perf_counter() - t1) * 1e3)
    ctx.set_option("profile", 1)

    last = 1                                           # MSMs in the last launch sequence (window-sharded batches)

    def sharded_pass(inputs_of, pipe=pipe, batch=batch, steps=args.steps):
        """`steps` window-sharded MSMs, `depth` launch sequences of up to `batch` MSMs in flight: all-gather, read-back and
        host tail of one sequence overlap the device work of the next; returns (seconds, MSMs in the last sequence, result)"""
        res, k = None, 1
        sync()
        t0 = time.perf_counter()
        tickets, sent = [], 0
        while sent < steps:
            k = min(batch, steps - sent)
            tickets.append(pipe.submit_batch(inputs_of(k)))
            sent += k
            if len(tickets) >= depth:
                res = pipe.collect_batch(tickets.pop(0))[-1]
        while tickets:
            res = pipe.collect_batch(tickets.pop(0))[-1]
        sync()
        el = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, k, res

    distinct_line = None

    def timed_pass():
        """EXACTLY args.steps MSMs, bracketed by barrier + synchronize on both sides, max over ranks: (seconds, result)"""
        if pipelined and sharded:
            el, k, res = sharded_pass(batch_inputs)
            return el, res, k
        sync()
        t0 = time.perf_counter()
        if pipelined:
            # K independent MSMs back to back, `depth` in flight on as many streams: host tail and device work of consecutive
            # MSMs overlap, and on the GPU the gaps and latency-bound tail of one are filled by the wide kernels of another
            _, res = pipelined_pass(ctx, d_pts.data_ptr(), d_sc.data_ptr(), n, args.steps, depth, note_stage)
        else:
            res = None
            for _ in range(args.steps):
                res = step()
                note_stage()
        sync()
        el = time.perf_counter() - t0
        if sharded:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if dist.get_backend() != "gloo" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, res, 1

    # The timed region runs args.repeats (3) times, each EXACTLY args.steps steps; `value` / `ms_per_step` are the MEDIAN pass
    # (the driver's 20 steps are 20 ms of GPU time: one pass alone carries the box's clock wobble straight into `value`); all
    # passes are in "passes_ms_per_step".
    passes = []
    for _ in range(max(1, args.repeats)):
        el, result, last = timed_pass()
        passes.append(el)
    if pipelined and sharded:
        note_stage()                                   # events of the last launch sequence (`last` MSMs): a sample, not the mean
    elapsed = sorted(passes)[len(passes) // 2]
    if pipelined and sharded and batch > 1 and len(base_copies) == 1:
        # the same again with a point buffer of its own per MSM of a batch (same bytes at distinct addresses): a shared
        # buffer is converted once per launch sequence, distinct ones once each -- both figures belong in the line
        copies = [d_pts] + [d_pts.clone() for _ in range(batch - 1)]
        torch.cuda.synchronize()
        el2, _, res2 = sharded_pass(lambda k: [(copies[m], d_sc) for m in range(k)])
        assert res2 == result
        distinct_line = {"ms_per_step": el2 * 1e3 / args.steps, "value": args.steps / el2,
                         "note": "every MSM of a batch names its own copy of the point buffer: %d conversions per launch sequence instead of one" % batch}
        del copies

    ms_per_step = elapsed * 1e3 / args.steps
    # the dominant kernel in the timed region, per LAUNCH (a launch carries `last` MSMs in a window-sharded batch): means over
    # the sampled launches -- a span on the device clock (first wave in .. last wave out) and the HIP-event interval around
    # the launch.  With `depth` MSMs in flight two accumulations usually share the GPU: these spans are stretched by the
    # concurrency and can exceed ms_per_step; the roofline uses the alone figure below.
    cnt = max(stage_cnt[0], 1)
    timed = {"kernel_ms_span_device_clock": stage_acc.get("accumulate_on_device", 0.0) / cnt or None,
             "kernel_ms_event_interval": stage_acc.get("accumulate", 0.0) / cnt or None,
             "core_clock_ghz": stage_acc.get("accumulate_core_clock_ghz", 0.0) / cnt or None,
             "launches_sampled": stage_cnt[0], "msms_per_launch": last, "in_flight": depth if pipelined else 1,
             "note": "concurrency-stretched: %d launch sequences share the GPU" % (depth if pipelined else 1)}
    # full per-stage breakdown from a few extra, untimed steps with the GPU to itself (an event at every stage boundary)
    stage_ms = alone_pass(ctx, step, 6)
    # the points this rank's accumulation gathers per MSM: non-zero digits of ITS windows, counted by the engine
    entries_rank = ctx.get_option("entries_accumulated")
    div = rehearse or world
    whole_bytes, _ = algorithmic_bytes(n, W, B, bls, entries_rank * div)       # (whole MSM: the ranks' windows are alike to 2^-c)
    _, acc_bytes_rank = algorithmic_bytes(n, W / div, B, bls, entries_rank)    # windows are sharded
    traffic, traffic_info = (None, None) if bls else measured_traffic(
        args.log2n, c, world if not rehearse else rehearse,
        args.digits == "signed" and args.scalars == "uniform" and args.points == "random" and not args.segment_len)

    out = {
        "metric": "MSMs/sec at n=2^%d %s (pipelined throughput = 1000/ms_per_step; single-MSM latency in latency_ms)" % (args.log2n, "BLS12-377 G1" if bls else "Twisted-Edwards BLS12"),
        "value": args.steps / elapsed,
        "unit": "MSM/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "passes_ms_per_step": [e * 1e3 / args.steps for e in passes],
        "value_is": "median of %d timed passes of %d steps each" % (len(passes), args.steps),
        "latency_ms": min(lat),
        "latency_ms_single_msm": min(lat),
        "host_buffers_ms": None,
        "mode": (("pipelined: %d MSMs in flight (te_msm_submit_device / te_msm_collect)" if world == 1 and not sharded else ("pipelined: %%d launch sequences of %d window-sharded MSM(s) each in flight per rank" % batch)) % depth) if pipelined else "synchronous: one MSM at a time",
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": "n=2^%d %s MSM, %d-bit %s windows (%d windows x %d buckets), points=%s, scalars=%s, inputs resident in HBM"
                               % (args.log2n, "BLS12-377 G1" if bls else "TE-BLS12", c, args.digits, W, B, args.points, args.scalars),
                   "points_note": {"random": "SURVEY 8d set (R): n independent subgroup points a_i*G, a_i seeded-random (te_msm_synth_inputs, fixed-base table on the host cores)",
                                   "chain": "n distinct subgroup points (a + i*b)*G, an arithmetic progression (te_msm_synth_inputs)",
                                   "fixed": "harness mode (H): one fixed point replicated (ui/AllBenchmarks.tsx:105-112)"}[args.points],
                   "parallelism": "windows sharded over %d rank(s), one %s all-gather of %d B partial sums per launch sequence of %d MSM(s), %s point buffers"
                                  % (world, "RCCL" if (world > 1 and dist.get_backend() == "nccl") else "gloo (rehearsal: ranks share one GPU)", batch * W * 720, batch,
                                     "one shared" if len(base_copies) == 1 else "distinct") if world > 1 else "single GPU"},
        "roofline": roofline_block(acc_bytes_rank, stage_ms, timed, bls, entries_rank / 64.0, traffic, traffic_info, entries=entries_rank),
        "msm_algorithmic_bytes": whole_bytes,
        "msm_algorithmic_gbps": whole_bytes / (ms_per_step * 1e-3) / 1e9,
        "stage_ms_untimed_pass": {k: v for k, v in stage_ms.items() if not k.endswith("_ghz")},
        "result_x": str(int.from_bytes(result[:32], "little")),
    }
    if distribution_error:
        out["input_distribution_error"] = distribution_error
    out["inputs"] = inputs_mode
    if pipe is not None and distribution_ms is None and not distribution_error and not bls and world > 1:
        # resident inputs (the default): the distribution is still measured, as a side figure behind the timed region -- guarded,
        # it must never cost the line -- and its result compared with the buffers the steps ran on
        try:
            ts = []
            for _ in range(3):
                dist.barrier(); torch.cuda.synchronize()
                t1 = time.perf_counter()
                hp, hs = pipe.load_host(pts, sc)
                torch.cuda.synchronize()
                tt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda" if dist.get_backend() != "gloo" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                ts.append(float(tt.item()) * 1e3)
            distribution_ms = min(ts)
            out["input_distribution_parity"] = "assembled buffers == the resident inputs" if (torch.equal(hp, d_pts) and torch.equal(hs, d_sc)) else "MISMATCH"
            del hp, hs
        except Exception as e:
            out["input_distribution_error"] = "%s: %s" % (type(e).__name__, e)
    if distribution_ms is not None:
        out["input_distribution_ms"] = distribution_ms
        out["input_distribution_note"] = ("ShardedPipeline.load_host, best of 3, max over ranks: each rank uploads %d of the %d points (+ scalars) from pageable host "
                                          "memory and two all-gathers (points, scalars) assemble the %d MB on every GPU; not part of the timed steps" % (
                                              (n + world - 1) // world, n, (len(pts) + len(sc)) >> 20))
    if sharded:
        out["rccl_ranks"] = dist.get_world_size()
        out["backend"] = dist.get_backend()
        out["batch_bases"] = "shared" if len(base_copies) == 1 else "distinct"
        if distinct_line:
            out["batch_distinct_bases"] = distinct_line
        # per-rank dominant-kernel time (alone pass), gathered so that rank 0 reports all of them
        km = torch.tensor([stage_ms.get("accumulate", 0.0)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        kms = [torch.zeros_like(km) for _ in range(world)]
        dist.all_gather(kms, km)
        out["per_rank_kernel_ms"] = [float(x.item()) for x in kms]

    def size_rate(entry, m, cbits, Wn):
        """MSM/s and the whole MSM's algorithmic bytes (SURVEY 8d, W * n gathered points) over 8 TB/s for a `sizes` entry"""
        whole_m, _ = algorithmic_bytes(m, Wn, 1 << (cbits - 1 if args.digits == "signed" else cbits), bls)
        entry["msm_per_s"] = 1e3 / entry["ms_per_step"]
        entry["msm_algorithmic_hbm_frac"] = whole_m / (entry["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS

    def host_buffer_ms(cx, p, s, reps=3, all_times=None):
        """one te_msm_run from pageable host buffers, default options (profile 0: the chunked-upload path of te_msm_run)"""
        prof = cx.get_option("profile")
        cx.set_option("profile", 0)
        ts = []
        try:
            for _ in range(reps):
                t1 = time.perf_counter()
                r = cx.run(p, s)
                ts.append((time.perf_counter() - t1) * 1e3)
        finally:
            cx.set_option("profile", prof)
        if all_times is not None:
            all_times.extend(ts)
        return min(ts), r

    if rank == 0 and world == 1 and not sharded and not args.no_host_buffers:
        # the boundary hands over host buffers (te_msm_run): PCIe-inclusive latency, reported but never `value`
        hb, r_host = host_buffer_ms(ctx, pts, sc)
        out["host_buffers_ms"] = hb
        out["pcie_inclusive_ms_host_buffers"] = hb
        out["host_buffers_path"] = ("te_msm_run, profile 0, host_chunks=%d (0 = automatic: the buffers are uploaded and processed in 3 pieces "
                                    "from 3 * 2^18 points, 2 from 2^17, whole below)" % ctx.get_option("host_chunks"))
        out["host_buffers_gbps"] = (len(pts) + len(sc)) / (hb * 1e-3) / 1e9
        assert r_host == result
        # the same boundary with several calls in flight (te_msm_submit: what concurrent compute_msm promises map onto): the upload
        # of MSM k+1 overlaps the device work of MSM k -- bound by the one PCIe link
        ctx.set_option("profile", 0)
        infl = pkg.WORKSETS
        for t in [ctx.submit_async(pts, sc) for _ in range(infl)]:
            assert ctx.collect(t) == result
        in_flight_passes = []
        for _ in range(3):
            t1 = time.perf_counter()
            tk = []
            for _ in range(32):
                tk.append(ctx.submit_async(pts, sc))
                if len(tk) >= infl:
                    assert ctx.collect(tk.pop(0)) == result
            while tk:
                assert ctx.collect(tk.pop(0)) == result
            in_flight_passes.append((time.perf_counter() - t1) * 1e3 / 32)
        out["host_buffers_in_flight_ms"] = min(in_flight_passes)
        out["host_buffers_in_flight_note"] = ("best of 3 passes of 32 te_msm_submit_async calls from pageable host buffers (what the N-API addon turns concurrent "
                                              "compute_msm promises into), %d in flight, %d upload threads, per MSM; passes: %s" % (
                                                  infl, ctx.get_option("upload_threads"), " ".join("%.3f" % x for x in in_flight_passes)))
        ctx.set_option("profile", 1)
        # resident bases: the same boundary with the points bound once (opt-in beside compute_msm: setBases in the addon)
        br = bases_resident_figures(ctx, pts, sc, result, d_sc.data_ptr(), n, depth)
        out["bases_resident"] = br
        out["bases_resident_in_flight_ms"] = br["in_flight_ms"]
        out["bases_resident_latency_ms"] = br["latency_ms"]
        out["bases_resident_device_scalars_ms"] = br.get("device_scalars_ms")
        out["bases_resident_note"] = ("te_msm_bind_points once (%.1f ms), then te_msm_submit_scalars (8 in flight, per MSM) / te_msm_run_scalars (one at a time) "
                                      "from pageable host scalars: 32 of the 96 bytes per point cross PCIe, no conversion; device_scalars: scalars in HBM, %d in flight"
                                      % (br["bind_ms"], depth))
        out["sizes"] = {str(args.log2n): {"ms_per_step": ms_per_step, "latency_ms": min(lat), "host_buffers_ms": hb, "window_bits": c,
                                          "core_clock_ghz": stage_ms.get("accumulate_core_clock_ghz"),
                                          "bases_resident_in_flight_ms": br["in_flight_ms"], "bases_resident_latency_ms": br["latency_ms"]}}
        size_rate(out["sizes"][str(args.log2n)], n, c, W)
        if not args.no_sizes and not bls and args.log2n == 20:
            # the other harness sizes (full_benchmarks.ts:13-15), short runs, window size chosen by the engine, on the same
            # context (its work sets own the buffers already)
            sx = ctx
            sx.set_option("window_bits", 0)
            for lg in (16, 17, 18, 19):
                m = 1 << lg
                p2, s2 = make_inputs(lg)
                dp2 = torch.frombuffer(bytearray(p2), dtype=torch.uint8).cuda()
                ds2 = torch.frombuffer(bytearray(s2), dtype=torch.uint8).cuda()
                torch.cuda.synchronize()
                sx.set_option("profile", 0)
                for t in [sx.submit_device(dp2.data_ptr(), ds2.data_ptr(), m) for _ in range(depth)]:
                    ref = sx.collect(t)
                l2 = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    assert sx.run_device(dp2.data_ptr(), ds2.data_ptr(), m) == ref
                    l2.append((time.perf_counter() - t1) * 1e3)
                el, _ = pipelined_pass(sx, dp2.data_ptr(), ds2.data_ptr(), m, 40, depth)
                hb2, r2 = host_buffer_ms(sx, p2, s2)
                assert r2 == ref
                sx.set_option("profile", 1)            # the kernel stamps its own clock: one more MSM for the core clock at this size
                assert sx.run_device(dp2.data_ptr(), ds2.data_ptr(), m) == ref
                ghz2 = sx.stage_ms().get("accumulate_core_clock_ghz")
                br2 = bases_resident_figures(sx, p2, s2, ref, None, m, depth)
                out["sizes"][str(lg)] = {"ms_per_step": el * 1e3 / 40, "latency_ms": min(l2), "host_buffers_ms": hb2, "window_bits": sx.plan(m)[0],
                                         "core_clock_ghz": ghz2,
                                         "bases_resident_in_flight_ms": br2["in_flight_ms"], "bases_resident_latency_ms": br2["latency_ms"]}
                size_rate(out["sizes"][str(lg)], m, *sx.plan(m))
            sx.set_option("window_bits", args.window_bits)
            sx.set_option("profile", 1)
    if pipe is not None and not rehearse and not bls and world > 1 and not args.no_sizes and args.log2n == 20:
        # the other harness sizes on N GPUs (the north star asks for n = 2^16 .. 2^20 at every GPU count; full_benchmarks.ts:13-15
        # runs 16..20): short passes, window size chosen by the engine, same sharding and batching rule, every rank's result
        # compared with the oracle-checked single-GPU engine on that rank's own GPU
        out["sizes"] = {str(args.log2n): {"ms_per_step": ms_per_step, "window_bits": c, "batch": batch}}
        size_rate(out["sizes"][str(args.log2n)], n, c, W)
        ctx.set_option("window_bits", 0)
        for lg in (16, 17, 18, 19):
            m = 1 << lg
            p2, s2 = make_inputs(lg)
            dp2 = torch.frombuffer(bytearray(p2), dtype=torch.uint8).cuda()
            ds2 = torch.frombuffer(bytearray(s2), dtype=torch.uint8).cuda()
            torch.cuda.synchronize()
            c2, W2 = ctx.plan(m)
            own2 = -(-W2 // world)                      # (17 windows over 4 ranks: 5 + 4 + 4 + 4 -- the busiest rank decides, for all)
            b2 = args.batch or max(1, min(pkg.MAX_BATCH, W2 // max(own2, 1)))
            if not args.batch and b2 < 4:
                b2 = 1
            p2pipe = pkg.ShardedPipeline(ctx, m, dist, depth=depth, batch=b2)
            one = lambda k: [(dp2, ds2)] * k
            for t in [p2pipe.submit_batch(one(b2)) for _ in range(depth)]:
                p2pipe.collect_batch(t)
            el2, _, r2 = sharded_pass(one, p2pipe, b2, 64)
            with pkg.MsmContext((dev,)) as solo:
                same2 = solo.run_device(dp2.data_ptr(), ds2.data_ptr(), m) == r2
            out["sizes"][str(lg)] = {"ms_per_step": el2 * 1e3 / 64, "window_bits": c2, "batch": b2, "parity_this_rank": "identical to the single-GPU result" if same2 else "MISMATCH"}
            size_rate(out["sizes"][str(lg)], m, c2, W2)
            del p2pipe, dp2, ds2
        ctx.set_option("window_bits", args.window_bits)
    if world > 1 and not rehearse and not bls and not args.no_host_buffers and (share or cpu_group is not None):
        # compute_msm(Buffer, Buffer) on all N GPUs from ONE process: rank 0 opens an n_dev = N context (te_msm_run: point
        # slices, one upload thread and PCIe link per device) while the other ranks wait -- the reference's boundary (host buffers,
        # upload inside the call: cuzk/gpu.ts:33-46) on N devices.  PCIe-inclusive: reported, never `value`.
        sync()
        # (rehearsal with every rank on one GPU, TE_BENCH_SHARE_GPU=1: the one device named `world` times -- the code path, not a speed-up)
        host_ids = tuple([dev] * world) if share else tuple(range(world))
        if rank == 0 and (share or torch.cuda.device_count() >= world):
            try:
                with pkg.MsmContext((dev,)) as one:
                    one.set_option("signed_digits", 1 if args.digits == "signed" else 0)
                    one.run(pts, sc)
                    hb1, r1 = host_buffer_ms(one, pts, sc, reps=5)
                with pkg.MsmContext(host_ids) as mc:
                    mc.set_option("signed_digits", 1 if args.digits == "signed" else 0)
                    r_host = mc.run(pts, sc)                      # buffers, staging areas, the per-device host threads
                    # the same boundary with MSMs in flight: whole-MSM tickets, one per device and upload thread
                    # (te_msm_submit_async: what concurrent compute_msm promises map onto under the N-API addon)
                    infl = 2 * world
                    for t in [mc.submit_async(pts, sc) for _ in range(infl)]:          # every work set's buffers and staging area
                        assert mc.collect(t) == r_host
                    # the first ~9 calls of a PROCESS in which eight threads copy at once carry stalls of the runtime's copy path
                    # (DESIGN.md section 6): the lone call is timed behind them
                    for _ in range(8):
                        mc.run(pts, sc)
                    lone = []
                    hb, r_host = host_buffer_ms(mc, pts, sc, reps=7, all_times=lone)
                    out["host_buffers_ms_median"] = sorted(lone)[len(lone) // 2]
                    cb, Wb = mc.plan((n + world - 1) // world)
                    k_in = 6 * world
                    t1 = time.perf_counter()
                    tk = []
                    for _ in range(k_in):
                        tk.append(mc.submit_async(pts, sc))
                        if len(tk) >= infl:
                            assert mc.collect(tk.pop(0)) == r_host
                    while tk:
                        assert mc.collect(tk.pop(0)) == r_host
                    out["host_buffers_in_flight_ms"] = (time.perf_counter() - t1) * 1e3 / k_in
                    out["host_buffers_in_flight_note"] = ("%d te_msm_submit_async calls from pageable host buffers on the n_dev = %d context, %d in flight "
                                                          "(two whole MSMs per device, one upload thread and PCIe link per device), per MSM" % (k_in, world, infl))
                out["host_buffers_ms"] = hb
                out["pcie_inclusive_ms_host_buffers"] = hb
                out["host_buffers_ms_one_device"] = hb1
                out["host_buffers_path"] = ("te_msm_run on ONE n_dev = %d context in rank 0's process: %d point slices of %d points, %d-bit windows, "
                                            "one host thread + PCIe link per device, rows summed in the host tail" % (world, world, (n + world - 1) // world, cb))
                out["host_buffers_parity"] = "identical to the window-sharded result" if (r_host == result and r1 == result) else "MISMATCH"
                out["host_buffers_devices"] = list(host_ids)
            except Exception as e:                                   # this leg is a side figure: it must never cost the line
                out["host_buffers_error"] = "%s: %s" % (type(e).__name__, e)
            torch.cuda.set_device(dev)                                # (the C-ABI restores the caller's device itself since round 5; belt and braces)
        if cpu_group is not None:
            dist.barrier(group=cpu_group)                         # ranks 1.. wait here, on the host
        elif share:
            dist.barrier()                                        # gloo: a CPU-side wait already
        sync()
    exp = None
    if rank == 0 and world == 1 and not sharded and not args.no_cpu_baseline:
        t0 = time.perf_counter()
        if bls:
            from oracle import oracle377       # cpu_baseline leg (and result checker) only
            exp = oracle377.msm(pts, sc, c=16 if n >= 65536 else 4, threads=threads)
        else:
            from oracle import oracle          # cpu_baseline leg (and result checker) only
            exp = oracle.msm(pts, sc, c=16 if n >= 65536 else 4, bpr_mode=1, threads=threads)
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "MSM/s", "cores": threads, "kind": "port",
                               "host_cpu_count": os.cpu_count(),
                               "sample": "1 full MSM at n=2^%d (C restatement of the reference pipeline, oracle/%s), %.1f s"
                                         % (args.log2n, "bls377_oracle.c; parity unpinned by the reference" if bls else "te_oracle.c", cpu_s)}
        if not bls and args.log2n >= 18:
            # the same restatement on ONE core ("also on 1 core"), on a bounded sample: the first 2^18 points of the
            # workload (a whole 2^20 MSM is ~15 s of one core); the per-window bucket reduction (2 x 2^15 additions against
            # 2^18) is then counted 2^(log2n - 18) times when scaled to n: a slight under-estimate of the one-core rate
            m1 = 1 << 18
            t0 = time.perf_counter()
            oracle.msm(pts[:64 * m1], sc[:32 * m1], c=16, bpr_mode=1, threads=1)
            one_s = time.perf_counter() - t0
            out["cpu_baseline"]["one_core"] = {"value": 1.0 / (one_s * (n / m1)), "unit": "MSM/s", "cores": 1,
                                               "sample": "first 2^18 points of the workload, 1 thread: %.2f s; x%d to n=2^%d" % (one_s, n // m1, args.log2n)}
        out["parity"] = "bit-exact vs oracle" if exp == result else "MISMATCH vs oracle"
        if exp != result:
            print(json.dumps(out))
            raise SystemExit("GPU result differs from the oracle")
    if "parity" not in out and not sharded:
        out["parity"] = "not checked (--no-cpu-baseline)"
    default_run = (rank == 0 and world == 1 and not sharded and not bls and args.log2n == 20 and args.digits == "signed"
                   and args.scalars == "uniform" and args.points == "random" and pipelined)
    if default_run and not args.no_configs and not args.no_cpu_baseline:
        # BASELINE configs 2 and 5 in the driver's one line: short passes, each checked against its oracle
        ctx.close()
        ctx = None
        out["configs"] = {}
        bad = []
        cu, ok = side_config(pkg, dev, "unsigned", "te", "unsigned", 20, depth, threads, 40, expect=exp, inputs=(pts, sc))
        out["configs"]["unsigned"] = cu
        if not ok:
            bad.append("unsigned")
        cb, ok = side_config(pkg, dev, "bls12_377", "bls12-377", "signed", 20, depth, threads, 24)
        out["configs"]["bls12_377"] = cb
        if not ok:
            bad.append("bls12_377")
        # harness mode (H): the UI's random mode replicates ONE point n times (ui/AllBenchmarks.tsx:105-112), same scalars
        ph, _ = pkg.synth_inputs(0x5EED0000 + 20, n, fixed_point="fixed", scalars=False)
        chh, ok = side_config(pkg, dev, "harness_fixed_point", "te", "signed", 20, depth, threads, 40, inputs=(ph, sc),
                              workload_note=", harness mode: one fixed point replicated")
        out["configs"]["harness_fixed_point"] = chh
        if not ok:
            bad.append("harness_fixed_point")
        # SURVEY 8f rank 3, a prover's witness: a quarter zeros, a quarter ones, the rest uniform (bucket 0 of window 0 holds n / 4 points)
        import numpy as np
        aw = np.frombuffer(sc, dtype=np.uint8).reshape(n, sb).copy()
        aw[0::4] = 0
        aw[1::4] = 0
        aw[1::4, 0] = 1
        cw, ok = side_config(pkg, dev, "witness_scalars", "te", "signed", 20, depth, threads, 40, inputs=(pts, aw.tobytes()),
                             workload_note=", scalars: a quarter zeros, a quarter ones, the rest uniform")
        out["configs"]["witness_scalars"] = cw
        if not ok:
            bad.append("witness_scalars")
        del aw
        # set (R) against the arithmetic progression the earlier rounds timed: the same kernels on equally spread field elements
        pc, _ = pkg.synth_inputs(0x5EED0000 + 20, n, fixed_point="chain", scalars=False)
        dpc = torch.frombuffer(bytearray(pc), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        with pkg.MsmContext((dev,)) as cx:
            cx.set_option("window_bits", args.window_bits)
            for t in [cx.submit_device(dpc.data_ptr(), d_sc.data_ptr(), n) for _ in range(depth)]:
                cx.collect(t)
            elc, _ = pipelined_pass(cx, dpc.data_ptr(), d_sc.data_ptr(), n, 40, depth)
            elr, _ = pipelined_pass(cx, d_pts.data_ptr(), d_sc.data_ptr(), n, 40, depth)
        out["points_chain_vs_random"] = {"chain_ms_per_step": elc * 1e3 / 40, "random_ms_per_step": elr * 1e3 / 40,
                                         "note": "40 steps each on one fresh context, back to back: the arithmetic-progression points of rounds 1-3 and set (R) time the same"}
        del dpc
        if bad:
            print(json.dumps(out))
            raise SystemExit("GPU result differs from the oracle in configs: %s" % bad)
    if sharded and rehearse:
        out["parity"] = "not checked: rehearsal of rank 0 of %d (partial sum)" % rehearse
        out["config"]["parallelism"] = "REHEARSAL: the per-rank step of a %d-GPU run on one GPU (%s point buffers per batch)" % (
            rehearse, "one shared" if len(base_copies) == 1 else "distinct")
    elif sharded:
        # untimed cross-check of the sharded path: the same MSM on this rank's GPU alone must give the same point
        with pkg.MsmContext((dev,)) as solo:
            solo.set_option("window_bits", args.window_bits)
            solo.set_option("signed_digits", 1 if args.digits == "signed" else 0)
            same = solo.run_device(d_pts.data_ptr(), d_sc.data_ptr(), n) == result
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda")
        if dist.get_backend() == "gloo":
            flag = flag.cpu()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        out["parity"] = "identical to the single-GPU result on every rank" if int(flag.item()) == 1 else "MISMATCH vs single-GPU result"
        if int(flag.item()) != 1:
            if rank == 0:
                print(json.dumps(out))
            raise SystemExit("window-sharded result differs from the single-GPU result")
    # the driver's record keeps `config`, `roofline` and `cpu_baseline` of this line: the figures of the boundary go there too
    for k in ("latency_ms", "host_buffers_ms", "host_buffers_in_flight_ms", "host_buffers_ms_one_device", "bases_resident_in_flight_ms",
              "bases_resident_latency_ms", "bases_resident_device_scalars_ms", "sizes"):
        if out.get(k) is not None:
            out["config"][k] = out[k]
    if rank == 0:
        print(json.dumps(out))
    if ctx is not None:
        ctx.close()
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
