"""Rehearsal of te_msm_run on D devices (point shards) on a ONE-GPU box: what ONE device of D does -- its slice of n / D points
from host buffers, all windows (te_msm_run of n / D points) -- next to the whole call on one device, plus the in-process
multi-"device" call with the one GPU named D times (D threads, D uploads sharing ONE link and one GPU: correctness and the
host-side overhead of threads and the D-set host tail, not a speed-up).  Also the node-visible pipelined form:
k te_msm_submit tickets in flight.   Run on the GPU box:  python tools/rehearse_point_shards.py [log2n]"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)


def timed(f, reps=9):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return r, ts[0], ts[len(ts) // 2]


with pkg.MsmContext((0,)) as c:
    ref = c.run(pts, sc)
    _, best, med = timed(lambda: c.run(pts, sc))
    print("n = 2^%d  one device, whole call: best %.3f ms, median %.3f ms" % (lg, best, med))
    whole = med
    for D in (2, 4, 8):
        m = n // D
        p1, s1 = pts[:64 * m], sc[:32 * m]                 # sliced once, outside the timed call
        c.run(p1, s1)
        _, best, med = timed(lambda: c.run(p1, s1))
        cb, W = c.plan(m)
        print("  one device's share at D = %d: %7d points, %d-bit windows: best %.3f ms, median %.3f ms  -> %.2fx the whole call (before the D-set host tail)"
              % (D, m, cb, best, med, whole / med))
    # tickets in flight from host buffers (what concurrent compute_msm promises get)
    for k in (1, 2, 4):
        def go():
            ts = [c.submit(pts, sc) for _ in range(k)]
            return [c.collect(t) for t in ts]
        go()
        r, best, med = timed(go, 7)
        assert all(x == ref for x in r)
        print("  %d te_msm_submit ticket(s) in flight: best %.3f ms, median %.3f ms per MSM" % (k, best / k, med / k))
    # the same with the upload on the device's host thread (te_msm_submit_async: what the N-API addon uses since round 5)
    for k in (2, 4, 8):
        def go_async():
            ts = [c.submit_async(pts, sc) for _ in range(k)]
            return [c.collect(t) for t in ts]
        go_async()
        r, best, med = timed(go_async, 7)
        assert all(x == ref for x in r)
        print("  %d te_msm_submit_async ticket(s) in flight: best %.3f ms, median %.3f ms per MSM" % (k, best / k, med / k))
    for lanes in (1, 2, 4, 8):
        c.set_option("upload_threads", lanes)
        def go_lanes():
            tk, out = [], []
            for _ in range(32):
                tk.append(c.submit_async(pts, sc))
                if len(tk) >= 8:
                    out.append(c.collect(tk.pop(0)))
            while tk:
                out.append(c.collect(tk.pop(0)))
            return out
        go_lanes()
        r, best, med = timed(go_lanes, 5)
        assert all(x == ref for x in r)
        print("  upload_threads = %d: 32 te_msm_submit_async tickets, 8 in flight: best %.3f ms, median %.3f ms per MSM" % (lanes, best / 32, med / 32))
    c.set_option("upload_threads", 4)
    def stream_async(total=24, depth=4):
        tk, out = [], []
        for _ in range(total):
            tk.append(c.submit_async(pts, sc))
            if len(tk) >= depth:
                out.append(c.collect(tk.pop(0)))
        while tk:
            out.append(c.collect(tk.pop(0)))
        return out
    r, best, med = timed(stream_async, 5)
    assert all(x == ref for x in r)
    print("  24 te_msm_submit_async tickets, 4 in flight, collected as they come: best %.3f ms, median %.3f ms per MSM" % (best / 24, med / 24))

for D in (2, 4, 8):
    with pkg.MsmContext((0,) * D) as c:
        assert c.run(pts, sc) == ref
        _, best, med = timed(lambda: c.run(pts, sc), 7)
        print("in-process te_msm_run, the one GPU named %d times (D uploads share one link): best %.3f ms, median %.3f ms" % (D, best, med))

# ---- round 5: whole-MSM tickets on D "devices" (te_msm_submit_async: one upload thread per device; on this box they share one
# link and one GPU: the code path and its host-side cost, not a speed-up), and every single time of the D = 8 lone call
for D in (2, 4, 8):
    with pkg.MsmContext((0,) * D) as c:
        def go():
            ts = [c.submit_async(pts, sc) for _ in range(2 * D)]
            return [c.collect(t) for t in ts]
        assert all(x == ref for x in go())
        r, best, med = timed(go, 5)
        print("tickets on the one GPU named %d times, %d whole MSMs in flight (te_msm_submit_async): best %.3f ms, median %.3f ms per MSM" % (D, 2 * D, best / (2 * D), med / (2 * D)))
with pkg.MsmContext((0,) * 8) as c:
    c.run(pts, sc)
    ts = []
    for _ in range(25):
        t0 = time.perf_counter(); c.run(pts, sc); ts.append((time.perf_counter() - t0) * 1e3)
    print("in-process te_msm_run, D = 8, 25 calls in order (ms):", " ".join("%.2f" % t for t in ts))
