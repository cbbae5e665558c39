"""te_msm_run / te_msm_submit / te_msm_run_scalars from PAGEABLE host buffers (the calling thread uploads): stream waits behind events recorded on
the copy stream (TE_MSM_CALLER_HOST_WAITS=0) against the build (the thread waits for the copy stream; no event, no stream wait in a hardware queue).
Child processes, alternating, three rounds; n = 2^20 and 2^18.
python tools/exp_caller_host_waits.py            (parent)
python tools/exp_caller_host_waits.py child"""
import importlib, os, statistics, subprocess, sys, time


def child():
    sys.path.insert(0, '.')
    pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
    out = []
    for lg in (20, 18):
        n = 1 << lg
        pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="random")

        def latency(f, reps=40):
            for _ in range(5):
                f()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
            return statistics.median(ts)

        def in_flight(c, submit, depth, steps=48):
            for t in [submit() for _ in range(depth)]:
                c.collect(t)
            ps = []
            for _ in range(3):
                t0 = time.perf_counter(); tk = []
                for _ in range(steps):
                    tk.append(submit())
                    if len(tk) >= depth:
                        c.collect(tk.pop(0))
                while tk:
                    c.collect(tk.pop(0))
                ps.append((time.perf_counter() - t0) * 1e3 / steps)
            return min(ps)

        with pkg.MsmContext((0,)) as c:
            b = c.bind_points(pts)
            want = c.run(pts, sc)
            assert c.run_scalars(b, sc) == want and c.collect(c.submit(pts, sc)) == want
            out.append("2^%d: run %.3f  run_scalars %.3f  submit x8 %.3f  x4 %.3f" % (
                lg, latency(lambda: c.run(pts, sc)), latency(lambda: c.run_scalars(b, sc)), in_flight(c, lambda: c.submit(pts, sc), 8), in_flight(c, lambda: c.submit(pts, sc), 4)))
    print("   ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rnd in range(3):
            for name, env in (("stream waits", {"TE_MSM_CALLER_HOST_WAITS": "0"}), ("host waits", {"TE_MSM_CALLER_HOST_WAITS": "1"})):
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=300)
                print("round %d %-12s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
