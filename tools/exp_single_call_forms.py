"""One call at a time at n = 2^20 (the reference's harness awaits each call, ui/Benchmark.tsx:32): the synchronous calls against the same MSM as a ticket
that is collected at once (what the N-API addon does for a lone promise), medians of 40, ms.  python tools/exp_single_call_forms.py"""
import importlib, statistics, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")


def latency(f, reps=40):
    for _ in range(5):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(ts)


with pkg.MsmContext((0,)) as c:
    b = c.bind_points(pts)
    for rnd in range(3):
        print("round %d  run %.3f  collect(submit) %.3f  collect(submit_async) %.3f   run_scalars %.3f  collect(submit_scalars) %.3f" % (
            rnd, latency(lambda: c.run(pts, sc)), latency(lambda: c.collect(c.submit(pts, sc))), latency(lambda: c.collect(c.submit_async(pts, sc))),
            latency(lambda: c.run_scalars(b, sc)), latency(lambda: c.collect(c.submit_scalars(b, sc)))), flush=True)
