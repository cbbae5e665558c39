"""One-GPU box: the per-thread stamps of the in-process multi-"device" te_msm_run (TE_MSM_TRACE_HOST=1): for every device
thread of every call "scalars staged", "points staged i", "piece enqueued i" with the time since its slice started and the
absolute time.  Prints, per call, its duration and the slowest single staging step -- what round 4's 9.5 ms median of the
D = 8 call (best 3.8) has to be explained from.   python tools/trace_point_shards.py [D] [log2n] [calls] 2> stamps.txt"""
import importlib, os, sys, time
os.environ["TE_MSM_TRACE_HOST"] = "1"
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
D = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lg = int(sys.argv[2]) if len(sys.argv) > 2 else 20
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = 1 << lg
pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)
with pkg.MsmContext((0,) * D) as c:
    for k in (int(x) for x in os.environ.get("TE_TRACE_OPTS", "").split(",") if x):
        c.set_option("host_chunks", k)
    c.run(pts, sc); c.run(pts, sc)
    for i in range(calls):
        sys.stderr.write("==== call %d begins (t = %.1f us)\n" % (i, time.monotonic() * 1e6)); sys.stderr.flush()
        t0 = time.perf_counter(); c.run(pts, sc); dt = (time.perf_counter() - t0) * 1e3
        sys.stderr.write("==== call %d took %.3f ms\n" % (i, dt)); sys.stderr.flush()
        print("call %2d: %.3f ms" % (i, dt))
