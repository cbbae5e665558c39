"""host stamps (TE_MSM_TRACE_HOST=1) of bound-bases tickets from host scalars in flight: where does a lane thread spend its time?
TE_MSM_TRACE_HOST=1 python tools/trace_bound_tickets.py [upload_threads [depth [steps]]] 2> stamps.txt"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
n = 1 << 20
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
with pkg.MsmContext((0,)) as c:
    c.set_option("upload_threads", int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    b = c.bind_points(pts)
    want = c.run_scalars(b, sc)
    for t in [c.submit_scalars(b, sc) for _ in range(depth)]:
        c.collect(t)
    sys.stderr.write("==== timed loop\n"); sys.stderr.flush()
    t0 = time.perf_counter(); tk = []
    for _ in range(steps):
        tk.append(c.submit_scalars(b, sc))
        if len(tk) >= depth:
            c.collect(tk.pop(0))
    while tk:
        c.collect(tk.pop(0))
    print("%.4f ms per MSM" % ((time.perf_counter() - t0) * 1e3 / steps))
