"""Bound bases, host scalars, 8 tickets in flight at n = 2^20 and 2^18: upload lanes 1..6 (option "upload_threads") with the lane thread waiting for
its upload (lane_host_waits = 1), three rounds alternating.  python tools/exp_bound_lanes.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
for lg in (20, 18):
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="random")
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    def in_flight(c, submit, depth, steps=96):
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
        return "%.4f (%s)" % (min(ps), " ".join("%.3f" % x for x in ps))

    with pkg.MsmContext((0,)) as c:
        b = c.bind_points(pts)
        print("n=2^%d device scalars, 4 in flight: %s" % (lg, in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4)), flush=True)
        for rnd in range(3):
            for lanes in (1, 2, 3, 4, 6):
                c.set_option("upload_threads", lanes)
                print("n=2^%d round %d upload_threads %d: host scalars, 8 in flight %s   host buffers (submit_async) 8 in flight %s" % (
                    lg, rnd, lanes, in_flight(c, lambda: c.submit_scalars(b, sc), 8), in_flight(c, lambda: c.submit_async(pts, sc), 8, 32)), flush=True)
