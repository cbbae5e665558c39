"""Bound bases, host scalars at n = 2^20 (and host-buffer tickets): upload lanes 1 / 2 / 3 / 4 x tickets in flight 2 / 4 / 6 / 8, two rounds, after the
marker of the copy stream went away (tools/exp_bound_copy_queue.py).  python tools/exp_bound_lanes_depth.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")


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
    return min(ps)


with pkg.MsmContext((0,)) as c:
    b = c.bind_points(pts)
    for rnd in range(2):
        for lanes in (1, 2, 3, 4):
            c.set_option("upload_threads", lanes)
            print("round %d lanes %d: host scalars in flight  %s   host buffers  %s" % (
                rnd, lanes, "  ".join("x%d %.4f" % (dp, in_flight(c, lambda: c.submit_scalars(b, sc), dp)) for dp in (2, 4, 6, 8)),
                "  ".join("x%d %.4f" % (dp, in_flight(c, lambda: c.submit_async(pts, sc), dp, 32)) for dp in (4, 8))), flush=True)
