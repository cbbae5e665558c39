"""A/B of option "lane_host_waits" (csrc/te_msm.hip, lane_wait): asynchronous tickets whose lane thread WAITS for each upload before it
enqueues the kernels behind it, against a stream wait in front of those kernels.  Tickets in flight from pageable host memory at n = 2^20:
bound bases (32 MB of scalars per MSM) and the ordinary host-buffer tickets (96 MB), alternating, next to device-scalar tickets.
python tools/exp_lane_host_waits.py > profiles/r06_lane_host_waits_raw.txt"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
for lg in (20, 18, 16):
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="random")
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    def in_flight(c, submit, depth, steps=96):
        for t in [submit() for _ in range(depth)]:
            assert c.collect(t) == want
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
        return min(ps), sorted(ps)[1]

    with pkg.MsmContext((0,)) as c:
        b = c.bind_points(pts)
        want = c.run_scalars(b, sc)
        print("n=2^%d  device scalars, bound bases, 4 in flight: %.4f ms per MSM (median %.4f)" % ((lg,) + in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4)), flush=True)
        for rnd in range(3):
            for hw in (0, 1):
                c.set_option("lane_host_waits", hw)
                a = in_flight(c, lambda: c.submit_scalars(b, sc), 8)
                h = in_flight(c, lambda: c.submit_async(pts, sc), 8, 48)
                print("n=2^%d  round %d lane_host_waits %d:  bound bases, host scalars, 8 in flight %.4f (median %.4f)   host buffers (submit_async), 8 in flight %.4f (median %.4f) ms per MSM"
                      % (lg, rnd, hw, a[0], a[1], h[0], h[1]), flush=True)
