"""Why do the driver-shaped lines read 1.04-1.07 ms per MSM for bound-bases tickets from host scalars where the dedicated A/B reads 0.92-0.96?
The same measurement (8 tickets in flight, 96-step passes, n = 2^20) in the states bench.py goes through: a fresh context; after the host-buffer
tickets (te_msm_submit_async, 96 B per point: every work set gets its staging buffers); after lone run_scalars calls; with 4 instead of 8 in flight.
python tools/exp_bound_in_flight_order.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
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
    c.set_option("window_bits", 16)
    b = c.bind_points(pts)
    print("fresh context: bound, host scalars, 8 in flight      ", in_flight(c, lambda: c.submit_scalars(b, sc), 8), flush=True)
    print("               bound, device scalars, 4 in flight    ", in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4), flush=True)
    print("               per-call, device inputs, 4 in flight  ", in_flight(c, lambda: c.submit_device(dp.data_ptr(), ds.data_ptr(), n), 4), flush=True)
    print("               bound, host scalars, 8 in flight      ", in_flight(c, lambda: c.submit_scalars(b, sc), 8), flush=True)
    print("               host buffers (submit_async), 8        ", in_flight(c, lambda: c.submit_async(pts, sc), 8, 32), flush=True)
    print("after them:    bound, host scalars, 8 in flight      ", in_flight(c, lambda: c.submit_scalars(b, sc), 8), flush=True)
    for _ in range(8):
        c.run_scalars(b, sc)
    print("after lone calls: bound, host scalars, 8 in flight   ", in_flight(c, lambda: c.submit_scalars(b, sc), 8), flush=True)
    print("                  bound, host scalars, 4 in flight   ", in_flight(c, lambda: c.submit_scalars(b, sc), 4), flush=True)
    print("                  bound, host scalars, 6 in flight   ", in_flight(c, lambda: c.submit_scalars(b, sc), 6), flush=True)
    c.trim(0)
    print("after trim(0):    bound, host scalars, 8 in flight   ", in_flight(c, lambda: c.submit_scalars(b, sc), 8), flush=True)
