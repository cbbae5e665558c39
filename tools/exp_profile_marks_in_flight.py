"""Do the two HIP events around k_accumulate (option "profile" = 1: what bench.py's roofline.kernel_ms is measured with, inside the timed region as
the contract asks) cost throughput with four MSMs in flight?  n = 2^20, resident inputs, te_msm_submit_device, profile 0 / 1 / 2 alternating, four rounds.
python tools/exp_profile_marks_in_flight.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
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
    want = c.run(pts, sc)
    assert c.collect(c.submit_device(dp.data_ptr(), ds.data_ptr(), n)) == want
    for rnd in range(4):
        row = []
        for prof in (0, 1, 2):
            c.set_option("profile", prof)
            row.append("profile %d: %s" % (prof, in_flight(c, lambda: c.submit_device(dp.data_ptr(), ds.data_ptr(), n), 4)))
        print("round %d  %s" % (rnd, "   ".join(row)), flush=True)
