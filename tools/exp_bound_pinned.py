"""Bound bases, 8 tickets in flight at n = 2^20: pageable host scalars against scalars in PINNED host memory (torch pin_memory) and device scalars,
three rounds alternating (final build: lane threads wait for their uploads).  python tools/exp_bound_pinned.py"""
import ctypes, importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
pin = torch.frombuffer(bytearray(sc), dtype=torch.uint8).pin_memory()
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
    want = c.run_scalars(b, sc)
    L = c._L

    def submit_pinned():
        t = ctypes.c_uint64()
        assert L.te_msm_submit_scalars(c._h, b._h, ctypes.cast(pin.data_ptr(), ctypes.c_char_p), ctypes.byref(t)) == 0
        return t.value
    assert c.collect(submit_pinned()) == want
    for rnd in range(3):
        print("round %d  device scalars x4 %s | pageable host scalars x8 %s | pinned host scalars x8 %s | pinned x4 %s" % (
            rnd, in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4), in_flight(c, lambda: c.submit_scalars(b, sc), 8),
            in_flight(c, submit_pinned, 8), in_flight(c, submit_pinned, 4)), flush=True)
    t = []
    for _ in range(9):
        t0 = time.perf_counter()
        out = ctypes.create_string_buffer(96)
        assert L.te_msm_run_scalars(c._h, b._h, ctypes.cast(pin.data_ptr(), ctypes.c_char_p), out) == 0
        t.append((time.perf_counter() - t0) * 1e3)
    assert out.raw[:64] == want
    print("lone call from pinned scalars: %.4f ms (median %.4f)" % (min(t), sorted(t)[4]))
