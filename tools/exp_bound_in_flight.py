"""Bound bases, host scalars, tickets in flight at n = 2^20: where do the ~0.13 ms per MSM between te_msm_submit_scalars (host, 1.02-1.09 ms)
and te_msm_submit_scalars_device (0.89-0.92 ms) go?  Neither the link (32 MB = 0.62 ms) nor the device is saturated.  Sweeps the upload lanes,
the pieces, the engine's own pinned staging, the number in flight, and pinned caller memory.  python tools/exp_bound_in_flight.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
torch.cuda.synchronize()


def in_flight(c, submit, depth, steps=64):
    for t in [submit() for _ in range(depth)]:
        c.collect(t)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); tk = []
        for _ in range(steps):
            tk.append(submit())
            if len(tk) >= depth:
                c.collect(tk.pop(0))
        while tk:
            c.collect(tk.pop(0))
        best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
    return best


with pkg.MsmContext((0,)) as c:
    b = c.bind_points(pts)
    want = c.run_scalars(b, sc)
    print("device scalars, 4 in flight: %.4f ms per MSM" % in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4), flush=True)
    for lanes in (1, 2, 4, 8):
        for chunks in (1, 2, 3):
            for staging in (0, 1):
                c.set_option("upload_threads", lanes); c.set_option("scalar_chunks", chunks); c.set_option("host_staging", staging)
                r = {d: in_flight(c, lambda: c.submit_scalars(b, sc), d) for d in (4, 8)}
                print("host scalars: upload_threads %d  scalar_chunks %d  host_staging %d :  4 in flight %.4f   8 in flight %.4f ms per MSM" % (lanes, chunks, staging, r[4], r[8]), flush=True)
    c.set_option("upload_threads", 4); c.set_option("scalar_chunks", 0); c.set_option("host_staging", 0)
    # the caller's scalars in pinned memory: truly asynchronous copies
    pin = torch.frombuffer(bytearray(sc), dtype=torch.uint8).pin_memory()
    import ctypes
    L = c._L
    def submit_pinned():
        t = ctypes.c_uint64()
        rc = L.te_msm_submit_scalars(c._h, b._h, ctypes.cast(pin.data_ptr(), ctypes.c_char_p), ctypes.byref(t))
        assert rc == 0
        return t.value
    for chunks in (0, 1, 3):
        c.set_option("scalar_chunks", chunks)
        print("host scalars in PINNED memory, scalar_chunks %d: 4 in flight %.4f   8 in flight %.4f ms per MSM" % (chunks, in_flight(c, submit_pinned, 4), in_flight(c, submit_pinned, 8)), flush=True)
    assert c.collect(submit_pinned()) == want
