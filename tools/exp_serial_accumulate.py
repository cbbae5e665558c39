"""One k_accumulate at a time?  With four MSMs in flight two accumulations usually share the GPU at half speed each (pipelined duration 1.4 ms against
0.72 alone).  TE_MSM_SERIAL_ACCUMULATE=1 chains the accumulate launches of a device (each waits for the one enqueued before it on another stream).
Child processes alternating, three rounds; n = 2^20, resident inputs (te_msm_submit_device) 4 and 8 in flight, bound bases from device and host scalars.
python tools/exp_serial_accumulate.py"""
import importlib, os, subprocess, sys, time


def child():
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
        return "%.4f" % min(ps)

    with pkg.MsmContext((0,)) as c:
        want = c.run(pts, sc)
        b = c.bind_points(pts)
        assert c.collect(c.submit_device(dp.data_ptr(), ds.data_ptr(), n)) == want and c.collect(c.submit_scalars(b, sc)) == want
        f = lambda: c.submit_device(dp.data_ptr(), ds.data_ptr(), n)
        print("resident x2 %s  x3 %s  x4 %s  x6 %s  x8 %s   bound, device scalars x4 %s   bound, host scalars x8 %s" % (
            in_flight(c, f, 2), in_flight(c, f, 3), in_flight(c, f, 4), in_flight(c, f, 6), in_flight(c, f, 8),
            in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4), in_flight(c, lambda: c.submit_scalars(b, sc), 8)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rnd in range(3):
            for name, env in (("free", {}), ("chained", {"TE_MSM_SERIAL_ACCUMULATE": "1"})):
                r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
                print("round %d %-8s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
