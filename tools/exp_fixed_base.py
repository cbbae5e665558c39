"""Fixed-base windows (option "bind_fixed_base" = c; round-5 verdict item 7) against the ordinary windows over the same bound set:
bind time and table size, the stages of one MSM alone (HIP events at every stage boundary), k_accumulate's own clock, latency, and
MSMs in flight from device and from host scalars.  python tools/exp_fixed_base.py [log2n ...] > profiles/r06_fixed_base_windows_raw.txt"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
sizes = [int(a) for a in sys.argv[1:]] or [20, 18, 16]
for lg in sizes:
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="random")
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    want = None
    for c in (0, 16, 17, 18, 19, 20, 21, 0, 20):
        with pkg.MsmContext((0,)) as cx:
            cx.set_option("bind_fixed_base", c)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            b = cx.bind_points(pts)
            bind_ms = (time.perf_counter() - t0) * 1e3
            r = cx.run_scalars_device(b, ds.data_ptr())
            want = want or r
            assert r == want, (lg, c)
            for _ in range(3):
                cx.run_scalars_device(b, ds.data_ptr())
            lat = []
            for _ in range(8):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                cx.run_scalars_device(b, ds.data_ptr())
                lat.append((time.perf_counter() - t0) * 1e3)
            cx.set_option("profile", 2)
            acc = {}
            for _ in range(6):
                cx.run_scalars_device(b, ds.data_ptr())
                for k, v in cx.stage_ms().items():
                    acc[k] = acc.get(k, 0.0) + v / 6
            entries = cx.get_option("entries_accumulated")
            cx.set_option("profile", 0)

            def in_flight(submit, depth, steps=96):
                for t in [submit() for _ in range(depth)]:
                    assert cx.collect(t) == want
                best = 1e9
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter(); tk = []
                    for _ in range(steps):
                        tk.append(submit())
                        if len(tk) >= depth:
                            cx.collect(tk.pop(0))
                    while tk:
                        cx.collect(tk.pop(0))
                    best = min(best, (time.perf_counter() - t0) * 1e3 / steps)
                return best
            dev4 = in_flight(lambda: cx.submit_scalars_device(b, ds.data_ptr()), 4)
            host8 = in_flight(lambda: cx.submit_scalars(b, sc), 8)
            hl = []
            for _ in range(5):
                t0 = time.perf_counter(); cx.run_scalars(b, sc); hl.append((time.perf_counter() - t0) * 1e3)
            st = " ".join("%s %.1f" % (k.replace("accumulate_core_clock_ghz", "clk"), v * (1 if k.endswith("ghz") else 1e3)) for k, v in acc.items() if k != "accumulate_on_device")
            print("n=2^%d c=%-2s table %6.0f MB bind %7.1f ms | entries %9d | latency dev %.4f host %.4f ms | in flight: dev x4 %.4f ms = %6.1f MSM/s, host x8 %.4f ms | fallbacks %d | stages us: %s"
                  % (lg, c or "-", cx.get_option("bases_bytes") / 2**20, bind_ms, entries, min(lat), min(hl), dev4, 1e3 / dev4, host8, cx.get_option("fixed_base_fallbacks"), st), flush=True)
