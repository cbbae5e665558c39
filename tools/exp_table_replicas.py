"""Fixed-base windows (round-5 verdict item 7), the question that decides it BEFORE anything is built: per-window tables
2^(c w) P_i make all windows share one bucket set (c = 20: 13 n + 2^20 additions instead of 16 n + 2^20 + ..: about -17 %), but
the accumulation then gathers from W x 128 MB of records instead of 128 MB that live in the 256 MB Infinity Cache.  What does the
gather cost when its footprint leaves the cache?  Option "exp_table_replicas" = R keeps R copies of the bound records and lets the
windows of an MSM gather from different copies (same records, same result): the arithmetic is unchanged, only the footprint grows
to R x 128 MB.  Measured: k_accumulate alone (HIP events, its own core clock) and MSM/s with four device-scalar tickets in flight.
python tools/exp_table_replicas.py > profiles/r06_fixed_base_gather_raw.txt"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
for curve, label in ((pkg.CURVE_TE_BLS12, "TE-BLS12, 128-B records"),):
    want = None
    for rep in (1, 2, 4, 8, 1, 8):
        with pkg.MsmContext((0,)) as c:
            c.set_option("window_bits", 16)
            c.set_option("exp_table_replicas", rep)
            b = c.bind_points(pts)
            r = c.run_scalars_device(b, ds.data_ptr())
            want = want or r
            assert r == want
            c.set_option("profile", 1)
            acc, clk = [], []
            for _ in range(12):
                assert c.run_scalars_device(b, ds.data_ptr()) == want
                st = c.stage_ms(); acc.append(st["accumulate"]); clk.append(st["accumulate_core_clock_ghz"])
            c.set_option("profile", 0)
            for t in [c.submit_scalars_device(b, ds.data_ptr()) for _ in range(4)]:
                c.collect(t)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); tk = []
                for _ in range(100):
                    tk.append(c.submit_scalars_device(b, ds.data_ptr()))
                    if len(tk) >= 4:
                        c.collect(tk.pop(0))
                while tk:
                    c.collect(tk.pop(0))
                best = min(best, (time.perf_counter() - t0) * 1e3 / 100)
            print("%s  copies of the records %d (gather footprint %4d MB): k_accumulate alone %.4f ms (min %.4f) at %.3f GHz = %.3f G cycles;  4 tickets in flight %.4f ms per MSM = %.1f MSM/s"
                  % (label, rep, rep * 128, sum(acc) / len(acc), min(acc), sum(clk) / len(clk), sum(a * k for a, k in zip(acc, clk)) / len(acc) * 1e-3 * 1e3, best, 1e3 / best), flush=True)
