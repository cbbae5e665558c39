"""te_msm_run_scalars: pieces the scalars of a bound set are uploaded and processed in (option "scalar_chunks"), latency of the lone call
from pageable host scalars; and te_msm_submit_scalars tickets in flight with the pieces forced.  python tools/sweep_scalar_chunks.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
for lg in (20, 19, 18, 17, 16):
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="chain")
    with pkg.MsmContext((0,)) as c:
        b = c.bind_points(pts)
        want = c.run_scalars(b, sc)
        out = []
        for k in (1, 2, 3, 4, 5, 6, 8):
            c.set_option("scalar_chunks", k)
            for _ in range(3):
                assert c.run_scalars(b, sc) == want
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); c.run_scalars(b, sc); ts.append((time.perf_counter() - t0) * 1e3)
            for t in [c.submit_scalars(b, sc) for _ in range(8)]:
                c.collect(t)
            t0 = time.perf_counter(); tk = []
            for _ in range(48):
                tk.append(c.submit_scalars(b, sc))
                if len(tk) >= 8:
                    c.collect(tk.pop(0))
            while tk:
                c.collect(tk.pop(0))
            fl = (time.perf_counter() - t0) * 1e3 / 48
            out.append("K=%d %.3f (median %.3f) / %.3f" % (k, min(ts), sorted(ts)[4], fl))
        print("n=2^%d  lone call ms (best, median) / per MSM with 8 tickets in flight:  %s" % (lg, "   ".join(out)), flush=True)
