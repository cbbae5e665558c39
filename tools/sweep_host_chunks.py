"""te_msm_run from host buffers: pieces against n (which `host_chunks` the automatic rule should pick).  GPU box: python tools/sweep_host_chunks.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
with pkg.MsmContext((0,)) as ctx:
    for lg in (16, 17, 18, 19, 20):
        n = 1 << lg
        pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)
        ctx.set_option("host_chunks", 1)
        ref = ctx.run(pts, sc)
        out = []
        for k in (0, 1, 2, 3, 4):
            ctx.set_option("host_chunks", k)
            ts = []
            for _ in range(8):
                t0 = time.perf_counter(); r = ctx.run(pts, sc); ts.append((time.perf_counter() - t0) * 1e3)
            assert r == ref
            out.append("K=%s %.3f" % ("auto" if k == 0 else k, min(ts)))
        print("n=2^%d  " % lg + "   ".join(out), flush=True)
