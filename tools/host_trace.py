import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n)
with pkg.MsmContext((0,)) as ctx:
    ctx.set_option("window_bits", 16)
    for k in (3, 4):
        ctx.set_option("host_chunks", k)
        for _ in range(3): ctx.run(pts, sc)
        sys.stderr.write("---- K=%d\n" % k); sys.stderr.flush()
        os.environ["TE_MSM_TRACE_HOST"] = "1"
        ctx.run(pts, sc)
        del os.environ["TE_MSM_TRACE_HOST"]
