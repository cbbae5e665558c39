"""Is the ONE submitting thread what bounds small MSMs in flight?  te_msm_submit_device enqueues ~10 launches per MSM on the calling
thread (~100 us); option "enqueue_async" = 1 hands that to the device's host threads ("upload_threads").  Pipelined throughput from
device-resident inputs, n = 2^16 .. 2^20, both settings.   python tools/exp_enqueue_async.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch


def run(c, dp, ds, n, steps, depth):
    tk, res = [], None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tk.append(c.submit_device(dp, ds, n))
        if len(tk) >= depth:
            res = c.collect(tk.pop(0))
    while tk:
        res = c.collect(tk.pop(0))
    return (time.perf_counter() - t0) * 1e3 / steps, res


for lg in (16, 17, 18, 19, 20):
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n, fixed_point="chain" if lg < 20 else "random")
    a = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); b = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    steps = 400 if lg <= 17 else 200 if lg <= 19 else 100
    line = []
    for mode, threads, depth in ((0, 4, 4), (0, 4, 8), (1, 2, 4), (1, 4, 4), (1, 4, 8), (1, 8, 8), (0, 4, 4)):
        with pkg.MsmContext((0,)) as c:
            c.set_option("enqueue_async", mode)      # (the experimental build only: the option was removed again)
            c.set_option("upload_threads", threads)
            ref = c.run_device(a.data_ptr(), b.data_ptr(), n)
            run(c, a.data_ptr(), b.data_ptr(), n, 2 * depth, depth)
            best = 1e9
            for _ in range(3):
                ms, res = run(c, a.data_ptr(), b.data_ptr(), n, steps, depth)
                assert res == ref
                best = min(best, ms)
            line.append("%s/%dthr/%dfl %.4f" % ("async" if mode else "inline", threads, depth, best))
    print("n=2^%d ms per MSM: %s" % (lg, "  ".join(line)), flush=True)
