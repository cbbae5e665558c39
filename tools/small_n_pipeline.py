"""Pipelined MSMs at small n: is the rate set by the device or by the host's enqueue cost?  (eager launches vs HIP-graph replay,
4 or 8 MSMs in flight; host time spent inside te_msm_submit_device per MSM)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
for log2n in (14, 16, 18):
    n = 1 << log2n
    pts, sc = pkg.synth_inputs(0x5EED0000 + log2n, n, fixed_point="chain")
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    for graph in (0, 1):
        for depth in (4, 8):
            with pkg.MsmContext((0,)) as ctx:
                ctx.set_option("graph", graph)
                for t in [ctx.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(depth)]:
                    ctx.collect(t)
                for t in [ctx.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(depth)]:
                    ctx.collect(t)
                steps, sub, tickets = 400, 0.0, []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    a = time.perf_counter()
                    tickets.append(ctx.submit_device(dp.data_ptr(), ds.data_ptr(), n))
                    sub += time.perf_counter() - a
                    if len(tickets) >= depth:
                        ctx.collect(tickets.pop(0))
                while tickets:
                    ctx.collect(tickets.pop(0))
                el = time.perf_counter() - t0
                print("n=2^%d graph=%d in flight %d: %.1f us per MSM (%.0f MSM/s), of which the host spends %.1f us inside submit" % (
                    log2n, graph, depth, el / steps * 1e6, steps / el, sub / steps * 1e6), flush=True)
