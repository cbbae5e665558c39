"""Experiment (not part of the product): how much latency would one MSM gain if its windows ran as G groups on G streams?

Emulation with what exists: G contexts, context g owns windows {w : w mod G = g} (te_msm_set_window_shard), all G launch
sequences are enqueued back to back on their own streams, the rows are merged and folded once.  Every context converts
the points and reads the scalars itself, so the emulation does G-1 record conversions and digit passes too many -- an upper
bound on the time of the real thing (one conversion, one digit pass, shared records).

    python tools/exp_groups.py [log2n] [G ...]
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch  # noqa: E402


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as solo:
        ref = solo.run_device(dp.data_ptr(), ds.data_ptr(), n)
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            solo.run_device(dp.data_ptr(), ds.data_ptr(), n)
            ts.append((time.perf_counter() - t0) * 1e3)
        print("n=2^%d  run_device (one sequence): %.3f ms (min of 8), median %.3f" % (lg, min(ts), sorted(ts)[4]))
        c, W = solo.plan(n)
    for G in groups:
        ctxs = [pkg.MsmContext((0,)) for _ in range(G)]
        parts = [torch.zeros(W * pkg.PARTIAL_BYTES, dtype=torch.uint8, device="cuda") for _ in range(G)]
        host = [torch.zeros(W * pkg.PARTIAL_BYTES, dtype=torch.uint8).pin_memory() for _ in range(G)]
        for g, cx in enumerate(ctxs):
            cx.set_window_shard(g, G)
        streams = [torch.cuda.Stream() for _ in range(G)]

        def once():
            for g, cx in enumerate(ctxs):
                cx.partial_device(dp.data_ptr(), ds.data_ptr(), n, parts[g].data_ptr(), streams[g].cuda_stream)
                with torch.cuda.stream(streams[g]):
                    host[g].copy_(parts[g], non_blocking=True)
            for s in streams:
                s.synchronize()
            rows = pkg.merge_partials([h.numpy().tobytes() for h in host], W, G)
            return pkg.finalize_host(rows, c, W)

        assert once() == ref, "merged result differs"
        ts = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            once()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("G=%d window groups on %d streams (emulated with %d contexts): %.3f ms (min of 8), median %.3f" % (G, G, G, min(ts), sorted(ts)[4]))
        for cx in ctxs:
            cx.close()


if __name__ == "__main__":
    main()
