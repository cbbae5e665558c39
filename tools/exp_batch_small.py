"""What would whole-MSM BATCH tickets buy at the small harness sizes?  te_msm_partial_device_batch (the window-shard building
block) already runs the windows of up to eight MSMs of one size through ONE launch sequence; with the shard 0 / 1 that is eight
whole MSMs.  Device time per MSM, batches back to back on two work sets, against tickets (te_msm_submit_device, 4 in flight).
python tools/exp_batch_small.py"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
for lg in (14, 16, 17, 18, 19):
    n = 1 << lg
    sets = [pkg.synth_inputs(0x5EED0000 + lg + 100 * m, n, fixed_point="chain") for m in range(8)]
    dev = [(torch.frombuffer(bytearray(p), dtype=torch.uint8).cuda(), torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()) for p, s in sets]
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        cb, W = c.plan(n)
        refs = [c.run_device(a.data_ptr(), b.data_ptr(), n) for a, b in dev]
        # tickets
        def tickets(steps=160, depth=4):
            tk = []
            t0 = time.perf_counter()
            for i in range(steps):
                a, b = dev[i % 8]
                tk.append(c.submit_device(a.data_ptr(), b.data_ptr(), n))
                if len(tk) >= depth:
                    c.collect(tk.pop(0))
            while tk:
                c.collect(tk.pop(0))
            return (time.perf_counter() - t0) * 1e3 / steps
        tickets(16); t_tk = min(tickets() for _ in range(3))
        line = ["tickets/4fl %.4f" % t_tk]
        for count in (2, 4, 8):
            parts = [torch.zeros(count * W * c.row_bytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
            host = [torch.zeros(count * W * c.row_bytes, dtype=torch.uint8).pin_memory() for _ in range(2)]
            def batches(nb=24):
                t0 = time.perf_counter()
                for i in range(nb):
                    k = i % 2
                    c.set_option("workset", k)
                    if i >= 2:
                        c.partial_wait(k)
                        res = [pkg.finalize_host(host[k][m * W * c.row_bytes:(m + 1) * W * c.row_bytes].numpy().tobytes(), cb, W) for m in range(count)]
                    c.partial_device_batch([dev[m][0].data_ptr() for m in range(count)], [dev[m][1].data_ptr() for m in range(count)], n, parts[k].data_ptr())
                    st, _ = c.workset_stream(k)
                    with torch.cuda.stream(torch.cuda.ExternalStream(st)):
                        host[k].copy_(parts[k], non_blocking=True)
                for k in range(2):
                    c.set_option("workset", k); c.partial_wait(k)
                torch.cuda.synchronize()
                res = [pkg.finalize_host(host[1][m * W * c.row_bytes:(m + 1) * W * c.row_bytes].numpy().tobytes(), cb, W) for m in range(count)]
                assert res == refs[:count], "batch result"
                return (time.perf_counter() - t0) * 1e3 / (nb * count)
            batches(4)
            line.append("batch of %d %.4f" % (count, min(batches() for _ in range(3))))
            c.set_option("workset", 0)
        print("n=2^%d (c=%d, W=%d) ms per MSM: %s" % (lg, cb, W, "  ".join(line)), flush=True)
