"""Where the time of te_msm_run (host buffers, what compute_msm(Buffer, Buffer) delivers) goes at n = 2^20:
PCIe alone (pageable and pinned sources), page-locking the caller's buffers, and the whole call for 1..8 pieces.
Run on the GPU box:  python tools/host_path.py"""
import ctypes, importlib, sys, time
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n)
nbytes = len(pts) + len(sc)
hp = torch.frombuffer(bytearray(pts), dtype=torch.uint8); hs = torch.frombuffer(bytearray(sc), dtype=torch.uint8)
dp = torch.empty(len(pts), dtype=torch.uint8, device="cuda"); ds = torch.empty(len(sc), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def best(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3

t = best(lambda: (dp.copy_(hp), ds.copy_(hs)))
print("pageable H2D of %d MB (points + scalars): %.3f ms = %.1f GB/s" % (nbytes >> 20, t, nbytes / t / 1e6))
pp, ps = hp.pin_memory(), hs.pin_memory()
t = best(lambda: (dp.copy_(pp, non_blocking=True), ds.copy_(ps, non_blocking=True)))
print("pinned   H2D of %d MB: %.3f ms = %.1f GB/s" % (nbytes >> 20, t, nbytes / t / 1e6))
rt = torch.cuda.cudart()
t0 = time.perf_counter(); r1 = rt.cudaHostRegister(hp.data_ptr(), hp.numel(), 0); r2 = rt.cudaHostRegister(hs.data_ptr(), hs.numel(), 0); t1 = time.perf_counter()
print("hipHostRegister of the caller's two buffers: %.3f ms (rc %s %s)" % ((t1 - t0) * 1e3, r1, r2))
if int(r1) == 0: rt.cudaHostUnregister(hp.data_ptr())
if int(r2) == 0: rt.cudaHostUnregister(hs.data_ptr())
with pkg.MsmContext((0,)) as ctx:
    ctx.set_option("window_bits", 16)
    ref = ctx.run(pts, sc)
    import os
    for k, split in ((1, None), (2, None), (3, None), (4, None), (5, None), (6, None), (8, None), (3, "40,35,25"), (3, "45,33,22"), (4, "32,27,23,18"), (4, "35,28,22,15"), (5, "28,24,20,16,12")):
        ctx.set_option("host_chunks", k)
        if split:
            os.environ["TE_MSM_HOST_SPLIT"] = split
        else:
            os.environ.pop("TE_MSM_HOST_SPLIT", None)
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); r = ctx.run(pts, sc); ts.append((time.perf_counter() - t0) * 1e3)
        assert r == ref
        print("te_msm_run, %d piece(s)%s: best %.3f ms, median %.3f ms  (%.1f GB/s of input)" % (k, (" split " + split) if split else "", min(ts), sorted(ts)[len(ts) // 2], nbytes / min(ts) / 1e6))
    os.environ.pop("TE_MSM_HOST_SPLIT", None)
    dpts = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); dsc = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); r = ctx.run_device(dpts.data_ptr(), dsc.data_ptr(), n); ts.append((time.perf_counter() - t0) * 1e3)
    print("te_msm_run_device (inputs resident): best %.3f ms" % min(ts))
