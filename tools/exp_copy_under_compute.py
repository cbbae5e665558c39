"""Do host-to-device copies and the MSM kernels overlap at all on this runtime?  A thread copies 32 MB from PINNED host memory to the device in a loop
on its own stream (torch: hipMemcpyAsync + stream synchronise) and reports its rate -- alone, and while four bound-bases MSMs from DEVICE scalars are
in flight (no host traffic of their own); the MSM rate is reported alone and beside the copies.  python tools/exp_copy_under_compute.py"""
import importlib, sys, threading, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
pin = torch.frombuffer(bytearray(sc), dtype=torch.uint8).pin_memory()
pag = torch.frombuffer(bytearray(sc), dtype=torch.uint8)
dst = torch.empty_like(ds)
torch.cuda.synchronize()
stop = False
rates = []


def copier(src):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        while not stop:
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            st.synchronize()
            rates.append((time.perf_counter() - t0) * 1e3)


def msms(c, b, secs):
    tk, done, t0 = [], 0, time.perf_counter()
    while time.perf_counter() - t0 < secs:
        tk.append(c.submit_scalars_device(b, ds.data_ptr()))
        if len(tk) >= 4:
            c.collect(tk.pop(0)); done += 1
    while tk:
        c.collect(tk.pop(0)); done += 1
    return (time.perf_counter() - t0) * 1e3 / done


with pkg.MsmContext((0,)) as c:
    b = c.bind_points(pts)
    msms(c, b, 0.3)
    print("MSMs alone (device scalars, 4 in flight): %.4f ms per MSM" % msms(c, b, 1.0), flush=True)
    for label, src in (("pinned", pin), ("pageable", pag)):
        stop = False; rates.clear()
        th = threading.Thread(target=copier, args=(src,)); th.start()
        time.sleep(0.5)
        alone = sorted(rates[len(rates) // 2:])
        rates.clear()
        per = msms(c, b, 1.5)
        beside = sorted(rates[2:])
        stop = True; th.join()
        print("%-8s 32 MB copies: alone %.3f ms median (%.1f GB/s); beside the MSMs %.3f ms median (%.1f GB/s), %d copies; MSMs beside the copies: %.4f ms per MSM" % (
            label, alone[len(alone) // 2], 33.554432 / alone[len(alone) // 2], beside[len(beside) // 2], 33.554432 / beside[len(beside) // 2], len(beside), per), flush=True)
