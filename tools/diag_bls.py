"""diagnosis: BLS12-377 at n = 2^20 -- which input / call form disagrees with the oracle?"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
from oracle import oracle377 as o
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
sp, ss = pkg.synth_inputs(0x5EED0000 + lg, n, curve=pkg.CURVE_BLS12_377_G1)
op, os_ = o.gen_points(0x5EED0014, n), o.gen_scalars(0x5EED0014, n)
def dev(b):
    return torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
for name, (p, s) in {"synth/synth": (sp, ss), "oracle/oracle": (op, os_)}.items():
    exp = o.msm(p, s, c=16, threads=16)
    exp4 = o.msm(p, s, c=13, threads=16)
    dp, ds = dev(p), dev(s)
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as cx:
        cx.set_option("curve", pkg.CURVE_BLS12_377_G1)
        cx.set_option("window_bits", 16)
        if os.environ.get("DIAG_PREZERO") == "0":
            cx.set_option("prezero", 0)
        r_host = cx.run(p, s)
        r_dev = cx.run_device(dp.data_ptr(), ds.data_ptr(), n)
        r_pipe = []
        for _ in range(3):
            ts = [cx.submit_device(dp.data_ptr(), ds.data_ptr(), n) for _ in range(4)]
            r_pipe += [cx.collect(t) for t in ts]
        cx.set_option("window_bits", 13)
        r13 = cx.run_device(dp.data_ptr(), ds.data_ptr(), n)
    print(name, "oracle c16==c13:", exp == exp4, "| host:", r_host == exp, "device:", r_dev == exp, "pipelined:", [r == exp for r in r_pipe], "c13:", r13 == exp, flush=True)
