import importlib, sys, torch
sys.path.insert(0,'/root/repo')
pkg=importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle import oracle
n=1<<20
pts=oracle.gen_points(1,n); sc=oracle.gen_scalars(1,n)
dp=torch.frombuffer(bytearray(pts),dtype=torch.uint8).cuda(); ds=torch.frombuffer(bytearray(sc),dtype=torch.uint8).cuda()
ctx=pkg.MsmContext((0,)); ctx.set_option("window_bits",16); ctx.set_option("profile",2)
for cut in (0,):
    ctx.set_option("debug_cut",cut)
    for _ in range(3):
        try: ctx.run_device(dp.data_ptr(),ds.data_ptr(),n)
        except Exception as e: pass
    print("cut",cut, {k:round(v*1000) for k,v in ctx.stage_ms().items() if True})
