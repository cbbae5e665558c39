"""Host-side cost of one MSM submission (enqueue) and collection, eager launches vs HIP graphs."""
import importlib, sys, time, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle import oracle
for log2n in (12, 16, 20):
    n = 1 << log2n
    pts, sc = oracle.gen_points(1, n), oracle.gen_scalars(1, n)
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    for graph in (0, 1):
        with pkg.MsmContext((0,)) as ctx:
            ctx.set_option("graph", graph)
            for _ in range(3):
                ctx.run_device(dp.data_ptr(), ds.data_ptr(), n)
            torch.cuda.synchronize()
            sub = col = 0.0; K = 50
            t_all = time.perf_counter()
            for _ in range(K):
                t0 = time.perf_counter(); t = ctx.submit_device(dp.data_ptr(), ds.data_ptr(), n); t1 = time.perf_counter()
                ctx.collect(t); t2 = time.perf_counter()
                sub += t1 - t0; col += t2 - t1
            t_all = time.perf_counter() - t_all
            print(f"n=2^{log2n} graph={graph}: submit {sub/K*1e6:7.1f} us  collect(wait+tail) {col/K*1e6:7.1f} us  total {t_all/K*1e6:7.1f} us", flush=True)
