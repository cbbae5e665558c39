"""Bounding experiment for the sort's bytes (round-4 verdict, item 8): what do the digit rows cost?  k_digits writes them (32 MB at
n = 2^20) and the level-1 scatter reads them back once (32 MB) -- 64 of the ~490 MB the sort moves.  A build of k_digits WITHOUT the
digit stores (its results are wrong: timing only) bounds what any scheme that keeps the digits out of memory can save on the write
side; the read side is bounded by the scatter's share of bytes.   TE_MSM_LIB=<lib> python tools/exp_sort_bytes.py"""
import importlib, os, sys
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
for lg in (20, 18, 16):
    n = 1 << lg
    pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        c.set_option("profile", 2)
        try:
            c.run_device(dp.data_ptr(), ds.data_ptr(), n)
        except pkg.MsmError:
            pass
        acc = {}
        reps = 12
        for _ in range(reps):
            try:
                c.run_device(dp.data_ptr(), ds.data_ptr(), n)
            except pkg.MsmError:
                pass
            for k, v in c.stage_ms().items():
                acc[k] = acc.get(k, 0.0) + v * 1e3 / reps
        print("%s n=2^%d: digits %.1f us, part_scatter %.1f us, bucket_sort %.1f us, prep_points %.1f us" % (
            os.path.basename(pkg.library_path()), lg, acc.get("digits", 0), acc.get("part_scatter", 0), acc.get("bucket_sort", 0), acc.get("prep_points", 0)))
