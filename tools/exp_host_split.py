"""te_msm_run from pageable host buffers at n = 2^20: piece weights of the upload (TE_MSM_HOST_SPLIT, option "host_chunks") -- equal thirds (the build)
against falling sizes and four / five pieces; medians of 30 calls, child processes, two rounds.  The link alone needs 1.80 ms (100.7 MB at 56 GB/s).
python tools/exp_host_split.py"""
import importlib, os, statistics, subprocess, sys, time

CASES = [("3 equal", 3, ""), ("3: 40/35/25", 3, "40,35,25"), ("3: 45/35/20", 3, "45,35,20"), ("3: 50/30/20", 3, "50,30,20"), ("4 equal", 4, ""),
         ("4: 35/30/20/15", 4, "35,30,20,15"), ("4: 40/30/20/10", 4, "40,30,20,10"), ("5: 30/25/20/15/10", 5, "30,25,20,15,10"), ("2: 60/40", 2, "60,40")]


def child(K):
    sys.path.insert(0, '.')
    pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
    with pkg.MsmContext((0,)) as c:
        c.set_option("host_chunks", K)
        want = None
        for _ in range(6):
            want = c.run(pts, sc)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); r = c.run(pts, sc); ts.append((time.perf_counter() - t0) * 1e3)
            assert r == want
        print("%.3f (min %.3f) %s" % (statistics.median(ts), min(ts), want[:8].hex()), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
    else:
        for rnd in range(2):
            for name, K, split in CASES:
                e = dict(os.environ)
                if split:
                    e["TE_MSM_HOST_SPLIT"] = split
                r = subprocess.run([sys.executable, __file__, "child", str(K)], env=e, capture_output=True, text=True, timeout=300)
                print("round %d %-20s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
