"""The drop-in end to end: compute_msm promises through the N-API addon (js/run_concurrent.js) at n = 2^20, random points, k = 1 / 2 / 4 / 8 in flight,
with and without setBases.  python tools/node_in_flight.py"""
import importlib, json, os, shutil, subprocess, sys, tempfile
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
node = shutil.which("node") or shutil.which("nodejs")
js = os.path.join(os.path.dirname(pkg.__file__), "js")
n = 1 << 20
pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
with tempfile.TemporaryDirectory() as t:
    open(os.path.join(t, "p.bin"), "wb").write(pts); open(os.path.join(t, "s.bin"), "wb").write(sc)
    for rnd in range(2):
        for k in (1, 2, 4, 8):
            for bases in ((), ("-", "bases")):
                r = subprocess.run([node, os.path.join(js, "run_concurrent.js"), os.path.join(t, "p.bin"), os.path.join(t, "s.bin"), str(k)] + list(bases),
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
                out = json.loads(r.stdout.decode().strip().splitlines()[-1])
                assert out.get("all_equal"), (out, r.stderr.decode()[-1000:])
                print("round %d k=%d %-6s single %.3f ms   %d in flight: %.3f ms = %.3f ms per MSM   stats %s" % (
                    rnd, k, "bases" if bases else "", out["single_ms"], k, out["concurrent_ms"], out["concurrent_ms"] / k, out["stats"]), flush=True)
