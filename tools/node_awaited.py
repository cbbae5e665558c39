"""The reference harness's call pattern through the N-API addon (every compute_msm awaited, ui/Benchmark.tsx:29-39): median of 30 calls at n = 2^20 and
2^16, with and without setBases; the lone promise submitted from the JavaScript thread (the build) against TE_MSM_LONE_IN_ENTER=0 (from its pool thread).
python tools/node_awaited.py"""
import importlib, json, os, shutil, subprocess, sys, tempfile
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
node = shutil.which("node") or shutil.which("nodejs")
js = os.path.join(os.path.dirname(pkg.__file__), "js")
with tempfile.TemporaryDirectory() as t:
    for lg in (20, 16):
        pts, sc = pkg.synth_inputs(0x5EED0000 + lg, 1 << lg, fixed_point="random")
        open(os.path.join(t, "p.bin"), "wb").write(pts); open(os.path.join(t, "s.bin"), "wb").write(sc)
        for rnd in range(3):
            for name, env in (("lone in pool thread", {"TE_MSM_LONE_IN_ENTER": "0"}), ("lone in enter", {})):
                row = []
                for bases in ((), ("bases",)):
                    r = subprocess.run([node, os.path.join(js, "run_awaited.js"), os.path.join(t, "p.bin"), os.path.join(t, "s.bin"), "30"] + list(bases),
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=dict(os.environ, **env))
                    out = json.loads(r.stdout.decode().strip().splitlines()[-1])
                    assert "x" in out, (out, r.stderr.decode()[-1000:])
                    row.append("%s median %.3f min %.3f (in enter %d, in execute %d)" % ("bases" if bases else "plain", out["median_ms"], out["min_ms"],
                                                                                      out["stats"]["submittedInEnter"], out["stats"]["submittedInExecute"]))
                print("2^%d round %d %-20s %s" % (lg, rnd, name, "   ".join(row)), flush=True)
