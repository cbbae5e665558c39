"""One-off parity check beyond the test suite's sizes: n = 2^24 from host buffers against the oracle (about a minute;
round 1 on an MI355X: 33.6 ms per MSM including the 1.5 GB upload, bit-exact)."""
import importlib, sys, time
sys.path.insert(0, ".")
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
from oracle import oracle
n = 1 << 24
t = time.time(); pts, sc = pkg.synth_inputs(99, n); print("inputs", round(time.time() - t, 1), "s", flush=True)
with pkg.MsmContext((0,)) as c:
    t = time.time(); got = c.run(pts, sc); print("gpu run (first, with allocation)", round((time.time() - t) * 1e3, 1), "ms", flush=True)
    t = time.time(); got2 = c.run(pts, sc); print("gpu run", round((time.time() - t) * 1e3, 1), "ms", flush=True)
t = time.time(); exp = oracle.msm(pts, sc, c=16, threads=16); print("oracle", round(time.time() - t, 1), "s", flush=True)
print("n=2^24 parity:", got == exp and got2 == exp)
