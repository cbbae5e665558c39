"""Does the host-buffer path depend on the caller REUSING its buffers?  The runtime may keep user pages pinned (or remember
them) between copies from the same addresses; a prover hands over different buffers every time.  Times te_msm_run with
(a) the same two buffers every call, (b) a rotation of 8 distinct copies, (c) a freshly allocated copy per call (allocation and
fill outside the timed region) -- on one device, on the one GPU named 8 times (point slices, 8 upload threads) and as
asynchronous tickets.   python tools/h2d_fresh_buffers.py [log2n]"""
import importlib, sys, time
sys.path.insert(0, '.')
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << lg
pts, sc = pkg.synth_inputs(0x5EED0000 + lg, n)
rot = [(bytes(bytearray(pts)), bytes(bytearray(sc))) for _ in range(8)]


def series(name, call, pick, k=24):
    ts = []
    for i in range(k):
        p, s = pick(i)
        t0 = time.perf_counter(); r = call(p, s); ts.append((time.perf_counter() - t0) * 1e3)
        assert r == ref
    srt = sorted(ts)
    print("%-64s best %.3f median %.3f worst %.3f | %s" % (name, srt[0], srt[len(srt) // 2], srt[-1], " ".join("%.2f" % t for t in ts)))


import os
for ids, staging in (((0,), 0), ((0,), 1), ((0,) * 8, 0), ((0,) * 8, 1)):
    if os.environ.get("TE_H2D_ONLY_STAGED") and not staging:
        continue
    if os.environ.get("TE_H2D_D1_ONLY") and len(ids) > 1:
        continue
    with pkg.MsmContext(ids) as c:
        c.set_option("host_staging", staging)
        ref = c.run(pts, sc)
        tag = "D = %d te_msm_run%s" % (len(ids), ", host_staging" if staging else "")
        series(tag + ", same buffers", c.run, lambda i: (pts, sc))
        series(tag + ", rotation of 8 copies (first round = first touch)", c.run, lambda i: rot[i % 8])
        series(tag + ", rotation of 8 copies again", c.run, lambda i: rot[i % 8])
        series(tag + ", a fresh copy per call", c.run, lambda i: (bytes(bytearray(pts)), bytes(bytearray(sc))), 12)
        series(tag + ", same buffers again", c.run, lambda i: (pts, sc), 8)
for staging in (() if os.environ.get("TE_H2D_D1_ONLY") else (0, 1)):
  with pkg.MsmContext((0,) * 8) as c:
    c.set_option("host_staging", staging)
    print("-- tickets, host_staging = %d" % staging)
    for name, pick in (("same buffers", lambda i: (pts, sc)), ("rotation of 8 copies", lambda i: rot[i % 8])):
          for rep in range(3):
              t0 = time.perf_counter()
              tk = []
              for i in range(48):
                  tk.append(c.submit_async(*pick(i)))
                  if len(tk) >= 16:
                      assert c.collect(tk.pop(0)) == ref
              while tk:
                  assert c.collect(tk.pop(0)) == ref
              print("D = 8 tickets (te_msm_submit_async, 16 in flight), %s: %.3f ms per MSM" % (name, (time.perf_counter() - t0) * 1e3 / 48))
