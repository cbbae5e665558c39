"""Bound bases, host scalars, 8 tickets in flight at n = 2^20: does the upload stand behind kernels of a work set that shares its hardware
queue?  The host stamps (tools/trace_bound_tickets.py) show a third of the 32 MB pageable copies taking 1.6-1.9 ms instead of 0.61.
Child processes, alternating, three rounds:
  marker     TE_MSM_COPY_MARKER=1 TE_MSM_LANE_EVENT_WAITS=1  (the uploads of every ticket wait for a marker on the work set's stream, and the lane
             thread for an event recorded behind the upload: the behaviour before this experiment)
  events     TE_MSM_LANE_EVENT_WAITS=1  (no marker for a work set whose previous MSM has delivered its result; the lane thread waits for an event)
  parallel   TE_MSM_SCALAR_UPLOADS_SERIAL=0  (the lanes' scalar uploads side by side on the link)
  ev+par     both of the above
  base       the default build          (no marker; the lane thread waits for the copy stream -- no packet of the upload path enters a hardware queue;
             scalars-only tickets cross the link one at a time)
  prio+      TE_MSM_COPY_PRIORITY=1   (copy streams of the greatest priority: hardware queues of their own)
  prio-      TE_MSM_COPY_PRIORITY=-1  (least priority)
  staging    option host_staging = 1 (uploads through the work sets' pinned rings)
  queues8    GPU_MAX_HW_QUEUES=8
python tools/exp_bound_copy_queue.py            (parent)
python tools/exp_bound_copy_queue.py child      (one measurement)"""
import importlib, os, subprocess, sys, time

CONFIGS = [("marker", {"TE_MSM_COPY_MARKER": "1", "TE_MSM_LANE_EVENT_WAITS": "1"}), ("events", {"TE_MSM_LANE_EVENT_WAITS": "1"}), ("parallel", {"TE_MSM_SCALAR_UPLOADS_SERIAL": "0"}),
           ("ev+par", {"TE_MSM_LANE_EVENT_WAITS": "1", "TE_MSM_SCALAR_UPLOADS_SERIAL": "0"}), ("base", {}), ("prio+", {"TE_MSM_COPY_PRIORITY": "1"}), ("prio-", {"TE_MSM_COPY_PRIORITY": "-1"}),
           ("staging", {"TE_MSM_HOST_STAGING": "1"}), ("queues8", {"GPU_MAX_HW_QUEUES": "8"})]
if os.environ.get("EXP_CONFIGS"):
    CONFIGS = [c for c in CONFIGS if c[0] in os.environ["EXP_CONFIGS"].split(",")]


def child():
    sys.path.insert(0, '.')
    pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
    import torch
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0x5EED0014, n, fixed_point="random")
    ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()

    def in_flight(c, submit, depth, steps=96):
        for t in [submit() for _ in range(depth)]:
            c.collect(t)
        ps = []
        for _ in range(3):
            t0 = time.perf_counter(); tk = []
            for _ in range(steps):
                tk.append(submit())
                if len(tk) >= depth:
                    c.collect(tk.pop(0))
            while tk:
                c.collect(tk.pop(0))
            ps.append((time.perf_counter() - t0) * 1e3 / steps)
        return "%.4f (%s)" % (min(ps), " ".join("%.3f" % x for x in ps))

    with pkg.MsmContext((0,)) as c:
        b = c.bind_points(pts)
        want = c.run_scalars(b, sc)
        assert c.collect(c.submit_scalars(b, sc)) == want
        print("device scalars x4 %s   host scalars x8 %s   x4 %s   x2 %s   host buffers x8 %s" % (
            in_flight(c, lambda: c.submit_scalars_device(b, ds.data_ptr()), 4), in_flight(c, lambda: c.submit_scalars(b, sc), 8),
            in_flight(c, lambda: c.submit_scalars(b, sc), 4), in_flight(c, lambda: c.submit_scalars(b, sc), 2), in_flight(c, lambda: c.submit_async(pts, sc), 8, 32)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rnd in range(3):
            for name, env in CONFIGS:
                e = dict(os.environ); e.update(env)
                r = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True, timeout=300)
                print("round %d %-8s %s" % (rnd, name, (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
