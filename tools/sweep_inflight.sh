#!/bin/bash
# MSMs in flight (bench.py --inflight k) at n = 2^20 from resident inputs, two rounds: ms per MSM of the driver's line for k = 2 .. 8
# tools/sweep_inflight.sh  -> gpurun_out/sweep_inflight.txt
out=gpurun_out/sweep_inflight.txt; : > $out
for rnd in 1 2; do
  for k in 2 3 4 5 6 8; do
    python bench.py --inflight $k --steps 48 --warmup 8 --no-cpu-baseline --no-sizes --no-configs --no-host-buffers > gpurun_out/_sweep_inflight.json
    python3 - $rnd $k >> $out <<'PY'
import json, sys
d = json.loads(open('gpurun_out/_sweep_inflight.json').read().strip().splitlines()[-1])
print("round %s in flight %s: %.4f ms per MSM = %.1f MSM/s   clock %.3f GHz" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["value"], d.get("roofline", {}).get("core_clock_ghz") or d.get("binding_roofline", {}).get("core_clock_ghz") or 0))
PY
  done
done
cat $out
