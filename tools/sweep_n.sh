#!/bin/bash
# MSM latency / throughput for n = 2^16..2^20 and a few window sizes
for lg in 16 17 18 19 20; do
  for c in 0 12 13 14 15 16; do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --log2n $lg --window-bits $c 2>/dev/null | tail -1 > /tmp/sw.json
    python - $lg $c <<'PY'
import json, sys
d = json.load(open("/tmp/sw.json"))
print("log2n", sys.argv[1], "c", sys.argv[2], "->", d["config"]["workload"].split(",")[1].strip()[:40], "ms/step", round(d["ms_per_step"], 4), "latency", round(d["latency_ms_single_msm"], 4), "acc", round(d["roofline"]["kernel_ms"], 4))
PY
  done
done
