#!/bin/bash
# Window size against n on the current build (auto_window_bits' boundaries were measured in rounds 1 and 4; sort and reduction have changed since):
# bench.py --log2n L --window-bits c for L = 16..19, c = 13..16; ms per MSM in flight / latency of one.   -> gpurun_out/sweep_window_bits.txt
out=gpurun_out/sweep_window_bits.txt; : > $out
for L in 16 17 18 19; do
  for c in 13 14 15 16; do
    python bench.py --log2n $L --window-bits $c --steps 60 --warmup 8 --no-cpu-baseline --no-sizes --no-configs --no-host-buffers > gpurun_out/_sweep_wb.json 2>gpurun_out/_sweep_wb.err || { echo "2^$L c=$c failed" >> $out; continue; }
    python3 - $L $c >> $out <<'PY'
import json, sys
d = json.loads(open('gpurun_out/_sweep_wb.json').read().strip().splitlines()[-1])
print("n=2^%s c=%s: %.4f ms per MSM in flight, latency %.4f ms" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d.get("latency_ms") or d["config"].get("latency_ms")))
PY
  done
done
cat $out
