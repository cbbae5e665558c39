#!/bin/bash
# segment length against the LATENCY of one MSM and the pipelined throughput, n = 2^16..2^20 (window bits chosen by the engine)
mkdir -p gpurun_out
out=gpurun_out/sweep_seg_latency.txt; : > $out
for lg in 16 17 18 19 20; do
  for seg in 8 16 32 64; do
    timeout -k 5 100 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs --log2n $lg --window-bits 0 --segment-len $seg > gpurun_out/_s.log 2>&1 || { echo "FAILED lg=$lg seg=$seg" >> $out; continue; }
    python - "$lg" "$seg" <<'PY' >> $out
import json, sys
for l in open("gpurun_out/_s.log"):
    if l.startswith("{"):
        j = json.loads(l); st = j["stage_ms_untimed_pass"]
        print("n=2^%s seg=%-3s  latency %.4f ms  %.4f ms/step  accumulate alone %4d us  marginal %4d us" % (sys.argv[1], sys.argv[2], j["latency_ms"], j["ms_per_step"], st["accumulate"]*1000, st["marginal_sums"]*1000))
PY
  done
done
cat $out
