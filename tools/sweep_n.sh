#!/bin/bash
# MSM throughput / latency for n = 2^16..2^20: automatic window size and the neighbouring ones, four MSMs
# in flight.  Writes gpurun_out/sweep_n.txt (copied to profiles/ per round).
mkdir -p gpurun_out
out=gpurun_out/sweep_n.txt
echo "# n sweep, one MI355X, bench.py --inflight 4 --steps 100 (window bits 0 = automatic rule)" > $out
for lg in ${LGS:-16 17 18 19 20}; do
  for c in ${CS:-0 14 15 16}; do
    timeout -k 5 120 python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs --log2n $lg --window-bits $c 2>/dev/null | tail -1 > gpurun_out/_sw.json || { echo "FAILED lg=$lg c=$c" >> $out; exit 1; }
    python - $lg $c <<'PY' >> $out
import json, sys
d = json.load(open("gpurun_out/_sw.json"))
st = d["stage_ms_untimed_pass"]
print("n=2^%s c=%-2s -> %-28s %8.4f ms/MSM %8.1f MSM/s  latency %.3f ms  accumulate alone %.3f ms  whole-MSM algorithmic %.0f GB/s"
      % (sys.argv[1], sys.argv[2], d["config"]["workload"].split(",")[1].strip()[:28], d["ms_per_step"], d["value"],
         d["latency_ms_single_msm"], st["accumulate"], d["msm_algorithmic_gbps"]))
PY
  done
done
cat $out
