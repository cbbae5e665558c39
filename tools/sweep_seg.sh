#!/bin/bash
# segment length sweep: per-rank step of a D-GPU window-sharded run (rehearsed on one GPU) and small single-GPU sizes
mkdir -p gpurun_out
out=gpurun_out/sweep_seg.txt; : > $out
run() {  # label, env D, log2n, seg
  local D=$1 lg=$2 seg=$3
  if [ "$D" = "1" ]; then
    timeout -k 5 100 python bench.py --steps 60 --warmup 5 --inflight 4 --no-cpu-baseline --no-sizes --no-host-buffers --log2n $lg --window-bits 0 --segment-len $seg > gpurun_out/_s.log 2>&1 || { echo "FAILED D=$D lg=$lg seg=$seg" >> $out; return 1; }
  else
    TE_BENCH_FORCE_DIST=1 TE_BENCH_REHEARSE_WORLD=$D timeout -k 5 100 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-sizes --no-host-buffers --log2n $lg --segment-len $seg > gpurun_out/_s.log 2>&1 || { echo "FAILED D=$D lg=$lg seg=$seg" >> $out; return 1; }
  fi
  python - "$D" "$lg" "$seg" <<'PY' >> $out
import json, sys
for l in open("gpurun_out/_s.log"):
    if l.startswith("{"):
        j = json.loads(l); st = j["stage_ms_untimed_pass"]
        print("D=%s n=2^%s seg=%-3s  %.4f ms/step  %7.1f MSM/s  accumulate %4d us  marginal %4d us" % (sys.argv[1], sys.argv[2], sys.argv[3], j["ms_per_step"], j["value"], st["accumulate"]*1000, st["marginal_sums"]*1000))
PY
}
for seg in 8 12 16 24 32 64; do run 8 20 $seg || exit 1; done
for seg in 12 16 24 32 64; do run 4 20 $seg || exit 1; done
for seg in 16 24 32 48 64; do run 2 20 $seg || exit 1; done
for seg in 32 48 64 96 128; do run 1 20 $seg || exit 1; done
for seg in 8 16 32 64; do run 1 18 $seg || exit 1; done
for seg in 4 8 16 32 64; do run 1 16 $seg || exit 1; done
cat $out
