#!/bin/bash
# Round-6 A/B (verdict item 4): the sort kernels compiled for a footprint that fits beside three resident accumulation waves per SIMD
# (libtemsm_smallsort.so: make -C webgpu-msm-twisted-edwards_amd/csrc smallsort) against the default build, same box, alternating:
# pipelined MSM/s (bench.py, 4 in flight) and the overlap report of a kernel trace of each.  -> gpurun_out/r06_ab_smallsort.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
A="$REPO/webgpu-msm-twisted-edwards_amd/libtemsm.so"; B="$REPO/webgpu-msm-twisted-edwards_amd/libtemsm_smallsort.so"
OUT=gpurun_out/r06_ab_smallsort.txt; : > $OUT
ARGS="--steps 100 --warmup 5 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs"
for round in 1 2 3; do
  for lib in "$A" "$B"; do
    TE_MSM_LIB=$lib python3 bench.py $ARGS 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('round $round %-28s %.1f MSM/s  passes %s  accumulate alone %.4f ms  clock alone %.3f GHz  clock pipelined %.3f GHz' % ('$(basename $lib)', d['value'], ' '.join('%.4f' % x for x in d['passes_ms_per_step']), r['kernel_ms'], r['binding_roofline']['core_clock_ghz'], r['timed_region']['core_clock_ghz']))" >> $OUT
  done
done
for lib in "$A" "$B"; do
  export TE_MSM_LIB=$lib
  bash tools/profile_trace.sh --no-sizes --no-host-buffers --no-configs --steps 60 > gpurun_out/r06_ab_trace_$(basename $lib .so).txt 2>&1
  echo "==== overlap, $(basename $lib)" >> $OUT
  python3 tools/overlap_report.py gpurun_out/prof/trace >> $OUT 2>&1
done
unset TE_MSM_LIB
cat $OUT
