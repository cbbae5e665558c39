#!/bin/bash
# A/B of two builds of the library on one box, alternating: tools/ab_lib.sh <tag> <variant.so> [rounds]   -> gpurun_out/<tag>.txt
# (TE_MSM_LIB selects the build; the default build is libtemsm.so)
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
TAG=$1; B="$REPO/webgpu-msm-twisted-edwards_amd/$2"; A="$REPO/webgpu-msm-twisted-edwards_amd/libtemsm.so"; R=${3:-3}
OUT=gpurun_out/$TAG.txt; : > $OUT
ARGS="--steps 100 --warmup 5 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs"
for round in $(seq 1 $R); do
  for lib in "$A" "$B"; do
    TE_MSM_LIB=$lib python3 bench.py $ARGS 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('round $round %-26s %7.1f MSM/s  passes %s  latency %.4f  accumulate alone %.4f ms  clock alone %.3f  in flight %.3f GHz' % ('$(basename $lib)', d['value'], ' '.join('%.4f' % x for x in d['passes_ms_per_step']), d['latency_ms'], r['kernel_ms'], r['binding_roofline']['core_clock_ghz'], r['timed_region']['core_clock_ghz']))" >> $OUT
  done
done
cat $OUT
