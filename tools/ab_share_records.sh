#!/bin/bash
# Round-6 A/B: shared record slabs for MSMs in flight that name the same device-resident point buffer (TE_MSM_SHARE_RECORDS=0/1), same box, alternating:
# the headline (bench.py, 4 in flight), then 2 and 8 in flight and the other harness sizes.  -> gpurun_out/r06_ab_share_records.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
OUT=gpurun_out/r06_ab_share_records.txt; : > $OUT
line() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-34s %7.1f MSM/s  passes %s  latency %.4f  accumulate alone %.4f ms  clock alone %.3f  clock in flight %.3f GHz' % ('$1', d['value'], ' '.join('%.4f' % x for x in d['passes_ms_per_step']), d['latency_ms'], r['kernel_ms'], r['binding_roofline']['core_clock_ghz'], (r.get('timed_region') or {}).get('core_clock_ghz') or 0))"; }
ARGS="--steps 100 --warmup 5 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs"
for round in 1 2 3; do
  for v in 0 1; do
    TE_MSM_SHARE_RECORDS=$v python3 bench.py $ARGS 2>/dev/null | grep '^{' | tail -1 | line "round $round share_records=$v 4 in flight" >> $OUT
  done
done
for extra in "--inflight 2" "--inflight 8" "--log2n 18 --window-bits 0" "--log2n 16 --window-bits 0" "--points fixed"; do
  for v in 0 1; do
    TE_MSM_SHARE_RECORDS=$v python3 bench.py $ARGS $extra 2>/dev/null | grep '^{' | tail -1 | line "share_records=$v $extra" >> $OUT
  done
done
cat $OUT
