#!/bin/bash
# Round-6 A/B: shared record slabs in the WINDOW-SHARDED building-block path (te_msm_partial_device[_batch]: ShardedPipeline), one rank's step of a
# D-GPU run rehearsed on one GPU (TE_BENCH_FORCE_DIST=1 TE_BENCH_REHEARSE_WORLD=D, RCCL with one rank).  -> gpurun_out/r06_ab_share_records_shards.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
OUT=gpurun_out/r06_ab_share_records_shards.txt; : > $OUT
export TE_BENCH_FORCE_DIST=1
for D in 8 4 2; do
  for round in 1 2; do
    for v in 0 1; do
      TE_BENCH_REHEARSE_WORLD=$D TE_MSM_SHARE_RECORDS=$v python3 bench.py --steps 512 --warmup 3 --no-cpu-baseline --no-sizes --no-host-buffers 2>/dev/null | grep '^{' | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('D=$D round $round share_records=$v: per-rank step %.4f ms per MSM (passes %s)  distinct bases %s  mode: %s' % (d['ms_per_step'], ' '.join('%.4f' % x for x in d['passes_ms_per_step']), (d.get('batch_distinct_bases') or {}).get('ms_per_step'), d['mode'][:70]))" >> $OUT
    done
  done
done
cat $OUT
