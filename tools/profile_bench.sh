#!/bin/bash
# rocprofv3 passes over the default bench workload (n = 2^20).  Run on the GPU box via gpurun.
#   pass 1: kernel trace + stats (per-kernel time)       pass 2/3: PMC FETCH_SIZE / WRITE_SIZE (own runs)
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="$REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-sizes --no-host-buffers"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
find "$OUT" -name "*.csv" | head -20
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
