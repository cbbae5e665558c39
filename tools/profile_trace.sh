#!/bin/bash
# kernel-trace-only rocprofv3 pass over the default bench workload; prints the per-kernel summary
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof"; rm -rf "$OUT/trace"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 "$REPO/tools/summarize_prof.py" "$OUT" 2>/dev/null | sed -n '1,22p'
