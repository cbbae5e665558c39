#!/bin/bash
# one PMC pass over the bench workload: tools/profile_pmc.sh "<counters>" [kernel-substring] [extra bench.py arguments]
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof/pmc_x"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d "$OUT" -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${3:-} > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" "${2:-k_accumulate}" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-28s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
