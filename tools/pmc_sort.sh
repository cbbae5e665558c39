#!/bin/bash
# SQ counters of the sort kernels, one MSM at a time: tools/pmc_sort.sh "<counters>"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof/pmc_sort"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc $1 --kernel-trace --output-format csv -d "$OUT" -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline --no-sizes --no-host-buffers --no-configs --repeats 1 > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("%-24s" % "kernel" + "".join("%22s" % c for c in names))
for k in ("k_digits", "k_part_scatter_prep", "k_l2_local", "k_l2_place_order", "k_accumulate", "k_seg_combine_all", "k_sum_groups", "k_sum_groups_team", "k_reduce_tail"):
    if k in acc:
        print("%-24s" % k + "".join("%22.4g" % (sum(acc[k][c]) / max(len(acc[k][c]), 1)) for c in names))
PY
