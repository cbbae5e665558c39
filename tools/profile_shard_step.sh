#!/bin/bash
# kernel trace of ONE rank's step of a D-GPU window-sharded run, rehearsed on one GPU (see bench.py TE_BENCH_REHEARSE_WORLD)
# usage: tools/profile_shard_step.sh D
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof_shard"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp TE_BENCH_FORCE_DIST=1 TE_BENCH_REHEARSE_WORLD=${1:-8}
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 $REPO/bench.py --steps 1024 --warmup 3 --no-cpu-baseline --no-sizes --no-host-buffers > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]))
rows.sort()
# steady state: between two k_accumulate launches inside the pipelined (timed) loop -- the middle of the launch sequence
acc_starts = [r[0] for r in rows if r[2] == "k_accumulate"]
i0, i1 = int(len(acc_starts) * 0.35), int(len(acc_starts) * 0.75)
lo, hi = acc_starts[i0], acc_starts[i1]
sel = [r for r in rows if r[0] >= lo and r[0] < hi]
dur = defaultdict(list)
for s, e, k in sel:
    dur[k].append((e - s) / 1e3)
busy, cur_s, cur_e = 0, None, None
for s, e, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = hi - lo
n_seq = i1 - i0
# MSMs per launch sequence: one k_digits launch covers every MSM of a batch since round 3, so the count comes from the bench line
import json
per_seq = 1
for line in open(os.path.join(sys.argv[1], "trace.log")):
    if line.startswith("{"):
        per_seq = int(json.loads(line).get("roofline", {}).get("timed_region", {}).get("msms_per_launch", 1) or 1)
n_acc = n_seq * per_seq
print("window %.2f ms, %d launch sequences (k_accumulate launches) carrying %d MSMs -> %.4f ms per MSM; GPU has at least one kernel running %.0f %% of the time" % (span / 1e6, n_seq, n_acc, span / 1e6 / max(n_acc, 1), 100.0 * busy / span))
print("sum of kernel durations per MSM: %.0f us (overlapped kernels counted separately)" % (sum(sum(v) for v in dur.values()) / max(n_acc, 1)))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s n=%4d mean %7.1f us  per MSM %7.1f us" % (k[:28], len(v), sum(v) / len(v), sum(v) / max(n_acc, 1)))
PY
