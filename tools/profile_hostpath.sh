#!/bin/bash
# te_msm_run from host buffers at n = 2^20: the library's own host stamps (TE_MSM_TRACE_HOST) and a rocprofv3 kernel + memory-copy
# trace of the same calls -> gpurun_out/hostpath/
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/hostpath"; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$REPO" && python3 tools/host_trace.py > "$OUT/stamps.txt" 2>&1
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/trace" -- python3 $REPO/tools/host_trace.py > "$OUT/trace.log" 2>&1 || { tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]))
for f in glob.glob(os.path.join(root, "trace", "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))))
ev.sort()
# the last te_msm_run of the K = 3 block: find groups separated by > 2 ms of silence, print the 4th group (3 warm runs + traced one)
groups, cur = [], []
for e in ev:
    if cur and e[0] - cur[-1][1] > 1_500_000:
        groups.append(cur); cur = []
    cur.append(e)
groups.append(cur)
big = [g for g in groups if len(g) > 20]
g = big[3] if len(big) > 3 else big[-1]
t0 = g[0][0]
print("timeline of one te_msm_run (3 pieces), us from the first device activity; %d events" % len(g))
for s, e, k in g:
    print("%9.1f %9.1f  %7.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k))
PY
