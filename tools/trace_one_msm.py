#!/usr/bin/env python3
"""Timeline of ONE MSM from a rocprofv3 kernel trace of `bench.py --no-pipeline` (tools/profile_trace.sh --no-pipeline):
per launch the mean duration and the mean idle gap in front of it, over the MSMs of the steady state.
usage: tools/trace_one_msm.py <dir with *kernel_trace.csv> [first-kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "fillBuffer"
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if name.startswith("void at::") or "elementwise" in name:
            continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].split("::")[-1].split("<")[0]))
rows.sort()
# split into MSMs at every `first` kernel
msms, cur = [], []
for s, e, k in rows:
    if first in k and cur:
        msms.append(cur); cur = []
    cur.append((s, e, k))
if cur:
    msms.append(cur)
# steady state: the most common launch sequence
seqs = defaultdict(list)
for m in msms:
    seqs[tuple(k for _, _, k in m)].append(m)
seq, group = max(seqs.items(), key=lambda kv: len(kv[1]))
group = group[len(group) // 4:]                      # drop warm-up
print("%d MSMs with the common sequence of %d launches (of %d MSM-like groups in the trace)" % (len(group), len(seq), len(msms)))
span = sum(m[-1][1] - m[0][0] for m in group) / len(group) / 1e3
busy = sum(sum(e - s for s, e, _ in m) for m in group) / len(group) / 1e3
print("device span first launch -> last launch end: %.1f us; sum of durations %.1f us; idle between launches %.1f us" % (span, busy, span - busy))
print("%-28s %10s %10s" % ("launch", "mean us", "gap before"))
for i, k in enumerate(seq):
    d = sum(m[i][1] - m[i][0] for m in group) / len(group) / 1e3
    g = sum((m[i][0] - m[i - 1][1]) for m in group) / len(group) / 1e3 if i else 0.0
    print("%-28s %10.1f %10.1f" % (k[:28], d, g))
