#!/usr/bin/env python3
"""How the kernels of several MSMs in flight share the GPU: from a rocprofv3 kernel trace of the pipelined bench
(tools/profile_trace.sh), the time with 0 / 1 / 2 / ... kernels running, the time k_accumulate runs alone or beside another
kernel, and per kernel the mean duration in the pipelined steady state against its duration alone (--alone <dir>, a trace of
`bench.py --no-pipeline`).
usage: tools/overlap_report.py <pipelined trace dir> [--alone <no-pipeline trace dir>]"""
import csv, glob, os, sys
from collections import defaultdict


def load(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if name.startswith("void at::") or "elementwise" in name:
                continue
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].split("::")[-1].split("<")[0]))
    rows.sort()
    return rows


def main():
    rows = load(sys.argv[1])
    acc = [r for r in rows if r[2] == "k_accumulate"]
    lo, hi = acc[len(acc) // 3][0], acc[(2 * len(acc)) // 3][0]          # middle third: steady state
    n_msm = sum(1 for r in acc if lo <= r[0] < hi)
    ev = []
    for s, e, k in rows:
        if e <= lo or s >= hi:
            continue
        ev.append((max(s, lo), 1, k)); ev.append((min(e, hi), -1, k))
    ev.sort()
    running = defaultdict(int)
    t_prev, by_count, acc_alone, acc_with, no_acc = lo, defaultdict(int), 0, 0, 0
    for t, d, k in ev:
        dt = t - t_prev
        if dt > 0:
            c = sum(running.values())
            by_count[c] += dt
            if running["k_accumulate"] > 0:
                if c == running["k_accumulate"]: acc_alone += dt
                else: acc_with += dt
            elif c > 0:
                no_acc += dt
        running[k] += d
        t_prev = t
    span = hi - lo
    print("steady state: %d MSMs in %.2f ms -> %.3f ms per MSM" % (n_msm, span / 1e6, span / 1e6 / n_msm))
    print("kernels running at once: " + "  ".join("%d: %.0f %%" % (c, 100.0 * v / span) for c, v in sorted(by_count.items())))
    print("k_accumulate running: %.0f %% of the time (%.0f %% with nothing else, %.0f %% beside other kernels); other kernels without it: %.0f %%; idle: %.0f %%"
          % (100.0 * (acc_alone + acc_with) / span, 100.0 * acc_alone / span, 100.0 * acc_with / span, 100.0 * no_acc / span, 100.0 * by_count[0] / span))
    dur = defaultdict(list)
    for s, e, k in rows:
        if lo <= s < hi:
            dur[k].append((e - s) / 1e3)
    alone = {}
    if "--alone" in sys.argv:
        d2 = defaultdict(list)
        for s, e, k in load(sys.argv[sys.argv.index("--alone") + 1]):
            d2[k].append((e - s) / 1e3)
        alone = {k: sorted(v)[len(v) // 2] for k, v in d2.items()}
    print("%-28s %10s %12s" % ("kernel", "pipelined", "alone (median)"))
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        print("%-28s %8.1f us %10s" % (k[:28], sum(v) / len(v), ("%.1f us" % alone[k]) if k in alone else ""))
    print("sum of pipelined durations per MSM: %.0f us" % (sum(sum(v) for v in dur.values()) / n_msm))


if __name__ == "__main__":
    main()
