"""Summarises the rocprofv3 CSVs written by tools/profile_bench.sh: per-kernel mean duration (kernel
trace) and per-kernel mean FETCH_SIZE / WRITE_SIZE (PMC passes), corrected as MI355X_MICROARCH.md
prescribes (FETCH_SIZE reads half of a wide coalesced stream on gfx950: reported raw and x2)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    return sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))


def main(root):
    # kernel trace
    dur = defaultdict(list)
    for f in find(os.path.join(root, "trace"), "*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("== kernel trace (us): name, calls, mean, min, max, total")
    tot = sum(sum(v) for v in dur.values())
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if k.startswith("void at::") or "elementwise" in k:
            continue
        print("%-60s %5d %10.1f %10.1f %10.1f %12.1f  %5.1f%%" % (k[:60], len(v), sum(v) / len(v), min(v), max(v), sum(v), 100 * sum(v) / tot))
    # launches per MSM: every dispatch of the engine (kernels, fills, copies) over the number of k_accumulate launches
    n_msm = sum(len(v) for k, v in dur.items() if "k_accumulate" in k or "k377_accumulate" in k)
    if n_msm:
        mine = {k: len(v) for k, v in dur.items() if not (k.startswith("void at::") or "elementwise" in k)}
        print("== launches per MSM: %.2f  (%d dispatches / %d MSMs; kernels %.2f, fills %.2f, copies %.2f)" % (
            sum(mine.values()) / n_msm, sum(mine.values()), n_msm,
            sum(v for k, v in mine.items() if not k.startswith("__amd_rocclr")) / n_msm,
            sum(v for k, v in mine.items() if "fillBuffer" in k) / n_msm, sum(v for k, v in mine.items() if "copyBuffer" in k) / n_msm))
    # pmc
    traffic = {}
    for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        acc = defaultdict(list)
        for f in find(os.path.join(root, sub), "*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == name:
                    acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        print("== %s (counter units are KiB per dispatch; mean over dispatches)" % name)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            if k.startswith("void at::") or "elementwise" in k:
                continue
            m = sum(v) / len(v)
            extra = "  x2(gfx950 wide-read correction) = %.1f MB" % (2 * m * 1024 / 1e6) if name == "FETCH_SIZE" else ""
            print("%-60s %5d %14.1f KiB = %10.1f MB%s" % (k[:60], len(v), m, m * 1024 / 1e6, extra))
            short = k.split("(")[0].split("::")[-1].split("<")[0]
            traffic.setdefault(short, {})[name] = m * 1024
    # HBM bytes per launch: FETCH_SIZE x factor + WRITE_SIZE.  MI355X_MICROARCH.md prescribes factor 2 for wide coalesced
    # streaming reads on gfx950 and calibration for anything else; tools/calibrate_fetch.sh measures the factor for
    # k_accumulate's 128-byte record gather (profiles/r02_fetch_calibration.txt) -- TE_FETCH_FACTOR carries it here.
    factor = float(os.environ.get("TE_FETCH_FACTOR", "2.0"))
    out = {k: {"fetch_bytes_raw": v.get("FETCH_SIZE"), "write_bytes": v.get("WRITE_SIZE"),
               "hbm_bytes_per_launch": factor * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0),
               "mean_us": (sum(dur[n]) / len(dur[n])) if (n := next((d for d in dur if d.split("(")[0].split("::")[-1].split("<")[0] == k), None)) else None}
           for k, v in traffic.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webgpu-msm-twisted-edwards_amd", "csrc")
    import re
    for f in ("kernels.hip.hpp", "curve.hpp", "fp.hpp"):          # the same list and the same rule as bench.py kernel_sources_sha
        text = open(os.path.join(csrc, f), "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    json.dump({"workload": "bench.py default (n = 2^20, c = 16, one GPU)", "commit": os.environ.get("TE_COMMIT", "unknown"),
               "kernel_sources_sha": h.hexdigest()[:16],
               "correction": "hbm_bytes_per_launch = %.2f x FETCH_SIZE + WRITE_SIZE (factor: see profiles/r02_fetch_calibration.txt)" % factor,
               "kernels": out}, open(os.path.join(root, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
