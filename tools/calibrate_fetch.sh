#!/bin/bash
# FETCH_SIZE calibration on known byte counts (tools/ubench_gather.hip).  Run on the GPU box via gpurun.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/cal"; rm -rf "$OUT"; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o "$OUT/ubench_gather" "$REPO/tools/ubench_gather.hip" || exit 1
export TMPDIR=/tmp; cd /tmp
"$OUT/ubench_gather" > "$OUT/plain.txt" 2>&1 || { cat "$OUT/plain.txt"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc" -- "$OUT/ubench_gather" > "$OUT/pmc.log" 2>&1 || { tail -5 "$OUT/pmc.log"; exit 1; }
python3 - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, os, re, sys
out = sys.argv[1]
known = {}
for l in open(os.path.join(out, "plain.txt")):
    m = re.match(r"CAL (k_stream|k_gather<\d>)\s+(.*?)\s+known_bytes (\d+)\s+([\d.]+) us", l)
    if m:
        known[m.group(1)] = (float(m.group(3)), m.group(2).strip(), float(m.group(4)))
print("# FETCH_SIZE calibration, MI355X (tools/ubench_gather.hip under rocprofv3 --pmc FETCH_SIZE); counter unit = KiB")
print("# kernel | access pattern | known bytes (lines x 128 B) | FETCH_SIZE bytes | ratio FETCH/known | factor to apply")
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "FETCH_SIZE":
            continue
        name = r["Kernel_Name"]
        key = "k_stream" if "k_stream" in name else None
        m = re.search(r"k_gather<(\d)>", name)
        if m: key = "k_gather<%s>" % m.group(1)
        if key in known:
            kb, what, us = known[key]
            v = float(r["Counter_Value"]) * 1024.0
            print("%-12s | %-58s | %14.0f | %14.0f | %.3f | x%.2f" % (key, what, kb, v, v / kb, kb / v if v else float("nan")))
PY
