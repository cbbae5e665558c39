"""Merges the per-config PMC passes of tools/profile_configs.sh into <prof>/traffic.json (written by tools/summarize_prof.py):
"configs": {name: {"workload": ..., "kernels": {short kernel name: {fetch_bytes_raw, write_bytes, hbm_bytes_per_launch}}}}.
Same correction as the default workload: hbm_bytes_per_launch = TE_FETCH_FACTOR (2.0) x FETCH_SIZE + WRITE_SIZE."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

WORKLOADS = {"unsigned": "bench.py --digits unsigned (n = 2^20, 16-bit unsigned windows, one GPU)",
             "bls12_377": "bench.py --curve bls12-377 (n = 2^20, BLS12-377 G1, 16-bit signed windows, one GPU)",
             "harness_fixed_point": "bench.py --points fixed (n = 2^20, one fixed point replicated, 16-bit signed windows, one GPU)",
             "witness_scalars": "bench.py --scalars mixed (n = 2^20, a quarter zeros, a quarter ones, the rest uniform, 16-bit signed windows, one GPU)"}


def short(k):
    return k.split("(")[0].split("::")[-1].split("<")[0]


def main(root):
    factor = float(os.environ.get("TE_FETCH_FACTOR", "2.0"))
    tj = os.path.join(root, "traffic.json")
    j = json.load(open(tj)) if os.path.exists(tj) else {}
    j.setdefault("configs", {})
    for name, wl in WORKLOADS.items():
        per = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            acc = defaultdict(list)
            for f in glob.glob(os.path.join(root, "cfg_%s_%s" % (name, ctr), "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") == ctr and not (r["Kernel_Name"].startswith("void at::") or "elementwise" in r["Kernel_Name"]):
                        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)      # counter unit: KiB per dispatch
            for k, v in acc.items():
                per.setdefault(k, {})[ctr] = sum(v) / len(v)
        kern = {k: {"fetch_bytes_raw": v["FETCH_SIZE"], "write_bytes": v["WRITE_SIZE"],
                    "hbm_bytes_per_launch": factor * v["FETCH_SIZE"] + v["WRITE_SIZE"]}
                for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
        if kern:
            j["configs"][name] = {"workload": wl, "kernels": kern}
            a = kern.get("k_accumulate")
            if a:
                print("%-22s k_accumulate: fetch raw %.1f MB, write %.1f MB -> %.3f GB per launch" % (name, a["fetch_bytes_raw"] / 1e6, a["write_bytes"] / 1e6, a["hbm_bytes_per_launch"] / 1e9))
    json.dump(j, open(tj, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
