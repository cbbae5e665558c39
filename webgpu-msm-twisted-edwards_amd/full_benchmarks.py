"""Benchmark protocol of the reference (submission/miscellaneous/full_benchmarks.ts:6-163) over this engine:
for every power 16..20 one first run with `force_recompile` (here: a fresh context, i.e. buffer allocation and
module load -- the counterpart of WGSL compilation), then NUM_RUNS more runs DELAY ms apart, every run a full
`compute_msm(bufferPoints, bufferScalars)` from host buffers (upload + device stages + read-back + host tail), and
the same Markdown table: | MSM size | 1st run | Run 1..5 | Average (incl 1st) | Average (excl 1st) |.

Inputs: the ZPrize files if `--data DIR` (or $TE_ZPRIZE_DATA) holds them (test-data/testCases.ts:35-52; results are
then compared with the five expected points), else the engine's seeded synthetic inputs (te_msm_synth_inputs).

`--csv FILE` also writes the harness's export (ui/CSVExportButton.tsx:8-23 over the rows ui/AllBenchmarks.tsx:45-52
collects): a header row "InputSize","MSM Func","Time (MS)" and one quoted row per timed call.

    python -m webgpu-msm-twisted-edwards_amd.full_benchmarks        (python full_benchmarks.py works too)
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

DELAY_MS = 100          # full_benchmarks.ts:10
NUM_RUNS = 5            # full_benchmarks.ts:11
START_POWER, END_POWER = 16, 20
CSV_HEADER = ["InputSize", "MSM Func", "Time (MS)"]          # ui/AllBenchmarks.tsx:45
MSM_FUNC_NAME = "Submission"                                  # the name the harness gives compute_msm (ui/AllBenchmarks.tsx:213-222)


def to_csv(rows) -> str:
    """convertToCSV of ui/CSVExportButton.tsx:9-11: every cell in double quotes, cells joined by ',', rows by newline"""
    return "\n".join(",".join('"%s"' % cell for cell in row) for row in rows)


def csv_rows(all_results):
    """the rows postResult collects (ui/AllBenchmarks.tsx:49-52): [input size (the power), function name, time in ms]"""
    rows = [list(CSV_HEADER)]
    for power in sorted(all_results):
        r = all_results[power]
        for t in [r["first_run_elapsed"]] + list(r["subsequent_runs"]):
            rows.append([power, MSM_FUNC_NAME, t])
    return rows


def _pkg():
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    return importlib.import_module(os.path.basename(here))


def load_case(pkg, power: int, data_dir: str | None):
    """(bufferPoints, bufferScalars, expected {x, y} or None)"""
    if data_dir:
        from .testdata import load_test_case, expected_result
        pp = os.path.join(data_dir, "points", f"{power}-power-points.txt")
        sp = os.path.join(data_dir, "scalars", f"{power}-power-scalars.txt")
        if os.path.exists(pp) and os.path.exists(sp):
            pts, sc = load_test_case(pp, sp)
            return pts, sc, expected_result(power)
    pts, sc = pkg.synth_inputs(0x5EED0000 + power, 1 << power)
    return pts, sc, None


def run(powers, data_dir=None, num_runs=NUM_RUNS, delay_ms=DELAY_MS, out=sys.stdout):
    pkg = _pkg()
    print(f"Running benchmarks for powers {powers[0]} to {powers[-1]} (inclusive)", file=out)
    cases = {p: load_case(pkg, p, data_dir) for p in powers}           # load test cases in advance (:24-41)
    all_results = {}
    do_recompile = True
    for power in powers:
        print(f"Running {num_runs + 1} invocations of compute_msm() for 2^{power} inputs, please wait...", file=out)
        pts, sc, expected = cases[power]
        t0 = time.perf_counter()
        msm = pkg.compute_msm(pts, sc, False, do_recompile)
        first = (time.perf_counter() - t0) * 1e3
        do_recompile = False                                           # only on the first run, whatever the power (:70-72)
        if expected is not None and (msm["x"] != expected["x"] or msm["y"] != expected["y"]):
            print(f"WARNING: the result of compute_msm is incorrect for 2^{power}", file=out)
        time.sleep(delay_ms / 1e3)
        runs = []
        for _ in range(num_runs):
            t0 = time.perf_counter()
            pkg.compute_msm(pts, sc, False, False)
            runs.append((time.perf_counter() - t0) * 1e3)
            time.sleep(delay_ms / 1e3)
        all_results[power] = {"first_run_elapsed": first, "subsequent_runs": runs,
                              "full_average": (first + sum(runs)) / (1 + len(runs)),
                              "subsequent_average": sum(runs) / len(runs),
                              "checked_against_expected": expected is not None}
    header = "| MSM size | 1st run |" + "".join(f" Run {i + 1} |" for i in range(num_runs))
    header += " Average (incl 1st) | Average (excl 1st) |\n|-|-|-|-|" + "-|" * num_runs + "\n"
    body = ""
    for power in powers:
        r = all_results[power]
        body += f"| 2^{power} | `{r['first_run_elapsed']:.2f}` |" + "".join(f" `{v:.2f}` |" for v in r["subsequent_runs"])
        body += f" **`{r['full_average']:.2f}`** | **`{r['subsequent_average']:.2f}`** |\n"
    print(header + body.strip(), file=out)
    print("times in ms, host buffers in / affine point out; the first run of the first size includes context creation", file=out)
    return all_results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=os.environ.get("TE_ZPRIZE_DATA"), help="directory with points/ and scalars/ (ZPrize test data)")
    ap.add_argument("--start", type=int, default=START_POWER)
    ap.add_argument("--end", type=int, default=END_POWER)
    ap.add_argument("--runs", type=int, default=NUM_RUNS)
    ap.add_argument("--json", action="store_true", help="also print the results as one JSON line")
    ap.add_argument("--csv", default=None, help="write the harness's CSV export (InputSize, MSM Func, Time (MS)) to this file")
    args = ap.parse_args()
    res = run(list(range(args.start, args.end + 1)), args.data, args.runs)
    if args.csv:
        with open(args.csv, "w") as f:
            f.write(to_csv(csv_rows(res)))
    if args.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
