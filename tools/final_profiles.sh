#!/bin/bash
# Round-end evidence on ONE box: kernel trace + PMC passes of the default bench command, the same trace with one MSM at a time
# (what roofline.kernel_ms must agree with), timelines of one MSM at 2^20 and 2^16, the overlap report.  -> gpurun_out/final/
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
F=gpurun_out/final; rm -rf $F; mkdir -p $F
export TE_COMMIT=${TE_COMMIT:-unknown}
bash tools/profile_bench.sh > $F/profile_bench.txt 2>&1 || { echo "profile_bench failed"; tail -5 $F/profile_bench.txt; exit 1; }
bash tools/profile_configs.sh > $F/profile_configs.txt 2>&1 || { echo "profile_configs failed"; tail -5 $F/profile_configs.txt; exit 1; }
cp gpurun_out/prof/summary.txt $F/rocprofv3_summary.txt; cp gpurun_out/prof/traffic.json $F/pmc_traffic.json
cp $(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1) $F/rocprofv3_kernel_stats.csv
rm -rf gpurun_out/prof_pipe; cp -r gpurun_out/prof/trace gpurun_out/prof_pipe
bash tools/profile_trace.sh --no-pipeline --no-sizes --no-host-buffers --no-configs --steps 30 > $F/rocprofv3_summary_no_pipeline.txt 2>&1 || exit 1
python3 tools/trace_one_msm.py gpurun_out/prof/trace k_digits > $F/single_msm_timeline_2_20.txt
python3 tools/overlap_report.py gpurun_out/prof_pipe --alone gpurun_out/prof/trace > $F/pipelined_overlap.txt 2>&1
bash tools/profile_trace.sh --no-pipeline --no-sizes --no-host-buffers --no-configs --steps 30 --log2n 16 --window-bits 0 > /dev/null 2>&1 || exit 1
python3 tools/trace_one_msm.py gpurun_out/prof/trace k_digits > $F/single_msm_timeline_2_16.txt
ls -la $F
# the driver's line on the SAME box (roofline.kernel_ms there must agree with the one-MSM-at-a-time trace above)
cp $F/pmc_traffic.json profiles/pmc_traffic.json      # (the line reads the traffic figure from there: this run's own, same sources)
python3 bench.py --steps 20 --warmup 5 2> /dev/null | grep '^{' | tail -1 > $F/bench_default_line.json
python3 - $F <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/bench_default_line.json"))
print("bench line on this box: %.1f MSM/s, latency %.3f ms, k_accumulate alone %.4f ms (frac %.3f), traffic %s" % (
    d["value"], d["latency_ms"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"]))
PY
