#!/bin/bash
# round 5, first GPU session: the new ticket tests, the whole GPU suite, the default bench line, the rehearsal with stamps
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_tickets.py -x -q -s > gpurun_out/r05_s1_tickets.log 2>&1; echo "tickets rc=$?" | tee -a gpurun_out/r05_s1_summary.txt
tail -5 gpurun_out/r05_s1_tickets.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_s1_gpu.log 2>&1; echo "gpu suite rc=$?" | tee -a gpurun_out/r05_s1_summary.txt
tail -5 gpurun_out/r05_s1_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/r05_s1_bench.json 2> gpurun_out/r05_s1_bench.err; echo "bench rc=$?" | tee -a gpurun_out/r05_s1_summary.txt
timeout -k 10 300 python tools/rehearse_point_shards.py 20 > gpurun_out/r05_s1_rehearse20.txt 2>&1; echo "rehearse rc=$?" | tee -a gpurun_out/r05_s1_summary.txt
timeout -k 10 200 python tools/trace_point_shards.py 8 20 14 > gpurun_out/r05_s1_trace8.txt 2> gpurun_out/r05_s1_trace8_stamps.txt; echo "trace rc=$?" | tee -a gpurun_out/r05_s1_summary.txt
cat gpurun_out/r05_s1_trace8.txt
