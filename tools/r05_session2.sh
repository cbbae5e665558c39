#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_tickets.py -x -q -k "window_count" > gpurun_out/r05_s2_test.log 2>&1; echo "test rc=$?"
tail -3 gpurun_out/r05_s2_test.log
timeout -k 10 400 python tools/h2d_fresh_buffers.py 20 > gpurun_out/r05_s2_fresh20.txt 2>&1; echo "fresh rc=$?"
cat gpurun_out/r05_s2_fresh20.txt
timeout -k 10 200 python tools/trace_point_shards.py 8 20 40 > gpurun_out/r05_s2_trace8.txt 2> gpurun_out/r05_s2_trace8_stamps.txt; echo "trace rc=$?"
tr '\n' ' ' < gpurun_out/r05_s2_trace8.txt
