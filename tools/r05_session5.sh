#!/bin/bash
# round 5, session 5: what the first calls of the eight-thread lone call wait for (hardware queues?), both libraries in the suite
set -o pipefail
mkdir -p gpurun_out
for q in 2 4 8; do
  for st in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$q host_staging=$st" | tee -a gpurun_out/r05_s5_hwq.txt
    GPU_MAX_HW_QUEUES=$q TE_MSM_HOST_STAGING=$st timeout -k 10 120 python tools/trace_point_shards.py 8 20 16 2> gpurun_out/r05_s5_stamps_q${q}_s${st}.txt | awk '{printf "%s ", $3} END {print ""}' | tee -a gpurun_out/r05_s5_hwq.txt
  done
done
echo "== default queues, D = 4" | tee -a gpurun_out/r05_s5_hwq.txt
timeout -k 10 120 python tools/trace_point_shards.py 4 20 16 2>/dev/null | awk '{printf "%s ", $3} END {print ""}' | tee -a gpurun_out/r05_s5_hwq.txt
timeout -k 10 900 python -m pytest tests/test_gpu_handoffs.py tests/test_gpu_tickets.py -x -q -s > gpurun_out/r05_s5_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r05_s5_tests.log
