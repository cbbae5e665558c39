#!/bin/bash
# round 5, session 3: new tests (host_staging, hand-offs under both builds, load_host over RCCL), the staged upload path measured
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_handoffs.py tests/test_gpu_stage_models.py -x -q -s > gpurun_out/r05_s3_tests.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r05_s3_tests.log
timeout -k 10 500 python tools/h2d_fresh_buffers.py 20 > gpurun_out/r05_s3_fresh20.txt 2>&1; echo "fresh rc=$?"
cat gpurun_out/r05_s3_fresh20.txt | cut -c1-330
