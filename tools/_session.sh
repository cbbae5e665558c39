#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/rehearse_point_shards.py 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_s10_rehearse20.txt
timeout -k 10 400 python tools/rehearse_point_shards.py 18 2>&1 | grep -v amdgpu.ids | grep "upload_threads\|submit ticket" | tee gpurun_out/r05_s10_rehearse18.txt
timeout -k 10 600 python -m pytest tests/test_gpu_tickets.py tests/test_gpu_bls377.py -x -q > gpurun_out/r05_s10_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r05_s10_tests.log
