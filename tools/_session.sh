#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for st in 4 8 12 16; do
  echo "== TE_MSM_STAGERS=$st" | tee -a gpurun_out/r05_s8_stagers.txt
  TE_MSM_STAGERS=$st TE_H2D_ONLY_STAGED=1 TE_H2D_D1_ONLY=1 timeout -k 10 200 python tools/h2d_fresh_buffers.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tee -a gpurun_out/r05_s8_stagers.txt
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_s8_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r05_s8_gpu.log
