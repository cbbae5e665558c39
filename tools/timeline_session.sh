#!/bin/bash
# single-MSM timelines at 2^20 and 2^16 from kernel traces of `bench.py --no-pipeline` -> gpurun_out/<tag>_timeline_2_20.txt / _2_16.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
TAG=${1:-tl}
bash tools/profile_trace.sh --no-pipeline --no-sizes --no-host-buffers --no-configs --repeats 1 --steps 30 > gpurun_out/${TAG}_summary_2_20.txt 2>&1 || exit 1
python3 tools/trace_one_msm.py gpurun_out/prof/trace k_digits > gpurun_out/${TAG}_timeline_2_20.txt
bash tools/profile_trace.sh --no-pipeline --no-sizes --no-host-buffers --no-configs --repeats 1 --steps 30 --log2n 16 --window-bits 0 > gpurun_out/${TAG}_summary_2_16.txt 2>&1 || exit 1
python3 tools/trace_one_msm.py gpurun_out/prof/trace k_digits > gpurun_out/${TAG}_timeline_2_16.txt
cat gpurun_out/${TAG}_timeline_2_20.txt gpurun_out/${TAG}_timeline_2_16.txt
