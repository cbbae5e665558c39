#!/bin/bash
# A/B of one environment switch on the single-MSM timeline at 2^20 (AB_ARGS="--log2n 16 --window-bits 0": other bench arguments):
#   tools/ab_timeline.sh TAG "ENV=VAL" ["ENV=VAL" ...]
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
TAG=$1; shift
for kv in "$@"; do
  name=$(echo "$kv" | sed 's:.*/::' | tr '= .' '___')
  env $kv bash tools/profile_trace.sh --no-pipeline --no-sizes --no-host-buffers --no-configs --repeats 1 --steps 30 ${AB_ARGS:-} > gpurun_out/${TAG}_${name}_summary.txt 2>&1 || exit 1
  python3 tools/trace_one_msm.py gpurun_out/prof/trace k_digits > gpurun_out/${TAG}_${name}_timeline.txt
  echo "== $kv"; cat gpurun_out/${TAG}_${name}_timeline.txt
done
