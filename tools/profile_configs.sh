#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: own runs) over the SIDE configs of the driver's line -- BASELINE config 2
# (unsigned windows), config 5 (BLS12-377 G1) and the harness mode (one fixed point) -- so that configs.*.roofline.traffic is a
# measured figure too.  Run on the GPU box via gpurun AFTER tools/profile_bench.sh (it merges into that run's traffic.json):
#   gpurun_out/prof/traffic.json gains "configs": {name: {"workload", "kernels": {k_accumulate: {...}, ...}}}
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/prof"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
COMMON="--steps 6 --warmup 1 --repeats 1 --no-cpu-baseline --no-sizes --no-host-buffers --no-configs"
for spec in "unsigned|--digits unsigned" "bls12_377|--curve bls12-377" "harness_fixed_point|--points fixed" "witness_scalars|--scalars mixed"; do
  name="${spec%%|*}"; flags="${spec#*|}"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d="$OUT/cfg_${name}_${ctr}"; rm -rf "$d"
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$d" -- python3 $REPO/bench.py $COMMON $flags > "$d.log" 2>&1 || { echo "$name $ctr pass failed"; tail -5 "$d.log"; exit 1; }
    echo "$name $ctr done"
  done
done
python3 "$REPO/tools/merge_config_traffic.py" "$OUT"
