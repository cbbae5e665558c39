#!/bin/bash
# round 5, session 6: the bounding experiment of the sort's bytes; bench.py --gpus 4 rehearsed with four ranks on the one GPU
# (gloo exchange; --inputs host: load_host timed, the 4-device context's tickets and lone call); the round's evidence set
set -o pipefail
mkdir -p gpurun_out
PKGDIR=$PWD/webgpu-msm-twisted-edwards_amd
for lib in libtemsm.so libtemsm_exp_nodigitstore.so libtemsm.so libtemsm_exp_nodigitstore.so; do
  TE_MSM_LIB=$PKGDIR/$lib timeout -k 10 200 python tools/exp_sort_bytes.py 2>/dev/null | tee -a gpurun_out/r05_s6_sort_bytes.txt
done
TE_BENCH_SHARE_GPU=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 64 --warmup 3 --inputs host > gpurun_out/r05_s6_gpus4.log 2> gpurun_out/r05_s6_gpus4.err; echo "gpus4 rc=$?"
grep '^{' gpurun_out/r05_s6_gpus4.log | tail -1 > gpurun_out/r05_s6_bench_gpus4_one_gpu_rehearsal.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_s6_bench_gpus4_one_gpu_rehearsal.json"))
print({k: d.get(k) for k in ("value", "ms_per_step", "input_distribution_ms", "host_buffers_ms", "host_buffers_ms_median", "host_buffers_in_flight_ms", "host_buffers_ms_one_device", "parity", "host_buffers_parity")})
PY
TE_COMMIT=$(cat .te_commit 2>/dev/null || echo unknown) timeout -k 10 1500 bash tools/final_profiles.sh > gpurun_out/r05_s6_final.txt 2>&1; echo "final rc=$?"; tail -15 gpurun_out/r05_s6_final.txt
