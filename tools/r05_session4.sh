#!/bin/bash
# round 5, session 4: default build against the fenced twin with four MSMs in flight; the staging copy (streaming stores against
# memcpy); the differential soak with the new ticket modes; the whole GPU suite
set -o pipefail
mkdir -p gpurun_out
B="python bench.py --no-cpu-baseline --no-sizes --no-configs --no-host-buffers --steps 100 --repeats 5"
PKGDIR=webgpu-msm-twisted-edwards_amd
for round in 1 2 3; do
  for lib in libtemsm.so libtemsm_fenced.so; do
    TE_MSM_LIB=$PWD/$PKGDIR/$lib timeout -k 10 200 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib round $round: %.1f MSM/s  passes %s  latency %.3f ms  k_accumulate alone %.4f ms' % (d['value'], ' '.join('%.4f' % x for x in d['passes_ms_per_step']), d['latency_ms'], d['roofline']['kernel_ms']))" | tee -a gpurun_out/r05_s4_fenced_ab.txt
  done
done
for mode in streaming memcpy; do
  echo "== staging copy: $mode" | tee -a gpurun_out/r05_s4_staging_copy.txt
  TE_MSM_STAGING_COPY=$mode TE_H2D_ONLY_STAGED=1 timeout -k 10 300 python tools/h2d_fresh_buffers.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-140 | tee -a gpurun_out/r05_s4_staging_copy.txt
done
timeout -k 10 400 python tools/soak.py 240 20251005 > gpurun_out/r05_s4_soak.txt 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r05_s4_soak.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_s4_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -4 gpurun_out/r05_s4_gpu.log
