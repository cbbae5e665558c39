#!/bin/bash
# round 5, session 7: verification at HEAD -- smoke, the whole GPU suite, ten minutes of soak, the default bench line,
# bench.py --gpus 2 with two ranks sharing the GPU (the guarded side legs of the N > 1 path)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r05_s7_smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r05_s7_smoke.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05_s7_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r05_s7_gpu.log
timeout -k 10 800 python tools/soak.py 600 5051 > gpurun_out/r05_s7_soak.txt 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/r05_s7_soak.txt
timeout -k 10 600 python bench.py > gpurun_out/r05_s7_bench.json 2> gpurun_out/r05_s7_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_s7_bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "latency_ms", "host_buffers_ms", "host_buffers_in_flight_ms", "parity")})
print({k: (round(v["value"], 1), v["parity"][:9], round(v["roofline"]["binding_roofline"]["frac_at_measured_clock"], 3)) for k, v in d["configs"].items()})
PY
TE_BENCH_SHARE_GPU=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 40 --warmup 3 > gpurun_out/r05_s7_gpus2.log 2> gpurun_out/r05_s7_gpus2.err; echo "gpus2 rc=$?"
python - <<'PY'
import json
l = [x for x in open("gpurun_out/r05_s7_gpus2.log") if x.startswith("{")]
d = json.loads(l[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "input_distribution_ms", "input_distribution_parity", "input_distribution_error", "host_buffers_ms", "host_buffers_ms_median", "host_buffers_in_flight_ms", "host_buffers_error", "parity")})
PY
