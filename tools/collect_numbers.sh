#!/bin/bash
# Every bench configuration DESIGN.md section 6 quotes, on ONE box (the pool's MI355X differ by a few per cent), into
# gpurun_out/numbers/: bench_<name>.json (the bench line), lines.json (all of them), rehearsal.txt (per-rank steps).
# usage (GPU box): bash tools/collect_numbers.sh
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
OUT=gpurun_out/numbers; mkdir -p $OUT
line() {  # name, bench args...
  local name=$1; shift
  timeout -k 10 400 python bench.py "$@" > $OUT/_log.txt 2> $OUT/_err.txt || { echo "FAILED $name"; tail -5 $OUT/_err.txt; return 1; }
  grep '^{' $OUT/_log.txt | tail -1 > $OUT/bench_$name.json
  python - $name $OUT/bench_$name.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print("%-14s %8.1f MSM/s  %.4f ms/step  latency %.3f ms  host-buffers %s" % (sys.argv[1], j["value"], j["ms_per_step"], j["latency_ms"], j.get("host_buffers_ms")))
PY
}
line default --steps 200 --warmup 10 || exit 1
line nopipe --steps 60 --warmup 5 --no-pipeline --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
line unsigned --steps 100 --warmup 5 --digits unsigned --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
line equal --steps 100 --warmup 5 --scalars equal --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
line small --steps 100 --warmup 5 --scalars small --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
line fixed --steps 100 --warmup 5 --points fixed --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
line bls --steps 60 --warmup 5 --curve bls12-377 --no-sizes --no-cpu-baseline || exit 1
line bls_eq --steps 60 --warmup 5 --curve bls12-377 --scalars equal --no-sizes --no-host-buffers --no-cpu-baseline || exit 1
: > $OUT/rehearsal.txt
for D in 1 2 4 8; do
  for b in 1 0; do
    TE_BENCH_FORCE_DIST=1 TE_BENCH_REHEARSE_WORLD=$D timeout -k 10 300 python bench.py --steps 256 --warmup 3 --no-cpu-baseline --no-sizes --no-host-buffers --batch $b > $OUT/_log.txt 2> $OUT/_err.txt || { echo "FAILED D=$D"; exit 1; }
    python - $D $b $OUT/_log.txt <<'PY' | tee -a $OUT/rehearsal.txt
import json, sys
for l in open(sys.argv[3]):
    if l.startswith("{"):
        j = json.loads(l)
        dl = j.get("batch_distinct_bases")
        print("D=%s --batch %s  %.4f ms per MSM and rank (%s)  %7.1f MSM/s upper bound  latency of one step %.3f ms  [%s]" % (
            sys.argv[1], sys.argv[2], j["ms_per_step"], ("shared bases; distinct bases %.4f" % dl["ms_per_step"]) if dl else "one MSM per sequence",
            j["value"], j["latency_ms"], j["mode"]))
PY
  done
done
python - $OUT <<'PY'
import glob, json, os, sys
d = {os.path.basename(f)[6:-5]: json.load(open(f)) for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json")))}
json.dump(d, open(os.path.join(sys.argv[1], "lines.json"), "w"), indent=1)
PY
