#!/bin/bash
# usage: tools/sweep.sh "<flag>" v1 v2 ...   -> one summary line per value of bench.py <flag> <value>
flag="$1"; shift
for v in "$@"; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-sizes --no-host-buffers $flag $v 2>/dev/null | tail -1 > /tmp/sweep.json
  python - "$flag" "$v" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json"))
print(sys.argv[1], sys.argv[2], "ms/step", round(d["ms_per_step"], 4), "acc_ms", round(d["roofline"]["kernel_ms"], 4),
      {k: round(v * 1000) for k, v in d["stage_ms_untimed_pass"].items()})
PY
done
