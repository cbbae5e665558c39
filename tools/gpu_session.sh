#!/bin/bash
# Runs a list of GPU steps on the gpurun box; ordinary failures (rc != 124/137) do not stop the
# session, a timeout / kill does (no further GPU step after a hang).
# usage: tools/gpu_session.sh "name|timeout_s|command" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== $name (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== $name rc=$rc in $(( $(date +%s) - start ))s"
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping the session"; exit $rc; fi
done
exit 0
