#!/bin/bash
# Runs a list of GPU steps in order on the gpurun box; each step is "name|timeout_seconds|command".
# Stops at the first step that was killed / timed out / crashed (never starts another GPU step after that),
# but carries on after an ordinary non-zero exit (e.g. a failing test) so later measurements still happen.
# Output of every step goes to gpurun_out/<name>.log.
mkdir -p gpurun_out
overall=0
for step in "$@"; do
  name="${step%%|*}"; rest="${step#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -o pipefail -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s"
  tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then
    echo "=== [$name] was killed or crashed: not starting further GPU steps"
    exit $rc
  fi
  [ $rc -ne 0 ] && overall=$rc
done
exit $overall
