#!/bin/bash
# split sweep for given workloads: tools/ab_split2.sh "<workload> ..." "<splits>"
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_split2; mkdir -p $O
for w in $1; do
  for s in $2; do
    MI355FFT_XCD_SPLIT=$s timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 10 > $O/${w}_s$s.json 2> $O/${w}_s$s.err
    python - "$w split=$s" "$O/${w}_s$s.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms", d["config"]["route"], flush=True)
PY
  done
done
