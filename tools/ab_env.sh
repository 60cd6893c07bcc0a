#!/bin/bash
# same-box sweep: tools/ab_env.sh "<workload> ..." "<ENV=.. settings separated by ;>"   e.g. "MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=4"
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_env; mkdir -p $O
IFS=';' read -ra SETS <<< "$2"
i=0
for w in $1; do
  for e in "${SETS[@]}"; do
    i=$((i+1))
    env $e timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 10 > $O/run$i.json 2> $O/run$i.err
    python - "$w [$e]" "$O/run$i.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms", d["config"]["route"], flush=True)
PY
  done
done
