#!/bin/bash
# same-box A/B of the fused routes against the two-launch routes over sizes: tools/ab_routes.sh "<workload> ..."
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_routes; mkdir -p $O
for w in $1; do
  for f in 1 0; do
    MI355FFT_XCD_FUSED=$f timeout -k 10 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 10 > $O/${w}_f$f.json 2> $O/${w}_f$f.err
    python - "$w fused=$f" "$O/${w}_f$f.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms", d["config"]["route"], flush=True)
PY
  done
done
