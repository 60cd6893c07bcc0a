#!/bin/bash
# same-box A/B of the fused kernel's group split and nontemporal variant on the headline bench
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_split; mkdir -p $O
run() { # name, split, lib
  MI355FFT_XCD_SPLIT=$2 MI355FFT_LIB=$3 timeout -k 10 200 python $R/bench.py --no-cpu-baseline --steps 10 > $O/$1.json 2> $O/$1.err
  python - "$1" "$O/$1.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], flush=True)
PY
}
MAIN=$R/webgpu-fft_amd/lib/libmi355fft.so; NT1=$R/webgpu-fft_amd/lib/variants/libmi355fft.so
for rep in 1 2; do
  run s2_$rep 2 $MAIN; run s3_$rep 3 $MAIN; run s2nt_$rep 2 $NT1; run s4nt_$rep 4 $NT1
done
