#!/bin/bash
# full GPU tier + the three BASELINE bench lines + a few grouping A/Bs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "gpu_suite|1100|python3 -m pytest tests -x -q -m gpu" \
  "smoke|300|python3 -c 'import __graft_entry__ as g; g.smoke()'" \
  "headline_full|400|python3 bench.py" > gpurun_out/r03_full_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/r03_full_steps.log
tail -4 gpurun_out/gpu_suite.log; tail -2 gpurun_out/smoke.log
grep -o '"value": [0-9.]*' gpurun_out/headline_full.log | head -1; grep -o '"frac": [0-9.]*' gpurun_out/headline_full.log | head -1
tools/ab_env.sh "c2c_2p22_b512 r2c_2p22_b1024 c2c_2p21_b1024" "MI355FFT_XCD_RT=1;MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_groups_ab.log
tools/ab_env.sh "c2c_s1024x1024_b4096 c2c_s512x512_b16384" "MI355FFT_XCD_RT=1;MI355FFT_XCD_SLOTS=2" 2>&1 | tee -a gpurun_out/r03_groups_ab.log
