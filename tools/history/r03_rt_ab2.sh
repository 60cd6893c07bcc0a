#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "rt_tests2|900|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'r2c_four_step or cfg5'" > gpurun_out/r03_rt_steps2.log 2>&1
tail -5 gpurun_out/rt_tests2.log
tools/ab_env.sh "r2c_2p22_b1024" "MI355FFT_XCD_RT=0;MI355FFT_XCD_RT=1;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_rt_ab2.log
