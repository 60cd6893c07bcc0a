#!/bin/bash
# r03: register-tile passes for 2048-point sides (kern_regtile.hpp) vs the routes they replace, same box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "rt_tests|900|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'two_pass or four_step or fused_many or cfg5'" > gpurun_out/r03_rt_steps.log 2>&1
tail -5 gpurun_out/rt_tests.log
tools/ab_env.sh "c2c_2p21_b1024 c2c_2p22_b512 r2c_2p22_b1024 c2r_2p22_b1024 c2c_2p20_b4096" "MI355FFT_XCD_RT=0;MI355FFT_XCD_RT=1" 2>&1 | tee gpurun_out/r03_rt_ab.log
