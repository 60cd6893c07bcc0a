#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/gpu_steps.sh "rt_tests3|900|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'r2c_four_step or cfg5 or two_workgroups or two_pass or fused_many'" > gpurun_out/r03_rt_steps3.log 2>&1
tail -3 gpurun_out/rt_tests3.log
tools/ab_env.sh "r2c_2p22_b1024" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_enosh/libmi355fft.so;MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_enosh/libmi355fft.so" 2>&1 | tee gpurun_out/r03_rt_ab8.log
