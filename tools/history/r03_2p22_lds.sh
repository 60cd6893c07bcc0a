#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "lds22_tests|600|MI355FFT_XCD_RT=0 python3 -m pytest tests/test_gpu_parity.py -x -q -k 'register_tile_sizes_full and 22'" > gpurun_out/r03_lds22_steps.log 2>&1
tail -3 gpurun_out/lds22_tests.log
tools/ab_env.sh "c2c_2p22_b512" "MI355FFT_XCD_RT=1;MI355FFT_XCD_RT=0;MI355FFT_XCD_RT=0 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_RT=0 MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_RT=1" 2>&1 | tee gpurun_out/r03_2p22_lds.log
