#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "o21_tests|600|MI355FFT_XCD_RT=2 python3 -m pytest tests/test_gpu_parity.py -x -q -k 'fused_many or register_tile_sizes_full'" > gpurun_out/r03_o21_steps.log 2>&1
tail -3 gpurun_out/o21_tests.log
tools/ab_env.sh "c2c_2p21_b1024" "MI355FFT_XCD_RT=1;MI355FFT_XCD_RT=2;MI355FFT_XCD_RT=2 MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2;MI355FFT_XCD_RT=2 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_RT=1" 2>&1 | tee gpurun_out/r03_2p21_orient.log
