#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "r2c21_tests|900|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'r2c_four_step or register_tile_sizes_full or pipeline_full_size'" > gpurun_out/r03_r2c21_steps.log 2>&1
tail -3 gpurun_out/r2c21_tests.log
tools/ab_env.sh "r2c_2p21_b2048" "MI355FFT_XCD_RT=0;MI355FFT_XCD_RT=1;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_r2c21_ab.log
