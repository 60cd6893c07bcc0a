#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "c2r21_tests|900|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'c2r_four_step or cfg5'" > gpurun_out/r03_c2r21_steps.log 2>&1
tail -3 gpurun_out/c2r21_tests.log
tools/ab_env.sh "c2r_2p21_b2048" "MI355FFT_XCD_RT=0;MI355FFT_XCD_RT=1;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=2" 2>&1 | tee gpurun_out/r03_c2r21_ab.log
