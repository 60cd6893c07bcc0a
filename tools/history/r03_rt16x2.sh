#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "rt16_parity|300|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'two_workgroups'" > gpurun_out/r03_rt16_steps.log 2>&1
tail -3 gpurun_out/rt16_parity.log
tools/ab_env.sh "c2c_2p20_b4096" "MI355FFT_XCD_HX=2;MI355FFT_XCD_HX=3;MI355FFT_XCD_HX=3 MI355FFT_XCD_SPLIT=8;MI355FFT_XCD_HX=3 MI355FFT_XCD_SPLIT=2;MI355FFT_XCD_HX=3 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=2;MI355FFT_XCD_HX=2" 2>&1 | tee gpurun_out/r03_rt16x2_ab.log
