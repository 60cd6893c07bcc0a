#!/bin/bash
# r03: headline size on 32-line register tiles (256-byte segments, MI355FFT_XCD_HX=2) vs the shipped LDS-resident fused kernel, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "rt32_parity|300|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'two_workgroups'" > gpurun_out/r03_rt32_steps.log 2>&1
tail -3 gpurun_out/rt32_parity.log
tools/ab_env.sh "c2c_2p20_b4096" "MI355FFT_XCD_HX=0;MI355FFT_XCD_HX=2;MI355FFT_XCD_HX=2 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_HX=2 MI355FFT_XCD_SPLIT=1;MI355FFT_XCD_HX=0;MI355FFT_XCD_HX=2" 2>&1 | tee gpurun_out/r03_rt32_ab.log
