#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "mixed_test|400|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'mixed'" > gpurun_out/r03_mixed_steps.log 2>&1
tail -3 gpurun_out/mixed_test.log
tools/ab_env.sh "c2c_n2187_b200000 c2c_n3000_b150000 c2c_n1500_b300000" "MI355FFT_MIXED_CT=2;MI355FFT_MIXED_CT=1" 2>&1 | tee gpurun_out/r03_mixed_ab.log
