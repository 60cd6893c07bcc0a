#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/gpu_steps.sh "lanes_test|300|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'whdcn or lane'" > gpurun_out/r03_lanes_steps.log 2>&1
tail -3 gpurun_out/lanes_test.log
tools/ab_env.sh "c2r_2p14_b65536 c2r_2p15_b32768" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_epv1_1/libmi355fft.so;MI355FFT_LIB=$L/lib_epv2_4/libmi355fft.so;MI355FFT_LIB=$L/lib_epv4_8/libmi355fft.so" 2>&1 | tee gpurun_out/r03_c2r_pv.log
