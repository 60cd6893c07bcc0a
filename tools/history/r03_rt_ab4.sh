#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "r2c_2p22_b1024 c2c_2p22_b512" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_epa/libmi355fft.so;MI355FFT_LIB=$L/lib_epb/libmi355fft.so;MI355FFT_LIB=$L/lib_enomath/libmi355fft.so;MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_rt_ab4.log
