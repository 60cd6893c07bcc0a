#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "r2c_2p22_b1024" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_easc/libmi355fft.so;MI355FFT_LIB=$L/lib_epb/libmi355fft.so;MI355FFT_LIB=$L/lib_epbasc/libmi355fft.so;MI355FFT_LIB=$L/lib_epbnm/libmi355fft.so" 2>&1 | tee gpurun_out/r03_rt_ab6.log
