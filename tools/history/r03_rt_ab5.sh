#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "r2c_2p22_b1024" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_eskl/libmi355fft.so;MI355FFT_LIB=$L/lib_ealn/libmi355fft.so;MI355FFT_LIB=$L/lib_eboth/libmi355fft.so" 2>&1 | tee gpurun_out/r03_rt_ab5.log
