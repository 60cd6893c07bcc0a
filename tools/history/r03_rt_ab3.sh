#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "r2c_2p22_b1024 c2c_2p22_b512" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_ent0/libmi355fft.so;MI355FFT_LIB=$L/lib_eni0/libmi355fft.so;MI355FFT_LIB=$L/lib_epf16/libmi355fft.so;MI355FFT_LIB=$L/lib_epf32/libmi355fft.so" 2>&1 | tee gpurun_out/r03_rt_ab3.log
tools/ab_env.sh "c2c_2p21_b1024" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_epf16/libmi355fft.so;MI355FFT_LIB=$L/lib_epf32/libmi355fft.so" 2>&1 | tee -a gpurun_out/r03_rt_ab3.log
