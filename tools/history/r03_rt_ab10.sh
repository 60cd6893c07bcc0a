#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "c2r_2p22_b1024" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_ec2rni0/libmi355fft.so;MI355FFT_XCD_RT=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_LIB=$L/lib_ec2rni0/libmi355fft.so MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_rt_ab10.log
