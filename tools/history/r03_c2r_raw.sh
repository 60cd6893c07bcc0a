#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "c2r_2p14_b65536 c2r_2p15_b32768" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_ec2rraw/libmi355fft.so;MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_ec2rraw/libmi355fft.so" 2>&1 | tee gpurun_out/r03_c2r_raw.log
tools/ab_env.sh "r2c_2p14_b65536 r2c_2p15_b32768" "MI355FFT_XCD_RT=1" 2>&1 | tee -a gpurun_out/r03_c2r_raw.log
