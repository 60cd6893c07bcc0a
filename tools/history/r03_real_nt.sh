#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "r2c_2p20_b4096 c2r_2p20_b4096 r2c_2p19_b8192 c2r_2p19_b8192 r2c_2p18_b16384 c2r_2p18_b16384 r2c_2p17_b32768" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_erealnt0/libmi355fft.so" 2>&1 | tee gpurun_out/r03_real_nt.log
