#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/ab_env.sh "c2c_2p20_b4096" "MI355FFT_XCD_HX=2;MI355FFT_LIB=$L/lib_ek_pf0/libmi355fft.so;MI355FFT_XCD_HX=2;MI355FFT_LIB=$L/lib_ek_pf0/libmi355fft.so" 2>&1 | tee gpurun_out/r03_rt32_pf.log
