#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
E="MI355FFT_XCD_HX=2 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1"
tools/ab_env.sh "c2c_2p20_b4096" "$E;$E MI355FFT_LIB=$L/lib_ek_nto0/libmi355fft.so;$E MI355FFT_LIB=$L/lib_ek_nti0/libmi355fft.so;$E MI355FFT_LIB=$L/lib_ek_wnt/libmi355fft.so;$E MI355FFT_LIB=$L/lib_ek_nomath/libmi355fft.so;$E" 2>&1 | tee gpurun_out/r03_rt32_ab3.log
tools/ab_env.sh "c2c_2p19_b8192 c2c_2p18_b16384 c2c_2p17_b32768" "MI355FFT_XCD_HX=0;MI355FFT_XCD_SPLIT=8 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=16 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1" 2>&1 | tee -a gpurun_out/r03_rt32_ab3.log
tools/ab_env.sh "r2c_2p20_b4096 c2r_2p20_b4096 r2c_2p21_b2048" "MI355FFT_XCD_HX=0;MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=8 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1" 2>&1 | tee -a gpurun_out/r03_rt32_ab3.log
