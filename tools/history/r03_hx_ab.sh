#!/bin/bash
# r03: headline size on the two-workgroups-per-CU register-tile kernel vs the shipped LDS-resident fused kernel, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
tools/gpu_steps.sh "hx_parity|300|MI355FFT_XCD_HX=1 python3 -m pytest tests/test_gpu_parity.py -x -q -k 'two_pass and 20 or fused_many or cfg3'" > gpurun_out/r03_hx_steps.log 2>&1
tail -3 gpurun_out/hx_parity.log
tools/ab_env.sh "c2c_2p20_b4096" "MI355FFT_XCD_HX=0;MI355FFT_XCD_HX=1;MI355FFT_XCD_HX=1 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_HX=1 MI355FFT_XCD_SPLIT=4;MI355FFT_XCD_HX=1 MI355FFT_XCD_SPLIT=1;MI355FFT_XCD_HX=1 MI355FFT_XCD_SPLIT=8 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_HX=1 MI355FFT_LIB=$L/lib_ehx3/libmi355fft.so" 2>&1 | tee gpurun_out/r03_hx_ab.log
