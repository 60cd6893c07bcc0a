#!/bin/bash
# r03: fftconv of 2^20-point lines: one-launch pipeline (fft_xcd_conv1m_kernel) vs forward + pointwise + inverse launches, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh "conv_parity|600|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'fftconv'" > gpurun_out/r03_conv_steps.log 2>&1
tail -3 gpurun_out/conv_parity.log
tools/ab_env.sh "fftconv_2p20_b512 fftconv_2p20_b2048" "MI355FFT_CONV_PIPELINE=0;MI355FFT_CONV_PIPELINE=1;MI355FFT_CONV_PIPELINE=1 MI355FFT_XCD_SLOTS=1;MI355FFT_CONV_PIPELINE=1 MI355FFT_XCD_SLOTS=1 MI355FFT_XCD_SPLIT=2;MI355FFT_CONV_PIPELINE=1 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=2" 2>&1 | tee gpurun_out/r03_conv_ab.log
