#!/bin/bash
# r03: LDS layout of the three-stage ROW shapes with a 16-point first stage (2048, 4096): padded (shipped) vs swizzled (MI355_LDS_SWZ16=1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/webgpu-fft_amd
W="c2c_2p11_b262144 c2c_2p12_b131072 r2c_2p12_b262144 r2c_2p13_b131072 c2r_2p12_b262144 c2r_2p13_b131072 dct2_2p12_b65536"
tools/ab_env.sh "$W" "MI355FFT_XCD_RT=1;MI355FFT_LIB=$L/lib_eswz/libmi355fft.so" 2>&1 | tee gpurun_out/r03_swz_ab.log
echo "## bank conflicts, padded" | tee -a gpurun_out/r03_swz_ab.log
PMC_LDS_WORKLOADS="c2c_2p11_b262144 c2c_2p12_b131072 r2c_2p13_b131072 c2r_2p12_b262144" tools/pmc_lds.sh 2>&1 | tee -a gpurun_out/r03_swz_ab.log
echo "## bank conflicts, swizzled" | tee -a gpurun_out/r03_swz_ab.log
MI355FFT_LIB=$L/lib_eswz/libmi355fft.so PMC_LDS_WORKLOADS="c2c_2p11_b262144 c2c_2p12_b131072 r2c_2p13_b131072 c2r_2p12_b262144" tools/pmc_lds.sh 2>&1 | tee -a gpurun_out/r03_swz_ab.log
