#!/bin/bash
# ROW line kernel of 2048 points (c2c 2048, r2c / c2r / DCT 4096): 32*32*2 T=4 (shipped, one 256-thread workgroup per CU) vs T=8 / T=2 / 16*16*8 T=2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_2p11_b131072 r2c_2p12_b65536 c2r_2p12_b65536 dct2_2p12_b65536 dct3_2p12_b65536 fftconv_2p11_b131072"
for w in $W; do for v in lib e1 e2 e3; do
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"k${v}_$w|60|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh $P > gpurun_out/row2k_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/row2k_steps.log | grep -v "exit 0" | tail
for w in $W; do echo "== $w: $(for v in lib e1 e2 e3; do echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/k${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"; done
