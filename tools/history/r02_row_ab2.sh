#!/bin/bash
# ROW line kernels, occupancy variants: lib = shipped (2048 as 16*16*8 T=2); e1 = last-stage table of 4096 left in global (4 workgroups per CU);
# e2 = the same for 2048 and 4096; e3 = 1024 as 16*16*4 T=4 (3 workgroups per CU, two exchanges)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_1024_b65536 r2c_2p11_b131072 dct2_2p11_b131072 fftconv_2p10_b262144 c2c_2p11_b131072 r2c_2p12_b65536 dct2_2p12_b65536 c2c_2p12_b65536 r2c_2p13_b32768 c2r_2p13_b32768 dct2_2p13_b32768"
for w in $W; do for v in lib e1 e2 e3; do
  case "$v:$w" in e3:*2p12*|e3:*2p13*|e1:*1024*|e1:*2p10*|e1:*2p11*|e2:*1024*|e2:*2p10*|e2:r2c_2p11*|e2:dct2_2p11*) continue;; esac
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"q${v}_$w|60|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh $P > gpurun_out/row_ab2_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/row_ab2_steps.log | grep -v "exit 0" | tail
for w in $W; do echo "== $w: $(for v in lib e1 e2 e3; do [ -f gpurun_out/q${v}_$w.log ] && echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/q${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"; done
