#!/bin/bash
# ROW line kernel of 1024 points (config 2): 32*32 T=8 (shipped) vs 16*16*4 T=8 (512 threads) / 16*16*4 T=4 / 32*32 T=16 (512 threads)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_1024_b65536 c2c_2p10_b524288 r2c_2p11_b524288 c2r_2p11_b524288 dct2_2p11_b131072 fftconv_2p10_b262144"
for r in 1 2; do for w in $W; do for v in lib e1 e2 e3; do
  [ $r = 2 ] && [ $w != c2c_1024_b65536 ] && continue
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"y${r}${v}_$w|60|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""; done; done; done
eval tools/gpu_steps.sh $P > gpurun_out/row1k_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/row1k_steps.log | grep -v "exit 0" | tail
for w in $W; do echo "== $w: $(for r in 1 2; do for v in lib e1 e2 e3; do [ -f gpurun_out/y${r}${v}_$w.log ] && echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/y${r}${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done; done)"; done
