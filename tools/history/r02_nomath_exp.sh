#!/bin/bash
# timing-only: every kernel with its butterflies and complex products removed (all global / LDS traffic and barriers as shipped)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_2p20_b4096 c2c_1024_b65536 r2c_2p22_b1024 c2c_2p12_b131072 c2c_2p14_b32768 c2c_2p15_b16384 c2c_2p17_b4096 c2c_2p21_b2048 r2c_2p12_b262144 r2c_2p20_b4096"
for w in $W; do for v in lib e1; do
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"nm${v}_$w|120|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh $P > gpurun_out/nomath_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/nomath_steps.log | grep -v "exit 0" | tail -3
for w in $W; do echo "== $w: $(for v in lib e1; do echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/nm${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"; done
