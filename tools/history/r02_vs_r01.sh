#!/bin/bash
# same-box regression check: this round's library vs the round-1 library (built from commit 1420184) on sizes whose sweep figures moved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_2p12_b131072 c2c_2p13_b65536 c2c_2p14_b32768 c2c_2p15_b16384 c2c_2p16_b8192 c2c_2p17_b4096 c2c_2p18_b2048 r2c_2p16_b16384 r2c_2p17_b8192 c2c_2p20_b512"
for w in $W; do for v in lib r01; do
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"o${v}_$w|60|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh $P > gpurun_out/vsr01_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/vsr01_steps.log | grep -v "exit 0" | tail -3
for w in $W; do echo "== $w: $(for v in lib r01; do echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/o${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' gpurun_out/o${v}_$w.log | head -1 | cut -d'"' -f4)] "; done)"; done
