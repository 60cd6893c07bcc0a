#!/bin/bash
# same-box A/B of the batched split loads in the r2c / c2r line kernels: batch 1 (lib_ab) vs 4 (lib, shipped) vs 8 (lib_ab2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
for w in r2c_2p10_b262144 r2c_2p12_b65536 r2c_2p13_b32768 r2c_2p14_b16384 r2c_2p15_b8192 c2r_2p10_b262144 c2r_2p12_b65536 c2r_2p13_b32768 c2r_2p14_b16384 dct2_2p12_b65536 dct3_2p12_b65536; do
  for v in ab:1 lib:4 ab2:8; do
    d=${v%%:*}; n=${v##*:}
    L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$d/libmi355fft.so; [ $d = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
    P="$P \"b${n}_$w|100|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""
  done
done
eval tools/gpu_steps.sh $P > gpurun_out/batch_ab_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/batch_ab_steps.log | grep -v "exit 0" | tail
for w in r2c_2p10_b262144 r2c_2p12_b65536 r2c_2p13_b32768 r2c_2p14_b16384 r2c_2p15_b8192 c2r_2p10_b262144 c2r_2p12_b65536 c2r_2p13_b32768 c2r_2p14_b16384 dct2_2p12_b65536 dct3_2p12_b65536; do
  echo "== $w: batch1 $(grep -o '"value": [0-9.]*' gpurun_out/b1_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) batch4 $(grep -o '"value": [0-9.]*' gpurun_out/b4_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) batch8 $(grep -o '"value": [0-9.]*' gpurun_out/b8_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' gpurun_out/b4_$w.log | head -1 | cut -d'"' -f4)]"
done
