#!/bin/bash
# where does the r2c line kernel's time go: shipped vs (1) no root-table loads (2) no mirrored store (3) no split loop at all — timing only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="r2c_2p10_b262144 r2c_2p12_b65536 r2c_2p13_b32768 r2c_2p14_b16384 c2c_2p9_b524288 c2c_2p11_b131072 c2c_2p12_b65536 c2c_2p13_b32768"
for w in $W; do
  for v in lib e1 e2 e3; do
    L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
    case $w in c2c*) [ $v != lib ] && continue;; esac
    P="$P \"x${v}_$w|100|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""
  done
done
eval tools/gpu_steps.sh $P > gpurun_out/r2c_exp_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/r2c_exp_steps.log | grep -v "exit 0" | tail
for w in $W; do
  echo "== $w: $(for v in lib e1 e2 e3; do [ -f gpurun_out/x${v}_$w.log ] && echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/x${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"
done
