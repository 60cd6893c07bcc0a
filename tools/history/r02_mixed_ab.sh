#!/bin/bash
# compile-time mixed-radix plans: tile shape variants (lines per workgroup / threads), same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="240 384 1000 1001 1280 1500 1536 1920 2000 2187 2560 3000 3072"
for n in $W; do b=$(( (1<<28) / n )); for v in lib e1 e2; do
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"ma${v}_$n|60|MI355FFT_LIB=$L python3 bench.py --workload c2c_n${n}_b$b --steps 10 --warmup 2 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh $P > gpurun_out/mixed_ab_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/mixed_ab_steps.log | grep -v "exit 0" | tail
for n in $W; do echo "== N=$n: $(for v in lib e1 e2; do echo -n "$(grep -o '"value": [0-9.]*' gpurun_out/ma${v}_$n.log | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' gpurun_out/ma${v}_$n.log | head -1 | cut -d'"' -f4 | sed 's/mixed-ct//')] "; done)"; done
