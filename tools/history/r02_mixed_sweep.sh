#!/bin/bash
# compile-time mixed-radix plans: one probe per length, ~2^28 points each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="96 105 160 192 240 320 360 384 480 500 640 720 768 960 1000 1001 1280 1500 1536 1920 2000 2187 2560 3000 3072"
for n in $W; do b=$(( (1<<28) / n )); P="$P \"m_$n|60|python3 bench.py --workload c2c_n${n}_b$b --steps 10 --warmup 2 --no-cpu-baseline\""; done
eval tools/gpu_steps.sh $P > gpurun_out/mixed_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/mixed_steps.log | grep -v "exit 0" | tail
for n in $W; do echo "== N=$n: $(grep -o '"value": [0-9.]*' gpurun_out/m_$n.log | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' gpurun_out/m_$n.log | head -1 | cut -d'"' -f4)]"; done
