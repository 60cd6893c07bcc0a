#!/bin/bash
# round-2 power-of-two sweep on one box: c2c / r2c / c2r, 2^6 ... 2^22, 2^29 points (c2c) / 2^30 real points per step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""; W=""
for lg in 6 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22; do
  W="$W c2c_2p${lg}_b$(( 1 << (29 - lg) )) r2c_2p${lg}_b$(( 1 << (30 - lg) )) c2r_2p${lg}_b$(( 1 << (30 - lg) ))"
done
for w in $W; do P="$P \"z_$w|90|python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline\""; done
eval tools/gpu_steps.sh $P > gpurun_out/sweep_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/sweep_steps.log | grep -v "exit 0" | tail
for w in $W; do echo "$w $(grep -o '"value": [0-9.]*' gpurun_out/z_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/z_$w.log | head -1 | cut -d' ' -f2 | cut -c1-7) ms $(grep -o '"route": "[^"]*"' gpurun_out/z_$w.log | head -1 | cut -d'"' -f4)"; done
