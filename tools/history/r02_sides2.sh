#!/bin/bash
# r2c / c2r / fftconv sides fused into the line kernels (round 2, second batch) vs the staging route
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
for w in r2c_s4096_b65536_view r2c_s1024x1024_b128_view fftconvlin_n1000k25_b262144 fftconvlin_n4000k97_b65536; do
  P="$P \"s2f_$w|120|python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""
  P="$P \"s2s_$w|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh \
  "'sides2_tests|600|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"ioview or zeropad or strided or whdcn or lane or fftconv or r2c or c2r\"'" \
  $P > gpurun_out/sides2_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/sides2_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/sides2_steps.log | tail -4
for f in gpurun_out/s2[fs]_*.log; do w=$(basename $f .log); echo "== $w: $(grep -o '"value": [0-9.]*' $f | head -1 | cut -d' ' -f2 | cut -c1-6) $(grep -o '"launches_per_step": [0-9]*' $f | head -1) [$(grep -o '"route": "[^"]*"' $f | head -1 | cut -d'"' -f4)]"; done
