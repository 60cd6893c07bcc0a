#!/bin/bash
# second pass on the fused real sides (8-byte pair fast paths) and the batched c2r pre-split loads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
for w in r2c_s4096_b65536_view r2c_s1024x1024_b128_view c2r_2p8_b1048576 c2r_2p10_b262144 c2r_2p11_b131072 c2r_2p12_b65536 c2r_2p13_b32768 c2r_2p14_b16384 dct3_2p12_b65536 dct3_2p14_b16384; do
  P="$P \"s3_$w|120|python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh \
  "'sides3_tests|600|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"ioview or zeropad or strided or whdcn or lane or fftconv or r2c or c2r or dct or dst\"'" \
  $P > gpurun_out/sides3_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/sides3_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/sides3_steps.log | tail -4
for f in gpurun_out/s3_*.log; do w=$(basename $f .log); echo "== $w: $(grep -o '"value": [0-9.]*' $f | head -1 | cut -d' ' -f2 | cut -c1-6) $(grep -o '"launches_per_step": [0-9]*' $f | head -1) [$(grep -o '"route": "[^"]*"' $f | head -1 | cut -d'"' -f4)]"; done
