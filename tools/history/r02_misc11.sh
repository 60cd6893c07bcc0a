#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=""
for w in dct2_2p12_b65536 dct3_2p12_b65536 dct2_2p10_b262144 dct2_2p14_b16384 dct3_2p14_b16384 dct2_2p8_b1048576; do
  S="$S \"f_$w|100|python3 bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh \
  "'trig_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"dct or dst or trig\"'" \
  $S > gpurun_out/misc11_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/misc11_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/misc11_steps.log | tail -4
for f in gpurun_out/f_d*.log; do w=$(basename $f .log); w=${w#f_}; echo "== $w: fused(table) $(grep -o '"value": [0-9.]*' $f | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' $f | head -1 | cut -d'"' -f4)]"; done
