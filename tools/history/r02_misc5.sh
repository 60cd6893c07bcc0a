#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "conv_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'fftconv or conv'" \
  "conv_1024|120|python3 bench.py --workload fftconv_2p10_b65536 --steps 20 --warmup 3 --no-cpu-baseline" \
  "conv_8192|120|python3 bench.py --workload fftconv_2p13_b8192 --steps 20 --warmup 3 --no-cpu-baseline" \
  "conv_256|120|python3 bench.py --workload fftconv_2p8_b262144 --steps 20 --warmup 3 --no-cpu-baseline" \
  "conv_2p20|120|python3 bench.py --workload fftconv_2p20_b256 --steps 10 --warmup 2 --no-cpu-baseline" > gpurun_out/misc5_steps.log 2>&1
grep -E "^=== .*exit|passed|failed" gpurun_out/misc5_steps.log | tail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_conv -- python3 $R/bench.py --workload fftconv_2p10_b65536 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_conv.log 2>&1
cd $R
head -8 $(ls gpurun_out/prof_conv/*/*kernel_stats.csv | head -1) | cut -c1-200
for f in conv_1024 conv_8192 conv_256 conv_2p20; do
  echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"
done
