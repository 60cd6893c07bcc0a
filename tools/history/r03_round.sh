#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "copy_test|200|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'copy_buffer'" \
  "headline_full|400|python3 bench.py" \
  "cfg5_line|200|python3 bench.py --workload r2c_2p22_b1024 --no-cpu-baseline" > gpurun_out/r03_round_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/r03_round_steps.log
tail -2 gpurun_out/copy_test.log
grep -o '"value": [0-9.]*' gpurun_out/headline_full.log; grep -o '"attainable": {[^}]*}' gpurun_out/headline_full.log | cut -c1-900
grep -o '"value": [0-9.]*' gpurun_out/cfg5_line.log; grep -o '"attainable": {[^}]*}' gpurun_out/cfg5_line.log | cut -c1-500
tools/r03_size_sweep.sh > gpurun_out/r03_size_sweep.log 2>&1
tail -50 gpurun_out/r03_size_sweep.log
