#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab/libmi355fft.so
tools/gpu_steps.sh \
  "gpus2|400|MI355FFT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --workload c2c_2p20_b512 --steps 5 --warmup 1" \
  "gpus2_lines|300|MI355FFT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --workload c2c_1024_b65536 --steps 20 --warmup 3" \
  "r2c22_ship|200|python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "r2c22_postnt|200|MI355FFT_LIB=$AB python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "c2r22_ship|200|python3 bench.py --workload c2r_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "c2r22_postnt|200|MI355FFT_LIB=$AB python3 bench.py --workload c2r_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "r2c_n3000|120|python3 bench.py --workload r2c_n3000_b400000 --steps 10 --warmup 2 --no-cpu-baseline" \
  "r2c_n3000_postnt|120|MI355FFT_LIB=$AB python3 bench.py --workload r2c_n3000_b400000 --steps 10 --warmup 2 --no-cpu-baseline" \
  "smoke|300|python3 -c 'import __graft_entry__ as g; g.smoke()'" \
  "headline_full|300|python3 bench.py" > gpurun_out/misc9_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/misc9_steps.log | grep -v "exit 0" | tail
for f in gpus2 gpus2_lines r2c22_ship r2c22_postnt c2r22_ship c2r22_postnt r2c_n3000 r2c_n3000_postnt; do echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"n_gpus": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"ranks_seen": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"collective_backend": "[^"]*"' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"; done
tail -3 gpurun_out/smoke.log
grep -o '"cpu_baseline": {[^}]*}' gpurun_out/headline_full.log | cut -c1-600
grep -o '"cpu_baseline_port": {[^}]*}' gpurun_out/headline_full.log | cut -c1-400
