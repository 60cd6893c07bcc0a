#!/bin/bash
# cfg2 family: nontemporal global accesses (lib_ab) and NT + T=4 tiles at N=1024 (lib_ab2) vs the shipped build, one-shot grids
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab/libmi355fft.so
AB2=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab2/libmi355fft.so
S=""
for w in c2c_1024_b65536 c2c_2p8_b262144 c2c_2p9_b131072 c2c_2p6_b1048576 c2c_2p11_b32768 c2c_2p12_b16384 r2c_2p10_b131072; do
  S="$S \"ship_$w|120|python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  S="$S \"nt_$w|120|MI355FFT_LIB=$AB python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh $S \
  "'ntt4_1024|120|MI355FFT_LIB=$AB2 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'ntt4_1024_2|120|MI355FFT_LIB=$AB2 MI355FFT_LINES_TILES_PER_WG=2 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'nt_headline|200|MI355FFT_LIB=$AB python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'nt_r2c22|200|MI355FFT_LIB=$AB python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'ship_r2c22|200|python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" > gpurun_out/misc6_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/misc6_steps.log | grep -v "exit 0" | tail
for f in gpurun_out/ship_*.log gpurun_out/nt_*.log gpurun_out/ntt4_*.log; do
  echo "== $(basename $f .log): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"route": "[^"]*"' $f | head -1)"
done
