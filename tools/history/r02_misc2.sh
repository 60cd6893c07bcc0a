#!/bin/bash
# A/B: ROW line kernels with the next tile's loads prefetched (lib) vs without (lib_ab), resident vs one-shot grids;
# r2c N=2^22 with the 16-byte split kernel; JS GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab/libmi355fft.so
S=""
for w in c2c_1024_b65536 c2c_2p8_b262144 c2c_2p9_b131072 c2c_2p11_b32768 c2c_2p6_b1048576 r2c_2p10_b131072; do
  S="$S \"pf_$w|120|python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  S="$S \"pf1s_$w|120|MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  S="$S \"nopf_$w|120|MI355FFT_LIB=$AB python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  S="$S \"nopf1s_$w|120|MI355FFT_LIB=$AB MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh $S \
  "'r2c22|200|python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'c2r22|200|python3 bench.py --workload c2r_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'js_gpu|400|cd webgpu-fft_amd/js && node test/gpu_parity.test.mjs'" \
  "'real_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"r2c or c2r or cfg5 or real\"'" > gpurun_out/misc2_steps.log 2>&1
tail -30 gpurun_out/misc2_steps.log
for f in gpurun_out/pf_*.log gpurun_out/pf1s_*.log gpurun_out/nopf_*.log gpurun_out/nopf1s_*.log gpurun_out/r2c22.log gpurun_out/c2r22.log; do
  echo "== $(basename $f .log): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"route": "[^"]*"' $f | head -1)"
done
