#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=""
for nb in 384:1000000 1920:200000 2000:200000 3072:125000 720:500000 240:1600000; do
  n=${nb%%:*}; b=${nb##*:}
  S="$S \"mct_$n|100|python3 bench.py --workload c2c_n${n}_b${b} --steps 10 --warmup 2 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh $S \
  "'mixed_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"mixed or bluestein or dct or dst or trig or strided or lane or whdcn or fftconv\"'" \
  "'dct2_2p20|120|python3 bench.py --workload dct2_2p20_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'dct4_2p20|120|python3 bench.py --workload dct4_2p20_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'dct2_4096|120|python3 bench.py --workload dct2_2p12_b65536 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'dct2_2d|120|python3 bench.py --workload dct2_s1024x1024_b256 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'dct2_1000|120|python3 bench.py --workload dct2_n1000_b400000 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'view2d_staged|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload c2c_s1024x1024_b256_view --steps 20 --warmup 3 --no-cpu-baseline'" \
  "'r2c22|200|python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'cfg2|120|python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'headline|200|python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline'" > gpurun_out/misc8_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/misc8_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/misc8_steps.log | tail -3
for f in mct_384 mct_1920 mct_2000 mct_3072 mct_720 mct_240 dct2_2p20 dct4_2p20 dct2_4096 dct2_2d dct2_1000 view2d_staged r2c22 cfg2 headline; do echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"; done
