#!/bin/bash
# mixed-radix: compile-time plans vs the runtime-plan kernel vs the stage route; ROW kernels with selective NT
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
S=""
for nb in 96:4000000 192:2000000 384:1000000 768:500000 1536:250000 3072:125000 160:2400000 320:1200000 640:600000 1280:300000 2560:150000 1000:400000 2000:200000 3000:130000 105:4000000 1001:400000 360:1000000 1920:200000 2187:180000 720:500000; do
  n=${nb%%:*}; b=${nb##*:}
  S="$S \"mct_$n|100|python3 bench.py --workload c2c_n${n}_b${b} --steps 10 --warmup 2 --no-cpu-baseline\""
  S="$S \"mrt_$n|100|MI355FFT_MIXED_CT=0 python3 bench.py --workload c2c_n${n}_b${b} --steps 10 --warmup 2 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh $S \
  "'mixed_tests|400|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"mixed or bluestein or lines\"'" \
  "'cfg2|120|python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'c2c_256|120|python3 bench.py --workload c2c_2p8_b262144 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'c2c_2048|120|python3 bench.py --workload c2c_2p11_b32768 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'c2c_64|120|python3 bench.py --workload c2c_2p6_b1048576 --steps 50 --warmup 5 --no-cpu-baseline'" > gpurun_out/misc7_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/misc7_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/misc7_steps.log | tail -3
for f in gpurun_out/mct_*.log; do n=$(basename $f .log); n=${n#mct_}; echo "== N=$n: ct $(grep -o '"value": [0-9.]*' $f | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' $f | head -1 | cut -d'"' -f4)] runtime-plan $(grep -o '"value": [0-9.]*' gpurun_out/mrt_$n.log | head -1 | cut -d' ' -f2 | cut -c1-6) [$(grep -o '"route": "[^"]*"' gpurun_out/mrt_$n.log | head -1 | cut -d'"' -f4)]"; done
for f in cfg2 c2c_256 c2c_2048 c2c_64; do echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1)"; done
