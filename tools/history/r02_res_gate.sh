#!/bin/bash
# round-2 gate of the XCD-resident kernel: parity first, then the skeleton and the real kernel at each exchange depth,
# then the shipped fused route on the same box, then PMC passes of the best candidate
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --workload c2c_2p20_b4096 --steps 10 --warmup 2 --no-cpu-baseline"
tools/gpu_steps.sh \
  "res_parity|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k xcd_resident" \
  "res_skel_d4|200|MI355FFT_XCD_RES=2 MI355FFT_XCD_RES_DEPTH=4 $B" \
  "res_skel_d2|200|MI355FFT_XCD_RES=2 MI355FFT_XCD_RES_DEPTH=2 $B" \
  "res_skel_d1|200|MI355FFT_XCD_RES=2 MI355FFT_XCD_RES_DEPTH=1 $B" \
  "res_d4|200|MI355FFT_XCD_RES=1 MI355FFT_XCD_RES_DEPTH=4 $B" \
  "res_d2|200|MI355FFT_XCD_RES=1 MI355FFT_XCD_RES_DEPTH=2 $B" \
  "res_d1|200|MI355FFT_XCD_RES=1 MI355FFT_XCD_RES_DEPTH=1 $B" \
  "fused_ref|200|$B" \
  "cfg2_ref|200|python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline" \
  "cfg2_oneshot1|200|MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline" \
  "cfg2_oneshot2|200|MI355FFT_LINES_TILES_PER_WG=2 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline" \
  "cfg2_oneshot4|200|MI355FFT_LINES_TILES_PER_WG=4 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline"
for f in res_skel_d4 res_skel_d2 res_skel_d1 res_d4 res_d2 res_d1 fused_ref cfg2_ref cfg2_oneshot1 cfg2_oneshot2 cfg2_oneshot4; do
  echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"
done
