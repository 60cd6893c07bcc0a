#!/bin/bash
# hand-off release A/B (lib_ab = MI355_XCD_RELEASE=1), trig phase f32, 1-D view probe, refreshed ioview test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab/libmi355fft.so
B="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline"
tools/gpu_steps.sh \
  "rel_shipped|200|$B" \
  "rel_release|200|MI355FFT_LIB=$AB $B" \
  "rel_shipped2|200|$B" \
  "rel_release2|200|MI355FFT_LIB=$AB $B" \
  "trig_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'dct or dst or trig or ioview'" \
  "dct2_2p20|120|python3 bench.py --workload dct2_2p20_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "dct4_2p20|120|python3 bench.py --workload dct4_2p20_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "dct2_4096|120|python3 bench.py --workload dct2_2p12_b65536 --steps 10 --warmup 2 --no-cpu-baseline" \
  "dct2_2d|120|python3 bench.py --workload dct2_s1024x1024_b256 --steps 10 --warmup 2 --no-cpu-baseline" \
  "view1d_fused|120|python3 bench.py --workload c2c_s4096_b65536_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view1d_staged|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload c2c_s4096_b65536_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view1d_plain|120|python3 bench.py --workload c2c_2p12_b65536 --steps 20 --warmup 3 --no-cpu-baseline" > gpurun_out/misc4_steps.log 2>&1
grep -E "^=== .*exit|passed|failed" gpurun_out/misc4_steps.log | tail -20
for f in rel_shipped rel_release rel_shipped2 rel_release2 dct2_2p20 dct4_2p20 dct2_4096 dct2_2d view1d_fused view1d_staged view1d_plain; do
  echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"launches_per_step": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"
done
