#!/bin/bash
# fused sides (views / zero ranges / strided layouts in the line kernels' first load and last store) vs the staging route
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "views_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k 'ioview or zeropad or strided or whdcn or lane or nd'" \
  "views_tests_staging|500|MI355FFT_FUSE_VIEWS=0 python3 -m pytest tests/test_gpu_parity.py -x -q -k 'ioview or zeropad or strided or whdcn or lane'" \
  "view1d_fused|120|python3 bench.py --workload c2c_s4096_b65536_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view1d_staged|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload c2c_s4096_b65536_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view2d_fused|120|python3 bench.py --workload c2c_s1024x1024_b256_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view2d_staged|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload c2c_s1024x1024_b256_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view2d_plain|120|python3 bench.py --workload c2c_s1024x1024_b256 --steps 20 --warmup 3 --no-cpu-baseline" \
  "view3d_fused|120|python3 bench.py --workload c2c_s256x256x64_b64_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "view3d_staged|120|MI355FFT_FUSE_VIEWS=0 python3 bench.py --workload c2c_s256x256x64_b64_view --steps 20 --warmup 3 --no-cpu-baseline" \
  "js_gpu|400|cd webgpu-fft_amd/js && node test/gpu_parity.test.mjs" > gpurun_out/views_steps.log 2>&1
grep -E "^=== |passed|failed" gpurun_out/views_steps.log | tail -30
for f in view1d_fused view1d_staged view2d_fused view2d_staged view2d_plain view3d_fused view3d_staged; do
  echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"launches_per_step": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"
done
