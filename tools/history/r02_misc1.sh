#!/bin/bash
# PMC traffic of the XCD-resident kernel (does the exchange stay in the L2?), and one-shot grids on the two-pass route
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python3 bench.py --workload c2c_2p20_b4096 --steps 10 --warmup 2 --no-cpu-baseline"
tools/gpu_steps.sh \
  "twopass_ref|200|MI355FFT_XCD_FUSED=0 $B" \
  "twopass_oneshot1|200|MI355FFT_XCD_FUSED=0 MI355FFT_LINES_TILES_PER_WG=1 $B" \
  "twopass_oneshot2|200|MI355FFT_XCD_FUSED=0 MI355FFT_LINES_TILES_PER_WG=2 $B" \
  "r2c22_ref|200|python3 bench.py --workload r2c_2p22_b1024 --steps 10 --warmup 2 --no-cpu-baseline" \
  "c2c_4096_oneshot|200|MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload c2c_2p12_b16384 --steps 20 --warmup 3 --no-cpu-baseline" \
  "c2c_4096_ref|200|python3 bench.py --workload c2c_2p12_b16384 --steps 20 --warmup 3 --no-cpu-baseline" \
  "c2c_256_oneshot|200|MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload c2c_2p8_b262144 --steps 20 --warmup 3 --no-cpu-baseline" \
  "c2c_256_ref|200|python3 bench.py --workload c2c_2p8_b262144 --steps 20 --warmup 3 --no-cpu-baseline" \
  "c2c_16384_oneshot|200|MI355FFT_LINES_TILES_PER_WG=1 python3 bench.py --workload c2c_2p14_b4096 --steps 20 --warmup 3 --no-cpu-baseline" \
  "c2c_16384_ref|200|python3 bench.py --workload c2c_2p14_b4096 --steps 20 --warmup 3 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MI355FFT_XCD_RES=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_res_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_res_fetch.log 2>&1
MI355FFT_XCD_RES=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_res_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_res_write.log 2>&1
cd $R
python3 tools/pmc_summary.py $(ls gpurun_out/pmc_res_fetch/*/*counter_collection.csv | head -1) $(ls gpurun_out/pmc_res_write/*/*counter_collection.csv | head -1) c2c_2p20_b4096 '{"fft_xcd_res_kernel": 1}' > gpurun_out/pmc_res_summary.json
cat gpurun_out/pmc_res_summary.json
for f in twopass_ref twopass_oneshot1 twopass_oneshot2 r2c22_ref c2c_4096_oneshot c2c_4096_ref c2c_256_oneshot c2c_256_ref c2c_16384_oneshot c2c_16384_ref; do
  echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"
done
