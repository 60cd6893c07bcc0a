#!/bin/bash
# PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) and kernel trace of the register-resident line kernels
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {
  local T=$1 W=$2 L=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3_kt_$T -- python3 $R/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p3_kt_$T.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p3_fetch_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_fetch_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p3_write_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_write_$T.log 2>&1
  python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/p3_fetch_$T/*/*counter_collection.csv | head -1) $(ls $R/gpurun_out/p3_write_$T/*/*counter_collection.csv | head -1) $W "$L" > $R/gpurun_out/p3_pmc_$T.json
  cp $(ls $R/gpurun_out/p3_kt_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/p3_kernel_stats_$T.csv
  echo "== $T"; head -3 $R/gpurun_out/p3_kernel_stats_$T.csv | cut -c1-200; grep -o '"hbm_bytes_per_step": [0-9.]*' $R/gpurun_out/p3_pmc_$T.json
}
run c2c_2p15 c2c_2p15_b16384 '{"fft_line32k_kernel": 1}'
run c2c_2p13 c2c_2p13_b65536 '{"fft_line_reg_kernel": 1}'
