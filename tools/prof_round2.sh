#!/bin/bash
# round-2 evidence: rocprofv3 kernel-trace summaries + separate PMC passes (FETCH_SIZE, WRITE_SIZE) for the headline (config 3),
# config 2 and the config-5 shard; bench lines of the same builds.  Output under gpurun_out/, copied into profiles/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {  # tag workload steps launches-json
  local T=$1 W=$2 K=$3 L=$4
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p2_kt_$T -- python3 $R/bench.py --workload $W --steps $K --warmup 2 --no-cpu-baseline > $R/gpurun_out/p2_kt_$T.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p2_fetch_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p2_fetch_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p2_write_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p2_write_$T.log 2>&1
  python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/p2_fetch_$T/*/*counter_collection.csv | head -1) $(ls $R/gpurun_out/p2_write_$T/*/*counter_collection.csv | head -1) $W "$L" > $R/gpurun_out/p2_pmc_$T.json
  cp $(ls $R/gpurun_out/p2_kt_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/p2_kernel_stats_$T.csv
  echo "== $T"; head -4 $R/gpurun_out/p2_kernel_stats_$T.csv | cut -c1-220; grep -o '"hbm_bytes_per_step": [0-9.]*' $R/gpurun_out/p2_pmc_$T.json
  tail -1 $R/gpurun_out/p2_kt_$T.log | cut -c1-200
}
run cfg3 c2c_2p20_b4096 5 '{"fft_xcd_fused_kernel": 1}'
run cfg2 c2c_1024_b65536 20 '{"fft_lines_kernel": 1}'
run cfg5 r2c_2p22_b1024 5 '{"fft_xcd_fused_kernel": 1, "r2c_post": 1}'
