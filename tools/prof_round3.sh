#!/bin/bash
# round-3 evidence: rocprofv3 kernel-trace summaries + separate PMC passes (FETCH_SIZE, WRITE_SIZE); bench lines of the same build.
# Output under gpurun_out/p3_*, copied into profiles/r03_* by hand.   usage: tools/prof_round3.sh [tags...]  (default: all)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
run() {  # tag workload steps launches-json
  local T=$1 W=$2 K=$3 L=$4
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3_kt_$T -- python3 $R/bench.py --workload $W --steps $K --warmup 2 --no-cpu-baseline > $R/gpurun_out/p3_kt_$T.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p3_fetch_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_fetch_$T.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p3_write_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_write_$T.log 2>&1 || return 1
  python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/p3_fetch_$T/*/*counter_collection.csv | head -1) $(ls $R/gpurun_out/p3_write_$T/*/*counter_collection.csv | head -1) $W "$L" > $R/gpurun_out/p3_pmc_$T.json
  cp $(ls $R/gpurun_out/p3_kt_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/p3_kernel_stats_$T.csv
  echo "== $T"; head -4 $R/gpurun_out/p3_kernel_stats_$T.csv | cut -c1-220; grep -o '"hbm_bytes_per_step": [0-9.]*' $R/gpurun_out/p3_pmc_$T.json
  tail -1 $R/gpurun_out/p3_kt_$T.log | cut -c1-240
}
want() { [ $# -eq 0 ] && return 0; local t=$1; shift; for a in $TAGS; do [ "$a" = "$t" ] && return 0; done; return 1; }
TAGS="$*"
sel() { [ -z "$TAGS" ] && return 0; for a in $TAGS; do [ "$a" = "$1" ] && return 0; done; return 1; }
sel cfg3 && run cfg3 c2c_2p20_b4096 5 '{"fft_xcd_rt1k_kernel": 1, "fft_xcd_fused_kernel": 1}'
sel cfg2 && run cfg2 c2c_1024_b65536 20 '{"fft_lines_kernel": 1}'
sel cfg5 && run cfg5 r2c_2p22_b1024 5 '{"fft_xcd_rt_r2c_kernel": 1}'
sel c2c22 && run c2c22 c2c_2p22_b512 5 '{"fft_xcd_rt_kernel": 1}'
sel c2r22 && run c2r22 c2r_2p22_b1024 5 '{"fft_xcd_rt_c2r_kernel": 1}'
sel c2c16 && run c2c16 c2c_2p16_b8192 5 '{"fft_xcd_fused_kernel": 1}'
sel c2c17 && run c2c17 c2c_2p17_b4096 5 '{"fft_xcd_fused_kernel": 1}'
sel c2c18 && run c2c18 c2c_2p18_b2048 5 '{"fft_xcd_fused_kernel": 1}'
true
