#!/bin/bash
# r03: N = 2^16 .. 2^18 with a small-footprint shared mode (all live workspace of an XCD <= 1-2 MiB, i.e. L2-sized) vs the shipped routes; same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/ab_env.sh "c2c_2p16_b8192" "MI355FFT_XCD_RT=1;MI355FFT_SOLO_MAX_KB=256 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1;MI355FFT_SOLO_MAX_KB=256 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=2;MI355FFT_SOLO_MAX_KB=256 MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2;MI355FFT_SOLO_MAX_KB=256 MI355FFT_XCD_SPLIT=4 MI355FFT_XCD_SLOTS=1" 2>&1 | tee gpurun_out/r03_small_l2.log
tools/ab_env.sh "c2c_2p17_b4096" "MI355FFT_XCD_RT=1;MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2;MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1" 2>&1 | tee -a gpurun_out/r03_small_l2.log
tools/ab_env.sh "c2c_2p18_b2048" "MI355FFT_XCD_RT=1;MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=1;MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=2" 2>&1 | tee -a gpurun_out/r03_small_l2.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "s2k1|MI355FFT_SOLO_MAX_KB=256 MI355FFT_XCD_SPLIT=2 MI355FFT_XCD_SLOTS=1|c2c_2p16_b8192" "s1k1|MI355FFT_XCD_SPLIT=1 MI355FFT_XCD_SLOTS=1|c2c_2p17_b4096"; do
  T=${cfg%%|*}; rest=${cfg#*|}; E=${rest%%|*}; W=${rest#*|}
  export $E
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p3_fetch_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_fetch_$T.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p3_write_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_write_$T.log 2>&1
  python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/p3_fetch_$T/*/*counter_collection.csv | head -1) $(ls $R/gpurun_out/p3_write_$T/*/*counter_collection.csv | head -1) $W '{"fft_xcd_fused_kernel": 1}' > $R/gpurun_out/p3_pmc_$T.json
  echo "== PMC $W [$E]: $(grep -o '"hbm_bytes_per_step": [0-9.]*' $R/gpurun_out/p3_pmc_$T.json)" | tee -a $R/gpurun_out/r03_small_l2.log
  for v in $E; do unset ${v%%=*}; done
done
