#!/bin/bash
# r03 evidence batch: latency configs, 2-rank rehearsal on one GPU, PMC traffic of the fftconv pipeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/gpu_steps.sh \
  "latency|300|python3 tools/latency_bench.py" \
  "gpus2|400|MI355FFT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --workload c2c_2p20_b512 --steps 5 --warmup 1" \
  "gpus2_cfg5|400|MI355FFT_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --workload r2c_2p22_b128 --steps 5 --warmup 1" > gpurun_out/r03_evidence_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/r03_evidence_steps.log
tail -6 gpurun_out/latency.log | cut -c1-300
for f in gpus2 gpus2_cfg5; do echo "== $f: $(grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1) $(grep -o '"n_gpus": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"ranks_seen": [0-9]*' gpurun_out/$f.log | head -1) $(grep -o '"collective_backend": "[^"]*"' gpurun_out/$f.log | head -1) $(grep -o '"route": "[^"]*"' gpurun_out/$f.log | head -1)"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=conv; W=fftconv_2p20_b512
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p3_kt_$T -- python3 $R/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/p3_kt_$T.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p3_fetch_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_fetch_$T.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p3_write_$T -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/p3_write_$T.log 2>&1
python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/p3_fetch_$T/*/*counter_collection.csv | head -1) $(ls $R/gpurun_out/p3_write_$T/*/*counter_collection.csv | head -1) $W '{"fft_xcd_conv1m_kernel": 1, "fft_xcd_rt1k_kernel": 1}' > $R/gpurun_out/p3_pmc_$T.json
cp $(ls $R/gpurun_out/p3_kt_$T/*/*kernel_stats.csv | head -1) $R/gpurun_out/p3_kernel_stats_$T.csv
head -4 $R/gpurun_out/p3_kernel_stats_$T.csv | cut -c1-200; grep -o '"hbm_bytes_per_step": [0-9.]*' $R/gpurun_out/p3_pmc_$T.json
