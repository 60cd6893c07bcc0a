#!/bin/bash
# rocprofv3 kernel-trace summary + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the headline bench
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_kt_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write_bench.log 2>&1
[ -n "$HEADLINE_ONLY" ] && { ls -R $R/gpurun_out/prof_kt | head; exit 0; }
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt2 -- python3 $R/bench.py --workload c2c_1024_b65536 --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_kt2_bench.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch2 -- python3 $R/bench.py --workload c2c_1024_b65536 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch2_bench.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write2 -- python3 $R/bench.py --workload c2c_1024_b65536 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write2_bench.log 2>&1
ls -R $R/gpurun_out | head -50
