#!/bin/bash
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE) for one bench workload: tools/prof_pmc.sh <workload> <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=$1; T=$2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${T}_fetch -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${T}_write -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${T}_write.log 2>&1
ls $R/gpurun_out/pmc_${T}_fetch/*/ $R/gpurun_out/pmc_${T}_write/*/
