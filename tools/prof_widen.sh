#!/bin/bash
# rocprofv3 kernel-trace summaries of the widened rows' probes (fftconv line route, 2-D dct2): per-kernel time split
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in fftconv_2p10_b65536 fftconv_2p13_b8192 dct2_s1024x1024_b256 dct2_2p20_b1024; do
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$w.log 2>&1
  f=$(find $R/gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1)
  echo "== $w"; tail -1 $R/gpurun_out/prof_$w.log | cut -c1-300; head -8 "$f"
done
