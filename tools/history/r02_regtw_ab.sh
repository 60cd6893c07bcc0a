#!/bin/bash
# one-shot ROW kernels: stage-1 roots straight into registers behind the tile loads (shipped) vs table staged in LDS first (lib_e1, -DMI355_LINES_REGTW=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2c_1024_b65536 c2c_2p9_b131072 c2c_2p8_b262144 c2c_2p7_b524288 c2c_2p6_b1048576"
for r in 1 2; do for w in $W; do
  for v in lib e1; do
    L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
    P="$P \"t${r}${v}_$w|100|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  done
done; done
eval tools/gpu_steps.sh "'lines_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"lines or golden or cfg or c2c_pow2 or headline\"'" $P > gpurun_out/regtw_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/regtw_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/regtw_steps.log | tail -2
for w in $W; do
  echo "== $w: $(for r in 1 2; do for v in lib e1; do echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/t${r}${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done; done)"
done
