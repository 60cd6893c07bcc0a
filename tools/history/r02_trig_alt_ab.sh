#!/bin/bash
# one-launch DCT / DST: alternate ROW shapes (MI355FFT_TRIG_ALT=1, shipped) vs the shapes the plain transforms use (=0); same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="dct2_2p10_b262144 dct3_2p10_b262144 dct2_2p11_b131072 dct3_2p11_b131072 dst2_2p11_b131072 dct2_2p12_b65536 dct3_2p12_b65536 dct2_2p13_b32768 dct3_2p13_b32768"
for w in $W; do for v in 1 0; do
  P="$P \"ta${v}_$w|60|MI355FFT_TRIG_ALT=$v python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh "'trigalt_tests|500|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"dct or dst or trig\"'" $P > gpurun_out/trigalt_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/trigalt_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/trigalt_steps.log | tail -2
for w in $W; do echo "== $w: $(for v in 1 0; do echo -n "alt=$v $(grep -o '"value": [0-9.]*' gpurun_out/ta${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"; done
