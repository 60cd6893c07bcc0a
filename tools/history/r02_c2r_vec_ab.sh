#!/bin/bash
# c2r line kernels: pre-split pass with two adjacent bins per lane and 16-byte stores (shipped) vs one pair per lane (lib_e1, -DMI355_C2R_PRE_VEC=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
P=""
W="c2r_2p8_b4194304 c2r_2p9_b2097152 c2r_2p10_b1048576 c2r_2p11_b524288 c2r_2p12_b262144 c2r_2p13_b131072 c2r_2p14_b65536 c2r_2p15_b32768"
for w in $W; do for v in lib e1; do
  L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_$v/libmi355fft.so; [ $v = lib ] && L=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib/libmi355fft.so
  P="$P \"u${v}_$w|60|MI355FFT_LIB=$L python3 bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline\""; done; done
eval tools/gpu_steps.sh "'vec_tests|400|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"c2r or golden or cfg5\"'" $P > gpurun_out/vec2_steps.log 2>&1
grep -E "^=== .*exit" gpurun_out/vec2_steps.log | grep -v "exit 0" | tail
grep -E "passed|failed" gpurun_out/vec2_steps.log | tail -2
for w in $W; do echo "== $w: $(for v in lib e1; do echo -n "$v $(grep -o '"value": [0-9.]*' gpurun_out/u${v}_$w.log | head -1 | cut -d' ' -f2 | cut -c1-6) "; done)"; done
