#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AB=$GRAFT_REPO_ROOT/webgpu-fft_amd/lib_ab/libmi355fft.so
S=""
for w in c2c_1024_b65536 c2c_2p8_b262144 c2c_2p9_b131072 c2c_2p6_b1048576 c2c_2p7_b524288 c2c_2p5_b2097152 c2c_2p11_b32768; do
  S="$S \"new_$w|120|python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
  S="$S \"res_$w|120|MI355FFT_LINES_TILES_PER_WG=-1 python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline\""
done
eval tools/gpu_steps.sh $S \
  "'t4_1s|120|MI355FFT_LIB=$AB python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'t4_res|120|MI355FFT_LIB=$AB MI355FFT_LINES_TILES_PER_WG=-1 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'t4_2|120|MI355FFT_LIB=$AB MI355FFT_LINES_TILES_PER_WG=2 python3 bench.py --workload c2c_1024_b65536 --steps 50 --warmup 5 --no-cpu-baseline'" \
  "'headline|200|python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'split1|200|MI355FFT_XCD_SPLIT=1 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline'" \
  "'lines_tests|400|python3 -m pytest tests/test_gpu_parity.py -x -q -k \"lines or cfg2 or fftconv\"'" > gpurun_out/misc3_steps.log 2>&1
tail -12 gpurun_out/misc3_steps.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MI355FFT_XCD_SPLIT=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_split1_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_split1_fetch.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
for p in glob.glob("gpurun_out/pmc_split1_fetch/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(p)):
        acc.setdefault(r["Kernel_Name"][:50], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("split1 FETCH_SIZE x2 GB per launch:", k, 2 * sum(v) / len(v) * 1024 / 1e9)
PY
for f in gpurun_out/new_*.log gpurun_out/res_c2c*.log gpurun_out/t4_*.log gpurun_out/headline.log gpurun_out/split1.log; do
  echo "== $(basename $f .log): $(grep -o '"value": [0-9.]*' $f | head -1) $(grep -o '"route": "[^"]*"' $f | head -1)"
done
