#!/bin/bash
# chunk-size sweep of the two-pass route (how much inter-pass intermediate stays in Infinity Cache)
for mb in 8 16 32 64 128 256 1024 65536; do
  echo "--- chunk ${mb} MiB"
  MI355FFT_CHUNK_BYTES=$((mb*1048576)) python bench.py --steps 5 --warmup 1 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['route'], 'GPt/s=%.1f'%d['value'], 'ms=%.2f'%d['ms_per_step'], 'frac=%.3f'%d['roofline']['frac'])"
done
echo "--- no graph, 64 MiB"
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-graph | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['config']['route'], 'GPt/s=%.1f'%d['value'], 'ms=%.2f'%d['ms_per_step'])"
